"""What would a pop that starts its walk BEFORE the previous expansion's pushes are in see?  (the costed next step
for k_astar2 -- a second wavefront runs pop k + 1's hole descent, which does not depend on the value being sifted, beside the first
wavefront's pushes of expansion k.)  CPU model: the reference's search (astar.cpp:75-135) over an explicit array heap with libstdc++'s
__adjust_heap / __push_heap index operations, pops and pushes checked against the oracle; per iteration it compares the walk taken on
the heap as pop k left it (slots behind the heap read as +inf, the length as it will be) with the walk on the heap after the pushes.
Prints, per search: pops, the share of iterations whose early walk reads a position a push wrote (the conservative test a kernel would
make), the share whose early walk really differs, and where pushes land (levels above the slot)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
import oracle_lib

INF = 1 << 30


def walk(keys, n):
    """hole descent of __adjust_heap over keys[0:n] (n = length without the removed last element): positions visited below the root,
    and every position whose key a decision read"""
    path, read = [], []
    hole, second = 0, 0
    while second < (n - 1) // 2:
        second = 2 * (second + 1)
        read.append(second); read.append(second - 1)
        if keys[second] > keys[second - 1]:
            second -= 1
        path.append(second)
        hole = second
    if (n & 1) == 0 and second == (n - 2) // 2:
        second = 2 * (second + 1)
        path.append(second - 1)
    return path, read


def run(name, case, radius=0.1, cells=None, origin=None, mpc=None, start=None, goal=None, maxd=None):
    orc = oracle_lib.load_oracle()
    cpm = helpers.CPM_DEFAULT
    if cells is None:
        m = helpers.load_reference_maps()["astar_" + name]
        cells, origin, mpc = m["cells"], m["origin"], m["mpc"]
        row = helpers.load_astar_cases()[name][case]
        start, goal = row["start"], row["goal"]
    maxd = 10.0 * radius if maxd is None else maxd
    dist = orc.set_distances(cells, mpc, cpm, origin)
    exp, est = orc.search(orc.pose(*start, 0.0), orc.pose(*goal, 0.0), dist, mpc, cpm, origin, radius, maxd)
    H, W = dist.shape
    f = np.float32
    def cell_of(v, o):                               # global_position_to_grid_cell on the pose's FLOAT coordinate (grid_utils.hpp:33-38)
        return int((float(f(v)) - float(f(o))) * float(f(cpm)))
    ex, ey = cell_of(goal[0], origin[0]), cell_of(goal[1], origin[1])
    sx, sy = cell_of(start[0], origin[0]), cell_of(start[1], origin[1])
    valid = dist > np.float64(radius) * 1.000001
    ocost = np.zeros((H, W), np.int64)
    band = (dist > radius) & (dist < maxd)
    ocost[band] = (np.float64(maxd) - (dist[band] * f(2000)).astype(np.float64)).astype(np.int64)      # exponent 1: the float product, then double
    closed = np.zeros((H, W), bool)
    keys, cell, g = [0], [(sx, sy)], [0]
    pops = pushes = 0
    conservative = differs = root_land = 0
    rises = np.zeros(40, np.int64)
    early = None
    while keys:
        n = len(keys)
        # ---- pop k: the true walk on the heap as it stands (compare with the walk taken early, before the last pushes)
        if n > 1:
            true_path, _ = walk(keys, n - 1)
            if early is not None:
                e_path, e_read, written = early
                if any(p in written for p in e_read):
                    conservative += 1
                if e_path != true_path:
                    differs += 1
        top_k, top_c, top_g = keys[0], cell[0], g[0]
        vk, vc, vg = keys[-1], cell[-1], g[-1]
        keys.pop(); cell.pop(); g.pop()
        n -= 1
        if n > 0:
            path, _ = walk(keys, n)
            hole = 0
            for p in path:
                keys[hole], cell[hole], g[hole] = keys[p], cell[p], g[p]
                hole = p
            while hole > 0 and keys[(hole - 1) // 2] > vk:
                par = (hole - 1) // 2
                keys[hole], cell[hole], g[hole] = keys[par], cell[par], g[par]
                hole = par
            keys[hole], cell[hole], g[hole] = vk, vc, vg
        pops += 1
        closed[top_c[1], top_c[0]] = True
        # ---- the early walk of pop k + 1: heap after pop k, slots behind it +inf, length as it will be once the pushes are in
        new = []
        done = False
        for dx, dy in ((1, 0), (-1, 0), (0, 1), (0, -1)):
            kx, ky = top_c[0] + dx, top_c[1] + dy
            if not (0 <= kx < W and 0 <= ky < H) or not valid[ky, kx]:
                continue
            if kx == ex and ky == ey:
                done = True
                pushes += len(new)                           # (the neighbours in front of the goal neighbour were pushed: astar.cpp:117-129)
                break
            if closed[ky, kx]:
                continue
            ax, ay = abs(ex - kx), abs(ey - ky)
            h = 14 * ay + 10 * (ax - ay) if ax >= ay else 14 * ax + 10 * (ay - ax)
            fn = top_g + 10 + h + int(ocost[ky, kx])
            if fn < 32767:
                new.append((fn, (kx, ky), top_g + 10))
        if done:
            break
        n_after = len(keys) + len(new)
        if n_after > 1:
            padded = keys + [INF] * len(new)
            e_path, e_read = walk(padded, n_after - 1)
        written = set()
        for fn, c, gg in new:
            keys.append(fn); cell.append(c); g.append(gg)
            hole = len(keys) - 1
            written.add(hole)
            up = 0
            while hole > 0 and keys[(hole - 1) // 2] > fn:
                par = (hole - 1) // 2
                keys[hole], cell[hole], g[hole] = keys[par], cell[par], g[par]
                hole = par
                written.add(hole)
                up += 1
            keys[hole], cell[hole], g[hole] = fn, c, gg
            rises[min(up, 39)] += 1
            root_land += hole == 0
            pushes += 1
        early = (e_path, e_read, written) if n_after > 1 else None
    ok = (pops, pushes) == tuple(est)
    tot = max(pops, 1)
    r = rises / max(rises.sum(), 1)
    stats = dict(pops=pops, pushes=pushes, oracle=tuple(est), early_walk_reads_a_written_position=conservative / tot, early_walk_differs=differs / tot,
                 pushes_landing_at_the_root=root_land / max(pushes, 1), pushes_that_do_not_rise=float(r[0]))
    print(f"{name} {case}: pops {pops} pushes {pushes} (oracle {tuple(est)}: {'equal' if ok else 'DIFFERENT'}) | early walk reads a written "
          f"position {100.0 * conservative / tot:.2f} %, really differs {100.0 * differs / tot:.2f} % | pushes landing at the root "
          f"{100.0 * root_land / max(pushes, 1):.2f} % | rise 0: {100 * r[0]:.0f} %, 1-2: {100 * r[1:3].sum():.0f} %, 3-5: {100 * r[3:6].sum():.0f} %, "
          f"6-9: {100 * r[6:10].sum():.0f} %, 10+: {100 * r[10:].sum():.0f} %", flush=True)
    return stats


if __name__ == "__main__":
    for name, case in (("maze", 0), ("maze", 2), ("maze", 1), ("narrow", 0), ("narrow", 1), ("wide", 1), ("wide", 2), ("convex", 0)):
        run(name, case)
