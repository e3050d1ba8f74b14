"""Generates tests/golden/astar_random_cases.json: start / goal pairs on seeded random worlds (the generator of
tests/test_gpu_parity.py::test_astar_random_maps_equal_oracle) whose search the CPU oracle finishes within a few seconds.  The
reference's cost function without open-list de-duplication makes many nearby, reachable goals cost 1e7 .. 1e9 pops; a pair is kept
only if the oracle, run in a child process with a time limit, needs at most MAX_POPS.  Data only: seeds, poses, pop counts."""
import json, multiprocessing as mp, os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
MAX_POPS = 2_000_000


def world(seed):
    rng = np.random.default_rng(seed)
    W, H = 320, 240
    cells = np.full((H, W), -80, np.int8)
    for _ in range(60):
        x, y = int(rng.integers(0, W - 12)), int(rng.integers(0, H - 12))
        cells[y:y + int(rng.integers(2, 12)), x:x + int(rng.integers(2, 12))] = 90
    for _ in range(6):
        if rng.random() < 0.5:
            y = int(rng.integers(20, H - 20)); x0 = int(rng.integers(0, W // 2)); cells[y:y + 2, x0:x0 + int(rng.integers(40, W // 2))] = 90
        else:
            x = int(rng.integers(20, W - 20)); y0 = int(rng.integers(0, H // 2)); cells[y0:y0 + int(rng.integers(40, H // 2)), x:x + 2] = 90
    cells[0, :] = cells[-1, :] = 90; cells[:, 0] = cells[:, -1] = 90
    return cells, rng


def one(args, q):
    import helpers, oracle_lib
    seed, radius, sp, gp = args
    orc = oracle_lib.load_oracle()
    cells, _ = world(seed)
    mpc = np.float32(0.05); origin = (np.float32(-8.0), np.float32(-6.0))
    dist = orc.set_distances(cells, mpc, helpers.CPM_DEFAULT, origin)
    exp, est = orc.search(orc.pose(*sp, 0.3), orc.pose(*gp, 0.0), dist, mpc, helpers.CPM_DEFAULT, origin, radius, 10.0 * radius, cap=1 << 16)
    q.put((int(est[0]), int(est[1]), len(exp)))


if __name__ == "__main__":
    import helpers, oracle_lib
    from scipy import ndimage
    orc = oracle_lib.load_oracle()
    out = []
    for seed in (1, 2, 3, 4):
        cells, rng = world(seed)
        mpc = np.float32(0.05); origin = (np.float32(-8.0), np.float32(-6.0))
        dist = orc.set_distances(cells, mpc, helpers.CPM_DEFAULT, origin)
        for radius in (0.1, 0.2):
            comp, _ = ndimage.label(dist > np.float32(radius) * np.float32(1.000001))
            ys, xs = np.nonzero(dist > radius * 1.5)
            kept = 0
            for _ in range(14):
                if kept >= 4:
                    break
                a = int(rng.integers(0, xs.size))
                near = np.nonzero((comp[ys, xs] == comp[ys[a], xs[a]]) & (np.abs(xs - xs[a]) + np.abs(ys - ys[a]) <= 90))[0]
                b = int(near[rng.integers(0, near.size)])
                sp = (float(origin[0]) + (xs[a] + 0.5) * 0.05, float(origin[1]) + (ys[a] + 0.5) * 0.05)
                gp = (float(origin[0]) + (xs[b] + 0.5) * 0.05, float(origin[1]) + (ys[b] + 0.5) * 0.05)
                q = mp.Queue()
                p = mp.Process(target=one, args=((seed, radius, sp, gp), q))
                p.start(); p.join(4.0)
                if p.is_alive():
                    p.terminate(); p.join()
                    continue
                pops, pushes, n = q.get()
                if pops > MAX_POPS:
                    continue
                out.append({"seed": seed, "radius": radius, "start": sp, "goal": gp, "pops": pops, "pushes": pushes, "poses": n})
                kept += 1
                print(out[-1], flush=True)
    json.dump(out, open(os.path.join(os.path.dirname(HERE), "golden", "astar_random_cases.json"), "w"), indent=0)
    print(len(out), "cases")
