"""Timing probe for row f3: find_map_frontiers + plan_path_to_frontier, GPU vs the CPU oracle, on the cut 200x200
reference map and on partially explored tiled worlds (2000^2, 4096^2)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import botlab_amd as bl, helpers, oracle_lib
from botlab_amd import synth, _capi

o = oracle_lib.load_oracle(); maps = helpers.load_reference_maps()
ctx = bl.default_context()


def explored_world(S, radius_cells):
    # open hall with 4x4-cell pillars every 40 cells (a maze of narrow corridors makes most candidate goals unreachable
    # for a robot-sized clearance, and the reference's sweep then runs exhaustive searches)
    cells = np.full((S, S), -100, np.int8)
    for oy in range(20, S, 40):
        for ox in range(20, S, 40):
            cells[oy:oy + 4, ox:ox + 4] = 100
    cells[0, :] = cells[-1, :] = 100; cells[:, 0] = cells[:, -1] = 100
    yy, xx = np.mgrid[0:S, 0:S]
    cx = cy = S // 2 + 6
    far = (xx - cx) ** 2 + (yy - cy) ** 2 > radius_cells ** 2
    cells[far] = 0
    half = S * 0.05 / 2
    # the robot stands 2 m inside the explored disc's rim (a goal 20+ m away costs the reference's A* millions of pops)
    return cells, (np.float32(-half), np.float32(-half)), np.float32(0.05), (-half + (cx + radius_cells - 40 + 0.5) * 0.05, -half + (cy + 0.5) * 0.05, 0.0)


cases = []
m = maps["obstacle_slam_10mx10m_5cm"]; c = m["cells"].copy(); c[:, 110:] = 0
cases.append(("cut200", c, m["origin"], m["mpc"], (-0.75, 0.2, 0.4), 0.2))
for S, r in ((2000, 400), (4096, 1500)):
    cells, origin, mpc, robot = explored_world(S, r)
    cases.append((f"tiled{S}", cells, origin, mpc, robot, 0.1))

for name, cells, origin, mpc, robot, radius in cases:
    grid = bl.OccupancyGrid.from_cells(cells, origin, mpc, cellsPerMeter=helpers.CPM_DEFAULT, ctx=ctx)
    pl = bl.MotionPlanner(bl.MotionPlannerParams(radius), ctx=ctx); pl.setMap(grid)
    rp = bl.make_pose(*robot)
    for rep in range(3):
        ctx.sync(); t0 = time.perf_counter(); fr = bl.find_map_frontiers(grid, rp); t_find = time.perf_counter() - t0
    lists = fr.cells(); bfs = fr.stats()
    pl.setNumFrontiers(len(lists))
    t0 = time.perf_counter(); exp = o.find_frontiers(cells, mpc, helpers.CPM_DEFAULT, origin, o.pose(*robot)); t_ofind = time.perf_counter() - t0
    same = len(exp) == len(lists) and all(a.tobytes() == b.tobytes() for a, b in zip(exp, lists))
    msg = (f"{name}: frontiers {len(lists)} cells {sum(len(f) for f in lists)} bfs cells/levels {bfs} | find GPU {t_find*1e3:.2f} ms, "
           f"oracle {t_ofind*1e3:.2f} ms, same={same}")
    if cells.size <= 200 * 200:      # on the big open halls the reference's A* cost terms make every search flood the hall (10^7 pops)
        for rep in range(2):
            t0 = time.perf_counter(); path, goal, st = bl.plan_path_to_frontier(fr, rp, grid, pl, return_info=True); t_plan = time.perf_counter() - t0
        dist = o.set_distances(cells, mpc, helpers.CPM_DEFAULT, origin)
        sp = pl.searchParams_
        t0 = time.perf_counter()
        epath, egoal, est = o.plan_path_to_frontier(exp, o.pose(*robot), dist, mpc, helpers.CPM_DEFAULT, origin, radius, sp.minDistanceToObstacle,
                                                    sp.maxDistanceWithCost, 1.0, num_frontiers=len(exp))
        t_oplan = time.perf_counter() - t0
        psame = len(epath) == len(path) and all((a.x, a.y, a.theta) == (b["x"], b["y"], b["theta"]) for a, b in zip(path, epath))
        msg += (f" | plan GPU {t_plan*1e3:.2f} ms ({st[2]} searches, {st[0]} pops), oracle {t_oplan*1e3:.2f} ms ({est[0]} pops), "
                f"path {len(path)} same={psame}")
    print(msg, flush=True)
