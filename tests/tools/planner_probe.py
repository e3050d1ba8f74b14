"""How long is ONE replan on a large grid, outside the pipeline?  setDistances and search_for_path on the bench's tiled-maze
world (python tests/tools/planner_probe.py [grid_side]), host-timed call by call, plus the same searches as one batch."""
import os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import bench
import botlab_amd as bl
from botlab_amd import host

side = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
args = types.SimpleNamespace(grid=side, goal_l1=40)
m, truth, poses, odo, scans, rands = bench.build_inputs(args, 40)
ctx = bl.default_context()
grid = bl.OccupancyGrid.from_cells(m["cells"], m["origin"], m["mpc"], cellsPerMeter=20.0, ctx=ctx)
planner = bl.MotionPlanner(ctx=ctx)
t0 = time.perf_counter(); planner.setMap(grid); ctx.sync(); t1 = time.perf_counter()
for _ in range(5):
    planner.distances_.forget(); planner.setMap(grid)
ctx.sync(); t2 = time.perf_counter()
print(f"setDistances {side}^2: first {1e3 * (t1 - t0):.2f} ms, then {1e3 * (t2 - t1) / 5:.3f} ms")
goal = bench.pick_goal(planner.distances_.cells(), m["origin"], poses[0][:2], 0.2, args.goal_l1)
gp = bl.make_pose(goal[0], goal[1], 0.0)
tt, pp = [], []
for k in range(0, 40, 2):
    sp = bl.make_pose(*poses[k])
    t0 = time.perf_counter()
    path, st = host.search_for_path(sp, gp, planner.distances_, planner.searchParams_, return_stats=True)
    tt.append(time.perf_counter() - t0); pp.append(st[0])
tt, pp = np.array(tt[2:]), np.array(pp[2:])
print(f"search_for_path: {1e3 * tt.mean():.3f} ms mean, {pp.mean():.0f} pops mean -> {1e6 * tt.sum() / max(pp.sum(), 1):.2f} us/pop "
      f"(min {1e3 * tt.min():.3f} ms / {pp[tt.argmin()]} pops, max {1e3 * tt.max():.3f} ms / {pp[tt.argmax()]} pops)")
