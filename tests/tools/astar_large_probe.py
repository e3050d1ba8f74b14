"""One search on the bench's 2000 x 2000 (or GRID=) tiled world, goal GOAL_L1 cells from the start: pops, ms, us per pop, with
the 147 KB heap of a search that runs alone and (SMALL=1) with the replanner's 40 KB footprint.  STAMPS=1 loads the stamped build."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import botlab_amd._capi as capi
if os.environ.get("STAMPS"):
    capi.LIB_PATH = capi.LIB_PATH.replace("libbotlab_hip.so", "libbotlab_hip_stamps.so")
import botlab_amd as bl
from botlab_amd import synth
import bench

grid = int(os.environ.get("GRID", "2000"))
cells = synth.tile_world(bench.load_map("astar_maze")["cells"], grid)
half = grid * 0.05 / 2.0
origin = (np.float32(-half), np.float32(-half))
cells = np.where(cells > 0, 127, -100).astype(np.int8)
ctx = bl.default_context()
cpm = np.float32(1.0 / np.float64(np.float32(0.05)))
g = bl.OccupancyGrid.from_cells(cells, origin, np.float32(0.05), cellsPerMeter=cpm, ctx=ctx)
pl = bl.MotionPlanner(ctx=ctx)
pl.setMap(g)
for l1 in [int(v) for v in os.environ.get("GOAL_L1", "40,400").split(",")]:
    goal = bench.pick_goal(pl.distances_.cells(), origin, (0.3, 0.3), 0.2, l1)
    s, gl = bl.make_pose(0.3, 0.3, 0.0), bl.make_pose(goal[0], goal[1], 0.0)
    best = None
    for rep in range(4):
        pl.setMap(g)
        ctx.sync()
        t0 = time.perf_counter()
        path, st = bl.search_for_path(s, gl, pl.distances_, pl.searchParams_, return_stats=True)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    print("grid", grid, "goal_l1", l1, "pops", st[0], "pushes", st[1], "len", len(path), "%.2f ms  %.3f us/pop" % (best * 1e3, best * 1e6 / max(1, st[0])),
          ("stamps " + " ".join(str(v) for v in st[2:])) if len(st) > 2 else "")
