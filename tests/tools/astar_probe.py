import os, sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import botlab_amd._capi as capi
if os.environ.get("STAMPS"):
    capi.LIB_PATH = capi.LIB_PATH.replace("libbotlab_hip.so", "libbotlab_hip_stamps.so")
import botlab_amd as bl, helpers
maps = helpers.load_reference_maps()
ctx = bl.default_context()
for name, case in (("astar_maze", 0), ("astar_maze", 2), ("astar_maze", 1), ("astar_wide", 2), ("astar_convex", 2)):
    m = maps[name]
    g = bl.OccupancyGrid.from_cells(m["cells"], m["origin"], m["mpc"], cellsPerMeter=helpers.CPM_DEFAULT, ctx=ctx)
    pl = bl.MotionPlanner(bl.MotionPlannerParams(0.1), ctx=ctx); pl.setMap(g)
    row = helpers.load_astar_cases()[name.split("_")[1]][case]
    s = bl.make_pose(*row["start"], 0.0); gl = bl.make_pose(*row["goal"], 0.0)
    for rep in range(3):
        t0 = time.perf_counter()
        path, st = bl.search_for_path(s, gl, pl.distances_, pl.searchParams_, return_stats=True)
        dt = time.perf_counter() - t0
    print(name, case, "pops", st[0], "pushes", st[1], "len", len(path), "%.2f ms  %.3f us/pop" % (dt * 1e3, dt * 1e6 / st[0]))
