"""One long fixture search (wide case 2 by default) inside grids of growing size: the fixture's 200 x 200 map pasted into the centre of
an otherwise occupied S x S grid, the same start / goal shifted with it.  us per pop by grid size; with STAMPS=1 the stamped build's
line per search.  python tests/tools/big_grid_search_probe.py [name:case] [sizes...]"""
import os, sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import botlab_amd._capi as capi
if os.environ.get("STAMPS"):
    capi.LIB_PATH = capi.LIB_PATH.replace("libbotlab_hip.so", "libbotlab_hip_stamps.so")
import botlab_amd as bl, helpers
maps = helpers.load_reference_maps()
ctx = bl.default_context()
arg = sys.argv[1] if len(sys.argv) > 1 and ":" in sys.argv[1] else "wide:2"
sizes = [int(a) for a in sys.argv[1:] if ":" not in a] or [0, 1000, 2000, 4096]      # 0: the map itself
name, case = arg.split(":")
m = maps["astar_" + name]
row = helpers.load_astar_cases()[name][int(case)]
h, w = m["cells"].shape
for S in sizes:
    S = S or max(h, w)
    cells = np.full((S, S), 127, np.int8)
    oy, ox = (S - h) // 2, (S - w) // 2
    cells[oy:oy + h, ox:ox + w] = m["cells"]
    mpc = float(m["mpc"])
    origin = (np.float32(float(m["origin"][0]) - ox * mpc), np.float32(float(m["origin"][1]) - oy * mpc))
    g = bl.OccupancyGrid.from_cells(cells, origin, m["mpc"], cellsPerMeter=helpers.CPM_DEFAULT, ctx=ctx)
    pl = bl.MotionPlanner(bl.MotionPlannerParams(0.1), ctx=ctx); pl.setMap(g)
    s = bl.make_pose(*row["start"], 0.0); gl = bl.make_pose(*row["goal"], 0.0)
    best = 1e9
    for rep in range(2):
        t0 = time.perf_counter()
        path, st = bl.search_for_path(s, gl, pl.distances_, pl.searchParams_, return_stats=True)
        best = min(best, time.perf_counter() - t0)
    print("grid %d: pops %d pushes %d (%.2f per pop) poses %d  %.2f ms  %.3f us/pop" % (S, st[0], st[1], st[1] / max(st[0], 1), len(path), best * 1e3, best * 1e6 / max(st[0], 1)), flush=True)
    pl.close() if hasattr(pl, "close") else None
    g.close()
