"""data/astar/narrow_poses.txt:4 (0 -5 -> 0 5, shouldExist 0): the constriction is narrower than the robot, so the reference's search
must exhaust the whole free side before it answers "no path" (2.6e8 pops in its own algorithm: excluded from the fixture tests on
both sides).  What the HIP path does with it: runs it once, prints status, pops, pushes and the time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C
import botlab_amd as bl, helpers
from botlab_amd import _capi
maps = helpers.load_reference_maps()
ctx = bl.default_context()
m = maps["astar_narrow"]
g = bl.OccupancyGrid.from_cells(m["cells"], m["origin"], m["mpc"], cellsPerMeter=helpers.CPM_DEFAULT, ctx=ctx)
pl = bl.MotionPlanner(bl.MotionPlannerParams(0.1), ctx=ctx); pl.setMap(g)
row = helpers.load_astar_cases()["narrow"][2]
s = bl.make_pose(*row["start"], 0.0); gl = bl.make_pose(*row["goal"], 0.0)
if len(sys.argv) > 1:
    ctx.lib.bl_astar_set_open_capacity(ctx.h, int(sys.argv[1]))
buf = (_capi.Pose * 4096)(); n = C.c_int(0); stats = (C.c_int64 * 2)()
t0 = time.perf_counter()
rc = ctx.lib.bl_astar_search(ctx.h, pl.distances_.h, C.byref(s), C.byref(gl), C.byref(pl.searchParams_), buf, 4096, C.byref(n), stats)
dt = time.perf_counter() - t0
msg = ctx.lib.bl_last_error()
print("narrow case 2: rc", rc, "path poses", n.value, "pops", stats[0], "pushes", stats[1], "%.1f s" % dt, "%.3f us/pop" % (dt * 1e6 / max(1, stats[0])),
      "|", msg.decode() if rc and msg else "")
