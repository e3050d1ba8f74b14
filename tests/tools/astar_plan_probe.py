"""The longest search of plan_path_to_frontier on the cut arena (5.4e5 pops), alone: us per pop and (STAMPS=1) the cycle shares."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import botlab_amd._capi as capi
if os.environ.get("PROBE_LIB"):          # A/B runs: another build of the library beside the tree's (libbotlab_hip_<name>.so)
    capi.LIB_PATH = capi.LIB_PATH.replace("libbotlab_hip.so", "libbotlab_hip_%s.so" % os.environ["PROBE_LIB"])
if os.environ.get("STAMPS"):
    capi.LIB_PATH = capi.LIB_PATH.replace("libbotlab_hip.so", "libbotlab_hip_stamps.so")
import botlab_amd as bl, helpers
maps = helpers.load_reference_maps()
m = maps["obstacle_slam_10mx10m_5cm"]; c = m["cells"].copy(); c[:, 110:] = 0
ctx = bl.default_context()
g = bl.OccupancyGrid.from_cells(c, m["origin"], m["mpc"], cellsPerMeter=helpers.CPM_DEFAULT, ctx=ctx)
pl = bl.MotionPlanner(bl.MotionPlannerParams(0.2), ctx=ctx); pl.setMap(g)
rp = bl.make_pose(-0.75, 0.2, 0.4)
fr = bl.find_map_frontiers(g, rp)
pl.setNumFrontiers(len(fr))
path, goal, st = bl.plan_path_to_frontier(fr, rp, g, pl, return_info=True)
print("plan:", st, "goal", goal.x, goal.y)
# the candidates of the deciding ring around the chosen goal: search each alone
for dx in (-0.05, 0.0, 0.05):
    for dy in (-0.05, 0.0, 0.05):
        gl = bl.make_pose(goal.x + dx, goal.y + dy, 0.0)
        if not pl.isValidGoal(gl):
            continue
        t0 = time.perf_counter()
        p, s2 = bl.search_for_path(rp, gl, pl.distances_, pl.searchParams_, return_stats=True)
        dt = time.perf_counter() - t0
        print("goal %+.2f %+.2f: pops %d pushes %d len %d  %.1f ms  %.3f us/pop" % (dx, dy, s2[0], s2[1], len(p), dt * 1e3, dt * 1e6 / max(1, s2[0])), flush=True)
