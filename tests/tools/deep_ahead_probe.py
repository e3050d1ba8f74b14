"""Fixture searches against the oracle for whatever form of the search the environment selects (BOTLAB_ASTAR_*), with timings:
   python tests/tools/deep_ahead_probe.py maze            -> all maze cases
   python tests/tools/deep_ahead_probe.py wide:2 convex:2 -> single cases"""
import os, sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import botlab_amd._capi as capi
if os.environ.get("PROBE_LIB"):          # A/B runs: another build of the library beside the tree's (libbotlab_hip_<name>.so)
    capi.LIB_PATH = capi.LIB_PATH.replace("libbotlab_hip.so", "libbotlab_hip_%s.so" % os.environ["PROBE_LIB"])
import botlab_amd as bl, helpers, oracle_lib
orc = oracle_lib.load_oracle()
maps = helpers.load_reference_maps()
ctx = bl.default_context()
bad = 0
for arg in sys.argv[1:] or ["maze"]:
    name, _, case = arg.partition(":")
    m = maps["astar_" + name]
    g = bl.OccupancyGrid.from_cells(m["cells"], m["origin"], m["mpc"], cellsPerMeter=helpers.CPM_DEFAULT, ctx=ctx)
    pl = bl.MotionPlanner(bl.MotionPlannerParams(0.1), ctx=ctx); pl.setMap(g)
    dist = orc.set_distances(m["cells"], m["mpc"], helpers.CPM_DEFAULT, m["origin"])
    rows = helpers.load_astar_cases()[name]
    for i, row in enumerate(rows):
        if case and int(case) != i: continue
        if (name, i) == ("narrow", 2): continue
        s = bl.make_pose(*row["start"], 0.0); gl = bl.make_pose(*row["goal"], 0.0)
        best = 1e9
        for rep in range(2):
            t0 = time.perf_counter()
            path, st = bl.search_for_path(s, gl, pl.distances_, pl.searchParams_, return_stats=True)
            best = min(best, time.perf_counter() - t0)
        exp, est = orc.search(orc.pose(*row["start"], 0.0), orc.pose(*row["goal"], 0.0), dist, m["mpc"], helpers.CPM_DEFAULT, m["origin"], 0.1, 1.0)
        got = np.array([(p.utime, p.x, p.y, p.theta) for p in path], dtype=exp.dtype)
        ok = tuple(st) == tuple(est) and got.tobytes() == exp.tobytes()
        bad += not ok
        print(name, i, "OK" if ok else "MISMATCH got %s exp %s" % (tuple(st), tuple(est)), "pops", st[0], "pushes", st[1],
              "%.2f ms  %.3f us/pop" % (best * 1e3, best * 1e6 / max(st[0], 1)), flush=True)
sys.exit(1 if bad else 0)
