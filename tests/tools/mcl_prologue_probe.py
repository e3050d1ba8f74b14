"""How much of k_mcl_main is not ray work?  Times the kernel with a normal scan and with a scan whose rays are all dropped
(range <= 0.15): resample search + gather + action model + map staging + partial sums only."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import botlab_amd as bl, helpers
from botlab_amd import synth, _capi
maps = helpers.load_reference_maps()
m = maps["obstacle_slam_10mx10m_5cm"]
truth = np.where(m["cells"] > 0, 127, -127).astype(np.int8)
ctx = bl.default_context()
g = bl.OccupancyGrid.from_cells(m["cells"], m["origin"], m["mpc"], cellsPerMeter=helpers.CPM_DEFAULT, ctx=ctx)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
poses = synth.square_trajectory((-0.75, 0.2, 0.0), 60, step_len=0.02, turn=0.05, side=0.8)
for label, kill in (("290 rays", False), ("0 rays", True), ("290 rays", False)):
    pf = bl.ParticleFilter(N, ctx=ctx)
    pf.initializeFilterAtPose(bl.make_pose(-0.75, 0.2, 0.0, utime=1), seed=3)
    scans = [synth.raycast_scan(truth, m["origin"], 0.05, poses[k - 1], poses[k], 1000 + k * 100) for k in range(1, 41)]
    if kill:
        for s in scans:
            s.ranges[:] = 0.1
    for k in range(5):
        pf.updateFilter(bl.make_pose(*poses[k + 1], utime=scans[k].utime), scans[k], g, rand_value=k, want_pose=False)
    ctx.sync(); ctx.timing_reset(); ctx.timing_enable(True, kernels=[_capi.BL_K_MCL_MAIN])
    for k in range(5, 40):
        pf.updateFilter(bl.make_pose(*poses[k + 1], utime=scans[k].utime), scans[k], g, rand_value=k, want_pose=False)
    ctx.sync(); ctx.timing_enable(False)
    ms, n = ctx.timing_get(_capi.BL_K_MCL_MAIN)
    print(f"{label}: k_mcl_main {ms / n * 1e3:.1f} us over {n} launches (N={N})")
