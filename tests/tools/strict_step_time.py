"""A 100 000-particle SLAM step (filter update with its end riding in the map kernel) with the default resampling rule and in strict mode
(bl_pf_set_strict_resampling): host-timed over 230 steps.  Under rocprofv3 --kernel-trace --stats its kernel table gives the three strict
kernels' durations (profiles/r05_strict_kernels.csv)."""
import os, sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import botlab_amd as bl, helpers
from botlab_amd import synth
maps = helpers.load_reference_maps()
ctx = bl.default_context()
m = maps["obstacle_slam_10mx10m_5cm"]
truth = np.where(m["cells"] > 0, 127, -127).astype(np.int8)
poses = synth.square_trajectory((-0.75, 0.2, 0.0), 260, step_len=0.02, turn=0.05, side=0.8)
scans = [synth.raycast_scan(truth, m["origin"], 0.05, poses[k - 1], poses[k], 1_000_000 + k * 100_000) for k in range(1, 261)]
for strict in (False, True):
    g = bl.OccupancyGrid.from_cells(m["cells"], m["origin"], m["mpc"], cellsPerMeter=helpers.CPM_DEFAULT, ctx=ctx)
    pf = bl.ParticleFilter(100_000, ctx=ctx)
    pf.initializeFilterAtPose(bl.make_pose(-0.75, 0.2, 0.0, utime=int(scans[0].times[0])), seed=11)
    pf.setStrictResampling(strict)
    mapper = bl.Mapping(5.0, 4, 1, ctx=ctx)
    def step(k):
        sc = scans[k]
        pf.updateBegin(bl.make_pose(*poses[k + 1], utime=sc.utime), sc, g, 77 + k)
        if k + 1 < len(scans): ctx.scanPrefetch(scans[k + 1])
        mapper.updateMapFinishingFilter(sc, pf, sc.utime, g)
    for k in range(30): step(k)
    ctx.sync()
    t0 = time.perf_counter()
    for k in range(30, 260): step(k)
    ctx.sync()
    print("strict" if strict else "default", "%.1f us per SLAM step (filter end riding in the map kernel)" % ((time.perf_counter() - t0) / 230 * 1e6), flush=True)
