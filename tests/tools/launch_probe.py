"""Times back-to-back enqueues of single stages (no per-step sync) to separate GPU-side per-operation cost from kernel time."""
import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import botlab_amd as bl, helpers
from botlab_amd import synth
m = helpers.load_reference_maps()["obstacle_slam_10mx10m_5cm"]
truth = np.where(m["cells"] > 0, 127, -127).astype(np.int8)
import os
if os.environ.get("PROBE_TORCH_STREAM"):
    import torch
    torch.cuda.set_device(0)
    _s = torch.cuda.Stream(torch.device("cuda", 0))
    ctx = bl.Context(0, stream=_s.cuda_stream)
else:
    ctx = bl.default_context()
if os.environ.get("PROBE_TIMING"):
    ctx.timing_enable(True, kernels=[0])
g = bl.OccupancyGrid.from_cells(m["cells"], m["origin"], m["mpc"], cellsPerMeter=helpers.CPM_DEFAULT, ctx=ctx)
poses = synth.square_trajectory((-0.75, 0.2, 0.0), 300, step_len=0.02, turn=0.05, side=0.8)
scans = [synth.raycast_scan(truth, m["origin"], 0.05, poses[k - 1], poses[k], 1000 + k * 100000) for k in range(1, 301)]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
pf = bl.ParticleFilter(N, ctx=ctx)
pf.initializeFilterAtPose(bl.make_pose(-0.75, 0.2, 0.0, utime=1000), seed=1)
mp = bl.Mapping(5.0, 4, 1, ctx=ctx)
dist = bl.ObstacleDistanceGrid(ctx=ctx)
def timeit(name, fn, n=200):
    for k in range(10): fn(k)
    ctx.sync()
    t0 = time.perf_counter()
    for k in range(10, 10 + n): fn(k)
    t1 = time.perf_counter()
    ctx.sync()
    t2 = time.perf_counter()
    print("%-28s host enqueue %.1f us/call, total %.1f us/call" % (name, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
timeit("pf.updateFilter(no pose)", lambda k: pf.updateFilter(bl.make_pose(*poses[k + 1], utime=scans[k].utime), scans[k], g, rand_value=k, want_pose=False))
timeit("mapper.updateMapDevicePose", lambda k: mp.updateMapDevicePose(scans[k], pf.poseDevicePtr(), scans[k].utime, g))
timeit("dist.setDistances", lambda k: dist.setDistances(g))
timeit("pf+map", lambda k: (pf.updateFilter(bl.make_pose(*poses[k + 1], utime=scans[k].utime), scans[k], g, rand_value=k, want_pose=False), mp.updateMapDevicePose(scans[k], pf.poseDevicePtr(), scans[k].utime, g)))
