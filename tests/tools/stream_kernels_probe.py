"""The streaming (HBM-bound) kernels of the path at their large sizes, for a rocprofv3 --kernel-trace run: particles() /
LCM encode and the record-based finish at 1M particles, the replanner snapshot and setDistances at 4096^2."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["BOTLAB_MCL_NO_FUSED_FINISH"] = "1"
import numpy as np
import botlab_amd as bl, helpers
from botlab_amd import synth
maps = helpers.load_reference_maps()
m = maps["obstacle_slam_10mx10m_5cm"]
truth = np.where(m["cells"] > 0, 127, -127).astype(np.int8)
ctx = bl.default_context()
g = bl.OccupancyGrid.from_cells(m["cells"], m["origin"], m["mpc"], cellsPerMeter=helpers.CPM_DEFAULT, ctx=ctx)
N = 1_000_000
pf = bl.ParticleFilter(N, ctx=ctx)
pf.initializeFilterAtPose(bl.make_pose(-0.75, 0.2, 0.0, utime=1), seed=3)
poses = synth.square_trajectory((-0.75, 0.2, 0.0), 6, step_len=0.02, turn=0.05, side=0.8)
for k in range(1, 6):
    sc = synth.raycast_scan(truth, m["origin"], 0.05, poses[k - 1], poses[k], 1000 + k * 100)
    pf.updateFilter(bl.make_pose(*poses[k], utime=sc.utime), sc, g, rand_value=k, want_pose=False)
buf = (C.c_uint8 * (20 + 48 * N))()
for _ in range(5):
    pf.particles()
    assert ctx.lib.bl_pf_encode_particles_lcm(pf.h, 7, buf, len(buf)) == len(buf)
S = 4096
world = synth.tile_world(maps["astar_maze"]["cells"], S)
big = bl.OccupancyGrid.from_cells(world, (-100.0, -100.0), 0.05, ctx=ctx)
d = bl.ObstacleDistanceGrid(ctx=ctx)
ap = bl.AsyncPlanner(ctx=ctx, lanes=1)
for _ in range(5):
    d.forget()                              # the full transform every time (an unchanged map would be recognised)
    d.setDistances(big)
    ap.submit(big, pf.poseDevicePtr(), bl.make_pose(0.3, 0.3, 0.0))
    ap.fetch()
d.cells()                                   # k_dist_floats, once, after the timed transforms
ctx.sync()
print("done")
