import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import botlab_amd._capi as capi
if os.environ.get("STAMPS"):
    capi.LIB_PATH = capi.LIB_PATH.replace("libbotlab_hip.so", "libbotlab_hip_stamps.so")
import botlab_amd as bl, helpers
from botlab_amd import synth
m = helpers.load_reference_maps()["obstacle_slam_10mx10m_5cm"]
truth = np.where(m["cells"] > 0, 127, -127).astype(np.int8)
ctx = bl.default_context()
g = bl.OccupancyGrid.from_cells(m["cells"], m["origin"], m["mpc"], cellsPerMeter=helpers.CPM_DEFAULT, ctx=ctx)
mp = bl.Mapping(5.0, 4, 1, ctx=ctx)
poses = synth.square_trajectory((-0.75, 0.2, 0.0), 6, step_len=0.02, turn=0.05, side=0.8)
for k in range(1, 7):
    sc = synth.raycast_scan(truth, m["origin"], 0.05, poses[k - 1], poses[k], 1000 + k * 100000)
    mp.updateMap(sc, bl.make_pose(*poses[k], utime=sc.utime), g)
ctx.sync()
