"""Where do k_mcl_main's microseconds go?  With the stamped build (-DMCL_STAMPS; BOTLAB_HIP_LIB points at it) every workgroup
leaves the 100 MHz clock at entry, behind its first barrier, behind the ray loop and at its end; this prints the distribution
over the workgroups of the last launch of a short pipelined run (the bench's own step, replanner beside it)."""
import ctypes as C, os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import botlab_amd as bl
from botlab_amd import _capi
import bench, types
args = types.SimpleNamespace(grid=200, max_range=8.0, map="obstacle_slam_10mx10m_5cm", start=None, rays=290, explore=False, particles=int(os.environ.get("N", "100000")))
ctx = bl.default_context()
m, truth, poses, odo, scans, rands = bench.build_inputs(args, 80, ctx)
cpm = np.float32(1.0 / np.float64(np.float32(0.05)))
g = bl.OccupancyGrid.from_cells(m["cells"], m["origin"], m["mpc"], cellsPerMeter=cpm, ctx=ctx)
pf = bl.ParticleFilter(int(os.environ.get("N", "100000")), ctx=ctx)
pf.initializeFilterAtPose(bl.make_pose(*odo[0], utime=int(scans[0].times[0])), seed=42)
mapper = bl.Mapping(5.0, 4, 1, ctx=ctx)
for k in range(60):
    sc = scans[k]
    pf.updateBegin(bl.make_pose(*odo[k + 1], utime=sc.utime), sc, g, int(rands[k]))
    mapper.updateMapFinishingFilter(sc, pf, sc.utime, g)
ctx.sync()
lib = ctx.lib
nb = 1400
buf = (C.c_ulonglong * (nb * 12))()
lib.bl_debug_mcl_stamps.restype = C.c_int
assert lib.bl_debug_mcl_stamps(buf, nb) == 0
t = np.array(buf[:], dtype=np.float64).reshape(nb, 12)
t = t[t[:, 0] > 0]
t0 = t[:, 0].min()
t[t == 0] = t0
us = (t - t0) * 0.01
main = us[:744]; tail = us[744:]
def q(v): return "min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f" % (v.min(), np.percentile(v, 10), np.median(v), np.percentile(v, 90), v.max())
print("workgroups seen:", len(us), "(first 744 = region 1)")
for name, r in (("region 1", main), ("region 2", tail)):
    if len(r) == 0: continue
    print(name, "entry           ", q(r[:, 0]))
    print(name, "entry -> barrier", q(r[:, 1] - r[:, 0]))
    print(name, "  ray table formed    ", q(r[:, 8] - r[:, 0]))
    print(name, "  staging loads issued", q(r[:, 4] - r[:, 0]))
    print(name, "  staging loads landed", q(r[:, 9] - r[:, 0]))
    print(name, "  bracket known       ", q(r[:, 5] - r[:, 0]))
    print(name, "  bisection done      ", q(r[:, 6] - r[:, 0]))
    print(name, "  prologue arithmetic ", q(r[:, 7] - r[:, 0]))
    print(name, "ray loop        ", q(r[:, 2] - r[:, 1]))
    print(name, "epilogue        ", q(r[:, 3] - r[:, 2]))
    print(name, "end             ", q(r[:, 3]))
