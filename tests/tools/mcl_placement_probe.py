"""Which compute unit did every workgroup of k_mcl_main run on, and when did it end?  Stamped build with -DMCL_STAMPS
-DMCL_STAMPS_HW (BOTLAB_HIP_LIB points at it).  Prints workgroups per CU and the end time against the CU's load."""
import ctypes as C, os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import botlab_amd as bl
import bench
N = int(os.environ.get("N", "100000"))
args = types.SimpleNamespace(grid=200, max_range=8.0, map="obstacle_slam_10mx10m_5cm", start=None, rays=290, explore=False, particles=N)
ctx = bl.default_context()
m, truth, poses, odo, scans, rands = bench.build_inputs(args, 80, ctx)
cpm = np.float32(1.0 / np.float64(np.float32(0.05)))
g = bl.OccupancyGrid.from_cells(m["cells"], m["origin"], m["mpc"], cellsPerMeter=cpm, ctx=ctx)
pf = bl.ParticleFilter(N, ctx=ctx)
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
t = np.array(buf[:], dtype=np.uint64).reshape(nb, 12)
t = t[t[:, 0] > 0]
t0 = t[:, 0].min()
hw = t[:, 6]
xcc = (hw >> np.uint64(32)) & np.uint64(0xf)
cu = (hw >> np.uint64(8)) & np.uint64(0xf)
sh = (hw >> np.uint64(12)) & np.uint64(1)
se = (hw >> np.uint64(13)) & np.uint64(7)
key = (xcc.astype(np.int64) << 12) | (se.astype(np.int64) << 8) | (sh.astype(np.int64) << 4) | cu.astype(np.int64)
end = (t[:, 3] - t0).astype(np.float64) * 0.01
start = (t[:, 0] - t0).astype(np.float64) * 0.01
loop = (t[:, 2].astype(np.float64) - t[:, 1].astype(np.float64)) * 0.01
main = np.arange(len(t)) < 744
print("workgroups seen", len(t), "distinct CUs", len(np.unique(key)), "distinct XCCs", len(np.unique(xcc)))
per_cu = {}
for i in range(len(t)):
    per_cu.setdefault(int(key[i]), []).append(i)
hist = {}
for k_, v in per_cu.items():
    n1 = sum(1 for i in v if main[i])
    hist.setdefault(n1, []).append(k_)
for n1 in sorted(hist):
    cus = hist[n1]
    idx = [i for k_ in cus for i in per_cu[k_] if main[i]]
    if not idx: 
        print(f"CUs with {n1} region-1 workgroups: {len(cus)}"); continue
    print(f"CUs with {n1} region-1 workgroups: {len(cus):3d}   ray loop median {np.median(loop[idx]):5.1f} us   end median {np.median(end[idx]):5.1f}  max {np.max(end[idx]):5.1f}")
print("per XCC: region-1 workgroups, median end")
for x in sorted(set(int(v) for v in xcc)):
    idx = [i for i in range(len(t)) if int(xcc[i]) == x and main[i]]
    print(f"  xcc {x}: {len(idx):3d} workgroups on {len(set(int(key[i]) for i in idx)):2d} CUs, end median {np.median(end[idx]):5.1f} max {np.max(end[idx]):5.1f}")
# the first 40 workgroups in launch order: where they went
print("launch order -> (xcc, se, sh, cu):", [(int(xcc[i]), int(se[i]), int(sh[i]), int(cu[i])) for i in range(24)])
