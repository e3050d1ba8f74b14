"""Timeline of one k_dist_fused launch: per tile the steps of the in-tile transform and of the exchange, per wave the steps of the row
pass (s_memrealtime stamps).  Needs the side library `make -C botlab_amd/csrc stamps` builds (libbotlab_hip_stamps.so: bl_planning.hip
with -DDF_STAMPS -DBL_ASTAR_STAMPS); the product library carries no stamps."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import botlab_amd._capi as capi
capi.LIB_PATH = capi.LIB_PATH.replace("libbotlab_hip.so", "libbotlab_hip_stamps.so")
import botlab_amd as bl, torch
ctx = bl.default_context()
side = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
rng = np.random.default_rng(side)
cells = np.where(rng.random((side, side)) < 0.01, 50, -7).astype(np.int8)
g = bl.OccupancyGrid.from_cells(cells, (0.0, 0.0), 0.05, ctx=ctx)
d = bl.ObstacleDistanceGrid(ctx=ctx)
for _ in range(6):
    d.forget(); d.setDistances(g)
torch.cuda.synchronize()
T = ((side + 127) // 128) ** 2
buf = (C.c_ulonglong * (48 * T))()
fn = ctx.lib.bl_dist_debug_fused_stamps
fn.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]; fn.restype = C.c_int
n = fn(d.h, buf, 48 * T)
allb = np.array(buf[:n], dtype=np.uint64).astype(np.int64)
a = allb[:16 * T].reshape(-1, 8)
wv = allb[16 * T:48 * T].reshape(T, 8, 4)
t0 = a[:, 0].min()
us = (a - t0) / 100.0                       # s_memrealtime: 100 MHz
S, A = us[:T], us[T:]
def q(v): return "min %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f" % (v.min(), np.median(v), np.percentile(v, 90), v.max())
names_s = {0: "start (loads issued)", 1: "row pass", 2: "rows in LDS, barrier", 3: "column ends, barrier", 4: "words out, barrier", 7: "D_T done"}
names_a = {0: "start", 4: "first loads back", 5: "all words there", 1: "decoded, in LDS", 2: "barrier", 3: "ring formed, barrier", 6: "ramps", 7: "stores issued"}
print("in-tile transform (us since the first workgroup's start; then the step lengths)")
prev = 0
for k in sorted(names_s): print("  %-22s %s   | step %s" % (names_s[k], q(S[:, k]), q(S[:, k] - S[:, prev]))); prev = k
print("exchange and ramps")
prev = 0
for k in (0, 4, 5, 1, 2, 3, 6, 7): print("  %-22s %s   | step %s" % (names_a[k], q(A[:, k]), q(A[:, k] - A[:, prev]))); prev = k
wus = (wv - t0) / 100.0
print("per wave, in-tile pass: cells in / row pass done / rows in LDS / row ends + quadrant numbers in LDS (us, median over tiles)")
for w in range(8): print("  wave %d: %s" % (w, "  ".join("%.2f" % np.median(wus[:, w, k]) for k in range(4))), " | max over tiles:", "  ".join("%.2f" % wus[:, w, k].max() for k in range(4)))
