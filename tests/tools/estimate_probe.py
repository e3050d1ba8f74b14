"""How long estimatePosteriorPose takes as its own launch (k_scan_tile_sums + k_mcl_finish) and how many sub-tiles / phases the
two float sums replay, for clouds centred at various distances from the axes."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import botlab_amd as bl
from botlab_amd import _capi
from botlab_amd.host import PARTICLE_DTYPE

ctx = bl.default_context()
rng = np.random.default_rng(3)
for N in (100_000, 1_000_000):
    for cx, cy, spread in ((-0.75, 0.2, 0.02), (0.02, 0.7, 0.02), (0.001, -0.0005, 0.02), (0.0, 0.0, 0.05), (37.5, -12.0, 0.3)):
        p = np.zeros(N, PARTICLE_DTYPE)
        p["x"] = (cx + spread * rng.standard_normal(N)).astype(np.float32)
        p["y"] = (cy + spread * rng.standard_normal(N)).astype(np.float32)
        p["theta"] = (0.1 * rng.standard_normal(N)).astype(np.float32)
        units = (1000 * rng.integers(20, 400, N)).astype(np.uint32)
        pf = bl.ParticleFilter(N, ctx=ctx)
        pf.setParticles(p, units)
        pf.estimatePosteriorPose()
        ctx.timing_reset(); ctx.timing_enable(True)
        for _ in range(20):
            est = pf.estimatePosteriorPose()
        ctx.timing_enable(False)
        ms, n = ctx.timing_get(_capi.BL_K_MCL_SCAN)
        print(f"N {N:8d} centre ({cx:8.4f},{cy:8.4f}) spread {spread}: {1e3 * ms / n:7.1f} us per estimate  {list(pf.debugEstimateStats())}  -> ({est.x:.6f}, {est.y:.6f})", flush=True)
        pf.close()
