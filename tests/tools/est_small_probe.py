import os, sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import botlab_amd as bl
from botlab_amd import _capi
from botlab_amd.host import PARTICLE_DTYPE
ctx = bl.default_context()
rng = np.random.default_rng(3)
for N in (100_000,):
    for cx, cy, spread in ((-0.75, 0.2, 0.02), (0.0, 0.0, 0.05)):
        p = np.zeros(N, PARTICLE_DTYPE)
        p["x"] = (cx + spread * rng.standard_normal(N)).astype(np.float32)
        p["y"] = (cy + spread * rng.standard_normal(N)).astype(np.float32)
        units = (1000 * rng.integers(20, 400, N)).astype(np.uint32)
        pf = bl.ParticleFilter(N, ctx=ctx)
        pf.setParticles(p, units)
        pf.estimatePosteriorPose()
        ctx.timing_reset(); ctx.timing_enable(True)
        for _ in range(5):
            est = pf.estimatePosteriorPose()
        ctx.timing_enable(False)
        ms, n = ctx.timing_get(_capi.BL_K_MCL_SCAN)
        print(os.environ.get("BOTLAB_MCL_NO_WILD"), f"N {N} centre ({cx},{cy}): {1e3*ms/n:9.1f} us  {list(pf.debugEstimateStats())}", flush=True)
        pf.close()
