"""estimatePosteriorPose on clouds centred on the origin / an axis (the float sums hover around zero): time per estimate (the
finish as its own launches) and the chain's counters -- per axis: generic replays, their phases, tables / maps taken, gaps walked."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import botlab_amd as bl
from botlab_amd import _capi
from botlab_amd.host import PARTICLE_DTYPE
ctx = bl.default_context()
for N in (100_000, 1_000_000):
    for centre in ((0.0, 0.0), (0.0, 0.6), (-0.9, 0.0), (-0.75, 0.2)):
        rng = np.random.default_rng(17)
        p = np.zeros(N, PARTICLE_DTYPE)
        p["x"] = (centre[0] + 0.05 * rng.standard_normal(N)).astype(np.float32)
        p["y"] = (centre[1] + 0.05 * rng.standard_normal(N)).astype(np.float32)
        p["theta"] = (0.1 * rng.standard_normal(N)).astype(np.float32)
        units = (1000 * rng.integers(20, 400, N)).astype(np.uint32)
        pf = bl.ParticleFilter(N, ctx=ctx)
        pf.setParticles(p, units)
        pf.estimatePosteriorPose()
        st = pf.debugEstimateStats()
        ctx.timing_reset(); ctx.timing_enable(True, kernels=[_capi.BL_K_MCL_SCAN])
        for _ in range(5):
            pf.estimatePosteriorPose()
        ctx.timing_enable(False)
        ms, n = ctx.timing_get(_capi.BL_K_MCL_SCAN)
        print("N %7d centre %-12s %7.1f us per estimate | x: replays %d phases %d maps %d gaps %d | y: replays %d phases %d maps %d gaps %d"
              % ((N, str(centre), 1e3 * ms / n) + tuple(st)), flush=True)
        pf.close()
