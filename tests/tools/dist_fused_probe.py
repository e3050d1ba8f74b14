"""Whole-grid setDistances: the one-launch form against the four-launch form, HIP-event time per transform."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import botlab_amd as bl
from botlab_amd import _capi
ctx = bl.default_context()
for side in ([int(a) for a in sys.argv[1:]] or [1024, 2000, 4096, 8176]):
    rng = np.random.default_rng(side)
    cells = np.where(rng.random((side, side)) < 0.01, 50, -7).astype(np.int8)
    g = bl.OccupancyGrid.from_cells(cells, (0.0, 0.0), 0.05, ctx=ctx)
    res = {}
    for form in ("fused", "four"):
        if form == "four": os.environ["BOTLAB_DIST_NO_FUSED"] = "1"
        else: os.environ.pop("BOTLAB_DIST_NO_FUSED", None)
        d = bl.ObstacleDistanceGrid(ctx=ctx)
        for _ in range(5):
            d.forget(); d.setDistances(g)
        torch.cuda.synchronize()
        ctx.timing_reset(); ctx.timing_stride(1); ctx.timing_enable(True, kernels=[_capi.BL_K_DIST])
        for _ in range(20):
            d.forget(); d.setDistances(g)
        torch.cuda.synchronize()
        ctx.timing_enable(False)
        ms, n = ctx.timing_get(_capi.BL_K_DIST)
        t0 = time.perf_counter()
        for _ in range(50):
            d.forget(); d.setDistances(g)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / 50
        res[form] = (ms / n * 1e3, wall * 1e6, d.cells().copy())
        d.close()
    same = np.array_equal(res["fused"][2].view(np.uint32), res["four"][2].view(np.uint32))
    cellsn = side * side
    print("%5d^2: fused %.1f us (events) %.1f us (back to back)  |  four launches %.1f us / %.1f us  | equal %s | fused: %.2f TB/s at its 3 B/cell, %.2f at the four-launch form's 9"
          % (side, res["fused"][0], res["fused"][1], res["four"][0], res["four"][1], same, 3 * cellsn / res["fused"][1] / 1e6, 9 * cellsn / res["fused"][1] / 1e6), flush=True)
    g.close()
