"""BASELINE.json configs[2] as written: --localization-only MCL on data/convex_10mx10m_5cm.map (src/slam/slam.cpp:36-45 loads it)
with 1 000 000 particles -- against the CPU oracle consuming the same noise (src/slam/particle_filter.cpp:37-160; ~11 s of
oracle time per moved update at this size).

Two filters run beside ONE oracle pass: the default resampler (exact integer prefix) and the strict one
(bl_pf_set_strict_resampling: the reference's own rounded cumulative).
  * strict mode: resampling indices, likelihoods, particle poses and the estimate equal the oracle's everywhere;
  * default mode: the source indices that differ from the oracle's are COUNTED (hard part 4, predicts a
    handful per update at this size), each must be off by exactly one, and their number must stay within the stated bound;
    every other particle is bit-equal, and the filter is re-seated on the oracle's particles before the next update when any
    differed, so that the second update is compared from the same state."""
import json
import os

import numpy as np
import pytest

import helpers
import oracle_lib
import botlab_amd as bl
from botlab_amd import synth

pytestmark = pytest.mark.gpu

N = 1_000_000
MAX_INDEX_DISAGREEMENTS = 16          # per update, default mode (measured: see gpurun_out/config3_1m.json / DESIGN.md section 7)


def _units_from_raw(raw):
    half = np.rint(raw * 2.0).astype(np.int64)              # likelihoods are sums of k or k / 2: exact half-integers
    return np.where(half > 0, half * 1000, 2).astype(np.uint32)


def _bits(pose):
    return tuple(np.array([pose.x, pose.y, pose.theta], np.float32).view(np.uint32).tolist())


def test_config3_localization_on_the_convex_map_against_the_oracle(oracle, maps, gpu_ctx):
    m = maps["convex_10mx10m_5cm"]
    cells = m["cells"]
    truth = np.where(cells > 0, 127, -127).astype(np.int8)
    g = bl.OccupancyGrid.from_cells(cells, m["origin"], m["mpc"], cellsPerMeter=helpers.CPM_DEFAULT, ctx=gpu_ctx)
    poses = synth.square_trajectory((-0.4, -0.4, 0.0), 3, step_len=0.02, turn=0.05, side=0.8)
    scans = [synth.raycast_scan(truth, m["origin"], 0.05, poses[k - 1], poses[k], 1_000_000 + k * 100_000) for k in range(1, 4)]
    opf = oracle_lib.OraclePF(oracle, N)
    opf.init_at_pose(oracle.pose(-0.4, -0.4, 0.0, utime=int(scans[0].times[0])), 3)
    start = opf.particles()
    filters = {}
    for mode in ("default", "strict"):
        pf = bl.ParticleFilter(N, ctx=gpu_ctx)
        if mode == "strict":
            pf.setStrictResampling(True)
        pf.setParticles(start)
        pf.debugEnable(True)
        filters[mode] = pf
    report = {"particles": N, "map": "convex_10mx10m_5cm", "updates": []}
    rands = (1804289383, 846930886, 1681692777)             # glibc rand(), unseeded (particle_filter.cpp:92)
    moved = 0
    for k, sc in enumerate(scans):
        o = poses[k + 1]
        res = opf.update(oracle.pose(*o, utime=sc.utime), sc, cells, m["mpc"], helpers.CPM_DEFAULT, m["origin"], rands[k])
        exp = opf.particles() if res["moved"] else None
        for mode, pf in filters.items():
            pose = pf.updateFilter(bl.make_pose(*o, utime=sc.utime), sc, g, rand_value=rands[k], noise=res["noise"])
            assert pose.utime == res["pose"].utime == sc.utime
            if not res["moved"]:
                continue
            idx, like = pf.debugLast()
            bad = np.nonzero(idx != res["idx"])[0]
            same = np.ones(N, bool)
            same[bad] = False
            got = pf.particles()
            if mode == "strict":
                assert bad.size == 0, f"strict resampling: {bad.size} indices differ at update {k}"
            else:
                assert bad.size <= MAX_INDEX_DISAGREEMENTS, f"{bad.size} indices differ at update {k}"
                assert np.all(np.abs(idx[bad].astype(np.int64) - res["idx"][bad]) == 1)
            report["updates"].append({"update": k, "mode": mode, "index_disagreements": int(bad.size)})
            assert np.array_equal(like[same].astype(np.float64) * 0.5, res["raw"][same]), (mode, k)     # exact half-integers
            for f in ("x", "y", "theta", "p_x", "p_y", "p_theta"):
                assert np.array_equal(got[f][same], exp[f][same]), (mode, k, f)
            assert np.array_equal(got["utime"], exp["utime"]) and np.array_equal(got["p_utime"], exp["p_utime"])
            if bad.size == 0:
                assert np.allclose(got["weight"], exp["weight"], rtol=1e-5, atol=0)      # north_star: weights within 1e-5 relative
                assert _bits(pose) == _bits(res["pose"]), (mode, k)                       # x, y: the serial float sums, bit for bit
            else:
                # another source for a few particles: the estimate is then formed over another set; it must still be the
                # reference's loop over THAT set, and within the weight of the differing particles of the oracle's
                import ctypes as C
                want = oracle_lib.OPose()
                oracle.lib.orc_estimate_pose(np.ascontiguousarray(got).ctypes.data, N, C.byref(want))
                assert _bits(pose)[:2] == _bits(want)[:2]
                assert abs(pose.x - res["pose"].x) < 1e-4 and abs(pose.y - res["pose"].y) < 1e-4
                pf.setParticles(exp, _units_from_raw(res["raw"]))                        # the next update starts from the oracle's state
        if res["moved"]:
            moved += 1
            assert res["raw"].max() > 100.0                  # the cloud sits on the map: scores are real
    assert moved == 2
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", "config3_1m.json"), "w") as fh:
        json.dump(report, fh)
    for pf in filters.values():
        pf.close()
    g.close()
