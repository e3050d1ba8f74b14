"""How often does the resampler pick another source particle than the reference?

resamplePosteriorDistribution (particle_filter.cpp:84-103) compares U_m = r + m / N with a cumulative weight c_i that is a
SEQUENTIALLY ROUNDED double sum; the kernels compare U_m * S with an exact integer prefix of the weight units.  The two rules can
part only where U_m lies within the rounding error of c_i of a partial sum.  This sweep counts the output particles whose
source index differs from the oracle's, over rand() values {0, 1, 1000, 2^30, RAND_MAX, glibc's first sixteen} and N in {4096,
100k, 300k}, for three kinds of weights, and asserts the bounds DESIGN.md quotes:
  * weights as an update leaves them (integer likelihoods, a few floored to 0.001): no difference at all;
  * weights of a fresh filter (all equal -- D2): every U_m sits ON a partial sum when r is 0, nearly 0 or exactly 1 / N
    (rand() <= ~1000 or == RAND_MAX: a 5e-7 chance on the ONE update that follows initialisation), so there the choice is
    decided by rounding on both sides; the differences are counted (half to three quarters of the particles, each off by
    one index) and bounded; for every other r: none."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import botlab_amd as bl
from botlab_amd.host import PARTICLE_DTYPE

pytestmark = pytest.mark.gpu

RAND_MAX = 2147483647
GLIBC = [1804289383, 846930886, 1681692777, 1714636915, 1957747793, 424238335, 719885386, 1649760492, 596516649, 1189641421,
         1025202362, 1350490027, 783368690, 1102520059, 2044897763, 1967513926]
EDGE = [0, 1, 1000, 1 << 30, RAND_MAX]


def _sweep(oracle, ctx, N, units, strict=False):
    p = np.zeros(N, PARTICLE_DTYPE)
    rng = np.random.default_rng(N)
    p["x"] = rng.standard_normal(N).astype(np.float32)
    pf = bl.ParticleFilter(N, ctx=ctx)
    out = {}
    try:
        pf.setStrictResampling(strict)
        pf.setParticles(p, units)
        host = np.ascontiguousarray(pf.particles())          # the weights units / S as doubles: what the reference would hold
        want = np.empty(N, np.int32)
        for rv in EDGE + GLIBC:
            got = pf.debugResample(rv)
            oracle.lib.orc_resample_indices(host.ctypes.data, N, rv, want.ctypes.data)
            d = np.nonzero(got != want)[0]
            assert np.all(np.abs(got[d].astype(np.int64) - want[d]) <= 1), (N, rv)        # never farther than the neighbour
            out[rv] = int(d.size)
    finally:
        pf.close()
    return out


@pytest.mark.parametrize("N", [200, 4096, 100_000, 300_000])
def test_resample_index_disagreements_are_counted_and_bounded(oracle, gpu_ctx, N):
    rng = np.random.default_rng(7 * N)
    report = {}
    # (a) weights as updateFilter leaves them: likelihood sums of 290 rays in half-units (x1000), some particles at the floor (2)
    units = (1000 * rng.integers(40, 36000, N)).astype(np.uint32)
    units[rng.random(N) < 0.02] = 2
    report["after_update"] = _sweep(oracle, gpu_ctx, N, units)
    assert sum(report["after_update"].values()) == 0, report
    # (b) a narrower spread (a converged filter: nearly equal likelihoods)
    units = (1000 * rng.integers(30000, 30400, N)).astype(np.uint32)
    report["converged"] = _sweep(oracle, gpu_ctx, N, units)
    assert sum(report["converged"].values()) == 0, report
    # (c) a fresh filter: all weights 1 / N.  r = 0, r ~ 0 and r = 1 / N (rand() == RAND_MAX) put EVERY U_m on a partial sum m / N of
    # the equal weights: which side the reference's rounded cumulative falls on is rounding noise, and the integer rule took the
    # neighbouring source for about half of the particles there (rounds 2-4; BOTLAB_NO_AUTO_STRICT=1 brings that rule back).  Equal
    # weights are detected (a fresh filter, an upload of equal weights) and resampled against the reference's own cumulative.
    report["uniform"] = _sweep(oracle, gpu_ctx, N, np.ones(N, np.uint32))
    assert sum(report["uniform"].values()) == 0, report["uniform"]
    report["uniform_other_value"] = _sweep(oracle, gpu_ctx, N, np.full(N, 30_000_000, np.uint32))
    assert sum(report["uniform_other_value"].values()) == 0, report["uniform_other_value"]
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", f"resample_sweep_{N}.json"), "w") as fh:
        json.dump({k: {str(r): c for r, c in v.items()} for k, v in report.items()}, fh)


@pytest.mark.parametrize("N", [4096, 100_000, 300_000])
def test_strict_resampling_has_no_disagreements(oracle, gpu_ctx, N):
    """bl_pf_set_strict_resampling: the resampler searches the reference's own sequentially rounded cumulative (formed bit for
    bit by k_pf_cumulative_strict) -- identical source indices for EVERY rand() value of the sweep and all three kinds of
    weights, the degenerate ones included."""
    rng = np.random.default_rng(11 * N)
    kinds = {"after_update": (1000 * rng.integers(40, 36000, N)).astype(np.uint32), "uniform": np.ones(N, np.uint32),
             "few_dominant": np.where(np.arange(N) % 997 == 0, 1000 * 36830, 2).astype(np.uint32)}
    kinds["after_update"][rng.random(N) < 0.02] = 2
    for name, units in kinds.items():
        rep = _sweep(oracle, gpu_ctx, N, units, strict=True)
        assert sum(rep.values()) == 0, (name, rep)


def test_strict_mode_updates_match_the_oracle(oracle, maps, gpu_ctx):
    """Whole updates in strict mode (the stand-alone finish + the cumulative kernel in front of every k_mcl_main): indices,
    likelihoods, particles and estimate as in the default mode's parity tests -- with rand() values the default mode cannot
    take on a fresh filter (0, RAND_MAX)."""
    import helpers
    import oracle_lib
    from botlab_amd import synth
    N = 20_000
    m = maps["obstacle_slam_10mx10m_5cm"]
    truth = np.where(m["cells"] > 0, 127, -127).astype(np.int8)
    g = bl.OccupancyGrid.from_cells(m["cells"], m["origin"], m["mpc"], cellsPerMeter=helpers.CPM_DEFAULT, ctx=gpu_ctx)
    poses = synth.square_trajectory((-0.75, 0.2, 0.0), 4, step_len=0.02, turn=0.05, side=0.8)
    scans = [synth.raycast_scan(truth, m["origin"], 0.05, poses[k - 1], poses[k], 1_000_000 + k * 100_000) for k in range(1, 5)]
    opf = oracle_lib.OraclePF(oracle, N)
    opf.init_at_pose(oracle.pose(-0.75, 0.2, 0.0, utime=int(scans[0].times[0])), 5)
    pf = bl.ParticleFilter(N, ctx=gpu_ctx)
    pf.setStrictResampling(True)
    pf.setParticles(opf.particles())
    pf.debugEnable(True)
    moved = 0
    for k, sc in enumerate(scans):
        o = poses[k + 1]
        rv = (0, 0, RAND_MAX, 1)[k]
        res = opf.update(oracle.pose(*o, utime=sc.utime), sc, m["cells"], m["mpc"], helpers.CPM_DEFAULT, m["origin"], rv)
        pose = pf.updateFilter(bl.make_pose(*o, utime=sc.utime), sc, g, rand_value=rv, noise=res["noise"])
        if not res["moved"]:
            continue
        moved += 1
        idx, like = pf.debugLast()
        assert np.array_equal(idx, res["idx"]), k
        assert np.array_equal(like.astype(np.float64) * 0.5, res["raw"]), k
        got, exp = pf.particles(), opf.particles()
        for f in ("x", "y", "theta", "p_x", "p_y", "p_theta"):
            assert np.array_equal(got[f], exp[f]), (k, f)
        assert (np.float32(pose.x), np.float32(pose.y), np.float32(pose.theta)) == (np.float32(res["pose"].x), np.float32(res["pose"].y), np.float32(res["pose"].theta))
    assert moved == 3
    pf.close()


def test_strict_mode_with_the_filter_end_riding_in_the_map_kernel(maps, gpu_ctx):
    """Strict mode keeps the riding form of the update's end (bl_mapping_update_finishing_pf): the cumulative's launches go behind the
    map kernel that carries the finish.  Same indices, particles, estimates and maps as updateFilter + updateMap in strict mode
    (which test_strict_mode_updates_match_the_oracle holds against the oracle), with rand() values that make the two resampling
    rules differ on equal weights."""
    import helpers
    from botlab_amd import synth
    N = 20_000
    m = maps["obstacle_slam_10mx10m_5cm"]
    truth = np.where(m["cells"] > 0, 127, -127).astype(np.int8)
    poses = synth.square_trajectory((-0.75, 0.2, 0.0), 6, step_len=0.02, turn=0.05, side=0.8)
    scans = [synth.raycast_scan(truth, m["origin"], 0.05, poses[k - 1], poses[k], 1_000_000 + k * 100_000) for k in range(1, 7)]
    res = []
    for riding in (False, True):
        g = bl.OccupancyGrid.from_cells(m["cells"], m["origin"], m["mpc"], cellsPerMeter=helpers.CPM_DEFAULT, ctx=gpu_ctx)
        pf = bl.ParticleFilter(N, ctx=gpu_ctx)
        pf.initializeFilterAtPose(bl.make_pose(-0.75, 0.2, 0.0, utime=int(scans[0].times[0])), seed=11)
        pf.setNoiseSeed(4)
        pf.setStrictResampling(True)
        pf.debugEnable(True)
        mapper = bl.Mapping(5.0, 4, 1, ctx=gpu_ctx)
        rec = []
        for k, sc in enumerate(scans):
            odo = bl.make_pose(*poses[k + 1], utime=sc.utime)
            rv = (0, 0, RAND_MAX, 1, 12345, 0)[k]
            if riding:
                pf.updateBegin(odo, sc, g, rv)
                mapper.updateMapFinishingFilter(sc, pf, sc.utime, g)
            else:
                pf.updateFilter(odo, sc, g, rand_value=rv, want_pose=False)
                mapper.updateMapDevicePose(sc, pf.poseDevicePtr(), sc.utime, g)
            p = pf.poseEstimate()
            idx, like = pf.debugLast()
            rec.append(((p.utime, p.x, p.y, p.theta), idx.copy(), like.copy(), pf.particles().copy(), g.cells().copy()))
        res.append(rec)
        pf.close(); g.close()
    for k, (a, b) in enumerate(zip(*res)):
        assert a[0] == b[0]
        if k > 0:                                   # (the very first update never moves -- action_model.cpp:26-31: nothing was resampled)
            assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
        assert a[3].tobytes() == b[3].tobytes()
        assert np.array_equal(a[4], b[4])


@pytest.mark.parametrize("N", [200, 1000, 4096, 12_345, 100_000])            # (200: the reference's default particle count, slam_main.cpp:21)
def test_all_floor_weights_are_resampled_as_the_reference_does(oracle, maps, gpu_ctx, N):
    """A lost filter: on a map that knows nothing (every cell 0) every particle's likelihood is 0, computeNormalizedPosterior
    (particle_filter.cpp:116-141) leaves N EQUAL weights 0.001 / wSum, and the next resampling compares U_m with partial sums that lie
    within rounding of every U_m when rand() is 0, nearly 0 or RAND_MAX -- where the integer rule took the neighbouring source for about
    half of the particles.  The launch that writes the total recognises the set (2 N units) and leaves the runs of the reference's own
    cumulative; whole updates then equal the oracle's: indices, particles, estimate -- for those rand() values, riding finish included."""
    import helpers
    import oracle_lib
    from botlab_amd import synth
    m = maps["obstacle_slam_10mx10m_5cm"]
    cells = np.zeros_like(m["cells"])
    truth = np.where(m["cells"] > 0, 127, -127).astype(np.int8)
    rvs = [12345, 0, 1, 1000, RAND_MAX, 1804289383, 0]
    poses = synth.square_trajectory((-0.75, 0.2, 0.0), len(rvs) + 1, step_len=0.02, turn=0.05, side=0.8)
    scans = [synth.raycast_scan(truth, m["origin"], 0.05, poses[k - 1], poses[k], 1_000_000 + k * 100_000) for k in range(1, len(rvs) + 2)]
    for riding in (False, True):
        g = bl.OccupancyGrid.from_cells(cells, m["origin"], m["mpc"], cellsPerMeter=helpers.CPM_DEFAULT, ctx=gpu_ctx)
        opf = oracle_lib.OraclePF(oracle, N)
        opf.init_at_pose(oracle.pose(-0.75, 0.2, 0.0, utime=int(scans[0].times[0])), 5)
        pf = bl.ParticleFilter(N, ctx=gpu_ctx)
        pf.setParticles(opf.particles())
        pf.debugEnable(True)
        assert pf.debugUniformRuns() > 0                       # a fresh filter: equal weights, known to the host
        mapper = bl.Mapping(5.0, 4, 1, ctx=gpu_ctx)
        moved = 0
        # update 0 latches the odometry (never moves), update 1 resamples the fresh filter and leaves the all-floor set, the others follow one
        for k, rv in enumerate([777] + rvs):
            sc, o = scans[k], poses[k + 1]
            res = opf.update(oracle.pose(*o, utime=sc.utime), sc, cells, m["mpc"], helpers.CPM_DEFAULT, m["origin"], rv)
            odo = bl.make_pose(*o, utime=sc.utime)
            if riding:
                pf.updateBegin(odo, sc, g, rv, noise=res["noise"])
                mapper.updateMapFinishingFilter(sc, pf, sc.utime, g)
                g.upload(cells)                                   # (the map stays unknown: only the filter is under test)
                pose = pf.poseEstimate()
            else:
                pose = pf.updateFilter(odo, sc, g, rand_value=rv, noise=res["noise"])
            if not res["moved"]:
                continue
            moved += 1
            idx, like = pf.debugLast()
            assert not like.any()                                 # every likelihood 0: every weight at the floor
            assert np.array_equal(idx, res["idx"]), (riding, k, rv, int(np.count_nonzero(idx != res["idx"])))
            got, exp = pf.particles(), opf.particles()
            for f in ("x", "y", "theta", "p_x", "p_y", "p_theta"):
                assert np.array_equal(got[f], exp[f]), (riding, k, f)
            assert (np.float32(pose.x), np.float32(pose.y), np.float32(pose.theta)) == (np.float32(res["pose"].x), np.float32(res["pose"].y), np.float32(res["pose"].theta))
            assert pf.debugUniformRuns() > 0, (riding, k)      # ... and the set it leaves is recognised again
        assert moved == len(rvs)
        pf.close(); g.close()


def test_ordinary_weights_leave_no_uniform_runs(oracle, maps, gpu_ctx):
    """... and an update on a real map does not: the prefix search stays in force (one compare and one store is all the recognition costs)."""
    import helpers
    from botlab_amd import synth
    m = maps["obstacle_slam_10mx10m_5cm"]
    truth = np.where(m["cells"] > 0, 127, -127).astype(np.int8)
    g = bl.OccupancyGrid.from_cells(m["cells"], m["origin"], m["mpc"], cellsPerMeter=helpers.CPM_DEFAULT, ctx=gpu_ctx)
    poses = synth.square_trajectory((-0.75, 0.2, 0.0), 4, step_len=0.02, turn=0.05, side=0.8)
    pf = bl.ParticleFilter(20_000, ctx=gpu_ctx)
    pf.initializeFilterAtPose(bl.make_pose(-0.75, 0.2, 0.0, utime=1_000_000), seed=3)
    assert pf.debugUniformRuns() > 0
    for k in range(1, 4):
        sc = synth.raycast_scan(truth, m["origin"], 0.05, poses[k - 1], poses[k], 1_000_000 + k * 100_000)
        pf.updateFilter(bl.make_pose(*poses[k], utime=sc.utime), sc, g, rand_value=99 + k)
    assert pf.debugUniformRuns() == 0
    pf.close(); g.close()
