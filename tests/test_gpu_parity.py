"""Parity tests proper: the HIP path (through the C ABI) against the CPU oracle on the same inputs.  Bit-exact for
int8 grids, distance floats, resample indices, likelihoods and A* paths; particle poses / weights / pose estimate
within the tolerances north_star states (1e-5 relative), written next to each assertion."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import helpers
import oracle_lib
import botlab_amd as bl
from botlab_amd import synth

pytestmark = pytest.mark.gpu

REL = 1e-5      # north_star: particle poses/weights within 1e-5 relative


def _grid_from_map(m, ctx):
    return bl.OccupancyGrid.from_cells(m["cells"], m["origin"], m["mpc"], cellsPerMeter=helpers.CPM_DEFAULT, ctx=ctx)


# ------------------------------------------------------------------ ObstacleDistanceGrid
@pytest.mark.parametrize("name", helpers.ALL_MAPS)
def test_distance_grid_bit_exact_on_shipped_maps(oracle, maps, gpu_ctx, name):
    m = maps[name]
    g = _grid_from_map(m, gpu_ctx)
    d = bl.ObstacleDistanceGrid(ctx=gpu_ctx)
    d.setDistances(g)
    exp = oracle.set_distances(m["cells"], m["mpc"], helpers.CPM_DEFAULT, m["origin"])
    assert np.array_equal(d.cells().view(np.uint32), exp.view(np.uint32))
    d.close(); g.close()


def test_distance_grid_reference_test_grid(oracle, gpu_ctx):
    # generate_grid of obstacle_distance_grid_test.cpp:172-196
    from test_oracle_pins import _generate_grid
    cells, lo, hi = _generate_grid()
    g = bl.OccupancyGrid(2.5, 2.5, 0.1, ctx=gpu_ctx)
    assert (g.width, g.height) == (25, 25)
    g.upload(cells)
    d = bl.ObstacleDistanceGrid(ctx=gpu_ctx)
    d.setDistances(g)
    got = d.cells()
    exp = oracle.set_distances(cells, g.mpc, g.cpm, g.origin)
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))
    assert np.all(got[cells == 0] == 0.0) and np.all(got[cells > 0] == 0.0)


@pytest.mark.parametrize("shape", [(1, 1), (1, 37), (53, 1), (64, 64), (257, 129), (300, 1000),
                                   (40, 8192),        # wide rows: the row pass chains two 4096-cell chunks
                                   (70, 12304),       # ... three chunks plus a ragged fourth (12304 = 3 * 4096 + 16)
                                   (600, 1040),       # large-grid column pass, last macro strip partial (600 = 4 * 128 + 88)
                                   (1029, 1024),      # ... one row into the ninth macro strip
                                   (515, 1026),       # W not a multiple of 16: wide rows off, large-grid column pass on
                                   (4230, 512)])      # more than 32 macro strips: no carry kernel, the apply kernel chains the summaries itself
def test_distance_grid_ragged_shapes(oracle, gpu_ctx, shape):
    rng = np.random.default_rng(shape[0] * 1000 + shape[1])
    h, w = shape
    density = 0.02 if h * w < 100000 else 0.0003          # sparse sources on the big ones: long carries across chunks / strips
    cells = np.where(rng.random(shape) < density, rng.integers(0, 100, shape), -rng.integers(1, 100, shape)).astype(np.int8)
    for variant in (cells, np.full(shape, -5, np.int8), np.zeros(shape, np.int8)):
        g = bl.OccupancyGrid.from_cells(variant, (-1.0, -2.0), 0.05, ctx=gpu_ctx)
        d = bl.ObstacleDistanceGrid(ctx=gpu_ctx)
        d.setDistances(g)
        exp = oracle.set_distances(variant, g.mpc, g.cpm, g.origin)
        assert np.array_equal(d.cells().view(np.uint32), exp.view(np.uint32))
        d.close(); g.close()


def test_distance_grid_2000x2000_tiled_maze(oracle, maps, gpu_ctx):
    world = synth.tile_world(maps["astar_maze"]["cells"], 2000)
    g = bl.OccupancyGrid.from_cells(world, (-50.0, -50.0), 0.05, ctx=gpu_ctx)
    d = bl.ObstacleDistanceGrid(ctx=gpu_ctx)
    d.setDistances(g)
    exp = oracle.set_distances(world, g.mpc, g.cpm, g.origin)
    assert np.array_equal(d.cells().view(np.uint32), exp.view(np.uint32))


# ------------------------------------------------------------------ Mapping
def _drive(maps, name, steps, seed):
    m = maps[name]
    truth = np.where(m["cells"] > 0, 127, -127).astype(np.int8)
    rng = np.random.default_rng(seed)
    poses = synth.square_trajectory((0.0, 0.0, 0.0), steps, step_len=0.04, turn=0.1, side=0.4)
    scans = []
    for k in range(1, len(poses)):
        t = 1_000_000 + k * 100_000
        scans.append(synth.raycast_scan(truth, m["origin"], float(m["mpc"]), poses[k - 1], poses[k], t, noise_sigma=0.005, rng=rng))
    return m, truth, poses, scans


@pytest.mark.parametrize("name,hit,miss", [("obstacle_slam_10mx10m_5cm", 4, 1), ("astar_maze", 3, 2), ("convex_10mx10m_5cm", 60, 45)])
def test_mapping_bit_exact(oracle, maps, gpu_ctx, name, hit, miss):
    m, truth, poses, scans = _drive(maps, name, 25, 11)
    g = bl.OccupancyGrid(10.0, 10.0, 0.05, ctx=gpu_ctx)                   # slam.cpp:23
    mapper = bl.Mapping(5.0, hit, miss, ctx=gpu_ctx)                        # slam.cpp:24
    om = oracle_lib.OracleMapping(oracle, 5.0, hit, miss)
    ref = np.zeros((g.height, g.width), np.int8)
    for k, scan in enumerate(scans):
        p = poses[k + 1]
        # a ranges array with invalid (<= 0.15) and beyond-max entries exercises both cuts
        if k == 3:
            scan.ranges[::7] = 0.1
            scan.ranges[5::11] = 7.5
        mapper.updateMap(scan, bl.make_pose(p[0], p[1], p[2], utime=scan.times[-1]), g)
        om.update(scan, oracle.pose(p[0], p[1], p[2], utime=scan.times[-1]), ref, g.mpc, g.cpm, g.origin)
        assert np.array_equal(g.cells(), ref), f"step {k}"
    assert (ref != 0).sum() > 300


def test_mapping_off_grid_and_empty_scan(oracle, gpu_ctx, maps):
    m, truth, poses, scans = _drive(maps, "obstacle_slam_10mx10m_5cm", 4, 5)
    g = bl.OccupancyGrid(10.0, 10.0, 0.05, ctx=gpu_ctx)
    mapper = bl.Mapping(5.0, 4, 1, ctx=gpu_ctx)
    om = oracle_lib.OracleMapping(oracle, 5.0, 4, 1)
    ref = np.zeros((g.height, g.width), np.int8)
    far = [(4.9, 4.9, 0.3), (4.95, 4.8, 0.5), (6.0, 6.0, 1.0), (-4.99, -4.99, 2.0)]     # rays leave the grid / robot outside it
    for k, (scan, p) in enumerate(zip(scans, far)):
        mapper.updateMap(scan, bl.make_pose(*p, utime=scan.times[-1]), g)
        om.update(scan, oracle.pose(*p, utime=scan.times[-1]), ref, g.mpc, g.cpm, g.origin)
        assert np.array_equal(g.cells(), ref), k
    empty = bl.LidarScan(np.zeros(0, np.float32), np.zeros(0, np.float32), np.zeros(0, np.int64), utime=5)
    mapper.updateMap(empty, bl.make_pose(0, 0, 0, utime=5), g)
    assert np.array_equal(g.cells(), ref)


@pytest.mark.parametrize("rays,mpc,max_laser", [(1500, 0.05, 5.0),      # more rays than threads: the per-ray walk + direct hits path
                                                 (290, 0.01, 8.0),       # 1 cm cells, 8 m: the counter window needs strips
                                                 (700, 0.02, 6.0)])
def test_mapping_unusual_shapes(oracle, maps, gpu_ctx, rays, mpc, max_laser):
    m = maps["obstacle_slam_10mx10m_5cm"]
    truth = np.where(m["cells"] > 0, 127, -127).astype(np.int8)
    poses = synth.square_trajectory((-0.75, 0.2, 0.0), 6, step_len=0.03, turn=0.08, side=0.8)
    g = bl.OccupancyGrid(10.0, 10.0, mpc, ctx=gpu_ctx)
    mapper = bl.Mapping(max_laser, 4, 1, ctx=gpu_ctx)
    om = oracle_lib.OracleMapping(oracle, max_laser, 4, 1)
    ref = np.zeros((g.height, g.width), np.int8)
    for k in range(1, len(poses)):
        scan = synth.raycast_scan(truth, m["origin"], 0.05, poses[k - 1], poses[k], 1_000_000 + k * 100_000, rays=rays)
        p = poses[k]
        mapper.updateMap(scan, bl.make_pose(p[0], p[1], p[2], utime=scan.utime), g)
        om.update(scan, oracle.pose(p[0], p[1], p[2], utime=scan.utime), ref, g.mpc, g.cpm, g.origin)
        assert np.array_equal(g.cells(), ref), f"step {k}"
    assert (ref != 0).sum() > 1000


# ------------------------------------------------------------------ ParticleFilter
def _assert_estimate_bit_equal(pose, want, where):
    """estimatePosteriorPose (particle_filter.cpp:144-160): pose.x / pose.y are the reference's serially rounded FLOAT
    accumulators, reproduced bit for bit (bl_serial_sum.h); theta = (float)atan2 of two double sums."""
    got = np.array([pose.x, pose.y, pose.theta], np.float32).view(np.uint32)
    exp = np.array([want.x, want.y, want.theta], np.float32).view(np.uint32)
    assert np.array_equal(got, exp), (where, (pose.x, pose.y, pose.theta), (want.x, want.y, want.theta))


def _mcl_sequence(oracle, maps, gpu_ctx, N, steps, name="obstacle_slam_10mx10m_5cm", seed=1):
    m, truth, poses, scans = _drive(maps, name, steps, seed)
    rng = np.random.default_rng(seed + 100)
    odo = synth.odometry_from_truth(poses, rng)
    cells = m["cells"]
    g = _grid_from_map(m, gpu_ctx)
    opf = oracle_lib.OraclePF(oracle, N)
    t0 = int(scans[0].times[0])
    opf.init_at_pose(oracle.pose(0.0, 0.0, 0.0, utime=t0), 1234)
    pf = bl.ParticleFilter(N, ctx=gpu_ctx)
    pf.setParticles(opf.particles())
    pf.debugEnable(True)
    rands = [1804289383, 846930886, 1681692777, 1714636915, 1957747793, 424238335, 719885386, 1649760492, 596516649,
             1189641421, 1025202362, 1350490027, 783368690, 1102520059, 2044897763, 1967513926]     # glibc rand(), unseeded
    return m, g, opf, pf, odo, scans, rands, cells


@pytest.mark.parametrize("N", [200, 4096])
def test_mcl_parity_with_oracle_noise(oracle, maps, gpu_ctx, N):
    m, g, opf, pf, odo, scans, rands, cells = _mcl_sequence(oracle, maps, gpu_ctx, N, 10)
    moved_updates = 0
    for k, scan in enumerate(scans):
        o = odo[k + 1] if k != 4 else odo[k]              # step 4 repeats the odometry: "robot did not move" branch
        if k == 5:
            o = odo[k + 1]
        t = int(scan.times[-1])
        res = opf.update(oracle.pose(o[0], o[1], o[2], utime=t), scan, cells, m["mpc"], helpers.CPM_DEFAULT, m["origin"], rands[k])
        pose = pf.updateFilter(bl.make_pose(o[0], o[1], o[2], utime=t), scan, g, rand_value=rands[k], noise=res["noise"])
        assert pose.utime == res["pose"].utime == t
        if not res["moved"]:
            continue
        moved_updates += 1
        idx, like = pf.debugLast()
        assert np.array_equal(idx, res["idx"]), f"resample indices differ at step {k}"            # index work: exact
        assert np.array_equal(like.astype(np.float64) * 0.5, res["raw"]), f"likelihoods differ at step {k}"   # exact half-integers
        got, exp = pf.particles(), opf.particles()
        for f in ("x", "y", "theta", "p_x", "p_y", "p_theta"):
            assert np.allclose(got[f], exp[f], rtol=REL, atol=1e-7), (k, f)
            assert np.array_equal(got[f], exp[f]), (k, f)      # in practice bit-equal; a failure here is informational
        assert np.array_equal(got["utime"], exp["utime"]) and np.array_equal(got["p_utime"], exp["p_utime"])
        assert np.allclose(got["weight"], exp["weight"], rtol=REL, atol=0)
        _assert_estimate_bit_equal(pose, res["pose"], (N, k))
    assert moved_updates >= 7


@pytest.mark.parametrize("N", [100_000, 300_000])
def test_mcl_parity_at_the_headline_size(oracle, maps, gpu_ctx, N):
    """BASELINE.json's configuration itself -- 100 000 particles, 290 rays, the shipped obstacle_slam map -- against the
    oracle consuming the same noise: resampling indices and likelihoods exact, particle poses bit-equal, weights within 1e-5
    relative, pose estimate bit-equal (the launch shape of this size: 4 lanes per particle, shared prologue, both regions);
    and 300 000 particles, the one-lane-per-particle shape of the large configurations."""
    m = maps["obstacle_slam_10mx10m_5cm"]
    truth = np.where(m["cells"] > 0, 127, -127).astype(np.int8)
    g = _grid_from_map(m, gpu_ctx)
    poses = synth.square_trajectory((-0.75, 0.2, 0.0), 3, step_len=0.02, turn=0.05, side=0.8)
    scans = [synth.raycast_scan(truth, m["origin"], 0.05, poses[k - 1], poses[k], 1_000_000 + k * 100_000) for k in range(1, 4)]
    opf = oracle_lib.OraclePF(oracle, N)
    opf.init_at_pose(oracle.pose(-0.75, 0.2, 0.0, utime=int(scans[0].times[0])), 5)
    pf = bl.ParticleFilter(N, ctx=gpu_ctx)
    pf.setParticles(opf.particles())
    pf.debugEnable(True)
    moved = 0
    for k, sc in enumerate(scans):
        o = poses[k + 1]
        # rand() values as glibc draws them (a value near 0 puts every U_m within rounding distance of a partial sum of the
        # uniform initial weights, where the reference's sequentially rounded sum and the exact integer rule may part:
        # DESIGN.md section 7, "Resampling rule")
        rv = (1804289383, 846930886, 1681692777)[k]
        res = opf.update(oracle.pose(*o, utime=sc.utime), sc, m["cells"], m["mpc"], helpers.CPM_DEFAULT, m["origin"], rv)
        pose = pf.updateFilter(bl.make_pose(*o, utime=sc.utime), sc, g, rand_value=rv, noise=res["noise"])
        if not res["moved"]:
            continue
        moved += 1
        idx, like = pf.debugLast()
        assert np.array_equal(idx, res["idx"])
        assert np.array_equal(like.astype(np.float64) * 0.5, res["raw"])
        got, exp = pf.particles(), opf.particles()
        for f in ("x", "y", "theta", "p_x", "p_y", "p_theta"):
            assert np.array_equal(got[f], exp[f]), (k, f)
        assert np.allclose(got["weight"], exp["weight"], rtol=REL, atol=0)
        _assert_estimate_bit_equal(pose, res["pose"], (N, k))
    assert moved == 2


@pytest.mark.parametrize("N,rays", [(300, 1700), (5000, 900), (40, 2400)])
def test_mcl_scans_longer_than_the_lds_ray_table(oracle, maps, gpu_ctx, N, rays):
    """The ray loop reads (range, theta) from an LDS table of 768 entries; longer scans go through it in chunks with the
    table refilled behind barriers -- with 4, 64 and 1 lanes per particle here.  Likelihoods and indices stay exact."""
    m = maps["obstacle_slam_10mx10m_5cm"]
    truth = np.where(m["cells"] > 0, 127, -127).astype(np.int8)
    g = _grid_from_map(m, gpu_ctx)
    poses = synth.square_trajectory((-0.75, 0.2, 0.0), 4, step_len=0.02, turn=0.05, side=0.8)
    scans = [synth.raycast_scan(truth, m["origin"], 0.05, poses[k - 1], poses[k], 1_000_000 + k * 100_000, rays=rays) for k in range(1, 5)]
    assert scans[0].num_ranges == rays
    opf = oracle_lib.OraclePF(oracle, N)
    opf.init_at_pose(oracle.pose(-0.75, 0.2, 0.0, utime=int(scans[0].times[0])), 9)
    pf = bl.ParticleFilter(N, ctx=gpu_ctx)
    pf.setParticles(opf.particles())
    pf.debugEnable(True)
    moved = 0
    for k, sc in enumerate(scans):
        o = poses[k + 1]
        res = opf.update(oracle.pose(*o, utime=sc.utime), sc, m["cells"], m["mpc"], helpers.CPM_DEFAULT, m["origin"], 31 + k)
        pf.updateFilter(bl.make_pose(*o, utime=sc.utime), sc, g, rand_value=31 + k, noise=res["noise"])
        if not res["moved"]:
            continue
        moved += 1
        idx, like = pf.debugLast()
        assert np.array_equal(idx, res["idx"]), k
        assert np.array_equal(like.astype(np.float64) * 0.5, res["raw"]), k
    assert moved >= 2


@pytest.mark.parametrize("case", ["far_particles", "long_ray", "thetas_negative", "thetas_many_turns"])
def test_mcl_packed_scoring_fallbacks(oracle, maps, gpu_ctx, case):
    """The whole-grid LDS mode scores rays in packed int16 arithmetic only while every cell coordinate provably fits:
    particles more than 8191 cells from the grid origin, or a scan with a ray longer than 4000 cells, must take the
    int32 path -- per lane in the first case, for the whole launch in the second -- and still match the oracle exactly.
    The ray loop also skips the general wrap_to_pi when every theta of the scan lies in [0, 6.2831]: scans whose thetas
    are negative or several turns away take the general form (downward and repeated 2*pi steps)."""
    N = 512
    m, g, opf, pf, odo, scans, rands, cells = _mcl_sequence(oracle, maps, gpu_ctx, N, 4)
    parts = opf.particles()
    if case == "far_particles":                              # every 5th particle 450-600 m away (|cell| > 8191)
        parts["x"][::5] += np.float32(450.0) * np.where(np.arange(len(parts["x"][::5])) % 2, 1, -1).astype(np.float32)
        parts["y"][2::5] -= np.float32(600.0)
        parts["p_x"][:] = parts["x"]; parts["p_y"][:] = parts["y"]
        opf.set_particles(parts)
        pf.setParticles(parts)
    moved = 0
    for k, scan in enumerate(scans):
        if case == "long_ray":
            scan.ranges[7] = np.float32(260.0)               # 5200 cells at 5 cm
        if case == "thetas_negative":
            scan.thetas[:] = (scan.thetas - np.float32(6.0)).astype(np.float32)
        if case == "thetas_many_turns":
            scan.thetas[::3] = (scan.thetas[::3] + np.float32(4 * np.pi)).astype(np.float32)
            scan.thetas[1::3] = (scan.thetas[1::3] - np.float32(6 * np.pi)).astype(np.float32)
        o = odo[k + 1]
        t = int(scan.times[-1])
        res = opf.update(oracle.pose(o[0], o[1], o[2], utime=t), scan, cells, m["mpc"], helpers.CPM_DEFAULT, m["origin"], rands[k])
        pf.updateFilter(bl.make_pose(o[0], o[1], o[2], utime=t), scan, g, rand_value=rands[k], noise=res["noise"])
        if not res["moved"]:
            continue
        moved += 1
        idx, like = pf.debugLast()
        assert np.array_equal(idx, res["idx"]), k
        assert np.array_equal(like.astype(np.float64) * 0.5, res["raw"]), k
    assert moved >= 2


@pytest.mark.parametrize("start,kidnap,env", [
    ((3.3, -7.1, 0.4), False, {}),
    ((-11.0, 6.0, 2.0), True, {}),
    ((3.3, -7.1, 0.4), False, {"BOTLAB_MCL_WINDOW": "40"}),          # a window far smaller than the scan's reach: most rays miss it
    ("corner_lo", False, {}), ("corner_hi", False, {"BOTLAB_MCL_WINDOW": "120"}),   # window hanging over the grid's edges
    ((3.3, -7.1, 0.4), False, {"BOTLAB_MCL_NO_WINDOW": "1"}),         # every gather through L2 from the zero-framed copy
    ((3.3, -7.1, 0.4), False, {"BOTLAB_MCL_NO_FRAMED": "1"}),         # ... from the grid itself (unpacked scoring)
])
def test_mcl_parity_large_grid_lds_window(oracle, maps, gpu_ctx, start, kidnap, env, monkeypatch):
    """A 1000x1000 grid does not fit the whole-grid LDS staging: the kernel stages a window of the zero-framed copy around
    the predicted pose, sized to the scan's reach, and gathers from the copy through L2 outside it.  kidnap=True puts the
    particle cloud far from the pose the window is centred on, so every gather takes the second path."""
    N = 3000
    for k_, v_ in env.items():
        monkeypatch.setenv(k_, v_)                      # read when the filter is created
    world = synth.tile_world(maps["astar_maze"]["cells"], 1000)
    origin, mpc, cpm = (np.float32(-25.0), np.float32(-25.0)), np.float32(0.05), helpers.CPM_DEFAULT
    corner = isinstance(start, str)
    if corner:                                          # a free cell next to the grid's first / last corner
        free = np.argwhere(world[:40, :40] <= 0) if start == "corner_lo" else np.argwhere(world[-40:, -40:] <= 0) + 960
        cy_, cx_ = free[len(free) // 2]
        start = (-25.0 + (cx_ + 0.5) * 0.05, -25.0 + (cy_ + 0.5) * 0.05, 0.7)
    cells = np.where(world > 0, 100, -60).astype(np.int8)
    g = bl.OccupancyGrid.from_cells(cells, origin, mpc, cellsPerMeter=cpm, ctx=gpu_ctx)
    poses = synth.square_trajectory(start, 6, step_len=0.05, turn=0.1, side=0.2)
    rng = np.random.default_rng(5)
    odo = synth.odometry_from_truth(poses, rng)
    opf = oracle_lib.OraclePF(oracle, N)
    opf.init_at_pose(oracle.pose(*start, utime=1000), 99)
    parts = opf.particles()
    if kidnap:      # the window is centred on the LAST particle's pose (bl_pf_set_particles); move it far away
        parts["x"][-1] += 12.0
        parts["y"][-1] -= 9.0
        opf.set_particles(parts)
    pf = bl.ParticleFilter(N, ctx=gpu_ctx)
    pf.setParticles(parts)
    pf.debugEnable(True)
    for k in range(1, len(poses)):
        scan = synth.raycast_scan(world, origin, 0.05, poses[k - 1], poses[k], 1000 + 100000 * k)
        o = odo[k]
        res = opf.update(oracle.pose(*o, utime=scan.utime), scan, cells, mpc, cpm, origin, 777 + k)
        pose = pf.updateFilter(bl.make_pose(*o, utime=scan.utime), scan, g, rand_value=777 + k, noise=res["noise"])
        if not res["moved"]:
            continue
        idx, like = pf.debugLast()
        assert np.array_equal(idx, res["idx"])
        assert np.array_equal(like.astype(np.float64) * 0.5, res["raw"])
        got, exp = pf.particles(), opf.particles()
        for f in ("x", "y", "theta"):
            assert np.allclose(got[f], exp[f], rtol=REL, atol=1e-7)
        assert np.allclose(got["weight"], exp["weight"], rtol=REL, atol=0)
        _assert_estimate_bit_equal(pose, res["pose"], (start, kidnap, k))
        assert corner or (res["raw"] > 0).sum() > N // 2        # the scans really hit the map


@pytest.mark.parametrize("where", ["centre", "corner"])
def test_mcl_parity_at_config5_shape(oracle, maps, gpu_ctx, where):
    """BASELINE.json configs[4]'s shape: a 4096x4096 grid (16 MiB: LDS window mode at full stride, zero-framed copy) and
    256 000 particles (one lane per particle, the large finish groups), two moved updates against the oracle consuming the same
    noise: resampling indices, likelihoods, particle poses and the pose estimate exact.  "corner": the cloud sits in the grid's
    last corner, so the window hangs over two edges of the grid at that stride and most rays leave the map."""
    N, size = 256_000, 4096
    world = synth.tile_world(maps["astar_maze"]["cells"], size)
    half = size * 0.05 / 2.0
    origin, mpc, cpm = (np.float32(-half), np.float32(-half)), np.float32(0.05), helpers.CPM_DEFAULT
    if where == "corner":
        free = np.argwhere(world[-40:, -40:] <= 0) + (size - 40)
        cy_, cx_ = free[len(free) // 2]
        start = (-half + (cx_ + 0.5) * 0.05, -half + (cy_ + 0.5) * 0.05, 0.7)
    else:
        start = (0.3, 0.3, 0.0)
    cells = np.where(world > 0, 100, -60).astype(np.int8)
    g = bl.OccupancyGrid.from_cells(cells, origin, mpc, cellsPerMeter=cpm, ctx=gpu_ctx)
    poses = synth.square_trajectory(start, 3, step_len=0.05, turn=0.1, side=0.2)
    odo = synth.odometry_from_truth(poses, np.random.default_rng(15))
    opf = oracle_lib.OraclePF(oracle, N)
    opf.init_at_pose(oracle.pose(*start, utime=1000), 21)
    pf = bl.ParticleFilter(N, ctx=gpu_ctx)
    pf.setParticles(opf.particles())
    pf.debugEnable(True)
    moved = 0
    for k in range(1, len(poses)):
        scan = synth.raycast_scan(world, origin, 0.05, poses[k - 1], poses[k], 1000 + 100000 * k)
        o = odo[k]
        rv = (1804289383, 846930886, 1681692777)[k - 1]
        res = opf.update(oracle.pose(*o, utime=scan.utime), scan, cells, mpc, cpm, origin, rv)
        pose = pf.updateFilter(bl.make_pose(*o, utime=scan.utime), scan, g, rand_value=rv, noise=res["noise"])
        if not res["moved"]:
            continue
        moved += 1
        idx, like = pf.debugLast()
        assert np.array_equal(idx, res["idx"]), (where, k)
        assert np.array_equal(like.astype(np.float64) * 0.5, res["raw"]), (where, k)
        got, exp = pf.particles(), opf.particles()
        for f in ("x", "y", "theta", "p_x", "p_y", "p_theta"):
            assert np.array_equal(got[f], exp[f]), (where, k, f)
        assert np.allclose(got["weight"], exp["weight"], rtol=REL, atol=0)
        _assert_estimate_bit_equal(pose, res["pose"], (where, k))
        assert (res["raw"] > 0).sum() > N // 2                 # the scans really hit the map
    assert moved == 2


def test_mcl_action_only(oracle, maps, gpu_ctx):
    N = 512
    m, g, opf, pf, odo, scans, rands, cells = _mcl_sequence(oracle, maps, gpu_ctx, N, 5)
    for k in range(4):
        o = odo[k + 1]
        t = int(scans[k].times[-1])
        exp_pose, noise = opf.update_action_only(oracle.pose(o[0], o[1], o[2], utime=t))
        got_pose = pf.updateFilterActionOnly(bl.make_pose(o[0], o[1], o[2], utime=t), noise=noise)
        assert (got_pose.x, got_pose.y, got_pose.theta, got_pose.utime) == (exp_pose.x, exp_pose.y, exp_pose.theta, exp_pose.utime)
        got, exp = pf.particles(), opf.particles()
        for f in ("x", "y", "theta", "p_x", "p_y", "p_theta"):
            assert np.allclose(got[f], exp[f], rtol=REL, atol=1e-7), (k, f)
        assert np.allclose(got["weight"], exp["weight"], rtol=REL)


def test_mcl_philox_mode_statistics(maps, gpu_ctx):
    """Perf mode (device Philox noise) has no serial-RNG counterpart in the reference: check the proposal moments and
    that the filter tracks the truth."""
    N = 20000
    m = maps["obstacle_slam_10mx10m_5cm"]
    truth = np.where(m["cells"] > 0, 127, -127).astype(np.int8)
    g = _grid_from_map(m, gpu_ctx)
    pf = bl.ParticleFilter(N, ctx=gpu_ctx)
    pf.initializeFilterAtPose(bl.make_pose(0, 0, 0, utime=1000), seed=42)
    p0 = pf.particles()
    assert abs(p0["x"].mean()) < 5e-4 and abs(p0["x"].std() - 0.01) < 5e-4 and abs(p0["theta"].std() - 0.01) < 5e-4
    assert p0["x"][-1] == 0 and p0["theta"][-1] == 0 and np.allclose(p0["weight"], 1.0 / N)
    poses = synth.square_trajectory((0.0, 0.0, 0.0), 12, step_len=0.04, turn=0.1, side=0.4)
    pf.updateFilter(bl.make_pose(0, 0, 0, utime=1000), synth.raycast_scan(truth, m["origin"], 0.05, poses[0], poses[0], 1000), g)
    for k in range(1, len(poses)):
        scan = synth.raycast_scan(truth, m["origin"], 0.05, poses[k - 1], poses[k], 1000 + k * 100000)
        est = pf.updateFilter(bl.make_pose(*poses[k], utime=scan.utime), scan, g)
    assert abs(est.x - poses[-1][0]) < 0.05 and abs(est.y - poses[-1][1]) < 0.05
    w = pf.particles()["weight"]
    assert abs(w.sum() - 1.0) < 1e-9 and (w > 0).all()


# ------------------------------------------------------------------ search_for_path
# narrow case 2 needs 2.6e8 pops: excluded on BOTH sides (the oracle pin skips it too, tests/test_oracle_pins.py EXPECTED).
# convex case 2 (1.8e6 pops) and wide case 2 (5.3e5 pops) run once each in test_astar_long_cases.
ASTAR_GPU_SKIP = {("narrow", 2): "2.6e8 pops, excluded on both sides", ("convex", 2): "runs in test_astar_long_cases",
                  ("wide", 2): "runs in test_astar_long_cases"}


@pytest.mark.parametrize("name", ["empty", "filled", "narrow", "wide", "convex", "maze"])
def test_astar_paths_bit_exact_on_reference_fixtures(oracle, maps, gpu_ctx, name):
    m = maps["astar_" + name]
    cpm = helpers.CPM_DEFAULT
    g = _grid_from_map(m, gpu_ctx)
    params = bl.MotionPlannerParams(0.1)                                    # astar_test.cpp:227-228
    planner = bl.MotionPlanner(params, ctx=gpu_ctx)
    planner.setMap(g)
    dist = oracle.set_distances(m["cells"], m["mpc"], cpm, m["origin"])
    assert np.array_equal(planner.distances_.cells().view(np.uint32), dist.view(np.uint32))
    for i, row in enumerate(helpers.load_astar_cases()[name]):
        if (name, i) in ASTAR_GPU_SKIP:
            continue
        s = bl.make_pose(row["start"][0], row["start"][1], 0.0)
        gl = bl.make_pose(row["goal"][0], row["goal"][1], 0.0)
        os_, og = oracle.pose(*row["start"], 0.0), oracle.pose(*row["goal"], 0.0)
        assert planner.isValidGoal(gl) == oracle.is_valid_goal(og, dist, m["mpc"], cpm, m["origin"], 0.1, 0.1)
        path, stats = bl.search_for_path(s, gl, planner.distances_, planner.searchParams_, return_stats=True)
        exp, est = oracle.search(os_, og, dist, m["mpc"], cpm, m["origin"], 0.1, 1.0)
        assert len(path) == len(exp), (name, i)
        assert stats == est, (name, i, stats, est)
        got = np.array([(p.utime, p.x, p.y, p.theta) for p in path], dtype=exp.dtype)
        assert got.tobytes() == exp.tobytes(), (name, i)


def test_astar_narrow_case_2_ends_with_capacity_error(maps):
    """data/astar/narrow_poses.txt:4 (0 -5 -> 0 5, shouldExist 0): the gap is narrower than the robot, so the reference's search has to
    exhaust the whole near side before it answers "no path" -- 2.6e8 pops of its algorithm, an open list of as many entries
    (tests/tools/astar_narrow2_probe.py: the HIP path's default 16 M-entry list is full after 16 828 121 pops, 25.6 s).  The defined
    outcome of the drop-in: BL_ERR_CAPACITY with the 1-pose path of a failed plan, at the same pop count every time -- never a
    wrong path, never an endless search.  Run here with a 1 M-entry list."""
    import ctypes as C
    from botlab_amd import _capi
    ctx = bl.Context(0)
    m = maps["astar_narrow"]
    g = _grid_from_map(m, ctx)
    planner = bl.MotionPlanner(bl.MotionPlannerParams(0.1), ctx=ctx)
    planner.setMap(g)
    row = helpers.load_astar_cases()["narrow"][2]
    assert row["should_exist"] is False
    s, gl = bl.make_pose(*row["start"], 0.0), bl.make_pose(*row["goal"], 0.0)
    assert ctx.lib.bl_astar_set_open_capacity(ctx.h, 1 << 20) == 0
    seen = []
    for _ in range(2):
        buf = (_capi.Pose * 64)(); n = C.c_int(0); stats = (C.c_int64 * 2)()
        rc = ctx.lib.bl_astar_search(ctx.h, planner.distances_.h, C.byref(s), C.byref(gl), C.byref(planner.searchParams_), buf, 64, C.byref(n), stats)
        assert rc == _capi.BL_ERR_CAPACITY and n.value == 1
        assert (buf[0].x, buf[0].y) == (s.x, s.y)
        seen.append((stats[0], stats[1]))
    assert seen[0] == seen[1] and seen[0][0] > 1_000_000 and seen[0][1] - seen[0][0] >= (1 << 20) - 4


@pytest.mark.parametrize("name,case,counts", [("wide", 2, (526431, 763405)), ("convex", 2, None)])
def test_astar_long_cases(oracle, maps, gpu_ctx, name, case, counts):
    """The two long searches of the reference's fixtures (data/astar/wide_poses.txt, convex_poses.txt read with the token-stream
    semantics of astar_test.cpp:236): path, pop count and push count equal the oracle's."""
    m = maps["astar_" + name]
    row = helpers.load_astar_cases()[name][case]
    g = _grid_from_map(m, gpu_ctx)
    planner = bl.MotionPlanner(bl.MotionPlannerParams(0.1), ctx=gpu_ctx)
    planner.setMap(g)
    dist = oracle.set_distances(m["cells"], m["mpc"], helpers.CPM_DEFAULT, m["origin"])
    path, stats = bl.search_for_path(bl.make_pose(*row["start"], 0.0), bl.make_pose(*row["goal"], 0.0), planner.distances_,
                                     planner.searchParams_, return_stats=True)
    exp, est = oracle.search(oracle.pose(*row["start"], 0.0), oracle.pose(*row["goal"], 0.0), dist, m["mpc"],
                             helpers.CPM_DEFAULT, m["origin"], 0.1, 1.0)
    assert stats == est
    if counts is not None:
        assert stats == counts
    else:
        assert stats[0] > 1_000_000
    got = np.array([(p.utime, p.x, p.y, p.theta) for p in path], dtype=exp.dtype)
    assert got.tobytes() == exp.tobytes()


def test_astar_pipelined_results_longer_than_the_pinned_head(oracle, maps, monkeypatch):
    """Searches enqueued ahead of fetching (the replanner's pattern): a path longer than the head that travels with the result
    record must come back whole although later searches on the same ctx have run meanwhile -- every result slot keeps its own
    copy of the path.  The head is shrunk to 8 cells (BOTLAB_ASTAR_PATH_HEAD) so that the shipped maze's 60-76-cell paths take
    that route; under the reference's own fCost < 32767 rule a path can barely exceed the real head of 4096 cells."""
    monkeypatch.setenv("BOTLAB_ASTAR_PATH_HEAD", "8")
    ctx = bl.Context()
    try:
        m = maps["astar_maze"]
        g = bl.OccupancyGrid.from_cells(m["cells"], m["origin"], m["mpc"], cellsPerMeter=helpers.CPM_DEFAULT, ctx=ctx)
        planner = bl.MotionPlanner(bl.MotionPlannerParams(0.1), ctx=ctx)
        planner.setMap(g)
        dist = oracle.set_distances(m["cells"], m["mpc"], helpers.CPM_DEFAULT, m["origin"])
        rows = [helpers.load_astar_cases()["maze"][i] for i in (0, 2, 3, 0)]
        for row in rows:                                       # four searches in flight
            bl.search_for_path_begin(bl.make_pose(*row["goal"], 0.0), planner.distances_, planner.searchParams_, start=bl.make_pose(*row["start"], 0.0))
        for row in rows:
            path, stats = bl.search_for_path_end(planner.distances_, return_stats=True)
            exp, est = oracle.search(oracle.pose(*row["start"], 0.0), oracle.pose(*row["goal"], 0.0), dist, m["mpc"], helpers.CPM_DEFAULT,
                                     m["origin"], 0.1, 1.0)
            assert stats == est and len(path) == len(exp) > 9
            got = np.array([(p.utime, p.x, p.y, p.theta) for p in path], dtype=exp.dtype)
            assert got.tobytes() == exp.tobytes()
    finally:
        ctx.close()


def test_astar_on_slam_map_with_default_radius(oracle, maps, gpu_ctx):
    """MotionPlanner() default robotRadius 0.2 (motion_planner.hpp:31) on a SLAM-built map, several goals."""
    m = maps["obstacle_slam_10mx10m_5cm"]
    g = _grid_from_map(m, gpu_ctx)
    planner = bl.MotionPlanner(ctx=gpu_ctx)
    planner.setMap(g)
    dist = oracle.set_distances(m["cells"], m["mpc"], helpers.CPM_DEFAULT, m["origin"])
    free = np.argwhere(dist > 0.25)
    rng = np.random.default_rng(2)
    n_found = 0
    for _ in range(12):
        (sy, sx), (gy, gx) = free[rng.integers(len(free))], free[rng.integers(len(free))]
        sp = (m["origin"][0] + (sx + 0.5) * 0.05, m["origin"][1] + (sy + 0.5) * 0.05)
        gp = (m["origin"][0] + (gx + 0.5) * 0.05, m["origin"][1] + (gy + 0.5) * 0.05)
        path, stats = bl.search_for_path(bl.make_pose(*sp, 0.3), bl.make_pose(*gp, 0.0), planner.distances_,
                                         planner.searchParams_, return_stats=True)
        exp, est = oracle.search(oracle.pose(*sp, 0.3), oracle.pose(*gp, 0.0), dist, m["mpc"], helpers.CPM_DEFAULT,
                                 m["origin"], 0.2, 2.0)
        assert stats == est
        got = np.array([(p.utime, p.x, p.y, p.theta) for p in path], dtype=exp.dtype)
        assert got.tobytes() == exp.tobytes()
        n_found += len(path) > 1
    assert n_found >= 3


def test_step_pipeline_with_device_pose_equals_host_pose_path(oracle, maps, gpu_ctx):
    """The single-sync step pipeline (pose estimate stays on the device and feeds the map update and the A* start) gives
    the same map, pose and path as the call-by-call host-pose form."""
    m = maps["obstacle_slam_10mx10m_5cm"]
    truth = np.where(m["cells"] > 0, 127, -127).astype(np.int8)
    poses = synth.square_trajectory((-0.75, 0.2, 0.0), 8, step_len=0.02, turn=0.05, side=0.8)
    scans = [synth.raycast_scan(truth, m["origin"], 0.05, poses[k - 1], poses[k], 1_000_000 + k * 100_000) for k in range(1, 9)]
    goal = bl.make_pose(-0.35, 0.2, 0.0)
    out = []
    # "batched*": a lane collects several submissions and searches them in one launch (lag 3 < batch x lanes: some batches
    # are sent off partly filled by the fetch; lag 6: every batch fills, the last one is flushed by the drain)
    forms = {"async": (3, 1, 3), "fused": (3, 1, 3), "batched2": (3, 2, 3), "batched3": (2, 3, 6), "batched4": (1, 4, 2),
             "riding": (3, 1, 3), "riding_deep": (3, 1, 6), "riding_prefetch": (2, 2, 6), "fused_prefetch": (3, 1, 3),
             "riding_wrong_prefetch": (2, 2, 6), "batched8": (1, 8, 7)}
    for dev in (False, True, "async", "fused", "batched2", "batched3", "batched4", "batched8", "riding", "riding_deep", "riding_prefetch",
                "fused_prefetch", "riding_wrong_prefetch"):
        g = _grid_from_map(m, gpu_ctx)
        lanes, batch, lag = forms.get(dev, (0, 0, 0))
        aplanner = bl.AsyncPlanner(ctx=gpu_ctx, lanes=lanes, batch=batch) if dev in forms else None
        lagged = []
        pf = bl.ParticleFilter(2000, ctx=gpu_ctx)
        pf.initializeFilterAtPose(bl.make_pose(-0.75, 0.2, 0.0, utime=int(scans[0].times[0])), seed=5)
        pf.setNoiseSeed(9)
        mapper = bl.Mapping(5.0, 4, 1, ctx=gpu_ctx)
        planner = bl.MotionPlanner(ctx=gpu_ctx)
        rec = []
        for k, sc in enumerate(scans):
            odo = bl.make_pose(*poses[k + 1], utime=sc.utime)
            if dev in forms:
                # replanner on its own stream against snapshots; results fetched three steps late, in order
                riding = dev in ("riding", "riding_deep", "riding_prefetch", "riding_wrong_prefetch")
                if riding:                                           # the end of the filter update rides in the map kernel
                    pf.updateBegin(odo, sc, g, 1000 + k)
                else:
                    pf.updateFilter(odo, sc, g, rand_value=1000 + k, want_pose=False)
                # the next scan handed over early rides in the map kernel too; handing over another scan than the one that
                # comes next (here: the one after it, or the current one again) must change nothing
                if dev in ("riding_prefetch", "fused_prefetch") and k + 1 < len(scans):
                    gpu_ctx.scanPrefetch(scans[k + 1])
                if dev == "riding_wrong_prefetch":
                    gpu_ctx.scanPrefetch(scans[k + 2] if (k % 2 == 0 and k + 2 < len(scans)) else sc)
                if riding:
                    aplanner.submit_with_map_update_finishing(mapper, sc, pf, sc.utime, g, goal)
                if riding:
                    pass
                elif dev != "async":                                 # map update + snapshot in one call (bench.py's form)
                    aplanner.submit_with_map_update(mapper, sc, pf.poseDevicePtr(), sc.utime, g, goal)
                else:
                    mapper.updateMapDevicePose(sc, pf.poseDevicePtr(), sc.utime, g)
                    aplanner.submit(g, pf.poseDevicePtr(), goal)
                lagged.append(k)
                if len(lagged) > lag:
                    lagged.pop(0)
                    path = aplanner.fetch()
                    rec.append(((path[0].utime, path[0].x, path[0].y, path[0].theta), [(p.x, p.y, p.theta) for p in path]))
                if k == len(scans) - 1:
                    while lagged:
                        lagged.pop(0)
                        path = aplanner.fetch()
                        rec.append(((path[0].utime, path[0].x, path[0].y, path[0].theta), [(p.x, p.y, p.theta) for p in path]))
                continue
            elif dev:
                pf.updateFilter(odo, sc, g, rand_value=1000 + k, want_pose=False)
                mapper.updateMapDevicePose(sc, pf.poseDevicePtr(), sc.utime, g)
                planner.setMap(g)
                bl.search_for_path_begin(goal, planner.distances_, planner.searchParams_, start_dev=pf.poseDevicePtr())
                path = bl.search_for_path_end(planner.distances_)
                pose = path[0]
            else:
                pose = pf.updateFilter(odo, sc, g, rand_value=1000 + k)
                mapper.updateMap(sc, pose, g)
                planner.setMap(g)
                path = bl.search_for_path(pose, goal, planner.distances_, planner.searchParams_)
            rec.append(((pose.utime, pose.x, pose.y, pose.theta), [(p.x, p.y, p.theta) for p in path]))
        out.append((rec, g.cells().copy()))
    for o in out[1:]:
        assert o[0] == out[0][0]
        assert np.array_equal(o[1], out[0][1])
    assert max(len(r[1]) for r in out[0][0]) > 3


def test_step_pipeline_batched32_equals_synchronous_steps(maps, gpu_ctx):
    """A replanner lane that collects 32 or 64 submissions per launch (bench.py's presets for the large grids: up to 3 x 64 searches in flight):
    40 steps with the filter's end riding in the map kernel, the first launch a full batch of 32 and the rest flushed by the drain,
    give the poses, paths and map of the call-by-call device-pose form."""
    m = maps["obstacle_slam_10mx10m_5cm"]
    truth = np.where(m["cells"] > 0, 127, -127).astype(np.int8)
    n = 40
    poses = synth.square_trajectory((-0.75, 0.2, 0.0), n, step_len=0.02, turn=0.05, side=0.8)
    scans = [synth.raycast_scan(truth, m["origin"], 0.05, poses[k - 1], poses[k], 1_000_000 + k * 100_000) for k in range(1, n + 1)]
    goal = bl.make_pose(-0.35, 0.2, 0.0)
    out = []
    for form in ("sync", "batched32", "batched32x2", "batched64"):       # (64: one launch of 40 searches, the distance grids in two)
        g = _grid_from_map(m, gpu_ctx)
        pf = bl.ParticleFilter(2000, ctx=gpu_ctx)
        pf.initializeFilterAtPose(bl.make_pose(-0.75, 0.2, 0.0, utime=int(scans[0].times[0])), seed=5)
        pf.setNoiseSeed(9)
        mapper = bl.Mapping(5.0, 4, 1, ctx=gpu_ctx)
        planner = bl.MotionPlanner(ctx=gpu_ctx)
        aplanner = bl.AsyncPlanner(ctx=gpu_ctx, lanes=2 if form == "batched32x2" else 1, batch=64 if form == "batched64" else 32) if form != "sync" else None
        rec = []
        for k, sc in enumerate(scans):
            odo = bl.make_pose(*poses[k + 1], utime=sc.utime)
            if aplanner is None:
                pf.updateFilter(odo, sc, g, rand_value=1000 + k, want_pose=False)
                mapper.updateMapDevicePose(sc, pf.poseDevicePtr(), sc.utime, g)
                planner.setMap(g)
                bl.search_for_path_begin(goal, planner.distances_, planner.searchParams_, start_dev=pf.poseDevicePtr())
                path = bl.search_for_path_end(planner.distances_)
                rec.append(((path[0].utime, path[0].x, path[0].y, path[0].theta), [(p.x, p.y, p.theta) for p in path]))
            else:
                pf.updateBegin(odo, sc, g, 1000 + k)
                aplanner.submit_with_map_update_finishing(mapper, sc, pf, sc.utime, g, goal)
        if aplanner is not None:
            for _ in scans:                                       # every result, in submission order
                path = aplanner.fetch()
                rec.append(((path[0].utime, path[0].x, path[0].y, path[0].theta), [(p.x, p.y, p.theta) for p in path]))
        out.append((rec, g.cells().copy()))
    for o in out[1:]:
        assert o[0] == out[0][0]
        assert np.array_equal(o[1], out[0][1])
    assert max(len(r[1]) for r in out[0][0]) > 3


@pytest.mark.parametrize("N,rays", [(1000, 290), (100_000, 290), (300_000, 290), (3000, 1500)])
def test_filter_end_riding_in_map_kernel_equals_separate_calls(maps, gpu_ctx, N, rays):
    """bl_mapping_update_finishing_pf: updateFilter's end (weight prefix, unit total, pose estimate) computed inside the map
    kernel's launch gives the same particles, weights, estimate, resampling and map, bit for bit, as bl_pf_update_end
    followed by the map update -- for launch shapes with 64, 4 and 1 lanes per particle (tiles of 8, 128 and 512 particles,
    with and without the second region)."""
    m = maps["obstacle_slam_10mx10m_5cm"]
    truth = np.where(m["cells"] > 0, 127, -127).astype(np.int8)
    poses = synth.square_trajectory((-0.75, 0.2, 0.0), 5, step_len=0.02, turn=0.05, side=0.8)
    # (1500 rays: more rays than the map kernel has threads -- its per-ray walk form and the copy-loop snapshot -- and a ray
    # table that the particle filter reads in chunks)
    scans = [synth.raycast_scan(truth, m["origin"], 0.05, poses[k - 1], poses[k], 1_000_000 + k * 100_000, rays=rays) for k in range(1, 6)]
    res = []
    for riding in (False, True):
        g = _grid_from_map(m, gpu_ctx)
        pf = bl.ParticleFilter(N, ctx=gpu_ctx)
        pf.initializeFilterAtPose(bl.make_pose(-0.75, 0.2, 0.0, utime=int(scans[0].times[0])), seed=11)
        pf.setNoiseSeed(4)
        pf.debugEnable(True)
        mapper = bl.Mapping(5.0, 4, 1, ctx=gpu_ctx)
        rec = []
        for k, sc in enumerate(scans):
            odo = bl.make_pose(*poses[k + 1], utime=sc.utime) if k != 2 else bl.make_pose(*poses[k], utime=sc.utime)   # step 2: not moved
            if riding:
                pf.updateBegin(odo, sc, g, 77 + k)
                if k + 1 < len(scans):
                    gpu_ctx.scanPrefetch(scans[k + 1])               # the next scan rides along as well
                mapper.updateMapFinishingFilter(sc, pf, sc.utime, g)
            else:
                pf.updateFilter(odo, sc, g, rand_value=77 + k, want_pose=False)
                mapper.updateMapDevicePose(sc, pf.poseDevicePtr(), sc.utime, g)
            p = pf.poseEstimate()
            idx, like = pf.debugLast()
            rec.append(((p.utime, p.x, p.y, p.theta), idx.copy(), like.copy(), pf.particles().copy(), g.cells().copy()))
        res.append(rec)
    for a, b in zip(*res):
        assert a[0] == b[0]
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
        assert a[3].tobytes() == b[3].tobytes()
        assert np.array_equal(a[4], b[4])


def test_shard_engine_single_rank_with_nccl_collectives_matches_plain_filter(maps, gpu_ctx):
    """HipShardEngine + ShardedParticleFilter (the multi-GPU driver) with world_size 1 over the real nccl/RCCL backend:
    same particles as the plain ParticleFilter, collectives issued on the engine's own stream."""
    import os
    import torch
    import torch.distributed as dist
    from botlab_amd import sharded
    m = maps["obstacle_slam_10mx10m_5cm"]
    truth = np.where(m["cells"] > 0, 127, -127).astype(np.int8)
    poses = synth.square_trajectory((-0.75, 0.2, 0.0), 5, step_len=0.02, turn=0.05, side=0.8)
    scans = [synth.raycast_scan(truth, m["origin"], 0.05, poses[k - 1], poses[k], 1_000_000 + k * 100_000) for k in range(1, 6)]
    N = 4000
    os.environ["BOTLAB_FORCE_COLLECTIVES"] = "1"
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29533", rank=0, world_size=1,
                                device_id=torch.device("cuda", 0))
        created = True
    try:
        eng = sharded.HipShardEngine(N, 0, 1, 0)
        spf = sharded.ShardedParticleFilter(eng)
        assert spf.force_collectives
        g1 = _grid_from_map(m, eng.ctx)
        g2 = _grid_from_map(m, gpu_ctx)
        pf = bl.ParticleFilter(N, ctx=gpu_ctx)
        init = bl.make_pose(-0.75, 0.2, 0.0, utime=int(scans[0].times[0]))
        spf.initializeFilterAtPose(init, seed=3)
        pf.initializeFilterAtPose(init, seed=3)
        for k, sc in enumerate(scans):
            odo = bl.make_pose(*poses[k + 1], utime=sc.utime)
            a = spf.updateFilter(odo, sc, g1, 500 + k)
            b = pf.updateFilter(odo, sc, g2, rand_value=500 + k)
            assert (a.x, a.y, a.theta, a.utime) == (b.x, b.y, b.theta, b.utime)
        assert spf.particles().tobytes() == pf.particles().tobytes()
    finally:
        os.environ.pop("BOTLAB_FORCE_COLLECTIVES", None)
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize("side,form", [(1000, "riding"), (1000, "host_pose"), (1001, "host_pose")])
def test_large_grid_mirror_kept_current_by_map_updates(maps, gpu_ctx, side, form, monkeypatch):
    """Grids that do not fit LDS are scored from a zero-framed mirror of the grid.  The filter builds it once; every map update
    then stores the cells it changes into both images (the endpoints it changes are exactly the cells the next scan scores).
    A SLAM loop on such a grid must give the same particles, poses and map whether the mirror is reused, rebuilt before every
    update (BOTLAB_MCL_NO_MIRROR_REUSE) or not used at all (BOTLAB_MCL_NO_FRAMED: gathers from the grid itself) -- including
    across an upload that replaces the cells mid-run and, side 1001, rows that are not whole dwords."""
    N = 4000
    world = synth.tile_world(maps["astar_maze"]["cells"], side)
    half = side * 0.05 / 2.0
    origin, mpc, cpm = (np.float32(-half), np.float32(-half)), np.float32(0.05), helpers.CPM_DEFAULT
    start = (3.3, -7.1, 0.4)
    truth = np.where(world > 0, 127, -127).astype(np.int8)
    first = np.where(world > 0, 30, -20).astype(np.int8)
    first[:, : side // 2] = 0                                           # half the map still unknown: the updates matter
    poses = synth.square_trajectory(start, 10, step_len=0.05, turn=0.1, side=0.2)
    scans = [synth.raycast_scan(truth, origin, 0.05, poses[k - 1], poses[k], 1_000_000 + k * 100_000) for k in range(1, len(poses))]
    out = []
    for env in ({}, {"BOTLAB_MCL_NO_MIRROR_REUSE": "1"}, {"BOTLAB_MCL_NO_FRAMED": "1"}):
        with monkeypatch.context() as mp:
            for k_, v_ in env.items():
                mp.setenv(k_, v_)                                       # read when the filter is created
            g = bl.OccupancyGrid.from_cells(first, origin, mpc, cellsPerMeter=cpm, ctx=gpu_ctx)
            pf = bl.ParticleFilter(N, ctx=gpu_ctx)
            pf.initializeFilterAtPose(bl.make_pose(*start, utime=int(scans[0].times[0])), seed=5)
            pf.setNoiseSeed(9)
            mapper = bl.Mapping(5.0, 4, 1, ctx=gpu_ctx)
            aplanner = bl.AsyncPlanner(ctx=gpu_ctx, lanes=1, batch=1) if form == "riding" else None
            rec = []
            for k, sc in enumerate(scans):
                odo = bl.make_pose(*poses[k + 1], utime=sc.utime)
                if k == 5:                                              # the cells are replaced behind the mirror's back
                    g.upload(np.where(world > 0, 10, -5).astype(np.int8))
                if form == "riding":
                    pf.updateBegin(odo, sc, g, 1000 + k)
                    aplanner.submit_with_map_update_finishing(mapper, sc, pf, sc.utime, g, bl.make_pose(start[0] + 0.5, start[1], 0.0))
                    aplanner.fetch()
                    pose = pf.poseEstimate()
                else:
                    pose = pf.updateFilter(odo, sc, g, rand_value=1000 + k)
                    mapper.updateMap(sc, pose, g)
                parts = pf.particles()
                rec.append((None if pose is None else (pose.x, pose.y, pose.theta), parts["x"].copy(), parts["y"].copy(),
                            parts["theta"].copy(), parts["weight"].copy()))
            out.append((rec, g.cells().copy()))
    for o in out[1:]:
        assert np.array_equal(o[1], out[0][1])
        for a_, b_ in zip(o[0], out[0][0]):
            assert a_[0] == b_[0]
            for u_, v_ in zip(a_[1:], b_[1:]):
                assert np.array_equal(u_, v_)
    assert (out[0][1] != first).sum() > 1000                           # the map really changed


@pytest.mark.parametrize("form", ["riding", "host_pose"])
def test_whole_grid_staging_forms_agree(maps, gpu_ctx, form, monkeypatch):
    """A 200 x 200 grid is staged whole in LDS by every workgroup of k_mcl_main: from the grid's zero-framed copy in 16-byte pieces
    (round 6, the default: the copy is built once and kept current by the map updates), row by row from the grid itself by LDS-DMA
    (BOTLAB_MCL_NO_STAGE_X4) or through registers (BOTLAB_MCL_NO_STAGE_DMA).  A SLAM loop gives the same particles, poses and map in
    all three -- across map updates that change the cells the next scan scores and an upload that replaces them behind the copy's back."""
    N = 6000
    m = maps["obstacle_slam_10mx10m_5cm"]
    truth = np.where(m["cells"] > 0, 127, -127).astype(np.int8)
    first = m["cells"].copy()
    first[:, 100:] = 0                                                  # half the map still unknown: the updates matter
    poses = synth.square_trajectory((-0.75, 0.2, 0.0), 10, step_len=0.04, turn=0.1, side=0.6)
    scans = [synth.raycast_scan(truth, m["origin"], 0.05, poses[k - 1], poses[k], 1_000_000 + k * 100_000) for k in range(1, len(poses))]
    out = []
    for env in ({}, {"BOTLAB_MCL_NO_STAGE_X4": "1"}, {"BOTLAB_MCL_NO_STAGE_DMA": "1"}):
        with monkeypatch.context() as mp:
            for k_, v_ in env.items():
                mp.setenv(k_, v_)                                       # read when the filter is created
            g = bl.OccupancyGrid.from_cells(first, m["origin"], m["mpc"], cellsPerMeter=helpers.CPM_DEFAULT, ctx=gpu_ctx)
            pf = bl.ParticleFilter(N, ctx=gpu_ctx)
            pf.initializeFilterAtPose(bl.make_pose(-0.75, 0.2, 0.0, utime=int(scans[0].times[0])), seed=5)
            pf.setNoiseSeed(9)
            mapper = bl.Mapping(5.0, 4, 1, ctx=gpu_ctx)
            rec = []
            for k, sc in enumerate(scans):
                odo = bl.make_pose(*poses[k + 1], utime=sc.utime)
                if k == 5:                                              # the cells are replaced behind the copy's back
                    g.upload(np.where(m["cells"] > 0, 10, -5).astype(np.int8))
                if form == "riding":
                    pf.updateBegin(odo, sc, g, 1000 + k)
                    mapper.updateMapFinishingFilter(sc, pf, sc.utime, g)
                    pose = pf.poseEstimate()
                else:
                    pose = pf.updateFilter(odo, sc, g, rand_value=1000 + k)
                    mapper.updateMap(sc, pose, g)
                parts = pf.particles()
                rec.append(((pose.x, pose.y, pose.theta), parts["x"].copy(), parts["y"].copy(), parts["theta"].copy(), parts["weight"].copy()))
            out.append((rec, g.cells().copy()))
            pf.close(); g.close()
    for o in out[1:]:
        assert np.array_equal(o[1], out[0][1])
        for a_, b_ in zip(o[0], out[0][0]):
            assert a_[0] == b_[0]
            for u_, v_ in zip(a_[1:], b_[1:]):
                assert np.array_equal(u_, v_)
    assert (out[0][1] != first).sum() > 500                            # the map really changed


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_astar_random_maps_equal_oracle(oracle, gpu_ctx, seed):
    """search_for_path on seeded random worlds (blocks and walls on a 320 x 240 grid, robot radius 0.1 and 0.2): poses, pops and pushes
    equal the oracle's -- open lists from a handful to hundreds of thousands of entries.  The start / goal pairs are those of
    tests/golden/astar_random_cases.json (tests/tools/make_astar_random_cases.py: pairs whose search the oracle finishes within
    2e6 pops -- with the reference's cost function and no open-list de-duplication many nearby goals cost 1e7 .. 1e9)."""
    import json
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
    import make_astar_random_cases as gen
    cases = [c for c in json.load(open(os.path.join(helpers.GOLDEN, "astar_random_cases.json"))) if c["seed"] == seed]
    assert len(cases) >= 6
    cells, _ = gen.world(seed)
    mpc = np.float32(0.05)
    origin = (np.float32(-8.0), np.float32(-6.0))
    cpm = helpers.CPM_DEFAULT
    g = bl.OccupancyGrid.from_cells(cells, origin, mpc, cellsPerMeter=cpm, ctx=gpu_ctx)
    dist = oracle.set_distances(cells, mpc, cpm, origin)
    found = 0
    for c in cases:
        radius = c["radius"]
        planner = bl.MotionPlanner(bl.MotionPlannerParams(radius), ctx=gpu_ctx)
        planner.setMap(g)
        assert np.array_equal(planner.distances_.cells().view(np.uint32), dist.view(np.uint32))
        sp, gp = c["start"], c["goal"]
        exp, est = oracle.search(oracle.pose(*sp, 0.3), oracle.pose(*gp, 0.0), dist, mpc, cpm, origin, radius, 10.0 * radius, cap=1 << 16)
        assert tuple(est) == (c["pops"], c["pushes"]) and len(exp) == c["poses"]          # the fixture is what the oracle says today
        path, stats = bl.search_for_path(bl.make_pose(*sp, 0.3), bl.make_pose(*gp, 0.0), planner.distances_, planner.searchParams_,
                                         return_stats=True)
        assert tuple(stats) == tuple(est), (seed, c, stats, est)
        got = np.array([(p.utime, p.x, p.y, p.theta) for p in path], dtype=exp.dtype)
        assert got.tobytes() == exp.tobytes(), (seed, c)
        found += 1 if len(path) > 1 else 0
    assert found >= 4


def _fixtures_in_child(queue_path):
    """the fixture searches with the C++ forms of k_astar2 only (the switch is read once per process)"""
    import json
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import helpers as h
    import oracle_lib
    import botlab_amd as b
    orc = oracle_lib.load_oracle()
    maps = h.load_reference_maps()
    ctx = b.default_context()
    out = {}
    for name in ("narrow", "wide", "convex", "maze"):
        m = maps["astar_" + name]
        g = b.OccupancyGrid.from_cells(m["cells"], m["origin"], m["mpc"], cellsPerMeter=h.CPM_DEFAULT, ctx=ctx)
        planner = b.MotionPlanner(b.MotionPlannerParams(0.1), ctx=ctx)
        planner.setMap(g)
        dist = orc.set_distances(m["cells"], m["mpc"], h.CPM_DEFAULT, m["origin"])
        for i, row in enumerate(h.load_astar_cases()[name]):
            if (name, i) == ("narrow", 2):
                continue
            path, stats = b.search_for_path(b.make_pose(*row["start"], 0.0), b.make_pose(*row["goal"], 0.0), planner.distances_,
                                            planner.searchParams_, return_stats=True)
            exp, est = orc.search(orc.pose(*row["start"], 0.0), orc.pose(*row["goal"], 0.0), dist, m["mpc"], h.CPM_DEFAULT, m["origin"], 0.1, 1.0)
            got = np.array([(p.utime, p.x, p.y, p.theta) for p in path], dtype=exp.dtype)
            out[f"{name}{i}"] = bool(tuple(stats) == tuple(est) and got.tobytes() == exp.tobytes())
    json.dump(out, open(queue_path, "w"))


@pytest.mark.parametrize("env", ["BOTLAB_ASTAR_NO_TURBO", "BOTLAB_ASTAR_V1", "BOTLAB_ASTAR_DUO=0", "BOTLAB_ASTAR_AHEAD=1", "BOTLAB_ASTAR_AHEAD=0",
                                 "BOTLAB_ASTAR_DEEP_AHEAD=0", "BOTLAB_ASTAR_SMALL_LDS=3", "BOTLAB_ASTAR_SMALL_LDS=3,BOTLAB_ASTAR_DEEP_AHEAD=0"])
def test_astar_fixtures_with_the_other_forms_of_the_search(tmp_path, env):
    """the same fixtures through k_astar2's C++ forms (no straight-line loop), through round 4's k_astar (8-byte entries) -- the
    forms a search falls back to for lists of 0-1 entries, cost tables beyond LDS or below the 16-bit key range -- and through the
    one-wave straight-line loop (what the replanner's units run), through round 5's two-wave loop (bl_astar2_duo.h, BOTLAB_ASTAR_AHEAD=0)
    and through the two-wave form of the loop single searches take by default (bl_astar2_ahead.h: the next pop's walk beside the
    pushes, expansions made ahead; three waves by default); with the one-wave loop beyond LDS in place of the three-wave one
    (BOTLAB_ASTAR_DEEP_AHEAD=0), and with the 40 KB footprint on three waves (BOTLAB_ASTAR_SMALL_LDS=3: every maze search of more than
    ~5 000 pops then crosses into the deep regime and back, and the three-wave loops run with the small footprint's tree shape)"""
    import json
    import subprocess
    out = str(tmp_path / "res.json")
    code = ("import sys, os; sys.path.insert(0, %r); sys.path.insert(0, %r); import test_gpu_parity as t; t._fixtures_in_child(%r)"
            % (os.path.dirname(os.path.abspath(__file__)), os.path.dirname(os.path.dirname(os.path.abspath(__file__))), out))
    e = dict(os.environ)
    for item in env.split(","):
        k, _, v = item.partition("=")
        e[k] = v or "1"
    subprocess.check_call([sys.executable, "-c", code], env=e)
    res = json.load(open(out))
    assert len(res) >= 16 and all(res.values()), {k: v for k, v in res.items() if not v}


def test_astar_cost_table_below_16_bit_keys_takes_k_astar(gpu_ctx, oracle, maps):
    """The fallback reached by the condition itself, not by BOTLAB_ASTAR_V1: a cost table whose obstacle cost can take an fCost to
    -32768 or below (astar.cpp:181-186: (int)pow(maxDist - d * 2000, exponent) -- maxDistanceWithCost 20 m gives -39 980 at the table's
    far end; exponent 3 gives -6e7 two cells from a wall) cannot live in k_astar2's 16-bit keys: the host must launch k_astar, and the
    result must still be the oracle's (poses, pops, pushes).  The reference's own parameters take k_astar2 -- checked beside it."""
    import botlab_amd as b
    import helpers as h
    lib = gpu_ctx.lib
    m = maps["astar_maze"]
    g = b.OccupancyGrid.from_cells(m["cells"], m["origin"], m["mpc"], cellsPerMeter=h.CPM_DEFAULT, ctx=gpu_ctx)
    planner = b.MotionPlanner(b.MotionPlannerParams(0.1), ctx=gpu_ctx)
    planner.setMap(g)
    dist = oracle.set_distances(m["cells"], m["mpc"], h.CPM_DEFAULT, m["origin"])
    rows = h.load_astar_cases()["maze"]
    for (mind, maxd, expo), want_kernel in (((0.1, 1.0, 1.0), 2), ((0.1, 20.0, 1.0), 1), ((0.1, 0.6, 3.0), 1), ((0.1, 1.0, 3.0), 1)):   # (the last one: D11 from 0.65 m on)
        sp = b._capi.SearchParams(mind, maxd, expo)
        for i in (0, 2):
            row = rows[i]
            path, stats = b.search_for_path(b.make_pose(*row["start"], 0.0), b.make_pose(*row["goal"], 0.0), planner.distances_, sp, return_stats=True)
            assert lib.bl_astar_debug_last_kernel(gpu_ctx.h) == want_kernel, (mind, maxd, expo)
            exp, est = oracle.search(oracle.pose(*row["start"], 0.0), oracle.pose(*row["goal"], 0.0), dist, m["mpc"], h.CPM_DEFAULT, m["origin"],
                                     mind, maxd, exponent=expo)
            got = np.array([(p.utime, p.x, p.y, p.theta) for p in path], dtype=exp.dtype)
            assert tuple(stats) == tuple(est), ((mind, maxd, expo), i, stats, est)
            assert got.tobytes() == exp.tobytes(), ((mind, maxd, expo), i)
