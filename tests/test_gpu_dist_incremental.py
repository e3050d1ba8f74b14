"""ObstacleDistanceGrid::setDistances (src/planning/obstacle_distance_grid.cpp:73-91) on a map that Mapping::updateMap keeps
changing: the library transforms only the window the updates since its last transform can influence (bl_planning.hip,
"incremental") -- the result must be the full transform's, bit for bit, at every step.  Checked over SLAM runs at 2000 x 2000
and 4096 x 4096 against a second distance grid that is forced to transform the whole map every time, and (at 2000 x 2000, every
40th step) against the CPU oracle; with maps that lose sources (unknown cells becoming free: distances GROW, and so must the
bound the window is built from), through the replanner's snapshots, and across the events that end a lineage."""
import ctypes as C

import numpy as np
import pytest

import helpers
import botlab_amd as bl
from botlab_amd import synth

pytestmark = pytest.mark.gpu


def _l1_of(d):
    return d.cells().view(np.uint32)


def _world(maps, size, explored_half):
    world = synth.tile_world(maps["astar_maze"]["cells"], size)
    half = size * 0.05 / 2.0
    origin = (np.float32(-half), np.float32(-half))
    yy, xx = np.mgrid[0:size, 0:size]
    known = (np.abs(xx - size // 2) < explored_half) & (np.abs(yy - size // 2) < explored_half)
    # explored part: walls 127, free -100; the rest unknown (0: a SOURCE of the transform, obstacle_distance_grid.cpp:125-128)
    cells = np.where(known, np.where(world > 0, 127, -100), 0).astype(np.int8)
    return world, cells, origin


@pytest.mark.parametrize("size,steps,explored", [(2000, 200, 60), (4096, 200, 2000)])
def test_incremental_transform_equals_full_over_a_slam_run(oracle, maps, gpu_ctx, monkeypatch, size, steps, explored):
    world, cells, origin = _world(maps, size, explored)
    g = bl.OccupancyGrid.from_cells(cells, origin, np.float32(0.05), cellsPerMeter=helpers.CPM_DEFAULT, ctx=gpu_ctx)
    mapper = bl.Mapping(5.0, 4, 1, ctx=gpu_ctx)
    inc = bl.ObstacleDistanceGrid(ctx=gpu_ctx)
    poses = synth.square_trajectory((0.3, 0.3, 0.0), steps, step_len=0.05, turn=0.1, side=1.5)
    scans = synth.raycast_scans_gpu(world, origin, 0.05, poses, 1_000_000, 100_000, gpu_ctx)
    inc.setDistances(g)                                       # the first transform is a full one
    for k in range(1, len(poses)):
        p = poses[k]
        mapper.updateMap(scans[k - 1], bl.make_pose(p[0], p[1], p[2], utime=scans[k - 1].utime), g)
        if k % 7 == 3:
            continue                                          # several updates between two transforms
        inc.setDistances(g)
        if k % 5 == 0 or k < 12:
            full = bl.ObstacleDistanceGrid(ctx=gpu_ctx)       # a fresh grid knows no lineage: whole-grid kernels
            full.setDistances(g)
            assert np.array_equal(_l1_of(inc), _l1_of(full)), f"step {k}"
            full.close()
        if size == 2000 and k % 40 == 0:
            exp = oracle.set_distances(g.cells(), g.mpc, g.cpm, g.origin)
            assert np.array_equal(_l1_of(inc), exp.view(np.uint32)), f"oracle, step {k}"
    st = inc.stats()
    assert st["incremental"] >= steps // 2, st                # the run really took the incremental path
    assert st["window"] >= steps // 2, st                     # ... and on the device it ended as a window, not as the whole grid
    inc.close(); g.close()


def test_incremental_transform_when_sources_disappear_and_lineages_end(oracle, maps, gpu_ctx):
    """A robot uncovering unknown space: the cells it sees turn from 0 (a source) to free, distances in the hall grow step by
    step, and the bound D must follow.  Then the events that end a lineage -- upload, reset, a raw device pointer -- each of
    which must send the next transform over the whole grid."""
    size = 1536
    cells = np.zeros((size, size), np.int8)                   # all unknown
    cells[0, :] = cells[-1, :] = 127
    cells[:, 0] = cells[:, -1] = 127
    world = np.full((size, size), -127, np.int8)              # truth: one empty hall
    world[0, :] = world[-1, :] = 127
    world[:, 0] = world[:, -1] = 127
    half = size * 0.05 / 2.0
    origin = (np.float32(-half), np.float32(-half))
    g = bl.OccupancyGrid.from_cells(cells, origin, np.float32(0.05), cellsPerMeter=helpers.CPM_DEFAULT, ctx=gpu_ctx)
    mapper = bl.Mapping(5.0, 4, 6, ctx=gpu_ctx)               # miss odds 6: a crossed cell is free at once
    inc = bl.ObstacleDistanceGrid(ctx=gpu_ctx)
    poses = synth.square_trajectory((0.0, 0.0, 0.0), 60, step_len=0.25, turn=0.4, side=6.0)
    scans = synth.raycast_scans_gpu(world, origin, 0.05, poses, 1_000_000, 100_000, gpu_ctx, max_range=4.5)
    inc.setDistances(g)
    grew = 0
    last_max = 0.0
    for k in range(1, len(poses)):
        p = poses[k]
        mapper.updateMap(scans[k - 1], bl.make_pose(p[0], p[1], p[2], utime=scans[k - 1].utime), g)
        inc.setDistances(g)
        if k % 3 == 0:
            exp = oracle.set_distances(g.cells(), g.mpc, g.cpm, g.origin)
            assert np.array_equal(_l1_of(inc), exp.view(np.uint32)), f"step {k}"
            grew += exp.max() > last_max
            last_max = float(exp.max())
    assert grew >= 5 and last_max > 2.0                       # the free space, and with it the largest distance, kept growing
    assert inc.stats()["incremental"] >= 40
    # ---- lineage ends
    c2 = g.cells().copy()
    c2[700:720, 700:900] = 127
    g.upload(c2)
    inc.setDistances(g)
    assert np.array_equal(_l1_of(inc), oracle.set_distances(c2, g.mpc, g.cpm, g.origin).view(np.uint32))
    g.reset()
    inc.setDistances(g)
    assert np.array_equal(_l1_of(inc), oracle.set_distances(np.zeros_like(c2), g.mpc, g.cpm, g.origin).view(np.uint32))
    n_full = inc.stats()["full"]
    inc.setDistances(g)                                       # nothing happened in between: no launch at all
    assert inc.stats()["unchanged"] >= 1 and inc.stats()["full"] == n_full
    inc.close(); g.close()


def test_replanner_snapshots_carry_the_lineage(maps, gpu_ctx):
    """The replanner's units transform SNAPSHOTS of the map, each unit every (lanes x batch)-th one: a snapshot counts as the
    version of the map it was copied from, so the units take the incremental path too -- and the paths they return are the ones
    a planner that transforms the whole grid every time returns (BOTLAB_DIST_NO_INCREMENTAL is read once per process, so the
    comparison is against the synchronous search on a fresh distance grid)."""
    size = 2000
    world, cells, origin = _world(maps, size, 2000)
    g = bl.OccupancyGrid.from_cells(cells, origin, np.float32(0.05), cellsPerMeter=helpers.CPM_DEFAULT, ctx=gpu_ctx)
    mapper = bl.Mapping(5.0, 4, 1, ctx=gpu_ctx)
    pf = bl.ParticleFilter(2000, ctx=gpu_ctx)
    poses = synth.square_trajectory((0.3, 0.3, 0.0), 40, step_len=0.05, turn=0.1, side=1.0)
    scans = synth.raycast_scans_gpu(world, origin, 0.05, poses, 1_000_000, 100_000, gpu_ctx)
    pf.initializeFilterAtPose(bl.make_pose(*poses[0], utime=int(scans[0].times[0])), seed=3)
    ap = bl.AsyncPlanner(ctx=gpu_ctx, lanes=2, batch=3)
    goal = bl.make_pose(0.3 + 1.2, 0.3 + 0.4, 0.0)
    got, maps_at = [], []
    for k in range(1, len(poses)):
        sc = scans[k - 1]
        pf.updateFilter(bl.make_pose(*poses[k], utime=sc.utime), sc, g, rand_value=100 + k, want_pose=False)
        ap.submit_with_map_update(mapper, sc, pf.poseDevicePtr(), sc.utime, g, goal)
        maps_at.append((g.cells().copy(), pf.poseEstimate()))
        if len(maps_at) - len(got) > 8:
            got.append(ap.fetch(return_stats=True))
    while len(got) < len(maps_at):
        got.append(ap.fetch(return_stats=True))
    planner = bl.MotionPlanner(ctx=gpu_ctx)
    for k in (0, 7, 19, len(maps_at) - 1):
        c, pose = maps_at[k]
        g2 = bl.OccupancyGrid.from_cells(c, origin, np.float32(0.05), cellsPerMeter=helpers.CPM_DEFAULT, ctx=gpu_ctx)
        planner.setMap(g2)
        exp, est = bl.search_for_path(pose, goal, planner.distances_, planner.searchParams_, return_stats=True)
        path, st = got[k]
        assert st == est, k
        assert [(q.x, q.y, q.theta) for q in path] == [(q.x, q.y, q.theta) for q in exp], k
        g2.close()
    ap.close(); pf.close(); g.close()


@pytest.mark.parametrize("lattice", [128, 0])
def test_incremental_window_is_dilated_by_the_true_bound(oracle, gpu_ctx, lattice):
    """A fully explored free hall (every cell known free, so no source hides behind the scan box's edge), wider than the scan box;
    a hit turns a free cell into a source inside it.  Cells far OUTSIDE the box are then nearer to the new source than to their
    old one: the window must be the box dilated by the bound D of the whole grid (+ 1), not by whatever the last window saw.
    With a pillar lattice every 128 cells D is small enough for a window; without it the window exceeds DINC_MAX and the device
    falls back to the whole grid -- both must equal the oracle.  (The whole-grid kernels leave no bound behind: the first
    incremental transform after one has to form it.)"""
    size = 1536
    cells = np.full((size, size), -1, np.int8)                # known free, one hit away from being a source
    cells[0, :] = cells[-1, :] = 127
    cells[:, 0] = cells[:, -1] = 127
    if lattice:
        for y in range(lattice, size - 2, lattice):
            for x in range(lattice, size - 2, lattice):
                cells[y:y + 2, x:x + 2] = 127
    world = np.where(cells > 0, 127, -127).astype(np.int8)
    c = size // 2
    world[c + 20:c + 26, c + 30:c + 36] = 127                  # an obstacle the map does not hold yet, 1.5 m from the start
    half = size * 0.05 / 2.0
    origin = (np.float32(-half), np.float32(-half))
    g = bl.OccupancyGrid.from_cells(cells, origin, np.float32(0.05), cellsPerMeter=helpers.CPM_DEFAULT, ctx=gpu_ctx)
    mapper = bl.Mapping(5.0, 4, 1, ctx=gpu_ctx)
    inc = bl.ObstacleDistanceGrid(ctx=gpu_ctx)
    poses = synth.square_trajectory((0.2, 0.2, 0.0), 12, step_len=0.05, turn=0.1, side=1.0)
    scans = synth.raycast_scans_gpu(world, origin, 0.05, poses, 1_000_000, 100_000, gpu_ctx, max_range=4.5)
    inc.setDistances(g)                                       # whole grid
    before = oracle.set_distances(g.cells(), g.mpc, g.cpm, g.origin)
    assert np.array_equal(_l1_of(inc), before.view(np.uint32))
    changed_far = 0
    for k in range(1, len(poses)):
        p = poses[k]
        mapper.updateMap(scans[k - 1], bl.make_pose(p[0], p[1], p[2], utime=scans[k - 1].utime), g)
        inc.setDistances(g)
        exp = oracle.set_distances(g.cells(), g.mpc, g.cpm, g.origin)
        assert np.array_equal(_l1_of(inc), exp.view(np.uint32)), f"step {k}"
        formed, D = inc.bound()
        finite = exp[exp >= 0]
        # (after a fall-back to the whole grid the host stays with the whole-grid kernels for a while: no bound then)
        assert formed or not lattice
        if formed:
            assert D >= int(round(float(finite.max()) / 0.1)) - 1, (k, D, float(finite.max()))
        # cells more than 5 m + 2 cells from the robot (outside any scan box dilated by 1) whose distance changed
        yy, xx = np.nonzero(exp != before)
        if len(yy):
            far = np.maximum(np.abs(xx - c), np.abs(yy - c)) > 110 + 20
            changed_far += int(far.sum())
        before = exp
    assert changed_far > 0                                    # the case really reaches beyond the box
    st = inc.stats()
    if lattice:
        assert st["incremental"] == len(poses) - 1 and st["window"] >= len(poses) - 3, st
    else:
        assert st["incremental"] >= 1 and st["fallback"] >= 1, st
    inc.close(); g.close()
