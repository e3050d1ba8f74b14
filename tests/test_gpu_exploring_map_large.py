"""Exploration::executeExploringMap (src/planning/exploration.cpp:277-369) composed at BASELINE.json configs[4]'s size: a 4096 x 4096
grid whose centre tile holds the shipped obstacle_slam arena, uncovered strip by strip while the robot drives along the planned
paths; everything else is unknown.  Next state, status, frontier lists (byte for byte), path and target equal the CPU oracle's
composition at every step -- through the synchronous ExploringMap (host.py) and through the asynchronous explorer lanes
(bl_explorer: snapshot + device pose + side streams), whose steps are submitted two at a time.

The explored region is an arena of the reference's own size on purpose: with the reference's cost function (negative obstacle
costs, no open-list de-duplication: astar.cpp:181-186, 118-127) every pop of a duplicate pushes its open neighbours again, and
plan_path_to_frontier costs ~100 pops per explored free cell -- 1e6 pops on this 3 300-cell arena, 1.2e7 and an overflowing open
list on a 7.5 m explored disc (measured), where the reference itself runs out of memory.  Floods of millions of cells are covered by
tests/test_gpu_frontiers.py."""
import numpy as np
import pytest

import helpers
import botlab_amd as bl

pytestmark = pytest.mark.gpu


def _same_frontiers(got, exp):
    assert len(got) == len(exp)
    for a, b in zip(got, exp):
        assert a.tobytes() == b.tobytes()


def _dev_pose(pose):
    import torch
    t = torch.from_numpy(np.frombuffer(bytes(pose), np.uint8).copy()).cuda()
    torch.cuda.synchronize()
    return t


@pytest.mark.parametrize("size", [4096])
def test_exploring_map_large_grid_matches_oracle(oracle, maps, gpu_ctx, size):
    import oracle_lib
    cpm = helpers.CPM_DEFAULT
    mpc = np.float32(0.05)
    m = maps["obstacle_slam_10mx10m_5cm"]
    half = size * 0.05 / 2.0
    origin = (np.float32(-half), np.float32(-half))
    ty = tx = (size // 2 // 200) * 200                        # the arena's tile: cells tx..tx+199
    # the arena's own frame starts at (-5, -5): a pose (x, y) there is this far from the big grid's origin
    off = (float(origin[0]) + tx * 0.05 + 5.0, float(origin[1]) + ty * 0.05 + 5.0)

    def world(cut):
        cells = np.zeros((size, size), np.int8)
        a = m["cells"].copy()
        a[:, cut:] = 0
        cells[ty:ty + 200, tx:tx + 200] = a
        return cells

    cuts = (110, 110, 125, 140, 200)
    worlds = [world(c) for c in cuts]
    pl = bl.MotionPlanner(bl.MotionPlannerParams(0.2), ctx=gpu_ctx)
    ex = bl.ExploringMap(pl)
    oex = oracle_lib.OracleExploringMap(oracle, 0.2)
    robot = (-0.75 + off[0], 0.2 + off[1], 0.4)
    robots, expected = [], []
    for k, cells in enumerate(worlds):
        grid = bl.OccupancyGrid.from_cells(cells, origin, mpc, cellsPerMeter=cpm, ctx=gpu_ctx)
        nxt = ex.execute(grid, bl.make_pose(*robot))
        enxt, efr = oex.execute(cells, mpc, cpm, origin, oracle.pose(*robot))
        _same_frontiers(ex.frontiers_.cells(), efr)
        assert (nxt, ex.status) == (enxt, oex.status), (k, nxt, ex.status, enxt, oex.status)
        assert len(ex.currentPath_) == len(oex.path)
        for a, b in zip(ex.currentPath_, oex.path):
            assert (a.utime, a.x, a.y, a.theta) == (int(b["utime"]), b["x"], b["y"], b["theta"])
        assert (np.float32(ex.currentTarget_.x), np.float32(ex.currentTarget_.y)) == oex.target
        robots.append(robot)
        expected.append((enxt, oex.status, efr, [(int(b["utime"]), b["x"], b["y"], b["theta"]) for b in oex.path], oex.target))
        if len(ex.currentPath_) > 1:                          # drive most of the way along the path (inside / outside the 0.5 m rule)
            p = ex.currentPath_[(3 * len(ex.currentPath_)) // 4] if cuts[k] != 125 else ex.currentPath_[1]
            robot = (float(p.x), float(p.y), float(p.theta))
        grid.close()
    assert expected[0][0] == bl.host.STATE_EXPLORING_MAP and expected[-1][0] == bl.host.STATE_RETURNING_HOME
    assert len(expected[0][2]) >= 1 and len(expected[0][3]) > 1

    # ---- the same steps through the explorer lanes, two submissions in flight
    axp = bl.AsyncExplorer(ctx=gpu_ctx, lanes=2, robotRadius=0.2)
    grids = [bl.OccupancyGrid.from_cells(c, origin, mpc, cellsPerMeter=cpm, ctx=gpu_ctx) for c in worlds]
    poses = [_dev_pose(bl.make_pose(*r)) for r in robots]
    got = []
    axp.submit(grids[0], poses[0].data_ptr())
    for k in range(1, len(grids) + 1):
        if k < len(grids):
            axp.submit(grids[k], poses[k].data_ptr())
        res, path = axp.fetch()
        got.append((res, path, axp.frontiers().cells()))
    assert axp.pending() == 0
    for k, (res, path, fr) in enumerate(got):
        enxt, est, efr, epath, etarget = expected[k]
        _same_frontiers(fr, efr)
        assert (res.next_state, res.status, res.num_frontiers) == (enxt, est, len(efr)), k
        assert [(p.utime, p.x, p.y, p.theta) for p in path] == epath, (k, robots[k], (res.pose.x, res.pose.y, res.pose.theta), res.pops, res.searches, [(p.x, p.y, p.theta) for p in path][10:16], epath[10:16])
        assert (np.float32(res.target.x), np.float32(res.target.y)) == etarget, k
        assert res.frontiers_ms > 0.0 and res.bfs_cells > 1000
    assert sum(g[0].planned for g in got) >= 2 and sum(g[0].pops for g in got) > 100_000
    axp.close()

    # ---- once more with submit and fetch on DIFFERENT threads (the reference's exploration process beside its SLAM process): the
    # submitting thread waits for a free lane, the fetching thread takes the steps back; same maps in the same order, same results
    import threading
    import time
    axp = bl.AsyncExplorer(ctx=gpu_ctx, lanes=2, robotRadius=0.2)
    got2, errors = [], []

    def fetcher():
        try:
            for _ in range(len(grids)):
                while axp.pending() == 0:
                    time.sleep(0.0005)
                res, path = axp.fetch()
                got2.append((res.next_state, res.status, res.num_frontiers, [(p.utime, p.x, p.y, p.theta) for p in path],
                             (np.float32(res.target.x), np.float32(res.target.y))))
        except Exception as e:                          # noqa: BLE001 -- reported by the main thread
            errors.append(e)

    th = threading.Thread(target=fetcher)
    th.start()
    for k in range(len(grids)):
        while axp.pending() >= 2:
            time.sleep(0.0005)
        axp.submit(grids[k], poses[k].data_ptr())
    th.join(timeout=300)
    assert not th.is_alive() and not errors, errors
    for k, g2 in enumerate(got2):
        enxt, est, efr, epath, etarget = expected[k]
        assert g2 == (enxt, est, len(efr), epath, etarget), k
    axp.close()
    for g in grids:
        g.close()
