"""Row f4: the simulator's lidar beam march on the GPU against oracle/sim_lidar.py (a line-by-line
restatement of src/sim/lidar.py + map.py).  Double arithmetic in the reference's order on both sides: ranges must be
identical, including beams that leave the map (Map.at_xy's unchecked index arithmetic) and beams that never hit."""
import math
import os
import sys

import numpy as np
import pytest

import helpers
from botlab_amd import sim, synth

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import sim_lidar  # noqa: E402

pytestmark = pytest.mark.gpu


def _worlds(maps):
    m = maps["obstacle_slam_10mx10m_5cm"]
    yield "obstacle_slam", m["cells"], float(m["origin"][0]), float(m["origin"][1]), 0.05
    m = maps["convex_10mx10m_5cm_offcenter"]
    yield "offcenter", m["cells"], float(m["origin"][0]), float(m["origin"][1]), 0.05
    yield "tiled", synth.tile_world(maps["astar_maze"]["cells"], 333), -3.7, 1.25, 0.03


def test_beam_march_matches_oracle(maps, gpu_ctx):
    rng = np.random.default_rng(8)
    for name, cells, ox, oy, mpc in _worlds(maps):
        lid = sim.SimLidar(cells, ox, oy, mpc, ctx=gpu_ctx)
        world = sim_lidar.Map(cells, ox, oy, mpc)
        h, w = cells.shape
        n = 1500
        x = rng.uniform(ox - 1.0, ox + w * mpc + 1.0, n)          # some beams start outside the map
        y = rng.uniform(oy - 1.0, oy + h * mpc + 1.0, n)
        pose_theta = rng.uniform(-math.pi, math.pi, n)
        theta = rng.uniform(0, 2 * math.pi, n)
        ang = sim._clamp(pose_theta - theta)
        got = lid.cast(x, y, ang)
        exp = np.array([sim_lidar.beam_scan(world, float(x[i]), float(y[i]), float(pose_theta[i]), float(theta[i])) for i in range(n)])
        assert got.tobytes() == exp.tobytes(), name
        assert (exp < 8).sum() > 200 and (exp == 8).sum() > 20, name


def test_scan_loop_matches_oracle(maps, gpu_ctx):
    m = maps["obstacle_slam_10mx10m_5cm"]
    ox, oy = float(m["origin"][0]), float(m["origin"][1])
    lid = sim.SimLidar(m["cells"], ox, oy, 0.05, ctx=gpu_ctx)
    world = sim_lidar.Map(m["cells"], ox, oy, 0.05)

    def pose_at(t):                                      # a robot turning while it drives
        return (-0.75 + 0.2 * (t - 100.0), 0.2, 0.3 * (t - 100.0))

    got = lid.scans(pose_at, [100.0, 100.1])
    for k, now in enumerate((100.0, 100.1)):
        th, rg, tm = sim_lidar.scan(world, pose_at, now)
        assert got[k][0].tolist() == th and got[k][1].tolist() == rg and got[k][2].tolist() == tm
    msg = lid.scan_message(pose_at, 100.0)
    assert msg.num_ranges == 290 and msg.ranges.dtype == np.float32 and np.all(np.diff(msg.times) < 0)
