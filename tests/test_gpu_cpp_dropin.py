"""The C++ drop-in classes (include/botlab/botlab_dropin.hpp) driven the way OccupancyGridSLAM::runSLAMIteration and
MotionPlanner drive the reference classes, checked against the oracle: the map must equal the oracle's Mapping replayed
on the poses the filter produced, the distance grid and the A* path must be bit-exact on that map."""
import os
import struct
import subprocess
import tempfile

import numpy as np
import pytest

import helpers
import oracle_lib
from botlab_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_dropin_end_to_end(oracle, maps):
    exe = os.path.join(ROOT, "tests", "cpp", "dropin_test")
    subprocess.check_call(["g++", "-std=c++11", "-O2", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "dropin_test.cpp"), "-L" + os.path.join(ROOT, "botlab_amd"),
                           "-lbotlab_hip", "-Wl,-rpath," + os.path.join(ROOT, "botlab_amd"), "-o", exe])
    m = maps["obstacle_slam_10mx10m_5cm"]
    truth = np.where(m["cells"] > 0, 127, -127).astype(np.int8)
    steps, N, R = 12, 5000, 290
    poses = synth.square_trajectory((-0.75, 0.2, 0.0), steps, step_len=0.02, turn=0.05, side=0.8)
    rng = np.random.default_rng(3)
    odo = synth.odometry_from_truth(poses, rng)
    scans = [synth.raycast_scan(truth, m["origin"], 0.05, poses[k - 1], poses[k], 1_000_000 + k * 100_000) for k in range(1, steps + 1)]
    with tempfile.TemporaryDirectory() as td:
        map_path, script, outp = os.path.join(td, "in.map"), os.path.join(td, "script.bin"), os.path.join(td, "out.bin")
        with open(map_path, "w") as f:                      # .map text format (occupancy_grid.cpp:111-136)
            f.write(f"{m['origin'][0]:g} {m['origin'][1]:g} 200 200 {m['mpc']:g}\n")
            for row in m["cells"]:
                f.write(" ".join(str(int(v)) for v in row) + " \n")
        with open(script, "wb") as f:
            f.write(struct.pack("<iii", steps, N, R))
            f.write(struct.pack("<qfff", int(scans[0].times[0]), *[np.float32(v) for v in odo[0]]))
            for k in range(steps):
                sc = scans[k]
                f.write(struct.pack("<q", sc.utime) + sc.ranges.tobytes() + sc.thetas.tobytes() + sc.times.tobytes())
                f.write(struct.pack("<qfff", sc.utime, *[np.float32(v) for v in odo[k + 1]]))
        out = subprocess.check_output([exe, map_path, script, outp]).decode()
        assert "dropin_test ok" in out
        raw = open(outp, "rb").read()
    off = 0
    est = np.frombuffer(raw, np.float32, steps * 3, off).reshape(steps, 3); off += steps * 12
    wsum = struct.unpack_from("<d", raw, off)[0]; off += 8
    a, b = struct.unpack_from("<bb", raw, off); off += 2
    cells = np.frombuffer(raw, np.int8, 40000, off).reshape(200, 200); off += 40000
    dist = np.frombuffer(raw, np.float32, 40000, off).reshape(200, 200); off += 160000
    plen = struct.unpack_from("<i", raw, off)[0]; off += 4
    path = np.frombuffer(raw, np.float32, plen * 3, off).reshape(plen, 3)

    assert abs(wsum - 1.0) < 1e-9
    assert np.abs(est[-1, :2] - poses[-1][:2]).max() < 0.05               # the filter tracks the truth
    assert a == 77 and b == cells[4, 3]                                     # host write went to the copy only
    # Mapping replayed by the oracle on the SAME poses: first updateMap call latches, the rest must match bit for bit
    cpm = helpers.CPM_DEFAULT
    ref = m["cells"].copy()
    om = oracle_lib.OracleMapping(oracle, 5.0, 4, 1)
    for k in range(steps):
        om.update(scans[k], oracle.pose(est[k, 0], est[k, 1], est[k, 2], utime=scans[k].utime), ref, m["mpc"], cpm, m["origin"])
    assert np.array_equal(cells, ref)
    exp_dist = oracle.set_distances(ref, m["mpc"], cpm, m["origin"])
    assert np.array_equal(dist.view(np.uint32), exp_dist.view(np.uint32))
    exp_path, _ = oracle.search(oracle.pose(est[-1, 0], est[-1, 1], est[-1, 2], utime=scans[-1].utime), oracle.pose(-0.35, 0.2, 0.0),
                                exp_dist, m["mpc"], cpm, m["origin"], 0.2, 2.0)
    assert plen == len(exp_path)
    got = np.stack([path[:, 0], path[:, 1], path[:, 2]], 1)
    exp = np.stack([exp_path["x"], exp_path["y"], exp_path["theta"]], 1)
    assert got.tobytes() == exp.tobytes()
