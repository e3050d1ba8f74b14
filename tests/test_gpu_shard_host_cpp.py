"""include/botlab/sharded_filter.hpp -- a C++ host that drives R particle shards from one process through the C ABI alone (composed
finish, peer-store exchange; no Python, no torch.distributed in the loop) -- compiled with g++, linked against libbotlab_hip.so
and run on a script of scans: tests/cpp/shard_host_test.cpp compares every rank's estimate after every step, the particle set and
every rank's map with ONE rank's, bit for bit.  What is sharded: /root/reference/src/slam/particle_filter.cpp:84-160."""
import os
import struct
import subprocess
import tempfile

import numpy as np
import pytest

import helpers
from botlab_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world,n", [(2, 30001), (4, 100_000), (8, 100_000)])
def test_cpp_shard_host_matches_single_rank(maps, world, n):
    exe = os.path.join(ROOT, "tests", "cpp", "shard_host_test")
    subprocess.check_call(["g++", "-std=c++11", "-O2", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "shard_host_test.cpp"), "-L" + os.path.join(ROOT, "botlab_amd"),
                           "-lbotlab_hip", "-Wl,-rpath," + os.path.join(ROOT, "botlab_amd"), "-o", exe])
    m = maps["obstacle_slam_10mx10m_5cm"]
    truth = np.where(m["cells"] > 0, 127, -127).astype(np.int8)
    steps = 6
    poses = synth.square_trajectory((-0.75, 0.2, 0.0), steps, step_len=0.03, turn=0.05, side=0.8)
    scans = [synth.raycast_scan(truth, m["origin"], 0.05, poses[k - 1], poses[k], 1_000_000 + k * 100_000) for k in range(1, steps + 1)]
    with tempfile.TemporaryDirectory() as td:
        script = os.path.join(td, "script.bin")
        with open(script, "wb") as f:
            f.write(struct.pack("<iiiii", n, world, 200, 200, steps))
            f.write(struct.pack("<ffff", np.float32(m["mpc"]), helpers.CPM_DEFAULT, np.float32(m["origin"][0]), np.float32(m["origin"][1])))
            f.write(np.ascontiguousarray(m["cells"], np.int8).tobytes())
            f.write(struct.pack("<qfffxxxx", int(scans[0].times[0]), -0.75, 0.2, 0.0))
            for k, sc in enumerate(scans):
                f.write(struct.pack("<qfffxxxx", sc.utime, *[np.float32(v) for v in poses[k + 1]]))
                f.write(struct.pack("<qii", sc.utime, sc.num_ranges, 900 + k))
                f.write(np.ascontiguousarray(sc.ranges, np.float32).tobytes() + np.ascontiguousarray(sc.thetas, np.float32).tobytes() +
                        np.ascontiguousarray(sc.times, np.int64).tobytes())
        env = dict(os.environ, BOTLAB_MCL_NO_FUSED_FINISH="1")       # the single rank takes the record-based finish every shard count shares
        r = subprocess.run([exe, script], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        assert f"shard_host_test ok: {world} ranks" in r.stdout.decode()
