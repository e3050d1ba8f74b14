"""PoseTraceT (include/botlab/slam_driver.hpp; pure host code) against the oracle's restatement of
src/common/pose_trace.cpp: poseAt interpolation/extrapolation, containsPoseAtTime, eraseTraceUntil, setReferencePose.
Runs the C++ test driver in trace-only mode, which never touches the GPU."""
import ctypes as C
import os
import struct
import subprocess
import tempfile

import numpy as np

import oracle_lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pose_trace_matches_oracle(oracle):
    exe = os.path.join(ROOT, "tests", "cpp", "slam_driver_test")
    subprocess.check_call(["g++", "-std=c++11", "-O2", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "slam_driver_test.cpp"), "-L" + os.path.join(ROOT, "botlab_amd"),
                           "-lbotlab_hip", "-Wl,-rpath," + os.path.join(ROOT, "botlab_amd"), "-o", exe])
    rng = np.random.default_rng(11)
    ev = []
    t = 5_000
    times = []
    for k in range(40):
        t += int(rng.integers(1, 200_000))
        times.append(t)
        ev.append(("P", (t, np.float32(rng.normal()), np.float32(rng.normal()), np.float32(rng.uniform(-3.14, 3.14)))))
        if k % 3 == 2:
            for q in (t, t - 1, t + 1, int(rng.integers(times[0] - 10, t + 10)), (times[-2] + t) // 2):
                ev.append(("Q", q))
        if k in (9, 30):                 # several samples with one time stamp: the bisection must pick the scan's first pair
            ev.append(("P", (t, np.float32(rng.normal()), np.float32(rng.normal()), np.float32(rng.uniform(-3.14, 3.14)))))
            ev.append(("P", (t, np.float32(rng.normal()), np.float32(rng.normal()), np.float32(rng.uniform(-3.14, 3.14)))))
            for q in (t, t - 1, times[-2], times[-2] + 1):
                ev.append(("Q", q))
        if k == 36:                       # a late sample: the stamps no longer ascend, poseAt falls back to the scan
            ev.append(("P", (t - 150_000, np.float32(0.5), np.float32(0.25), np.float32(-1.0))))
            for q in (t, t - 1, t - 150_000, t - 150_001, t - 75_000, times[3] + 5):
                ev.append(("Q", q))
        if k == 20:
            ev.append(("R", (0, np.float32(0.4), np.float32(-1.5), np.float32(2.2))))
        if k in (25, 33):
            ev.append(("X", times[k - 12] + (k == 33)))
            ev.append(("Q", times[k - 14]))
            ev.append(("Q", times[k - 11] + 3))
    ev.append(("X", t + 1))                                          # erases everything
    ev.append(("Q", t))
    with tempfile.TemporaryDirectory() as td:
        sp, op = os.path.join(td, "s.bin"), os.path.join(td, "o.bin")
        with open(sp, "wb") as f:
            f.write(struct.pack("<iiii", -1, 0, len(ev), 0))
            for kind, x in ev:
                f.write(kind.encode())
                f.write(struct.pack("<q", int(x)) if kind in "QX" else struct.pack("<qfff", int(x[0]), x[1], x[2], x[3]))
        env = dict(os.environ, HIP_VISIBLE_DEVICES="")
        out = subprocess.check_output([exe, sp, op], env=env).decode()
        assert "trace only" in out
        raw = open(op, "rb").read()
    L = oracle.lib
    tr = L.orc_trace_create()
    off = 0
    nq = 0
    for kind, x in ev:
        if kind in "PR":
            p = oracle.pose(x[1], x[2], x[3], utime=x[0])
            (L.orc_trace_add if kind == "P" else L.orc_trace_set_reference)(tr, C.byref(p))
        elif kind == "Q":
            got = struct.unpack_from("<qfffi", raw, off); off += 24
            e = oracle_lib.OPose()
            L.orc_trace_pose_at(tr, int(x), C.byref(e))
            assert got == (e.utime, e.x, e.y, e.theta, L.orc_trace_contains(tr, int(x))), (nq, x)
            nq += 1
        else:
            got = struct.unpack_from("<ii", raw, off); off += 8
            assert got == (L.orc_trace_erase_until(tr, int(x)), L.orc_trace_size(tr))
    assert raw[off:off + 1] == b"E" and nq > 60
    L.orc_trace_destroy(tr)
