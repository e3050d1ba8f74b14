"""Row f1: the LCM wire codec of libbotlab_hip.so (host code, no GPU needed) against the
independent restatement in oracle/lcm_codec.py.  PARITY UNPINNED against real LCM 1.4.0 output (none exists in the
reference); what is pinned here: the member tables against the reference's .lcm files (tests/golden/lcm_types.json),
byte-identical encodings, decode round trips, error behaviour on truncated / foreign messages, log-event framing."""
import ctypes as C
import json
import os
import struct
import sys

import numpy as np
import pytest

import helpers
from botlab_amd import _capi
from botlab_amd.host import PARTICLE_DTYPE, POSE_DTYPE

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import lcm_codec  # noqa: E402

ORDER = ["pose_xyt_t", "odometry_t", "lidar_t", "particle_t", "particles_t", "occupancy_grid_t", "robot_path_t"]


@pytest.fixture(scope="module")
def lib():
    return _capi.load()


def _buf(n):
    return (C.c_uint8 * n)()


def test_member_tables_match_the_reference_lcm_files():
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "lcm_types.json")))
    assert {k: [list(m) for m in v] for k, v in lcm_codec.TYPES.items()} == golden


def test_fingerprints(lib):
    for i, name in enumerate(ORDER):
        assert lib.bl_lcm_fingerprint(i) == lcm_codec.fingerprint(name), name
    assert lib.bl_lcm_fingerprint(0) == lib.bl_lcm_fingerprint(1)            # same members: the struct name is not hashed
    assert len({lib.bl_lcm_fingerprint(i) for i in range(7)}) == 6
    assert lib.bl_lcm_fingerprint(99) == 0


def _pose_dict(p):
    return {"utime": int(p["utime"]), "x": float(p["x"]), "y": float(p["y"]), "theta": float(p["theta"])}


def test_pose_and_odometry_round_trip(lib):
    p = _capi.Pose(-123456789012, 1.25, -3.5e-3, 3.0)
    for typ, name in ((0, "pose_xyt_t"), (1, "odometry_t")):
        n = lib.bl_lcm_encode_pose(typ, C.byref(p), None, 0)
        assert n == 28
        b = _buf(n)
        assert lib.bl_lcm_encode_pose(typ, C.byref(p), b, n) == n
        assert bytes(b) == lcm_codec.encode(name, {"utime": p.utime, "x": p.x, "y": p.y, "theta": p.theta})
        q = _capi.Pose()
        _capi.check(lib.bl_lcm_decode_pose(typ, b, n, C.byref(q)))
        assert (q.utime, q.x, q.y, q.theta) == (p.utime, p.x, p.y, p.theta)
        assert lib.bl_lcm_encode_pose(typ, C.byref(p), b, n - 1) == -_capi.BL_ERR_CAPACITY
        assert lib.bl_lcm_decode_pose(typ, b, n - 1, C.byref(q)) != 0            # truncated
    assert lib.bl_lcm_encode_pose(2, C.byref(p), None, 0) == -_capi.BL_ERR_ARG


def test_lidar_round_trip(lib):
    rng = np.random.default_rng(1)
    for n in (0, 1, 290):
        ranges = rng.uniform(0, 8, n).astype(np.float32); thetas = rng.uniform(0, 6.3, n).astype(np.float32)
        times = rng.integers(0, 2**50, n).astype(np.int64); inten = rng.uniform(0, 1, n).astype(np.float32)
        scan = _capi.Lidar(77, n, ranges.ctypes.data_as(C.POINTER(C.c_float)), thetas.ctypes.data_as(C.POINTER(C.c_float)),
                           times.ctypes.data_as(C.POINTER(C.c_int64)))
        size = lib.bl_lcm_encode_lidar(C.byref(scan), inten.ctypes.data, None, 0)
        assert size == 8 + 8 + 4 + 20 * n
        b = _buf(size)
        assert lib.bl_lcm_encode_lidar(C.byref(scan), inten.ctypes.data, b, size) == size
        exp = lcm_codec.encode("lidar_t", {"utime": 77, "num_ranges": n, "ranges": ranges.tolist(), "thetas": thetas.tolist(),
                                           "times": times.tolist(), "intensities": inten.tolist()})
        assert bytes(b) == exp
        ut, cnt = C.c_int64(), C.c_int32()
        r2, t2, i2 = np.zeros(n + 1, np.float32), np.zeros(n + 1, np.float32), np.zeros(n + 1, np.float32)
        tm2 = np.zeros(n + 1, np.int64)
        _capi.check(lib.bl_lcm_decode_lidar(b, size, C.byref(ut), C.byref(cnt), r2.ctypes.data, t2.ctypes.data, tm2.ctypes.data, i2.ctypes.data, n + 1))
        assert (ut.value, cnt.value) == (77, n)
        assert np.array_equal(r2[:n], ranges) and np.array_equal(t2[:n], thetas) and np.array_equal(tm2[:n], times) and np.array_equal(i2[:n], inten)
        # intensities == NULL encodes zeros, as the simulator publishes them (src/sim/lidar.py:147)
        assert lib.bl_lcm_encode_lidar(C.byref(scan), None, b, size) == size
        assert bytes(b)[size - 4 * n:] == b"\x00" * (4 * n)


def test_particles_grid_path_round_trip(lib):
    rng = np.random.default_rng(2)
    n = 257
    parts = np.zeros(n, PARTICLE_DTYPE)
    for f in ("x", "y", "theta", "p_x", "p_y", "p_theta"):
        parts[f] = rng.normal(size=n).astype(np.float32)
    parts["utime"] = 5; parts["p_utime"] = rng.integers(0, 2**40, n); parts["weight"] = rng.uniform(0, 1, n)
    size = lib.bl_lcm_encode_particles(99, parts.ctypes.data, n, None, 0)
    assert size == 20 + 48 * n
    b = _buf(size)
    assert lib.bl_lcm_encode_particles(99, parts.ctypes.data, n, b, size) == size
    exp = lcm_codec.encode("particles_t", {"utime": 99, "num_particles": n, "particles": [
        {"pose": {"utime": int(p["utime"]), "x": float(p["x"]), "y": float(p["y"]), "theta": float(p["theta"])},
         "parent_pose": {"utime": int(p["p_utime"]), "x": float(p["p_x"]), "y": float(p["p_y"]), "theta": float(p["p_theta"])},
         "weight": float(p["weight"])} for p in parts]})
    assert bytes(b) == exp
    out = np.zeros(n, PARTICLE_DTYPE)
    ut, cnt = C.c_int64(), C.c_int32()
    _capi.check(lib.bl_lcm_decode_particles(b, size, C.byref(ut), C.byref(cnt), out.ctypes.data, n))
    assert (ut.value, cnt.value) == (99, n) and out.tobytes() == parts.tobytes()
    bad = bytearray(bytes(b)); bad[3] ^= 1
    assert lib.bl_lcm_decode_particles((C.c_uint8 * size).from_buffer(bad), size, C.byref(ut), C.byref(cnt), out.ctypes.data, n) != 0

    cells = rng.integers(-128, 128, size=(37, 53)).astype(np.int8)
    size = lib.bl_lcm_encode_grid(7, np.float32(-1.5), np.float32(2.25), np.float32(0.05), 53, 37, cells.ctypes.data, None, 0)
    b = _buf(size)
    assert lib.bl_lcm_encode_grid(7, np.float32(-1.5), np.float32(2.25), np.float32(0.05), 53, 37, cells.ctypes.data, b, size) == size
    exp = lcm_codec.encode("occupancy_grid_t", {"utime": 7, "origin_x": -1.5, "origin_y": 2.25, "meters_per_cell": float(np.float32(0.05)),
                                                "width": 53, "height": 37, "num_cells": 53 * 37, "cells": cells.reshape(-1).tolist()})
    assert bytes(b) == exp
    ut = C.c_int64(); fr = (C.c_float * 3)(); dims = (C.c_int32 * 3)(); c2 = np.zeros(53 * 37, np.int8)
    _capi.check(lib.bl_lcm_decode_grid(b, size, C.byref(ut), fr, dims, c2.ctypes.data, c2.size))
    assert ut.value == 7 and list(dims) == [53, 37, 53 * 37] and np.array_equal(c2.reshape(37, 53), cells)

    path = np.zeros(12, POSE_DTYPE)
    path["x"] = rng.normal(size=12).astype(np.float32); path["y"] = rng.normal(size=12).astype(np.float32); path["theta"] = 0.5
    size = lib.bl_lcm_encode_path(1234, path.ctypes.data, 12, None, 0)
    b = _buf(size)
    assert lib.bl_lcm_encode_path(1234, path.ctypes.data, 12, b, size) == size
    assert bytes(b) == lcm_codec.encode("robot_path_t", {"utime": 1234, "path_length": 12, "path": [_pose_dict(p) for p in path]})
    out = np.zeros(12, POSE_DTYPE); cnt = C.c_int32()
    _capi.check(lib.bl_lcm_decode_path(b, size, C.byref(ut), C.byref(cnt), out.ctypes.data, 12))
    assert cnt.value == 12 and out.tobytes() == path.tobytes()
    assert lcm_codec.decode("robot_path_t", bytes(b))["path"][3]["x"] == float(path["x"][3])


def test_log_event_framing(lib):
    events = [(0, 1000, "LIDAR", os.urandom(37)), (1, 1500, "ODOMETRY", b""), (2, 99999999999, "SLAM_POSE", os.urandom(28))]
    blob = b""
    for num, ts, ch, data in events:
        size = lib.bl_lcm_log_event_size(len(ch), len(data))
        b = _buf(size)
        d = (C.c_uint8 * max(len(data), 1)).from_buffer_copy(data or b"\x00")
        assert lib.bl_lcm_log_write_event(num, ts, ch.encode(), d, len(data), b, size) == size
        assert bytes(b) == lcm_codec.log_event(num, ts, ch, data)
        blob += bytes(b)
    raw = (C.c_uint8 * len(blob)).from_buffer_copy(blob)
    off, got = 0, []
    while off < len(blob):
        en, ts, co, do = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        cl, dl = C.c_int32(), C.c_int32()
        n = lib.bl_lcm_log_read_event(C.byref(raw, off), len(blob) - off, C.byref(en), C.byref(ts), C.byref(co), C.byref(cl), C.byref(do), C.byref(dl))
        assert n > 0
        got.append((en.value, ts.value, blob[off + co.value:off + co.value + cl.value].decode(), blob[off + do.value:off + do.value + dl.value]))
        off += n
    assert got == events
    assert lib.bl_lcm_log_read_event(raw, 30, None, None, None, None, None, None) == 0          # incomplete event: wait for more bytes
    assert lib.bl_lcm_log_read_event(C.byref(raw, 1), len(blob) - 1, None, None, None, None, None, None) < 0   # no sync word here
