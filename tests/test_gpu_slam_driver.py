"""Row f2: PoseTrace and the OccupancyGridSLAM step scheduler (include/botlab/slam_driver.hpp,
C++ over the C ABI) against the oracle's restatement of src/common/pose_trace.cpp and src/slam/slam.cpp, fed the same
event sequence.  Mapping-only mode is fully deterministic: every per-iteration pose, the scan-queue bookkeeping and the
final map must be identical.  Full-SLAM mode uses different random streams (reference: random_device / mt19937), so it
is checked for the control flow and for tracking the truth."""
import ctypes as C
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


def _build():
    exe = os.path.join(ROOT, "tests", "cpp", "slam_driver_test")
    subprocess.check_call(["g++", "-std=c++11", "-O2", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "slam_driver_test.cpp"), "-L" + os.path.join(ROOT, "botlab_amd"),
                           "-lbotlab_hip", "-Wl,-rpath," + os.path.join(ROOT, "botlab_amd"), "-o", exe])
    return exe


def _events(maps, mode, steps=14, name="obstacle_slam_10mx10m_5cm", start=(-0.75, 0.2, 0.0)):
    """An event list with the hazards the reference's handlers deal with: scans before any pose/odometry (ignored), a
    scan whose pose has not arrived yet (queued, not ready), a short scan (< 100 ranges: skipped with an error)."""
    m = maps[name]
    truth = np.where(m["cells"] > 0, 127, -127).astype(np.int8)
    poses = synth.square_trajectory(start, steps, step_len=0.03, turn=0.05, side=0.8)
    ev = []
    t0 = 1_000_000
    sc0 = synth.raycast_scan(truth, m["origin"], 0.05, poses[0], poses[0], t0 - 150_000)
    ev.append(("L", sc0))                                        # before any pose: ignored
    for k in range(len(poses)):
        t = t0 + k * 100_000
        p = (t, *[np.float32(v) for v in poses[k]])
        if mode == 0:
            ev.append(("P", p))
        else:
            ev.append(("O", p))
        if k >= 1:
            sc = synth.raycast_scan(truth, m["origin"], 0.05, poses[k - 1], poses[k], t)
            if k == 5:
                short = synth.raycast_scan(truth, m["origin"], 0.05, poses[k - 1], poses[k], t, rays=90)
                ev.append(("L", short))                          # invalid scan (num_ranges <= 100)
            if k == 7:
                late = synth.raycast_scan(truth, m["origin"], 0.05, poses[k], poses[k], t + 100_000)
                ev.append(("L", sc))
                ev.append(("L", late))                           # its pose arrives with the next event: queued until then
                continue
            ev.append(("L", sc))
    return m, poses, ev


def _write_script(path, mode, nparticles, ev, queries=()):
    with open(path, "wb") as f:
        f.write(struct.pack("<iiii", mode, nparticles, len(ev) + len(queries), 0))
        for kind, x in ev:
            f.write(kind.encode())
            if kind in "OPTR":
                f.write(struct.pack("<qfff", int(x[0]), x[1], x[2], x[3]))
            elif kind == "L":
                f.write(struct.pack("<qi", x.utime, x.num_ranges) + x.ranges.tobytes() + x.thetas.tobytes() + x.times.tobytes())
        for kind, x in queries:
            f.write(kind.encode())
            f.write(struct.pack("<q", int(x)) if kind == "Q" else struct.pack("<qfff", int(x[0]), x[1], x[2], x[3]))


def test_mapping_only_driver_matches_oracle(oracle, maps):
    exe = _build()
    m, poses, ev = _events(maps, mode=0)
    ts = [e[1][0] for e in ev if e[0] == "P"]
    queries = [("Q", ts[0] - 5), ("Q", ts[0]), ("Q", (ts[2] + ts[3]) // 2), ("Q", ts[-1]), ("Q", ts[-1] + 9),
               ("R", (0, np.float32(1.0), np.float32(-2.0), np.float32(0.7))), ("Q", (ts[4] + ts[5]) // 2 + 17)]
    with tempfile.TemporaryDirectory() as td:
        script, outp = os.path.join(td, "s.bin"), os.path.join(td, "o.bin")
        _write_script(script, 0, 200, ev, queries)
        out = subprocess.check_output([exe, script, outp], stderr=subprocess.DEVNULL).decode()
        assert "slam_driver_test ok" in out
        raw = open(outp, "rb").read()
    # ---- the same events through the oracle's driver
    L = oracle.lib
    d = L.orc_slam_create(200, 4, 1, 0, 1, 0, None, 1)
    tr = L.orc_trace_create()
    exp_iters = []
    for kind, x in ev:
        if kind == "P":
            p = oracle.pose(x[1], x[2], x[3], utime=x[0])
            L.orc_slam_handle_pose(d, C.byref(p)); L.orc_trace_add(tr, C.byref(p))
        elif kind == "L":
            l = oracle.lidar(x)
            L.orc_slam_handle_laser(d, C.byref(l))
        while L.orc_slam_ready(d):
            L.orc_slam_iterate(d, 0)
            st = (C.c_int * 6)()
            cp = oracle_lib.OPose()
            L.orc_slam_state(d, st, C.byref(cp), None)
            exp_iters.append(((cp.utime, cp.x, cp.y, cp.theta), (st[0], st[1], st[2])))
    # iterations: poses and queue bookkeeping identical
    off = 0
    got_iters = []
    while raw[off:off + 1] == b"I":
        t, x, y, th, a, b, c = struct.unpack_from("<qfffiii", raw, off + 1)
        got_iters.append(((t, x, y, th), (a, b, c)))
        off += 33
    assert len(got_iters) == len(exp_iters) >= 12
    assert got_iters == exp_iters
    # PoseTrace queries
    for kind, x in queries:
        if kind == "R":
            p = oracle.pose(x[1], x[2], x[3], utime=x[0])
            L.orc_trace_set_reference(tr, C.byref(p))
            continue
        t, px, py, pth, c = struct.unpack_from("<qfffi", raw, off)
        off += 24
        e = oracle_lib.OPose()
        L.orc_trace_pose_at(tr, int(x), C.byref(e))
        assert (t, px, py, pth) == (e.utime, e.x, e.y, e.theta), (kind, x)
        assert c == L.orc_trace_contains(tr, int(x))
    assert raw[off:off + 1] == b"E"
    fin = struct.unpack_from("<iiiii", raw, off + 1)
    cells = np.frombuffer(raw, np.int8, 40000, off + 21).reshape(200, 200)
    st = (C.c_int * 6)()
    exp_cells = np.zeros((200, 200), np.int8)
    L.orc_slam_state(d, st, None, exp_cells.ctypes.data)
    assert fin[:3] == (st[0], st[1], st[2]) and fin[4] == st[4] and fin[3] == 0       # mapping-only publishes no pose
    assert np.array_equal(cells, exp_cells) and (cells != 0).sum() > 500
    L.orc_slam_destroy(d); L.orc_trace_destroy(tr)


def test_full_slam_driver_control_flow_and_tracking(oracle, maps):
    exe = _build()
    m, poses, ev = _events(maps, mode=3, steps=24)
    with tempfile.TemporaryDirectory() as td:
        script, outp = os.path.join(td, "s.bin"), os.path.join(td, "o.bin")
        _write_script(script, 3, 3000, ev)
        out = subprocess.check_output([exe, script, outp], stderr=subprocess.DEVNULL).decode()
        raw = open(outp, "rb").read()
    L = oracle.lib
    d = L.orc_slam_create(300, 4, 1, 0, 0, 0, None, 1)
    exp = []
    for kind, x in ev:
        if kind == "O":
            p = oracle.pose(x[1], x[2], x[3], utime=x[0]); L.orc_slam_handle_odometry(d, C.byref(p))
        elif kind == "L":
            l = oracle.lidar(x); L.orc_slam_handle_laser(d, C.byref(l))
        while L.orc_slam_ready(d):
            L.orc_slam_iterate(d, 12345)
            st = (C.c_int * 6)(); cp = oracle_lib.OPose()
            L.orc_slam_state(d, st, C.byref(cp), None)
            exp.append((cp.utime, (st[0], st[1], st[2])))
    off, got = 0, []
    while raw[off:off + 1] == b"I":
        t, x, y, th, a, b, c = struct.unpack_from("<qfffiii", raw, off + 1)
        got.append((t, (a, b, c), (x, y, th)))
        off += 33
    assert [(g[0], g[1]) for g in got] == exp                     # same iterations, same timestamps, same queue state
    fin = struct.unpack_from("<iiiii", raw, off + 1)
    st = (C.c_int * 6)(); L.orc_slam_state(d, st, None, None)
    assert fin[3] == st[3] and fin[4] == st[4]                    # SLAM_POSE / SLAM_MAP publish counts (every 5th map)
    # Full SLAM starts its own frame at (0, 0, 0) (slam.cpp:158-166), the truth starts at poses[0]; the reference's
    # filter lags the truth a few cm while the map is still thin.  The two drivers use different random streams, so the
    # estimates agree statistically, not bitwise: both within 2 cm of each other and 10 cm of the truth.
    last = got[-1][2]
    truth = (poses[-1][0] - poses[0][0], poses[-1][1] - poses[0][1])
    assert abs(last[0] - cp.x) < 0.02 and abs(last[1] - cp.y) < 0.02
    assert abs(last[0] - truth[0]) < 0.10 and abs(last[1] - truth[1]) < 0.10
    L.orc_slam_destroy(d)


def _write_map_file(path, m):
    """The reference's ASCII .map format (occupancy_grid.cpp:111-136), as the shipped data/*.map files are written."""
    c = m["cells"]
    with open(path, "w") as f:
        f.write(f"{float(m['origin'][0]):g} {float(m['origin'][1]):g} {c.shape[1]} {c.shape[0]} {float(m['mpc']):g}\n")
        for row in c:
            f.write(" ".join(str(int(v)) for v in row) + " \n")


def _oracle_run(oracle, m, ev, n, action_only, rand_value=12345):
    L = oracle.lib
    g = oracle.grid(m["cells"].copy(), m["mpc"], helpers.CPM_DEFAULT, m["origin"])
    d = L.orc_slam_create(n, 4, 1, 0, 0, 1 if action_only else 0, C.byref(g), 1)
    its = []
    for kind, x in ev:
        if kind == "O":
            p = oracle.pose(x[1], x[2], x[3], utime=x[0]); L.orc_slam_handle_odometry(d, C.byref(p))
        elif kind == "L":
            l = oracle.lidar(x); L.orc_slam_handle_laser(d, C.byref(l))
        while L.orc_slam_ready(d):
            L.orc_slam_iterate(d, rand_value)
            st = (C.c_int * 6)(); cp = oracle_lib.OPose()
            L.orc_slam_state(d, st, C.byref(cp), None)
            its.append(((cp.utime, cp.x, cp.y, cp.theta), (st[0], st[1], st[2])))
    st = (C.c_int * 6)()
    cells = np.zeros((200, 200), np.int8)
    L.orc_slam_state(d, st, None, cells.ctypes.data)
    L.orc_slam_destroy(d)
    return its, tuple(st), cells


def _driver_run(exe, mode, nparticles, ev, mapfile):
    with tempfile.TemporaryDirectory() as td:
        script, outp = os.path.join(td, "s.bin"), os.path.join(td, "o.bin")
        _write_script(script, mode, nparticles, ev)
        out = subprocess.check_output([exe, script, outp, mapfile], stderr=subprocess.DEVNULL).decode()
        assert "slam_driver_test ok" in out
        raw = open(outp, "rb").read()
    off, its = 0, []
    while raw[off:off + 1] == b"I":
        t, x, y, th, a, b, c = struct.unpack_from("<qfffiii", raw, off + 1)
        its.append(((t, x, y, th), (a, b, c)))
        off += 33
    assert raw[off:off + 1] == b"E"
    fin = struct.unpack_from("<iiiii", raw, off + 1)
    cells = np.frombuffer(raw, np.int8, 40000, off + 21).reshape(200, 200)
    return its, fin, cells


def test_action_only_driver_matches_oracle(oracle, maps, tmp_path):
    """--action-only on a loaded map (slam.cpp:36-45, 253-262): updateFilterActionOnly returns the odometry itself
    (particle_filter.cpp:54-65), so every iteration's pose, the queue bookkeeping, the publish counts and the map the
    (always-on, slam.cpp:276) map update leaves are deterministic: all identical to the oracle driver's."""
    exe = _build()
    m, poses, ev = _events(maps, mode=2, steps=16, name="convex_10mx10m_5cm", start=(0.0, 0.0, 0.0))
    mapfile = str(tmp_path / "convex.map")
    _write_map_file(mapfile, m)
    got_its, fin, cells = _driver_run(exe, 2, 500, ev, mapfile)
    exp_its, st, exp_cells = _oracle_run(oracle, m, ev, 500, action_only=True)
    assert len(got_its) == len(exp_its) >= 12
    assert got_its == exp_its
    assert fin[:3] == st[:3] and fin[3] == st[3] and fin[4] == st[4]
    assert np.array_equal(cells, exp_cells)
    assert (cells != m["cells"]).sum() > 20                   # the loaded map was extended: the mode test of slam.cpp:276 is always true


def test_localization_only_driver_control_flow_and_tracking(oracle, maps, tmp_path):
    """--localization-only on data/convex_10mx10m_5cm.map (BASELINE.json configs[2]'s mode, slam.cpp:36-45): updateFilter runs
    from the first iteration on (haveMap_ from the file), the map is still extended every iteration.  Different random
    streams on the two sides (reference: random_device / mt19937), so control flow and publish counts are compared exactly
    and the estimates statistically."""
    exe = _build()
    m, poses, ev = _events(maps, mode=1, steps=24, name="convex_10mx10m_5cm", start=(0.0, 0.0, 0.0))    # the reference's own start: the map frame's origin
    mapfile = str(tmp_path / "convex.map")
    _write_map_file(mapfile, m)
    got_its, fin, cells = _driver_run(exe, 1, 3000, ev, mapfile)
    exp_its, st, exp_cells = _oracle_run(oracle, m, ev, 300, action_only=False)
    assert [(g[0][0], g[1]) for g in got_its] == [(e[0][0], e[1]) for e in exp_its]      # same iterations, stamps, queue state
    assert len(got_its) >= 20
    assert fin[3] == st[3] > 0                                                             # a SLAM_POSE per localised iteration
    assert fin[4] == st[4]
    last, ol = got_its[-1][0], exp_its[-1][0]
    truth = (poses[-1][0] - poses[0][0], poses[-1][1] - poses[0][1])
    assert abs(last[1] - ol[1]) < 0.03 and abs(last[2] - ol[2]) < 0.03
    assert abs(last[1] - truth[0]) < 0.10 and abs(last[2] - truth[1]) < 0.10
    assert (cells != m["cells"]).sum() > 20                   # localization-only still maps (slam.cpp:276)
