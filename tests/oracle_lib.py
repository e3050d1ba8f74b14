"""ctypes binding of the CPU oracle (oracle/liboracle.so) and of oracle/_ref/libref_math.so.
TEST INFRASTRUCTURE ONLY: nothing under botlab_amd/ imports this."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")


class OPose(C.Structure):
    _fields_ = [("utime", C.c_int64), ("x", C.c_float), ("y", C.c_float), ("theta", C.c_float)]


class OParticle(C.Structure):
    _fields_ = [("pose", OPose), ("parent_pose", OPose), ("weight", C.c_double)]


class OGrid(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("mpc", C.c_float), ("cpm", C.c_float),
                ("ox", C.c_float), ("oy", C.c_float), ("cells", C.c_void_p)]


class OLidar(C.Structure):
    _fields_ = [("utime", C.c_int64), ("num_ranges", C.c_int32), ("ranges", C.c_void_p), ("thetas", C.c_void_p),
                ("times", C.c_void_p)]


class ORay(C.Structure):
    _fields_ = [("ox", C.c_float), ("oy", C.c_float), ("range", C.c_float), ("theta", C.c_float)]


class OSearchParams(C.Structure):
    _fields_ = [("minDistanceToObstacle", C.c_double), ("maxDistanceWithCost", C.c_double),
                ("distanceCostExponent", C.c_double)]


PARTICLE_DTYPE = np.dtype([("utime", "<i8"), ("x", "<f4"), ("y", "<f4"), ("theta", "<f4"), ("_pad0", "<f4"),
                           ("p_utime", "<i8"), ("p_x", "<f4"), ("p_y", "<f4"), ("p_theta", "<f4"), ("_pad1", "<f4"),
                           ("weight", "<f8")])


def _build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        L = lib
        L.orc_wrap_to_pi.restype = C.c_float
        L.orc_wrap_to_pi.argtypes = [C.c_float]
        L.orc_angle_diff.restype = C.c_double
        L.orc_angle_diff.argtypes = [C.c_double, C.c_double]
        L.orc_angle_sum.restype = C.c_double
        L.orc_angle_sum.argtypes = [C.c_double, C.c_double]
        L.orc_cosf.restype = C.c_float
        L.orc_cosf.argtypes = [C.c_float]
        L.orc_sinf.restype = C.c_float
        L.orc_sinf.argtypes = [C.c_float]
        L.orc_interpolate_pose.argtypes = [C.c_int64, C.POINTER(OPose), C.POINTER(OPose), C.POINTER(OPose)]
        L.orc_moving_scan.restype = C.c_int
        L.orc_moving_scan.argtypes = [C.POINTER(OLidar), C.POINTER(OPose), C.POINTER(OPose), C.c_void_p, C.c_int]
        L.orc_mapping_create.restype = C.c_void_p
        L.orc_mapping_create.argtypes = [C.c_float, C.c_int8, C.c_int8]
        L.orc_mapping_destroy.argtypes = [C.c_void_p]
        L.orc_mapping_update.argtypes = [C.c_void_p, C.POINTER(OLidar), C.POINTER(OPose), C.POINTER(OGrid)]
        L.orc_pf_create.restype = C.c_void_p
        L.orc_pf_create.argtypes = [C.c_int]
        L.orc_pf_destroy.argtypes = [C.c_void_p]
        L.orc_pf_set_particles.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_pf_get_particles.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_pf_init_at_pose.argtypes = [C.c_void_p, C.POINTER(OPose), C.c_uint32]
        L.orc_pf_update.argtypes = [C.c_void_p, C.POINTER(OPose), C.POINTER(OLidar), C.POINTER(OGrid), C.c_int, C.c_int,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(OPose), C.POINTER(C.c_int)]
        L.orc_pf_update_action_only.argtypes = [C.c_void_p, C.POINTER(OPose), C.c_int, C.c_void_p, C.POINTER(OPose)]
        L.orc_pf_action_state.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        L.orc_likelihood.argtypes = [C.c_void_p, C.c_int, C.POINTER(OLidar), C.POINTER(OGrid), C.c_void_p]
        L.orc_estimate_pose.argtypes = [C.c_void_p, C.c_int, C.POINTER(OPose)]
        L.orc_resample_indices.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_set_distances.argtypes = [C.POINTER(OGrid), C.POINTER(OGrid)]
        L.orc_search_for_path.restype = C.c_int
        L.orc_search_for_path.argtypes = [C.POINTER(OPose), C.POINTER(OPose), C.POINTER(OGrid), C.POINTER(OSearchParams),
                                          C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_heap_replay.restype = C.c_int
        L.orc_heap_replay.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p]
        L.orc_action_create.restype = C.c_void_p
        L.orc_action_destroy.argtypes = [C.c_void_p]
        L.orc_action_update.restype = C.c_int
        L.orc_action_update.argtypes = [C.c_void_p, C.POINTER(OPose)]
        L.orc_action_apply_noise.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_find_frontiers.restype = C.c_int
        L.orc_find_frontiers.argtypes = [C.POINTER(OGrid), C.POINTER(OPose), C.c_double, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                         C.POINTER(C.c_int)]
        L.orc_is_path_safe.restype = C.c_int
        L.orc_is_path_safe.argtypes = [C.c_void_p, C.c_int, C.POINTER(OGrid), C.c_double]
        L.orc_plan_path_to_frontier.restype = C.c_int
        L.orc_plan_path_to_frontier.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(OPose), C.POINTER(OGrid), C.c_double,
                                                C.POINTER(OSearchParams), C.c_int, C.POINTER(OPose), C.c_void_p, C.c_int,
                                                C.POINTER(OPose), C.c_void_p]
        L.orc_trace_create.restype = C.c_void_p
        L.orc_trace_destroy.argtypes = [C.c_void_p]
        L.orc_trace_add.argtypes = [C.c_void_p, C.POINTER(OPose)]
        L.orc_trace_erase_until.restype = C.c_int
        L.orc_trace_erase_until.argtypes = [C.c_void_p, C.c_int64]
        L.orc_trace_pose_at.argtypes = [C.c_void_p, C.c_int64, C.POINTER(OPose)]
        L.orc_trace_contains.restype = C.c_int
        L.orc_trace_contains.argtypes = [C.c_void_p, C.c_int64]
        L.orc_trace_set_reference.argtypes = [C.c_void_p, C.POINTER(OPose)]
        L.orc_trace_size.restype = C.c_int
        L.orc_trace_size.argtypes = [C.c_void_p]
        L.orc_trace_get.argtypes = [C.c_void_p, C.c_int, C.POINTER(OPose)]
        L.orc_slam_create.restype = C.c_void_p
        L.orc_slam_create.argtypes = [C.c_int, C.c_int8, C.c_int8, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint32]
        L.orc_slam_destroy.argtypes = [C.c_void_p]
        L.orc_slam_handle_laser.argtypes = [C.c_void_p, C.POINTER(OLidar)]
        L.orc_slam_handle_odometry.argtypes = [C.c_void_p, C.POINTER(OPose)]
        L.orc_slam_handle_pose.argtypes = [C.c_void_p, C.POINTER(OPose)]
        L.orc_slam_handle_optitrack.argtypes = [C.c_void_p, C.POINTER(OPose)]
        L.orc_slam_ready.restype = C.c_int
        L.orc_slam_ready.argtypes = [C.c_void_p]
        L.orc_slam_iterate.argtypes = [C.c_void_p, C.c_int]
        L.orc_slam_state.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(OPose), C.c_void_p]
        L.orc_is_valid_goal.restype = C.c_int
        L.orc_is_valid_goal.argtypes = [C.POINTER(OPose), C.POINTER(OGrid), C.c_double, C.c_double, C.c_int, C.POINTER(OPose)]

    # ---- helpers building the C views
    @staticmethod
    def grid(cells, mpc, cpm, origin):
        cells = np.ascontiguousarray(cells)
        g = OGrid(cells.shape[1], cells.shape[0], np.float32(mpc), np.float32(cpm), np.float32(origin[0]),
                  np.float32(origin[1]), cells.ctypes.data)
        g._keep = cells
        return g

    @staticmethod
    def lidar(scan):
        l = OLidar(scan.utime, scan.num_ranges, scan.ranges.ctypes.data, scan.thetas.ctypes.data, scan.times.ctypes.data)
        l._keep = scan
        return l

    @staticmethod
    def pose(x=0.0, y=0.0, theta=0.0, utime=0):
        return OPose(int(utime), float(np.float32(x)), float(np.float32(y)), float(np.float32(theta)))

    # ---- operations
    def moving_scan(self, scan, begin, end):
        out = (ORay * max(1, scan.num_ranges))()
        l = self.lidar(scan)
        n = self.lib.orc_moving_scan(C.byref(l), C.byref(begin), C.byref(end), out, scan.num_ranges)
        return np.array([(r.ox, r.oy, r.range, r.theta) for r in out[:n]], dtype=np.float32).reshape(-1, 4)

    def set_distances(self, cells, mpc, cpm, origin):
        g = self.grid(cells.astype(np.int8), mpc, cpm, origin)
        out = np.zeros(cells.shape, dtype=np.float32)
        d = self.grid(out, mpc, cpm, origin)
        self.lib.orc_set_distances(C.byref(g), C.byref(d))
        return out

    def search(self, start, goal, dist, mpc, cpm, origin, min_dist, max_dist, exponent=1.0, literal=0, cap=1 << 20):
        d = self.grid(np.ascontiguousarray(dist, dtype=np.float32), mpc, cpm, origin)
        sp = OSearchParams(min_dist, max_dist, exponent)
        out = (OPose * cap)()
        st = (C.c_int64 * 2)()
        n = self.lib.orc_search_for_path(C.byref(start), C.byref(goal), C.byref(d), C.byref(sp), literal, out, cap, st)
        path = np.array([(p.utime, p.x, p.y, p.theta) for p in out[:n]],
                        dtype=[("utime", "<i8"), ("x", "<f4"), ("y", "<f4"), ("theta", "<f4")])
        return path, (st[0], st[1])

    def find_frontiers(self, cells, mpc, cpm, origin, robot, min_len=0.35):
        """find_map_frontiers: list of (n, 2) float32 arrays of cell coordinates, in the reference's order."""
        cells = np.ascontiguousarray(cells, dtype=np.int8)
        g = self.grid(cells, mpc, cpm, origin)
        cap = cells.size + 1
        offs = np.zeros(cap + 1, np.int32)
        xy = np.zeros((cap, 2), np.float32)
        tot = C.c_int(0)
        n = self.lib.orc_find_frontiers(C.byref(g), C.byref(robot), float(min_len), offs.ctypes.data, cap, xy.ctypes.data, cap, C.byref(tot))
        return [xy[offs[k]:offs[k + 1]].copy() for k in range(n)]

    def plan_path_to_frontier(self, frontiers, robot, dist, mpc, cpm, origin, robot_radius, min_dist, max_dist, exponent=1.0,
                              num_frontiers=None, prev_goal=None, cap=1 << 16):
        """plan_path_to_frontier (frontiers.cpp:104-214): (path, chosen goal, (pops, pushes))."""
        d = self.grid(np.ascontiguousarray(dist, dtype=np.float32), mpc, cpm, origin)
        offs = np.zeros(len(frontiers) + 1, np.int32)
        for k, f in enumerate(frontiers):
            offs[k + 1] = offs[k] + len(f)
        xy = np.ascontiguousarray(np.concatenate(frontiers) if frontiers else np.zeros((0, 2)), dtype=np.float32)
        sp = OSearchParams(min_dist, max_dist, exponent)
        pg = prev_goal if prev_goal is not None else self.pose(1e9, 1e9, 0.0)
        out = (OPose * cap)()
        goal = OPose()
        st = (C.c_int64 * 2)()
        n = self.lib.orc_plan_path_to_frontier(offs.ctypes.data, len(frontiers), xy.ctypes.data, C.byref(robot), C.byref(d),
                                               float(robot_radius), C.byref(sp), len(frontiers) if num_frontiers is None else num_frontiers,
                                               C.byref(pg), out, cap, C.byref(goal), st)
        path = np.array([(p.utime, p.x, p.y, p.theta) for p in out[:n]],
                        dtype=[("utime", "<i8"), ("x", "<f4"), ("y", "<f4"), ("theta", "<f4")])
        return path, (goal.x, goal.y, goal.theta), (st[0], st[1])

    def is_valid_goal(self, goal, dist, mpc, cpm, origin, robot_radius, min_dist, num_frontiers=1, prev_goal=None):
        d = self.grid(np.ascontiguousarray(dist, dtype=np.float32), mpc, cpm, origin)
        pg = prev_goal or self.pose(1e9, 1e9, 0)
        return bool(self.lib.orc_is_valid_goal(C.byref(goal), C.byref(d), robot_radius, min_dist, num_frontiers, C.byref(pg)))


class OracleMapping:
    def __init__(self, orc, max_laser, hit, miss):
        self.o = orc
        self.h = orc.lib.orc_mapping_create(np.float32(max_laser), hit, miss)

    def update(self, scan, pose, cells, mpc, cpm, origin):
        """cells (H, W) int8, modified in place."""
        g = self.o.grid(cells, mpc, cpm, origin)
        l = self.o.lidar(scan)
        self.o.lib.orc_mapping_update(self.h, C.byref(l), C.byref(pose), C.byref(g))

    def __del__(self):
        if self.h:
            self.o.lib.orc_mapping_destroy(self.h)
            self.h = None


class OraclePF:
    def __init__(self, orc, n):
        self.o = orc
        self.N = n
        self.h = orc.lib.orc_pf_create(n)

    def init_at_pose(self, pose, seed):
        self.o.lib.orc_pf_init_at_pose(self.h, C.byref(pose), seed)

    def set_particles(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.dtype.itemsize == 56 and arr.size == self.N
        self.o.lib.orc_pf_set_particles(self.h, arr.ctypes.data)

    def particles(self):
        out = np.zeros(self.N, dtype=PARTICLE_DTYPE)
        self.o.lib.orc_pf_get_particles(self.h, out.ctypes.data)
        return out

    def update(self, odom, scan, cells, mpc, cpm, origin, rand_value, noise_in=None):
        """Returns dict(pose, moved, noise, idx, raw).  noise_in: consume these samples instead of drawing."""
        g = self.o.grid(cells, mpc, cpm, origin)
        l = self.o.lidar(scan)
        noise = np.zeros(3 * self.N, np.float32) if noise_in is None else np.ascontiguousarray(noise_in, np.float32).copy()
        idx = np.zeros(self.N, np.int32)
        raw = np.zeros(self.N, np.float64)
        out = OPose()
        moved = C.c_int()
        self.o.lib.orc_pf_update(self.h, C.byref(odom), C.byref(l), C.byref(g), int(rand_value), 0 if noise_in is None else 1,
                                 noise.ctypes.data, idx.ctypes.data, raw.ctypes.data, C.byref(out), C.byref(moved))
        return dict(pose=out, moved=bool(moved.value), noise=noise, idx=idx, raw=raw)

    def update_action_only(self, odom, noise_in=None):
        noise = np.zeros(3 * self.N, np.float32) if noise_in is None else np.ascontiguousarray(noise_in, np.float32).copy()
        out = OPose()
        self.o.lib.orc_pf_update_action_only(self.h, C.byref(odom), 0 if noise_in is None else 1, noise.ctypes.data, C.byref(out))
        return out, noise

    def __del__(self):
        if self.h:
            self.o.lib.orc_pf_destroy(self.h)
            self.h = None


_oracle = None


def load_oracle():
    global _oracle
    if _oracle is None:
        path = os.path.join(ORACLE_DIR, "liboracle.so")
        if not os.path.exists(path):
            _build()
        _oracle = Oracle(C.CDLL(path))
    return _oracle


def load_ref_math():
    """oracle/_ref/libref_math.so (compiled from the reference's own headers), or None when it was not built."""
    path = os.path.join(ORACLE_DIR, "_ref", "libref_math.so")
    if not os.path.exists(path):
        return None
    L = C.CDLL(path)
    L.ref_wrap_to_pi.restype = C.c_float
    L.ref_wrap_to_pi.argtypes = [C.c_float]
    L.ref_angle_diff.restype = C.c_double
    L.ref_angle_diff.argtypes = [C.c_double, C.c_double]
    L.ref_angle_sum.restype = C.c_double
    L.ref_angle_sum.argtypes = [C.c_double, C.c_double]
    L.ref_interpolate_pose.argtypes = [C.c_int64, C.POINTER(OPose), C.POINTER(OPose), C.POINTER(OPose)]
    return L


class OracleExploringMap:
    """Exploration::executeExploringMap (src/planning/exploration.cpp:277-369) over the oracle's own pieces (set_distances,
    find_frontiers, plan_path_to_frontier): TEST INFRASTRUCTURE, the checker of botlab_amd.host.ExploringMap.
    States / statuses are the constants of lcmtypes/exploration_status_t.lcm:3-11; the unset status of :344-347 is taken as
    FAILED (definition D10: the switch of :352-368 ends in its default branch)."""

    def __init__(self, orc, robot_radius=0.2, prev_goal=None):
        self.orc = orc
        self.r = float(robot_radius)
        self.prev_goal = prev_goal if prev_goal is not None else orc.pose(1e9, 1e9, 0.0)
        self.target = (np.float32(0.0), np.float32(0.0))
        self.path = []
        self.status = None

    def execute(self, cells, mpc, cpm, origin, pose):
        orc = self.orc
        dist = orc.set_distances(cells, mpc, cpm, origin)                                   # :299
        fr = orc.find_frontiers(cells, mpc, cpm, origin, pose)                              # :300
        if self.target[0] != 0 or self.target[1] != 0:                                      # :307-311
            dx = float(np.float32(pose.x) - self.target[0]); dy = float(np.float32(pose.y) - self.target[1])
            cur = float(np.float32(np.sqrt(dx * dx + dy * dy)))
        else:
            cur = 0.0
        if cur <= 0.5 and len(fr) > 0:                                                      # :316-321
            path, goal, _ = orc.plan_path_to_frontier(fr, pose, dist, mpc, cpm, origin, self.r, self.r, 10.0 * self.r, 1.0,
                                                      num_frontiers=len(fr), prev_goal=self.prev_goal)
            self.path = path
            if len(path) > 1:
                self.target = (np.float32(path[-1]["x"]), np.float32(path[-1]["y"]))
        if len(fr) == 0:                                                                    # :335-347
            self.status = 1
        elif len(self.path) > 1:
            self.status = 0
        else:
            self.status = 2
        return {0: 1, 1: 2, 2: 4}[self.status], fr                                          # :352-368
