"""Host-side mirror of the reference's C++ surfaces for the hot path, over the C ABI of libbotlab_hip.so.

Same names, argument meaning and error behaviour as the reference classes (the C++ drop-in with identical signatures
lives in include/botlab/; this module is what the parity tests and bench.py drive):

  OccupancyGrid          src/slam/occupancy_grid.hpp:51-209
  Mapping                src/slam/mapping.hpp:25-34
  ParticleFilter         src/slam/particle_filter.hpp:38-77
  ObstacleDistanceGrid   src/planning/obstacle_distance_grid.hpp:28-96
  search_for_path        src/planning/astar.hpp:58-61
  MotionPlanner          src/planning/motion_planner.hpp:89-130 (setMap / planPath / isValidGoal / setParams)

All arithmetic runs in the HIP library; nothing here computes a result on the CPU.
"""
import ctypes as C
import math

import numpy as np

from . import _capi
from ._capi import Lidar, Particle, Pose, SearchParams, check

_libc = C.CDLL(None)
_libc.rand.restype = C.c_int

PARTICLE_DTYPE = np.dtype([("utime", "<i8"), ("x", "<f4"), ("y", "<f4"), ("theta", "<f4"), ("_pad0", "<f4"),
                           ("p_utime", "<i8"), ("p_x", "<f4"), ("p_y", "<f4"), ("p_theta", "<f4"), ("_pad1", "<f4"),
                           ("weight", "<f8")])
POSE_DTYPE = np.dtype([("utime", "<i8"), ("x", "<f4"), ("y", "<f4"), ("theta", "<f4"), ("_pad", "<f4")])
assert PARTICLE_DTYPE.itemsize == 56 and POSE_DTYPE.itemsize == 24


def make_pose(x=0.0, y=0.0, theta=0.0, utime=0):
    return Pose(int(utime), float(np.float32(x)), float(np.float32(y)), float(np.float32(theta)))


class LidarScan:
    """lidar_t (lcmtypes/lidar_t.lcm): utime, ranges[], thetas[], times[]."""

    def __init__(self, ranges, thetas, times, utime=0):
        self.ranges = np.ascontiguousarray(ranges, dtype=np.float32)
        self.thetas = np.ascontiguousarray(thetas, dtype=np.float32)
        self.times = np.ascontiguousarray(times, dtype=np.int64)
        assert self.ranges.shape == self.thetas.shape == self.times.shape and self.ranges.ndim == 1
        self.utime = int(utime)
        self.num_ranges = int(self.ranges.size)

    def as_c(self):
        return Lidar(self.utime, self.num_ranges, self.ranges.ctypes.data_as(C.POINTER(C.c_float)),
                     self.thetas.ctypes.data_as(C.POINTER(C.c_float)), self.times.ctypes.data_as(C.POINTER(C.c_int64)),
                     None)


class Context:
    """One HIP stream's worth of state on one device (bl_ctx)."""

    def __init__(self, device=0, stream=None):
        self.lib = _capi.load()
        h = C.c_void_p()
        check(self.lib.bl_ctx_create(int(device), C.c_void_p(stream) if stream else None, C.byref(h)))
        self.h = h
        self.device = device

    def scanPrefetch(self, scan):
        """Hands the NEXT lidar scan over early: the next map kernel of this context copies it to the device beside its own
        work, and the update that later brings the same scan launches no fetch kernel (bl_scan_prefetch)."""
        c = scan.as_c()
        check(self.lib.bl_scan_prefetch(self.h, C.byref(c)))

    def sync(self):
        check(self.lib.bl_ctx_sync(self.h))

    def timing_enable(self, on=True, kernels=None):
        """kernels: iterable of BL_K_* ids to time (default: all)."""
        flag = 0 if not on else (1 if kernels is None else sum(1 << k for k in kernels))
        if flag == 1 and kernels is not None:      # only kernel 0 requested: bit 0 alone is spelled 1 | (1 << 31)
            flag = 1 | (1 << 30)
        check(self.lib.bl_ctx_timing_enable(self.h, flag))

    def timing_stride(self, every):
        """Time only every `every`-th launch of each enabled kernel."""
        check(self.lib.bl_ctx_timing_stride(self.h, int(every)))

    def timing_reset(self):
        check(self.lib.bl_ctx_timing_reset(self.h))

    def timing_get(self, kernel_id):
        ms, n = C.c_double(), C.c_int64()
        check(self.lib.bl_ctx_timing_get(self.h, kernel_id, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def close(self):
        if self.h:
            self.lib.bl_ctx_destroy(self.h)
            self.h = None


_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


class OccupancyGrid:
    """Device-resident int8 log-odds grid (occupancy_grid.hpp).  Float members follow the reference's float32
    arithmetic (occupancy_grid.cpp:19-36)."""

    def __init__(self, widthInMeters=None, heightInMeters=None, metersPerCell=0.05, ctx=None, _raw=None):
        self.ctx = ctx or default_context()
        lib = self.ctx.lib
        if _raw is not None:
            width, height, mpc, cpm, ox, oy = _raw
        else:
            assert widthInMeters > 0 and heightInMeters > 0            # occupancy_grid.cpp:25-28
            mpc = np.float32(metersPerCell)
            assert mpc <= np.float32(widthInMeters) and mpc <= np.float32(heightInMeters)
            cpm = np.float32(1.0) / mpc
            width = int(np.float32(widthInMeters) * cpm)
            height = int(np.float32(heightInMeters) * cpm)
            ox = -np.float32(widthInMeters) / np.float32(2.0)
            oy = -np.float32(heightInMeters) / np.float32(2.0)
        self.width, self.height = int(width), int(height)
        self.mpc, self.cpm = np.float32(mpc), np.float32(cpm)
        self.origin = (np.float32(ox), np.float32(oy))
        h = C.c_void_p()
        check(lib.bl_grid_create(self.ctx.h, self.width, self.height, self.mpc, self.cpm, self.origin[0], self.origin[1],
                                 C.byref(h)))
        self.h = h

    @classmethod
    def from_cells(cls, cells, origin, metersPerCell, cellsPerMeter=None, ctx=None):
        """loadFromFile / fromLCM equivalent: cells is an (H, W) int8 array.  loadFromFile keeps the cellsPerMeter_ of
        the default constructor (20.0f, occupancy_grid.cpp:9-16,138-175); fromLCM sets 1.0f/mpc (:99-108)."""
        cells = np.ascontiguousarray(cells, dtype=np.int8)
        hgt, wid = cells.shape
        mpc = np.float32(metersPerCell)
        cpm = np.float32(cellsPerMeter) if cellsPerMeter is not None else np.float32(1.0) / mpc
        g = cls(ctx=ctx, _raw=(wid, hgt, mpc, cpm, np.float32(origin[0]), np.float32(origin[1])))
        g.upload(cells)
        return g

    # accessors (occupancy_grid.hpp:80-91)
    def widthInCells(self): return self.width
    def heightInCells(self): return self.height
    def metersPerCell(self): return self.mpc
    def cellsPerMeter(self): return self.cpm
    def originInGlobalFrame(self): return self.origin
    def isCellInGrid(self, x, y): return 0 <= x < self.width and 0 <= y < self.height

    def upload(self, cells):
        cells = np.ascontiguousarray(cells, dtype=np.int8)
        assert cells.shape == (self.height, self.width)
        check(self.ctx.lib.bl_grid_upload(self.h, cells.ctypes.data))

    def cells(self):
        out = np.empty((self.height, self.width), dtype=np.int8)
        check(self.ctx.lib.bl_grid_download(self.h, out.ctypes.data))
        return out

    def reset(self):
        check(self.ctx.lib.bl_grid_reset(self.h))

    def logOdds(self, x, y):
        """Host read of one cell (downloads the grid; test convenience only)."""
        return int(self.cells()[y, x]) if self.isCellInGrid(x, y) else 0

    def saveToFile(self, filename):
        """ASCII .map format (occupancy_grid.cpp:111-136)."""
        c = self.cells()
        with open(filename, "w") as f:
            f.write(f"{_fmt(self.origin[0])} {_fmt(self.origin[1])} {self.width} {self.height} {_fmt(self.mpc)}\n")
            for y in range(self.height):
                f.write(" ".join(str(int(v)) for v in c[y]) + " \n")
        return True

    @classmethod
    def loadFromFile(cls, filename, ctx=None):
        with open(filename) as f:
            tok = f.read().split()
        ox, oy, w, h, mpc = np.float32(tok[0]), np.float32(tok[1]), int(tok[2]), int(tok[3]), np.float32(tok[4])
        cells = np.array(tok[5:5 + w * h], dtype=np.int64).astype(np.int8).reshape(h, w)
        return cls.from_cells(cells, (ox, oy), mpc, cellsPerMeter=np.float32(1.0 / np.float64(np.float32(0.05))), ctx=ctx)

    def close(self):
        if self.h:
            self.ctx.lib.bl_grid_destroy(self.h)
            self.h = None


def _fmt(v):
    return np.format_float_positional(np.float32(v), precision=6, unique=True, trim="-")


class Mapping:
    """Mapping(maxLaserDistance, hitOdds, missOdds).updateMap(scan, pose, map) (mapping.hpp:25-34)."""

    def __init__(self, maxLaserDistance, hitOdds, missOdds, ctx=None):
        self.ctx = ctx or default_context()
        h = C.c_void_p()
        check(self.ctx.lib.bl_mapping_create(self.ctx.h, np.float32(maxLaserDistance), int(hitOdds), int(missOdds), C.byref(h)))
        self.h = h

    def updateMap(self, scan, pose, grid):
        ls = scan.as_c()
        check(self.ctx.lib.bl_mapping_update(self.h, C.byref(ls), C.byref(pose), grid.h))

    def updateMapDevicePose(self, scan, d_pose_ptr, pose_utime, grid):
        ls = scan.as_c()
        check(self.ctx.lib.bl_mapping_update_dev_pose(self.h, C.byref(ls), d_pose_ptr, int(pose_utime), grid.h))

    def updateMapFinishingFilter(self, scan, pf, pose_utime, grid):
        """pf.updateEnd(want_pose=False) + updateMapDevicePose(scan, pf.poseDevicePtr(), ...) in ONE launch: the end of the
        filter update (pose estimate + weight prefix) rides in the map kernel (bl_mapping_update_finishing_pf)."""
        ls = scan.as_c()
        check(self.ctx.lib.bl_mapping_update_finishing_pf(self.h, C.byref(ls), pf.h, int(pose_utime), grid.h))

    def close(self):
        if self.h:
            self.ctx.lib.bl_mapping_destroy(self.h)
            self.h = None


class ParticleFilter:
    """ParticleFilter(numParticles) (particle_filter.hpp:38-77).  shard=(lo, hi) keeps only those particles' private
    state on this device (botlab_amd.sharded drives the exchange)."""

    def __init__(self, numParticles, ctx=None, shard=None):
        assert numParticles > 1                                         # particle_filter.cpp:11
        self.ctx = ctx or default_context()
        self.N = int(numParticles)
        self.lo, self.hi = shard if shard else (0, self.N)
        h = C.c_void_p()
        check(self.ctx.lib.bl_pf_create(self.ctx.h, self.N, self.lo, self.hi, C.byref(h)))
        self.h = h

    def initializeFilterAtPose(self, pose, seed=None):
        if seed is None:                                                # reference: std::random_device
            seed = int.from_bytes(np.random.bytes(8), "little")
        check(self.ctx.lib.bl_pf_init_at_pose(self.h, C.byref(pose), C.c_uint64(seed)))

    def setParticles(self, particles, units=None):
        """particles: structured array (PARTICLE_DTYPE) of all N particles."""
        p = np.ascontiguousarray(particles)
        assert p.dtype.itemsize == 56 and p.size == self.N
        u = None
        if units is not None:
            u = np.ascontiguousarray(units, dtype=np.uint32)
            assert u.size == self.N
        check(self.ctx.lib.bl_pf_set_particles(self.h, p.ctypes.data, u.ctypes.data if u is not None else None))

    def setNoiseSeed(self, seed):
        check(self.ctx.lib.bl_pf_set_noise_seed(self.h, C.c_uint64(seed)))

    def updateFilter(self, odometry, laser, grid, rand_value=None, noise=None, want_pose=True):
        """pose_xyt_t updateFilter(odometry, laser, map).  rand_value defaults to libc rand(), as the reference's
        resampler calls it (particle_filter.cpp:92); noise (3*N float32) replaces the Philox action noise."""
        if rand_value is None:
            rand_value = _libc.rand()
        ls = laser.as_c()
        nz = None
        if noise is not None:
            nz = np.ascontiguousarray(noise, dtype=np.float32)
            assert nz.size == 3 * self.N
        out = Pose()
        check(self.ctx.lib.bl_pf_update(self.h, C.byref(odometry), C.byref(ls), grid.h, int(rand_value),
                                        nz.ctypes.data if nz is not None else None, C.byref(out) if want_pose else None))
        return out if want_pose else None

    def updateBegin(self, odometry, laser, grid, rand_value, noise=None):
        ls = laser.as_c()
        nz = None
        if noise is not None:
            nz = np.ascontiguousarray(noise, dtype=np.float32)
        moved = C.c_int()
        check(self.ctx.lib.bl_pf_update_begin(self.h, C.byref(odometry), C.byref(ls), grid.h, int(rand_value),
                                              nz.ctypes.data if nz is not None else None, C.byref(moved)))
        return bool(moved.value)

    def updateEnd(self, want_pose=True):
        out = Pose()
        check(self.ctx.lib.bl_pf_update_end(self.h, C.byref(out) if want_pose else None))
        return out if want_pose else None

    def updateFilterActionOnly(self, odometry, noise=None):
        nz = None
        if noise is not None:
            nz = np.ascontiguousarray(noise, dtype=np.float32)
        out = Pose()
        check(self.ctx.lib.bl_pf_update_action_only(self.h, C.byref(odometry), nz.ctypes.data if nz is not None else None,
                                                    C.byref(out)))
        return out

    def poseEstimate(self):
        out = Pose()
        check(self.ctx.lib.bl_pf_pose_estimate(self.h, C.byref(out)))
        return out

    def poseDevicePtr(self):
        return self.ctx.lib.bl_pf_pose_device_ptr(self.h)

    def estimatePosteriorPose(self):
        """estimatePosteriorPose(posterior_) of the particles as they stand (particle_filter.cpp:144-160)."""
        out = Pose()
        check(self.ctx.lib.bl_pf_estimate_posterior_pose(self.h, C.byref(out)))
        return out

    def setStrictResampling(self, on=True):
        """Resample against the reference's own sequentially rounded cumulative weight (bl_pf_set_strict_resampling)."""
        check(self.ctx.lib.bl_pf_set_strict_resampling(self.h, 1 if on else 0))

    def debugResample(self, rand_value):
        """Source index of every output particle of resamplePosteriorDistribution for this rand() value (particle_filter.cpp:84-103)."""
        idx = np.empty(self.N, np.int32)
        check(self.ctx.lib.bl_pf_debug_resample(self.h, int(rand_value), idx.ctypes.data))
        return idx

    def debugUniformRuns(self):
        """Runs of the equal-weight cumulative in force for the next resampling (0: the weights are not known to be equal)."""
        n = C.c_int()
        check(self.ctx.lib.bl_pf_debug_uniform_runs(self.h, C.byref(n)))
        return n.value

    def debugEstimateStats(self):
        """Per axis (x, then y): generic replays, their phases, table replays, gaps walked the slow way."""
        out = np.zeros(8, np.uint32)
        check(self.ctx.lib.bl_pf_debug_estimate_stats(self.h, out.ctypes.data))
        return [int(v) for v in out]

    def particles(self):
        """particles_t.particles of the local shard as a structured array."""
        out = np.zeros(self.hi - self.lo, dtype=PARTICLE_DTYPE)
        check(self.ctx.lib.bl_pf_get_particles(self.h, out.ctypes.data))
        return out

    def debugEnable(self, on=True):
        check(self.ctx.lib.bl_pf_debug_enable(self.h, 1 if on else 0))

    def debugLast(self):
        n = self.hi - self.lo
        idx = np.empty(n, np.int32)
        like = np.empty(n, np.int32)
        check(self.ctx.lib.bl_pf_debug_last(self.h, idx.ctypes.data, like.ctypes.data))
        return idx, like

    def close(self):
        if self.h:
            self.ctx.lib.bl_pf_destroy(self.h)
            self.h = None


class ObstacleDistanceGrid:
    """ObstacleDistanceGrid().setDistances(map); operator()(x, y) (obstacle_distance_grid.hpp:28-96)."""

    def __init__(self, ctx=None):
        self.ctx = ctx or default_context()
        h = C.c_void_p()
        check(self.ctx.lib.bl_dist_create(self.ctx.h, C.byref(h)))
        self.h = h
        self._host = None

    def setDistances(self, grid):
        check(self.ctx.lib.bl_dist_set_distances(self.h, grid.h))
        self._host = None

    def forget(self):
        """The next setDistances transforms the whole map (bl_dist_forget)."""
        check(self.ctx.lib.bl_dist_forget(self.h))

    def stats(self):
        """How the transforms of this grid went out (bl_dist_debug_stats)."""
        v = (C.c_int64 * 6)()
        check(self.ctx.lib.bl_dist_debug_stats(self.h, v))
        return dict(incremental=v[0], full=v[1], unchanged=v[2], nothing=v[3], window=v[4], fallback=v[5])

    def bound(self):
        """(formed, D): the bound the next incremental transform dilates its window by (bl_dist_debug_bound)."""
        f, b = C.c_int(), C.c_uint()
        check(self.ctx.lib.bl_dist_debug_bound(self.h, C.byref(f), C.byref(b)))
        return bool(f.value), int(b.value)

    def fusedStats(self):
        """(gave_up, helped) of the one-launch whole-grid transform (bl_dist_debug_fused)."""
        v = (C.c_int64 * 2)()
        check(self.ctx.lib.bl_dist_debug_fused(self.h, v))
        return int(v[0]), int(v[1])

    def shape(self):
        w, h = C.c_int(), C.c_int()
        check(self.ctx.lib.bl_dist_shape(self.h, C.byref(w), C.byref(h)))
        return w.value, h.value

    def widthInCells(self): return self.shape()[0]
    def heightInCells(self): return self.shape()[1]

    def frame(self):
        v = [C.c_float() for _ in range(4)]
        check(self.ctx.lib.bl_dist_frame(self.h, *[C.byref(x) for x in v]))
        return tuple(np.float32(x.value) for x in v)      # mpc, cpm, ox, oy

    def cells(self):
        if self._host is None:
            w, h = self.shape()
            out = np.empty((h, w), dtype=np.float32)
            check(self.ctx.lib.bl_dist_download(self.h, out.ctypes.data))
            self._host = out
        return self._host

    def isCellInGrid(self, x, y):
        w, h = self.shape()
        return 0 <= x < w and 0 <= y < h

    def __call__(self, x, y):
        return self.cells()[y, x]

    def close(self):
        if self.h:
            self.ctx.lib.bl_dist_destroy(self.h)
            self.h = None


def search_for_path(start, goal, distances, params, return_stats=False, cap=1 << 20):
    """robot_path_t search_for_path(start, goal, distances, params) (astar.hpp:58-61).  Returns the list of poses
    (path_length == len(result); length 1 == no path)."""
    ctx = distances.ctx
    buf = (Pose * cap)()
    n = C.c_int()
    stats = (C.c_int64 * 2)()
    check(ctx.lib.bl_astar_search(ctx.h, distances.h, C.byref(start), C.byref(goal), C.byref(params), buf, cap, C.byref(n), stats))
    path = [Pose(p.utime, p.x, p.y, p.theta) for p in buf[:min(n.value, cap)]]
    if return_stats:
        return path, (stats[0], stats[1])
    return path


def search_for_path_begin(goal, distances, params, start=None, start_dev=None):
    """Asynchronous form: enqueue the search (start pose from the host, or read on the device from start_dev, e.g.
    ParticleFilter.poseDevicePtr()); fetch it with search_for_path_end."""
    ctx = distances.ctx
    if start_dev is not None:
        check(ctx.lib.bl_astar_search_async_dev_start(ctx.h, distances.h, start_dev, C.byref(goal), C.byref(params)))
    else:
        check(ctx.lib.bl_astar_search_async(ctx.h, distances.h, C.byref(start), C.byref(goal), C.byref(params)))


_path_bufs = {}


def search_for_path_end(distances, cap=4097, return_stats=False):
    ctx = distances.ctx
    buf = _path_bufs.get(cap)
    if buf is None:
        buf = _path_bufs[cap] = (Pose * cap)()
    n = C.c_int()
    stats = (C.c_int64 * 2)()
    check(ctx.lib.bl_astar_search_result(ctx.h, buf, cap, C.byref(n), stats))
    if n.value > cap:
        raise _capi.BotlabHipError(f"path of {n.value} poses does not fit the {cap}-pose buffer")
    path = [Pose(p.utime, p.x, p.y, p.theta) for p in buf[:n.value]]
    return (path, (stats[0], stats[1])) if return_stats else path


def search_for_path_batch(start, goals, distances, params, cap_each=4097, return_stats=False):
    """n independent search_for_path calls from one start, run concurrently on the device (bl_astar_search_batch).
    Returns a list of paths (each a list of poses)."""
    ctx = distances.ctx
    n = len(goals)
    g = (Pose * max(n, 1))(*goals)
    buf = (Pose * (max(n, 1) * cap_each))()
    lens = (C.c_int * max(n, 1))()
    stats = (C.c_int64 * (2 * max(n, 1)))()
    check(ctx.lib.bl_astar_search_batch(ctx.h, distances.h, C.byref(start), g, n, C.byref(params), buf, cap_each, lens, stats))
    paths = []
    for i in range(n):
        if lens[i] > cap_each:
            raise _capi.BotlabHipError(f"path {i} of {lens[i]} poses does not fit the {cap_each}-pose buffer")
        paths.append([Pose(p.utime, p.x, p.y, p.theta) for p in buf[i * cap_each:i * cap_each + lens[i]]])
    if return_stats:
        return paths, [(stats[2 * i], stats[2 * i + 1]) for i in range(n)]
    return paths


class Frontiers:
    """std::vector<frontier_t> (frontiers.hpp:17-20) as a library handle; .cells() gives the list of (n, 2) float32 arrays."""

    def __init__(self, ctx, h):
        self.ctx, self.h = ctx, h

    @classmethod
    def from_lists(cls, ctx, frontiers):
        offs = np.zeros(len(frontiers) + 1, np.int32)
        for k, f in enumerate(frontiers):
            offs[k + 1] = offs[k] + len(f)
        xy = np.ascontiguousarray(np.concatenate(frontiers) if len(frontiers) else np.zeros((0, 2)), dtype=np.float32)
        h = C.c_void_p()
        check(ctx.lib.bl_frontiers_from_host(offs.ctypes.data, len(frontiers), xy.ctypes.data, C.byref(h)))
        return cls(ctx, h)

    def __len__(self):
        return self.ctx.lib.bl_frontiers_count(self.h)

    def cells(self):
        n, tot = len(self), self.ctx.lib.bl_frontiers_total_cells(self.h)
        offs = np.zeros(n + 1, np.int32)
        xy = np.zeros((max(tot, 1), 2), np.float32)
        check(self.ctx.lib.bl_frontiers_get(self.h, offs.ctypes.data, xy.ctypes.data))
        return [xy[offs[k]:offs[k + 1]].copy() for k in range(n)]

    def stats(self):
        a, b = C.c_int(), C.c_int()
        check(self.ctx.lib.bl_frontiers_stats(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def sweep_kernel(self):
        """which kernels grew these frontiers (bl_frontiers_debug_sweep_kernel): 0 / 1 one workgroup (small / large grid),
        2 k_frontier_grow, 3 k_frontier_grow2"""
        return int(self.ctx.lib.bl_frontiers_debug_sweep_kernel(self.h))

    def close(self):
        if self.h:
            self.ctx.lib.bl_frontiers_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def find_map_frontiers(grid, robotPose, minFrontierLength=0.35):
    """find_map_frontiers (frontiers.hpp:34-36)."""
    h = C.c_void_p()
    check(grid.ctx.lib.bl_frontiers_find(grid.ctx.h, grid.h, C.byref(robotPose), float(minFrontierLength), C.byref(h)))
    return Frontiers(grid.ctx, h)


def plan_path_to_frontier(frontiers, robotPose, grid, planner, cap=1 << 16, return_info=False):
    """plan_path_to_frontier (frontiers.hpp:50-53).  `grid` is unused by the reference too; the planner's distance grid is."""
    ctx = planner.distances_.ctx
    if not isinstance(frontiers, Frontiers):
        frontiers = Frontiers.from_lists(ctx, frontiers)
    st = _capi.MotionPlannerState(planner.params_.robotRadius, planner.searchParams_, planner.num_frontiers, planner.prev_goal)
    buf = (Pose * cap)()
    n = C.c_int()
    goal = Pose()
    stats = (C.c_int64 * 3)()
    check(ctx.lib.bl_plan_path_to_frontier(ctx.h, frontiers.h, C.byref(robotPose), planner.distances_.h, C.byref(st), buf, cap,
                                           C.byref(n), C.byref(goal), stats))
    if n.value > cap:
        raise _capi.BotlabHipError(f"path of {n.value} poses does not fit the {cap}-pose buffer")
    path = [Pose(p.utime, p.x, p.y, p.theta) for p in buf[:n.value]]
    if return_info:
        return path, goal, (stats[0], stats[1], stats[2])
    return path


# exploration_status_t (lcmtypes/exploration_status_t.lcm:3-11)
STATE_INITIALIZING, STATE_EXPLORING_MAP, STATE_RETURNING_HOME, STATE_COMPLETED_EXPLORATION, STATE_FAILED_EXPLORATION = 0, 1, 2, 3, 4
STATUS_IN_PROGRESS, STATUS_COMPLETE, STATUS_FAILED = 0, 1, 2


class ExploringMap:
    """Exploration::executeExploringMap (src/planning/exploration.cpp:277-369), the per-map step of the exploration loop,
    without its LCM calls: setMap (distance transform), find_map_frontiers, and -- when the robot is within 0.5 m of the
    current target or has none -- plan_path_to_frontier; then the status / next-state rule of :332-368.  The reference
    leaves status.status unset when frontiers remain but no path was found (:344-347 is commented out) and so falls into the
    default branch of :365-367: FAILED_EXPLORATION (definition D10; `status` then reports STATUS_FAILED)."""

    def __init__(self, planner):
        self.planner_ = planner
        self.currentTarget_ = make_pose(0.0, 0.0, 0.0)
        self.currentPath_ = []
        self.frontiers_ = None
        self.status = None

    def execute(self, currentMap, currentPose):
        self.planner_.setMap(currentMap)                                             # :299
        self.frontiers_ = find_map_frontiers(currentMap, currentPose)                # :300
        lists = self.frontiers_.cells()
        self.planner_.setNumFrontiers(len(lists))                                    # :302
        t = self.currentTarget_
        if t.x != 0 or t.y != 0:                                                      # :307-311: double pow/sqrt, stored to a float
            dx = float(np.float32(currentPose.x) - np.float32(t.x))
            dy = float(np.float32(currentPose.y) - np.float32(t.y))
            currDist = float(np.float32(np.sqrt(dx * dx + dy * dy)))
        else:
            currDist = 0.0
        if currDist <= float(np.float32(0.5)) and len(lists) > 0:                     # :316-321
            self.currentPath_ = plan_path_to_frontier(self.frontiers_, currentPose, currentMap, self.planner_)
            if len(self.currentPath_) > 1:
                p = self.currentPath_[-1]
                self.currentTarget_ = Pose(p.utime, p.x, p.y, p.theta)
        if len(lists) == 0:                                                           # :335-347
            self.status = STATUS_COMPLETE
        elif len(self.currentPath_) > 1:
            self.status = STATUS_IN_PROGRESS
        else:
            self.status = STATUS_FAILED                                               # D10
        return {STATUS_IN_PROGRESS: STATE_EXPLORING_MAP, STATUS_COMPLETE: STATE_RETURNING_HOME,
                STATUS_FAILED: STATE_FAILED_EXPLORATION}[self.status]                # :352-368


class AsyncExplorer:
    """bl_explorer: ExploringMap's step (exploration.cpp:277-369) on side streams -- submit(map, device pose) snapshots both on
    the SLAM stream; a lane runs setMap + find_map_frontiers against the snapshot; fetch() hands the steps back in order and
    applies the 0.5 m re-planning rule with the state they share, running plan_path_to_frontier on that lane when it is due."""

    def __init__(self, ctx=None, lanes=1, robotRadius=0.2):
        self.ctx = ctx or default_context()
        self.lanes = int(lanes)
        h = C.c_void_p()
        check(self.ctx.lib.bl_explorer_create(self.ctx.h, self.lanes, float(robotRadius), C.byref(h)))
        self.h = h
        self._buf = (Pose * 65536)()

    def submit(self, grid, pose_dev):
        check(self.ctx.lib.bl_explorer_submit(self.h, grid.h, pose_dev))

    def pending(self):
        return self.ctx.lib.bl_explorer_pending(self.h)

    def fetch(self, want_path=True):
        """-> (bl_explore_result_t as _capi.ExploreResult, currentPath_ as a list of poses or None)"""
        r = _capi.ExploreResult()
        check(self.ctx.lib.bl_explorer_fetch(self.h, C.byref(r), self._buf, 65536 if want_path else 0))
        path = [Pose(p.utime, p.x, p.y, p.theta) for p in self._buf[:min(r.path_length, 65536)]] if want_path else None
        return r, path

    def frontiers(self):
        h = C.c_void_p()
        check(self.ctx.lib.bl_explorer_frontiers(self.h, C.byref(h)))
        return Frontiers(self.ctx, h)

    def setState(self, target=None, prev_goal=None):
        check(self.ctx.lib.bl_explorer_set_state(self.h, C.byref(target) if target is not None else None,
                                                 C.byref(prev_goal) if prev_goal is not None else None))

    def close(self):
        if self.h:
            self.ctx.lib.bl_explorer_destroy(self.h)
            self.h = None


class AsyncPlanner:
    """bl_planner: MotionPlanner.setMap + planPath run on a second stream against a snapshot of the map and of the
    device-resident pose (the reference's planner process, src/planning/exploration.cpp:300-317)."""

    def __init__(self, ctx=None, params=None, lanes=1, batch=1):
        """lanes side streams; each collects `batch` submissions and searches them in one launch (bl_planner_create_batched)."""
        self.ctx = ctx or default_context()
        self.params_ = params or MotionPlannerParams()
        self.searchParams_ = SearchParams(self.params_.robotRadius, 10.0 * self.params_.robotRadius, 1.0)   # motion_planner.cpp:105-110
        self.lanes = int(lanes)
        h = C.c_void_p()
        self.batch = int(batch)
        check(self.ctx.lib.bl_planner_create_batched(self.ctx.h, self.lanes, self.batch, C.byref(h)))
        self.h = h
        self._buf = (Pose * 4097)()

    def submit(self, grid, start_dev, goal):
        check(self.ctx.lib.bl_planner_submit(self.h, grid.h, start_dev, C.byref(goal), C.byref(self.searchParams_)))

    def submit_with_map_update(self, mapping, scan, pose_dev, pose_utime, grid, goal):
        """mapping.updateMapDevicePose(scan, pose_dev, pose_utime, grid) then submit(grid, pose_dev, goal) as one library call
        (the map kernel leaves the snapshot behind on grids up to 256 K cells)."""
        c = scan.as_c()
        check(self.ctx.lib.bl_planner_submit_with_map_update(self.h, mapping.h, C.byref(c), pose_dev, int(pose_utime), grid.h,
                                                             C.byref(goal), C.byref(self.searchParams_)))

    def submit_with_map_update_finishing(self, mapping, scan, pf, pose_utime, grid, goal):
        """submit_with_map_update whose map kernel also ends the filter update begun with pf.updateBegin()."""
        c = scan.as_c()
        check(self.ctx.lib.bl_planner_submit_with_map_update_finishing_pf(self.h, mapping.h, C.byref(c), pf.h, int(pose_utime),
                                                                          grid.h, C.byref(goal), C.byref(self.searchParams_)))

    def fetch(self, return_stats=False):
        n = C.c_int()
        stats = (C.c_int64 * 2)()
        check(self.ctx.lib.bl_planner_fetch(self.h, self._buf, 4097, C.byref(n), stats))
        path = [Pose(p.utime, p.x, p.y, p.theta) for p in self._buf[:min(n.value, 4097)]]
        return (path, (stats[0], stats[1])) if return_stats else path

    def flush(self):
        """End of input: the batch every lane is still collecting goes out as it is (bl_planner_flush)."""
        check(self.ctx.lib.bl_planner_flush(self.h))

    def timing(self, on=-1):
        d, a, n = C.c_double(), C.c_double(), C.c_int64()
        check(self.ctx.lib.bl_planner_timing(self.h, on, C.byref(d), C.byref(a), C.byref(n)))
        return d.value, a.value, n.value

    def close(self):
        if self.h:
            self.ctx.lib.bl_planner_destroy(self.h)
            self.h = None


class MotionPlannerParams:
    def __init__(self, robotRadius=0.2):                                 # motion_planner.hpp:27-35
        self.robotRadius = float(robotRadius)


class MotionPlanner:
    """MotionPlanner (motion_planner.cpp:9-110): setMap, planPath, isValidGoal, setParams quirks included."""

    def __init__(self, params=None, ctx=None):
        self.params_ = params or MotionPlannerParams()
        self.distances_ = ObstacleDistanceGrid(ctx=ctx)
        self.searchParams_ = SearchParams()
        self.num_frontiers = 1           # uninitialised in the reference (motion_planner.hpp:164); callers set it
        self.prev_goal = make_pose(1e9, 1e9, 0.0)
        self.setParams(self.params_)

    def setParams(self, params):
        # motion_planner.cpp:105-110 reads params_ (the constructor's copy), not the argument
        self.searchParams_.minDistanceToObstacle = self.params_.robotRadius
        self.searchParams_.maxDistanceWithCost = 10.0 * self.searchParams_.minDistanceToObstacle
        self.searchParams_.distanceCostExponent = 1.0

    def setMap(self, grid):
        self.distances_.setDistances(grid)

    def setPrevGoal(self, goal): self.prev_goal = goal
    def setNumFrontiers(self, n): self.num_frontiers = int(n)

    def isValidGoal(self, goal):
        # motion_planner.cpp:52-74
        dx = np.float32(goal.x) - np.float32(self.prev_goal.x)
        dy = np.float32(goal.y) - np.float32(self.prev_goal.y)
        dist_prev = np.sqrt(np.float32(dx * dx + dy * dy), dtype=np.float32)
        if self.num_frontiers != 1 and float(dist_prev) < 2 * self.searchParams_.minDistanceToObstacle:
            return False
        mpc, cpm, ox, oy = self.distances_.frame()
        gx = int((float(np.float32(goal.x)) - float(ox)) * float(cpm))
        gy = int((float(np.float32(goal.y)) - float(oy)) * float(cpm))
        if self.distances_.isCellInGrid(gx, gy):
            return float(self.distances_(gx, gy)) > self.params_.robotRadius
        return False

    def planPath(self, start, goal, searchParams=None, return_stats=False):
        if not self.isValidGoal(goal):
            failed = [Pose(start.utime, start.x, start.y, start.theta)]   # failedPath (motion_planner.cpp:28-40)
            return (failed, (0, 0)) if return_stats else failed
        return search_for_path(start, goal, self.distances_, searchParams or self.searchParams_, return_stats=return_stats)

    def isPathSafe(self, path):
        # motion_planner.cpp:77-96 (one gather for all poses); a pose outside the grid is unsafe
        mpc = np.float32(self.distances_.frame()[0])
        w, h = self.distances_.shape()
        q = np.zeros((len(path), 2), np.int32)
        for i, p in enumerate(path):
            q[i, 0] = int(np.float32(np.float32(p.x) / mpc) + np.float32(w // 2))
            q[i, 1] = int(np.float32(np.float32(p.y) / mpc) + np.float32(h // 2))
        out = np.zeros(len(path), np.float32)
        check(self.distances_.ctx.lib.bl_dist_gather(self.distances_.h, q.ctypes.data, len(path), out.ctypes.data))
        return bool(np.all(out > self.searchParams_.minDistanceToObstacle))
