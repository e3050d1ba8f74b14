"""Simulator lidar on the GPU: the scan loop of src/sim/lidar.py:74-104 (noise off) with the
beam march of lidar.py:106-138 / map.py:80-87 run by libbotlab_hip.so for all beams of all requested scans in one launch.
The host side mirrors the reference's Python: beam angles accumulate (theta += step), beam times run backwards from the
scan time, pose.theta - theta goes through geometry.clamp."""
import ctypes as C
import math

import numpy as np

from . import host
from ._capi import check


def _clamp(a):
    """geometry.clamp (src/sim/geometry.py:5-10) element-wise, same operations in the same order."""
    a = np.array(a, dtype=np.float64)
    while True:
        m = a > math.pi
        if not m.any():
            break
        a[m] -= 2 * math.pi
    while True:
        m = a <= -math.pi
        if not m.any():
            break
        a[m] += 2 * math.pi
    return a


class SimLidar:
    def __init__(self, world_cells, origin_x, origin_y, meters_per_cell, ctx=None, num_ranges=290, max_distance=8, scan_rate=10):
        self.ctx = ctx or host.default_context()
        cells = np.ascontiguousarray(world_cells, dtype=np.int8)
        self.ox, self.oy, self.mpc = float(origin_x), float(origin_y), float(meters_per_cell)
        # the grid object is only the device storage of the truth cells; the simulator's own (double) frame is passed per call
        self.grid = host.OccupancyGrid.from_cells(cells, (np.float32(self.ox), np.float32(self.oy)), np.float32(self.mpc), ctx=self.ctx)
        self.num_ranges, self.max_distance, self.scan_rate = int(num_ranges), max_distance, scan_rate

    def cast(self, x, y, angle):
        """Lidar._beam_scan for beams starting at (x[i], y[i]) along angle[i] (already clamped): distances, float64."""
        x = np.ascontiguousarray(x, dtype=np.float64); y = np.ascontiguousarray(y, dtype=np.float64)
        angle = np.ascontiguousarray(angle, dtype=np.float64)
        out = np.zeros(x.size, np.float64)
        check(self.ctx.lib.bl_sim_cast_beams(self.ctx.h, self.grid.h, self.ox, self.oy, self.mpc, x.ctypes.data, y.ctypes.data,
                                             angle.ctypes.data, int(x.size), float(self.max_distance), out.ctypes.data))
        return out

    def scans(self, pose_at, nows):
        """One lidar scan per entry of `nows` (seconds): pose_at(t) -> (x, y, theta) is evaluated per beam, as Lidar.scan does
        through get_current_pose(at_time).  Returns a list of (thetas, ranges, times) with float64 thetas / ranges and
        integer microsecond times, all beams marched in one launch."""
        R = self.num_ranges
        theta_step = 2 * math.pi / R
        beam_period = 1 / (R * self.scan_rate)
        xs, ys, angs, metas = [], [], [], []
        for now in nows:
            theta = 0
            thetas, times = [], []
            for _ in range(R):
                px, py, pth = pose_at(now)
                thetas.append(theta)
                times.append(int(1e6 * now))
                xs.append(px); ys.append(py); angs.append(pth - theta)
                now -= beam_period
                theta += theta_step
            metas.append((thetas, times))
        ranges = self.cast(xs, ys, _clamp(angs))
        return [(np.array(th), ranges[k * R:(k + 1) * R].copy(), np.array(tm, dtype=np.int64)) for k, (th, tm) in enumerate(metas)]

    def scan_message(self, pose_at, now):
        """One scan as the lidar_t the simulator publishes (float32 ranges / thetas: lcmtypes/lidar_t.lcm)."""
        th, rg, tm = self.scans(pose_at, [now])[0]
        return host.LidarScan(rg.astype(np.float32), th.astype(np.float32), tm, utime=int(tm[0]))
