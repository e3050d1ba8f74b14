"""CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE) for row f4 of SURVEY.md section 8: the simulator's lidar.

A plain-Python restatement of src/sim/lidar.py:74-138 (Lidar.scan / _beam_scan / _beam_step_generator, noise off) and
src/sim/map.py:63-87 (Map.load_from_file's occupied set, Map.at_xy).  The reference module cannot be imported here (it
needs pygame and lcm), so this file follows it line by line; only tests/ may import it.

PARITY UNPINNED: the reference has no test for its simulator, and its trigonometry is numpy.cos / numpy.sin on Python
floats, whose last bit depends on the numpy build (SIMD kernels or libm).  This restatement -- and the product, on the
host -- use libm's cos / sin (math.cos / math.sin); everything after the trigonometry is exact double arithmetic.

Faithful quirks:
  * Map.at_xy has no bounds check: index = row * width + col is looked up in the set of occupied indices, so a column
    outside [0, width) aliases into a neighbouring row and anything outside [0, width*height) is simply free.  A beam
    that leaves the map keeps marching to max_distance.
  * dist, x and y are ACCUMULATED (x += dx ...), the test runs at dist = 0 first, and the loop condition is dist <= max.
  * beam times decrease along the scan (now -= beam_period) and thetas accumulate (theta += theta_step_size).
"""
import math


def clamp(rads):                                   # geometry.py:5-10
    while rads > math.pi:
        rads -= 2 * math.pi
    while rads <= -math.pi:
        rads += 2 * math.pi
    return rads


class Map:
    """The occupied set of map.py:63-78 kept as a flat truth array: occupied[index] = cell value > 0."""

    def __init__(self, cells, origin_x, origin_y, meters_per_cell):
        self.height, self.width = cells.shape
        self.occupied = (cells.reshape(-1) > 0)
        self.ox, self.oy, self.mpc = float(origin_x), float(origin_y), float(meters_per_cell)

    def at_xy(self, x, y):                          # map.py:80-87
        row = math.floor((y - self.oy) / self.mpc)
        col = math.floor((x - self.ox) / self.mpc)
        index = row * self.width + col
        return 0 <= index < self.occupied.size and bool(self.occupied[index])


def beam_scan(world, x, y, pose_theta, theta, max_distance=8):
    """Lidar._beam_scan (lidar.py:106-126), noise off: returns the measured distance."""
    ang = clamp(pose_theta - theta)                 # pose.theta -= theta through the clamping setter (geometry.py:24-26)
    step_size = world.mpc / 2                       # _beam_step_generator (lidar.py:128-138)
    dx = math.cos(ang) * step_size
    dy = math.sin(ang) * step_size
    dist = 0
    while dist <= max_distance:
        if world.at_xy(x, y):
            return dist
        x += dx
        y += dy
        dist += step_size
    return max_distance


def scan(world, pose_at, now, num_ranges=290, scan_rate=10, max_distance=8):
    """Lidar.scan's body (lidar.py:74-92), noise off: (thetas, ranges, times) as Python lists; pose_at(t) -> (x, y, theta)."""
    theta = 0
    theta_step_size = 2 * math.pi / num_ranges
    beam_period = 1 / (num_ranges * scan_rate)
    thetas, ranges, times = [], [], []
    for _ in range(num_ranges):
        px, py, pth = pose_at(now)
        thetas.append(theta)
        ranges.append(beam_scan(world, px, py, pth, theta, max_distance))
        times.append(int(1e6 * now))
        now -= beam_period
        theta += theta_step_size
    return thetas, ranges, times
