"""Synthetic world / lidar / odometry generator for tests and bench.py (host side, numpy).

Semantics follow the reference's simulator, which cannot be imported here (pygame/lcm missing) and never travels:
  * lidar: R = 290 rays per scan, 8 m max range, ray-marched at half a cell, first occupied cell stops the ray
    (src/sim/lidar.py:23-26,106-138); thetas[i] = 2*pi*i/R, beam direction = pose.theta - thetas[i]
    (src/slam/moving_laser_scan.cpp:33); times end at the scan time.
  * trajectory: 1 m square at 0.02 m / 0.01 rad per step style motion (src/mbot/drive_square.cpp:24-56).
  * odometry: truth + small Gaussian drift (src/sim/sim.py:188-193).
Inputs to the hot path only -- nothing here is part of the measured or parity-checked computation.
"""
import numpy as np

from .host import LidarScan

RAYS = 290
MAX_RANGE = 8.0
RAY_DT_US = 345


def tile_world(cells, size):
    """Deterministic size x size truth world: the occupied cells (> 0) of `cells` tiled, 1-cell solid border
    (SURVEY.md section 8d)."""
    h, w = cells.shape
    reps = (size + h - 1) // h, (size + w - 1) // w
    occ = np.tile((cells > 0), reps)[:size, :size]
    world = np.where(occ, 127, -127).astype(np.int8)
    world[0, :] = 127
    world[-1, :] = 127
    world[:, 0] = 127
    world[:, -1] = 127
    return world


def raycast_scan(truth, origin, mpc, pose_begin, pose_end, t_end_us, rays=RAYS, max_range=MAX_RANGE, noise_sigma=0.0, rng=None):
    """One lidar_t: ray i is cast from the pose interpolated between pose_begin and pose_end."""
    h, w = truth.shape
    i = np.arange(rays)
    frac = (i + 1) / rays
    px = pose_begin[0] + (pose_end[0] - pose_begin[0]) * frac
    py = pose_begin[1] + (pose_end[1] - pose_begin[1]) * frac
    dth = np.arctan2(np.sin(pose_end[2] - pose_begin[2]), np.cos(pose_end[2] - pose_begin[2]))
    pth = pose_begin[2] + dth * frac
    thetas = (2.0 * np.pi * i / rays).astype(np.float32)
    ang = pth - thetas
    step = mpc / 2.0
    nsteps = int(max_range / step)
    k = np.arange(1, nsteps + 1) * step                       # (nsteps,)
    xs = px[:, None] + np.cos(ang)[:, None] * k[None, :]
    ys = py[:, None] + np.sin(ang)[:, None] * k[None, :]
    cx = np.floor((xs - origin[0]) / mpc).astype(np.int64)
    cy = np.floor((ys - origin[1]) / mpc).astype(np.int64)
    inside = (cx >= 0) & (cx < w) & (cy >= 0) & (cy < h)
    occ = np.zeros_like(inside)
    occ[inside] = truth[cy[inside], cx[inside]] > 0
    stop = occ | ~inside
    first = np.where(stop.any(axis=1), stop.argmax(axis=1), nsteps - 1)
    ranges = k[first]
    hit_inside = inside[np.arange(rays), first] & occ[np.arange(rays), first]
    ranges = np.where(hit_inside, ranges, max_range)
    if noise_sigma > 0.0:
        ranges = ranges + rng.normal(0.0, noise_sigma, size=rays)
    times = (t_end_us - (rays - 1 - i) * RAY_DT_US).astype(np.int64)
    return LidarScan(ranges.astype(np.float32), thetas, times, utime=int(t_end_us))


def square_trajectory(start, steps, step_len=0.02, turn=0.01 * 5, side=1.0):
    """Truth poses (x, y, theta) of a robot driving a square of `side` metres: straight segments of step_len per step,
    in-place turns of `turn` rad per step."""
    poses = [np.array(start, dtype=np.float64)]
    x, y, th = start
    travelled, turning, turned = 0.0, False, 0.0
    for _ in range(steps):
        if not turning:
            x += step_len * np.cos(th)
            y += step_len * np.sin(th)
            travelled += step_len
            if travelled >= side - 1e-9:
                turning, turned, travelled = True, 0.0, 0.0
        else:
            th += turn
            turned += turn
            if turned >= np.pi / 2 - 1e-9:
                turning = False
        poses.append(np.array([x, y, np.arctan2(np.sin(th), np.cos(th))]))
    return poses


def odometry_from_truth(poses, rng, sigma_trans=1e-3, sigma_rot=3e-3):
    """Odometry = truth motion + per-step Gaussian drift, accumulated in the odometry frame."""
    odo = [np.array(poses[0], dtype=np.float64)]
    for a, b in zip(poses[:-1], poses[1:]):
        d = b - a
        dist = np.hypot(d[0], d[1])
        head = np.arctan2(d[1], d[0]) - a[2] if dist > 1e-12 else 0.0
        dth = np.arctan2(np.sin(d[2]), np.cos(d[2]))
        dist_n = dist + (rng.normal(0, sigma_trans) if dist > 0 else 0.0)
        dth_n = dth + rng.normal(0, sigma_rot)
        p = odo[-1]
        nx = p[0] + dist_n * np.cos(p[2] + head)
        ny = p[1] + dist_n * np.sin(p[2] + head)
        nth = np.arctan2(np.sin(p[2] + dth_n), np.cos(p[2] + dth_n))
        odo.append(np.array([nx, ny, nth]))
    return odo


def raycast_scans_gpu(truth, origin, mpc, poses, t0_us, dt_us, ctx, rays=RAYS, max_range=MAX_RANGE, noise_sigma=0.0, rng=None):
    """raycast_scan for every consecutive pose pair of `poses` at once: the beams of all scans are marched by the simulator's
    lidar kernel (botlab_amd.sim.SimLidar.cast, half-cell steps as src/sim/lidar.py:106-138) in one launch.  Scan k ends at
    t0_us + k * dt_us.  Inputs to the hot path only."""
    from .sim import SimLidar, _clamp
    n = len(poses) - 1
    i = np.arange(rays)
    frac = (i + 1) / rays
    pb = np.array(poses[:-1], dtype=np.float64)
    pe = np.array(poses[1:], dtype=np.float64)
    px = pb[:, None, 0] + (pe[:, None, 0] - pb[:, None, 0]) * frac[None, :]
    py = pb[:, None, 1] + (pe[:, None, 1] - pb[:, None, 1]) * frac[None, :]
    dth = np.arctan2(np.sin(pe[:, 2] - pb[:, 2]), np.cos(pe[:, 2] - pb[:, 2]))
    pth = pb[:, None, 2] + dth[:, None] * frac[None, :]
    thetas = (2.0 * np.pi * i / rays).astype(np.float32)
    ang = pth - thetas[None, :]
    lidar = SimLidar(truth, float(origin[0]), float(origin[1]), float(mpc), ctx=ctx, num_ranges=rays, max_distance=max_range)
    ranges = lidar.cast(px.ravel(), py.ravel(), _clamp(ang.ravel())).reshape(n, rays)
    ranges = np.minimum(ranges, max_range)
    if noise_sigma > 0.0:
        ranges = ranges + rng.normal(0.0, noise_sigma, size=ranges.shape)
    scans = []
    for k in range(n):
        t_end = t0_us + (k + 1) * dt_us
        times = (t_end - (rays - 1 - i) * RAY_DT_US).astype(np.int64)
        scans.append(LidarScan(ranges[k].astype(np.float32), thetas, times, utime=int(t_end)))
    return scans
