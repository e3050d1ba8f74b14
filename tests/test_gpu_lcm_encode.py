"""Row f1, device part: particles_t and occupancy_grid_t encoded straight from device state must be byte-identical to the
ORACLE's encoding (oracle/lcm_codec.py, struct.pack + its own fingerprints) of particles() / the downloaded grid -- and, as a
second check, to the library's host codec."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import botlab_amd as bl
import helpers
from botlab_amd import _capi, synth

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import lcm_codec  # noqa: E402

pytestmark = pytest.mark.gpu


def _pose(p, pre):          # pre: "" for particle_t.pose, "p_" for parent_pose (host.PARTICLE_DTYPE field names)
    return {"utime": int(p[pre + "utime"]), "x": float(p[pre + "x"]), "y": float(p[pre + "y"]), "theta": float(p[pre + "theta"])}


def test_device_encoders_match_host_codec(maps, gpu_ctx):
    lib = gpu_ctx.lib
    m = maps["obstacle_slam_10mx10m_5cm"]
    truth = np.where(m["cells"] > 0, 127, -127).astype(np.int8)
    grid = bl.OccupancyGrid.from_cells(m["cells"], m["origin"], m["mpc"], cellsPerMeter=helpers.CPM_DEFAULT, ctx=gpu_ctx)
    N = 3001
    pf = bl.ParticleFilter(N, ctx=gpu_ctx)
    poses = synth.square_trajectory((-0.75, 0.2, 0.0), 3, step_len=0.02, turn=0.05, side=0.8)
    scans = [synth.raycast_scan(truth, m["origin"], 0.05, poses[k - 1], poses[k], 1_000_000 + k * 100_000) for k in range(1, 4)]
    pf.initializeFilterAtPose(bl.make_pose(-0.75, 0.2, 0.0, utime=int(scans[0].times[0])), seed=3)
    for k, sc in enumerate(scans):
        pf.updateFilter(bl.make_pose(*poses[k + 1], utime=sc.utime), sc, grid, rand_value=50 + k)
    parts = pf.particles()
    size = 20 + 48 * N
    dev = (C.c_uint8 * size)()
    assert lib.bl_pf_encode_particles_lcm(pf.h, 4242, dev, size) == size
    hostb = (C.c_uint8 * size)()
    assert lib.bl_lcm_encode_particles(4242, parts.ctypes.data, N, hostb, size) == size
    assert bytes(dev) == bytes(hostb)
    want = lcm_codec.encode("particles_t", {"utime": 4242, "num_particles": N, "particles": [
        {"pose": _pose(p, ""), "parent_pose": _pose(p, "p_"), "weight": float(p["weight"])} for p in parts]})
    assert bytes(dev) == want
    assert lib.bl_pf_encode_particles_lcm(pf.h, 4242, dev, size - 1) == -_capi.BL_ERR_CAPACITY

    cells = grid.cells()
    gsize = 8 + 8 + 12 + 12 + cells.size
    gdev, ghost = (C.c_uint8 * gsize)(), (C.c_uint8 * gsize)()
    assert lib.bl_grid_encode_lcm(grid.h, 17, gdev, gsize) == gsize
    assert lib.bl_lcm_encode_grid(17, m["origin"][0], m["origin"][1], m["mpc"], cells.shape[1], cells.shape[0], cells.ctypes.data, ghost, gsize) == gsize
    assert bytes(gdev) == bytes(ghost)
    want = lcm_codec.encode("occupancy_grid_t", {"utime": 17, "origin_x": float(m["origin"][0]), "origin_y": float(m["origin"][1]),
                                                  "meters_per_cell": float(m["mpc"]), "width": cells.shape[1], "height": cells.shape[0],
                                                  "num_cells": cells.size, "cells": cells.ravel().tolist()})
    assert bytes(gdev) == want
