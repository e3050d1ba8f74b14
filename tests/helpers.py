"""Shared test helpers: golden fixtures and synthetic inputs."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")

SLAM_MAPS = ["convex_10mx10m_5cm", "convex_10mx10m_5cm_offcenter", "drive_square_10mx10m_5cm", "obstacle_slam_10mx10m_5cm"]
ASTAR_MAPS = ["astar_convex", "astar_empty", "astar_filled", "astar_maze", "astar_narrow", "astar_wide"]
GEN_MAPS = ["empty", "filled", "narrow", "wide"]
ALL_MAPS = SLAM_MAPS + GEN_MAPS + ASTAR_MAPS

CPM_DEFAULT = np.float32(1.0 / np.float64(np.float32(0.05)))    # cellsPerMeter_ after OccupancyGrid() + loadFromFile


def load_reference_maps():
    z = np.load(os.path.join(GOLDEN, "reference_maps.npz"))
    out = {}
    for name in ALL_MAPS:
        out[name] = dict(cells=z[name + "__cells"], origin=tuple(z[name + "__origin"]), mpc=z[name + "__mpc"][0])
    return out


def load_astar_cases():
    with open(os.path.join(GOLDEN, "astar_cases.json")) as f:
        return json.load(f)
