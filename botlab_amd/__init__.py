"""botlab_amd -- MI355X (gfx950) implementation of botLab's occupancy-grid SLAM / MCL / A* hot path.

The product is libbotlab_hip.so (hand-written HIP kernels behind the C ABI in include/botlab_hip.h).  This package is
the Python host mirror of the reference's class surfaces used by tests and bench.py.  There is no CPU fallback: every
call raises if the HIP library is missing or no GPU is visible.
"""
from ._capi import BotlabHipError, Lidar, Particle, Pose, SearchParams, load  # noqa: F401
from .host import (AsyncExplorer, AsyncPlanner, Context, LidarScan, Mapping, MotionPlanner, MotionPlannerParams, ObstacleDistanceGrid,  # noqa: F401
                   OccupancyGrid, ParticleFilter, PARTICLE_DTYPE, POSE_DTYPE, default_context, make_pose,
                   search_for_path, search_for_path_begin, search_for_path_end, search_for_path_batch, Frontiers,
                   find_map_frontiers, plan_path_to_frontier, ExploringMap)

__all__ = ["AsyncExplorer", "AsyncPlanner", "BotlabHipError", "Lidar", "Particle", "Pose", "SearchParams", "load", "Context", "LidarScan", "Mapping",
           "MotionPlanner", "MotionPlannerParams", "ObstacleDistanceGrid", "OccupancyGrid", "ParticleFilter",
           "PARTICLE_DTYPE", "POSE_DTYPE", "default_context", "make_pose", "search_for_path", "search_for_path_begin",
           "search_for_path_end", "search_for_path_batch", "Frontiers", "find_map_frontiers", "plan_path_to_frontier", "ExploringMap"]
