"""ctypes binding of libbotlab_hip.so (include/botlab_hip.h).  Loading fails loudly when the HIP library has not
been built: there is no CPU fallback anywhere in this package."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BOTLAB_HIP_LIB") or os.path.join(_HERE, "libbotlab_hip.so")   # BOTLAB_HIP_LIB: a diagnostic build (probes)


class Pose(C.Structure):
    """lcmtypes/pose_xyt_t.lcm (24 bytes)."""
    _fields_ = [("utime", C.c_int64), ("x", C.c_float), ("y", C.c_float), ("theta", C.c_float)]

    def __repr__(self):
        return f"Pose(utime={self.utime}, x={self.x!r}, y={self.y!r}, theta={self.theta!r})"


class Particle(C.Structure):
    """lcmtypes/particle_t.lcm (56 bytes)."""
    _fields_ = [("pose", Pose), ("parent_pose", Pose), ("weight", C.c_double)]


class Lidar(C.Structure):
    """lcmtypes/lidar_t.lcm, arrays as host pointers."""
    _fields_ = [("utime", C.c_int64), ("num_ranges", C.c_int32), ("ranges", C.POINTER(C.c_float)),
                ("thetas", C.POINTER(C.c_float)), ("times", C.POINTER(C.c_int64)),
                ("intensities", C.POINTER(C.c_float))]


class SearchParams(C.Structure):
    """src/planning/astar.hpp:15-27."""
    _fields_ = [("minDistanceToObstacle", C.c_double), ("maxDistanceWithCost", C.c_double),
                ("distanceCostExponent", C.c_double)]


assert C.sizeof(Pose) == 24 and C.sizeof(Particle) == 56

class MotionPlannerState(C.Structure):
    """bl_motion_planner_t: the MotionPlanner members plan_path_to_frontier reads (motion_planner.hpp:153-165)."""
    _fields_ = [("robot_radius", C.c_double), ("search", SearchParams), ("num_frontiers", C.c_int32), ("prev_goal", Pose)]


class ExploreResult(C.Structure):
    """bl_explore_result_t: one Exploration::executeExploringMap step (exploration.cpp:277-369)."""
    _fields_ = [("next_state", C.c_int32), ("status", C.c_int32), ("num_frontiers", C.c_int32), ("frontier_cells", C.c_int32),
                ("planned", C.c_int32), ("path_length", C.c_int32), ("pops", C.c_int64), ("pushes", C.c_int64), ("searches", C.c_int64),
                ("bfs_cells", C.c_int32), ("bfs_levels", C.c_int32), ("pose", Pose), ("target", Pose), ("frontiers_ms", C.c_float),
                ("plan_ms", C.c_float)]


BL_K_MCL_MAIN, BL_K_MCL_SCAN, BL_K_MAP, BL_K_DIST, BL_K_ASTAR, BL_K_FRONTIERS = range(6)
BL_K_DIST_ROWS, BL_K_DIST_COLS_SUMMARY, BL_K_DIST_COLS_APPLY, BL_K_SNAPSHOT, BL_K_DIST_FUSED = range(6, 11)
BL_OK, BL_ERR_HIP, BL_ERR_ARG, BL_ERR_CAPACITY, BL_ERR_STATE = range(5)

_vp = C.c_void_p
_P = C.POINTER

# name -> (restype, argtypes); every symbol include/botlab_hip.h declares
SIGNATURES = {
    "bl_last_error": (C.c_char_p, []),
    "bl_version": (C.c_char_p, []),
    "bl_ctx_create": (C.c_int, [C.c_int, _vp, _P(_vp)]),
    "bl_ctx_destroy": (None, [_vp]),
    "bl_ctx_sync": (C.c_int, [_vp]),
    "bl_ctx_timing_enable": (C.c_int, [_vp, C.c_int]),
    "bl_ctx_timing_stride": (C.c_int, [_vp, C.c_int]),
    "bl_ctx_timing_get": (C.c_int, [_vp, C.c_int, _P(C.c_double), _P(C.c_int64)]),
    "bl_ctx_timing_reset": (C.c_int, [_vp]),
    "bl_grid_create": (C.c_int, [_vp, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, _P(_vp)]),
    "bl_grid_destroy": (None, [_vp]),
    "bl_grid_upload": (C.c_int, [_vp, _vp]),
    "bl_grid_download": (C.c_int, [_vp, _vp]),
    "bl_grid_reset": (C.c_int, [_vp]),
    "bl_grid_set_frame": (C.c_int, [_vp, C.c_float, C.c_float, C.c_float, C.c_float]),
    "bl_grid_copy": (C.c_int, [_vp, _vp]),
    "bl_grid_device_ptr": (_vp, [_vp]),
    "bl_grid_shape": (C.c_int, [_vp, _P(C.c_int), _P(C.c_int)]),
    "bl_mapping_create": (C.c_int, [_vp, C.c_float, C.c_int8, C.c_int8, _P(_vp)]),
    "bl_mapping_destroy": (None, [_vp]),
    "bl_mapping_update": (C.c_int, [_vp, _P(Lidar), _P(Pose), _vp]),
    "bl_mapping_update_dev_pose": (C.c_int, [_vp, _P(Lidar), _vp, C.c_int64, _vp]),
    "bl_pf_create": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _P(_vp)]),
    "bl_pf_destroy": (None, [_vp]),
    "bl_pf_set_exchange_buffers": (C.c_int, [_vp, _vp, _vp]),
    "bl_pf_exchange_rec_ptr": (_vp, [_vp]),
    "bl_pf_init_at_pose": (C.c_int, [_vp, _P(Pose), C.c_uint64]),
    "bl_pf_set_particles": (C.c_int, [_vp, _vp, _vp]),
    "bl_pf_get_particles": (C.c_int, [_vp, _vp]),
    "bl_pf_set_noise_seed": (C.c_int, [_vp, C.c_uint64]),
    "bl_pf_update": (C.c_int, [_vp, _P(Pose), _P(Lidar), _vp, C.c_int, _vp, _P(Pose)]),
    "bl_pf_update_begin": (C.c_int, [_vp, _P(Pose), _P(Lidar), _vp, C.c_int, _vp, _P(C.c_int)]),
    "bl_pf_update_end": (C.c_int, [_vp, _P(Pose)]),
    "bl_pf_update_action_only": (C.c_int, [_vp, _P(Pose), _vp, _P(Pose)]),
    "bl_pf_pose_estimate": (C.c_int, [_vp, _P(Pose)]),
    "bl_pf_pose_device_ptr": (_vp, [_vp]),
    "bl_pf_estimate_posterior_pose": (C.c_int, [_vp, _vp]),
    "bl_pf_debug_estimate_stats": (C.c_int, [_vp, _vp]),
    "bl_pf_debug_set_finish_generation": (C.c_int, [_vp, C.c_uint32]),
    "bl_pf_set_strict_resampling": (C.c_int, [_vp, C.c_int]),
    "bl_pf_debug_resample": (C.c_int, [_vp, C.c_int, _vp]),
    "bl_pf_debug_enable": (C.c_int, [_vp, C.c_int]),
    "bl_pf_debug_last": (C.c_int, [_vp, _vp, _vp]),
    "bl_pf_debug_uniform_runs": (C.c_int, [_vp, _P(C.c_int)]),
    "bl_debug_trig_probe": (C.c_int, [_vp, _P(C.c_float), _P(C.c_float), _P(C.c_float), _P(C.c_uint64)]),
    "bl_debug_trig_addition_probe": (C.c_int, [_vp, C.c_uint64, C.c_uint32, _P(C.c_float), _P(C.c_float), _P(C.c_float), _P(C.c_uint64)]),
    "bl_dist_create": (C.c_int, [_vp, _P(_vp)]),
    "bl_dist_destroy": (None, [_vp]),
    "bl_dist_set_distances": (C.c_int, [_vp, _vp]),
    "bl_dist_download": (C.c_int, [_vp, _vp]),
    "bl_dist_debug_stats": (C.c_int, [_vp, _vp]),
    "bl_dist_forget": (C.c_int, [_vp]),
    "bl_dist_debug_bound": (C.c_int, [_vp, _vp, _vp]),
    "bl_dist_debug_fused": (C.c_int, [_vp, _vp]),
    "bl_dist_shape": (C.c_int, [_vp, _P(C.c_int), _P(C.c_int)]),
    "bl_dist_frame": (C.c_int, [_vp, _P(C.c_float), _P(C.c_float), _P(C.c_float), _P(C.c_float)]),
    "bl_dist_device_ptr": (_vp, [_vp]),
    "bl_astar_search": (C.c_int, [_vp, _vp, _P(Pose), _P(Pose), _P(SearchParams), _vp, C.c_int, _P(C.c_int),
                                  _P(C.c_int64)]),
    "bl_astar_set_open_capacity": (C.c_int, [_vp, C.c_int64]),
    "bl_astar_debug_last_kernel": (C.c_int, [_vp]),
    "bl_debug_heap2_replay": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int64, _vp, _vp, _P(C.c_int), _vp]),
    "bl_astar_search_async": (C.c_int, [_vp, _vp, _P(Pose), _P(Pose), _P(SearchParams)]),
    "bl_astar_search_async_dev_start": (C.c_int, [_vp, _vp, _vp, _P(Pose), _P(SearchParams)]),
    "bl_astar_search_result": (C.c_int, [_vp, _vp, C.c_int, _P(C.c_int), _P(C.c_int64)]),
    "bl_planner_create": (C.c_int, [_vp, C.c_int, _P(_vp)]),
    "bl_planner_create_batched": (C.c_int, [_vp, C.c_int, C.c_int, _P(_vp)]),
    "bl_planner_destroy": (None, [_vp]),
    "bl_planner_submit": (C.c_int, [_vp, _vp, _vp, _P(Pose), _P(SearchParams)]),
    "bl_planner_fetch": (C.c_int, [_vp, _vp, C.c_int, _P(C.c_int), _P(C.c_int64)]),
    "bl_planner_flush": (C.c_int, [_vp]),
    "bl_planner_timing": (C.c_int, [_vp, C.c_int, _P(C.c_double), _P(C.c_double), _P(C.c_int64)]),
    "bl_planner_submit_with_map_update": (C.c_int, [_vp, _vp, _P(Lidar), _vp, C.c_int64, _vp, _P(Pose), _P(SearchParams)]),
    "bl_mapping_update_finishing_pf": (C.c_int, [_vp, _P(Lidar), _vp, C.c_int64, _vp]),
    "bl_scan_prefetch": (C.c_int, [_vp, _P(Lidar)]),
    "bl_comm_load": (C.c_int, [C.c_char_p]),
    "bl_comm_unique_id": (C.c_int, [C.c_char_p, C.c_char_p]),
    "bl_comm_create": (C.c_int, [_vp, C.c_char_p, C.c_char_p, C.c_int, C.c_int, _P(_vp)]),
    "bl_comm_destroy": (None, [_vp]),
    "bl_comm_all_gather_inplace": (C.c_int, [_vp, _vp, C.c_size_t]),
    "bl_dev_enable_peer_access": (C.c_int, [C.c_int, C.c_int]),
    "bl_dev_alloc": (C.c_int, [_vp, C.c_size_t, _P(_vp)]),
    "bl_dev_word": (C.c_int, [_vp, _vp, C.c_int, _P(C.c_uint32)]),
    "bl_dev_free": (C.c_int, [_vp]),
    "bl_ipc_export": (C.c_int, [_vp, C.c_char_p]),
    "bl_ipc_open": (C.c_int, [C.c_char_p, _P(_vp)]),
    "bl_ipc_close": (C.c_int, [_vp]),
    "bl_pf_shard_setup": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int]),
    "bl_pf_shard_local_ptrs": (C.c_int, [_vp, _P(_vp), _P(_vp), _P(_vp)]),
    "bl_pf_shard_set_peer": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp]),
    "bl_pf_shard_commit": (C.c_int, [_vp]),
    "bl_pf_shard_buffers": (C.c_int, [_vp, _P(_vp), _P(C.c_size_t), _P(_vp), _P(C.c_size_t)]),
    "bl_pf_shard_stage": (C.c_int, [_vp, C.c_int]),
    "bl_pf_shard_exchange": (C.c_int, [_vp, _vp]),
    "bl_pf_shard_traffic": (C.c_int, [_vp, _vp]),
    "bl_pf_shard_local_ptrs_peer": (C.c_int, [_vp, _P(_vp), _P(_vp), _P(_vp)]),
    "bl_pf_shard_set_peer_buffers": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp]),
    "bl_pf_shard_peer_commit": (C.c_int, [_vp]),
    "bl_pf_shard_peer_active": (C.c_int, [_vp]),
    "bl_pf_shard_peer_selftest": (C.c_int, [_vp, _P(C.c_int)]),
    "bl_pf_shard_peer_reset": (C.c_int, [_vp, C.c_int]),
    "bl_pf_shard_exchange_peer": (C.c_int, [_vp]),
    "bl_pf_shard_exchange_peer_phase": (C.c_int, [_vp, C.c_int]),
    "bl_planner_submit_with_map_update_finishing_pf": (C.c_int, [_vp, _vp, _P(Lidar), _vp, C.c_int64, _vp, _P(Pose), _P(SearchParams)]),
    "bl_astar_search_batch": (C.c_int, [_vp, _vp, _P(Pose), _vp, C.c_int, _P(SearchParams), _vp, C.c_int, _vp, _vp]),
    "bl_dist_gather": (C.c_int, [_vp, _vp, C.c_int, _vp]),
    "bl_frontiers_find": (C.c_int, [_vp, _vp, _P(Pose), C.c_double, _P(_vp)]),
    "bl_frontiers_from_host": (C.c_int, [_vp, C.c_int, _vp, _P(_vp)]),
    "bl_frontiers_count": (C.c_int, [_vp]),
    "bl_frontiers_total_cells": (C.c_int, [_vp]),
    "bl_frontiers_get": (C.c_int, [_vp, _vp, _vp]),
    "bl_frontiers_stats": (C.c_int, [_vp, _P(C.c_int), _P(C.c_int)]),
    "bl_frontiers_debug_sweep_kernel": (C.c_int, [_vp]),
    "bl_frontiers_destroy": (None, [_vp]),
    "bl_lcm_fingerprint": (C.c_uint64, [C.c_int]),
    "bl_lcm_encode_pose": (C.c_int64, [C.c_int, _P(Pose), _vp, C.c_int64]),
    "bl_lcm_encode_lidar": (C.c_int64, [_P(Lidar), _vp, _vp, C.c_int64]),
    "bl_lcm_encode_particles": (C.c_int64, [C.c_int64, _vp, C.c_int32, _vp, C.c_int64]),
    "bl_lcm_encode_grid": (C.c_int64, [C.c_int64, C.c_float, C.c_float, C.c_float, C.c_int32, C.c_int32, _vp, _vp, C.c_int64]),
    "bl_lcm_encode_path": (C.c_int64, [C.c_int64, _vp, C.c_int32, _vp, C.c_int64]),
    "bl_lcm_decode_pose": (C.c_int, [C.c_int, _vp, C.c_int64, _P(Pose)]),
    "bl_lcm_decode_lidar": (C.c_int, [_vp, C.c_int64, _P(C.c_int64), _P(C.c_int32), _vp, _vp, _vp, _vp, C.c_int32]),
    "bl_lcm_decode_particles": (C.c_int, [_vp, C.c_int64, _P(C.c_int64), _P(C.c_int32), _vp, C.c_int32]),
    "bl_lcm_decode_grid": (C.c_int, [_vp, C.c_int64, _P(C.c_int64), _vp, _vp, _vp, C.c_int64]),
    "bl_lcm_decode_path": (C.c_int, [_vp, C.c_int64, _P(C.c_int64), _P(C.c_int32), _vp, C.c_int32]),
    "bl_lcm_log_event_size": (C.c_int64, [C.c_int32, C.c_int32]),
    "bl_lcm_log_write_event": (C.c_int64, [C.c_int64, C.c_int64, C.c_char_p, _vp, C.c_int32, _vp, C.c_int64]),
    "bl_lcm_log_read_event": (C.c_int64, [_vp, C.c_int64, _P(C.c_int64), _P(C.c_int64), _P(C.c_int64), _P(C.c_int32), _P(C.c_int64),
                                          _P(C.c_int32)]),
    "bl_pf_encode_particles_lcm": (C.c_int64, [_vp, C.c_int64, _vp, C.c_int64]),
    "bl_grid_encode_lcm": (C.c_int64, [_vp, C.c_int64, _vp, C.c_int64]),
    "bl_sim_cast_beams": (C.c_int, [_vp, _vp, C.c_double, C.c_double, C.c_double, _vp, _vp, _vp, C.c_int, C.c_double, _vp]),
    "bl_explorer_create": (C.c_int, [_vp, C.c_int, C.c_double, _P(_vp)]),
    "bl_explorer_destroy": (None, [_vp]),
    "bl_explorer_set_state": (C.c_int, [_vp, _P(Pose), _P(Pose)]),
    "bl_explorer_submit": (C.c_int, [_vp, _vp, _vp]),
    "bl_explorer_pending": (C.c_int, [_vp]),
    "bl_explorer_fetch": (C.c_int, [_vp, _P(ExploreResult), _vp, C.c_int]),
    "bl_explorer_frontiers": (C.c_int, [_vp, _P(_vp)]),
    "bl_plan_path_to_frontier": (C.c_int, [_vp, _vp, _P(Pose), _vp, _P(MotionPlannerState), _vp, C.c_int, _P(C.c_int), _P(Pose),
                                           _vp]),
}

_lib = None


class BotlabHipError(RuntimeError):
    pass


def load():
    """Returns the loaded library with argtypes set; raises if the extension is missing (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BotlabHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). botlab_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().bl_last_error()
        raise BotlabHipError(f"botlab_hip call failed (status {rc}): {msg.decode() if msg else ''}")
