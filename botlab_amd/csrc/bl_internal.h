// bl_internal.h -- private structures of libbotlab_hip.so (not part of the ABI).
#ifndef BL_INTERNAL_H
#define BL_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/botlab_hip.h"
#include "bl_math.h"

void bl_set_error(const char* fmt, ...);

#define BL_HIP(call)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            bl_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return BL_ERR_HIP;                                                                    \
        }                                                                                         \
    } while (0)

#define BL_CHECK_ARG(cond)                                                            \
    do {                                                                              \
        if (!(cond)) {                                                                \
            bl_set_error("bad argument: %s (%s:%d)", #cond, __FILE__, __LINE__);      \
            return BL_ERR_ARG;                                                        \
        }                                                                             \
    } while (0)

struct bl_timer {
    double total_ms = 0;
    int64_t launches = 0;
    int64_t seen = 0;                       // launches offered to the timer (timed: every timing_stride-th)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;   // recorded, not yet read
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
};

// device-side scan description shared by the mapping and the MCL kernels (uploaded once per scan)
struct bl_scan_dev {
    int capacity = 0;        // rays allocated
    float* ranges = nullptr;
    float* thetas = nullptr;
    double* ratio = nullptr; // per-ray interpolation ratio for the current (begin, end) utime pair
    void* staging = nullptr; // pinned host staging: ranges | thetas | ratio
    size_t staging_bytes = 0;
};

struct bl_astar_state;
struct bl_frontier_scratch;

struct bl_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    bool timing = false;
    unsigned int timing_mask = 0xffffffffu;
    int timing_stride = 1;
    bl_timer timers[BL_K_COUNT];
    bl_astar_state* astar = nullptr;
    int64_t astar_capacity = 0;
    bool astar_small_lds = false;      // k_astar with the 40 KB LDS footprint (co-running searches)
    bl_frontier_scratch* frontier = nullptr;
};

// find_map_frontiers' result: frontier k = cells offsets[k]..offsets[k+1] of xy (x, y per cell, global metres)
struct bl_frontiers {
    std::vector<int32_t> offsets;
    std::vector<float> xy;
    int bfs_cells = 0, bfs_levels = 0;
};

struct bl_grid {
    bl_ctx* ctx;
    bl_frame frame;
    int8_t* cells;      // device
};

// RAII-less helpers
int bl_timer_begin(bl_ctx* ctx, int id, hipEvent_t* a, hipEvent_t* b);
int bl_timer_end(bl_ctx* ctx, int id, hipEvent_t a, hipEvent_t b);
int bl_scan_upload(bl_ctx* ctx, bl_scan_dev* sd, const bl_lidar_t* scan, int64_t begin_utime, int64_t end_utime,
                   int* num_rays);
void bl_scan_free(bl_scan_dev* sd);

#endif
