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
struct bl_ctx;
int bl_ctx_create_low_priority(int device, bl_ctx** out);      // bl_ctx.hip: own stream of the lowest priority (long-running work)

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

// device-side scan block of a ctx, shared by the mapping and the MCL kernels: the kept rays of the last lidar_t handed to
// either of them.  A call that brings the same scan again (updateFilter then updateMap of one SLAM step) reuses it.
struct bl_scan_dev {
    int capacity = 0;        // rays allocated
    float* ranges = nullptr;
    float* thetas = nullptr;
    int64_t* times = nullptr;   // per-ray time stamps: interpolateRatio = (double)(t - begin) / (double)(end - begin) is formed in the kernels
    void* staging = nullptr;    // pinned host slots: times | ranges | thetas
    size_t staging_bytes = 0;
    int kept = 0;               // rays in the block
    float max_range = 0;        // largest kept range (bounds the cell offsets a ray can produce)
    bool thetas_simple = false; // every kept theta lies in [0, 6.2831]: a wrapped pose angle less a theta needs at most one upward 2*pi step
    // second device block + a packed scan waiting for it (bl_scan_prefetch): the next map kernel brings it over beside its own
    // work, and the blocks swap
    void* base = nullptr;       // the allocation both blocks live in
    float* alt_ranges = nullptr; float* alt_thetas = nullptr; int64_t* alt_times = nullptr;
    bool pre_pending = false; int pre_slot = 0, pre_kept = 0; float pre_max_range = 0; bool pre_thetas_simple = false;
};

struct bl_scan_prefetch_args {
    const int64_t* h_times; const float* h_ranges; const float* h_thetas; int kept;
    int64_t* d_times; float* d_ranges; float* d_thetas;
    unsigned long long* h_seq; unsigned long long seq;
};
// bl_ctx.hip: 1 and `out` filled if a packed scan waits for a kernel to carry it (the bookkeeping is done: the caller MUST
// launch the copy, behind everything that reads the current block); 0 otherwise.  Call it AFTER taking the current block's
// pointers for the launch that carries the copy.
int bl_scan_prefetch_take(struct bl_ctx* ctx, bl_scan_prefetch_args* out);

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
    bl_scan_dev scan;
};

// find_map_frontiers' result: frontier k = cells offsets[k]..offsets[k+1] of xy (x, y per cell, global metres)
struct bl_frontiers {
    std::vector<int32_t> offsets;
    std::vector<float> xy;
    int bfs_cells = 0, bfs_levels = 0;
    int sweep_kernel = 0;             // bl_frontiers_debug_sweep_kernel
};

// Zero-framed mirror of a grid too large to stage whole in LDS (rows -BL_MIRROR_FRAME..H+BL_MIRROR_FRAME-1, columns
// -4..stride-5, stride = ((W + 3) & ~3) + 8; zeros outside the grid): the image k_mcl_main gathers from and cuts its LDS window
// out of.  The particle filter builds it (k_mcl_frame) when it is not current; k_map_update then stores every cell it changes
// into both images, so that a SLAM loop pays for the copy once, not once per step.  Anything else that writes the cells
// (upload, reset, copy, a replanner snapshot) marks it stale, and a caller that has taken the raw device pointer may write
// behind the library's back: from then on the mirror is rebuilt before every use.
#define BL_MIRROR_FRAME 3

// Which cells the map updates of a grid's LINEAGE have touched, kept on the device so that nothing on the host has to wait for a
// kernel: Mapping::updateMap number v (1-based) of the lineage leaves the bounding box of the cells it may have changed in entry
// v % BL_DIRTY_LOG -- (x0 | y0 << 16, x1 | y1 << 16, v low, v high), x1 < x0 for an update that changed nothing.  A lineage is a
// sequence of cell states each of which comes from the one before by a map update; an upload, a reset or a copy from elsewhere
// starts a new one (new id).  The replanner's snapshots carry the id and version of the map they were copied from and share its
// log, and a distance grid remembers which (id, version) it last transformed: when the next map it is given is a later version
// of the same lineage, only the window those boxes can influence is transformed again (bl_planning.hip, "incremental").
#define BL_DIRTY_LOG 1024
struct bl_dirty_log { int refs; int4* dev; };
struct bl_grid {
    bl_ctx* ctx;
    bl_frame frame;
    int8_t* cells;      // device
    mutable int8_t* mirror;           // device, first byte of the framed image (or null)
    mutable size_t mirror_cap;
    mutable int mirror_stride;
    mutable bool mirror_valid;        // the mirror equals the cells as of the work enqueued so far
    bool mirror_external;             // bl_grid_device_ptr has been handed out
    uint64_t id;                      // lineage of the cells (0: none yet -- assigned on first need)
    uint64_t version;                 // map updates logged in this lineage
    bl_dirty_log* log;                // the lineage's log (shared with snapshots), or null before the first logged update
};
uint64_t bl_grid_new_lineage(bl_grid* g);                  // bl_ctx.hip: the cells were rewritten wholesale
uint64_t bl_grid_lineage_id(const bl_grid* g);             // the grid's lineage (assigned now if it has none yet)
void bl_grid_adopt_lineage(bl_grid* snap, const bl_grid* src);   // snap's cells are (about to be) a copy of src's
// a map update is about to be enqueued on g: its log entry (null when the log cannot be had) and the version it will carry
int4* bl_grid_log_next(bl_grid* g, uint64_t* version);

// bl_planning.hip: map + device pose snapshot on main's stream (whole grid, or the dirty cells when snap holds an earlier version)
int bl_snapshot_enqueue(struct bl_ctx* main, const bl_grid* map, bl_grid* snap, const void* d_pose, bl_pose_xyt_t* snap_pose,
                        unsigned int* done_count, unsigned long long* flag, unsigned long long seq);

// where a replanner submission wants its map + pose snapshot, and the number to publish in *flag when it is complete
struct bl_planner_snap {
    bl_grid* grid;                    // the snapshot grid itself (its lineage is set once the copy is enqueued)
    int8_t* cells; bl_pose_xyt_t* pose;
    unsigned long long* flag; unsigned long long seq;
    unsigned int* done_count;
};
int bl_planner_reserve(bl_planner* p, const bl_grid* map, bl_planner_snap* out);                    // bl_planning.hip
int bl_planner_commit(bl_planner* p, const bl_pose_xyt_t* goal, const bl_search_params_t* params);
void bl_planner_cancel(bl_planner* p);

// RAII-less helpers
int bl_timer_begin(bl_ctx* ctx, int id, hipEvent_t* a, hipEvent_t* b);
int bl_timer_end(bl_ctx* ctx, int id, hipEvent_t a, hipEvent_t b);
int bl_timer_pair(bl_ctx* ctx, int id, hipEvent_t* a, hipEvent_t* b);     // for launches that take start/stop events themselves
int bl_timer_commit(bl_ctx* ctx, int id, hipEvent_t a, hipEvent_t b);
int bl_scan_upload(bl_ctx* ctx, const bl_lidar_t* scan, int* num_rays);
void bl_scan_free(bl_ctx* ctx);

// interpolateRatio of interpolate_pose_by_time (src/common/interpolation.hpp:35) for a ray stamped t and the pose pair
// (begin, end), den = (double)(end - begin)
__host__ __device__ inline double bl_interp_ratio(int64_t t, int64_t begin, double den) { return (double)(t - begin) / den; }

#endif
