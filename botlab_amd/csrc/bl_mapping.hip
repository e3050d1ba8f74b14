// bl_mapping.hip -- Mapping::updateMap (src/slam/mapping.cpp:17-127) as one gfx950 workgroup.
//
// Reference semantics: pass 1 adds hitOdds (saturating at 127) at the end cell of every ray with range <= max laser
// distance; pass 2 walks the reference's Bresenham variant from the truncated start cell to the end cell (end cell
// excluded) subtracting missOdds (saturating at -128).  Because every pass-1 update precedes every pass-2 update and
// both saturate monotonically, a cell that is the end of H rays and is crossed by M rays ends at
//     v' = max(-128, min(127, v + hit*H) - miss*M)
// which is order-independent.  The kernel counts H per end cell and M per crossed cell in an LDS window (uint16
// counters; the ray whose increment finds an end cell's counter at zero becomes that cell's leader), walks the rays in
// independent 16-cell segments spread over all 1024 threads (closed form of the Bresenham variant), then applies the
// closed form with one owner thread per cell -- no global atomics, bit-exact int8 results.
//
// Launch shape: ONE workgroup of 1024 threads.  The whole update touches ~20 KB (290 rays x <=142 cells @5 cm / 5 m);
// it is latency-bound, not bandwidth-bound, and a single workgroup keeps every phase boundary a __syncthreads().
// The LDS window is the bounding box of all ray cells clipped to the grid; if it exceeds the LDS budget it is
// processed in horizontal strips (each strip re-walks the rays), so any resolution / laser range is handled.
#include <stdio.h>

#include "bl_internal.h"
#include "bl_mcl_finish.h"

#define MAP_THREADS 1024
#define MAP_LDS_COUNTERS (72 * 1024)         // uint16 counters -> 144 KB of the 160 KB LDS
#define MAP_MAX_RAYS 8192
#define MAP_CELL_LIMIT (1 << 24)             // |cell coordinate| bound for a ray to be traced (rejects NaN/inf geometry)

struct bl_mapping {
    bl_ctx* ctx;
    float max_laser;
    int hit, miss;
    bool initialized;
    int64_t prev_utime;
    bl_pose_xyt_t* d_prev;      // previousPose_ (device)
    int4* d_rays;               // scratch: (x0, y0, x1, y1) per ray
    int ray_capacity;
};

struct map_args {
    int8_t* cells;
    bl_frame frame;
    const float* ranges;
    const float* thetas;
    const int64_t* times;           // per-ray stamps; interpolateRatio = (t - t_begin) / t_den
    int64_t t_begin; double t_den;
    int R;
    bl_pose_xyt_t* prev;            // device previousPose_
    const bl_pose_xyt_t* cur_dev;   // device pose of this update, or null
    bl_pose_xyt_t cur_host;         // pose of this update when cur_dev is null
    int64_t cur_utime;              // utime to record with the pose
    int interp;                     // prev.utime != cur.utime
    int apply;                      // initialized_
    float max_laser;
    int hit, miss;
    int4* rays;
    long long* stamps;              // diagnostic build only (-DBL_MAP_STAMPS)
    int8_t* mirror; int mirror_stride;   // cell (0, 0) of the grid's zero-framed mirror when it is current (bl_internal.h), or null
    int4* dirty_entry; unsigned long long dirty_version;   // the lineage's log entry of this update (bl_internal.h), or null
    // optional tail: copy the updated grid and the pose to a replanner snapshot and publish its submission number
    // (bl_planner_submit_with_map_update: saves a dependent launch on the SLAM stream)
    int8_t* snap_cells; bl_pose_xyt_t* snap_pose; const bl_pose_xyt_t* snap_pose_src;
    unsigned long long* snap_flag; unsigned long long snap_seq;
    // optional: the end of the particle-filter update whose pose estimate this map update uses (bl_mcl_finish.h).
    // Workgroup 0 forms the estimate before it reads the pose; workgroups 1.. write the weight prefix meanwhile.
    int fin_on; mcl_finish_args fin;
    // optional: the next scan, packed in a pinned slot (bl_scan_prefetch), brought to the ctx's second device block by the last
    // workgroup of this launch
    int pre_on; bl_scan_prefetch_args pre;
};

#ifdef MCLF_STAMPS
#define XSTAMP(i) do { if (threadIdx.x == 0 && a.fin_on) a.fin.state->xstamps[i] = MCLF_NOW(); } while (0)
#else
#define XSTAMP(i) do { } while (0)
#endif
#ifdef BL_MAP_STAMPS
#define MSTAMP(i) do { if (threadIdx.x == 0) a.stamps[i] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define MSTAMP(i) do { } while (0)
#endif

// where cell number idx (row-major in the grid, row y) lies in the zero-framed mirror, relative to the mirror's cell (0, 0)
__device__ __forceinline__ size_t mirror_at(const map_args& a, size_t idx, int y)
{
    return idx + (size_t)y * (size_t)(a.mirror_stride - a.frame.width);
}

__device__ __forceinline__ bool cell_in_grid(const bl_frame& f, int x, int y)
{
    return x >= 0 && x < f.width && y >= 0 && y < f.height;
}

// Cell k (k = 0 .. K-1, K = max(dx, dy); start cell included, end cell excluded) of the reference's Bresenham walk
// (mapping.cpp:101-127: e2 = 2*err; if (e2 >= -dy) {err -= dy; x += sx;} if (e2 <= dx) {err += dx; y += sy;}) has a closed
// form: the major axis advances k, the minor axis floor((2*k*dmin + dmaj) / (2*dmaj)).  Checked against the loop for every
// dx, dy < 230 and all sign combinations (tests/tools/bresenham_closed_form.py), so the walk of one ray can be cut
// into independent segments.
#define MAP_SEG 16                            // cells per walk segment
#define MAP_SEG_RAYS 1024                     // rays whose segment table fits the static LDS array (more rays: serial walk per ray)

__device__ __forceinline__ unsigned int half_of(unsigned int pair, int ci) { return (ci & 1) ? (pair >> 16) : (pair & 0xffffu); }

// snap != nullptr: every store to the grid is mirrored into the replanner snapshot, whose bulk copy (the grid as it was when
// the kernel started) the caller holds in sv[] and map_update_body stores once its first loads are under way.
#define MAP_EARLY_VEC 4                       // int4 per thread held for the early snapshot copy: grids up to 64 KB
__device__ __forceinline__ void map_update_body(const map_args& a, int8_t* snap, const int4 (&sv)[MAP_EARLY_VEC], bl_pose_xyt_t* lds_pose, bool provisional);

// Workgroups of a launch that carries a filter's end: 0 the map update, 1 the pre-chain, 2 the finisher, 3.. the groups, last
// (optional) the scan prefetch.  The map update does not wait idle for the finisher's exact x, y: it forms the estimate's double
// reduction itself (same theta, x / y within a few 1e-6), runs everything up to the first grid store with that pose -- ray geometry,
// hit and miss counts in LDS: integer results of the pose -- and, when the exact x, y arrive, recomputes the rays' integer cells
// with them: equal (9 scans in 10 at 100k particles) means the counts are the reference's, otherwise the phases run again.
#define MAP_RIDER_WGS 3
__global__ __launch_bounds__(MAP_THREADS) void k_map_update(map_args a)
{
    __shared__ mclf_smem s_fin;
    // the finish that rides here belongs to a sharded particle set whose exchange gave up (k_shard_wait, bl_mcl.hip): no finish, and
    // no map store from a pose that was never formed -- the scan's way to the device (the last rider) still runs
    const bool broken = a.fin_on && a.fin.sh && __hip_atomic_load(&a.fin.state->shard_broken, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
    if (broken && !(a.pre_on && blockIdx.x == gridDim.x - 1 && blockIdx.x > 0)) return;
    if (blockIdx.x > 0) {                                       // riders
        if (a.pre_on && blockIdx.x == gridDim.x - 1) {
            for (int i = threadIdx.x; i < a.pre.kept; i += MAP_THREADS) {
                a.pre.d_times[i] = a.pre.h_times[i];
                a.pre.d_ranges[i] = a.pre.h_ranges[i];
                a.pre.d_thetas[i] = a.pre.h_thetas[i];
            }
            __syncthreads();                                    // every lane's loads from the slot have returned
            if (threadIdx.x == 0) __hip_atomic_store(a.pre.h_seq, a.pre.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        } else if (blockIdx.x == 1) {
            mclf_pre_chain(a.fin, s_fin);
        } else if (blockIdx.x == 2) {
            extern __shared__ __align__(16) unsigned int s_dyn_fin[];         // this workgroup's own dynamic LDS
            mclf_pose(a.fin, s_fin, (char*)s_dyn_fin, (size_t)MAP_LDS_COUNTERS * 2, true);
        } else {
            mclf_prefix_group(a.fin, (int)blockIdx.x - MAP_RIDER_WGS, s_fin);
        }
        return;
    }
    const size_t n = (size_t)a.frame.width * a.frame.height;
    const size_t n16 = n / 16;
    // Early snapshot (the usual case: a 200x200 grid, a 290-ray scan): the copy of the grid is loaded before anything else
    // and stored while the ray geometry is computed; the cells this update changes are then written to both.  Otherwise
    // the snapshot is a copy loop behind the update.
    const bool early = a.snap_cells != nullptr && a.apply && a.R <= MAP_THREADS && (a.frame.width & 3) == 0 &&
                       (n & 15) == 0 && n16 <= (size_t)MAP_EARLY_VEC * MAP_THREADS;
    int4 sv[MAP_EARLY_VEC];
#pragma unroll
    for (int u = 0; u < MAP_EARLY_VEC; ++u) sv[u] = make_int4(0, 0, 0, 0);
    if (early) {
#pragma unroll
        for (int u = 0; u < MAP_EARLY_VEC; ++u) {
            const size_t i = (size_t)u * MAP_THREADS + threadIdx.x;
            if (i < n16) sv[u] = ((const int4*)a.cells)[i];
        }
    }
    __shared__ bl_pose_xyt_t s_fin_pose;
    if (a.fin_on) {
        mclf_reduce_partials(a.fin, s_fin);
        __syncthreads();
        if (threadIdx.x == 0) {
            double tot[5];
            mclf_block_totals(s_fin, tot);
            s_fin_pose = mclf_approx_pose(tot, a.fin.utime);    // theta final, x / y provisional
        }
        __syncthreads();
    }
    // every return inside is uniform over the workgroup
    map_update_body(a, early ? a.snap_cells : nullptr, sv, a.fin_on ? &s_fin_pose : nullptr, a.fin_on != 0);
    if (a.snap_cells) {
        if (!early) {
            __syncthreads();                                    // the grid stores of this workgroup are visible to its own loads
            const int4* s4 = (const int4*)a.cells;
            int4* d4 = (int4*)a.snap_cells;
            for (size_t base = 0; base < n16; base += 4 * MAP_THREADS) {       // four loads in flight per thread, then the stores
                int4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const size_t i = base + (size_t)u * MAP_THREADS + threadIdx.x; if (i < n16) v[u] = s4[i]; }
#pragma unroll
                for (int u = 0; u < 4; ++u) { const size_t i = base + (size_t)u * MAP_THREADS + threadIdx.x; if (i < n16) d4[i] = v[u]; }
            }
            for (size_t i = n16 * 16 + threadIdx.x; i < n; i += MAP_THREADS) a.snap_cells[i] = a.cells[i];
        }
        // (a riding finish: the estimate was written by another workgroup of this launch -- take it from this one's LDS copy)
        if (threadIdx.x == 0) *a.snap_pose = a.fin_on ? s_fin_pose : *a.snap_pose_src;
        if (a.snap_flag) {                                              // flag hand-off only; an event hand-off needs nothing here
            __threadfence();
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_store(a.snap_flag, a.snap_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// start cell and end cell of one ray (moving_laser_scan.cpp:22-37, mapping.cpp:45-49); x == 0x7fffffff: the ray takes no part
__device__ __forceinline__ int4 map_ray_cells(const map_args& a, const bl_pose3& pb, const bl_pose3& pe, float range, float ray_theta, int64_t ray_time)
{
    int4 ray = make_int4(0x7fffffff, 0, 0, 0);
    if (range <= a.max_laser) {                                 // rays with range <= 0.15f were dropped on the host
        bl_pose3 rp = a.interp ? bl_interpolate_pose(pb, pe, bl_interp_ratio(ray_time, a.t_begin, a.t_den)) : pe;
        float theta = bl_wrap_to_pi(rp.theta - ray_theta);
        float sn, cs, sx, sy;
        bl_sincosf(theta, &sn, &cs);
        bl_global_to_grid(rp.x, rp.y, a.frame, &sx, &sy);
        float fx = (range * cs * a.frame.cpm) + sx;
        float fy = (range * sn * a.frame.cpm) + sy;
        const float lim = (float)MAP_CELL_LIMIT;
        if (fx > -lim && fx < lim && fy > -lim && fy < lim && sx > -lim && sx < lim && sy > -lim && sy < lim)
            ray = make_int4((int)sx, (int)sy, (int)fx, (int)fy);        // float -> int truncation at the bresenham() call
    }
    return ray;
}

// lds_pose: the pose of this update in shared memory (or null: a.cur_dev / a.cur_host).  provisional: its x, y are the map
// workgroup's own double reduction; the reference's x, y come from the finisher (mclf_wait_pose) -- everything up to the first
// grid store runs with the provisional pose, then the rays' integer cells are recomputed with the exact one and compared.
__device__ __forceinline__ void map_update_body(const map_args& a, int8_t* snap, const int4 (&sv)[MAP_EARLY_VEC], bl_pose_xyt_t* lds_pose, bool provisional)
{
    extern __shared__ __align__(16) unsigned int s_cnt[];     // MAP_LDS_COUNTERS/2 dwords, two uint16 counters each
    __shared__ int s_box[4];                                   // xmin, ymin, xmax, ymax over all traced cells
    __shared__ float s_pose[6];                                // prev x,y,theta ; cur x,y,theta
    __shared__ int s_segp[MAP_SEG_RAYS + 1];                   // exclusive prefix of the rays' segment counts
    __shared__ int s_wsum[MAP_THREADS / 64];
    __shared__ int s_redo;
    __shared__ int s_dirty[4];                                 // the box of cells this update may change (the last pass's)

    const int tid = threadIdx.x;
    MSTAMP(0);
    // the first ray of every thread is loaded before the barrier the pose arrives behind (the two round trips overlap)
    float pre_range = 0.0f, pre_theta = 0.0f;
    int64_t pre_time = 0;
    if (a.apply && tid < a.R) {
        pre_range = a.ranges[tid]; pre_theta = a.thetas[tid];
        if (a.interp) pre_time = a.times[tid];
    }
    // the exact pose into *lds_pose (one thread waits for the finisher), visible to the workgroup behind the barrier
    auto take_exact_pose = [&]() {
        if (tid == 0) { float x = lds_pose->x, y = lds_pose->y; (void)mclf_wait_pose(a.fin, &x, &y); lds_pose->x = x; lds_pose->y = y; }
        __syncthreads();
    };
    bool snap_stored = false;
    // One pass over everything.  Returns 1 when it ran with the provisional pose and must run again with the exact one (which
    // is in *lds_pose by then), 0 when the update is complete.
    auto run = [&](bool prov) -> int {
    if (tid == 0) {
        bl_pose_xyt_t cur = lds_pose ? *lds_pose : (a.cur_dev ? *a.cur_dev : a.cur_host);
        bl_pose_xyt_t prev = a.apply ? *a.prev : cur;          // mapping.cpp:19-21: first call uses pose for both
        s_pose[0] = prev.x; s_pose[1] = prev.y; s_pose[2] = prev.theta;
        s_pose[3] = cur.x; s_pose[4] = cur.y; s_pose[5] = cur.theta;
        s_box[0] = 0x7fffffff; s_box[1] = 0x7fffffff; s_box[2] = -0x7fffffff; s_box[3] = -0x7fffffff;
        s_redo = 0;
        s_dirty[0] = 1; s_dirty[1] = 1; s_dirty[2] = 0; s_dirty[3] = 0;      // empty
    }
    __syncthreads();
    if (!a.apply) { if (prov) take_exact_pose(); return 0; }    // increase/decreaseCellOdds do nothing (mapping.cpp:74,88)

    const bl_pose3 pb = {s_pose[0], s_pose[1], s_pose[2]};
    const bl_pose3 pe = {s_pose[3], s_pose[4], s_pose[5]};
    const bool seg_walk = a.R <= MAP_SEG_RAYS;                  // then thread tid owns ray tid in phases A and B

    MSTAMP(1);
    // ---- phase A: ray geometry (moving_laser_scan.cpp:22-37, mapping.cpp:45-49)
    int bx_lo = 0x7fffffff, by_lo = 0x7fffffff, bx_hi = -0x7fffffff, by_hi = -0x7fffffff;
    int4 my_ray = make_int4(0x7fffffff, 0, 0, 0);
    for (int r = tid; r < a.R; r += MAP_THREADS) {
        const bool first = r == tid;
        const int4 ray = map_ray_cells(a, pb, pe, first ? pre_range : a.ranges[r], first ? pre_theta : a.thetas[r],
                                       a.interp ? (first ? pre_time : a.times[r]) : 0);
        if (ray.x != 0x7fffffff) {
            bx_lo = min(bx_lo, min(ray.x, ray.z)); by_lo = min(by_lo, min(ray.y, ray.w));
            bx_hi = max(bx_hi, max(ray.x, ray.z)); by_hi = max(by_hi, max(ray.y, ray.w));
        }
        a.rays[r] = ray;
        my_ray = ray;
    }
    // one LDS atomic per wave and bound (290 lanes hammering four words serialised this phase)
    for (int off = 32; off > 0; off >>= 1) {
        bx_lo = min(bx_lo, __shfl_xor(bx_lo, off, 64)); by_lo = min(by_lo, __shfl_xor(by_lo, off, 64));
        bx_hi = max(bx_hi, __shfl_xor(bx_hi, off, 64)); by_hi = max(by_hi, __shfl_xor(by_hi, off, 64));
    }
    if ((tid & 63) == 0 && bx_lo != 0x7fffffff) {
        atomicMin(&s_box[0], bx_lo); atomicMin(&s_box[1], by_lo);
        atomicMax(&s_box[2], bx_hi); atomicMax(&s_box[3], by_hi);
    }
    // segment table: ray r contributes ceil(K / MAP_SEG) segments, K = max(dx, dy)
    if (seg_walk) {
        int segs = 0;
        if (tid < a.R && my_ray.x != 0x7fffffff) {
            const int K = max(abs(my_ray.z - my_ray.x), abs(my_ray.w - my_ray.y));
            segs = (K + MAP_SEG - 1) / MAP_SEG;
        }
        int incl = segs;
        for (int off = 1; off < 64; off <<= 1) {
            int t = __shfl_up(incl, off, 64);
            if ((tid & 63) >= off) incl += t;
        }
        if ((tid & 63) == 63) s_wsum[tid >> 6] = incl;
        __syncthreads();
        int base = 0;
        for (int w = 0; w < (tid >> 6); ++w) base += s_wsum[w];
        if (tid < a.R) s_segp[tid] = base + incl - segs;
        if (tid == a.R - 1) s_segp[a.R] = base + incl;
    }
    __syncthreads();

    if (snap && !snap_stored) {
        const size_t n16 = ((size_t)a.frame.width * a.frame.height) / 16;
#pragma unroll
        for (int u = 0; u < MAP_EARLY_VEC; ++u) {
            const size_t i = (size_t)u * MAP_THREADS + tid;
            if (i < n16) ((int4*)snap)[i] = sv[u];
        }
        __syncthreads();                                        // bulk copy before the mirrored stores of changed cells below
        snap_stored = true;
    }
    MSTAMP(2);
    // ---- the update runs through an LDS window of uint16 counters: the bounding box of all ray cells clipped to the grid,
    // in horizontal strips when it exceeds the LDS budget
    int bx0 = max(s_box[0], 0), by0 = max(s_box[1], 0);
    int bx1 = min(s_box[2], a.frame.width - 1), by1 = min(s_box[3], a.frame.height - 1);
    if (bx1 < bx0 || by1 < by0) {                               // nothing inside the grid (with this pose)
        if (prov) { take_exact_pose(); return 1; }
        return 0;
    }
    // grid rows of whole dwords: the window takes whole dwords too, and the free-space pass below updates four cells per access
    const bool dword_rows = (a.frame.width & 3) == 0 && seg_walk;
    if (dword_rows) { bx0 &= ~3; bx1 |= 3; }
    if (tid == 0) { s_dirty[0] = bx0; s_dirty[1] = by0; s_dirty[2] = bx1; s_dirty[3] = by1; }     // every store below lies inside
    const int ww = bx1 - bx0 + 1;
    const int wh = by1 - by0 + 1;
    int rows_per_strip = MAP_LDS_COUNTERS / ww;
    if (rows_per_strip < 1) rows_per_strip = 1;                 // ww > 73728 cannot happen for int32 grids < 2^31 cells with sane width
    if (rows_per_strip > wh) rows_per_strip = wh;
    const int strip_cells_max = rows_per_strip * ww;
    if (strip_cells_max > MAP_LDS_COUNTERS) { if (prov) take_exact_pose(); return 0; }      // defensive: never index past the LDS window
    // the look-ahead covers the usual shape only: one strip, a thread per ray
    if (prov && (rows_per_strip < wh || a.R > MAP_THREADS)) { take_exact_pose(); return 1; }

    for (int sy0 = by0; sy0 <= by1; sy0 += rows_per_strip) {
        const int sy1 = min(sy0 + rows_per_strip - 1, by1);
        const int ncell = (sy1 - sy0 + 1) * ww;
        for (int i = tid; i < (ncell + 1) / 2; i += MAP_THREADS) s_cnt[i] = 0;
        __syncthreads();
        // ---- phase B: endpoint pass (mapping.cpp:42-57).  H = rays ending in a cell; the ray whose increment found the
        // counter at zero is the cell's leader and applies the whole closed form for that cell after the walk.
        MSTAMP(3);
        for (int r0 = 0; r0 < a.R; r0 += MAP_THREADS) {         // one round when R <= 1024
            const int r = r0 + tid;
            int ci = -1;
            bool leader = false;
            if (r < a.R) {
                const int4 me = seg_walk ? my_ray : a.rays[r];
                if (me.x != 0x7fffffff && me.z >= bx0 && me.z <= bx1 && me.w >= sy0 && me.w <= sy1) {   // window = clipped grid
                    ci = (me.w - sy0) * ww + (me.z - bx0);
                    const unsigned int old = atomicAdd(&s_cnt[ci >> 1], (ci & 1) ? 0x10000u : 1u);
                    leader = half_of(old, ci) == 0u;
                }
            }
            __syncthreads();
            int H = 0, idx_y = 0;
            size_t idx = 0;
            if (leader) {
                H = (int)half_of(s_cnt[ci >> 1], ci);
                const int4 me = seg_walk ? my_ray : a.rays[r];
                idx = (size_t)me.w * a.frame.width + me.z;
                idx_y = me.w;
            }
            __syncthreads();
            if (a.R > MAP_THREADS) {
                // more rays than threads (never the case for a 290-ray lidar): hits of this round go straight to the grid;
                // the free-space pass of an end cell then sees the already saturated value, as in the reference
                if (leader) {
                    int v = a.cells[idx];
                    a.cells[idx] = (int8_t)min(127, v + a.hit * H);
                    if (a.mirror) a.mirror[mirror_at(a, idx, idx_y)] = a.cells[idx];
                }
                if (ci >= 0) atomicAnd(&s_cnt[ci >> 1], (ci & 1) ? 0x0000ffffu : 0xffff0000u);
                __syncthreads();
                continue;
            }
            // ---- phase C: free-space pass (mapping.cpp:59-71, 101-127): M = rays crossing a cell
            for (int i = tid; i < (ncell + 1) / 2; i += MAP_THREADS) s_cnt[i] = 0;
            __syncthreads();
            MSTAMP(4);
            const int nseg = s_segp[a.R];
            for (int sg = tid; sg < nseg; sg += MAP_THREADS) {
                int lo = 0, hi = a.R - 1;                       // last ray whose prefix <= sg
                while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (s_segp[mid] <= sg) lo = mid; else hi = mid - 1; }
                const int4 ray = a.rays[lo];
                const int k0 = (sg - s_segp[lo]) * MAP_SEG;
                const int dx = abs(ray.z - ray.x), dy = abs(ray.w - ray.y);
                const int sx = ray.x < ray.z ? 1 : -1, sy = ray.y < ray.w ? 1 : -1;
                const int K = max(dx, dy), k1 = min(K, k0 + MAP_SEG);
                const bool xmajor = dx >= dy;
                const int dmaj = xmajor ? dx : dy, dmin = xmajor ? dy : dx;
                const long long num = 2ll * k0 * dmin + dmaj;
                int n = (int)(num / (2ll * dmaj));
                int rem = (int)(num - (long long)n * 2ll * dmaj);
                for (int k = k0; k < k1; ++k) {
                    const int x = xmajor ? ray.x + sx * k : ray.x + sx * n;
                    const int y = xmajor ? ray.y + sy * n : ray.y + sy * k;
                    if (x >= bx0 && x <= bx1 && y >= sy0 && y <= sy1) {   // window is already clipped to the grid
                        const int c = (y - sy0) * ww + (x - bx0);
                        atomicAdd(&s_cnt[c >> 1], (c & 1) ? 0x10000u : 1u);
                    }
                    rem += 2 * dmin;
                    if (rem >= 2 * dmaj) { rem -= 2 * dmaj; n += 1; }
                }
            }
            __syncthreads();
            MSTAMP(5);
            if (prov) {
                // hit and miss counts stand in LDS, nothing has been stored to the grid yet: now the reference's x, y
                XSTAMP(15);
                take_exact_pose();
                XSTAMP(11);
                const bl_pose3 pe2 = {lds_pose->x, lds_pose->y, pe.theta};
                if (tid < a.R) {
                    const int4 ray2 = map_ray_cells(a, pb, pe2, pre_range, pre_theta, pre_time);
                    if (ray2.x != my_ray.x || ray2.y != my_ray.y || ray2.z != my_ray.z || ray2.w != my_ray.w) atomicOr(&s_redo, 1);
                }
                __syncthreads();
                if (s_redo) return 1;                                 // some ray's cells moved: count again with the exact pose
                prov = false;                                         // every ray has the reference's cells: the counts are the reference's
                XSTAMP(12);
            }
            // leaders finish their end cell: v' = max(-128, min(127, v + hit*H) - miss*M), then hide it from the window pass
            if (leader) {
                const int M = (int)half_of(s_cnt[ci >> 1], ci);
                int v = a.cells[idx];
                v = max(-128, min(127, v + a.hit * H) - a.miss * M);
                a.cells[idx] = (int8_t)v;
                if (snap) snap[idx] = (int8_t)v;
                if (a.mirror) a.mirror[mirror_at(a, idx, idx_y)] = (int8_t)v;
                atomicAnd(&s_cnt[ci >> 1], (ci & 1) ? 0x0000ffffu : 0xffff0000u);
            }
            __syncthreads();
        }
        if (a.R > MAP_THREADS) {
            // serial walk per ray (the pre-segment form), all hits already applied
            for (int r = tid; r < a.R; r += MAP_THREADS) {
                int4 ray = a.rays[r];
                if (ray.x == 0x7fffffff) continue;
                int x = ray.x, y = ray.y;
                const int x2 = ray.z, y2 = ray.w;
                const int dx = abs(x2 - x), dy = abs(y2 - y);
                const int sx = x < x2 ? 1 : -1, sy = y < y2 ? 1 : -1;
                int err = dx - dy;
                int guard = dx + dy + 2;
                while ((x != x2 || y != y2) && guard-- > 0) {
                    if (x >= bx0 && x <= bx1 && y >= sy0 && y <= sy1) {
                        int c = (y - sy0) * ww + (x - bx0);
                        atomicAdd(&s_cnt[c >> 1], (c & 1) ? 0x10000u : 1u);
                    }
                    int e2 = 2 * err;
                    if (e2 >= -dy) { err -= dy; x += sx; }
                    if (e2 <= dx) { err += dx; y += sy; }
                }
            }
            __syncthreads();
        }
        XSTAMP(13);
        // window pass: thread (tx, ty) owns column bx0 + tx (+256, ...) and every 4th row; 8 rows per batch so the byte loads
        // of a batch are in flight together (a serial load -> store chain per cell cost ~1 us per cell per thread)
        const int tx = tid & 255, ty = tid >> 8;
        const int nrows = sy1 - sy0 + 1;
        if (dword_rows) {
            // four cells (one dword of the grid, two dwords of counters) per item, four items in flight per thread
            const int wq = ww >> 2;
            const int nitems = nrows * wq;
            for (int it0 = tid; it0 < nitems; it0 += 4 * MAP_THREADS) {
                unsigned long long c[4];
                size_t gi[4];
                int v[4], gy[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int it = it0 + u * MAP_THREADS;
                    c[u] = 0ull; gi[u] = 0; v[u] = 0; gy[u] = 0;
                    if (it < nitems) {
                        const int ry = it / wq, dq = it - ry * wq;
                        c[u] = *(const unsigned long long*)&s_cnt[(ry * ww + 4 * dq) >> 1];
                        gi[u] = (size_t)(sy0 + ry) * a.frame.width + bx0 + 4 * dq;
                        gy[u] = sy0 + ry;
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (c[u]) v[u] = *(const int*)(a.cells + gi[u]);
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (c[u]) {
                        unsigned int nv = 0;
#pragma unroll
                        for (int b = 0; b < 4; ++b) {
                            const int M = (int)((c[u] >> (16 * b)) & 0xffffull);
                            const int val = (int)(int8_t)(v[u] >> (8 * b));
                            nv |= ((unsigned int)max(-128, val - a.miss * M) & 0xffu) << (8 * b);
                        }
                        *(int*)(a.cells + gi[u]) = (int)nv;
                        if (snap) *(int*)(snap + gi[u]) = (int)nv;
                        if (a.mirror) *(int*)(a.mirror + mirror_at(a, gi[u], gy[u])) = (int)nv;
                    }
            }
        } else
        for (int cx = tx; cx < ww; cx += 256) {
            for (int r0 = ty; r0 < nrows; r0 += 32) {
                int M[8], v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int ry = r0 + 4 * u;
                    M[u] = 0;
                    if (ry < nrows) {
                        const int ci2 = ry * ww + cx;
                        M[u] = (int)half_of(s_cnt[ci2 >> 1], ci2);
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (M[u] > 0) v[u] = a.cells[(size_t)(sy0 + r0 + 4 * u) * a.frame.width + bx0 + cx];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (M[u] > 0) {
                        const size_t at = (size_t)(sy0 + r0 + 4 * u) * a.frame.width + bx0 + cx;
                        a.cells[at] = (int8_t)max(-128, v[u] - a.miss * M[u]);
                        if (a.mirror) a.mirror[mirror_at(a, at, sy0 + r0 + 4 * u)] = a.cells[at];
                    }
            }
        }
        __syncthreads();
        XSTAMP(14);
        MSTAMP(6);
    }
    return 0;
    };   // run

    const bool ahead = provisional && lds_pose != nullptr;
    const int again = run(ahead);
    if (again == 1) (void)run(false);
    if (ahead && tid == 0) { a.fin.state->lookahead[0] += 1; a.fin.state->lookahead[1] += again; }
    // previousPose_ = pose (mapping.cpp:38) -- the exact one
    if (tid == 0) {
        bl_pose_xyt_t rec = lds_pose ? *lds_pose : (a.cur_dev ? *a.cur_dev : a.cur_host);
        rec.utime = a.cur_utime;
        *a.prev = rec;
        if (a.dirty_entry)                                     // one 16-byte store: box and the version it belongs to
            *a.dirty_entry = make_int4(s_dirty[0] | (s_dirty[1] << 16), s_dirty[2] | (s_dirty[3] << 16), (int)(unsigned int)a.dirty_version,
                                       (int)(unsigned int)(a.dirty_version >> 32));
    }
}

// ---------------------------------------------------------------- host side
extern "C" int bl_mapping_create(bl_ctx* ctx, float max_laser_distance, int8_t hit_odds, int8_t miss_odds, bl_mapping** out)
{
    BL_CHECK_ARG(ctx != nullptr && out != nullptr);
    BL_CHECK_ARG(hit_odds >= 0 && miss_odds >= 0);
    BL_HIP(hipSetDevice(ctx->device));
    bl_mapping* m = new bl_mapping();
    m->ctx = ctx;
    m->max_laser = max_laser_distance;
    m->hit = hit_odds; m->miss = miss_odds;
    m->initialized = false;
    m->prev_utime = 0;
    m->d_prev = nullptr; m->d_rays = nullptr; m->ray_capacity = 0;
    hipError_t e = hipMalloc((void**)&m->d_prev, sizeof(bl_pose_xyt_t));
    if (e != hipSuccess) { bl_set_error("hipMalloc failed: %s", hipGetErrorString(e)); delete m; return BL_ERR_HIP; }
    BL_HIP(hipMemsetAsync(m->d_prev, 0, sizeof(bl_pose_xyt_t), ctx->stream));
    static bool attr_set = false;
    if (!attr_set) {
        BL_HIP(hipFuncSetAttribute((const void*)k_map_update, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   MAP_LDS_COUNTERS * 2));
        attr_set = true;
    }
    *out = m;
    return BL_OK;
}

extern "C" void bl_mapping_destroy(bl_mapping* m)
{
    if (!m) return;
    (void)hipStreamSynchronize(m->ctx->stream);
    (void)hipFree(m->d_prev);
    if (m->d_rays) (void)hipFree(m->d_rays);
    delete m;
}

#define MAP_SNAPSHOT_IN_KERNEL_CELLS (256 * 1024)      // larger grids: one workgroup would copy for too long

// what mapping_update_impl checks of its arguments, ahead of taking a filter's finish (a finish that has been taken MUST be
// launched: the filter's end-of-update bookkeeping is done by then)
static int mapping_check_args(const bl_mapping* m, const bl_lidar_t* scan, const bl_grid* map)
{
    BL_CHECK_ARG(m != nullptr && scan != nullptr && map != nullptr);
    BL_CHECK_ARG(scan->num_ranges >= 0 && scan->num_ranges <= MAP_MAX_RAYS);
    BL_CHECK_ARG(map->frame.width <= 65535 && map->frame.height <= 65535);       // end cells are packed 16+16 bits in LDS
    return BL_OK;
}

static int mapping_update_impl(bl_mapping* m, const bl_lidar_t* scan, const bl_pose_xyt_t* h_pose, const void* d_pose,
                               int64_t pose_utime, bl_grid* map, const bl_planner_snap* snap = nullptr,
                               const mcl_finish_args* fin = nullptr)
{
    { const int rc0 = mapping_check_args(m, scan, map); if (rc0) return rc0; }
    bl_ctx* ctx = m->ctx;
    BL_HIP(hipSetDevice(ctx->device));
    int64_t begin = m->initialized ? m->prev_utime : pose_utime;
    int R = 0;
    int rc = bl_scan_upload(ctx, scan, &R);
    if (rc) return rc;
    if (R > m->ray_capacity) {
        if (m->d_rays) { BL_HIP(hipStreamSynchronize(ctx->stream)); BL_HIP(hipFree(m->d_rays)); }
        int cap = R < 1024 ? 1024 : R;
        BL_HIP(hipMalloc((void**)&m->d_rays, (size_t)cap * sizeof(int4)));
        m->ray_capacity = cap;
    }
    map_args a;
    a.cells = map->cells;
    a.frame = map->frame;
    a.ranges = ctx->scan.ranges; a.thetas = ctx->scan.thetas; a.times = ctx->scan.times;
    a.t_begin = begin; a.t_den = (begin != pose_utime) ? (double)(pose_utime - begin) : 1.0;
    a.R = R;
    a.prev = m->d_prev;
    a.cur_dev = (const bl_pose_xyt_t*)d_pose;
    if (h_pose) a.cur_host = *h_pose; else { a.cur_host.utime = 0; a.cur_host.x = a.cur_host.y = a.cur_host.theta = 0; }
    a.cur_utime = pose_utime;
    a.interp = (begin != pose_utime) ? 1 : 0;
    a.apply = m->initialized ? 1 : 0;
    a.max_laser = m->max_laser;
    a.hit = m->hit; a.miss = m->miss;
    a.rays = m->d_rays;
    a.stamps = nullptr;
    // the zero-framed mirror the particle filter gathers from, kept current while this update's stream is the one it was built on
    a.mirror = nullptr; a.mirror_stride = 0;
    if (map->mirror && map->mirror_valid && !map->mirror_external && map->ctx == ctx) {
        a.mirror = map->mirror + BL_MIRROR_FRAME * map->mirror_stride + 4;
        a.mirror_stride = map->mirror_stride;
    } else map->mirror_valid = false;
    // the cells this update may change go to the lineage's log (grids up to 65535 a side: checked above)
    a.dirty_entry = bl_grid_log_next(map, (uint64_t*)&a.dirty_version);
    a.snap_cells = nullptr; a.snap_pose = nullptr; a.snap_pose_src = nullptr; a.snap_flag = nullptr; a.snap_seq = 0;
    if (snap) {
        a.snap_cells = snap->cells; a.snap_pose = snap->pose; a.snap_pose_src = (const bl_pose_xyt_t*)d_pose;
        a.snap_flag = snap->flag; a.snap_seq = snap->seq;
    }
#ifdef BL_MAP_STAMPS
    static long long* d_st = nullptr;
    if (!d_st) BL_HIP(hipMalloc((void**)&d_st, 64));
    a.stamps = d_st;
#endif
    hipEvent_t e0, e1;
    rc = bl_timer_begin(ctx, BL_K_MAP, &e0, &e1);
    if (rc) return rc;
    a.fin_on = fin ? 1 : 0;
    if (fin) a.fin = *fin; else a.fin = mcl_finish_args{};
    // the current scan block's pointers are in `a` already: a waiting prefetch may now take the other block
    a.pre = bl_scan_prefetch_args{};
    a.pre_on = bl_scan_prefetch_take(ctx, &a.pre);
    static_assert(MCLF_WG == MAP_THREADS, "the riding finish uses the map kernel's workgroup size");
    hipLaunchKernelGGL(k_map_update, dim3(1 + (fin ? fin->groups_wait + MAP_RIDER_WGS - 1 : 0) + a.pre_on), dim3(MAP_THREADS), MAP_LDS_COUNTERS * 2, ctx->stream, a);
    BL_HIP(hipGetLastError());
    rc = bl_timer_end(ctx, BL_K_MAP, e0, e1);
    if (rc) return rc;
#ifdef BL_MAP_STAMPS
    {
        long long h[8];
        BL_HIP(hipMemcpy(h, a.stamps, 56, hipMemcpyDeviceToHost));
        fprintf(stderr, "[map stamps] pose %lld rays %lld hits %lld zero %lld walk %lld apply %lld cycles\n", h[1] - h[0], h[2] - h[1],
                h[3] - h[2], h[4] - h[3], h[5] - h[4], h[6] - h[5]);
    }
#endif
    m->initialized = true;
    m->prev_utime = pose_utime;
    return BL_OK;
}

extern "C" int bl_mapping_update(bl_mapping* m, const bl_lidar_t* scan, const bl_pose_xyt_t* pose, bl_grid* map)
{
    BL_CHECK_ARG(pose != nullptr);
    return mapping_update_impl(m, scan, pose, nullptr, pose->utime, map);
}

extern "C" int bl_mapping_update_dev_pose(bl_mapping* m, const bl_lidar_t* scan, const void* d_pose, int64_t pose_utime,
                                          bl_grid* map)
{
    BL_CHECK_ARG(d_pose != nullptr);
    return mapping_update_impl(m, scan, nullptr, d_pose, pose_utime, map);
}

// Mapping::updateMap with the pose read on the device, followed by a replanner submission (bl_planner_submit) whose map +
// pose snapshot is taken by the map kernel itself when the grid is small enough for one workgroup to copy.
extern "C" int bl_planner_submit_with_map_update(bl_planner* p, bl_mapping* m, const bl_lidar_t* scan, const void* d_pose,
                                                 int64_t pose_utime, bl_grid* map, const bl_pose_xyt_t* goal,
                                                 const bl_search_params_t* params)
{
    BL_CHECK_ARG(p != nullptr && m != nullptr && d_pose != nullptr && map != nullptr && goal != nullptr && params != nullptr);
    if ((size_t)map->frame.width * map->frame.height > (size_t)MAP_SNAPSHOT_IN_KERNEL_CELLS) {
        int rc = mapping_update_impl(m, scan, nullptr, d_pose, pose_utime, map);
        if (rc) return rc;
        return bl_planner_submit(p, map, d_pose, goal, params);
    }
    bl_planner_snap sn;
    int rc = bl_planner_reserve(p, map, &sn);
    if (rc) return rc;
    rc = mapping_update_impl(m, scan, nullptr, d_pose, pose_utime, map, &sn);
    if (rc) { bl_planner_cancel(p); return rc; }
    bl_grid_adopt_lineage(sn.grid, map);                        // the snapshot holds the map as this update leaves it
    return bl_planner_commit(p, goal, params);
}

// The same two calls with the END of the particle-filter update folded in: `pf` has an update begun (bl_pf_update_begin)
// and this call is its bl_pf_update_end(pf, NULL) -- the pose estimate is formed by the map kernel's own workgroup right
// before Mapping::updateMap uses it, the weight prefix is written by further workgroups of the same launch, and the SLAM
// stream carries one kernel less per step.  A filter with nothing pending (the robot did not move) or whose finish cannot
// ride (sharded particle set) is ended the ordinary way first; results are bit-identical either way.
static int finishing_pf_prepare(bl_pf* pf, bl_mapping* m, const bl_lidar_t* scan, const bl_grid* map, mcl_finish_args* fin, bool* ride)
{
    *ride = false;
    if (bl_pf_ctx(pf) != m->ctx) { bl_set_error("filter and mapping belong to different contexts"); return BL_ERR_ARG; }
    int rc = mapping_check_args(m, scan, map);                  // before the filter's bookkeeping: a refused call leaves it pending
    if (rc) return rc;
    const int t = bl_pf_take_finish(pf, fin);
    if (t > 0) { *ride = true; return BL_OK; }
    return bl_pf_update_end(pf, nullptr);                       // no-op when nothing is pending
}

// the map update that carries a taken finish failed before its launch (scan upload, allocation): the finish goes out on its own
static int finish_after_failure(bl_pf* pf, const mcl_finish_args* fin, bool ride, int rc)
{
    if (rc && ride) (void)bl_pf_launch_taken_finish(pf, fin);
    return rc;
}

extern "C" int bl_mapping_update_finishing_pf(bl_mapping* m, const bl_lidar_t* scan, bl_pf* pf, int64_t pose_utime, bl_grid* map)
{
    BL_CHECK_ARG(m != nullptr && pf != nullptr && map != nullptr);
    mcl_finish_args fin; bool ride;
    int rc = finishing_pf_prepare(pf, m, scan, map, &fin, &ride);
    if (rc) return rc;
    rc = mapping_update_impl(m, scan, nullptr, bl_pf_pose_device_ptr(pf), pose_utime, map, nullptr, ride ? &fin : nullptr);
    if (!rc && ride) bl_pf_ride_launched(pf);
    return finish_after_failure(pf, &fin, ride, rc);
}

extern "C" int bl_planner_submit_with_map_update_finishing_pf(bl_planner* p, bl_mapping* m, const bl_lidar_t* scan, bl_pf* pf,
                                                              int64_t pose_utime, bl_grid* map, const bl_pose_xyt_t* goal,
                                                              const bl_search_params_t* params)
{
    BL_CHECK_ARG(p != nullptr && m != nullptr && pf != nullptr && map != nullptr && goal != nullptr && params != nullptr);
    mcl_finish_args fin; bool ride;
    int rc = finishing_pf_prepare(pf, m, scan, map, &fin, &ride);
    if (rc) return rc;
    const void* d_pose = bl_pf_pose_device_ptr(pf);
    if ((size_t)map->frame.width * map->frame.height > (size_t)MAP_SNAPSHOT_IN_KERNEL_CELLS) {
        rc = mapping_update_impl(m, scan, nullptr, d_pose, pose_utime, map, nullptr, ride ? &fin : nullptr);
        if (rc) return finish_after_failure(pf, &fin, ride, rc);
        if (ride) bl_pf_ride_launched(pf);
        return bl_planner_submit(p, map, d_pose, goal, params);
    }
    bl_planner_snap sn;
    rc = bl_planner_reserve(p, map, &sn);
    if (rc) {
        // the filter's bookkeeping is already done: its finish must still be launched (with the map update, which is this step's)
        if (ride) { const int rc2 = mapping_update_impl(m, scan, nullptr, d_pose, pose_utime, map, nullptr, &fin); if (!rc2) bl_pf_ride_launched(pf); (void)finish_after_failure(pf, &fin, true, rc2); }
        return rc;
    }
    rc = mapping_update_impl(m, scan, nullptr, d_pose, pose_utime, map, &sn, ride ? &fin : nullptr);
    if (rc) { bl_planner_cancel(p); return finish_after_failure(pf, &fin, ride, rc); }
    if (ride) bl_pf_ride_launched(pf);
    bl_grid_adopt_lineage(sn.grid, map);
    return bl_planner_commit(p, goal, params);
}
