// bl_planning.hip -- ObstacleDistanceGrid::setDistances (src/planning/obstacle_distance_grid.cpp:73-181) and
// search_for_path (src/planning/astar.cpp:9-274) as gfx950 kernels.
//
// Distance grid.  The reference floods a 4-connected min-heap from every cell with log-odds >= 0 (free cells start at
// -1) and gives each newly reached cell parent + 0.1f.  Pops come in non-decreasing distance, so a cell whose
// 4-connected (L1) distance to the nearest non-free cell is n receives f[n], f[0] = 0, f[n] = f[n-1] + 0.1f (float);
// a map with no non-free cell keeps -1 everywhere.  The kernels compute the exact integer L1 distance transform
// (separable: nearest source within the row, then a min-plus sweep down and up the columns) and map it through the
// float table f -- bit-identical floats, no heap.
//
// A*.  The reference's result depends on the pop order of libstdc++'s binary heap among equal fCost and on its
// re-expansion of duplicate open-list entries, so the search is executed with exactly those heap index operations
// (std::push_heap / std::pop_heap semantics, stl_heap.h) by one wavefront; the closed list is an int32 grid of parent moves
// (first closing of a cell wins, which is all is_member/get_member ever observe).  Single-search latency is bound by
// dependent memory accesses, not bandwidth (DESIGN.md "A*").
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <cmath>
#include <deque>
#include <map>

#include "bl_internal.h"

// =============================================================================================== distance grid
struct bl_dist {
    bl_ctx* ctx;
    bl_frame frame;
    size_t capacity;          // cells allocated
    uint16_t* row;            // per-row nearest-source distance (0xFFFF: none in row)
    uint16_t* l1;             // L1 distance (0xFFFF: no source anywhere)
    float* cells;             // float distances handed to callers: f[l1], allocated and formed when a caller first asks (floats_valid)
    bool floats_valid;
    bool floats_handed_out;   // bl_dist_device_ptr has been called: a caller may read the floats without asking again (see there)
    float* lut;               // device f[n]
    int32_t* closed;          // A* closed-cell scratch of this grid: (generation << 3) | move code; an entry of another generation
                              // than the running search's is "not closed" -- no clearing between searches (k_astar)
    uint32_t closed_gen;      // generation of the last search launched on this grid (0: none yet; the array starts zeroed)
    int* sum_f; int* sum_b; size_t sum_cap;   // per macro strip and column: chain summaries of the large-grid column pass
    int lut_n;
    std::vector<float>* lut_host;
    bool valid;
    // incremental transforms (see "incremental" below): what l1 is the transform of, and the device / pinned words of the scheme
    uint64_t src_id, src_version;
    bool bound_ok;            // state[DST_DUB] bounds l1's distances (the merged column pass formed it)
    unsigned int* state;      // device: DST_* words
    unsigned int* h_status;   // pinned: the plan the last incremental launch settled on (DST_MODE_*), written by the device
    unsigned int* h_status_dev;
    int inc_holdoff;          // incremental launches to skip (the device kept falling back to the whole grid)
    // k_dist_fused: the tiles' summary and claim words, the grid they were laid out for, the last launch's tag
    unsigned int* fwords; size_t fwords_cap; int f_w, f_h; unsigned int f_tag;
    int64_t n_inc, n_full, n_same;   // transforms by kind (diagnostic)
};

#define DIST_INF (1 << 28)

// The grids of one launch: blockIdx.z picks the unit.  A replanner lane transforms the snapshots of a whole batch in one set of
// launches instead of one set per snapshot (eight 200 x 200 grids: 0.17 ms of back-to-back small kernels on the lane -> 0.03);
// a single setDistances is a batch of one.
#define DIST_MAX_BATCH 32
struct dist_batch {
    const int8_t* cells[DIST_MAX_BATCH]; uint16_t* row[DIST_MAX_BATCH]; uint16_t* l1[DIST_MAX_BATCH];
    float* out[DIST_MAX_BATCH]; const float* lut[DIST_MAX_BATCH];      // small grids: the float grid, written by the column pass when a
                                                                       // caller holds its device pointer (bl_dist_device_ptr); else null
    int* sum_f[DIST_MAX_BATCH]; int* sum_b[DIST_MAX_BATCH];
    unsigned int* state[DIST_MAX_BATCH]; unsigned int* hstat[DIST_MAX_BATCH];
    const int4* log[DIST_MAX_BATCH]; unsigned int from[DIST_MAX_BATCH], to[DIST_MAX_BATCH];   // incremental: map updates from + 1 .. to of the log
};

// ---- incremental transform: the words of bl_dist::state
// A map update changes cells inside a box B (the lineage's dirty log, bl_internal.h).  With D an upper bound of every finite
// distance before AND after, a cell farther than D from B keeps its distance: its nearest source is nearer than any changed cell
// (removed sources do not matter) and no new source is nearer than that.  So only the window W = B dilated by D + 1 is
// transformed again, with the OLD distances of the ring of cells around W as seeds (an L1 path from a cell of W to a source
// outside W crosses the ring at a cell q with d(q) + |p - q| = d(p), and d(q) is unchanged) -- exact, bit for bit the full
// transform (tests/test_gpu_dist_incremental.py).  The window is decided ON THE DEVICE from the log (the host has not seen the
// poses the boxes depend on): a window wider than DINC_MAX cells, a log entry that is not there any more, or no bound D make the
// same launch transform the whole grid instead (slower than the dedicated kernels; the host then stays away from the
// incremental form for a while, h_status).
#define DST_DUB 0            // D: upper bound of the finite distances of l1 (0xFFFF: unknown / a cell without any source)
#define DST_MODE 1           // plan of the running transform: DST_MODE_*
#define DST_X0 2             // ... its region, inclusive (x0 a multiple of 16, x1 + 1 a multiple of 16)
#define DST_Y0 3
#define DST_X1 4
#define DST_Y1 5
#define DST_MAX_GROUPS 64    // grids up to 8192 columns / rows in the region kernels (macro strips whose summaries fit LDS)
#define DST_STATS 8          // [3] incremental launches that ended as: nothing to do, a window, the whole grid (diagnostic)
#define DST_FAILED (DST_STATS + 3)   // k_dist_fused: workgroups that gave up (a summary that never came)
#define DST_HELPED (DST_STATS + 4)   // k_dist_fused: summaries computed in place of a workgroup that had not claimed its tile yet
#define DST_WORDS (DST_STATS + 5)
#define DST_MODE_NONE 0      // nothing changed
#define DST_MODE_WINDOW 1
#define DST_MODE_FULL 2
#define DST_MODE_BROKEN 0xDEADu   // (h_status only) a whole-grid launch of k_dist_fused gave up: the grid's distances are not valid
#define DINC_MAX 1024        // widest / tallest window

// ---- plans ------------------------------------------------------------------------------------------------------------
// agent-scope relaxed accesses through the L2 (sc1): what one workgroup hands to another inside a launch (the XCDs have separate
// L2s; a release fence would write a whole L2 back)
__device__ __forceinline__ void dst_store(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int dst_load(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned int dst_load_u(const unsigned int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// The whole grid as the plan of the transform the column pass behind this kernel runs (one thread of the launch's first
// workgroup per unit); the bound D starts again from the values that pass writes.
__device__ __forceinline__ void dist_plan_full(unsigned int* state, int W, int H)
{
    if (!state) return;
    state[DST_DUB] = 0u;
    state[DST_MODE] = DST_MODE_FULL;
    state[DST_X0] = 0u; state[DST_Y0] = 0u; state[DST_X1] = (unsigned int)(W - 1); state[DST_Y1] = (unsigned int)(H - 1);
}

// Row pass: one workgroup per row; d_row[x] = min over sources x' in the row of |x - x'|.
__global__ __launch_bounds__(256) void k_dist_rows(dist_batch db, int W)
{
    const int8_t* __restrict__ cells = db.cells[blockIdx.z];
    uint16_t* __restrict__ row = db.row[blockIdx.z];
    __shared__ int s_wave[4];
    __shared__ int s_carry;
    if (blockIdx.x == 0 && threadIdx.x == 0) dist_plan_full(db.state[blockIdx.z], W, (int)gridDim.x);
    const int y = blockIdx.x;
    const int8_t* c = cells + (size_t)y * W;
    uint16_t* out = row + (size_t)y * W;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // forward: nearest source at or left of x
    if (threadIdx.x == 0) s_carry = -DIST_INF;
    __syncthreads();
    for (int base = 0; base < W; base += 256) {
        int x = base + threadIdx.x;
        int v = (x < W && c[x] >= 0) ? x : -DIST_INF;          // is_cell_occupied: logOdds >= 0 (obstacle_distance_grid.cpp:125-128)
        int incl = v;
        for (int off = 1; off < 64; off <<= 1) {
            int t = __shfl_up(incl, off, 64);
            if (lane >= off) incl = max(incl, t);
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        int best = max(incl, s_carry);
        for (int w = 0; w < wave; ++w) best = max(best, s_wave[w]);
        if (x < W) out[x] = (uint16_t)min(x - best, 0xFFFF);
        __syncthreads();
        if (threadIdx.x == 255) s_carry = best;
        __syncthreads();
    }
    // backward: nearest source at or right of x
    if (threadIdx.x == 0) s_carry = DIST_INF;
    __syncthreads();
    const int nseg = (W + 255) / 256;
    for (int seg = nseg - 1; seg >= 0; --seg) {
        int x = seg * 256 + threadIdx.x;
        int v = (x < W && c[x] >= 0) ? x : DIST_INF;
        int incl = v;                                           // suffix min within the wave
        for (int off = 1; off < 64; off <<= 1) {
            int t = __shfl_down(incl, off, 64);
            if (lane + off < 64) incl = min(incl, t);
        }
        if (lane == 0) s_wave[wave] = incl;
        __syncthreads();
        int best = min(incl, s_carry);
        for (int w = wave + 1; w < 4; ++w) best = min(best, s_wave[w]);
        if (x < W) {
            int d = min(best - x, 0xFFFF);
            out[x] = (uint16_t)min((int)out[x], d);
        }
        __syncthreads();
        if (threadIdx.x == 0) s_carry = best;
        __syncthreads();
    }
}

// Row pass for narrow grids (W <= 256, a multiple of 4: the 200 x 200 maps): a WAVE owns a row, a lane four consecutive cells
// (one 4-byte load, one 8-byte store); both scans run on the same registers, nothing goes through LDS and nothing waits for a
// barrier -- k_dist_rows spends a 256-thread workgroup, eight barriers and a read-back of its own output on such a row.
__global__ __launch_bounds__(256) void k_dist_rows_narrow(dist_batch db, int W, int H)
{
    const int8_t* __restrict__ cells = db.cells[blockIdx.z];
    uint16_t* __restrict__ row = db.row[blockIdx.z];
    if (blockIdx.x == 0 && threadIdx.x == 0) dist_plan_full(db.state[blockIdx.z], W, H);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int y = blockIdx.x * 4 + wave;
    if (y >= H) return;
    const int x0 = lane * 4;
    int raw = -1;                                               // 0xFF bytes: free cells, no source
    if (x0 < W) raw = *(const int*)(cells + (size_t)y * W + x0);
    int b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) b[i] = (int)(int8_t)(raw >> (8 * i));
    int last = -DIST_INF, first = DIST_INF;
#pragma unroll
    for (int i = 0; i < 4; ++i) if (b[i] >= 0) last = x0 + i;       // is_cell_occupied: logOdds >= 0 (obstacle_distance_grid.cpp:125-128)
#pragma unroll
    for (int i = 3; i >= 0; --i) if (b[i] >= 0) first = x0 + i;
    int il = last, ir = first;
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(il, off, 64); if (lane >= off) il = max(il, t);
        const int u = __shfl_down(ir, off, 64); if (lane + off < 64) ir = min(ir, u);
    }
    int bl = __shfl_up(il, 1, 64), br = __shfl_down(ir, 1, 64);
    if (lane == 0) bl = -DIST_INF;
    if (lane == 63) br = DIST_INF;
    if (x0 < W) {
        unsigned int o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { if (b[i] >= 0) bl = x0 + i; o[i] = (unsigned int)min(x0 + i - bl, 0xFFFF); }
#pragma unroll
        for (int i = 3; i >= 0; --i) { if (b[i] >= 0) br = x0 + i; o[i] = min(o[i], (unsigned int)min(br - (x0 + i), 0xFFFF)); }
        *(uint2*)(row + (size_t)y * W + x0) = make_uint2(o[0] | (o[1] << 16), o[2] | (o[3] << 16));
    }
}

// Row pass for wide grids (W a multiple of 16): a thread owns 16 consecutive cells (one 16-byte load, two 16-byte
// stores), a workgroup a 4096-cell chunk of the row; nearest source to the left = exclusive max-scan of the threads' last
// source index (wave shuffles + one LDS exchange), to the right the mirrored min-scan; chunks of rows wider than 4096 are
// chained left-to-right and right-to-left through a carry.  3 B of traffic per cell, four barriers per chunk (k_dist_rows
// re-reads its own output and takes eight barriers per 256 cells: 80 us at 4096^2, 0.6 TB/s).
#define DR2_CELLS 16
#define DR2_CHUNK (256 * DR2_CELLS)
__global__ __launch_bounds__(256) void k_dist_rows_wide(dist_batch db, int W)
{
    const int8_t* __restrict__ cells = db.cells[blockIdx.z];
    uint16_t* __restrict__ row = db.row[blockIdx.z];
    __shared__ int s_wl[4], s_wr[4];
    __shared__ int s_carry_l, s_carry_r;
    if (blockIdx.x == 0 && threadIdx.x == 0) dist_plan_full(db.state[blockIdx.z], W, (int)gridDim.x);
    const int y = blockIdx.x;
    const int8_t* c = cells + (size_t)y * W;
    uint16_t* out = row + (size_t)y * W;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nchunk = (W + DR2_CHUNK - 1) / DR2_CHUNK;
    if (nchunk == 1) {
        // the whole row is one chunk: both scans on the same registers, 1 B read + 2 B written per cell
        const int x0 = threadIdx.x * DR2_CELLS;
        int4 raw = make_int4(-1, -1, -1, -1);
        if (x0 < W) raw = *(const int4*)(c + x0);
        const int8_t* b = (const int8_t*)&raw;
        int last = -DIST_INF, first = DIST_INF;
#pragma unroll
        for (int i = 0; i < DR2_CELLS; ++i) if (b[i] >= 0) last = x0 + i;
#pragma unroll
        for (int i = DR2_CELLS - 1; i >= 0; --i) if (b[i] >= 0) first = x0 + i;
        int il = last, ir = first;
        for (int off = 1; off < 64; off <<= 1) {
            int t = __shfl_up(il, off, 64); if (lane >= off) il = max(il, t);
            int u = __shfl_down(ir, off, 64); if (lane + off < 64) ir = min(ir, u);
        }
        if (lane == 63) s_wl[wave] = il;
        if (lane == 0) s_wr[wave] = ir;
        __syncthreads();
        int bl = __shfl_up(il, 1, 64), br = __shfl_down(ir, 1, 64);
        if (lane == 0) bl = -DIST_INF;
        if (lane == 63) br = DIST_INF;
        for (int w = 0; w < wave; ++w) bl = max(bl, s_wl[w]);
        for (int w = wave + 1; w < 4; ++w) br = min(br, s_wr[w]);
        if (x0 < W) {
            uint16_t o[DR2_CELLS];
#pragma unroll
            for (int i = 0; i < DR2_CELLS; ++i) { if (b[i] >= 0) bl = x0 + i; o[i] = (uint16_t)min(x0 + i - bl, 0xFFFF); }
#pragma unroll
            for (int i = DR2_CELLS - 1; i >= 0; --i) { if (b[i] >= 0) br = x0 + i; o[i] = (uint16_t)min((int)o[i], min(br - (x0 + i), 0xFFFF)); }
            *(int4*)(out + x0) = *(const int4*)&o[0];
            *(int4*)(out + x0 + 8) = *(const int4*)&o[8];
        }
        return;
    }
    // ---- pass 1 (left to right): nearest source at or left of every cell, kept as a distance in `out`
    if (threadIdx.x == 0) { s_carry_l = -DIST_INF; s_carry_r = DIST_INF; }
    __syncthreads();
    for (int ch = 0; ch < nchunk; ++ch) {
        const int x0 = ch * DR2_CHUNK + threadIdx.x * DR2_CELLS;
        int4 raw = make_int4(-1, -1, -1, -1);                   // 0xFF bytes: free cells (log-odds < 0), i.e. no source
        if (x0 < W) raw = *(const int4*)(c + x0);
        const int8_t* b = (const int8_t*)&raw;
        int last = -DIST_INF;                                   // last source index inside this thread's cells
#pragma unroll
        for (int i = 0; i < DR2_CELLS; ++i) if (b[i] >= 0) last = x0 + i;   // is_cell_occupied: logOdds >= 0
        int incl = last;
        for (int off = 1; off < 64; off <<= 1) { int t = __shfl_up(incl, off, 64); if (lane >= off) incl = max(incl, t); }
        if (lane == 63) s_wl[wave] = incl;
        __syncthreads();
        int before = __shfl_up(incl, 1, 64);
        if (lane == 0) before = -DIST_INF;
        int best = max(before, s_carry_l);
        for (int w = 0; w < wave; ++w) best = max(best, s_wl[w]);
        if (x0 < W) {
            uint16_t o[DR2_CELLS];
#pragma unroll
            for (int i = 0; i < DR2_CELLS; ++i) { if (b[i] >= 0) best = x0 + i; o[i] = (uint16_t)min(x0 + i - best, 0xFFFF); }
            *(int4*)(out + x0) = *(const int4*)&o[0];
            *(int4*)(out + x0 + 8) = *(const int4*)&o[8];
        }
        __syncthreads();
        if (threadIdx.x == 255) s_carry_l = max(max(incl, s_carry_l), max(max(s_wl[0], s_wl[1]), s_wl[2]));
        __syncthreads();
    }
    // ---- pass 2 (right to left): nearest source at or right of every cell, merged with pass 1
    for (int ch = nchunk - 1; ch >= 0; --ch) {
        const int x0 = ch * DR2_CHUNK + threadIdx.x * DR2_CELLS;
        int4 raw = make_int4(-1, -1, -1, -1);
        if (x0 < W) raw = *(const int4*)(c + x0);
        const int8_t* b = (const int8_t*)&raw;
        int first = DIST_INF;
#pragma unroll
        for (int i = DR2_CELLS - 1; i >= 0; --i) if (b[i] >= 0) first = x0 + i;
        int incl = first;                                       // suffix min within the wave
        for (int off = 1; off < 64; off <<= 1) { int t = __shfl_down(incl, off, 64); if (lane + off < 64) incl = min(incl, t); }
        if (lane == 0) s_wr[wave] = incl;
        __syncthreads();
        int after = __shfl_down(incl, 1, 64);
        if (lane == 63) after = DIST_INF;
        int best = min(after, s_carry_r);
        for (int w = wave + 1; w < 4; ++w) best = min(best, s_wr[w]);
        if (x0 < W) {
            uint16_t o[DR2_CELLS];
            *(int4*)&o[0] = *(const int4*)(out + x0);
            *(int4*)&o[8] = *(const int4*)(out + x0 + 8);
#pragma unroll
            for (int i = DR2_CELLS - 1; i >= 0; --i) {
                if (b[i] >= 0) best = x0 + i;
                o[i] = (uint16_t)min((int)o[i], min(best - (x0 + i), 0xFFFF));
            }
            *(int4*)(out + x0) = *(const int4*)&o[0];
            *(int4*)(out + x0 + 8) = *(const int4*)&o[8];
        }
        __syncthreads();
        if (threadIdx.x == 0) s_carry_r = min(min(incl, s_carry_r), min(min(s_wr[1], s_wr[2]), s_wr[3]));
        __syncthreads();
    }
}

// Column pass: a workgroup owns 64 columns; its 16 thread rows split the H rows into strips.  d[y] = min(row[y],
// d[y-1]+1) downwards and the mirror upwards; strips are chained through LDS summaries.
#define DCOL_TX 64
#define DCOL_TY 16
__global__ __launch_bounds__(DCOL_TX * DCOL_TY) void k_dist_cols(dist_batch db, int W, int H)
{
    const uint16_t* __restrict__ row = db.row[blockIdx.z];
    uint16_t* __restrict__ l1 = db.l1[blockIdx.z];
    float* __restrict__ out = db.out[blockIdx.z];
    const float* __restrict__ lut = db.lut[blockIdx.z];
    __shared__ int s_fwd[DCOL_TY][DCOL_TX];
    __shared__ int s_bwd[DCOL_TY][DCOL_TX];
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int x = blockIdx.x * DCOL_TX + tx;
    const int rows_per = (H + DCOL_TY - 1) / DCOL_TY;
    const int y0 = min(ty * rows_per, H), y1 = min(y0 + rows_per, H);
    const bool live = x < W;
    int a_f = DIST_INF, a_b = DIST_INF;
    if (live) {
        for (int y = y0; y < y1; ++y) {
            int g = row[(size_t)y * W + x];
            if (g == 0xFFFF) g = DIST_INF;
            a_f = min(a_f, g + (y1 - 1 - y));
            a_b = min(a_b, g + (y - y0));
        }
    }
    s_fwd[ty][tx] = a_f;
    s_bwd[ty][tx] = a_b;
    __syncthreads();
    if (!live) return;
    // E = distance at the last row of the previous strip (downward chain); B = at the first row of the next strip
    int E = DIST_INF;
    for (int s = 0; s < ty; ++s) {
        int sy0 = min(s * rows_per, H), sy1 = min(sy0 + rows_per, H);
        E = min(s_fwd[s][tx], E + (sy1 - sy0));
    }
    int B = DIST_INF;
    for (int s = DCOL_TY - 1; s > ty; --s) {
        int sy0 = min(s * rows_per, H), sy1 = min(sy0 + rows_per, H);
        B = min(s_bwd[s][tx], B + (sy1 - sy0));
    }
    int d = E;
    for (int y = y0; y < y1; ++y) {
        int g = row[(size_t)y * W + x];
        if (g == 0xFFFF) g = DIST_INF;
        d = min(g, d + 1);
        l1[(size_t)y * W + x] = (uint16_t)min(d, 0xFFFF);
    }
    int b = B;
    for (int y = y1 - 1; y >= y0; --y) {
        int g = row[(size_t)y * W + x];
        if (g == 0xFFFF) g = DIST_INF;
        b = min(g, b + 1);
        int f = l1[(size_t)y * W + x];
        if (f == 0xFFFF) f = DIST_INF;
        int v = min(f, b);
        bool none = v >= 0xFFFF;
        l1[(size_t)y * W + x] = none ? (uint16_t)0xFFFF : (uint16_t)v;
        if (out) out[(size_t)y * W + x] = none ? -1.0f : lut[v];
    }
}

// The same for grids of up to DCOL_TY x DCOLS_ROWS rows (the 200 x 200 maps of the headline configuration): a thread's rows live in
// registers -- every load of the kernel is requested before the first is used, where the loops above made ~40 dependent trips
// through L2 per thread (13 rows, three passes: 23 us alone for 40 000 cells, the longer half of every replan's setDistances).
#define DCOLS_ROWS 16
__global__ __launch_bounds__(DCOL_TX * DCOL_TY) void k_dist_cols_small(dist_batch db, int W, int H)
{
    const uint16_t* __restrict__ row = db.row[blockIdx.z];
    uint16_t* __restrict__ l1 = db.l1[blockIdx.z];
    float* __restrict__ out = db.out[blockIdx.z];
    const float* __restrict__ lut = db.lut[blockIdx.z];
    __shared__ int s_fwd[DCOL_TY][DCOL_TX];
    __shared__ int s_bwd[DCOL_TY][DCOL_TX];
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int x = blockIdx.x * DCOL_TX + tx;
    const int rows_per = (H + DCOL_TY - 1) / DCOL_TY;           // <= DCOLS_ROWS (the host checks)
    const int y0 = min(ty * rows_per, H), y1 = min(y0 + rows_per, H);
    const bool live = x < W;
    int g[DCOLS_ROWS];
#pragma unroll
    for (int i = 0; i < DCOLS_ROWS; ++i) {
        const int y = y0 + i;
        int v = 0xFFFF;
        if (live && y < y1) v = row[(size_t)y * W + x];
        g[i] = v == 0xFFFF ? DIST_INF : v;
    }
    int a_f = DIST_INF, a_b = DIST_INF;
#pragma unroll
    for (int i = 0; i < DCOLS_ROWS; ++i)
        if (y0 + i < y1) { a_f = min(a_f, g[i] + (y1 - 1 - (y0 + i))); a_b = min(a_b, g[i] + i); }
    s_fwd[ty][tx] = a_f;
    s_bwd[ty][tx] = a_b;
    __syncthreads();
    if (!live) return;
    int E = DIST_INF;
    for (int s = 0; s < ty; ++s) {
        const int sy0 = min(s * rows_per, H), sy1 = min(sy0 + rows_per, H);
        E = min(s_fwd[s][tx], E + (sy1 - sy0));
    }
    int B = DIST_INF;
    for (int s = DCOL_TY - 1; s > ty; --s) {
        const int sy0 = min(s * rows_per, H), sy1 = min(sy0 + rows_per, H);
        B = min(s_bwd[s][tx], B + (sy1 - sy0));
    }
    int f[DCOLS_ROWS];
    int d = E;
#pragma unroll
    for (int i = 0; i < DCOLS_ROWS; ++i) { d = min(g[i], d + 1); f[i] = d; }
    int b = B;
    int v[DCOLS_ROWS];
#pragma unroll
    for (int i = DCOLS_ROWS - 1; i >= 0; --i) {
        if (y0 + i < y1) { b = min(g[i], b + 1); v[i] = min(min(f[i], b), 0xFFFF); } else v[i] = 0xFFFF;
    }
    float fo[DCOLS_ROWS];
    if (out) {
#pragma unroll
        for (int i = 0; i < DCOLS_ROWS; ++i) fo[i] = (y0 + i < y1 && v[i] < 0xFFFF) ? lut[v[i]] : -1.0f;
    }
#pragma unroll
    for (int i = 0; i < DCOLS_ROWS; ++i)
        if (y0 + i < y1) {
            const size_t at = (size_t)(y0 + i) * W + x;
            l1[at] = (uint16_t)v[i];
            if (out) out[at] = fo[i];
        }
}

// ---- column pass for large grids ------------------------------------------------------------------------------------
// k_dist_cols has W / 64 workgroups: 64 of them on a 4096-wide grid, a quarter of the CUs, each streaming narrow 128-byte
// rows (measured 0.62 ms = 0.45 TB/s of traffic at 4096^2).  Here a workgroup owns 128 columns x 128 rows (a "macro
// strip"; 64 lanes x 2 columns, 16 thread rows x 8 rows, everything a thread touches stays in registers):
//   k_dist_cols_summary  per macro strip and column: distance to the nearest source inside the strip, seen from its last
//                        row (downward chain) and from its first row (upward chain)             -- reads `row` once
//   k_dist_cols_carry    (grids up to 4096 rows) per column the summaries turned into the carries entering each strip
//   k_dist_cols_apply    takes the carry entering its strip (or chains the other strips' summaries itself), then the
//                        same down / up passes as k_dist_cols                                   -- reads `row` once
// 4 B read + 10 B written per cell instead of 6 + 12, and (W / 128) x (H / 128) workgroups.
#define DC2_TX 64
#define DC2_TY 16
#define DC2_SUB 8                              // rows per thread
#define DC2_ROWS (DC2_TY * DC2_SUB)            // rows per macro strip
#define DC2_STAGE 32                           // macro strips whose summaries k_dist_cols_apply stages in LDS (grids up to 4096 rows)

__device__ __forceinline__ void dc2_load(const uint16_t* __restrict__ row, int W, int H, int x, int y0, int g[DC2_SUB][2])
{
#pragma unroll
    for (int i = 0; i < DC2_SUB; ++i) {
        unsigned int v = 0xFFFFFFFFu;
        if (y0 + i < H) v = *(const unsigned int*)(row + (size_t)(y0 + i) * W + x);
        const int a = (int)(v & 0xFFFFu), b = (int)(v >> 16);
        g[i][0] = a == 0xFFFF ? DIST_INF : a;
        g[i][1] = b == 0xFFFF ? DIST_INF : b;
    }
}

__global__ __launch_bounds__(DC2_TX * DC2_TY) void k_dist_cols_summary(dist_batch db, int W, int H)
{
    const uint16_t* __restrict__ row = db.row[blockIdx.z];
    int* __restrict__ sum_f = db.sum_f[blockIdx.z];
    int* __restrict__ sum_b = db.sum_b[blockIdx.z];
    __shared__ int s_f[DC2_TY][2 * DC2_TX];
    __shared__ int s_b[DC2_TY][2 * DC2_TX];
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int x = blockIdx.x * 2 * DC2_TX + 2 * tx;
    const int Y0 = blockIdx.y * DC2_ROWS;
    const int y0 = Y0 + ty * DC2_SUB;
    int a_f[2] = {DIST_INF, DIST_INF}, a_b[2] = {DIST_INF, DIST_INF};
    if (x < W) {
        int g[DC2_SUB][2];
        dc2_load(row, W, H, x, y0, g);
        const int y1 = min(H, y0 + DC2_SUB);
#pragma unroll
        for (int i = 0; i < DC2_SUB; ++i)
            if (y0 + i < H)
                for (int c = 0; c < 2; ++c) {
                    a_f[c] = min(a_f[c], g[i][c] + (y1 - 1 - (y0 + i)));
                    a_b[c] = min(a_b[c], g[i][c] + i);
                }
    }
    for (int c = 0; c < 2; ++c) { s_f[ty][2 * tx + c] = a_f[c]; s_b[ty][2 * tx + c] = a_b[c]; }
    __syncthreads();
    if (ty == 0 && x < W) {
        for (int c = 0; c < 2; ++c) {
            int F = DIST_INF, B = DIST_INF;
            for (int s = 0; s < DC2_TY; ++s) {
                const int sy0 = Y0 + s * DC2_SUB, len = max(0, min(H, sy0 + DC2_SUB) - sy0);
                if (len > 0) F = min(s_f[s][2 * tx + c], F + len);
            }
            for (int s = DC2_TY - 1; s >= 0; --s) {
                const int sy0 = Y0 + s * DC2_SUB, len = max(0, min(H, sy0 + DC2_SUB) - sy0);
                if (len > 0) B = min(s_b[s][2 * tx + c], B + len);
            }
            sum_f[(size_t)blockIdx.y * W + x + c] = min(F, DIST_INF);
            sum_b[(size_t)blockIdx.y * W + x + c] = min(B, DIST_INF);
        }
    }
}

// Between the two: the summaries of a column's macro strips turned, in place, into the carries ENTERING each strip -- sum_f[m] =
// distance at the last row of strip m - 1 seen from above (the downward chain through strips 0 .. m - 1), sum_b[m] = distance at
// the first row of strip m + 1 seen from below.  A thread per column, every summary of the column requested before the chains
// run (up to DC2_STAGE strips: grids up to 4096 rows).  1 MB at 4096^2; with it k_dist_cols_apply reads two values per column
// instead of staging 32 KB of other strips' summaries per workgroup and chaining through them in every thread.
__global__ __launch_bounds__(256) void k_dist_cols_carry(dist_batch db, int W, int H)
{
    int* __restrict__ sum_f = db.sum_f[blockIdx.z];
    int* __restrict__ sum_b = db.sum_b[blockIdx.z];
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int nmacro = (H + DC2_ROWS - 1) / DC2_ROWS;
    if (x >= W) return;
    int f[DC2_STAGE], b[DC2_STAGE];
#pragma unroll
    for (int m = 0; m < DC2_STAGE; ++m) {
        f[m] = DIST_INF; b[m] = DIST_INF;
        if (m < nmacro) { f[m] = sum_f[(size_t)m * W + x]; b[m] = sum_b[(size_t)m * W + x]; }
    }
    int e = DIST_INF;
#pragma unroll
    for (int m = 0; m < DC2_STAGE; ++m)
        if (m < nmacro) {
            const int len = min(H, (m + 1) * DC2_ROWS) - m * DC2_ROWS;
            sum_f[(size_t)m * W + x] = min(e, DIST_INF);
            e = min(f[m], e + len);
        }
    int c = DIST_INF;
#pragma unroll
    for (int m = DC2_STAGE - 1; m >= 0; --m)
        if (m < nmacro) {
            const int len = min(H, (m + 1) * DC2_ROWS) - m * DC2_ROWS;
            sum_b[(size_t)m * W + x] = min(c, DIST_INF);
            c = min(b[m], c + len);
        }
}

// CARRIED: sum_f / sum_b hold the carries entering each strip (k_dist_cols_carry ran); otherwise the strips' summaries
template <bool CARRIED>
__global__ __launch_bounds__(DC2_TX * DC2_TY) void k_dist_cols_apply(dist_batch db, int W, int H)
{
    const uint16_t* __restrict__ row = db.row[blockIdx.z];
    const int* __restrict__ sum_f = db.sum_f[blockIdx.z];
    const int* __restrict__ sum_b = db.sum_b[blockIdx.z];
    uint16_t* __restrict__ l1 = db.l1[blockIdx.z];
    __shared__ int s_f[DC2_TY][2 * DC2_TX];
    __shared__ int s_b[DC2_TY][2 * DC2_TX];
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int x = blockIdx.x * 2 * DC2_TX + 2 * tx;
    const int Y0 = blockIdx.y * DC2_ROWS;
    const int y0 = Y0 + ty * DC2_SUB;
    const int nmacro = (H + DC2_ROWS - 1) / DC2_ROWS;
    const bool live = x < W;
    int g[DC2_SUB][2];
    int a_f[2] = {DIST_INF, DIST_INF}, a_b[2] = {DIST_INF, DIST_INF};
    if (live) {
        dc2_load(row, W, H, x, y0, g);
        const int y1 = min(H, y0 + DC2_SUB);
#pragma unroll
        for (int i = 0; i < DC2_SUB; ++i)
            if (y0 + i < H)
                for (int c = 0; c < 2; ++c) {
                    a_f[c] = min(a_f[c], g[i][c] + (y1 - 1 - (y0 + i)));
                    a_b[c] = min(a_b[c], g[i][c] + i);
                }
    }
    for (int c = 0; c < 2; ++c) { s_f[ty][2 * tx + c] = a_f[c]; s_b[ty][2 * tx + c] = a_b[c]; }
    // The other strips' summaries of this workgroup's 128 columns, staged in LDS with every load in flight at once: chained
    // straight from memory they were up to 2 x 31 dependent L2 round trips per thread -- 20 of the kernel's 45 us at 4096^2.
    __shared__ int s_sf[CARRIED ? 1 : DC2_STAGE][2 * DC2_TX], s_sb[CARRIED ? 1 : DC2_STAGE][2 * DC2_TX];
    const bool staged = !CARRIED && nmacro <= DC2_STAGE;
    int2 cf = make_int2(DIST_INF, DIST_INF), cb = make_int2(DIST_INF, DIST_INF);
    if (CARRIED && live) {                                      // (x is even, W is even: one 8-byte load per array)
        cf = *(const int2*)(sum_f + (size_t)blockIdx.y * W + x);
        cb = *(const int2*)(sum_b + (size_t)blockIdx.y * W + x);
    }
    if (staged) {
        const int t = ty * DC2_TX + tx;
        for (int i = t; i < nmacro * 2 * DC2_TX; i += DC2_TX * DC2_TY) {
            const int m = i / (2 * DC2_TX), cx = i - m * (2 * DC2_TX);
            const int xg = (int)blockIdx.x * 2 * DC2_TX + cx;
            s_sf[m][cx] = xg < W ? sum_f[(size_t)m * W + xg] : DIST_INF;
            s_sb[m][cx] = xg < W ? sum_b[(size_t)m * W + xg] : DIST_INF;
        }
    }
    __syncthreads();
    if (live && y0 < H) {
    int E[2], B[2];
    for (int c = 0; c < 2; ++c) {
        // carry entering the macro strip from above (distance at row Y0 - 1) and from below (distance at row Y1)
        int e = DIST_INF, b = DIST_INF;
        if (CARRIED) { e = c ? cf.y : cf.x; b = c ? cb.y : cb.x; }
        else {
        for (int m = 0; m < (int)blockIdx.y; ++m) {
            const int len = min(H, (m + 1) * DC2_ROWS) - m * DC2_ROWS;
            e = min(staged ? s_sf[m][2 * tx + c] : sum_f[(size_t)m * W + x + c], e + len);
        }
        for (int m = nmacro - 1; m > (int)blockIdx.y; --m) {
            const int len = min(H, (m + 1) * DC2_ROWS) - m * DC2_ROWS;
            b = min(staged ? s_sb[m][2 * tx + c] : sum_b[(size_t)m * W + x + c], b + len);
        }
        }
        // ... then through the thread rows above / below this one inside the strip
        for (int s = 0; s < ty; ++s) {
            const int sy0 = Y0 + s * DC2_SUB, len = max(0, min(H, sy0 + DC2_SUB) - sy0);
            if (len > 0) e = min(s_f[s][2 * tx + c], e + len);
        }
        for (int s = DC2_TY - 1; s > ty; --s) {
            const int sy0 = Y0 + s * DC2_SUB, len = max(0, min(H, sy0 + DC2_SUB) - sy0);
            if (len > 0) b = min(s_b[s][2 * tx + c], b + len);
        }
        E[c] = min(e, DIST_INF); B[c] = min(b, DIST_INF);
    }
    int f[DC2_SUB][2];
#pragma unroll
    for (int i = 0; i < DC2_SUB; ++i)
        for (int c = 0; c < 2; ++c) { E[c] = min(g[i][c], E[c] + 1); f[i][c] = E[c]; }
#pragma unroll
    for (int i = DC2_SUB - 1; i >= 0; --i) {
        if (y0 + i >= H) continue;
        int v[2];
        for (int c = 0; c < 2; ++c) { B[c] = min(g[i][c], B[c] + 1); v[c] = min(min(f[i][c], B[c]), 0xFFFF); }
        const size_t at = (size_t)(y0 + i) * W + x;
        *(unsigned int*)(l1 + at) = (unsigned int)v[0] | ((unsigned int)v[1] << 16);
    }
    }
}

// ---- whole-grid transform in ONE launch --------------------------------------------------------------------------------
// The L1 distance of a cell p to the sources of another 128 x 128 tile T depends on T only through a short summary, each part of
// it a piece of T's own distance transform D_T (the transform of T's cells alone):
//   T in p's row band      D_T along T's last and first column: a path from p enters T through one of them         (128 words per tile)
//   T in p's column band   D_T along T's last and first row                                                         (128 words)
//   T diagonal to p        |dx| + |dy| has fixed signs, so one number per quadrant: max (sx + sy), max (sx - sy), max (-sx + sy),
//                          max (-sx - sy) over T's sources                                                          (4 words)
// So a workgroup computes D_T of its tile (cells in; row pass, transposition through LDS, column pass: the distances stay in
// registers), publishes T's summary on the way, takes in the summaries of the other tiles -- the nearest outside source as seen
// from each cell of the ring around the tile: left and right per row, above and below per column, the diagonal tiles folded
// into the latter -- and finishes with four ramps per cell: d = min (D_T, left + c + 1, right + 128 - c, above + r + 1, below +
// 128 - r).  ONE hand-over, the intermediate `row` grid never exists: 1 B read and 2 B written per cell where the four-launch
// form moves 9.
// One launch: workgroup b owns tiles b, b + G, ... (at most DF_MAXK, their D_T all held in registers: grids up to 4096 x 4096;
// 2000 x 2000: one tile per CU).  Every published word is tag << 26 | payload with a tag per launch, so no fence orders data
// against a flag, and what a workgroup may wait for is spelled out at the wait: tiles are CLAIMED by a marker word, and a
// workgroup computes the summary of every tile it finds unclaimed itself before its first wait.
// All distances are pairs of uint16 in one register (two rows in the row pass, two columns in the column pass), 0xFFFF = no
// source, sums saturating: v_pk_add_u16 clamp / v_pk_min_u16, two cells per instruction.
#define DF_T 128                               // tile side
#define DF_NT 512                              // threads: row layout 64 row pairs x 8 threads x 16 columns; column layout 8 x 16 rows x 64 column pairs
#define DF_WORDS 256                           // band words per tile: [0, 128) rows, [128, 256) columns; then 4 quadrant words per tile, then a claim word per tile
#define DF_GP DF_T                             // LDS pitch of the tile of row distances (uint16): no padding, 16-byte chunks swizzled (df_at)
#define DF_TAG_SHIFT 26
#define DF_NOPOT 511u                          // 9-bit potentials (finite ones are at most 254)
#define DF_CORNER_BIAS 20000
#define DF_SPIN_CAP (1 << 18)
#define DF_MAX_SIDE 8176                       // quadrant numbers within 16 bits around the bias
#define DF_GSLOTS 4                            // 16-byte loads per thread and band in the gather: bands of up to 64 tiles
#define DF_SCAN 2                              // claim words per thread: DF_MAXK * DF_MAX_WGS tiles
#define DF_MAX_WGS 256
#define DF_MAXK 4
struct dist_fused_batch { unsigned int* words[DIST_MAX_BATCH]; unsigned int tag[DIST_MAX_BATCH]; unsigned int test_delay; };

typedef unsigned short df_u2 __attribute__((ext_vector_type(2)));
typedef short df_s2 __attribute__((ext_vector_type(2)));
#define DF_INF2 0xFFFFFFFFu
__device__ __forceinline__ unsigned int df_min(unsigned int a, unsigned int b)
{
    return __builtin_bit_cast(unsigned int, __builtin_elementwise_min(__builtin_bit_cast(df_u2, a), __builtin_bit_cast(df_u2, b)));
}
__device__ __forceinline__ unsigned int df_adds(unsigned int a, unsigned int b)          // saturating: 0xFFFF stays 0xFFFF
{
    return __builtin_bit_cast(unsigned int, __builtin_elementwise_add_sat(__builtin_bit_cast(df_u2, a), __builtin_bit_cast(df_u2, b)));
}
__device__ __forceinline__ unsigned int df_both(int v) { return (unsigned int)v | ((unsigned int)v << 16); }
__device__ __forceinline__ unsigned int df_pair(int lo, int hi) { return (unsigned int)min(lo, 0xFFFF) | ((unsigned int)min(hi, 0xFFFF) << 16); }

typedef unsigned int df_u4 __attribute__((ext_vector_type(4)));
// 16 bytes through the L2, not waited for: the caller's `s_waitcnt vmcnt(0)` statement names the registers, and nothing that could
// make the compiler move them may stand between the two (the operand is read-write so that a load under a condition redefines the
// SAME register, not a copy the compiler merges afterwards -- the data lands asynchronously, a copy made before the wait is stale)
__device__ __forceinline__ void df_load16(df_u4& w, const unsigned int* p) { asm volatile("global_load_dwordx4 %0, %1, off sc1" : "+v"(w) : "v"(p) : "memory"); }

// A workgroup barrier that waits for the wave's LDS traffic only: __syncthreads() also waits for every global access the wave has
// in flight -- the tile's cells on their way in, the published words on their way out -- although nothing behind these barriers
// depends on them (the loads are waited for where their registers are used; the words are read by other workgroups, and each
// carries its own tag).
__device__ __forceinline__ void df_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// max over the wave of non-negative values, in lane 63 (DPP: shifts inside the 16-lane rows fill with zero, then the rows' last
// lanes are handed on) -- six VALU pairs and no LDS round trip, where a shuffle tree makes six dependent ones
__device__ __forceinline__ int df_wave_max(int v)
{
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, true));      // row_bcast:15 into rows 1 and 3
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, true));      // row_bcast:31 into rows 2 and 3
    return v;
}

template <int OFF>
__device__ __forceinline__ void df_scan_step(unsigned int& inc_f, unsigned int& inc_b, int rq)
{
    const unsigned int t = (unsigned int)__builtin_amdgcn_update_dpp((int)DF_INF2, (int)inc_f, 0x110 + OFF, 0xf, 0xf, false);     // row_shr
    const unsigned int u = (unsigned int)__builtin_amdgcn_update_dpp((int)DF_INF2, (int)inc_b, 0x100 + OFF, 0xf, 0xf, false);     // row_shl
    if (rq >= OFF) inc_f = df_min(inc_f, df_adds(t, df_both(16 * OFF)));
    if (rq + OFF < 8) inc_b = df_min(inc_b, df_adds(u, df_both(16 * OFF)));
}

// The row pass of a tile in the row layout: the thread holds the same 16 columns of two rows (a: 16 bytes of the even row, b: of
// the odd row), its 8-lane group the two whole tile rows.  v[i]: the distances of column i of both rows to the nearest source of
// their own row inside the tile; row_f / row_b (valid in every lane of the group): those at the tile's last / first column.
__device__ __forceinline__ void df_row_pass(const int4 a, const int4 b, int rq, unsigned int v[16], unsigned int& row_f, unsigned int& row_b)
{
    const unsigned int aw[4] = {(unsigned int)a.x, (unsigned int)a.y, (unsigned int)a.z, (unsigned int)a.w};
    const unsigned int bw[4] = {(unsigned int)b.x, (unsigned int)b.y, (unsigned int)b.z, (unsigned int)b.w};
    const unsigned int one = df_both(1);
    unsigned int s[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        // byte i of both rows into the high bytes of the two halves; its sign spread over the half: 0xFFFF free, 0 a source
        // (is_cell_occupied: logOdds >= 0, obstacle_distance_grid.cpp:125-128)
        const unsigned int p = __builtin_amdgcn_perm(bw[i >> 2], aw[i >> 2], 0x000c000cu | ((unsigned int)(i & 3) << 8) | ((unsigned int)(4 + (i & 3)) << 24));
        s[i] = __builtin_bit_cast(unsigned int, __builtin_bit_cast(df_s2, p) >> 15);
    }
    unsigned int d = DF_INF2;
#pragma unroll
    for (int i = 0; i < 16; ++i) { d = df_min(df_adds(d, one), s[i]); v[i] = d; }
    unsigned int inc_f = d;                                  // distance at the thread's last column to its own nearest source
    d = DF_INF2;
#pragma unroll
    for (int i = 15; i >= 0; --i) { d = df_min(df_adds(d, one), s[i]); v[i] = df_min(v[i], d); }
    unsigned int inc_b = d;
    // min-plus scans over the 8 threads of the row pair (row_shr / row_shl stay inside the 16-lane row; lanes whose source would
    // lie in the other group keep their value)
    df_scan_step<1>(inc_f, inc_b, rq); df_scan_step<2>(inc_f, inc_b, rq); df_scan_step<4>(inc_f, inc_b, rq);
    unsigned int cf = (unsigned int)__builtin_amdgcn_update_dpp((int)DF_INF2, (int)inc_f, 0x111, 0xf, 0xf, false);
    unsigned int cb = (unsigned int)__builtin_amdgcn_update_dpp((int)DF_INF2, (int)inc_b, 0x101, 0xf, 0xf, false);
    if (rq == 0) cf = DF_INF2;                               // distance at the column just left of the thread's first
    if (rq == 7) cb = DF_INF2;
    // lane 7 of the group holds the whole row's forward value, lane 0 the backward one: to every lane
    row_f = (unsigned int)__shfl((int)inc_f, (threadIdx.x & 56) | 7, 64);
    row_b = (unsigned int)__shfl((int)inc_b, threadIdx.x & 56, 64);
#pragma unroll
    for (int i = 0; i < 16; ++i) { cf = df_adds(cf, one); v[i] = df_min(v[i], cf); }
#pragma unroll
    for (int i = 15; i >= 0; --i) { cb = df_adds(cb, one); v[i] = df_min(v[i], cb); }
}

// The LDS tile of row distances: uint16, row-major, 256 bytes a row, the sixteen 16-byte chunks of rows 2, 3, 6, 7, ... swapped
// in pairs (chunk ^ 1).  The row layout writes a chunk per lane (ds_write_b128): lanes of one row pair cover every other chunk,
// and with the swap the lanes of the NEXT row pair cover the chunks between -- 16 lanes, 64 banks, no conflict (unswizzled, padded
// or not, rows two apart start on the same bank however the pitch is chosen as long as chunks stay 16-byte aligned: the writes of
// the eight waves took 1.7 us of an in-tile pass of 4.4).  The column layout reads a dword per lane along a row: any order of
// the chunks is conflict-free.
__device__ __forceinline__ int df_at(int row, int col) { return row * DF_GP + ((((col >> 3) ^ ((row >> 1) & 1)) << 3) | (col & 7)); }

// the row distances of the thread's two rows into the LDS tile
__device__ __forceinline__ void df_rows_to_lds(uint16_t* s_g, int rp, int rq, const unsigned int v[16])
{
    unsigned int lo[8], hi[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        lo[j] = __builtin_amdgcn_perm(v[2 * j + 1], v[2 * j], 0x05040100u);       // even row: columns 2j, 2j + 1
        hi[j] = __builtin_amdgcn_perm(v[2 * j + 1], v[2 * j], 0x07060302u);       // odd row
    }
    *(uint4*)&s_g[df_at(2 * rp, 16 * rq)] = make_uint4(lo[0], lo[1], lo[2], lo[3]); *(uint4*)&s_g[df_at(2 * rp, 16 * rq + 8)] = make_uint4(lo[4], lo[5], lo[6], lo[7]);
    *(uint4*)&s_g[df_at(2 * rp + 1, 16 * rq)] = make_uint4(hi[0], hi[1], hi[2], hi[3]); *(uint4*)&s_g[df_at(2 * rp + 1, 16 * rq + 8)] = make_uint4(hi[4], hi[5], hi[6], hi[7]);
}

// K: tiles per workgroup (tiles b + k G, k < K, those below T)
template <int K>
__global__ __launch_bounds__(DF_NT) void k_dist_fused(dist_batch db, dist_fused_batch fb, int W, int H)
{
    const int z = blockIdx.z;
    const int t = threadIdx.x;
    const int nJ = (W + DF_T - 1) / DF_T, nI = (H + DF_T - 1) / DF_T, T = nI * nJ, G = (int)gridDim.x;
    // Workgroup b's k-th tile.  Workgroups are placed on the 8 XCDs round-robin, and a tile's 128-byte rows straddle two cache
    // lines unless W is a multiple of 128: an XCD takes a run of consecutive tiles (whole row bands where T / 8 is a multiple of
    // the bands' length), so that the lines two neighbours share come through one L2 once -- not through two L2s from memory
    // twice (the tiles' loads, all issued in the launch's first half microsecond, are a burst the fabric has to carry).
    const int per_xcd = (T + 7) >> 3;
    auto tile_of = [&](int k) -> int {
        const int q = ((int)blockIdx.x >> 3) + k * (G >> 3);
        const int tile = ((int)blockIdx.x & 7) * per_xcd + q;
        return q < per_xcd && tile < T ? tile : -1;
    };
    __shared__ __attribute__((aligned(16))) uint16_t s_g[DF_T * DF_GP];
    __shared__ unsigned int s_f[8][64], s_b[8][64], s_ein[8][64], s_bin[8][64];
    __shared__ unsigned int s_rw[DF_T];                 // per tile row: D_T's row pass at the last (low half) / first (high half) column
    __shared__ __attribute__((aligned(16))) uint16_t s_part[4][16][DF_T];    // the ring by slice of the gather: left, right, above, below (0xFFFF: nothing)
    __shared__ unsigned int s_L2[DF_T], s_R2[DF_T], s_E2[64], s_B2[64];
    __shared__ __attribute__((aligned(16))) int s_Qw[8][4];          // per wave: its rows' share of the tile's quadrant numbers
    __shared__ int s_fail, s_open;
    unsigned int* state = db.state[z];
    unsigned int* words = fb.words[z];
    unsigned int* corners = words + (size_t)T * DF_WORDS;           // [T][4]
    unsigned int* claims = corners + (size_t)T * 4;                 // [T]: the tag of the launch in which the tile's summary was last taken on
    const unsigned int tag = fb.tag[z];
    const int8_t* __restrict__ cells = db.cells[z];
    const int rp = t >> 3, rq = t & 7;                  // row layout: rows 2 rp, 2 rp + 1 of the tile, columns 16 rq .. 16 rq + 15
    const int tx = t & 63, ty = t >> 6;                 // column layout: columns 2 tx, 2 tx + 1, rows 16 ty .. 16 ty + 15
    const unsigned int one = df_both(1);
#ifdef DF_STAMPS
    unsigned long long* stamps = (unsigned long long*)(claims + T);
#define DF_NOW(var) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define DFS(slot, k) do { unsigned long long now_; DF_NOW(now_); if (t == 0) stamps[8 * (slot) + (k)] = now_; } while (0)
    unsigned long long* wstamps = stamps + 16 * (size_t)T;           // [T][8 waves][4]
#define DFW(tile, k) do { unsigned long long now_; DF_NOW(now_); if ((t & 63) == 0) wstamps[32 * (tile) + 4 * (t >> 6) + (k)] = now_; } while (0)
#else
#define DFS(slot, k) do { } while (0)
#define DFW(tile, k) do { } while (0)
#endif
    if (t == 0) { s_fail = 0; s_open = 0; if (blockIdx.x == 0) dist_plan_full(state, W, H); }
    if (fb.test_delay && (blockIdx.x & 1)) for (unsigned int i = 0; i < fb.test_delay; ++i) __builtin_amdgcn_s_sleep(64);     // (tests: a workgroup that starts late)
    df_barrier();
    auto load_tile = [&](int tile, int4& a, int4& b) {
        const int I = tile / nJ, J = tile - I * nJ;
        const int x = J * DF_T + 16 * rq, y = I * DF_T + 2 * rp;
        a = make_int4(-1, -1, -1, -1); b = make_int4(-1, -1, -1, -1);          // 0xFF bytes: free cells (log-odds < 0), i.e. no source
        if (x < W && y < H) a = *(const int4*)(cells + (size_t)y * W + x);
        if (x < W && y + 1 < H) b = *(const int4*)(cells + (size_t)(y + 1) * W + x);
    };
    // a claim on a tile's summary: a plain marker, not an election -- two workgroups that take on the same tile publish the same words
    auto claim = [&](int tile) { if (t == 0) __hip_atomic_store(&claims[tile], tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    // D_T of the tile whose cells the threads hold, into d0 (column layout), its summary published on the way.  Ends behind a
    // barrier.  scan: look at every tile's claim word on the way (the loads ride behind the column work); returns whether this
    // thread saw one without this launch's tag.
    auto in_tile = [&](int tile, const int4 a, const int4 b, unsigned int d0[16], bool scan) -> bool {
        const int I = tile / nJ, J = tile - I * nJ;
        const int X0 = J * DF_T, Y0 = I * DF_T;
        {
            unsigned int v[16], row_f, row_b;
#ifdef DF_STAMPS
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            DFW(tile, 0);
#endif
            df_row_pass(a, b, rq, v, row_f, row_b);
#ifdef DF_STAMPS
            asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
            asm volatile("" : "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]), "+v"(row_f), "+v"(row_b));
#endif
            DFS(tile, 1);
            DFW(tile, 1);
            df_rows_to_lds(s_g, rp, rq, v);
            DFW(tile, 2);
            // the two rows' ends, and the wave's share of the quadrant numbers: per row the extreme source columns decide (every
            // lane of a row pair's group holds them: reduced over the wave by DPP, one lane writes -- left to
            // atomicMax on an LDS word the compiler loops over the active lanes, 1.4 us here)
            int q[4] = {0, 0, 0, 0};                    // biased; 0: no source
            for (int c = 0; c < 2; ++c) {
                const int rf = (int)((row_f >> (16 * c)) & 0xFFFFu), rbk = (int)((row_b >> (16 * c)) & 0xFFFFu);
                if (rf != 0xFFFF) {
                    const int y = Y0 + 2 * rp + c, last = X0 + DF_T - 1 - rf, first = X0 + rbk;
                    q[0] = max(q[0], last + y + DF_CORNER_BIAS); q[1] = max(q[1], last - y + DF_CORNER_BIAS);
                    q[2] = max(q[2], -first + y + DF_CORNER_BIAS); q[3] = max(q[3], -first - y + DF_CORNER_BIAS);
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) q[k] = df_wave_max(q[k]);
            if (rq == 7) *(uint2*)&s_rw[2 * rp] = make_uint2((row_f & 0xFFFFu) | (row_b << 16), (row_f >> 16) | (row_b & 0xFFFF0000u));
            if ((t & 63) == 63) *(int4*)&s_Qw[t >> 6][0] = make_int4(q[0], q[1], q[2], q[3]);
        }
        DFW(tile, 3);
        df_barrier();
        DFS(tile, 2);
        unsigned int cl[DF_SCAN];
        if (scan) {
#pragma unroll
            for (int q = 0; q < DF_SCAN; ++q) cl[q] = t + q * DF_NT < T ? dst_load_u(&claims[t + q * DF_NT]) : tag;
        }
        {   // column pass, first half: per thread the ends of the two scans over its 16 rows
#pragma unroll
            for (int i = 0; i < 16; ++i) d0[i] = *(const unsigned int*)&s_g[df_at(16 * ty + i, 2 * tx)];
            unsigned int df_ = DF_INF2, dbk = DF_INF2;
#pragma unroll
            for (int i = 0; i < 16; ++i) df_ = df_min(df_adds(df_, one), d0[i]);
#pragma unroll
            for (int i = 15; i >= 0; --i) dbk = df_min(df_adds(dbk, one), d0[i]);
            s_f[ty][tx] = df_; s_b[ty][tx] = dbk;
        }
        df_barrier();
        DFS(tile, 3);
        if (t < 64) {
            // the carries entering each thread row of a column pair from above and from below; the ends of the chains are D_T at
            // the tile's last / first row: the column words
            unsigned int e = DF_INF2, bk = DF_INF2;
#pragma unroll
            for (int s2 = 0; s2 < 8; ++s2) { s_ein[s2][t] = e; e = df_min(s_f[s2][t], df_adds(e, df_both(16))); }
#pragma unroll
            for (int s2 = 7; s2 >= 0; --s2) { s_bin[s2][t] = bk; bk = df_min(s_b[s2][t], df_adds(bk, df_both(16))); }
            unsigned int c2[2];
            for (int c = 0; c < 2; ++c)
                c2[c] = (tag << DF_TAG_SHIFT) | (min((bk >> (16 * c)) & 0xFFFFu, DF_NOPOT) << 9) | min((e >> (16 * c)) & 0xFFFFu, DF_NOPOT);
            __hip_atomic_store((unsigned long long*)&words[(size_t)tile * DF_WORDS + DF_T + 2 * t], (unsigned long long)c2[0] | ((unsigned long long)c2[1] << 32),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (t < 128) {
            // D_T along the tile's last / first column: the row pass's ends spread over the rows (the column pass restricted to
            // those two columns) -- lane l: rows 2l, 2l + 1; low half the last column, high half the first
            const int l = t - 64;
            unsigned int x0 = s_rw[2 * l], x1 = s_rw[2 * l + 1];
            x1 = df_min(x1, df_adds(x0, one));
            unsigned int S = x1;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const unsigned int u = (unsigned int)__shfl_up((int)S, off, 64); if (l >= off) S = df_min(S, df_adds(u, df_both(2 * off))); }
            unsigned int c = (unsigned int)__shfl_up((int)S, 1, 64);
            if (l == 0) c = DF_INF2;
            x0 = df_min(x0, df_adds(c, one)); x1 = df_min(x1, df_adds(c, df_both(2)));
            x0 = df_min(x0, df_adds(x1, one));
            S = x0;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const unsigned int u = (unsigned int)__shfl_down((int)S, off, 64); if (l + off < 64) S = df_min(S, df_adds(u, df_both(2 * off))); }
            c = (unsigned int)__shfl_down((int)S, 1, 64);
            if (l == 63) c = DF_INF2;
            x1 = df_min(x1, df_adds(c, one)); x0 = df_min(x0, df_adds(c, df_both(2)));
            const unsigned int w0 = (tag << DF_TAG_SHIFT) | (min(x0 >> 16, DF_NOPOT) << 9) | min(x0 & 0xFFFFu, DF_NOPOT);
            const unsigned int w1 = (tag << DF_TAG_SHIFT) | (min(x1 >> 16, DF_NOPOT) << 9) | min(x1 & 0xFFFFu, DF_NOPOT);
            __hip_atomic_store((unsigned long long*)&words[(size_t)tile * DF_WORDS + 2 * l], (unsigned long long)w0 | ((unsigned long long)w1 << 32),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (t < 132) {
            int qm = 0;
#pragma unroll
            for (int w = 0; w < 8; ++w) qm = max(qm, s_Qw[w][t - 128]);
            __hip_atomic_store(&corners[(size_t)tile * 4 + (t - 128)], (tag << DF_TAG_SHIFT) | (unsigned int)qm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        df_barrier();
        DFS(tile, 4);
        {   // column pass, second half, in place: down with the carry from above, then up (the forward values bound the true
            // ones from above, so the second sweep may start from them)
            unsigned int e = s_ein[ty][tx], bk = s_bin[ty][tx];
#pragma unroll
            for (int i = 0; i < 16; ++i) { e = df_min(df_adds(e, one), d0[i]); d0[i] = e; }
#pragma unroll
            for (int i = 15; i >= 0; --i) { bk = df_min(df_adds(bk, one), d0[i]); d0[i] = bk; }
        }
        bool open = false;
        if (scan) {
#pragma unroll
            for (int q = 0; q < DF_SCAN; ++q) open = open || cl[q] != tag;
        }
        DFS(tile, 7);
        return open;
    };

    // ================================================================================ D_T of the workgroup's tiles
    unsigned int d0[K][16];
    bool early_open = false;
    {
        int4 ra = make_int4(-1, -1, -1, -1), rb = ra, na = ra, nb = ra;
        if (tile_of(0) >= 0) load_tile(tile_of(0), ra, rb);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int tile = tile_of(k);
            if (tile >= 0) {
                DFS(tile, 0);
                claim(tile);
                const int next = k + 1 < K ? tile_of(k + 1) : -1;
                if (next >= 0) load_tile(next, na, nb);                          // (in flight behind this tile's work)
                const bool o = in_tile(tile, ra, rb, d0[k], next < 0);           // (the last one: by then the others' claims are out)
                if (next < 0) early_open = o;
                ra = na; rb = nb;
            }
        }
    }
    // ==================================================================================== the other tiles' summaries; the ramps
    // A workgroup waits for words of EVERY other tile.  Waiting is only safe for tiles some running workgroup has taken on (it
    // publishes without waiting for anybody); a tile nobody has claimed yet belongs to a workgroup that may not become resident
    // while this one spins -- so before its first wait a workgroup looks at every claim word and computes the summaries of the
    // unclaimed tiles itself.  The launch cannot deadlock however few of its workgroups are resident together.
    if (early_open) s_open = 1;
    df_barrier();
    bool all_claimed = s_open == 0;                     // every claim word carried the tag when the last tile's pass looked
    df_barrier();
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int tile = tile_of(k);
        if (tile < 0) break;
        const int I = tile / nJ, J = tile - I * nJ;
        const int X0 = J * DF_T, Y0 = I * DF_T;
        DFS(T + tile, 0);
        if (t == 0) s_open = T;
        df_barrier();
        bool failed = false;
        // The words this tile needs, every load of a thread in flight before the first is looked at (16 bytes each: four rows or
        // four columns of one tile; L2-served, never from this CU's L1 -- a word polled once too early would be polled from there
        // forever).  slice = t >> 5 takes the tiles slice, slice + 16, ... of the row band and of the column band, u = t & 31
        // their entries 4u .. 4u + 3; thread t the diagonal tiles t, t + 512 (one quadrant number each).
        const int slice = t >> 5, u = t & 31;
        const unsigned int* prow[DF_GSLOTS];
        const unsigned int* pcol[DF_GSLOTS];
        const unsigned int* pcor[2];
        df_u4 wrow[DF_GSLOTS], wcol[DF_GSLOTS];
        unsigned int wcor[2];
#pragma unroll
        for (int q = 0; q < DF_GSLOTS; ++q) {
            const int j2 = slice + 16 * q, i2 = slice + 16 * q;
            prow[q] = j2 < nJ && j2 != J ? &words[(size_t)(I * nJ + j2) * DF_WORDS + 4 * u] : nullptr;
            pcol[q] = i2 < nI && i2 != I ? &words[(size_t)(i2 * nJ + J) * DF_WORDS + DF_T + 4 * u] : nullptr;
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int k2 = t + q * DF_NT, i2 = k2 / nJ, j2 = k2 - i2 * nJ;
            pcor[q] = k2 < T && i2 != I && j2 != J ? &corners[(size_t)k2 * 4 + (i2 < I ? (j2 < J ? 0 : 2) : (j2 < J ? 1 : 3))] : nullptr;
        }
#pragma unroll
        for (int q = 0; q < DF_GSLOTS; ++q) { wrow[q] = df_u4{0u, 0u, 0u, 0u}; wcol[q] = wrow[q]; if (prow[q]) df_load16(wrow[q], prow[q]); if (pcol[q]) df_load16(wcol[q], pcol[q]); }
#pragma unroll
        for (int q = 0; q < 2; ++q) wcor[q] = pcor[q] ? dst_load_u(pcor[q]) : 0u;
        if (!all_claimed) {
            // (uniform over the workgroup: at most once per workgroup.  The words asked for above are awaited first: the registers
            // they land in are the compiler's to move once the code below needs room)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(wrow[0]), "+v"(wrow[1]), "+v"(wrow[2]), "+v"(wrow[3]), "+v"(wcol[0]), "+v"(wcol[1]), "+v"(wcol[2]), "+v"(wcol[3]) :: "memory");
            const int first = tile_of(0);                   // (workgroups start their search at their own tile: they take different ones)
            while (true) {
                for (int k2 = t; k2 < T; k2 += DF_NT) if (dst_load_u(&claims[k2]) != tag) atomicMin(&s_open, k2 >= first ? k2 - first : k2 + T - first);
                df_barrier();
                const int rot = s_open;
                df_barrier();
                if (rot >= T) break;
                const int open = rot + first < T ? rot + first : rot + first - T;
                if (t == 0) s_open = T;
                int4 ha, hb;
                unsigned int scratch[16];
                load_tile(open, ha, hb);
                claim(open);
                if (t == 0) atomicAdd(&state[DST_HELPED], 1u);
                in_tile(open, ha, hb, scratch, false);
                df_barrier();
            }
            all_claimed = true;
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(wrow[0]), "+v"(wrow[1]), "+v"(wrow[2]), "+v"(wrow[3]), "+v"(wcol[0]), "+v"(wcol[1]), "+v"(wcol[2]), "+v"(wcol[3]) :: "memory");
        DFS(T + tile, 4);
        // the words that have not come yet: asked for again together, not one after the other
        {
            auto ready = [&](const df_u4& w) { return (w.x >> DF_TAG_SHIFT) == tag && (w.y >> DF_TAG_SHIFT) == tag && (w.z >> DF_TAG_SHIFT) == tag && (w.w >> DF_TAG_SHIFT) == tag; };
            unsigned int pend = 0;
#pragma unroll
            for (int q = 0; q < DF_GSLOTS; ++q) { if (prow[q] && !ready(wrow[q])) pend |= 1u << q; if (pcol[q] && !ready(wcol[q])) pend |= 1u << (8 + q); }
#pragma unroll
            for (int q = 0; q < 2; ++q) if (pcor[q] && (wcor[q] >> DF_TAG_SHIFT) != tag) pend |= 1u << (16 + q);
            int spins = 0;
            while (pend) {
                if (++spins > DF_SPIN_CAP) { failed = true; break; }
                __builtin_amdgcn_s_sleep(2);
#pragma unroll
                for (int q = 0; q < DF_GSLOTS; ++q) { if ((pend >> q) & 1u) df_load16(wrow[q], prow[q]); if ((pend >> (8 + q)) & 1u) df_load16(wcol[q], pcol[q]); }
#pragma unroll
                for (int q = 0; q < 2; ++q) if ((pend >> (16 + q)) & 1u) wcor[q] = dst_load_u(pcor[q]);
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(wrow[0]), "+v"(wrow[1]), "+v"(wrow[2]), "+v"(wrow[3]), "+v"(wcol[0]), "+v"(wcol[1]), "+v"(wcol[2]), "+v"(wcol[3]) :: "memory");
#pragma unroll
                for (int q = 0; q < DF_GSLOTS; ++q) { if (((pend >> q) & 1u) && ready(wrow[q])) pend &= ~(1u << q); if (((pend >> (8 + q)) & 1u) && ready(wcol[q])) pend &= ~(1u << (8 + q)); }
#pragma unroll
                for (int q = 0; q < 2; ++q) if (((pend >> (16 + q)) & 1u) && (wcor[q] >> DF_TAG_SHIFT) == tag) pend &= ~(1u << (16 + q));
            }
        }
        DFS(T + tile, 5);
        {   // per slice: the nearest outside source as seen from the ring, entries 4u .. 4u + 3 (a potential p of a tile n tiles away:
            // p + 128 n).  A slot's tile lies on ONE side of this tile: one field of each word counts.
            unsigned int L[4] = {0xFFFFu, 0xFFFFu, 0xFFFFu, 0xFFFFu}, R[4] = {0xFFFFu, 0xFFFFu, 0xFFFFu, 0xFFFFu};
            unsigned int E[4] = {0xFFFFu, 0xFFFFu, 0xFFFFu, 0xFFFFu}, B[4] = {0xFFFFu, 0xFFFFu, 0xFFFFu, 0xFFFFu};
#pragma unroll
            for (int q = 0; q < DF_GSLOTS; ++q) {
                const int o2 = slice + 16 * q;                                       // the slot's tile in the row band / in the column band
                if (prow[q]) {
                    const unsigned int wr[4] = {wrow[q].x, wrow[q].y, wrow[q].z, wrow[q].w};
                    const bool left = o2 < J;
                    const unsigned int sh = left ? 0u : 9u, add = (unsigned int)(DF_T * (left ? J - 1 - o2 : o2 - J - 1));
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const unsigned int f = (wr[e] >> sh) & 0x1FFu, c = f == DF_NOPOT ? 0xFFFFu : f + add;
                        if (left) L[e] = min(L[e], c); else R[e] = min(R[e], c);
                    }
                }
                if (pcol[q]) {
                    const unsigned int wc[4] = {wcol[q].x, wcol[q].y, wcol[q].z, wcol[q].w};
                    const bool above = o2 < I;
                    const unsigned int sh = above ? 0u : 9u, add = (unsigned int)(DF_T * (above ? I - 1 - o2 : o2 - I - 1));
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const unsigned int f = (wc[e] >> sh) & 0x1FFu, c = f == DF_NOPOT ? 0xFFFFu : f + add;
                        if (above) E[e] = min(E[e], c); else B[e] = min(B[e], c);
                    }
                }
            }
            *(uint2*)&s_part[0][slice][4 * u] = make_uint2(L[0] | (L[1] << 16), L[2] | (L[3] << 16));
            *(uint2*)&s_part[1][slice][4 * u] = make_uint2(R[0] | (R[1] << 16), R[2] | (R[3] << 16));
            *(uint2*)&s_part[2][slice][4 * u] = make_uint2(E[0] | (E[1] << 16), E[2] | (E[3] << 16));
            *(uint2*)&s_part[3][slice][4 * u] = make_uint2(B[0] | (B[1] << 16), B[2] | (B[3] << 16));
            // the diagonal tiles' numbers by quadrant: the wave's maxima
            int q4[4] = {0, 0, 0, 0};
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int k2 = t + q * DF_NT, i2 = k2 / nJ, j2 = k2 - i2 * nJ;
                const int which = i2 < I ? (j2 < J ? 0 : 2) : (j2 < J ? 1 : 3);
                const int enc = pcor[q] ? (int)(wcor[q] & 0xFFFFu) : 0;
#pragma unroll
                for (int w = 0; w < 4; ++w) if (which == w) q4[w] = max(q4[w], enc);
            }
#pragma unroll
            for (int w = 0; w < 4; ++w) q4[w] = df_wave_max(q4[w]);
            if ((t & 63) == 63) *(int4*)&s_Qw[t >> 6][0] = make_int4(q4[0], q4[1], q4[2], q4[3]);
        }
        if (failed) s_fail = 1;
        DFS(T + tile, 1);
        df_barrier();
        DFS(T + tile, 2);
        if (s_fail) {                                   // a summary never arrived: say so, write nothing more
            if (t == 0) { atomicAdd(&state[DST_FAILED], 1u); if (db.hstat[z]) __hip_atomic_store(db.hstat[z], DST_MODE_BROKEN, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
            return;
        }
        // the ring around the tile in the form the ramps take it: per row both halves the same value; per column pair the pair,
        // the diagonal tiles' numbers folded in
        if (t < 2 * DF_T) {
            const int a = t >> 7, r = t & (DF_T - 1);
            int m = 0xFFFF;
#pragma unroll
            for (int s2 = 0; s2 < 16; ++s2) m = min(m, (int)s_part[a][s2][r]);
            if (a == 0) s_L2[r] = df_both(m); else s_R2[r] = df_both(m);
        } else if (t < 4 * DF_T) {
            const int down = t < 3 * DF_T, c = t & (DF_T - 1), xa = X0 + c;
            int e = 0xFFFF;
#pragma unroll
            for (int s2 = 0; s2 < 16; ++s2) e = min(e, (int)s_part[down ? 2 : 3][s2][c]);
            if (e == 0xFFFF) e = DIST_INF;
            int qa = 0, qb = 0;
#pragma unroll
            for (int w = 0; w < 8; ++w) { qa = max(qa, s_Qw[w][down ? 0 : 1]); qb = max(qb, s_Qw[w][down ? 2 : 3]); }
            if (down) {
                if (qa) e = min(e, xa + (Y0 - 1) - (qa - DF_CORNER_BIAS));                     // sources above left:  px + py - max (sx + sy)
                if (qb) e = min(e, (Y0 - 1) - xa - (qb - DF_CORNER_BIAS));                     // above right:          py - px - max (sy - sx)
            } else {
                const int Yb = Y0 + DF_T;                                                      // the row just below the tile
                if (qa) e = min(e, xa - Yb - (qa - DF_CORNER_BIAS));                           // below left:           px - py - max (sx - sy)
                if (qb) e = min(e, -xa - Yb - (qb - DF_CORNER_BIAS));                          // below right:         -px - py - max (-sx - sy)
            }
            ((uint16_t*)(down ? s_E2 : s_B2))[c] = (uint16_t)min(e, 0xFFFF);
        }
        df_barrier();
        DFS(T + tile, 3);
        {
            const unsigned int off_l = (unsigned int)(2 * tx + 1) | ((unsigned int)(2 * tx + 2) << 16);
            const unsigned int off_r = (unsigned int)(DF_T - 2 * tx) | ((unsigned int)(DF_T - 1 - 2 * tx) << 16);
            unsigned int e = df_adds(s_E2[tx], df_both(16 * ty)), bk = df_adds(s_B2[tx], df_both(DF_T - 16 - 16 * ty));
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                e = df_adds(e, one);
                unsigned int v = df_min(d0[k][i], e);
                v = df_min(v, df_adds(s_L2[16 * ty + i], off_l));
                d0[k][i] = df_min(v, df_adds(s_R2[16 * ty + i], off_r));
            }
#pragma unroll
            for (int i = 15; i >= 0; --i) { bk = df_adds(bk, one); d0[k][i] = df_min(d0[k][i], bk); }
        }
        DFS(T + tile, 6);
        uint16_t* __restrict__ l1 = db.l1[z];
        const int xc = X0 + 2 * tx;
        if (xc < W) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int yy = Y0 + 16 * ty + i;
                if (yy < H) *(unsigned int*)(l1 + (size_t)yy * W + xc) = d0[k][i];
            }
        }
        DFS(T + tile, 7);
        df_barrier();                                   // the LDS arrays are free for the next tile
    }
}

// The plan of an incremental transform, by the first wave of a workgroup (every workgroup of k_dist_rows_inc forms the same one
// from the same inputs: the log entries from + 1 .. to and the bound D as the previous transform left it).  plan[0..4] = mode,
// x0, y0, x1, y1.  W is a multiple of 16.
__device__ __forceinline__ void dist_plan_inc(const dist_batch& db, int z, int W, int H, int lane, int* plan)
{
    const unsigned int from = db.from[z], to = db.to[z];
    const int4* log = db.log[z];
    int bx0 = 0x7fffffff, by0 = 0x7fffffff, bx1 = -1, by1 = -1;
    bool lost = log == nullptr || to - from > (unsigned int)(BL_DIRTY_LOG - 32);
    if (!lost)
        for (unsigned int v = from + 1u + (unsigned int)lane; v <= to; v += 64u) {
            const int4 e = log[v % BL_DIRTY_LOG];
            if ((unsigned int)e.z != v) { lost = true; continue; }            // overwritten, or never written: no knowledge
            const int x0 = e.x & 0xffff, y0 = (int)((unsigned int)e.x >> 16), x1 = e.y & 0xffff, y1 = (int)((unsigned int)e.y >> 16);
            if (x1 < x0 || y1 < y0) continue;                                 // that update changed nothing
            bx0 = min(bx0, x0); by0 = min(by0, y0); bx1 = max(bx1, x1); by1 = max(by1, y1);
        }
    for (int off = 32; off > 0; off >>= 1) {
        bx0 = min(bx0, __shfl_xor(bx0, off, 64)); by0 = min(by0, __shfl_xor(by0, off, 64));
        bx1 = max(bx1, __shfl_xor(bx1, off, 64)); by1 = max(by1, __shfl_xor(by1, off, 64));
    }
    lost = __builtin_amdgcn_ballot_w64(lost) != 0ull;
    if (lane != 0) return;
    const unsigned int dub = db.state[z][DST_DUB];
    int mode = DST_MODE_WINDOW, x0 = 0, y0 = 0, x1 = W - 1, y1 = H - 1;
    if (lost || dub >= 0xFFFFu) mode = DST_MODE_FULL;
    else if (bx1 < bx0) mode = DST_MODE_NONE;
    else {
        const int R = (int)dub + 1;
        x0 = max(0, bx0 - R) & ~15; x1 = min(W - 1, bx1 + R) | 15;
        y0 = max(0, by0 - R); y1 = min(H - 1, by1 + R);
        if (x1 > W - 1) x1 = W - 1;
        if (x1 - x0 + 1 > DINC_MAX || y1 - y0 + 1 > DINC_MAX) { mode = DST_MODE_FULL; x0 = 0; y0 = 0; x1 = W - 1; y1 = H - 1; }
    }
    plan[0] = mode; plan[1] = x0; plan[2] = y0; plan[3] = x1; plan[4] = y1;
}

// Row pass of an incremental transform: row[y][x] for the plan's region = min over the sources x' of the region's row of |x - x'|
// and over the two ring cells of the row (just left / right of the region) of their OLD distance + the way to them -- one wave
// per row, a lane per 16 cells, regions wider than 1024 cells in chunks chained through a carry.  Min-plus prefix scans: a value
// v at position p reaches x > p with (v - p) + x and x < p with (v + p) - x.  The first workgroup of a unit publishes the plan.
#define DRI_WAVES 4
__global__ __launch_bounds__(64 * DRI_WAVES) void k_dist_rows_inc(dist_batch db, int W, int H)
{
    const int z = blockIdx.z;
    __shared__ int s_plan[5];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave == 0) dist_plan_inc(db, z, W, H, lane, s_plan);
    __syncthreads();
    const int mode = s_plan[0], x0 = s_plan[1], y0 = s_plan[2], x1 = s_plan[3], y1 = s_plan[4];
    unsigned int* state = db.state[z];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        state[DST_MODE] = (unsigned int)mode;
        state[DST_X0] = (unsigned int)x0; state[DST_Y0] = (unsigned int)y0; state[DST_X1] = (unsigned int)x1; state[DST_Y1] = (unsigned int)y1;
        // (DST_DUB is NOT touched here: workgroups of this launch that start later still form their plan from it.  A whole-grid
        // plan has it reset by the first column launch, which starts when every workgroup of this one has ended.)
        state[DST_STATS + mode] += 1u;
        if (db.hstat[z]) __hip_atomic_store(db.hstat[z], (unsigned int)mode, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (mode == DST_MODE_NONE) return;
    const int8_t* __restrict__ cells = db.cells[z];
    uint16_t* __restrict__ row = db.row[z];
    const uint16_t* __restrict__ old = db.l1[z];
    const int nrows = y1 - y0 + 1, w = x1 - x0 + 1;
    const int nchunk = (w + 1023) >> 10;
    for (int r = blockIdx.x * DRI_WAVES + wave; r < nrows; r += gridDim.x * DRI_WAVES) {
        const int y = y0 + r;
        const int8_t* c = cells + (size_t)y * W;
        uint16_t* out = row + (size_t)y * W;
        int carry_f = DIST_INF, carry_b = DIST_INF;                           // (v - p) of the left ring cell, (v + p) of the right one
        if (mode == DST_MODE_WINDOW) {
            if (x0 > 0) { const int v = old[(size_t)y * W + x0 - 1]; if (v != 0xFFFF) carry_f = v - (x0 - 1); }
            if (x1 < W - 1) { const int v = old[(size_t)y * W + x1 + 1]; if (v != 0xFFFF) carry_b = v + (x1 + 1); }
        }
        // ---- left to right: nearest seed at or left of every cell, kept as a distance in `out` (or in registers: one chunk)
        int4 raw = make_int4(-1, -1, -1, -1);
        uint16_t o[16];
        for (int ch = 0; ch < nchunk; ++ch) {
            const int xb = x0 + (ch << 10) + lane * 16;
            raw = make_int4(-1, -1, -1, -1);                                  // 0xFF bytes: free cells, no source
            if (xb <= x1) raw = *(const int4*)(c + xb);
            const int8_t* b = (const int8_t*)&raw;
            int la = DIST_INF;
#pragma unroll
            for (int i = 0; i < 16; ++i) if (b[i] >= 0) la = -(xb + i);       // is_cell_occupied: logOdds >= 0; the rightmost one wins
            int incl = la;
            for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off, 64); if (lane >= off) incl = min(incl, t); }
            int run = __shfl_up(incl, 1, 64);
            if (lane == 0) run = DIST_INF;
            run = min(run, carry_f);
#pragma unroll
            for (int i = 0; i < 16; ++i) { if (b[i] >= 0) run = -(xb + i); o[i] = (uint16_t)min(run + xb + i, 0xFFFF); }
            carry_f = min(carry_f, __builtin_amdgcn_readlane(incl, 63));
            if (nchunk > 1 && xb <= x1) { *(int4*)(out + xb) = *(const int4*)&o[0]; *(int4*)(out + xb + 8) = *(const int4*)&o[8]; }
        }
        // ---- right to left, merged with the first sweep
        for (int ch = nchunk - 1; ch >= 0; --ch) {
            const int xb = x0 + (ch << 10) + lane * 16;
            if (nchunk > 1) {
                raw = make_int4(-1, -1, -1, -1);
                if (xb <= x1) { raw = *(const int4*)(c + xb); *(int4*)&o[0] = *(const int4*)(out + xb); *(int4*)&o[8] = *(const int4*)(out + xb + 8); }
            }
            const int8_t* b = (const int8_t*)&raw;
            int lb = DIST_INF;
#pragma unroll
            for (int i = 15; i >= 0; --i) if (b[i] >= 0) lb = xb + i;         // the leftmost one wins
            int incl = lb;
            for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_down(incl, off, 64); if (lane + off < 64) incl = min(incl, t); }
            int run = __shfl_down(incl, 1, 64);
            if (lane == 63) run = DIST_INF;
            run = min(run, carry_b);
#pragma unroll
            for (int i = 15; i >= 0; --i) { if (b[i] >= 0) run = xb + i; o[i] = (uint16_t)min((int)o[i], min(run - (xb + i), 0xFFFF)); }
            carry_b = min(carry_b, __builtin_amdgcn_readlane(incl, 0));
            if (xb <= x1) { *(int4*)(out + xb) = *(const int4*)&o[0]; *(int4*)(out + xb + 8) = *(const int4*)&o[8]; }
        }
    }
}

// Column pass over the plan's region, as two launches of one kernel: PHASE 0 leaves, per macro strip (128 rows) and column of the
// region, the distance to the nearest seed inside the strip as seen from its last and from its first row; PHASE 1 chains the
// other strips' summaries into the carries entering a strip from above and below and writes l1.  In a window plan the rows just
// above / below the region seed the chains with their OLD distances.  A workgroup takes the items (macro strip, column group of
// 128 columns) blockIdx.x, blockIdx.x + gridDim.x, ... -- column groups of one strip side by side, so that the workgroups running
// together stream whole rows; the launch is sized for the widest window and loops when the device settled on the whole grid.
// Nothing in either launch waits for another workgroup (a first form did both phases in one launch, a workgroup waiting for the
// summaries of its column group: beside the particle filter's kernels too few of its workgroups became resident together, and
// the waits ran into their cap).
template <int PHASE>
__global__ __launch_bounds__(DC2_TX * DC2_TY) void k_dist_cols_region(dist_batch db, int W, int H)
{
    const int z = blockIdx.z;
    unsigned int* state = db.state[z];
    const int mode = (int)state[DST_MODE];
    if (mode == DST_MODE_NONE) return;
    // a whole-grid plan forms the bound D anew: PHASE 0 (which neither reads nor raises it) clears it between the row launch,
    // whose workgroups all planned from the old value, and PHASE 1, which raises it to the largest distance written
    if (PHASE == 0 && mode == DST_MODE_FULL && blockIdx.x == 0 && threadIdx.x == 0 && threadIdx.y == 0) state[DST_DUB] = 0u;
    const int x0 = (int)state[DST_X0], y0 = (int)state[DST_Y0], x1 = (int)state[DST_X1], y1 = (int)state[DST_Y1];
    const uint16_t* __restrict__ row = db.row[z];
    uint16_t* __restrict__ l1 = db.l1[z];
    int* __restrict__ sum_f = db.sum_f[z];
    int* __restrict__ sum_b = db.sum_b[z];
    __shared__ int s_f[DC2_TY][2 * DC2_TX];
    __shared__ int s_b[DC2_TY][2 * DC2_TX];
    extern __shared__ int s_dcm_sum[];                               // PHASE 1: [2][strips of the grid][128], sized by the launch
    const int strips_cap = (H + DC2_ROWS - 1) / DC2_ROWS;
    int (*s_sf)[2 * DC2_TX] = (int (*)[2 * DC2_TX])s_dcm_sum;
    int (*s_sb)[2 * DC2_TX] = (int (*)[2 * DC2_TX])(s_dcm_sum + (size_t)strips_cap * 2 * DC2_TX);
    const int tx = threadIdx.x, ty = threadIdx.y, t = ty * DC2_TX + tx;
    const int w = x1 - x0 + 1, h = y1 - y0 + 1;
    const int ngroups = (w + 2 * DC2_TX - 1) / (2 * DC2_TX), nstrips = (h + DC2_ROWS - 1) / DC2_ROWS;
    const int nitems = ngroups * nstrips;
    const bool ring = mode == DST_MODE_WINDOW;
    int dmax = 0;
    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
        const int st = item / ngroups, cg = item - st * ngroups;
        const int x = x0 + cg * 2 * DC2_TX + 2 * tx;
        const int Y0 = y0 + st * DC2_ROWS;
        const int ya = Y0 + ty * DC2_SUB;
        const bool live = x <= x1;                                   // (x0 even, x1 odd: a live pair lies inside)
        int g[DC2_SUB][2];
        int a_f[2] = {DIST_INF, DIST_INF}, a_b[2] = {DIST_INF, DIST_INF};
#pragma unroll
        for (int i = 0; i < DC2_SUB; ++i) {
            unsigned int v = 0xFFFFFFFFu;
            if (live && ya + i <= y1) v = *(const unsigned int*)(row + (size_t)(ya + i) * W + x);
            const int a = (int)(v & 0xFFFFu), b = (int)(v >> 16);
            g[i][0] = a == 0xFFFF ? DIST_INF : a;
            g[i][1] = b == 0xFFFF ? DIST_INF : b;
        }
        if (PHASE == 1) {
            // the strips' summaries of this column group (PHASE 0's launch), every load in flight beside the tile's
            for (int i = t; i < nstrips * 2 * DC2_TX; i += DC2_TX * DC2_TY) {
                const int m = i / (2 * DC2_TX), cx = i - m * (2 * DC2_TX);
                const int xg = x0 + cg * 2 * DC2_TX + cx;
                s_sf[m][cx] = xg <= x1 ? sum_f[(size_t)m * W + xg] : DIST_INF;
                s_sb[m][cx] = xg <= x1 ? sum_b[(size_t)m * W + xg] : DIST_INF;
            }
        }
        const int yend = min(y1 + 1, ya + DC2_SUB);                  // one past this thread row's last row
#pragma unroll
        for (int i = 0; i < DC2_SUB; ++i)
            if (ya + i <= y1)
                for (int c = 0; c < 2; ++c) {
                    a_f[c] = min(a_f[c], g[i][c] + (yend - 1 - (ya + i)));
                    a_b[c] = min(a_b[c], g[i][c] + i);
                }
        for (int c = 0; c < 2; ++c) { s_f[ty][2 * tx + c] = a_f[c]; s_b[ty][2 * tx + c] = a_b[c]; }
        __syncthreads();
        if (PHASE == 0) {
            if (ty == 0 && live) {
                for (int c = 0; c < 2; ++c) {
                    int F = DIST_INF, B = DIST_INF;
                    for (int s2 = 0; s2 < DC2_TY; ++s2) {
                        const int sy0 = Y0 + s2 * DC2_SUB, len = max(0, min(y1 + 1, sy0 + DC2_SUB) - sy0);
                        if (len > 0) F = min(s_f[s2][2 * tx + c], F + len);
                    }
                    for (int s2 = DC2_TY - 1; s2 >= 0; --s2) {
                        const int sy0 = Y0 + s2 * DC2_SUB, len = max(0, min(y1 + 1, sy0 + DC2_SUB) - sy0);
                        if (len > 0) B = min(s_b[s2][2 * tx + c], B + len);
                    }
                    sum_f[(size_t)st * W + x + c] = min(F, DIST_INF);
                    sum_b[(size_t)st * W + x + c] = min(B, DIST_INF);
                }
            }
        } else if (live && ya <= y1) {
            int E[2], B[2];
            for (int c = 0; c < 2; ++c) {
                // carries entering the strip: the distance at the row just above it / just below it
                int e = DIST_INF, b = DIST_INF;
                if (ring && y0 > 0) { const int v = l1[(size_t)(y0 - 1) * W + x + c]; if (v != 0xFFFF) e = v; }
                if (ring && y1 < H - 1) { const int v = l1[(size_t)(y1 + 1) * W + x + c]; if (v != 0xFFFF) b = v; }
                for (int m = 0; m < st; ++m) e = min(s_sf[m][2 * tx + c], e + DC2_ROWS);
                for (int m = nstrips - 1; m > st; --m) {
                    const int len = min(h, (m + 1) * DC2_ROWS) - m * DC2_ROWS;
                    b = min(s_sb[m][2 * tx + c], b + len);
                }
                for (int s2 = 0; s2 < ty; ++s2) e = min(s_f[s2][2 * tx + c], e + DC2_SUB);
                for (int s2 = DC2_TY - 1; s2 > ty; --s2) {
                    const int sy0 = Y0 + s2 * DC2_SUB, len = max(0, min(y1 + 1, sy0 + DC2_SUB) - sy0);
                    if (len > 0) b = min(s_b[s2][2 * tx + c], b + len);
                }
                E[c] = min(e, DIST_INF); B[c] = min(b, DIST_INF);
            }
            int f[DC2_SUB][2];
#pragma unroll
            for (int i = 0; i < DC2_SUB; ++i)
                for (int c = 0; c < 2; ++c) { E[c] = min(g[i][c], E[c] + 1); f[i][c] = E[c]; }
#pragma unroll
            for (int i = DC2_SUB - 1; i >= 0; --i) {
                if (ya + i > y1) continue;
                int v[2];
                for (int c = 0; c < 2; ++c) { B[c] = min(g[i][c], B[c] + 1); v[c] = min(min(f[i][c], B[c]), 0xFFFF); }
                dmax = max(dmax, max(v[0], v[1]));
                *(unsigned int*)(l1 + (size_t)(ya + i) * W + x) = (unsigned int)v[0] | ((unsigned int)v[1] << 16);
            }
        }
        __syncthreads();                                             // the LDS arrays are free for the next item
    }
    if (PHASE == 1) {
        // the bound D: largest value written (0xFFFF where no seed reaches) -- one atomic per workgroup, and only one that would
        // raise it (same-address atomics serialise in the L2: one per wave of a whole-grid pass cost 150 us)
        __shared__ int s_dmax;
        if (t == 0) s_dmax = 0;
        __syncthreads();
        for (int off = 32; off > 0; off >>= 1) dmax = max(dmax, __shfl_xor(dmax, off, 64));
        if (tx == 0 && dmax > 0) atomicMax(&s_dmax, dmax);
        __syncthreads();
        if (t == 0 && (unsigned int)s_dmax > dst_load_u(&state[DST_DUB])) atomicMax(&state[DST_DUB], (unsigned int)s_dmax);
    }
}

// The bound D of a transform the whole-grid kernels made (they keep none: an atomic per workgroup would cost them more than this
// pass over l1 costs the first incremental transform that needs it).  The host clears state[DST_DUB] on the stream before it
// (dist_set_distances_batch, the units with !bound_ok).
__global__ __launch_bounds__(256) void k_dist_bound(const uint16_t* __restrict__ l1, size_t n8, unsigned int* state)
{
    __shared__ int s_dmax;
    if (threadIdx.x == 0) s_dmax = 0;
    __syncthreads();
    unsigned int m = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        const int4 v = ((const int4*)l1)[i];
        const unsigned int w[4] = {(unsigned int)v.x, (unsigned int)v.y, (unsigned int)v.z, (unsigned int)v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) m = max(m, max(w[k] & 0xFFFFu, w[k] >> 16));
    }
    for (int off = 32; off > 0; off >>= 1) m = max(m, (unsigned int)__shfl_xor((int)m, off, 64));
    if ((threadIdx.x & 63) == 0 && m > 0) atomicMax(&s_dmax, (int)m);
    __syncthreads();
    if (threadIdx.x == 0 && s_dmax > 0) atomicMax(&state[DST_DUB], (unsigned int)s_dmax);
}

extern "C" int bl_dist_create(bl_ctx* ctx, bl_dist** out)
{
    BL_CHECK_ARG(ctx != nullptr && out != nullptr);
    bl_dist* d = new bl_dist();
    memset((void*)d, 0, sizeof(*d));
    d->ctx = ctx;
    d->frame.mpc = 0.05f; d->frame.cpm = 20.0f;          // ObstacleDistanceGrid() (obstacle_distance_grid.cpp:31-37)
    d->lut_host = new std::vector<float>();
    *out = d;
    return BL_OK;
}

extern "C" void bl_dist_destroy(bl_dist* d)
{
    if (!d) return;
    (void)hipStreamSynchronize(d->ctx->stream);
    if (d->row) (void)hipFree(d->row);
    if (d->l1) (void)hipFree(d->l1);
    if (d->cells) (void)hipFree(d->cells);
    if (d->closed) (void)hipFree(d->closed);
    if (d->sum_f) (void)hipFree(d->sum_f);
    if (d->sum_b) (void)hipFree(d->sum_b);
    if (d->lut) (void)hipFree(d->lut);
    if (d->state) (void)hipFree(d->state);
    if (d->fwords) (void)hipFree(d->fwords);
    if (d->h_status) (void)hipHostFree(d->h_status);
    delete d->lut_host;
    delete d;
}

// The float grid the reference's callers see: f[L1 distance], -1 where no source exists.  k_astar reads the integer distances,
// so a replan never needs it: it is formed when a caller first asks (bl_dist_download, bl_dist_device_ptr, bl_dist_gather) --
// 4 of the 12 bytes per cell the column pass used to move.
__global__ __launch_bounds__(256) void k_dist_floats(const uint16_t* __restrict__ l1, const float* __restrict__ lut, float* __restrict__ out, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int v = l1[i];
        out[i] = v == 0xFFFF ? -1.0f : lut[v];
    }
}

static int dist_floats(bl_dist* d)
{
    if (d->floats_valid) return BL_OK;
    const size_t n = (size_t)d->frame.width * d->frame.height;
    if (!d->cells) BL_HIP(hipMalloc((void**)&d->cells, d->capacity * 4));      // a replan never needs the floats: allocated on first request
    size_t blocks = (n + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(k_dist_floats, dim3((unsigned int)blocks), dim3(256), 0, d->ctx->stream, d->l1, d->lut, d->cells, n);
    BL_HIP(hipGetLastError());
    d->floats_valid = true;
    return BL_OK;
}

// resetGrid (obstacle_distance_grid.cpp:100-118) and the scratch a transform of `map` needs; no launch
static int dist_prepare(bl_dist* d, const bl_grid* map)
{
    bl_ctx* ctx = d->ctx;
    const int W = map->frame.width, H = map->frame.height;
    BL_CHECK_ARG(W + H < 0xFFFF);
    size_t n = (size_t)W * H;
    if (n > d->capacity) {
        BL_HIP(hipStreamSynchronize(ctx->stream));
        if (d->row) BL_HIP(hipFree(d->row));
        if (d->l1) BL_HIP(hipFree(d->l1));
        if (d->cells) BL_HIP(hipFree(d->cells));
        if (d->closed) BL_HIP(hipFree(d->closed));
        d->row = nullptr; d->l1 = nullptr; d->cells = nullptr; d->closed = nullptr;
        BL_HIP(hipMalloc((void**)&d->row, n * 2));
        BL_HIP(hipMalloc((void**)&d->l1, n * 2));
        BL_HIP(hipMalloc((void**)&d->closed, n * 4));
        BL_HIP(hipMemsetAsync(d->closed, 0, n * 4, ctx->stream));           // generation 0: nothing closed
        d->closed_gen = 0;
        d->capacity = n;
        d->valid = false;
        d->src_id = 0;
    }
    if (!d->state) {
        BL_HIP(hipMalloc((void**)&d->state, DST_WORDS * sizeof(unsigned int)));
        BL_HIP(hipMemsetAsync(d->state, 0, DST_WORDS * sizeof(unsigned int), ctx->stream));
        BL_HIP(hipHostMalloc((void**)&d->h_status, 64, hipHostMallocDefault));
        memset(d->h_status, 0, 64);
        BL_HIP(hipHostGetDevicePointer((void**)&d->h_status_dev, d->h_status, 0));
    }
    d->frame = map->frame;
    {
        const size_t need = (size_t)((H + DC2_ROWS - 1) / DC2_ROWS) * W;
        if (need > d->sum_cap) {
            BL_HIP(hipStreamSynchronize(ctx->stream));
            if (d->sum_f) BL_HIP(hipFree(d->sum_f));
            if (d->sum_b) BL_HIP(hipFree(d->sum_b));
            d->sum_f = nullptr; d->sum_b = nullptr;
            BL_HIP(hipMalloc((void**)&d->sum_f, need * 4));
            BL_HIP(hipMalloc((void**)&d->sum_b, need * 4));
            d->sum_cap = need;
        }
    }
    if (d->lut_n < W + H + 1) {
        BL_HIP(hipStreamSynchronize(ctx->stream));
        if (d->lut) BL_HIP(hipFree(d->lut));
        d->lut = nullptr;
        int ln = W + H + 1;
        d->lut_host->resize(ln);
        float f = 0.0f;                                   // f[n] = f[n-1] + 0.1f (obstacle_distance_grid.cpp:174)
        for (int i = 0; i < ln; ++i) { (*d->lut_host)[i] = f; f = f + 0.1f; }
        BL_HIP(hipMalloc((void**)&d->lut, (size_t)ln * 4));
        BL_HIP(hipMemcpy(d->lut, d->lut_host->data(), (size_t)ln * 4, hipMemcpyHostToDevice));
        d->lut_n = ln;
    }
    return BL_OK;
}

// May the transform of `map` start from d's current one?  (same lineage, a later version whose log entries are still there,
// a grid the window kernels handle: rows of whole 16-cell groups, at least DINC_MAX a side, at most 8192 rows)
static bool dist_can_increment(const bl_dist* d, const bl_grid* map)
{
    static const bool off = getenv("BOTLAB_DIST_NO_INCREMENTAL") != nullptr;
    const int W = map->frame.width, H = map->frame.height;
    if (off || !d->valid || d->src_id == 0 || d->src_id != map->id || map->mirror_external || !map->log || !d->state) return false;
    if (d->frame.width != W || d->frame.height != H) return false;
    if ((W & 15) != 0 || W < DINC_MAX || H < DINC_MAX || (H + DC2_ROWS - 1) / DC2_ROWS > DST_MAX_GROUPS || (W + 2 * DC2_TX - 1) / (2 * DC2_TX) > DST_MAX_GROUPS) return false;
    if (map->version < d->src_version || map->version - d->src_version > (uint64_t)(BL_DIRTY_LOG - 64)) return false;
    return d->inc_holdoff == 0;
}

// setDistances of n grids of one size, all on ds[0]'s stream, as one set of launches (blockIdx.z = grid)
static int dist_set_distances_batch(int n, bl_dist* const* ds, const bl_grid* const* maps)
{
    BL_CHECK_ARG(n >= 1 && n <= DIST_MAX_BATCH);
    bl_ctx* ctx = ds[0]->ctx;
    BL_HIP(hipSetDevice(ctx->device));
    const int W = maps[0]->frame.width, H = maps[0]->frame.height;
    dist_batch b;
    memset((void*)&b, 0, sizeof(b));
    bool all_inc = true, all_same = true;
    for (int u = 0; u < n; ++u) {
        BL_CHECK_ARG(ds[u] != nullptr && maps[u] != nullptr && ds[u]->ctx->stream == ctx->stream);
        BL_CHECK_ARG(maps[u]->frame.width == W && maps[u]->frame.height == H);
        bl_dist* d = ds[u];
        // what the device said about the last incremental launch (no waiting: the word is whatever has landed by now)
        if (d->h_status && *d->h_status == (unsigned int)DST_MODE_FULL) { *d->h_status = 0; d->inc_holdoff = 16; }
        if (d->h_status && *d->h_status == DST_MODE_BROKEN) {
            *d->h_status = 0; d->valid = false; d->src_id = 0;
            bl_set_error("setDistances: the one-launch transform gave up (a tile summary never arrived); the distances it left are not valid");
            return BL_ERR_STATE;
        }
        const bool inc = dist_can_increment(d, maps[u]);
        if (d->inc_holdoff > 0) d->inc_holdoff -= 1;
        // the very state of the very map d holds the transform of (any grid size; no log needed)
        const bool same = d->valid && d->src_id != 0 && d->src_id == maps[u]->id && !maps[u]->mirror_external && maps[u]->version == d->src_version &&
                          d->frame.width == W && d->frame.height == H;
        const uint64_t from = d->src_version;
        int rc = dist_prepare(d, maps[u]);
        if (rc) return rc;
        all_inc = all_inc && inc; all_same = all_same && same;
        b.cells[u] = maps[u]->cells; b.row[u] = d->row; b.l1[u] = d->l1;
        b.out[u] = nullptr; b.lut[u] = d->lut;
        b.sum_f[u] = d->sum_f; b.sum_b[u] = d->sum_b; b.state[u] = d->state; b.hstat[u] = d->h_status_dev;
        b.log[u] = maps[u]->log ? maps[u]->log->dev : nullptr;
        b.from[u] = (unsigned int)from; b.to[u] = (unsigned int)maps[u]->version;
    }
    if (all_same) {                                    // the very maps these grids hold the transforms of
        for (int u = 0; u < n; ++u) ds[u]->n_same += 1;
        return BL_OK;
    }
    for (int u = 0; u < n; ++u) ds[u]->floats_valid = false;
    const bool merged = H >= 4 * DC2_ROWS && W >= 4 * DC2_TX && (W & 1) == 0 && (H + DC2_ROWS - 1) / DC2_ROWS <= DST_MAX_GROUPS;
    const int strips = (H + DC2_ROWS - 1) / DC2_ROWS;
    const size_t merged_lds = (size_t)2 * strips * 2 * DC2_TX * sizeof(int);
    {   // the attribute belongs to the device that is current when it is set: once per device this process drives
        static unsigned long long attr_set_devices = 0ull;
        const unsigned long long bit = 1ull << (ctx->device & 63);
        if (!(attr_set_devices & bit)) {
            BL_HIP(hipFuncSetAttribute((const void*)k_dist_cols_region<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * DST_MAX_GROUPS * 2 * DC2_TX * (int)sizeof(int)));
            attr_set_devices |= bit;
        }
    }
    hipEvent_t e0, e1;
    int rc = bl_timer_begin(ctx, BL_K_DIST, &e0, &e1);
    if (rc) return rc;
    hipEvent_t f0, f1;
    // BOTLAB_DIST_REGION_COLS=1: a transform of the whole grid goes through the region kernels too (their plan = the grid): tests
    static const bool region_kernels = getenv("BOTLAB_DIST_REGION_COLS") != nullptr;
    const bool bound_kept = all_inc || (merged && region_kernels);                       // the column pass leaves the bound D behind
    if (all_inc) {
        // The window is B dilated by D + 1, so D must bound l1 as it stands: the whole-grid kernels leave none behind
        // (dist_plan_full: zero) -- the first incremental transform after one of them forms it, one read of l1.
        for (int u = 0; u < n; ++u) {
            bl_dist* d = ds[u];
            if (d->bound_ok) continue;
            const size_t n8 = (size_t)W * H / 8;                                         // (W a multiple of 16: dist_can_increment)
            BL_HIP(hipMemsetAsync(d->state + DST_DUB, 0, sizeof(unsigned int), ctx->stream));
            hipLaunchKernelGGL(k_dist_bound, dim3((unsigned int)std::min<size_t>((n8 + 255) / 256, 2048)), dim3(256), 0, ctx->stream, d->l1, n8, d->state);
            d->bound_ok = true;
        }
        // the window (or, failing that, the whole grid) is settled by the kernels themselves
        rc = bl_timer_begin(ctx, BL_K_DIST_ROWS, &f0, &f1);
        if (rc) return rc;
        hipLaunchKernelGGL(k_dist_rows_inc, dim3(DINC_MAX / DRI_WAVES, 1, n), dim3(64 * DRI_WAVES), 0, ctx->stream, b, W, H);
        rc = bl_timer_end(ctx, BL_K_DIST_ROWS, f0, f1);
        if (rc) return rc;
        const int win_items = (DINC_MAX / (2 * DC2_TX)) * (DINC_MAX / DC2_ROWS);       // the widest window's items
        rc = bl_timer_begin(ctx, BL_K_DIST_COLS_SUMMARY, &f0, &f1);
        if (rc) return rc;
        hipLaunchKernelGGL(k_dist_cols_region<0>, dim3(win_items, 1, n), dim3(DC2_TX, DC2_TY), 0, ctx->stream, b, W, H);
        rc = bl_timer_end(ctx, BL_K_DIST_COLS_SUMMARY, f0, f1);
        if (rc) return rc;
        rc = bl_timer_begin(ctx, BL_K_DIST_COLS_APPLY, &f0, &f1);
        if (rc) return rc;
        hipLaunchKernelGGL(k_dist_cols_region<1>, dim3(win_items, 1, n), dim3(DC2_TX, DC2_TY), merged_lds, ctx->stream, b, W, H);
        rc = bl_timer_end(ctx, BL_K_DIST_COLS_APPLY, f0, f1);
        if (rc) return rc;
    } else {
        // grids of whole 16-cell groups, at least 4 x 4 and at most DF_MAXK x 256 tiles: the whole transform in one launch
        // (BOTLAB_DIST_NO_FUSED=1: the four-launch form)
        const bool no_fused = getenv("BOTLAB_DIST_NO_FUSED") != nullptr;        // (read per call: tests switch forms inside one process)
        const int tiles = ((W + DF_T - 1) / DF_T) * ((H + DF_T - 1) / DF_T);
        const bool fused = !no_fused && !region_kernels && (W & 15) == 0 && W >= 4 * DF_T && H >= 4 * DF_T && W <= DF_MAX_SIDE && H <= DF_MAX_SIDE &&
                           tiles <= DF_MAXK * DF_MAX_WGS;
        if (fused) {
            const int per_wg = (tiles + DF_MAX_WGS - 1) / DF_MAX_WGS;           // tiles a workgroup keeps in registers
            const int per_xcd = (tiles + 7) / 8;                               // (the kernel's tile_of: an XCD's workgroups share a run of tiles)
            const int fused_wgs = 8 * ((per_xcd + per_wg - 1) / per_wg);
            dist_fused_batch fb;
            memset((void*)&fb, 0, sizeof(fb));
            if (const char* e = getenv("BOTLAB_DIST_FUSED_TEST_DELAY")) fb.test_delay = (unsigned int)atoi(e);
            for (int u = 0; u < n; ++u) {
                bl_dist* d = ds[u];
#ifdef DF_STAMPS
                const size_t need = (size_t)tiles * (DF_WORDS + 5) + (size_t)tiles * 32 + (size_t)tiles * 64;      // + [2 T][8] + [T][8][4] 64-bit stamps
#else
                const size_t need = (size_t)tiles * (DF_WORDS + 5);             // per tile: band words, 4 quadrant words, a claim word
#endif
                if (need > d->fwords_cap) {
                    BL_HIP(hipStreamSynchronize(ctx->stream));
                    if (d->fwords) BL_HIP(hipFree(d->fwords));
                    d->fwords = nullptr;
                    BL_HIP(hipMalloc((void**)&d->fwords, need * sizeof(unsigned int)));
                    d->fwords_cap = need; d->f_w = 0;
                }
                if (d->f_w != W || d->f_h != H) {        // words laid out for another grid could carry this launch's tag: tag 0 everywhere
                    BL_HIP(hipMemsetAsync(d->fwords, 0, d->fwords_cap * sizeof(unsigned int), ctx->stream));
                    d->f_w = W; d->f_h = H;
                }
                d->f_tag = d->f_tag % 63u + 1u;
                fb.words[u] = d->fwords; fb.tag[u] = d->f_tag;
            }
            rc = bl_timer_begin(ctx, BL_K_DIST_FUSED, &f0, &f1);
            if (rc) return rc;
            const dim3 fgrid(fused_wgs, 1, n);
            if (per_wg == 1) hipLaunchKernelGGL(k_dist_fused<1>, fgrid, dim3(DF_NT), 0, ctx->stream, b, fb, W, H);
            else if (per_wg == 2) hipLaunchKernelGGL(k_dist_fused<2>, fgrid, dim3(DF_NT), 0, ctx->stream, b, fb, W, H);
            else hipLaunchKernelGGL(k_dist_fused<DF_MAXK>, fgrid, dim3(DF_NT), 0, ctx->stream, b, fb, W, H);
            rc = bl_timer_end(ctx, BL_K_DIST_FUSED, f0, f1);
            if (rc) return rc;
        } else {
        if (!merged) for (int u = 0; u < n; ++u) b.state[u] = nullptr;      // (the small-grid column pass keeps no plan or bound)
        rc = bl_timer_begin(ctx, BL_K_DIST_ROWS, &f0, &f1);
        if (rc) return rc;
        if (W >= 1024 && (W & 15) == 0) hipLaunchKernelGGL(k_dist_rows_wide, dim3(H, 1, n), dim3(256), 0, ctx->stream, b, W);
        else if (W <= 256 && (W & 3) == 0) hipLaunchKernelGGL(k_dist_rows_narrow, dim3((H + 3) / 4, 1, n), dim3(256), 0, ctx->stream, b, W, H);
        else hipLaunchKernelGGL(k_dist_rows, dim3(H, 1, n), dim3(256), 0, ctx->stream, b, W);
        rc = bl_timer_end(ctx, BL_K_DIST_ROWS, f0, f1);
        if (rc) return rc;
        if (merged && region_kernels) {
            const int groups = (W + 2 * DC2_TX - 1) / (2 * DC2_TX);
            rc = bl_timer_begin(ctx, BL_K_DIST_COLS_SUMMARY, &f0, &f1);
            if (rc) return rc;
            hipLaunchKernelGGL(k_dist_cols_region<0>, dim3(groups * strips, 1, n), dim3(DC2_TX, DC2_TY), 0, ctx->stream, b, W, H);
            rc = bl_timer_end(ctx, BL_K_DIST_COLS_SUMMARY, f0, f1);
            if (rc) return rc;
            rc = bl_timer_begin(ctx, BL_K_DIST_COLS_APPLY, &f0, &f1);
            if (rc) return rc;
            hipLaunchKernelGGL(k_dist_cols_region<1>, dim3(groups * strips, 1, n), dim3(DC2_TX, DC2_TY), merged_lds, ctx->stream, b, W, H);
            rc = bl_timer_end(ctx, BL_K_DIST_COLS_APPLY, f0, f1);
            if (rc) return rc;
        } else if (H >= 4 * DC2_ROWS && W >= 4 * DC2_TX && (W & 1) == 0) {
            const dim3 grid2((W + 2 * DC2_TX - 1) / (2 * DC2_TX), strips, n);
            rc = bl_timer_begin(ctx, BL_K_DIST_COLS_SUMMARY, &f0, &f1);
            if (rc) return rc;
            hipLaunchKernelGGL(k_dist_cols_summary, grid2, dim3(DC2_TX, DC2_TY), 0, ctx->stream, b, W, H);
            const bool carried = (H + DC2_ROWS - 1) / DC2_ROWS <= DC2_STAGE;
            // (the small carry kernel is timed with the summaries it turns into carries)
            if (carried) hipLaunchKernelGGL(k_dist_cols_carry, dim3((W + 255) / 256, 1, n), dim3(256), 0, ctx->stream, b, W, H);
            rc = bl_timer_end(ctx, BL_K_DIST_COLS_SUMMARY, f0, f1);
            if (rc) return rc;
            rc = bl_timer_begin(ctx, BL_K_DIST_COLS_APPLY, &f0, &f1);
            if (rc) return rc;
            if (carried) hipLaunchKernelGGL(k_dist_cols_apply<true>, grid2, dim3(DC2_TX, DC2_TY), 0, ctx->stream, b, W, H);
            else hipLaunchKernelGGL(k_dist_cols_apply<false>, grid2, dim3(DC2_TX, DC2_TY), 0, ctx->stream, b, W, H);
            rc = bl_timer_end(ctx, BL_K_DIST_COLS_APPLY, f0, f1);
            if (rc) return rc;
        } else {
            rc = bl_timer_begin(ctx, BL_K_DIST_COLS_APPLY, &f0, &f1);
            if (rc) return rc;
            if (H <= DCOL_TY * DCOLS_ROWS) hipLaunchKernelGGL(k_dist_cols_small, dim3((W + DCOL_TX - 1) / DCOL_TX, 1, n), dim3(DCOL_TX, DCOL_TY), 0, ctx->stream, b, W, H);
            else hipLaunchKernelGGL(k_dist_cols, dim3((W + DCOL_TX - 1) / DCOL_TX, 1, n), dim3(DCOL_TX, DCOL_TY), 0, ctx->stream, b, W, H);
            rc = bl_timer_end(ctx, BL_K_DIST_COLS_APPLY, f0, f1);
            if (rc) return rc;
        }
        }
    }
    BL_HIP(hipGetLastError());
    rc = bl_timer_end(ctx, BL_K_DIST, e0, e1);
    if (rc) return rc;
    for (int u = 0; u < n; ++u) {
        bl_dist* d = ds[u];
        d->valid = true;
        // a caller that holds the float grid's device pointer gets it formed with every transform (it may not ask again)
        if (d->floats_handed_out) { const int rcf = dist_floats(d); if (rcf) return rcf; }
        if (all_inc) d->n_inc += 1; else d->n_full += 1;
        // l1 is now the transform of this version of this lineage (the small-grid and two-pass column kernels keep no bound D: the
        // next transform is a full one as well)
        d->src_id = bl_grid_lineage_id(maps[u]); d->src_version = maps[u]->version;
        d->bound_ok = bound_kept;
    }
    return BL_OK;
}

extern "C" int bl_dist_set_distances(bl_dist* d, const bl_grid* map)
{
    BL_CHECK_ARG(d != nullptr && map != nullptr);
    return dist_set_distances_batch(1, &d, &map);
}

// the next setDistances transforms the whole map, whatever `d` holds now (timing the whole-grid kernels; tests)
extern "C" int bl_dist_forget(bl_dist* d)
{
    BL_CHECK_ARG(d != nullptr);
    d->src_id = 0; d->bound_ok = false;
    return BL_OK;
}

// transforms by kind: {incremental launches, whole-grid launches, calls that found the map unchanged, and of the incremental
// launches those the device ended as: nothing to do, a window, the whole grid after all}  (synchronises; diagnostic)
extern "C" int bl_dist_debug_stats(bl_dist* d, int64_t* out6)
{
    BL_CHECK_ARG(d != nullptr && out6 != nullptr);
    out6[0] = d->n_inc; out6[1] = d->n_full; out6[2] = d->n_same; out6[3] = out6[4] = out6[5] = 0;
    if (d->state) {
        unsigned int st[3];
        BL_HIP(hipMemcpyAsync(st, d->state + DST_STATS, sizeof(st), hipMemcpyDeviceToHost, d->ctx->stream));
        BL_HIP(hipStreamSynchronize(d->ctx->stream));
        out6[3] = st[0]; out6[4] = st[1]; out6[5] = st[2];
    }
    return BL_OK;
}

extern "C" int bl_dist_debug_fused(bl_dist* d, int64_t* out2)
{
    BL_CHECK_ARG(d != nullptr && out2 != nullptr);
    out2[0] = out2[1] = 0;
    if (d->state) {
        unsigned int st[2];
        BL_HIP(hipMemcpyAsync(st, d->state + DST_FAILED, sizeof(st), hipMemcpyDeviceToHost, d->ctx->stream));
        BL_HIP(hipStreamSynchronize(d->ctx->stream));
        out2[0] = st[0]; out2[1] = st[1];
    }
    return BL_OK;
}

// the bound D the next incremental transform would build its window from, and whether the host counts it as formed
// (synchronises; diagnostic: tests assert D >= every finite distance of the grid)
extern "C" int bl_dist_debug_bound(bl_dist* d, int* formed, unsigned int* bound)
{
    BL_CHECK_ARG(d != nullptr && formed != nullptr && bound != nullptr);
    *formed = d->bound_ok ? 1 : 0; *bound = 0;
    if (d->state) {
        BL_HIP(hipMemcpyAsync(bound, d->state + DST_DUB, sizeof(unsigned int), hipMemcpyDeviceToHost, d->ctx->stream));
        BL_HIP(hipStreamSynchronize(d->ctx->stream));
    }
    return BL_OK;
}

#ifdef DF_STAMPS
extern "C" int bl_dist_debug_fused_stamps(bl_dist* d, unsigned long long* out, int n)
{
    const int tiles = ((d->f_w + DF_T - 1) / DF_T) * ((d->f_h + DF_T - 1) / DF_T);
    if (n > tiles * 48) n = tiles * 48;
    BL_HIP(hipMemcpyAsync(out, d->fwords + (size_t)tiles * (DF_WORDS + 5), (size_t)n * 8, hipMemcpyDeviceToHost, d->ctx->stream));
    BL_HIP(hipStreamSynchronize(d->ctx->stream));
    return n;
}
#endif

// A whole-grid launch of k_dist_fused that gave up says so in the pinned status word (DST_MODE_BROKEN).  Every entry that has just
// synchronised the grid's stream and is about to hand out something computed FROM the distances looks at it: the grid is marked
// invalid and the call returns BL_ERR_STATE instead of results from distances that were never finished.
static int dist_check_broken(bl_dist* d)
{
    if (d && d->h_status && *d->h_status == DST_MODE_BROKEN) {
        *d->h_status = 0; d->valid = false; d->src_id = 0;
        bl_set_error("setDistances: the one-launch transform gave up (a tile summary never arrived); the distances it left are not valid");
        return BL_ERR_STATE;
    }
    return BL_OK;
}

extern "C" int bl_dist_download(bl_dist* d, float* cells)
{
    BL_CHECK_ARG(d != nullptr && cells != nullptr && d->valid);
    { int rc = dist_floats(d); if (rc) return rc; }
    BL_HIP(hipMemcpyAsync(cells, d->cells, (size_t)d->frame.width * d->frame.height * 4, hipMemcpyDeviceToHost, d->ctx->stream));
    BL_HIP(hipStreamSynchronize(d->ctx->stream));
    return dist_check_broken(d);
}

extern "C" int bl_dist_shape(const bl_dist* d, int* width, int* height)
{
    BL_CHECK_ARG(d != nullptr);
    if (width) *width = d->frame.width;
    if (height) *height = d->frame.height;
    return BL_OK;
}

extern "C" int bl_dist_frame(const bl_dist* d, float* mpc, float* cpm, float* ox, float* oy)
{
    BL_CHECK_ARG(d != nullptr);
    if (mpc) *mpc = d->frame.mpc;
    if (cpm) *cpm = d->frame.cpm;
    if (ox) *ox = d->frame.ox;
    if (oy) *oy = d->frame.oy;
    return BL_OK;
}

extern "C" void* bl_dist_device_ptr(bl_dist* d)
{
    if (!d) return nullptr;
    d->floats_handed_out = true;                       // from now on every setDistances forms the floats (on d's stream, behind the transform)
    if (!d->valid) return (void*)d->cells;
    if (dist_floats(d) != BL_OK) return nullptr;
    return (void*)d->cells;
}

// =============================================================================================== A*
#define ASTAR_INVALID_COST INT32_MIN
#define ASTAR_ST_FOUND 0
#define ASTAR_ST_NOPATH 1        // early exit or open list exhausted: 1-pose path
#define ASTAR_ST_CAPACITY 2
#define ASTAR_ST_LIMIT 3
#define ASTAR_ST_BROKEN 4          // the two wavefronts of a search lost each other (bl_astar2_duo.h: never observed; ends the search instead of hanging)

struct astar_result { int status; int path_len; long long pops; long long pushes; bl_pose_xyt_t start; long long stamps[6];
                      long long path_off; };           // batch form: where the path went in the shared pool

// Diagnostic build only (-DBL_ASTAR_STAMPS): s_memtime shares of the search loop, written to the result record's
// stamps[] (never to an output the search computes from).
#ifdef BL_ASTAR_STAMPS
#define STAMP(var) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define STAMP(var) do { } while (0)
#endif

// Open-list entry: x = fCost, y = (cy << 17) | (cx << 2) | k where the cell (cx, cy) was generated from its parent by
// move k (cell = parent + delta[k]).  gCost is not stored: fCost = gCost + hCost(cell) + oCost(cell) and the last two
// are functions of the cell alone, so gCost is recovered when the entry is popped.
// Two LDS footprints: BIG keeps heap levels 0..13 (16383 entries, 128 KB) + an 8000-entry cost table in LDS -- the
// whole CU's LDS, for a search that runs alone; SMALL keeps levels 0..11 (4095 entries, 32 KB) + 2000 cost entries
// (40 KB in all) so the search can start on a CU that a co-running kernel (the particle filter: 40 KB per workgroup)
// already occupies instead of waiting for one to drain.  Deeper levels live in HBM/L2.
#define AH_LDS 16383
#define AH_COST_LDS 8000
#define AH_LDS_BYTES ((AH_LDS + 1) * 8 + AH_COST_LDS * 4)
#define AH_LDS_SMALL 4095
#define AH_COST_LDS_SMALL 2000
#define AH_LDS_SMALL_BYTES ((AH_LDS_SMALL + 1) * 8 + AH_COST_LDS_SMALL * 4)
#define AH_MAX_DIM 32767
#define AH_MAX_CAP (1 << 25)

#define ASTAR_SLOTS 4

struct bl_astar_state {
    int2* heap; int64_t heap_cap;
    // closed cells live with the distance grid (bl_dist::closed): -1 not closed; 0..3 move that produced the FIRST
    // closed entry; 4 start.  d_out = [astar_result, padded to ASTAR_HDR bytes][path cells]: one D2H brings back the
    // result record and the head of the path.
    char* d_out; size_t path_cap;
    // every pending search also leaves its path in a buffer of its result slot (8 * (W + H) cells, at least the head): a path
    // longer than the pinned head is fetched from there under the slot's event, whatever was launched behind it
    int32_t* d_slot_path[ASTAR_SLOTS]; size_t slot_path_cap; int64_t slot_search[ASTAR_SLOTS];
    int path_head;                         // cells of the path that travel with the result record (BOTLAB_ASTAR_PATH_HEAD: tests)
    int32_t* cost_lut; int cost_lut_cap;
    struct astar_unit* h_units;            // pinned [ASTAR_SLOTS][ASTAR_MAX_UNITS]: per-workgroup arguments of the unit form
    char* h_out[ASTAR_SLOTS];              // pinned result ring ([result][path head]): searches may be enqueued ahead of fetching
    char* h_out_dev[ASTAR_SLOTS];          // the same slots as the device sees them: k_astar writes its result there itself
    hipEvent_t done[ASTAR_SLOTS];
    bl_frame slot_frame[ASTAR_SLOTS];
    bl_dist* slot_dist[ASTAR_SLOTS];       // the distance grid each pending search runs on (its status word is looked at when the result is fetched)
    int64_t launched, fetched;
    int32_t* h_cost;                   // pinned staging for the cost table
    bool pending;
    bl_pose_xyt_t start;
    bl_frame frame;
    // cost table cache key
    bool lut_valid; bl_search_params_t lut_params; int lut_n; const void* lut_owner;
    int lut_eff;                       // entries [lut_eff - 1, lut_n) of the cost table are all equal: the kernel clamps the index to lut_eff - 1
    int last_kernel = 0;               // 2: k_astar2, 1: k_astar took the last launch (bl_astar_debug_last_kernel)
    int lut_min;                       // smallest obstacle cost of a valid cell: fCost >= lut_min (k_astar2 keeps fCost in 16 bits)
    // batch form
    int b_cap; size_t b_cells; int64_t b_heap_each; size_t b_path_each;
    int2* b_heap; int32_t* b_closed; int32_t* b_path; int32_t* b_pool; char* b_results; int2* b_goals;
    unsigned long long* b_cursor;
    char* hb_results; int2* hb_goals; int32_t* hb_pool; size_t hb_pool_cap;
    // distance gather (bl_dist_gather)
    int g_cap; int2* g_cells; float* g_vals; int2* hg_cells; float* hg_vals;
};

#define ASTAR_PATH_HEAD 4096
#define ASTAR_HDR 256
static_assert(sizeof(astar_result) <= ASTAR_HDR, "result record must fit the header of the output buffer");

#define ASTAR_MAX_UNITS 64
struct astar_unit {
    const uint16_t* l1; const int32_t* cost_lut; int2* heap; int32_t* closed; int32_t* path; astar_result* result;
    const bl_pose_xyt_t* start_dev; bl_pose_xyt_t start_host; int sx, sy, gx, gy;
    char* host_out; int32_t* slot_path;
    int cost_n;
    unsigned int closed_gen;
};

struct astar_args {
    const uint16_t* l1; int W, H;
    const int32_t* cost_lut; int cost_n;   // per L1 distance min(n, cost_n - 1): obstacle cost, or ASTAR_INVALID_COST if the cell is not valid
    int2* heap; int heap_cap;
    int32_t* closed; unsigned int closed_gen;     // closed cells: entries of generation closed_gen, (closed_gen << 3) | move
    int32_t* path; long long path_cap;
    int32_t* slot_path; long long slot_path_cap; int path_head;      // per-result-slot copy of the path; cells that go to host_out
    astar_result* result;
    // pinned host slot ([result][path head]) the search leaves its outcome in directly (no copy command behind the kernel:
    // an asynchronous device-to-host copy blocked the enqueueing thread for ~7 ms once every few hundred calls), or null
    char* host_out;
    int sx, sy, gx, gy;
    long long max_pops;
    const bl_pose_xyt_t* start_dev;    // when non-null the start cell is derived on the device from this pose
    bl_pose_xyt_t start_host;
    bl_frame frame;
    // batch form (bl_astar_search_batch): workgroup b runs search b -- same start, own goal cell, own heap / closed grid /
    // path scratch / result record at the strides below; found paths are compacted into one pool for a single D2H.
    const int2* batch_goals;           // null: single search
    long long heap_stride, closed_stride, path_stride;
    int32_t* pool; unsigned long long* pool_cursor;
    // unit form (the replanner's batches): workgroup b runs an unrelated search -- own distance grid, cost table, start
    // pose, goal and scratch -- on grids of one size
    // (an array in pinned host memory, read once per workgroup: indexing a by-value kernel argument with blockIdx moves the
    // whole argument block out of scalar registers and slowed every search by a third)
    const astar_unit* units;
};

// The LDS part of the heap is addressed through an address_space(3) pointer: a two-way select between an LDS and a
// global pointer makes hipcc emit FLAT accesses, which reach LDS through the aperture check at several times the
// ds_read latency.  The slow (mixed) accessors are only used for rounds that touch entries beyond AH_LDS.
extern __shared__ int2 s_heap[];                    // [AH_LDS entries][1 dummy][cost table]
typedef __attribute__((address_space(3))) long long lds_ll_t;
typedef __attribute__((address_space(3))) int lds_int_t;

__device__ __forceinline__ int2 lds_entry(int i)
{
    long long raw = ((volatile lds_ll_t*)s_heap)[i];
    return make_int2((int)(raw & 0xffffffffll), (int)(raw >> 32));
}
__device__ __forceinline__ void lds_entry_store(int i, int2 v)
{
    ((volatile lds_ll_t*)s_heap)[i] = ((long long)(unsigned int)v.x) | ((long long)v.y << 32);
}
// The part of the heap in HBM through a pointer that SAYS global memory: what comes out of the argument block is a generic pointer
// to the compiler, a generic access is a FLAT instruction, and a flat load counts on the LDS counter too -- every wait for an LDS
// read then also waited for the heap's loads in flight, and the other way round.
typedef __attribute__((address_space(1))) int2 gint2_t;
__device__ __forceinline__ int2 gheap_load(const int2* g_heap, int i) { const gint2_t* p = (const gint2_t*)(g_heap + i); return make_int2(p->x, p->y); }
__device__ __forceinline__ void gheap_store(int2* g_heap, int i, int2 v) { gint2_t* p = (gint2_t*)(g_heap + i); p->x = v.x; p->y = v.y; }

template <int LDSN>
__device__ __forceinline__ int2 heap_read(const int2* g_heap, int i)
{
    int2 v = lds_entry(i < LDSN ? i : LDSN);
    if (i >= LDSN) v = gheap_load(g_heap, i);
    return v;
}
template <int LDSN>
__device__ __forceinline__ void heap_write(int2* g_heap, int i, int2 v)
{
    lds_entry_store(i < LDSN ? i : LDSN, v);
    if (i >= LDSN) gheap_store(g_heap, i, v);
}

// std::__push_heap(first, hole, 0, value, greater-by-fCost) (stl_heap.h:128-146) by one wavefront: lane a holds the
// (a+1)-th ancestor of the hole; the value rises past the leading run of ancestors with a larger fCost, each of which
// drops one level.  IN_LDS: the hole (and therefore its whole ancestor chain) is below AH_LDS -- that instantiation
// contains no vector-memory instruction, so the loads the caller left in flight are not waited for here.
template <bool IN_LDS, int LDSN>
__device__ __forceinline__ void heap_sift_up(int2* g_heap, int hole, int2 value, int lane)
{
    const unsigned int hp = (unsigned int)hole + 1u;
    const int D = 31 - __clz(hp);                       // ancestors of the hole
    const bool v = lane < D;
    const int anc = (int)(hp >> (lane + 1)) - 1;
    const int below = (int)(hp >> lane) - 1;            // where this ancestor lands if it drops one level
    int2 e = make_int2(0, 0);
    if (IN_LDS) e = lds_entry(v ? anc : 0);               // (unconditional: a clamped index costs less than a branch around the read)
    else if (v) e = heap_read<LDSN>(g_heap, anc);
    // (the lanes' compare as a mask, ANDed with the mask of the lanes that hold an ancestor: ballot(a && b) makes hipcc put the
    // flag into a register and compare it again -- a few more links in a chain that a lone wave walks at ~15 cycles a link)
    const unsigned long long m = __builtin_amdgcn_sicmp(e.x, value.x, 38 /* signed > */) & (D >= 64 ? ~0ull : ((1ull << D) - 1ull));
    const int t = __ffsll((long long)~m) - 1;           // length of the leading run
    if (IN_LDS) {
        // (no branches around the stores: lanes with nothing to write aim at the dummy entry behind the LDS levels -- a lone wave
        // pays more for a skipped branch region than for a store)
        lds_entry_store(lane < t ? below : LDSN, e);
        lds_entry_store(lane == 0 ? (int)(hp >> t) - 1 : LDSN, value);
    } else {
        if (lane < t) heap_write<LDSN>(g_heap, below, e);
        if (lane == 0) heap_write<LDSN>(g_heap, (int)(hp >> t) - 1, value);
        __threadfence_block();                          // drain the wave's HBM stores before dependent loads
    }
    __builtin_amdgcn_wave_barrier();
}

// std::__adjust_heap(first, 0, len, value) (stl_heap.h:214-250): the hole walks to a leaf always taking the child for
// which comp(right, left) is false, then value is pushed up from there.
// A round handles the 6-level subtree under the hole with 63 lanes: lane l (level k, index j in its level) loads its
// own entry and, if internal, the fCost of its two children; bit l of M = ballot("prefers right child").  Lane l is on
// the walk iff it is a valid node and every ancestor inside the subtree prefers the branch toward it -- a compare of M
// against two per-lane constant masks (amask: its ancestors' lanes, areq: the branch bit required at each).  Each
// on-path lane moves its entry to its parent; the deepest on-path lane is the new hole.
// IN_LDS: len <= AH_LDS, the whole heap is in LDS (no vector-memory instruction in that instantiation).
template <bool IN_LDS, int LDSN>
__device__ __forceinline__ void heap_adjust(int2* g_heap, int len, int2 value, int lane, int lk, int ljm1,
                                            unsigned long long amask, unsigned long long areq)
{
    int hole = 0;
    while (true) {
        const int hp = hole + 1;
        const int node = (hp << lk) + ljm1;
        const bool valid = lane < 63 && node < len;
        const int cl = 2 * node + 1;
        const bool two = lane < 31 && cl + 1 < len;     // both children exist
        int2 e = make_int2(0, 0);
        int fl = 0, fr = 0;
        if (IN_LDS) {
            // unconditional reads at clamped indices (entry 0 always exists here): no branches around the three loads
            e = lds_entry(valid ? node : 0);
            fl = lds_entry(two && valid ? cl : 0).x;
            fr = lds_entry(two && valid ? cl + 1 : 0).x;
        } else if (valid) {
            e = heap_read<LDSN>(g_heap, node);
            if (two) { fl = heap_read<LDSN>(g_heap, cl).x; fr = heap_read<LDSN>(g_heap, cl + 1).x; }
        }
        // right child preferred unless comp(right, left), i.e. right.fCost > left.fCost; a lone left child -> left
        const unsigned long long vmask = __builtin_amdgcn_ballot_w64(valid), tmask = __builtin_amdgcn_ballot_w64(two);
        const unsigned long long M = __builtin_amdgcn_sicmp(fr, fl, 41 /* signed <= */) & vmask & tmask;
        const unsigned long long P = __builtin_amdgcn_uicmpl((M ^ areq) & amask, 0ull, 32 /* == */) & vmask;
        const bool on_path = ((P >> lane) & 1ull) != 0ull;
        const int cur = 63 - __clzll((long long)P);     // deepest on-path lane
        if (IN_LDS) lds_entry_store(on_path && lane > 0 ? (node - 1) >> 1 : LDSN, e);      // (unconditional: see heap_sift_up)
        else if (on_path && lane > 0) heap_write<LDSN>(g_heap, (node - 1) >> 1, e);
        hole = __builtin_amdgcn_readlane(node, cur);
        if (!(cur >= 31 && 2 * hole + 1 < len)) break;
    }
    if (!IN_LDS) __threadfence_block();
    __builtin_amdgcn_wave_barrier();
    heap_sift_up<IN_LDS, LDSN>(g_heap, hole, value, lane);
}

// std::__adjust_heap as ONE pass for heaps with levels outside LDS.  heap_adjust above moves every entry on the walk up one
// level and then lets heap_sift_up pull the lowest t of them down again -- two rounds of stores, a fence between them (the
// sift-up re-reads what the walk has just written, from memory that is not coherent within the wave without it) and a second
// read of the same entries: four global round trips per pop once the walk leaves the LDS levels.  Here the walk stores nothing:
// every round notes the entries it passes in a small LDS array indexed by heap level (s_path[j] = the old entry of the walk's
// node at level j); the climb is decided on that array (the ancestors' "new" entries are exactly s_path[k], s_path[k-1], ...),
// and only the net result is written: the walk's nodes above the landing level take their child's old entry, the landing node
// takes `value`, the nodes below keep theirs.  Same final heap, entry for entry; two global round trips (the last round's loads,
// the closing stores' fence).
template <int LDSN>
__device__ __forceinline__ void heap_adjust_fused(int2* g_heap, int len, int2 value, int lane, int lk, int ljm1,
                                                  unsigned long long amask, unsigned long long areq, volatile lds_ll_t* s_path)
{
    int hole = 0;
    while (true) {
        const int hp = hole + 1;
        const int node = (hp << lk) + ljm1;
        const bool valid = lane < 63 && node < len;
        const int cl = 2 * node + 1;
        const bool two = lane < 31 && cl + 1 < len;     // both children exist
        // the three reads of a lane, issued together: LDS at clamped indices, and for nodes past the LDS levels the global
        // loads back to back (one heap_read after the other waits for each load before it issues the next)
        const bool vt = valid && two;
        int2 e = lds_entry(valid && node < LDSN ? node : LDSN);
        int fl = lds_entry(vt && cl < LDSN ? cl : LDSN).x, fr = lds_entry(vt && cl + 1 < LDSN ? cl + 1 : LDSN).x;
        int2 eg = make_int2(0, 0), flg = make_int2(0, 0), frg = make_int2(0, 0);
        if (valid && node >= LDSN) eg = gheap_load(g_heap, node);
        if (vt && cl >= LDSN) flg = gheap_load(g_heap, cl);
        if (vt && cl + 1 >= LDSN) frg = gheap_load(g_heap, cl + 1);
        if (valid && node >= LDSN) e = eg;
        if (vt && cl >= LDSN) fl = flg.x;
        if (vt && cl + 1 >= LDSN) fr = frg.x;
        const unsigned long long M = __ballot(valid && two && !(fr > fl));
        const bool on_path = valid && (((M ^ areq) & amask) == 0ull);
        const unsigned long long P = __ballot(on_path);
        const int cur = 63 - __clzll((long long)P);     // deepest on-path lane
        if (on_path) s_path[31 - __clz(node + 1)] = ((long long)(unsigned int)e.x) | ((long long)e.y << 32);
        hole = __builtin_amdgcn_readlane(node, cur);
        if (!(cur >= 31 && 2 * hole + 1 < len)) break;
    }
    __builtin_amdgcn_wave_barrier();
    // the climb (std::__push_heap from the leaf): lane a looks at the ancestor a + 1 levels above the hole, whose entry after
    // the walk would be the old entry of the walk's node one level below it
    const unsigned int hpf = (unsigned int)hole + 1u;
    const int k = 31 - __clz(hpf);                      // level of the leaf = number of ancestors
    const bool v = lane < k;
    long long raw = 0;
    if (v) raw = s_path[k - lane];
    const int ex = (int)(raw & 0xffffffffll);
    const unsigned long long m = __ballot(v && (ex > value.x));
    const int t = __ffsll((long long)~m) - 1;           // ancestors that end up where they were
    // net stores: the walk's node at level j < k - t takes s_path[j + 1] (lane a holds s_path[k - a], whose new home is the
    // node a + 1 levels above the leaf); the node at level k - t takes `value`; the t nodes below keep their entries
    if (v && lane >= t) heap_write<LDSN>(g_heap, (int)(hpf >> (lane + 1)) - 1, make_int2(ex, (int)(raw >> 32)));
    if (lane == 0) heap_write<LDSN>(g_heap, (int)(hpf >> t) - 1, value);
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();
}

// One wavefront runs the reference's search loop (astar.cpp:75-135) with libstdc++'s heap operations executed
// cooperatively; lanes 0..3 evaluate the four neighbours of the popped node, lane 4 re-derives its gCost.
// Closed cells: closed[] holds (generation << 3) | move; an entry of this search's generation is written once (first closing wins:
// lane 4 loads the popped cell's own entry with the neighbours') and read with
// L1-bypassing loads, so no lane ever waits on the closing store of the popped cell.
template <int LDSN, int COSTN>
__global__ __launch_bounds__(64) void k_astar(astar_args a)
{
    if (a.units) {
        const astar_unit u = a.units[blockIdx.x];
        a.l1 = u.l1; a.cost_lut = u.cost_lut; a.heap = u.heap; a.closed = u.closed; a.path = u.path; a.result = u.result;
        a.start_dev = u.start_dev; a.start_host = u.start_host; a.sx = u.sx; a.sy = u.sy; a.gx = u.gx; a.gy = u.gy;
        a.host_out = u.host_out; a.slot_path = u.slot_path; a.cost_n = u.cost_n; a.closed_gen = u.closed_gen;
    }
    if (a.batch_goals) {
        const long long b = blockIdx.x;
        const int2 g = a.batch_goals[b];
        a.gx = g.x; a.gy = g.y;
        a.heap += b * a.heap_stride; a.closed += b * a.closed_stride; a.path += b * a.path_stride;
        a.result = (astar_result*)((char*)a.result + b * ASTAR_HDR);
        a.host_out = nullptr; a.slot_path = nullptr; a.slot_path_cap = 0;
    }
    int2* g_heap = a.heap;
    __shared__ long long s_path[40];                    // heap_adjust_fused: the entries a walk passes, by heap level
    const int lane = threadIdx.x;
    __builtin_amdgcn_s_setprio(3);          // a lone latency-bound wave: win issue arbitration against co-resident kernels
    astar_result res; res.status = ASTAR_ST_NOPATH; res.path_len = 0; res.pops = 0; res.pushes = 0;
    for (int q = 0; q < 6; ++q) res.stamps[q] = 0;
    res.path_off = 0;
    res.start = a.start_host;
    if (a.start_dev) {                                   // global_position_to_grid_cell of the device-resident pose
        res.start = *a.start_dev;
        bl_global_to_cell((double)res.start.x, (double)res.start.y, a.frame, &a.sx, &a.sy);
    }
    const bool cost_in_lds = a.cost_n <= COSTN;
    lds_int_t* s_cost = (lds_int_t*)s_heap + 2 * (LDSN + 1);
    if (cost_in_lds) for (int i = lane; i < a.cost_n; i += 64) s_cost[i] = a.cost_lut[i];
    __syncthreads();
    // isValid (astar.cpp:140-149, D5) folded with get_oCost (astar.cpp:181-186): both depend only on the cell's distance
    // value, i.e. on its integer L1 distance
    auto cell_cost = [&](int x, int y) -> int {
        if (x < 0 || y < 0 || x >= a.W || y >= a.H) return ASTAR_INVALID_COST;
        int n = a.l1[(size_t)y * a.W + x];
        if (n == 0xFFFF) return ASTAR_INVALID_COST;                 // distance -1: never > minDist
        return a.cost_lut[min(n, a.cost_n - 1)];
    };
    const bool ok = cell_cost(a.gx, a.gy) != ASTAR_INVALID_COST       // astar.cpp:40-44
                    && cell_cost(a.sx, a.sy) != ASTAR_INVALID_COST    // :46-50
                    && !(a.sx == a.gx && a.sy == a.gy);               // :52-56
    if (!ok) { if (lane == 0) { *a.result = res; if (a.host_out) *(astar_result*)a.host_out = res; } return; }

    // ---- per-lane constants of the sift-down rounds
    const int lk = 31 - __clz(lane + 1);                // level of this lane in a 6-level subtree (lane 63 unused)
    const int lj = (lane + 1) - (1 << lk);
    unsigned long long amask = 0, areq = 0;
    for (int t = 0; t < lk; ++t) {
        const int anc_lane = ((1 << t) - 1) + (lj >> (lk - t));
        amask |= 1ull << anc_lane;
        areq |= (unsigned long long)((lj >> (lk - t - 1)) & 1) << anc_lane;
    }
    // lane-constant neighbour offsets: xDeltas {1,-1,0,0}, yDeltas {0,0,1,-1} (astar.cpp:215-216); lane 4: the cell itself
    const int ddx = lane == 0 ? 1 : (lane == 1 ? -1 : 0);
    const int ddy = lane == 2 ? 1 : (lane == 3 ? -1 : 0);

    int len = 1;
    if (lane == 0) lds_entry_store(0, make_int2(0, (a.sy << 17) | (a.sx << 2)));      // firstNode: all costs 0
    __builtin_amdgcn_wave_barrier();
    bool done = false;
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, ta = 0, tb = 0, tr0 = 0, tr1 = 0, acc_adj = 0, acc_nb = 0, acc_all = 0, acc_wait = 0, acc_push = 0;
    (void)t0; (void)t1; (void)t2; (void)t3; (void)ta; (void)tb; (void)tr0; (void)tr1; (void)acc_adj; (void)acc_nb; (void)acc_all; (void)acc_wait; (void)acc_push;
#ifdef BL_ASTAR_STAMPS
    tr0 = __builtin_amdgcn_s_memrealtime();
#endif
    while (len > 0 && !done) {
        if (res.pops >= a.max_pops) { res.status = ASTAR_ST_LIMIT; break; }
        STAMP(t0);
        __threadfence_block();                           // the previous closing CAS has landed before closed[] is re-read
        const int2 top = lds_entry(0);
        const int cx = (top.y >> 2) & 0x7fff, cy = (int)((unsigned)top.y >> 17), tdir = top.y & 3;
        // ---- issue the per-lane loads of this expansion first; they stay in flight across the heap work below.  The two
        // loads are inline asm so that hipcc does not wait for them at the next join (it would: both results are
        // "atomic"/L1-bypassing reads); the matching s_waitcnt is the asm statement after the pop.
        const int nx = cx + ddx, ny = cy + ddy;
        const bool inb = lane < 5 && nx >= 0 && ny >= 0 && nx < a.W && ny < a.H;
        const int ncell = inb ? ny * a.W + nx : 0;                             // lanes without a neighbour read cell 0, unused
        const uint16_t* l1_addr = a.l1 + ncell;
        const int32_t* cl_addr = a.closed + ncell;
        int my_l1, my_closed;
        asm volatile("global_load_ushort %0, %2, off\n\tglobal_load_dword %1, %3, off sc1"
                     : "=&v"(my_l1), "=&v"(my_closed) : "v"(l1_addr), "v"(cl_addr) : "memory");
        // ---- openList.pop(): std::pop_heap + pop_back
        STAMP(t1);
        len -= 1;
        if (len > 0) {
            if (len < LDSN) {
                // (the last entry straight from LDS: read through heap_read -- which may take it from HBM -- the value carries a wait
                // for EVERY load in flight into the sift-up, the expansion's neighbour loads included: they then ran beside the
                // walk only, not beside the whole pop)
                const int2 value = lds_entry(len);
                heap_adjust<true, LDSN>(g_heap, len, value, lane, lk, lj - 1, amask, areq);
            } else {
                const int2 value = heap_read<LDSN>(g_heap, len);
                heap_adjust_fused<LDSN>(g_heap, len, value, lane, lk, lj - 1, amask, areq, (volatile lds_ll_t*)s_path);
            }
        }
        STAMP(ta);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(my_l1), "+v"(my_closed) :: "memory");
        // closedList.push_back(nNode): only the first entry per cell is ever observed (is_member / get_member).  Lane 4 has
        // loaded the popped cell's own entry: it is closed now unless an earlier pop of this search closed it
        if (lane == 4 && inb && ((unsigned int)my_closed >> 3) != a.closed_gen)
            *(__attribute__((address_space(1))) int*)(a.closed + (cy * a.W + cx)) = (int)((a.closed_gen << 3) | (unsigned int)(res.pops == 0 ? 4 : tdir));      // (a global, not a flat store: gheap_load)
        const bool nclosed = inb && lane < 4 && ((unsigned int)my_closed >> 3) == a.closed_gen;
        if (!inb) my_l1 = 0xFFFF;
        STAMP(t2);
        res.pops += 1;
        // ---- the four neighbours, one per lane (expand_node order is the lane order, astar.cpp:213-233)
        int my_cost = ASTAR_INVALID_COST;
        if (inb && my_l1 != 0xFFFF) {
            const int ci = min(my_l1, a.cost_n - 1);
            if (cost_in_lds) my_cost = s_cost[ci]; else my_cost = *(const __attribute__((address_space(1))) int*)(a.cost_lut + ci);
        }
        // gCost of the popped node: fCost - hCost - oCost of its cell (the start node carries zeros, astar.cpp:66-69)
        const int ax = abs(a.gx - nx), ay = abs(a.gy - ny);                     // get_hCost (:170-179); lane 4: the cell itself
        // 14 * min + 10 * (max - min) = 10 * max + 4 * min, as shifts and adds (a 32-bit v_mul_lo issues at quarter rate)
        const int hmx = max(ax, ay), hmn = min(ax, ay);
        const int hc = (hmx << 3) + (hmx << 1) + (hmn << 2);
        const int c_cost = __builtin_amdgcn_readlane(my_cost, 4);
        const int c_h = __builtin_amdgcn_readlane(hc, 4);
        const int tg = (res.pops == 1) ? 0 : top.x - c_h - c_cost;
        const bool nvalid = lane < 4 && my_cost != ASTAR_INVALID_COST;          // in grid and isValid
        const int f = tg + 10 + hc + (nvalid ? my_cost : 0);                    // get_gCost: 4-connected step
        const unsigned int goal_m = (unsigned int)__ballot(nvalid && nx == a.gx && ny == a.gy);
        // a valid neighbour is pushed unless it is closed (:123) or fNew >= INT16_MAX (:103,124)
        unsigned int push_m = (unsigned int)__ballot(nvalid && !nclosed && 32767 > f);
        const int ey = (ny << 17) | (nx << 2) | lane;
        if (goal_m) push_m &= (goal_m & (0u - goal_m)) - 1u;                    // neighbours before the goal neighbour only
        STAMP(tb);
        {
            // the usual case -- room for all of them and every hole inside the LDS levels -- without the per-push checks (each a
            // branch of this lone wave)
            const int npush = __popc(push_m);
            if (npush > 0 && len + npush <= LDSN && len + npush <= a.heap_cap) {
                while (push_m) {
                    const int kk = __ffs((int)push_m) - 1;
                    push_m &= push_m - 1u;
                    heap_sift_up<true, LDSN>(g_heap, len, make_int2(__builtin_amdgcn_readlane(f, kk), __builtin_amdgcn_readlane(ey, kk)), lane);
                    len += 1;
                }
                res.pushes += npush;
            }
        }
        while (push_m) {
            const int kk = __ffs((int)push_m) - 1;
            push_m &= push_m - 1u;
            if (len >= a.heap_cap) { res.status = ASTAR_ST_CAPACITY; done = true; break; }
            const int fk = __builtin_amdgcn_readlane(f, kk);
            const int yk = __builtin_amdgcn_readlane(ey, kk);
            if (len < LDSN) heap_sift_up<true, LDSN>(g_heap, len, make_int2(fk, yk), lane);      // push_back + std::push_heap
            else heap_sift_up<false, LDSN>(g_heap, len, make_int2(fk, yk), lane);
            len += 1;
            res.pushes += 1;
        }
        if (goal_m && !done) {                                                  // :107-114 -> makePath (:235-274)
            const int kk = __ffs((int)goal_m) - 1;
            if (lane == 0) {
                const int start = a.sy * a.W + a.sx;
                const int gnx = cx + (kk == 0 ? 1 : (kk == 1 ? -1 : 0)), gny = cy + (kk == 2 ? 1 : (kk == 3 ? -1 : 0));
                long long n = 0;
                int cell = gny * a.W + gnx, parent = cy * a.W + cx;
                while (cell != start) {
                    if (n < a.path_cap) a.path[n] = cell;
                    if (n < a.slot_path_cap) a.slot_path[n] = cell;
                    if (a.host_out && n < a.path_head) ((int32_t*)(a.host_out + ASTAR_HDR))[n] = cell;
                    n += 1;
                    cell = parent;
                    const unsigned int cw = (unsigned int)__hip_atomic_load(&a.closed[cell], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const int d = (cw >> 3) == a.closed_gen ? (int)(cw & 7u) : 4;
                    if (d >= 4) break;                                          // reached the start entry
                    parent = cell - (d == 0 ? 1 : (d == 1 ? -1 : (d == 2 ? a.W : -a.W)));
                }
                res.path_len = (int)n;
            }
            res.status = ASTAR_ST_FOUND;
            done = true;
        }
        STAMP(t3);
#ifdef BL_ASTAR_STAMPS
        acc_adj += ta - t1; acc_wait += t2 - ta; acc_nb += tb - t2; acc_push += t3 - tb; acc_all += t3 - t0;
#endif
    }
#ifdef BL_ASTAR_STAMPS
    tr1 = __builtin_amdgcn_s_memrealtime();
    res.stamps[0] = (long long)acc_all; res.stamps[1] = (long long)acc_adj; res.stamps[2] = (long long)acc_nb;
    res.stamps[3] = (long long)(tr1 - tr0); res.stamps[4] = (long long)acc_wait; res.stamps[5] = (long long)acc_push;
#endif
    if (a.pool && res.status == ASTAR_ST_FOUND) {
        // lane 0 wrote the path to this search's scratch; the wave moves it into the shared pool
        __threadfence();
        int n = __builtin_amdgcn_readfirstlane(res.path_len);
        if ((long long)n > a.path_cap) n = (int)a.path_cap;
        unsigned long long off = 0;
        if (lane == 0) off = atomicAdd(a.pool_cursor, (unsigned long long)n);
        off = ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(off >> 32)) << 32) | (unsigned int)__builtin_amdgcn_readfirstlane((int)off);
        for (int i = lane; i < n; i += 64)
            a.pool[off + i] = __hip_atomic_load(&a.path[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        res.path_off = (long long)off;
    }
    if (lane == 0) { *a.result = res; if (a.host_out) *(astar_result*)a.host_out = res; }
}

#include "bl_astar2.h"

// Test entry: n heap operations (key >= 0: push of (key, payload), key in [1, 65534]; key < 0: pop) on k_astar2's open list with
// the storage tiers of configuration cfg (0: a2_big, 1: a2_small, 2: a2_test); out_*: the popped entries in order; cycles[4]:
// device cycles and counts of pushes and pops.
extern "C" int bl_debug_heap2_replay(bl_ctx* ctx, const int32_t* keys, const uint32_t* pays, int n, int cfg, int64_t cap,
                                     uint32_t* out_keys, uint32_t* out_pays, int* out_n, uint64_t* cycles)
{
    const int cfg_in = cfg;
    cfg &= 15;                    // (+16: the general forms only, without the hand-scheduled ones)
    BL_CHECK_ARG(ctx != nullptr && keys != nullptr && pays != nullptr && n >= 0 && cfg >= 0 && cfg <= 2 && cap >= 2 && cap <= AH_MAX_CAP);
    BL_CHECK_ARG(out_keys != nullptr && out_pays != nullptr && out_n != nullptr);
    BL_HIP(hipSetDevice(ctx->device));
    int* d_keys = nullptr; unsigned* d_pays = nullptr; int2* d_heap = nullptr; unsigned* d_ok = nullptr; unsigned* d_op = nullptr; int* d_n = nullptr;
    unsigned long long* d_cyc = nullptr;
    const size_t nn = (size_t)(n > 0 ? n : 1);
    BL_HIP(hipMalloc((void**)&d_keys, nn * 4)); BL_HIP(hipMalloc((void**)&d_pays, nn * 4));
    BL_HIP(hipMalloc((void**)&d_heap, (size_t)cap * 8)); BL_HIP(hipMalloc((void**)&d_ok, nn * 4)); BL_HIP(hipMalloc((void**)&d_op, nn * 4));
    BL_HIP(hipMalloc((void**)&d_n, 4)); BL_HIP(hipMalloc((void**)&d_cyc, 32));
    BL_HIP(hipMemcpyAsync(d_keys, keys, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    BL_HIP(hipMemcpyAsync(d_pays, pays, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    BL_HIP(hipMemsetAsync(d_heap, 0xA5, (size_t)cap * 8, ctx->stream));
    if (cfg == 0) {
        BL_HIP(hipFuncSetAttribute((const void*)k_heap2_probe<a2_big>, hipFuncAttributeMaxDynamicSharedMemorySize, a2_big::BYTES));
        hipLaunchKernelGGL((k_heap2_probe<a2_big>), dim3(1), dim3(64), a2_big::BYTES, ctx->stream, d_keys, d_pays, n, d_heap, (int)cap, d_ok, d_op, d_n, d_cyc, (cfg_in & 16) ? 0 : 1);
    } else if (cfg == 1) hipLaunchKernelGGL((k_heap2_probe<a2_small>), dim3(1), dim3(64), a2_small::BYTES, ctx->stream, d_keys, d_pays, n, d_heap, (int)cap, d_ok, d_op, d_n, d_cyc, (cfg_in & 16) ? 0 : 1);
    else hipLaunchKernelGGL((k_heap2_probe<a2_test>), dim3(1), dim3(64), a2_test::BYTES, ctx->stream, d_keys, d_pays, n, d_heap, (int)cap, d_ok, d_op, d_n, d_cyc, (cfg_in & 16) ? 0 : 1);
    BL_HIP(hipGetLastError());
    BL_HIP(hipMemcpyAsync(out_n, d_n, 4, hipMemcpyDeviceToHost, ctx->stream));
    BL_HIP(hipStreamSynchronize(ctx->stream));
    if (*out_n > 0) { BL_HIP(hipMemcpy(out_keys, d_ok, (size_t)*out_n * 4, hipMemcpyDeviceToHost)); BL_HIP(hipMemcpy(out_pays, d_op, (size_t)*out_n * 4, hipMemcpyDeviceToHost)); }
    if (cycles) BL_HIP(hipMemcpy(cycles, d_cyc, 32, hipMemcpyDeviceToHost));
    void* all[] = {d_keys, d_pays, d_heap, d_ok, d_op, d_n, d_cyc};
    for (void* q : all) BL_HIP(hipFree(q));
    return BL_OK;
}

void bl_astar_free(bl_ctx* ctx)
{
    bl_astar_state* s = ctx->astar;
    if (!s) return;
    if (s->heap) (void)hipFree(s->heap);
    if (s->d_out) (void)hipFree(s->d_out);
    for (int i = 0; i < ASTAR_SLOTS; ++i) if (s->d_slot_path[i]) (void)hipFree(s->d_slot_path[i]);
    if (s->cost_lut) (void)hipFree(s->cost_lut);
    if (s->h_units) (void)hipHostFree(s->h_units);
    for (int i = 0; i < ASTAR_SLOTS; ++i) {
        if (s->h_out[i]) (void)hipHostFree(s->h_out[i]);
        if (s->done[i]) (void)hipEventDestroy(s->done[i]);
    }
    if (s->h_cost) (void)hipHostFree(s->h_cost);
    void* dev[] = {s->b_heap, s->b_closed, s->b_path, s->b_pool, s->b_results, s->b_goals, s->b_cursor, s->g_cells, s->g_vals};
    for (void* q : dev) if (q) (void)hipFree(q);
    void* hst[] = {s->hb_results, s->hb_goals, s->hb_pool, s->hg_cells, s->hg_vals};
    for (void* q : hst) if (q) (void)hipHostFree(q);
    delete s;
    ctx->astar = nullptr;
}

extern "C" int bl_astar_set_open_capacity(bl_ctx* ctx, int64_t nodes)
{
    BL_CHECK_ARG(ctx != nullptr && nodes >= 0);
    ctx->astar_capacity = nodes;
    return BL_OK;
}

static int astar_cells_to_path(const bl_frame& frame, const bl_pose_xyt_t& start, const int32_t* cells, int n,
                               bl_pose_xyt_t* out_path, int cap);

static int astar_prepare(bl_ctx* ctx, const bl_dist* d)
{
    if (!ctx->astar) {
        ctx->astar = new bl_astar_state();
        memset((void*)ctx->astar, 0, sizeof(bl_astar_state));
        ctx->astar->path_head = ASTAR_PATH_HEAD;
        if (getenv("BOTLAB_ASTAR_NO_TURBO")) { const bool off = false; BL_HIP(hipMemcpyToSymbol(HIP_SYMBOL(a2_turbo_enabled), &off, sizeof(off))); }
        if (getenv("BOTLAB_ASTAR_DEEP_AHEAD") && atoi(getenv("BOTLAB_ASTAR_DEEP_AHEAD")) == 0) { const bool off = false; BL_HIP(hipMemcpyToSymbol(HIP_SYMBOL(a2_deep_ahead_enabled), &off, sizeof(off))); }
        if (getenv("BOTLAB_ASTAR_AHEAD") && atoi(getenv("BOTLAB_ASTAR_AHEAD")) == 0) { const bool off = false; BL_HIP(hipMemcpyToSymbol(HIP_SYMBOL(a2_walk_ahead_enabled), &off, sizeof(off))); }
        if (const char* e = getenv("BOTLAB_ASTAR_PATH_HEAD")) { const int v = atoi(e); if (v >= 1 && v <= ASTAR_PATH_HEAD) ctx->astar->path_head = v; }
        for (int i = 0; i < ASTAR_SLOTS; ++i) {
            BL_HIP(hipHostMalloc((void**)&ctx->astar->h_out[i], ASTAR_HDR + ASTAR_PATH_HEAD * 4, hipHostMallocDefault));
            BL_HIP(hipHostGetDevicePointer((void**)&ctx->astar->h_out_dev[i], ctx->astar->h_out[i], 0));
            BL_HIP(hipEventCreateWithFlags(&ctx->astar->done[i], hipEventDisableTiming));
        }
    }
    bl_astar_state* s = ctx->astar;
    int64_t want = ctx->astar_capacity > 0 ? ctx->astar_capacity : (int64_t)1 << 24;     // 16M entries = 128 MB
    if (want > AH_MAX_CAP) want = AH_MAX_CAP;
    if (want < AH_LDS + 1) want = AH_LDS + 1;
    if (s->heap_cap != want) {
        BL_HIP(hipStreamSynchronize(ctx->stream));
        if (s->heap) BL_HIP(hipFree(s->heap));
        s->heap = nullptr;
        BL_HIP(hipMalloc((void**)&s->heap, (size_t)want * sizeof(int2)));
        s->heap_cap = want;
        BL_HIP(hipFuncSetAttribute((const void*)k_astar<AH_LDS, AH_COST_LDS>, hipFuncAttributeMaxDynamicSharedMemorySize, AH_LDS_BYTES));
        BL_HIP(hipFuncSetAttribute((const void*)k_astar2<a2_big>, hipFuncAttributeMaxDynamicSharedMemorySize, a2_big::BYTES));
    }
    size_t n = (size_t)d->frame.width * d->frame.height;
    if (n < ASTAR_PATH_HEAD) n = ASTAR_PATH_HEAD;
    if (n > s->path_cap) {
        BL_HIP(hipStreamSynchronize(ctx->stream));
        if (s->d_out) BL_HIP(hipFree(s->d_out));
        s->d_out = nullptr;
        BL_HIP(hipMalloc((void**)&s->d_out, ASTAR_HDR + n * 4));
        s->path_cap = n;
    }
    size_t sp = (size_t)8 * ((size_t)d->frame.width + d->frame.height);
    if (sp < ASTAR_PATH_HEAD) sp = ASTAR_PATH_HEAD;
    if (sp > s->slot_path_cap) {
        BL_HIP(hipStreamSynchronize(ctx->stream));
        for (int i = 0; i < ASTAR_SLOTS; ++i) {
            if (s->d_slot_path[i]) BL_HIP(hipFree(s->d_slot_path[i]));
            s->d_slot_path[i] = nullptr;
            BL_HIP(hipMalloc((void**)&s->d_slot_path[i], sp * 4));
        }
        s->slot_path_cap = sp;
    }
    int ln = d->frame.width + d->frame.height + 1;
    if (ln > s->cost_lut_cap) {
        BL_HIP(hipStreamSynchronize(ctx->stream));
        if (s->cost_lut) BL_HIP(hipFree(s->cost_lut));
        if (s->h_cost) BL_HIP(hipHostFree(s->h_cost));
        s->cost_lut = nullptr; s->h_cost = nullptr;
        BL_HIP(hipMalloc((void**)&s->cost_lut, (size_t)ln * 4));
        BL_HIP(hipHostMalloc((void**)&s->h_cost, (size_t)ln * 4, hipHostMallocDefault));
        s->cost_lut_cap = ln;
        s->lut_valid = false;
    }
    return BL_OK;
}

// per-distance cell table: validity (astar.cpp:141) and obstacle cost (astar.cpp:181-186) from the float value f[n] a
// cell at L1 distance n holds.  The host's pow() is the reference's pow().  Rebuilt only when the search parameters or
// the table length change.
static int astar_prepare_lut(bl_ctx* ctx, const bl_dist* d, const bl_search_params_t* params)
{
    bl_astar_state* s = ctx->astar;
    const int ln = d->frame.width + d->frame.height + 1;
    const bool same = s->lut_valid && s->lut_n == ln && s->lut_owner == (const void*)d &&
                      memcmp(&s->lut_params, params, sizeof(*params)) == 0;
    if (same) return BL_OK;
    BL_HIP(hipStreamSynchronize(ctx->stream));      // h_cost may still be the source of an earlier copy
    const std::vector<float>& f = *d->lut_host;
    for (int n = 0; n < ln; ++n) {
        float dist = f[n];
        int32_t c;
        if (!(dist > params->minDistanceToObstacle * 1.000001)) c = ASTAR_INVALID_COST;
        else {
            c = 0;
            if (dist > params->minDistanceToObstacle && dist < params->maxDistanceWithCost) {
                double v = pow(params->maxDistanceWithCost - dist * 2000, params->distanceCostExponent);   // float product
                c = (v == v && fabs(v) < 2.0e9) ? static_cast<int>(v) : 0;                                 // (D11: the cast is undefined beyond int)
                if (c == ASTAR_INVALID_COST) c = ASTAR_INVALID_COST + 1;
            }
        }
        s->h_cost[n] = c;
    }
    BL_HIP(hipMemcpyAsync(s->cost_lut, s->h_cost, (size_t)ln * 4, hipMemcpyHostToDevice, ctx->stream));
    // the cost is 0 from maxDistanceWithCost on: the table's constant tail is not worth LDS (or a gather per expansion on grids
    // whose W + H + 1 entries do not fit it)
    int eff = ln;
    while (eff > 1 && s->h_cost[eff - 2] == s->h_cost[ln - 1]) eff--;
    s->lut_eff = eff;
    int lo = 0;
    for (int n = 0; n < ln; ++n) if (s->h_cost[n] != ASTAR_INVALID_COST && s->h_cost[n] < lo) lo = s->h_cost[n];
    s->lut_min = lo;
    s->lut_valid = true; s->lut_n = ln; s->lut_owner = (const void*)d; s->lut_params = *params;
    return BL_OK;
}

// Everything of a search up to the launch: scratch, cost table, closed grid reset, and the kernel arguments.
static int astar_fill(bl_ctx* ctx, const bl_dist* d, const bl_pose_xyt_t* start, const void* d_start,
                      const bl_pose_xyt_t* goal, const bl_search_params_t* params, astar_args* out)
{
    BL_CHECK_ARG(ctx != nullptr && d != nullptr && goal != nullptr && params != nullptr);
    BL_CHECK_ARG(start != nullptr || d_start != nullptr);
    BL_CHECK_ARG(d->valid && d->ctx == ctx);
    BL_CHECK_ARG(d->frame.width <= AH_MAX_DIM && d->frame.height <= AH_MAX_DIM);
    BL_HIP(hipSetDevice(ctx->device));
    int rc = astar_prepare(ctx, d);
    if (rc) return rc;
    bl_astar_state* s = ctx->astar;
    if (s->launched - s->fetched >= ASTAR_SLOTS) {
        bl_set_error("%d A* searches are already pending on this ctx; fetch a result first", ASTAR_SLOTS);
        return BL_ERR_STATE;
    }
    rc = astar_prepare_lut(ctx, d, params);
    if (rc) return rc;
    astar_args& a = *out;
    a.l1 = d->l1; a.W = d->frame.width; a.H = d->frame.height;
    a.cost_lut = s->cost_lut; a.cost_n = s->lut_eff;
    a.heap = s->heap; a.heap_cap = (int)s->heap_cap;
    a.closed = d->closed;
    a.path = (int32_t*)(s->d_out + ASTAR_HDR); a.path_cap = (long long)s->path_cap;
    a.result = (astar_result*)s->d_out;
    a.host_out = s->h_out_dev[s->launched % ASTAR_SLOTS];
    a.slot_path = s->d_slot_path[s->launched % ASTAR_SLOTS]; a.slot_path_cap = (long long)s->slot_path_cap; a.path_head = s->path_head;
    a.frame = d->frame;
    a.batch_goals = nullptr; a.heap_stride = a.closed_stride = a.path_stride = 0; a.pool = nullptr; a.pool_cursor = nullptr;
    a.units = nullptr;
    bl_global_to_cell((double)goal->x, (double)goal->y, d->frame, &a.gx, &a.gy);     // astar.cpp:23-33
    a.sx = 0; a.sy = 0;
    a.start_dev = (const bl_pose_xyt_t*)d_start;
    if (start) {
        a.start_host = *start;
        bl_global_to_cell((double)start->x, (double)start->y, d->frame, &a.sx, &a.sy);
    } else {
        memset(&a.start_host, 0, sizeof(a.start_host));
    }
    a.max_pops = 1ll << 31;
    if (const char* e = getenv("BOTLAB_ASTAR_MAX_POPS")) { const long long v = atoll(e); if (v > 0) a.max_pops = v; }      // probes: stop after v pops
    bl_dist* dm = const_cast<bl_dist*>(d);           // closed[] is search scratch that travels with the grid
    if (++dm->closed_gen >= (1u << 28)) {            // (a wrap every 2.7e8 searches: start over from a zeroed array)
        BL_HIP(hipMemsetAsync(dm->closed, 0, (size_t)a.W * a.H * 4, ctx->stream));
        dm->closed_gen = 1;
    }
    a.closed_gen = dm->closed_gen;
    return BL_OK;
}

// After the launch: the result record + path head go to the pinned ring, the event marks this search done.
static int astar_after(bl_ctx* ctx, const bl_dist* d)
{
    bl_astar_state* s = ctx->astar;
    const int slot = (int)(s->launched % ASTAR_SLOTS);
    // the kernel has written [result][path head] into the pinned slot itself
    BL_HIP(hipEventRecord(s->done[slot], ctx->stream));
    s->slot_frame[slot] = d->frame;
    s->slot_dist[slot] = const_cast<bl_dist*>(d);
    s->slot_search[slot] = s->launched;
    s->launched += 1;
    s->pending = true;
    return BL_OK;
}

// k_astar2 keeps an entry's fCost in 16 bits: usable when no valid cell's obstacle cost can take an fCost to -32768 or below
// (the reference's own parameters give -3998 at the least).  BOTLAB_ASTAR_V1=1: k_astar's 8-byte entries (probes, A/B runs).
static bool astar_split_ok(const bl_astar_state* s)
{
    static const bool force_v1 = getenv("BOTLAB_ASTAR_V1") != nullptr;
    return !force_v1 && s->lut_valid && s->lut_min > -32768;
}

extern "C" int bl_astar_debug_last_kernel(bl_ctx* ctx) { return ctx && ctx->astar ? ctx->astar->last_kernel : 0; }

// Threads of a k_astar2 workgroup.  128: a second wavefront runs the expansions of the LDS-regime loop beside the first
// (bl_astar2_duo.h: -4 .. -9 % per pop there).  Only for searches that have their compute unit to themselves (the 147 KB heap):
// the replanner's units share CUs four at a time, where a second wave per search would take issue slots from the others.
// BOTLAB_ASTAR_DUO=0: one wave everywhere (A/B runs, tests).
static int astar2_threads(bool shares_cu = false)
{
    static const bool duo = !(getenv("BOTLAB_ASTAR_DUO") && atoi(getenv("BOTLAB_ASTAR_DUO")) == 0);
    // (bl_astar2_ahead.h: pops / pushes / expansions on three waves unless BOTLAB_ASTAR_AHEAD says 0 -- the duo loop -- or 1 -- its two-wave form)
    static const bool three = !(getenv("BOTLAB_ASTAR_AHEAD") && atoi(getenv("BOTLAB_ASTAR_AHEAD")) != 2);
    return duo && !shares_cu ? (three ? 192 : 128) : 64;
}

static void astar_launch_kernel(bl_ctx* ctx, const astar_args& a, int workgroups, bool split)
{
    // The 40 KB footprint is for searches that share CUs with the particle filter (the replanner's units): beside its
    // whole-grid LDS image or its LDS window (3 x ~50 KB per CU) a 147 KB heap needs a CU of its own, and a dozen searches in
    // flight then take a dozen CUs out of the filter's single round (4096 x 4096 / 256k particles: k_mcl_main 0.43 -> 0.37 ms
    // with the small footprint, the searches themselves no slower).  A search that runs alone takes the 147 KB heap: an open
    // list spilling past the LDS levels pays an HBM round trip per heap level.
    static const bool force_small = getenv("BOTLAB_ASTAR_SMALL_LDS") != nullptr;     // probes: the replanner's footprint on a lone search
    const bool small = ctx->astar_small_lds || force_small;
    if (ctx->astar) ctx->astar->last_kernel = split ? 2 : 1;
    // (BOTLAB_ASTAR_SMALL_LDS=3: the small footprint on three waves -- tests reach the deep regime's three-wave loop with short searches)
    static const bool small3 = force_small && atoi(getenv("BOTLAB_ASTAR_SMALL_LDS")) == 3;
    if (split && small) hipLaunchKernelGGL((k_astar2<a2_small>), dim3(workgroups), dim3(astar2_threads(!(small3 && !ctx->astar_small_lds))), a2_small::BYTES, ctx->stream, a);
    else if (split) hipLaunchKernelGGL((k_astar2<a2_big>), dim3(workgroups), dim3(astar2_threads()), a2_big::BYTES, ctx->stream, a);
    else if (small) hipLaunchKernelGGL((k_astar<AH_LDS_SMALL, AH_COST_LDS_SMALL>), dim3(workgroups), dim3(64), AH_LDS_SMALL_BYTES, ctx->stream, a);
    else hipLaunchKernelGGL((k_astar<AH_LDS, AH_COST_LDS>), dim3(workgroups), dim3(64), AH_LDS_BYTES, ctx->stream, a);
}

static int astar_launch(bl_ctx* ctx, const bl_dist* d, const bl_pose_xyt_t* start, const void* d_start,
                        const bl_pose_xyt_t* goal, const bl_search_params_t* params)
{
    astar_args a;
    int rc = astar_fill(ctx, d, start, d_start, goal, params, &a);
    if (rc) return rc;
    hipEvent_t e0, e1;
    rc = bl_timer_begin(ctx, BL_K_ASTAR, &e0, &e1);
    if (rc) return rc;
    astar_launch_kernel(ctx, a, 1, astar_split_ok(ctx->astar));
    BL_HIP(hipGetLastError());
    rc = bl_timer_end(ctx, BL_K_ASTAR, e0, e1);
    if (rc) return rc;
    return astar_after(ctx, d);
}

// n unrelated searches (own ctx state and distance grid each, all on the stream the ctxs share, grids of one size) as
// ONE launch, a workgroup each: what a replanner lane does with the submissions it has collected.
static int astar_launch_units(int n, bl_ctx* const* ctxs, bl_dist* const* dists, const void* const* d_starts,
                              const bl_pose_xyt_t* goals, const bl_search_params_t* params)
{
    BL_CHECK_ARG(n >= 1 && n <= ASTAR_MAX_UNITS);
    if (n == 1) return astar_launch(ctxs[0], dists[0], nullptr, d_starts[0], &goals[0], &params[0]);
    astar_args a;
    bool split = true;
    int rc0 = astar_prepare(ctxs[0], dists[0]);
    if (rc0) return rc0;
    bl_astar_state* s0 = ctxs[0]->astar;
    if (!s0->h_units) BL_HIP(hipHostMalloc((void**)&s0->h_units, sizeof(astar_unit) * ASTAR_MAX_UNITS * ASTAR_SLOTS, hipHostMallocDefault));
    // at most ASTAR_SLOTS launches of this ctx are pending, so the ring entry of the oldest is free again
    astar_unit* units = s0->h_units + (size_t)(s0->launched % ASTAR_SLOTS) * ASTAR_MAX_UNITS;
    for (int b = n - 1; b >= 0; --b) {               // unit 0 last: its arguments stay in `a` for the launch
        BL_CHECK_ARG(ctxs[b]->stream == ctxs[0]->stream);
        BL_CHECK_ARG(dists[b]->frame.width == dists[0]->frame.width && dists[b]->frame.height == dists[0]->frame.height);
        int rc = astar_fill(ctxs[b], dists[b], nullptr, d_starts[b], &goals[b], &params[b], &a);
        if (rc) return rc;
        split = split && astar_split_ok(ctxs[b]->astar);
        astar_unit& u = units[b];
        u.l1 = a.l1; u.cost_lut = a.cost_lut; u.heap = a.heap; u.closed = a.closed; u.path = a.path; u.result = a.result;
        u.start_dev = a.start_dev; u.start_host = a.start_host; u.sx = a.sx; u.sy = a.sy; u.gx = a.gx; u.gy = a.gy;
        u.host_out = a.host_out; u.slot_path = a.slot_path; u.cost_n = a.cost_n; u.closed_gen = a.closed_gen;
    }
    a.units = units;
    bl_ctx* ctx = ctxs[0];
    hipEvent_t e0, e1;
    int rc = bl_timer_begin(ctx, BL_K_ASTAR, &e0, &e1);
    if (rc) return rc;
    astar_launch_kernel(ctx, a, n, split);
    BL_HIP(hipGetLastError());
    rc = bl_timer_end(ctx, BL_K_ASTAR, e0, e1);
    if (rc) return rc;
    for (int b = 0; b < n; ++b) { rc = astar_after(ctxs[b], dists[b]); if (rc) return rc; }
    return BL_OK;
}

extern "C" int bl_astar_search_async(bl_ctx* ctx, const bl_dist* d, const bl_pose_xyt_t* start, const bl_pose_xyt_t* goal,
                                     const bl_search_params_t* params)
{
    BL_CHECK_ARG(start != nullptr);
    return astar_launch(ctx, d, start, nullptr, goal, params);
}

extern "C" int bl_astar_search_async_dev_start(bl_ctx* ctx, const bl_dist* d, const void* d_start, const bl_pose_xyt_t* goal,
                                               const bl_search_params_t* params)
{
    BL_CHECK_ARG(d_start != nullptr);
    return astar_launch(ctx, d, nullptr, d_start, goal, params);
}

extern "C" int bl_astar_search_result(bl_ctx* ctx, bl_pose_xyt_t* out_path, int cap, int* out_len, int64_t* stats)
{
    BL_CHECK_ARG(ctx != nullptr && out_path != nullptr && cap >= 1 && out_len != nullptr);
    bl_astar_state* s = ctx->astar;
    if (!s || s->launched == s->fetched) { bl_set_error("no A* search pending"); return BL_ERR_STATE; }
    const int slot = (int)(s->fetched % ASTAR_SLOTS);           // results come back in launch order
    BL_HIP(hipEventSynchronize(s->done[slot]));                 // waits for THIS search only; later work keeps running
    s->fetched += 1;
    s->pending = s->launched != s->fetched;
    s->frame = s->slot_frame[slot];
    astar_result r = *(const astar_result*)s->h_out[slot];
    { const int rcb = dist_check_broken(s->slot_dist[slot]); if (rcb) { out_path[0] = r.start; *out_len = 1; return rcb; } }
    if (stats) { stats[0] = r.pops; stats[1] = r.pushes; }
#ifdef BL_ASTAR_STAMPS
    {
        const double pp = (double)(r.pops ? r.pops : 1);
        if (astar2_threads() >= 128 && !(ctx->astar_small_lds || getenv("BOTLAB_ASTAR_SMALL_LDS")) && !(getenv("BOTLAB_ASTAR_AHEAD") && atoi(getenv("BOTLAB_ASTAR_AHEAD")) == 0))
            fprintf(stderr, "[astar ahead stamps] pops %lld: cycles/pop inside barriers -- wave 0: B1 %.0f, B2 %.0f; wave 1: B1 %.0f, B2 %.0f | walks taken again %.4f per pop | search %.0f cycles/pop | expansions made ahead %lld, not %lld\n", r.pops,
                    (double)r.stamps[0] / pp, (double)r.stamps[1] / pp, (double)r.stamps[2] / pp, (double)r.stamps[4] / pp, (double)r.stamps[5] / pp,
                    (double)r.stamps[3] * 1e-8 * 2.4e9 / pp, (long long)(r.path_off & 0xffffffffll), (long long)(r.path_off >> 32));
        else if (astar2_threads() >= 128 && !(ctx->astar_small_lds || getenv("BOTLAB_ASTAR_SMALL_LDS")))
            fprintf(stderr, "[astar duo stamps] pops %lld: cycles/pop -- wave 1: Z to the wait %.0f, the wait %.0f, expansion %.0f, record %.0f; wave 0 inside Y %.0f | search %.0f cycles/pop | tops foreseen %lld, not %lld\n", r.pops,
                    (double)r.stamps[0] / pp, (double)r.stamps[1] / pp, (double)r.stamps[2] / pp, (double)r.stamps[4] / pp, (double)r.stamps[5] / pp,
                    (double)r.stamps[3] * 1e-8 * 2.4e9 / pp, (long long)(r.path_off & 0xffffffffll), (long long)(r.path_off >> 32));
        else
        fprintf(stderr, "[astar stamps] pops %lld pushes %lld cycles/pop: all %.0f = issue %.0f + adjust %.0f + loadwait %.0f + expand %.0f + pushes %.0f | clock %.2f GHz\n",
                r.pops, r.pushes, (double)r.stamps[0] / pp,
                ((double)r.stamps[0] - (double)r.stamps[1] - (double)r.stamps[2] - (double)r.stamps[4] - (double)r.stamps[5]) / pp,
                (double)r.stamps[1] / pp, (double)r.stamps[4] / pp, (double)r.stamps[2] / pp, (double)r.stamps[5] / pp,
                r.stamps[3] ? (double)r.stamps[0] / ((double)r.stamps[3] * 10.0) : 0.0);
    }
#endif
    s->start = r.start;
    out_path[0] = s->start;                                            // path.path.push_back(start) (astar.cpp:21)
    *out_len = 1;
    if (r.status == ASTAR_ST_CAPACITY) { bl_set_error("A* open list exceeded its capacity (%lld pops)", r.pops); return BL_ERR_CAPACITY; }
    if (r.status == ASTAR_ST_LIMIT) { bl_set_error("A* pop limit reached"); return BL_ERR_CAPACITY; }
    if (r.status == ASTAR_ST_BROKEN) { bl_set_error("A* search gave up: its two wavefronts lost each other (BOTLAB_ASTAR_DUO=0 takes the one-wave loop)"); return BL_ERR_STATE; }
    if (r.status != ASTAR_ST_FOUND) return BL_OK;
    // makePath (astar.cpp:235-274): cells come goal-first; poses are emitted start-side first
    std::vector<int32_t> cells((size_t)r.path_len);
    if (r.path_len <= s->path_head) memcpy(cells.data(), s->h_out[slot] + ASTAR_HDR, (size_t)r.path_len * 4);
    else if ((size_t)r.path_len <= s->slot_path_cap) {
        // longer than the head that came with the result record: the slot's own copy of the path is complete (the event has
        // fired) and stays untouched until ASTAR_SLOTS further searches have been launched, which cannot happen before this fetch
        BL_HIP(hipMemcpy(cells.data(), s->d_slot_path[slot], (size_t)r.path_len * 4, hipMemcpyDeviceToHost));
    } else {
        // longer even than 8 * (W + H) cells: only the search scratch holds all of it, and a later search reuses that
        if (s->pending) { bl_set_error("A* path of %d cells exceeds the per-result path buffer (%zu) while later searches are pending", r.path_len, s->slot_path_cap); return BL_ERR_CAPACITY; }
        BL_HIP(hipStreamSynchronize(ctx->stream));
        BL_HIP(hipMemcpy(cells.data(), s->d_out + ASTAR_HDR, (size_t)r.path_len * 4, hipMemcpyDeviceToHost));
    }
    const int total = astar_cells_to_path(s->frame, s->start, cells.data(), r.path_len, out_path, cap);
    *out_len = total;
    return BL_OK;
}

extern "C" int bl_astar_search(bl_ctx* ctx, const bl_dist* d, const bl_pose_xyt_t* start, const bl_pose_xyt_t* goal,
                               const bl_search_params_t* params, bl_pose_xyt_t* out_path, int cap, int* out_len, int64_t* stats)
{
    int rc = bl_astar_search_async(ctx, d, start, goal, params);
    if (rc) return rc;
    return bl_astar_search_result(ctx, out_path, cap, out_len, stats);
}

// makePath's pose list (astar.cpp:235-274) from the goal-first cell chain the kernel produced: [start pose] + cells from
// the start side to the goal.  Returns the path length; writes at most cap poses.
static int astar_cells_to_path(const bl_frame& frame, const bl_pose_xyt_t& start, const int32_t* cells, int n,
                               bl_pose_xyt_t* out_path, int cap)
{
    if (cap >= 1) out_path[0] = start;                                 // path.path.push_back(start) (astar.cpp:21)
    float prevX = 0, prevY = 0;
    for (int i = 0; i < n; ++i) {
        int cx = cells[i] % frame.width, cy = cells[i] / frame.width;
        bl_pose_xyt_t p;
        p.utime = 0;                                                   // D6
        p.x = (float)((double)frame.ox + (double)cx * (double)frame.mpc);     // grid_utils.hpp:14-19
        p.y = (float)((double)frame.oy + (double)cy * (double)frame.mpc);
        if (i == 0) p.theta = (float)(double)start.theta;
        else p.theta = (float)atan2((double)prevY - (double)cy, (double)prevX - (double)cx);
        prevX = (float)cx; prevY = (float)cy;
        const int at = n - i;                                          // poses are emitted start-side first
        if (at < cap) out_path[at] = p;
    }
    return 1 + n;
}

// ------------------------------------------------------------------------------------------------ batched searches
// n independent search_for_path calls from ONE start on one distance grid, one wavefront (workgroup) each, concurrently.
// This is what plan_path_to_frontier's candidate sweep (frontiers.cpp:145-204: up to 164 planPath calls per ring) and
// any "try several goals" caller needs; each search is the same exact emulation as the single form.
#define ASTAR_BATCH_MAX 64
static int astar_batch_prepare(bl_ctx* ctx, const bl_dist* d, int want)
{
    bl_astar_state* s = ctx->astar;
    const size_t cells = (size_t)d->frame.width * d->frame.height;
    // searches per launch: bounded by 8 GB of closed grids (64 searches at 4096 x 4096)
    int cap = ASTAR_BATCH_MAX;
    while (cap > 1 && (size_t)cap * cells * 4 > ((size_t)8 << 30)) cap >>= 1;
    if (want < cap) cap = want < 8 ? 8 : want;
    if (cap > ASTAR_BATCH_MAX) cap = ASTAR_BATCH_MAX;
    if (s->b_cap >= cap && s->b_cells >= cells) return BL_OK;
    BL_HIP(hipStreamSynchronize(ctx->stream));
    void* dev[] = {s->b_heap, s->b_closed, s->b_path, s->b_pool, s->b_results, s->b_goals, s->b_cursor};
    for (void* q : dev) if (q) BL_HIP(hipFree(q));
    void* hst[] = {s->hb_results, s->hb_goals, s->hb_pool};
    for (void* q : hst) if (q) BL_HIP(hipHostFree(q));
    s->b_heap = nullptr; s->b_closed = nullptr; s->b_path = nullptr; s->b_pool = nullptr; s->b_results = nullptr;
    s->b_goals = nullptr; s->b_cursor = nullptr; s->hb_results = nullptr; s->hb_goals = nullptr; s->hb_pool = nullptr;
    s->b_cap = 0;
    if (cap < s->b_cap) cap = s->b_cap;
    int64_t heap_each = ctx->astar_capacity > 0 ? ctx->astar_capacity : (int64_t)1 << 23;     // 8M entries = 64 MB per search
    if (heap_each > AH_MAX_CAP) heap_each = AH_MAX_CAP;
    if (heap_each < AH_LDS + 1) heap_each = AH_LDS + 1;
    size_t path_each = cells < 65536 ? cells : 65536;
    if (path_each < 64) path_each = 64;
    BL_HIP(hipMalloc((void**)&s->b_heap, (size_t)cap * heap_each * sizeof(int2)));
    BL_HIP(hipMalloc((void**)&s->b_closed, (size_t)cap * cells * 4));
    BL_HIP(hipMalloc((void**)&s->b_path, (size_t)cap * path_each * 4));
    BL_HIP(hipMalloc((void**)&s->b_pool, (size_t)cap * path_each * 4));
    BL_HIP(hipMalloc((void**)&s->b_results, (size_t)cap * ASTAR_HDR));
    BL_HIP(hipMalloc((void**)&s->b_goals, (size_t)cap * sizeof(int2)));
    BL_HIP(hipMalloc((void**)&s->b_cursor, 8));
    BL_HIP(hipHostMalloc((void**)&s->hb_results, (size_t)cap * ASTAR_HDR, hipHostMallocDefault));
    BL_HIP(hipHostMalloc((void**)&s->hb_goals, (size_t)cap * sizeof(int2), hipHostMallocDefault));
    s->hb_pool_cap = (size_t)cap * path_each;
    BL_HIP(hipHostMalloc((void**)&s->hb_pool, s->hb_pool_cap * 4, hipHostMallocDefault));
    s->b_cap = cap; s->b_cells = cells; s->b_heap_each = heap_each; s->b_path_each = path_each;
    BL_HIP(hipFuncSetAttribute((const void*)k_astar<AH_LDS, AH_COST_LDS>, hipFuncAttributeMaxDynamicSharedMemorySize, AH_LDS_BYTES));
    BL_HIP(hipFuncSetAttribute((const void*)k_astar2<a2_big>, hipFuncAttributeMaxDynamicSharedMemorySize, a2_big::BYTES));
    return BL_OK;
}

// Runs the searches for goal CELLS goals[0..n) (duplicates allowed) and hands each result to `sink(i, status, cells, len, pops, pushes)`.
template <class Sink>
static int astar_batch_cells(bl_ctx* ctx, const bl_dist* d, const bl_pose_xyt_t* start, const int2* goals, int n,
                             const bl_search_params_t* params, Sink sink)
{
    BL_CHECK_ARG(ctx != nullptr && d != nullptr && start != nullptr && params != nullptr && n >= 0);
    BL_CHECK_ARG(d->valid && d->ctx == ctx);
    BL_CHECK_ARG(d->frame.width <= AH_MAX_DIM && d->frame.height <= AH_MAX_DIM);
    if (n == 0) return BL_OK;
    BL_HIP(hipSetDevice(ctx->device));
    int rc = astar_prepare(ctx, d);
    if (rc) return rc;
    bl_astar_state* s = ctx->astar;
    if (s->launched != s->fetched) { bl_set_error("a single A* search is pending on this ctx; fetch it before a batch"); return BL_ERR_STATE; }
    rc = astar_batch_prepare(ctx, d, n);
    if (rc) return rc;
    rc = astar_prepare_lut(ctx, d, params);
    if (rc) return rc;
    const size_t cells = (size_t)d->frame.width * d->frame.height;
    for (int base = 0; base < n; base += s->b_cap) {
        const int m = (n - base) < s->b_cap ? (n - base) : s->b_cap;
        memcpy(s->hb_goals, goals + base, (size_t)m * sizeof(int2));
        BL_HIP(hipMemcpyAsync(s->b_goals, s->hb_goals, (size_t)m * sizeof(int2), hipMemcpyHostToDevice, ctx->stream));
        BL_HIP(hipMemsetAsync(s->b_closed, 0, (size_t)m * cells * 4, ctx->stream));
        BL_HIP(hipMemsetAsync(s->b_cursor, 0, 8, ctx->stream));
        astar_args a;
        a.l1 = d->l1; a.W = d->frame.width; a.H = d->frame.height;
        a.cost_lut = s->cost_lut; a.cost_n = s->lut_eff;
        a.heap = s->b_heap; a.heap_cap = (int)s->b_heap_each;
        a.closed = s->b_closed; a.closed_gen = 1;
        a.path = s->b_path; a.path_cap = (long long)s->b_path_each;
        a.slot_path = nullptr; a.slot_path_cap = 0; a.path_head = 0; a.host_out = nullptr;
        a.result = (astar_result*)s->b_results;
        a.frame = d->frame;
        a.gx = a.gy = 0;
        a.start_dev = nullptr;
        a.start_host = *start;
        bl_global_to_cell((double)start->x, (double)start->y, d->frame, &a.sx, &a.sy);
        a.max_pops = 1ll << 31;
        a.units = nullptr;
        a.batch_goals = s->b_goals;
        a.heap_stride = s->b_heap_each; a.closed_stride = (long long)cells; a.path_stride = (long long)s->b_path_each;
        a.pool = s->b_pool; a.pool_cursor = s->b_cursor;
        hipEvent_t e0, e1;
        rc = bl_timer_begin(ctx, BL_K_ASTAR, &e0, &e1);
        if (rc) return rc;
        if (astar_split_ok(s)) hipLaunchKernelGGL((k_astar2<a2_big>), dim3(m), dim3(astar2_threads()), a2_big::BYTES, ctx->stream, a);
        else hipLaunchKernelGGL((k_astar<AH_LDS, AH_COST_LDS>), dim3(m), dim3(64), AH_LDS_BYTES, ctx->stream, a);
        BL_HIP(hipGetLastError());
        rc = bl_timer_end(ctx, BL_K_ASTAR, e0, e1);
        if (rc) return rc;
        BL_HIP(hipMemcpyAsync(s->hb_results, s->b_results, (size_t)m * ASTAR_HDR, hipMemcpyDeviceToHost, ctx->stream));
        BL_HIP(hipStreamSynchronize(ctx->stream));
        rc = dist_check_broken(const_cast<bl_dist*>(d));           // searches on distances that were never finished are no results
        if (rc) return rc;
        size_t total = 0;
        for (int i = 0; i < m; ++i) {
            const astar_result* r = (const astar_result*)(s->hb_results + (size_t)i * ASTAR_HDR);
            if (r->status == ASTAR_ST_CAPACITY) { bl_set_error("A* open list exceeded its capacity (%lld pops)", r->pops); return BL_ERR_CAPACITY; }
            if (r->status == ASTAR_ST_LIMIT) { bl_set_error("A* pop limit reached"); return BL_ERR_CAPACITY; }
            if (r->status == ASTAR_ST_BROKEN) { bl_set_error("A* search gave up: its two wavefronts lost each other (BOTLAB_ASTAR_DUO=0 takes the one-wave loop)"); return BL_ERR_STATE; }
            if (r->status == ASTAR_ST_FOUND) {
                if ((size_t)r->path_len > s->b_path_each) { bl_set_error("A* path of %d cells exceeds the batch path capacity", r->path_len); return BL_ERR_CAPACITY; }
                total += (size_t)r->path_len;
            }
        }
        if (total > 0) {
            BL_HIP(hipMemcpyAsync(s->hb_pool, s->b_pool, total * 4, hipMemcpyDeviceToHost, ctx->stream));
            BL_HIP(hipStreamSynchronize(ctx->stream));
        }
        for (int i = 0; i < m; ++i) {
            const astar_result* r = (const astar_result*)(s->hb_results + (size_t)i * ASTAR_HDR);
            const bool found = r->status == ASTAR_ST_FOUND;
            sink(base + i, found, found ? s->hb_pool + r->path_off : nullptr, found ? r->path_len : 0, r->pops, r->pushes);
        }
    }
    return BL_OK;
}

extern "C" int bl_astar_search_batch(bl_ctx* ctx, const bl_dist* d, const bl_pose_xyt_t* start, const bl_pose_xyt_t* goals, int n,
                                     const bl_search_params_t* params, bl_pose_xyt_t* out_paths, int cap_each, int* out_lens,
                                     int64_t* stats)
{
    BL_CHECK_ARG(goals != nullptr && out_paths != nullptr && out_lens != nullptr && cap_each >= 1 && d != nullptr);
    std::vector<int2> cells((size_t)(n > 0 ? n : 0));
    for (int i = 0; i < n; ++i) bl_global_to_cell((double)goals[i].x, (double)goals[i].y, d->frame, &cells[i].x, &cells[i].y);   // astar.cpp:23-33
    const bl_frame frame = d->frame;
    const bl_pose_xyt_t st = *start;
    return astar_batch_cells(ctx, d, start, cells.data(), n, params,
                             [&](int i, bool found, const int32_t* pc, int len, long long pops, long long pushes) {
                                 (void)found;
                                 out_lens[i] = astar_cells_to_path(frame, st, pc, len, out_paths + (size_t)i * cap_each, cap_each);
                                 if (stats) { stats[2 * i] = pops; stats[2 * i + 1] = pushes; }
                             });
}

// distances_(x, y) for n cells in one round trip (isValidGoal / isPathSafe of many candidates); off-grid -> NaN
__global__ void k_dist_gather(const float* __restrict__ cells, int W, int H, const int2* __restrict__ q, int n, float* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int2 c = q[i];
    out[i] = (c.x >= 0 && c.y >= 0 && c.x < W && c.y < H) ? cells[(size_t)c.y * W + c.x] : __int_as_float(0x7fc00000);
}

extern "C" int bl_dist_gather(bl_dist* d, const int32_t* xy_cells, int n, float* out)
{
    BL_CHECK_ARG(d != nullptr && d->valid && n >= 0 && (n == 0 || (xy_cells != nullptr && out != nullptr)));
    if (n == 0) return BL_OK;
    bl_ctx* ctx = d->ctx;
    BL_HIP(hipSetDevice(ctx->device));
    int rc = astar_prepare(ctx, d);
    if (rc) return rc;
    rc = dist_floats(d);
    if (rc) return rc;
    bl_astar_state* s = ctx->astar;
    if (s->g_cap < n) {
        BL_HIP(hipStreamSynchronize(ctx->stream));
        if (s->g_cells) BL_HIP(hipFree(s->g_cells));
        if (s->g_vals) BL_HIP(hipFree(s->g_vals));
        if (s->hg_cells) BL_HIP(hipHostFree(s->hg_cells));
        if (s->hg_vals) BL_HIP(hipHostFree(s->hg_vals));
        s->g_cells = nullptr; s->g_vals = nullptr; s->hg_cells = nullptr; s->hg_vals = nullptr; s->g_cap = 0;
        int cap = n < 4096 ? 4096 : n * 2;
        BL_HIP(hipMalloc((void**)&s->g_cells, (size_t)cap * sizeof(int2)));
        BL_HIP(hipMalloc((void**)&s->g_vals, (size_t)cap * 4));
        BL_HIP(hipHostMalloc((void**)&s->hg_cells, (size_t)cap * sizeof(int2), hipHostMallocDefault));
        BL_HIP(hipHostMalloc((void**)&s->hg_vals, (size_t)cap * 4, hipHostMallocDefault));
        s->g_cap = cap;
    }
    memcpy(s->hg_cells, xy_cells, (size_t)n * sizeof(int2));
    BL_HIP(hipMemcpyAsync(s->g_cells, s->hg_cells, (size_t)n * sizeof(int2), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_dist_gather, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, d->cells, d->frame.width, d->frame.height,
                       s->g_cells, n, s->g_vals);
    BL_HIP(hipGetLastError());
    BL_HIP(hipMemcpyAsync(s->hg_vals, s->g_vals, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    BL_HIP(hipStreamSynchronize(ctx->stream));
    rc = dist_check_broken(d);
    if (rc) return rc;
    memcpy(out, s->hg_vals, (size_t)n * 4);
    return BL_OK;
}

// =============================================================================================== asynchronous replanner
// The reference runs its planner in a separate process that consumes the maps and poses the SLAM process publishes
// (src/planning/exploration.cpp:300-317 on every SLAM_MAP message).  bl_planner is that arrangement on one device: a
// second HIP stream runs setDistances + search_for_path against a SNAPSHOT of the map and of the pose taken on the
// SLAM stream, so the one-wavefront A* overlaps the next scan's particle-filter kernels instead of serialising with
// them.  Two snapshot slots; every hand-off between the two streams is an event.
#define PLANNER_SLOTS 2
#define PLANNER_MAX_LANES 4
#define PLANNER_MAX_BATCH ASTAR_MAX_UNITS

// A unit = what one replan needs: a ctx (A* scratch + result ring) on the lane's stream, a distance grid, snapshot slots.
// A lane = one side stream with `batch` units.  Consecutive submissions fill the units of one lane; when the last one is
// in, their searches go out as ONE k_astar launch, a workgroup each, and the next lane takes over.  So up to lanes x batch
// replans (independent searches on independent snapshots) run concurrently, each one wavefront on its own CU, although
// the runtime has only four hardware queues (a fifth stream would share one and serialise).  batch = 1 is a launch per
// submission (lowest latency: the 200x200 case, where a search is shorter than a step); results always come back in
// submission order.
struct planner_unit {
    bl_ctx* ctx;
    bl_dist* dist;
    bl_grid* snap[PLANNER_SLOTS];
    bl_pose_xyt_t* pose[PLANNER_SLOTS];
    hipEvent_t snap_ready[PLANNER_SLOTS];   // recorded on main after the snapshot copy
    bl_pose_xyt_t goal;
    bl_search_params_t params;
};

struct planner_lane {
    bl_ctx* side;                           // unit[0].ctx: owns the lane's stream
    planner_unit unit[PLANNER_MAX_BATCH];
    hipEvent_t slot_free[PLANNER_SLOTS];    // recorded on side after the searches that read the slot
    bool slot_used[PLANNER_SLOTS];
    int64_t batches;                        // batches launched on this lane
    int filled;                             // units of the current batch whose snapshot + distance grid are enqueued
};

struct planner_ticket { int lane, unit; };

struct bl_planner {
    bl_ctx* main;                       // the SLAM ctx (not owned)
    int lanes, batch;
    planner_lane lane[PLANNER_MAX_LANES];
    int cur_lane;                       // the lane collecting submissions
    int64_t submitted, fetched;
    std::deque<planner_ticket>* tickets;    // outstanding submissions, oldest first
    unsigned long long* d_flag;         // number of the last submission whose snapshot is complete (written by the snapshot kernel)
    unsigned int* d_done;               // workgroup counter of the multi-workgroup snapshot kernel
    bool reserved;
    bool handoff_flag;                  // lane waits on the flag word (hipStreamWaitValue64) instead of an event
};

extern "C" int bl_planner_create_batched(bl_ctx* ctx, int lanes, int batch, bl_planner** out)
{
    BL_CHECK_ARG(ctx != nullptr && out != nullptr && lanes >= 1 && lanes <= PLANNER_MAX_LANES);
    BL_CHECK_ARG(batch >= 1 && batch <= PLANNER_MAX_BATCH);
    BL_HIP(hipSetDevice(ctx->device));
    bl_planner* p = new bl_planner();
    memset((void*)p, 0, sizeof(*p));
    p->main = ctx;
    p->lanes = lanes;
    p->batch = batch;
    p->tickets = new std::deque<planner_ticket>();
    p->handoff_flag = getenv("BOTLAB_PLANNER_HANDOFF_FLAG") != nullptr;
    BL_HIP(hipMalloc((void**)&p->d_flag, 8));
    BL_HIP(hipMalloc((void**)&p->d_done, 4));
    BL_HIP(hipMemsetAsync(p->d_flag, 0, 8, ctx->stream));        // on the SLAM stream: the null stream (and its hardware queue) stays untouched
    BL_HIP(hipMemsetAsync(p->d_done, 0, 4, ctx->stream));
    BL_HIP(hipStreamSynchronize(ctx->stream));
    for (int l = 0; l < lanes; ++l) {
        planner_lane& L = p->lane[l];
        for (int u = 0; u < batch; ++u) {
            planner_unit& U = L.unit[u];
            int rc = bl_ctx_create(ctx->device, u == 0 ? nullptr : (void*)L.side->stream, &U.ctx);
            if (rc) return rc;
            if (u == 0) L.side = U.ctx;
            // searches that co-run with the SLAM stream's kernels take the small LDS footprint; a planner with one search in flight
            // (the closed loop: every path fetched before the next step) gives it a CU's whole LDS
            U.ctx->astar_small_lds = lanes * batch > 1;
            // open list of a unit: 4 M entries (32 MB) unless BOTLAB_PLANNER_OPEN_CAPACITY says otherwise -- up to 4 x 32 units exist
            U.ctx->astar_capacity = getenv("BOTLAB_PLANNER_OPEN_CAPACITY") ? atoll(getenv("BOTLAB_PLANNER_OPEN_CAPACITY")) : ((int64_t)1 << 22);
            rc = bl_dist_create(U.ctx, &U.dist);
            if (rc) return rc;
            for (int i = 0; i < PLANNER_SLOTS; ++i) {
                BL_HIP(hipMalloc((void**)&U.pose[i], sizeof(bl_pose_xyt_t)));
                BL_HIP(hipEventCreateWithFlags(&U.snap_ready[i], hipEventDisableTiming));
            }
        }
        for (int i = 0; i < PLANNER_SLOTS; ++i) BL_HIP(hipEventCreateWithFlags(&L.slot_free[i], hipEventDisableTiming));
    }
    *out = p;
    return BL_OK;
}

extern "C" int bl_planner_create(bl_ctx* ctx, int lanes, bl_planner** out) { return bl_planner_create_batched(ctx, lanes, 1, out); }

extern "C" void bl_planner_destroy(bl_planner* p)
{
    if (!p) return;
    (void)hipStreamSynchronize(p->main->stream);
    for (int l = 0; l < p->lanes; ++l) {
        planner_lane& L = p->lane[l];
        if (!L.side) continue;
        (void)hipStreamSynchronize(L.side->stream);
        for (int u = p->batch - 1; u >= 0; --u) {            // unit 0 owns the stream: last
            planner_unit& U = L.unit[u];
            if (!U.ctx) continue;
            for (int i = 0; i < PLANNER_SLOTS; ++i) {
                if (U.snap[i]) bl_grid_destroy(U.snap[i]);
                if (U.pose[i]) (void)hipFree(U.pose[i]);
                if (U.snap_ready[i]) (void)hipEventDestroy(U.snap_ready[i]);
            }
            if (U.dist) bl_dist_destroy(U.dist);
            bl_ctx_destroy(U.ctx);
        }
        for (int i = 0; i < PLANNER_SLOTS; ++i) if (L.slot_free[i]) (void)hipEventDestroy(L.slot_free[i]);
    }
    if (p->d_flag) (void)hipFree(p->d_flag);
    if (p->d_done) (void)hipFree(p->d_done);
    delete p->tickets;
    delete p;
}

// map + pose snapshot as ONE kernel on the SLAM stream (two hipMemcpyAsync D2D cost two copy-engine handshakes there).
// The last workgroup to finish publishes the submission number to the planner's flag word; the lane stream waits for it
// with hipStreamWaitValue64 -- nothing is recorded on the SLAM stream (an event record costs ~4-7 us of stream time).
__global__ __launch_bounds__(256) void k_planner_snapshot(const int8_t* __restrict__ src, int8_t* __restrict__ dst, size_t n,
                                                          const bl_pose_xyt_t* __restrict__ src_pose, bl_pose_xyt_t* __restrict__ dst_pose,
                                                          unsigned int* done_count, unsigned long long* flag, unsigned long long seq)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t n16 = n / 16;
    const int4* s4 = (const int4*)src;
    int4* d4 = (int4*)dst;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) d4[i] = s4[i];
    for (size_t i = n16 * 16 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[i];
    if (blockIdx.x == 0 && threadIdx.x == 0) *dst_pose = *src_pose;
    if (flag) {                                     // flag hand-off only; an event hand-off needs nothing here
        // one fence per workgroup, after its barrier (a fence in every thread writes L2 back half a million times on a
        // 16 MB grid: the copy then took 230 us instead of ~10)
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            const unsigned int old = atomicAdd(done_count, 1u);
            if (old == gridDim.x - 1) {
                *done_count = 0;
                __threadfence();
                __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

// The searches of the lane's collected submissions as one launch; the next lane collects from here on.
static int planner_launch_lane(bl_planner* p, int l)
{
    planner_lane& L = p->lane[l];
    if (L.filled == 0) return BL_OK;
    const int slot = (int)(L.batches % PLANNER_SLOTS);
    if (p->batch > 1 && !p->handoff_flag) {
        // the deferred hand-over of bl_planner_commit: one event behind the newest snapshot covers every snapshot of the batch
        BL_HIP(hipEventRecord(L.unit[0].snap_ready[slot], p->main->stream));
        BL_HIP(hipStreamWaitEvent(L.side->stream, L.unit[0].snap_ready[slot], 0));
        bl_dist* bd[PLANNER_MAX_BATCH]; const bl_grid* bm[PLANNER_MAX_BATCH];
        for (int u = 0; u < L.filled; ++u) { bd[u] = L.unit[u].dist; bm[u] = L.unit[u].snap[slot]; }
        // (the transform's kernels take their grids' pointers by value: DIST_MAX_BATCH of them per launch)
        for (int u0 = 0; u0 < L.filled; u0 += DIST_MAX_BATCH) {
            const int rc = dist_set_distances_batch(L.filled - u0 < DIST_MAX_BATCH ? L.filled - u0 : DIST_MAX_BATCH, bd + u0, bm + u0);
            if (rc) return rc;
        }
    }
    bl_ctx* ctxs[PLANNER_MAX_BATCH]; bl_dist* dists[PLANNER_MAX_BATCH]; const void* starts[PLANNER_MAX_BATCH];
    bl_pose_xyt_t goals[PLANNER_MAX_BATCH]; bl_search_params_t params[PLANNER_MAX_BATCH];
    for (int u = 0; u < L.filled; ++u) {
        planner_unit& U = L.unit[u];
        ctxs[u] = U.ctx; dists[u] = U.dist; starts[u] = U.pose[slot]; goals[u] = U.goal; params[u] = U.params;
    }
    int rc = astar_launch_units(L.filled, ctxs, dists, starts, goals, params);
    if (rc) return rc;
    BL_HIP(hipEventRecord(L.slot_free[slot], L.side->stream));
    L.slot_used[slot] = true;
    L.batches += 1;
    L.filled = 0;
    if (l == p->cur_lane) p->cur_lane = (p->cur_lane + 1) % p->lanes;
    return BL_OK;
}

// First half of a submission: pick the lane, unit and snapshot slot, make the SLAM stream safe to overwrite the slot, and
// say where the snapshot goes and which number to publish when it is complete.
int bl_planner_reserve(bl_planner* p, const bl_grid* map, bl_planner_snap* out)
{
    BL_CHECK_ARG(p != nullptr && map != nullptr && out != nullptr);
    BL_CHECK_ARG(map->ctx == p->main);
    if (p->reserved) { bl_set_error("a replanner submission is already reserved"); return BL_ERR_STATE; }
    BL_HIP(hipSetDevice(p->main->device));
    {   // a batch holds grids of one size: a map of another size starts a new one
        planner_lane& C = p->lane[p->cur_lane];
        if (C.filled > 0) {
            const bl_frame& f0 = C.unit[0].dist->frame;
            if (f0.width != map->frame.width || f0.height != map->frame.height) {
                int rc = planner_launch_lane(p, p->cur_lane);
                if (rc) return rc;
            }
        }
    }
    planner_lane& L = p->lane[p->cur_lane];
    planner_unit& U = L.unit[L.filled];
    const int slot = (int)(L.batches % PLANNER_SLOTS);
    bl_grid*& snap = U.snap[slot];
    if (snap && (snap->frame.width != map->frame.width || snap->frame.height != map->frame.height)) {
        BL_HIP(hipStreamSynchronize(L.side->stream));
        bl_grid_destroy(snap);
        snap = nullptr;
    }
    if (!snap) {
        int rc = bl_grid_create(U.ctx, map->frame.width, map->frame.height, map->frame.mpc, map->frame.cpm, map->frame.ox,
                                map->frame.oy, &snap);
        if (rc) return rc;
        BL_HIP(hipStreamSynchronize(L.side->stream));          // its zero-fill ran on the side stream
    }
    snap->frame = map->frame;
    // the lane must have finished with this slot; in steady state it has, long ago -- only then is a stream wait enqueued
    if (L.slot_used[slot] && hipEventQuery(L.slot_free[slot]) != hipSuccess)
        BL_HIP(hipStreamWaitEvent(p->main->stream, L.slot_free[slot], 0));
    out->grid = snap;
    out->cells = snap->cells;
    snap->mirror_valid = false;                              // (nobody localises on a snapshot, but its cells are about to change)
    out->pose = U.pose[slot];
    out->flag = p->handoff_flag ? p->d_flag : nullptr;
    out->seq = (unsigned long long)p->submitted + 1ull;
    out->done_count = p->d_done;
    p->reserved = true;
    return BL_OK;
}

void bl_planner_cancel(bl_planner* p) { if (p) p->reserved = false; }

// Second half: the lane stream waits for the published number and runs setDistances on the snapshot; search_for_path goes
// out with the lane's batch (at once when batch == 1).
int bl_planner_commit(bl_planner* p, const bl_pose_xyt_t* goal, const bl_search_params_t* params)
{
    BL_CHECK_ARG(p != nullptr && goal != nullptr && params != nullptr);
    if (!p->reserved) { bl_set_error("bl_planner_commit without bl_planner_reserve"); return BL_ERR_STATE; }
    p->reserved = false;
    const int l = p->cur_lane;
    planner_lane& L = p->lane[l];
    planner_unit& U = L.unit[L.filled];
    const int slot = (int)(L.batches % PLANNER_SLOTS);
    // A lane that collects a batch hands over ONCE, when the batch goes out (planner_launch_lane): an event record costs
    // the SLAM stream several microseconds, and the lane cannot start the batch's searches before its last snapshot anyway.
    const bool deferred = p->batch > 1 && !p->handoff_flag;
    int rc = BL_OK;
    if (!deferred) {
        if (p->handoff_flag) {
            BL_HIP(hipStreamWaitValue64(L.side->stream, p->d_flag, (uint64_t)p->submitted + 1ull, hipStreamWaitValueGte, 0xffffffffffffffffull));
        } else {
            BL_HIP(hipEventRecord(U.snap_ready[slot], p->main->stream));
            BL_HIP(hipStreamWaitEvent(L.side->stream, U.snap_ready[slot], 0));
        }
        rc = bl_dist_set_distances(U.dist, U.snap[slot]);
        if (rc) return rc;
    }
    U.goal = *goal; U.params = *params;
    p->tickets->push_back(planner_ticket{l, L.filled});
    L.filled += 1;
    p->submitted += 1;
    if (L.filled == p->batch) return planner_launch_lane(p, l);
    return BL_OK;
}

// The same snapshot when the slot still holds an EARLIER version of this very map (a slot is reused every 2 x lanes x batch
// submissions): only the cells the map updates in between may have changed are copied -- the union of their boxes in the
// lineage's log (bl_internal.h), settled on the device; a log entry that is gone makes the same launch copy the whole grid.
// 16 MB per step become ~100 KB on a 4096 x 4096 map (k_planner_snapshot: 9.8 us of the SLAM stream per step there).
#define SNAP_INC_WGS 128
__global__ __launch_bounds__(256) void k_planner_snapshot_inc(const int8_t* __restrict__ src, int8_t* __restrict__ dst, int W, int H,
                                                              const int4* __restrict__ log, unsigned int from, unsigned int to,
                                                              const bl_pose_xyt_t* __restrict__ src_pose, bl_pose_xyt_t* __restrict__ dst_pose)
{
    __shared__ int s_box[5];
    const int lane = threadIdx.x & 63;
    if (threadIdx.x < 64) {
        int bx0 = 0x7fffffff, by0 = 0x7fffffff, bx1 = -1, by1 = -1;
        bool lost = to - from > (unsigned int)(BL_DIRTY_LOG - 32);
        if (!lost)
            for (unsigned int v = from + 1u + (unsigned int)lane; v <= to; v += 64u) {
                const int4 e = log[v % BL_DIRTY_LOG];
                if ((unsigned int)e.z != v) { lost = true; continue; }
                const int x0 = e.x & 0xffff, y0 = (int)((unsigned int)e.x >> 16), x1 = e.y & 0xffff, y1 = (int)((unsigned int)e.y >> 16);
                if (x1 < x0 || y1 < y0) continue;
                bx0 = min(bx0, x0); by0 = min(by0, y0); bx1 = max(bx1, x1); by1 = max(by1, y1);
            }
        for (int off = 32; off > 0; off >>= 1) {
            bx0 = min(bx0, __shfl_xor(bx0, off, 64)); by0 = min(by0, __shfl_xor(by0, off, 64));
            bx1 = max(bx1, __shfl_xor(bx1, off, 64)); by1 = max(by1, __shfl_xor(by1, off, 64));
        }
        lost = __builtin_amdgcn_ballot_w64(lost) != 0ull;
        if (lane == 0) {
            if (lost) { bx0 = 0; by0 = 0; bx1 = W - 1; by1 = H - 1; }
            s_box[0] = bx0 & ~15; s_box[1] = by0; s_box[2] = min(bx1 | 15, W - 1); s_box[3] = by1; s_box[4] = bx1 >= bx0 ? 1 : 0;
        }
    }
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0) *dst_pose = *src_pose;
    if (!s_box[4]) return;
    const int x0 = s_box[0], y0 = s_box[1], x1 = s_box[2], y1 = s_box[3];
    const int q = (x1 - x0 + 1) >> 4;                              // 16-byte pieces per row (W is a multiple of 16)
    const long long total = (long long)q * (y1 - y0 + 1);
    for (long long base = (long long)blockIdx.x * blockDim.x + threadIdx.x; base < total; base += 4ll * gridDim.x * blockDim.x) {
        int4 v[4];
        size_t at[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long i = base + (long long)u * gridDim.x * blockDim.x;
            at[u] = 0;
            if (i < total) { const int r = (int)(i / q), c = (int)(i - (long long)r * q); at[u] = (size_t)(y0 + r) * W + x0 + 16 * c; v[u] = *(const int4*)(src + at[u]); }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long i = base + (long long)u * gridDim.x * blockDim.x;
            if (i < total) *(int4*)(dst + at[u]) = v[u];
        }
    }
}

// Map + pose snapshot on main's stream into a grid of the same size (the whole grid, or -- when `snap` still holds an earlier
// version of this very map -- the cells the updates in between may have changed); the snapshot then carries the map's lineage.
// flag / done_count / seq: the replanner's flag hand-over (null / 0: none).
int bl_snapshot_enqueue(bl_ctx* main, const bl_grid* map, bl_grid* snap, const void* d_pose, bl_pose_xyt_t* snap_pose,
                        unsigned int* done_count, unsigned long long* flag, unsigned long long seq)
{
    const size_t n = (size_t)map->frame.width * map->frame.height;
    hipEvent_t f0, f1;
    int rc = bl_timer_begin(main, BL_K_SNAPSHOT, &f0, &f1);
    if (rc) return rc;
    static const bool no_inc = getenv("BOTLAB_SNAPSHOT_NO_INCREMENTAL") != nullptr;
    const bl_grid* old = snap;
    const bool inc = !no_inc && !flag && old->id != 0 && old->id == map->id && !map->mirror_external && map->log != nullptr &&
                     old->version <= map->version && map->version - old->version <= (uint64_t)(BL_DIRTY_LOG - 64) &&
                     (map->frame.width & 15) == 0 && n >= ((size_t)1 << 20);
    if (inc) {
        hipLaunchKernelGGL(k_planner_snapshot_inc, dim3(SNAP_INC_WGS), dim3(256), 0, main->stream, map->cells, snap->cells, map->frame.width,
                           map->frame.height, map->log->dev, (unsigned int)old->version, (unsigned int)map->version,
                           (const bl_pose_xyt_t*)d_pose, snap_pose);
    } else {
        int blocks = (int)((n / 16 + 255) / 256);
        if (blocks < 1) blocks = 1;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(k_planner_snapshot, dim3(blocks), dim3(256), 0, main->stream, map->cells, snap->cells, n,
                           (const bl_pose_xyt_t*)d_pose, snap_pose, done_count, flag, seq);
    }
    BL_HIP(hipGetLastError());
    rc = bl_timer_end(main, BL_K_SNAPSHOT, f0, f1);
    if (rc) return rc;
    bl_grid_adopt_lineage(snap, map);                  // the snapshot holds this version of the map
    return BL_OK;
}

extern "C" int bl_planner_submit(bl_planner* p, const bl_grid* map, const void* d_start_pose, const bl_pose_xyt_t* goal,
                                 const bl_search_params_t* params)
{
    BL_CHECK_ARG(p != nullptr && map != nullptr && d_start_pose != nullptr && goal != nullptr && params != nullptr);
    bl_planner_snap sn;
    int rc = bl_planner_reserve(p, map, &sn);
    if (rc) return rc;
    rc = bl_snapshot_enqueue(p->main, map, sn.grid, d_start_pose, sn.pose, sn.done_count, sn.flag, sn.seq);
    if (rc) return rc;
    return bl_planner_commit(p, goal, params);
}

extern "C" int bl_planner_fetch(bl_planner* p, bl_pose_xyt_t* out_path, int cap, int* out_len, int64_t* stats)
{
    BL_CHECK_ARG(p != nullptr);
    if (p->fetched == p->submitted) { bl_set_error("no replan pending"); return BL_ERR_STATE; }
    const planner_ticket t = p->tickets->front();              // results come back in submission order
    p->tickets->pop_front();
    planner_lane& L = p->lane[t.lane];
    // asked for before its batch is complete (the caller keeps fewer submissions in flight than a batch holds, or is
    // draining): the batch goes out with what it has.  A lane's earlier batches are launched, so a ticket whose ctx has
    // nothing pending can only belong to the batch being collected.
    if (L.filled > 0 && t.unit < L.filled) {
        bl_astar_state* st = L.unit[t.unit].ctx->astar;
        if (!st || st->launched == st->fetched) {
            int rc = planner_launch_lane(p, t.lane);
            if (rc) return rc;
        }
    }
    int rc = bl_astar_search_result(L.unit[t.unit].ctx, out_path, cap, out_len, stats);
    p->fetched += 1;
    return rc;
}

// End of input: no further submission will come soon, so the batch every lane is collecting goes out as it is -- its distance grids
// and searches then run beside the SLAM stream's remaining steps instead of behind the fetch that would have sent it off.
extern "C" int bl_planner_flush(bl_planner* p)
{
    BL_CHECK_ARG(p != nullptr);
    for (int l = 0; l < p->lanes; ++l) {
        if (p->lane[l].filled == 0) continue;
        bl_astar_state* st = p->lane[l].unit[0].ctx->astar;
        if (st && st->launched != st->fetched) continue;       // (an earlier batch of the lane still holds its units' result slots: the fetch will send this one)
        const int rc = planner_launch_lane(p, l);
        if (rc) return rc;
    }
    return BL_OK;
}

extern "C" int bl_planner_timing(bl_planner* p, int on, double* dist_ms, double* astar_ms, int64_t* launches)
{
    BL_CHECK_ARG(p != nullptr);
    double d = 0, a = 0; int64_t n = 0;
    for (int l = 0; l < p->lanes; ++l)
    for (int u = 0; u < p->batch; ++u) {
        bl_ctx* c = p->lane[l].unit[u].ctx;
        if (on >= 0) { int rc = bl_ctx_timing_enable(c, on); if (rc) return rc; if (on) { rc = bl_ctx_timing_reset(c); if (rc) return rc; } }
        double dl = 0, al = 0; int64_t nl = 0;
        int rc = bl_ctx_timing_get(c, BL_K_DIST, &dl, &nl); if (rc) return rc;
        rc = bl_ctx_timing_get(c, BL_K_ASTAR, &al, &nl); if (rc) return rc;
        d += dl; a += al; n += nl;
    }
    if (dist_ms) *dist_ms = d;
    if (astar_ms) *astar_ms = a;
    if (launches) *launches = n;
    return BL_OK;
}

// =============================================================================================== plan_path_to_frontier
// Host logic of frontiers.cpp:87-214 over the batched search: the candidate goals of one ring of the expanding square
// are checked together -- isValidGoal (one gather of the goal cells' distances), planPath (one batch of searches, one
// per distinct goal cell: the path does not depend on where inside the cell the goal lies), isPathSafe (one gather of
// every path pose's cell) -- and the reference's "last valid candidate of the sweep wins" rule is then applied in order.
namespace {
struct frontier_search { bool done = false; std::vector<bl_pose_xyt_t> path; };

// MotionPlanner::isValidGoal (motion_planner.cpp:52-74) given distances_(goalCell) (NaN when outside the grid)
inline bool mp_is_valid_goal(const bl_motion_planner_t& pl, float gx, float gy, float dist_at_cell)
{
    float dx = gx - pl.prev_goal.x, dy = gy - pl.prev_goal.y;
    float distanceFromPrev = std::sqrt(dx * dx + dy * dy);
    if (pl.num_frontiers != 1 && distanceFromPrev < 2 * pl.search.minDistanceToObstacle) return false;
    if (dist_at_cell != dist_at_cell) return false;                     // not in the grid
    return dist_at_cell > pl.robot_radius;
}
}  // namespace

extern "C" int bl_plan_path_to_frontier(bl_ctx* ctx, const bl_frontiers* frontiers, const bl_pose_xyt_t* robot_pose, bl_dist* dist,
                                        const bl_motion_planner_t* planner, bl_pose_xyt_t* out_path, int cap, int* out_len,
                                        bl_pose_xyt_t* chosen_goal, int64_t* stats)
{
    BL_CHECK_ARG(ctx != nullptr && frontiers != nullptr && robot_pose != nullptr && dist != nullptr && planner != nullptr);
    BL_CHECK_ARG(out_path != nullptr && cap >= 1 && out_len != nullptr && dist->valid && dist->ctx == ctx);
    const bl_motion_planner_t& pl = *planner;
    const bl_pose_xyt_t robotPose = *robot_pose;
    const bl_frame frame = dist->frame;
    int64_t pops = 0, pushes = 0, searches = 0;
    if (stats) stats[0] = stats[1] = stats[2] = 0;
    if (chosen_goal) memset(chosen_goal, 0, sizeof(*chosen_goal));
    *out_len = 0;
    const int nfr = (int)frontiers->offsets.size() - 1;
    if (nfr <= 0) return BL_OK;                                         // emptyPath (:118-120)
    // ---- closest frontier by squared distance to any of its cells, first minimum wins (:122-138); then its middle cell (:140)
    float min_dist = (float)99999999999999LL;                           // float min_dist = 99999999999999 (:122)
    int closest = -1;
    for (int k = 0; k < nfr; ++k)
        for (int i = frontiers->offsets[k]; i < frontiers->offsets[k + 1]; ++i) {
            const float px = frontiers->xy[2 * (size_t)i], py = frontiers->xy[2 * (size_t)i + 1];
            const float distance_sq = (robotPose.x - px) * (robotPose.x - px) + (robotPose.y - py) * (robotPose.y - py);
            if (distance_sq < min_dist) { closest = k; min_dist = distance_sq; }
        }
    out_path[0] = robotPose;
    if (closest < 0) { *out_len = 1; return BL_OK; }                    // no cell beat min_dist (reference: out-of-bounds read)
    const int nc = frontiers->offsets[closest + 1] - frontiers->offsets[closest];
    const int mid = frontiers->offsets[closest] + (int)((size_t)(nc - 1) / 2);
    const float cpx = frontiers->xy[2 * (size_t)mid], cpy = frontiers->xy[2 * (size_t)mid + 1];

    std::map<long long, frontier_search> cache;                         // goal cell -> planPath result
    struct cand { float x, y; int cx, cy; bool valid; int ring; };
    std::vector<cand> cands;
    std::vector<int32_t> q;
    std::vector<float> qv;

    // check_valid (:87-102), first half, for the candidates cands[from..): isValidGoal by one gather of the goal cells' distances.
    // Returns the number of goal cells among them that no search has been run for yet (they are entered into `todo`).
    std::vector<int2> todo;
    std::vector<long long> todo_key;
    auto validity = [&](size_t from, int* fresh) -> int {
        const int n = (int)(cands.size() - from);
        *fresh = 0;
        if (n == 0) return BL_OK;
        q.resize(2 * (size_t)n); qv.resize((size_t)n);
        for (int i = 0; i < n; ++i) {
            cand& c = cands[from + (size_t)i];
            bl_global_to_cell((double)c.x, (double)c.y, frame, &c.cx, &c.cy);    // motion_planner.cpp:61
            q[2 * (size_t)i] = c.cx; q[2 * (size_t)i + 1] = c.cy;
        }
        int rc = bl_dist_gather(dist, q.data(), n, qv.data());
        if (rc) return rc;
        for (int i = 0; i < n; ++i) {
            cand& c = cands[from + (size_t)i];
            c.valid = mp_is_valid_goal(pl, c.x, c.y, qv[i]);
            if (!c.valid) continue;
            const long long key = ((long long)c.cy << 32) | (unsigned int)c.cx;
            if (cache.find(key) == cache.end()) { cache[key] = frontier_search(); todo.push_back(make_int2(c.cx, c.cy)); todo_key.push_back(key); *fresh += 1; }
        }
        return BL_OK;
    };
    // ... second half, for every candidate gathered so far: planPath (one batch of searches, one per goal cell not searched yet)
    // and isPathSafe (one gather over all path poses)
    auto evaluate = [&]() -> int {
        const int n = (int)cands.size();
        int rc = BL_OK;
        if (!todo.empty()) {
            static const bool trace = getenv("BOTLAB_PLAN_TRACE") != nullptr;
            const auto w0 = std::chrono::steady_clock::now();
            long long max_pops = 0, sum_pops = 0; int n_found = 0;
            rc = astar_batch_cells(ctx, dist, &robotPose, todo.data(), (int)todo.size(), &pl.search,
                                   [&](int i, bool found, const int32_t* pc, int len, long long po, long long pu) {
                                       if (po > max_pops) max_pops = po;
                                       sum_pops += po; n_found += found ? 1 : 0;
                                       frontier_search& fs = cache[todo_key[i]];
                                       fs.path.resize((size_t)1 + len);
                                       astar_cells_to_path(frame, robotPose, pc, len, fs.path.data(), 1 + len);
                                       fs.done = true;
                                       pops += po; pushes += pu; searches += 1;
                                   });
            if (rc) return rc;
            if (trace)
                fprintf(stderr, "[plan] batch of %d searches: %d found, %lld pops in all, longest %lld, %.1f ms\n", (int)todo.size(), n_found, sum_pops,
                        max_pops, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count());
            todo.clear(); todo_key.clear();
        }
        // isPathSafe (motion_planner.cpp:77-96) of every candidate path with >= 3 poses: one gather for all poses
        q.clear();
        std::vector<int> first((size_t)n, -1);
        for (int i = 0; i < n; ++i) {
            if (!cands[i].valid) continue;
            const frontier_search& fs = cache[((long long)cands[i].cy << 32) | (unsigned int)cands[i].cx];
            if (fs.path.size() < 3) { cands[i].valid = false; continue; }     // temp_path.path_length < 3 (:95)
            first[i] = (int)(q.size() / 2);
            for (const bl_pose_xyt_t& p : fs.path) {
                const int x = p.x / frame.mpc + frame.width / 2;              // float / float + int -> float -> int (:83-84)
                const int y = p.y / frame.mpc + frame.height / 2;
                q.push_back(x); q.push_back(y);
            }
        }
        qv.resize(q.size() / 2);
        rc = bl_dist_gather(dist, q.data(), (int)(q.size() / 2), qv.data());
        if (rc) return rc;
        for (int i = 0; i < n; ++i) {
            if (first[i] < 0) continue;
            const size_t np = cache[((long long)cands[i].cy << 32) | (unsigned int)cands[i].cx].path.size();
            for (size_t k = 0; k < np; ++k) {
                const float dv = qv[(size_t)first[i] + k];
                if (dv != dv || dv <= pl.search.minDistanceToObstacle) { cands[i].valid = false; break; }   // D9: outside the grid is unsafe
            }
        }
        return BL_OK;
    };

    // The sweep of :143-199 takes one ring of the expanding square after the other and stops at the first ring that holds a valid
    // candidate.  The rings are independent of each other, so several CAN be evaluated ahead of the rule (BOTLAB_FRONTIER_SPECULATE=n:
    // rings are gathered until they ask for n goal cells not searched yet, their searches run side by side as one batch, and the
    // rule is then applied ring by ring, in the reference's order, to the same outcome).  Off by default (n = 1: one ring per
    // batch, the reference's own sequence), because it does not pay: a plan's searches already arrive as ONE batch (the first ring
    // with goals of enough clearance), whose length is its longest search (5.4e5 pops, 1.4 s on the cut arena), while the rings
    // behind hold goals that are expensive or unreachable (68 searches, 1.0e7 pops, 8.7 s with n = 64 there).
    bool foundPose = false;
    float square_radius = .025;
    float sq_len = .025;
    bl_pose_xyt_t goal_pose; memset(&goal_pose, 0, sizeof(goal_pose));    // D1
    int wraps = 0;
    static const int speculate = getenv("BOTLAB_FRONTIER_SPECULATE") ? atoi(getenv("BOTLAB_FRONTIER_SPECULATE")) : 1;
    while (!foundPose) {
        // ---- gather rings ahead (the radius sequence of :194-197, the D8 cut after the second wrap included)
        cands.clear();
        struct ring_info { size_t begin, end; bool wraps_here; };
        std::vector<ring_info> rings;
        float r_spec = square_radius;
        int wraps_spec = wraps, pending = (int)todo.size();
        while (true) {
            const float top_height = cpy + r_spec;
            const float bot_height = cpy - r_spec;
            const float left_bound = cpy + r_spec;          // sic: built from the y coordinate (:176-177)
            const float right_bound = cpy - r_spec;
            const size_t begin = cands.size();
            const int ring = (int)rings.size();
            for (float i = -r_spec; i <= r_spec; i += sq_len) {
                cands.push_back(cand{cpx + i, top_height, 0, 0, false, ring});
                cands.push_back(cand{cpx + i, bot_height, 0, 0, false, ring});
            }
            for (float i = -r_spec; i <= r_spec; i += sq_len) {
                cands.push_back(cand{right_bound, cpy + i, 0, 0, false, ring});
                cands.push_back(cand{left_bound, cpy + i, 0, 0, false, ring});
            }
            int fresh = 0;
            int rc = validity(begin, &fresh);
            if (rc) return rc;
            pending += fresh;
            bool wrap = false, last = false;
            if (r_spec < 0.5) r_spec += sq_len;
            else { r_spec = 0.05; wrap = true; if (++wraps_spec == 2) last = true; }     // (D8: the sweep ends there if nothing was found)
            rings.push_back(ring_info{begin, cands.size(), wrap});
            if (last || pending >= speculate || speculate <= 1) break;
        }
        int rc = evaluate();
        if (rc) return rc;
        // ---- the rule, ring by ring: first of each pair if valid, else the second (:153-193); then the radius update (:194-197)
        for (const ring_info& rg : rings) {
            for (size_t i = rg.begin; i + 1 < rg.end; i += 2) {
                if (cands[i].valid) { foundPose = true; goal_pose.x = cands[i].x; goal_pose.y = cands[i].y; }
                else if (cands[i + 1].valid) { foundPose = true; goal_pose.x = cands[i + 1].x; goal_pose.y = cands[i + 1].y; }
            }
            if (square_radius < 0.5) square_radius += sq_len;
            else {
                square_radius = 0.05;
                if (++wraps == 2 && !foundPose) {                                    // D8
                    if (stats) { stats[0] = pops; stats[1] = pushes; stats[2] = searches; }
                    *out_len = 1;
                    return BL_OK;
                }
            }
            if (foundPose) break;
        }
    }
    goal_pose.theta = robotPose.theta;                                           // :209
    if (chosen_goal) *chosen_goal = goal_pose;
    // planner.planPath(robotPose, goal_pose) (:210): the chosen candidate's search has already run
    int gcx, gcy;
    bl_global_to_cell((double)goal_pose.x, (double)goal_pose.y, frame, &gcx, &gcy);
    const frontier_search& fs = cache[((long long)gcy << 32) | (unsigned int)gcx];
    const int n = (int)fs.path.size();
    for (int i = 0; i < n && i < cap; ++i) out_path[i] = fs.path[i];
    *out_len = n;
    if (stats) { stats[0] = pops; stats[1] = pushes; stats[2] = searches; }
    return BL_OK;
}
