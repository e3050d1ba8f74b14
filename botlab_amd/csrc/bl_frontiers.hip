// bl_frontiers.hip -- find_map_frontiers (src/planning/frontiers.cpp:25-85, 217-288) on the device-resident map.
//
// The reference is two nested FIFO breadth-first searches sharing one visited set: a 4-connected flood of free space from
// the robot cell and, whenever that flood touches an unvisited frontier cell, an 8-connected flood of the frontier it
// belongs to.  Its OUTPUT ORDER is part of the contract (plan_path_to_frontier breaks distance ties by it and picks the
// middle cell of the chosen frontier), so the kernel reproduces the exact queue order, level by level:
//   * a level of a FIFO BFS is the concatenation, over the previous level in queue order, of each cell's newly visited
//     neighbours in neighbour order.  Every (queue position p, neighbour n) pair gets the key 4p+n (8q+n in a frontier);
//     all lanes claim their neighbours with atomicMin(key) -- the smallest key is the serial code's first visit -- and
//     the winners are appended by a stable compaction in key order.
//   * frontier cells are never entered by the free-space flood; the first (smallest-key) touch of a frontier component
//     is where the serial code calls grow_frontier, and that touched cell is its seed.  A second sweep over the finished
//     queue visits the touches in key order and grows every component not grown yet from its seed.
// The depth of the computation is the number of BFS levels (inherent to the FIFO order); each level is a few barriers
// of one 1024-thread workgroup, with all per-cell state in L2-resident arrays.
#include <string.h>

#include "bl_internal.h"

#define FR_T 1024
#define FR_INF 0xFFFFFFFFu

struct frontier_args {
    const int8_t* cells; int W, H;
    int rx, ry;                 // robot cell (may lie outside the grid)
    uint8_t* cls;               // 0 other, 1 free (flooded), 2 frontier cell, 3 robot cell
    unsigned int* claim;        // free cells: winning claim key; frontier cells: first touch key
    unsigned int* fclaim;       // frontier growth claim key
    int32_t* queue;             // free-space queue, position 0 = the robot cell (coordinates in rx, ry)
    int32_t* out_cells;         // frontier cells, frontier after frontier, each in growth-queue order
    int32_t* out_offsets; int cap_frontiers;
    int32_t* counts;            // [0] frontiers, [1] frontier cells, [2] free cells reached (+1), [3] levels, [4] overflow flag
};

__device__ __forceinline__ unsigned int ld_claim(const unsigned int* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // atomics resolve at L2: read there
}

// exclusive scan of a small per-thread count over the workgroup; returns the offset, *total = sum
__device__ __forceinline__ int block_excl_scan(int v, int* s_wave, int* total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = v;
    for (int off = 1; off < 64; off <<= 1) {
        int t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    __syncthreads();
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int base = 0, tot = 0;
    for (int w = 0; w < FR_T / 64; ++w) { int x = s_wave[w]; if (w < wave) base += x; tot += x; }
    *total = tot;
    return base + incl - v;
}

__device__ __forceinline__ unsigned int block_min(unsigned int v, unsigned int* s_wave)
{
    for (int off = 32; off > 0; off >>= 1) { unsigned int t = __shfl_xor(v, off, 64); v = t < v ? t : v; }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = v;
    __syncthreads();
    unsigned int m = FR_INF;
    for (int w = 0; w < FR_T / 64; ++w) { unsigned int x = s_wave[w]; m = x < m ? x : m; }
    return m;
}

// OccupancyGrid::logOdds: 0 outside the grid (occupancy_grid.cpp:63-71)
__device__ __forceinline__ int fr_log_odds(const frontier_args& a, int x, int y)
{
    return (x >= 0 && y >= 0 && x < a.W && y < a.H) ? (int)a.cells[(size_t)y * a.W + x] : 0;
}

// CLS_LDS: the class bytes of the whole grid live in LDS (grids up to FR_CLS_LDS cells); otherwise in the global scratch.
// All communication is inside ONE workgroup (one CU, one vector L1): __syncthreads() orders plain stores/loads and makes
// the claim atomics (performed at L2) complete; claim words are read back with L2-scope loads only.
#define FR_CLS_LDS (96 * 1024)
#define FR_LQ 2048                // next-level queue entries mirrored in LDS (wider levels are re-read from the global queue)

template <bool CLS_LDS>
__global__ __launch_bounds__(FR_T) void k_frontiers(frontier_args a)
{
    extern __shared__ uint8_t s_cls[];
    __shared__ int s_wave[FR_T / 64];
    __shared__ unsigned int s_umin[FR_T / 64];
    __shared__ int s_q[2][FR_LQ];
    const int tid = threadIdx.x;
    const long long ncell = (long long)a.W * a.H;
    uint8_t* cls = CLS_LDS ? s_cls : a.cls;
    // ---- classification: is_frontier_cell (frontiers.cpp:217-246) / free (:77)
    for (long long c = tid; c < ncell; c += FR_T) {
        const int x = (int)(c % a.W), y = (int)(c / a.W);
        const int v = a.cells[c];
        int k = 0;
        if (!(v > 0 || v < -5)) {                          // (map(x,y) > .1 || map(x,y) < -5) on an int8
            if (fr_log_odds(a, x - 1, y) < 0 || fr_log_odds(a, x + 1, y) < 0 || fr_log_odds(a, x, y + 1) < 0 || fr_log_odds(a, x, y - 1) < 0) k = 2;
        }
        if (k == 0 && v < 0) k = 1;
        if (x == a.rx && y == a.ry) k = 3;                  // visitedCells.insert(robotCell) before anything else (:42)
        cls[c] = (uint8_t)k;
        a.claim[c] = FR_INF;
        a.fclaim[c] = FR_INF;
    }
    __threadfence();
    __syncthreads();
    // ---- free-space flood (:47-82), xDeltas {-1,1,0,0}, yDeltas {0,0,1,-1}
    int lo = 0, hi = 1, levels = 0, cur = 0;
    while (lo < hi) {
        const bool one_pass = hi - lo <= FR_T;              // the common case: this thread's neighbours stay in registers
        const bool from_lds = hi - lo <= FR_LQ && lo > 0;
        int rc[4], rn = 0; unsigned int rk[4];
        for (int p = lo + tid; p < hi; p += FR_T) {
            int x, y;
            if (p == 0) { x = a.rx; y = a.ry; } else { const int c = from_lds ? s_q[cur][p - lo] : a.queue[p]; x = c % a.W; y = c / a.W; }
            rn = 0;
            for (int n = 0; n < 4; ++n) {
                const int nx = x + (n == 0 ? -1 : (n == 1 ? 1 : 0)), ny = y + (n == 2 ? 1 : (n == 3 ? -1 : 0));
                if (nx < 0 || ny < 0 || nx >= a.W || ny >= a.H) continue;
                const int c = ny * a.W + nx;
                const int k = cls[c];
                if (k == 1 || k == 2) {
                    const unsigned int key = ((unsigned int)p << 2) | (unsigned int)n;
                    atomicMin(&a.claim[c], key);
                    if (k == 1) { rc[rn] = c; rk[rn] = key; rn++; }
                }
            }
        }
        __syncthreads();
        int newhi = hi;
        for (int base = lo; base < hi; base += FR_T) {
            const int p = base + tid;
            int wins = 0, wc[4];
            if (p < hi) {
                if (one_pass) {
                    for (int j = 0; j < rn; ++j)
                        if (ld_claim(&a.claim[rc[j]]) == rk[j]) wc[wins++] = rc[j];
                } else {
                    const int c0 = a.queue[p];
                    const int x = c0 % a.W, y = c0 / a.W;
                    for (int n = 0; n < 4; ++n) {
                        const int nx = x + (n == 0 ? -1 : (n == 1 ? 1 : 0)), ny = y + (n == 2 ? 1 : (n == 3 ? -1 : 0));
                        if (nx < 0 || ny < 0 || nx >= a.W || ny >= a.H) continue;
                        const int c = ny * a.W + nx;
                        if (cls[c] == 1 && ld_claim(&a.claim[c]) == (((unsigned int)p << 2) | (unsigned int)n)) wc[wins++] = c;
                    }
                }
            }
            int total;
            const int off = block_excl_scan(wins, s_wave, &total);
            for (int j = 0; j < wins; ++j) {
                const int at = newhi + off + j;
                a.queue[at] = wc[j];
                cls[wc[j]] = 4;                             // visited: never claimed again (its claim key stays the smallest anyway)
                if (at - hi < FR_LQ) s_q[cur ^ 1][at - hi] = wc[j];
            }
            newhi += total;
        }
        __syncthreads();
        lo = hi; hi = newhi; levels += 1; cur ^= 1;
    }
    const int qn = hi;
    // ---- frontiers in discovery order (:66-75): touches in key order; a touched frontier cell that is not part of a grown
    // frontier yet is the seed of the next one (grow_frontier, :249-288; xDeltas {-1,-1,-1,1,1,1,0,0}, yDeltas {0,1,-1,0,1,-1,1,-1})
    int nf = 0, total_cells = 0, overflow = 0;
    for (int base = 0; base < qn; base += FR_T) {
        const int p = base + tid;
        int cand = 0, cc[4]; unsigned int ck[4];
        if (p < qn) {
            int x, y;
            if (p == 0) { x = a.rx; y = a.ry; } else { const int c = a.queue[p]; x = c % a.W; y = c / a.W; }
            for (int n = 0; n < 4; ++n) {
                const int nx = x + (n == 0 ? -1 : (n == 1 ? 1 : 0)), ny = y + (n == 2 ? 1 : (n == 3 ? -1 : 0));
                if (nx < 0 || ny < 0 || nx >= a.W || ny >= a.H) continue;
                const int c = ny * a.W + nx;
                const unsigned int key = ((unsigned int)p << 2) | (unsigned int)n;
                if (cls[c] == 2 && ld_claim(&a.claim[c]) == key) { cc[cand] = c; ck[cand] = key; cand++; }
            }
        }
        while (true) {
            unsigned int mine = FR_INF;
            for (int j = 0; j < cand; ++j)
                if (ck[j] < mine && cls[cc[j]] == 2) mine = ck[j];          // class 5 = already part of a grown frontier
            const unsigned int best = block_min(mine, s_umin);
            if (best == FR_INF) break;
            // the seed is the neighbour (best & 3) of queue position (best >> 2)
            int seed;
            {
                const int sp = (int)(best >> 2), n = (int)(best & 3u);
                int x, y;
                if (sp == 0) { x = a.rx; y = a.ry; } else { const int c = a.queue[sp]; x = c % a.W; y = c / a.W; }
                seed = (y + (n == 2 ? 1 : (n == 3 ? -1 : 0))) * a.W + x + (n == 0 ? -1 : (n == 1 ? 1 : 0));
            }
            int32_t* fq = a.out_cells + total_cells;
            if (tid == 0) { fq[0] = seed; cls[seed] = 5; s_q[0][0] = seed; }
            __syncthreads();
            int flo = 0, fhi = 1, fcur = 0;
            while (flo < fhi) {
                const bool one_pass = fhi - flo <= FR_T;
                const bool from_lds = fhi - flo <= FR_LQ;
                int rc[8], rn = 0; unsigned int rk[8];
                for (int q = flo + tid; q < fhi; q += FR_T) {
                    const int c = from_lds ? s_q[fcur][q - flo] : fq[q];
                    const int x = c % a.W, y = c / a.W;
                    rn = 0;
                    for (int n = 0; n < 8; ++n) {
                        const int nx = x + (n < 3 ? -1 : (n < 6 ? 1 : 0));
                        const int ny = y + ((n == 1 || n == 4 || n == 6) ? 1 : ((n == 2 || n == 5 || n == 7) ? -1 : 0));
                        if (nx < 0 || ny < 0 || nx >= a.W || ny >= a.H) continue;
                        const int nc = ny * a.W + nx;
                        if (cls[nc] == 2) {
                            const unsigned int key = ((unsigned int)q << 3) | (unsigned int)n;
                            atomicMin(&a.fclaim[nc], key);
                            rc[rn] = nc; rk[rn] = key; rn++;
                        }
                    }
                }
                __syncthreads();
                int fnew = fhi;
                for (int fb = flo; fb < fhi; fb += FR_T) {
                    const int q = fb + tid;
                    int wins = 0, wc[8];
                    if (q < fhi) {
                        if (one_pass) {
                            for (int j = 0; j < rn; ++j)
                                if (ld_claim(&a.fclaim[rc[j]]) == rk[j]) wc[wins++] = rc[j];
                        } else {
                            const int c = fq[q];
                            const int x = c % a.W, y = c / a.W;
                            for (int n = 0; n < 8; ++n) {
                                const int nx = x + (n < 3 ? -1 : (n < 6 ? 1 : 0));
                                const int ny = y + ((n == 1 || n == 4 || n == 6) ? 1 : ((n == 2 || n == 5 || n == 7) ? -1 : 0));
                                if (nx < 0 || ny < 0 || nx >= a.W || ny >= a.H) continue;
                                const int nc = ny * a.W + nx;
                                if (cls[nc] == 2 && ld_claim(&a.fclaim[nc]) == (((unsigned int)q << 3) | (unsigned int)n)) wc[wins++] = nc;
                            }
                        }
                    }
                    int total;
                    const int off = block_excl_scan(wins, s_wave, &total);
                    for (int j = 0; j < wins; ++j) {
                        const int at = fnew + off + j;
                        fq[at] = wc[j];
                        if (at - fhi < FR_LQ) s_q[fcur ^ 1][at - fhi] = wc[j];
                    }
                    fnew += total;
                }
                __syncthreads();
                // winners leave class 2 only now: a cell claimed in this level must still look unvisited to every
                // claimer of the level (the smallest key wins), but visited to the next level
                for (int q = fhi + tid; q < fnew; q += FR_T) cls[fq[q]] = 5;
                __syncthreads();
                flo = fhi; fhi = fnew; fcur ^= 1;
            }
            if (nf < a.cap_frontiers) { if (tid == 0) a.out_offsets[nf] = total_cells; } else overflow = 1;
            nf += 1;
            total_cells += fhi;
        }
    }
    if (tid == 0) {
        if (nf <= a.cap_frontiers) a.out_offsets[nf] = total_cells;
        a.counts[0] = nf; a.counts[1] = total_cells; a.counts[2] = qn; a.counts[3] = levels; a.counts[4] = overflow;
    }
}

struct bl_frontier_scratch {
    size_t cells = 0;
    uint8_t* cls = nullptr; unsigned int* claim = nullptr; unsigned int* fclaim = nullptr;
    int32_t* queue = nullptr; int32_t* out_cells = nullptr; int32_t* out_offsets = nullptr; int32_t* counts = nullptr;
    int cap_frontiers = 0;
    int32_t* h_counts = nullptr;
};

void bl_frontier_scratch_free(bl_ctx* ctx)
{
    bl_frontier_scratch* s = ctx->frontier;
    if (!s) return;
    void* dev[] = {s->cls, s->claim, s->fclaim, s->queue, s->out_cells, s->out_offsets, s->counts};
    for (void* q : dev) if (q) (void)hipFree(q);
    if (s->h_counts) (void)hipHostFree(s->h_counts);
    delete s;
    ctx->frontier = nullptr;
}

extern "C" int bl_frontiers_find(bl_ctx* ctx, const bl_grid* map, const bl_pose_xyt_t* robot_pose, double min_frontier_length,
                                 bl_frontiers** out)
{
    BL_CHECK_ARG(ctx != nullptr && map != nullptr && robot_pose != nullptr && out != nullptr);
    BL_CHECK_ARG(map->ctx == ctx);
    const int W = map->frame.width, H = map->frame.height;
    BL_CHECK_ARG(W >= 1 && H >= 1 && (int64_t)W * H < ((int64_t)1 << 28));          // claim keys are 4 * queue position + n
    BL_HIP(hipSetDevice(ctx->device));
    if (!ctx->frontier) ctx->frontier = new bl_frontier_scratch();
    bl_frontier_scratch* s = ctx->frontier;
    const size_t n = (size_t)W * H;
    if (s->cells < n) {
        BL_HIP(hipStreamSynchronize(ctx->stream));
        void* dev[] = {s->cls, s->claim, s->fclaim, s->queue, s->out_cells, s->out_offsets};
        for (void* q : dev) if (q) BL_HIP(hipFree(q));
        s->cls = nullptr; s->claim = nullptr; s->fclaim = nullptr; s->queue = nullptr; s->out_cells = nullptr; s->out_offsets = nullptr;
        s->cells = 0;
        s->cap_frontiers = (int)(n / 4 + 4);               // components are 8-separated: at most one per 2x2 block
        BL_HIP(hipMalloc((void**)&s->cls, n));
        BL_HIP(hipMalloc((void**)&s->claim, n * 4));
        BL_HIP(hipMalloc((void**)&s->fclaim, n * 4));
        BL_HIP(hipMalloc((void**)&s->queue, (n + 1) * 4));
        BL_HIP(hipMalloc((void**)&s->out_cells, n * 4));
        BL_HIP(hipMalloc((void**)&s->out_offsets, ((size_t)s->cap_frontiers + 1) * 4));
        if (!s->counts) BL_HIP(hipMalloc((void**)&s->counts, 8 * 4));
        if (!s->h_counts) BL_HIP(hipHostMalloc((void**)&s->h_counts, 8 * 4, hipHostMallocDefault));
        s->cells = n;
    }
    frontier_args a;
    a.cells = map->cells; a.W = W; a.H = H;
    bl_global_to_cell((double)robot_pose->x, (double)robot_pose->y, map->frame, &a.rx, &a.ry);      // :39
    a.cls = s->cls; a.claim = s->claim; a.fclaim = s->fclaim; a.queue = s->queue;
    a.out_cells = s->out_cells; a.out_offsets = s->out_offsets; a.cap_frontiers = s->cap_frontiers; a.counts = s->counts;
    hipEvent_t e0, e1;
    int rc = bl_timer_begin(ctx, BL_K_FRONTIERS, &e0, &e1);
    if (rc) return rc;
    if (n <= (size_t)FR_CLS_LDS) {
        static bool attr_set = false;
        if (!attr_set) {
            BL_HIP(hipFuncSetAttribute((const void*)k_frontiers<true>, hipFuncAttributeMaxDynamicSharedMemorySize, FR_CLS_LDS));
            attr_set = true;
        }
        hipLaunchKernelGGL(k_frontiers<true>, dim3(1), dim3(FR_T), (n + 15) & ~(size_t)15, ctx->stream, a);
    } else {
        hipLaunchKernelGGL(k_frontiers<false>, dim3(1), dim3(FR_T), 0, ctx->stream, a);
    }
    BL_HIP(hipGetLastError());
    rc = bl_timer_end(ctx, BL_K_FRONTIERS, e0, e1);
    if (rc) return rc;
    BL_HIP(hipMemcpyAsync(s->h_counts, s->counts, 8 * 4, hipMemcpyDeviceToHost, ctx->stream));
    BL_HIP(hipStreamSynchronize(ctx->stream));
    const int nf = s->h_counts[0], total = s->h_counts[1];
    if (s->h_counts[4] || nf > s->cap_frontiers) { bl_set_error("internal: frontier table overflow (%d frontiers)", nf); return BL_ERR_CAPACITY; }
    std::vector<int32_t> offs((size_t)nf + 1), cells((size_t)total);
    BL_HIP(hipMemcpy(offs.data(), s->out_offsets, ((size_t)nf + 1) * 4, hipMemcpyDeviceToHost));
    if (total > 0) BL_HIP(hipMemcpy(cells.data(), s->out_cells, (size_t)total * 4, hipMemcpyDeviceToHost));
    bl_frontiers* f = new bl_frontiers();
    f->bfs_cells = s->h_counts[2]; f->bfs_levels = s->h_counts[3];
    f->offsets.push_back(0);
    for (int k = 0; k < nf; ++k) {
        const int cnt = offs[k + 1] - offs[k];
        // f.cells.size() * map.metersPerCell() >= minFrontierLength: size_t * float -> float, compared as double (:69)
        if (!((double)((float)(size_t)cnt * map->frame.mpc) >= min_frontier_length)) continue;
        for (int i = offs[k]; i < offs[k + 1]; ++i) {
            const int cx = cells[i] % W, cy = cells[i] / W;
            // grid_position_to_global_position(Point<int>) narrowed to Point<float> (grid_utils.hpp:14-19, frontiers.cpp:268)
            f->xy.push_back((float)((double)map->frame.ox + (double)cx * (double)map->frame.mpc));
            f->xy.push_back((float)((double)map->frame.oy + (double)cy * (double)map->frame.mpc));
        }
        f->offsets.push_back((int32_t)(f->xy.size() / 2));
    }
    *out = f;
    return BL_OK;
}

extern "C" int bl_frontiers_from_host(const int32_t* offsets, int count, const float* xy, bl_frontiers** out)
{
    BL_CHECK_ARG(out != nullptr && count >= 0 && (count == 0 || (offsets != nullptr && xy != nullptr)));
    bl_frontiers* f = new bl_frontiers();
    f->offsets.push_back(0);
    for (int k = 0; k < count; ++k) {
        if (offsets[k + 1] < offsets[k]) { delete f; bl_set_error("frontier offsets must not decrease"); return BL_ERR_ARG; }
        f->xy.insert(f->xy.end(), xy + 2 * (size_t)offsets[k], xy + 2 * (size_t)offsets[k + 1]);
        f->offsets.push_back((int32_t)(f->xy.size() / 2));
    }
    *out = f;
    return BL_OK;
}

extern "C" int bl_frontiers_count(const bl_frontiers* f) { return f ? (int)f->offsets.size() - 1 : 0; }
extern "C" int bl_frontiers_total_cells(const bl_frontiers* f) { return f ? (int)(f->xy.size() / 2) : 0; }
extern "C" int bl_frontiers_get(const bl_frontiers* f, int32_t* offsets, float* xy)
{
    BL_CHECK_ARG(f != nullptr && offsets != nullptr);
    memcpy(offsets, f->offsets.data(), f->offsets.size() * 4);
    if (!f->xy.empty()) { BL_CHECK_ARG(xy != nullptr); memcpy(xy, f->xy.data(), f->xy.size() * 4); }
    return BL_OK;
}
extern "C" int bl_frontiers_stats(const bl_frontiers* f, int* bfs_cells, int* bfs_levels)
{
    BL_CHECK_ARG(f != nullptr);
    if (bfs_cells) *bfs_cells = f->bfs_cells;
    if (bfs_levels) *bfs_levels = f->bfs_levels;
    return BL_OK;
}
extern "C" void bl_frontiers_destroy(bl_frontiers* f) { delete f; }
