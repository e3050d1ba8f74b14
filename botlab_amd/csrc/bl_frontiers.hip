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
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <deque>
#include <mutex>
#include <vector>

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
    int32_t* counts;            // [0] frontiers, [1] frontier cells, [2] free cells reached (+1), [3] levels, [4] overflow flag,
                                // [5], [6] time stamps, [8] "take the one-workgroup sweep" flag, [10] "k_frontier_grow2 declined: k_frontier_grow"
                                // flag, [12] frontier-class cells found, [13] touches found (k_frontier_touches: one 64-bit counter)
    int phase;                  // k_frontiers: 0 flood + sweep (small grids), 1 flood only, 2 sweep only (and only if counts[8] is set)
    uint2* touch;               // (key, cell) of every frontier cell the flood touched, in no order; FR_TOUCH_MAX entries
    int32_t* fcell;             // every frontier-class cell of the grid, in no order; FG_CELL_MAX entries (k_frontier_touches; counts[12])
    int grow_v1;                // 1, 2: take k_frontier_grow (visited set only, classes from global memory) whatever the map holds
    uint8_t* nb;                // large grids: per cell, the static classes of its four neighbours, 2 bits each (k_frontier_nb)
    const bl_pose_xyt_t* d_pose; // the robot pose in device memory (then rx, ry are formed by every kernel that needs them), or null
    bl_frame frame;
};

// robotCell = global_position_to_grid_cell(robotPose) (frontiers.cpp:39) when the pose lives on the device
__device__ __forceinline__ void fr_robot_cell(frontier_args& a)
{
    if (a.d_pose) { const bl_pose_xyt_t rp = *a.d_pose; bl_global_to_cell((double)rp.x, (double)rp.y, a.frame, &a.rx, &a.ry); }
}

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
#define FR_LQ 4096                // next-level queue entries mirrored in LDS (wider levels are re-read from the global queue)
#define FR_B 4                    // queue positions per thread in a batched level
#define FR_CH 8192                // slots of the LDS claim table of a level
#define FR_CH_PROBES 128           // linear probes after which the table counts as full

// Classification of every cell -- is_frontier_cell (frontiers.cpp:217-246) / free (:77) -- and the reset of the claim words.
// Independent per cell: for grids whose classes do not fit LDS this runs as its own launch over the whole device in front of
// the flood (one workgroup walking 16 M cells with five dependent loads each cost more than the flood itself).
__device__ __forceinline__ void fr_classify(const frontier_args& a, uint8_t* cls, long long c)
{
    const int x = (int)(c % a.W), y = (int)(c / a.W);
    const int v = a.cells[c];
    int k = 0;
    if (!(v > 0 || v < -5)) {                          // (map(x,y) > .1 || map(x,y) < -5) on an int8
        if (fr_log_odds(a, x - 1, y) < 0 || fr_log_odds(a, x + 1, y) < 0 || fr_log_odds(a, x, y + 1) < 0 || fr_log_odds(a, x, y - 1) < 0) k = 2;
    }
    if (k == 0 && v < 0) k = 1;
    if (x == a.rx && y == a.ry) k = 3;                  // visitedCells.insert(robotCell) before anything else (:42)
    cls[c] = (uint8_t)k;
    // the claim words are only ever read or written at cells of class 1 or 2 (a cell leaves those classes for 4 / 5 and never enters
    // them), the growth claims only at class 2: nothing else is reset -- 9 bytes written per cell became 1 on an unknown map
    if (k == 1 || k == 2) a.claim[c] = FR_INF;
    if (k == 2) a.fclaim[c] = FR_INF;
}

__global__ __launch_bounds__(256) void k_frontier_classify(frontier_args a)
{
    fr_robot_cell(a);
    const long long ncell = (long long)a.W * a.H;
    for (long long c = (long long)blockIdx.x * 256 + threadIdx.x; c < ncell; c += (long long)gridDim.x * 256) fr_classify(a, a.cls, c);
}

// One flood level of up to B x 1024 cells on a large grid, with the claims on free cells settled in an LDS table (cell -> smallest
// key) instead of atomicMin + read-back through L2: one global round trip (the class bytes, every load of the level in flight
// together) and the closing stores per level instead of three round trips.  Thread t owns the B consecutive queue positions
// lo + B t ...: thread order = queue order, so one scan places the winners.  A free cell that is still class 1 at the start of a
// level has never been claimed (every claimed free cell has a winner, and winners become class 4 before the next level starts),
// so claim[] is only needed for frontier cells -- their first-touch keys, read after the flood.  The table holds one entry per
// distinct claimed cell (about the size of the next level); returns the size of the next level, or -1 with nothing stored if the
// table got too crowded (the caller empties it).
template <int B>
__device__ __forceinline__ int fr_level_lds(const frontier_args& a, uint8_t* cls, const int* s_cur, int* s_next, int* s_hc, unsigned int* s_hk,
                                            int* s_wave, int* s_full, int lo, int hi)
{
    const int tid = threadIdx.x;
    int nc[B][4], kk[B][4], slot[B][4];
#pragma unroll
    for (int j = 0; j < B; ++j) {
        const int p = lo + B * tid + j;
#pragma unroll
        for (int n = 0; n < 4; ++n) { nc[j][n] = -1; slot[j][n] = -1; }
        if (p < hi) {
            int x, y;
            if (p == 0) { x = a.rx; y = a.ry; } else { const int c = lo > 0 ? s_cur[p - lo] : a.queue[p]; x = c % a.W; y = c / a.W; }
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const int nx = x + (n == 0 ? -1 : (n == 1 ? 1 : 0)), ny = y + (n == 2 ? 1 : (n == 3 ? -1 : 0));
                if (nx >= 0 && ny >= 0 && nx < a.W && ny < a.H) nc[j][n] = ny * a.W + nx;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < B; ++j)
#pragma unroll
        for (int n = 0; n < 4; ++n) kk[j][n] = nc[j][n] >= 0 ? (int)cls[nc[j][n]] : 0;
    bool full = false;
#pragma unroll
    for (int j = 0; j < B; ++j)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const unsigned int key = ((unsigned int)(lo + B * tid + j) << 2) | (unsigned int)n;
            if (kk[j][n] == 2) atomicMin(&a.claim[nc[j][n]], key);                // (harmless if the level is taken again)
            if (kk[j][n] == 1 && !full) {
                int sl = (int)(((unsigned int)nc[j][n] * 2654435761u) >> (32 - 13));             // FR_CH = 2^13
                int probes = 0;
                while (true) {
                    const int old_tag = atomicCAS(&s_hc[sl], -1, nc[j][n]);
                    if (old_tag == nc[j][n] || old_tag == -1) break;
                    if (B > 1 && ++probes > FR_CH_PROBES) { full = true; break; }    // a table this crowded: give the level up (B = 1: at most half full)
                    sl = (sl + 1) & (FR_CH - 1);
                }
                if (!full) { atomicMin(&s_hk[sl], key); slot[j][n] = sl; }
            }
        }
    if (B > 1 && full) *s_full = 1;
    __syncthreads();
    if (B > 1 && *s_full) { __syncthreads(); if (tid == 0) *s_full = 0; return -1; }
    int wins = 0;
    unsigned int winmask = 0;
#pragma unroll
    for (int j = 0; j < B; ++j)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const bool w = slot[j][n] >= 0 && s_hk[slot[j][n]] == ((((unsigned int)(lo + B * tid + j)) << 2) | (unsigned int)n);
            winmask |= (w ? 1u : 0u) << (4 * j + n);
            wins += w ? 1 : 0;
        }
    int total;
    int at = hi + block_excl_scan(wins, s_wave, &total);                            // (its barriers: every thread has read its slots)
#pragma unroll
    for (int j = 0; j < B; ++j)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            if (slot[j][n] >= 0) { s_hc[slot[j][n]] = -1; s_hk[slot[j][n]] = FR_INF; }       // the table is empty again for the next level
            if ((winmask >> (4 * j + n)) & 1u) {
                a.queue[at] = nc[j][n];
                cls[nc[j][n]] = 4;                                                  // visited
                if (at - hi < FR_LQ) s_next[at - hi] = nc[j][n];
                at += 1;
            }
        }
    __syncthreads();
    return total;
}

template <bool CLS_LDS>
__global__ __launch_bounds__(FR_T) void k_frontiers(frontier_args a)
{
    fr_robot_cell(a);
    const long long t_begin = wall_clock64();
    extern __shared__ uint8_t s_cls[];
    __shared__ int s_wave[FR_T / 64];
    __shared__ unsigned int s_umin[FR_T / 64];
    __shared__ int s_q[2][FR_LQ];
    // claims of a narrow level (large grids only: the small-grid form spends its LDS on the class bytes)
    __shared__ int s_hc[CLS_LDS ? 1 : FR_CH];
    __shared__ unsigned int s_hk[CLS_LDS ? 1 : FR_CH];
    __shared__ int s_full;
    const int tid = threadIdx.x;
    const long long ncell = (long long)a.W * a.H;
    uint8_t* cls = CLS_LDS ? s_cls : a.cls;
    // ---- classification (large grids: done by k_frontier_classify in front of this launch)
    if (CLS_LDS)
        for (long long c = tid; c < ncell; c += FR_T) fr_classify(a, cls, c);
    __threadfence();
    __syncthreads();
    // ---- free-space flood (:47-82), xDeltas {-1,1,0,0}, yDeltas {0,0,1,-1}
    int lo = 0, hi = 1, levels = 0, cur = 0;
    if (!CLS_LDS) {
        for (int i = tid; i < FR_CH; i += FR_T) { s_hc[i] = -1; s_hk[i] = FR_INF; }
        if (tid == 0) s_full = 0;
        __syncthreads();
    }
    if (a.phase == 2) {                                     // the flood has run (phase 1): only the sweep, and only when asked for
        if (a.counts[8] == 0) return;
        lo = hi = a.counts[2]; levels = a.counts[3];
    }
    while (lo < hi) {
        if (!CLS_LDS && hi - lo <= FR_B * FR_T) {
            // ---- large grids, levels of up to FR_B x 1024 cells: claims on free cells settled in LDS (fr_level_lds)
            const int total = hi - lo <= FR_T ? fr_level_lds<1>(a, cls, s_q[cur], s_q[cur ^ 1], s_hc, s_hk, s_wave, &s_full, lo, hi)
                                              : fr_level_lds<FR_B>(a, cls, s_q[cur], s_q[cur ^ 1], s_hc, s_hk, s_wave, &s_full, lo, hi);
            if (total >= 0) { lo = hi; hi += total; levels += 1; cur ^= 1; continue; }
            // the table filled up (a level that claims several times its own size): empty it, take the level through claim[]
            for (int i = tid; i < FR_CH; i += FR_T) { s_hc[i] = -1; s_hk[i] = FR_INF; }
            __syncthreads();
        }
        if (hi - lo > FR_T && hi - lo <= FR_B * FR_T) {
            // ---- a level of 1025 .. FR_B x 1024 cells in one batch: thread t owns the FR_B consecutive queue positions lo + FR_B t ...
            // (thread order = queue order, so one scan places the winners), everything in registers, every load of a phase in
            // flight together: three global round trips per level whatever its width (the per-position loop below pays them
            // once per 1024 positions and reads the queue back from memory)
            const bool from_lds = hi - lo <= FR_LQ && lo > 0;
            int nc[FR_B][4], kk[FR_B][4];
#pragma unroll
            for (int j = 0; j < FR_B; ++j) {
                const int p = lo + FR_B * tid + j;
#pragma unroll
                for (int n = 0; n < 4; ++n) nc[j][n] = -1;
                if (p < hi) {
                    int x, y;
                    if (p == 0) { x = a.rx; y = a.ry; } else { const int c = from_lds ? s_q[cur][p - lo] : a.queue[p]; x = c % a.W; y = c / a.W; }
#pragma unroll
                    for (int n = 0; n < 4; ++n) {
                        const int nx = x + (n == 0 ? -1 : (n == 1 ? 1 : 0)), ny = y + (n == 2 ? 1 : (n == 3 ? -1 : 0));
                        if (nx >= 0 && ny >= 0 && nx < a.W && ny < a.H) nc[j][n] = ny * a.W + nx;
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < FR_B; ++j)
#pragma unroll
                for (int n = 0; n < 4; ++n) kk[j][n] = nc[j][n] >= 0 ? (int)cls[nc[j][n]] : 0;
#pragma unroll
            for (int j = 0; j < FR_B; ++j)
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    if (kk[j][n] == 1 || kk[j][n] == 2) atomicMin(&a.claim[nc[j][n]], ((unsigned int)(lo + FR_B * tid + j) << 2) | (unsigned int)n);
            __syncthreads();
            unsigned int got[FR_B][4];
#pragma unroll
            for (int j = 0; j < FR_B; ++j)
#pragma unroll
                for (int n = 0; n < 4; ++n) got[j][n] = kk[j][n] == 1 ? ld_claim(&a.claim[nc[j][n]]) : FR_INF;
            int wins = 0;
#pragma unroll
            for (int j = 0; j < FR_B; ++j)
#pragma unroll
                for (int n = 0; n < 4; ++n) wins += got[j][n] == ((((unsigned int)(lo + FR_B * tid + j)) << 2) | (unsigned int)n) ? 1 : 0;
            int total;
            int at = hi + block_excl_scan(wins, s_wave, &total);
#pragma unroll
            for (int j = 0; j < FR_B; ++j)
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    if (got[j][n] == ((((unsigned int)(lo + FR_B * tid + j)) << 2) | (unsigned int)n)) {
                        a.queue[at] = nc[j][n];
                        cls[nc[j][n]] = 4;                  // visited: never claimed again (its claim key stays the smallest anyway)
                        if (at - hi < FR_LQ) s_q[cur ^ 1][at - hi] = nc[j][n];
                        at += 1;
                    }
            __syncthreads();
            lo = hi; hi += total; levels += 1; cur ^= 1;
            continue;
        }
        const bool one_pass = hi - lo <= FR_T;              // the common case on small maps: this thread's neighbours stay in registers
        const bool from_lds = hi - lo <= FR_LQ && lo > 0;   // (levels wider than FR_B x 1024: 1024 positions at a time)
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
    const long long t_flood = wall_clock64();
    if (a.phase == 1) {
        if (tid == 0) { a.counts[2] = qn; a.counts[3] = levels; a.counts[5] = (int)(t_flood - t_begin); a.counts[8] = 0; a.counts[10] = 0; a.counts[12] = 0; a.counts[13] = 0; }
        return;
    }
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
        if (a.phase == 0) a.counts[5] = (int)(t_flood - t_begin);
        a.counts[6] = (int)(wall_clock64() - t_flood);     // 100 MHz ticks: flood, frontier sweep
    }
}

// ---- large grids: the flood with everything a level needs in LDS ---------------------------------------------------------------
// The level loop of k_frontiers<false> pays a global round trip per level for the class bytes of the neighbours (39 ms for 3582
// levels at 4096^2, 11 us a level).  Here a level needs NO global load that was not issued a level earlier:
//   * k_frontier_nb leaves, per cell, the STATIC classes of its four neighbours in one byte (free / frontier / other); a queue
//     entry carries its cell's byte, loaded when the cell was claimed (the load flies while the level's scan runs);
//   * what is dynamic -- which free neighbours are already visited -- needs no memory either: the 4-connected grid is
//     bipartite, so every visited neighbour of an unvisited cell nc reached from level L lies in level L itself (a visited
//     neighbour at a lower level would have claimed nc earlier).  The cells of the running level sit in an LDS hash
//     (cell -> queue position); a claimer (p, n) of nc looks up nc's three other neighbours there: those it finds are nc's
//     other claimers (it wins iff none has a smaller position: the key 4p+n order) and, with p itself, nc's visited-neighbour
//     mask, which travels in nc's queue entry and spares the next level the claims on its own parents.
// No atomics decide anything (the hash is insert-once, lookup-only), the queue order is the thread order, one scan places the
// winners: four workgroup barriers and no exposed global latency per level.  Entries are x | y << 14 | mask << 28.  Levels wider
// than FL_QMAX cells, and the first one (the robot cell may lie off the grid), take the generic path over cls[] / claim[].
#define FL_T 1024
#define FL_MAXB 8                          // chunks of 1024 positions a level may have
#define FL_QMAX (FL_T * FL_MAXB)
#define FL_HS 16384                        // hash slots: twice the widest level
#define FL_LDS_BYTES (2 * FL_QMAX * 4 + 2 * FL_QMAX + (FL_HS + 32) * 4 + 256)
#define FL_XY(x, y) ((unsigned int)(x) | ((unsigned int)(y) << 14))

__global__ __launch_bounds__(256) void k_frontier_nb(frontier_args a)
{
    const long long ncell = (long long)a.W * a.H;
    for (long long c = (long long)blockIdx.x * 256 + threadIdx.x; c < ncell; c += (long long)gridDim.x * 256) {
        const int x = (int)(c % a.W), y = (int)(c / a.W);
        unsigned int b = 0;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int nx = x + (n == 0 ? -1 : (n == 1 ? 1 : 0)), ny = y + (n == 2 ? 1 : (n == 3 ? -1 : 0));
            if (nx < 0 || ny < 0 || nx >= a.W || ny >= a.H) continue;
            const unsigned int k = a.cls[(size_t)ny * a.W + nx];
            if (k == 1u || k == 2u) b |= k << (2 * n);                 // (the robot cell, class 3, is visited from the start: "other")
        }
        a.nb[c] = (uint8_t)b;
    }
}

// A workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding global store and load of
// the wave (s_waitcnt vmcnt(0)): with the level's queue / class stores and the next level's neighbour bytes in flight that is two
// exposed global round trips per level -- the very cost this kernel exists to avoid.  Nothing a level reads from global memory
// was written by the same launch (the generic path, which does, keeps __syncthreads and is entered behind one).
__device__ __forceinline__ void fl_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ int fl_excl_scan(int v, int* s_wave, int* total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = v;
    for (int off = 1; off < 64; off <<= 1) {
        int t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    if (lane == 63) s_wave[wave] = incl;               // (s_wave is read only between this barrier and the level's last one)
    fl_barrier();
    int base = 0, tot = 0;
    for (int w = 0; w < FL_T / 64; ++w) { int x = s_wave[w]; if (w < wave) base += x; tot += x; }
    *total = tot;
    return base + incl - v;
}

// The level's hash, one 32-bit word per slot.  home = (x + FL_HC y) mod FL_HS -- the homes of a cell's neighbours are the cell's
// home plus a constant, lanes that hold the cells of a diagonal or straight run of the front read different LDS banks, and no
// multiplication is spent (the flood is bound by VALU issue: 16 waves on 4 SIMDs, four cycles an instruction) -- and an entry that
// sits d slots behind its home stores y | d << 14, which with the slot names its cell exactly, beside the cell's queue position:
// y | d << 14 | position << 19.  Linear probing without wrap-around (FL_PAD spare slots behind the table), insert-once (one
// atomicCAS), never deleted while lookups run: a lookup ends at the first empty slot.  d <= FL_DMAX, or the level is given up.
// A cell code off the grid (x - 1 at x = 0 borrows from y) names no cell of the level as long as W, H <= FL_MAX_SIDE.
#define FL_EMPTY 0xFFFFFFFFu
#define FL_DMAX 30
#define FL_PAD 32
#define FL_HC 90
#define FL_MAX_SIDE 16380
__device__ __forceinline__ unsigned int fl_home(unsigned int x, unsigned int y) { return (x + FL_HC * y) & (FL_HS - 1); }

#ifdef FL_STAMPS
__device__ long long g_fl_stamp[8];
#define FLS(i) do { if (threadIdx.x == 0) { const long long t_ = wall_clock64(); g_fl_stamp[i] += t_ - t_prev; t_prev = t_; } } while (0)
#else
#define FLS(i) do { } while (0)
#endif

// exclusive prefix of a per-lane count 0..3 over the workgroup (ballots inside the wave, one LDS word per wave, a 16-lane DPP scan
// over the waves): *total = the workgroup's sum.  One barrier.
__device__ __forceinline__ int fl_scan_small(int v, int* s_wave, int* total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long b0 = __ballot(v & 1), b1 = __ballot(v & 2);
    const int below = (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(b0 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)b0, 0u)) +
                      2 * (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)b1, 0u));
    if (lane == 0) s_wave[wave] = __popcll(b0) + 2 * __popcll(b1);
    fl_barrier();
    int t = s_wave[lane & 15];                               // lanes 0..15 of every row: the sixteen waves' sums
    t += __builtin_amdgcn_update_dpp(0, t, 0x111, 0xf, 0xf, false);      // row_shr:1 ... inclusive scan inside the row of 16
    t += __builtin_amdgcn_update_dpp(0, t, 0x112, 0xf, 0xf, false);
    t += __builtin_amdgcn_update_dpp(0, t, 0x114, 0xf, 0xf, false);
    t += __builtin_amdgcn_update_dpp(0, t, 0x118, 0xf, 0xf, false);
    *total = __builtin_amdgcn_readlane(t, 15);
    const int base = wave == 0 ? 0 : __builtin_amdgcn_readlane(t, wave - 1);
    return base + below;
}

// A level of hi - lo <= FL_QMAX cells whose entries are in s_qc / s_nbc; leaves the next level's in s_qn / s_nbn.  Thread t takes
// the positions t, t + 1024, ... (a chunk of 1024 at a time: inside a chunk thread order = queue order, one scan per chunk places
// its winners behind those of the chunks before).  Returns the next level's size (beyond FL_QMAX only a.queue holds it), or -1
// with nothing changed but the (emptied) table when an entry would not fit within FL_DMAX slots of its home.
__device__ __forceinline__ int fl_level(const frontier_args& a, const unsigned int* s_qc, unsigned int* s_qn, const uint8_t* s_nbc, uint8_t* s_nbn,
                                        unsigned int* s_tab, int* s_wave, int* s_flag, int lo, int hi)
{
#ifdef FL_STAMPS
    long long t_prev = wall_clock64();
#endif
    const int tid = threadIdx.x, width = hi - lo;
    const int nchunk = (width + FL_T - 1) / FL_T;
    // ---- the level's cells into the table
    bool overflow = false;
    for (int ch = 0; ch < nchunk; ++ch) {
        const int p = ch * FL_T + tid;
        if (p >= width) break;
        const unsigned int e = s_qc[p];
        const unsigned int x = e & 0x3FFFu, y = (e >> 14) & 0x3FFFu;
        const unsigned int home = fl_home(x, y), body = y | ((unsigned int)p << 19);
        unsigned int d = 0;
        for (; d <= FL_DMAX; ++d)
            if (atomicCAS(&s_tab[home + d], FL_EMPTY, body | (d << 14)) == FL_EMPTY) break;
        if (d > FL_DMAX) overflow = true;
    }
    if (overflow) *s_flag = 1;
    FLS(0);
    fl_barrier();
    FLS(1);
    int base = 0;
    const bool gave_up = *s_flag != 0;
    if (!gave_up) {
    for (int ch = 0; ch < nchunk; ++ch) {
        const int p = ch * FL_T + tid;
        unsigned int wxy[4], wnb[4];
        unsigned int winmask = 0;
        if (p < width) {
            const unsigned int ent = s_qc[p], nbb = s_nbc[p];
            const unsigned int x = ent & 0x3FFFu, y = (ent >> 14) & 0x3FFFu, xy = ent & 0x0FFFFFFFu;
            const unsigned int vmask = ent >> 28;
            unsigned int f = nbb & ~(nbb >> 1) & 0x55u;                 // bit 2n: neighbour n is free
            const unsigned int t2 = (nbb >> 1) & ~nbb & 0x55u;          // bit 2n: neighbour n is a frontier cell
            f = (f | (f >> 1)) & 0x33u; f = (f | (f >> 2)) & 0x0Fu;
            const unsigned int cand = f & ~vmask;                      // free neighbours that are not parents
            if (t2 != 0u) {
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    if ((t2 >> (2 * n)) & 1u) {
                        const int nx = (int)x + (n == 0 ? -1 : (n == 1 ? 1 : 0)), ny = (int)y + (n == 2 ? 1 : (n == 3 ? -1 : 0));
                        atomicMin(&a.claim[(size_t)ny * a.W + nx], ((unsigned int)(lo + p) << 2) | (unsigned int)n);
                    }
            }
            if (cand != 0u) {
                // Who else of this level touches the cells p can claim?  The three other neighbours of p + d_n are among the eight
                // cells at distance 2 of p: index 0..3 = p + 2 d_n; 4 = (-1,+1), 5 = (-1,-1), 6 = (+1,+1), 7 = (+1,-1); cell i
                // matters to the candidates cm[i].  Looked up once per position, first and second probe in one LDS read.
                const unsigned int home = fl_home(x, y);
                unsigned int found = 0, earlier = 0, unresolved = 0;
                // every read of the position first (cells no candidate asks for read the position's own home: harmless) ...
                unsigned int w0[8], w1[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const unsigned int cm = i < 4 ? (1u << i) : (i == 4 ? 0x5u : (i == 5 ? 0x9u : (i == 6 ? 0x6u : 0xAu)));
                    const int dx = i == 0 ? -2 : (i == 1 ? 2 : (i < 4 ? 0 : (i < 6 ? -1 : 1)));
                    const int dy = i == 2 ? 2 : (i == 3 ? -2 : (i < 4 ? 0 : ((i == 4 || i == 6) ? 1 : -1)));
                    const unsigned int hq = (home + ((cand & cm) ? (unsigned int)((dx + FL_HC * dy) & (FL_HS - 1)) : 0u)) & (FL_HS - 1);
                    w0[i] = s_tab[hq]; w1[i] = s_tab[hq + 1];
                }
                // ... then what they say, without a branch
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const unsigned int cm = i < 4 ? (1u << i) : (i == 4 ? 0x5u : (i == 5 ? 0x9u : (i == 6 ? 0x6u : 0xAu)));
                    const int dy = i == 2 ? 2 : (i == 3 ? -2 : (i < 4 ? 0 : ((i == 4 || i == 6) ? 1 : -1)));
                    const unsigned int k13 = (y + (unsigned int)dy) << 13;             // the key y | 0 << 14, shifted to the top
                    const bool need = (cand & cm) != 0u;
                    const bool m0 = (w0[i] << 13) == k13, m1 = (w1[i] << 13) == k13 + (1u << 27);
                    const bool hit = need && (m0 || m1);
                    const unsigned int pos = (m0 ? w0[i] : w1[i]) >> 19;
                    found |= hit ? (1u << i) : 0u;
                    earlier |= (hit && pos < (unsigned int)p) ? (1u << i) : 0u;
                    unresolved |= (need && !(m0 || m1) && max(w0[i], w1[i]) != FL_EMPTY) ? (1u << i) : 0u;   // neither probe ended the chain
                }
                while (unresolved) {                                   // third and later probes: rare
                    const int i0 = __ffs((int)unresolved) - 1;
                    unresolved &= unresolved - 1u;
                    const int dx = i0 == 0 ? -2 : (i0 == 1 ? 2 : (i0 < 4 ? 0 : (i0 < 6 ? -1 : 1)));
                    const int dy = i0 == 2 ? 2 : (i0 == 3 ? -2 : (i0 < 4 ? 0 : ((i0 == 4 || i0 == 6) ? 1 : -1)));
                    const unsigned int hq = (home + (unsigned int)((dx + FL_HC * dy) & (FL_HS - 1))) & (FL_HS - 1);
                    const unsigned int k13 = (y + (unsigned int)dy) << 13;
                    for (unsigned int d = 2; d <= FL_DMAX; ++d) {
                        const unsigned int w = s_tab[hq + d];
                        if (w == FL_EMPTY) break;
                        if ((w << 13) == k13 + (d << 27)) { found |= 1u << i0; if ((w >> 19) < (unsigned int)p) earlier |= 1u << i0; break; }
                    }
                }
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    if (!((cand >> n) & 1u)) continue;
                    // neighbours of nc = p + d_n other than p: m = n (straight on: cell n) and the two perpendicular ones
                    // n = 0 (-1,0): m = 2 -> cell 4, m = 3 -> cell 5;   n = 1 (+1,0): m = 2 -> 6, m = 3 -> 7
                    // n = 2 (0,+1): m = 0 -> cell 4, m = 1 -> cell 6;   n = 3 (0,-1): m = 0 -> 5, m = 1 -> 7
                    const int ma = n < 2 ? 2 : 0, mb = n < 2 ? 3 : 1;
                    const int ia = n == 0 ? 4 : (n == 1 ? 6 : (n == 2 ? 4 : 5));
                    const int ib = n == 0 ? 5 : (n == 1 ? 7 : (n == 2 ? 6 : 7));
                    if (earlier & ((1u << n) | (1u << ia) | (1u << ib))) continue;     // a claimer in front of p
                    const unsigned int vm = (1u << (n ^ 1)) | (((found >> n) & 1u) << n) | (((found >> ia) & 1u) << ma) | (((found >> ib) & 1u) << mb);
                    const int dxy = n == 0 ? -1 : (n == 1 ? 1 : (n == 2 ? (1 << 14) : -(1 << 14)));
                    const int dc = n == 0 ? -1 : (n == 1 ? 1 : (n == 2 ? a.W : -a.W));
                    wxy[n] = (xy + (unsigned int)dxy) | (vm << 28);
                    wnb[n] = a.nb[(int)(y * (unsigned int)a.W + x) + dc];              // in flight while the scan runs
                    winmask |= 1u << n;
                }
            }
        }
        int total;
        FLS(2);
        int at = base + fl_scan_small(__popc(winmask), s_wave + (ch & 1) * (FL_T / 64), &total);
        FLS(3);
#pragma unroll
        for (int n = 0; n < 4; ++n)
            if ((winmask >> n) & 1u) {
                const unsigned int e = wxy[n];
                if (at < FL_QMAX) { s_qn[at] = e; s_nbn[at] = (uint8_t)wnb[n]; }
                a.queue[hi + at] = (int)((e >> 14) & 0x3FFFu) * a.W + (int)(e & 0x3FFFu);
                at += 1;
            }
        base += total;
        FLS(4);
    }
    }
    // ---- the table is emptied (every lookup of the level lies before the last scan's barrier -- or, when the level was given
    // up, before the barrier above): 16 slots per thread, whole
    {
        uint4* t4 = (uint4*)s_tab;
        const uint4 e4 = make_uint4(FL_EMPTY, FL_EMPTY, FL_EMPTY, FL_EMPTY);
#pragma unroll
        for (int i = 0; i < FL_HS / 4 / FL_T; ++i) t4[i * FL_T + tid] = e4;
        if (tid < FL_PAD / 4) t4[FL_HS / 4 + tid] = e4;
    }
    fl_barrier();
    if (gave_up) { if (tid == 0) *s_flag = 0; fl_barrier(); return -1; }
    FLS(5);
    return base;
}

// entries of the level lo..hi (cells in a.queue, marks in cls[]) into s_q / s_nb: after a level of the generic path
__device__ __forceinline__ void fl_refill(const frontier_args& a, unsigned int* s_q, uint8_t* s_nb, int lo, int hi)
{
    for (int p = (int)threadIdx.x; p < hi - lo; p += FL_T) {
        const int c = a.queue[lo + p];
        const int x = c % a.W, y = c / a.W;
        unsigned int vm = 0;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int nx = x + (n == 0 ? -1 : (n == 1 ? 1 : 0)), ny = y + (n == 2 ? 1 : (n == 3 ? -1 : 0));
            if (nx < 0 || ny < 0 || nx >= a.W || ny >= a.H) continue;
            if (a.cls[(size_t)ny * a.W + nx] == 4) vm |= 1u << n;
        }
        s_q[p] = FL_XY(x, y) | (vm << 28);
        s_nb[p] = a.nb[c];
    }
    __syncthreads();
}

// a level of any width over cls[] / claim[] (the level loop of k_frontiers, without its LDS mirrors)
__device__ __forceinline__ int fl_level_generic(const frontier_args& a, int* s_wave, int lo, int hi)
{
    const int tid = threadIdx.x;
    for (int p = lo + tid; p < hi; p += FL_T) {
        int x, y;
        if (p == 0) { x = a.rx; y = a.ry; } else { const int c = a.queue[p]; x = c % a.W; y = c / a.W; }
        for (int n = 0; n < 4; ++n) {
            const int nx = x + (n == 0 ? -1 : (n == 1 ? 1 : 0)), ny = y + (n == 2 ? 1 : (n == 3 ? -1 : 0));
            if (nx < 0 || ny < 0 || nx >= a.W || ny >= a.H) continue;
            const int c = ny * a.W + nx;
            const int k = a.cls[c];
            if (k == 1 || k == 2) atomicMin(&a.claim[c], ((unsigned int)p << 2) | (unsigned int)n);
        }
    }
    __syncthreads();
    int newhi = hi;
    for (int base = lo; base < hi; base += FL_T) {
        const int p = base + tid;
        int wins = 0, wc[4];
        if (p < hi) {
            int x, y;
            if (p == 0) { x = a.rx; y = a.ry; } else { const int c0 = a.queue[p]; x = c0 % a.W; y = c0 / a.W; }
            for (int n = 0; n < 4; ++n) {
                const int nx = x + (n == 0 ? -1 : (n == 1 ? 1 : 0)), ny = y + (n == 2 ? 1 : (n == 3 ? -1 : 0));
                if (nx < 0 || ny < 0 || nx >= a.W || ny >= a.H) continue;
                const int c = ny * a.W + nx;
                if (a.cls[c] == 1 && ld_claim(&a.claim[c]) == (((unsigned int)p << 2) | (unsigned int)n)) wc[wins++] = c;
            }
        }
        int total;
        const int off = block_excl_scan(wins, s_wave, &total);
        for (int j = 0; j < wins; ++j) a.queue[newhi + off + j] = wc[j];
        newhi += total;
    }
    __syncthreads();
    // winners become visited only now: a cell claimed in this level must look unvisited to every claimer of the level
    for (int q = hi + tid; q < newhi; q += FL_T) a.cls[a.queue[q]] = 4;
    __syncthreads();
    return newhi - hi;
}

__global__ __launch_bounds__(FL_T) void k_frontier_flood(frontier_args a)
{
    fr_robot_cell(a);
    const long long t_begin = wall_clock64();
    extern __shared__ __align__(16) uint8_t s_fl[];
    unsigned int* s_q0 = (unsigned int*)s_fl;                                  // [2][FL_QMAX]
    unsigned int* s_tab = (unsigned int*)(s_fl + 2 * FL_QMAX * 4);
    uint8_t* s_nb0 = s_fl + 2 * FL_QMAX * 4 + (FL_HS + FL_PAD) * 4;             // [2][FL_QMAX]
    int* s_wave = (int*)(s_fl + 2 * FL_QMAX * 4 + (FL_HS + FL_PAD) * 4 + 2 * FL_QMAX);     // [2][16]
    int* s_flag = s_wave + 2 * (FL_T / 64);
    const int tid = threadIdx.x;
    for (int i = tid; i < FL_HS + FL_PAD; i += FL_T) s_tab[i] = FL_EMPTY;
    if (tid == 0) *s_flag = 0;
    __syncthreads();
    int lo = 0, hi = 1, levels = 0, cur = 0;
    int marked = 1;                                        // queue positions below it carry the visited mark (cls 4)
    bool packed = false;                                   // s_q[cur] / s_nb[cur] hold the level lo..hi
    // the packed levels leave the visited marks to whoever needs them: the generic path and the refill read cls[]
    auto catch_up = [&]() {
        __syncthreads();                                   // the packed levels' queue stores have landed
        for (int q = marked + tid; q < hi; q += FL_T) a.cls[a.queue[q]] = 4;
        __syncthreads();
        marked = hi;
    };
    while (lo < hi) {
        const int width = hi - lo;
        int total = -1;
        if (lo > 0 && width <= FL_QMAX) {
            unsigned int* qc = s_q0 + cur * FL_QMAX; unsigned int* qn = s_q0 + (cur ^ 1) * FL_QMAX;
            uint8_t* nc = s_nb0 + cur * FL_QMAX; uint8_t* nn = s_nb0 + (cur ^ 1) * FL_QMAX;
            if (!packed) { catch_up(); fl_refill(a, qc, nc, lo, hi); }
            total = fl_level(a, qc, qn, nc, nn, s_tab, s_wave, s_flag, lo, hi);
            packed = total >= 0 && total <= FL_QMAX;
            if (total >= 0) cur ^= 1;
        }
        if (total < 0) {
            catch_up();
            total = fl_level_generic(a, s_wave, lo, hi);
            marked = hi + total;
            packed = false;
        }
        lo = hi; hi += total; levels += 1;
    }
    if (tid == 0) { a.counts[2] = hi; a.counts[3] = levels; a.counts[5] = (int)(wall_clock64() - t_begin); a.counts[8] = 0; a.counts[10] = 0; a.counts[12] = 0; a.counts[13] = 0; }
#ifdef FL_STAMPS
    if (tid == 0) {
        printf("[flood stamps, 100 MHz ticks] insert %lld  barrier1 %lld  claims %lld  scan %lld  write %lld  barrier4 %lld\n",
               g_fl_stamp[0], g_fl_stamp[1], g_fl_stamp[2], g_fl_stamp[3], g_fl_stamp[4], g_fl_stamp[5]);
        for (int i = 0; i < 8; ++i) g_fl_stamp[i] = 0;
    }
#endif
}

// ---- large grids: the frontier sweep as two launches ------------------------------------------------------------------------
// The one-workgroup sweep walks the whole free-space queue again (7 M positions at 4096^2) to find the touches in key order, and
// grows each frontier level by level with three global round trips per level -- frontiers are thin curves, so a level holds two
// cells (measured: 6.7 of 12.9 ms at 2000^2, 42 of 133 ms at 4096^2).  Instead:
//   k_frontier_touches  every frontier cell the flood touched carries its first-touch key in claim[]: all workgroups collect
//                       (key, cell) pairs, in no order (a few thousand on a real map);
//   k_frontier_grow     one workgroup: the live touch with the smallest key is the next seed (a touch is dead once its cell is
//                       part of a grown frontier); ONE WAVE then runs the reference's FIFO growth serially, eight lanes looking
//                       at the eight neighbours of up to eight queued cells per global round trip, the visited set an LDS hash.
// A map with more touches than FR_TOUCH_MAX or a frontier larger than the hash holds takes the one-workgroup sweep instead
// (k_frontiers phase 2, launched behind this one: it returns at once unless counts[8] is set).
#define FR_TOUCH_MAX 16384
#define FR_TOUCH_PER_THREAD (FR_TOUCH_MAX / FR_T)
#define FR_HASH 16384                       // slots of the visited set (a power of two); a frontier may fill half of them
#define FR_RING 1024                        // growth queue entries mirrored in LDS
#define FG_SLOTS 32768                      // k_frontier_grow2: slots of the set of ALL frontier-class cells (a power of two) ...
#define FG_CELL_MAX 16384                   // ... which it fills to one half at most
#define FG_DMAX 64                          // an insert that finds no room within so many slots of its home gives the sweep to k_frontier_grow
#define FG_PAD 128                          // no wrap-around: spare slots behind the last home
#define FG_LDS_BYTES ((FG_SLOTS + FG_PAD) * 4 + FR_RING * 4)

__global__ __launch_bounds__(256) void k_frontier_touches(frontier_args a)
{
    // four class bytes per thread and round; a wave that sees no frontier-class cell (nearly all do not) goes on after one ballot
    const long long ncell = (long long)a.W * a.H;
    const int lane = threadIdx.x & 63;
    for (long long c0 = ((long long)blockIdx.x * 256) * 4; c0 < ncell; c0 += (long long)gridDim.x * 256 * 4) {
        const long long c = c0 + 4 * (long long)threadIdx.x;
        unsigned int w = 0;
        if (c + 3 < ncell) w = *(const unsigned int*)(a.cls + c);
        else for (int b = 0; b < 4; ++b) if (c + b < ncell) w |= (unsigned int)a.cls[c + b] << (8 * b);
        const unsigned int x = w ^ 0x02020202u;
        const bool any = ((x - 0x01010101u) & ~x & 0x80808080u) != 0u;          // some byte of w is 2
        if (__ballot(any) == 0ull) continue;
        // every frontier-class cell goes on k_frontier_grow2's list, every touched one on the touch list.  The two counters are this
        // kernel's bottleneck (every add returns, and adds to one cache line take ~10 ns each at the L2): ONE 64-bit add per wave
        // and round carries both counts
        unsigned int key[4];
        unsigned long long m2[4], mt[4];
        int n2 = 0, nt = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const bool is2 = ((w >> (8 * b)) & 0xFFu) == 2u;
            key[b] = FR_INF;
            if (is2) key[b] = a.claim[c + b];
            m2[b] = __ballot(is2); mt[b] = __ballot(key[b] != FR_INF);
            n2 += __popcll(m2[b]); nt += __popcll(mt[b]);
        }
        unsigned long long base = 0ull;
        if (lane == 0) base = atomicAdd((unsigned long long*)&a.counts[12], (unsigned long long)(unsigned int)n2 | ((unsigned long long)(unsigned int)nt << 32));
        base = __shfl(base, 0, 64);
        int at2 = (int)(unsigned int)base, at = (int)(base >> 32);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            if ((m2[b] >> lane) & 1ull) {
                const int i2 = at2 + (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(m2[b] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m2[b], 0u));
                if (i2 < FG_CELL_MAX) a.fcell[i2] = (int)(c + b);
            }
            if ((mt[b] >> lane) & 1ull) {
                const int it = at + (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(mt[b] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)mt[b], 0u));
                if (it < FR_TOUCH_MAX) a.touch[it] = make_uint2(key[b], (unsigned int)(c + b));
            }
            at2 += __popcll(m2[b]); at += __popcll(mt[b]);
        }
    }
}

// wave-uniform: is x in the set, and if not, put it there.  Returns 1 (was there), 0 (inserted), -1 (no room within 64 slots)
__device__ __forceinline__ int fr_hash_test_and_set(int* s_hash, int x, int lane)
{
    const unsigned int h = ((unsigned int)x * 2654435761u) >> (32 - 14);          // FR_HASH = 2^14
    const int slot = (int)((h + (unsigned int)lane) & (FR_HASH - 1));
    const int v = s_hash[slot];
    const unsigned long long hit = __ballot(v == x), empty = __ballot(v == -1);
    const int first_empty = empty ? __ffsll((long long)empty) - 1 : 64;
    if (hit && (__ffsll((long long)hit) - 1) < first_empty) return 1;
    if (first_empty == 64) return -1;
    if (lane == first_empty) s_hash[slot] = x;
    __builtin_amdgcn_wave_barrier();
    return 0;
}

__global__ __launch_bounds__(FR_T) void k_frontier_grow(frontier_args a)
{
    __shared__ int s_hash[FR_HASH];
    __shared__ int s_ring[FR_RING];
    __shared__ unsigned int s_umin[FR_T / 64];
    __shared__ int s_seed, s_cnt, s_fail;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long t_begin = wall_clock64();
    const int T = a.counts[13];
    if (a.counts[10] == 0) return;                         // k_frontier_grow2 has grown the frontiers
    if (T > FR_TOUCH_MAX || a.W > 65535 || a.H > 32767) { if (tid == 0) a.counts[8] = 1; return; }      // (the ring packs x | y << 16)
    unsigned int tk[FR_TOUCH_PER_THREAD]; int tc[FR_TOUCH_PER_THREAD];
#pragma unroll
    for (int j = 0; j < FR_TOUCH_PER_THREAD; ++j) {
        const int i = j * FR_T + tid;
        tk[j] = FR_INF; tc[j] = 0;
        if (i < T) { const uint2 t = a.touch[i]; tk[j] = t.x; tc[j] = (int)t.y; }
    }
    int nf = 0, total = 0, overflow = 0;
    while (true) {
        // ---- the next seed: the live touch with the smallest key (frontiers.cpp:66-75 meets them in that order)
        unsigned int mine = FR_INF; int mine_c = 0;
#pragma unroll
        for (int j = 0; j < FR_TOUCH_PER_THREAD; ++j) {
            if (tk[j] == FR_INF) continue;
            if (a.cls[tc[j]] != 2) { tk[j] = FR_INF; continue; }                 // its frontier has been grown
            if (tk[j] < mine) { mine = tk[j]; mine_c = tc[j]; }
        }
        const unsigned int best = block_min(mine, s_umin);
        if (best == FR_INF) break;
        if (mine == best) s_seed = mine_c;                                      // keys are unique
        for (int i = tid; i < FR_HASH; i += FR_T) s_hash[i] = -1;
        if (tid == 0) { s_cnt = 0; s_fail = 0; }
        __syncthreads();
        int32_t* fq = a.out_cells + total;
        if (wave == 0) {
            // ---- grow_frontier (:249-288) by one wave, serially in queue order; xDeltas {-1,-1,-1,1,1,1,0,0}, yDeltas {0,1,-1,0,1,-1,1,-1}
            const int seed = s_seed;
            int head = 0, tail = 1, fail = 0;
            // (the ring holds x | y << 16: a division by the grid's width per queue entry and lane is forty instructions of this lone wave)
            if (lane == 0) { fq[0] = seed; s_ring[0] = (seed % a.W) | ((seed / a.W) << 16); }
            (void)fr_hash_test_and_set(s_hash, seed, lane);
            const int n = lane & 7, qi = lane >> 3;
            const int dx = n < 3 ? -1 : (n < 6 ? 1 : 0);
            const int dy = (n == 1 || n == 4 || n == 6) ? 1 : ((n == 2 || n == 5 || n == 7) ? -1 : 0);
            while (head < tail && !fail) {
                const int nb = min(tail - head, 8);
                int nc = -1, nxy = 0;
                if (qi < nb) {
                    const int q = head + qi;
                    int cx, cy;
                    if (tail - q <= FR_RING) { const int e = s_ring[q & (FR_RING - 1)]; cx = e & 0xFFFF; cy = e >> 16; }
                    else { const int c = __hip_atomic_load(&fq[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); cx = c % a.W; cy = c / a.W; }
                    const int x = cx + dx, y = cy + dy;
                    if (x >= 0 && y >= 0 && x < a.W && y < a.H) { nc = y * a.W + x; nxy = x | (y << 16); }
                }
                const bool isf = nc >= 0 && a.cls[nc] == 2;                     // class 2 does not change while a frontier grows
                unsigned long long m = __ballot(isf);
                m &= nb >= 8 ? ~0ull : ((1ull << (8 * nb)) - 1ull);
                while (m) {
                    const int l = __ffsll((long long)m) - 1;                     // ascending lane = queue order, then neighbour order
                    m &= m - 1ull;
                    const int x = __builtin_amdgcn_readlane(nc, l), xy = __builtin_amdgcn_readlane(nxy, l);
                    const int r = fr_hash_test_and_set(s_hash, x, lane);
                    if (r < 0 || tail >= FR_HASH / 2) { fail = 1; break; }
                    if (r == 0) {
                        if (lane == 0) { __hip_atomic_store(&fq[tail], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); s_ring[tail & (FR_RING - 1)] = xy; }
                        tail += 1;
                    }
                }
                __builtin_amdgcn_wave_barrier();
                head += nb;
            }
            if (lane == 0) { s_cnt = tail; s_fail = fail; }
        }
        __syncthreads();
        if (s_fail) {
            // a frontier too large for the visited set: undo the marks and hand the whole sweep to the one-workgroup form
            for (int i = tid; i < total; i += FR_T) a.cls[a.out_cells[i]] = 2;
            if (tid == 0) a.counts[8] = 1;
            return;
        }
        const int cnt = s_cnt;
        for (int i = tid; i < cnt; i += FR_T) a.cls[__hip_atomic_load(&fq[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)] = 5;
        __threadfence();
        __syncthreads();
        if (nf < a.cap_frontiers) { if (tid == 0) a.out_offsets[nf] = total; } else overflow = 1;
        nf += 1;
        total += cnt;
    }
    if (tid == 0) {
        if (nf <= a.cap_frontiers) a.out_offsets[nf] = total;
        a.counts[0] = nf; a.counts[1] = total; a.counts[4] = overflow;
        a.counts[6] = (int)(wall_clock64() - t_begin);
    }
}


// ---- the same sweep with nothing of a frontier's growth in global memory -----------------------------------------------------
// k_frontier_grow pays a global round trip per step of the growth queue for the classes of the neighbours (a frontier is a thin
// curve: two cells per step; 1.35 us per step, 5.7 ms for the 8 400 cells of an explored disc at 4096^2), resets its visited set per
// frontier and reads cls[] once per live touch and frontier.  Here ALL frontier-class cells of the grid (k_frontier_touches lists
// them: counts[12], fcell[]) sit in ONE LDS set, a word per cell: cell | visited << 31.  "Is the neighbour a frontier cell?" and "has
// it been grown?" are the same lookup -- one lane per (queued cell, neighbour), linear probing from a multiplicative home, one or
// two LDS round trips -- the visited bits outlive the frontier (they are the reference's class mark), and a touch is dead when the
// bit of its cell's slot (found once, in front of the first frontier) is set.  Growth order as before: ascending lane = queue order,
// then neighbour order; lanes that name one cell in the same step are told apart in registers.
// More than FG_CELL_MAX frontier-class cells, or a home region without room: k_frontier_grow takes the sweep (counts[10]).
// (The home: a multiplicative hash alone maps the cells of a column -- an arithmetic progression of stride W -- to an arithmetic
// progression of homes, with a stride of 0.63 slots at W = 10 946, a Fibonacci number: a vertical frontier of 200 cells overflowed
// its home region.  One xor-shift and a second multiplication end that; tests/test_gpu_frontiers.py has the width.)
__device__ __forceinline__ unsigned int fg_home(int c)
{
    unsigned int h = (unsigned int)c * 2654435761u;
    h ^= h >> 15;
    return (h * 0x85EBCA6Bu) >> (32 - 15);                                     // FG_SLOTS = 2^15
}

__global__ __launch_bounds__(FR_T) void k_frontier_grow2(frontier_args a)
{
    extern __shared__ __align__(16) int s_fg[];
    int* s_set = s_fg;
    int* s_ring = s_fg + FG_SLOTS + FG_PAD;
    __shared__ unsigned int s_umin[FR_T / 64];
    __shared__ int s_seed, s_cnt, s_fail;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long t_begin = wall_clock64();
    const int F = a.counts[12], T = a.counts[13];
    if (T > FR_TOUCH_MAX || a.W > 65535 || a.H > 32767) { if (tid == 0) a.counts[8] = 1; return; }      // (the ring packs x | y << 16)
    if (F > FG_CELL_MAX || a.grow_v1 == 1) { if (tid == 0) a.counts[10] = 1; return; }
    for (int i = tid; i < FG_SLOTS + FG_PAD; i += FR_T) s_set[i] = -1;
    if (tid == 0) s_fail = 0;
    __syncthreads();
    for (int i = tid; i < F; i += FR_T) {
        const int c = a.fcell[i];
        const unsigned int h = fg_home(c);
        int d = 0;
#pragma unroll 1
        for (; d < FG_DMAX; ++d)
            if (atomicCAS(&s_set[h + (unsigned int)d], -1, c) == -1) break;
        if (d == FG_DMAX) s_fail = 1;
    }
    __syncthreads();
    if (s_fail || a.grow_v1 == 2) { if (tid == 0) a.counts[10] = 1; return; }           // (2: the tests' way into this exit)
    // ---- the touches of this thread: key and the slot of the cell (every touched cell is a frontier-class cell: it is there)
    unsigned int tk[FR_TOUCH_PER_THREAD]; int ts[FR_TOUCH_PER_THREAD];
#pragma unroll
    for (int j = 0; j < FR_TOUCH_PER_THREAD; ++j) {
        const int i = j * FR_T + tid;
        tk[j] = FR_INF; ts[j] = 0;
        if (i < T) {
            const uint2 t = a.touch[i];
            const unsigned int h = fg_home((int)t.y);
#pragma unroll 1
            for (int d = 0; d < FG_DMAX; ++d) {
                const int sl = (int)(h + (unsigned int)d);
                if (s_set[sl] == (int)t.y) { tk[j] = t.x; ts[j] = sl; break; }
            }
        }
    }
    int nf = 0, total = 0, overflow = 0;
    while (true) {
        // ---- the next seed: the live touch with the smallest key (frontiers.cpp:66-75 meets them in that order)
        unsigned int mine = FR_INF; int mine_s = 0;
#pragma unroll
        for (int j = 0; j < FR_TOUCH_PER_THREAD; ++j) {
            if (tk[j] == FR_INF) continue;
            if (s_set[ts[j]] < 0) { tk[j] = FR_INF; continue; }                  // its frontier has been grown
            if (tk[j] < mine) { mine = tk[j]; mine_s = ts[j]; }
        }
        const unsigned int best = block_min(mine, s_umin);
        if (best == FR_INF) break;
        if (mine == best) s_seed = mine_s;                                      // keys are unique
        __syncthreads();
        int32_t* fq = a.out_cells + total;
        if (wave == 0) {
            // ---- grow_frontier (:249-288) by one wave, serially in queue order; xDeltas {-1,-1,-1,1,1,1,0,0}, yDeltas {0,1,-1,0,1,-1,1,-1}
            const int seed_slot = s_seed;
            const int seed = __builtin_amdgcn_readfirstlane(s_set[seed_slot]);
            int head = 0, tail = 1;
            int last_xy = (seed % a.W) | ((seed / a.W) << 16);                   // the entry at the back of the queue, in a scalar register
            if (lane == 0) { fq[0] = seed; s_ring[0] = last_xy; s_set[seed_slot] = seed | (int)0x80000000; }
            const int n = lane & 7, qi = lane >> 3;
            const int dx = n < 3 ? -1 : (n < 6 ? 1 : 0);
            const int dy = (n == 1 || n == 4 || n == 6) ? 1 : ((n == 2 || n == 5 || n == 7) ? -1 : 0);
            while (head < tail) {
                const int live = tail - head;
                const int nb = min(live, 8);
                int cx = 0, cy = 0;
                if (live == 1) { cx = last_xy & 0xFFFF; cy = last_xy >> 16; }   // a thin curve: the one queued cell is the one just written
                else if (live <= FR_RING) { if (qi < nb) { const int e = s_ring[(head + qi) & (FR_RING - 1)]; cx = e & 0xFFFF; cy = e >> 16; } }
                else if (qi < nb) { const int c = __hip_atomic_load(&fq[head + qi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); cx = c % a.W; cy = c / a.W; }
                const int x = cx + dx, y = cy + dy;
                const bool in = qi < nb && x >= 0 && y >= 0 && x < a.W && y < a.H;
                const int nc = in ? y * a.W + x : -2, nxy = x | (y << 16);      // (-2: no slot ever holds it, empty ones hold -1)
                // the neighbour in the set: a frontier-class cell, grown or not; not there: any other class.  Home slot and the next
                // in one read.  A word EQUAL to the cell is the cell not yet grown (the visited bit is clear) -- and a cell one slot
                // behind its home implies an occupied home (nothing is ever removed): two compares decide the common case.  Both
                // slots taken by others: the chain goes on, lane by lane (rare at a quarter load)
                const unsigned int h = fg_home(nc);
                const int v0 = s_set[h], v1 = s_set[h + 1u];
                int slot = (int)h + (v0 == nc ? 0 : 1);
                bool fresh = v0 == nc || v1 == nc;
                if (__ballot(((unsigned int)v0 > (unsigned int)v1 ? (unsigned int)v0 : (unsigned int)v1) != 0xFFFFFFFFu)) {      // some lane: both slots taken
                    if (in && !fresh && v0 != -1 && v1 != -1 && (v0 & 0x7FFFFFFF) != nc && (v1 & 0x7FFFFFFF) != nc) {
#pragma unroll 1
                        for (int d = 2; d < FG_DMAX; ++d) {
                            const int v = s_set[h + (unsigned int)d];
                            if (v == -1) break;
                            if ((v & 0x7FFFFFFF) == nc) { slot = (int)h + d; fresh = v >= 0; break; }
                        }
                    }
                }
                unsigned long long m = __ballot(fresh);
                // ascending lane = queue order, then neighbour order; later lanes of this step that name the same cell drop out with it
#define FG_TAKE() do { \
                    l_last = __ffsll((long long)m) - 1; \
                    const int xc = __builtin_amdgcn_readlane(nc, l_last); \
                    m &= ~__ballot(nc == xc); \
                    if (lane == l_last) {                                        /* the lane that found it has slot, cell and x | y << 16 */ \
                        s_set[slot] = nc | (int)0x80000000; \
                        __hip_atomic_store(&fq[tail], nc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
                        s_ring[tail & (FR_RING - 1)] = nxy; \
                    } \
                    tail += 1; \
                } while (0)
                if (m) {                                                         // (a taken branch costs this lone wave ten instructions: the first two in line)
                    int l_last;
                    FG_TAKE(); if (m) { FG_TAKE(); while (m) FG_TAKE(); }
                    last_xy = __builtin_amdgcn_readlane(nxy, l_last);
                }
#undef FG_TAKE
                __builtin_amdgcn_wave_barrier();
                head += nb;
            }
            if (lane == 0) s_cnt = tail;
        }
        __syncthreads();
        const int cnt = s_cnt;
        if (nf < a.cap_frontiers) { if (tid == 0) a.out_offsets[nf] = total; } else overflow = 1;
        nf += 1;
        total += cnt;
    }
    if (tid == 0) {
        if (nf <= a.cap_frontiers) a.out_offsets[nf] = total;
        a.counts[0] = nf; a.counts[1] = total; a.counts[4] = overflow;
        a.counts[6] = (int)(wall_clock64() - t_begin);
    }
}

struct bl_frontier_scratch {
    size_t cells = 0;
    uint8_t* cls = nullptr; unsigned int* claim = nullptr; unsigned int* fclaim = nullptr;
    int32_t* queue = nullptr; int32_t* out_cells = nullptr; int32_t* out_offsets = nullptr; int32_t* counts = nullptr;
    int cap_frontiers = 0;
    int32_t* h_counts = nullptr;
    uint2* touch = nullptr;
    uint8_t* nb = nullptr;
    int32_t* fcell = nullptr;
    int form = 0;                     // the last launch: 0 one workgroup with the classes in LDS, 1 one workgroup (large grid), 2 the multi-launch form
};

void bl_frontier_scratch_free(bl_ctx* ctx)
{
    bl_frontier_scratch* s = ctx->frontier;
    if (!s) return;
    void* dev[] = {s->cls, s->claim, s->fclaim, s->queue, s->out_cells, s->out_offsets, s->counts, s->touch, s->nb, s->fcell};
    for (void* q : dev) if (q) (void)hipFree(q);
    if (s->h_counts) (void)hipHostFree(s->h_counts);
    delete s;
    ctx->frontier = nullptr;
}

// The kernels of one find_map_frontiers on ctx's stream, the robot pose from the host or read on the device; nothing waits.
static int frontiers_launch(bl_ctx* ctx, const bl_grid* map, const bl_pose_xyt_t* robot_pose, const bl_pose_xyt_t* d_pose)
{
    const int W = map->frame.width, H = map->frame.height;
    BL_CHECK_ARG(W >= 1 && H >= 1 && (int64_t)W * H < ((int64_t)1 << 28));          // claim keys are 4 * queue position + n
    BL_HIP(hipSetDevice(ctx->device));
    if (!ctx->frontier) ctx->frontier = new bl_frontier_scratch();
    bl_frontier_scratch* s = ctx->frontier;
    const size_t n = (size_t)W * H;
    if (s->cells < n) {
        BL_HIP(hipStreamSynchronize(ctx->stream));
        void* dev[] = {s->cls, s->claim, s->fclaim, s->queue, s->out_cells, s->out_offsets, s->nb};
        for (void* q : dev) if (q) BL_HIP(hipFree(q));
        s->cls = nullptr; s->claim = nullptr; s->fclaim = nullptr; s->queue = nullptr; s->out_cells = nullptr; s->out_offsets = nullptr; s->nb = nullptr;
        s->cells = 0;
        s->cap_frontiers = (int)(n / 4 + 4);               // components are 8-separated: at most one per 2x2 block
        BL_HIP(hipMalloc((void**)&s->cls, n));
        BL_HIP(hipMalloc((void**)&s->nb, n));
        BL_HIP(hipMalloc((void**)&s->claim, n * 4));
        BL_HIP(hipMalloc((void**)&s->fclaim, n * 4));
        BL_HIP(hipMalloc((void**)&s->queue, (n + 1) * 4));
        BL_HIP(hipMalloc((void**)&s->out_cells, n * 4));
        BL_HIP(hipMalloc((void**)&s->out_offsets, ((size_t)s->cap_frontiers + 1) * 4));
        if (!s->counts) BL_HIP(hipMalloc((void**)&s->counts, 16 * 4));
        if (!s->h_counts) BL_HIP(hipHostMalloc((void**)&s->h_counts, 16 * 4, hipHostMallocDefault));
        if (!s->touch) BL_HIP(hipMalloc((void**)&s->touch, (size_t)FR_TOUCH_MAX * sizeof(uint2)));
        if (!s->fcell) BL_HIP(hipMalloc((void**)&s->fcell, (size_t)FG_CELL_MAX * 4));
        s->cells = n;
    }
    frontier_args a;
    a.cells = map->cells; a.W = W; a.H = H;
    a.rx = a.ry = 0; a.d_pose = d_pose; a.frame = map->frame;
    if (!d_pose) bl_global_to_cell((double)robot_pose->x, (double)robot_pose->y, map->frame, &a.rx, &a.ry);      // :39
    a.cls = s->cls; a.claim = s->claim; a.fclaim = s->fclaim; a.queue = s->queue;
    a.out_cells = s->out_cells; a.out_offsets = s->out_offsets; a.cap_frontiers = s->cap_frontiers; a.counts = s->counts;
    a.phase = 0; a.touch = s->touch; a.nb = s->nb; a.fcell = s->fcell;
    static const int grow_v1 = getenv("BOTLAB_FRONTIER_GROW_V1") ? atoi(getenv("BOTLAB_FRONTIER_GROW_V1")) : 0;     // A/B runs and tests of that form
    a.grow_v1 = grow_v1;                                   // 1: k_frontier_grow from the start; 2: k_frontier_grow2 gives up once its set is built
    hipEvent_t e0, e1;
    int rc = bl_timer_begin(ctx, BL_K_FRONTIERS, &e0, &e1);
    if (rc) return rc;
    if (n <= (size_t)FR_CLS_LDS) {
        static unsigned long long attr_set_devices = 0ull;
        const unsigned long long bit = 1ull << (ctx->device & 63);
        if (!(attr_set_devices & bit)) {
            BL_HIP(hipFuncSetAttribute((const void*)k_frontiers<true>, hipFuncAttributeMaxDynamicSharedMemorySize, FR_CLS_LDS));
            attr_set_devices |= bit;
        }
        hipLaunchKernelGGL(k_frontiers<true>, dim3(1), dim3(FR_T), (n + 15) & ~(size_t)15, ctx->stream, a);
        s->form = 0;
    } else {
        long long cblocks = ((long long)n + 255) / 256;
        if (cblocks > 8192) cblocks = 8192;
        hipLaunchKernelGGL(k_frontier_classify, dim3((unsigned int)cblocks), dim3(256), 0, ctx->stream, a);
        static const bool one_wg_sweep = getenv("BOTLAB_FRONTIER_ONE_WG_SWEEP") != nullptr;       // A/B runs and tests of the fallback
        s->form = one_wg_sweep ? 1 : 2;
        if (one_wg_sweep) {
            hipLaunchKernelGGL(k_frontiers<false>, dim3(1), dim3(FR_T), 0, ctx->stream, a);
        } else {
            a.phase = 1;
            static const bool old_flood = getenv("BOTLAB_FRONTIER_OLD_FLOOD") != nullptr;                 // A/B runs
            if (!old_flood && W <= FL_MAX_SIDE && H <= FL_MAX_SIDE) {
                // the flood with its levels' state in LDS (k_frontier_flood): entries are 14-bit coordinates
                static unsigned long long attr_set_devices = 0ull;
                const unsigned long long bit = 1ull << (ctx->device & 63);
                if (!(attr_set_devices & bit)) {
                    BL_HIP(hipFuncSetAttribute((const void*)k_frontier_flood, hipFuncAttributeMaxDynamicSharedMemorySize, FL_LDS_BYTES));
                    attr_set_devices |= bit;
                }
                hipLaunchKernelGGL(k_frontier_nb, dim3((unsigned int)cblocks), dim3(256), 0, ctx->stream, a);
                hipLaunchKernelGGL(k_frontier_flood, dim3(1), dim3(FL_T), FL_LDS_BYTES, ctx->stream, a);
            } else {
                hipLaunchKernelGGL(k_frontiers<false>, dim3(1), dim3(FR_T), 0, ctx->stream, a);
            }
            hipLaunchKernelGGL(k_frontier_touches, dim3((unsigned int)((cblocks + 3) / 4)), dim3(256), 0, ctx->stream, a);       // (four cells per thread)
            {
                static unsigned long long attr_set_devices = 0ull;
                const unsigned long long bit = 1ull << (ctx->device & 63);
                if (!(attr_set_devices & bit)) {
                    BL_HIP(hipFuncSetAttribute((const void*)k_frontier_grow2, hipFuncAttributeMaxDynamicSharedMemorySize, FG_LDS_BYTES));
                    attr_set_devices |= bit;
                }
            }
            hipLaunchKernelGGL(k_frontier_grow2, dim3(1), dim3(FR_T), FG_LDS_BYTES, ctx->stream, a);
            hipLaunchKernelGGL(k_frontier_grow, dim3(1), dim3(FR_T), 0, ctx->stream, a);         // (returns at once unless counts[10] is set)
            a.phase = 2;
            hipLaunchKernelGGL(k_frontiers<false>, dim3(1), dim3(FR_T), 0, ctx->stream, a);
        }
    }
    BL_HIP(hipGetLastError());
    rc = bl_timer_end(ctx, BL_K_FRONTIERS, e0, e1);
    if (rc) return rc;
    BL_HIP(hipMemcpyAsync(s->h_counts, s->counts, 16 * 4, hipMemcpyDeviceToHost, ctx->stream));
    return BL_OK;
}

// ... and their result, once the stream has run them (frontiers.cpp:66-72: the length filter; cell centres as global points)
static int frontiers_collect(bl_ctx* ctx, const bl_frame& frame, double min_frontier_length, bl_frontiers** out)
{
    bl_frontier_scratch* s = ctx->frontier;
    const int W = frame.width;
    BL_HIP(hipStreamSynchronize(ctx->stream));
    const int nf = s->h_counts[0], total = s->h_counts[1];
    if (getenv("BOTLAB_FRONTIER_STAMPS"))
        fprintf(stderr, "[frontiers] flood %.3f ms (%d cells, %d levels), frontier sweep %.3f ms (%d frontiers, %d cells; %d touches%s)\n",
                s->h_counts[5] * 1e-5, s->h_counts[2], s->h_counts[3], s->h_counts[6] * 1e-5, nf, total, s->h_counts[13],
                s->h_counts[8] ? ", one-workgroup sweep" : (s->h_counts[10] ? ", k_frontier_grow" : ""));
    if (s->h_counts[4] || nf > s->cap_frontiers) { bl_set_error("internal: frontier table overflow (%d frontiers)", nf); return BL_ERR_CAPACITY; }
    std::vector<int32_t> offs((size_t)nf + 1), cells((size_t)total);
    BL_HIP(hipMemcpy(offs.data(), s->out_offsets, ((size_t)nf + 1) * 4, hipMemcpyDeviceToHost));
    if (total > 0) BL_HIP(hipMemcpy(cells.data(), s->out_cells, (size_t)total * 4, hipMemcpyDeviceToHost));
    bl_frontiers* f = new bl_frontiers();
    f->bfs_cells = s->h_counts[2]; f->bfs_levels = s->h_counts[3];
    f->sweep_kernel = s->form < 2 ? s->form : (s->h_counts[8] ? 1 : (s->h_counts[10] ? 2 : 3));
    f->offsets.push_back(0);
    for (int k = 0; k < nf; ++k) {
        const int cnt = offs[k + 1] - offs[k];
        // f.cells.size() * map.metersPerCell() >= minFrontierLength: size_t * float -> float, compared as double (:69)
        if (!((double)((float)(size_t)cnt * frame.mpc) >= min_frontier_length)) continue;
        for (int i = offs[k]; i < offs[k + 1]; ++i) {
            const int cx = cells[i] % W, cy = cells[i] / W;
            // grid_position_to_global_position(Point<int>) narrowed to Point<float> (grid_utils.hpp:14-19, frontiers.cpp:268)
            f->xy.push_back((float)((double)frame.ox + (double)cx * (double)frame.mpc));
            f->xy.push_back((float)((double)frame.oy + (double)cy * (double)frame.mpc));
        }
        f->offsets.push_back((int32_t)(f->xy.size() / 2));
    }
    *out = f;
    return BL_OK;
}

extern "C" int bl_frontiers_find(bl_ctx* ctx, const bl_grid* map, const bl_pose_xyt_t* robot_pose, double min_frontier_length,
                                 bl_frontiers** out)
{
    BL_CHECK_ARG(ctx != nullptr && map != nullptr && robot_pose != nullptr && out != nullptr);
    BL_CHECK_ARG(map->ctx == ctx);
    int rc = frontiers_launch(ctx, map, robot_pose, nullptr);
    if (rc) return rc;
    return frontiers_collect(ctx, map->frame, min_frontier_length, out);
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
extern "C" int bl_frontiers_debug_sweep_kernel(const bl_frontiers* f) { return f ? f->sweep_kernel : -1; }
extern "C" void bl_frontiers_destroy(bl_frontiers* f) { delete f; }

// =============================================================================================== the exploration step on side streams
// Exploration::executeExploringMap (src/planning/exploration.cpp:277-369) consumes every map the SLAM process publishes: setMap
// (distance transform), find_map_frontiers, and -- when the robot is within 0.5 m of its target or has none -- plan_path_to_frontier.
// bl_explorer is that arrangement on one device, as the replanner (bl_planner) is for a fixed goal: a submission snapshots the map
// and the device-resident pose on the SLAM stream; a lane (a ctx with its own stream, distance grid and frontier scratch) runs
// setDistances + the frontier kernels against the snapshot while the SLAM stream goes on; the fetch -- in submission order -- takes
// the frontier lists to the host, applies the re-planning rule with the state the steps share (currentTarget_, currentPath_) and,
// when a plan is due, runs plan_path_to_frontier on that lane.  One flood is one workgroup: the lanes' floods run side by side.
#define EXPLORER_MAX_LANES 16
struct explorer_lane {
    bl_ctx* ctx; bl_dist* dist; bl_grid* snap;
    bl_pose_xyt_t* d_pose; bl_pose_xyt_t* h_pose;      // the snapshot's pose: device, and pinned host copy
    hipEvent_t snap_ready, t0, t1;
    bool busy;
};
struct bl_explorer {
    bl_ctx* main;
    int lanes, next;
    explorer_lane lane[EXPLORER_MAX_LANES];
    std::deque<int>* order;                             // lanes with a submission, oldest first
    bl_motion_planner_t planner;                        // MotionPlanner: robot radius, search parameters, num_frontiers, prev_goal
    bl_pose_xyt_t target;                               // currentTarget_
    std::vector<bl_pose_xyt_t>* path;                   // currentPath_
    bl_frontiers* last;                                 // frontiers_ of the last fetched step
    double min_frontier_length;
    // submit (the SLAM thread) and fetch (possibly another thread: the reference's exploration PROCESS) share `order` and the lanes'
    // busy flags; a lane's other members belong to submit while it is idle and to fetch while it is busy
    std::mutex* mu;
    // the exploration state (planner, target, path, last): written by fetch, read and written by bl_explorer_set_state /
    // bl_explorer_frontiers from any thread.  Never held across a plan (seconds): fetch copies the planner out and its results in
    std::mutex* smu;
};

extern "C" void bl_explorer_destroy(bl_explorer* e);

extern "C" int bl_explorer_create(bl_ctx* ctx, int lanes, double robot_radius, bl_explorer** out)
{
    BL_CHECK_ARG(ctx != nullptr && out != nullptr && lanes >= 1 && lanes <= EXPLORER_MAX_LANES && robot_radius > 0.0);
    BL_HIP(hipSetDevice(ctx->device));
    bl_explorer* e = new bl_explorer();
    memset((void*)e, 0, sizeof(*e));
    e->main = ctx; e->lanes = lanes;
    e->order = new std::deque<int>();
    e->mu = new std::mutex();
    e->smu = new std::mutex();
    e->path = new std::vector<bl_pose_xyt_t>();
    e->min_frontier_length = 0.35;                      // kMinFrontierLength's default (frontiers.hpp:36)
    // MotionPlanner(params) + setParams (motion_planner.cpp:9-16, 105-110): minDist = robotRadius, maxDist = 10 minDist, exponent 1
    e->planner.robot_radius = robot_radius;
    e->planner.search.minDistanceToObstacle = robot_radius;
    e->planner.search.maxDistanceWithCost = 10.0 * robot_radius;
    e->planner.search.distanceCostExponent = 1.0;
    e->planner.num_frontiers = 1;
    e->planner.prev_goal.x = 1e9f; e->planner.prev_goal.y = 1e9f;   // never set by the reference's exploration loop (D5)
    // (a failure below hands the partly built explorer to bl_explorer_destroy, which tolerates missing members: nothing leaks)
    auto build_lane = [&](explorer_lane& L) -> int {
        // a lane's stream has the lowest priority (BOTLAB_EXPLORER_NORMAL_PRIORITY=1: the default one): a plan to a frontier is one
        // kernel of up to seconds, and streams of one priority share the runtime's four hardware queues -- with its stream behind
        // a lane's on one queue the SLAM loop ran at 182 instead of 3 700 steps/s (4096 x 4096, the explorer on every newest map)
        int rc = getenv("BOTLAB_EXPLORER_NORMAL_PRIORITY") ? bl_ctx_create(ctx->device, nullptr, &L.ctx) : bl_ctx_create_low_priority(ctx->device, &L.ctx);
        if (rc) return rc;
        L.ctx->astar_small_lds = true;                  // its searches co-run with the SLAM stream's kernels
        rc = bl_dist_create(L.ctx, &L.dist);
        if (rc) return rc;
        BL_HIP(hipMalloc((void**)&L.d_pose, sizeof(bl_pose_xyt_t)));
        BL_HIP(hipHostMalloc((void**)&L.h_pose, sizeof(bl_pose_xyt_t), hipHostMallocDefault));
        BL_HIP(hipEventCreateWithFlags(&L.snap_ready, hipEventDisableTiming));
        BL_HIP(hipEventCreate(&L.t0));
        BL_HIP(hipEventCreate(&L.t1));
        return BL_OK;
    };
    for (int l = 0; l < lanes; ++l) {
        const int rc = build_lane(e->lane[l]);
        if (rc) { bl_explorer_destroy(e); return rc; }
    }
    *out = e;
    return BL_OK;
}

extern "C" void bl_explorer_destroy(bl_explorer* e)
{
    if (!e) return;
    (void)hipStreamSynchronize(e->main->stream);
    for (int l = 0; l < e->lanes; ++l) {
        explorer_lane& L = e->lane[l];
        if (!L.ctx) continue;
        (void)hipStreamSynchronize(L.ctx->stream);
        if (L.snap) bl_grid_destroy(L.snap);
        if (L.dist) bl_dist_destroy(L.dist);
        if (L.d_pose) (void)hipFree(L.d_pose);
        if (L.h_pose) (void)hipHostFree(L.h_pose);
        if (L.snap_ready) (void)hipEventDestroy(L.snap_ready);
        if (L.t0) (void)hipEventDestroy(L.t0);
        if (L.t1) (void)hipEventDestroy(L.t1);
        bl_ctx_destroy(L.ctx);
    }
    if (e->last) bl_frontiers_destroy(e->last);
    delete e->order; delete e->path; delete e->mu; delete e->smu;
    delete e;
}

extern "C" int bl_explorer_set_state(bl_explorer* e, const bl_pose_xyt_t* target, const bl_pose_xyt_t* prev_goal)
{
    BL_CHECK_ARG(e != nullptr);
    std::lock_guard<std::mutex> g(*e->smu);             // (fetch, on another thread, reads and writes the same state under it)
    if (target) e->target = *target;
    if (prev_goal) e->planner.prev_goal = *prev_goal;
    return BL_OK;
}

extern "C" int bl_explorer_submit(bl_explorer* e, const bl_grid* map, const void* d_pose)
{
    BL_CHECK_ARG(e != nullptr && map != nullptr && d_pose != nullptr && map->ctx == e->main);
    explorer_lane& L = e->lane[e->next];
    {
        std::lock_guard<std::mutex> g(*e->mu);
        if (L.busy) { bl_set_error("every explorer lane holds a submission: fetch first"); return BL_ERR_STATE; }
    }
    BL_HIP(hipSetDevice(e->main->device));
    if (L.snap && (L.snap->frame.width != map->frame.width || L.snap->frame.height != map->frame.height)) { bl_grid_destroy(L.snap); L.snap = nullptr; }
    if (!L.snap) {
        int rc = bl_grid_create(L.ctx, map->frame.width, map->frame.height, map->frame.mpc, map->frame.cpm, map->frame.ox, map->frame.oy, &L.snap);
        if (rc) return rc;
        BL_HIP(hipStreamSynchronize(L.ctx->stream));    // its zero-fill ran on the lane's stream
    }
    L.snap->frame = map->frame;
    L.snap->mirror_valid = false;
    // (the lane is idle: its last submission has been fetched, and a fetch leaves nothing behind on the lane's stream)
    int rc = bl_snapshot_enqueue(e->main, map, L.snap, d_pose, L.d_pose, nullptr, nullptr, 0ull);
    if (rc) return rc;
    BL_HIP(hipEventRecord(L.snap_ready, e->main->stream));
    BL_HIP(hipStreamWaitEvent(L.ctx->stream, L.snap_ready, 0));
    rc = bl_dist_set_distances(L.dist, L.snap);         // planner_.setMap(currentMap_) (:299)
    if (rc) return rc;
    BL_HIP(hipEventRecord(L.t0, L.ctx->stream));
    rc = frontiers_launch(L.ctx, L.snap, nullptr, L.d_pose);   // find_map_frontiers(currentMap_, currentPose_) (:300)
    if (rc) return rc;
    BL_HIP(hipEventRecord(L.t1, L.ctx->stream));
    BL_HIP(hipMemcpyAsync(L.h_pose, L.d_pose, sizeof(bl_pose_xyt_t), hipMemcpyDeviceToHost, L.ctx->stream));
    {
        std::lock_guard<std::mutex> g(*e->mu);
        L.busy = true;
        e->order->push_back(e->next);
    }
    e->next = (e->next + 1) % e->lanes;
    return BL_OK;
}

// submissions not handed back yet (the one a fetch is working on included)
extern "C" int bl_explorer_pending(const bl_explorer* e)
{
    if (!e) return 0;
    std::lock_guard<std::mutex> g(*e->mu);
    int n = 0;
    for (int l = 0; l < e->lanes; ++l) n += e->lane[l].busy ? 1 : 0;
    return n;
}

extern "C" int bl_explorer_fetch(bl_explorer* e, bl_explore_result_t* out, bl_pose_xyt_t* out_path, int cap)
{
    BL_CHECK_ARG(e != nullptr && out != nullptr && (cap == 0 || out_path != nullptr) && cap >= 0);
    int l;
    {
        std::lock_guard<std::mutex> g(*e->mu);
        if (e->order->empty()) { bl_set_error("no exploration step pending"); return BL_ERR_STATE; }
        l = e->order->front();
        e->order->pop_front();
    }
    explorer_lane& L = e->lane[l];
    // the lane is idle again when this call returns, whatever way
    struct release { bl_explorer* e; explorer_lane* L; ~release() { std::lock_guard<std::mutex> g(*e->mu); L->busy = false; } } rel{e, &L};
    BL_HIP(hipSetDevice(e->main->device));
    memset(out, 0, sizeof(*out));
    bl_frontiers* fr = nullptr;
    int rc = frontiers_collect(L.ctx, L.snap->frame, e->min_frontier_length, &fr);      // (waits for the lane's stream)
    if (rc) return rc;
    float fms = 0.0f;
    (void)hipEventElapsedTime(&fms, L.t0, L.t1);
    const bl_pose_xyt_t pose = *L.h_pose;
    const int nf = (int)fr->offsets.size() - 1;
    bl_motion_planner_t planner;
    bl_pose_xyt_t target;
    {
        std::lock_guard<std::mutex> g(*e->smu);
        if (e->last) bl_frontiers_destroy(e->last);
        e->last = new bl_frontiers(*fr);                                                 // (the plan below reads fr itself, outside the lock)
        e->planner.num_frontiers = nf;                                                   // planner_.setNumFrontiers (:302)
        planner = e->planner; target = e->target;
    }
    struct drop { bl_frontiers* f; ~drop() { bl_frontiers_destroy(f); } } drop_fr{fr};
    // :307-311 -- sqrt(pow(dx, 2) + pow(dy, 2)) in double from float differences, stored to a float
    float currDist = 0.0f;
    if (target.x != 0 || target.y != 0) {
        const double dx = (double)(pose.x - target.x), dy = (double)(pose.y - target.y);
        currDist = (float)std::sqrt(dx * dx + dy * dy);
    }
    int64_t st[3] = {0, 0, 0};
    double plan_ms = 0.0;
    if (currDist <= 0.5f && nf > 0) {                                                    // :316-321
        const auto w0 = std::chrono::steady_clock::now();
        std::vector<bl_pose_xyt_t> buf((size_t)1 << 16);
        int len = 0;
        bl_pose_xyt_t goal;
        rc = bl_plan_path_to_frontier(L.ctx, fr, &pose, L.dist, &planner, buf.data(), (int)buf.size(), &len, &goal, st);
        if (rc) return rc;
        if (len > (int)buf.size()) { bl_set_error("path of %d poses does not fit", len); return BL_ERR_CAPACITY; }
        std::lock_guard<std::mutex> g(*e->smu);
        e->path->assign(buf.begin(), buf.begin() + len);
        if (len > 1) e->target = (*e->path)[(size_t)len - 1];
        out->planned = 1;
        plan_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count();
    }
    std::lock_guard<std::mutex> gs(*e->smu);
    // :335-368 (D10: the status the reference leaves unset when frontiers remain but no path was found is FAILED)
    const int path_len = (int)e->path->size();
    out->status = nf == 0 ? 1 : (path_len > 1 ? 0 : 2);                                   // COMPLETE / IN_PROGRESS / FAILED
    out->next_state = out->status == 0 ? 1 : (out->status == 1 ? 2 : 4);                  // EXPLORING_MAP / RETURNING_HOME / FAILED_EXPLORATION
    out->num_frontiers = nf;
    out->frontier_cells = (int)(fr->xy.size() / 2);
    out->path_length = path_len;
    out->pops = st[0]; out->pushes = st[1]; out->searches = st[2];
    out->bfs_cells = fr->bfs_cells; out->bfs_levels = fr->bfs_levels;
    out->pose = pose; out->target = e->target;
    out->frontiers_ms = fms; out->plan_ms = (float)plan_ms;
    for (int i = 0; i < path_len && i < cap; ++i) out_path[i] = (*e->path)[(size_t)i];
    return BL_OK;
}

// frontiers_ of the last fetched step (a copy the caller owns)
extern "C" int bl_explorer_frontiers(const bl_explorer* e, bl_frontiers** out)
{
    BL_CHECK_ARG(e != nullptr && out != nullptr);
    std::lock_guard<std::mutex> g(*e->smu);
    if (!e->last) { bl_set_error("no exploration step fetched yet"); return BL_ERR_STATE; }
    *out = new bl_frontiers(*e->last);
    return BL_OK;
}
