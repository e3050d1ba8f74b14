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
                                // [5], [6] time stamps, [7] touches found (k_frontier_touches), [8] "take the one-workgroup sweep" flag
    int phase;                  // k_frontiers: 0 flood + sweep (small grids), 1 flood only, 2 sweep only (and only if counts[8] is set)
    uint2* touch;               // (key, cell) of every frontier cell the flood touched, in no order; FR_TOUCH_MAX entries
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
    a.claim[c] = FR_INF;
    a.fclaim[c] = FR_INF;
}

__global__ __launch_bounds__(256) void k_frontier_classify(frontier_args a)
{
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
        if (tid == 0) { a.counts[2] = qn; a.counts[3] = levels; a.counts[5] = (int)(t_flood - t_begin); a.counts[7] = 0; a.counts[8] = 0; }
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

__global__ __launch_bounds__(256) void k_frontier_touches(frontier_args a)
{
    const long long ncell = (long long)a.W * a.H;
    for (long long c = (long long)blockIdx.x * 256 + threadIdx.x; c < ncell; c += (long long)gridDim.x * 256) {
        if (a.cls[c] != 2) continue;
        const unsigned int key = a.claim[c];
        if (key == FR_INF) continue;
        const int at = atomicAdd(&a.counts[7], 1);
        if (at < FR_TOUCH_MAX) a.touch[at] = make_uint2(key, (unsigned int)c);
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
    const int T = a.counts[7];
    if (T > FR_TOUCH_MAX) { if (tid == 0) a.counts[8] = 1; return; }
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
            if (lane == 0) { fq[0] = seed; s_ring[0] = seed; }
            (void)fr_hash_test_and_set(s_hash, seed, lane);
            const int n = lane & 7, qi = lane >> 3;
            const int dx = n < 3 ? -1 : (n < 6 ? 1 : 0);
            const int dy = (n == 1 || n == 4 || n == 6) ? 1 : ((n == 2 || n == 5 || n == 7) ? -1 : 0);
            while (head < tail && !fail) {
                const int nb = min(tail - head, 8);
                int nc = -1;
                if (qi < nb) {
                    const int q = head + qi;
                    int c;
                    if (tail - q <= FR_RING) c = s_ring[q & (FR_RING - 1)];
                    else c = __hip_atomic_load(&fq[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const int x = c % a.W + dx, y = c / a.W + dy;
                    if (x >= 0 && y >= 0 && x < a.W && y < a.H) nc = y * a.W + x;
                }
                const bool isf = nc >= 0 && a.cls[nc] == 2;                     // class 2 does not change while a frontier grows
                unsigned long long m = __ballot(isf);
                m &= nb >= 8 ? ~0ull : ((1ull << (8 * nb)) - 1ull);
                while (m) {
                    const int l = __ffsll((long long)m) - 1;                     // ascending lane = queue order, then neighbour order
                    m &= m - 1ull;
                    const int x = __builtin_amdgcn_readlane(nc, l);
                    const int r = fr_hash_test_and_set(s_hash, x, lane);
                    if (r < 0 || tail >= FR_HASH / 2) { fail = 1; break; }
                    if (r == 0) {
                        if (lane == 0) { __hip_atomic_store(&fq[tail], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); s_ring[tail & (FR_RING - 1)] = x; }
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

struct bl_frontier_scratch {
    size_t cells = 0;
    uint8_t* cls = nullptr; unsigned int* claim = nullptr; unsigned int* fclaim = nullptr;
    int32_t* queue = nullptr; int32_t* out_cells = nullptr; int32_t* out_offsets = nullptr; int32_t* counts = nullptr;
    int cap_frontiers = 0;
    int32_t* h_counts = nullptr;
    uint2* touch = nullptr;
};

void bl_frontier_scratch_free(bl_ctx* ctx)
{
    bl_frontier_scratch* s = ctx->frontier;
    if (!s) return;
    void* dev[] = {s->cls, s->claim, s->fclaim, s->queue, s->out_cells, s->out_offsets, s->counts, s->touch};
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
        if (!s->counts) BL_HIP(hipMalloc((void**)&s->counts, 16 * 4));
        if (!s->h_counts) BL_HIP(hipHostMalloc((void**)&s->h_counts, 16 * 4, hipHostMallocDefault));
        if (!s->touch) BL_HIP(hipMalloc((void**)&s->touch, (size_t)FR_TOUCH_MAX * sizeof(uint2)));
        s->cells = n;
    }
    frontier_args a;
    a.cells = map->cells; a.W = W; a.H = H;
    bl_global_to_cell((double)robot_pose->x, (double)robot_pose->y, map->frame, &a.rx, &a.ry);      // :39
    a.cls = s->cls; a.claim = s->claim; a.fclaim = s->fclaim; a.queue = s->queue;
    a.out_cells = s->out_cells; a.out_offsets = s->out_offsets; a.cap_frontiers = s->cap_frontiers; a.counts = s->counts;
    a.phase = 0; a.touch = s->touch;
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
        long long cblocks = ((long long)n + 255) / 256;
        if (cblocks > 8192) cblocks = 8192;
        hipLaunchKernelGGL(k_frontier_classify, dim3((unsigned int)cblocks), dim3(256), 0, ctx->stream, a);
        static const bool one_wg_sweep = getenv("BOTLAB_FRONTIER_ONE_WG_SWEEP") != nullptr;       // A/B runs and tests of the fallback
        if (one_wg_sweep) {
            hipLaunchKernelGGL(k_frontiers<false>, dim3(1), dim3(FR_T), 0, ctx->stream, a);
        } else {
            a.phase = 1;
            hipLaunchKernelGGL(k_frontiers<false>, dim3(1), dim3(FR_T), 0, ctx->stream, a);
            hipLaunchKernelGGL(k_frontier_touches, dim3((unsigned int)cblocks), dim3(256), 0, ctx->stream, a);
            hipLaunchKernelGGL(k_frontier_grow, dim3(1), dim3(FR_T), 0, ctx->stream, a);
            a.phase = 2;
            hipLaunchKernelGGL(k_frontiers<false>, dim3(1), dim3(FR_T), 0, ctx->stream, a);
        }
    }
    BL_HIP(hipGetLastError());
    rc = bl_timer_end(ctx, BL_K_FRONTIERS, e0, e1);
    if (rc) return rc;
    BL_HIP(hipMemcpyAsync(s->h_counts, s->counts, 16 * 4, hipMemcpyDeviceToHost, ctx->stream));
    BL_HIP(hipStreamSynchronize(ctx->stream));
    const int nf = s->h_counts[0], total = s->h_counts[1];
    if (getenv("BOTLAB_FRONTIER_STAMPS"))
        fprintf(stderr, "[frontiers] flood %.3f ms (%d cells, %d levels), frontier sweep %.3f ms (%d frontiers, %d cells; %d touches%s)\n",
                s->h_counts[5] * 1e-5, s->h_counts[2], s->h_counts[3], s->h_counts[6] * 1e-5, nf, total, s->h_counts[7],
                s->h_counts[8] ? ", one-workgroup sweep" : "");
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
