// bl_mcl_finish.h -- the end of a particle-filter update (weight-unit prefix + estimatePosteriorPose, particle_filter.cpp:116-160)
// as device functions, so that it can ride in another kernel's launch: k_map_update runs it beside Mapping::updateMap
// (bl_mapping_update_finishing_pf), which needs the pose estimate and nothing else of it.  k_mcl_finish_prefix (bl_mcl.hip) is
// the stand-alone launch of the same arithmetic; both give bit-identical prefix, unit total and pose.
#ifndef BL_MCL_FINISH_H
#define BL_MCL_FINISH_H

#include "bl_internal.h"

struct pf_state {
    double S;                 // total weight units of rec[cur]
    bl_pose_xyt_t pose;       // posteriorPose_
    double sums_used[5];      // the sums the estimate was formed from (diagnostic)
};

// partials[b][5]: per k_mcl_main workgroup b the sums of units, units*x, units*y, units*sin(theta), units*cos(theta).
// Workgroups [0, main_blocks) of that launch own `tile` particles each from 0 on (clipped to main_particles), the rest own
// `tail_tile` particles each from main_particles on.
struct mcl_finish_args {
    const double* partials; int nblocks;
    const float4* rec; int N;
    int tile, main_blocks, main_particles, tail_tile;
    unsigned long long* prefix;
    pf_state* state;
    int64_t utime;
};

#define MCLF_POSE_THREADS 256                 // the estimate's addition order is that of a 256-thread workgroup, whoever runs it
#define MCLF_WG 1024                          // prefix workgroups of the riding form
#define MCLF_ITEMS 8
#define MCLF_CHUNK (MCLF_WG * MCLF_ITEMS)     // particles per prefix workgroup: a whole number of k_mcl_main tiles (tiles are powers of two <= 1024)

// prefix workgroups the riding form needs
static inline int mclf_groups(const mcl_finish_args& f)
{
    const int tpc_main = MCLF_CHUNK / f.tile;
    const int tail_blocks = f.nblocks - f.main_blocks;
    int g = (f.main_blocks + tpc_main - 1) / tpc_main;
    if (tail_blocks > 0) { const int tpc_tail = MCLF_CHUNK / f.tail_tile; g += (tail_blocks + tpc_tail - 1) / tpc_tail; }
    return g;
}

#if defined(__HIPCC__)
__device__ __forceinline__ double mclf_wave_sum(double v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// The five sums in a fixed order (thread-strided over the partials with a stride of 256, wave shuffles, waves in order) and
// the estimate.  Called by EVERY thread of a workgroup of >= 256 threads (it contains a barrier); threads beyond the first
// 256 only take part in the barrier.  s_red: shared double[4][5].  s_pose_out (shared memory, optional) receives the estimate too.
__device__ __forceinline__ void mclf_pose(const mcl_finish_args& f, double (*s_red)[5], bl_pose_xyt_t* s_pose_out = nullptr)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < MCLF_POSE_THREADS) {
        double v[5] = {0, 0, 0, 0, 0};
        for (int b = tid; b < f.nblocks; b += MCLF_POSE_THREADS)
            for (int k = 0; k < 5; ++k) v[k] += f.partials[(size_t)b * 5 + k];
        for (int k = 0; k < 5; ++k) v[k] = mclf_wave_sum(v[k]);
        if (lane == 0) for (int k = 0; k < 5; ++k) s_red[wave][k] = v[k];
    }
    __syncthreads();
    if (tid == 0) {
        double tot[5] = {0, 0, 0, 0, 0};
        for (int w = 0; w < MCLF_POSE_THREADS / 64; ++w) for (int k = 0; k < 5; ++k) tot[k] += s_red[w][k];
        f.state->S = tot[0];                                 // the unit total: an exact integer below 2^53, any order gives it
        bl_pose_xyt_t p;
        p.utime = f.utime;
        p.x = (float)(tot[1] / tot[0]);
        p.y = (float)(tot[2] / tot[0]);
        p.theta = (float)atan2(tot[3], tot[4]);
        f.state->pose = p;
        if (s_pose_out) *s_pose_out = p;                     // shared memory: the caller's workgroup reads it behind its next barrier
        for (int k = 0; k < 5; ++k) f.state->sums_used[k] = tot[k];
    }
}

// Prefix workgroup g of the riding form: MCLF_WG threads, MCLF_CHUNK particles, exact integers throughout.
// s_u64: shared unsigned long long[2 * MCLF_WG / 64].
__device__ __forceinline__ void mclf_prefix_group(const mcl_finish_args& f, int g, unsigned long long* s_u64)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned long long* s_off = s_u64;
    unsigned long long* s_wave = s_u64 + MCLF_WG / 64;
    const int tpc_main = MCLF_CHUNK / f.tile;
    const int main_groups = (f.main_blocks + tpc_main - 1) / tpc_main;
    int first_block, lo, hi;
    if (g < main_groups) {
        first_block = g * tpc_main;
        lo = first_block * f.tile;
        hi = min(f.main_particles, lo + MCLF_CHUNK);
    } else {
        const int gt = g - main_groups;
        first_block = f.main_blocks + gt * (MCLF_CHUNK / f.tail_tile);
        lo = f.main_particles + gt * MCLF_CHUNK;
        hi = min(f.N, lo + MCLF_CHUNK);
    }
    // units of everything before this chunk: the unit sums of the k_mcl_main workgroups before it
    unsigned long long before = 0;
    for (int j = tid; j < first_block; j += MCLF_WG) before += (unsigned long long)f.partials[(size_t)j * 5];
    for (int off = 32; off > 0; off >>= 1) before += __shfl_xor(before, off, 64);
    if (lane == 0) s_off[wave] = before;
    const int base = lo + tid * MCLF_ITEMS;
    unsigned int u[MCLF_ITEMS];
#pragma unroll
    for (int k = 0; k < MCLF_ITEMS; ++k) u[k] = (base + k < hi) ? __float_as_uint(f.rec[base + k].w) : 0u;
    unsigned long long loc[MCLF_ITEMS];
    unsigned long long run = 0;
#pragma unroll
    for (int k = 0; k < MCLF_ITEMS; ++k) { run += u[k]; loc[k] = run; }
    unsigned long long incl = run;
    for (int off = 1; off < 64; off <<= 1) {
        unsigned long long t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    unsigned long long off0 = incl - run;
    for (int w = 0; w < MCLF_WG / 64; ++w) { off0 += s_off[w]; if (w < wave) off0 += s_wave[w]; }
#pragma unroll
    for (int k = 0; k < MCLF_ITEMS; ++k)
        if (base + k < hi) f.prefix[base + k] = off0 + loc[k];
}
#endif  // __HIPCC__

// bl_mcl.hip: if `pf` has an update begun (bl_pf_update_begin) whose end can ride in another launch (the whole particle set
// on this device, the partial-sum form of the finish), does the end-of-update bookkeeping, fills `out` and returns 1; the
// caller MUST then launch the finish.  Returns 0 when there is nothing to take (no update pending: the robot did not move)
// and a negative status when the pending update cannot ride (the caller falls back to bl_pf_update_end).
int bl_pf_take_finish(bl_pf* pf, mcl_finish_args* out);
bl_ctx* bl_pf_ctx(bl_pf* pf);

#endif
