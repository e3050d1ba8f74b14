// bl_mcl.hip -- ParticleFilter (src/slam/particle_filter.cpp) with ActionModel (src/slam/action_model.cpp) and
// SensorModel (src/slam/sensor_model.cpp) as gfx950 kernels.
//
// Data layout in HBM (per filter):
//   rec[2]   float4[N]   exchange record of ALL N particles, double-buffered: (x, y, theta, weight-units as uint32 bits).
//                        rec[cur] is the posterior of the previous update; k_mcl_main writes rec[cur^1].
//   prefix   uint64[N]   inclusive prefix sum of the weight-units of rec[cur] (exact integers)
//   parent   float4[n]   parent pose of the local shard (x, y, theta, -)
//   state    pf_state    S (sum of units), pose estimate
//
// Weights as exact integers.  A raw particle weight in the reference is max(likelihood, 0.001) with likelihood a sum
// of terms k or k/2, k an int8 log-odds (sensor_model.cpp:40-57, particle_filter.cpp:125-133).  In units of 0.0005
// every raw weight is an integer: 1000 * (2*likelihood) or 2.  Sums and prefix sums of units are exact in uint64, so
// the normalisation, the resampling cumulative and the result are independent of launch shape and GPU count.
//
// Resampling rule (particle_filter.cpp:84-103): output particle m takes the first source i whose cumulative normalised
// weight c_i satisfies U_m <= c_i, U_m = r + m/N.  Here: first i with U_m * S <= prefix[i] (double product against an
// exact integer), i clamped to N-1.  The reference accumulates c_i sequentially in double; the two rules can differ
// only when U_m lies within rounding distance of a partial sum (DESIGN.md "Resampling").
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <new>

#include <hip/hip_ext.h>
#include "bl_internal.h"
#include "bl_mcl_finish.h"

#define MCL_LDS_RAYS 384                      // rays whose (range, theta, cos theta, sin theta) table is staged in LDS at a time (longer scans: several passes)
#define MCL_MIN_BLOCKS 512                    // split rays over lanes until the launch has at least this many workgroups (2 per CU)
#define MCL_WIN_SMALL_BYTES (64 * 1024)       // whole-grid staging budget (200x200 int8 framed = 41 KB -> three workgroups per CU)
#define MCL_WIN_MAX 208                       // window side: 208^2 = 42 KB, three workgroups per CU like the whole-grid image of a 200x200 map
#define MCL_WIN_MARGIN 24                     // cells added to the scan's reach on every side of the window for the spread of the cloud
#define SCAN_THREADS 256
#define SCAN_ITEMS 2                          // tiles of 512 particles: a whole number of them makes a finish group (bl_mcl_finish.h)
#define SCAN_TILE (SCAN_THREADS * SCAN_ITEMS)

// pf_state: bl_mcl_finish.h

struct bl_pf {
    bl_ctx* ctx;
    int N, lo, hi, n_local;
    float4* rec[2];
    bool rec_external;
    int cur;
    double* tile_partials;    // [scan_blocks][5]: per-tile sums of units, units*(x, y, sin, cos) from the record (record-based finish)
    ss_rec* fin_recs;         // [2][fin_subs_cap]: sub-tile records of the pose sums (bl_mcl_finish.h)
    ss_wild* fin_wild;        // [2][fin_subs_cap]: wild maps of the sub-tiles whose sum is predicted to cross binades
    int fin_subs_cap;
    mclf_tab_elem* fin_tabs;  // [2][MCLF_TSLOTS][MCLF_SUB]: tables of the risky sub-tiles
    unsigned long long* fin_sync;     // the finish launches' sync word (zero between launches)
    unsigned int fin_gen;             // generation of the last finish launch's records (mcl_finish_args.tag; the record slots start out as zeros)
    int fin_last_nrec;                // sub-tiles per axis of that launch: another count lays the records out differently (slots are zeroed)
    unsigned long long* prefix;
    float4* parent;
    pf_state* state;
    double* partials;         // [blocks][5]
    int partials_cap;
    bool use_lds;
    int last_blocks, last_tile;   // launch shape of the last k_mcl_main
    int last_main_blocks, last_main_particles, last_tail_tile;
    bool fused_finish, no_fused_finish, no_packed, no_balance, no_framed, no_window, no_fast_trig, no_mirror_reuse, no_stage_dma, no_stage_x4;
    int window_override;          // window side in cells (0: from the scan's reach)
    int cus;                      // compute units of the device
    int split_log2_override;  // -1: automatic
    int block_override;       // 0: automatic
    bool debug;               // record resample index / likelihood per particle (parity tests)
    bool strict;              // strict resampling: prefix[] is overwritten with the reference's rounded double cumulative after every finish
    void* strict_recs; double* strict_starts;     // ... by chunks side by side: a record and the true start per chunk of 128 particles
    bool uniform_now;         // every particle of rec[cur] carries the same weight (a fresh filter, equal uploaded weights): the one kind of
                              // weights on which the integer rule and the reference's rounded cumulative part ways (rand() near 0 or
                              // RAND_MAX puts every U on a partial sum) -- the next resampling then takes the reference's cumulative
    bool prefix_is_strict;    // prefix[] holds that cumulative (doubles), not the integer prefix
    double w_floor;           // 0.001 / wSum of N floor weights as the reference's loop rounds it (particle_filter.cpp:116-141): the all-floor set's weight
    unsigned long long* block_sums;   // scan scratch
    int scan_blocks;
    int32_t* dbg_idx;
    int32_t* dbg_like;
    float* d_noise;           // 3 * n_local (parity mode)
    bl_particle_t* d_export;  // n_local
    // uniform utimes of the particle set (every particle carries the same pair; DESIGN.md "Particle utimes")
    int64_t pose_utime, parent_utime;
    // ActionModel state (action_model.hpp:60-78)
    bl_pose_xyt_t prev_odom;
    bool action_initialized, moved;
    double rot1, trans, rot2, rot1Std, transStd, rot2Std;
    uint64_t noise_seed;
    uint32_t step;
    bool initialized;
    bool pending_end;         // update_begin issued, update_end not yet
    int64_t pending_utime;
    // composed finish of a sharded set (bl_pf_shard_*): rank, world, particles per rank; the other ranks' exchange records and
    // weight prefixes as this process sees them (its own included); the exchange blocks of the finish
    int sh_rank, sh_world, sh_block, sh_world_pending;
    const float4* sh_peer_rec[2][BL_MAX_SHARDS];                  // host copies of the tables below
    const unsigned long long* sh_peer_prefix[BL_MAX_SHARDS];
    bool sh_stage_sums, sh_stage_groups;       // the stages of the running update's exchange that have been enqueued
    struct mcl_shard_tab* sh_tab;              // device [2]: the tables k_mcl_main reads sources through, for cur = 0 / 1
    mclf_shards* sh_fin;                       // device [2]: the finish's view, for the record written by an update from cur = 0 / 1
    char* sh_xchg; size_t sh_xchg_stride;      // the exchange blocks of the finish (this rank's and, gathered, every rank's)
    int sh_subs_per_rank;
    // peer-store form of the exchange (bl_pf_shard_peer_*): no collective -- every rank pushes its slice of the tile sums and its
    // exchange block into every other rank's buffers (their memory, mapped into this process) and raises a per-source counter
    // there; the consuming launches are preceded by a one-wave wait on those counters
    bool sh_peer;                              // the form is set up (every rank's buffers are mapped)
    bool sh_broken;                            // the host has seen pf_state::shard_broken (sticky until the shards are set up again)
    unsigned long long* sh_flags;              // device: [2][BL_MAX_SHARDS] -- the update number rank r's sums / block are here for
    double* sh_peer_sums[BL_MAX_SHARDS];       // every rank's tile-sum buffer (both parities), exchange blocks and counters as THIS process sees them
    char* sh_peer_xchg[BL_MAX_SHARDS];
    unsigned long long* sh_peer_flags[BL_MAX_SHARDS];
    struct shard_peers* sh_peers_dev;          // device copy of the three tables
    unsigned long long sh_gen;                 // moved updates exchanged so far
    int sh_tiles_all;                          // tiles of the padded particle set (one parity of the tile-sum buffer)
    int64_t sh_bytes_pulled_bound;             // (diagnostic) upper bound of the bytes the last k_mcl_main may have read from other ranks
};

// ---------------------------------------------------------------- device helpers
__device__ __forceinline__ double wave_sum(double v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// Map window staged in LDS (hard part 5): every gather of one update falls inside the bounding box
// of the particle cloud dilated by the laser range.  win = (x0, y0, w, h) in cells, already clipped to the grid; stride
// is the LDS row pitch in bytes.  A cell outside the window (cloud wider than expected) is read from HBM/L2 instead,
// so the window is a cache, never a correctness condition.
struct map_window { int x0, y0, w, h, stride; };

typedef __attribute__((address_space(3))) signed char lds_i8_t;

// min(max(x, -1), hi) for hi >= -1
__device__ __forceinline__ int clamp_from_m1(int x, int hi)
{
    int r;
    asm("v_med3_i32 %0, %1, -1, %2" : "=v"(r) : "v"(x), "s"(hi));
    return r;
}

// MAP_MODE: 0 = gathers from HBM/L2 only; 1 = the whole grid is staged in LDS; 2 = an LDS window of the zero-framed copy,
// cells outside it gathered from that copy through L2 (packed scoring only).
// Branch-free in modes 0 and 1: an off-grid cell reads a dummy slot and is masked to 0 (OccupancyGrid::logOdds,
// occupancy_grid.cpp:63-71).
template <int MAP_MODE>
__device__ __forceinline__ int grid_odds(const int8_t* __restrict__ cells, const lds_i8_t* s_map, const map_window& win,
                                         const bl_frame& f, int x, int y)
{
    if (MAP_MODE == 1) {
        // whole grid staged with a zero frame (one row above/below, four columns left, >= four right): clamping the cell
        // to the frame replaces the bounds test, the index select and the result mask
        // to the frame replaces the bounds test; v_med3_i32 is the clamp in one instruction, and the row offset is a 24-bit
        // multiply (v_mul_lo_u32 issues at quarter rate; rows and stride are far below 2^24)
        // s_map points at cell (0, 0) of the framed image here, so row -1 / column -1 are plain negative offsets
        const int cx = clamp_from_m1(x, f.width), cy = clamp_from_m1(y, f.height);
        return s_map[__mul24(cy, win.stride) + cx];
    }
    // modes 0 and 2 (lanes of the window mode that cannot take the packed path): bounds test + gather through L2
    const bool in = (unsigned int)x < (unsigned int)f.width && (unsigned int)y < (unsigned int)f.height;
    const int v = cells[in ? (size_t)y * f.width + x : (size_t)0];
    return in ? v : 0;
}

// SensorModel::scoreRay (sensor_model.cpp:28-59) in half-units: returns 2*odds, o1 or o2 (score = that / 2).
// The two neighbour cells are gathered unconditionally: within a 64-lane wave some ray always needs them, so the
// early-out of the reference would only add divergence.
template <int MAP_MODE>
__device__ __forceinline__ int score_ray_half_units(const int8_t* __restrict__ cells, const lds_i8_t* s_map,
                                                     const map_window& win, const bl_frame& f, float sx, float sy,
                                                     int isx, int isy, float range, float cs, float sn)
{
    const float tx = range * cs * f.cpm, ty = range * sn * f.cpm;
    const int ex = (int)(tx + sx);
    const int ey = (int)(ty + sy);
    // (2 * range * cos) * cpm == 2 * ((range * cos) * cpm) bit for bit: scaling by two is exact and commutes with
    // rounding for normal floats (range > 0.15 and |cos| >= 4e-8 keep every product far from the subnormal range)
    const int xx = (int)((2.0f * tx) + sx);
    const int xy = (int)((2.0f * ty) + sy);
    int ax, ay, bx, by;
    bl_bresenham_first_step(ex, ey, isx, isy, &ax, &ay);
    bl_bresenham_first_step(ex, ey, xx, xy, &bx, &by);
    const int odds = grid_odds<MAP_MODE>(cells, s_map, win, f, ex, ey);
    const int o1 = grid_odds<MAP_MODE>(cells, s_map, win, f, ax, ay);
    const int o2 = grid_odds<MAP_MODE>(cells, s_map, win, f, bx, by);
    return odds > 0 ? 2 * odds : (o1 > 0 ? o1 : (o2 > 0 ? o2 : 0));
}

// ---- packed 16-bit form of the scoring arithmetic (whole-grid LDS mode, no pose interpolation) ----------------------------
// Cell coordinates travel as (x, y) int16 pairs, so one v_pk_* instruction serves both axes: the two Bresenham first steps,
// the clamps to the zero frame and (v_dot2_i32_i16 with (1, stride)) the LDS offset.  Exactly the int32 arithmetic as long as
// nothing leaves int16: the host enables it only for grids up to 8192 cells a side and scans whose longest ray spans at most
// 4000 cells, and a lane uses it only if its start cell lies within +-8191 -- then |end| <= 12191, |2x-range point| <= 16191
// and every doubled difference stays below 2^15.
typedef short short2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ short2_t pk_sign_fill(short2_t v)              // per half: -1 if negative else 0
{
    short2_t r;
    // kept opaque: as plain C the compiler turns (v >> 15) & x into per-half compares and selects
    asm("v_pk_ashrrev_i16 %0, 15, %1 op_sel_hi:[0,1]" : "=v"(r) : "v"(v));
    return r;
}

// bl_bresenham_first_step on packed cells: step x iff 2dx - dy >= 0, step y iff 2dy - dx >= 0, toward the target
__device__ __forceinline__ short2_t first_step_sg(short2_t e, short2_t target)     // the step itself: (+-1 or 0, +-1 or 0)
{
    const short2_t d = target - e;
    const short2_t nd = -d;
    const short2_t nad = __builtin_elementwise_min(d, nd);              // -|d|
    short2_t t;                                                           // 2|dx| - |dy|, 2|dy| - |dx| = -2 nad + nad.yx in one v_pk_mad_i16
    asm("v_pk_mad_i16 %0, %1, -2, %1 op_sel:[0,0,1] op_sel_hi:[1,0,0]" : "=v"(t) : "v"(nad));
    const short2_t nostep = pk_sign_fill(t);
    const short2_t dpos = pk_sign_fill(nd);                               // d > 0 ? -1 : 0 (the sign of -d, which |d| needed anyway)
    // (~dpos | 1) & ~nostep, i.e. (d > 0 ? 1 : -1) where the axis steps, in one v_bitop3 (truth table a=dpos, b=nostep, c=1: bits 0, 1, 5)
    const int sg = __builtin_amdgcn_bitop3_b32(__builtin_bit_cast(int, dpos), __builtin_bit_cast(int, nostep), 0x00010001, 0x23);
    return __builtin_bit_cast(short2_t, sg);
}
__device__ __forceinline__ short2_t first_step_pk(short2_t e, short2_t target) { return e + first_step_sg(e, target); }

// A wave-uniform value held in a vector register for the whole ray loop (an asm operand of pk_dot2 that the compiler only
// knows as a scalar is re-materialised with a v_mov in front of every use)
__device__ __forceinline__ int vgpr_of(int s)
{
    int v;
    asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(s));
    return v;
}

// a.x * b.x + a.y * b.y + c: the three-address form (hipcc lowers __builtin_amdgcn_sdot2 to v_dot2c, which accumulates
// into its destination and so needs a v_mov of c in front of it)
__device__ __forceinline__ int pk_dot2(short2_t a, short2_t b, int c)
{
    int r;
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// The zero frame around a staged image is MCL_FRAME cells wide (rows; 4 or more columns).  Scoring clamps the ENDPOINT cell
// to [-2, size + 1] once; its two Bresenham neighbours are then formed from the clamped cell and need no clamp of their
// own: they stay within the frame, and whenever the clamp moved the endpoint all three true cells lie outside the grid
// (the endpoint by >= 2 cells, so its neighbours by >= 1) and all three cells read are frame zeros.
#define MCL_FRAME BL_MIRROR_FRAME
struct pk_map { int base; short2_t K; short2_t hi; };                     // LDS address of cell (0,0); (1, stride); (W + 1, H + 1)
struct pk_map_global { const int8_t* base; short2_t K; short2_t hi; };    // the same over a zero-framed copy in device memory

__device__ __forceinline__ short2_t pk_clamp_endpoint(short2_t c, short2_t hi)
{
    const short2_t lo = {(short)-2, (short)-2};
    return __builtin_elementwise_min(__builtin_elementwise_max(c, lo), hi);
}

__device__ __forceinline__ int pk_read(const pk_map& pm, short2_t c)
{
    return *(const lds_i8_t*)(size_t)(unsigned int)pk_dot2(c, pm.K, pm.base);
}

__device__ __forceinline__ int pk_read(const pk_map_global& pm, short2_t c)
{
    return pm.base[pk_dot2(c, pm.K, 0)];                                   // |offset| < 2^27
}
// the same reads by position: where cell c lies (the LDS address, or the offset in the framed copy), and a cell one step sg from
// a position -- the dot product is linear, so the neighbour's position is dot2(sg, K) + the cell's: one instruction where forming
// the neighbour cell and its position takes two
__device__ __forceinline__ int pk_pos(const pk_map& pm, short2_t c) { return pk_dot2(c, pm.K, pm.base); }
__device__ __forceinline__ int pk_pos(const pk_map_global& pm, short2_t c) { return pk_dot2(c, pm.K, 0); }
__device__ __forceinline__ int pk_at(const pk_map&, int pos) { return *(const lds_i8_t*)(size_t)(unsigned int)pos; }
__device__ __forceinline__ int pk_at(const pk_map_global& pm, int pos) { return pm.base[pos]; }

// SensorModel::scoreRay in half-units (see score_ray_half_units), packed form.  The float endpoint arithmetic is written on
// (x, y) pairs as well (v_pk_mul_f32 / v_pk_add_f32): each lane-wise operation is the reference's own IEEE operation
// (2.0f * t == t + t exactly).
typedef float float2_t __attribute__((ext_vector_type(2)));

// bl_wrap_to_pi for the ray angles, without divergent loops: the first upward step of the reference's loop as a select (the
// same IEEE operation), and the full function only for a wave in which some lane needs anything else.
__device__ __forceinline__ float wrap_to_pi_cells(float x, bool simple)
{
    const float PI_F = 0x1.921fb6p+1f;
    // a pose angle in (-pi, pi) less a scan angle in [0, 2pi) only ever needs the upward step; anything else (x >= pi, or a
    // second step) takes the full function under the wave-uniform branch.  `simple` (wave-uniform): the host has checked
    // that every theta of the scan lies in [0, 6.2831], and x = wrapped angle - theta: then x < pi and x + 2pi > -pi hold
    // for every ray and the test is skipped.
    const float up = (float)((double)x + 2.0 * BL_PI);
    float w = x <= -PI_F ? up : x;
    if (!simple) {
        if (__builtin_amdgcn_ballot_w64(__builtin_fabsf(w) >= PI_F)) {
            if (__builtin_fabsf(w) >= PI_F) w = bl_wrap_to_pi(x);
        }
    }
    return w;
}

// the endpoint and the point at twice the range (whose direction the second neighbour is taken in), in float cells
__device__ __forceinline__ void ray_points_pk(float2_t start, float cpm, float range, float cs, float sn, float2_t& e, float2_t& x)
{
    const float2_t dir = {cs, sn};
    const float2_t t = (range * dir) * cpm;                               // (range * cos) * cpm, (range * sin) * cpm
    e = t + start;
    x = (t + t) + start;
}
// (int)e.x | (int)e.y << 16 in two instructions each: the float -> int conversion writes its low half-word straight into
// the selected half of the pair (SDWA destination select).  No saturation is needed: the packed path's preconditions keep
// every cell inside int16.  A half-word write must not be read by the very next instruction (gfx940 dst_sel forwarding
// hazard, invisible to the compiler inside inline asm), hence the interleaving and the closing s_nop.
__device__ __forceinline__ void ray_points_to_cells(float2_t e, float2_t x, short2_t& E, short2_t& X)
{
    int ei, xi;
    asm("v_cvt_i32_f32_sdwa %0, %2 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD\n\t"
        "v_cvt_i32_f32_sdwa %1, %4 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD\n\t"
        "v_cvt_i32_f32_sdwa %0, %3 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
        "v_cvt_i32_f32_sdwa %1, %5 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\t"
        "s_nop 0"
        : "=&v"(ei), "=&v"(xi) : "v"(e.x), "v"(e.y), "v"(x.x), "v"(x.y));
    E = __builtin_bit_cast(short2_t, ei);
    X = __builtin_bit_cast(short2_t, xi);
}
// the endpoint cell and the cell at twice the range
__device__ __forceinline__ void ray_cells_pk(float2_t start, float cpm, float range, float cs, float sn, short2_t& E, short2_t& X)
{
    float2_t e, x;
    ray_points_pk(start, cpm, range, cs, sn, e, x);
    ray_points_to_cells(e, x, E, X);
}

// ---- the same two cells WITHOUT the exact sinf / cosf for almost every ray.
// The reference takes sinf / cosf of theta' = wrap_to_pi(d), d = fl(pose.theta - ray theta).  Here the direction comes from the
// addition theorems: cos(p - r) = cp cr + sp sr, sin(p - r) = sp cr - cp sr, with (cp, sp) the particle's and (cr, sr) the ray's
// pair (the ray's formed in double and rounded once, ray_table_entry; the particle's by bl_sincosf, which is libm's sinf / cosf bit
// for bit) -- two packed instructions per ray where v_sin_f32 and v_cos_f32 (quarter
// rate each, and a multiply in front) cost nine issue slots.  How far that is from the reference's float values, for every float
// p in [-pi, pi] and r in [0, BL_THETA_SIMPLE_MAX = 6.2831] (theta_simple: a single wrap), with u = 2^-24:
//     the reference's angle: |fl(p - r) - (p - r)| <= 4.8e-7 (half an ulp below 16: p - r reaches -3 pi), the wrap's own rounding
//         <= 1.2e-7 (half an ulp below 4: it adds 2 pi in double and rounds once); sinf / cosf of it, rounded: + 6e-8    <= 6.6e-7
//     the four table values: each within 0.56 ulp = 3.3e-8 of the true value (libm's sinf / cosf); into the two products: <= 3.3e-8
//         (|cr| + |cp| + |sr| + |sp|) <= 9.4e-8; the product's and the fma's own roundings: 2 x 3e-8                    <= 1.6e-7
// together <= 8.2e-7 < MCL_TRIG_EPS = 9.3e-7 (measured maximum over 1e10 random pairs: bl_debug_trig_addition_probe,
// profiles/r04_trig_addition_probe.json).
// pcs scaled (round 6): the particle's pair is carried as fl(cpm cp), fl(cpm sp), so that the two packed instructions give cpm * dir and
// range * that is t in one multiplication where the reference has two.  The scaling's rounding is half an ulp of each of cp, sp:
// <= 6e-8 (|cp cr| + |sp sr|) <= 6e-8 into the direction, together <= 8.8e-7 < MCL_TRIG_EPS; t' = fl(range * (cpm dir')) has ONE
// rounding where t = fl(fl(range * dir) * cpm) has two, inside the 4u the second line below allows.  The probe measures this form.
// The guard band below is the one the hardware-sine form was proven with (tests/tools/sincos_hw_probe.hip measured 8.81e-7 for it);
// it is kept, so the exact path behind it is taken as often as before.
// With dir' = dir + eta, |eta| <= eps, and u = 2^-24 the relative error of one float operation:
//     |fl(range * dir') - fl(range * dir)|      <= range * (eps + 2u)
//     |t' - t|, t = fl(fl(range * dir) * cpm)   <= range * cpm * (eps + 4u) (1 + 2u)
//     |e' - e|, e = fl(t + start)                <= range * cpm * (eps + 4u) (1 + 2u) + 2u * max(|e|, |e'|)
//     |x' - x|, x = fl(fl(t + t) + start)        <= 2 * range * cpm * (eps + 4u) (1 + 2u) + 2u * max(|x|, |x'|)
//     round 6: e' = fl(range * dc + start), x'' = fl(range * dc + e') with dc = cpm * dir' -- one fused multiply-add each.  With
//     T = range * dc (no rounding): |T - t| <= range * cpm * (eps + 2u) (t has two roundings), e' = (T + start)(1 + d1), x'' = (T + e')(1 + d2),
//     and t + t exact in x = (2t + start)(1 + d):
//     |e' - e|                                    <= range * cpm * (eps + 2u) + 2u * max(|e|, |e'|)
//     |x'' - x|                                   <= 2 * range * cpm * (eps + 2u) + 3u * max(|x|, |x''|, |e'|)
// so the truncated cells agree whenever e' (x'') is farther than B1 (B2) from every integer.  With max(|x|, |x''|, |e'|) <= |start| +
// 2 range cpm + 8 the far point's band -- the one the test uses for all four coordinates -- is
//     B2 = range * cpm * (2.04 (eps + 4u) + 6.12 u) + 3.06 u (|start| + 8)
// (until round 6 the coordinate bound used the LONGEST ray for every ray: the constant part was 2.5 times what it is now and, at the
// scan's typical 2 - 3 m, a quarter of the band).  A ray inside a band (1-4 in a thousand) makes its whole wave take the exact path for that round and keeps the exact
// cells; everything downstream is integer.  The test itself (round 6): the distance of c to the nearest integer is 0.5 - |fract(c) - 0.5|,
// exactly, so "both coordinates farther than B" is max(|fract - 0.5|) < 0.5 - B -- two v_fract, one packed add, one max per point;
// the threshold 0.5 - B is formed with 1.2e-7 taken off (its two roundings are half an ulp of 0.5 each), which only widens the band.
// MCL_TRIG_EPS: 5.5 % above the larger maximum measured for the hardware form (bl_debug_trig_probe, kept: BOTLAB_MCL_HW_TRIG=1 takes
// that form), 13 % above the addition form's analytic bound and 45 % above its measured maximum; tests/test_gpu_trig_guard.py
// asserts the measured maxima of both forms stay below it.
#define MCL_TRIG_EPS 9.3e-7f
__device__ __forceinline__ void hw_sincos_unwrapped(float d, float* sn, float* cs)
{
    const float rev = d * 0.15915494309189535f;
    asm("v_sin_f32 %0, %1" : "=v"(*sn) : "v"(rev));
    asm("v_cos_f32 %0, %1" : "=v"(*cs) : "v"(rev));
}
// (range, theta, cos theta, sin theta): cosf / sinf as libm rounds them (bl_sincosf_cells: a double polynomial rounded once,
// within 0.56 ulp -- a double sincos here cost the staging waves more than the transcendentals it replaces saved the ray loop)
__device__ __forceinline__ float4 ray_table_entry(float range, float theta)
{
    float sn, cs;
    bl_sincosf_cells(theta, &sn, &cs);
    return make_float4(range, theta, cs, sn);
}
// (cos, sin)(p - r) from (cp, sp) and (cr, sr): v_pk_mul_f32 + v_pk_fma_f32
__device__ __forceinline__ float2_t trig_by_addition(float2_t pcs, float cr, float sr)
{
    const float2_t a = {pcs.y, -pcs.x};                      // (sp, -cp)
    const float2_t t = a * float2_t{sr, sr};
    return __builtin_elementwise_fma(pcs, float2_t{cr, cr}, t);           // (cp cr + sp sr, sp cr - cp sr)
}
// (p, r: the particle's and the ray's angle -- their difference is formed only where it is used, the hardware form and the exact path;
// hw: wave-uniform, a kernel argument)
template <bool HW>
__device__ __forceinline__ void ray_cells_fast(float2_t start, float cpm, float range, float p, float r, float2_t pcs, float cr, float sr,
                                               float k1, float kh2, short2_t& E, short2_t& X)
{
    float2_t dc;                                             // cpm * (cos, sin)(p - r)
    if (HW) {                                                // (BOTLAB_MCL_HW_TRIG: a loop of its own, chosen outside it; the marker
        float sn, cs;                                        // keeps the optimiser from folding the two loops back into one with this
        asm volatile("; hardware sine / cosine");            // branch inside)
        hw_sincos_unwrapped(p - r, &sn, &cs);
        dc = float2_t{cs, sn} * cpm;
    } else
        dc = trig_by_addition(pcs, cr, sr);                  // pcs = cpm (cos p, sin p): the direction comes out in cells per metre ("pcs scaled" above)
    // the two points by two packed multiply-adds (e = range dc + start, x = range dc + e) where the reference has a product, its
    // double and two additions: see "x'' = " above -- one rounding per point instead of two or three, inside the same bounds
    const float2_t rr = {range, range};
    const float2_t e = __builtin_elementwise_fma(rr, dc, start), x = __builtin_elementwise_fma(rr, dc, e);
    // the distance of a coordinate c to the nearest integer is 0.5 - |fract(c) - 0.5|, every step of it exact in float (fract(c) is
    // c - floor(c), a multiple of c's ulp below 1): "farther than B from every integer" is max(|fract - 0.5|) < 0.5 - B.  kh2 = 0.5 -
    // 1.5 k2 - 1.2e-7: the threshold's own rounding (half an ulp of 0.5, twice) only ever widens the band.
    // (one threshold for the four coordinates, the far point's -- B2 >= B1, so the endpoint is only tested more strictly than it
    // needs: a third more rays in the band, two instructions less on every ray)
    const float C2 = __builtin_fmaf(range, -k1, kh2);
    const float2_t fe = float2_t{__builtin_amdgcn_fractf(e.x), __builtin_amdgcn_fractf(e.y)} - 0.5f;
    const float2_t fx = float2_t{__builtin_amdgcn_fractf(x.x), __builtin_amdgcn_fractf(x.y)} - 0.5f;
    const float mx = __builtin_fmaxf(__builtin_fabsf(fx.x), __builtin_fabsf(fx.y));
    const float m4 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(fe.x), __builtin_fabsf(fe.y)), mx);
    const bool near = !(m4 < C2);                                        // (a nan lands here too)
    ray_points_to_cells(e, x, E, X);
    if (__builtin_amdgcn_ballot_w64(near)) {
        float sn2, cs2;
        float p2 = p;
        asm volatile("" : "+v"(p2));                          // (keeps the subtraction inside the branch: 1-4 rays in a thousand come here)
        bl_sincosf_cells(wrap_to_pi_cells(p2 - r, true), &sn2, &cs2);
        short2_t E2, X2;
        ray_cells_pk(start, cpm, range, cs2, sn2, E2, X2);
        E = near ? E2 : E;
        X = near ? X2 : X;
    }
}

__device__ __forceinline__ int score_pick(int odds, int o1, int o2)
{
    return odds > 0 ? 2 * odds : (o1 > 0 ? o1 : (o2 > 0 ? o2 : 0));
}

template <class PM>
__device__ __forceinline__ int score_cells_pk(const PM& pm, short2_t S, short2_t E, short2_t X)
{
    const short2_t Ec = pk_clamp_endpoint(E, pm.hi);
    const int p0 = pk_pos(pm, Ec);
    return score_pick(pk_at(pm, p0), pk_at(pm, pk_dot2(first_step_sg(Ec, S), pm.K, p0)), pk_at(pm, pk_dot2(first_step_sg(Ec, X), pm.K, p0)));
}
template <class PM>
__device__ __forceinline__ int score_ray_pk(const PM& pm, float2_t start, short2_t S, float cpm, float range, float cs, float sn)
{
    short2_t E, X;
    ray_cells_pk(start, cpm, range, cs, sn, E, X);
    return score_cells_pk(pm, S, E, X);
}

// Window form: an LDS copy of the rectangle [org, org + size) of the zero-framed image.  The endpoint is clamped to the
// rectangle less a one-cell rim, so the two neighbours formed from the clamped cell stay inside it; if the clamp moved the
// endpoint the ray is scored again from the framed image itself -- under a wave-uniform branch, rare when the window covers
// the scan's reach around the particle cloud.
struct pk_map_window { int base; short2_t K; short2_t org; short2_t hi; pk_map_global g; };   // hi = size - 2

__device__ __forceinline__ int score_cells_pk(const pk_map_window& pm, short2_t S, short2_t E, short2_t X)
{
    const short2_t one = {(short)1, (short)1};
    const short2_t Ew = E - pm.org;
    const short2_t Ec = __builtin_elementwise_min(__builtin_elementwise_max(Ew, one), pm.hi);
    const bool miss = __builtin_bit_cast(int, Ec) != __builtin_bit_cast(int, Ew);
    // the directions toward S and X are differences: taken from the endpoint itself in grid coordinates -- where the clamp moved it
    // (miss) all three values are read again below
    const int p0 = pk_dot2(Ec, pm.K, pm.base);
    int odds = *(const lds_i8_t*)(size_t)(unsigned int)p0;
    int o1 = *(const lds_i8_t*)(size_t)(unsigned int)pk_dot2(first_step_sg(E, S), pm.K, p0);
    int o2 = *(const lds_i8_t*)(size_t)(unsigned int)pk_dot2(first_step_sg(E, X), pm.K, p0);
    if (__builtin_amdgcn_ballot_w64(miss)) {
        if (miss) {
            const short2_t Eg = pk_clamp_endpoint(E, pm.g.hi);
            odds = pk_read(pm.g, Eg); o1 = pk_read(pm.g, first_step_pk(Eg, S)); o2 = pk_read(pm.g, first_step_pk(Eg, X));
        }
    }
    return score_pick(odds, o1, o2);
}
__device__ __forceinline__ int score_ray_pk_window(const pk_map_window& pm, float2_t start, short2_t S, float cpm, float range,
                                                   float cs, float sn)
{
    short2_t E, X;
    ray_cells_pk(start, cpm, range, cs, sn, E, X);
    return score_cells_pk(pm, S, E, X);
}

// Zero-framed copy of the grid (rows -MCL_FRAME..H+MCL_FRAME-1, columns -4..stride-5), one dword per thread: the image
// k_mcl_main stages in LDS for small grids, kept in device memory for grids that do not fit.
__global__ __launch_bounds__(256) void k_mcl_frame(const int8_t* __restrict__ cells, int W, int H, int stride, int* __restrict__ framed)
{
    const int wq = stride >> 2;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= wq * (H + 2 * MCL_FRAME)) return;
    const int ry = i / wq, q = i - ry * wq;
    const int y = ry - MCL_FRAME, x = 4 * q - 4;
    int v = 0;
    if (y >= 0 && y < H && x >= 0 && x < W) {
        const size_t g = (size_t)y * W + x;
        if ((W & 3) == 0) v = *(const int*)(cells + g);
        else
            for (int b = 0; b < 4; ++b)
                if (x + b < W) v |= ((int)(unsigned char)cells[g + b]) << (8 * b);
    }
    framed[i] = v;
}

struct mcl_args {
    const float4* src;            // rec[cur]     (all N)
    float4* dst;                  // rec[cur ^ 1] (all N; this shard writes [lo, hi))
    const unsigned long long* prefix;
    const pf_state* state;
    float4* parent;               // local
    double* partials;             // [gridDim.x][5]
    int32_t* dbg_idx;
    int32_t* dbg_like;
    const int8_t* cells;
    const int8_t* framed;         // cell (0, 0) of the zero-framed copy (MAP_MODE 0 with packed scoring) or null
    int framed_stride;
    bl_frame frame;
    const float* ranges;
    const float* thetas;
    const int64_t* times;         // per-ray stamps; interpolateRatio = (t - t_begin) / t_den (first moved update only)
    int64_t t_begin; double t_den;
    int R;
    int N, lo, n_local;
    double r;                     // (rand/RAND_MAX) * (1/N)
    double M_inv;                 // 1/N
    double rot1, trans, rot2, rot1Std, transStd, rot2Std;
    const float* noise;           // 3 * n_local or null
    uint32_t seed_lo, seed_hi, step;
    int interp;                   // parent utime != pose utime (first moved update)
    int resample;                 // 0: action-only (source = own index)
    int strict;                   // prefix[] holds the reference's rounded double cumulative (bl_pf_set_strict_resampling)
    int win_w, win_h;             // LDS map window size in cells (0: no staging); >= grid size means the whole grid
    int split_log2;               // each particle's rays are spread over 2^split_log2 adjacent lanes
    int pk_ok;                    // grid and scan admit the packed 16-bit scoring path (see score_ray_pk)
    int theta_simple;             // every theta of the scan lies in [0, 6.2831] (see wrap_to_pi_cells)
    int fast_trig;                // ray directions without the exact sinf / cosf, a guard band and the exact path behind it (ray_cells_fast):
                                  // 1 by the addition theorems, 2 by the hardware sine / cosine, 0 off
    int stage_dma;                // whole-grid staging by LDS-DMA (rows of whole dwords, at most 64 of them)
    float max_range_cells;        // longest kept ray in cells
    int main_blocks, main_particles;   // region 1: main_blocks workgroups cover particles [0, main_particles) of the shard
    // composed finish of a sharded set (null otherwise): source records and weight prefix of particle i lie with rank i / block, as
    // this process sees that rank's arrays (its own, or another rank's through an IPC mapping), indexed by the global i.  A table
    // in device memory: indexing a table inside this by-value argument per lane would move the argument into scratch.
    const struct mcl_shard_tab* sh;
};
struct mcl_shard_tab { int world, block; const float4* src[BL_MAX_SHARDS]; const unsigned long long* prefix[BL_MAX_SHARDS]; };

__device__ __forceinline__ float4 mcl_src_at(const mcl_args& a, int i)
{
    if (a.sh) return a.sh->src[i / a.sh->block][i];
    return a.src[i];
}
struct mcl_prefix_view {              // what the resampling search reads the cumulative through
    const mcl_shard_tab* sh; const unsigned long long* flat;
    __device__ __forceinline__ unsigned long long operator[](int i) const { return sh ? sh->prefix[i / sh->block][i] : flat[i]; }
};

__device__ __forceinline__ void philox_normals3(uint32_t m, uint32_t step, uint32_t k0, uint32_t k1, float z[3])
{
    uint32_t o[4];
    bl_philox4x32(m, step, 0x6d636c31u, 0, k0, k1, o);
    // (0,1] uniforms; Box-Muller
    float u1 = ((float)(o[0] >> 8) + 1.0f) * (1.0f / 16777216.0f);
    float u2 = ((float)(o[1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    float u3 = ((float)(o[2] >> 8) + 1.0f) * (1.0f / 16777216.0f);
    float u4 = ((float)(o[3] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    float ra = sqrtf(-2.0f * logf(u1)), rb = sqrtf(-2.0f * logf(u3));
    float s2, c2, c4;
    sincosf(6.2831853071795864769f * u2, &s2, &c2);
    c4 = cosf(6.2831853071795864769f * u4);
    z[0] = ra * c2; z[1] = ra * s2; z[2] = rb * c4;
}

// The packed ray loops: lane `sub` of a particle's group takes rays sub, sub + split, ...  All lanes of a wave run the same
// number of whole rounds (a scalar loop counter; the ray index only lives on as the LDS address of the table entry), and the
// lanes with one more ray take it in a masked round behind the loop.
#define MCL_RAY_LOOP(SCORE_EXPR)                                                        \
    do {                                                                                \
        const int rounds_ = cnt >> sl2;                                                 \
        int off_ = sub * 16;                         /* byte offset of the table entry */ \
        for (int k_ = 0; k_ < rounds_; ++k_, off_ += split * 16) {                      \
            const float2 rt = *(const float2*)((const char*)s_ray + off_);              \
            float sn, cs;                                                               \
            bl_sincosf_cells(wrap_to_pi_cells(pth_r - rt.y, theta_simple), &sn, &cs);                   \
            acc += SCORE_EXPR;                                                          \
        }                                                                               \
        if ((off_ >> 4) < cnt) {                                                        \
            const float2 rt = *(const float2*)((const char*)s_ray + off_);              \
            float sn, cs;                                                               \
            bl_sincosf_cells(wrap_to_pi_cells(pth_r - rt.y, theta_simple), &sn, &cs);                   \
            acc += SCORE_EXPR;                                                          \
        }                                                                               \
    } while (0)

// The same loops with the two cells from ray_cells_fast (theta_simple scans only): the direction by the addition theorems from
// the particle's (cos, sin) pair pcs_ and the table's.  Rays [LO, HI) of the chunk (LO a multiple of the split).
#define MCL_RAY_LOOP_FAST_RANGE(PM, LO, HI, HW)                                         \
    do {                                                                                \
        const int rounds_ = ((HI) - (LO)) >> sl2;                                       \
        int off_ = ((LO) + sub) * 16;                                                   \
        for (int k_ = 0; k_ < rounds_; ++k_, off_ += split * 16) {                      \
            const float4 rt = *(const float4*)((const char*)s_ray + off_);              \
            short2_t E_, X_;                                                            \
            ray_cells_fast<HW>(start, a.frame.cpm, rt.x, pth_r, rt.y, pcs_, rt.z, rt.w, trig_k1, trig_kh2, E_, X_);   \
            acc += score_cells_pk(PM, S, E_, X_);                                       \
        }                                                                               \
        if ((off_ >> 4) < (HI)) {                                                       \
            const float4 rt = *(const float4*)((const char*)s_ray + off_);              \
            short2_t E_, X_;                                                            \
            ray_cells_fast<HW>(start, a.frame.cpm, rt.x, pth_r, rt.y, pcs_, rt.z, rt.w, trig_k1, trig_kh2, E_, X_);   \
            acc += score_cells_pk(PM, S, E_, X_);                                       \
        }                                                                               \
    } while (0)
#define MCL_RAY_LOOP_FAST(PM) do { if (hw_trig_) MCL_RAY_LOOP_FAST_RANGE(PM, 0, cnt, true); else MCL_RAY_LOOP_FAST_RANGE(PM, 0, cnt, false); } while (0)

// ---- resampling search, first part: a wave narrows the range its lanes have to bisect -----------------------------------
// The lanes of a wave resample consecutive particles, so their targets T ascend and every lane's source index lies
// between the first and the last active lane's.  Those two are found by the whole wave at once -- 64 probes per round and
// target, the range shrinking 64-fold: three dependent L2 round trips at 100k particles instead of the bisection's 17 --
// and each lane then bisects only inside [first, last] (about as many entries as the wave has particles).
// index(T) = first i with T <= prefix[i], or N - 1 (particle_filter.cpp:84-103 with D4's clamp).
__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ int kary_pos(int lo, int hi, int step, int k) { return min(lo + (k + 1) * step - 1, hi); }

// The cumulative the search runs on: the exact integer prefix of the weight units (compared with T = U * S), or -- strict
// mode -- the reference's own sequentially rounded double cumulative c_i (k_pf_cumulative_strict; compared with T = U).
__device__ __forceinline__ bool resample_reaches(double T, unsigned long long v, bool strict)
{
    return strict ? T <= __longlong_as_double((long long)v) : T <= (double)v;
}

// Q probes per lane and search in a round: 64 Q probes narrow a range by that factor per dependent round trip.  Measured at 100 000
// particles (round 6, profiles/r06_mcl_timeline.txt): Q = 8 -- two rounds instead of three -- has the bracket at 7.0 us after the
// workgroup's entry where Q = 1 has it at 3.1: every one of the launch's 1 488 prologue waves probes the same positions in its first
// round and different lines per lane in every round, so a round is bound by the L2's request rate (64 Q lines per wave and search),
// not by its latency.  Q = 1 stands.
#define MCL_BRACKET_Q 1
// m (round 6): the lane's own output index, or -1.  U_m = r + m / N, so with weights that do not differ wildly the source index lies
// near m: ONE round of 64 probes, 64 entries apart, around the first lane's m -- probe 0 the guard below the window, probe 63 its top
// -- brackets both ends of the wave when the guard does not reach the first target and the top reaches the last; the lanes then
// bisect a bracket of up to ~200 entries (8 steps instead of 6 - 7) and two of the three dependent rounds are gone.  Anything else
// (a cumulative far from linear, the array's ends) takes the rounds below from the full range, as before.
#define MCL_BRACKET_WINDOW 64
template <int Q, class Prefix>
__device__ __forceinline__ void resample_bracket(const Prefix& prefix, int N, double T, bool active,
                                                 int lane, int* out_lo, int* out_hi, bool strict = false, int m = -1)
{
    *out_lo = 0; *out_hi = N - 1;
    const unsigned long long act = __builtin_amdgcn_ballot_w64(active);
    if (!act) return;
    const double T0 = readlane_f64(T, __ffsll((long long)act) - 1), T1 = readlane_f64(T, 63 - __clzll((long long)act));
    const int m0 = __builtin_amdgcn_readlane(m, __ffsll((long long)act) - 1);
    if (m0 >= 0) {
        // probes at g, g + W, ..., g + 63 W with g = m0 - 31 W (the wave's targets span about 64 entries around m0 + 32)
        const int g = m0 - 31 * MCL_BRACKET_WINDOW;
        if (g >= 0 && g + 63 * MCL_BRACKET_WINDOW < N) {
            const unsigned long long v = prefix[g + lane * MCL_BRACKET_WINDOW];
            const unsigned long long b0 = __builtin_amdgcn_ballot_w64(resample_reaches(T0, v, strict));
            const unsigned long long b1 = __builtin_amdgcn_ballot_w64(resample_reaches(T1, v, strict));
            // the probes ascend, so each ballot is a run of ones at the top: its lowest set bit f is the first probe that reaches T
            if (b0 && !(b0 & 1ull) && b1) {
                const int f0 = __ffsll((long long)b0) - 1, f1 = __ffsll((long long)b1) - 1;
                *out_lo = g + (f0 - 1) * MCL_BRACKET_WINDOW + 1;        // the probe below f0 does not reach T0: index(T0) lies above it
                *out_hi = g + f1 * MCL_BRACKET_WINDOW;                  // probe f1 reaches T1: index(T1) is at most there
                return;
            }
        }
    }
    int lo0 = 0, hi0 = N - 1, lo1 = 0, hi1 = N - 1;
    for (int round = 0; round < 8 && (lo0 < hi0 || lo1 < hi1); ++round) {     // 64^8 entries: the cap only bounds the loop
        const int step0 = (hi0 - lo0 + 64 * Q) / (64 * Q), step1 = (hi1 - lo1 + 64 * Q) / (64 * Q);
        unsigned long long v0[Q], v1[Q];
#pragma unroll
        for (int u = 0; u < Q; ++u) {                                          // probe k = 64 u + lane: ascending in (u, lane)
            v0[u] = prefix[kary_pos(lo0, hi0, step0, 64 * u + lane)];
            v1[u] = prefix[kary_pos(lo1, hi1, step1, 64 * u + lane)];
        }
        int f0 = -1, f1 = -1;                                                  // first probe that reaches T (-1: none)
#pragma unroll
        for (int u = Q - 1; u >= 0; --u) {
            const unsigned long long b0 = __builtin_amdgcn_ballot_w64(resample_reaches(T0, v0[u], strict));
            const unsigned long long b1 = __builtin_amdgcn_ballot_w64(resample_reaches(T1, v1[u], strict));
            if (b0) f0 = 64 * u + __ffsll((long long)b0) - 1;
            if (b1) f1 = 64 * u + __ffsll((long long)b1) - 1;
        }
        if (f0 < 0) lo0 = hi0;                                                 // nothing reaches T: the clamp
        else { const int nl = f0 ? kary_pos(lo0, hi0, step0, f0 - 1) + 1 : lo0; hi0 = kary_pos(lo0, hi0, step0, f0); lo0 = nl; }
        if (f1 < 0) lo1 = hi1;
        else { const int nl = f1 ? kary_pos(lo1, hi1, step1, f1 - 1) + 1 : lo1; hi1 = kary_pos(lo1, hi1, step1, f1); lo1 = nl; }
    }
    if (lo0 == hi0 && lo1 == hi1) { *out_lo = lo0; *out_hi = lo1; }             // (else the full range stands)
}
template <class Prefix>
__device__ __forceinline__ void resample_bracket(const Prefix& prefix, int N, double T, bool active,
                                                 int lane, int* out_lo, int* out_hi, bool strict = false, int m = -1)
{
    resample_bracket<1>(prefix, N, T, active, lane, out_lo, out_hi, strict, m);
}

// second part: the lane's own bisection inside the bracket.  index(T) = first i with T <= prefix[i] (clamped by the bracket).
template <class Prefix>
__device__ __forceinline__ int resample_bisect(const Prefix& prefix, double T, int lo, int hi, bool strict = false)
{
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (resample_reaches(T, prefix[mid], strict)) hi = mid; else lo = mid + 1;
    }
    return lo;
}

// Row-wise staging by `n_sw` whole waves (this one is number `sw` of them, wave-uniform): lane = dword column of the LDS row,
// `MCL_STAGE_ROWS` rows in flight per wave.  The row index is scalar, so a load costs no address arithmetic beyond one add
// and the frame rows are a scalar branch (the thread-strided form spent ~20 vector instructions per dword on indices and
// clamps).  src_row(ry) returns the dword pointer of LDS row ry's source or nullptr for an all-zero row; column q of the row
// holds source dword q - q_lo when q_lo <= q < q_hi and zero otherwise.
#define MCL_STAGE_ROWS 6
template <class SrcRow>
__device__ __forceinline__ void stage_rows(int* s_map32, int rows, int wq, int q_lo, int q_hi, int sw, int n_sw, int lane, SrcRow src_row)
{
    for (int q0 = 0; q0 < wq; q0 += 64) {                       // one pass for rows of up to 64 dwords (256 cells)
        const int q = q0 + lane;
        const bool qin = q < wq, qdata = q >= q_lo && q < q_hi;
        for (int r0 = sw; r0 < rows; r0 += n_sw * MCL_STAGE_ROWS) {
            int v[MCL_STAGE_ROWS];
#pragma unroll
            for (int u = 0; u < MCL_STAGE_ROWS; ++u) {
                const int ry = r0 + u * n_sw;
                v[u] = 0;
                if (ry < rows) {
                    const int* src = src_row(ry);
                    if (src != nullptr && qdata) v[u] = src[q - q_lo];
                }
            }
#pragma unroll
            for (int u = 0; u < MCL_STAGE_ROWS; ++u) {
                const int ry = r0 + u * n_sw;
                if (ry < rows && qin) s_map32[ry * wq + q] = v[u];
            }
        }
    }
}

// One lane group per output particle m of the shard: low-variance resample (gather), ActionModel::applyAction,
// SensorModel::likelihood, weight units, and the block's partial sums for normalisation + pose estimate.
// Launch shape: BLOCK threads; a particle occupies `split` = 2^split_log2 adjacent lanes of one wave (a whole wave in the
// second region of the launch), lane `sub` of them taking rays sub, sub + split, ...  (the host picks split so that the
// launch has >= ~512 workgroups: at 100k particles a one-thread-per-particle launch is 6 waves per CU and latency-bound on
// its serial 290-ray loop).
// Shared prologue (split >= 4, no pose interpolation): the per-particle work -- resampling bisection, gather, Philox noise,
// action model -- is done ONCE per particle by the first P = BLOCK / split threads of the workgroup (one particle per lane)
// while the other waves stage the ray table and the map; the result reaches the particle's lanes through a small LDS table,
// and the same P threads write the particle and feed the partial sums after the ray loop.  (With every lane of a group
// repeating the prologue, as the first form of this kernel did, a wave spent a quarter of its instructions outside the ray
// loop; the kernel is VALU-bound.)
#ifdef MCL_STAMPS
// diagnostic build: per workgroup, the 100 MHz clock at entry, behind the first barrier, behind the ray loop and at the end
#define MCL_STAMP_SLOTS 12
__device__ unsigned long long g_mcl_stamps[4096 * MCL_STAMP_SLOTS];
#define MCL_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 4096) g_mcl_stamps[blockIdx.x * MCL_STAMP_SLOTS + (k)] = wall_clock64(); } while (0)
#define MCL_STAMP_T(k, T) do { if ((int)threadIdx.x == (T) && blockIdx.x < 4096) g_mcl_stamps[blockIdx.x * MCL_STAMP_SLOTS + (k)] = wall_clock64(); } while (0)
#ifdef MCL_STAMPS_HW
// ... and, in place of the "bisection done" stamp, where the workgroup ran: HW_ID (se, sh, cu, simd of wave 0) | XCC_ID << 32
// (tests/tools/mcl_placement_probe.py)
#define MCL_STAMP_HW() do { if (threadIdx.x == 0 && blockIdx.x < 4096) g_mcl_stamps[blockIdx.x * MCL_STAMP_SLOTS + 6] = \
    (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32); } while (0)
#else
#define MCL_STAMP_HW() do { } while (0)
#endif
extern "C" int bl_debug_mcl_stamps(unsigned long long* out, int n)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mcl_stamps), (size_t)n * MCL_STAMP_SLOTS * 8) == hipSuccess ? 0 : 1;
}
#else
#define MCL_STAMP(k) do { } while (0)
#define MCL_STAMP_T(k, T) do { } while (0)
#define MCL_STAMP_HW() do { } while (0)
#endif

template <int INTERP, int BLOCK, int MAP_MODE>
__global__ __launch_bounds__(BLOCK) void k_mcl_main(mcl_args a)
{
    extern __shared__ __align__(16) signed char s_dyn[];
    __shared__ double s_part[BLOCK / 64][5];
    __shared__ map_window s_win;
    // (range, theta, cos theta, sin theta) of the kept rays, MCL_LDS_RAYS at a time: one ds_read_b128 per ray.  (A table that is
    // read from LDS or, past its size, from global memory makes every access a FLAT load behind two scalar branches.)  The cosine
    // and sine are those of the ray's own angle, formed in double and rounded once (ray_table_entry): the fast ray loop gets the
    // direction of pose.theta - theta from them by the addition theorems instead of two transcendentals per particle-ray.
    __shared__ float4 s_ray[MCL_LDS_RAYS];
    __shared__ float4 s_pp[BLOCK / 4];                      // shared prologue: (theta, start x, start y, -) per particle
    __shared__ float2 s_pcs[BLOCK / 4];                     // ... and (cos theta, sin theta), formed in double and rounded once
    __shared__ int s_acc[BLOCK / 4];                        // shared prologue: half-unit score per particle
    const lds_i8_t* s_map = (const lds_i8_t*)s_dyn;
    int* s_map32 = (int*)s_dyn;
    map_window win = {0, 0, 0, 0, 0};
    const int tid = (int)threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    MCL_STAMP(0);

    // Two regions in one launch (see "Whole rounds" in pf_launch_main): workgroups [0, main_blocks) take BLOCK >> split_log2
    // particles each, 2^split_log2 lanes per particle; the workgroups after them take the remaining particles one per
    // wave (64 lanes over the rays), so that the last partial round of the machine lasts a tenth of a full one.
    // composed shard whose exchange gave up (k_shard_wait): the other ranks' records and prefix are not what this update needs
    if (a.sh && __hip_atomic_load(&a.state->shard_broken, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    const bool tail = (int)blockIdx.x >= a.main_blocks;
    const int sl2 = tail ? 6 : a.split_log2;
    const int split = 1 << sl2;
    const int P = BLOCK >> sl2;                                                     // particles of this workgroup
    const int jbase = tail ? a.main_particles + ((int)blockIdx.x - a.main_blocks) * (BLOCK >> 6) : (int)blockIdx.x * P;
    const int jl = tid >> sl2;
    const int j = jbase + jl;
    const int sub = tid & (split - 1);
    const bool active = j < a.n_local && (tail || j < a.main_particles);
    const bool shared_pro = !INTERP && sl2 >= 2;                                    // workgroup-uniform
    const int pw = shared_pro ? (P + 63) >> 6 : 0;                                  // waves that run the prologue only
    const int n_sw = BLOCK / 64 - pw;                                               // waves that stage
    // the particle whose prologue and epilogue this thread runs
    const int jp = shared_pro ? jbase + tid : j;
    const bool pro_active = shared_pro ? (tid < P && jp < a.n_local && (tail || jp < a.main_particles)) : active;
    const int mp = a.lo + jp;

    if (MAP_MODE == 2) {
        // ---- the window of the zero-framed copy to stage: win_w x win_h cells centred on the cell the previous pose estimate
        // moves to under the odometry action (it may hang over the grid into the zero frame)
        if (tid == 0) {
            map_window w;
            const bl_pose_xyt_t p = a.state->pose;
            const float ex = (float)((double)p.x + a.trans * cos((double)p.theta + a.rot1));
            const float ey = (float)((double)p.y + a.trans * sin((double)p.theta + a.rot1));
            float gx, gy;
            bl_global_to_grid(ex, ey, a.frame, &gx, &gy);
            const int cx = (gx > -1.0e6f && gx < 1.0e6f) ? (int)gx : 0, cy = (gy > -1.0e6f && gy < 1.0e6f) ? (int)gy : 0;
            // framed image: columns [-4, framed_stride - 4), rows [-MCL_FRAME, H + MCL_FRAME); window columns stay dword-aligned
            w.x0 = max(-4, min(cx - a.win_w / 2, a.framed_stride - 4 - a.win_w)) & ~3;
            w.y0 = max(-MCL_FRAME, min(cy - a.win_h / 2, a.frame.height + MCL_FRAME - a.win_h));
            w.w = a.win_w; w.h = a.win_h; w.stride = a.win_w;
            s_win = w;
        }
        __syncthreads();
        win = s_win;
    }
    if (MAP_MODE == 1) {
        win.stride = ((a.frame.width + 3) & ~3) + 8;
        s_map += MCL_FRAME * win.stride + 4;                  // cell (0, 0) of the framed image (grid_odds<1> indexes from it)
    }

    // ---- phase 1a: the staging waves (every wave when the prologue is not shared) bring the ray table and the map into LDS
    if (wave >= pw) {
        const int sw = wave - pw;
        const int st = tid - pw * 64, n_st = n_sw * 64;
        const int cnt0 = a.R < MCL_LDS_RAYS ? a.R : MCL_LDS_RAYS;
        for (int n = st; n < cnt0; n += n_st) s_ray[n] = ray_table_entry(a.ranges[n], a.thetas[n]);
        MCL_STAMP_T(8, pw * 64);                                 // this staging wave's ray-table entries are formed
        if (MAP_MODE == 1) {
            // the whole grid as a framed image: rows -MCL_FRAME..H+MCL_FRAME-1, columns -4..stride-5 (zeros outside)
            const int wq = win.stride >> 2;
            const int rows = a.frame.height + 2 * MCL_FRAME;
            if (a.framed && a.stage_dma) {
                // the framed copy in device memory, as it stands: whole 16-byte pieces, 64 per instruction (pf_launch_main has
                // checked the size); a piece is 16 contiguous bytes of both images
                const int pieces = (wq * rows) >> 2;
                const char* src = (const char*)(a.framed - MCL_FRAME * a.framed_stride - 4);
                for (int p0 = sw * 64; p0 < pieces; p0 += n_sw * 64)
                    if (p0 + lane < pieces)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)(p0 + lane) * 16),
                                                         (__attribute__((address_space(3))) void*)((char*)s_map32 + (size_t)p0 * 16), 16, 0, 0);
            } else if ((a.frame.width & 3) == 0 && (a.frame.width >> 2) <= 64 && a.stage_dma) {
                // Rows of up to 64 dwords by LDS-DMA: one global_load_lds_dword per row (lane = dword column; the wave-uniform
                // destination is the row's first data dword), no register between the load and LDS, so EVERY row of a wave
                // is in flight at once -- the register-staged form below keeps six rows per wave in flight and took 8 us of a
                // workgroup's first 11 (tests/tools/mcl_timeline_probe.py); more rows in flight there cost registers the
                // ray loop needs.  Frame rows and the two frame columns are plain zero stores.  The barrier behind the
                // prologue waits for the DMAs (hipcc drains vmcnt in front of it).
                const int wdw = a.frame.width >> 2;
                const int H = a.frame.height;
                for (int ry = sw; ry < rows; ry += n_sw) {
                    const int y = ry - MCL_FRAME;
                    int* row = s_map32 + ry * wq;
                    if (y < 0 || y >= H) {
                        for (int q = lane; q < wq; q += 64) row[q] = 0;
                    } else {
                        if (lane == 0) { row[0] = 0; row[wdw + 1] = 0; }
                        if (lane < wdw)
                            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const int*)a.cells + (size_t)y * wdw + lane),
                                                             (__attribute__((address_space(3))) void*)(row + 1), 4, 0, 0);
                    }
                }
            } else if ((a.frame.width & 3) == 0) {
                const int wdw = a.frame.width >> 2;
                const int* cells32 = (const int*)a.cells;
                const int H = a.frame.height;
                stage_rows(s_map32, rows, wq, 1, 1 + wdw, sw, n_sw, lane, [&](int ry) -> const int* {
                    const int y = ry - MCL_FRAME;
                    return (y >= 0 && y < H) ? cells32 + (size_t)y * wdw : nullptr;
                });
            } else {
                for (int i = st; i < wq * rows; i += n_st) {    // rows that are not whole dwords: byte gathers
                    const int ry = i / wq, q = i - ry * wq;
                    const int y = ry - MCL_FRAME, x = 4 * q - 4;
                    int v = 0;
                    if (y >= 0 && y < a.frame.height && x >= 0 && x < a.frame.width) {
                        const size_t g = (size_t)y * a.frame.width + x;
                        for (int b = 0; b < 4; ++b)
                            if (x + b < a.frame.width) v |= ((int)(unsigned char)a.cells[g + b]) << (8 * b);
                    }
                    s_map32[i] = v;
                }
            }
        }
        if (MAP_MODE == 2) {
            const int wq = win.w >> 2, fq = a.framed_stride >> 2;
            const int* f32 = (const int*)(a.framed - MCL_FRAME * a.framed_stride - 4);   // first row, column -4 of the framed image
            const int* org = f32 + (size_t)(win.y0 + MCL_FRAME) * fq + ((win.x0 + 4) >> 2);
            if (wq <= 64 && a.stage_dma) {                          // the window lies inside the mirror: every row by LDS-DMA (see mode 1)
                if (lane < wq)
                    for (int ry = sw; ry < win.h; ry += n_sw)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(org + (size_t)ry * fq + lane),
                                                         (__attribute__((address_space(3))) void*)(s_map32 + ry * wq), 4, 0, 0);
            } else
            stage_rows(s_map32, win.h, wq, 0, wq, sw, n_sw, lane, [&](int ry) -> const int* { return org + (size_t)ry * fq; });
        }
    }

    MCL_STAMP_T(4, pw * 64);                                     // a staging wave is through (its loads ISSUED)
#ifdef MCL_STAMPS
    if (wave >= pw) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    MCL_STAMP_T(9, pw * 64);                                     // ... and landed (stamped build only: the shipped one waits at the barrier)
#endif
    // ---- phase 1b: per-particle prologue
    double t_units = 0, t_x = 0, t_y = 0, t_s = 0, t_c = 0;
    float2 pcs_own = make_float2(2.0f, 0.0f);               // (2, -): "take the hardware sine / cosine" (a.fast_trig == 2)
    float pth_sin = 0.0f, pth_cos = 1.0f;
    int i = mp;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    float px = 0.f, py = 0.f, pth = 0.f, sx0 = 0.f, sy0 = 0.f;
    // ---- resamplePosteriorDistribution (particle_filter.cpp:84-103): first index with T <= prefix[i], clamped to N-1
    double rs_T = 0.0;
    int rs_lo = 0, rs_hi = a.N - 1;
    const mcl_prefix_view pview = {a.sh, a.prefix};
    // equal weights (a fresh filter, every particle at the likelihood floor): U against the runs of the reference's own rounded
    // cumulative, no prefix (uni_seg, bl_mcl_finish.h).  The same value in every lane of the launch.
    const int uni_n = a.resample ? a.state->uni_n : 0;
    if (a.resample && (!shared_pro || wave < pw)) {                  // whole waves: the narrowing is cooperative
        if (pro_active) { rs_T = a.r + mp * a.M_inv; if (!a.strict && uni_n <= 0) rs_T *= a.state->S; }      // U (particle_filter.cpp:95), or U * S
        if (uni_n > 0) { }
        else if (a.sh) resample_bracket(pview, a.N, rs_T, pro_active, lane, &rs_lo, &rs_hi, a.strict != 0);
        else resample_bracket<MCL_BRACKET_Q>(a.prefix, a.N, rs_T, pro_active, lane, &rs_lo, &rs_hi, a.strict != 0, pro_active ? mp : -1);
    }
    MCL_STAMP(5);                                                // the bracket is known
    // (Round 6, measured and dropped: the bracket's entries -- usually ~100 -- brought into the wave's quarter of the particle table in
    // ONE round trip and the lanes' searches run on LDS instead of seven or eight dependent loads each: prologue done at 6.4 us after
    // the workgroup's entry against 6.3 -- those loads hit lines the last bracket round brought in; profiles/r06_mcl_timeline.txt.)
    if (pro_active) {
        if (a.resample && uni_n > 0) i = uni_search(a.state, uni_n, rs_T, a.N);
        else if (a.resample) i = a.sh ? resample_bisect(pview, rs_T, rs_lo, rs_hi, a.strict != 0) : resample_bisect(a.prefix, rs_T, rs_lo, rs_hi, a.strict != 0);
        MCL_STAMP(6);
        s = mcl_src_at(a, i);
        // ---- ActionModel::applyAction (action_model.cpp:78-103)
        float n1, n2, n3;
        if (a.noise) {
            n1 = a.noise[3 * jp]; n2 = a.noise[3 * jp + 1]; n3 = a.noise[3 * jp + 2];
        } else {
            float z[3];
            philox_normals3((uint32_t)mp, a.step, a.seed_lo, a.seed_hi, z);
            n1 = (float)(a.rot1 + a.rot1Std * (double)z[0]);
            n2 = (float)(a.trans + a.transStd * (double)z[1]);
            n3 = (float)(a.rot2 + a.rot2Std * (double)z[2]);
        }
        const float head = s.z + n1;                        // float sum, then double libm cos/sin of it
        double hs, hc;
        sincos((double)head, &hs, &hc);                     // one argument reduction for both
        px = (float)((double)s.x + (double)n2 * hc);
        py = (float)((double)s.y + (double)n2 * hs);
        pth = bl_wrap_to_pi(s.z + n1 + n3);
        if (a.cells) bl_global_to_grid(px, py, a.frame, &sx0, &sy0);
        bl_sincosf(pth, &pth_sin, &pth_cos);                 // (the estimate's partial sums need them anyway: particle_filter.cpp:151-152)
        if (a.cells && a.fast_trig == 1) pcs_own = make_float2(a.frame.cpm * pth_cos, a.frame.cpm * pth_sin);    // ("pcs scaled", ray_cells_fast)
    }
    // guard-band offset of the fast trig path for this particle: 2.04 u times a bound on its cell coordinates (ray_cells_fast)
    // (the part of the band that does not grow with the range: 2.04 u times the particle's own cell coordinates; what a ray adds to
    // the coordinates -- up to twice its length -- is in the per-metre coefficient, trig_k1 below)
    const float k2_own = 1.2159e-7f * (__builtin_fmaxf(__builtin_fabsf(sx0), __builtin_fabsf(sy0)) + 8.0f);
    MCL_STAMP(7);                                                // prologue arithmetic done
    if (shared_pro && tid < P) { s_pp[tid] = make_float4(pth, sx0, sy0, k2_own); s_pcs[tid] = pcs_own; }
    __syncthreads();                                            // map, ray table and particle table are in place
    MCL_STAMP(1);

    // what the ray loop needs of this lane's particle
    float r_pth = pth, r_sx0 = sx0, r_sy0 = sy0, trig_k2 = k2_own;
    float2 r_pcs = pcs_own;
    if (shared_pro) { const float4 e = s_pp[jl]; r_pth = e.x; r_sx0 = e.y; r_sy0 = e.z; trig_k2 = e.w; r_pcs = s_pcs[jl]; }
    const float2_t pcs_ = {r_pcs.x, r_pcs.y};
    // the far point's band per metre of range, in cells: 2 x 1.02 cpm (eps + 4u) for the direction and the products (4u = 2^-22; the
    // fused form needs 2u, the hardware-trig form 3u), + 3.06 u x 2 cpm for the roundings at coordinates that a ray moves by up to 2 cpm
    // cells per metre (6.12 u = 3.648e-7)
    const float trig_k1 = a.frame.cpm * (2.04f * (MCL_TRIG_EPS + 2.3842e-7f) + 3.648e-7f);
    const float trig_kh2 = 0.5f - 1.5f * trig_k2 - 1.2e-7f;                        // (ray_cells_fast: the band as a threshold on |fract - 0.5|; the far point's, formed as e + t)
    const bool fast_trig = a.fast_trig != 0;                                       // wave-uniform
    const bool hw_trig_ = a.fast_trig == 2;
    const int isx0 = (int)r_sx0, isy0 = (int)r_sy0;

    // ---- SensorModel::likelihood (sensor_model.cpp:14-25) over MovingLaserScan(scan, parent_pose, pose)
    int acc = 0;                                            // half-units: likelihood = acc / 2 exactly
    if (a.cells) {
        const bool theta_simple = a.theta_simple != 0;
        const bool pk_lane = !INTERP && a.pk_ok && isx0 >= -8191 && isx0 <= 8191 && isy0 >= -8191 && isy0 <= 8191;
        const short2_t S = {(short)isx0, (short)isy0};
        const float2_t start = {r_sx0, r_sy0};
        const float pth_r = r_pth;
        for (int base = 0; base < a.R; base += MCL_LDS_RAYS) {      // one pass for scans of up to MCL_LDS_RAYS kept rays
            const int cnt = a.R - base < MCL_LDS_RAYS ? a.R - base : MCL_LDS_RAYS;
            if (base > 0) {
                __syncthreads();                                    // every lane is done with the previous chunk
                for (int n = tid; n < cnt; n += BLOCK) s_ray[n] = ray_table_entry(a.ranges[base + n], a.thetas[base + n]);
                __syncthreads();
            }
            if (!active) continue;
            if (MAP_MODE == 1 && pk_lane) {
                pk_map pm;
                pm.base = vgpr_of((int)(unsigned int)(size_t)s_map);
                pm.K = __builtin_bit_cast(short2_t, vgpr_of(1 | (win.stride << 16)));
                pm.hi = short2_t{(short)(a.frame.width + 1), (short)(a.frame.height + 1)};
                if (fast_trig) MCL_RAY_LOOP_FAST(pm); else
                MCL_RAY_LOOP(score_ray_pk(pm, start, S, a.frame.cpm, rt.x, cs, sn));
            } else if (MAP_MODE == 2 && pk_lane) {
                pk_map_window pm;
                pm.base = vgpr_of((int)(unsigned int)(size_t)s_map);
                pm.K = __builtin_bit_cast(short2_t, vgpr_of(1 | (win.stride << 16)));
                pm.org = short2_t{(short)win.x0, (short)win.y0};
                pm.hi = short2_t{(short)(win.w - 2), (short)(win.h - 2)};
                pm.g.base = a.framed;
                pm.g.K = short2_t{(short)1, (short)a.framed_stride};
                pm.g.hi = short2_t{(short)(a.frame.width + 1), (short)(a.frame.height + 1)};
                if (fast_trig) MCL_RAY_LOOP_FAST(pm); else
                MCL_RAY_LOOP(score_ray_pk_window(pm, start, S, a.frame.cpm, rt.x, cs, sn));
            } else if (MAP_MODE == 0 && pk_lane && a.framed) {
                pk_map_global pm;
                pm.base = a.framed;
                pm.K = short2_t{(short)1, (short)a.framed_stride};
                pm.hi = short2_t{(short)(a.frame.width + 1), (short)(a.frame.height + 1)};
                if (fast_trig) MCL_RAY_LOOP_FAST(pm); else
                MCL_RAY_LOOP(score_ray_pk(pm, start, S, a.frame.cpm, rt.x, cs, sn));
            } else {
                const bl_pose3 pb = {s.x, s.y, s.z};            // INTERP: the prologue is not shared, these are this lane's own
                const bl_pose3 pe = {px, py, pth};
                for (int n = sub; n < cnt; n += split) {    // the host uploads only rays with range > 0.15f (moving_laser_scan.cpp:24)
                    const float4 rt = s_ray[n];
                    float theta, sx, sy;
                    int isx, isy;
                    if (INTERP) {
                        bl_pose3 rp = bl_interpolate_pose(pb, pe, bl_interp_ratio(a.times[base + n], a.t_begin, a.t_den));
                        theta = bl_wrap_to_pi(rp.theta - rt.y);
                        bl_global_to_grid(rp.x, rp.y, a.frame, &sx, &sy);
                        isx = (int)sx; isy = (int)sy;
                    } else {
                        theta = bl_wrap_to_pi(pth_r - rt.y);
                        sx = r_sx0; sy = r_sy0; isx = isx0; isy = isy0;
                    }
                    float sn, cs;
                    bl_sincosf(theta, &sn, &cs);
                    acc += score_ray_half_units<MAP_MODE>(a.cells, s_map, win, a.frame, sx, sy, isx, isy, rt.x, cs, sn);
                }
            }
        }
    }
    MCL_STAMP(2);
    if (active)
        for (int off = split >> 1; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);     // exact integer sum over the group
    if (shared_pro) {
        if (active && sub == 0) s_acc[jl] = acc;
        __syncthreads();
        if (pro_active) acc = s_acc[tid];
    }
    if (shared_pro ? pro_active : (active && sub == 0)) {
        // ---- computeNormalizedPosterior (particle_filter.cpp:116-141): w = max(likelihood, 0.001) in units of 0.0005
        const uint32_t units = a.cells ? (acc > 0 ? (uint32_t)acc * 1000u : 2u) : __float_as_uint(s.w);
        a.dst[mp] = make_float4(px, py, pth, __uint_as_float(units));
        a.parent[jp] = make_float4(s.x, s.y, s.z, 0.0f);
        if (a.dbg_idx) { a.dbg_idx[jp] = i; a.dbg_like[jp] = acc; }
        // ---- estimatePosteriorPose (particle_filter.cpp:144-160) partial sums
        const float sth = pth_sin, cth = pth_cos;           // sinf / cosf of the particle's heading, from the prologue
        t_units = (double)units;
        t_x = t_units * (double)px;
        t_y = t_units * (double)py;
        t_s = t_units * (double)sth;
        t_c = t_units * (double)cth;
    }
    // the workgroup's five sums: lanes in a wave by shuffles, then the waves that hold particles in order
    const int nred = shared_pro ? pw : BLOCK / 64;
    if (wave < nred) {
        t_units = wave_sum(t_units); t_x = wave_sum(t_x); t_y = wave_sum(t_y); t_s = wave_sum(t_s); t_c = wave_sum(t_c);
        if (lane == 0) { s_part[wave][0] = t_units; s_part[wave][1] = t_x; s_part[wave][2] = t_y; s_part[wave][3] = t_s; s_part[wave][4] = t_c; }
    }
    __syncthreads();
    if (tid < 5) {
        double v = 0;
        for (int w = 0; w < nred; ++w) v += s_part[w][tid];
        a.partials[(size_t)blockIdx.x * 5 + tid] = v;
    }
    MCL_STAMP(3);
    MCL_STAMP_HW();
}

// ---------------------------------------------------------------- weight-unit prefix scan over all N from the record (2 launches)
// Tile sums of the weight units and, when tile_partials != nullptr, the block sums k_mcl_main would have left for tiles of
// SCAN_TILE particles: units, units*(x, y, sinf(theta), cosf(theta)) in double (the finish takes the unit sums as exact integers,
// theta from the sin / cos sums and the binade predictions of the x / y accumulators from the others).  The order of every
// addition is a function of N alone (items in a thread, shuffle tree in a wave, waves in order), so the estimate does not
// depend on how the particles were sharded: after the all-gather every rank derives it from the record itself.
// (tile0: the first tile this launch sums -- a rank of a composed finish sums the tiles of its own block only; clear_word: a word
// the launch zeroes on the way, that rank's table counter)
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_tile_sums(const float4* __restrict__ rec, int N,
                                                                 unsigned long long* __restrict__ tile_sums,
                                                                 double* __restrict__ tile_partials, int tile0 = 0,
                                                                 unsigned long long* clear_word = nullptr)
{
    __shared__ unsigned long long s[SCAN_THREADS / 64];
    __shared__ double s_pose[SCAN_THREADS / 64][4];
    if (clear_word && blockIdx.x == 0 && threadIdx.x == 0) *clear_word = 0ull;
    const int tile = tile0 + (int)blockIdx.x;
    const int base = tile * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    unsigned long long v = 0;
    double e[4] = {0, 0, 0, 0};
    if (tile_partials) {
        for (int k = 0; k < SCAN_ITEMS; ++k)
            if (base + k < N) {
                const float4 r = rec[base + k];
                const uint32_t u = __float_as_uint(r.w);
                v += u;
                float sth, cth;
                bl_sincosf(r.z, &sth, &cth);
                const double du = (double)u;
                e[0] += du * (double)r.x; e[1] += du * (double)r.y; e[2] += du * (double)sth; e[3] += du * (double)cth;
            }
        for (int k = 0; k < 4; ++k) e[k] = wave_sum(e[k]);
    } else {
        for (int k = 0; k < SCAN_ITEMS; ++k)
            if (base + k < N) v += __float_as_uint(rec[base + k].w);
    }
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) {
        s[threadIdx.x >> 6] = v;
        for (int k = 0; k < 4; ++k) s_pose[threadIdx.x >> 6][k] = e[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int w = 0; w < SCAN_THREADS / 64; ++w) t += s[w];
        tile_sums[tile] = t;
        if (tile_partials) tile_partials[(size_t)tile * 5] = (double)t;
    }
    if (tile_partials && threadIdx.x < 4) {
        double t = 0;
        for (int w = 0; w < SCAN_THREADS / 64; ++w) t += s_pose[w][threadIdx.x];
        tile_partials[(size_t)tile * 5 + 1 + threadIdx.x] = t;
    }
}

// Prefix-only scan (particle uploads, initialisation, action-only updates: no estimate is formed): workgroup b sums the unit
// totals of the tiles before it (exact integers), scans its tile and writes the prefix; workgroup 0 records the unit total.
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_write_prefix(const float4* __restrict__ rec, int N,
                                                                    const unsigned long long* __restrict__ tile_sums, int ntiles,
                                                                    unsigned long long* __restrict__ prefix, pf_state* state, int uni_mode, double w_floor)
{
    __shared__ unsigned long long s_wave[SCAN_THREADS / 64];
    __shared__ unsigned long long s_off[SCAN_THREADS / 64];
    __shared__ unsigned long long s_tot[SCAN_THREADS / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long before = 0, all = 0;
    for (int j = threadIdx.x; j < ntiles; j += SCAN_THREADS) {
        const unsigned long long t = tile_sums[j];
        if (j < (int)blockIdx.x) before += t;
        all += t;
    }
    for (int off = 32; off > 0; off >>= 1) { before += __shfl_xor(before, off, 64); all += __shfl_xor(all, off, 64); }
    if (lane == 0) { s_off[wave] = before; s_tot[wave] = all; }
    const int base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    unsigned long long loc[SCAN_ITEMS];
    unsigned long long run = 0;
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        if (base + k < N) run += __float_as_uint(rec[base + k].w);
        loc[k] = run;
    }
    unsigned long long incl = run;
    for (int off = 1; off < 64; off <<= 1) {
        unsigned long long t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    unsigned long long off0 = incl - run, total = 0;
    for (int w = 0; w < SCAN_THREADS / 64; ++w) { off0 += s_off[w]; total += s_tot[w]; if (w < wave) off0 += s_wave[w]; }
    for (int k = 0; k < SCAN_ITEMS; ++k)
        if (base + k < N) prefix[base + k] = off0 + loc[k];
    if (blockIdx.x == 0 && threadIdx.x == 0) { state->S = (double)total; uni_update(state, N, (double)total, uni_mode, w_floor); }     // (equal weights: bl_mcl_finish.h, uni_seg)
}

// Stand-alone launch of the end of an update (bl_mcl_finish.h): workgroup 0 is the finisher (estimatePosteriorPose,
// particle_filter.cpp:144-160), workgroups 1.. are the groups (weight prefix + sub-tile records).  k_map_update carries the
// same functions when the end rides in the map update's launch.
__global__ __launch_bounds__(MCLF_WG) void k_mcl_finish(mcl_finish_args f)
{
    __shared__ mclf_smem sm;
    if (f.sh && __hip_atomic_load(&f.state->shard_broken, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;     // (an exchange of this set gave up: k_shard_wait)
    extern __shared__ __align__(16) char s_fin_scratch[];                  // MCLF_LDS_BYTES (the groups do not touch it)
    if (blockIdx.x == 0) mclf_pose(f, sm, s_fin_scratch, (size_t)MCLF_LDS_BYTES);
    else if (blockIdx.x == 1) mclf_pre_chain(f, sm);
    else mclf_prefix_group(f, (int)blockIdx.x - MCLF_EXTRA_WGS, sm);
}

// The groups of one rank of a composed finish (bl_mcl_finish.h, mclf_shards): weight prefix of its own particles -- global values:
// the tile sums of every rank are here by then -- and their sub-tile records and tables, left in the rank's exchange block.
__global__ __launch_bounds__(MCLF_WG) void k_shard_groups(mcl_finish_args f, int group0)
{
    __shared__ mclf_smem sm;
    if (__hip_atomic_load(&f.state->shard_broken, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    mclf_prefix_group(f, group0 + (int)blockIdx.x, sm);
}

// resamplePosteriorDistribution alone (particle_filter.cpp:84-103): the source index of every output particle, by exactly the
// search k_mcl_main runs (diagnostic entry bl_pf_debug_resample)
__global__ __launch_bounds__(256) void k_pf_resample_only(const unsigned long long* __restrict__ prefix, const pf_state* __restrict__ state,
                                                          int N, double r, double M_inv, int strict, int32_t* __restrict__ out)
{
    const int m = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63;
    const bool on = m < N;
    double T = on ? r + m * M_inv : 0.0;
    const int uni_n = state->uni_n;                                 // equal weights: the runs of the reference's cumulative (uni_seg)
    if (uni_n > 0) { if (on) out[m] = uni_search(state, uni_n, T, N); return; }
    if (!strict) T *= state->S;
    int lo, hi;
    resample_bracket(prefix, N, T, on, lane, &lo, &hi, strict != 0);
    if (on) out[m] = resample_bisect(prefix, T, lo, hi, strict != 0);
}

// ---- strict resampling: the reference's cumulative weight, bit for bit.
// resamplePosteriorDistribution (particle_filter.cpp:84-103) compares U with c, c = w_0, then c += w_i: a SEQUENTIALLY ROUNDED
// double sum of the normalised weights w_i = fl64(units_i / S).  One wave reproduces every c_i: inside a binade the sum is an
// integer prefix of quantized terms (bl_serial_sum.h, double form), a step that leaves the binade or ties is taken in real
// arithmetic (strict_chunk).  Strict mode only (bl_pf_set_strict_resampling).  One wave walking the whole sum took 0.7 ms at 100k
// particles and 7.3 ms at 1M; the chunks side by side (k_strict_records / _chain / _fill below) take ~50 / ~140 us.
__device__ __forceinline__ long long mclf_scan_add_i64(long long v)
{
#define MCLF_STEP(C, R) { const int lo_ = mclf_dpp<C, R>(0, (int)(unsigned int)(unsigned long long)v);                       \
                          const int hi_ = mclf_dpp<C, R>(0, (int)(unsigned int)((unsigned long long)v >> 32));                 \
                          v += (long long)(((unsigned long long)(unsigned int)hi_ << 32) | (unsigned long long)(unsigned int)lo_); }
    MCLF_DPP_STEPS(MCLF_STEP)
#undef MCLF_STEP
    return v;
}
__device__ __forceinline__ long long mclf_readlane_i64(long long v, int lane)
{
    const int lo = __builtin_amdgcn_readlane((int)(unsigned int)(unsigned long long)v, lane);
    const int hi = __builtin_amdgcn_readlane((int)(unsigned int)((unsigned long long)v >> 32), lane);
    return (long long)(((unsigned long long)(unsigned int)hi << 32) | (unsigned long long)(unsigned int)lo);
}

// One chunk of up to 128 terms (lane l holds terms 2l, 2l + 1 as w[0], w[1]) stepped from the true accumulator `acc` (wave-uniform):
// in-binade integer prefix sums up to the first step that leaves the binade, ties or is too large, that step in real arithmetic,
// and on.  out: where c_i of the chunk's term i goes (out[i]), or null.  Returns the accumulator behind the chunk.
__device__ __forceinline__ double strict_chunk(double acc, const double (&w)[2], int n, int lane, double* __restrict__ out)
{
    int pos = 0;
    while (pos < n) {
        const int key = __builtin_amdgcn_readfirstlane(ssd_key(acc));
        if (!key) {                                            // the very first term (c = w_0), or nothing usable
            acc = ssd_exact_step(acc, mclf_readlane_f64((pos & 1) ? w[1] : w[0], pos >> 1));
            if (out && lane == (pos >> 1)) out[pos] = acc;
            pos++;
            continue;
        }
        const long long M = ssd_mag(acc);
        const ssd_bin b = ssd_bin_of(key);
        long long p[2], run = 0;
        int bad[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int i = 2 * lane + k;
            int bk = 0;
            const long long d = ssd_quantize(b, w[k], &bk);
            const bool on = i >= pos && i < n;
            run += on ? d : 0; bad[k] = on ? bk : 0;
            p[k] = run;
        }
        const long long excl = mclf_scan_add_i64(run) - run;
        long long Mi[2];
        bool ex[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int i = 2 * lane + k;
            Mi[k] = M + excl + p[k];
            ex[k] = i >= pos && i < n && (bad[k] || Mi[k] <= SSD_MLO || Mi[k] >= SSD_MHI);
        }
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(ex[0] || ex[1]);
        int j = n;
        if (mask) {
            const int fl = __ffsll((long long)mask) - 1;
            j = 2 * fl + (__builtin_amdgcn_readlane(ex[0] ? 1 : 0, fl) ? 0 : 1);
        }
        if (out) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int i = 2 * lane + k;
                if (i >= pos && i < j) out[i] = ssd_from(key, Mi[k]);
            }
        }
        if (j == n) { acc = ssd_from(key, mclf_readlane_i64((n - 1) & 1 ? Mi[1] : Mi[0], (n - 1) >> 1)); break; }
        double before = acc;
        if (j > pos) before = ssd_from(key, mclf_readlane_i64((j - 1) & 1 ? Mi[1] : Mi[0], (j - 1) >> 1));
        acc = ssd_exact_step(before, mclf_readlane_f64((j & 1) ? w[1] : w[0], j >> 1));
        if (out && lane == (j >> 1)) out[j] = acc;
        pos = j + 1;
    }
    return acc;
}

// the chunk's two terms of a lane: w_i = fl64(units_i / S) (particle_filter.cpp:134-141 normalises, :94-99 accumulates)
__device__ __forceinline__ void strict_terms(const float4* __restrict__ rec, int N, int base, int lane, double S, double (&w)[2])
{
    const int i = base + 2 * lane;
    w[0] = i < N ? (double)__float_as_uint(rec[i].w) / S : 0.0;
    w[1] = i + 1 < N ? (double)__float_as_uint(rec[i + 1].w) / S : 0.0;
}

// The whole sum by ONE wave, chunk after chunk: small particle sets (below STRICT_PAR_MIN), and the form the three kernels below are
// checked against.
__global__ __launch_bounds__(64) void k_pf_cumulative_strict(const float4* __restrict__ rec, int N, const pf_state* __restrict__ state,
                                                             double* __restrict__ out)
{
    const int lane = threadIdx.x;
    const double S = state->S;
    double acc = 0.0;                                              // wave-uniform
    for (int base = 0; base < N; base += 128) {
        double w[2];
        strict_terms(rec, N, base, lane, S, w);
        acc = strict_chunk(acc, w, min(128, N - base), lane, out + base);
    }
}

// ---- the same sum with the chunks side by side.  The weights are positive, so the sum only
// grows: it stays in one binade for long stretches (half of all particles lie in the last one) and changes binade ~17 times in all.
//   A  k_strict_records   a wave per chunk: with the binade the chunk's start is PREDICTED to lie in -- from the exact integer prefix of
//                         the weight units, prefix[base - 1] / S -- the chunk's terms as integers of that binade's ulp, their sum D,
//                         and "bad" if any term ties or is too large there
//   B  k_strict_chain     one wave walks the chunks' records with the TRUE accumulator, 64 at a time: a run of chunks whose predicted
//                         binade is the accumulator's, without bad terms and ending below the binade's end, is an integer prefix sum
//                         (every step in it rounds to the ulp without a tie, so c advances by exactly its integer); the first chunk
//                         that is not (a crossing, a tie, a misprediction, chunk 0 with the sum's first steps) is stepped from its
//                         terms (strict_chunk) and the walk goes on behind it.  Leaves every chunk's true start.
//   C  k_strict_fill      a wave per chunk: strict_chunk from the true start, c_i written out
// prefix[] is read by A (integers) and overwritten by C (doubles).
#define STRICT_PAR_MIN 4096
struct strict_rec { long long D; int key; int bad; };

__global__ __launch_bounds__(256) void k_strict_records(const float4* __restrict__ rec, int N, const pf_state* __restrict__ state,
                                                        const unsigned long long* __restrict__ prefix, strict_rec* __restrict__ recs,
                                                        double* __restrict__ first)
{
    const int lane = threadIdx.x & 63;
    const int t = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    const int base = t * 128;
    if (base >= N) return;
    const int n = min(128, N - base);
    const double S = state->S;
    strict_rec r; r.D = 0; r.key = 0; r.bad = 1;
    if (t == 0) {
        // the sum's first chunk starts from zero and crosses a binade every few terms (a phase of strict_chunk each: 7 us in all): its
        // 128 terms are simply added one after the other here, beside the other chunks' records (1.3 us), and kept for k_strict_fill;
        // the record is the accumulator behind the chunk (bad = 2)
        double w[2];
        strict_terms(rec, N, 0, lane, S, w);
        double acc = 0.0, k0 = 0.0, k1 = 0.0;                    // (terms beyond n are zeros: the sum simply stays)
#pragma unroll 4
        for (int l = 0; l < 64; ++l) {
            acc = ssd_exact_step(acc, mclf_readlane_f64(w[0], l));
            k0 = lane == l ? acc : k0;
            acc = ssd_exact_step(acc, mclf_readlane_f64(w[1], l));
            k1 = lane == l ? acc : k1;
        }
        if (2 * lane < n) first[2 * lane] = k0;
        if (2 * lane + 1 < n) first[2 * lane + 1] = k1;
        r.D = __double_as_longlong(acc); r.bad = 2;
    } else {
        double w[2];
        strict_terms(rec, N, base, lane, S, w);
        const int key = ssd_key((double)prefix[base - 1] / S);
        if (key) {
            const ssd_bin b = ssd_bin_of(key);
            int bad = 0;
            long long d = 0;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                int bk = 0;
                const long long q = ssd_quantize(b, w[k], &bk);
                if (2 * lane + k < n) { d += q; bad |= bk; }
            }
            const long long tot = mclf_readlane_i64(mclf_scan_add_i64(d), 63);
            r.D = tot; r.key = key; r.bad = __builtin_amdgcn_ballot_w64(bad != 0) != 0ull ? 1 : 0;
        }
    }
    if (lane == 0) recs[t] = r;
}

__global__ __launch_bounds__(64) void k_strict_chain(const float4* __restrict__ rec, int N, const pf_state* __restrict__ state,
                                                     const strict_rec* __restrict__ recs, double* __restrict__ starts)
{
    const int lane = threadIdx.x;
    const double S = state->S;
    const int nchunks = (N + 127) / 128;
    double acc = 0.0;                                              // wave-uniform: the true accumulator in front of chunk g0 + pos
    strict_rec nxt; nxt.D = 0; nxt.key = 0; nxt.bad = 1;
    if (lane < nchunks) nxt = recs[lane];
    for (int g0 = 0; g0 < nchunks; g0 += 64) {
        const int c = g0 + lane;
        strict_rec r = nxt;
        nxt.D = 0; nxt.key = 0; nxt.bad = 1;
        if (c + 64 < nchunks) nxt = recs[c + 64];                   // (the next 64 records travel while these are walked)
        const int cnt = min(64, nchunks - g0);
        int pos = 0;
        if (g0 == 0) {                                              // chunk 0 was stepped by k_strict_records: its record is the sum behind it
            if (lane == 0) starts[0] = 0.0;
            acc = __longlong_as_double(mclf_readlane_i64(r.D, 0));
            pos = 1;
        }
        while (pos < cnt) {
            const int key = __builtin_amdgcn_readfirstlane(ssd_key(acc));
            const long long M = key ? ssd_mag(acc) : 0;
            const bool ok = key != 0 && lane >= pos && lane < cnt && !r.bad && r.key == key;
            const long long Dm = ok ? r.D : 0;
            const long long incl = mclf_scan_add_i64(Dm);
            const long long Mi = M + incl;
            const bool fail = lane >= pos && (!ok || Mi >= SSD_MHI);
            const unsigned long long mask = __builtin_amdgcn_ballot_w64(fail);
            const int j = __builtin_amdgcn_readfirstlane(mask ? __ffsll((long long)mask) - 1 : 64);      // (lanes from cnt on fail: j <= cnt)
            if (lane >= pos && lane < j) starts[c] = ssd_from(key, Mi - Dm);
            if (j > pos) acc = ssd_from(key, mclf_readlane_i64(Mi, j - 1));
            if (j < cnt) {                                          // chunk g0 + j does not go by its record: from its terms
                if (lane == 0) starts[g0 + j] = acc;
                const int base = (g0 + j) * 128;
                double w[2];
                strict_terms(rec, N, base, lane, S, w);
                acc = strict_chunk(acc, w, min(128, N - base), lane, nullptr);
                pos = j + 1;
            } else pos = cnt;
        }
    }
}

__global__ __launch_bounds__(256) void k_strict_fill(const float4* __restrict__ rec, int N, const pf_state* __restrict__ state,
                                                     const double* __restrict__ starts, const double* __restrict__ first,
                                                     double* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int t = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    const int base = t * 128;
    if (base >= N) return;
    if (t == 0) {                                                   // (stepped by k_strict_records: first[])
        for (int i = lane; i < min(128, N); i += 64) out[i] = first[i];
        return;
    }
    double w[2];
    strict_terms(rec, N, base, lane, state->S, w);
    (void)strict_chunk(starts[t], w, min(128, N - base), lane, out + base);
}

// ---------------------------------------------------------------- init / export / small state kernels
__global__ void k_pf_init(float4* rec, float4* parent, int N, int lo, int n_local, bl_pose_xyt_t pose, uint32_t k0, uint32_t k1)
{
    // initializeFilterAtPose (particle_filter.cpp:16-34): every rank fills the WHOLE record (it is replicated)
    int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= N) return;
    float z[3];
    philox_normals3((uint32_t)m, 0xffffffffu, k0, k1, z);
    float x = (float)((double)pose.x + 0.01 * (double)z[0]);
    float y = (float)((double)pose.y + 0.01 * (double)z[1]);
    float th = bl_wrap_to_pi((float)((double)pose.theta + 0.01 * (double)z[2]));
    if (m == N - 1) { x = pose.x; y = pose.y; th = pose.theta; }     // posterior_.back().pose = pose
    rec[m] = make_float4(x, y, th, __uint_as_float(1u));
    int j = m - lo;
    if (j >= 0 && j < n_local) parent[j] = make_float4(x, y, th, 0.0f);
}

__global__ void k_pf_export(const float4* rec, const float4* parent, const pf_state* state, int lo, int n_local,
                            int64_t pose_utime, int64_t parent_utime, bl_particle_t* out)
{
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_local) return;
    float4 r = rec[lo + j];
    float4 p = parent[j];
    bl_particle_t o;
    memset(&o, 0, sizeof(o));                                // the struct's padding bytes leave the device defined (zero)
    o.pose.utime = pose_utime; o.pose.x = r.x; o.pose.y = r.y; o.pose.theta = r.z;
    o.parent_pose.utime = parent_utime; o.parent_pose.x = p.x; o.parent_pose.y = p.y; o.parent_pose.theta = p.z;
    o.weight = (double)__float_as_uint(r.w) / state->S;
    out[j] = o;
}

__device__ __forceinline__ uint32_t bl_bswap32(uint32_t v) { return __builtin_bswap32(v); }

// particles_t body on the LCM wire: 48 bytes = 12 big-endian dwords per particle (pose utime hi/lo, x, y, theta; parent
// the same; weight as a double, high dword first), one thread per dword so that a wave writes 256 consecutive bytes.
__global__ __launch_bounds__(256) void k_pf_encode_lcm(const float4* __restrict__ rec, const float4* __restrict__ parent,
                                                       const pf_state* __restrict__ state, int lo, int n_local,
                                                       int64_t pose_utime, int64_t parent_utime, uint32_t* __restrict__ out)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)n_local * 12) return;
    const int j = (int)(t / 12), d = (int)(t % 12);
    uint32_t v;
    if (d < 5) {
        const float4 r = rec[lo + j];
        v = d == 0 ? (uint32_t)((uint64_t)pose_utime >> 32) : d == 1 ? (uint32_t)pose_utime
          : __float_as_uint(d == 2 ? r.x : (d == 3 ? r.y : r.z));
    } else if (d < 10) {
        const float4 p = parent[j];
        v = d == 5 ? (uint32_t)((uint64_t)parent_utime >> 32) : d == 6 ? (uint32_t)parent_utime
          : __float_as_uint(d == 7 ? p.x : (d == 8 ? p.y : p.z));
    } else {
        const double w = (double)__float_as_uint(rec[lo + j].w) / state->S;     // the weight k_pf_export hands out
        const uint64_t b = (uint64_t)__double_as_longlong(w);
        v = d == 10 ? (uint32_t)(b >> 32) : (uint32_t)b;
    }
    out[t] = bl_bswap32(v);
}

__global__ void k_pf_set_pose(pf_state* state, bl_pose_xyt_t pose, int only_utime)
{
    if (only_utime) state->pose.utime = pose.utime;
    else state->pose = pose;
}

// ---------------------------------------------------------------- host side
static int pf_alloc(bl_pf* pf)
{
    size_t N = pf->N, n = pf->n_local;
    if (!pf->rec[0]) {
        BL_HIP(hipMalloc((void**)&pf->rec[0], N * sizeof(float4)));
        BL_HIP(hipMalloc((void**)&pf->rec[1], N * sizeof(float4)));
    }
    BL_HIP(hipMalloc((void**)&pf->prefix, N * sizeof(unsigned long long)));
    BL_HIP(hipMalloc((void**)&pf->parent, n * sizeof(float4)));
    BL_HIP(hipMalloc((void**)&pf->state, sizeof(pf_state)));
    int blocks = (int)((n * 64 + 255) / 256) + 1;          // worst case: every particle spread over a whole wave, 256-thread blocks
    BL_HIP(hipMalloc((void**)&pf->partials, (size_t)blocks * 5 * sizeof(double)));
    pf->partials_cap = blocks;
    static bool attr_set = false;
    if (!attr_set) {
        const int big = 384 * 384;                     // BOTLAB_MCL_WINDOW may ask for up to 384 cells a side
        BL_HIP(hipFuncSetAttribute((const void*)k_mcl_main<0, 256, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
        BL_HIP(hipFuncSetAttribute((const void*)k_mcl_main<0, 512, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
        BL_HIP(hipFuncSetAttribute((const void*)k_mcl_main<0, 1024, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
        BL_HIP(hipFuncSetAttribute((const void*)k_mcl_finish, hipFuncAttributeMaxDynamicSharedMemorySize, MCLF_LDS_BYTES));
        attr_set = true;
    }
    pf->scan_blocks = (int)((N + SCAN_TILE - 1) / SCAN_TILE);
    BL_HIP(hipMalloc((void**)&pf->block_sums, (size_t)pf->scan_blocks * sizeof(unsigned long long)));
    BL_HIP(hipMalloc((void**)&pf->tile_partials, (size_t)pf->scan_blocks * 5 * sizeof(double)));
    pf->fin_subs_cap = (int)(N / MCLF_SUB) + 4 * (MCLF_WG / 64);      // main region + tail region, each rounded up to whole groups
    BL_HIP(hipMalloc((void**)&pf->fin_recs, (size_t)2 * pf->fin_subs_cap * sizeof(ss_rec)));
    if (!getenv("BOTLAB_MCL_NO_WILD")) BL_HIP(hipMalloc((void**)&pf->fin_wild, (size_t)2 * pf->fin_subs_cap * sizeof(ss_wild)));
    BL_HIP(hipMalloc((void**)&pf->fin_tabs, (size_t)2 * MCLF_TSLOTS * MCLF_SUB * sizeof(mclf_tab_elem)));
    BL_HIP(hipMalloc((void**)&pf->fin_sync, MCLF_SYNC_WORDS * sizeof(unsigned long long)));
    BL_HIP(hipMemsetAsync(pf->fin_sync, 0, MCLF_SYNC_WORDS * sizeof(unsigned long long), pf->ctx->stream));
    BL_HIP(hipMalloc((void**)&pf->dbg_idx, n * sizeof(int32_t)));
    BL_HIP(hipMalloc((void**)&pf->dbg_like, n * sizeof(int32_t)));
    BL_HIP(hipMemsetAsync(pf->dbg_idx, 0, n * sizeof(int32_t), pf->ctx->stream));
    BL_HIP(hipMemsetAsync(pf->dbg_like, 0, n * sizeof(int32_t), pf->ctx->stream));
    BL_HIP(hipMemsetAsync(pf->state, 0, sizeof(pf_state), pf->ctx->stream));
    return BL_OK;
}

extern "C" int bl_pf_create(bl_ctx* ctx, int num_particles, int shard_lo, int shard_hi, bl_pf** out)
{
    BL_CHECK_ARG(ctx != nullptr && out != nullptr);
    BL_CHECK_ARG(num_particles > 1);                         // particle_filter.cpp:11
    BL_CHECK_ARG(shard_lo >= 0 && shard_lo < shard_hi && shard_hi <= num_particles);
    BL_HIP(hipSetDevice(ctx->device));
    bl_pf* pf = new bl_pf();
    memset((void*)pf, 0, sizeof(*pf));
    pf->ctx = ctx;
    pf->N = num_particles; pf->lo = shard_lo; pf->hi = shard_hi; pf->n_local = shard_hi - shard_lo;
    {   // computeNormalizedPosterior on N floor weights: wSum += 0.001 N times, then 0.001 / wSum (volatile: no vectorised reassociation)
        volatile double wsum = 0.0;
        for (int i = 0; i < num_particles; ++i) wsum = wsum + 0.001;
        pf->w_floor = 0.001 / wsum;
    }
    pf->noise_seed = 0x243F6A8885A308D3ull;
    pf->use_lds = getenv("BOTLAB_MCL_NO_LDS") == nullptr;
    pf->no_fused_finish = getenv("BOTLAB_MCL_NO_FUSED_FINISH") != nullptr;
    pf->no_packed = getenv("BOTLAB_MCL_NO_PACKED") != nullptr;
    pf->no_balance = getenv("BOTLAB_MCL_NO_BALANCE") != nullptr;
    pf->no_framed = getenv("BOTLAB_MCL_NO_FRAMED") != nullptr;
    pf->no_mirror_reuse = getenv("BOTLAB_MCL_NO_MIRROR_REUSE") != nullptr;
    pf->no_stage_dma = getenv("BOTLAB_MCL_NO_STAGE_DMA") != nullptr;
    pf->no_stage_x4 = getenv("BOTLAB_MCL_NO_STAGE_X4") != nullptr;
    pf->no_window = getenv("BOTLAB_MCL_NO_WINDOW") != nullptr;
    pf->no_fast_trig = getenv("BOTLAB_MCL_NO_FAST_TRIG") != nullptr;
    pf->window_override = getenv("BOTLAB_MCL_WINDOW") ? atoi(getenv("BOTLAB_MCL_WINDOW")) : 0;
    { int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device) == hipSuccess) pf->cus = cus; }
    pf->split_log2_override = getenv("BOTLAB_MCL_SPLIT_LOG2") ? atoi(getenv("BOTLAB_MCL_SPLIT_LOG2")) : -1;
    if (pf->split_log2_override > 6) pf->split_log2_override = 6;
    pf->block_override = getenv("BOTLAB_MCL_BLOCK") ? atoi(getenv("BOTLAB_MCL_BLOCK")) : 0;
    if (pf->block_override != 0 && pf->block_override != 256 && pf->block_override != 512 && pf->block_override != 1024) pf->block_override = 0;
    *out = pf;
    return BL_OK;
}

extern "C" void bl_pf_destroy(bl_pf* pf)
{
    if (!pf) return;
    (void)hipStreamSynchronize(pf->ctx->stream);
    if (!pf->rec_external) { if (pf->rec[0]) (void)hipFree(pf->rec[0]); if (pf->rec[1]) (void)hipFree(pf->rec[1]); }
    void* ptrs[] = {pf->fin_wild, pf->tile_partials, pf->fin_recs, pf->fin_tabs, pf->fin_sync, pf->prefix, pf->parent, pf->state, pf->partials, pf->block_sums, pf->dbg_idx, pf->dbg_like,
                    pf->d_noise, pf->d_export, pf->sh_xchg, pf->sh_tab, pf->sh_fin, pf->sh_flags, pf->sh_peers_dev, pf->strict_recs, pf->strict_starts};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    delete pf;
}

extern "C" int bl_pf_set_exchange_buffers(bl_pf* pf, void* d_rec0, void* d_rec1)
{
    BL_CHECK_ARG(pf != nullptr && d_rec0 && d_rec1);
    if (pf->prefix) { bl_set_error("exchange buffers must be set before the filter is initialised"); return BL_ERR_STATE; }
    pf->rec[0] = (float4*)d_rec0; pf->rec[1] = (float4*)d_rec1; pf->rec_external = true;
    return BL_OK;
}

extern "C" void* bl_pf_exchange_rec_ptr(bl_pf* pf) { return pf && pf->prefix ? (void*)pf->rec[pf->pending_end ? pf->cur ^ 1 : pf->cur] : nullptr; }
extern "C" const void* bl_pf_pose_device_ptr(bl_pf* pf) { return pf && pf->state ? (const void*)&pf->state->pose : nullptr; }

// what the launch that writes the weight total is told about equal weights (uni_update): 1 = the host knows the record's weights are
// all equal (only a launch over rec[cur] as uploaded / initialised may be told so), 0 = recognise the all-floor set, -1 = off
// (BOTLAB_NO_AUTO_STRICT=1: the integer rule everywhere -- tests, A/B runs)
static int pf_uni_mode(const bl_pf* pf, bool plain_scan)
{
    static const bool no_auto = getenv("BOTLAB_NO_AUTO_STRICT") != nullptr;
    if (no_auto) return -1;
    return plain_scan && pf->uniform_now ? 1 : 0;
}

// strict resampling: the integer prefix just written gives way to the reference's own cumulative (same buffer, as doubles)
static void pf_strict_cumulative(bl_pf* pf, int which)
{
    // (equal weights -- a fresh filter, an upload, an all-floor update -- no longer come here: the launch that writes the total leaves
    // the runs of the reference's cumulative and k_mcl_main searches those, uni_seg in bl_mcl_finish.h)
    pf->prefix_is_strict = pf->strict;
    if (!pf->prefix_is_strict) return;
    static const bool one_wave = getenv("BOTLAB_STRICT_ONE_WAVE") != nullptr;        // tests, A/B runs
    const int nchunks = (pf->N + 127) / 128;
    if (pf->N >= STRICT_PAR_MIN && !one_wave) {
        if (!pf->strict_recs) {
            if (hipMalloc((void**)&pf->strict_recs, (size_t)nchunks * sizeof(strict_rec)) != hipSuccess) pf->strict_recs = nullptr;
            if (pf->strict_recs && hipMalloc((void**)&pf->strict_starts, ((size_t)nchunks + 128) * sizeof(double)) != hipSuccess) { (void)hipFree(pf->strict_recs); pf->strict_recs = nullptr; }
        }
        if (pf->strict_recs) {
            hipStream_t st = pf->ctx->stream;
            hipLaunchKernelGGL(k_strict_records, dim3((nchunks + 3) / 4), dim3(256), 0, st, pf->rec[which], pf->N, pf->state, pf->prefix, (strict_rec*)pf->strict_recs, pf->strict_starts + nchunks);
            hipLaunchKernelGGL(k_strict_chain, dim3(1), dim3(64), 0, st, pf->rec[which], pf->N, pf->state, (const strict_rec*)pf->strict_recs, pf->strict_starts);
            hipLaunchKernelGGL(k_strict_fill, dim3((nchunks + 3) / 4), dim3(256), 0, st, pf->rec[which], pf->N, pf->state, pf->strict_starts, pf->strict_starts + nchunks, (double*)pf->prefix);
            return;
        }
    }
    hipLaunchKernelGGL(k_pf_cumulative_strict, dim3(1), dim3(64), 0, pf->ctx->stream, pf->rec[which], pf->N, pf->state, (double*)pf->prefix);
}

// what a finish launch needs beyond the block sums: the group shape, record and table space, the sync word.  Returns the group
// count or -1 when the blocks do not tile a group.
static int pf_finish_fill(bl_pf* pf, mcl_finish_args* f)
{
    if (f->tile <= 0 || f->tail_tile <= 0) return -1;
    f->gthreads = mclf_gthreads(pf->N);
    for (int attempt = 0; attempt < 2; ++attempt) {
        const int chunk = mclf_chunk(f->gthreads);
        if (f->tile <= chunk && (chunk % f->tile) == 0 && f->tail_tile <= chunk && (chunk % f->tail_tile) == 0) break;
        if (attempt == 1 || f->gthreads == MCLF_GT_LARGE) return -1;
        f->gthreads = MCLF_GT_LARGE;
    }
    const int groups = mclf_groups(*f);
    f->groups = groups;
    f->groups_wait = groups;
    // the records' generation tag (bl_mcl_finish.h): every launch stores every slot of its layout, so a slot shows the previous
    // launch's tag until its group is through.  Zeroed slots read as tag 0: generations that are multiples of 256 are skipped,
    // and a launch whose layout differs from the last one's (first launch, another launch shape) starts from zeroed slots.
    if (((++pf->fin_gen) & 0xffu) == 0u) ++pf->fin_gen;
    f->tag = pf->fin_gen;
    const int nrec = groups * (f->gthreads >> 6);
    if (pf->sh_world <= 1 && nrec != pf->fin_last_nrec && (int64_t)nrec <= pf->fin_subs_cap) {
        if (hipMemsetAsync(pf->fin_recs, 0, (size_t)2 * pf->fin_subs_cap * sizeof(ss_rec), pf->ctx->stream) != hipSuccess) return -1;
        pf->fin_last_nrec = nrec;
    }
    f->sh = nullptr;
    f->uni_mode = pf_uni_mode(pf, false);
    f->w_floor = pf->w_floor;
    f->wild = pf->sh_world > 1 ? nullptr : pf->fin_wild;      // (a composed finish keeps to records, tables and replays)
    f->no_trees = getenv("BOTLAB_MCL_NO_TREES") != nullptr ? 1 : 0;
    f->recs = pf->fin_recs;
    f->tabs = pf->fin_tabs;
    f->sync = pf->fin_sync;
    if (pf->sh_world > 1) return groups;                     // (a composed finish keeps its records in the exchange blocks)
    if ((int64_t)groups * (f->gthreads >> 6) > pf->fin_subs_cap) return -1;
    return groups;
}

// record-based end of an update, or a plain prefix scan of rec[which] (write_pose == 0); timed as BL_K_MCL_SCAN
static int pf_scan(bl_pf* pf, int which, int write_pose, int64_t utime)
{
    bl_ctx* ctx = pf->ctx;
    hipEvent_t e0, e1;
    int rc = bl_timer_begin(ctx, BL_K_MCL_SCAN, &e0, &e1);
    if (rc) return rc;
    hipLaunchKernelGGL(k_scan_tile_sums, dim3(pf->scan_blocks), dim3(SCAN_THREADS), 0, ctx->stream, pf->rec[which], pf->N,
                       pf->block_sums, write_pose ? pf->tile_partials : (double*)nullptr);
    if (write_pose) {
        // the tiles of k_scan_tile_sums stand where k_mcl_main's workgroups stand in the fused form
        mcl_finish_args f;
        f.partials = pf->tile_partials; f.nblocks = pf->scan_blocks;
        f.rec = pf->rec[which]; f.N = pf->N;
        f.tile = SCAN_TILE; f.main_blocks = pf->scan_blocks; f.main_particles = pf->N; f.tail_tile = 1;
        f.prefix = pf->prefix; f.state = pf->state; f.utime = utime;
        const int groups = pf_finish_fill(pf, &f);
        if (groups < 0) { bl_set_error("internal: finish launch shape"); return BL_ERR_STATE; }
        hipLaunchKernelGGL(k_mcl_finish, dim3(MCLF_EXTRA_WGS + f.groups_wait), dim3(MCLF_WG), MCLF_LDS_BYTES, ctx->stream, f);
    } else {
        hipLaunchKernelGGL(k_scan_write_prefix, dim3(pf->scan_blocks), dim3(SCAN_THREADS), 0, ctx->stream, pf->rec[which], pf->N,
                           pf->block_sums, pf->scan_blocks, pf->prefix, pf->state, pf_uni_mode(pf, true), pf->w_floor);
    }
    BL_HIP(hipGetLastError());
    pf_strict_cumulative(pf, which);
    return bl_timer_end(ctx, BL_K_MCL_SCAN, e0, e1);
}

static void pf_reset_action(bl_pf* pf)
{
    pf->action_initialized = false; pf->moved = false;
    pf->rot1 = pf->trans = pf->rot2 = 0;
    pf->step = 0;
    pf->pending_end = false;
}

extern "C" int bl_pf_init_at_pose(bl_pf* pf, const bl_pose_xyt_t* pose, uint64_t seed)
{
    BL_CHECK_ARG(pf != nullptr && pose != nullptr);
    BL_HIP(hipSetDevice(pf->ctx->device));
    if (!pf->prefix) { int rc = pf_alloc(pf); if (rc) return rc; }
    pf->cur = 0;
    if (pf->sh_broken) { pf->sh_broken = false; BL_HIP(hipMemsetAsync(&pf->state->shard_broken, 0, sizeof(unsigned int), pf->ctx->stream)); }
    hipLaunchKernelGGL(k_pf_init, dim3((pf->N + 255) / 256), dim3(256), 0, pf->ctx->stream, pf->rec[0], pf->parent, pf->N,
                       pf->lo, pf->n_local, *pose, (uint32_t)seed, (uint32_t)(seed >> 32));
    hipLaunchKernelGGL(k_pf_set_pose, dim3(1), dim3(1), 0, pf->ctx->stream, pf->state, *pose, 0);
    BL_HIP(hipGetLastError());
    pf->pose_utime = pose->utime; pf->parent_utime = pose->utime;
    pf->initialized = true;
    pf->uniform_now = true;                      // weights 1 / N (particle_filter.cpp:25)
    // the reference does not reset its ActionModel here; a filter is initialised once (slam.cpp:232-250)
    return pf_scan(pf, 0, 0, 0);
}

extern "C" int bl_pf_set_particles(bl_pf* pf, const bl_particle_t* particles, const uint32_t* units)
{
    BL_CHECK_ARG(pf != nullptr && particles != nullptr);
    BL_HIP(hipSetDevice(pf->ctx->device));
    if (!pf->prefix) { int rc = pf_alloc(pf); if (rc) return rc; }
    std::vector<float4> rec(pf->N), par(pf->n_local);
    pf->uniform_now = true;
    for (int m = 0; m < pf->N; ++m) {
        uint32_t u = units ? units[m] : 1u;
        if (units && units[m] != units[0]) pf->uniform_now = false;
        float w; memcpy(&w, &u, 4);
        rec[m] = make_float4(particles[m].pose.x, particles[m].pose.y, particles[m].pose.theta, w);
    }
    for (int j = 0; j < pf->n_local; ++j) {
        const bl_pose_xyt_t& pp = particles[pf->lo + j].parent_pose;
        par[j] = make_float4(pp.x, pp.y, pp.theta, 0.0f);
    }
    pf->cur = 0;
    BL_HIP(hipMemcpyAsync(pf->rec[0], rec.data(), rec.size() * sizeof(float4), hipMemcpyHostToDevice, pf->ctx->stream));
    BL_HIP(hipMemcpyAsync(pf->parent, par.data(), par.size() * sizeof(float4), hipMemcpyHostToDevice, pf->ctx->stream));
    BL_HIP(hipStreamSynchronize(pf->ctx->stream));
    pf->pose_utime = particles[0].pose.utime;
    pf->parent_utime = particles[0].parent_pose.utime;
    pf->initialized = true;
    pf->pending_end = false;
    // posteriorPose_ is not defined by a particle upload; the last particle's pose seeds it (it only centres the LDS
    // map window of the next update and is what poseEstimate() returns until that update)
    hipLaunchKernelGGL(k_pf_set_pose, dim3(1), dim3(1), 0, pf->ctx->stream, pf->state, particles[pf->N - 1].pose, 0);
    BL_HIP(hipGetLastError());
    return pf_scan(pf, 0, 0, 0);
}

extern "C" int bl_pf_get_particles(bl_pf* pf, bl_particle_t* out_local)
{
    BL_CHECK_ARG(pf != nullptr && out_local != nullptr);
    if (!pf->initialized || pf->pending_end) { bl_set_error("filter not initialised or update pending"); return BL_ERR_STATE; }
    BL_HIP(hipSetDevice(pf->ctx->device));
    if (!pf->d_export) BL_HIP(hipMalloc((void**)&pf->d_export, (size_t)pf->n_local * sizeof(bl_particle_t)));
    hipLaunchKernelGGL(k_pf_export, dim3((pf->n_local + 255) / 256), dim3(256), 0, pf->ctx->stream, pf->rec[pf->cur],
                       pf->parent, pf->state, pf->lo, pf->n_local, pf->pose_utime, pf->parent_utime, pf->d_export);
    BL_HIP(hipGetLastError());
    BL_HIP(hipMemcpyAsync(out_local, pf->d_export, (size_t)pf->n_local * sizeof(bl_particle_t), hipMemcpyDeviceToHost,
                          pf->ctx->stream));
    unsigned int broken = 0;
    if (pf->sh_world > 1) BL_HIP(hipMemcpyAsync(&broken, &pf->state->shard_broken, sizeof(broken), hipMemcpyDeviceToHost, pf->ctx->stream));
    BL_HIP(hipStreamSynchronize(pf->ctx->stream));
    if (broken) pf->sh_broken = true;
    if (pf->sh_broken) { bl_set_error("the shard exchange of this particle set gave up: its particles are not a valid posterior; set the shards up again or re-initialise the filter"); return BL_ERR_STATE; }
    return BL_OK;
}

// particles() + particles_t::encode of slam.cpp:265-268 in one step: the wire bytes are produced on the device (48 B per
// particle instead of the 56-byte host struct) and land in the caller's message buffer with one D2H.
extern "C" int64_t bl_pf_encode_particles_lcm(bl_pf* pf, int64_t utime, uint8_t* buf, int64_t cap)
{
    if (!pf || !buf) { bl_set_error("bad argument"); return -(int64_t)BL_ERR_ARG; }
    if (!pf->initialized || pf->pending_end) { bl_set_error("filter not initialised or update pending"); return -(int64_t)BL_ERR_STATE; }
    if (pf->sh_broken) { bl_set_error("the shard exchange of this particle set gave up: its particles are not a valid posterior"); return -(int64_t)BL_ERR_STATE; }
    const int64_t body = (int64_t)pf->n_local * 48, total = 20 + body;
    if (total > cap) { bl_set_error("LCM encode: %lld bytes do not fit the %lld-byte buffer", (long long)total, (long long)cap); return -(int64_t)BL_ERR_CAPACITY; }
    if (hipSetDevice(pf->ctx->device) != hipSuccess) return -(int64_t)BL_ERR_HIP;
    if (!pf->d_export) { if (hipMalloc((void**)&pf->d_export, (size_t)pf->n_local * sizeof(bl_particle_t)) != hipSuccess) { bl_set_error("hipMalloc failed"); return -(int64_t)BL_ERR_HIP; } }
    const long long dwords = (long long)pf->n_local * 12;
    hipLaunchKernelGGL(k_pf_encode_lcm, dim3((unsigned)((dwords + 255) / 256)), dim3(256), 0, pf->ctx->stream, pf->rec[pf->cur], pf->parent,
                       pf->state, pf->lo, pf->n_local, pf->pose_utime, pf->parent_utime, (uint32_t*)pf->d_export);
    // header: fingerprint, utime, num_particles (big-endian)
    const uint64_t fp = bl_lcm_fingerprint(BL_LCM_PARTICLES);
    for (int i = 0; i < 8; ++i) buf[i] = (uint8_t)(fp >> (56 - 8 * i));
    for (int i = 0; i < 8; ++i) buf[8 + i] = (uint8_t)((uint64_t)utime >> (56 - 8 * i));
    for (int i = 0; i < 4; ++i) buf[16 + i] = (uint8_t)((uint32_t)pf->n_local >> (24 - 8 * i));
    if (hipGetLastError() != hipSuccess || hipMemcpyAsync(buf + 20, pf->d_export, (size_t)body, hipMemcpyDeviceToHost, pf->ctx->stream) != hipSuccess ||
        hipStreamSynchronize(pf->ctx->stream) != hipSuccess) { bl_set_error("bl_pf_encode_particles_lcm: HIP call failed"); return -(int64_t)BL_ERR_HIP; }
    return total;
}

extern "C" int bl_pf_set_noise_seed(bl_pf* pf, uint64_t seed)
{
    BL_CHECK_ARG(pf != nullptr);
    pf->noise_seed = seed;
    return BL_OK;
}

// ActionModel::updateAction (action_model.cpp:22-75) -- host scalars, same libm calls as the reference
static bool action_update(bl_pf* pf, const bl_pose_xyt_t& odometry)
{
    if (!pf->action_initialized) { pf->prev_odom = odometry; pf->action_initialized = true; }
    float deltaX = odometry.x - pf->prev_odom.x;
    float deltaY = odometry.y - pf->prev_odom.y;
    float deltaTheta = (float)bl_angle_diff(odometry.theta, pf->prev_odom.theta);
    float dir = 1.0;
    pf->rot1 = bl_angle_diff(atan2f(deltaY, deltaX), pf->prev_odom.theta);
    pf->trans = sqrtf(deltaX * deltaX + deltaY * deltaY);
    if (fabs(pf->trans) < 0.0001) { pf->rot1 = 0.0f; }
    else if (fabs(pf->rot1) > BL_PI / 2.0) { pf->rot1 = -bl_angle_diff(BL_PI, pf->rot1); dir = -1.0; }
    // (the reference's third branch, |rot1| < -pi/2, can never be taken: action_model.cpp:44-47)
    pf->trans *= dir;
    pf->rot2 = bl_angle_diff(deltaTheta, pf->rot1);
    pf->moved = !((fabs(pf->trans) + fabs(pf->rot2)) < 0.00001f);
    pf->rot1Std = 0.05; pf->transStd = 0.005; pf->rot2Std = 0.05;
    pf->prev_odom = odometry;
    return pf->moved;
}

static int pf_upload_noise(bl_pf* pf, const float* noise)
{
    if (!pf->d_noise) BL_HIP(hipMalloc((void**)&pf->d_noise, (size_t)pf->n_local * 3 * sizeof(float)));
    BL_HIP(hipMemcpyAsync(pf->d_noise, noise + (size_t)pf->lo * 3, (size_t)pf->n_local * 3 * sizeof(float),
                          hipMemcpyHostToDevice, pf->ctx->stream));
    BL_HIP(hipStreamSynchronize(pf->ctx->stream));          // caller-owned pageable buffer
    return BL_OK;
}

static int pf_launch_main(bl_pf* pf, const bl_grid* map, int R, int rand_value, const float* noise, int resample)
{
    bl_ctx* ctx = pf->ctx;
    mcl_args a;
    a.src = pf->rec[pf->cur];
    a.dst = pf->rec[pf->cur ^ 1];
    a.prefix = pf->prefix;
    a.state = pf->state;
    a.parent = pf->parent;
    a.partials = pf->partials;
    a.dbg_idx = pf->debug ? pf->dbg_idx : nullptr; a.dbg_like = pf->debug ? pf->dbg_like : nullptr;
    a.cells = map ? map->cells : nullptr;
    if (map) a.frame = map->frame; else memset(&a.frame, 0, sizeof(a.frame));
    a.ranges = ctx->scan.ranges; a.thetas = ctx->scan.thetas; a.times = ctx->scan.times;
    // MovingLaserScan(scan, parent_pose, pose) (sensor_model.cpp:18): begin = parent utime, end = ActionModel::utime_ = 0 (D3)
    a.t_begin = pf->pose_utime; a.t_den = (pf->pose_utime != 0) ? (double)(0 - pf->pose_utime) : 1.0;
    a.R = R;
    a.N = pf->N; a.lo = pf->lo; a.n_local = pf->n_local;
    a.M_inv = 1.0 / pf->N;                                           // particle_filter.cpp:89
    a.r = (((double)rand_value) / (double)RAND_MAX) * a.M_inv;       // particle_filter.cpp:92
    a.rot1 = pf->rot1; a.trans = pf->trans; a.rot2 = pf->rot2;
    a.rot1Std = pf->rot1Std; a.transStd = pf->transStd; a.rot2Std = pf->rot2Std;
    a.noise = noise ? pf->d_noise : nullptr;
    a.seed_lo = (uint32_t)pf->noise_seed; a.seed_hi = (uint32_t)(pf->noise_seed >> 32);
    a.step = pf->step;
    a.resample = resample;
    a.strict = pf->prefix_is_strict ? 1 : 0;
    if (getenv("BOTLAB_MCL_DIAG_NOSEARCH")) a.resample = 0;
    // MovingLaserScan(scan, parent_pose, pose): parent_pose.utime is the particle's previous pose utime, pose.utime is
    // ActionModel::utime_ == 0 (D3); they differ only on the first moved update after initialisation.
    a.interp = (map && pf->pose_utime != 0) ? 1 : 0;
    // packed 16-bit scoring: grid up to 8192 a side, the longest kept ray spans at most 4000 cells (see score_ray_pk)
    a.theta_simple = ctx->scan.thetas_simple ? 1 : 0;
    a.max_range_cells = map ? ctx->scan.max_range * a.frame.cpm : 0.0f;
    static const bool hw_trig = getenv("BOTLAB_MCL_HW_TRIG") != nullptr;        // the round-3 form of the fast path (A/B runs, tests)
    a.fast_trig = (a.theta_simple && !pf->no_fast_trig) ? (hw_trig ? 2 : 1) : 0;
    a.stage_dma = pf->no_stage_dma ? 0 : 1;
    a.pk_ok = (map && a.frame.width <= 8192 && a.frame.height <= 8192 && ctx->scan.max_range * a.frame.cpm <= 4000.0f &&
               !pf->no_packed) ? 1 : 0;
    // Where the gathers go.  Mode 1: the whole grid, zero-framed, staged in LDS by every workgroup (grids up to 64 KB).
    // Larger grids: a zero-framed copy in device memory, made by k_mcl_frame in front of this launch, and
    //   mode 2: an LDS window of it around the predicted pose, the scan's reach plus MCL_WIN_MARGIN cells for the spread of
    //           the cloud on every side but at most MCL_WIN_MAX cells (cells outside it are gathered from the copy through L2), or
    //   mode 0: every gather through L2 (window switched off).
    // Without the packed path (longer rays, larger grids, the interpolating first update) mode 0 gathers from the grid itself.
    int lds_bytes = 0, mode = 0;
    a.win_w = 0; a.win_h = 0;
    a.framed = nullptr; a.framed_stride = 0;
    if (map) {
        const int W = map->frame.width, H = map->frame.height;
        const int stride = ((W + 3) & ~3) + 8;
        const size_t whole = (size_t)stride * (H + 2 * MCL_FRAME);      // framed image
        // the zero-framed copy in device memory, current as of the work enqueued on this stream (k_map_update keeps it so)
        auto ensure_mirror = [&]() -> int {
            if (whole > map->mirror_cap) {
                if (map->mirror) { BL_HIP(hipStreamSynchronize(ctx->stream)); BL_HIP(hipFree(map->mirror)); map->mirror = nullptr; map->mirror_cap = 0; }
                BL_HIP(hipMalloc((void**)&map->mirror, whole));
                map->mirror_cap = whole;
                map->mirror_valid = false;
            }
            if (!map->mirror_valid || map->mirror_stride != stride || map->mirror_external || pf->no_mirror_reuse) {
                const int dwords = (stride >> 2) * (H + 2 * MCL_FRAME);
                hipLaunchKernelGGL(k_mcl_frame, dim3((dwords + 255) / 256), dim3(256), 0, ctx->stream, map->cells, W, H, stride, (int*)map->mirror);
                map->mirror_stride = stride;
                map->mirror_valid = map->ctx == ctx;          // k_map_update keeps it current from here on (same stream only)
            }
            a.framed = map->mirror + MCL_FRAME * stride + 4;
            a.framed_stride = stride;
            return BL_OK;
        };
        if (pf->use_lds && whole <= MCL_WIN_SMALL_BYTES) {
            a.win_w = W; a.win_h = H;
            lds_bytes = (int)whole;
            mode = 1;
            // Round 6: the LDS image of mode 1 IS the framed copy, byte for byte -- so it is staged from that copy in 16-byte pieces
            // (global_load_lds_dwordx4: 1 KB per wave instruction, 42 instructions per workgroup) instead of row by row from the
            // grid in dwords (206 instructions of 200 bytes): the staging waves' loads were issue-bound, 35 per wave in 6 us
            // (profiles/r06_mcl_timeline.txt).  Needs the image to be whole 16-byte pieces and the map to live on this ctx's
            // stream (the copy is kept current by k_map_update there); BOTLAB_MCL_NO_STAGE_X4=1: the row form.
            if (!pf->no_stage_x4 && a.stage_dma && (whole & 15) == 0 && map->ctx == ctx && !map->mirror_external && !pf->no_framed && !pf->no_mirror_reuse) {
                int rc_m = ensure_mirror();
                if (rc_m) return rc_m;
            }
        } else if (a.pk_ok && !a.interp && !pf->no_framed) {
            { int rc_m = ensure_mirror(); if (rc_m) return rc_m; }
            // Measured at 100k-1M particles on 2000^2 / 4096^2 grids: with the rays inside it a 208-cell window is 8-22 %
            // faster than gathering everything through L2, with nearly every ray leaving it (8 m rays) it is within
            // -3 .. +6 %; larger windows lose more to occupancy (264: two workgroups per CU) than they gain in hits.
            const int reach = (int)ceilf(ctx->scan.max_range * a.frame.cpm) + MCL_WIN_MARGIN;
            int side = (2 * reach + 3) & ~3;
            if (side > MCL_WIN_MAX) side = MCL_WIN_MAX;
            if (pf->window_override > 0) side = (pf->window_override + 3) & ~3;
            if (side > 384) side = 384;
            if (pf->use_lds && !pf->no_window) {
                a.win_w = side < stride ? side : stride;
                a.win_h = side < H + 2 * MCL_FRAME ? side : H + 2 * MCL_FRAME;
                lds_bytes = a.win_w * a.win_h;
                mode = 2;
                // (Tried in round 4: the scan's rays packed longest first and the rounds of rays longer than the window's half side
                // scored from the framed image directly, without the window attempt and its ballot -- 0.1156 ms against 0.1133 at
                // 2000 x 2000 / 100k particles: what a large grid costs is the L2 gathers of the long rays themselves.)
            }
        }
    }
    // Launch shape.  Rays of one particle go over 2^split_log2 adjacent lanes: the smallest split that gives
    // >= MCL_MIN_BLOCKS workgroups (at most a wave).  512-thread workgroups (whole grid staged: 40 KB, up to 3 per CU).
    int block = 512;                             // measured at 100k and 1M particles, 200x200: 512 is within 3 % of the best
    if (pf->block_override > 0) block = pf->block_override;
    a.split_log2 = 0;
    if (map && pf->split_log2_override >= 0) a.split_log2 = pf->split_log2_override;
    else if (map) {
        while (a.split_log2 < 6 && ((int64_t)pf->n_local << a.split_log2) < (int64_t)MCL_MIN_BLOCKS * block) a.split_log2++;
        // ... and, up to four lanes per particle, until the launch is about two rounds of the machine: 256 000 particles at two
        // lanes are 1000 workgroups -- one round and a third -- and took 0.309 ms where four lanes (2000 workgroups) take 0.252
        while (a.split_log2 < 2 && ((int64_t)pf->n_local << a.split_log2) < (int64_t)1400 * block) a.split_log2++;
    }
    while (a.split_log2 > 0 && (1 << a.split_log2) > R) a.split_log2--;
    // Whole rounds.  All workgroups of this VALU-bound kernel take about the same time T, so 782 workgroups on a machine
    // that holds 768 at a time run for a full round plus a nearly empty one (measured: 98 304 particles 0.102 ms, 100 000
    // particles 0.128 ms; the straggling round lasts T/4, the time one wave needs for its particles' rays on an idle CU).
    // When the count exceeds whole rounds by less than ~60 % of a round, the excess particles therefore go to a second
    // region of the same launch, dispatched last, in which every particle has a whole wave (5 rays per lane instead of
    // 73): those workgroups last ~T/10.  One round is counted a little short of what the device holds (below), because the
    // replanner's kernels on the other streams occupy a few workgroup slots.
    const int gpb = block >> a.split_log2;                  // particles per region-1 workgroup
    int64_t main_blocks = ((int64_t)pf->n_local + gpb - 1) / gpb, tail_blocks = 0;
    int64_t main_particles = pf->n_local;
    const int tail_tile = block >> 6;                       // particles per region-2 workgroup
    if (map && !pf->no_balance && gpb >= 1) {
        const int cus = pf->cus > 0 ? pf->cus : 256;
        int per_cu = 32 / (block >> 6);
        const int lds_per_wg = lds_bytes + 9 * 1024;        // + static LDS (ray table, partial sums)
        if ((160 * 1024) / lds_per_wg < per_cu) per_cu = (160 * 1024) / lds_per_wg;
        if (per_cu < 1) per_cu = 1;
        // (round 6: 1/128 short -- six slots -- where it was 1/32 until the ray loop lost a quarter of its instructions: 11 930 -> 12 250
        // steps/s on the headline, profiles/r06_mcl_round_split.txt; BOTLAB_MCL_ROUND_SHORT = the divisor, 0 = a full round)
        static const int short_div = getenv("BOTLAB_MCL_ROUND_SHORT") ? atoi(getenv("BOTLAB_MCL_ROUND_SHORT")) : 128;
        const int64_t round = (int64_t)cus * per_cu - (short_div > 0 ? (int64_t)cus * per_cu / short_div : 0);
        const int64_t full = main_blocks / round, excess = main_blocks - full * round;
        // (a particle of the second region costs a wave about a tenth of a region-1 workgroup's time, and the device runs ~6000
        // such waves at once: past ~25 000 excess particles the second region outlasts the straggling round it replaces --
        // 65 536 of them made the 256 000-particle launch of BASELINE.json's configs[4] 17 % slower)
        if (full >= 1 && excess > 0 && excess * 8 < round * 5 && excess * gpb <= 24576) {
            main_blocks = full * round;
            main_particles = main_blocks * gpb;
            tail_blocks = ((int64_t)pf->n_local - main_particles + tail_tile - 1) / tail_tile;
        }
    }
    a.main_blocks = (int)main_blocks; a.main_particles = (int)main_particles;
    a.sh = pf->sh_world > 1 ? pf->sh_tab + pf->cur : nullptr;       // composed finish: sources lie with their owners
    pf->sh_stage_sums = pf->sh_stage_groups = false;
    int blocks = (int)(main_blocks + tail_blocks);
    if (blocks > pf->partials_cap) { bl_set_error("internal: partials buffer too small"); return BL_ERR_STATE; }
    hipEvent_t e0, e1;
    int rc = bl_timer_pair(ctx, BL_K_MCL_MAIN, &e0, &e1);     // a timed launch carries its own start/stop events
    if (rc) return rc;
#define MCL_LAUNCH(B, M)                                                                                          \
    do {                                                                                                          \
        if (a.interp) hipExtLaunchKernelGGL((k_mcl_main<1, B, (M) == 2 ? 0 : (M)>), dim3(blocks), dim3(B), lds_bytes, ctx->stream, e0, e1, 0, a); \
        else hipExtLaunchKernelGGL((k_mcl_main<0, B, M>), dim3(blocks), dim3(B), lds_bytes, ctx->stream, e0, e1, 0, a);          \
    } while (0)
#define MCL_LAUNCH_MODE(B)                                    \
    do {                                                      \
        if (mode == 0) MCL_LAUNCH(B, 0);                      \
        else if (mode == 1) MCL_LAUNCH(B, 1);                 \
        else MCL_LAUNCH(B, 2);                                \
    } while (0)
    if (block == 256) MCL_LAUNCH_MODE(256);
    else if (block == 512) MCL_LAUNCH_MODE(512);
    else MCL_LAUNCH_MODE(1024);
#undef MCL_LAUNCH_MODE
#undef MCL_LAUNCH
    BL_HIP(hipGetLastError());
    rc = bl_timer_commit(ctx, BL_K_MCL_MAIN, e0, e1);
    if (rc) return rc;
    pf->last_blocks = blocks;
    pf->last_tile = gpb;                                  // particles per region-1 workgroup of k_mcl_main
    pf->last_main_blocks = a.main_blocks; pf->last_main_particles = a.main_particles; pf->last_tail_tile = tail_tile;
    pf->fused_finish = (pf->n_local == pf->N) && pf->last_tile >= 1 && pf->last_tile <= mclf_chunk(MCLF_GT_LARGE) && !pf->no_fused_finish;
    return BL_OK;
}

// the finish of the update k_mcl_main has begun, from that launch's per-workgroup sums
static int pf_fused_args(bl_pf* pf, int which, int64_t utime, mcl_finish_args* f)
{
    f->partials = pf->partials; f->nblocks = pf->last_blocks;
    f->rec = pf->rec[which]; f->N = pf->N;
    f->tile = pf->last_tile; f->main_blocks = pf->last_main_blocks; f->main_particles = pf->last_main_particles;
    f->tail_tile = pf->last_tail_tile > 0 ? pf->last_tail_tile : 1;
    f->prefix = pf->prefix; f->state = pf->state; f->utime = utime;
    return pf_finish_fill(pf, f);
}

// single-shard finish of an update as its own launch (timed as BL_K_MCL_SCAN)
static int pf_finish_fused(bl_pf* pf, int which, int64_t utime)
{
    bl_ctx* ctx = pf->ctx;
    mcl_finish_args f;
    const int groups = pf_fused_args(pf, which, utime, &f);
    if (groups < 0) return pf_scan(pf, which, 1, utime);
    hipEvent_t e0, e1;
    int rc = bl_timer_begin(ctx, BL_K_MCL_SCAN, &e0, &e1);
    if (rc) return rc;
    hipLaunchKernelGGL(k_mcl_finish, dim3(MCLF_EXTRA_WGS + f.groups_wait), dim3(MCLF_WG), MCLF_LDS_BYTES, ctx->stream, f);
    BL_HIP(hipGetLastError());
    pf_strict_cumulative(pf, which);
    return bl_timer_end(ctx, BL_K_MCL_SCAN, e0, e1);
}

static int pf_shard_fin_args(bl_pf* pf, int which, int64_t utime, mcl_finish_args* f);

extern "C" int bl_pf_update_begin(bl_pf* pf, const bl_pose_xyt_t* odometry, const bl_lidar_t* scan, const bl_grid* map,
                                  int rand_value, const float* noise, int* moved)
{
    BL_CHECK_ARG(pf != nullptr && odometry != nullptr && scan != nullptr && map != nullptr);
    if (!pf->initialized) { bl_set_error("bl_pf_update before bl_pf_init_at_pose / bl_pf_set_particles"); return BL_ERR_STATE; }
    if (pf->pending_end) { bl_set_error("bl_pf_update_begin called twice without bl_pf_update_end"); return BL_ERR_STATE; }
    if (pf->sh_broken) { bl_set_error("the shard exchange of this particle set gave up earlier: set the shards up again or re-initialise the filter"); return BL_ERR_STATE; }
    BL_HIP(hipSetDevice(pf->ctx->device));
    bool mv = action_update(pf, *odometry);
    if (moved) *moved = mv ? 1 : 0;
    pf->pending_utime = odometry->utime;
    if (!mv) {
        // posteriorPose_.utime = odometry.utime (particle_filter.cpp:50)
        bl_pose_xyt_t p; p.utime = odometry->utime; p.x = p.y = p.theta = 0;
        hipLaunchKernelGGL(k_pf_set_pose, dim3(1), dim3(1), 0, pf->ctx->stream, pf->state, p, 1);
        BL_HIP(hipGetLastError());
        return BL_OK;
    }
    int R = 0;
    int rc = bl_scan_upload(pf->ctx, scan, &R);
    if (rc) return rc;
    if (noise) { rc = pf_upload_noise(pf, noise); if (rc) return rc; }
    rc = pf_launch_main(pf, map, R, rand_value, noise, 1);
    if (rc) return rc;
    pf->uniform_now = false;                     // the record this update writes carries the sensor model's weights
    pf->pending_end = true;
    return BL_OK;
}

extern "C" int bl_pf_update_end(bl_pf* pf, bl_pose_xyt_t* out_pose)
{
    BL_CHECK_ARG(pf != nullptr);
    BL_HIP(hipSetDevice(pf->ctx->device));
    if (pf->pending_end) {
        int rc;
        if (pf->sh_world > 1) {
            // composed finish: the exchange (bl_pf_shard_exchange, or the two stages with the caller's all-gathers) lies behind;
            // pre-chain + finisher over every rank's blocks
            if (!pf->sh_stage_groups) { bl_set_error("bl_pf_update_end of a composed shard before its exchange (bl_pf_shard_exchange / bl_pf_shard_stage)"); return BL_ERR_STATE; }
            mcl_finish_args f;
            if (pf_shard_fin_args(pf, pf->cur ^ 1, pf->pending_utime, &f) < 0) { bl_set_error("internal: finish launch shape"); return BL_ERR_STATE; }
            hipEvent_t e0, e1;
            rc = bl_timer_begin(pf->ctx, BL_K_MCL_SCAN, &e0, &e1);
            if (rc) return rc;
            hipLaunchKernelGGL(k_mcl_finish, dim3(MCLF_EXTRA_WGS), dim3(MCLF_WG), MCLF_LDS_BYTES, pf->ctx->stream, f);
            BL_HIP(hipGetLastError());
            pf->prefix_is_strict = false;        // (a composed shard keeps the integer rule)
            rc = bl_timer_end(pf->ctx, BL_K_MCL_SCAN, e0, e1);
        } else
        rc = pf->fused_finish ? pf_finish_fused(pf, pf->cur ^ 1, pf->pending_utime)
                              : pf_scan(pf, pf->cur ^ 1, 1, pf->pending_utime);
        if (rc) return rc;
        pf->cur ^= 1;
        pf->parent_utime = pf->pose_utime;       // parent_pose = sample.pose (action_model.cpp:92)
        pf->pose_utime = 0;                      // pose.utime = utime_ (D3)
        pf->step += 1;
        pf->pending_end = false;
    }
    if (out_pose) return bl_pf_pose_estimate(pf, out_pose);
    return BL_OK;
}

// The end of a begun update handed to another launch (bl_mcl_finish.h)
int bl_pf_take_finish(bl_pf* pf, mcl_finish_args* out)
{
    if (!pf || !pf->pending_end) return 0;
    // (strict resampling: the cumulative's launches follow the launch that carries the finish -- bl_pf_ride_launched; a sharded
    // set in strict mode ends its update the ordinary way)
    if (pf->strict && pf->sh_world > 1) return -1;
    if (pf->sh_world > 1) {
        // composed finish: groups and both all-gathers lie behind (else the caller ends the update the ordinary way, which says so)
        if (!pf->sh_stage_groups) return -1;
        if (pf_shard_fin_args(pf, pf->cur ^ 1, pf->pending_utime, out) < 0) return -1;
    } else if (pf->fused_finish) {
        if (pf_fused_args(pf, pf->cur ^ 1, pf->pending_utime, out) < 0) return -1;
    } else {
        // a shard (the record has just been gathered from every rank) or a launch shape the groups do not tile: the tile sums
        // come from the record itself, in an order that depends on N alone (k_scan_tile_sums, launched here), and the rest of
        // the finish -- groups, pre-chain, finisher -- rides in the caller's kernel exactly as in the fused form
        if (pf->no_fused_finish) return -1;
        const int which = pf->cur ^ 1;
        mcl_finish_args& f = *out;
        f.partials = pf->tile_partials; f.nblocks = pf->scan_blocks;
        f.rec = pf->rec[which]; f.N = pf->N;
        f.tile = SCAN_TILE; f.main_blocks = pf->scan_blocks; f.main_particles = pf->N; f.tail_tile = 1;
        f.prefix = pf->prefix; f.state = pf->state; f.utime = pf->pending_utime;
        if (pf_finish_fill(pf, &f) < 0) return -1;
        hipEvent_t e0, e1;
        if (bl_timer_begin(pf->ctx, BL_K_MCL_SCAN, &e0, &e1)) return -1;
        hipLaunchKernelGGL(k_scan_tile_sums, dim3(pf->scan_blocks), dim3(SCAN_THREADS), 0, pf->ctx->stream, pf->rec[which], pf->N,
                           pf->block_sums, pf->tile_partials);
        if (bl_timer_end(pf->ctx, BL_K_MCL_SCAN, e0, e1)) return -1;
    }
    pf->cur ^= 1;
    pf->parent_utime = pf->pose_utime;       // parent_pose = sample.pose (action_model.cpp:92)
    pf->pose_utime = 0;                      // pose.utime = utime_ (D3)
    pf->step += 1;
    pf->pending_end = false;
    return 1;
}


// ================================================================================================ composed finish of a sharded set
// SURVEY.md section 8e / DESIGN.md section 6.  Rank r owns the output particles [r * block, min(N, (r + 1) * block)) and keeps the
// record and the weight prefix of ITS particles only.  Per moved update:
//   k_mcl_main          own block; the resampling search and the source gather read other ranks' prefix / record where the
//                       sources lie (their memory, mapped into this process: an interval around the own block, not N x 16 B)
//   stage 1             tile sums of the own block                      -> all-gather #1 (40 B per 512 particles)
//   stage 2             groups of the own block: global weight prefix, sub-tile records, tables (the sums of EVERY tile are
//                       there now: S, the offsets, the binade predictions)   -> all-gather #2 (32 B per 128 particles + tables)
//   finish              pre-chain + finisher over every rank's blocks (riding in the map kernel, or k_mcl_finish): the 10-us
//                       part, replicated -- every rank owns the identical estimate, bit for bit the single rank's
// The two all-gathers are the caller's (bl_pf_shard_exchange runs them on a bl_comm; botlab_amd/sharded.py can also use
// torch.distributed): they are what orders one rank's kernels against the other ranks' reads of its memory.
extern "C" int bl_ipc_export(const void* dev_ptr, char* out64)
{
    BL_CHECK_ARG(dev_ptr != nullptr && out64 != nullptr);
    static_assert(sizeof(hipIpcMemHandle_t) <= 64, "IPC handle size");
    hipIpcMemHandle_t h;
    BL_HIP(hipIpcGetMemHandle(&h, const_cast<void*>(dev_ptr)));
    memset(out64, 0, 64);
    memcpy(out64, &h, sizeof(h));
    return BL_OK;
}
extern "C" int bl_ipc_open(const char* handle64, void** out)
{
    BL_CHECK_ARG(handle64 != nullptr && out != nullptr);
    hipIpcMemHandle_t h;
    memcpy(&h, handle64, sizeof(h));
    BL_HIP(hipIpcOpenMemHandle(out, h, hipIpcMemLazyEnablePeerAccess));
    return BL_OK;
}
extern "C" int bl_ipc_close(void* p)
{
    if (p) BL_HIP(hipIpcCloseMemHandle(p));
    return BL_OK;
}

extern "C" int bl_pf_shard_setup(bl_pf* pf, int rank, int world, int block)
{
    BL_CHECK_ARG(pf != nullptr && world >= 2 && world <= BL_MAX_SHARDS && rank >= 0 && rank < world);
    BL_CHECK_ARG(block > 0 && block % mclf_chunk(mclf_gthreads(pf->N)) == 0 && block % SCAN_TILE == 0);   // whole finish groups, whole scan tiles
    BL_CHECK_ARG(pf->lo == rank * block && pf->hi == (pf->N < (rank + 1) * block ? pf->N : (rank + 1) * block));
    BL_CHECK_ARG((int64_t)(world - 1) * block < pf->N);            // every rank owns particles
    if (pf->rec_external) { bl_set_error("a composed finish keeps its exchange records in the library's own allocations"); return BL_ERR_STATE; }
    if (pf->strict) { bl_set_error("strict resampling needs the whole particle set on one device"); return BL_ERR_STATE; }
    BL_HIP(hipSetDevice(pf->ctx->device));
    if (!pf->prefix) { int rc = pf_alloc(pf); if (rc) return rc; }
    BL_HIP(hipStreamSynchronize(pf->ctx->stream));
    // tile sums over the padded particle count (tiles past N sum to zero), one slice per rank
    const int tiles = world * (block / SCAN_TILE);
    if (pf->tile_partials) BL_HIP(hipFree(pf->tile_partials));
    if (pf->block_sums) BL_HIP(hipFree(pf->block_sums));
    pf->tile_partials = nullptr; pf->block_sums = nullptr;
    BL_HIP(hipMalloc((void**)&pf->tile_partials, (size_t)2 * tiles * 5 * sizeof(double)));      // (two parities: the peer-store form)
    BL_HIP(hipMalloc((void**)&pf->block_sums, (size_t)tiles * sizeof(unsigned long long)));
    BL_HIP(hipMemsetAsync(pf->tile_partials, 0, (size_t)2 * tiles * 5 * sizeof(double), pf->ctx->stream));
    pf->scan_blocks = tiles;
    pf->sh_tiles_all = tiles;
    pf->sh_peer = false; pf->sh_gen = 0;
    pf->sh_broken = false;                                         // a set whose exchange gave up starts over here
    BL_HIP(hipMemsetAsync(&pf->state->shard_broken, 0, sizeof(unsigned int), pf->ctx->stream));
    BL_HIP(hipMemsetAsync(&pf->state->wait_timeouts, 0, sizeof(unsigned int), pf->ctx->stream));
    if (!pf->sh_flags) {
        // The counters are POLLED while other devices store into them: fine-grained device memory, which this device's L2 does not
        // keep (a coarse-grained line, once fetched by a poll, would be served from the L2 for ever: a remote store does not pass
        // through it).  The data buffers are read by launches that START after the wait (a launch begins with an acquire).
        if (hipExtMallocWithFlags((void**)&pf->sh_flags, 4096, hipDeviceMallocFinegrained) != hipSuccess) {
            (void)hipGetLastError();
            BL_HIP(hipMalloc((void**)&pf->sh_flags, 4096));
        }
    }
    BL_HIP(hipMemsetAsync(pf->sh_flags, 0, 2 * BL_MAX_SHARDS * sizeof(unsigned long long), pf->ctx->stream));
    memset(pf->sh_peer_sums, 0, sizeof(pf->sh_peer_sums)); memset(pf->sh_peer_xchg, 0, sizeof(pf->sh_peer_xchg));
    memset(pf->sh_peer_flags, 0, sizeof(pf->sh_peer_flags));
    pf->sh_subs_per_rank = block / MCLF_SUB;
    pf->sh_xchg_stride = ((size_t)MCLF_XCHG_HDR + (size_t)2 * pf->sh_subs_per_rank * sizeof(ss_rec) +
                          (size_t)2 * MCLF_TSLOTS * MCLF_SUB * sizeof(mclf_tab_elem) + 255) & ~(size_t)255;
    if (pf->sh_xchg) BL_HIP(hipFree(pf->sh_xchg));
    BL_HIP(hipMalloc((void**)&pf->sh_xchg, pf->sh_xchg_stride * world));
    BL_HIP(hipMemsetAsync(pf->sh_xchg, 0, pf->sh_xchg_stride * world, pf->ctx->stream));
    if (!pf->sh_tab) BL_HIP(hipMalloc((void**)&pf->sh_tab, 2 * sizeof(mcl_shard_tab)));
    if (!pf->sh_fin) BL_HIP(hipMalloc((void**)&pf->sh_fin, 2 * sizeof(mclf_shards)));
    pf->sh_rank = rank; pf->sh_block = block;
    pf->sh_world = 0;                                              // composed from bl_pf_shard_commit on
    memset(pf->sh_peer_rec, 0, sizeof(pf->sh_peer_rec)); memset(pf->sh_peer_prefix, 0, sizeof(pf->sh_peer_prefix));
    pf->sh_world_pending = world;
    return BL_OK;
}

// what the other ranks need of this one: its two exchange records and its weight prefix (device pointers; export them with
// bl_ipc_export for another process)
extern "C" int bl_pf_shard_local_ptrs(bl_pf* pf, void** rec0, void** rec1, void** prefix)
{
    BL_CHECK_ARG(pf != nullptr && pf->prefix != nullptr && rec0 && rec1 && prefix);
    *rec0 = pf->rec[0]; *rec1 = pf->rec[1]; *prefix = pf->prefix;
    return BL_OK;
}

// rank `rank`'s arrays as THIS process sees them (its own pointers for its own rank)
extern "C" int bl_pf_shard_set_peer(bl_pf* pf, int rank, const void* rec0, const void* rec1, const void* prefix)
{
    BL_CHECK_ARG(pf != nullptr && pf->sh_world_pending >= 2 && rank >= 0 && rank < pf->sh_world_pending && rec0 && rec1 && prefix);
    pf->sh_peer_rec[0][rank] = (const float4*)rec0; pf->sh_peer_rec[1][rank] = (const float4*)rec1;
    pf->sh_peer_prefix[rank] = (const unsigned long long*)prefix;
    return BL_OK;
}

extern "C" int bl_pf_shard_commit(bl_pf* pf)
{
    BL_CHECK_ARG(pf != nullptr && pf->sh_world_pending >= 2);
    if (pf->pending_end) { bl_set_error("update pending"); return BL_ERR_STATE; }
    const int world = pf->sh_world_pending;
    for (int r = 0; r < world; ++r) BL_CHECK_ARG(pf->sh_peer_rec[0][r] && pf->sh_peer_rec[1][r] && pf->sh_peer_prefix[r]);
    BL_CHECK_ARG(pf->sh_peer_rec[0][pf->sh_rank] == pf->rec[0] && pf->sh_peer_prefix[pf->sh_rank] == pf->prefix);
    BL_HIP(hipSetDevice(pf->ctx->device));
    mcl_shard_tab tab[2];
    mclf_shards fin[2];
    memset(tab, 0, sizeof(tab)); memset((void*)fin, 0, sizeof(fin));
    for (int c = 0; c < 2; ++c) {
        tab[c].world = world; tab[c].block = pf->sh_block;
        fin[c].world = world; fin[c].rank = pf->sh_rank; fin[c].block = pf->sh_block; fin[c].subs_per_rank = pf->sh_subs_per_rank;
        fin[c].xchg = pf->sh_xchg; fin[c].xchg_stride = pf->sh_xchg_stride;
        for (int r = 0; r < world; ++r) {
            tab[c].src[r] = pf->sh_peer_rec[c][r];                 // an update from cur = c reads the records rec[c]
            tab[c].prefix[r] = pf->sh_peer_prefix[r];
            fin[c].rec[r] = pf->sh_peer_rec[c ^ 1][r];             // ... and its finish the ones it wrote, rec[c ^ 1]
        }
    }
    BL_HIP(hipMemcpyAsync(pf->sh_tab, tab, sizeof(tab), hipMemcpyHostToDevice, pf->ctx->stream));
    BL_HIP(hipMemcpyAsync(pf->sh_fin, fin, sizeof(fin), hipMemcpyHostToDevice, pf->ctx->stream));
    BL_HIP(hipStreamSynchronize(pf->ctx->stream));
    pf->sh_world = world;
    pf->sh_stage_sums = pf->sh_stage_groups = false;
    return BL_OK;
}

// The tile sums of the running update.  The collective forms keep one buffer (the all-gather orders its reuse); the peer-store
// form keeps two and alternates: a rank that runs ahead pushes the NEXT update's sums while this one's finish still reads these.
static double* pf_shard_partials(const bl_pf* pf)
{
    return pf->tile_partials + (pf->sh_peer ? (size_t)(pf->sh_gen & 1ull) * (size_t)pf->sh_tiles_all * 5 : 0);
}

// the two buffers of the exchange: every rank's slice of the tile sums and every rank's block of records / tables
extern "C" int bl_pf_shard_buffers(bl_pf* pf, void** sums, size_t* sums_bytes_per_rank, void** xchg, size_t* xchg_bytes_per_rank)
{
    BL_CHECK_ARG(pf != nullptr && (pf->sh_world >= 2 || pf->sh_world_pending >= 2) && sums && sums_bytes_per_rank && xchg && xchg_bytes_per_rank);
    *sums = pf_shard_partials(pf); *sums_bytes_per_rank = (size_t)(pf->sh_block / SCAN_TILE) * 5 * sizeof(double);
    *xchg = pf->sh_xchg; *xchg_bytes_per_rank = pf->sh_xchg_stride;
    return BL_OK;
}

// the finish arguments of the record an update from `cur` wrote (the update begun, or the particles as they stand)
static int pf_shard_fin_args(bl_pf* pf, int which, int64_t utime, mcl_finish_args* f)
{
    f->partials = pf_shard_partials(pf); f->nblocks = pf->scan_blocks;
    f->rec = pf->rec[which]; f->N = pf->N;
    f->tile = SCAN_TILE; f->main_blocks = pf->scan_blocks; f->main_particles = pf->N; f->tail_tile = 1;
    f->prefix = pf->prefix; f->state = pf->state; f->utime = utime;
    if (pf_finish_fill(pf, f) < 0) return -1;
    f->groups_wait = 0;
    f->sh = pf->sh_fin + (which ^ 1);
    f->recs = (ss_rec*)pf->sh_xchg; f->tabs = (mclf_tab_elem*)pf->sh_xchg;      // (non-null: the groups leave records and tables)
    return f->groups;
}

// stage 1 (tile sums of the own block) or stage 2 (its groups) of the running update's exchange; the caller all-gathers the
// corresponding buffer of bl_pf_shard_buffers behind each (in place: this rank's slice is where it belongs)
extern "C" int bl_pf_shard_stage(bl_pf* pf, int stage)
{
    BL_CHECK_ARG(pf != nullptr && pf->sh_world >= 2 && (stage == 1 || stage == 2));
    if (!pf->pending_end) { bl_set_error("bl_pf_shard_stage without an update begun"); return BL_ERR_STATE; }
    BL_HIP(hipSetDevice(pf->ctx->device));
    const int which = pf->cur ^ 1;
    hipEvent_t e0, e1;
    int rc = bl_timer_begin(pf->ctx, BL_K_MCL_SCAN, &e0, &e1);
    if (rc) return rc;
    if (stage == 1) {
        const int tiles = pf->sh_block / SCAN_TILE;
        hipLaunchKernelGGL(k_scan_tile_sums, dim3(tiles), dim3(SCAN_THREADS), 0, pf->ctx->stream, (const float4*)pf->rec[which], pf->N, pf->block_sums,
                           pf_shard_partials(pf), pf->sh_rank * tiles, (unsigned long long*)(pf->sh_xchg + (size_t)pf->sh_rank * pf->sh_xchg_stride));
        pf->sh_stage_sums = true;
    } else {
        if (!pf->sh_stage_sums) { bl_set_error("bl_pf_shard_stage(2) before stage 1"); return BL_ERR_STATE; }
        mcl_finish_args f;
        if (pf_shard_fin_args(pf, which, pf->pending_utime, &f) < 0) { bl_set_error("internal: finish launch shape"); return BL_ERR_STATE; }
        const int per_rank = pf->sh_block / mclf_chunk(f.gthreads);
        hipLaunchKernelGGL(k_shard_groups, dim3(per_rank), dim3(MCLF_WG), 0, pf->ctx->stream, f, pf->sh_rank * per_rank);
        pf->sh_stage_groups = true;
    }
    BL_HIP(hipGetLastError());
    return bl_timer_end(pf->ctx, BL_K_MCL_SCAN, e0, e1);
}

// both stages with their all-gathers on the library's own communicator (bl_comm.hip), all on the filter's stream
extern "C" int bl_pf_shard_exchange(bl_pf* pf, bl_comm* c)
{
    BL_CHECK_ARG(pf != nullptr && c != nullptr);
    void* sums; void* xchg; size_t sb, xb;
    int rc = bl_pf_shard_buffers(pf, &sums, &sb, &xchg, &xb);
    if (rc) return rc;
    rc = bl_pf_shard_stage(pf, 1);
    if (rc) return rc;
    rc = bl_comm_all_gather_inplace(c, sums, sb / sizeof(float));
    if (rc) return rc;
    rc = bl_pf_shard_stage(pf, 2);
    if (rc) return rc;
    return bl_comm_all_gather_inplace(c, xchg, xb / sizeof(float));
}

// ---- peer-store form: the exchange without a collective ---------------------------------------------------------------------
// Two RCCL all-gathers cost ~40 us of latency each whatever they carry (10 KB and 87 KB per rank at 1M particles / 8 ranks), next
// to ~130 us of kernels: by the builder's own budget 2.9x at 8 ranks where north_star asks for 6x.  Here a rank copies its slice
// of the tile sums, and later its exchange block, straight into the same place of every other rank's buffer -- their memory,
// mapped into this process (hipIpc), 16-byte system-scope stores over xGMI -- fences, and then stores the update's number into
// its slot of every rank's counter table.  A consuming launch is preceded by a one-wave launch that waits until every slot of the
// local table has reached the update's number (system-scope loads, a spin limit as everywhere).  Layouts, kernels and results are
// those of the collective form, bit for bit.  What orders the REUSE of a buffer without a collective:
//   * tile sums: two parities (a rank ahead pushes the next update's sums while this one's finish still reads these);
//   * exchange blocks: rank B's groups of update u + 1 need every rank's tile sums of u + 1, and rank A pushes those behind its
//     finish of update u on its stream -- so B cannot overwrite its block in A's memory while A still reads update u's;
//   * records / weight prefix read by another rank's k_mcl_main: as in the collective form they are complete when the owner's
//     block has arrived (the push is a later launch on the owner's stream), and they are rewritten only behind the owner's
//     groups of the next update, which wait for the reader's tile sums of that update.
struct shard_peers { double* sums[BL_MAX_SHARDS]; char* xchg[BL_MAX_SHARDS]; unsigned long long* flags[BL_MAX_SHARDS]; };

typedef int shard_i4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void shard_store_sys(void* dst, const shard_i4 v)
{
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(dst), "v"(v) : "memory");
}

typedef int shard_i2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void shard_store_sys8(void* dst, const shard_i2 v)
{
    asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" :: "v"(dst), "v"(v) : "memory");
}

// Workgroup r of `world`: `bytes` (a multiple of 8) at src + off -> rank r's buffer + off, then rank r's flags[kind][rank] = gen
// (workgroup `rank` has nothing to do: its own copy is the source).  A workgroup per destination: the links to the seven
// neighbours carry their copies side by side, and no workgroup waits for another.
__global__ __launch_bounds__(1024) void k_shard_push(const shard_peers* __restrict__ peers, int kind, size_t off, size_t bytes, int rank, int world,
                                                     unsigned long long gen)
{
    const int r = blockIdx.x;
    if (r == rank || r >= world) return;
    const char* src = (kind == 0 ? (const char*)peers->sums[rank] : (const char*)peers->xchg[rank]) + off;
    char* dst = (kind == 0 ? (char*)peers->sums[r] : peers->xchg[r]) + off;
    if (((off | bytes) & 15) == 0) {
        const size_t n16 = bytes / 16;
        for (size_t i = threadIdx.x; i < n16; i += blockDim.x) shard_store_sys(dst + i * 16, ((const shard_i4*)src)[i]);
    } else {                                                 // a slice of doubles that is not whole 16-byte pieces (an odd number of tiles)
        const size_t n8 = bytes / 8;
        for (size_t i = threadIdx.x; i < n8; i += blockDim.x) shard_store_sys8(dst + i * 8, ((const shard_i2*)src)[i]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0)
        __hip_atomic_store(peers->flags[r] + kind * BL_MAX_SHARDS + rank, gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// One wave: until every other rank's slot of this rank's counter table has reached `gen` (lane r watches rank r).  Ranks run
// skewed by whatever their hosts do (a first launch that loads code, a host that plans for seconds inside a fetch, another process
// on the device), so this wait has its own limit on the 100 MHz clock -- `limit_ticks`, tens of seconds unless
// BOTLAB_SHARD_WAIT_MS says otherwise -- not the spin count of the waits inside one launch.  When it does give up the particle
// set can no longer be trusted: it sets the STICKY pf_state::shard_broken, which turns this update's groups, finish, map store
// and every later k_mcl_main of the set into no-ops and makes every later wait return at once; the host reports BL_ERR_STATE from
// every call that fetches the pose until the shards are set up again (bl_pf_shard_setup) or the filter re-initialised.
__global__ __launch_bounds__(64) void k_shard_wait(const unsigned long long* __restrict__ flags, int kind, int rank, int world, unsigned long long gen,
                                                   pf_state* __restrict__ state, unsigned long long limit_ticks)
{
    const int r = threadIdx.x;
    if (r >= world || r == rank) return;
    if (__hip_atomic_load(&state->shard_broken, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__hip_atomic_load(flags + kind * BL_MAX_SHARDS + r, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < gen) {
        if (__builtin_amdgcn_s_memrealtime() - t0 > limit_ticks) {
            atomicAdd(&state->wait_timeouts, 1u);
            __hip_atomic_store(&state->shard_broken, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
        __builtin_amdgcn_s_sleep(8);
    }
}

// what the other ranks need of this one for the peer-store form: its tile-sum buffer, its exchange blocks and its counter table
extern "C" int bl_pf_shard_local_ptrs_peer(bl_pf* pf, void** sums, void** xchg, void** flags)
{
    BL_CHECK_ARG(pf != nullptr && pf->tile_partials != nullptr && pf->sh_xchg != nullptr && pf->sh_flags != nullptr && sums && xchg && flags);
    *sums = pf->tile_partials; *xchg = pf->sh_xchg; *flags = pf->sh_flags;
    return BL_OK;
}

extern "C" int bl_pf_shard_set_peer_buffers(bl_pf* pf, int rank, void* sums, void* xchg, void* flags)
{
    BL_CHECK_ARG(pf != nullptr && pf->sh_world_pending >= 2 && rank >= 0 && rank < pf->sh_world_pending && sums && xchg && flags);
    pf->sh_peer_sums[rank] = (double*)sums; pf->sh_peer_xchg[rank] = (char*)xchg; pf->sh_peer_flags[rank] = (unsigned long long*)flags;
    return BL_OK;
}

// after bl_pf_shard_commit: switch the exchange to the peer-store form (every rank's buffers have been handed in)
extern "C" int bl_pf_shard_peer_commit(bl_pf* pf)
{
    BL_CHECK_ARG(pf != nullptr && pf->sh_world >= 2);
    if (pf->pending_end) { bl_set_error("update pending"); return BL_ERR_STATE; }
    for (int r = 0; r < pf->sh_world; ++r) BL_CHECK_ARG(pf->sh_peer_sums[r] && pf->sh_peer_xchg[r] && pf->sh_peer_flags[r]);
    BL_CHECK_ARG(pf->sh_peer_sums[pf->sh_rank] == pf->tile_partials && pf->sh_peer_xchg[pf->sh_rank] == pf->sh_xchg && pf->sh_peer_flags[pf->sh_rank] == pf->sh_flags);
    BL_HIP(hipSetDevice(pf->ctx->device));
    shard_peers h;
    memset((void*)&h, 0, sizeof(h));
    for (int r = 0; r < pf->sh_world; ++r) { h.sums[r] = pf->sh_peer_sums[r]; h.xchg[r] = pf->sh_peer_xchg[r]; h.flags[r] = pf->sh_peer_flags[r]; }
    if (!pf->sh_peers_dev) BL_HIP(hipMalloc((void**)&pf->sh_peers_dev, sizeof(shard_peers)));
    BL_HIP(hipMemcpyAsync(pf->sh_peers_dev, &h, sizeof(h), hipMemcpyHostToDevice, pf->ctx->stream));
    BL_HIP(hipStreamSynchronize(pf->ctx->stream));
    pf->sh_peer = true;
    pf->sh_gen = 0;
    return BL_OK;
}

extern "C" int bl_pf_shard_peer_active(const bl_pf* pf) { return pf && pf->sh_peer ? 1 : 0; }

// First contact, before any update relies on it: every rank pushes a pattern into every other rank's tile-sum buffer and raises
// its counter (number `probe_gen`, counted like an update); then waits for everybody's counter and checks everybody's pattern.
// Returns BL_OK with *ok = 1 when this rank has seen every other rank's bytes arrive within the device-side spin limit; the caller
// agrees on the answer over its rendezvous (all ranks must take the same form) and calls bl_pf_shard_peer_reset.
__global__ __launch_bounds__(64) void k_shard_probe_fill(double* own_slice, int words, int rank, unsigned long long gen)
{
    for (int i = threadIdx.x; i < words; i += 64) own_slice[i] = (double)((rank + 1) * 1000003ull + gen * 7919ull + (unsigned long long)i);
}
__global__ __launch_bounds__(64) void k_shard_probe_check(const double* sums, int words_per_rank, int rank, int world, unsigned long long gen, int* bad)
{
    for (int r = 0; r < world; ++r) {
        if (r == rank) continue;
        const double* sl = sums + (size_t)r * words_per_rank;
        for (int i = threadIdx.x; i < words_per_rank; i += 64) {
            const double v = __hip_atomic_load(sl + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (v != (double)((r + 1) * 1000003ull + gen * 7919ull + (unsigned long long)i)) atomicAdd(bad, 1);
        }
    }
}
extern "C" int bl_pf_shard_peer_selftest(bl_pf* pf, int* ok)
{
    BL_CHECK_ARG(pf != nullptr && pf->sh_peer && ok != nullptr);
    *ok = 0;
    BL_HIP(hipSetDevice(pf->ctx->device));
    const int tiles = pf->sh_block / SCAN_TILE, words = tiles * 5;
    const unsigned long long gen = ++pf->sh_gen;
    double* mine = pf_shard_partials(pf) + (size_t)pf->sh_rank * words;
    const size_t off = (size_t)((char*)mine - (char*)pf->tile_partials);
    int* d_bad = nullptr;
    BL_HIP(hipMalloc((void**)&d_bad, 2 * sizeof(int)));
    BL_HIP(hipMemsetAsync(d_bad, 0, 2 * sizeof(int), pf->ctx->stream));
    unsigned int before = 0, after = 0;
    BL_HIP(hipMemcpyAsync(&before, &pf->state->wait_timeouts, 4, hipMemcpyDeviceToHost, pf->ctx->stream));
    hipLaunchKernelGGL(k_shard_probe_fill, dim3(1), dim3(64), 0, pf->ctx->stream, mine, words, pf->sh_rank, gen);
    hipLaunchKernelGGL(k_shard_push, dim3(pf->sh_world), dim3(1024), 0, pf->ctx->stream, pf->sh_peers_dev, 0, off, (size_t)words * 8, pf->sh_rank, pf->sh_world, gen);
    hipLaunchKernelGGL(k_shard_wait, dim3(1), dim3(64), 0, pf->ctx->stream, pf->sh_flags, 0, pf->sh_rank, pf->sh_world, gen, pf->state, (unsigned long long)100000 * 1000);      // (the probe: 1 s)
    hipLaunchKernelGGL(k_shard_probe_check, dim3(1), dim3(64), 0, pf->ctx->stream, pf_shard_partials(pf), words, pf->sh_rank, pf->sh_world, gen, d_bad);
    BL_HIP(hipGetLastError());
    int bad[2] = {0, 0};
    BL_HIP(hipMemcpyAsync(bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost, pf->ctx->stream));
    BL_HIP(hipMemcpyAsync(&after, &pf->state->wait_timeouts, 4, hipMemcpyDeviceToHost, pf->ctx->stream));
    BL_HIP(hipStreamSynchronize(pf->ctx->stream));
    (void)hipFree(d_bad);
    if (after != before) {                                   // (the probe's own time-out is its answer, not a broken set)
        BL_HIP(hipMemsetAsync(&pf->state->wait_timeouts, 0, 4, pf->ctx->stream));
        BL_HIP(hipMemsetAsync(&pf->state->shard_broken, 0, 4, pf->ctx->stream));
    }
    *ok = (bad[0] == 0 && after == before) ? 1 : 0;
    return BL_OK;
}

// leave the peer-store form (the ranks agreed to: a rank's self-test failed); the collective forms take over.  Counters restart.
extern "C" int bl_pf_shard_peer_reset(bl_pf* pf, int keep)
{
    BL_CHECK_ARG(pf != nullptr);
    if (pf->pending_end) { bl_set_error("update pending"); return BL_ERR_STATE; }
    if (!keep) pf->sh_peer = false;
    return BL_OK;
}

// limit of a cross-rank wait in 100 MHz ticks: 30 s, or BOTLAB_SHARD_WAIT_MS (tests)
static unsigned long long shard_wait_ticks()
{
    static const unsigned long long ticks = [] {
        const char* e = getenv("BOTLAB_SHARD_WAIT_MS");
        const double ms = e && atof(e) > 0 ? atof(e) : 30000.0;
        return (unsigned long long)(ms * 100000.0);
    }();
    return ticks;
}

// The running update's exchange in the peer-store form, on the filter's stream, in three phases: 0 = tile sums of the own block,
// pushed to every rank; 1 = wait for every rank's sums, the own block's groups, their block pushed to every rank; 2 = wait for
// every rank's block.  (Phases so that ONE process that drives several ranks -- tests -- can enqueue every rank's pushes before
// any rank's wait: streams that share a hardware queue run in submission order.)
extern "C" int bl_pf_shard_exchange_peer_phase(bl_pf* pf, int phase)
{
    BL_CHECK_ARG(pf != nullptr && pf->sh_world >= 2 && pf->sh_peer && phase >= 0 && phase <= 2);
    if (!pf->pending_end) { bl_set_error("bl_pf_shard_exchange_peer without an update begun"); return BL_ERR_STATE; }
    BL_HIP(hipSetDevice(pf->ctx->device));
    const int tiles = pf->sh_block / SCAN_TILE;
    hipEvent_t e0, e1;
    int rc;
    if (phase == 0) {
        if (pf->sh_stage_sums) { bl_set_error("exchange phase 0 twice in one update"); return BL_ERR_STATE; }
        pf->sh_gen += 1;                                     // (pf_shard_partials follows: this update's parity)
        rc = bl_pf_shard_stage(pf, 1);
        if (rc) return rc;
        rc = bl_timer_begin(pf->ctx, BL_K_MCL_SCAN, &e0, &e1);
        if (rc) return rc;
        const size_t sums_off = (size_t)((char*)(pf_shard_partials(pf) + (size_t)pf->sh_rank * tiles * 5) - (char*)pf->tile_partials);
        hipLaunchKernelGGL(k_shard_push, dim3(pf->sh_world), dim3(1024), 0, pf->ctx->stream, pf->sh_peers_dev, 0, sums_off,
                           (size_t)tiles * 5 * sizeof(double), pf->sh_rank, pf->sh_world, pf->sh_gen);
    } else if (phase == 1) {
        if (!pf->sh_stage_sums || pf->sh_stage_groups) { bl_set_error("exchange phase 1 out of order"); return BL_ERR_STATE; }
        hipLaunchKernelGGL(k_shard_wait, dim3(1), dim3(64), 0, pf->ctx->stream, pf->sh_flags, 0, pf->sh_rank, pf->sh_world, pf->sh_gen, pf->state, shard_wait_ticks());
        rc = bl_pf_shard_stage(pf, 2);
        if (rc) return rc;
        rc = bl_timer_begin(pf->ctx, BL_K_MCL_SCAN, &e0, &e1);
        if (rc) return rc;
        hipLaunchKernelGGL(k_shard_push, dim3(pf->sh_world), dim3(1024), 0, pf->ctx->stream, pf->sh_peers_dev, 1, (size_t)pf->sh_rank * pf->sh_xchg_stride,
                           pf->sh_xchg_stride, pf->sh_rank, pf->sh_world, pf->sh_gen);
    } else {
        if (!pf->sh_stage_groups) { bl_set_error("exchange phase 2 out of order"); return BL_ERR_STATE; }
        rc = bl_timer_begin(pf->ctx, BL_K_MCL_SCAN, &e0, &e1);
        if (rc) return rc;
        hipLaunchKernelGGL(k_shard_wait, dim3(1), dim3(64), 0, pf->ctx->stream, pf->sh_flags, 1, pf->sh_rank, pf->sh_world, pf->sh_gen, pf->state, shard_wait_ticks());
    }
    BL_HIP(hipGetLastError());
    return bl_timer_end(pf->ctx, BL_K_MCL_SCAN, e0, e1);
}

extern "C" int bl_pf_shard_exchange_peer(bl_pf* pf)
{
    for (int phase = 0; phase < 3; ++phase) { const int rc = bl_pf_shard_exchange_peer_phase(pf, phase); if (rc) return rc; }
    return BL_OK;
}

// bytes of the exchange per rank and update: what this rank sends into the two all-gathers, what it receives from them, and
// the size of its own block of particle records (its k_mcl_main reads about that much of source records, from wherever they
// lie; the replicated form receives N x 16 B instead)
extern "C" int bl_pf_shard_traffic(bl_pf* pf, int64_t* out3)
{
    BL_CHECK_ARG(pf != nullptr && out3 != nullptr && pf->sh_world >= 2);
    const int64_t sums = (int64_t)(pf->sh_block / SCAN_TILE) * 5 * (int64_t)sizeof(double);
    out3[0] = sums + (int64_t)pf->sh_xchg_stride;
    out3[1] = (int64_t)(pf->sh_world - 1) * out3[0];
    if (pf->sh_peer) out3[0] = out3[1];                      // peer-store form: the slice and the block are stored once per other rank
    out3[2] = (int64_t)pf->n_local * (int64_t)sizeof(float4);
    return BL_OK;
}

bl_ctx* bl_pf_ctx(bl_pf* pf) { return pf ? pf->ctx : nullptr; }

// the launch that carries a taken finish has been enqueued: what follows the finish in strict mode follows it
void bl_pf_ride_launched(bl_pf* pf)
{
    if (pf) pf_strict_cumulative(pf, pf->cur);               // (bl_pf_take_finish has flipped cur: the record the finish works on)
}

int bl_pf_launch_taken_finish(bl_pf* pf, const mcl_finish_args* fin)
{
    if (!pf || !fin) return BL_ERR_ARG;
    hipLaunchKernelGGL(k_mcl_finish, dim3(MCLF_EXTRA_WGS + fin->groups_wait), dim3(MCLF_WG), MCLF_LDS_BYTES, pf->ctx->stream, *fin);
    BL_HIP(hipGetLastError());
    bl_pf_ride_launched(pf);
    return BL_OK;
}

extern "C" int bl_pf_update(bl_pf* pf, const bl_pose_xyt_t* odometry, const bl_lidar_t* scan, const bl_grid* map,
                            int rand_value, const float* noise, bl_pose_xyt_t* out_pose)
{
    int moved = 0;
    int rc = bl_pf_update_begin(pf, odometry, scan, map, rand_value, noise, &moved);
    if (rc) return rc;
    return bl_pf_update_end(pf, out_pose);
}

extern "C" int bl_pf_update_action_only(bl_pf* pf, const bl_pose_xyt_t* odometry, const float* noise, bl_pose_xyt_t* out_pose)
{
    BL_CHECK_ARG(pf != nullptr && odometry != nullptr);
    if (!pf->initialized || pf->pending_end) { bl_set_error("filter not initialised or update pending"); return BL_ERR_STATE; }
    if (pf->n_local != pf->N) { bl_set_error("updateFilterActionOnly needs the whole particle set on one device"); return BL_ERR_ARG; }
    BL_HIP(hipSetDevice(pf->ctx->device));
    bool mv = action_update(pf, *odometry);
    if (mv) {
        if (noise) { int rc = pf_upload_noise(pf, noise); if (rc) return rc; }
        int rc = pf_launch_main(pf, nullptr, 0, 0, noise, 0);     // proposal = applyAction(posterior_) (particle_filter.cpp:60-61)
        if (rc) return rc;
        pf->fused_finish = false;                                 // weights are carried over: plain scan, no estimate
        rc = pf_scan(pf, pf->cur ^ 1, 0, 0);
        if (rc) return rc;
        pf->cur ^= 1;
        pf->parent_utime = pf->pose_utime;
        pf->pose_utime = 0;
        pf->step += 1;
    }
    hipLaunchKernelGGL(k_pf_set_pose, dim3(1), dim3(1), 0, pf->ctx->stream, pf->state, *odometry, 0);   // posteriorPose_ = odometry
    BL_HIP(hipGetLastError());
    if (out_pose) *out_pose = *odometry;
    return BL_OK;
}

extern "C" int bl_pf_pose_estimate(bl_pf* pf, bl_pose_xyt_t* out_pose)
{
    BL_CHECK_ARG(pf != nullptr && out_pose != nullptr && pf->state != nullptr);
    struct { bl_pose_xyt_t pose; unsigned int wait_timeouts; unsigned int shard_broken; } h;
    static_assert(offsetof(pf_state, wait_timeouts) == offsetof(pf_state, pose) + sizeof(bl_pose_xyt_t), "the counter is read with the pose");
    static_assert(offsetof(pf_state, shard_broken) == offsetof(pf_state, wait_timeouts) + sizeof(unsigned int), "the flag is read with the pose");
    if (pf->sh_broken) { bl_set_error("the shard exchange of this particle set gave up earlier: set the shards up again or re-initialise the filter"); return BL_ERR_STATE; }
    BL_HIP(hipMemcpyAsync(&h, &pf->state->pose, sizeof(h), hipMemcpyDeviceToHost, pf->ctx->stream));
    BL_HIP(hipStreamSynchronize(pf->ctx->stream));
    *out_pose = h.pose;
    if (h.shard_broken != 0) {
        // sticky, on the device and here: the launches that would have consumed the missing data did nothing (particles, weights,
        // pose and map are those of the last complete update), and nothing re-creates the set behind the caller's back
        pf->sh_broken = true;
        bl_set_error("a wait for another rank's part of the shard exchange gave up (%u): this update's groups, finish, map store and every "
                     "later resampling did nothing, the particle set is no valid posterior any more; set the shards up again "
                     "(bl_pf_shard_setup) or re-initialise the filter", h.wait_timeouts);
        return BL_ERR_STATE;
    }
    if (h.wait_timeouts != 0) {
        // reported once: the count belongs to the launches since the last estimate was fetched, not to every later one
        BL_HIP(hipMemsetAsync(&pf->state->wait_timeouts, 0, sizeof(unsigned int), pf->ctx->stream));
        bl_set_error("a wait inside a finish launch gave up (%u): the estimate is not valid", h.wait_timeouts);
        return BL_ERR_STATE;
    }
    return BL_OK;
}

// estimatePosteriorPose(posterior_) on demand (particle_filter.cpp:144-160): the record-based finish on the current record
extern "C" int bl_pf_estimate_posterior_pose(bl_pf* pf, bl_pose_xyt_t* out_pose)
{
    BL_CHECK_ARG(pf != nullptr);
    if (!pf->initialized || pf->pending_end) { bl_set_error("filter not initialised or update pending"); return BL_ERR_STATE; }
    if (pf->sh_world > 1) { bl_set_error("estimatePosteriorPose on demand needs the whole record on this device (composed shard)"); return BL_ERR_STATE; }
    BL_HIP(hipSetDevice(pf->ctx->device));
    int rc = pf_scan(pf, pf->cur, 1, pf->pending_utime);
    if (rc) return rc;
    if (out_pose) return bl_pf_pose_estimate(pf, out_pose);
    return BL_OK;
}

extern "C" int bl_pf_debug_estimate_stats(bl_pf* pf, uint32_t* out4)   /* eight values */
{
    BL_CHECK_ARG(pf != nullptr && out4 != nullptr && pf->state != nullptr);
    BL_HIP(hipMemcpyAsync(out4, pf->state->chain_stats, 8 * sizeof(uint32_t), hipMemcpyDeviceToHost, pf->ctx->stream));
    BL_HIP(hipStreamSynchronize(pf->ctx->stream));
    if (getenv("BOTLAB_FINISH_LOOKAHEAD")) {
        unsigned int la[2];
        BL_HIP(hipMemcpy(la, pf->state->lookahead, sizeof(la), hipMemcpyDeviceToHost));
        fprintf(stderr, "map updates ahead of the exact pose: %u, run again: %u\n", la[0], la[1]);
        unsigned int ps[2];
        BL_HIP(hipMemcpy(ps, pf->state->pre_stats, sizeof(ps), hipMemcpyDeviceToHost));
        fprintf(stderr, "pre-chain sub-tiles by map / replayed: x %u / %u, y %u / %u\n", ps[0] >> 16, ps[0] & 0xffffu, ps[1] >> 16, ps[1] & 0xffffu);
    }
    if (getenv("BOTLAB_FINISH_STAMPS")) {
        unsigned long long st[6];
        BL_HIP(hipMemcpy(st, pf->state->stamps, sizeof(st), hipMemcpyDeviceToHost));
        unsigned long long gs[8];
        BL_HIP(hipMemcpy(gs, pf->state->gstamps, sizeof(gs), hipMemcpyDeviceToHost));
        fprintf(stderr, "group timeline (us, from the finisher's entry): entry %+.2f", ((long long)gs[0] - (long long)st[0]) * 0.01);
        for (int k = 1; k < 8; ++k) fprintf(stderr, " | +%.2f", (gs[k] - gs[k - 1]) * 0.01);
        fprintf(stderr, "\n");
        unsigned long long cs[16];
        BL_HIP(hipMemcpy(cs, pf->state->cstamps, sizeof(cs), hipMemcpyDeviceToHost));
        fprintf(stderr, "x chain entries (us):");
        for (int k = 1; k < 16; ++k) fprintf(stderr, " %.2f", cs[k] > cs[k - 1] ? (cs[k] - cs[k - 1]) * 0.01 : -1.0);
        fprintf(stderr, "\n");
        unsigned long long xs[16];
        BL_HIP(hipMemcpy(xs, pf->state->xstamps, sizeof(xs), hipMemcpyDeviceToHost));
        fprintf(stderr, "stamped group: loads back %+.2f after its entry; finisher stage a: wave 0 has its first records %+.2f, is through %+.2f after the block sums\n",
                ((long long)xs[0] - (long long)gs[0]) * 0.01, ((long long)xs[1] - (long long)st[1]) * 0.01, ((long long)xs[2] - (long long)st[1]) * 0.01);
        fprintf(stderr, "pre-chain wave 0: block sums done %+.2f, past their barrier %+.2f, terms done %+.2f\n", ((long long)xs[8] - (long long)st[0]) * 0.01,
                ((long long)xs[9] - (long long)st[0]) * 0.01, ((long long)xs[10] - (long long)st[0]) * 0.01);
        fprintf(stderr, "pre-chain: sums barrier %+.2f, stepped %+.2f, maps barrier %+.2f after the finisher's entry\n", ((long long)xs[5] - (long long)st[0]) * 0.01,
                ((long long)xs[6] - (long long)st[0]) * 0.01, ((long long)xs[7] - (long long)st[0]) * 0.01);
        fprintf(stderr, "pre-chain published %+.2f, x chain had its start value %+.2f after the finisher's entry\n", ((long long)xs[3] - (long long)st[0]) * 0.01, ((long long)xs[4] - (long long)st[0]) * 0.01);
        fprintf(stderr, "map workgroup: counts stand %+.2f, exact pose taken %+.2f, verified %+.2f, leaders done %+.2f, window done %+.2f after the finisher's entry\n",
                ((long long)xs[15] - (long long)st[0]) * 0.01, ((long long)xs[11] - (long long)st[0]) * 0.01, ((long long)xs[12] - (long long)st[0]) * 0.01,
                ((long long)xs[13] - (long long)st[0]) * 0.01, ((long long)xs[14] - (long long)st[0]) * 0.01);
        fprintf(stderr, "finisher timeline (us): block sums +%.2f, records in and staged +%.2f, lists and gaps +%.2f, chains +%.2f, exit +%.2f\n", (st[1] - st[0]) * 0.01,
                (st[2] - st[1]) * 0.01, (st[3] - st[2]) * 0.01, (st[4] - st[3]) * 0.01, (st[5] - st[4]) * 0.01);
    }
    return BL_OK;
}

// The generation the NEXT finish launch's records are tagged with follows from this one (tests: the tag's wrap-around, where the
// host skips the generations whose tag reads like zeroed slots)
extern "C" int bl_pf_debug_set_finish_generation(bl_pf* pf, uint32_t generation)
{
    BL_CHECK_ARG(pf != nullptr);
    pf->fin_gen = generation;
    return BL_OK;
}

// Strict resampling on / off.  Takes effect with the next prefix the filter forms (an update's end, an upload of particles,
// initializeFilterAtPose); switching it on for a filter that already holds particles re-forms the prefix at once.
extern "C" int bl_pf_set_strict_resampling(bl_pf* pf, int on)
{
    BL_CHECK_ARG(pf != nullptr);
    if (pf->pending_end) { bl_set_error("update pending"); return BL_ERR_STATE; }
    const bool was = pf->strict;
    pf->strict = on != 0;
    if (pf->initialized && was != pf->strict) { BL_HIP(hipSetDevice(pf->ctx->device)); return pf_scan(pf, pf->cur, 0, 0); }
    return BL_OK;
}

extern "C" int bl_pf_debug_resample(bl_pf* pf, int rand_value, int32_t* out_idx)
{
    BL_CHECK_ARG(pf != nullptr && out_idx != nullptr);
    if (!pf->initialized || pf->pending_end) { bl_set_error("filter not initialised or update pending"); return BL_ERR_STATE; }
    if (pf->n_local != pf->N) { bl_set_error("bl_pf_debug_resample needs the whole particle set on one device"); return BL_ERR_ARG; }
    BL_HIP(hipSetDevice(pf->ctx->device));
    const double M_inv = 1.0 / pf->N;                                              // particle_filter.cpp:89
    const double r = (((double)rand_value) / (double)RAND_MAX) * M_inv;            // particle_filter.cpp:92
    hipLaunchKernelGGL(k_pf_resample_only, dim3((pf->N + 255) / 256), dim3(256), 0, pf->ctx->stream, pf->prefix, pf->state, pf->N, r, M_inv,
                       pf->prefix_is_strict ? 1 : 0, pf->dbg_idx);
    BL_HIP(hipGetLastError());
    BL_HIP(hipMemcpyAsync(out_idx, pf->dbg_idx, (size_t)pf->N * 4, hipMemcpyDeviceToHost, pf->ctx->stream));
    BL_HIP(hipStreamSynchronize(pf->ctx->stream));
    return BL_OK;
}

extern "C" int bl_pf_debug_uniform_runs(bl_pf* pf, int* out_runs)
{
    BL_CHECK_ARG(pf != nullptr && out_runs != nullptr);
    if (!pf->initialized) { bl_set_error("filter not initialised"); return BL_ERR_STATE; }
    BL_HIP(hipSetDevice(pf->ctx->device));
    BL_HIP(hipMemcpyAsync(out_runs, &pf->state->uni_n, sizeof(int), hipMemcpyDeviceToHost, pf->ctx->stream));
    BL_HIP(hipStreamSynchronize(pf->ctx->stream));
    return BL_OK;
}

// ---- the measurement MCL_TRIG_EPS rests on, as a call (tests/test_gpu_trig_guard.py runs it on every driver run): the largest
// |hw_sincos_unwrapped(d) - (sinf, cosf)(wrap_to_pi(d))| over EVERY float d in [-3 pi - 0.01, pi + 0.01], the range of a wrapped
// pose angle less a scan angle in [0, 6.2831] (sensor_model.cpp:34-37 through moving_laser_scan.cpp:33).  ~0.7 s on the device.
__global__ __launch_bounds__(256) void k_trig_probe(uint32_t lo_bits, uint32_t count, unsigned int* out_max)
{
    float ms = 0.f, mc = 0.f;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < count; i += (uint64_t)gridDim.x * blockDim.x) {
        const float d = __uint_as_float(lo_bits + (uint32_t)i);
        float sn, cs, hs, hc;
        bl_sincosf_cells(bl_wrap_to_pi(d), &sn, &cs);
        hw_sincos_unwrapped(d, &hs, &hc);
        ms = fmaxf(ms, fabsf(hs - sn)); mc = fmaxf(mc, fabsf(hc - cs));
        if (!(fabsf(hs - sn) <= 1.0f)) ms = 2.0f;               // a nan difference must not hide in fmaxf
        if (!(fabsf(hc - cs) <= 1.0f)) mc = 2.0f;
    }
    for (int off = 32; off > 0; off >>= 1) { ms = fmaxf(ms, __shfl_xor(ms, off, 64)); mc = fmaxf(mc, __shfl_xor(mc, off, 64)); }
    if ((threadIdx.x & 63) == 0) {                              // non-negative floats order like their bit patterns
        atomicMax(&out_max[0], __float_as_uint(ms));
        atomicMax(&out_max[1], __float_as_uint(mc));
    }
}

extern "C" int bl_debug_trig_probe(bl_ctx* ctx, float* max_sin_err, float* max_cos_err, float* eps_used, uint64_t* floats_checked)
{
    BL_CHECK_ARG(ctx != nullptr && max_sin_err != nullptr && max_cos_err != nullptr);
    BL_HIP(hipSetDevice(ctx->device));
    unsigned int* d_max = nullptr;
    BL_HIP(hipMalloc((void**)&d_max, 8));
    BL_HIP(hipMemsetAsync(d_max, 0, 8, ctx->stream));
    const float hi_pos = 3.1515927f, hi_neg = 9.4347780f;       // pi + 0.01, 3 pi + 0.01
    uint32_t bp, bn;
    memcpy(&bp, &hi_pos, 4); memcpy(&bn, &hi_neg, 4);
    // positive floats 0 .. hi_pos: bit patterns 0 .. bp; negative floats -0 .. -hi_neg: 0x80000000 .. 0x80000000 + bn
    hipLaunchKernelGGL(k_trig_probe, dim3(4096), dim3(256), 0, ctx->stream, 0u, bp + 1u, d_max);
    hipLaunchKernelGGL(k_trig_probe, dim3(4096), dim3(256), 0, ctx->stream, 0x80000000u, bn + 1u, d_max);
    BL_HIP(hipGetLastError());
    float h[2];
    BL_HIP(hipMemcpyAsync(h, d_max, 8, hipMemcpyDeviceToHost, ctx->stream));
    BL_HIP(hipStreamSynchronize(ctx->stream));
    BL_HIP(hipFree(d_max));
    *max_sin_err = h[0]; *max_cos_err = h[1];
    if (eps_used) *eps_used = MCL_TRIG_EPS;
    if (floats_checked) *floats_checked = (uint64_t)bp + 1ull + (uint64_t)bn + 1ull;
    return BL_OK;
}

// |direction by the addition theorems - the reference's sinf / cosf of wrap_to_pi(fl(p - r))| over random pairs: p a float of
// [-pi, pi] (a particle's wrapped heading), r a float of [0, 6.2831] (a theta_simple scan's ray angle), with the very functions the
// ray loop calls (ray_table_entry, trig_by_addition; bl_sincosf_cells / wrap_to_pi_cells on the exact side).
__global__ __launch_bounds__(256) void k_trig_addition_probe(unsigned long long pairs_per_thread, unsigned int seed, unsigned int* out_max)
{
    unsigned int ms = 0, mc = 0;
    // xorshift128+ per thread, seeded by a mix of the thread index
    unsigned long long s0 = (0x9E3779B97F4A7C15ull * ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x + 1ull)) ^ ((unsigned long long)seed << 17);
    unsigned long long s1 = 0xD1B54A32D192ED03ull * (s0 | 1ull);
    for (unsigned long long k = 0; k < pairs_per_thread; ++k) {
        unsigned long long x = s0; const unsigned long long y = s1;
        s0 = y; x ^= x << 23; s1 = x ^ y ^ (x >> 17) ^ (y >> 26);
        const unsigned long long rnd = s1 + y;
        const float up = (float)(unsigned int)(rnd >> 40) * (1.0f / 16777216.0f);           // 24 bits each
        const float ur = (float)(unsigned int)((rnd >> 16) & 0xFFFFFFull) * (1.0f / 16777216.0f);
        float p = bl_wrap_to_pi((up - 0.5f) * 6.2831855f);
        float r = ur * BL_THETA_SIMPLE_MAX;
        // every fourth pair from the corners uniform sampling rarely meets: p within a few ulps of +-pi, r within a few ulps of 0 or
        // of the cap (where p - r needs its one wrap and comes closest to a second), p - r within a few ulps of a multiple of pi / 2
        if ((rnd & 3ull) == 0ull) {
            const unsigned int j = (unsigned int)(rnd >> 2) & 7u, kind = (unsigned int)(rnd >> 5) & 7u;
            const float pi_f = 3.14159274f;
            if (kind & 1u) p = __uint_as_float(__float_as_uint(pi_f) - j) * ((kind & 2u) ? -1.0f : 1.0f);
            if (kind & 4u) r = (kind & 2u) ? __uint_as_float(__float_as_uint(BL_THETA_SIMPLE_MAX) - j) : __uint_as_float(j * 3u);
            else if (!(kind & 1u)) {
                const float q = 1.57079637f * (float)((int)((rnd >> 8) & 7ull) - 5);             // p - r near q
                r = __builtin_fminf(__builtin_fmaxf(p - q, 0.0f), BL_THETA_SIMPLE_MAX);
                r = __uint_as_float(__float_as_uint(r) + (r > 0.0f && r < BL_THETA_SIMPLE_MAX ? j : 0u) - (r >= BL_THETA_SIMPLE_MAX ? j : 0u));
            }
        }
        float ps, pc;
        bl_sincosf(p, &ps, &pc);
        // the form the ray loop runs: the pair scaled by the cells per metre (20 for the shipped maps; every fourth pair with a scale
        // that is no power of two times a short mantissa), the result taken back to a direction in double
        const float cpm = (rnd & 12ull) == 4ull ? 13.7f + (float)((rnd >> 44) & 1023ull) * 0.01f : 20.0f;
        const float2_t pcs = {cpm * pc, cpm * ps};
        const float4 rt = ray_table_entry(1.0f, r);
        const float2_t dc = trig_by_addition(pcs, rt.z, rt.w);
        float sn, cs;
        bl_sincosf_cells(wrap_to_pi_cells(p - r, true), &sn, &cs);
        const float es = (float)__builtin_fabs((double)dc.y / (double)cpm - (double)sn), ec = (float)__builtin_fabs((double)dc.x / (double)cpm - (double)cs);
        ms = max(ms, __float_as_uint(es)); mc = max(mc, __float_as_uint(ec));
    }
    for (int off = 32; off > 0; off >>= 1) { ms = max(ms, (unsigned int)__shfl_xor((int)ms, off, 64)); mc = max(mc, (unsigned int)__shfl_xor((int)mc, off, 64)); }
    if ((threadIdx.x & 63) == 0) { atomicMax(out_max, ms); atomicMax(out_max + 1, mc); }
}

extern "C" int bl_debug_trig_addition_probe(bl_ctx* ctx, uint64_t pairs, uint32_t seed, float* max_sin_err, float* max_cos_err, float* eps_used,
                                            uint64_t* pairs_checked)
{
    BL_CHECK_ARG(ctx != nullptr && max_sin_err != nullptr && max_cos_err != nullptr && pairs >= 1);
    BL_HIP(hipSetDevice(ctx->device));
    unsigned int* d_max = nullptr;
    BL_HIP(hipMalloc((void**)&d_max, 8));
    BL_HIP(hipMemsetAsync(d_max, 0, 8, ctx->stream));
    const unsigned long long threads = 8192ull * 256ull;
    const unsigned long long per = (pairs + threads - 1) / threads;
    hipLaunchKernelGGL(k_trig_addition_probe, dim3(8192), dim3(256), 0, ctx->stream, per, seed, d_max);
    BL_HIP(hipGetLastError());
    float h[2];
    BL_HIP(hipMemcpyAsync(h, d_max, 8, hipMemcpyDeviceToHost, ctx->stream));
    BL_HIP(hipStreamSynchronize(ctx->stream));
    BL_HIP(hipFree(d_max));
    *max_sin_err = h[0]; *max_cos_err = h[1];
    if (eps_used) *eps_used = MCL_TRIG_EPS;
    if (pairs_checked) *pairs_checked = per * threads;
    return BL_OK;
}

extern "C" int bl_pf_debug_enable(bl_pf* pf, int on)
{
    BL_CHECK_ARG(pf != nullptr);
    // (what bl_pf_debug_last returns before the first resampling update is zeros, not whatever the allocation held)
    if (on && !pf->debug && pf->dbg_idx) {
        BL_HIP(hipSetDevice(pf->ctx->device));
        BL_HIP(hipMemsetAsync(pf->dbg_idx, 0, (size_t)pf->n_local * sizeof(int32_t), pf->ctx->stream));
        BL_HIP(hipMemsetAsync(pf->dbg_like, 0, (size_t)pf->n_local * sizeof(int32_t), pf->ctx->stream));
    }
    pf->debug = on != 0;
    return BL_OK;
}

extern "C" int bl_pf_debug_last(bl_pf* pf, int32_t* resample_idx, int32_t* likelihood_half_units)
{
    BL_CHECK_ARG(pf != nullptr && pf->dbg_idx != nullptr);
    if (!pf->debug) { bl_set_error("bl_pf_debug_last needs bl_pf_debug_enable(pf, 1) before the update"); return BL_ERR_STATE; }
    if (resample_idx)
        BL_HIP(hipMemcpyAsync(resample_idx, pf->dbg_idx, (size_t)pf->n_local * 4, hipMemcpyDeviceToHost, pf->ctx->stream));
    if (likelihood_half_units)
        BL_HIP(hipMemcpyAsync(likelihood_half_units, pf->dbg_like, (size_t)pf->n_local * 4, hipMemcpyDeviceToHost, pf->ctx->stream));
    BL_HIP(hipStreamSynchronize(pf->ctx->stream));
    return BL_OK;
}
