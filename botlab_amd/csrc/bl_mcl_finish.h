// bl_mcl_finish.h -- the end of a particle-filter update (weight-unit prefix + estimatePosteriorPose, particle_filter.cpp:116-160)
// as device functions, so that it can ride in another kernel's launch: k_map_update runs it beside Mapping::updateMap
// (bl_mapping_update_finishing_pf), which needs the pose estimate and nothing else of it.  k_mcl_finish (bl_mcl.hip) is the
// stand-alone launch of the same functions.
//
// One launch = the FINISHER (workgroup 0) + `groups` GROUP workgroups (the first `gthreads` threads of each work):
//   group     exact integer prefix of its weight units (written out), and for each of its waves' 128 particles (a "sub-tile")
//             one RECORD per axis of the serially rounded float sums pose.x / pose.y (bl_serial_sum.h), formed for the binade
//             a double-precision running sum predicts.  A sub-tile in which that sum comes near a binade's end is RISKY: the
//             finisher will have to step through it, and the group leaves a TABLE for that (the terms, and their integer
//             prefix sums in the predicted binade and in the next one).  Then the group counts itself done.
//   finisher  waits for the groups; all its waves bring records and tables into LDS and join the records BETWEEN risky
//             sub-tiles into one composite per gap; then one wave per axis runs the chain with the TRUE accumulator: gap
//             (one check, one add), risky sub-tile (first step that leaves the binade from the table, that step in real
//             arithmetic, the rest of the sub-tile from the next binade's column), gap, ...  Whatever does not go by the book
//             (a gap that does not fit, a tie, a sign change, a second crossing) is walked / replayed generically from the
//             particle record: predictions and tables cost time when wrong, never correctness.
//             x, y are bit for bit the reference's float accumulators; theta comes from double sums.
//
// Visibility inside the launch: the MI355X's eight XCDs have separate L2s, and an agent-scope release fence (__threadfence)
// writes a whole L2 back -- 12 us measured with the prefix stores in it.  So everything a group hands to the finisher goes by
// agent-scope relaxed atomic stores / loads (write-through, read-through: sc1).  A record carries the launch's generation tag in
// both of its 8-byte halves: the finisher polls the record slots themselves and takes a record when both halves show this
// launch's tag -- no counter, no barrier, no wait in the groups; a wave that leaves a table or a wild map for a record waits for
// those stores before it stores the record.  Nothing else of the groups' output is read inside the launch.
#ifndef BL_MCL_FINISH_H
#define BL_MCL_FINISH_H

#include "bl_internal.h"
#include "bl_serial_sum.h"

// ---- equal weights.  On a set of N EQUAL weights w the reference's resampler (particle_filter.cpp:84-103) compares U_m = r + m / N with
// c_0 = w, c_i = fl(c_{i-1} + w): every U_m lies within the sum's rounding error of a partial sum when r is 0, nearly 0 or 1 / N, so an
// exact integer prefix picks the neighbouring source for about half of the particles there.  Equal weights are what a fresh filter holds
// (1 / N, D2), what an upload of equal weights holds, and what an update leaves when EVERY particle ends at the likelihood floor
// (computeNormalizedPosterior, :116-141: w = 0.001 / wSum, wSum the sequentially rounded sum of N times 0.001 -- a kidnapped or lost
// filter; the total of the weight units is then exactly 2 N, which no other set of weights gives: a unit is 0.0005, the floor 2 units,
// the smallest likelihood above it 1000).  For such a set the c_i have a closed form: inside a binade a step adds the same multiple of
// the binade's ulp every time (after one step inside it: a tie rounds to even, and from an even sum on the step is constant too), so the
// whole sequence is ~3 runs (c0, inc, n) per binade.  The launch that writes the total (the finisher, k_scan_write_prefix) forms the runs
// -- one thread, some tens of runs -- and the next k_mcl_main searches them instead of the prefix: U_m against the reference's own
// rounded cumulative, bit for bit, on any number of ranks (the runs need no particle data).
#define UNI_MAX 192
struct uni_seg { double c0, inc, last; int i0, n; };      // c_{i0 + k} = c0 + k * inc (exact), 0 <= k < n; last = c0 + (n - 1) * inc

struct pf_state {
    double S;                 // total weight units of rec[cur]
    bl_pose_xyt_t pose;       // posteriorPose_
    unsigned int wait_timeouts;    // waits inside a finish launch that ran into MCLF_SPIN_LIMIT (never, unless a launch lost workgroups); read with the pose
    unsigned int shard_broken;     // STICKY: a cross-rank wait of the peer-store exchange gave up (k_shard_wait): every later launch that would consume
                                   // another rank's data does nothing; only bl_pf_shard_setup / a new particle set clears it.  Read with the pose
    double sums_used[5];      // units, -, -, units*sin, units*cos the estimate was formed from (diagnostic)
    unsigned int chain_stats[8];   // x then y: generic replays, their phases, table replays, gaps walked the slow way (diagnostic)
    unsigned int lookahead[2];     // map updates that ran ahead of the exact pose; of those, the ones that had to run again (diagnostic)
    unsigned int pre_stats[2];     // x, y: sub-tiles the pre-chain took by a map << 16 | sub-tiles it replayed, summed over the launches (diagnostic)
    unsigned long long cstamps[16]; // the x chain, entry by entry (diagnostic, -DMCLF_STAMPS)
    unsigned long long gstamps[8]; // one group's timeline (diagnostic, -DMCLF_STAMPS)
    unsigned long long stamps[6];  // finisher timeline in 10 ns ticks (diagnostic, -DMCLF_STAMPS)
    int uni_n;                // > 0: rec[cur]'s weights are all equal and uni[0 .. uni_n) holds the reference's cumulative of them (see uni_seg)
    int uni_pad;
    uni_seg uni[UNI_MAX];
    unsigned long long xstamps[16]; // [0] the stamped group's loads are back; [1] the finisher's stage-a loads are back, [2] its wave 0 is through stage a; [3] the pre-chain has published, [4] the x chain has its start value
};

struct mclf_tab_elem { double t; int se, se1; };       // term; (inclusive prefix << 1 | bad) in the predicted binade and in the next

// Where the finish's inputs lie when the particle set is sharded over ranks and the finish is COMPOSED: every
// rank runs the groups over its own block of particles only and leaves their records, tables and table counts in its block of an
// exchange buffer; one small all-gather later every rank holds every block and runs pre-chain + finisher over all of them.
// Particle records of other ranks (the pre-chain's first sub-tiles, a generic replay) are read from the owner's memory.
#define BL_MAX_SHARDS 8
#define MCLF_XCHG_HDR 64                       // bytes in front of a rank's block: [0] its table counts (x: bits 0..15, y: 16..31)
struct mclf_shards {
    int world;
    int rank;
    int block;                                 // particles per rank: a multiple of the finish groups' chunk and of the scan tile
    int subs_per_rank;                         // sub-tile records per rank and axis
    const float4* rec[BL_MAX_SHARDS];          // per rank its exchange record, indexed by GLOBAL particle index (its own block is valid)
    char* xchg;                                // the ranks' blocks, xchg_stride bytes apart: [header][recs x][recs y][tables x][tables y]
    size_t xchg_stride;
};

// partials[b][5]: per block b the sums of units, units*x, units*y, units*sin(theta), units*cos(theta) of its particles (units
// exact; the others feed theta and the binade predictions only).  Blocks [0, main_blocks) own `tile` particles each from 0 on
// (clipped to main_particles), the rest own `tail_tile` particles each from main_particles on.
struct mcl_finish_args {
    const double* partials; int nblocks;
    const float4* rec; int N;
    int tile, main_blocks, main_particles, tail_tile;
    unsigned long long* prefix;
    pf_state* state;
    int64_t utime;
    int uni_mode;                          // equal weights (uni_seg): 0 = detect the all-floor set from the total, -1 = off (BOTLAB_NO_AUTO_STRICT)
    double w_floor;                        // the weight computeNormalizedPosterior leaves on N floor weights: 0.001 / (0.001 + ... + 0.001, N terms, rounded at every step); the host forms it once per filter
    ss_rec* recs;                          // [2][groups * subs]: x records, then y records
    mclf_tab_elem* tabs;                   // [2][MCLF_TSLOTS][MCLF_SUB]
    unsigned long long* sync;              // MCLF_SYNC_WORDS words, zero between launches:
                                           // [0] bits 0..15 / 16..31: x / y tables handed out
                                           // [1], [2] the x / y sum behind the first MCLF_PRE_SUBS sub-tiles: float bits | 1 << 32 ("it is there")
                                           // [3], [4] the finisher's exact x / y for the map workgroup of the same launch, likewise
                                           //     (mclf_pose with publish writes them, mclf_wait_pose reads and clears them)
    int groups, gthreads;                  // group workgroups (of the whole particle set); threads of each that work (256 or 1024)
    ss_wild* wild;                         // [2][sub-tiles]: the map of a sub-tile whose sum is predicted to cross binades (bl_serial_sum.h), or null
    int groups_wait;                       // groups of THIS launch the finisher waits for (composed finish: 0, they ran in an earlier launch)
    unsigned int tag;                      // generation of this launch's records: a record slot holds the previous launch's record (another
                                           // tag) until its group has stored this launch's (mclf_store_rec, mclf_decode_rec)
    const mclf_shards* sh;                 // device memory; null: one rank (a table in the argument block itself would be indexed
                                           // per lane, which moves a by-value argument into scratch for every thread of the kernel)
    int no_trees;                          // BOTLAB_MCL_NO_TREES: an overflowed list is walked by the chain's wave alone (tests, A/B)
};

#define MCLF_POSE_THREADS 256                 // the theta sums' addition order is that of a 256-thread workgroup, whoever runs it
#define MCLF_WG 1024                          // threads of every workgroup of the launch
#define MCLF_ITEMS 2
#define MCLF_SUB (64 * MCLF_ITEMS)            // particles per sub-tile
#define MCLF_GT_SMALL 256                     // working threads of a group, small / large particle counts
#define MCLF_GT_LARGE 1024
#define MCLF_GT_SWITCH 160000                 // particle count from which the large groups are used
#define MCLF_RISKY 0x1000                     // flag in a record's key
#define MCLF_WILD 0x2000                      // ... of a risky record: the sub-tile's wild map is in f.wild
#define MCLF_TSLOT_SHIFT 16                   // bits 16..23 of a record's key: 1 + the slot of its table
#define MCLF_KEY_MASK 0xfff
#define MCLF_MARGIN 4096                      // ulps of slack on the predicted start when deciding "risky"
#define MCLF_TSLOTS 20                        // per axis: tables
#define MCLF_MAXENT 32                        // per axis: risky sub-tiles the chain steps through by the list (more: the slow walk)
#define MCLF_PRE_SUBS 9                       // sub-tiles at the start of the sums that a workgroup of its own does while the groups run
#define MCLF_EXTRA_WGS 2                      // workgroups of the launch in front of the groups: finisher, pre-chain
#define MCLF_PRE_STEPPED 2                    // ... the first of them term by term (a binade change every few terms), the others in-binade
#define MCLF_TAB_WAVES 4                      // waves of the finisher that fetch tables while the others join gaps (four tables per wave and trip)
#define MCLF_SPIN_LIMIT (1u << 20)               // polls (~1.3 us each) after which a wait inside the launch gives up: ~1.4 s
#define MCLF_SYNC_WORDS 8                     // words of the sync block (one 64-byte line; five in use)
#define MCLF_LDS_BYTES (112 * 1024)           // scratch the finisher wants (the map kernel's counter window serves)

static inline int mclf_gthreads(int N) { return N >= MCLF_GT_SWITCH ? MCLF_GT_LARGE : MCLF_GT_SMALL; }
#define mclf_chunk(gthreads) ((gthreads) * MCLF_ITEMS)

// group workgroups a finish needs (tile and tail_tile must divide the chunk)
static inline int mclf_groups(const mcl_finish_args& f)
{
    const int chunk = mclf_chunk(f.gthreads);
    const int tpc_main = chunk / f.tile;
    const int tail_blocks = f.nblocks - f.main_blocks;
    int g = (f.main_blocks + tpc_main - 1) / tpc_main;
    if (tail_blocks > 0) { const int tpc_tail = chunk / f.tail_tile; g += (tail_blocks + tpc_tail - 1) / tpc_tail; }
    return g;
}

#if defined(__HIPCC__)
// ---- the runs of the equal-weight cumulative (uni_seg).  One thread; every addition below is the reference's own IEEE addition
// (the build has -ffp-contract=off; nothing here may be reassociated).
__device__ __forceinline__ int uni_exp(double v) { return (int)((__double_as_longlong(v) >> 52) & 0x7ffLL); }      // biased exponent

// c_0 = w, c_i = fl(c_{i-1} + w), i < N, as runs; returns their number (0: they do not fit `cap`, or w is not a positive normal
// number), *last = c_{N-1}.  seg == nullptr: only the last value is wanted (the reference's wSum of N floor weights).
// A run of constant increment starts at a sum that was reached BY A STEP INSIDE ITS BINADE (then a tie has rounded to even and every
// later step of the binade adds the same amount: with w = (q + 1/2) ulp and an even sum M, q even gives M + q -- even again --, q odd
// gives M + q + 1 -- even again; without a tie the step is RN(w / ulp) from any sum) and ends two units below the binade's top, so
// that every step of it rounds on the binade's own grid; the steps across a binade's top are taken one by one.
__device__ inline int uni_build(double w, int N, uni_seg* seg, int cap, double* last_out)
{
    if (!(w > 0.0) || uni_exp(w) == 0 || uni_exp(w) == 0x7ff || N <= 0) { *last_out = 0.0; return 0; }
    int ns = 0, i = 0;
    double c = w;                                              // c = c_i
    bool fits = true;
    auto emit = [&](int i0, double c0, double inc, int n) {
        if (seg) { if (ns < cap) { uni_seg s_; s_.c0 = c0; s_.inc = inc; s_.last = c0 + (double)(n - 1) * inc; s_.i0 = i0; s_.n = n; seg[ns] = s_; } else fits = false; }
        ns += 1;
    };
    while (true) {
        if (i >= N - 1) { emit(i, c, 0.0, 1); break; }
        const double c1 = c + w;                               // c_{i+1}
        const int e = uni_exp(c);
        bool run = uni_exp(c1) == e && i + 2 < N;
        double c2 = 0.0;
        if (run) { c2 = c1 + w; run = uni_exp(c2) == e; }
        if (!run) { emit(i, c, 0.0, 1); i += 1; c = c1; continue; }
        // c, c1, c2 in one binade: c1 was reached by a step inside it; from c1 on the step is inc while the sums stay inside
        const double inc = c2 - c1;                            // exact (both on the binade's grid)
        const double u = __longlong_as_double((long long)(e - 52) << 52);          // the binade's ulp, 2^(e - 1023 - 52) (e > 52 here: c >= w normal and ... see below)
        long long nrun = 1;
        if (e > 52) {
            const long long M1 = (long long)(c1 / u), Q = (long long)(inc / u);    // exact: integers below 2^53
            if (Q > 0) nrun = ((1LL << 53) - 2 - M1) / Q + 1;                      // c1 + k inc <= (2^53 - 2) u for k < nrun
            if (nrun < 1) nrun = 1;
        }
        if ((long long)i + 1 + nrun > (long long)N) nrun = (long long)N - (i + 1);
        emit(i, c, 0.0, 1);
        emit(i + 1, c1, inc, (int)nrun);
        const double tail = c1 + (double)(nrun - 1) * inc;     // exact
        i = i + 1 + (int)nrun;
        if (i >= N) { c = tail; break; }
        c = tail + w;                                          // c_i: a real step from the run's last sum
    }
    *last_out = c;
    return (fits || !seg) ? ns : 0;
}

// first i with T <= c_i, clamped to N - 1 (resamplePosteriorDistribution's `while (U > c)`, D4), from the runs
__device__ inline int uni_search(const pf_state* st, int uni_n, double T, int N)
{
    int lo = 0, hi = uni_n - 1;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (st->uni[mid].last >= T) hi = mid; else lo = mid + 1; }
    const uni_seg s_ = st->uni[lo];
    if (!(s_.last >= T)) return N - 1;
    if (T <= s_.c0 || s_.n <= 1) return s_.i0;
    long long k = (long long)((T - s_.c0) / s_.inc);
    if (k < 0) k = 0;
    if (k > s_.n - 1) k = s_.n - 1;
    while (k < s_.n - 1 && s_.c0 + (double)k * s_.inc < T) ++k;                    // (c0 + k inc is exact: a sum of the run)
    while (k > 0 && s_.c0 + (double)(k - 1) * s_.inc >= T) --k;
    return s_.i0 + (int)k;
}

// the launch that has just written the total S of the weight units decides whether the set is one of equal weights.  mode 1: the host
// knows it is (a fresh filter, an upload of equal weights: w = fl(units / S) = fl(1 / N)); 0: only the all-floor set is recognised
// (S == 2 N: w = w_floor = 0.001 / wSum as computeNormalizedPosterior leaves it, formed by the host); -1: never.
__device__ inline void uni_update(pf_state* st, int N, double S, int mode, double w_floor)
{
    int n = 0;
    if (mode >= 0) {
        double w = 0.0, last;
        if (mode == 1) w = 1.0 / (double)N;
        else if (S == 2.0 * (double)N) w = w_floor;
        if (w > 0.0) n = uni_build(w, N, st->uni, UNI_MAX, &last);
    }
    st->uni_n = n;
}

#define MCLF_MAXW (MCLF_WG / 64)
struct mclf_smem {
    unsigned long long off[MCLF_MAXW], wave[MCLF_MAXW], tot[MCLF_MAXW];
    double bx[MCLF_MAXW], by[MCLF_MAXW];          // sums of the blocks before the group, per wave
    double wx[MCLF_MAXW], wy[MCLF_MAXW];          // sums of the sub-tiles' terms
    double red[MCLF_POSE_THREADS / 64][5];
    float xy[2], first[2];                        // the sums; the sums behind the finisher's own sub-tiles
    unsigned int stats[8];
    int tbase[2][BL_MAX_SHARDS + 1];              // per axis: first staged slot of every rank's tables (composed finish; one rank: {0, MCLF_TSLOTS})
    int ntab_seen[2];                             // per axis: 1 + the highest table slot a record of this launch names (one rank)
    double pre[2][MCLF_PRE_STEPPED * MCLF_SUB];   // the terms the finisher steps one by one, per axis
    double psum[2][MCLF_PRE_SUBS];                // pre-chain: per axis the double sums of the first sub-tiles' terms (predictions)
    ss_wild pmap[2][MCLF_PRE_SUBS];               // pre-chain: per axis the wild maps of the sub-tiles behind the stepped ones
};

#ifdef MCLF_STAMPS
#define MCLF_NOW() ((unsigned long long)wall_clock64())
#else
#define MCLF_NOW() 0ull
#endif

__device__ __forceinline__ double mclf_wave_sum(double v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ double mclf_wave_sum_all(double v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ---- agent-scope relaxed accesses (sc1: through the L2 to memory, so that another XCD sees them without a cache writeback)
__device__ __forceinline__ void mclf_store_u64(unsigned long long* p, unsigned long long v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long mclf_load_u64(const unsigned long long* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// "my write-through stores have left": the sc1 stores above are ordered against a LATER flag store / counter add only if the
// wave really waits for them -- a workgroup-scope release fence emits no s_waitcnt vmcnt(0) on gfx950 outside tgsplit mode, and
// records and flag travel to different L2 channels.  No cache writeback is involved (the stores are write-through).
__device__ __forceinline__ void mclf_drain_stores()
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
// A record in memory: two 8-byte words, each stored by one relaxed atomic store (the unit the memory model keeps whole), each
// with the launch's tag in it -- word 0: key (24 bits; the empty run's key SS_ID as MCLF_KEY_ID) | tag bits 0..7 << 24 | D << 32;
// word 1: lo << 1 | tag bit 0, hi << 1 | tag bit 1 (|lo|, |hi| <= 2^29).  Every launch stores every record slot once, so a
// slot holds the previous launch's record or this one's, and the tags of consecutive launches differ in both words.
#define MCLF_KEY_ID 0x8000
__device__ __forceinline__ void mclf_store_rec(ss_rec* p, const ss_rec& r, unsigned int tag)
{
    unsigned long long* q = (unsigned long long*)p;
    const unsigned int k = (r.key == SS_ID ? (unsigned int)MCLF_KEY_ID : (unsigned int)r.key) | ((tag & 0xffu) << 24);
    const unsigned int lo = ((unsigned int)r.lo << 1) | (tag & 1u), hi = ((unsigned int)r.hi << 1) | ((tag >> 1) & 1u);
    mclf_store_u64(q, (unsigned long long)k | ((unsigned long long)(unsigned int)r.D << 32));
    mclf_store_u64(q + 1, (unsigned long long)lo | ((unsigned long long)hi << 32));
}
// the four dwords of a record slot; *current: both words carry `tag`
__device__ __forceinline__ ss_rec mclf_decode_rec(const int4& q, unsigned int tag, bool* current)
{
    const unsigned int k = (unsigned int)q.x;
    *current = (k >> 24) == (tag & 0xffu) && ((unsigned int)q.z & 1u) == (tag & 1u) && ((unsigned int)q.w & 1u) == ((tag >> 1) & 1u);
    const int key = (k & 0xffffffu) == (unsigned int)MCLF_KEY_ID ? SS_ID : (int)(k & 0xffffffu);
    return ss_rec_make(key, q.y, q.z >> 1, q.w >> 1);
}
// a record that is known to be there (the finisher has seen every slot with this launch's tag; a composed finish's arrived
// through the all-gather)
__device__ __forceinline__ ss_rec mclf_load_rec(const ss_rec* p)
{
    const unsigned long long* q = (const unsigned long long*)p;
    const unsigned long long a = mclf_load_u64(q), b = mclf_load_u64(q + 1);
    bool current;
    return mclf_decode_rec(make_int4((int)(unsigned int)a, (int)(unsigned int)(a >> 32), (int)(unsigned int)b, (int)(unsigned int)(b >> 32)), 0u, &current);
}
__device__ __forceinline__ void mclf_store_tab(mclf_tab_elem* p, double t, int se, int se1)
{
    unsigned long long* q = (unsigned long long*)p;
    mclf_store_u64(q, (unsigned long long)__double_as_longlong(t));
    mclf_store_u64(q + 1, (unsigned long long)(unsigned int)se | ((unsigned long long)(unsigned int)se1 << 32));
}
__device__ __forceinline__ mclf_tab_elem mclf_load_tab(const mclf_tab_elem* p)
{
    const unsigned long long* q = (const unsigned long long*)p;
    const unsigned long long a = mclf_load_u64(q), b = mclf_load_u64(q + 1);
    mclf_tab_elem e;
    e.t = __longlong_as_double((long long)a); e.se = (int)(unsigned int)b; e.se1 = (int)(unsigned int)(b >> 32);
    return e;
}

__device__ __forceinline__ void mclf_store_wild(ss_wild* p, const ss_wild& w)
{
    unsigned long long* q = (unsigned long long*)p;
    mclf_store_u64(q, (unsigned long long)(unsigned int)w.key_in | ((unsigned long long)(unsigned int)w.key_out << 32));
    mclf_store_u64(q + 1, (unsigned long long)(unsigned int)w.q | ((unsigned long long)(unsigned int)w.r << 32));
    mclf_store_u64(q + 2, (unsigned long long)w.a); mclf_store_u64(q + 3, (unsigned long long)w.c);
    mclf_store_u64(q + 4, (unsigned long long)w.L); mclf_store_u64(q + 5, (unsigned long long)w.H);
}
__device__ __forceinline__ ss_wild mclf_load_wild(const ss_wild* p)
{
    const unsigned long long* q = (const unsigned long long*)p;
    const unsigned long long k = mclf_load_u64(q), s2 = mclf_load_u64(q + 1);
    ss_wild w;
    w.key_in = (int)(unsigned int)k; w.key_out = (int)(unsigned int)(k >> 32); w.q = (int)(unsigned int)s2; w.r = (int)(unsigned int)(s2 >> 32);
    w.a = (long long)mclf_load_u64(q + 2); w.c = (long long)mclf_load_u64(q + 3);
    w.L = (long long)mclf_load_u64(q + 4); w.H = (long long)mclf_load_u64(q + 5);
    return w;
}
static_assert(sizeof(ss_wild) == 48, "wild records are stored as six 64-bit words");
__device__ __forceinline__ long long mclf_shfl_up_i64(long long v, int off)
{
    const int lo = __shfl_up((int)(unsigned int)(unsigned long long)v, off, 64), hi = __shfl_up((int)(unsigned int)((unsigned long long)v >> 32), off, 64);
    return (long long)(((unsigned long long)(unsigned int)hi << 32) | (unsigned long long)(unsigned int)lo);
}
__device__ __forceinline__ ss_wild mclf_shfl_up_wild(const ss_wild& w, int off)
{
    ss_wild o;
    o.key_in = __shfl_up(w.key_in, off, 64); o.key_out = __shfl_up(w.key_out, off, 64); o.q = __shfl_up(w.q, off, 64); o.r = __shfl_up(w.r, off, 64);
    o.a = mclf_shfl_up_i64(w.a, off); o.c = mclf_shfl_up_i64(w.c, off); o.L = mclf_shfl_up_i64(w.L, off); o.H = mclf_shfl_up_i64(w.H, off);
    return o;
}
__device__ __forceinline__ long long mclf_readlane_i64_(long long v, int lane)
{
    const int lo = __builtin_amdgcn_readlane((int)(unsigned int)(unsigned long long)v, lane), hi = __builtin_amdgcn_readlane((int)(unsigned int)((unsigned long long)v >> 32), lane);
    return (long long)(((unsigned long long)(unsigned int)hi << 32) | (unsigned long long)(unsigned int)lo);
}
__device__ __forceinline__ ss_wild mclf_readlane_wild(const ss_wild& w, int lane)
{
    ss_wild o;
    o.key_in = __builtin_amdgcn_readlane(w.key_in, lane); o.key_out = __builtin_amdgcn_readlane(w.key_out, lane);
    o.q = __builtin_amdgcn_readlane(w.q, lane); o.r = __builtin_amdgcn_readlane(w.r, lane);
    o.a = mclf_readlane_i64_(w.a, lane); o.c = mclf_readlane_i64_(w.c, lane); o.L = mclf_readlane_i64_(w.L, lane); o.H = mclf_readlane_i64_(w.H, lane);
    return o;
}

// ---- where things lie (one rank: the filter's own arrays; composed finish: the exchange blocks, mclf_shards)
// (read through a pointer that SAYS global memory: what comes out of the shard table is a generic pointer to the compiler, and a
// generic load is a flat load with a full wait behind it -- eighteen of those in a row held the pre-chain's first barrier up by 2 us)
typedef float mclf_f4 __attribute__((ext_vector_type(4)));
typedef const mclf_f4 __attribute__((address_space(1)))* mclf_gptr4;
__device__ __forceinline__ const float4* mclf_particle_ptr(const mcl_finish_args& f, int i)
{
    return f.sh ? f.sh->rec[i / f.sh->block] + i : f.rec + i;
}
__device__ __forceinline__ float4 mclf_particle(const mcl_finish_args& f, int i)
{
    const mclf_f4 v = *(mclf_gptr4)mclf_particle_ptr(f, i);
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ ss_rec* mclf_rec_ptr(const mcl_finish_args& f, int axis, int s)
{
    if (f.sh) {
        const int spr = f.sh->subs_per_rank, r = s / spr;
        return (ss_rec*)(f.sh->xchg + (size_t)r * f.sh->xchg_stride + MCLF_XCHG_HDR) + (size_t)axis * spr + (s - r * spr);
    }
    return f.recs + (size_t)axis * ((size_t)f.groups * (f.gthreads >> 6)) + s;
}
__device__ __forceinline__ mclf_tab_elem* mclf_tab_ptr(const mcl_finish_args& f, int axis, int rank, int slot)
{
    if (f.sh)
        return (mclf_tab_elem*)(f.sh->xchg + (size_t)rank * f.sh->xchg_stride + MCLF_XCHG_HDR + (size_t)2 * f.sh->subs_per_rank * sizeof(ss_rec)) +
               ((size_t)axis * MCLF_TSLOTS + slot) * MCLF_SUB;
    return f.tabs + ((size_t)axis * MCLF_TSLOTS + slot) * MCLF_SUB;
}
// the word tables are handed out from (x: bits 0..15, y: bits 16..31) -- this rank's own
__device__ __forceinline__ unsigned long long* mclf_tab_counter(const mcl_finish_args& f)
{
    if (f.sh) return (unsigned long long*)(f.sh->xchg + (size_t)f.sh->rank * f.sh->xchg_stride);
    return f.sync;
}
// tables rank r handed out on an axis (at most MCLF_TSLOTS have a slot)
__device__ __forceinline__ int mclf_tab_count(const mcl_finish_args& f, unsigned long long own_word, int r, int axis)
{
    unsigned long long w = own_word;
    if (f.sh) w = mclf_load_u64((const unsigned long long*)(f.sh->xchg + (size_t)r * f.sh->xchg_stride));
    return min((int)((w >> (axis ? 16 : 0)) & 0xffffull), MCLF_TSLOTS);
}
__device__ __forceinline__ int mclf_rank_of_sub(const mcl_finish_args& f, int s) { return f.sh ? s / f.sh->subs_per_rank : 0; }
__device__ __forceinline__ int mclf_world(const mcl_finish_args& f) { return f.sh ? f.sh->world : 1; }

// particles [lo, hi) of group g and the first block that belongs to it
__device__ __forceinline__ void mclf_group_range(const mcl_finish_args& f, int g, int* first_block, int* lo, int* hi)
{
    const int chunk = mclf_chunk(f.gthreads);
    const int tpc_main = chunk / f.tile;
    const int main_groups = (f.main_blocks + tpc_main - 1) / tpc_main;
    if (g < main_groups) {
        *first_block = g * tpc_main;
        *lo = *first_block * f.tile;
        *hi = min(f.main_particles, *lo + chunk);
    } else {
        const int gt = g - main_groups;
        *first_block = f.main_blocks + gt * (chunk / f.tail_tile);
        *lo = f.main_particles + gt * chunk;
        *hi = min(f.N, *lo + chunk);
    }
}
// particles [lo, hi) of sub-tile s (global index: group * waves-per-group + wave)
__device__ __forceinline__ void mclf_sub_range(const mcl_finish_args& f, int s, int* lo, int* hi)
{
    // (a group's chunk is its sub-tiles side by side, and the groups of the main region stand side by side: no division for a
    // sub-tile that lies wholly inside that region -- the pre-chain asks for its nine one after the other, in one wave)
    if ((s + 1) * MCLF_SUB <= f.main_particles) { *lo = s * MCLF_SUB; *hi = *lo + MCLF_SUB; return; }
    const int subs = f.gthreads >> 6;
    int first_block, glo, ghi;
    mclf_group_range(f, s / subs, &first_block, &glo, &ghi);
    *lo = glo + (s % subs) * MCLF_SUB;
    *hi = min(ghi, *lo + MCLF_SUB);
    if (*hi < *lo) *hi = *lo;
}

// The term of particle r on an axis: t = fl64(w * x), w = fl64(units / S) (particle_filter.cpp:136-138, 151-152).  With the
// units exact integers, units / S is the same real quotient as the reference's weight / wSum whenever no weight was floored to
// 0.001 (then wSum itself is a rounded sum; DESIGN.md "Pose estimate").
// The all-floor set (uni_seg) is the one case in which that matters to every term: the reference's weight is w_floor = 0.001 / wSum
// for all N particles (a few 1e-12 away from 2 / S = 1 / N: a last-bit difference in one term of a few thousand, and the float sum
// rounds differently there).  A NEGATIVE S says "the weight is -S" (mclf_term_S).
__device__ __forceinline__ double mclf_term(const float4& r, double S, int axis)
{
    const double w = S < 0.0 ? -S : (double)__float_as_uint(r.w) / S;
    return w * (double)(axis ? r.y : r.x);
}
__device__ __forceinline__ double mclf_term_S(const mcl_finish_args& f, double S)
{
    return (f.uni_mode >= 0 && S == 2.0 * (double)f.N) ? -f.w_floor : S;
}
__device__ __forceinline__ void mclf_load_terms(const mcl_finish_args& f, int axis, double S, int lo, int hi, int lane, double (&t)[MCLF_ITEMS])
{
#pragma unroll
    for (int k = 0; k < MCLF_ITEMS; ++k) {
        const int i = lo + lane * MCLF_ITEMS + k;
        t[k] = i < hi ? mclf_term(mclf_particle(f, i), S, axis) : 0.0;
    }
}

// ---- wave-wide integer scans on DPP (row shifts inside the 16-lane rows, then the two row broadcasts of gfx9): six dependent
// VALU operations instead of six LDS-crossbar round trips -- these sit on the serial path of the chain.
// update_dpp(old, src, ctrl, row_mask, bank_mask, bound_ctrl = false): a lane without a source lane, or outside the row mask,
// receives `old`, which is the operation's identity here.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int mclf_dpp(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, 0xf, false); }
#define MCLF_DPP_STEPS(STEP) \
    STEP(0x111, 0xf) STEP(0x112, 0xf) STEP(0x114, 0xf) STEP(0x118, 0xf) STEP(0x142, 0xa) STEP(0x143, 0xc)
// row_shr:1, 2, 4, 8; row_bcast:15 into rows 1 and 3; row_bcast:31 into rows 2 and 3

__device__ __forceinline__ int mclf_scan_add(int v)
{
#define MCLF_STEP(C, R) v += mclf_dpp<C, R>(0, v);
    MCLF_DPP_STEPS(MCLF_STEP)
#undef MCLF_STEP
    return v;
}
__device__ __forceinline__ int mclf_wave_min(int v)
{
#define MCLF_STEP(C, R) v = min(v, mclf_dpp<C, R>(0x7fffffff, v));
    MCLF_DPP_STEPS(MCLF_STEP)
#undef MCLF_STEP
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int mclf_wave_max(int v)
{
#define MCLF_STEP(C, R) v = max(v, mclf_dpp<C, R>((int)0x80000000, v));
    MCLF_DPP_STEPS(MCLF_STEP)
#undef MCLF_STEP
    return __builtin_amdgcn_readlane(v, 63);
}
// inclusive scan of records by composition (lane l: records 0..l of the wave joined in order)
__device__ __forceinline__ ss_rec mclf_scan_join(ss_rec r)
{
#define MCLF_STEP(C, R) { const ss_rec o = ss_rec_make(mclf_dpp<C, R>(SS_ID, r.key), mclf_dpp<C, R>(0, r.D), mclf_dpp<C, R>(0, r.lo), mclf_dpp<C, R>(0, r.hi)); \
                          r = ss_rec_join(o, r); }
    MCLF_DPP_STEPS(MCLF_STEP)
#undef MCLF_STEP
    return r;
}
// the same, restarting behind every lane whose `head` is set: lane l gets the join of the records from the last head <= l on
__device__ __forceinline__ ss_rec mclf_scan_join_segmented(ss_rec r, int head)
{
#define MCLF_STEP(C, R) { const ss_rec o = ss_rec_make(mclf_dpp<C, R>(SS_ID, r.key), mclf_dpp<C, R>(0, r.D), mclf_dpp<C, R>(0, r.lo), mclf_dpp<C, R>(0, r.hi)); \
                          const int oh = mclf_dpp<C, R>(0, head);                                                                                      \
                          const ss_rec j = ss_rec_join(o, r);                                                                                          \
                          r.key = head ? r.key : j.key; r.D = head ? r.D : j.D; r.lo = head ? r.lo : j.lo; r.hi = head ? r.hi : j.hi;                  \
                          head |= oh; }
    MCLF_DPP_STEPS(MCLF_STEP)
#undef MCLF_STEP
    return r;
}

// Quantized increments of this lane's terms in binade `key`, their inclusive prefix over the wave (p[k]), a bad flag per term.
__device__ __forceinline__ void mclf_prefix_in(int key, const double (&t)[MCLF_ITEMS], int cnt, int (&p)[MCLF_ITEMS], int (&bad)[MCLF_ITEMS])
{
    const ss_bin b = ss_bin_of(key);
    int run = 0;
#pragma unroll
    for (int k = 0; k < MCLF_ITEMS; ++k) {
        int bk = 0;
        const int d = ss_quantize(b, t[k], &bk);
        run += k < cnt ? d : 0; bad[k] = k < cnt ? bk : 0;
        p[k] = run;
    }
    const int excl = mclf_scan_add(run) - run;
#pragma unroll
    for (int k = 0; k < MCLF_ITEMS; ++k) p[k] += excl;
}

// The wild map of a sub-tile (bl_serial_sum.h): one scan of step maps over the wave, for the binade sequence a double-precision
// prefix sum of the terms predicts (wave-uniform; key_in 0: none that is valid for the predicted input binade `key`) ...
__device__ __forceinline__ ss_wild mclf_wild_map(const double (&t)[MCLF_ITEMS], int cnt, double predicted_start, int key, int lane)
{
    static_assert(MCLF_ITEMS == 2, "two terms per lane");
    const double ls = (cnt > 0 ? t[0] : 0.0) + (cnt > 1 ? t[1] : 0.0);
    double incl = ls;
    for (int off = 1; off < 64; off <<= 1) { const double o = __shfl_up(incl, off, 64); if (lane >= off) incl += o; }
    const double P1 = predicted_start + (incl - ls) + (cnt > 0 ? t[0] : 0.0), P2 = P1 + (cnt > 1 ? t[1] : 0.0);
    const int k2 = ss_key((float)P2);                                   // the key behind this lane's terms ...
    int k0 = __shfl_up(k2, 1, 64);                                      // ... is the key in front of the next lane's
    if (lane == 0) k0 = key;
    const int k1 = cnt > 1 ? ss_key((float)P1) : k2;
    ss_wild w = cnt > 0 ? ssw_step(k0, k1, t[0]) : ssw_identity(k0);
    if (cnt > 1) w = ssw_join(w, ssw_step(k1, k2, t[1]));
    for (int off = 1; off < 64; off <<= 1) {
        const ss_wild o = mclf_shfl_up_wild(w, off);
        if (lane >= off) w = ssw_join(o, w);
    }
    const ss_wild all = mclf_readlane_wild(w, 63);
    if (all.key_in == 0 || all.key_in != key) return ssw_invalid();
    return all;
}
// ... stored in f.wild when it is valid for the predicted input binade `key`
__device__ __forceinline__ bool mclf_build_wild(const mcl_finish_args& f, int axis, const double (&t)[MCLF_ITEMS], int cnt, double predicted_start,
                                             int key, int lane, int sub_index)
{
    const ss_wild all = mclf_wild_map(t, cnt, predicted_start, key, lane);
    if (all.key_in == 0) return false;
    const int nsub = f.groups * (f.gthreads >> 6);
    if (lane == 0) mclf_store_wild(f.wild + (size_t)axis * nsub + sub_index, all);
    return true;
}

// Record of one sub-tile (this wave's particles, MCLF_ITEMS consecutive ones per lane, `cnt` of them valid in this lane) for the
// binade of the predicted start value; a risky one also gets a table if a slot is left.
__device__ __forceinline__ ss_rec mclf_make_record(const mcl_finish_args& f, int axis, const double (&t)[MCLF_ITEMS], int cnt,
                                                   double predicted_start, bool very_first, int lane, int sub_index)
{
    // (the first sub-tiles of the sums are the finisher's own work, done while the groups run: the sums start from zero, so
    // nothing is needed from the groups there, and the binade changes every few terms.  Their records are empty runs: no gap
    // and no list entry ever covers them)
    if (very_first) return ss_rec_identity();
    const int key = __builtin_amdgcn_readfirstlane(ss_key((float)predicted_start));
    int p[MCLF_ITEMS] = {0, 0}, bad[MCLF_ITEMS] = {0, 0};
    bool risky = true, up_only = false, have_wild = false;
    ss_rec rec = ss_rec_make(MCLF_RISKY, 0, 0, 0);
    if (__builtin_amdgcn_ballot_w64(cnt > 0) == 0) return ss_rec_identity();                 // no particle in this sub-tile
    if (key) {
        mclf_prefix_in(key, t, cnt, p, bad);
        int lo = SS_SAT, hi = -SS_SAT;
#pragma unroll
        for (int k = 0; k < MCLF_ITEMS; ++k)
            if (k < cnt) { lo = min(lo, p[k]); hi = max(hi, p[k]); }
        lo = mclf_wave_min(lo); hi = mclf_wave_max(hi);
        const int D = __builtin_amdgcn_readlane(p[MCLF_ITEMS - 1], 63);
        const bool anybad = __builtin_amdgcn_ballot_w64((bad[0] | bad[1]) != 0) != 0;
        // "risky": with the predicted start magnitude and a margin for its error the run would not stay inside the binade
        const int Mp = ss_mag((float)predicted_start);
        risky = anybad || !(Mp + lo - MCLF_MARGIN > SS_MLO && Mp + hi + MCLF_MARGIN < SS_MHI);
        rec = anybad ? ss_rec_make(MCLF_RISKY, 0, 0, 0) : ss_rec_make(key | (risky ? MCLF_RISKY : 0), D, lo, hi);
        // the same prefix sums say HOW a risky sub-tile is predicted to leave, if at all: not downwards, and upwards no further
        // than the next binade (they keep measuring the magnitude there, in this binade's ulps) -- what a sum far from zero does
        // log2 N times, and what the sub-tiles right behind such a crossing look like (just above the binade's lower end)
        up_only = !anybad && Mp + lo > SS_MLO && Mp + hi + MCLF_MARGIN < 2 * SS_MHI && (key & 0xff) < 254;
    }
    // A risky sub-tile that only steps UP goes by a TABLE while slots last: the table finds the crossing with the true accumulator,
    // wherever it falls.  Every other one (a sum that hovers around zero leaves its binade up and down and through zero all the
    // time: the reference starts every run at the origin), and an upward one without a slot, gets a wild map below.
    const bool dirty = risky && f.wild != nullptr && key != 0;
    bool got_table = false;
    if (risky && f.tabs && (!dirty || up_only)) {
        // the table: terms, prefix in the predicted binade (if there is one), prefix in the next binade up
        unsigned long long got = 0;
        if (lane == 0) got = __hip_atomic_fetch_add(mclf_tab_counter(f), axis ? (1ull << 16) : 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int slot = __builtin_amdgcn_readfirstlane((int)((got >> (axis ? 16 : 0)) & 0xffffull));
        if (slot < MCLF_TSLOTS) {
            int p1[MCLF_ITEMS] = {0, 0}, bad1[MCLF_ITEMS] = {1, 1};
            const bool up = key != 0 && (key & 0xff) < 254;
            if (up) mclf_prefix_in(key + 1, t, cnt, p1, bad1);
            mclf_tab_elem* tab = mclf_tab_ptr(f, axis, f.sh ? f.sh->rank : 0, slot);
#pragma unroll
            for (int k = 0; k < MCLF_ITEMS; ++k) {
                const int se = key ? ((p[k] << 1) | bad[k]) : 1, se1 = up ? ((p1[k] << 1) | bad1[k]) : 1;
                mclf_store_tab(tab + lane * MCLF_ITEMS + k, t[k], se, se1);
            }
            rec.key |= (slot + 1) << MCLF_TSLOT_SHIFT;
            got_table = slot < MCLF_TSLOTS / 2;                // (a sum that has used up half of its tables is not one that only
                                                               // crosses a binade now and then: its sub-tiles get maps as well)
        }
    }
    if (dirty && !got_table) {
        have_wild = mclf_build_wild(f, axis, t, cnt, predicted_start, key, lane, sub_index);
        if (have_wild && !up_only) return ss_rec_make(MCLF_RISKY | MCLF_WILD, 0, 0, 0);
    }
    if (have_wild && (rec.key & MCLF_RISKY)) rec.key |= MCLF_WILD;          // (a record that is not risky goes by the book)
    return rec;
}

// Group g: the first f.gthreads threads of the workgroup work (the others leave at once); particles [lo, hi) of the group.
__device__ __forceinline__ void mclf_prefix_group(const mcl_finish_args& f, int g, mclf_smem& sm)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid >= f.gthreads) return;
    const int gt = f.gthreads, nw = gt >> 6;
#ifdef MCLF_STAMPS
    unsigned long long gs[8] = {MCLF_NOW(), 0, 0, 0, 0, 0, 0, 0};
#define MCLF_GSTAMP(i) gs[i] = MCLF_NOW()
#else
#define MCLF_GSTAMP(i) do { } while (0)
#endif
    int first_block, lo, hi;
    mclf_group_range(f, g, &first_block, &lo, &hi);
    // the group's particles are requested first, the block sums behind them: everything the group reads is in flight at once (the
    // sums lie in another XCD's memory: read one block per thread and trip, with the two prediction sums behind a branch, they
    // were four dependent trips of ~0.8 us each in front of everything else the group does)
    const int base = lo + tid * MCLF_ITEMS;
    float4 r[MCLF_ITEMS];
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < MCLF_ITEMS; ++k) {
        r[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (base + k < hi) { r[k] = f.rec[base + k]; cnt = k + 1; }
    }
    // sums over the blocks: units before the group and in total (exact integers), x / y sums before the group (prediction)
    unsigned long long before = 0, total = 0;
    double bx = 0.0, by = 0.0;
    for (int j0 = tid; j0 < f.nblocks; j0 += 4 * gt) {
        double pu[4], px4[4], py4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = j0 + q * gt;
            const size_t at = (size_t)(j < f.nblocks ? j : 0) * 5;              // (a clamped index: the loads carry no branch)
            pu[q] = f.partials[at]; px4[q] = f.partials[at + 1]; py4[q] = f.partials[at + 2];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = j0 + q * gt;
            if (j < f.nblocks) {
                const unsigned long long u = (unsigned long long)pu[q];
                total += u;
                if (j < first_block) { before += u; bx += px4[q]; by += py4[q]; }
            }
        }
    }
#ifdef MCLF_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0 && g == f.groups / 2) f.state->xstamps[0] = MCLF_NOW();
#endif
    for (int off = 32; off > 0; off >>= 1) { before += __shfl_xor(before, off, 64); total += __shfl_xor(total, off, 64); }
    bx = mclf_wave_sum_all(bx); by = mclf_wave_sum_all(by);
    if (lane == 0) { sm.off[wave] = before; sm.tot[wave] = total; sm.bx[wave] = bx; sm.by[wave] = by; }
    unsigned long long loc[MCLF_ITEMS];
    unsigned long long run = 0;
#pragma unroll
    for (int k = 0; k < MCLF_ITEMS; ++k) { run += __float_as_uint(r[k].w); loc[k] = run; }
    unsigned long long incl = run;
    for (int off = 1; off < 64; off <<= 1) {
        unsigned long long t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    if (lane == 63) sm.wave[wave] = incl;
    MCLF_GSTAMP(1);
    __syncthreads();
    MCLF_GSTAMP(2);
    unsigned long long off0 = incl - run, S_u = 0;
    double px = 0.0, py = 0.0;
    for (int w = 0; w < nw; ++w) { off0 += sm.off[w]; S_u += sm.tot[w]; px += sm.bx[w]; py += sm.by[w]; if (w < wave) off0 += sm.wave[w]; }
    // (the prefix is the NEXT kernel's input; the records are what the finisher of this launch is waiting for: they go first)
    if (!f.recs) {
#pragma unroll
        for (int k = 0; k < MCLF_ITEMS; ++k)
            if (base + k < hi) f.prefix[base + k] = off0 + loc[k];
        return;
    }
    MCLF_GSTAMP(3);
    // ---- the sub-tile records of the two float accumulators
    const double S = (double)S_u;
    const double St = mclf_term_S(f, S);
    double tx[MCLF_ITEMS], ty[MCLF_ITEMS];
    double sx = 0.0, sy = 0.0;
#pragma unroll
    for (int k = 0; k < MCLF_ITEMS; ++k) {
        tx[k] = mclf_term(r[k], St, 0); ty[k] = mclf_term(r[k], St, 1);
        sx += tx[k]; sy += ty[k];
    }
    sx = mclf_wave_sum_all(sx); sy = mclf_wave_sum_all(sy);
    if (lane == 0) { sm.wx[wave] = sx; sm.wy[wave] = sy; }
    MCLF_GSTAMP(4);
    __syncthreads();
    px /= S; py /= S;                                     // predicted accumulators where the group starts ...
    for (int w = 0; w < wave; ++w) { px += sm.wx[w]; py += sm.wy[w]; }      // ... and where this wave's sub-tile starts
    const bool very_first = g * nw + wave < MCLF_PRE_SUBS;
    const ss_rec rx = mclf_make_record(f, 0, tx, cnt, px, very_first, lane, g * nw + wave);
    const ss_rec ry = mclf_make_record(f, 1, ty, cnt, py, very_first, lane, g * nw + wave);
    MCLF_GSTAMP(5);
    // a record that names a table or a wild map must not be seen before them: this wave's write-through stores have left (no
    // cache writeback) before the record goes.  The record itself is its own flag (mclf_decode_rec): nothing waits for it here.
    const bool side_data = (rx.key != SS_ID && (rx.key & MCLF_RISKY) != 0) || (ry.key != SS_ID && (ry.key & MCLF_RISKY) != 0);
    if (side_data) mclf_drain_stores();
    if (lane == 0) {
        const int s = g * nw + wave;
        mclf_store_rec(mclf_rec_ptr(f, 0, s), rx, f.tag);
        mclf_store_rec(mclf_rec_ptr(f, 1, s), ry, f.tag);
    }
#pragma unroll
    for (int k = 0; k < MCLF_ITEMS; ++k)
        if (base + k < hi) f.prefix[base + k] = off0 + loc[k];
    MCLF_GSTAMP(6);
#ifdef MCLF_STAMPS
    if (tid == 0 && g == f.groups / 2) { gs[7] = MCLF_NOW(); for (int k = 0; k < 8; ++k) f.state->gstamps[k] = gs[k]; }
#endif
}

__device__ __forceinline__ double mclf_readlane_f64(double v, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// A sub-tile of n terms (lane l holds terms 2l, 2l+1) replayed from term `pos` on with the true accumulator by one wave (all 64
// lanes call it; acc is wave-uniform): in-binade integer prefix sums up to the first step that leaves the binade, ties or is
// too large, that step in real arithmetic, and on (bl_serial_sum.h).  Terms before `head` are simply stepped.
// lds_terms: MCLF_SUB doubles of LDS this wave may use, or null
__device__ __forceinline__ float mclf_replay(const double (&t)[MCLF_ITEMS], int n, int pos, int head, float acc, int lane, unsigned int* phases,
                                             double* lds_terms = nullptr)
{
    static_assert(MCLF_ITEMS == 2, "the replay indexes two terms per lane");
    n = __builtin_amdgcn_readfirstlane(n); pos = __builtin_amdgcn_readfirstlane(pos); head = __builtin_amdgcn_readfirstlane(head);
    if (pos < head) {
        // the first terms of a sum: the accumulator changes its binade every few terms, so they are stepped one by one (the
        // term reads do not depend on the accumulator; the loop carries three dependent operations per term)
        const int h = min(head, n);
        if (lds_terms) {
            // through LDS: the reads run ahead of the chain of roundings (30 cycles a term); fetched from the lanes' registers by
            // readlane with a computed lane, a term took 195 cycles -- 10.9 us for a sub-tile, of which a sum hovering around zero
            // has a dozen (100 000 particles) to a hundred (1 000 000)
            *(double2*)&lds_terms[2 * lane] = make_double2(t[0], t[1]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 8
            for (int i = pos; i < h; ++i) acc = ss_exact_step(acc, lds_terms[i]);       // (plain reads behind the statement above: the compiler batches them ahead of the roundings)
        } else {
#pragma unroll 8
            for (int i = pos; i < h; ++i) acc = ss_exact_step(acc, mclf_readlane_f64((i & 1) ? t[1] : t[0], i >> 1));
        }
        pos = h;
    }
    int budget = 4;                                                   // phases before the rest is simply stepped: a sub-tile that leaves
                                                                      // its binade again and again costs 0.7 us per phase, 2.5 us stepped
    while (pos < n) {
        const int key = __builtin_amdgcn_readfirstlane(ss_key(acc));
        if (key && budget == 0 && lds_terms) {                        // the phases are spent: the rest term by term, through LDS
            *(double2*)&lds_terms[2 * lane] = make_double2(t[0], t[1]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 8
            for (int i = pos; i < n; ++i) acc = ss_exact_step(acc, lds_terms[i]);
            break;
        }
        if (!key || budget == 0) {                                    // no usable binade (zero, tiny, not finite): the step itself
            const double tj = mclf_readlane_f64((pos & 1) ? t[1] : t[0], pos >> 1);
            acc = ss_exact_step(acc, tj);
            pos++;
            continue;
        }
        budget--;
        *phases += 1;
        const int M = ss_mag(acc);
        const ss_bin b = ss_bin_of(key);
        int bad[MCLF_ITEMS], p[MCLF_ITEMS];
        int run = 0;
#pragma unroll
        for (int k = 0; k < MCLF_ITEMS; ++k) {
            const int i = lane * MCLF_ITEMS + k;
            int bk = 0;
            const int d = ss_quantize(b, t[k], &bk);
            const bool on = i >= pos && i < n;
            run += on ? d : 0; bad[k] = on ? bk : 0;
            p[k] = run;
        }
        const int excl = mclf_scan_add(run) - run;
        int Mi[MCLF_ITEMS];
        bool out[MCLF_ITEMS];
#pragma unroll
        for (int k = 0; k < MCLF_ITEMS; ++k) {
            const int i = lane * MCLF_ITEMS + k;
            Mi[k] = M + excl + p[k];
            out[k] = i >= pos && i < n && (bad[k] || Mi[k] <= SS_MLO || Mi[k] >= SS_MHI);
        }
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(out[0] || out[1]);
        if (!mask) { acc = ss_from(key, __builtin_amdgcn_readlane(Mi[MCLF_ITEMS - 1], 63)); break; }
        const int fl = __ffsll((long long)mask) - 1;
        const int ek = __builtin_amdgcn_readlane(out[0] ? 1 : 0, fl) ? 0 : 1;
        const int j = fl * MCLF_ITEMS + ek;
        int Mb = M;
        if (j > pos) Mb = ek ? __builtin_amdgcn_readlane(Mi[0], fl) : __builtin_amdgcn_readlane(Mi[1], fl - 1);
        const double tj = mclf_readlane_f64(ek ? t[1] : t[0], fl);
        acc = ss_exact_step(ss_from(key, Mb), tj);
        pos = j + 1;
    }
    return acc;
}

// A risky sub-tile by its table (this lane's two rows e[0], e[1]; tkey = the binade the table was made for): the usual case is
// ONE step that leaves the binade upwards.  Returns through the generic replay whenever the sub-tile does something else.
__device__ __forceinline__ float mclf_replay_table(const mclf_tab_elem (&e)[MCLF_ITEMS], int tkey, int n, int head, float acc, int lane,
                                                   unsigned int* phases, unsigned int* by_table)
{
    double t[MCLF_ITEMS] = {e[0].t, e[1].t};
    const int key = __builtin_amdgcn_readfirstlane(ss_key(acc));
    tkey = __builtin_amdgcn_readfirstlane(tkey); n = __builtin_amdgcn_readfirstlane(n); head = __builtin_amdgcn_readfirstlane(head);
    if (key == 0 || key != tkey || head > 0) return mclf_replay(t, n, 0, head, acc, lane, phases);
    const int M = ss_mag(acc);
    int Mi[MCLF_ITEMS];
    bool out[MCLF_ITEMS];
#pragma unroll
    for (int k = 0; k < MCLF_ITEMS; ++k) {
        const int i = lane * MCLF_ITEMS + k;
        Mi[k] = M + (e[k].se >> 1);
        out[k] = i < n && ((e[k].se & 1) || Mi[k] <= SS_MLO || Mi[k] >= SS_MHI);
    }
    const unsigned long long mask = __builtin_amdgcn_ballot_w64(out[0] || out[1]);
    *by_table += 1;
    if (!mask) return ss_from(key, __builtin_amdgcn_readlane(Mi[MCLF_ITEMS - 1], 63));       // it stayed inside after all
    const int fl = __ffsll((long long)mask) - 1;
    const int ek = __builtin_amdgcn_readlane(out[0] ? 1 : 0, fl) ? 0 : 1;
    const int j = fl * MCLF_ITEMS + ek;
    int Mb = M;
    if (j > 0) Mb = ek ? __builtin_amdgcn_readlane(Mi[0], fl) : __builtin_amdgcn_readlane(Mi[1], fl - 1);
    const double tj = mclf_readlane_f64(ek ? t[1] : t[0], fl);
    const float acc1 = ss_exact_step(ss_from(key, Mb), tj);
    const int key1 = __builtin_amdgcn_readfirstlane(ss_key(acc1));
    if (key1 != key + 1 || (key & 0xff) >= 254) return mclf_replay(t, n, j + 1, 0, acc1, lane, phases);
    // the rest of the sub-tile in the next binade: prefix differences of the table's second column
    const int M1 = ss_mag(acc1);
    const int base1 = (ek ? __builtin_amdgcn_readlane(e[1].se1, fl) : __builtin_amdgcn_readlane(e[0].se1, fl)) >> 1;
    bool out1 = false;
    int last1 = 0;
#pragma unroll
    for (int k = 0; k < MCLF_ITEMS; ++k) {
        const int i = lane * MCLF_ITEMS + k;
        const int m = M1 + (e[k].se1 >> 1) - base1;
        out1 |= i > j && i < n && ((e[k].se1 & 1) || m <= SS_MLO || m >= SS_MHI);
        last1 = m;
    }
    if (__builtin_amdgcn_ballot_w64(out1)) return mclf_replay(t, n, j + 1, 0, acc1, lane, phases);
    return ss_from(key1, __builtin_amdgcn_readlane(last1, 63));
}

__device__ __forceinline__ int mclf_plain_key(int key) { return key == SS_ID ? SS_ID : (key & MCLF_KEY_MASK); }

// ---- LDS staging (per axis)
struct mclf_ent { int s, tslot, tkey, lo_n; ss_rec head; };     // a risky sub-tile: index, table slot or -1, the table's binade, first particle << 8 | particle count - 1; join of the records between the previous risky one (or its batch's start) and it
struct mclf_step { ss_rec gap; int s, tslot, tkey, lo_n; };     // what the chain reads per list entry, in sub-tile order: the gap in front of it, then the entry
struct mclf_stage {
    mclf_tab_elem* tab;           // [MCLF_TSLOTS][MCLF_SUB]
    ss_rec* comp;                 // [nbatch]: join of a batch's records; key 0 for a batch with risky records in it
    ss_rec* btail;                // [nbatch]: for such a batch, join of the records behind its last risky one
    mclf_step* step;              // [MCLF_MAXENT + 1]: entry k in sub-tile order with the gap in front of it; [nent]: the gap behind the last
    mclf_ent* ent;                // [MCLF_MAXENT], in arrival order
    int* order;                   // [MCLF_MAXENT]: entries by sub-tile index
    ss_wild* wmap;                // [MCLF_MAXENT], in arrival order: the wild map of an entry that has one (tslot = -2 - arrival index)
    int* nent;                    // entries handed out (may exceed MCLF_MAXENT: the surplus is not listed)
};
__device__ __forceinline__ size_t mclf_stage_bytes(int nbatch)
{
    return (size_t)MCLF_TSLOTS * MCLF_SUB * sizeof(mclf_tab_elem) + 2 * (size_t)nbatch * sizeof(ss_rec) + (MCLF_MAXENT + 1) * sizeof(mclf_step) +
           MCLF_MAXENT * sizeof(mclf_ent) + MCLF_MAXENT * sizeof(int) + MCLF_MAXENT * sizeof(ss_wild) + 16;
}
__device__ __forceinline__ mclf_stage mclf_stage_at(char* base, int nbatch)
{
    mclf_stage st;
    st.tab = (mclf_tab_elem*)base; base += (size_t)MCLF_TSLOTS * MCLF_SUB * sizeof(mclf_tab_elem);
    st.comp = (ss_rec*)base; base += (size_t)nbatch * sizeof(ss_rec);
    st.btail = (ss_rec*)base; base += (size_t)nbatch * sizeof(ss_rec);
    st.step = (mclf_step*)base; base += (MCLF_MAXENT + 1) * sizeof(mclf_step);
    st.ent = (mclf_ent*)base; base += MCLF_MAXENT * sizeof(mclf_ent);
    st.order = (int*)base; base += MCLF_MAXENT * sizeof(int);
    st.wmap = (ss_wild*)base; base += MCLF_MAXENT * sizeof(ss_wild);
    st.nent = (int*)base;
    return st;
}

// One batch of 64 records of an axis, by one wave: composite, and the list entries of its risky records.
// tbase: first staged slot of every rank's tables on this axis (a record names its table by its rank's own slot number)
__device__ __forceinline__ void mclf_stage_batch(const mcl_finish_args& f, const mclf_stage& st, ss_rec r, int b, int lane, const int* tbase, int axis, int* ntab_seen)
{
    {
        // The usual batch at large particle counts -- 64 plain records of ONE binade (a sum far from zero changes its binade
        // log2 N times in all): the join of the batch by the bare prefix arithmetic, D = a.D + b.D, lo = min(a.lo, a.D + b.lo),
        // hi = max(a.hi, a.D + b.hi) with the saturation of ss_rec_join -- a quarter of the general segmented scan's
        // instructions (at 1M particles the finisher's waves take sixteen batches each).
        const int k0 = __builtin_amdgcn_readfirstlane(r.key);
        const bool odd = r.key == SS_ID || (r.key & (MCLF_RISKY | MCLF_WILD)) != 0 || r.key != k0 || r.key == 0;
        if (__builtin_amdgcn_ballot_w64(odd) == 0ull) {
            int D = r.D, lo = r.lo, hi = r.hi;
#define MCLF_STEP(C, R) { const int oD = mclf_dpp<C, R>(0, D), olo = mclf_dpp<C, R>(SS_SAT, lo), ohi = mclf_dpp<C, R>(-SS_SAT, hi);   \
                          lo = min(olo, ss_sat_i(oD + lo)); hi = max(ohi, ss_sat_i(oD + hi)); D = ss_sat_i(oD + D); }
            MCLF_DPP_STEPS(MCLF_STEP)
#undef MCLF_STEP
            if (lane == 63) st.comp[b] = ss_rec_make(mclf_plain_key(k0), D, lo, hi);
            return;
        }
    }
    const bool risky = r.key != SS_ID && (r.key & MCLF_RISKY) != 0;
    int tslot = r.key == SS_ID ? -1 : ((r.key >> MCLF_TSLOT_SHIFT) & 0xff) - 1;
    if (tslot >= 0) { tslot += tbase[mclf_rank_of_sub(f, b * 64 + lane)]; atomicMax(ntab_seen, tslot + 1); }
    const int pkey = mclf_plain_key(r.key);
    const unsigned long long rmask = __builtin_amdgcn_ballot_w64(risky);
    ss_rec v = r;
    v.key = pkey;
    if (risky) v = ss_rec_identity();
    const int head = (lane == 0 || ((rmask >> (lane - 1)) & 1ull)) ? 1 : 0;        // a segment starts behind every risky record
    const ss_rec seg = mclf_scan_join_segmented(v, head);
    if (rmask == 0) {
        if (lane == 63) st.comp[b] = seg;
        return;
    }
    if (lane == 63) {
        st.comp[b] = ss_rec_make(0, 0, 0, 0);
        st.btail[b] = risky ? ss_rec_identity() : seg;
    }
    if (risky) {
        const int e = atomicAdd(st.nent, 1);
        if (e < MCLF_MAXENT) {
            mclf_ent en;
            int lo, hi;
            mclf_sub_range(f, b * 64 + lane, &lo, &hi);
            en.s = b * 64 + lane; en.tslot = tslot; en.tkey = pkey; en.lo_n = hi > lo ? ((lo << 8) | (hi - lo - 1)) : -1;
            en.head = seg;                                     // (its own record counts as the identity in the scan)
            if (tslot < 0 && (r.key & MCLF_WILD) && f.wild) {  // no table: the sub-tile's wild map comes along (one more round trip, per lane)
                st.wmap[e] = mclf_load_wild(f.wild + (size_t)axis * ((size_t)f.groups * (f.gthreads >> 6)) + b * 64 + lane);
                en.tslot = -2 - e;
            }
            st.ent[e] = en;
        }
    }
}

// join of comp[b0 .. b1] (inclusive; empty range: identity) by one wave; every lane returns the result
__device__ __forceinline__ ss_rec mclf_join_batches(const mclf_stage& st, int b0, int b1, int lane)
{
    ss_rec acc = ss_rec_identity();
    for (int b = b0; b <= b1; b += 64) {
        ss_rec r = (b + lane <= b1) ? st.comp[b + lane] : ss_rec_identity();
        r = mclf_scan_join(r);
        const ss_rec all = ss_rec_make(__builtin_amdgcn_readlane(r.key, 63), __builtin_amdgcn_readlane(r.D, 63),
                                       __builtin_amdgcn_readlane(r.lo, 63), __builtin_amdgcn_readlane(r.hi, 63));
        acc = ss_rec_join(acc, all);
    }
    return acc;
}

// Records [ra, rb) of an axis walked with the true accumulator straight from global memory, replaying whatever does not fit
// (the path for everything the staging did not foresee, and the whole chain when the scratch cannot hold the tables).
__device__ __forceinline__ float mclf_walk_plain(const mcl_finish_args& f, int axis, double S, int ra, int rb, float acc, int lane,
                                              unsigned int* replays, unsigned int* phases)
{
    int r0 = ra;
    while (r0 < rb) {
        ss_rec r = (r0 + lane < rb) ? mclf_load_rec(mclf_rec_ptr(f, axis, r0 + lane)) : ss_rec_identity();
        r.key = mclf_plain_key(r.key);
        const ss_rec pre = mclf_scan_join(r);
        const int key = __builtin_amdgcn_readfirstlane(ss_key(acc));
        const int M = ss_mag(acc);
        const bool fits = pre.key == SS_ID || (key != 0 && ss_rec_fits(pre, key, M));
        const unsigned long long nofit = __builtin_amdgcn_ballot_w64(!fits);
        const int nb = min(64, rb - r0);
        int fb = nofit ? __ffsll((long long)nofit) - 1 : 64;
        if (fb > nb) fb = nb;
        if (fb > 0) {
            const int pk = __builtin_amdgcn_readlane(pre.key, fb - 1), pD = __builtin_amdgcn_readlane(pre.D, fb - 1);
            if (pk != SS_ID) acc = ss_from(key, M + pD);
        }
        if (fb < nb) {
            const int s = r0 + fb;
            int lo, hi;
            mclf_sub_range(f, s, &lo, &hi);
            if (lo < hi) {
                double t[MCLF_ITEMS];
                mclf_load_terms(f, axis, S, lo, hi, lane, t);
                acc = mclf_replay(t, hi - lo, 0, 0, acc, lane, phases);
                *replays += 1;
            }
            r0 = s + 1;
        } else {
            r0 += nb;
        }
    }
    return acc;
}

// The walk for sums with MANY sub-tiles that do not go by a record (a sum that hovers around zero: a third of its sub-tiles
// cross binades): 64 records and their wild maps per round trip, the plain records between two risky ones joined once by a
// segmented scan, and then per risky record one gap (a check and an add) and one wild map (a check and a few integer
// operations), all from registers.  Whatever does not fit is replayed from the particle records as before.
__device__ __forceinline__ float mclf_walk(const mcl_finish_args& f, int axis, double S, int ra, int rb, float acc, int lane,
                                        unsigned int* replays, unsigned int* phases)
{
    if (!f.wild) return mclf_walk_plain(f, axis, S, ra, rb, acc, lane, replays, phases);
    const int nsub = f.groups * (f.gthreads >> 6);
    for (int r0 = ra; r0 < rb; r0 += 64) {
        const int nb = min(64, rb - r0);
        ss_rec r = ss_rec_identity();
        ss_wild w = ssw_invalid();
        if (lane < nb) {
            // record and map as four 16-byte loads through the L2 (sc1), issued together and waited for once: as eight relaxed
            // atomic loads they went out one behind the other -- eight round trips to memory per batch
            const ss_rec* rp = mclf_rec_ptr(f, axis, r0 + lane);
            const ss_wild* wp = f.wild + (size_t)axis * nsub + r0 + lane;
            int4 q0, q1, q2, q3;
            asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\t"
                         "global_load_dwordx4 %2, %5, off offset:16 sc1\n\tglobal_load_dwordx4 %3, %5, off offset:32 sc1\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3) : "v"(rp), "v"(wp) : "memory");
            bool current;
            r = mclf_decode_rec(q0, 0u, &current);
            w.key_in = q1.x; w.key_out = q1.y; w.q = q1.z; w.r = q1.w;
            w.a = (long long)(((unsigned long long)(unsigned int)q2.y << 32) | (unsigned int)q2.x); w.c = (long long)(((unsigned long long)(unsigned int)q2.w << 32) | (unsigned int)q2.z);
            w.L = (long long)(((unsigned long long)(unsigned int)q3.y << 32) | (unsigned int)q3.x); w.H = (long long)(((unsigned long long)(unsigned int)q3.w << 32) | (unsigned int)q3.z);
        }
        const bool risky = r.key != SS_ID && (r.key & MCLF_RISKY) != 0;
        const bool wild = risky && (r.key & MCLF_WILD) != 0;
        unsigned long long rmask = __builtin_amdgcn_ballot_w64(risky);
        const unsigned long long wmask = __builtin_amdgcn_ballot_w64(wild);
        ss_rec v = r;
        v.key = mclf_plain_key(r.key);
        if (risky) v = ss_rec_identity();
        const int head = (lane == 0 || ((rmask >> (lane - 1)) & 1ull)) ? 1 : 0;
        const ss_rec seg = mclf_scan_join_segmented(v, head);        // lane l: the plain records from the last risky one (exclusive) to l
        int prev = -1;                                               // last record of the batch taken care of
        // between two replays the accumulator travels as integers -- (key, M), wave-uniform: a gap is a compare and an add, a wild
        // map a compare and a few shifts; the float is only put together again for a replay and at the end of the batch
        int key = __builtin_amdgcn_readfirstlane(ss_key(acc));
        int M = __builtin_amdgcn_readfirstlane(ss_mag(acc));
        while (true) {
            const int next = rmask ? __ffsll((long long)rmask) - 1 : nb;
            if (next - 1 > prev) {                                   // the gap of plain records in front of `next`
                const ss_rec gap = ss_rec_make(__builtin_amdgcn_readlane(seg.key, next - 1), __builtin_amdgcn_readlane(seg.D, next - 1),
                                               __builtin_amdgcn_readlane(seg.lo, next - 1), __builtin_amdgcn_readlane(seg.hi, next - 1));
                if (gap.key != SS_ID) {
                    if (key != 0 && ss_rec_fits(gap, key, M)) M += gap.D;
                    else {
                        if (key != 0) acc = ss_from_bits(key, M);
                        acc = mclf_walk_plain(f, axis, S, r0 + prev + 1, r0 + next, acc, lane, replays, phases);
                        key = __builtin_amdgcn_readfirstlane(ss_key(acc)); M = __builtin_amdgcn_readfirstlane(ss_mag(acc));
                    }
                }
            }
            if (next >= nb) break;
            const int s = r0 + next;
            bool done = false;
            if ((wmask >> next) & 1ull) {
                const ss_wild wn = mclf_readlane_wild(w, next);
                const long long m = ssw_signed(key, M);
                if (key != 0 && ssw_fits(wn, key, m)) {
                    const long long mo = ssw_apply(wn, m);
                    key = wn.key_out; M = (int)(mo < 0 ? -mo : mo);
                    done = true;
                }
            }
            if (!done) {
                int lo, hi;
                mclf_sub_range(f, s, &lo, &hi);
                if (lo < hi) {
                    double t[MCLF_ITEMS];
                    mclf_load_terms(f, axis, S, lo, hi, lane, t);
                    // a wild sub-tile whose map does not fit crosses binades every few terms: stepping its 128 terms one by one
                    // (2.5 us) beats a phase per crossing (0.7 us each, dozens of them)
                    const bool was_wild = ((wmask >> next) & 1ull) != 0ull;
                    if (key != 0) acc = ss_from_bits(key, M);
                    acc = mclf_replay(t, hi - lo, 0, was_wild ? hi - lo : 0, acc, lane, phases);
                    key = __builtin_amdgcn_readfirstlane(ss_key(acc)); M = __builtin_amdgcn_readfirstlane(ss_mag(acc));
                    *replays += 1;
                }
            }
            prev = next;
            rmask &= rmask - 1ull;
        }
        if (key != 0) acc = ss_from_bits(key, M);
    }
    return acc;
}

// ---- the walk with composition trees ---------------------------------------------------------------------------------------
// A sum that hovers around zero hands the chain a few hundred sub-tiles that do not go by a record (mclf_walk).  One wave taking
// them one after the other -- a gap, a map, a gap, a map: 0.6 us each -- was 170 of the estimate's 200 us at 100 000 particles
// around the origin.  But every record is a map of the signed magnitude (a plain record m -> m +- D on the inputs it fits,
// ssw_from_rec; a wild sub-tile its wild map), maps compose (ssw_join), and the true accumulator fits a composite exactly when it
// fits every part in turn.  So while the chain's wave walks, the finisher's other waves -- idle until now -- compose: each takes
// a batch of 64 records, joins them pairwise up a binary tree (127 nodes: node (k, j) = records j 2^k .. (j + 1) 2^k - 1 of the
// batch) and leaves the tree in LDS; the chain takes a whole batch with ONE check of its root, and where the root does not fit
// (13 of 773 sub-tiles at 100k, 92 of 7 800 at 1M: the sum passing through zero, where no prediction of its binades holds) walks
// down to the record that does not, steps it, and climbs again -- a dozen checks instead of a scan.  A wrong map costs time,
// never correctness: nothing is applied without its check.
#define MCLF_TREE_NODES 128                    // 127 in use: level k at [128 - (128 >> k), ...)
#define MCLF_TREE_RING 6                       // trees in flight per axis (they live in the axis' table area: 20 tables = 40 KB)
#define MCLF_WILD_ID 0x7ffffff0                // key_in of a node that composes as the identity (no record below it)
struct mclf_trees { ss_wild* nodes; volatile int* ready; volatile int* consumed; volatile int* abort; };
__device__ __forceinline__ mclf_trees mclf_trees_at(const mclf_stage& st)
{
    mclf_trees t;
    t.nodes = (ss_wild*)st.tab;
    int* w = (int*)((char*)st.tab + (size_t)MCLF_TREE_RING * MCLF_TREE_NODES * sizeof(ss_wild));
    t.ready = w; t.consumed = w + MCLF_TREE_RING; t.abort = w + MCLF_TREE_RING + 1;
    return t;
}
static_assert((size_t)MCLF_TREE_RING * MCLF_TREE_NODES * sizeof(ss_wild) + (MCLF_TREE_RING + 2) * sizeof(int) <= (size_t)MCLF_TSLOTS * MCLF_SUB * sizeof(mclf_tab_elem),
              "the trees live in the table area of their axis");
__device__ __forceinline__ int mclf_tree_off(int k) { return MCLF_TREE_NODES - (MCLF_TREE_NODES >> k); }

__device__ __forceinline__ long long mclf_shfl_down_i64(long long v, int off)
{
    const int lo = __shfl_down((int)(unsigned int)(unsigned long long)v, off, 64), hi = __shfl_down((int)(unsigned int)((unsigned long long)v >> 32), off, 64);
    return (long long)(((unsigned long long)(unsigned int)hi << 32) | (unsigned long long)(unsigned int)lo);
}
__device__ __forceinline__ ss_wild mclf_shfl_down_wild(const ss_wild& w, int off)
{
    ss_wild o;
    o.key_in = __shfl_down(w.key_in, off, 64); o.key_out = __shfl_down(w.key_out, off, 64); o.q = __shfl_down(w.q, off, 64); o.r = __shfl_down(w.r, off, 64);
    o.a = mclf_shfl_down_i64(w.a, off); o.c = mclf_shfl_down_i64(w.c, off); o.L = mclf_shfl_down_i64(w.L, off); o.H = mclf_shfl_down_i64(w.H, off);
    return o;
}

// record r0 + lane of an axis and its wild map (lanes >= nb: the empty record), as four 16-byte loads through the L2 waited for once
__device__ __forceinline__ void mclf_load_rec_and_map(const mcl_finish_args& f, int axis, int r0, int nb, int lane, ss_rec* r, ss_wild* w)
{
    const int nsub = f.groups * (f.gthreads >> 6);
    *r = ss_rec_identity();
    *w = ssw_invalid();
    if (lane < nb) {
        const ss_rec* rp = mclf_rec_ptr(f, axis, r0 + lane);
        const ss_wild* wp = f.wild + (size_t)axis * nsub + r0 + lane;
        int4 q0, q1, q2, q3;
        asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\t"
                     "global_load_dwordx4 %2, %5, off offset:16 sc1\n\tglobal_load_dwordx4 %3, %5, off offset:32 sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3) : "v"(rp), "v"(wp) : "memory");
        bool current;
        *r = mclf_decode_rec(q0, 0u, &current);
        w->key_in = q1.x; w->key_out = q1.y; w->q = q1.z; w->r = q1.w;
        w->a = (long long)(((unsigned long long)(unsigned int)q2.y << 32) | (unsigned int)q2.x); w->c = (long long)(((unsigned long long)(unsigned int)q2.w << 32) | (unsigned int)q2.z);
        w->L = (long long)(((unsigned long long)(unsigned int)q3.y << 32) | (unsigned int)q3.x); w->H = (long long)(((unsigned long long)(unsigned int)q3.w << 32) | (unsigned int)q3.z);
    }
}

// A helper wave (index hidx of nh) of an axis: the trees of the walk's batches hidx, hidx + nh, ... (records [ra, rb) in batches of
// 64 from ra on), each into ring slot batch % MCLF_TREE_RING once the chain is through with the batch that had it.
__device__ __forceinline__ void mclf_tree_helper(const mcl_finish_args& f, const mclf_trees& tr, int axis, int ra, int rb, int hidx, int nh, int lane)
{
    const int nbat = (rb - ra + 63) >> 6;
    for (int bi = hidx; bi < nbat; bi += nh) {
        unsigned int spins = 0;
        while (*tr.consumed < bi - MCLF_TREE_RING + 1 && *tr.abort == 0) {
            if (++spins > MCLF_SPIN_LIMIT) return;
            __builtin_amdgcn_s_sleep(2);
        }
        if (*tr.abort != 0) return;
        const int r0 = ra + 64 * bi, nb = min(64, rb - r0);
        ss_rec r;
        ss_wild w;
        mclf_load_rec_and_map(f, axis, r0, nb, lane, &r, &w);
        const bool risky = r.key != SS_ID && (r.key & MCLF_RISKY) != 0;
        const bool wild = risky && (r.key & MCLF_WILD) != 0;
        ss_rec pr = r;
        pr.key = mclf_plain_key(r.key);
        ss_wild cur = ssw_invalid();
        bool cur_id = lane >= nb || r.key == SS_ID;
        if (!cur_id) cur = risky ? (wild ? w : ssw_invalid()) : ssw_from_rec(pr);
        ss_wild* slot = tr.nodes + (size_t)(bi % MCLF_TREE_RING) * MCLF_TREE_NODES;
        { ss_wild out = cur; if (cur_id) out.key_in = MCLF_WILD_ID; slot[lane] = out; }
#pragma unroll
        for (int k = 1; k <= 6; ++k) {
            const int h = 1 << (k - 1);
            const ss_wild o = mclf_shfl_down_wild(cur, h);
            const bool o_id = __shfl_down((int)cur_id, h, 64) != 0;
            if ((lane & ((1 << k) - 1)) == 0) {
                if (!o_id) {
                    if (cur_id) { cur = o; cur_id = false; }
                    else cur = ssw_join(cur, o);
                }
                ss_wild out = cur;
                if (cur_id) out.key_in = MCLF_WILD_ID;
                slot[mclf_tree_off(k) + (lane >> k)] = out;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // the tree stands in LDS before its flag does
        if (lane == 0) tr.ready[bi % MCLF_TREE_RING] = bi + 1;
    }
}

// mclf_walk's job by the trees the helper waves leave (records [ra, rb), the chain's wave, acc wave-uniform)
__device__ __forceinline__ float mclf_walk_trees(const mcl_finish_args& f, const mclf_trees& tr, int axis, double S, int ra, int rb, float acc, int lane,
                                                 unsigned int* replays, unsigned int* phases, double* lds_terms)
{
    const int nbat = (rb - ra + 63) >> 6;
    for (int bi = 0; bi < nbat; ++bi) {
        const int r0 = ra + 64 * bi, nb = min(64, rb - r0);
        unsigned int spins = 0;
        while (tr.ready[bi % MCLF_TREE_RING] != bi + 1) {
            if (++spins > MCLF_SPIN_LIMIT) {                        // (never, unless a helper died: the rest the old way)
                if (lane == 0) { *tr.abort = 1; atomicAdd(&f.state->wait_timeouts, 1u); }
                return mclf_walk(f, axis, S, r0, rb, acc, lane, replays, phases);
            }
            __builtin_amdgcn_s_sleep(1);
        }
        const ss_wild* slot = tr.nodes + (size_t)(bi % MCLF_TREE_RING) * MCLF_TREE_NODES;
        int key = __builtin_amdgcn_readfirstlane(ss_key(acc));
        int M = __builtin_amdgcn_readfirstlane(ss_mag(acc));
        int pos = 0;
        while (pos < nb) {
            // the nodes that start at record pos: (k, pos >> k) for k up to the number of trailing zeros of pos (the whole batch at
            // pos = 0) -- lane k tests node k, all in one LDS round trip; the largest that fits carries the accumulator 2^k records on
            const int kmax = pos ? min(__ffs(pos) - 1, 6) : 6;
            const bool have = lane <= kmax;
            ss_wild nd = ssw_invalid();
            if (have) nd = slot[mclf_tree_off(lane) + (pos >> lane)];
            const long long m = ssw_signed(key, M);
            const bool okl = have && (nd.key_in == MCLF_WILD_ID || (key != 0 && ssw_fits(nd, key, m)));
            const unsigned long long okm = __builtin_amdgcn_ballot_w64(okl);
            if (okm) {
                const int kb = 63 - __clzll((long long)okm);
                const ss_wild wn = mclf_readlane_wild(nd, kb);
                if (wn.key_in != MCLF_WILD_ID) {
                    const long long mo = ssw_apply(wn, m);
                    key = wn.key_out; M = (int)(mo < 0 ? -mo : mo);
                }
                pos += 1 << kb;
                continue;
            }
            // the record itself does not go by its map: stepped from the particle records
            {
                const int lk_in = __builtin_amdgcn_readlane(nd.key_in, 0), lk_out = __builtin_amdgcn_readlane(nd.key_out, 0);
                const int lq = __builtin_amdgcn_readlane(nd.q, 0), lr = __builtin_amdgcn_readlane(nd.r, 0);
                const int s = r0 + pos;
                int lo, hi;
                mclf_sub_range(f, s, &lo, &hi);
                if (lo < hi) {
                    double t[MCLF_ITEMS];
                    mclf_load_terms(f, axis, S, lo, hi, lane, t);
                    // (a wild sub-tile whose map does not fit crosses binades every few terms: its 128 terms one by one beat a phase
                    // per crossing; a plain one: the in-binade phases)
                    const bool was_wild = lk_in != lk_out || lq != 0 || lr != 0 || lk_in == 0;
                    if (key != 0) acc = ss_from_bits(key, M);
                    acc = mclf_replay(t, hi - lo, 0, was_wild ? hi - lo : 0, acc, lane, phases, lds_terms);
                    key = __builtin_amdgcn_readfirstlane(ss_key(acc)); M = __builtin_amdgcn_readfirstlane(ss_mag(acc));
                    *replays += 1;
                }
                pos += 1;
            }
        }
        if (key != 0) acc = ss_from_bits(key, M);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) *tr.consumed = bi + 1;
    }
    return acc;
}

// One axis' float accumulator over all particles, by one wave (all 64 lanes; the result is wave-uniform).
// `first`: the accumulator behind the first MCLF_PRE_SUBS sub-tiles (mclf_pose).
__device__ __forceinline__ float mclf_chain(const mcl_finish_args& f, const mclf_stage* st, int ntab, int axis, double S, float first, int lane,
                                            unsigned int* stats, bool trees = false, double* lds_terms = nullptr)
{
    const int nrec = f.groups * (f.gthreads >> 6);
    // (a list that overflowed is no list: the gaps between listed entries would skip the unlisted risky records)
    const int done = min(MCLF_PRE_SUBS, nrec);
    if (st && trees) { const mclf_trees tr = mclf_trees_at(*st); return mclf_walk_trees(f, tr, axis, S, done, nrec, first, lane, &stats[0], &stats[1], lds_terms); }
    if (!st || *st->nent > MCLF_MAXENT) return mclf_walk(f, axis, S, done, nrec, first, lane, &stats[0], &stats[1]);
    const int nent = __builtin_amdgcn_readfirstlane(*st->nent);
    float acc = first;                                           // the first sub-tiles are done (mclf_pose)
    int prev = done - 1;                                         // last sub-tile done
    for (int k = 0; k <= nent; ++k) {
#ifdef MCLF_STAMPS
        if (axis == 0 && k < 16 && lane == 0) f.state->cstamps[k] = MCLF_NOW();
#endif
        mclf_step sp = st->step[k];                              // wave-uniform LDS reads
        sp.gap.key = __builtin_amdgcn_readfirstlane(sp.gap.key); sp.gap.D = __builtin_amdgcn_readfirstlane(sp.gap.D);
        sp.gap.lo = __builtin_amdgcn_readfirstlane(sp.gap.lo); sp.gap.hi = __builtin_amdgcn_readfirstlane(sp.gap.hi);
        sp.s = __builtin_amdgcn_readfirstlane(sp.s); sp.tslot = __builtin_amdgcn_readfirstlane(sp.tslot);
        sp.tkey = __builtin_amdgcn_readfirstlane(sp.tkey); sp.lo_n = __builtin_amdgcn_readfirstlane(sp.lo_n);
        // the gap in front of entry k (behind the last one for k == nent)
        if (sp.gap.key != SS_ID) {
            const int key = __builtin_amdgcn_readfirstlane(ss_key(acc));
            const int M = ss_mag(acc);
            if (key != 0 && ss_rec_fits(sp.gap, key, M)) acc = ss_from(key, M + sp.gap.D);
            else { acc = mclf_walk(f, axis, S, prev + 1, sp.s, acc, lane, &stats[0], &stats[1]); stats[3] += 1; }
        }
        if (k == nent) break;
        // the risky sub-tile itself
        if (sp.lo_n >= 0) {
            const int lo = sp.lo_n >> 8, n = (sp.lo_n & 0xff) + 1;
            bool by_map = false;
            if (sp.tslot <= -2) {                                    // a wild sub-tile: its map, if the accumulator fits
                ss_wild wn = st->wmap[-2 - sp.tslot];                // (wave-uniform LDS read)
                wn = mclf_readlane_wild(wn, 0);
                const int key = __builtin_amdgcn_readfirstlane(ss_key(acc));
                const long long m = ssw_signed(key, ss_mag(acc));
                if (key != 0 && ssw_fits(wn, key, m)) {
                    const long long mo = ssw_apply(wn, m);
                    acc = ss_from(wn.key_out, (int)(mo < 0 ? -mo : mo));
                    stats[2] += 1;
                    by_map = true;
                }
            }
            if (by_map) { }
            else if (sp.tslot >= 0 && sp.tslot < ntab) {
                const mclf_tab_elem* row = st->tab + sp.tslot * MCLF_SUB + 2 * lane;
                mclf_tab_elem el[MCLF_ITEMS] = {row[0], row[1]};
                acc = mclf_replay_table(el, sp.tkey, n, 0, acc, lane, &stats[1], &stats[2]);
            } else {
                double t[MCLF_ITEMS];
                mclf_load_terms(f, axis, S, lo, lo + n, lane, t);
                acc = mclf_replay(t, n, 0, sp.tslot <= -2 ? n : 0, acc, lane, &stats[1], lds_terms);      // (a wild one that does not fit: stepped)
                stats[0] += 1;
            }
        }
        prev = sp.s;
    }
    return acc;
}

// The block sums in the fixed order of a 256-thread workgroup (thread-strided with a stride of 256, wave shuffles, waves in order)
// into sm.red; the caller puts a barrier behind it and adds the four rows up (mclf_block_totals).
__device__ __forceinline__ void mclf_reduce_partials(const mcl_finish_args& f, mclf_smem& sm)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < MCLF_POSE_THREADS) {
        double v[5] = {0, 0, 0, 0, 0};
        // (four blocks per thread requested together -- a trip to memory each otherwise; the additions keep their order)
        for (int b0 = tid; b0 < f.nblocks; b0 += 4 * MCLF_POSE_THREADS) {
            double p[4][5];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int b = b0 + q * MCLF_POSE_THREADS;
                const size_t at = (size_t)(b < f.nblocks ? b : 0) * 5;
#pragma unroll
                for (int k = 0; k < 5; ++k) p[q][k] = f.partials[at + k];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (b0 + q * MCLF_POSE_THREADS < f.nblocks)
                    for (int k = 0; k < 5; ++k) v[k] += p[q][k];
        }
        for (int k = 0; k < 5; ++k) v[k] = mclf_wave_sum(v[k]);
        if (lane == 0) for (int k = 0; k < 5; ++k) sm.red[wave][k] = v[k];
    }
}
__device__ __forceinline__ void mclf_block_totals(const mclf_smem& sm, double (&tot)[5])
{
    for (int k = 0; k < 5; ++k) tot[k] = 0.0;
    for (int w = 0; w < MCLF_POSE_THREADS / 64; ++w) for (int k = 0; k < 5; ++k) tot[k] += sm.red[w][k];
}
// theta of the estimate (final: the finisher forms it from the same sums in the same order) and the weighted means of x, y in
// double -- NOT the reference's x, y (those are the serially rounded float sums), but within a few 1e-6 of them at 100k particles
__device__ __forceinline__ bl_pose_xyt_t mclf_approx_pose(const double (&tot)[5], int64_t utime)
{
    bl_pose_xyt_t p;
    p.utime = utime;
    p.x = (float)(tot[1] / tot[0]);
    p.y = (float)(tot[2] / tot[0]);
    p.theta = (float)atan2(tot[3], tot[4]);
    return p;
}
// one thread: wait for the finisher's x, y (mclf_pose with publish), clear the mailbox for the next launch
// Returns false when the wait gave up (x, y untouched: the caller keeps its provisional pose; the count reaches the host with the
// next bl_pf_pose_estimate, which reports it once and clears it).
__device__ __forceinline__ bool mclf_wait_pose(const mcl_finish_args& f, float* x, float* y)
{
    unsigned long long wx, wy;
    unsigned int spins = 0;
    bool ok = true;
    while (true) {                                   // (both words requested together; each says by itself that it is there)
        wx = mclf_load_u64(f.sync + 3); wy = mclf_load_u64(f.sync + 4);
        if (((wx & wy) >> 32) & 1ull) break;
        if (++spins > MCLF_SPIN_LIMIT) { atomicAdd(&f.state->wait_timeouts, 1u); ok = false; break; }    // (every wait of the launch has an end)
        __builtin_amdgcn_s_sleep(1);
    }
    if (ok) {
        *x = __uint_as_float((unsigned int)wx);
        *y = __uint_as_float((unsigned int)wy);
    }
    mclf_store_u64(f.sync + 4, 0ull);
    mclf_store_u64(f.sync + 3, 0ull);
    return ok;
}

// The pre-chain workgroup: the float sums over the first MCLF_PRE_SUBS sub-tiles.  The sums start from zero, so nothing is
// needed from the groups there, and that is where the accumulator changes its binade every few terms.  Waves 0 / 1 (x / y) step
// the first MCLF_PRE_STEPPED sub-tiles term by term (terms through LDS, one wave-uniform loop of three dependent operations per
// term); meanwhile every other wave of the workgroup forms the WILD MAP of one (axis, sub-tile) behind them from the double
// prefix sums of the terms (bl_serial_sum.h: so early in a sum the float accumulator and its double prediction differ by a few
// ulps, and a step's validity interval is as wide as a term).  Waves 0 / 1 then take a sub-tile by its map -- a check and a few
// integer operations -- and replay it the in-binade way when the map does not fit (each replay phase is 0.7 us of one wave:
// seven sub-tiles were 11 us that the finisher's chains waited for).  The result goes to the finisher through f.sync[1..2].
// Called by every thread of the workgroup (barriers).
static_assert(MCLF_PRE_SUBS - MCLF_PRE_STEPPED <= (MCLF_WG / 64 - 2) / 2, "one wave per (axis, mapped sub-tile) of the pre-chain");
__device__ __forceinline__ void mclf_pre_chain(const mcl_finish_args& f, mclf_smem& sm)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nrec = f.groups * (f.gthreads >> 6);
    const int npre = min(MCLF_PRE_SUBS, nrec);
    const bool stepper = wave < 2;
    const int axis = stepper ? wave : ((wave - 2) & 1);
    const int bq = MCLF_PRE_STEPPED + ((wave - 2) >> 1);                // a builder's sub-tile
    const bool builder = !stepper && bq < npre;
    // the particles are requested before the block sums are reduced
    float4 pre_r[MCLF_PRE_SUBS][MCLF_ITEMS];
    int cnt[MCLF_PRE_SUBS];
    if (stepper) {
        int at[MCLF_PRE_SUBS][MCLF_ITEMS];
#pragma unroll
        for (int q = 0; q < MCLF_PRE_SUBS; ++q) {
            int lo = 0, hi = 0;
            if (q < npre) mclf_sub_range(f, q, &lo, &hi);
            cnt[q] = hi - lo;
#pragma unroll
            for (int k = 0; k < MCLF_ITEMS; ++k) { const int i = lo + lane * MCLF_ITEMS + k; at[q][k] = i < hi ? i : 0; }
        }
        // (the loads stand alone, without a branch between them: they go out together; the count decides what is used)
        if (!f.sh) {
#pragma unroll
            for (int q = 0; q < MCLF_PRE_SUBS; ++q)
#pragma unroll
                for (int k = 0; k < MCLF_ITEMS; ++k) { const mclf_f4 v = *(mclf_gptr4)(f.rec + at[q][k]); pre_r[q][k] = make_float4(v.x, v.y, v.z, v.w); }
        } else {
#pragma unroll
            for (int q = 0; q < MCLF_PRE_SUBS; ++q)
#pragma unroll
                for (int k = 0; k < MCLF_ITEMS; ++k) pre_r[q][k] = mclf_particle(f, at[q][k]);
        }
    }
    float4 br[MCLF_ITEMS] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
    int bcnt = 0;
    if (builder) {
        int lo, hi;
        mclf_sub_range(f, bq, &lo, &hi);
        bcnt = max(0, min(MCLF_ITEMS, hi - lo - lane * MCLF_ITEMS));
#pragma unroll
        for (int k = 0; k < MCLF_ITEMS; ++k) {
            const int i = k < bcnt ? lo + lane * MCLF_ITEMS + k : 0;
            if (!f.sh) { const mclf_f4 v = *(mclf_gptr4)(f.rec + i); br[k] = make_float4(v.x, v.y, v.z, v.w); }
            else br[k] = mclf_particle(f, i);
        }
    }
    if (tid < MCLF_POSE_THREADS) {
        double su = 0.0;
        for (int b0 = tid; b0 < f.nblocks; b0 += 4 * MCLF_POSE_THREADS) {
            double p[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int b = b0 + q * MCLF_POSE_THREADS; p[q] = f.partials[(size_t)(b < f.nblocks ? b : 0) * 5]; }
#pragma unroll
            for (int q = 0; q < 4; ++q) if (b0 + q * MCLF_POSE_THREADS < f.nblocks) su += p[q];
        }
        su = mclf_wave_sum_all(su);
        if (lane == 0) sm.red[wave][0] = su;
    }
#ifdef MCLF_STAMPS
    if (tid == 0) f.state->xstamps[8] = MCLF_NOW();
#endif
    __syncthreads();
#ifdef MCLF_STAMPS
    if (tid == 0) f.state->xstamps[9] = MCLF_NOW();
#endif
    double S = 0.0;
    for (int w = 0; w < MCLF_POSE_THREADS / 64; ++w) S += sm.red[w][0];          // exact integer below 2^53, any order gives it
    S = mclf_term_S(f, S);                                                       // (S feeds the terms only from here on)
    // ---- the terms; their double sums per sub-tile (the builders' predictions)
    double bt[MCLF_ITEMS] = {0.0, 0.0};
    double* pre = sm.pre[axis];
    int nstep = 0;
    if (stepper) {
#pragma unroll
        for (int q = 0; q < MCLF_PRE_STEPPED; ++q) {
            double qs = 0.0;
            if (q < npre) {
#pragma unroll
                for (int k = 0; k < MCLF_ITEMS; ++k) {
                    const double t = (lane * MCLF_ITEMS + k < cnt[q]) ? mclf_term(pre_r[q][k], S, axis) : 0.0;
                    pre[q * MCLF_SUB + 2 * lane + k] = t;
                    qs += t;
                }
                nstep = q * MCLF_SUB + cnt[q];                // (a short sub-tile can only be the last one)
            }
            qs = mclf_wave_sum_all(qs);
            if (lane == 0) sm.psum[axis][q] = qs;
        }
    } else if (builder) {
        double qs = 0.0;
#pragma unroll
        for (int k = 0; k < MCLF_ITEMS; ++k) { bt[k] = k < bcnt ? mclf_term(br[k], S, axis) : 0.0; qs += bt[k]; }
        qs = mclf_wave_sum_all(qs);
        if (lane == 0) sm.psum[axis][bq] = qs;
    }
#ifdef MCLF_STAMPS
    if (tid == 0) f.state->xstamps[10] = MCLF_NOW();
#endif
    __syncthreads();
#ifdef MCLF_STAMPS
    if (tid == 0) f.state->xstamps[5] = MCLF_NOW();
#endif
    unsigned int ph = 0;
    float v = 0.0f;
    if (stepper) {
        nstep = __builtin_amdgcn_readfirstlane(nstep);
        int i = 0;
        for (; i + 8 <= nstep; i += 8) {
            double w8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) w8[u] = pre[i + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) v = ss_exact_step(v, w8[u]);
        }
        for (; i < nstep; ++i) v = ss_exact_step(v, pre[i]);
    } else if (builder) {
        double start = 0.0;
        for (int q = 0; q < bq; ++q) start += sm.psum[axis][q];
        const int key = __builtin_amdgcn_readfirstlane(ss_key((float)start));
        ss_wild m = ssw_invalid();
        if (key != 0 && __builtin_amdgcn_ballot_w64(bcnt > 0) != 0ull) m = mclf_wild_map(bt, bcnt, start, key, lane);
        if (lane == 0) sm.pmap[axis][bq] = m;
    }
#ifdef MCLF_STAMPS
    if (tid == 0) f.state->xstamps[6] = MCLF_NOW();
#endif
    __syncthreads();
#ifdef MCLF_STAMPS
    if (tid == 0) f.state->xstamps[7] = MCLF_NOW();
#endif
    if (!stepper) return;
    unsigned int by_map = 0, replayed = 0;
#pragma unroll
    for (int q = MCLF_PRE_STEPPED; q < MCLF_PRE_SUBS; ++q) {
        if (q < npre && cnt[q] > 0) {
            ss_wild m = sm.pmap[axis][q];                                  // (wave-uniform LDS read)
            m = mclf_readlane_wild(m, 0);
            const int key = __builtin_amdgcn_readfirstlane(ss_key(v));
            const long long mag = ssw_signed(key, ss_mag(v));
            if (key != 0 && ssw_fits(m, key, mag)) {
                const long long mo = ssw_apply(m, mag);
                v = ss_from_bits(m.key_out, (int)(mo < 0 ? -mo : mo));
                by_map++;
            } else {
                double t[MCLF_ITEMS];
#pragma unroll
                for (int k = 0; k < MCLF_ITEMS; ++k) t[k] = (lane * MCLF_ITEMS + k < cnt[q]) ? mclf_term(pre_r[q][k], S, axis) : 0.0;
                v = mclf_replay(t, cnt[q], 0, 0, v, lane, &ph);
                replayed++;
            }
        }
    }
    if (lane == 0) { sm.first[wave] = v; f.state->pre_stats[wave] += (by_map << 16) + replayed; }
    __syncthreads();                                           // (waves 0 and 1 only: the others have left)
    if (tid == 0) {
        mclf_store_u64(f.sync + 1, (unsigned long long)__float_as_uint(sm.first[0]) | (1ull << 32));
        mclf_store_u64(f.sync + 2, (unsigned long long)__float_as_uint(sm.first[1]) | (1ull << 32));
#ifdef MCLF_STAMPS
        f.state->xstamps[3] = MCLF_NOW();
#endif
    }
}

// The finisher: waits for the groups of this launch, forms estimatePosteriorPose.  Called by EVERY thread of a workgroup of
// MCLF_WG threads (it contains barriers).  scratch / scratch_bytes: LDS the finisher may use until it returns (16-byte aligned).
// publish: x and y also go to f.sync[3..4] for another workgroup of the same launch (mclf_wait_pose).
__device__ __forceinline__ void mclf_pose(const mcl_finish_args& f, mclf_smem& sm, char* scratch, size_t scratch_bytes,
                                          bool publish = false)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef MCLF_STAMPS
    unsigned long long stamp[6] = {MCLF_NOW(), 0, 0, 0, 0, 0};
#define MCLF_STAMP(i) stamp[i] = MCLF_NOW()
#else
#define MCLF_STAMP(i) do { } while (0)
#endif
    const int nrec = f.groups * (f.gthreads >> 6), nbatch = (nrec + 63) >> 6;
    mclf_reduce_partials(f, sm);
    const size_t per_axis = (mclf_stage_bytes(nbatch) + 15) & ~(size_t)15;
    const bool staged = scratch != nullptr && 2 * per_axis <= scratch_bytes;
#define MCLF_STAGE(axis) mclf_stage_at(scratch + (size_t)(axis) * per_axis, nbatch)
    if (staged && tid < 2) *MCLF_STAGE(tid).nent = 0;
    if (tid == 0) {
        // tables: rank r's slot q of an axis is staged as slot tbase[axis][r] + q -- while it is below MCLF_TSLOTS.  A composed
        // finish knows every rank's count (its groups ran in an earlier launch, their blocks arrived by the all-gather behind it);
        // with the groups in this launch the records say which slots are in use (ntab_seen, below)
        const int world = mclf_world(f);
        for (int axis = 0; axis < 2; ++axis) {
            int run = 0;
            for (int r = 0; r < world; ++r) { sm.tbase[axis][r] = run; run += f.sh ? mclf_tab_count(f, 0ull, r, axis) : MCLF_TSLOTS; }
            sm.tbase[axis][world] = run;
            sm.ntab_seen[axis] = 0;
        }
    }
    __syncthreads();
    double S = 0.0;
    for (int w = 0; w < MCLF_POSE_THREADS / 64; ++w) S += sm.red[w][0];          // exact integer below 2^53, any order gives it
    S = mclf_term_S(f, S);                                                       // (S feeds the terms only from here on)
    MCLF_STAMP(1);
    const int sh_world = mclf_world(f);
    {
        // ---- stage a: every wave takes batches of 64 records of both axes, four per round, as 16-byte loads through the L2 (sc1)
        // that are waited for once (relaxed atomic 8-byte loads behind a branch per item became flat loads with a wait each).
        // The groups of this launch are still at work when the finisher gets here: a slot that does not show the launch's tag yet
        // is read again -- the load that finds the last record of a batch IS its staging load.  (A composed finish takes what the
        // all-gather brought.)  Lanes without a record read the first one.
        const bool poll = f.groups_wait > 0;
        const int nb2 = 2 * nbatch;
        const void* const dummy = (const void*)mclf_rec_ptr(f, 0, 0);
        for (int round = 0; round * 4 * MCLF_MAXW < nb2; ++round) {
            const void* pr[4];
            bool mine[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int it = wave + (4 * round + u) * MCLF_MAXW;
                pr[u] = dummy; mine[u] = false;
                if (it < nb2) {
                    const int axis = it >= nbatch ? 1 : 0, b = it - axis * nbatch;
                    if (b * 64 + lane < nrec) { pr[u] = (const void*)mclf_rec_ptr(f, axis, b * 64 + lane); mine[u] = true; }
                }
            }
            ss_rec rr[4];
            unsigned int spins = 0;
            while (true) {
                int4 q[4];
                asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\t"
                             "global_load_dwordx4 %2, %6, off sc1\n\tglobal_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
                             : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3]) : "v"(pr[0]), "v"(pr[1]), "v"(pr[2]), "v"(pr[3]) : "memory");
                bool late = false;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    bool current;
                    rr[u] = mclf_decode_rec(q[u], f.tag, &current);
                    if (!mine[u]) rr[u] = ss_rec_identity();
                    late |= mine[u] && poll && !current;
                }
                if (__builtin_amdgcn_ballot_w64(late) == 0ull) break;
                if (++spins > MCLF_SPIN_LIMIT) { if (lane == 0) atomicAdd(&f.state->wait_timeouts, 1u); break; }
                __builtin_amdgcn_s_sleep(1);
            }
#ifdef MCLF_STAMPS
            if (tid == 0 && round == 0) f.state->xstamps[1] = MCLF_NOW();
#endif
            if (staged) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int it = wave + (4 * round + u) * MCLF_MAXW;
                    if (it < nb2) {
                        const int axis = it >= nbatch ? 1 : 0, b = it - axis * nbatch;
                        mclf_stage_batch(f, MCLF_STAGE(axis), rr[u], b, lane, sm.tbase[axis], axis, &sm.ntab_seen[axis]);
                    }
                }
            }
        }
    }
#ifdef MCLF_STAMPS
    if (tid == 0) f.state->xstamps[2] = MCLF_NOW();
#endif
    __syncthreads();                                             // every record of this launch has been seen: the groups are through
    MCLF_STAMP(2);
    const int ntab[2] = {min(f.sh ? sm.tbase[0][sh_world] : sm.ntab_seen[0], MCLF_TSLOTS), min(f.sh ? sm.tbase[1][sh_world] : sm.ntab_seen[1], MCLF_TSLOTS)};
    if (staged) {
        // ---- stage b: the last MCLF_TAB_WAVES waves bring the tables the records name into LDS (a round trip to memory that nothing
        // below waits for before the chain's barrier) while waves 0, 1 put the lists in sub-tile order and the other waves join the gaps
        if (wave >= MCLF_MAXW - MCLF_TAB_WAVES) {
            const int ntabs = ntab[0] + ntab[1], tw = wave - (MCLF_MAXW - MCLF_TAB_WAVES), tnw = MCLF_TAB_WAVES;
            const void* const dummy = (const void*)mclf_rec_ptr(f, 0, 0);
            for (int round = 0; round * 4 * tnw < ntabs; ++round) {
                const void* pt[4][2];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int q = tw + (4 * round + u) * tnw;
                    pt[u][0] = dummy; pt[u][1] = dummy;
                    if (q < ntabs) {
                        const int axis = q >= ntab[0] ? 1 : 0, slot = q - axis * ntab[0];
                        int r = 0;
                        while (r + 1 < sh_world && sm.tbase[axis][r + 1] <= slot) ++r;             // the rank whose table this staged slot is
                        const mclf_tab_elem* src = mclf_tab_ptr(f, axis, r, slot - sm.tbase[axis][r]) + lane * MCLF_ITEMS;
                        pt[u][0] = (const void*)src; pt[u][1] = (const void*)(src + 1);
                    }
                }
                int4 qt[4][2];
                asm volatile("global_load_dwordx4 %0, %8, off sc1\n\tglobal_load_dwordx4 %1, %9, off sc1\n\t"
                             "global_load_dwordx4 %2, %10, off sc1\n\tglobal_load_dwordx4 %3, %11, off sc1\n\t"
                             "global_load_dwordx4 %4, %12, off sc1\n\tglobal_load_dwordx4 %5, %13, off sc1\n\t"
                             "global_load_dwordx4 %6, %14, off sc1\n\tglobal_load_dwordx4 %7, %15, off sc1\n\ts_waitcnt vmcnt(0)"
                             : "=&v"(qt[0][0]), "=&v"(qt[0][1]), "=&v"(qt[1][0]), "=&v"(qt[1][1]), "=&v"(qt[2][0]), "=&v"(qt[2][1]), "=&v"(qt[3][0]), "=&v"(qt[3][1])
                             : "v"(pt[0][0]), "v"(pt[0][1]), "v"(pt[1][0]), "v"(pt[1][1]), "v"(pt[2][0]), "v"(pt[2][1]), "v"(pt[3][0]), "v"(pt[3][1])
                             : "memory");
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int q = tw + (4 * round + u) * tnw;
                    if (q < ntabs) {
                        const int axis = q >= ntab[0] ? 1 : 0, slot = q - axis * ntab[0];
                        mclf_tab_elem* dst = MCLF_STAGE(axis).tab + slot * MCLF_SUB + lane * MCLF_ITEMS;
#pragma unroll
                        for (int k = 0; k < MCLF_ITEMS; ++k) {
                            mclf_tab_elem e;
                            e.t = __hiloint2double(qt[u][k].y, qt[u][k].x); e.se = qt[u][k].z; e.se1 = qt[u][k].w;
                            dst[k] = e;
                        }
                    }
                }
            }
        }
        if (wave < 2) {
            const mclf_stage a = MCLF_STAGE(wave);
            const int n = *a.nent > MCLF_MAXENT ? 0 : *a.nent;              // (an overflowed list is no list: the chain walks)
            const int mine = lane < n ? a.ent[lane].s : 0x7fffffff;
            int rank = 0;
            for (int j = 0; j < n; ++j) rank += (__builtin_amdgcn_readlane(mine, j) < mine) ? 1 : 0;
            if (lane < n) a.order[rank] = lane;
        }
        __syncthreads();
        if (wave < MCLF_MAXW - MCLF_TAB_WAVES) {
            const int n0 = *MCLF_STAGE(0).nent > MCLF_MAXENT ? 0 : *MCLF_STAGE(0).nent, n1 = *MCLF_STAGE(1).nent > MCLF_MAXENT ? 0 : *MCLF_STAGE(1).nent;
            for (int it = wave; it < n0 + n1 + 2; it += MCLF_MAXW - MCLF_TAB_WAVES) {
                const int axis = it > n0 ? 1 : 0, k = it - axis * (n0 + 1);
                const mclf_stage a = MCLF_STAGE(axis);
                const int n = axis ? n1 : n0;
                const int pb = k > 0 ? (a.ent[a.order[k - 1]].s >> 6) : -1;
                const int cb = k < n ? (a.ent[a.order[k]].s >> 6) : nbatch;
                ss_rec g;
                if (k > 0 && k < n && pb == cb) g = a.ent[a.order[k]].head;
                else {
                    g = k > 0 ? a.btail[pb] : ss_rec_identity();
                    g = ss_rec_join(g, mclf_join_batches(a, pb + 1, cb - 1, lane));
                    if (k < n) g = ss_rec_join(g, a.ent[a.order[k]].head);
                }
                if (lane == 0) {
                    mclf_step sp;
                    sp.gap = g;
                    if (k < n) { const mclf_ent en = a.ent[a.order[k]]; sp.s = en.s; sp.tslot = en.tslot; sp.tkey = en.tkey; sp.lo_n = en.lo_n; }
                    else { sp.s = nrec; sp.tslot = -1; sp.tkey = 0; sp.lo_n = -1; }
                    a.step[k] = sp;
                }
            }
        }
        __syncthreads();
    }
    MCLF_STAMP(3);
    // an axis whose list overflowed (the sum hovers around zero) is walked with composition trees: its table area holds them, the
    // waves that would idle through the chains build them (mclf_tree_helper)
    bool trees[2] = {false, false};
    if (staged && f.wild != nullptr && f.sh == nullptr && f.no_trees == 0) {
        for (int ax = 0; ax < 2; ++ax) trees[ax] = *MCLF_STAGE(ax).nent > MCLF_MAXENT && nrec > MCLF_PRE_SUBS;
        if (tid < 2 && trees[tid]) {
            const mclf_trees tr = mclf_trees_at(MCLF_STAGE(tid));
            for (int q = 0; q < MCLF_TREE_RING; ++q) tr.ready[q] = 0;
            *tr.consumed = 0; *tr.abort = 0;
        }
        if (trees[0] || trees[1]) __syncthreads();
    }
    if (wave >= 2 && (trees[0] || trees[1])) {
        const bool both = trees[0] && trees[1];
        const int ax = both ? (wave & 1) : (trees[0] ? 0 : 1);
        const int hidx = both ? (wave - 2) >> 1 : wave - 2, nh = both ? (MCLF_MAXW - 2) / 2 : MCLF_MAXW - 2;
        const mclf_stage hs = MCLF_STAGE(ax);
        mclf_tree_helper(f, mclf_trees_at(hs), ax, min(MCLF_PRE_SUBS, nrec), nrec, hidx, nh, lane);
    }
    if (wave < 2) {                                                                // wave 0: pose.x, wave 1: pose.y
        unsigned int stats[4] = {0, 0, 0, 0};
        // the sums behind the first sub-tiles, from the pre-chain workgroup (long done by now, as a rule)
        unsigned long long fw;
        unsigned int spins = 0;
        while ((((fw = mclf_load_u64(f.sync + 1 + wave)) >> 32) & 1ull) == 0ull) {
            if (++spins > MCLF_SPIN_LIMIT) { if (lane == 0) atomicAdd(&f.state->wait_timeouts, 1u); break; }
            __builtin_amdgcn_s_sleep(1);
        }
#ifdef MCLF_STAMPS
        if (tid == 0) f.state->xstamps[4] = MCLF_NOW();
#endif
        const float first = __uint_as_float((unsigned int)fw);
        const mclf_stage mine = MCLF_STAGE(wave);
        const float v = mclf_chain(f, staged ? &mine : nullptr, wave ? ntab[1] : ntab[0], wave, S, first, lane, stats, trees[wave], &sm.pre[wave][0]);      // (sm.pre: the pre-chain workgroup's buffer, idle in this one)
        if (lane == 0) { sm.xy[wave] = v; for (int k = 0; k < 4; ++k) sm.stats[4 * wave + k] = stats[k]; }
    }
    __syncthreads();
    MCLF_STAMP(4);
    if (tid == 0) {
        mclf_store_u64(f.sync, 0ull);                        // the next launch on this stream hands its tables out from slot 0 again
        mclf_store_u64(f.sync + 1, 0ull);
        mclf_store_u64(f.sync + 2, 0ull);
        double tot[5] = {0, 0, 0, 0, 0};
        for (int w = 0; w < MCLF_POSE_THREADS / 64; ++w) for (int k = 0; k < 5; ++k) tot[k] += sm.red[w][k];
        f.state->S = tot[0];
        bl_pose_xyt_t p;
        p.utime = f.utime;
        p.x = sm.xy[0];
        p.y = sm.xy[1];
        p.theta = (float)atan2(tot[3], tot[4]);
        f.state->pose = p;
        if (publish) {
            // another workgroup of this launch (the map update) is waiting for x and y: f.sync[3] / [4], which that workgroup clears
            mclf_store_u64(f.sync + 3, (unsigned long long)__float_as_uint(p.x) | (1ull << 32));
            mclf_store_u64(f.sync + 4, (unsigned long long)__float_as_uint(p.y) | (1ull << 32));
        }
        for (int k = 0; k < 5; ++k) f.state->sums_used[k] = tot[k];
        for (int k = 0; k < 8; ++k) f.state->chain_stats[k] = sm.stats[k];
        uni_update(f.state, f.N, tot[0], f.uni_mode, f.w_floor);        // (behind the published pose: one compare and one store unless every particle is at the floor)
#ifdef MCLF_STAMPS
        stamp[5] = MCLF_NOW();
        for (int k = 0; k < 6; ++k) f.state->stamps[k] = stamp[k];
#endif
    }
}
#endif  // __HIPCC__

// bl_mcl.hip: if `pf` has an update begun (bl_pf_update_begin) whose end can ride in another launch (the whole particle set
// on this device, the partial-sum form of the finish), does the end-of-update bookkeeping, fills `out` and returns 1; the
// caller MUST then launch the finish.  Returns 0 when there is nothing to take (no update pending: the robot did not move)
// and a negative status when the pending update cannot ride (the caller falls back to bl_pf_update_end).
int bl_pf_take_finish(bl_pf* pf, mcl_finish_args* out);
bl_ctx* bl_pf_ctx(bl_pf* pf);
// a finish that was taken and could not ride after all (its carrier failed before the launch): k_mcl_finish on its own
int bl_pf_launch_taken_finish(bl_pf* pf, const mcl_finish_args* fin);
// the launch that carries a taken finish has been enqueued (strict resampling: the cumulative's launches go behind it)
void bl_pf_ride_launched(bl_pf* pf);

#endif
