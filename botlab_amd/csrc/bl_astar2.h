// bl_astar2.h -- k_astar2: search_for_path (src/planning/astar.cpp:9-137) with the open list in split storage.
// Included by bl_planning.hip behind astar_args / astar_result.
//
// The search is the reference's loop, pop for pop (astar.cpp:75-135): libstdc++'s std::pop_heap / std::push_heap index
// operations (stl_heap.h:128-146, 214-250, comparator greater-by-fCost, astar.hpp:41-44) decide the order of equal fCosts, so
// they are executed exactly -- by one wavefront.  What k_astar2 changes against k_astar is where an entry lives:
//
//   * the KEY of entry i (its fCost, biased to unsigned: fCost + 32768, 16 bits) sits in slot i + 1 of a key array; slots
//     0 .. 2^(KLV+1) - 1 (heap levels 0 .. KLV) are LDS, deeper slots are global memory.  fCost < INT16_MAX is the reference's own
//     push condition (astar.cpp:103,124); fCost > -32768 holds because gCost, hCost >= 0 and the host checks the cost table's
//     minimum before it picks this kernel.  Slot 0 holds 0 (below every key): a lane without an ancestor reads it and stops the
//     climb.  Every LDS slot at or behind the heap's length holds 0xFFFF (above every key): a sift-down round reads the two
//     children of a node as ONE aligned 32-bit word (slots 2n + 2, 2n + 3) and needs no bounds checks.
//   * the PAYLOAD of entry i ((cy << 17) | (cx << 2) | move, as in k_astar) sits in entry i of a payload array; entries of
//     levels 0 .. PLV are LDS, deeper ones global.
//
// Every decision of a heap operation (the walk's path, where the value lands, how far a push rises) is taken from keys alone:
// four times the entries of k_astar's 8-byte layout fit the LDS (65 535 instead of 16 383; 16 383 instead of 4 095 beside the
// particle filter), and a round reads one word per lane.  Payloads only move; their loads are issued when the path is known and
// their stores wait until the end of the operation, so a payload that lives in global memory costs no round trip on the
// decision chain.  A pop stores nothing until the walk has reached its leaf (the walk's nodes above the landing level take their
// successor's entry, the landing node takes the value, the nodes below keep theirs: std::__adjust_heap + std::__push_heap in
// one net pass, proved against libstdc++ lane for lane by tests/tools/heap2_model.cpp).
#ifndef BL_ASTAR2_H
#define BL_ASTAR2_H

#include "bl_astar2_turbo.h"
#include "bl_astar2_deep.h"
#include "bl_astar2_duo.h"
#include "bl_astar2_ahead.h"

#define A2_COSTN 256
#define A2_INF 0xFFFFu
#define A2_MAXR 6

template <int KLV, int PLV> struct a2cfg {
    static constexpr int KSLOTS = 1 << (KLV + 1);            // key slots in LDS
    static constexpr int PLN = (1 << (PLV + 1)) - 1;         // payload entries in LDS (entry PLN: dummy)
    static constexpr int KEY_BYTES = KSLOTS * 2;
    static constexpr int PAY_OFF = KEY_BYTES;
    static constexpr int PAY_BYTES = (PLN + 1) * 4;
    static constexpr int COST_OFF = PAY_OFF + PAY_BYTES;
    static constexpr int TBL_OFF = COST_OFF + A2_COSTN * 4;  // constants of the straight-line loop (bl_astar2_turbo.h)
    static constexpr int BYTES = TBL_OFF + A2T_TBL_BYTES;
    static constexpr int LEV = KLV;                          // deepest heap level whose keys are LDS
    static constexpr int PLEV = PLV;                         // ... whose payloads are
    static constexpr int FD = KLV - 10;                      // levels the first sift-down round descends: rounds start at levels 0, FD, FD + 5, KLV
    static_assert(FD >= 1 && FD <= 5, "round layout");
    static_assert(PLV >= FD + 5 && PLV <= FD + 9, "payload tier boundary must fall into the third round");
};
typedef a2cfg<14, 13> a2_big;       // 64 KB keys (32 767 entries) + 64 KB payloads (16 383): a search that has a CU to itself
typedef a2cfg<12, 11> a2_small;     // 16 KB keys (8 191) + 16 KB payloads (4 095): beside the particle filter's workgroups
typedef a2cfg<11, 6> a2_test;       // tests: every storage tier within a few thousand entries

typedef __attribute__((address_space(3))) unsigned short a2_lds_u16;
typedef __attribute__((address_space(3))) unsigned int a2_lds_u32;
typedef unsigned int a2_v4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) a2_v4 a2_lds_u128;
typedef __attribute__((address_space(1))) unsigned short a2_g_u16;
typedef __attribute__((address_space(1))) unsigned int a2_g_u32;

__device__ __forceinline__ unsigned a2_lds16(unsigned addr) { return *(const a2_lds_u16*)(size_t)addr; }
__device__ __forceinline__ unsigned a2_lds32(unsigned addr) { return *(const a2_lds_u32*)(size_t)addr; }
// a lane mask as the scalar it is (free where the compiler already knows: the masks come from compares and lane reads)
__device__ __forceinline__ unsigned long long a2_uni(unsigned long long m)
{
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(m >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)m);
}
// Stores under a lane mask: exec is set from the mask around the instruction (the wave runs with all 64 lanes enabled everywhere
// in the search loop).  No address select, no branch; a zero mask makes the instruction a no-op.
__device__ __forceinline__ void a2_st_lds16(unsigned long long m, unsigned addr, unsigned val)
{
    asm volatile("s_mov_b64 exec, %0\n\tds_write_b16 %1, %2\n\ts_mov_b64 exec, -1" :: "s"(a2_uni(m)), "v"(addr), "v"(val) : "memory");
}
__device__ __forceinline__ void a2_st_lds32(unsigned long long m, unsigned addr, unsigned val)
{
    asm volatile("s_mov_b64 exec, %0\n\tds_write_b32 %1, %2\n\ts_mov_b64 exec, -1" :: "s"(a2_uni(m)), "v"(addr), "v"(val) : "memory");
}
__device__ __forceinline__ void a2_st_lds16_32(unsigned long long m, unsigned addr16, unsigned val16, unsigned addr32, unsigned val32)
{
    asm volatile("s_mov_b64 exec, %0\n\tds_write_b16 %1, %2\n\tds_write_b32 %3, %4\n\ts_mov_b64 exec, -1"
                 :: "s"(a2_uni(m)), "v"(addr16), "v"(val16), "v"(addr32), "v"(val32) : "memory");
}
__device__ __forceinline__ void a2_st_g16(unsigned long long m, a2_g_u16* p, unsigned val)
{
    asm volatile("s_mov_b64 exec, %0\n\tglobal_store_short %1, %2, off\n\ts_mov_b64 exec, -1" :: "s"(a2_uni(m)), "v"(p), "v"(val) : "memory");
}
__device__ __forceinline__ void a2_st_g32(unsigned long long m, a2_g_u32* p, unsigned val)
{
    asm volatile("s_mov_b64 exec, %0\n\tglobal_store_dword %1, %2, off\n\ts_mov_b64 exec, -1" :: "s"(a2_uni(m)), "v"(p), "v"(val) : "memory");
}
// Loads under a lane mask (lanes outside keep `old`): global addresses of idle lanes are never touched.
__device__ __forceinline__ unsigned a2_ld_g32(unsigned long long m, const a2_g_u32* p, unsigned old)
{
    asm volatile("s_mov_b64 exec, %1\n\tglobal_load_dword %0, %2, off\n\ts_mov_b64 exec, -1" : "+v"(old) : "s"(a2_uni(m)), "v"(p) : "memory");
    return old;
}
__device__ __forceinline__ unsigned a2_ld_g16(unsigned long long m, const a2_g_u16* p, unsigned old)
{
    asm volatile("s_mov_b64 exec, %1\n\tglobal_load_ushort %0, %2, off\n\ts_mov_b64 exec, -1" : "+v"(old) : "s"(a2_uni(m)), "v"(p) : "memory");
    return old;
}
// The results of the masked loads above are used behind one of these (the values are operands: every later use reads the register
// as it is after the wait).
__device__ __forceinline__ void a2_wait_vm1(unsigned& x) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(x) :: "memory"); }
__device__ __forceinline__ void a2_wait_vm4(unsigned& x, unsigned& y, unsigned& z, unsigned& w) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(x), "+v"(y), "+v"(z), "+v"(w) :: "memory"); }

#define A2_ULT 36
#define A2_ULE 37
#define A2_UGT 34
#define A2_EQ 32

// BOTLAB_ASTAR_NO_TURBO=1 (read by the host, astar_launch_kernel): the C++ loop everywhere (tests, A/B runs)
__device__ bool a2_turbo_enabled = true;
__device__ bool a2_deep_ahead_enabled = true;     // the same three waves beyond LDS (bl_astar2_ahead.h, "the deep regime": 0.88 - 0.90 us per pop where
                                                  // bl_astar2_deep.h's one wave takes 1.12 - 1.14); BOTLAB_ASTAR_DEEP_AHEAD=0: that one-wave loop
__device__ bool a2_walk_ahead_enabled = true;     // LDS-regime loop of single searches: the next pop's walk beside the pushes (bl_astar2_ahead.h: pops / pushes / expansions
                                                  // on three waves, 0.634 us per pop where bl_astar2_duo.h's two take 0.71 - 0.72); BOTLAB_ASTAR_AHEAD=0: the duo loop, =1: two waves

// per-lane constants of the wave-parallel heap operations
struct a2_lanes {
    int lane;
    int lk, ljm1;                  // sift-down rounds: this lane is the node at level lk, index ljm1 + 1 of the subtree under the hole
    unsigned amask, areq;          // ... and is on the walk iff (M & amask) == areq for the mask M of lanes that prefer their right child
    unsigned sh1, sh0;             // pushes: lane a looks at the hole's (a + 1)-th ancestor, slot hp >> sh1; it drops to slot hp >> sh0
};

// what a sift-down round leaves for the end of the pop
struct a2_round {
    unsigned long long Q;          // lanes on the walk whose preferred child is on the walk too ("positions")
    unsigned node, child, knext;   // this lane's node, its preferred child, that child's key
    unsigned pl, pg;               // the child's payload (LDS tier / global tier)
};

// One sift-down round over the 6-level (first round: FD + 1 levels) subtree under the hole hp - 1, keys in LDS: which lanes are on
// the walk, and for each its preferred child with key and payload.  Returns hp of the walk's node on the subtree's last level
// (or of the leaf the walk ended on: a round started there finds no position and changes nothing).
template <class C, int RI>
__device__ __forceinline__ unsigned a2_round_lds(a2_round& R, unsigned hp, unsigned len, unsigned kbase, unsigned pbase, const a2_g_u32* gp, const a2_lanes& ln)
{
    const unsigned long long lanes_ok = RI == 0 ? (2ull << ((1 << (C::FD + 1)) - 2)) - 1ull : 0x7FFFFFFFFFFFFFFFull;
    const unsigned node = (hp << ln.lk) + (unsigned)ln.ljm1;
    // the two children's keys as one word (slots 2n + 2, 2n + 3); lanes of the subtree's last level read a word nobody uses
    const unsigned pair = a2_lds32(min(kbase + 4u + 4u * node, kbase + (unsigned)C::KEY_BYTES - 4u));
    const unsigned long long V = __builtin_amdgcn_uicmp(node, len, A2_ULT) & lanes_ok;
    const unsigned fl = pair & 0xffffu, fr = pair >> 16;
    // right child preferred unless comp(right, left), i.e. right.fCost > left.fCost (stl_heap.h:224-226); a lone left child is
    // taken (stl_heap.h:231-237): its sibling's slot reads 0xFFFF
    const unsigned M = (unsigned)__builtin_amdgcn_uicmp(fr, fl, A2_ULE);
    const unsigned long long P = __builtin_amdgcn_uicmp(M & ln.amask, ln.areq, A2_EQ) & V;
    const int cur = 63 - __clzll((long long)P);
    R.Q = P & ~(1ull << cur);
    R.node = node;
    R.knext = min(fl, fr);
    R.child = 2u * node + 1u + (fr <= fl ? 1u : 0u);
    R.pl = a2_lds32(pbase + 4u * min(R.child, (unsigned)C::PLN));
    R.pg = 0;
    if (RI == 2) { if (((R.Q >> ln.lane) & 1ull) && R.child >= (unsigned)C::PLN) R.pg = gp[R.child]; }
    return (unsigned)__builtin_amdgcn_readlane((int)node, cur) + 1u;
}

// the same with the children's keys in global memory (no 0xFFFF behind the heap there: bounds are checked)
__device__ __forceinline__ unsigned a2_round_global(a2_round& R, unsigned hp, unsigned len, const a2_g_u32* gk32, const a2_g_u32* gp, const a2_lanes& ln, bool* more)
{
    const unsigned node = (hp << ln.lk) + (unsigned)ln.ljm1;
    const bool valid = ln.lane < 63 && node < len;
    unsigned pair = 0xFFFFFFFFu;
    if (valid && 2u * node + 1u < len) pair = gk32[node + 1u];
    const unsigned fl = pair & 0xffffu, fr = 2u * node + 2u < len ? (pair >> 16) : A2_INF;
    const unsigned M = (unsigned)__builtin_amdgcn_uicmp(fr, fl, A2_ULE);
    const unsigned long long P = __builtin_amdgcn_uicmp(M & ln.amask, ln.areq, A2_EQ) & __builtin_amdgcn_ballot_w64(valid);
    const int cur = 63 - __clzll((long long)P);
    R.Q = P & ~(1ull << cur);
    R.node = node;
    R.knext = min(fl, fr);
    R.child = 2u * node + 1u + (fr <= fl ? 1u : 0u);
    R.pl = 0; R.pg = 0;
    if ((R.Q >> ln.lane) & 1ull) R.pg = gp[R.child];
    const unsigned hole = (unsigned)__builtin_amdgcn_readlane((int)node, cur);
    *more = cur >= 31 && 2u * hole + 1u < len;
    return hole + 1u;
}

// positions W of round R take their successor's entry (RI: which storage tiers the round's nodes live in)
template <class C, int RI>
__device__ __forceinline__ void a2_round_store(const a2_round& R, unsigned long long W, unsigned kbase, unsigned pbase, a2_g_u16* gk, a2_g_u32* gp, int lane)
{
    if (RI < 2) a2_st_lds16_32(W, kbase + 2u + 2u * R.node, R.knext, pbase + 4u * R.node, R.pl);
    else if (RI == 2) {
        const unsigned long long nl = __builtin_amdgcn_uicmp(R.node, (unsigned)C::PLN, A2_ULT);
        const unsigned pv = R.child < (unsigned)C::PLN ? R.pl : R.pg;
        a2_st_lds16_32(W & nl, kbase + 2u + 2u * R.node, R.knext, pbase + 4u * R.node, pv);
        a2_st_lds16(W & ~nl, kbase + 2u + 2u * R.node, R.knext);
        if (((W & ~nl) >> lane) & 1ull) gp[R.node] = pv;
    } else {
        // (the first such round's lane 0 is the walk's node on level KLV: its key slot is an LDS slot, its children's are not)
        const unsigned long long kl = __builtin_amdgcn_uicmp(R.node + 1u, (unsigned)C::KSLOTS, A2_ULT);
        a2_st_lds16(W & kl, kbase + 2u + 2u * R.node, R.knext);
        if (((W & ~kl) >> lane) & 1ull) gk[R.node + 1u] = (unsigned short)R.knext;
        if ((W >> lane) & 1ull) gp[R.node] = R.pg;
    }
}

// the value lands on node `land`
template <class C>
__device__ __forceinline__ void a2_land(unsigned land, unsigned vk, unsigned vp, unsigned kbase, unsigned pbase, a2_g_u16* gk, a2_g_u32* gp, int lane)
{
    if (land + 1u < (unsigned)C::KSLOTS) a2_st_lds16(1ull, kbase + 2u + 2u * land, vk); else if (lane == 0) gk[land + 1u] = (unsigned short)vk;
    if (land < (unsigned)C::PLN) a2_st_lds32(1ull, pbase + 4u * land, vp); else if (lane == 0) gp[land] = vp;
}

// std::__adjust_heap(first, 0, len, value) + the std::__push_heap it ends with (stl_heap.h:214-250, 128-146) for a heap whose
// walk stays inside NR rounds of LDS keys: the rounds find the walk and store nothing; then the climb -- the deepest position
// whose successor's key is not greater than the value's -- decides where the value lands, and one net pass of stores moves the
// positions above it up by one level.
template <class C, int NR>
__device__ __forceinline__ void a2_pop_lds(unsigned len, unsigned vk, unsigned vp, unsigned kbase, unsigned pbase, a2_g_u16* gk, a2_g_u32* gp, const a2_lanes& ln)
{
    a2_round R[3];
    unsigned hp = 1;
    hp = a2_round_lds<C, 0>(R[0], hp, len, kbase, pbase, gp, ln);
    if (NR > 1) hp = a2_round_lds<C, 1>(R[1], hp, len, kbase, pbase, gp, ln);
    if (NR > 2) hp = a2_round_lds<C, 2>(R[2], hp, len, kbase, pbase, gp, ln);
    (void)hp;
    unsigned long long S = R[NR - 1].Q & ~__builtin_amdgcn_uicmp(R[NR - 1].knext, vk, A2_UGT);
    unsigned land = 0;
    if (S != 0ull) {
        // the usual case: the climb ends inside the last round
        const int L = 63 - __clzll((long long)S);
        land = (unsigned)__builtin_amdgcn_readlane((int)R[NR - 1].child, L);
        if (NR > 1) a2_round_store<C, 0>(R[0], R[0].Q, kbase, pbase, gk, gp, ln.lane);
        if (NR > 2) a2_round_store<C, 1>(R[1], R[1].Q, kbase, pbase, gk, gp, ln.lane);
        a2_round_store<C, NR - 1>(R[NR - 1], R[NR - 1].Q & ((2ull << L) - 1ull), kbase, pbase, gk, gp, ln.lane);
    } else {
        int rs = -1, L = 0;
        if (NR > 1) {
#pragma unroll
            for (int r = NR - 2; r >= 0; --r) {
                if (rs < 0) {
                    S = R[r].Q & ~__builtin_amdgcn_uicmp(R[r].knext, vk, A2_UGT);
                    if (S != 0ull) { rs = r; L = 63 - __clzll((long long)S); land = (unsigned)__builtin_amdgcn_readlane((int)R[r].child, L); }
                }
            }
        }
        if (rs >= 0) {
            const unsigned long long low = (2ull << L) - 1ull;
            a2_round_store<C, 0>(R[0], rs == 0 ? R[0].Q & low : R[0].Q, kbase, pbase, gk, gp, ln.lane);
            if (NR > 2 && rs >= 1) a2_round_store<C, 1>(R[1], R[1].Q & low, kbase, pbase, gk, gp, ln.lane);
        }
    }
    a2_land<C>(land, vk, vp, kbase, pbase, gk, gp, ln.lane);
}

// ... for a heap with key levels in global memory: three LDS rounds, then rounds of global loads
template <class C>
__device__ __forceinline__ void a2_pop_deep(unsigned len, unsigned vk, unsigned vp, unsigned kbase, unsigned pbase, a2_g_u16* gk, a2_g_u32* gp, const a2_lanes& ln)
{
    a2_round R[A2_MAXR];
    unsigned hp = 1;
    hp = a2_round_lds<C, 0>(R[0], hp, len, kbase, pbase, gp, ln);
    hp = a2_round_lds<C, 1>(R[1], hp, len, kbase, pbase, gp, ln);
    hp = a2_round_lds<C, 2>(R[2], hp, len, kbase, pbase, gp, ln);
    int nr = 3;
    bool more = 2u * (hp - 1u) + 1u < len;      // the walk's last node has a child: it sits on level KLV and the walk goes on in global memory
#pragma unroll
    for (int r = 3; r < A2_MAXR; ++r) {
        R[r].Q = 0; R[r].node = 0; R[r].child = 0; R[r].knext = 0; R[r].pl = 0; R[r].pg = 0;
        if (more) { hp = a2_round_global(R[r], hp, len, (const a2_g_u32*)gk, gp, ln, &more); nr = r + 1; }
    }
    int rs = -1, L = 0; unsigned land = 0;
#pragma unroll
    for (int r = A2_MAXR - 1; r >= 0; --r) {
        if (r < nr && rs < 0) {
            const unsigned long long S = R[r].Q & ~__builtin_amdgcn_uicmp(R[r].knext, vk, A2_UGT);
            if (S != 0ull) { rs = r; L = 63 - __clzll((long long)S); land = (unsigned)__builtin_amdgcn_readlane((int)R[r].child, L); }
        }
    }
    const unsigned long long low = (2ull << L) - 1ull;
    if (rs >= 0) a2_round_store<C, 0>(R[0], rs == 0 ? R[0].Q & low : R[0].Q, kbase, pbase, gk, gp, ln.lane);
    if (rs >= 1) a2_round_store<C, 1>(R[1], rs == 1 ? R[1].Q & low : R[1].Q, kbase, pbase, gk, gp, ln.lane);
    if (rs >= 2) a2_round_store<C, 2>(R[2], rs == 2 ? R[2].Q & low : R[2].Q, kbase, pbase, gk, gp, ln.lane);
#pragma unroll
    for (int r = 3; r < A2_MAXR; ++r)
        if (r < nr && rs >= r) a2_round_store<C, 3>(R[r], rs == r ? R[r].Q & low : R[r].Q, kbase, pbase, gk, gp, ln.lane);
    a2_land<C>(land, vk, vp, kbase, pbase, gk, gp, ln.lane);
}

// push_back + std::push_heap (stl_heap.h:128-146) of (kb, pv) onto a heap of len entries: lane a holds the (a + 1)-th ancestor of
// the hole; the value rises past the leading run of ancestors with a larger key, each of which drops one level.
template <class C>
__device__ __forceinline__ void a2_push_general(unsigned len, unsigned kb, unsigned pv, unsigned kbase, unsigned pbase, a2_g_u16* gk, a2_g_u32* gp, const a2_lanes& ln)
{
    const unsigned hp = len + 1u;
    const unsigned slot = hp >> ln.sh1;                                 // 0 for lanes without an ancestor: reads the key below every key
    unsigned ka;
    if ((hp >> 1) < (unsigned)C::KSLOTS) ka = a2_lds16(kbase + 2u * slot);           // every ancestor's slot is an LDS slot
    else {
        ka = a2_lds16(kbase + 2u * min(slot, (unsigned)C::KSLOTS - 1u));
        if (slot >= (unsigned)C::KSLOTS) ka = gk[slot];
    }
    const unsigned long long GT = __builtin_amdgcn_uicmp(ka, kb, A2_UGT);
    const int t = __ffsll((long long)~GT) - 1;
    const unsigned at = hp >> t;
    if (t > 0) {
        // the t nearest ancestors drop one level each (keys and payloads, whichever tier they live in)
        const unsigned below = hp >> ln.sh0;
        const unsigned long long mv = (1ull << t) - 1ull;
        const bool mine = ln.lane < t;
        unsigned pa = a2_lds32(pbase + 4u * min(slot - 1u, (unsigned)C::PLN));
        if (hp > (unsigned)C::PLN) { if (mine && slot - 1u >= (unsigned)C::PLN) pa = gp[slot - 1u]; }
        const unsigned long long kl = __builtin_amdgcn_uicmp(below, (unsigned)C::KSLOTS, A2_ULT);
        const unsigned long long pl = __builtin_amdgcn_uicmp(below - 1u, (unsigned)C::PLN, A2_ULT);
        a2_st_lds16(mv & kl, kbase + 2u * below, ka);
        a2_st_lds32(mv & pl, pbase + 4u * (below - 1u), pa);
        if (hp > (unsigned)C::PLN) {
            if (mine && below >= (unsigned)C::KSLOTS) gk[below] = (unsigned short)ka;
            if (mine && below - 1u >= (unsigned)C::PLN) gp[below - 1u] = pa;
        }
    }
    if (at < (unsigned)C::KSLOTS) a2_st_lds16(1ull, kbase + 2u * at, kb); else if (ln.lane == 0) gk[at] = (unsigned short)kb;
    if (at - 1u < (unsigned)C::PLN) a2_st_lds32(1ull, pbase + 4u * (at - 1u), pv); else if (ln.lane == 0) gp[at - 1u] = pv;
}


// ------------------------------------------------------------------------------------------------ hand-scheduled forms
// One wavefront alone on a CU pays ~4.2 cycles per instruction, ~50 per dependent LDS read, 28 - 60 per TAKEN branch and ~8 per
// hand-over between the vector and the scalar unit (tests/tools/lone_wave_probe.hip).  What hipcc makes of the functions above
// spends most of a heap operation in branches around tier checks, so the case that matters -- every key and payload the
// operation touches in LDS, which is every operation of a heap of up to PLN entries -- is written out as straight-line code:
// the same index operations, in the same order, checked by the same tests (tests/test_gpu_heap2.py runs every configuration
// with and without them).  They need the dynamic LDS segment at address 0 (the kernel has no static LDS; checked at its start).
// Scratch registers are fixed (v200 .. v239) and declared as clobbers.
#define A2_STR2(x) #x
#define A2_STR(x) A2_STR2(x)

// round R (0, 1, 2) on v(200 + 5 R): node, child, knext, payload of the child; Q mask into operand SQ; OK = that round's lane mask
#define A2_ASM_ROUND(N, C_, K, P, SQ, OK)                                                                     \
    "v_lshl_add_u32 " N ", %[hp], %[lk], %[ljm1]\n\t"                                                         \
    "v_lshl_add_u32 v220, " N ", 2, 4\n\t"                                                                    \
    "v_min_u32 v220, %[kmax], v220\n\t"                                                                       \
    "ds_read_b32 v221, v220\n\t"                                                                              \
    "v_cmp_gt_u32_e64 %[sV], %[len], " N "\n\t"                                                               \
    "s_and_b64 %[sV], %[sV], " OK "\n\t"                                                                      \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                \
    "v_cmp_le_u32_sdwa vcc, v221, v221 src0_sel:WORD_1 src1_sel:WORD_0\n\t"                                   \
    "v_min_u32_sdwa " K ", v221, v221 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n\t" \
    "v_lshl_add_u32 " C_ ", " N ", 1, 1\n\t"                                                                  \
    "v_and_b32 v222, vcc_lo, %[amask]\n\t"                                                                    \
    "v_addc_co_u32 " C_ ", vcc, 0, " C_ ", vcc\n\t"                                                           \
    "v_cmp_eq_u32 vcc, v222, %[areq]\n\t"                                                                     \
    "s_and_b64 " SQ ", vcc, %[sV]\n\t"                                                                        \
    "s_flbit_i32_b64 %[sT], " SQ "\n\t"                                                                       \
    "s_sub_i32 %[sT], 63, %[sT]\n\t"                                                                          \
    "s_bitset0_b64 " SQ ", %[sT]\n\t"                                                                         \
    "v_readlane_b32 %[hp], " N ", %[sT]\n\t"                                                                  \
    "s_add_i32 %[hp], %[hp], 1\n\t"                                                                           \
    "v_min_u32 v220, %[pln], " C_ "\n\t"                                                                      \
    "v_lshl_add_u32 v220, v220, 2, %[pb]\n\t"                                                                 \
    "ds_read_b32 " P ", v220\n\t"

// store addresses of a round's nodes (all lanes): key slot into KA, payload entry into N itself
#define A2_ASM_ADDR(N, KA)                                                                                    \
    "v_lshl_add_u32 " KA ", " N ", 1, 2\n\t"                                                                  \
    "v_lshl_add_u32 " N ", " N ", 2, %[pb]\n\t"
// positions of a round under mask M take their successor's entry
#define A2_ASM_STORE(N, KA, K, P, M)                                                                          \
    "s_mov_b64 exec, " M "\n\t"                                                                               \
    "ds_write_b16 " KA ", " K "\n\t"                                                                          \
    "ds_write_b32 " N ", " P "\n\t"

// the climb inside the last round (child CL, knext KL, mask SQL), the address computations (ADDRS) and stores (STORES: earlier
// rounds under their Q, the last one under %[sV]) and the landing; %[done] = 0 when the climb leaves the last round (the caller
// then runs the pop again in its general form: nothing has been stored)
#define A2_ASM_FINISH(CL, KL, SQL, ADDRS, STORES)                                                             \
    "v_cmp_lt_u32 vcc, %[vk], " KL "\n\t"                                                                     \
    "s_andn2_b64 %[sV], " SQL ", vcc\n\t"                                                                     \
    "s_mov_b32 %[done], 0\n\t"                                                                                \
    "s_cbranch_scc0 9f\n\t"                                                                                   \
    "s_flbit_i32_b64 %[sT], %[sV]\n\t"                                                                        \
    "s_sub_i32 %[sT], 63, %[sT]\n\t"                                                                          \
    "v_readlane_b32 %[done], " CL ", %[sT]\n\t"                                                               \
    "s_add_i32 %[sT], %[sT], 1\n\t"                                                                           \
    "s_bfm_b64 %[sV], %[sT], 0\n\t"                                                                           \
    "s_and_b64 %[sV], %[sV], " SQL "\n\t"                                                                     \
    ADDRS                                                                                                     \
    "s_lshl_b32 %[sT], %[done], 1\n\t"                                                                        \
    "s_add_i32 %[sT], %[sT], 2\n\t"                                                                           \
    "v_mov_b32 v220, %[sT]\n\t"                                                                               \
    "s_lshl_b32 %[sT], %[done], 2\n\t"                                                                        \
    "s_add_i32 %[sT], %[sT], %[pb]\n\t"                                                                       \
    "v_mov_b32 v221, %[sT]\n\t"                                                                               \
    "v_mov_b32 v222, %[vk]\n\t"                                                                               \
    "v_mov_b32 v223, %[vp]\n\t"                                                                               \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                \
    STORES                                                                                                    \
    "s_mov_b64 exec, 1\n\t"                                                                                   \
    "ds_write_b16 v220, v222\n\t"                                                                             \
    "ds_write_b32 v221, v223\n\t"                                                                             \
    "s_mov_b64 exec, -1\n\t"                                                                                  \
    "s_mov_b32 %[done], 1\n\t"                                                                                \
    "9:\n\t"

#define A2_ASM_POP_OPERANDS                                                                                   \
    : [hp] "+s"(hp), [done] "=&s"(done), [sV] "=&s"(sV), [sT] "=&s"(sT), [q0] "=&s"(q0), [q1] "=&s"(q1), [q2] "=&s"(q2) \
    : [len] "s"(len), [vk] "s"(vk), [vp] "s"(vp), [pb] "s"(pbase), [ok0] "s"(ok0), [ok] "s"(ok),               \
      [lk] "v"(ln.lk), [ljm1] "v"(ln.ljm1), [amask] "v"(ln.amask), [areq] "v"(ln.areq),                        \
      [kmax] "n"(C::KEY_BYTES - 4), [pln] "n"(C::PLN)                                                          \
    : "memory", "vcc", "scc", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v220", "v221", "v222", "v223"

// a2_pop_lds<C, NR> for a heap of at most PLN + 1 entries (every payload the walk touches in LDS); vk, vp uniform.  Returns false
// when the climb left the last round: nothing was stored, the caller runs the general form.
template <class C, int NR>
__device__ __forceinline__ bool a2_pop_fast(unsigned len, unsigned vk, unsigned vp, unsigned pbase, const a2_lanes& ln)
{
    unsigned hp = 1, done, sT;
    unsigned long long sV, q0, q1, q2;
    const unsigned long long ok0 = (2ull << ((1 << (C::FD + 1)) - 2)) - 1ull, ok = 0x7FFFFFFFFFFFFFFFull;
    if (NR == 1)
        asm volatile(A2_ASM_ROUND("v200", "v201", "v202", "v203", "%[q0]", "%[ok0]")
                     A2_ASM_FINISH("v201", "v202", "%[q0]", A2_ASM_ADDR("v200", "v204"), A2_ASM_STORE("v200", "v204", "v202", "v203", "%[sV]"))
                     A2_ASM_POP_OPERANDS);
    else if (NR == 2)
        asm volatile(A2_ASM_ROUND("v200", "v201", "v202", "v203", "%[q0]", "%[ok0]")
                     A2_ASM_ROUND("v205", "v206", "v207", "v208", "%[q1]", "%[ok]")
                     A2_ASM_FINISH("v206", "v207", "%[q1]", A2_ASM_ADDR("v200", "v204") A2_ASM_ADDR("v205", "v209"),
                                   A2_ASM_STORE("v200", "v204", "v202", "v203", "%[q0]") A2_ASM_STORE("v205", "v209", "v207", "v208", "%[sV]"))
                     A2_ASM_POP_OPERANDS);
    else
        asm volatile(A2_ASM_ROUND("v200", "v201", "v202", "v203", "%[q0]", "%[ok0]")
                     A2_ASM_ROUND("v205", "v206", "v207", "v208", "%[q1]", "%[ok]")
                     A2_ASM_ROUND("v210", "v211", "v212", "v213", "%[q2]", "%[ok]")
                     A2_ASM_FINISH("v211", "v212", "%[q2]", A2_ASM_ADDR("v200", "v204") A2_ASM_ADDR("v205", "v209") A2_ASM_ADDR("v210", "v214"),
                                   A2_ASM_STORE("v200", "v204", "v202", "v203", "%[q0]") A2_ASM_STORE("v205", "v209", "v207", "v208", "%[q1]")
                                   A2_ASM_STORE("v210", "v214", "v212", "v213", "%[sV]"))
                     A2_ASM_POP_OPERANDS);
    return done != 0;
}

// a2_push for a hole whose ancestors' keys and payloads are all LDS entries (hp = len + 1 <= PLN); kb, pv uniform
template <class C>
__device__ __forceinline__ void a2_push_fast(unsigned hp, unsigned kb, unsigned pv, unsigned pbase, const a2_lanes& ln)
{
    unsigned long long sM; unsigned sT, sU;
    asm volatile("v_lshrrev_b32_e64 v230, %[sh1], %[hp]\n\t"
                 "v_lshrrev_b32_e64 v231, %[sh0], %[hp]\n\t"
                 "v_lshlrev_b32 v232, 1, v230\n\t"
                 "v_add_u32 v233, -1, v230\n\t"
                 "ds_read_u16 v234, v232\n\t"
                 "v_min_u32 v233, %[pln], v233\n\t"
                 "v_lshl_add_u32 v233, v233, 2, %[pb]\n\t"
                 "ds_read_b32 v235, v233\n\t"
                 "v_lshlrev_b32 v236, 1, v231\n\t"
                 "v_lshl_add_u32 v237, v231, 2, %[pbm4]\n\t"
                 "s_waitcnt lgkmcnt(1)\n\t"
                 "v_cmp_lt_u32 vcc, %[kb], v234\n\t"
                 "s_not_b64 %[sM], vcc\n\t"
                 "s_ff1_i32_b64 %[sT], %[sM]\n\t"
                 "s_bfm_b64 %[sM], %[sT], 0\n\t"
                 "s_lshr_b32 %[sT], %[hp], %[sT]\n\t"
                 "s_lshl_b32 %[sU], %[sT], 1\n\t"
                 "s_lshl_b32 %[sT], %[sT], 2\n\t"
                 "s_add_i32 %[sT], %[sT], %[pbm4]\n\t"
                 "s_waitcnt lgkmcnt(0)\n\t"
                 "s_mov_b64 exec, %[sM]\n\t"
                 "ds_write_b16 v236, v234\n\t"
                 "ds_write_b32 v237, v235\n\t"
                 "s_mov_b64 exec, 1\n\t"
                 "v_mov_b32 v232, %[sU]\n\t"
                 "v_mov_b32 v233, %[kb]\n\t"
                 "v_mov_b32 v238, %[sT]\n\t"
                 "v_mov_b32 v239, %[pv]\n\t"
                 "ds_write_b16 v232, v233\n\t"
                 "ds_write_b32 v238, v239\n\t"
                 "s_mov_b64 exec, -1\n\t"
                 : [sM] "=&s"(sM), [sT] "=&s"(sT), [sU] "=&s"(sU)
                 : [hp] "s"(hp), [kb] "s"(kb), [pv] "s"(pv), [pb] "s"(pbase), [pbm4] "s"(pbase - 4u), [sh1] "v"(ln.sh1), [sh0] "v"(ln.sh0), [pln] "n"(C::PLN)
                 : "memory", "vcc", "scc", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239");
}

template <class C>
__device__ __forceinline__ void a2_push(unsigned len, unsigned kb, unsigned pv, unsigned kbase, unsigned pbase, a2_g_u16* gk, a2_g_u32* gp, const a2_lanes& ln, bool fast)
{
    if (fast && len < (unsigned)C::PLN) a2_push_fast<C>(len + 1u, kb, pv, pbase, ln);
    else a2_push_general<C>(len, kb, pv, kbase, pbase, gk, gp, ln);
}

// openList.pop() = std::pop_heap + pop_back on a heap of len >= 1 entries; (vk, vp) = what the caller has read from LDS for the
// entry at the back of the array (slot min(len, KSLOTS - 1), payload entry min(len - 1, PLN)): the value the sift-down places
// (stl_heap.h:254-262).  The slot that entry leaves is "behind the heap" from here on.
template <class C>
__device__ __forceinline__ void a2_pop(unsigned len, unsigned vk, unsigned vp, unsigned kbase, unsigned pbase, a2_g_u16* gk, a2_g_u32* gp, const a2_lanes& ln, bool fast)
{
    const unsigned last = len - 1u;
    if (len >= (unsigned)C::KSLOTS) vk = gk[len];
    if (last >= (unsigned)C::PLN) vp = gp[last];
    a2_st_lds16(len < (unsigned)C::KSLOTS ? 1ull : 0ull, kbase + 2u * len, A2_INF);
    if (last > 0) {
        const int D = 31 - __clz((int)last);             // level of the heap's last entry
        if (fast && len <= (unsigned)C::PLN + 1u) {
            // (uniform values as the scalars they are: operands of the hand-scheduled forms)
            const unsigned vks = (unsigned)__builtin_amdgcn_readfirstlane((int)vk), vps = (unsigned)__builtin_amdgcn_readfirstlane((int)vp);
            bool placed;
            if (D <= C::FD) placed = a2_pop_fast<C, 1>(last, vks, vps, pbase, ln);
            else if (D <= C::FD + 5) placed = a2_pop_fast<C, 2>(last, vks, vps, pbase, ln);
            else placed = a2_pop_fast<C, 3>(last, vks, vps, pbase, ln);
            if (placed) return;
        }
        if (D <= C::FD) a2_pop_lds<C, 1>(last, vk, vp, kbase, pbase, gk, gp, ln);
        else if (D <= C::FD + 5) a2_pop_lds<C, 2>(last, vk, vp, kbase, pbase, gk, gp, ln);
        else if (D <= C::LEV) a2_pop_lds<C, 3>(last, vk, vp, kbase, pbase, gk, gp, ln);
        else a2_pop_deep<C>(last, vk, vp, kbase, pbase, gk, gp, ln);
    }
}

__device__ __forceinline__ a2_lanes a2_make_lanes(int lane)
{
    a2_lanes ln;
    ln.lane = lane;
    ln.lk = 31 - __clz(lane + 1);
    const int lj = (lane + 1) - (1 << ln.lk);
    ln.amask = 0; ln.areq = 0;
    for (int t = 0; t < ln.lk && lane < 63; ++t) {
        const int anc_lane = ((1 << t) - 1) + (lj >> (ln.lk - t));
        ln.amask |= 1u << anc_lane;
        ln.areq |= (unsigned)((lj >> (ln.lk - t - 1)) & 1) << anc_lane;
    }
    ln.ljm1 = lj - 1;
    ln.sh1 = min(lane + 1, 31); ln.sh0 = min(lane, 31);
    return ln;
}

// every key slot "behind the heap", slot 0 below every key
template <class C>
__device__ __forceinline__ void a2_init_lds(unsigned kbase, int lane)
{
    for (int i = lane; i < C::KEY_BYTES / 16; i += 64) *(a2_lds_u128*)(size_t)(kbase + 16u * i) = (a2_v4)(~0u);
    __syncthreads();
    if (lane == 0) *(a2_lds_u16*)(size_t)kbase = 0;
    __syncthreads();
}

// Test entry (bl_debug_heap2_replay): replays a sequence of pushes (key >= 0: biased 16-bit key + payload) and pops (key < 0) on
// the split-storage heap and records every popped entry -- compared with libstdc++ by tests/test_gpu_heap2.py.
template <class C>
__global__ __launch_bounds__(64) void k_heap2_probe(const int* __restrict__ keys, const unsigned* __restrict__ pays, int n, int2* heap, int heap_cap,
                                                    unsigned* __restrict__ out_k, unsigned* __restrict__ out_p, int* out_n, unsigned long long* cycles, int use_fast)
{
    a2_g_u32* const gp = (a2_g_u32*)heap;
    a2_g_u16* const gk = (a2_g_u16*)((char*)heap + 4ll * heap_cap);
    const int lane = threadIdx.x;
    const unsigned kbase = (unsigned)(size_t)(__attribute__((address_space(3))) char*)s_heap;
    const unsigned pbase = kbase + C::PAY_OFF;
    a2_init_lds<C>(kbase, lane);
    const a2_lanes ln = a2_make_lanes(lane);
    const bool fast = use_fast != 0 && kbase == 0u;
    unsigned len = 0; int no = 0;
    unsigned long long c_push = 0, c_pop = 0, n_push = 0, n_pop = 0;
    for (int i = 0; i < n; ++i) {
        const int k = __builtin_amdgcn_readfirstlane(keys[i]);
        const unsigned pin = (unsigned)__builtin_amdgcn_readfirstlane((int)pays[i]);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" :: "s"(k), "s"(pin) : "memory");
        const unsigned long long c0 = __builtin_amdgcn_s_memtime();
        if (k >= 0) {
            if (len < (unsigned)heap_cap) { a2_push<C>(len, (unsigned)k, pin, kbase, pbase, gk, gp, ln, fast); len += 1; }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            c_push += __builtin_amdgcn_s_memtime() - c0; n_push += 1;
        } else if (len > 0) {
            const unsigned kt = a2_lds16(kbase + 2u), pt = a2_lds32(pbase);
            const unsigned vk = a2_lds16(kbase + 2u * min(len, (unsigned)C::KSLOTS - 1u));
            const unsigned vp = a2_lds32(pbase + 4u * min(len - 1u, (unsigned)C::PLN));
            if (lane == 0) { out_k[no] = kt; out_p[no] = pt; }
            no += 1;
            a2_pop<C>(len, vk, vp, kbase, pbase, gk, gp, ln, fast);
            len -= 1;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            c_pop += __builtin_amdgcn_s_memtime() - c0; n_pop += 1;
        }
    }
    if (lane == 0) { *out_n = no; cycles[0] = c_push; cycles[1] = n_push; cycles[2] = c_pop; cycles[3] = n_pop; }
}

// One wavefront runs the reference's search loop; lanes 0..3 evaluate the four neighbours of the popped node, lane 4 re-derives
// its gCost (as k_astar does).  Closed cells: closed[] holds (generation << 3) | move, written once per cell.
// Launched with 64 threads, or with 128: the second wavefront then runs the expansions of the LDS-regime loop beside the first
// (bl_astar2_duo.h) and waits at a barrier whenever the first is anywhere else.
template <class C>
__global__ __launch_bounds__(192) void k_astar2(astar_args a)
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
    // the scratch k_astar uses for 8-byte entries holds both arrays: payloads [heap_cap], then key slots [heap_cap + 2]
    a2_g_u32* const gp = (a2_g_u32*)a.heap;
    a2_g_u16* const gk = (a2_g_u16*)((char*)a.heap + 4ll * a.heap_cap);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    __builtin_amdgcn_s_setprio(3);
    astar_result res; res.status = ASTAR_ST_NOPATH; res.path_len = 0; res.pops = 0; res.pushes = 0;
    for (int q = 0; q < 6; ++q) res.stamps[q] = 0;
    res.path_off = 0;
    res.start = a.start_host;
    if (a.start_dev) {
        res.start = *a.start_dev;
        bl_global_to_cell((double)res.start.x, (double)res.start.y, a.frame, &a.sx, &a.sy);
    }
    const unsigned kbase = (unsigned)(size_t)(__attribute__((address_space(3))) char*)s_heap;
    const unsigned pbase = kbase + C::PAY_OFF;
    const unsigned cbase = kbase + C::COST_OFF;
    // ---- LDS: the key slots and the cost table
    a2_init_lds<C>(kbase, lane);
    const bool cost_in_lds = a.cost_n <= A2_COSTN;
    if (cost_in_lds) for (int i = lane; i < a.cost_n; i += 64) *(a2_lds_u32*)(size_t)(cbase + 4u * i) = (unsigned)a.cost_lut[i];
    __syncthreads();
    auto cell_cost = [&](int x, int y) -> int {
        if (x < 0 || y < 0 || x >= a.W || y >= a.H) return ASTAR_INVALID_COST;
        int n = a.l1[(size_t)y * a.W + x];
        if (n == 0xFFFF) return ASTAR_INVALID_COST;
        return a.cost_lut[min(n, a.cost_n - 1)];
    };
    const bool ok = cell_cost(a.gx, a.gy) != ASTAR_INVALID_COST       // astar.cpp:40-44
                    && cell_cost(a.sx, a.sy) != ASTAR_INVALID_COST    // :46-50
                    && !(a.sx == a.gx && a.sy == a.gy);               // :52-56
    if (!ok) { if (lane == 0 && wave == 0) { *a.result = res; if (a.host_out) *(astar_result*)a.host_out = res; } return; }

    const a2_lanes ln = a2_make_lanes(lane);
    const bool fast = kbase == 0u && a.max_pops >= 0;      // (hand-scheduled heap operations: the dynamic LDS segment starts at 0)
    // the straight-line loop's constants (bl_astar2_turbo.h): a row per lane, then the scalars
    const unsigned tbl = kbase + C::TBL_OFF;
    static_assert((C::TBL_OFF & 15) == 0, "table alignment");
    const bool turbo = fast && cost_in_lds && a2_turbo_enabled;
    const bool ahead = (long long)a.W * a.H * 6 > (3ll << 20);
    // (two rounds in global memory reach level LEV + 10 >= 22; the scratch bounds the list at 2^25 entries)
    const unsigned deep_max = (unsigned)min((long long)a.heap_cap, 1ll << min(C::LEV + 11, 25)) - 4u;
    const bool deep = turbo && C::PLEV == C::LEV - 1 && deep_max > (unsigned)C::PLN + 8u && deep_max < 0x7fffffffu;
    {
        a2_lds_u32* row = (a2_lds_u32*)(size_t)(tbl + 64u * (unsigned)lane);
        // cells two steps from the popped one (lanes 0..7): their lines are asked for one expansion ahead
        const int pdx = lane == 0 ? 2 : (lane == 1 ? -2 : (lane == 4 || lane == 5 ? 1 : (lane == 6 || lane == 7 ? -1 : 0)));
        const int pdy = lane == 2 ? 2 : (lane == 3 ? -2 : (lane == 4 || lane == 6 ? 1 : (lane == 5 || lane == 7 ? -1 : 0)));
        row[8] = (unsigned)pdx; row[9] = (unsigned)pdy;
        row[0] = (unsigned)ln.lk; row[1] = (unsigned)ln.ljm1; row[2] = ln.amask; row[3] = ln.areq; row[4] = ln.sh1; row[5] = ln.sh0;
        row[6] = (unsigned)(lane == 0 ? 1 : (lane == 1 ? -1 : 0)); row[7] = (unsigned)(lane == 2 ? 1 : (lane == 3 ? -1 : 0));
        if (lane == 0) {
            a2_lds_u32* sc = (a2_lds_u32*)(size_t)(tbl + 4096u);
            sc[A2T_SC_W] = (unsigned)a.W; sc[A2T_SC_H] = (unsigned)a.H; sc[A2T_SC_GX] = (unsigned)a.gx; sc[A2T_SC_GY] = (unsigned)a.gy;
            sc[A2T_SC_GEN] = a.closed_gen; sc[A2T_SC_CN1] = (unsigned)(a.cost_n - 1);
            sc[A2T_SC_MAXPOPS] = a.max_pops > 0x7fffffffll ? 0x7fffffffu : (unsigned)a.max_pops;
            sc[A2T_SC_LIM] = (unsigned)C::PLN - 5u;                       // the loop runs while 2 <= length <= PLN - 3
            sc[A2T_SC_L1] = (unsigned)(size_t)a.l1; sc[A2T_SC_L1 + 1] = (unsigned)((size_t)a.l1 >> 32);
            sc[A2T_SC_CLOSED] = (unsigned)(size_t)a.closed; sc[A2T_SC_CLOSED + 1] = (unsigned)((size_t)a.closed >> 32);
            sc[A2T_SC_PB] = pbase; sc[A2T_SC_CB] = cbase;
            sc[A2T_SC_LVL1] = 1u << (C::FD + 1); sc[A2T_SC_LVL2] = 1u << (C::FD + 6);
            sc[A2T_SC_GK] = (unsigned)(size_t)gk; sc[A2T_SC_GK + 1] = (unsigned)((size_t)gk >> 32);
            sc[A2T_SC_GP] = (unsigned)(size_t)gp; sc[A2T_SC_GP + 1] = (unsigned)((size_t)gp >> 32);
            sc[A2T_SC_DLIM] = deep_max - ((unsigned)C::PLN + 2u);                         // the deep loop runs while PLN + 2 <= length <= deep_max
            for (int q = 21; q < 64; ++q) sc[q] = 0;
            sc[A2W_RUN_WORD] = A2W_GO;
        }
        __syncthreads();
    }
    // ---- two wavefronts: the second runs the expansions of the LDS-regime loop until the first says QUIT (bl_astar2_duo.h)
    const bool duo = turbo && blockDim.x >= 128u;
    const bool walk_ahead = duo && a2_walk_ahead_enabled;
    const bool ahead3 = walk_ahead && blockDim.x == 192u;      // ... with the expansions on a third wave (BOTLAB_ASTAR_AHEAD=2)
    // (deep_max etc. are wave-uniform; the three waves agree on the forms they run)
    const bool deep3 = ahead3 && a2_deep_ahead_enabled && turbo && C::PLEV == C::LEV - 1;
    if (wave == 2) {
        if (ahead3) {
            if (ahead) asm volatile(A2A_BODY_EXPAND3(A2T_PREFETCH, "4", "3")
                         :: [tbl] "s"(__builtin_amdgcn_readfirstlane((int)tbl)), [kmax] "n"(C::KEY_BYTES - 4), [pln] "n"(C::PLN),
                            [ok0lo] "n"((unsigned)(((2ull << ((1 << (C::FD + 1)) - 2)) - 1ull) & 0xffffffffull)),
                            [ok0hi] "n"((unsigned)(((2ull << ((1 << (C::FD + 1)) - 2)) - 1ull) >> 32))
                         : A2A_EXPUSH_CLOBBERS);
            else asm volatile(A2A_BODY_EXPAND3("", "2", "1")
                         :: [tbl] "s"(__builtin_amdgcn_readfirstlane((int)tbl)), [kmax] "n"(C::KEY_BYTES - 4), [pln] "n"(C::PLN),
                            [ok0lo] "n"((unsigned)(((2ull << ((1 << (C::FD + 1)) - 2)) - 1ull) & 0xffffffffull)),
                            [ok0hi] "n"((unsigned)(((2ull << ((1 << (C::FD + 1)) - 2)) - 1ull) >> 32))
                         : A2A_EXPUSH_CLOBBERS);
        }
        return;
    }
    if (wave == 1) {
        if (deep3) {
            asm volatile(A2A_BODY_PUSH3D
                         :: [tbl] "s"(__builtin_amdgcn_readfirstlane((int)tbl)), [kmax] "n"(C::KEY_BYTES - 4), [kmax2] "n"(C::KEY_BYTES - 2), [pln] "n"(C::PLN),
                            [kslots] "n"(C::KSLOTS), [kslotsm1] "n"(C::KSLOTS - 1),
                            [ok0lo] "n"((unsigned)(((2ull << ((1 << (C::FD + 1)) - 2)) - 1ull) & 0xffffffffull)),
                            [ok0hi] "n"((unsigned)(((2ull << ((1 << (C::FD + 1)) - 2)) - 1ull) >> 32))
                         : A2A_PUSH3D_CLOBBERS);
        } else if (ahead3) {
            asm volatile(A2A_BODY_PUSH3
                         :: [tbl] "s"(__builtin_amdgcn_readfirstlane((int)tbl)), [kmax] "n"(C::KEY_BYTES - 4), [pln] "n"(C::PLN),
                            [ok0lo] "n"((unsigned)(((2ull << ((1 << (C::FD + 1)) - 2)) - 1ull) & 0xffffffffull)),
                            [ok0hi] "n"((unsigned)(((2ull << ((1 << (C::FD + 1)) - 2)) - 1ull) >> 32))
                         : A2A_PUSH3_CLOBBERS);
        } else if (walk_ahead) {
            if (ahead) asm volatile(A2A_BODY_EXPUSH(A2T_PREFETCH, "4", "3")
                         :: [tbl] "s"(__builtin_amdgcn_readfirstlane((int)tbl)), [kmax] "n"(C::KEY_BYTES - 4), [pln] "n"(C::PLN),
                            [ok0lo] "n"((unsigned)(((2ull << ((1 << (C::FD + 1)) - 2)) - 1ull) & 0xffffffffull)),
                            [ok0hi] "n"((unsigned)(((2ull << ((1 << (C::FD + 1)) - 2)) - 1ull) >> 32))
                         : A2A_EXPUSH_CLOBBERS);
            else asm volatile(A2A_BODY_EXPUSH("", "2", "1")
                         :: [tbl] "s"(__builtin_amdgcn_readfirstlane((int)tbl)), [kmax] "n"(C::KEY_BYTES - 4), [pln] "n"(C::PLN),
                            [ok0lo] "n"((unsigned)(((2ull << ((1 << (C::FD + 1)) - 2)) - 1ull) & 0xffffffffull)),
                            [ok0hi] "n"((unsigned)(((2ull << ((1 << (C::FD + 1)) - 2)) - 1ull) >> 32))
                         : A2A_EXPUSH_CLOBBERS);
        } else if (duo) {
            if (ahead) asm volatile(A2W_BODY_EXPAND(A2T_PREFETCH, "4")
                         :: [tbl] "s"(__builtin_amdgcn_readfirstlane((int)tbl)), [kmax] "n"(C::KEY_BYTES - 4), [pln] "n"(C::PLN),
                            [ok0lo] "n"((unsigned)(((2ull << ((1 << (C::FD + 1)) - 2)) - 1ull) & 0xffffffffull)),
                            [ok0hi] "n"((unsigned)(((2ull << ((1 << (C::FD + 1)) - 2)) - 1ull) >> 32))
                         : A2W_EXPAND_CLOBBERS);
            else asm volatile(A2W_BODY_EXPAND("", "2")
                         :: [tbl] "s"(__builtin_amdgcn_readfirstlane((int)tbl)), [kmax] "n"(C::KEY_BYTES - 4), [pln] "n"(C::PLN),
                            [ok0lo] "n"((unsigned)(((2ull << ((1 << (C::FD + 1)) - 2)) - 1ull) & 0xffffffffull)),
                            [ok0hi] "n"((unsigned)(((2ull << ((1 << (C::FD + 1)) - 2)) - 1ull) >> 32))
                         : A2W_EXPAND_CLOBBERS);
        }
        return;
    }
    // neighbour offsets: xDeltas {1,-1,0,0}, yDeltas {0,0,1,-1} (astar.cpp:215-216); lane 4: the cell itself
    const int ddx = lane == 0 ? 1 : (lane == 1 ? -1 : 0);
    const int ddy = lane == 2 ? 1 : (lane == 3 ? -1 : 0);
    const unsigned W = (unsigned)a.W, H = (unsigned)a.H;
    const unsigned max_pops = a.max_pops > 0x7fffffffll ? 0x7fffffffu : (unsigned)a.max_pops;
    const unsigned heap_cap = (unsigned)a.heap_cap;

    unsigned len = 1, pops = 0, pushes = 0;
    if (lane == 0) { *(a2_lds_u16*)(size_t)(kbase + 2) = 32768u; *(a2_lds_u32*)(size_t)pbase = (unsigned)((a.sy << 17) | (a.sx << 2)); }   // firstNode: all costs 0
    __builtin_amdgcn_wave_barrier();
    unsigned goal_m = 0;
    int cx = 0, cy = 0;
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, ta = 0, tb = 0, tr0 = 0, tr1 = 0, acc_adj = 0, acc_nb = 0, acc_all = 0, acc_wait = 0, acc_push = 0;
    (void)t0; (void)t1; (void)t2; (void)t3; (void)ta; (void)tb; (void)tr0; (void)tr1; (void)acc_adj; (void)acc_nb; (void)acc_all; (void)acc_wait; (void)acc_push;
#ifdef BL_ASTAR_STAMPS
    tr0 = __builtin_amdgcn_s_memrealtime();
#endif
    while (len > 0) {
        if (turbo && len >= 2u && len <= (unsigned)C::PLN - 3u) {
            // the regime of the straight-line loop: it runs until the open list leaves it, the goal is reached or the pop limit hit
            unsigned code, gm, ptop;
            unsigned s_len = (unsigned)__builtin_amdgcn_readfirstlane((int)len), s_pops = (unsigned)__builtin_amdgcn_readfirstlane((int)pops);
            unsigned s_pushes = (unsigned)__builtin_amdgcn_readfirstlane((int)pushes);
            // (grids whose distance + closed arrays fit the L2 gain nothing from asking for lines ahead)
            if (walk_ahead) asm volatile(A2A_BODY_POP
                         : [len] "+s"(s_len), [pops] "+s"(s_pops), [pushes] "+s"(s_pushes), [code] "=&s"(code), [gm] "=&s"(gm), [pt] "=&s"(ptop)
                         : [tbl] "s"(__builtin_amdgcn_readfirstlane((int)tbl)), [kmax] "n"(C::KEY_BYTES - 4), [pln] "n"(C::PLN),
                           [ok0lo] "n"((unsigned)(((2ull << ((1 << (C::FD + 1)) - 2)) - 1ull) & 0xffffffffull)),
                           [ok0hi] "n"((unsigned)(((2ull << ((1 << (C::FD + 1)) - 2)) - 1ull) >> 32))
                         : A2A_POP_CLOBBERS);
            else if (duo) asm volatile(A2W_BODY_HEAP
                         : [len] "+s"(s_len), [pops] "+s"(s_pops), [pushes] "+s"(s_pushes), [code] "=&s"(code), [gm] "=&s"(gm), [pt] "=&s"(ptop)
                         : [tbl] "s"(__builtin_amdgcn_readfirstlane((int)tbl)), [kmax] "n"(C::KEY_BYTES - 4), [pln] "n"(C::PLN),
                           [ok0lo] "n"((unsigned)(((2ull << ((1 << (C::FD + 1)) - 2)) - 1ull) & 0xffffffffull)),
                           [ok0hi] "n"((unsigned)(((2ull << ((1 << (C::FD + 1)) - 2)) - 1ull) >> 32))
                         : A2T_CLOBBERS);
            else if (ahead) asm volatile(A2T_BODY(A2T_PREFETCH, "2")
                         : [len] "+s"(s_len), [pops] "+s"(s_pops), [pushes] "+s"(s_pushes), [code] "=&s"(code), [gm] "=&s"(gm), [pt] "=&s"(ptop)
                         : [tbl] "s"(__builtin_amdgcn_readfirstlane((int)tbl)), [kmax] "n"(C::KEY_BYTES - 4), [pln] "n"(C::PLN),
                           [ok0lo] "n"((unsigned)(((2ull << ((1 << (C::FD + 1)) - 2)) - 1ull) & 0xffffffffull)),
                           [ok0hi] "n"((unsigned)(((2ull << ((1 << (C::FD + 1)) - 2)) - 1ull) >> 32))
                         : A2T_CLOBBERS);
            else asm volatile(A2T_BODY("", "0")
                         : [len] "+s"(s_len), [pops] "+s"(s_pops), [pushes] "+s"(s_pushes), [code] "=&s"(code), [gm] "=&s"(gm), [pt] "=&s"(ptop)
                         : [tbl] "s"(__builtin_amdgcn_readfirstlane((int)tbl)), [kmax] "n"(C::KEY_BYTES - 4), [pln] "n"(C::PLN),
                           [ok0lo] "n"((unsigned)(((2ull << ((1 << (C::FD + 1)) - 2)) - 1ull) & 0xffffffffull)),
                           [ok0hi] "n"((unsigned)(((2ull << ((1 << (C::FD + 1)) - 2)) - 1ull) >> 32))
                         : A2T_CLOBBERS);
            len = s_len; pops = s_pops; pushes = s_pushes;
            if (code == 2u) { goal_m = gm; cx = (int)((ptop >> 2) & 0x7fffu); cy = (int)(ptop >> 17); res.status = ASTAR_ST_FOUND; break; }
            if (code == 3u) { res.status = ASTAR_ST_LIMIT; break; }
            if (code == 4u) { res.status = ASTAR_ST_BROKEN; break; }          // (two-wave loop: the other wave's flag never came)
            if (len == 0) break;
        }
        if (deep && len >= (unsigned)C::PLN + 2u && len <= deep_max) {
            // ... and its form for open lists that reach into global memory (bl_astar2_deep.h)
            unsigned code, gm, ptop;
            unsigned s_len = (unsigned)__builtin_amdgcn_readfirstlane((int)len), s_pops = (unsigned)__builtin_amdgcn_readfirstlane((int)pops);
            unsigned s_pushes = (unsigned)__builtin_amdgcn_readfirstlane((int)pushes);
            if (deep3) asm volatile(A2A_BODY_POPD
                         : [len] "+s"(s_len), [pops] "+s"(s_pops), [pushes] "+s"(s_pushes), [code] "=&s"(code), [gm] "=&s"(gm), [pt] "=&s"(ptop)
                         : [tbl] "s"(__builtin_amdgcn_readfirstlane((int)tbl)), [kmax] "n"(C::KEY_BYTES - 4), [kmax2] "n"(C::KEY_BYTES - 2), [pln] "n"(C::PLN),
                           [kslots] "n"(C::KSLOTS), [kslotsm1] "n"(C::KSLOTS - 1), [dlo] "n"(C::PLN + 2),
                           [ok0lo] "n"((unsigned)(((2ull << ((1 << (C::FD + 1)) - 2)) - 1ull) & 0xffffffffull)),
                           [ok0hi] "n"((unsigned)(((2ull << ((1 << (C::FD + 1)) - 2)) - 1ull) >> 32))
                         : A2D_CLOBBERS);
            else asm volatile(A2D_BODY
                         : [len] "+s"(s_len), [pops] "+s"(s_pops), [pushes] "+s"(s_pushes), [code] "=&s"(code), [gm] "=&s"(gm), [pt] "=&s"(ptop)
                         : [tbl] "s"(__builtin_amdgcn_readfirstlane((int)tbl)), [kmax] "n"(C::KEY_BYTES - 4), [kmax2] "n"(C::KEY_BYTES - 2), [pln] "n"(C::PLN),
                           [kslots] "n"(C::KSLOTS), [kslotsm1] "n"(C::KSLOTS - 1), [dlo] "n"(C::PLN + 2),
                           [ok0lo] "n"((unsigned)(((2ull << ((1 << (C::FD + 1)) - 2)) - 1ull) & 0xffffffffull)),
                           [ok0hi] "n"((unsigned)(((2ull << ((1 << (C::FD + 1)) - 2)) - 1ull) >> 32))
                         : A2D_CLOBBERS);
            len = s_len; pops = s_pops; pushes = s_pushes;
            if (code == 2u) { goal_m = gm; cx = (int)((ptop >> 2) & 0x7fffu); cy = (int)(ptop >> 17); res.status = ASTAR_ST_FOUND; break; }
            if (code == 3u) { res.status = ASTAR_ST_LIMIT; break; }
            if (code == 4u) { res.status = ASTAR_ST_BROKEN; break; }
            if (len == 0) break;
            if (len >= 2u && len <= (unsigned)C::PLN - 3u) continue;        // back in the LDS regime
        }
        if (pops >= max_pops) { res.status = ASTAR_ST_LIMIT; break; }
        STAMP(t0);
        // ---- the top, and the entry at the back of the array: the value the pop's sift-down places (stl_heap.h:254-262)
        const unsigned last = len - 1u;
        const unsigned kt_v = a2_lds16(kbase + 2u), pt_v = a2_lds32(pbase);
        unsigned vk = a2_lds16(kbase + 2u * min(len, (unsigned)C::KSLOTS - 1u));
        unsigned vp = a2_lds32(pbase + 4u * min(last, (unsigned)C::PLN));
        const unsigned kb0 = (unsigned)__builtin_amdgcn_readfirstlane((int)kt_v);
        const unsigned pay0 = (unsigned)__builtin_amdgcn_readfirstlane((int)pt_v);
        cx = (int)((pay0 >> 2) & 0x7fffu); cy = (int)(pay0 >> 17);
        const int tdir = (int)(pay0 & 3u);
        // ---- the loads of this expansion, in flight across the pop (inline asm: hipcc must not wait for them at the next join)
        const int nx = cx + ddx, ny = cy + ddy;
        const bool inb = lane < 5 && (unsigned)nx < W && (unsigned)ny < H;
        const unsigned ncell = inb ? (unsigned)ny * W + (unsigned)nx : 0u;
        int my_l1, my_closed;
        asm volatile("global_load_ushort %0, %2, %4\n\tglobal_load_dword %1, %3, %5 sc1"
                     : "=&v"(my_l1), "=&v"(my_closed) : "v"(ncell * 2u), "v"(ncell * 4u), "s"(a.l1), "s"(a.closed) : "memory");
        STAMP(t1);
        // ---- openList.pop(): std::pop_heap + pop_back
        a2_pop<C>(len, vk, vp, kbase, pbase, gk, gp, ln, fast);
        len = last;
        STAMP(ta);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(my_l1), "+v"(my_closed) :: "memory");
        // closedList.push_back(nNode): only the first entry per cell is ever observed (is_member / get_member)
        if (lane == 4 && inb && ((unsigned int)my_closed >> 3) != a.closed_gen)
            *(__attribute__((address_space(1))) int*)(a.closed + ((unsigned)cy * W + (unsigned)cx)) = (int)((a.closed_gen << 3) | (unsigned int)(pops == 0 ? 4 : tdir));
        const bool nclosed = inb && lane < 4 && ((unsigned int)my_closed >> 3) == a.closed_gen;
        if (!inb) my_l1 = 0xFFFF;
        STAMP(t2);
        pops += 1;
        // ---- the four neighbours, one per lane (expand_node order is the lane order, astar.cpp:213-233)
        int my_cost = ASTAR_INVALID_COST;
        if (inb && my_l1 != 0xFFFF) {
            const int ci = min(my_l1, a.cost_n - 1);
            if (cost_in_lds) my_cost = (int)a2_lds32(cbase + 4u * (unsigned)ci); else my_cost = *(const __attribute__((address_space(1))) int*)(a.cost_lut + ci);
        }
        const int ax = abs(a.gx - nx), ay = abs(a.gy - ny);                     // get_hCost (:170-179); lane 4: the cell itself
        const int hmx = max(ax, ay), hmn = min(ax, ay);
        const int hc = (hmx << 3) + (hmx << 1) + (hmn << 2);                    // 14 * min + 10 * (max - min)
        // gCost of the popped node: fCost - hCost - oCost of its cell (the start node carries zeros, astar.cpp:66-69)
        const int c_g = __builtin_amdgcn_readlane(((int)kb0 - 32768) - hc - my_cost, 4);
        const int tg = (pops == 1) ? 0 : c_g;
        const bool nvalid = lane < 4 && my_cost != ASTAR_INVALID_COST;          // in grid and isValid
        const int f = tg + 10 + hc + (nvalid ? my_cost : 0);                    // get_gCost: 4-connected step
        goal_m = (unsigned int)__ballot(nvalid && nx == a.gx && ny == a.gy);
        // a valid neighbour is pushed unless it is closed (:123) or fNew >= INT16_MAX (:103,124)
        unsigned int push_m = (unsigned int)__ballot(nvalid && !nclosed && 32767 > f);
        const unsigned ey = (unsigned)((ny << 17) | (nx << 2) | lane);
        const unsigned fkey = (unsigned)(f + 32768);
        if (goal_m) push_m &= (goal_m & (0u - goal_m)) - 1u;                    // neighbours before the goal neighbour only
        STAMP(tb);
        bool full = false;
        while (push_m) {
            const int kk = __ffs((int)push_m) - 1;
            push_m &= push_m - 1u;
            if (len >= heap_cap) { full = true; break; }
            a2_push<C>(len, (unsigned)__builtin_amdgcn_readlane((int)fkey, kk), (unsigned)__builtin_amdgcn_readlane((int)ey, kk), kbase, pbase, gk, gp, ln, fast);
            len += 1;
            pushes += 1;
        }
        STAMP(t3);
#ifdef BL_ASTAR_STAMPS
        acc_adj += ta - t1; acc_wait += t2 - ta; acc_nb += tb - t2; acc_push += t3 - tb; acc_all += t3 - t0;
#endif
        if (full) { res.status = ASTAR_ST_CAPACITY; break; }
        if (goal_m) { res.status = ASTAR_ST_FOUND; break; }
    }
    if (duo) {                                                                  // the second wave waits at X: let it go home
        if (lane == 0) *(a2_lds_u32*)(size_t)(tbl + 4096u + 4u * A2W_RUN_WORD) = A2W_QUIT;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef BL_ASTAR_STAMPS
        for (int q = 0; q < 8; ++q) __builtin_amdgcn_s_sleep(100);           // (the second wave adds its sums on its way out)
#endif
    }
    res.pops = pops; res.pushes = pushes;
    if (res.status == ASTAR_ST_FOUND) {                                         // :107-114 -> makePath (:235-274)
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
                if (d >= 4) break;                                              // reached the start entry
                parent = cell - (d == 0 ? 1 : (d == 1 ? -1 : (d == 2 ? a.W : -a.W)));
            }
            res.path_len = (int)n;
        }
    }
#ifdef BL_ASTAR_STAMPS
    tr1 = __builtin_amdgcn_s_memrealtime();
    res.stamps[0] = (long long)acc_all; res.stamps[1] = (long long)acc_adj; res.stamps[2] = (long long)acc_nb;
    res.stamps[3] = (long long)(tr1 - tr0); res.stamps[4] = (long long)acc_wait; res.stamps[5] = (long long)acc_push;
    if (walk_ahead) {
        // walk-ahead loop (bl_astar2_ahead.h): cycles inside the barriers -- wave 0 in B1 / B2 -> [0] / [1], wave 1 in B1 / B2 -> [2] / [4];
        // walks taken again -> [5]; tops whose expansion had been made ahead | not -> path_off
        const a2_lds_u32* sc = (const a2_lds_u32*)(size_t)(tbl + 4096u);
        res.stamps[0] = sc[24]; res.stamps[1] = sc[26]; res.stamps[2] = sc[27]; res.stamps[4] = sc[28]; res.stamps[5] = sc[29];
        res.path_off = (long long)sc[30] | ((long long)sc[31] << 32);
        // (deep regime on three waves: the wave that pushes inside B1 in place of wave 2's B2; first pushes whose ancestors were read
        // again in place of the expansions not made ahead)
        if (deep3) { res.stamps[4] = sc[21]; res.path_off = (long long)sc[30] | ((long long)sc[22] << 32); }
    } else if (duo) {
        // two-wave loop: wave 0's cycles inside Y, X, Z; wave 1's inside X, Y (bl_astar2_duo.h).  Wave 1 adds its sums on its way out:
        // it has left by the time the barrier below is through... the sums it has added so far, then
        const a2_lds_u32* sc = (const a2_lds_u32*)(size_t)(tbl + 4096u);
        res.stamps[0] = sc[21]; res.stamps[1] = sc[22]; res.stamps[2] = sc[23]; res.stamps[4] = sc[25]; res.stamps[5] = sc[24];
        res.path_off = (long long)sc[30] | ((long long)sc[31] << 32);      // (diagnostic only) expansions out of registers | asked for
    } else if (turbo) {
        // the straight-line loop's own sums (table words 16 .. 21) on top: checks + top -> [0] (with the rest), pop -> [1], expansion -> [2], load wait -> [4], pushes -> [5]
        const a2_lds_u32* sc = (const a2_lds_u32*)(size_t)(tbl + 4096u);
        const long long m0 = sc[24], m1 = sc[25], m2 = sc[26], m3 = sc[27], m4 = sc[28], m5 = sc[29];
        res.stamps[0] += m0 + m1 + m2 + m3 + m4 + m5; res.stamps[1] += m2; res.stamps[4] += m3; res.stamps[2] += m4; res.stamps[5] += m5;
    }
#endif
    if (a.pool && res.status == ASTAR_ST_FOUND) {
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

#endif
