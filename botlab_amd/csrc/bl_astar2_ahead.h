// bl_astar2_ahead.h -- the search loop of k_astar2 on three (or two) wavefronts, with the NEXT pop's walk taken beside the pushes: the
// LDS regime first, its form for open lists that reach into global memory at the end of the file ("the deep regime").
// The default for searches that have their compute unit to themselves (round 6).
//
// std::pop_heap moves the hole from the root to a leaf along the smaller children (stl_heap.h __adjust_heap) BEFORE it looks at the
// value it re-inserts: the walk depends on the heap alone, not on the entry from the back of the array.  So pop k + 1's walk can be
// taken on the heap as pop k left it, while the other wavefront is still pushing expansion k's candidates -- as long as no decision
// of the walk read a position those pushes wrote.  A push writes an ancestor line: its slot and the ancestors that drop a level, up
// to the position t where the entry lands; the walk reads the two children of every node on its path; so a written position was read
// exactly when parent(t) lies on the walk's path (if a deeper written position's parent were on the path, so would parent(t) be: the
// path is closed under ancestors).  tests/tools/walk_ahead_model.py replays the reference's search on the astar fixtures with
// libstdc++'s index operations: the early walk reads a written position in 0.1 % (5e5 pops) .. 2.8 % (1e4 - 4e4 pops) of the
// iterations (a quarter of them in searches of ~1 000 pops: launch-bound anyway).  Such an iteration takes its walk again.
//
//   two waves:
//     wave 0 (POP)     B1 | test, the entry at the back, climb, stores       | B2 | walk of the NEXT pop (reads only) ....................... | B1
//     wave 1 (EXPUSH)  B1 | top, Z, "is it the top I expanded?", record      | B2 | pushes, where they landed, expansion of the top FORESEEN next | B1
//   three waves (the default): wave 1 keeps the pushes, wave 2 takes everything else of it; the candidates go from wave 2 to wave 1
//   through the record (words 48..55) before B2.
//
// B1 / B2 are workgroup barriers; Z is bl_astar2_duo.h's flag (wave 1 has read the top: wave 0 may store).  The expansion a pop needs
// is made an iteration AHEAD as well, for the top wave 1 foresees (the smaller child of the root, as bl_astar2_duo.h asks its loads
// ahead: 99.5 % of the tops here), with the popped cell's closedList entry held back until the top is seen to be that entry.  Same
// macros, same index operations in the same order as the one-wave loop (bl_astar2_turbo.h) and the two-wave loop without the early
// walk (bl_astar2_duo.h): the open list goes through the same states (the fixtures run through all of them).
//
// MEASURED (profiles/r06_astar_walk_ahead_stamps.txt, r06_astar_pop.csv; maze 2: 13 693 pops, 1.86 pushes per pop, bl_astar2_duo.h
// 0.71 - 0.72 us per pop):
//   two waves (BOTLAB_ASTAR_AHEAD=1, the forms above)                          0.705 - 0.720: both waves busy (~1 580 of ~1 750 cycles)
//   THREE waves (the default): wave 0 pops, wave 1 pushes, wave 2 expansions     0.69 as first built, 0.63 - 0.64 with wave 0 trimmed
// With three waves the pops alone are the chain (wave 0 waits ~55 cycles in either barrier -- the barrier's own cost --, the expansion
// wave ~400 + ~90), so every cycle off wave 0 counts: the four path tests as ONE vector test (lane j = push j, lane 3 = the entry
// at the back), no wait in front of B1 (the walk taken ahead only reads), the loop's checks made once per iteration, the next
// walk's first pairs asked for while the record is on its way; on wave 1 the first push's ancestor reads go out with the record
// reads (they need the length only).  What is left on wave 0, ~1 400 cycles: two LDS round trips in front of the climb (the entry
// at the back, the flag), the stores and the wait for them, the record's round trip, a walk of 2 - 3 rounds (~470).
//
// Hand-over (table words, record at tbl + 4096 + 128):
//     32..35  wave 1 -> wave 0 before B2: push mask, goal mask, popped payload, -;  word 35 on (re-)entry wave 0 -> wave 1: the length
//     36..38  wave 1 -> wave 0 before B1: per push max(landing node >> 1, 1) (1-based node index of the landing's parent; the root
//             when the entry rose to the top: every walk has it), 0x7fffffff for "no push"
//     48..55  (three waves) wave 2 -> wave 1 before B2: (key, payload) of the expansion's four candidates
//     44      run word: 1 go, 2 quit, 3 go and forget what you foresaw (wave 0 has been elsewhere; the length is in word 35), 4 park
//     45      Z
// Wave 0 leaves the loop at an iteration boundary only, behind a B1 it enters with the run word on "park": wave 1's pushes are in,
// and wave 1 goes back to B1 and waits there (as it does at the kernel's start) until wave 0 comes back or says quit.
#ifndef BL_ASTAR2_AHEAD_H
#define BL_ASTAR2_AHEAD_H

#define A2A_PARK 4u

// the rounds of a pop's walk on a heap of s40 entries (the entry at the back already taken off): round data in v200-v203 / v205-v208 /
// v240-v243, masks s[72:73] / s[74:75] / s[76:77]; s78 = 1 + the leaf the hole ends in.  Which rounds ran is a function of s40.
#define A2A_WALK(L)                                                                                           \
    "s_mov_b32 s78, 1\n\t"                                                                                    \
    A2T_ROUND("v200", "v201", "v202", "v203", "s[72:73]", "s[62:63]", "")                                     \
    "s_cmp_lt_u32 s40, s59\n\t"                                                                               \
    "s_cbranch_scc1 " L "9f\n\t"                                                                              \
    A2T_ROUND("v205", "v206", "v207", "v208", "s[74:75]", "s[64:65]", "")                                     \
    "s_cmp_ge_u32 s40, s60\n\t"                                                                               \
    "s_cbranch_scc0 " L "9f\n\t"                                                                              \
    A2T_ROUND("v240", "v241", "v242", "v243", "s[76:77]", "s[64:65]", "")                                     \
    L "9:\n\t"

// A2T_ROUND in two halves: the read of the nodes' child pairs needs the subtree's root only (s78), the rest also the length (s40) --
// the first round of a walk taken ahead asks for its pairs before the length is known (the record with the push count is on its way)
#define A2A_ROUND_ASK(N)                                                                                      \
    "v_lshl_add_u32 " N ", s78, v180, v181\n\t"                                                               \
    "v_lshl_add_u32 v220, " N ", 2, 4\n\t"                                                                    \
    "v_min_u32 v220, %[kmax], v220\n\t"                                                                       \
    "ds_read_b32 v221, v220\n\t"
#define A2A_ROUND_REST(N, C_, K, P, SQ, OK)                                                                   \
    "v_cmp_gt_u32_e64 s[68:69], s40, " N "\n\t"                                                               \
    "s_and_b64 s[68:69], s[68:69], " OK "\n\t"                                                                \
    "v_lshl_add_u32 " C_ ", " N ", 1, 1\n\t"                                                                  \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                \
    "v_cmp_le_u32_sdwa vcc, v221, v221 src0_sel:WORD_1 src1_sel:WORD_0\n\t"                                   \
    "v_min_u32_sdwa " K ", v221, v221 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n\t" \
    "s_nop 1\n\t"                                                                                             \
    "v_and_b32 v222, vcc_lo, v182\n\t"                                                                        \
    "v_addc_co_u32 " C_ ", vcc, 0, " C_ ", vcc\n\t"                                                           \
    "v_cmp_eq_u32 vcc, v222, v183\n\t"                                                                        \
    "s_and_b64 " SQ ", vcc, s[68:69]\n\t"                                                                     \
    "s_flbit_i32_b64 s70, " SQ "\n\t"                                                                         \
    "s_sub_i32 s70, 63, s70\n\t"                                                                              \
    "s_bitset0_b64 " SQ ", s70\n\t"                                                                           \
    "v_readlane_b32 s78, " N ", s70\n\t"                                                                      \
    "s_add_i32 s78, s78, 1\n\t"                                                                               \
    "v_min_u32 v220, %[pln], " C_ "\n\t"                                                                      \
    "v_lshl_add_u32 v220, v220, 2, s56\n\t"                                                                   \
    "ds_read_b32 " P ", v220\n\t"
// the walk behind a first round whose pairs have been asked for (A2A_ROUND_ASK("v200") with s78 = 1)
#define A2A_WALK_REST(L)                                                                                      \
    A2A_ROUND_REST("v200", "v201", "v202", "v203", "s[72:73]", "s[62:63]")                                    \
    "s_cmp_lt_u32 s40, s59\n\t"                                                                               \
    "s_cbranch_scc1 " L "9f\n\t"                                                                              \
    A2T_ROUND("v205", "v206", "v207", "v208", "s[74:75]", "s[64:65]", "")                                     \
    "s_cmp_ge_u32 s40, s60\n\t"                                                                               \
    "s_cbranch_scc0 " L "9f\n\t"                                                                              \
    A2T_ROUND("v240", "v241", "v242", "v243", "s[76:77]", "s[64:65]", "")                                     \
    L "9:\n\t"

// is the 1-based node A (an SGPR; 0x7fffffff: none) an ancestor-or-self of the walk's leaf s81 (1-based; s71 = its leading zeros)?
// yes -> TAKEN.  (A deeper than the leaf: A > s81 >= s81 >> anything, never equal.)
#define A2A_ON_PATH(A, TAKEN)                                                                                 \
    "s_flbit_i32_b32 s70, " A "\n\t"                                                                          \
    "s_sub_i32 s70, s70, s71\n\t"                                                                             \
    "s_lshr_b32 s70, s81, s70\n\t"                                                                            \
    "s_cmp_eq_u32 s70, " A "\n\t"                                                                             \
    "s_cbranch_scc1 " TAKEN "\n\t"

// ---------------------------------------------------------------------------------------------------------- wave 0: the pops
#define A2A_BODY_POP                                                                                          \
    "s_mov_b32 s40, %[len]\n\t"                                                                               \
    "s_mov_b32 s41, %[pops]\n\t"                                                                              \
    "s_mov_b32 s42, %[pushes]\n\t"                                                                            \
    A2W_ENTRY                                                                                                 \
    "s_mov_b32 s88, 0\n\t"                                                                                    \
    "s_mov_b32 s80, 0\n\t"                                                                                    \
    "s_mov_b32 s79, 0\n\t"                           /* no walk taken ahead */                                \
    A2W_ACC_ZERO("s36") A2W_ACC_ZERO("s38") A2W_ACC_ZERO("s86")                                               \
    "v_add_u32 v214, 4224, v191\n\t"                 /* the record */                                         \
    "v_mov_b32 v216, 3\n\t"                          /* run word: "go, and forget what you foresaw" (this wave has been elsewhere) */ \
    "v_mov_b32 v217, 1\n\t"                          /* ... "go" */                                           \
    "v_mov_b32 v219, 0\n\t"                                                                                   \
    "v_mov_b32 v225, 4\n\t"                          /* ... "park" */                                         \
    "v_mov_b32 v224, s40\n\t"                                                                                 \
    "s_mov_b64 exec, 1\n\t"                                                                                   \
    "ds_write_b32 v214, v216 offset:48\n\t"                                                                   \
    "ds_write_b32 v214, v224 offset:12\n\t"          /* the length, for the other wave */                     \
    "s_mov_b64 exec, -1\n\t"                                                                                  \
    "v_min_u32 v215, 3, v188\n\t"                                                                             \
    "v_lshl_add_u32 v215, v215, 2, v214\n\t"                                                                  \
    "v_add_u32 v215, 16, v215\n\t"                   /* the lane's word of where the pushes landed (lanes 0..2; lane 3's says "none") */ \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                \
    /* ================================================================== one iteration */                    \
    "1:\n\t"                                                                                                  \
    "s_cmp_lg_u32 s88, 0\n\t"                        /* the last expansion reached the goal */                \
    "s_cbranch_scc1 92f\n\t"                                                                                  \
    "s_cmp_ge_u32 s41, s50\n\t"                                                                               \
    "s_cbranch_scc1 93f\n\t"                                                                                  \
    "s_add_i32 s70, s40, -2\n\t"                                                                              \
    "s_cmp_gt_u32 s70, s58\n\t"                      /* len < 2 (wraps) or len - 2 > lim - 2 */               \
    "s_cbranch_scc1 91f\n\t"                                                                                  \
    /* (nothing this wave has WRITTEN is under way here: the walk taken ahead only reads) */                  \
    "2:\n\t"                                                                                                  \
    A2W_TIMED_BARRIER("s36")                     /* B1: the pushes are in; the other wave takes the top from here */ \
    "ds_read_b32 v210, v215\n\t"                     /* where they landed: lane j the j-th push */            \
    /* ---- the entry at the back of the array (key v193, payload v197): the value the pop's sift-down places */ \
    "s_lshl_b32 s70, s40, 1\n\t"                                                                              \
    "v_mov_b32 v191, s70\n\t"                                                                                 \
    "ds_read_u16 v193, v191\n\t"                                                                              \
    "s_lshl_b32 s71, s40, 2\n\t"                                                                              \
    "s_add_i32 s71, s71, s56\n\t"                                                                             \
    "s_add_i32 s71, s71, -4\n\t"                                                                              \
    "v_mov_b32 v195, s71\n\t"                                                                                 \
    "ds_read_b32 v197, v195\n\t"                                                                              \
    "s_add_i32 s40, s40, -1\n\t"                                                                              \
    /* the slot the last entry leaves is "behind the heap" from here on */                                    \
    "s_mov_b64 exec, 1\n\t"                                                                                   \
    "ds_write_b16 v191, v176\n\t"                                                                             \
    "s_mov_b64 exec, -1\n\t"                                                                                  \
    "s_cmp_eq_u32 s79, 0\n\t"                                                                                 \
    "s_cbranch_scc1 60f\n\t"                                                                                  \
    /* ---- the walk taken ahead: did it read what the pushes wrote, or the entry that has just left? */      \
    /* lanes 0..2: the parents of the pushes' landings; lane 3: the parent of the node the entry at the back held (1-based) --  */ \
    /* is any of them an ancestor-or-self of the walk's leaf s81?  a on the path <=> s81 >> (lvl(s81) - lvl(a)) == a            */ \
    "s_add_i32 s85, s40, 1\n\t"                                                                               \
    "s_lshr_b32 s85, s85, 1\n\t"                                                                              \
    "v_mov_b32 v211, s85\n\t"                                                                                 \
    "s_flbit_i32_b32 s71, s81\n\t"                                                                            \
    "s_mov_b64 s[82:83], 8\n\t"                                                                               \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                \
    "v_cndmask_b32_e64 v210, v210, v211, s[82:83]\n\t"                                                        \
    "v_ffbh_u32 v211, v210\n\t"                                                                               \
    "v_subrev_u32 v211, s71, v211\n\t"                                                                        \
    "v_lshrrev_b32_e64 v212, v211, s81\n\t"                                                                   \
    "v_cmp_eq_u32 vcc, v212, v210\n\t"                                                                        \
    "s_and_b64 s[82:83], vcc, 15\n\t"                                                                         \
    "s_cbranch_scc1 60f\n\t"                                                                                  \
    "61:\n\t"                                                                                                 \
    /* ---- the climb and one pass of stores, by the number of rounds the walk had */                         \
    "s_cmp_lt_u32 s40, s59\n\t"                                                                               \
    "s_cbranch_scc1 20f\n\t"                                                                                  \
    "s_cmp_ge_u32 s40, s60\n\t"                                                                               \
    "s_cbranch_scc1 30f\n\t"                                                                                  \
    /* two rounds */                                                                                          \
    A2T_CLIMB("v206", "v207", "s[74:75]", "25f")                                                              \
    "26:\n\t"                                                                                                 \
    A2W_GATE_ASK                                                                                              \
    A2T_ADDR("v200", "v204") A2T_ADDR("v205", "v209") A2T_LAND_ADDR                                           \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                \
    A2W_STORES_GATE("70")                                                                                     \
    A2T_STORE("v200", "v204", "v202", "v203", "s[72:73]") A2T_STORE("v205", "v209", "v207", "v208", "s[74:75]") \
    A2T_LAND_STORE                                                                                            \
    /* ---- the pop is in: the other wave pushes, this one takes the next pop's walk */                       \
    "40:\n\t"                                                                                                 \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                \
    A2W_TIMED_BARRIER("s38")                     /* B2 */                                                     \
    "ds_read_b128 v[210:213], v214\n\t"              /* the expansion's outcome: push mask, goal mask, popped payload */ \
    "s_add_i32 s41, s41, 1\n\t"                                                                               \
    "s_mov_b32 s79, 0\n\t"                                                                                    \
    "s_mov_b32 s78, 1\n\t"                                                                                    \
    A2A_ROUND_ASK("v200")                          /* (the next walk's first pairs: on their way while the record is waited for) */ \
    "s_waitcnt lgkmcnt(1)\n\t"                                                                                \
    "v_readfirstlane_b32 s87, v210\n\t"                                                                       \
    "v_readfirstlane_b32 s88, v211\n\t"                                                                       \
    "v_readfirstlane_b32 s80, v212\n\t"                                                                       \
    "s_bcnt1_i32_b32 s70, s87\n\t"                                                                            \
    "s_add_i32 s40, s40, s70\n\t"                    /* the length once the pushes are in */                  \
    "s_add_i32 s42, s42, s70\n\t"                                                                             \
    "s_cmp_lg_u32 s88, 0\n\t"                        /* the loop ends at the top of the next iteration: no walk */ \
    "s_cbranch_scc1 1b\n\t"                                                                                   \
    "s_cmp_ge_u32 s41, s50\n\t"                                                                               \
    "s_cbranch_scc1 1b\n\t"                                                                                   \
    "s_add_i32 s70, s40, -2\n\t"                                                                              \
    "s_cmp_gt_u32 s70, s58\n\t"                                                                               \
    "s_cbranch_scc1 1b\n\t"                                                                                   \
    "s_add_i32 s40, s40, -1\n\t"                     /* (the walk's heap: without the entry at the back) */   \
    A2A_WALK_REST("5")                                                                                        \
    "s_mov_b32 s81, s78\n\t"                                                                                  \
    "s_add_i32 s40, s40, 1\n\t"                                                                               \
    "s_mov_b32 s79, 1\n\t"                                                                                    \
    "s_branch 2b\n\t"                              /* (the checks of the loop's top have just been made) */ \
    /* ================================================================== out of line */                      \
    /* the walk, now: none was taken ahead, or the one taken read a position that has changed since */        \
    "60:\n\t"                                                                                                 \
    A2W_ACC_COUNT("s86")                                                                                      \
    A2A_WALK("6")                                                                                             \
    "s_branch 61b\n\t"                                                                                        \
    /* one round */                                                                                           \
    "20:\n\t"                                                                                                 \
    A2T_CLIMB("v201", "v202", "s[72:73]", "21f")                                                              \
    "22:\n\t"                                                                                                 \
    A2W_GATE_ASK                                                                                              \
    A2T_ADDR("v200", "v204") A2T_LAND_ADDR                                                                    \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                \
    A2W_STORES_GATE("75")                                                                                     \
    A2T_STORE("v200", "v204", "v202", "v203", "s[72:73]")                                                     \
    A2T_LAND_STORE                                                                                            \
    "s_branch 40b\n\t"                                                                                        \
    "21:\n\t"                                                                                                 \
    A2T_RARE_ROOT("s[72:73]", "22b")                                                                          \
    /* three rounds */                                                                                        \
    "30:\n\t"                                                                                                 \
    A2T_CLIMB("v241", "v242", "s[76:77]", "35f")                                                              \
    "36:\n\t"                                                                                                 \
    A2W_GATE_ASK                                                                                              \
    A2T_ADDR("v200", "v204") A2T_ADDR("v205", "v209") A2T_ADDR("v240", "v244") A2T_LAND_ADDR                  \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                \
    A2W_STORES_GATE("78")                                                                                     \
    A2T_STORE("v200", "v204", "v202", "v203", "s[72:73]") A2T_STORE("v205", "v209", "v207", "v208", "s[74:75]") \
    A2T_STORE("v240", "v244", "v242", "v243", "s[76:77]")                                                     \
    A2T_LAND_STORE                                                                                            \
    "s_branch 40b\n\t"                                                                                        \
    "35:\n\t"                                                                                                 \
    A2T_RARE_UP("s[76:77]", "v206", "v207", "s[74:75]", "36b", "37")                                          \
    A2T_RARE_UP("s[74:75]", "v201", "v202", "s[72:73]", "36b", "38")                                          \
    A2T_RARE_ROOT("s[72:73]", "36b")                                                                          \
    /* two rounds, the climb leaves the second */                                                             \
    "25:\n\t"                                                                                                 \
    A2T_RARE_UP("s[74:75]", "v201", "v202", "s[72:73]", "26b", "27")                                          \
    A2T_RARE_ROOT("s[72:73]", "26b")                                                                          \
    /* ---- exits: the other wave is parked behind the barrier (its pushes are in by then) */                 \
    "91:\n\t"                                                                                                 \
    "s_mov_b32 %[code], 1\n\t"                                                                                \
    "s_branch 98f\n\t"                                                                                        \
    "92:\n\t"                                                                                                 \
    "s_mov_b32 %[code], 2\n\t"                                                                                \
    "s_branch 98f\n\t"                                                                                        \
    "93:\n\t"                                                                                                 \
    "s_mov_b32 %[code], 3\n\t"                                                                                \
    "98:\n\t"                                                                                                 \
    "s_mov_b64 exec, 1\n\t"                                                                                   \
    "ds_write_b32 v214, v225 offset:48\n\t"                                                                   \
    "s_mov_b64 exec, -1\n\t"                                                                                  \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                \
    "s_barrier\n\t"                                                                                           \
    "s_branch 99f\n\t"                                                                                        \
    "94:\n\t"                                        /* the other wave's flag never came: no barrier would either */ \
    "s_mov_b32 %[code], 4\n\t"                                                                                \
    "99:\n\t"                                                                                                 \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                \
    A2W_ACC_OUT("s36", "4192") A2W_ACC_OUT("s38", "4200") A2W_ACC_OUT("s86", "4212")                          \
    "s_mov_b32 %[len], s40\n\t"                                                                               \
    "s_mov_b32 %[pops], s41\n\t"                                                                              \
    "s_mov_b32 %[pushes], s42\n\t"                                                                            \
    "s_mov_b32 %[gm], s88\n\t"                                                                                \
    "s_mov_b32 %[pt], s80\n\t"

#define A2A_POP_CLOBBERS A2T_CLOBBERS

// ---------------------------------------------------------------------------------------------------------- wave 1: expansions and pushes
// A2T_PUSH_REST, and where the entry landed: max(landing node >> 1, 1) into lane J of v167
#define A2A_PUSH_REST(J)                                                                                      \
    "s_ff1_i32_b32 s91, s87\n\t"                                                                              \
    "s_add_i32 s70, s87, -1\n\t"                                                                              \
    "s_and_b32 s87, s87, s70\n\t"                                                                             \
    "v_readlane_b32 s89, v226, s91\n\t"                                                                       \
    "s_waitcnt lgkmcnt(1)\n\t"                                                                                \
    "s_nop 1\n\t"                                                                                             \
    "v_cmp_lt_u32 vcc, s89, v234\n\t"                                                                         \
    "s_not_b64 s[92:93], vcc\n\t"                                                                             \
    "s_ff1_i32_b64 s70, s[92:93]\n\t"                                                                         \
    "s_bfm_b64 s[92:93], s70, 0\n\t"                                                                          \
    "s_lshr_b32 s70, s78, s70\n\t"                                                                            \
    "s_lshr_b32 s39, s70, 1\n\t"                                                                              \
    "s_max_u32 s39, s39, 1\n\t"                                                                               \
    "v_writelane_b32 v167, s39, " J "\n\t"                                                                    \
    "s_lshl_b32 s71, s70, 1\n\t"                                                                              \
    "s_lshl_b32 s70, s70, 2\n\t"                                                                              \
    "s_add_i32 s70, s70, s56\n\t"                                                                             \
    "s_add_i32 s70, s70, -4\n\t"                                                                              \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                \
    "v_mov_b32 v232, s71\n\t"                                                                                 \
    "v_mov_b32 v238, s70\n\t"                                                                                 \
    "s_mov_b64 exec, s[92:93]\n\t"                                                                            \
    "ds_write_b16 v236, v234\n\t"                                                                             \
    "ds_write_b32 v237, v235\n\t"                                                                             \
    "s_lshl_b64 exec, 1, s91\n\t"                    /* the entry itself: straight from the lane that holds it */ \
    "ds_write_b16 v232, v226\n\t"                                                                             \
    "ds_write_b32 v238, v227\n\t"                                                                             \
    "s_mov_b64 exec, -1\n\t"                                                                                  \
    "s_add_i32 s40, s40, 1\n\t"

// A2T_EXPAND without its store: the closedList entry of the popped cell (address v214, value v218, lane mask -> s[72:73]) is left to
// A2A_COMMIT -- the expansion below is made for a top that is only FORESEEN.
#define A2A_EXPAND_NOSTORE(TAG)                                                                               \
    "v_lshrrev_b32 v222, 3, v216\n\t"                                                                         \
    "v_cmp_ne_u32 vcc, s48, v222\n\t"           /* not closed by this search (lane 4: the popped cell itself) */ \
    "s_and_b64 s[72:73], vcc, s[96:97]\n\t"                                                                   \
    "s_andn2_b64 s[92:93], s[94:95], s[96:97]\n\t"   /* neighbour lanes inside the grid */                    \
    "s_and_b64 s[92:93], s[92:93], vcc\n\t"          /* ... and not closed */                                 \
    "v_min_u32 v223, s49, v215\n\t"                                                                           \
    "v_lshl_add_u32 v223, v223, 2, s57\n\t"                                                                   \
    "ds_read_b32 v224, v223\n\t"                     /* isValid + get_oCost by the cell's L1 distance */      \
    "v_cmp_ne_u32 vcc, 0xffff, v215\n\t"                                                                      \
    "s_and_b64 s[92:93], s[92:93], vcc\n\t"                                                                   \
    "s_and_b64 s[36:37], s[36:37], vcc\n\t"                                                                   \
    "s_and_b64 s[36:37], s[36:37], s[94:95]\n\t"                                                              \
    "s_andn2_b64 s[36:37], s[36:37], s[96:97]\n\t"   /* goal neighbours: in grid, lanes 0..3 */               \
    "v_add_u32 v217, 0xffff8000, v192\n\t"           /* fCost of the popped entry */                          \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                \
    "v_cmp_ne_u32 vcc, 0x80000000, v224\n\t"         /* the cell is valid */                                  \
    "s_and_b64 s[92:93], s[92:93], vcc\n\t"                                                                   \
    "s_and_b64 s[36:37], s[36:37], vcc\n\t"                                                                   \
    "v_add_u32 v225, v219, v224\n\t"                 /* hCost + oCost */                                      \
    "v_sub_u32 v226, v217, v225\n\t"                 /* lane 4: gCost of the popped node */                   \
    "s_nop 0\n\t"                                                                                             \
    "v_readlane_b32 s86, v226, 4\n\t"                                                                         \
    "s_add_i32 s86, s86, 0x800a\n\t"                 /* + 10 (get_gCost), + 32768 (key bias) */               \
    "v_add_u32 v226, s86, v225\n\t"                  /* key of the neighbour's entry */                       \
    "v_cmp_gt_u32 vcc, 0xffff, v226\n\t"             /* fNew < INT16_MAX (astar.cpp:103,124) */               \
    "s_and_b64 s[92:93], s[92:93], vcc\n\t"                                                                   \
    "s_mov_b32 s87, s92\n\t"                                                                                  \
    "s_mov_b32 s88, s36\n\t"                                                                                  \
    "s_cmp_eq_u32 s88, 0\n\t"                                                                                 \
    "s_cbranch_scc1 " TAG "f\n\t"                                                                             \
    "s_sub_i32 s70, 0, s88\n\t"                      /* neighbours before the goal neighbour only */          \
    "s_and_b32 s70, s70, s88\n\t"                                                                             \
    "s_add_i32 s70, s70, -1\n\t"                                                                              \
    "s_and_b32 s87, s87, s70\n\t"                                                                             \
    TAG ":\n\t"                                                                                               \
    "v_readlane_b32 s75, v212, 4\n\t"                /* the cell this expansion closes */

// the pushes of an expansion (mask s87, keys v226, payloads v227 in lanes 0..3) into a heap of s40 entries behind the pop, and where
// each landed (A2A_PUSH_REST) into the record's words 36..38 (v168 = the lane's word)
#define A2A_PUSHES                                                                                            \
    "s_add_i32 s40, s40, -1\n\t"                                                                              \
    "v_mov_b32 v167, 0x7fffffff\n\t"                                                                          \
    A2T_PUSH_CHECK("17f") A2T_PUSH_READ A2A_PUSH_REST("0")                                                    \
    A2T_PUSH_CHECK("17f") A2T_PUSH_READ A2A_PUSH_REST("1")                                                    \
    A2T_PUSH_CHECK("17f") A2T_PUSH_READ A2A_PUSH_REST("2")                                                    \
    "17:\n\t"                                                                                                 \
    "s_mov_b64 exec, 15\n\t"                                                                                  \
    "ds_write_b32 v168, v167\n\t"                                                                             \
    "s_mov_b64 exec, -1\n\t"
// (three waves) the candidates for the wave that pushes: (key, payload) of lanes 0..3 into the record's words 48..55
#define A2A_REC_CANDIDATES                                                                                    \
    "s_mov_b64 exec, 15\n\t"                                                                                  \
    "ds_write_b64 v171, v[226:227]\n\t"

// the foreseen top of the NEXT iteration (the smaller child of the root as read at B1, of equal keys the right one: payload v150,
// key v164) and the loads of its expansion, into (v160, v161)
#define A2A_FORESEE                                                                                           \
    "v_cmp_le_u32_sdwa vcc, v244, v244 src0_sel:WORD_1 src1_sel:WORD_0\n\t"                                   \
    "v_min_u32_sdwa v164, v244, v244 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n\t" \
    "s_nop 3\n\t"                                                                                             \
    "v_cndmask_b32 v150, v162, v163, vcc\n\t"                                                                 \
    A2W_NBR2_TO("v160", "v161")

// One iteration.  The expansion a pop needs is made one iteration AHEAD, for the top the wave foresees (bl_astar2_duo.h: right for
// 95 - 99.5 % of the tops), behind the pushes and beside the other wave's walk: key v192 / payload v196 of the entry it is for, the
// candidates' keys v226 / payloads v227, push mask s87, goal mask s88, and its closedList entry not stored yet (s[72:73], v214, v218,
// cell s75); s74 = "there is one".  At B1 the top really there is compared with it: the same entry (payload and key) -> the entry is
// stored and the record goes out at once; another (or none foreseen: wave 0 has been elsewhere) -> the expansion is made now, from
// loads asked for now (v158, v159).  v169 = the record's address, v168 = the lane's word of "where the pushes landed" (lanes 0..3).
#define A2A_XBODY(PREFETCH, VMWAIT, SPECWAIT, RECX, MID)                                                                           \
    "10:\n\t"                                                                                                 \
    A2W_TIMED_BARRIER("s41")                     /* B1: the heap is final */                                  \
    "ds_read_b32 v240, v169 offset:48\n\t"           /* the run word */                                       \
    "ds_read_b32 v241, v177\n\t"                     /* the top: payload, key */                              \
    "ds_read_u16 v246, v190\n\t"                                                                              \
    "ds_read_b32 v244, v165\n\t"                     /* the root's children: keys (slots 2, 3), payloads (entries 1, 2) */ \
    "ds_read_b64 v[162:163], v177 offset:4\n\t"                                                               \
    "ds_read_b32 v242, v169 offset:12\n\t"           /* the length (of use behind run word 3 only) */         \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                \
    "v_readfirstlane_b32 s70, v240\n\t"                                                                       \
    "s_cmp_eq_u32 s70, 2\n\t"                                                                                 \
    "s_cbranch_scc1 99f\n\t"                                                                                  \
    "s_cmp_eq_u32 s70, 4\n\t"                        /* parked: the other wave is elsewhere */                \
    "s_cbranch_scc1 10b\n\t"                                                                                  \
    "s_mov_b64 exec, 1\n\t"                      /* Z: the top and the children have been read -- wave 0 may store */ \
    "ds_write_b32 v169, v170 offset:52\n\t"                                                                   \
    "s_mov_b64 exec, -1\n\t"                                                                                  \
    "s_cmp_eq_u32 s70, 3\n\t"                        /* wave 0 has been elsewhere: what was foreseen is void, the length is its */ \
    "s_cbranch_scc0 14f\n\t"                                                                                  \
    "v_readfirstlane_b32 s40, v242\n\t"                                                                       \
    "s_branch 13f\n\t"                                                                                        \
    "14:\n\t"                                                                                                 \
    "s_cmp_eq_u32 s74, 0\n\t"                                                                                 \
    "s_cbranch_scc1 13f\n\t"                                                                                  \
    "v_readfirstlane_b32 s70, v241\n\t"                                                                       \
    "v_readfirstlane_b32 s71, v196\n\t"                                                                       \
    "s_cmp_eq_u32 s70, s71\n\t"                                                                               \
    "s_cbranch_scc0 13f\n\t"                                                                                  \
    "v_readfirstlane_b32 s70, v246\n\t"                                                                       \
    "v_readfirstlane_b32 s71, v192\n\t"                                                                       \
    "s_cmp_eq_u32 s70, s71\n\t"                                                                               \
    "s_cbranch_scc1 11f\n\t"                                                                                  \
    "13:\n\t"                                      /* not the foreseen top: its expansion now */              \
    A2W_ACC_COUNT("s81")                                                                                      \
    "v_mov_b32 v196, v241\n\t"                                                                                \
    "v_mov_b32 v192, v246\n\t"                                                                                \
    A2W_NBR_ADDR                                                                                              \
    "global_load_ushort v158, v213, s[52:53]\n\t"                                                             \
    "global_load_dword v159, v214, s[54:55] sc1\n\t"                                                          \
    A2A_FORESEE                                                                                               \
    PREFETCH                                                                                                  \
    A2T_FILL0                                                                                                 \
    "s_waitcnt vmcnt(" VMWAIT ")\n\t"              /* (what has just been asked for for the next top stays under way) */ \
    "v_mov_b32 v215, v158\n\t"                                                                                \
    "v_mov_b32 v216, v159\n\t"                                                                                \
    "v_cmp_eq_u32 vcc, s84, v212\n\t"                /* the cell the last expansion closed: closed */         \
    "s_nop 3\n\t"                                                                                             \
    "v_cndmask_b32 v216, v216, v166, vcc\n\t"                                                                 \
    A2A_EXPAND_NOSTORE("15")                                                                                  \
    "s_branch 12f\n\t"                                                                                        \
    "11:\n\t"                                      /* the foreseen top: its expansion has been made */        \
    A2W_ACC_COUNT("s79")                                                                                      \
    A2A_FORESEE                                                                                               \
    PREFETCH                                                                                                  \
    "12:\n\t"                                                                                                 \
    "s_mov_b64 exec, s[72:73]\n\t"                                                                            \
    "global_store_dword v214, v218, s[54:55]\n\t"    /* closedList.push_back: the first entry per cell is the one observed */ \
    "s_mov_b64 exec, -1\n\t"                                                                                  \
    "s_mov_b32 s84, s75\n\t"                                                                                  \
    "v_mov_b32 v230, s87\n\t"                                                                                 \
    "v_mov_b32 v231, s88\n\t"                                                                                 \
    "v_mov_b32 v232, v196\n\t"                                                                                \
    "s_mov_b64 exec, 1\n\t"                                                                                   \
    "ds_write_b96 v169, v[230:232]\n\t"              /* (word 35 -- the length for a re-entry -- is wave 0's) */ \
    RECX                                                                                                      \
    "s_mov_b64 exec, -1\n\t"                                                                                  \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                \
    A2W_TIMED_BARRIER("s42")                     /* B2: the pop is in */                                      \
    MID                                                                                                       \
    /* ---- the expansion of the top foreseen for the next iteration, from the loads asked for at B1 */       \
    "v_mov_b32 v196, v150\n\t"                                                                                \
    "v_mov_b32 v192, v164\n\t"                                                                                \
    "v_mov_b32 v228, v151\n\t"                                                                                \
    "v_mov_b32 v229, v152\n\t"                                                                                \
    "v_mov_b32 v210, v153\n\t"                                                                                \
    "v_mov_b32 v211, v154\n\t"                                                                                \
    "v_mov_b32 v212, v155\n\t"                                                                                \
    "v_mov_b32 v213, v156\n\t"                                                                                \
    "v_mov_b32 v214, v157\n\t"                                                                                \
    "s_mov_b64 s[94:95], s[82:83]\n\t"                                                                        \
    A2T_FILL0                                                                                                 \
    "s_waitcnt vmcnt(" SPECWAIT ")\n\t"          /* (the lines asked for ahead and the closedList entry's store may be under way) */ \
    "v_mov_b32 v215, v160\n\t"                                                                                \
    "v_mov_b32 v216, v161\n\t"                                                                                \
    "v_cmp_eq_u32 vcc, s84, v212\n\t"                /* the cell the last expansion closed: closed */         \
    "s_nop 3\n\t"                                                                                             \
    "v_cndmask_b32 v216, v216, v166, vcc\n\t"                                                                 \
    A2A_EXPAND_NOSTORE("16")                                                                                  \
    "s_mov_b32 s74, 1\n\t"                                                                                    \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                \
    "s_branch 10b\n\t"

#define A2A_BODY_EXPUSH(PREFETCH, VMWAIT, SPECWAIT)                                                                    \
    A2W_ENTRY                                                                                                 \
    "s_mov_b32 s40, 0\n\t"                                                                                    \
    "s_mov_b32 s74, 0\n\t"                           /* no expansion made ahead */                            \
    A2W_ACC_ZERO("s41") A2W_ACC_ZERO("s42") A2W_ACC_ZERO("s79") A2W_ACC_ZERO("s81")                           \
    "v_mov_b32 v190, 2\n\t"                                                                                   \
    "v_mov_b32 v165, 4\n\t"                                                                                   \
    "v_add_u32 v169, 4224, v191\n\t"                 /* the record */                                         \
    "v_min_u32 v168, 3, v188\n\t"                                                                             \
    "v_lshl_add_u32 v168, v168, 2, v169\n\t"                                                                  \
    "v_add_u32 v168, 16, v168\n\t"                   /* the lane's word of where the pushes landed (lanes 0..3) */ \
    "v_mov_b32 v166, s51\n\t"                        /* a closed entry of this search */                      \
    "v_mov_b32 v170, 1\n\t"                                                                                   \
    "s_mov_b32 s84, -1\n\t"                          /* no cell closed */                                     \
    A2A_XBODY(PREFETCH, VMWAIT, SPECWAIT, "", A2A_PUSHES)                                                     \
    "99:\n\t"                                                                                                 \
    A2W_ACC_OUT("s41", "4204") A2W_ACC_OUT("s42", "4208") A2W_ACC_OUT("s79", "4216") A2W_ACC_OUT("s81", "4220") \
    "s_waitcnt vmcnt(0)\n\t"

// ---------------------------------------------------------------------------------------------------------- three waves
// The same loop with the expansions on a wave of their own: wave 0 the pops (A2A_BODY_POP, unchanged), wave 1 the pushes, wave 2 the
// expansions (what wave 1 does above, without the pushes; the candidates go to wave 1 through the record's words 48..55).  All three
// meet at B1 and B2; wave 1 and wave 2 park at B1 together.
#define A2A_BODY_EXPAND3(PREFETCH, VMWAIT, SPECWAIT)                                                          \
    A2W_ENTRY                                                                                                 \
    "s_mov_b32 s40, 0\n\t"                                                                                    \
    "s_mov_b32 s74, 0\n\t"                           /* no expansion made ahead */                            \
    A2W_ACC_ZERO("s41") A2W_ACC_ZERO("s42") A2W_ACC_ZERO("s79") A2W_ACC_ZERO("s81")                           \
    "v_mov_b32 v190, 2\n\t"                                                                                   \
    "v_mov_b32 v165, 4\n\t"                                                                                   \
    "v_add_u32 v169, 4224, v191\n\t"                 /* the record */                                         \
    "v_min_u32 v171, 3, v188\n\t"                                                                             \
    "v_lshl_add_u32 v171, v171, 3, v169\n\t"                                                                  \
    "v_add_u32 v171, 64, v171\n\t"                   /* the lane's candidate (lanes 0..3): words 48..55 */    \
    "v_mov_b32 v166, s51\n\t"                        /* a closed entry of this search */                      \
    "v_mov_b32 v150, -1\n\t"                         /* no top foreseen */                                    \
    "v_mov_b32 v170, 1\n\t"                                                                                   \
    "s_mov_b32 s84, -1\n\t"                          /* no cell closed */                                     \
    A2A_XBODY(PREFETCH, VMWAIT, SPECWAIT, A2A_REC_CANDIDATES, "")                                             \
    "99:\n\t"                                                                                                 \
    A2W_ACC_OUT("s41", "4204") A2W_ACC_OUT("s42", "4208") A2W_ACC_OUT("s79", "4216") A2W_ACC_OUT("s81", "4220") \
    "s_waitcnt vmcnt(0)\n\t"

#define A2A_BODY_PUSH3                                                                                        \
    A2W_ENTRY                                                                                                 \
    "s_mov_b32 s40, 0\n\t"                                                                                    \
    "v_add_u32 v169, 4224, v191\n\t"                 /* the record */                                         \
    "v_min_u32 v168, 3, v188\n\t"                                                                             \
    "v_lshl_add_u32 v171, v168, 3, v169\n\t"                                                                  \
    "v_add_u32 v171, 64, v171\n\t"                   /* the lane's candidate (lanes 0..3): words 48..55 */    \
    "v_lshl_add_u32 v168, v168, 2, v169\n\t"                                                                  \
    "v_add_u32 v168, 16, v168\n\t"                   /* the lane's word of where the pushes landed */         \
    "10:\n\t"                                                                                                 \
    "s_barrier\n\t"                                  /* B1 */                                                 \
    "ds_read_b32 v240, v169 offset:48\n\t"           /* the run word */                                       \
    "ds_read_b32 v242, v169 offset:12\n\t"           /* the length (of use behind run word 3 only) */         \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                \
    "v_readfirstlane_b32 s70, v240\n\t"                                                                       \
    "s_cmp_eq_u32 s70, 2\n\t"                                                                                 \
    "s_cbranch_scc1 99f\n\t"                                                                                  \
    "s_cmp_eq_u32 s70, 4\n\t"                        /* parked: wave 0 is elsewhere */                        \
    "s_cbranch_scc1 10b\n\t"                                                                                  \
    "s_cmp_eq_u32 s70, 3\n\t"                                                                                 \
    "s_cbranch_scc0 11f\n\t"                                                                                  \
    "v_readfirstlane_b32 s40, v242\n\t"                                                                       \
    "11:\n\t"                                                                                                 \
    "s_barrier\n\t"                                  /* B2: the pop is in, the expansion's record is there */ \
    "ds_read_b32 v210, v169\n\t"                     /* the push mask */                                      \
    "ds_read_b64 v[226:227], v171\n\t"               /* the lane's candidate */                               \
    "s_add_i32 s40, s40, -1\n\t"                                                                              \
    "v_mov_b32 v167, 0x7fffffff\n\t"                                                                          \
    A2T_PUSH_READ                                  /* the first push's ancestors: they need the length only */ \
    "s_waitcnt lgkmcnt(2)\n\t"                                                                                \
    "v_readfirstlane_b32 s87, v210\n\t"                                                                       \
    A2T_PUSH_CHECK("17f") A2A_PUSH_REST("0")                                                                  \
    A2T_PUSH_CHECK("17f") A2T_PUSH_READ A2A_PUSH_REST("1")                                                    \
    A2T_PUSH_CHECK("17f") A2T_PUSH_READ A2A_PUSH_REST("2")                                                    \
    "17:\n\t"                                                                                                 \
    "s_mov_b64 exec, 15\n\t"                                                                                  \
    "ds_write_b32 v168, v167\n\t"                                                                             \
    "s_mov_b64 exec, -1\n\t"                                                                                  \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                \
    "s_branch 10b\n\t"                                                                                        \
    "99:\n\t"

// ---------------------------------------------------------------------------------------------------------- the deep regime
// The same three waves for open lists that reach into global memory (bl_astar2_deep.h: PLN + 2 <= length <= deep_max): wave 0's
// walk of the NEXT pop -- three rounds in LDS, one or two in global memory -- beside wave 1's pushes, wave 2 as above (the top and the
// root's children are LDS words in either regime).  Differences from the LDS forms:
//   * the entry at the back of the array comes from wave 1 (record words 40, 41: key, payload): out of its last push's registers,
//     read from memory only behind an expansion without pushes;
//   * neither wave waits for the ACKNOWLEDGEMENT of its global stores in front of the barrier that hands the heap over (~500 cycles
//     each): they are issued, and the other wave's loads of the same lines follow them through the same vector L1
//     (tests/tools/cross_wave_store_probe.hip: 0 stale reads in 2e8 per form, profiles/r06_cross_wave_store_probe.txt);
//   * wave 1 is idle while wave 0 finishes a pop and a push's ancestor line is a global round trip: all three lines are asked for at
//     B1, beside the pop, checked behind B2 against where the pop's value landed (record word 42: the pop wrote the nodes from the
//     root to there) and patched in registers for what the earlier pushes of the expansion wrote (A2P_*);
//   * s79 remembers where the walk taken ahead ended: 1 in LDS, 2 one global round, 3 two; record word 46 tells wave 1 the regime.
// MEASURED (profiles/r06_astar_deep_push_wave_variants.txt, r06_astar_pop.csv): wide 2 (5.3e5 pops) 1.13 -> 0.85 us per pop, convex 2
// (1.8e6) 1.15 -> 0.87; as first built (first line only, wave 1 waiting for its stores' acknowledgement) 0.89.
// Registers as A2D_BODY's (s28-s31, s34-s35, s82-s85, v150-v175); the path test's temporaries are s90-s92 and v224 here.
#define A2A_WALKD(FIRST, L)                                                                                   \
    FIRST                                                                                                     \
    A2T_ROUND("v205", "v206", "v207", "v208", "s[74:75]", "s[64:65]", "")                                     \
    A2T_ROUND("v240", "v241", "v242", "v243", "s[76:77]", "s[64:65]", "")                                     \
    A2D_ROUND2_PAYLOAD("v241", "s[76:77]")                                                                    \
    "s_mov_b32 s79, 1\n\t"                                                                                    \
    "s_mov_b32 s81, s78\n\t"                                                                                  \
    "s_cmp_ge_u32 s40, %[kslots]\n\t"                                                                         \
    "s_cbranch_scc0 " L "9f\n\t"                                                                              \
    A2D_GROUND_ASK("v155", "v156", "v159", "s[82:83]", "v[164:165]")                                          \
    "s_waitcnt vmcnt(0)\n\t"                                                                                  \
    A2D_GROUND_DECIDE("v155", "v156", "v157", "v158", "v159", "s[82:83]", "v164", "v165")                     \
    "v_readlane_b32 s78, v155, s70\n\t"                                                                       \
    "s_mov_b32 s79, 2\n\t"                                                                                    \
    "s_add_i32 s81, s78, 1\n\t"                                                                               \
    "s_lshl_b32 s71, s78, 1\n\t"                                                                              \
    "s_add_i32 s71, s71, 1\n\t"                                                                               \
    "s_cmp_lt_u32 s71, s40\n\t"                                                                               \
    "s_cselect_b32 s71, s70, 0\n\t"                                                                           \
    "s_cmp_ge_u32 s71, 31\n\t"                       /* the walk's node on the round's last level has a child: a second global round */ \
    "s_cbranch_scc0 " L "9f\n\t"                                                                              \
    "s_add_i32 s78, s78, 1\n\t"                                                                               \
    A2D_GROUND_ASK("v171", "v172", "v175", "s[34:35]", "v[166:167]")                                          \
    "s_waitcnt vmcnt(0)\n\t"                                                                                  \
    A2D_GROUND_DECIDE("v171", "v172", "v173", "v174", "v175", "s[34:35]", "v166", "v167")                     \
    "v_readlane_b32 s81, v171, s70\n\t"                                                                       \
    "s_add_i32 s81, s81, 1\n\t"                                                                               \
    "s_mov_b32 s79, 3\n\t"                                                                                    \
    L "9:\n\t"

// A2D_LAND, and where the value landed (1-based node) into the record's word 42 for wave 1: the pop wrote the nodes from the root to
// there.  The usual landing -- key slot and payload both global -- in 13 instructions instead of A2D_LAND's 21.
#define A2A_LANDD(TAG)                                                                                        \
    "s_mov_b64 exec, -1\n\t"                                                                                  \
    "s_add_i32 s70, s71, 1\n\t"                                                                               \
    "v_mov_b32 v224, s70\n\t"                                                                                 \
    "s_cmp_lt_u32 s70, %[kslots]\n\t"                                                                         \
    "s_cbranch_scc1 " TAG "1f\n\t"                                                                            \
    "s_lshl_b32 s70, s70, 1\n\t"                                                                              \
    "v_mov_b32 v220, s70\n\t"                                                                                 \
    "s_lshl_b32 s70, s71, 2\n\t"                                                                              \
    "v_mov_b32 v222, s70\n\t"                                                                                 \
    "s_mov_b64 exec, 1\n\t"                                                                                   \
    "global_store_short v220, v193, s[30:31]\n\t"                                                             \
    "global_store_dword v222, v197, s[28:29]\n\t"                                                             \
    "s_branch " TAG "2f\n\t"                                                                                  \
    TAG "1:\n\t"                                                                                              \
    A2D_LAND                                                                                                  \
    "s_mov_b64 exec, 1\n\t"                                                                                   \
    TAG "2:\n\t"                                                                                              \
    "ds_write_b32 v214, v224 offset:40\n\t"                                                                   \
    "s_mov_b64 exec, -1\n\t"

#define A2A_BODY_POPD                                                                                         \
    "s_mov_b32 s40, %[len]\n\t"                                                                               \
    "s_mov_b32 s41, %[pops]\n\t"                                                                              \
    "s_mov_b32 s42, %[pushes]\n\t"                                                                            \
    A2W_ENTRY                                                                                                 \
    "ds_read_b128 v[150:153], v191 offset:4160\n\t"                                                           \
    "ds_read_b32 v154, v191 offset:4176\n\t"                                                                  \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                \
    A2T_RSF("s30", "v150") A2T_RSF("s31", "v151") A2T_RSF("s28", "v152") A2T_RSF("s29", "v153") A2T_RSF("s58", "v154") \
    "s_mov_b32 s88, 0\n\t"                                                                                    \
    "s_mov_b32 s80, 0\n\t"                                                                                    \
    "s_mov_b32 s79, 0\n\t"                           /* no walk taken ahead */                                \
    A2W_ACC_ZERO("s94") A2W_ACC_ZERO("s95") A2W_ACC_ZERO("s86")                                               \
    "v_add_u32 v214, 4224, v191\n\t"                 /* the record */                                         \
    "v_mov_b32 v216, 3\n\t"                          /* run word: "go, and forget what you foresaw" */        \
    "v_mov_b32 v217, 1\n\t"                          /* ... "go" (and the regime word: deep) */               \
    "v_mov_b32 v219, 0\n\t"                                                                                   \
    "v_mov_b32 v225, 4\n\t"                          /* ... "park" */                                         \
    "v_mov_b32 v224, s40\n\t"                                                                                 \
    "s_mov_b64 exec, 1\n\t"                                                                                   \
    "ds_write_b32 v214, v216 offset:48\n\t"                                                                   \
    "ds_write_b32 v214, v224 offset:12\n\t"          /* the length, for the other waves */                    \
    "ds_write_b32 v214, v217 offset:56\n\t"          /* the regime, for wave 1 */                             \
    /* the entry at the back, this once by this wave (wave 1 hands over the later ones): key slot s40, payload entry s40 - 1 */ \
    "s_mov_b64 exec, -1\n\t"                                                                                  \
    "s_lshl_b32 s70, s40, 1\n\t"                                                                              \
    "v_mov_b32 v168, s70\n\t"                                                                                 \
    "s_min_u32 s71, s70, %[kmax2]\n\t"                                                                        \
    "v_mov_b32 v160, s71\n\t"                                                                                 \
    "ds_read_u16 v162, v160\n\t"                                                                              \
    "s_cmp_ge_u32 s40, %[kslots]\n\t"                                                                         \
    "s_cselect_b64 s[38:39], -1, 0\n\t"                                                                       \
    "s_mov_b64 exec, s[38:39]\n\t"                                                                            \
    "global_load_ushort v169, v168, s[30:31]\n\t"                                                             \
    "s_mov_b64 exec, -1\n\t"                                                                                  \
    "s_lshl_b32 s71, s40, 2\n\t"                                                                              \
    "s_add_i32 s71, s71, -4\n\t"                                                                              \
    "v_mov_b32 v170, s71\n\t"                                                                                 \
    "global_load_dword v163, v170, s[28:29]\n\t"                                                              \
    "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"                                                                       \
    "v_cndmask_b32_e64 v162, v162, v169, s[38:39]\n\t"                                                        \
    "s_mov_b64 exec, 1\n\t"                                                                                   \
    "ds_write_b64 v214, v[162:163] offset:32\n\t"                                                             \
    "s_mov_b64 exec, -1\n\t"                                                                                  \
    "v_min_u32 v215, 3, v188\n\t"                                                                             \
    "v_lshl_add_u32 v215, v215, 2, v214\n\t"                                                                  \
    "v_add_u32 v215, 16, v215\n\t"                   /* the lane's word of where the pushes landed */         \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                \
    /* ================================================================== one iteration */                    \
    "1:\n\t"                                                                                                  \
    "s_cmp_lg_u32 s88, 0\n\t"                        /* the last expansion reached the goal */                \
    "s_cbranch_scc1 92f\n\t"                                                                                  \
    "s_cmp_ge_u32 s41, s50\n\t"                                                                               \
    "s_cbranch_scc1 93f\n\t"                                                                                  \
    "s_sub_u32 s70, s40, %[dlo]\n\t"                                                                          \
    "s_cmp_gt_u32 s70, s58\n\t"                      /* len < PLN + 2 (wraps) or beyond the deep loop's depth */ \
    "s_cbranch_scc1 91f\n\t"                                                                                  \
    "2:\n\t"                                                                                                  \
    A2W_TIMED_BARRIER("s94")                     /* B1: the pushes are in; wave 2 takes the top from here */ \
    "ds_read_b32 v210, v215\n\t"                     /* where they landed: lane j the j-th push */            \
    "ds_read_b64 v[212:213], v214 offset:32\n\t"     /* the entry at the back: key, payload */                \
    "s_lshl_b32 s70, s40, 1\n\t"                                                                              \
    "s_min_u32 s71, s70, %[kmax2]\n\t"                                                                        \
    "v_mov_b32 v191, s71\n\t"                                                                                 \
    "s_cmp_ge_u32 s40, %[kslots]\n\t"                                                                         \
    "s_cselect_b64 s[38:39], -1, 0\n\t"              /* its key slot is global */                             \
    "s_add_i32 s40, s40, -1\n\t"                                                                              \
    /* the slot the last entry leaves is "behind the heap" from here on (an LDS slot: 0xFFFF) */              \
    "s_andn2_b64 exec, 1, s[38:39]\n\t"                                                                       \
    "ds_write_b16 v191, v176\n\t"                                                                             \
    "s_mov_b64 exec, -1\n\t"                                                                                  \
    "s_cmp_eq_u32 s79, 0\n\t"                                                                                 \
    "s_cbranch_scc1 60f\n\t"                                                                                  \
    /* ---- the walk taken ahead: did it read what the pushes wrote, or the entry that has just left?  (A2A_BODY_POP's test) */ \
    "s_add_i32 s92, s40, 1\n\t"                                                                               \
    "s_lshr_b32 s92, s92, 1\n\t"                                                                              \
    "v_mov_b32 v211, s92\n\t"                                                                                 \
    "s_flbit_i32_b32 s71, s81\n\t"                                                                            \
    "s_mov_b64 s[90:91], 8\n\t"                                                                               \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                \
    "v_cndmask_b32_e64 v210, v210, v211, s[90:91]\n\t"                                                        \
    "v_ffbh_u32 v211, v210\n\t"                                                                               \
    "v_subrev_u32 v211, s71, v211\n\t"                                                                        \
    "v_lshrrev_b32_e64 v224, v211, s81\n\t"                                                                   \
    "v_cmp_eq_u32 vcc, v224, v210\n\t"                                                                        \
    "s_and_b64 s[90:91], vcc, 15\n\t"                                                                         \
    "s_cbranch_scc1 60f\n\t"                                                                                  \
    "61:\n\t"                                                                                                 \
    "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"                                                                       \
    "v_mov_b32 v193, v212\n\t"                                                                                \
    "v_mov_b32 v197, v213\n\t"                                                                                \
    "v_cndmask_b32_e64 v243, v243, v154, s[84:85]\n\t"                                                        \
    /* ---- the climb and one pass of stores, by where the walk ended */                                      \
    "s_cmp_eq_u32 s79, 1\n\t"                                                                                 \
    "s_cbranch_scc1 30f\n\t"                                                                                  \
    "s_cmp_eq_u32 s79, 3\n\t"                                                                                 \
    "s_cbranch_scc1 50f\n\t"                                                                                  \
    /* one round in global memory */                                                                          \
    A2T_CLIMB("v156", "v157", "s[82:83]", "31f")                                                              \
    "32:\n\t"                                                                                                 \
    A2W_GATE_ASK                                                                                              \
    A2T_ADDR("v200", "v204") A2T_ADDR("v205", "v209") A2T_ADDR("v240", "v244")                                \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                \
    A2W_STORES_GATE("70")                                                                                     \
    A2T_STORE("v200", "v204", "v202", "v203", "s[72:73]") A2T_STORE("v205", "v209", "v207", "v208", "s[74:75]") \
    A2T_STORE("v240", "v244", "v242", "v243", "s[76:77]")                                                     \
    A2D_STORE3("s[82:83]")                                                                                    \
    A2A_LANDD("41")                                                                                           \
    /* ---- the pop is in (its global stores are on their way): wave 1 pushes, this one takes the next pop's walk */ \
    "40:\n\t"                                                                                                 \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                \
    A2W_TIMED_BARRIER("s95")                     /* B2 */                                                 \
    "ds_read_b128 v[210:213], v214\n\t"              /* the expansion's outcome: push mask, goal mask, popped payload */ \
    "s_add_i32 s41, s41, 1\n\t"                                                                               \
    "s_mov_b32 s79, 0\n\t"                                                                                    \
    "s_mov_b32 s78, 1\n\t"                                                                                    \
    A2A_ROUND_ASK("v200")                                                                                     \
    "s_waitcnt lgkmcnt(1)\n\t"                                                                                \
    "v_readfirstlane_b32 s87, v210\n\t"                                                                       \
    "v_readfirstlane_b32 s88, v211\n\t"                                                                       \
    "v_readfirstlane_b32 s80, v212\n\t"                                                                       \
    "s_bcnt1_i32_b32 s70, s87\n\t"                                                                            \
    "s_add_i32 s40, s40, s70\n\t"                    /* the length once the pushes are in */                  \
    "s_add_i32 s42, s42, s70\n\t"                                                                             \
    "s_cmp_lg_u32 s88, 0\n\t"                        /* the loop ends at the top of the next iteration: no walk */ \
    "s_cbranch_scc1 1b\n\t"                                                                                   \
    "s_cmp_ge_u32 s41, s50\n\t"                                                                               \
    "s_cbranch_scc1 1b\n\t"                                                                                   \
    "s_sub_u32 s70, s40, %[dlo]\n\t"                                                                          \
    "s_cmp_gt_u32 s70, s58\n\t"                                                                               \
    "s_cbranch_scc1 1b\n\t"                                                                                   \
    "s_add_i32 s40, s40, -1\n\t"                     /* (the walk's heap: without the entry at the back) */   \
    A2A_WALKD(A2A_ROUND_REST("v200", "v201", "v202", "v203", "s[72:73]", "s[62:63]"), "5")                    \
    "s_add_i32 s40, s40, 1\n\t"                                                                               \
    "s_branch 2b\n\t"                                                                                         \
    /* ================================================================== out of line */                      \
    "60:\n\t"                                                                                                 \
    A2W_ACC_COUNT("s86")                                                                                      \
    "s_mov_b32 s78, 1\n\t"                                                                                    \
    A2A_WALKD(A2T_ROUND("v200", "v201", "v202", "v203", "s[72:73]", "s[62:63]", ""), "6")                     \
    "s_branch 61b\n\t"                                                                                        \
    /* the walk ended inside LDS */                                                                           \
    "30:\n\t"                                                                                                 \
    A2T_CLIMB("v241", "v242", "s[76:77]", "35f")                                                              \
    "36:\n\t"                                                                                                 \
    A2W_GATE_ASK                                                                                              \
    A2T_ADDR("v200", "v204") A2T_ADDR("v205", "v209") A2T_ADDR("v240", "v244")                                \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                \
    A2W_STORES_GATE("75")                                                                                     \
    A2T_STORE("v200", "v204", "v202", "v203", "s[72:73]") A2T_STORE("v205", "v209", "v207", "v208", "s[74:75]") \
    A2T_STORE("v240", "v244", "v242", "v243", "s[76:77]")                                                     \
    A2A_LANDD("42")                                                                                           \
    "s_branch 40b\n\t"                                                                                        \
    /* two rounds in global memory */                                                                         \
    "50:\n\t"                                                                                                 \
    A2T_CLIMB("v172", "v173", "s[34:35]", "51f")                                                              \
    "52:\n\t"                                                                                                 \
    A2W_GATE_ASK                                                                                              \
    A2T_ADDR("v200", "v204") A2T_ADDR("v205", "v209") A2T_ADDR("v240", "v244")                                \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                \
    A2W_STORES_GATE("78")                                                                                     \
    A2T_STORE("v200", "v204", "v202", "v203", "s[72:73]") A2T_STORE("v205", "v209", "v207", "v208", "s[74:75]") \
    A2T_STORE("v240", "v244", "v242", "v243", "s[76:77]")                                                     \
    A2D_STORE3("s[82:83]")                                                                                    \
    A2D_STORE4("s[34:35]")                                                                                    \
    A2A_LANDD("43")                                                                                           \
    "s_branch 40b\n\t"                                                                                        \
    /* rare climbs */                                                                                         \
    "51:\n\t"                                                                                                 \
    A2T_RARE_UP("s[34:35]", "v156", "v157", "s[82:83]", "52b", "53")                                          \
    A2T_RARE_UP("s[82:83]", "v241", "v242", "s[76:77]", "52b", "54")                                          \
    A2T_RARE_UP("s[76:77]", "v206", "v207", "s[74:75]", "52b", "55")                                          \
    A2T_RARE_UP("s[74:75]", "v201", "v202", "s[72:73]", "52b", "56")                                          \
    A2T_RARE_ROOT("s[72:73]", "52b")                                                                          \
    "31:\n\t"                                                                                                 \
    A2T_RARE_UP("s[82:83]", "v241", "v242", "s[76:77]", "32b", "33")                                          \
    A2T_RARE_UP("s[76:77]", "v206", "v207", "s[74:75]", "32b", "34")                                          \
    A2T_RARE_UP("s[74:75]", "v201", "v202", "s[72:73]", "32b", "39")                                          \
    A2T_RARE_ROOT("s[72:73]", "32b")                                                                          \
    "35:\n\t"                                                                                                 \
    A2T_RARE_UP("s[76:77]", "v206", "v207", "s[74:75]", "36b", "37")                                          \
    A2T_RARE_UP("s[74:75]", "v201", "v202", "s[72:73]", "36b", "38")                                          \
    A2T_RARE_ROOT("s[72:73]", "36b")                                                                          \
    /* ---- exits: the other waves are parked behind the barrier (the pushes are in by then) */               \
    "91:\n\t"                                                                                                 \
    "s_mov_b32 %[code], 1\n\t"                                                                                \
    "s_branch 98f\n\t"                                                                                        \
    "92:\n\t"                                                                                                 \
    "s_mov_b32 %[code], 2\n\t"                                                                                \
    "s_branch 98f\n\t"                                                                                        \
    "93:\n\t"                                                                                                 \
    "s_mov_b32 %[code], 3\n\t"                                                                                \
    "98:\n\t"                                                                                                 \
    "s_mov_b64 exec, 1\n\t"                                                                                   \
    "ds_write_b32 v214, v225 offset:48\n\t"                                                                   \
    "ds_write_b32 v214, v219 offset:56\n\t"          /* the regime word back to "LDS" */                      \
    "s_mov_b64 exec, -1\n\t"                                                                                  \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                \
    "s_barrier\n\t"                                                                                           \
    "s_branch 99f\n\t"                                                                                        \
    "94:\n\t"                                        /* wave 2's flag never came: no barrier would either */  \
    "s_mov_b32 %[code], 4\n\t"                                                                                \
    "99:\n\t"                                                                                                 \
    "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"                                                                       \
    A2W_ACC_OUT("s94", "4192") A2W_ACC_OUT("s95", "4200") A2W_ACC_OUT("s86", "4212")                                                                       \
    "s_mov_b32 %[len], s40\n\t"                                                                               \
    "s_mov_b32 %[pops], s41\n\t"                                                                              \
    "s_mov_b32 %[pushes], s42\n\t"                                                                            \
    "s_mov_b32 %[gm], s88\n\t"                                                                                \
    "s_mov_b32 %[pt], s80\n\t"

// ---- the pushes of the deep regime out of ancestor lines read BESIDE THE POP (wave 1 idles while wave 0 finishes it; a line is a
// global round trip).  The slots the pushes go to follow from the length alone, so all three lines are asked for at B1, into a
// register set each (A = A2D_PUSH_*'s own registers).  Behind B2 a line is what memory holds now unless
//   * the pop wrote a node of it the push looks at.  The pop writes the nodes from the root to where its value landed (record word 42):
//     an ancestor-closed set, so it is enough to ask whether the ancestor that STOPS the entry lies on that path (for a patched
//     line: its parent, because a patch moves values one lane down).  Then the line is read again (A2D_PUSH_READ, as before);
//   * an earlier push of the same expansion wrote nodes of it: push i (slot s_i, d_i ancestors dropped) moved the values of the
//     heights 1 .. d_i of ITS line one level down and put its entry at height d_i; the lines of s_i and s_j are the same nodes from
//     height c = bit length of (s_i xor s_j) up.  So if d_i >= c, lanes c - 1 .. d_i - 2 of line j take their upper neighbour's
//     value (one wave shift) and lane d_i - 1 the entry of push i: A2P_PATCH, ~20 instructions where a second read is a round trip.
//   SL the slot, V231 where an ancestor would drop to, K / P its key / payload (LDS reads, then the merged values), GK / GP (global
//   loads), O236 / E165 / O166 / L237 the addresses of the drop, MK / MP the lanes whose key / payload is global
#define A2P_SET_A "s78", "v231", "v234", "v235", "v162", "v164", "v236", "v165", "v166", "v237", "s[34:35]", "s[38:39]"
#define A2P_SET_B "s79", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "s[72:73]", "s[74:75]"
#define A2P_SET_C "s86", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "s[76:77]", "s[80:81]"
#define A2P_READ(OFF, ...) A2P_READ_(OFF, __VA_ARGS__)
#define A2P_READ_(OFF, SL, V231, K, P, GK, GP, O236, E165, O166, L237, MK, MP)                                \
    "s_add_i32 " SL ", s40, " OFF "\n\t"                                                                      \
    "v_lshrrev_b32_e64 v230, v184, " SL "\n\t"       /* ancestor's slot (0: none) */                          \
    "v_lshrrev_b32_e64 " V231 ", v185, " SL "\n\t"   /* the slot it would drop to */                          \
    "v_min_u32 v232, %[kslotsm1], v230\n\t"                                                                   \
    "v_lshlrev_b32 v232, 1, v232\n\t"                                                                         \
    "ds_read_u16 " K ", v232\n\t"                                                                             \
    "v_add_u32 v233, -1, v230\n\t"                                                                            \
    "v_min_u32 v160, %[pln], v233\n\t"                                                                        \
    "v_lshl_add_u32 v160, v160, 2, s56\n\t"                                                                   \
    "ds_read_b32 " P ", v160\n\t"                                                                             \
    "v_cmp_le_u32 vcc, %[kslots], v230\n\t"                                                                   \
    "s_mov_b64 " MK ", vcc\n\t"                                                                               \
    "v_lshlrev_b32 v161, 1, v230\n\t"                                                                         \
    "s_mov_b64 exec, vcc\n\t"                                                                                 \
    "global_load_ushort " GK ", v161, s[30:31]\n\t"                                                           \
    "s_mov_b64 exec, -1\n\t"                                                                                  \
    "v_cmp_le_u32 vcc, %[pln], v233\n\t"                                                                      \
    "v_cmp_ne_u32_e64 s[68:69], 0, v230\n\t"                                                                  \
    "s_and_b64 " MP ", vcc, s[68:69]\n\t"                                                                     \
    "v_lshlrev_b32 v163, 2, v233\n\t"                                                                         \
    "s_mov_b64 exec, " MP "\n\t"                                                                              \
    "global_load_dword " GP ", v163, s[28:29]\n\t"                                                            \
    "s_mov_b64 exec, -1\n\t"                                                                                  \
    "v_lshlrev_b32 " O236 ", 1, " V231 "\n\t"                                                                 \
    "v_add_u32 " E165 ", -1, " V231 "\n\t"                                                                    \
    "v_lshlrev_b32 " O166 ", 2, " E165 "\n\t"                                                                 \
    "v_add_u32 " L237 ", s56, " O166 "\n\t"
// the two tiers of a line into K / P (everything asked for has arrived)
#define A2P_MERGE(...) A2P_MERGE_(__VA_ARGS__)
#define A2P_MERGE_(SL, V231, K, P, GK, GP, O236, E165, O166, L237, MK, MP)                                    \
    "v_cndmask_b32_e64 " K ", " K ", " GK ", " MK "\n\t"                                                      \
    "v_cndmask_b32_e64 " P ", " P ", " GP ", " MP "\n\t"
// how far the entry (key s89) rises on a merged line: s70 = the ancestors that drop, s[92:93] their lanes, s71 = the slot it takes
#define A2P_DECIDE(...) A2P_DECIDE_(__VA_ARGS__)
#define A2P_DECIDE_(SL, V231, K, P, GK, GP, O236, E165, O166, L237, MK, MP)                                   \
    "v_cmp_lt_u32 vcc, s89, " K "\n\t"                                                                        \
    "s_not_b64 s[92:93], vcc\n\t"                                                                             \
    "s_ff1_i32_b64 s70, s[92:93]\n\t"                                                                         \
    "s_bfm_b64 s[92:93], s70, 0\n\t"                                                                          \
    "s_lshr_b32 s71, " SL ", s70\n\t"
// A2D_PUSH_STORES over a set; the push's (drops, slot, key, payload) -> the four SGPRs of INFO for the patches of later lines
#define A2P_STORES(ID, IS, IK, IP, ...) A2P_STORES_(ID, IS, IK, IP, __VA_ARGS__)
#define A2P_STORES_(ID, IS, IK, IP, SL, V231, K, P, GK, GP, O236, E165, O166, L237, MK, MP)                   \
    "s_mov_b32 " ID ", s70\n\t"                                                                               \
    "s_mov_b32 " IS ", " SL "\n\t"                                                                            \
    "s_mov_b32 " IK ", s89\n\t"                                                                               \
    "s_mov_b32 " IP ", s90\n\t"                                                                               \
    "v_cmp_gt_u32 vcc, %[kslots], " V231 "\n\t"                                                               \
    "s_and_b64 exec, s[92:93], vcc\n\t"                                                                       \
    "ds_write_b16 " O236 ", " K "\n\t"                                                                        \
    "s_andn2_b64 exec, s[92:93], vcc\n\t"                                                                     \
    "global_store_short " O236 ", " K ", s[30:31]\n\t"                                                        \
    "s_mov_b64 exec, -1\n\t"                                                                                  \
    "v_cmp_gt_u32 vcc, %[pln], " E165 "\n\t"                                                                  \
    "s_and_b64 exec, s[92:93], vcc\n\t"                                                                       \
    "ds_write_b32 " L237 ", " P "\n\t"                                                                        \
    "s_andn2_b64 exec, s[92:93], vcc\n\t"                                                                     \
    "global_store_dword " O166 ", " P ", s[28:29]\n\t"                                                        \
    "s_mov_b64 exec, -1\n\t"                                                                                  \
    "s_lshl_b32 s70, s71, 1\n\t"                                                                              \
    "v_mov_b32 v232, s70\n\t"                                                                                 \
    "v_mov_b32 v233, s89\n\t"                                                                                 \
    "s_add_i32 s70, s71, -1\n\t"                                                                              \
    "s_lshl_b32 s70, s70, 2\n\t"                                                                              \
    "v_mov_b32 v238, s70\n\t"                                                                                 \
    "v_add_u32 v167, s56, v238\n\t"                                                                           \
    "v_mov_b32 v239, s90\n\t"                                                                                 \
    "s_cmp_lt_u32 s71, %[kslots]\n\t"                                                                         \
    "s_cselect_b64 s[68:69], 1, 0\n\t"                                                                        \
    "s_mov_b64 exec, s[68:69]\n\t"                                                                            \
    "ds_write_b16 v232, v233\n\t"                                                                             \
    "s_xor_b64 exec, s[68:69], 1\n\t"                                                                         \
    "global_store_short v232, v233, s[30:31]\n\t"                                                             \
    "s_add_i32 s70, s71, -1\n\t"                                                                              \
    "s_cmp_lt_u32 s70, %[pln]\n\t"                                                                            \
    "s_cselect_b64 s[68:69], 1, 0\n\t"                                                                        \
    "s_mov_b64 exec, s[68:69]\n\t"                                                                            \
    "ds_write_b32 v167, v239\n\t"                                                                             \
    "s_xor_b64 exec, s[68:69], 1\n\t"                                                                         \
    "global_store_dword v238, v239, s[28:29]\n\t"                                                             \
    "s_mov_b64 exec, -1\n\t"                                                                                  \
    "s_add_i32 s40, s40, 1\n\t"
// where the entry landed (max(landing >> 1, 1) into lane J of v172) and what the array's last slot holds now (s36 key, s37 payload: the
// entry itself if no ancestor dropped, else its parent = lane 0 of the line)
#define A2P_WHERE(J, ...) A2P_WHERE_(J, __VA_ARGS__)
#define A2P_WHERE_(J, SL, V231, K, P, GK, GP, O236, E165, O166, L237, MK, MP)                                 \
    "s_lshr_b32 s39, s71, 1\n\t"                                                                              \
    "s_max_u32 s39, s39, 1\n\t"                                                                               \
    "v_writelane_b32 v172, s39, " J "\n\t"                                                                    \
    "v_readlane_b32 s36, " K ", 0\n\t"                                                                        \
    "v_readlane_b32 s37, " P ", 0\n\t"                                                                        \
    "s_cmp_eq_u32 s71, " SL "\n\t"                                                                            \
    "s_cselect_b32 s36, s89, s36\n\t"                                                                         \
    "s_cselect_b32 s37, s90, s37\n\t"
// line SET as an earlier push (ID drops, slot IS, key IK, payload IP) left it
// (STALE: where to go when the two slots lie on different LEVELS of the heap -- the later one is, or follows, a power of two: the
// lines then share the root at different heights, which no lane shift expresses; such a line is read again.  The xor exceeds the
// earlier slot exactly then.  tests/test_push_lines_model_cpu.py found this case; no fixture ever met it with a push at the root.)
#define A2P_PATCH(TAG, STALE, ID, IS, IK, IP, ...) A2P_PATCH_(TAG, STALE, ID, IS, IK, IP, __VA_ARGS__)
#define A2P_PATCH_(TAG, STALE, ID, IS, IK, IP, SL, V231, K, P, GK, GP, O236, E165, O166, L237, MK, MP)        \
    "s_xor_b32 s82, " IS ", " SL "\n\t"                                                                       \
    "s_cmp_gt_u32 s82, " IS "\n\t"                                                                            \
    "s_cbranch_scc1 " STALE "\n\t"                                                                            \
    "s_flbit_i32_b32 s82, s82\n\t"                                                                            \
    "s_sub_i32 s82, 32, s82\n\t"                     /* c: the two lines are the same nodes from this height up */ \
    "s_cmp_ge_u32 " ID ", s82\n\t"                                                                            \
    "s_cbranch_scc0 " TAG "f\n\t"                                                                             \
    "s_add_i32 s83, s83, 1\n\t"                      /* (patched: the pop's test looks one node higher per patch) */ \
    "s_add_i32 s84, " ID ", -1\n\t"                                                                           \
    "s_bfm_b64 s[92:93], s84, 0\n\t"                 /* lanes 0 .. d - 2 */                                   \
    "s_add_i32 s82, s82, -1\n\t"                                                                              \
    "s_bfm_b64 s[68:69], s82, 0\n\t"                 /* lanes 0 .. c - 2 */                                   \
    "s_andn2_b64 s[92:93], s[92:93], s[68:69]\n\t"   /* lanes c - 1 .. d - 2: the upper neighbour's value */  \
    "s_lshl_b64 s[68:69], 1, s84\n\t"                /* lane d - 1: the earlier push's entry */               \
    "v_mov_b32 v150, " IK "\n\t"                                                                              \
    "v_mov_b32 v151, " IP "\n\t"                                                                              \
    "v_mov_b32_dpp v152, " K " wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"                                     \
    "v_mov_b32_dpp v153, " P " wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"                                     \
    "s_nop 1\n\t"                                                                                             \
    "v_cndmask_b32_e64 " K ", " K ", v152, s[92:93]\n\t"                                                      \
    "v_cndmask_b32_e64 " P ", " P ", v153, s[92:93]\n\t"                                                      \
    "v_cndmask_b32_e64 " K ", " K ", v150, s[68:69]\n\t"                                                      \
    "v_cndmask_b32_e64 " P ", " P ", v151, s[68:69]\n\t"                                                      \
    "s_nop 1\n\t"                                                                                             \
    TAG ":\n\t"
// did the pop (its value landed on the 1-based node s85) write the ancestor that stopped the entry (node s71 >> 1; for a line
// patched s83 times the node s83 levels above it: a patch moves values one lane down; none: the entry rose to the root)?  yes -> STALE
#define A2P_TEST(STALE)                                                                                       \
    "s_add_i32 s84, s83, 1\n\t"                                                                               \
    "s_lshr_b32 s84, s71, s84\n\t"                                                                            \
    "s_cmp_eq_u32 s84, 0\n\t"                                                                                 \
    "s_cbranch_scc1 " STALE "\n\t"                                                                            \
    "s_flbit_i32_b32 s82, s84\n\t"                                                                            \
    "s_flbit_i32_b32 s39, s85\n\t"                                                                            \
    "s_sub_i32 s82, s82, s39\n\t"                                                                             \
    "s_lshr_b32 s82, s85, s82\n\t"                   /* (a node deeper than the landing is larger than it: never equal) */ \
    "s_cmp_eq_u32 s82, s84\n\t"                                                                               \
    "s_cbranch_scc1 " STALE "\n\t"

// wave 1 for both regimes: the LDS pushes of A2A_BODY_PUSH3, or (regime word 1) the deep regime's (A2P_* above) with where they landed, and then
// the entry at the back of the array for wave 0's next pop.  s61 = the regime; v172 landings, v173 record, v174 / v175 the lane's
// landing word / candidate.
#define A2A_PUSH_REST_L(J)                                                                                    \
    "s_ff1_i32_b32 s91, s87\n\t"                                                                              \
    "s_add_i32 s70, s87, -1\n\t"                                                                              \
    "s_and_b32 s87, s87, s70\n\t"                                                                             \
    "v_readlane_b32 s89, v226, s91\n\t"                                                                       \
    "s_waitcnt lgkmcnt(1)\n\t"                                                                                \
    "s_nop 1\n\t"                                                                                             \
    "v_cmp_lt_u32 vcc, s89, v234\n\t"                                                                         \
    "s_not_b64 s[92:93], vcc\n\t"                                                                             \
    "s_ff1_i32_b64 s70, s[92:93]\n\t"                                                                         \
    "s_bfm_b64 s[92:93], s70, 0\n\t"                                                                          \
    "s_lshr_b32 s70, s78, s70\n\t"                                                                            \
    "s_lshr_b32 s39, s70, 1\n\t"                                                                              \
    "s_max_u32 s39, s39, 1\n\t"                                                                               \
    "v_writelane_b32 v172, s39, " J "\n\t"                                                                    \
    "s_lshl_b32 s71, s70, 1\n\t"                                                                              \
    "s_lshl_b32 s70, s70, 2\n\t"                                                                              \
    "s_add_i32 s70, s70, s56\n\t"                                                                             \
    "s_add_i32 s70, s70, -4\n\t"                                                                              \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                \
    "v_mov_b32 v232, s71\n\t"                                                                                 \
    "v_mov_b32 v238, s70\n\t"                                                                                 \
    "s_mov_b64 exec, s[92:93]\n\t"                                                                            \
    "ds_write_b16 v236, v234\n\t"                                                                             \
    "ds_write_b32 v237, v235\n\t"                                                                             \
    "s_lshl_b64 exec, 1, s91\n\t"                                                                             \
    "ds_write_b16 v232, v226\n\t"                                                                             \
    "ds_write_b32 v238, v227\n\t"                                                                             \
    "s_mov_b64 exec, -1\n\t"                                                                                  \
    "s_add_i32 s40, s40, 1\n\t"
#define A2A_BODY_PUSH3D                                                                                       \
    A2W_ENTRY                                                                                                 \
    "ds_read_b128 v[150:153], v191 offset:4160\n\t"                                                           \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                \
    A2T_RSF("s30", "v150") A2T_RSF("s31", "v151") A2T_RSF("s28", "v152") A2T_RSF("s29", "v153")               \
    "s_mov_b32 s40, 0\n\t"                                                                                    \
    "s_mov_b32 s61, 0\n\t"                                                                                    \
    A2W_ACC_ZERO("s94") A2W_ACC_ZERO("s95")                                                                                       \
    "v_add_u32 v173, 4224, v191\n\t"                 /* the record */                                         \
    "v_min_u32 v174, 3, v188\n\t"                                                                             \
    "v_lshl_add_u32 v175, v174, 3, v173\n\t"                                                                  \
    "v_add_u32 v175, 64, v175\n\t"                   /* the lane's candidate (lanes 0..3): words 48..55 */    \
    "v_lshl_add_u32 v174, v174, 2, v173\n\t"                                                                  \
    "v_add_u32 v174, 16, v174\n\t"                   /* the lane's word of where the pushes landed */         \
    "10:\n\t"                                                                                                 \
    A2W_TIMED_BARRIER("s94")                     /* B1 */                                                 \
    "ds_read_b32 v240, v173 offset:48\n\t"           /* the run word */                                       \
    "ds_read_b32 v242, v173 offset:12\n\t"           /* the length, the regime (of use behind run word 3 only) */ \
    "ds_read_b32 v241, v173 offset:56\n\t"                                                                    \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                \
    "v_readfirstlane_b32 s70, v240\n\t"                                                                       \
    "s_cmp_eq_u32 s70, 2\n\t"                                                                                 \
    "s_cbranch_scc1 99f\n\t"                                                                                  \
    "s_cmp_eq_u32 s70, 4\n\t"                        /* parked: wave 0 is elsewhere */                        \
    "s_cbranch_scc1 10b\n\t"                                                                                  \
    "s_cmp_eq_u32 s70, 3\n\t"                                                                                 \
    "s_cbranch_scc0 11f\n\t"                                                                                  \
    "v_readfirstlane_b32 s40, v242\n\t"                                                                       \
    "v_readfirstlane_b32 s61, v241\n\t"                                                                       \
    "11:\n\t"                                                                                                 \
    "s_add_i32 s40, s40, -1\n\t"                                                                              \
    "s_cmp_lg_u32 s61, 0\n\t"                                                                                 \
    "s_cbranch_scc0 12f\n\t"                                                                                  \
    /* deep regime: the pushes' ancestor lines (a global round trip each) are asked for NOW, beside the pop (A2P_*) */ \
    A2P_READ("1", A2P_SET_A) A2P_READ("2", A2P_SET_B) A2P_READ("3", A2P_SET_C)                                 \
    "12:\n\t"                                                                                                 \
    "s_barrier\n\t"                                  /* B2: the pop is in, the expansion's record is there */ \
    "ds_read_b32 v210, v173\n\t"                     /* the push mask */                                      \
    "ds_read_b64 v[226:227], v175\n\t"               /* the lane's candidate */                               \
    "v_mov_b32 v172, 0x7fffffff\n\t"                                                                          \
    "s_cmp_lg_u32 s61, 0\n\t"                                                                                 \
    "s_cbranch_scc1 20f\n\t"                                                                                  \
    A2T_PUSH_READ                                  /* the first push's ancestors: they need the length only */ \
    "s_waitcnt lgkmcnt(2)\n\t"                                                                                \
    "v_readfirstlane_b32 s87, v210\n\t"                                                                       \
    A2T_PUSH_CHECK("17f") A2A_PUSH_REST_L("0")                                                                \
    A2T_PUSH_CHECK("17f") A2T_PUSH_READ A2A_PUSH_REST_L("1")                                                  \
    A2T_PUSH_CHECK("17f") A2T_PUSH_READ A2A_PUSH_REST_L("2")                                                  \
    "17:\n\t"                                                                                                 \
    "s_mov_b64 exec, 15\n\t"                                                                                  \
    "ds_write_b32 v174, v172\n\t"                                                                             \
    "s_mov_b64 exec, -1\n\t"                                                                                  \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                \
    "s_branch 10b\n\t"                                                                                        \
    /* ---- the deep regime */                                                                                \
    "20:\n\t"                                                                                                 \
    "ds_read_b32 v241, v173 offset:40\n\t"           /* where the pop's value landed (1-based node) */        \
    "s_mov_b32 s36, -1\n\t"                          /* no push made */                                       \
    "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"              /* (what was asked for in front of B2 arrived long ago) */ \
    "v_readfirstlane_b32 s87, v210\n\t"                                                                       \
    "v_readfirstlane_b32 s85, v241\n\t"                                                                       \
    A2P_MERGE(A2P_SET_A) A2P_MERGE(A2P_SET_B) A2P_MERGE(A2P_SET_C)                                            \
    /* ---- the first push: line A */                                                                         \
    A2T_PUSH_CHECK("27f") A2D_PUSH_PICK                                                                       \
    "s_mov_b32 s83, 0\n\t"                                                                                    \
    A2P_DECIDE(A2P_SET_A)                                                                                     \
    A2P_TEST("21f")                                                                                           \
    "s_branch 22f\n\t"                                                                                        \
    "21:\n\t"                                        /* the pop wrote what it looked at: read again */        \
    A2W_ACC_COUNT("s95")                                                                                      \
    A2D_PUSH_READ A2D_PUSH_DECIDE                                                                             \
    "22:\n\t"                                                                                                 \
    A2P_STORES("s72", "s73", "s74", "s75", A2P_SET_A) A2P_WHERE("0", A2P_SET_A)                               \
    /* ---- the second: line B as the first push left it */                                                   \
    A2T_PUSH_CHECK("27f") A2D_PUSH_PICK                                                                       \
    "s_mov_b32 s83, 0\n\t"                                                                                    \
    A2P_PATCH("231", "23f", "s72", "s73", "s74", "s75", A2P_SET_B)                                                   \
    A2P_DECIDE(A2P_SET_B)                                                                                     \
    A2P_TEST("23f")                                                                                           \
    A2P_STORES("s76", "s77", "s80", "s81", A2P_SET_B) A2P_WHERE("1", A2P_SET_B)                               \
    "s_branch 24f\n\t"                                                                                        \
    "23:\n\t"                                                                                                 \
    A2W_ACC_COUNT("s95")                                                                                      \
    A2D_PUSH_READ A2D_PUSH_DECIDE                                                                             \
    A2P_STORES("s76", "s77", "s80", "s81", A2P_SET_A) A2P_WHERE("1", A2P_SET_A)                               \
    "24:\n\t"                                                                                                 \
    /* ---- the third: line C as the first two left it */                                                     \
    A2T_PUSH_CHECK("27f") A2D_PUSH_PICK                                                                       \
    "s_mov_b32 s83, 0\n\t"                                                                                    \
    A2P_PATCH("251", "25f", "s72", "s73", "s74", "s75", A2P_SET_C)                                                   \
    A2P_PATCH("252", "25f", "s76", "s77", "s80", "s81", A2P_SET_C)                                                   \
    A2P_DECIDE(A2P_SET_C)                                                                                     \
    A2P_TEST("25f")                                                                                           \
    A2P_STORES("s72", "s73", "s74", "s75", A2P_SET_C) A2P_WHERE("2", A2P_SET_C)                               \
    "s_branch 27f\n\t"                                                                                        \
    "25:\n\t"                                                                                                 \
    A2W_ACC_COUNT("s95")                                                                                      \
    A2D_PUSH_READ A2D_PUSH_DECIDE                                                                             \
    A2P_STORES("s72", "s73", "s74", "s75", A2P_SET_A) A2P_WHERE("2", A2P_SET_A)                               \
    "27:\n\t"                                                                                                 \
    "s_cmp_eq_u32 s36, -1\n\t"                                                                                \
    "s_cbranch_scc1 26f\n\t"                                                                                  \
    /* the entry at the back of the array as the pushes leave it: out of the last push's registers */         \
    "v_mov_b32 v162, s36\n\t"                                                                                 \
    "v_mov_b32 v163, s37\n\t"                                                                                 \
    /* (this wave's global stores are ISSUED in front of B1, as wave 0's are in front of B2: wave 0's loads follow them through the same L1) */ \
    "s_branch 28f\n\t"                                                                                        \
    "26:\n\t"                                                                                                 \
    /* no push: read it -- key slot s40 (LDS or global), payload entry s40 - 1 (global) */                    \
    "s_lshl_b32 s70, s40, 1\n\t"                                                                              \
    "v_mov_b32 v168, s70\n\t"                                                                                 \
    "s_min_u32 s71, s70, %[kmax2]\n\t"                                                                        \
    "v_mov_b32 v160, s71\n\t"                                                                                 \
    "ds_read_u16 v162, v160\n\t"                                                                              \
    "s_cmp_ge_u32 s40, %[kslots]\n\t"                                                                         \
    "s_cselect_b64 s[38:39], -1, 0\n\t"                                                                       \
    "s_mov_b64 exec, s[38:39]\n\t"                                                                            \
    "global_load_ushort v169, v168, s[30:31]\n\t"                                                             \
    "s_mov_b64 exec, -1\n\t"                                                                                  \
    "s_lshl_b32 s71, s40, 2\n\t"                                                                              \
    "s_add_i32 s71, s71, -4\n\t"                                                                              \
    "v_mov_b32 v170, s71\n\t"                                                                                 \
    "global_load_dword v163, v170, s[28:29]\n\t"                                                              \
    "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"              /* (this wave's global stores are acknowledged with it) */ \
    "v_cndmask_b32_e64 v162, v162, v169, s[38:39]\n\t"                                                        \
    "28:\n\t"                                                                                                 \
    "s_mov_b64 exec, 1\n\t"                                                                                   \
    "ds_write_b64 v173, v[162:163] offset:32\n\t"                                                             \
    "s_mov_b64 exec, 15\n\t"                                                                                  \
    "ds_write_b32 v174, v172\n\t"                                                                             \
    "s_mov_b64 exec, -1\n\t"                                                                                  \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                \
    "s_branch 10b\n\t"                                                                                        \
    "99:\n\t"                                                                                                 \
    A2W_ACC_OUT("s94", "4180") A2W_ACC_OUT("s95", "4184")

#define A2A_PUSH3D_CLOBBERS A2D_CLOBBERS

#define A2A_PUSH3_CLOBBERS A2T_CLOBBERS, "v167", "v168", "v169", "v171"

#define A2A_EXPUSH_CLOBBERS A2W_EXPAND_CLOBBERS, "v164", "v167", "v168", "v169", "v170", "v171", "v246"

#endif
