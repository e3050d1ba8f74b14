// bl_astar2_duo.h -- the LDS-regime search loop of k_astar2 (bl_astar2_turbo.h) split over TWO wavefronts of the workgroup.
//
// A lone wavefront issues one instruction per ~4.2 cycles (the CU's arbiter visits a SIMD every fourth cycle), and the straight-line
// loop is ~460 instructions per pop: it waits for its own instruction stream, not for memory.  Two waves of a workgroup sit on
// different SIMDs and each issues at that rate (tests/tools/two_wave_probe.hip: 4.4 cycles per instruction each; an s_barrier both
// reach ~10 cycles; LDS write -> barrier -> read ~150).  So:
//     wave 0 (HEAP)    the open list: the pop's walk, climb and stores, then the pushes -- everything that touches the heap
//     wave 1 (EXPAND)  the expansion of the popped node: neighbour loads, closedList entry, costs, the candidates' keys and payloads
//                      (astar.cpp:95-135, 213-233) -- it needs the popped top only, which is known before the pop begins
// Same macros, same order of heap operations, same stores as the one-wave loop: the open list goes through the same states.
//
// Hand-over, per iteration two barriers and a flag word:
//     X   (barrier) the heap is final: wave 0 has finished the previous iteration's pushes   -> wave 1 reads the top (payload, key)
//     Z   (flag)    wave 1 raises it when it HAS read the top; wave 0 looks at it in front of its stores into the heap -- ~800 cycles
//                   behind X, so it never waits, but the order is the flag's, not the clock's -- and lowers it again.  (A barrier here
//                   holds BOTH waves: wave 1 stood in it for the whole walk.)
//     Y   (barrier) wave 1's record is in LDS: push mask, goal mask, the popped payload, (key, payload) of up to four candidates
//                                                                                           -> wave 0 pushes
// Wave 1 never leaves its loop while the kernel runs: whenever wave 0 is outside this loop (the general iteration, the deep loop)
// wave 1 waits at X; on its way back in wave 0 sets the run word to 3 ("go, and forget what you foresaw"), behind its first gate
// to 1 ("go").  At the end wave 0 stores QUIT and passes X once more.
// Record: table words 32..35 = push mask, goal mask, popped payload, -; words 36..43 = (key, payload) x 4; word 44 = run word;
// word 45 = the flag.
#ifndef BL_ASTAR2_DUO_H
#define BL_ASTAR2_DUO_H

#define A2W_REC_OFF (4096 + 128)
#define A2W_LIST_OFF (4096 + 144)
#define A2W_RUN_WORD 44
#define A2W_GO 1u
#define A2W_QUIT 2u

// Diagnostic build (-DBL_ASTAR_STAMPS): cycles inside barriers and between marks (ACC an SGPR the loops leave alone), added into table
// words at the loop's exit.  Wave 0: word 24 = inside Y (waiting for the record); wave 1: words 21 / 22 / 23 / 25 = from the flag to the
// wait for its loads / the wait / the expansion / the record, 27 / 28 = inside X / Y, 30 / 31 = tops that were the foreseen one / not.
#ifdef BL_ASTAR_STAMPS
#define A2W_TIMED_BARRIER(ACC) "s_memtime s[100:101]\n\ts_waitcnt lgkmcnt(0)\n\ts_mov_b32 s43, s100\n\ts_barrier\n\ts_memtime s[100:101]\n\ts_waitcnt lgkmcnt(0)\n\ts_sub_u32 s43, s100, s43\n\ts_add_u32 " ACC ", " ACC ", s43\n\t"
#define A2W_ACC_OUT(ACC, OFF) "v_mov_b32 v245, %[tbl]\n\tv_mov_b32 v250, " ACC "\n\ts_mov_b64 exec, 1\n\tds_add_u32 v245, v250 offset:" OFF "\n\ts_mov_b64 exec, -1\n\ts_waitcnt lgkmcnt(0)\n\t"
#define A2W_ACC_ZERO(ACC) "s_mov_b32 " ACC ", 0\n\t"
#define A2W_ACC_COUNT(ACC) "s_add_u32 " ACC ", " ACC ", 1\n\t"
#define A2W_T0 "s_memtime s[100:101]\n\ts_waitcnt lgkmcnt(0)\n\ts_mov_b32 s43, s100\n\t"
#define A2W_T1(ACC) "s_memtime s[100:101]\n\ts_waitcnt lgkmcnt(0)\n\ts_sub_u32 s43, s100, s43\n\ts_add_u32 " ACC ", " ACC ", s43\n\ts_mov_b32 s43, s100\n\t"
#else
#define A2W_TIMED_BARRIER(ACC) "s_barrier\n\t"
#define A2W_ACC_OUT(ACC, OFF) ""
#define A2W_ACC_ZERO(ACC) ""
#define A2W_ACC_COUNT(ACC) ""
#define A2W_T0 ""
#define A2W_T1(ACC) ""
#endif

// constants of either wave (the one-wave loop's entry)
#define A2W_ENTRY                                                                                             \
    "v_mbcnt_lo_u32_b32 v188, -1, 0\n\t"                                                                      \
    "v_mbcnt_hi_u32_b32 v188, -1, v188\n\t"                                                                   \
    "v_lshlrev_b32 v190, 6, v188\n\t"                                                                         \
    "v_add_u32 v190, %[tbl], v190\n\t"                                                                        \
    "v_mov_b32 v191, %[tbl]\n\t"                                                                              \
    "ds_read_b128 v[180:183], v190\n\t"                                                                       \
    "ds_read_b128 v[184:187], v190 offset:16\n\t"                                                             \
    "ds_read_b64 v[178:179], v190 offset:32\n\t"                                                              \
    "ds_read_b128 v[192:195], v191 offset:4096\n\t"                                                           \
    "ds_read_b128 v[196:199], v191 offset:4112\n\t"                                                           \
    "ds_read_b128 v[200:203], v191 offset:4128\n\t"                                                           \
    "ds_read_b128 v[204:207], v191 offset:4144\n\t"                                                           \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                \
    A2T_RSF("s44", "v192") A2T_RSF("s45", "v193") A2T_RSF("s46", "v194") A2T_RSF("s47", "v195")               \
    A2T_RSF("s48", "v196") A2T_RSF("s49", "v197") A2T_RSF("s50", "v198") A2T_RSF("s58", "v199")               \
    A2T_RSF("s52", "v200") A2T_RSF("s53", "v201") A2T_RSF("s54", "v202") A2T_RSF("s55", "v203")               \
    A2T_RSF("s56", "v204") A2T_RSF("s57", "v205") A2T_RSF("s59", "v206") A2T_RSF("s60", "v207")               \
    "s_mov_b32 s62, %[ok0lo]\n\t"                                                                             \
    "s_mov_b32 s63, %[ok0hi]\n\t"                                                                             \
    "s_mov_b32 s64, -1\n\t"                                                                                   \
    "s_mov_b32 s65, 0x7fffffff\n\t"                                                                           \
    "s_mov_b64 s[66:67], 31\n\t"                                                                              \
    "s_mov_b64 s[96:97], 16\n\t"                                                                              \
    "s_mov_b64 s[98:99], 0xff\n\t"                                                                            \
    "s_lshl_b32 s51, s48, 3\n\t"                                                                              \
    "v_mov_b32 v177, s56\n\t"                                                                                 \
    "v_mov_b32 v176, 0xffff\n\t"

// ---------------------------------------------------------------------------------------------------------- wave 0: the heap
// Z is not a barrier (wave 1 would stand in it until wave 0 has finished its walk): wave 1 raises the flag word when it has read the
// top, wave 0 looks at it in front of its stores (the read goes out ahead of the address arithmetic; the flag has been up for
// hundreds of cycles by then) and lowers it again; the run word turns to "go" at the same place.
#define A2W_GATE_ASK "ds_read_b32 v218, v214 offset:52\n\t"
#define A2W_STORES_GATE(TAG)                                                                                  \
    "s_mov_b32 s39, 0\n\t"                                                                                    \
    TAG ":\n\t"                                                                                               \
    "v_readfirstlane_b32 s70, v218\n\t"                                                                       \
    "s_cmp_lg_u32 s70, 0\n\t"                                                                                 \
    "s_cbranch_scc1 " TAG "1f\n\t"                                                                            \
    "s_add_u32 s39, s39, 1\n\t"                      /* (never seen: the flag goes up a few instructions behind X.  A wave that */ \
    "s_cmp_gt_u32 s39, 0x400000\n\t"                 /* does not come back from a loop would hang the device: give up instead, */ \
    "s_cbranch_scc1 94f\n\t"                         /* ~1 s, the search ends as BROKEN) */                   \
    "ds_read_b32 v218, v214 offset:52\n\t"                                                                    \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                \
    "s_branch " TAG "b\n\t"                                                                                   \
    TAG "1:\n\t"                                                                                              \
    "s_mov_b64 exec, 1\n\t"                                                                                   \
    "ds_write_b32 v214, v219 offset:52\n\t"                                                                   \
    "ds_write_b32 v214, v217 offset:48\n\t"                                                                   \
    "s_mov_b64 exec, -1\n\t"
#define A2W_BODY_HEAP                                                                                         \
    "s_mov_b32 s40, %[len]\n\t"                                                                               \
    "s_mov_b32 s41, %[pops]\n\t"                                                                              \
    "s_mov_b32 s42, %[pushes]\n\t"                                                                            \
    A2W_ENTRY                                                                                                 \
    "s_mov_b32 s88, 0\n\t"                                                                                    \
    "s_mov_b32 s80, 0\n\t"                                                                                    \
    A2W_ACC_ZERO("s36") A2W_ACC_ZERO("s37") A2W_ACC_ZERO("s38")                                               \
    "v_add_u32 v214, 4224, v191\n\t"                 /* the record */                                         \
    "v_min_u32 v215, 3, v188\n\t"                                                                             \
    "v_lshl_add_u32 v215, v215, 3, v214\n\t"                                                                  \
    "v_add_u32 v215, 16, v215\n\t"                   /* the lane's candidate */                               \
    "v_mov_b32 v216, 3\n\t"                          /* run word: "go, and forget what you foresaw" (this wave has been elsewhere) */ \
    "v_mov_b32 v217, 1\n\t"                          /* ... "go" */                                           \
    "v_mov_b32 v219, 0\n\t"                                                                                   \
    "s_mov_b64 exec, 1\n\t"                                                                                   \
    "ds_write_b32 v214, v216 offset:48\n\t"                                                                   \
    "s_mov_b64 exec, -1\n\t"                                                                                  \
    /* ================================================================== one iteration */                    \
    "1:\n\t"                                                                                                  \
    "s_cmp_ge_u32 s41, s50\n\t"                                                                               \
    "s_cbranch_scc1 93f\n\t"                                                                                  \
    "s_add_i32 s70, s40, -2\n\t"                                                                              \
    "s_cmp_gt_u32 s70, s58\n\t"                 /* len < 2 (wraps) or len - 2 > lim - 2 */                    \
    "s_cbranch_scc1 91f\n\t"                                                                                  \
    "s_waitcnt lgkmcnt(0)\n\t"                     /* (the pushes' stores have been executed) */              \
    A2W_TIMED_BARRIER("s37")                     /* X: the other wave takes the top from here */              \
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
    /* ---- openList.pop(): rounds, the climb, one pass of stores */                                          \
    "s_mov_b32 s78, 1\n\t"                                                                                    \
    A2T_ROUND("v200", "v201", "v202", "v203", "s[72:73]", "s[62:63]", "")                                     \
    "s_cmp_lt_u32 s40, s59\n\t"                                                                               \
    "s_cbranch_scc1 20f\n\t"                                                                                  \
    A2T_ROUND("v205", "v206", "v207", "v208", "s[74:75]", "s[64:65]", "")                                     \
    "s_cmp_ge_u32 s40, s60\n\t"                                                                               \
    "s_cbranch_scc1 30f\n\t"                                                                                  \
    /* two rounds */                                                                                          \
    A2T_CLIMB("v206", "v207", "s[74:75]", "25f")                                                              \
    "26:\n\t"                                                                                                 \
    A2W_GATE_ASK                                                                                              \
    A2T_ADDR("v200", "v204") A2T_ADDR("v205", "v209") A2T_LAND_ADDR                                           \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                \
    A2W_STORES_GATE("70")                                                                                      \
    A2T_STORE("v200", "v204", "v202", "v203", "s[72:73]") A2T_STORE("v205", "v209", "v207", "v208", "s[74:75]") \
    A2T_LAND_STORE                                                                                            \
    /* ---- the pushes of the expansion the other wave has made */                                            \
    "40:\n\t"                                                                                                 \
    A2T_PUSH_READ                               /* (the first push's ancestors: on their way while the record is waited for) */ \
    "s_waitcnt lgkmcnt(2)\n\t"                     /* (the pop's stores have been executed: only the two reads above may be under way) */ \
    A2W_TIMED_BARRIER("s36")                     /* Y: the record is there; the other wave foresees the next top from the repaired root */ \
    "ds_read_b128 v[210:213], v214\n\t"                                                                       \
    "ds_read_b64 v[226:227], v215\n\t"                                                                        \
    "s_add_i32 s41, s41, 1\n\t"                                                                               \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                \
    "v_readfirstlane_b32 s87, v210\n\t"                                                                       \
    "v_readfirstlane_b32 s88, v211\n\t"                                                                       \
    "v_readfirstlane_b32 s80, v212\n\t"                                                                       \
    A2T_PUSH_CHECK("50f") A2T_PUSH_REST                                                                       \
    A2T_PUSH_CHECK("50f") A2T_PUSH_READ A2T_PUSH_REST                                                         \
    A2T_PUSH_CHECK("50f") A2T_PUSH_READ A2T_PUSH_REST                                                         \
    "50:\n\t"                                                                                                 \
    "s_cmp_lg_u32 s88, 0\n\t"                                                                                 \
    "s_cbranch_scc1 92f\n\t"                                                                                  \
    "s_branch 1b\n\t"                                                                                         \
    /* ================================================================== out of line */                      \
    /* one round */                                                                                           \
    "20:\n\t"                                                                                                 \
    A2T_CLIMB("v201", "v202", "s[72:73]", "21f")                                                              \
    "22:\n\t"                                                                                                 \
    A2W_GATE_ASK                                                                                              \
    A2T_ADDR("v200", "v204") A2T_LAND_ADDR                                                                    \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                \
    A2W_STORES_GATE("75")                                                                                      \
    A2T_STORE("v200", "v204", "v202", "v203", "s[72:73]")                                                     \
    A2T_LAND_STORE                                                                                            \
    "s_branch 40b\n\t"                                                                                        \
    "21:\n\t"                                                                                                 \
    A2T_RARE_ROOT("s[72:73]", "22b")                                                                          \
    /* three rounds */                                                                                        \
    "30:\n\t"                                                                                                 \
    A2T_ROUND("v240", "v241", "v242", "v243", "s[76:77]", "s[64:65]", "")                                     \
    A2T_CLIMB("v241", "v242", "s[76:77]", "35f")                                                              \
    "36:\n\t"                                                                                                 \
    A2W_GATE_ASK                                                                                              \
    A2T_ADDR("v200", "v204") A2T_ADDR("v205", "v209") A2T_ADDR("v240", "v244") A2T_LAND_ADDR                  \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                \
    A2W_STORES_GATE("78")                                                                                      \
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
    /* ---- exits (the other wave is at X, or on its way there) */                                            \
    "91:\n\t"                                                                                                 \
    "s_mov_b32 %[code], 1\n\t"                                                                                \
    "s_branch 99f\n\t"                                                                                        \
    "92:\n\t"                                                                                                 \
    "s_mov_b32 %[code], 2\n\t"                                                                                \
    "s_branch 99f\n\t"                                                                                        \
    "93:\n\t"                                                                                                 \
    "s_mov_b32 %[code], 3\n\t"                                                                                \
    "s_branch 99f\n\t"                                                                                        \
    "94:\n\t"                                                                                                 \
    "s_mov_b32 %[code], 4\n\t"                                                                                \
    "99:\n\t"                                                                                                 \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                \
    A2W_ACC_OUT("s36", "4192") A2W_ACC_OUT("s38", "4200")                          \
    "s_mov_b32 %[len], s40\n\t"                                                                               \
    "s_mov_b32 %[pops], s41\n\t"                                                                              \
    "s_mov_b32 %[pushes], s42\n\t"                                                                            \
    "s_mov_b32 %[gm], s88\n\t"                                                                                \
    "s_mov_b32 %[pt], s80\n\t"

// ---------------------------------------------------------------------------------------------------------- wave 1: the expansion
// A2T_NBR in two halves: coordinates, in-grid mask and addresses of the five cells; their two loads
#define A2W_NBR_ADDR                                                                                          \
    "v_bfe_u32 v228, v196, 2, 15\n\t"                                                                         \
    "v_lshrrev_b32 v229, 17, v196\n\t"                                                                        \
    "v_add_u32 v210, v228, v186\n\t"                                                                          \
    "v_add_u32 v211, v229, v187\n\t"                                                                          \
    "v_cmp_gt_u32 vcc, s44, v210\n\t"                                                                         \
    "v_cmp_gt_u32_e64 s[94:95], s45, v211\n\t"                                                                \
    "s_and_b64 s[94:95], s[94:95], vcc\n\t"                                                                   \
    "s_and_b64 s[94:95], s[94:95], s[66:67]\n\t"                                                              \
    "v_mad_u32_u24 v212, v211, s44, v210\n\t"                                                                 \
    "v_cndmask_b32_e64 v212, 0, v212, s[94:95]\n\t"                                                           \
    "v_lshlrev_b32 v213, 1, v212\n\t"                                                                         \
    "v_lshlrev_b32 v214, 2, v212\n\t"
// ... of the FORESEEN top (payload v150): coordinates, mask and addresses in a second set (v151-v157, s[82:83]), loads into (D0, D1)
#define A2W_NBR2_TO(D0, D1)                                                                                   \
    "v_bfe_u32 v151, v150, 2, 15\n\t"                                                                         \
    "v_lshrrev_b32 v152, 17, v150\n\t"                                                                        \
    "v_add_u32 v153, v151, v186\n\t"                                                                          \
    "v_add_u32 v154, v152, v187\n\t"                                                                          \
    "v_cmp_gt_u32 vcc, s44, v153\n\t"                                                                         \
    "v_cmp_gt_u32_e64 s[82:83], s45, v154\n\t"                                                                \
    "s_and_b64 s[82:83], s[82:83], vcc\n\t"                                                                   \
    "s_and_b64 s[82:83], s[82:83], s[66:67]\n\t"                                                              \
    "v_mad_u32_u24 v155, v154, s44, v153\n\t"                                                                 \
    "v_cndmask_b32_e64 v155, 0, v155, s[82:83]\n\t"                                                           \
    "v_lshlrev_b32 v156, 1, v155\n\t"                                                                         \
    "v_lshlrev_b32 v157, 2, v155\n\t"                                                                         \
    "global_load_ushort " D0 ", v156, s[52:53]\n\t"                                                           \
    "global_load_dword " D1 ", v157, s[54:55] sc1\n\t"

// One iteration of the expansion wave.  An expansion's ten values (L1 distance and closed entry of the four neighbours and of the
// cell itself) cost the L2's latency -- the closed entries go through it: ~1 000 cycles for a lone wave -- which the one-wave loop
// hides under the pop's walk; this wave has nothing to hide it under, so it asks a whole iteration AHEAD, for the top it FORESEES:
// the smaller child of the present root (of equal keys the right one, as __adjust_heap walks).  That is the next top unless the
// entry from the back of the array rises to the root or a candidate of this expansion gets a key below it; measured on the
// reference's maze searches it is 95 - 99.5 % of the tops (-DBL_ASTAR_STAMPS counts them).  The top then really read at X is
// compared with the foreseen one and, if it is another (or wave 0 has been elsewhere: run word 3), asked for on the spot.
// Loads asked for ahead left before the LAST expansion's closedList entry was stored: that one cell (s84) is made good in
// registers.  (C0, C1): the pair this iteration's values land in; (N0, N1): the pair asked into for the next (the pairs swap);
// L: a digit that makes the copy's labels its own; NEXT: the other copy's entry (label and direction).
#define A2W_XBODY(PREFETCH, VMWAIT, C0, C1, N0, N1, L, NEXT)                                                  \
    L "0:\n\t"                                                                                                \
    A2W_TIMED_BARRIER("s89")                     /* X: the heap is final */                                   \
    "ds_read_b32 v240, v238 offset:48\n\t"           /* the run word */                                       \
    "ds_read_b32 v241, v177\n\t"                     /* the top: payload, key */                              \
    "ds_read_u16 v192, v190\n\t"                                                                              \
    "ds_read_b32 v244, v165\n\t"                     /* the root's children: keys (slots 2, 3), payloads (entries 1, 2) */ \
    "ds_read_b64 v[162:163], v177 offset:4\n\t"                                                               \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                \
    "v_readfirstlane_b32 s70, v240\n\t"                                                                       \
    "s_cmp_eq_u32 s70, 2\n\t"                                                                                 \
    "s_cbranch_scc1 99f\n\t"                                                                                  \
    "s_mov_b64 exec, 1\n\t"                      /* Z: the top and the children have been read -- wave 0 may store */ \
    "ds_write_b32 v238, v234 offset:52\n\t"                                                                   \
    "s_mov_b64 exec, -1\n\t"                                                                                  \
    A2W_T0                                                                                                    \
    "s_cmp_eq_u32 s70, 3\n\t"                        /* wave 0 has been elsewhere: what was foreseen is void */ \
    "s_cbranch_scc1 " L "3f\n\t"                                                                              \
    "v_readfirstlane_b32 s70, v241\n\t"                                                                       \
    "v_readfirstlane_b32 s71, v150\n\t"                                                                       \
    "s_cmp_eq_u32 s70, s71\n\t"                                                                               \
    "s_cbranch_scc1 " L "1f\n\t"                                                                              \
    L "3:\n\t"                                     /* not the foreseen top: asked for now */                  \
    A2W_ACC_COUNT("s59")                                                                                      \
    "v_mov_b32 v196, v241\n\t"                                                                                \
    A2W_NBR_ADDR                                                                                              \
    "global_load_ushort " C0 ", v213, s[52:53]\n\t"                                                           \
    "global_load_dword " C1 ", v214, s[54:55] sc1\n\t"                                                        \
    "s_branch " L "2f\n\t"                                                                                    \
    L "1:\n\t"                                     /* the foreseen top: its loads have been under way for an iteration */ \
    A2W_ACC_COUNT("s58")                                                                                      \
    "v_mov_b32 v196, v150\n\t"                                                                                \
    "v_mov_b32 v228, v151\n\t"                                                                                \
    "v_mov_b32 v229, v152\n\t"                                                                                \
    "v_mov_b32 v210, v153\n\t"                                                                                \
    "v_mov_b32 v211, v154\n\t"                                                                                \
    "v_mov_b32 v212, v155\n\t"                                                                                \
    "v_mov_b32 v213, v156\n\t"                                                                                \
    "v_mov_b32 v214, v157\n\t"                                                                                \
    "s_mov_b64 s[94:95], s[82:83]\n\t"                                                                        \
    L "2:\n\t"                                                                                                \
    "v_cmp_le_u32_sdwa vcc, v244, v244 src0_sel:WORD_1 src1_sel:WORD_0\n\t"                                   \
    "s_nop 3\n\t"                                                                                             \
    "v_cndmask_b32 v150, v162, v163, vcc\n\t"                                                                 \
    A2W_NBR2_TO(N0, N1)                                                                                       \
    PREFETCH                                                                                                  \
    A2T_FILL0                                                                                                 \
    A2W_T1("s60")                                                                                             \
    "s_waitcnt vmcnt(" VMWAIT ")\n\t"              /* (what has just been asked for stays under way) */       \
    A2W_T1("s61")                                                                                             \
    "v_mov_b32 v215, " C0 "\n\t"                                                                              \
    "v_mov_b32 v216, " C1 "\n\t"                                                                              \
    "v_cmp_eq_u32 vcc, s84, v212\n\t"                /* the cell the last expansion closed: closed */         \
    "s_nop 3\n\t"                                                                                             \
    "v_cndmask_b32 v216, v216, v166, vcc\n\t"                                                                 \
    A2T_EXPAND(L "5")                                                                                         \
    "v_readlane_b32 s84, v212, 4\n\t"                                                                         \
    A2W_T1("s81")                                                                                             \
    "v_mov_b32 v230, s87\n\t"                                                                                 \
    "v_mov_b32 v231, s88\n\t"                                                                                 \
    "v_mov_b32 v232, v196\n\t"                                                                                \
    "s_mov_b64 exec, 1\n\t"                                                                                   \
    "ds_write_b128 v238, v[230:233]\n\t"                                                                      \
    "s_mov_b64 exec, 15\n\t"                                                                                  \
    "ds_write_b64 v239, v[226:227]\n\t"                                                                       \
    "s_mov_b64 exec, -1\n\t"                                                                                  \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                \
    A2W_T1("s85")                                                                                             \
    A2W_TIMED_BARRIER("s90")                     /* Y: the record is in LDS */                                \
    "s_branch " NEXT "\n\t"

#define A2W_BODY_EXPAND(PREFETCH, VMWAIT)                                                                     \
    A2W_ENTRY                                                                                                 \
    "v_mov_b32 v190, 2\n\t"                                                                                   \
    "v_mov_b32 v165, 4\n\t"                                                                                   \
    "v_add_u32 v238, 4224, v191\n\t"                 /* the record */                                         \
    "v_lshl_add_u32 v239, v188, 3, v238\n\t"                                                                  \
    "v_add_u32 v239, 16, v239\n\t"                   /* the lane's candidate (lanes 0..3) */                  \
    "v_mov_b32 v233, 0\n\t"                                                                                   \
    "v_mov_b32 v166, s51\n\t"                        /* a closed entry of this search */                      \
    "v_mov_b32 v150, -1\n\t"                         /* no top foreseen */                                    \
    "v_mov_b32 v234, 1\n\t"                                                                                   \
    "s_mov_b32 s84, -1\n\t"                          /* no cell closed */                                     \
    A2W_ACC_ZERO("s89") A2W_ACC_ZERO("s90") A2W_ACC_ZERO("s91") A2W_ACC_ZERO("s58") A2W_ACC_ZERO("s59") A2W_ACC_ZERO("s60") A2W_ACC_ZERO("s61") A2W_ACC_ZERO("s81") A2W_ACC_ZERO("s85") \
    A2W_XBODY(PREFETCH, VMWAIT, "v158", "v159", "v160", "v161", "1", "20f")                                   \
    A2W_XBODY(PREFETCH, VMWAIT, "v160", "v161", "v158", "v159", "2", "10b")                                   \
    "99:\n\t"                                                                                                 \
    A2W_ACC_OUT("s89", "4204") A2W_ACC_OUT("s90", "4208") A2W_ACC_OUT("s91", "4212") A2W_ACC_OUT("s58", "4216") A2W_ACC_OUT("s59", "4220") A2W_ACC_OUT("s60", "4180") A2W_ACC_OUT("s61", "4184") A2W_ACC_OUT("s81", "4188") A2W_ACC_OUT("s85", "4196") \
    "s_waitcnt vmcnt(0)\n\t"

#define A2W_EXPAND_CLOBBERS A2T_CLOBBERS, "s82", "s83", "s84", "s85", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", \
    "v160", "v161", "v162", "v163", "v165", "v166"


#endif
