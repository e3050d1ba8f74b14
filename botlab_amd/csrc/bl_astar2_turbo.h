// bl_astar2_turbo.h -- the search loop of k_astar2 as ONE straight-line instruction stream, for the regime in which every key and
// payload an iteration touches is an LDS entry (2 <= open-list length <= PLN - 3) and the cost table is in LDS.
//
// Why: a lone wavefront pays ~4.2 cycles per instruction, ~50 per dependent LDS read, 28 - 60 per TAKEN branch and ~8 per
// hand-over between its vector and scalar units (tests/tools/lone_wave_probe.hip, profiles/r05_lone_wave_probe.txt).  What hipcc
// makes of the C++ form of this loop (bl_astar2.h) spends more than half of an iteration in branches around tier checks and in
// 64-bit address arithmetic.  Here an iteration with a two-round pop and two pushes is ~250 instructions, one taken branch (the
// loop's own) and seven LDS waits.  Same index operations in the same order as the C++ form: pop = rounds that only read, the
// climb, one net pass of stores (a2_pop_lds); push = a2_push_general's LDS case; expansion as in k_astar (astar.cpp:75-135,
// 213-233).  Both forms run under the same tests (tests/test_gpu_parity.py astar cases with and without BOTLAB_ASTAR_NO_TURBO).
//
// The loop leaves (code in %[code]) at an iteration boundary only: 1 = open list outside the regime (the C++ loop goes on from
// the same state), 2 = goal reached by this expansion (%[gm] = goal neighbours, %[pt] = the popped entry's payload), 3 = pop
// limit, 4 = the pushes of this expansion found the list at the regime's end (cannot happen: the regime leaves room for three).
//
// Registers are fixed: constants s44-s67 / v180-v188 (loaded from the table the kernel leaves in LDS at %[tbl]), temporaries
// s68-s97 / v190-v239; all declared as clobbers.  The dynamic LDS segment starts at address 0 (checked by the caller).
#ifndef BL_ASTAR2_TURBO_H
#define BL_ASTAR2_TURBO_H

// lane table: a 64-byte row per lane at tbl + 64 * lane: lk, ljm1, amask, areq, sh1, sh0, ddx, ddy, pdx, pdy; scalars at tbl + 4096
#define A2T_SC_W 0
#define A2T_SC_H 1
#define A2T_SC_GX 2
#define A2T_SC_GY 3
#define A2T_SC_GEN 4
#define A2T_SC_CN1 5
#define A2T_SC_MAXPOPS 6
#define A2T_SC_LIM 7
#define A2T_SC_L1 8
#define A2T_SC_CLOSED 10
#define A2T_SC_PB 12
#define A2T_SC_CB 13
#define A2T_SC_LVL1 14
#define A2T_SC_LVL2 15
#define A2T_SC_GK 16
#define A2T_SC_GP 18
#define A2T_SC_DLIM 20
#define A2T_TBL_BYTES (4096 + 256)            // lane rows; 32 scalar words; the two-wave loop's record and run word (bl_astar2_duo.h)

// one sift-down round: node N, child C_, knext K, payload P (of the child), mask SQ, lane mask OK (see a2_round_lds); FILL =
// instructions of the expansion that need nothing from the round: they run while the round's LDS read is under way
#define A2T_ROUND(N, C_, K, P, SQ, OK, FILL)                                                                  \
    "v_lshl_add_u32 " N ", s78, v180, v181\n\t"                                                               \
    "v_lshl_add_u32 v220, " N ", 2, 4\n\t"                                                                    \
    "v_min_u32 v220, %[kmax], v220\n\t"                                                                       \
    "ds_read_b32 v221, v220\n\t"                                                                              \
    "v_cmp_gt_u32_e64 s[68:69], s40, " N "\n\t"                                                               \
    "s_and_b64 s[68:69], s[68:69], " OK "\n\t"                                                                \
    "v_lshl_add_u32 " C_ ", " N ", 1, 1\n\t"                                                                  \
    FILL                                                                                                      \
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

// (in the first round's shadow) get_hCost of the five cells (astar.cpp:170-179) -> v219; is_goal -> s[36:37]; the payloads of the
// neighbours' entries -> v227; what the popped cell's closed entry would be -> v218
#define A2T_FILL0                                                                                             \
    "v_sad_u32 v217, s46, v210, 0\n\t"                                                                        \
    "v_sad_u32 v218, s47, v211, 0\n\t"                                                                        \
    "v_max_u32 v219, v217, v218\n\t"                                                                          \
    "v_min_u32 v217, v217, v218\n\t"                                                                          \
    "v_mul_u32_u24 v219, 10, v219\n\t"                                                                        \
    "v_lshl_add_u32 v219, v217, 2, v219\n\t"         /* 14 min + 10 (max - min) */                            \
    "v_cmp_eq_u32 vcc, s46, v210\n\t"                                                                         \
    "v_cmp_eq_u32_e64 s[36:37], s47, v211\n\t"                                                                \
    "s_and_b64 s[36:37], s[36:37], vcc\n\t"                                                                   \
    "v_lshl_or_b32 v227, v210, 2, v188\n\t"                                                                   \
    "v_lshl_or_b32 v227, v211, 17, v227\n\t"                                                                  \
    "v_and_b32 v218, 3, v196\n\t"                                                                             \
    "v_or_b32 v218, s51, v218\n\t"

// the climb inside the last round (child CL, knext KL, mask SQL): mask of the positions that move, landing node in s71
#define A2T_CLIMB(CL, KL, SQL, RARE)                                                                          \
    "v_cmp_lt_u32 vcc, v193, " KL "\n\t"                                                                      \
    "s_andn2_b64 s[68:69], " SQL ", vcc\n\t"                                                                  \
    "s_cbranch_scc0 " RARE "\n\t"                                                                             \
    "s_flbit_i32_b64 s70, s[68:69]\n\t"                                                                       \
    "s_sub_i32 s70, 63, s70\n\t"                                                                              \
    "v_readlane_b32 s71, " CL ", s70\n\t"                                                                     \
    "s_add_i32 s70, s70, 1\n\t"                                                                               \
    "s_bfm_b64 s[68:69], s70, 0\n\t"                                                                          \
    "s_and_b64 " SQL ", " SQL ", s[68:69]\n\t"
// store addresses of a round's nodes: key slot into KA, payload entry into N itself
#define A2T_ADDR(N, KA)                                                                                       \
    "v_lshl_add_u32 " KA ", " N ", 1, 2\n\t"                                                                  \
    "v_lshl_add_u32 " N ", " N ", 2, s56\n\t"
#define A2T_STORE(N, KA, K, P, M)                                                                             \
    "s_mov_b64 exec, " M "\n\t"                                                                               \
    "ds_write_b16 " KA ", " K "\n\t"                                                                          \
    "ds_write_b32 " N ", " P "\n\t"
// the landing: node s71 takes the value (key v193, payload v197)
#define A2T_LAND_ADDR                                                                                         \
    "s_lshl_b32 s70, s71, 1\n\t"                                                                              \
    "s_add_i32 s70, s70, 2\n\t"                                                                               \
    "v_mov_b32 v220, s70\n\t"                                                                                 \
    "s_lshl_b32 s70, s71, 2\n\t"                                                                              \
    "s_add_i32 s70, s70, s56\n\t"                                                                             \
    "v_mov_b32 v221, s70\n\t"
#define A2T_LAND_STORE                                                                                        \
    "s_mov_b64 exec, 1\n\t"                                                                                   \
    "ds_write_b16 v220, v193\n\t"                                                                             \
    "ds_write_b32 v221, v197\n\t"                                                                             \
    "s_mov_b64 exec, -1\n\t"
// a rare climb, one round up: positions of the round just left do not move (their mask SQX := 0); look at round (CL, KL, SQL);
// found -> mask and landing as in A2T_CLIMB, then BACK; else fall through (to the next A2T_RARE_UP or A2T_RARE_ROOT)
#define A2T_RARE_UP(SQX, CL, KL, SQL, BACK, TAG)                                                              \
    "s_mov_b64 " SQX ", 0\n\t"                                                                                \
    "v_cmp_lt_u32 vcc, v193, " KL "\n\t"                                                                      \
    "s_andn2_b64 s[68:69], " SQL ", vcc\n\t"                                                                  \
    "s_cbranch_scc0 " TAG "f\n\t"                                                                             \
    "s_flbit_i32_b64 s70, s[68:69]\n\t"                                                                       \
    "s_sub_i32 s70, 63, s70\n\t"                                                                              \
    "v_readlane_b32 s71, " CL ", s70\n\t"                                                                     \
    "s_add_i32 s70, s70, 1\n\t"                                                                               \
    "s_bfm_b64 s[68:69], s70, 0\n\t"                                                                          \
    "s_and_b64 " SQL ", " SQL ", s[68:69]\n\t"                                                                \
    "s_branch " BACK "\n\t"                                                                                   \
    TAG ":\n\t"
// ... the value rises to the root: no position moves
#define A2T_RARE_ROOT(SQX, BACK)                                                                              \
    "s_mov_b64 " SQX ", 0\n\t"                                                                                \
    "s_mov_b32 s71, 0\n\t"                                                                                    \
    "s_branch " BACK "\n\t"

// push_back + std::push_heap (see a2_push_general).  READ: the keys and payloads of the ancestors of the hole s40 (lane a: the
// (a + 1)-th).  REST: the entry (key v226, payload v227) of the lowest lane of the mask s87 rises past the leading run of
// ancestors with a larger key, each of which drops one level.
#define A2T_PUSH_READ                                                                                         \
    "s_add_i32 s78, s40, 1\n\t"                                                                               \
    "v_lshrrev_b32_e64 v230, v184, s78\n\t"                                                                   \
    "v_lshrrev_b32_e64 v231, v185, s78\n\t"                                                                   \
    "v_lshlrev_b32 v232, 1, v230\n\t"                                                                         \
    "v_add_u32 v233, -1, v230\n\t"                                                                            \
    "ds_read_u16 v234, v232\n\t"                                                                              \
    "v_min_u32 v233, %[pln], v233\n\t"                                                                        \
    "v_lshl_add_u32 v233, v233, 2, s56\n\t"                                                                   \
    "ds_read_b32 v235, v233\n\t"                                                                              \
    "v_lshlrev_b32 v236, 1, v231\n\t"                                                                         \
    "v_lshl_add_u32 v237, v231, 2, s56\n\t"                                                                   \
    "v_add_u32 v237, -4, v237\n\t"
#define A2T_PUSH_REST                                                                                         \
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
    "s_add_i32 s40, s40, 1\n\t"                                                                               \
    "s_add_i32 s42, s42, 1\n\t"
#define A2T_PUSH_CHECK(DONE)                                                                                  \
    "s_cmp_eq_u32 s87, 0\n\t"                                                                                 \
    "s_cbranch_scc1 " DONE "\n\t"

// Diagnostic build (-DBL_ASTAR_STAMPS): cycles between the marks of an iteration, summed in lane 0 of v250 + k (k = 0: back edge and
// checks .. 5: pushes); the sums travel through the table's words 16 .. 21.  Each mark drains the LDS queue: the figures are
// shares, not the undisturbed loop's times.
#ifdef BL_ASTAR_STAMPS
#define A2T_STAMP(K) "s_memtime s[100:101]\n\ts_waitcnt lgkmcnt(0)\n\ts_sub_u32 s43, s100, s43\n\tv_add_u32 v25" K ", s43, v25" K "\n\ts_mov_b32 s43, s100\n\t"
#define A2T_STAMPS_IN "v_mov_b32 v245, %[tbl]\n\tds_read_b128 v[250:253], v245 offset:4192\n\tds_read_b64 v[254:255], v245 offset:4208\n\ts_memtime s[100:101]\n\ts_waitcnt lgkmcnt(0)\n\ts_mov_b32 s43, s100\n\t"
#define A2T_STAMPS_OUT "v_mov_b32 v245, %[tbl]\n\tds_write_b128 v245, v[250:253] offset:4192\n\tds_write_b64 v245, v[254:255] offset:4208\n\t"
#define A2T_STAMP_CLOBBERS , "s43", "s100", "s101", "v245", "v250", "v251", "v252", "v253", "v254", "v255"
#else
#define A2T_STAMP(K) ""
#define A2T_STAMPS_IN ""
#define A2T_STAMPS_OUT ""
#define A2T_STAMP_CLOBBERS
#endif

#define A2T_NBR                                                                                               \
    /* ---- the loads of this expansion: lanes 0..3 the neighbours (astar.cpp:215-216), lane 4 the cell itself */ \
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
    "v_lshlrev_b32 v214, 2, v212\n\t"                                                                         \
    "global_load_ushort v215, v213, s[52:53]\n\t"                                                             \
    "global_load_dword v216, v214, s[54:55] sc1\n\t"

// the lines of the cells two steps away, asked for now and never waited for (grids that do not fit the L2: the next expansions
// find them there)
#define A2T_PREFETCH                                                                                          \
    "v_add_u32 v246, v228, v178\n\t"                                                                          \
    "v_add_u32 v247, v229, v179\n\t"                                                                          \
    "v_cmp_gt_u32 vcc, s44, v246\n\t"                                                                         \
    "v_cmp_gt_u32_e64 s[68:69], s45, v247\n\t"                                                                \
    "s_and_b64 s[68:69], s[68:69], vcc\n\t"                                                                   \
    "s_and_b64 s[68:69], s[68:69], s[98:99]\n\t"                                                              \
    "v_mad_u32_u24 v246, v247, s44, v246\n\t"                                                                 \
    "v_cndmask_b32_e64 v246, 0, v246, s[68:69]\n\t"                                                           \
    "v_lshlrev_b32 v247, 2, v246\n\t"                                                                         \
    "v_lshlrev_b32 v246, 1, v246\n\t"                                                                         \
    "global_load_ushort v248, v246, s[52:53]\n\t"                                                             \
    "global_load_dword v249, v247, s[54:55]\n\t"

// the expansion proper (the loads of A2T_NBR have arrived): closes the popped cell, then key v226 / payload v227 of every neighbour's
// entry, the lanes to push -> s87, the goal neighbours -> s88 (astar.cpp:95-135, 213-233)
#define A2T_EXPAND(TAG)                                                                                      \
    "v_lshrrev_b32 v222, 3, v216\n\t"                                                                         \
    "v_cmp_ne_u32 vcc, s48, v222\n\t"           /* not closed by this search (lane 4: the popped cell itself) */ \
    "s_and_b64 s[68:69], vcc, s[96:97]\n\t"                                                                   \
    "s_andn2_b64 s[92:93], s[94:95], s[96:97]\n\t"   /* neighbour lanes inside the grid */                    \
    "s_and_b64 s[92:93], s[92:93], vcc\n\t"          /* ... and not closed */                                 \
    "s_mov_b64 exec, s[68:69]\n\t"                                                                            \
    "global_store_dword v214, v218, s[54:55]\n\t"    /* closedList.push_back: the first entry per cell is the one observed */ \
    "s_mov_b64 exec, -1\n\t"                                                                                  \
    "s_add_i32 s41, s41, 1\n\t"                                                                               \
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
    "v_sub_u32 v226, v217, v225\n\t"                 /* lane 4: gCost of the popped node (never the start node here: its */ \
    "s_nop 0\n\t"                                    /* expansion has a one-entry list; a lane read needs a wait state */ \
    "v_readlane_b32 s86, v226, 4\n\t"                /* behind the VALU write of its source on gfx950) */     \
    "s_add_i32 s86, s86, 0x800a\n\t"                 /* + 10 (get_gCost), + 32768 (key bias) */               \
    "v_add_u32 v226, s86, v225\n\t"                  /* key of the neighbour's entry */                       \
    "v_cmp_gt_u32 vcc, 0xffff, v226\n\t"             /* fNew < INT16_MAX (astar.cpp:103,124) */               \
    "s_and_b64 s[92:93], s[92:93], vcc\n\t"                                                                   \
    "s_mov_b32 s87, s92\n\t"                                                                                  \
    "s_mov_b32 s88, s36\n\t"                                                                                  \
    "s_cmp_eq_u32 s88, 0\n\t"                                                                                 \
    "s_cbranch_scc1 " TAG "f\n\t"                                                                                  \
    "s_sub_i32 s70, 0, s88\n\t"                      /* neighbours before the goal neighbour only */          \
    "s_and_b32 s70, s70, s88\n\t"                                                                             \
    "s_add_i32 s70, s70, -1\n\t"                                                                              \
    "s_and_b32 s87, s87, s70\n\t"                                                                             \
    TAG ":\n\t"

#define A2T_RSF(dst, idx) "v_readfirstlane_b32 " dst ", " idx "\n\t"

#define A2T_BODY(PREFETCH, VMWAIT)                                                                            \
    /* ---- entry: state and constants */                                                                     \
    "s_mov_b32 s40, %[len]\n\t"                                                                               \
    "s_mov_b32 s41, %[pops]\n\t"                                                                              \
    "s_mov_b32 s42, %[pushes]\n\t"                                                                            \
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
    "v_mov_b32 v176, 0xffff\n\t"                                                                              \
    "s_mov_b32 s88, 0\n\t"                                                                                    \
    "s_mov_b32 s80, 0\n\t"                                                                                    \
    A2T_STAMPS_IN                                                                                             \
    /* ================================================================== one iteration */                    \
    "1:\n\t"                                                                                                  \
    A2T_STAMP("0")                                                                                            \
    "s_cmp_ge_u32 s41, s50\n\t"                                                                               \
    "s_cbranch_scc1 93f\n\t"                                                                                  \
    "s_add_i32 s70, s40, -2\n\t"                                                                              \
    "s_cmp_gt_u32 s70, s58\n\t"                 /* len < 2 (wraps) or len - 2 > lim - 2 */                    \
    "s_cbranch_scc1 91f\n\t"                                                                                  \
    /* ---- the top (payload v196, key v192) and the entry at the back of the array (key v193, payload v197): the value the */ \
    /* pop's sift-down places; all four stay in vector registers (every lane reads the same word) */          \
    "v_mov_b32 v190, 2\n\t"                                                                                   \
    "s_lshl_b32 s70, s40, 1\n\t"                                                                              \
    "v_mov_b32 v191, s70\n\t"                                                                                 \
    "ds_read_b32 v196, v177\n\t"                                                                              \
    "ds_read_u16 v192, v190\n\t"                                                                              \
    "ds_read_u16 v193, v191\n\t"                                                                              \
    "s_lshl_b32 s71, s40, 2\n\t"                                                                              \
    "s_add_i32 s71, s71, s56\n\t"                                                                             \
    "s_add_i32 s71, s71, -4\n\t"                                                                              \
    "v_mov_b32 v195, s71\n\t"                                                                                 \
    "ds_read_b32 v197, v195\n\t"                                                                              \
    "s_add_i32 s40, s40, -1\n\t"                                                                              \
    "s_waitcnt lgkmcnt(3)\n\t"                                                                                \
    A2T_NBR                                                                                                   \
    PREFETCH                                                                                                  \
    /* the slot the last entry leaves is "behind the heap" from here on */                                    \
    "s_mov_b64 exec, 1\n\t"                                                                                   \
    "ds_write_b16 v191, v176\n\t"                                                                             \
    "s_mov_b64 exec, -1\n\t"                                                                                  \
    /* ---- openList.pop(): rounds, the climb, one pass of stores */                                          \
    A2T_STAMP("1")                                                                                            \
    "s_mov_b32 s78, 1\n\t"                                                                                    \
    A2T_ROUND("v200", "v201", "v202", "v203", "s[72:73]", "s[62:63]", A2T_FILL0)                              \
    "s_cmp_lt_u32 s40, s59\n\t"                                                                               \
    "s_cbranch_scc1 20f\n\t"                                                                                  \
    A2T_ROUND("v205", "v206", "v207", "v208", "s[74:75]", "s[64:65]", "")                                     \
    "s_cmp_ge_u32 s40, s60\n\t"                                                                               \
    "s_cbranch_scc1 30f\n\t"                                                                                  \
    /* two rounds */                                                                                          \
    A2T_CLIMB("v206", "v207", "s[74:75]", "25f")                                                              \
    "26:\n\t"                                                                                                 \
    A2T_ADDR("v200", "v204") A2T_ADDR("v205", "v209") A2T_LAND_ADDR                                           \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                \
    A2T_STORE("v200", "v204", "v202", "v203", "s[72:73]") A2T_STORE("v205", "v209", "v207", "v208", "s[74:75]") \
    A2T_LAND_STORE                                                                                            \
    /* ---- the expansion */                                                                                  \
    "40:\n\t"                                                                                                 \
    A2T_PUSH_READ                               /* (the first push's ancestors: read while the expansion computes) */ \
    A2T_STAMP("2")                                                                                            \
    "s_waitcnt vmcnt(" VMWAIT ")\n\t"                                                                         \
    A2T_STAMP("3")                                                                                            \
    A2T_EXPAND("45")                                                                                          \
    A2T_STAMP("4")                                                                                            \
    A2T_PUSH_CHECK("50f") A2T_PUSH_REST                                                                       \
    A2T_PUSH_CHECK("50f") A2T_PUSH_READ A2T_PUSH_REST                                                         \
    A2T_PUSH_CHECK("50f") A2T_PUSH_READ A2T_PUSH_REST                                                         \
    "50:\n\t"                                                                                                 \
    A2T_STAMP("5")                                                                                            \
    "s_cmp_lg_u32 s88, 0\n\t"                                                                                 \
    "s_cbranch_scc1 92f\n\t"                                                                                  \
    "s_branch 1b\n\t"                                                                                         \
    /* ================================================================== out of line */                      \
    /* one round */                                                                                           \
    "20:\n\t"                                                                                                 \
    A2T_CLIMB("v201", "v202", "s[72:73]", "21f")                                                              \
    "22:\n\t"                                                                                                 \
    A2T_ADDR("v200", "v204") A2T_LAND_ADDR                                                                    \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                \
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
    A2T_ADDR("v200", "v204") A2T_ADDR("v205", "v209") A2T_ADDR("v240", "v244") A2T_LAND_ADDR                  \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                \
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
    /* ---- exits */                                                                                          \
    "91:\n\t"                                                                                                 \
    "s_mov_b32 %[code], 1\n\t"                                                                                \
    "s_branch 99f\n\t"                                                                                        \
    "92:\n\t"                                                                                                 \
    "v_readfirstlane_b32 s80, v196\n\t"                                                                       \
    "s_mov_b32 %[code], 2\n\t"                                                                                \
    "s_branch 99f\n\t"                                                                                        \
    "93:\n\t"                                                                                                 \
    "s_mov_b32 %[code], 3\n\t"                                                                                \
    "99:\n\t"                                                                                                 \
    A2T_STAMPS_OUT                                                                                            \
    "s_waitcnt vmcnt(0)\n\t"                         /* (the lines asked for ahead land in scratch registers) */ \
    "s_mov_b32 %[len], s40\n\t"                                                                               \
    "s_mov_b32 %[pops], s41\n\t"                                                                              \
    "s_mov_b32 %[pushes], s42\n\t"                                                                            \
    "s_mov_b32 %[gm], s88\n\t"                                                                                \
    "s_mov_b32 %[pt], s80\n\t"

#define A2T_CLOBBERS                                                                                          \
    "memory", "vcc", "scc", "s36", "s37", "s38", "s39",                                                       \
    "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", \
    "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", \
    "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99", \
    "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v190", "v191", "v192", "v193",     \
    "v194", "v195", "v196", "v197",                                                                           \
    "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214",     \
    "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231",     \
    "v232", "v233",                                                                                           \
    "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v246", "v247", "v248", "v249" A2T_STAMP_CLOBBERS

#endif
