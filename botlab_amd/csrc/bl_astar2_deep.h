// bl_astar2_deep.h -- the search loop of k_astar2 as one instruction stream for open lists that reach into global memory:
// PLN + 2 <= length <= capacity - 4 (keys of levels 0 .. LEV and payloads of levels 0 .. LEV - 1 in LDS, the rest in the
// search's global scratch; the walk of a pop takes three rounds in LDS and one or two in global memory: up to 2^25 entries).  Same operations as
// bl_astar2.h's C++ forms (a2_pop_deep, a2_push_general) and the same expansion as bl_astar2_turbo.h (its macros are used here).
//
// What the order of an iteration buys: a global round trip costs a lone wavefront 300 - 900 cycles, and nothing in the
// EXPANSION depends on the pop -- so the walk's global round is asked for, the expansion runs while its keys travel, and the pop
// is finished behind it:
//     top + value (vp from global memory) -> neighbour loads -> rounds 1-3 (LDS) -> [round 4: loads asked for]
//     -> expansion -> [round 4 decided] -> climb, one pass of stores -> pushes (ancestors' keys + payloads in one round trip).
//
// Needs PLV == LEV - 1 (a2_big, a2_small): every node of the LDS rounds has its payload in LDS, every node of the global round in
// global memory; only the third round's CHILDREN straddle the payload tiers.  Registers: bl_astar2_turbo.h's, plus s28-s31, s34-s35 /
// s82-s85 and v150-v175.
#ifndef BL_ASTAR2_DEEP_H
#define BL_ASTAR2_DEEP_H

// the third round's children may be global payloads: the LDS read of A2T_ROUND stays (clamped), the global ones are asked for under
// s[84:85] into v154; selected at store time
#define A2D_ROUND2_PAYLOAD(C_, SQ)                                                                            \
    "v_cmp_le_u32 vcc, %[pln], " C_ "\n\t"                                                                    \
    "s_and_b64 s[84:85], vcc, " SQ "\n\t"                                                                     \
    "v_lshlrev_b32 v153, 2, " C_ "\n\t"                                                                       \
    "s_mov_b64 exec, s[84:85]\n\t"                                                                            \
    "global_load_dword v154, v153, s[28:29]\n\t"                                                              \
    "s_mov_b64 exec, -1\n\t"

// a round in global memory, first half: node N, 2 node + 1 -> C_, the children's keys asked for (PAIR; lanes without a left
// child keep 0xFFFFFFFF); valid lanes -> SQ
#define A2D_GROUND_ASK(N, C_, PAIR, SQ, PP)                                                                   \
    "v_lshl_add_u32 " N ", s78, v180, v181\n\t"                                                               \
    "v_cmp_gt_u32_e64 " SQ ", s40, " N "\n\t"                                                                 \
    "s_and_b64 " SQ ", " SQ ", s[64:65]\n\t"                                                                  \
    "v_lshl_add_u32 " C_ ", " N ", 1, 1\n\t"                                                                  \
    "v_cmp_gt_u32 vcc, s40, " C_ "\n\t"                                                                       \
    "s_and_b64 s[68:69], vcc, " SQ "\n\t"                                                                     \
    "v_lshl_add_u32 v220, " N ", 2, 4\n\t"                                                                    \
    "v_mov_b32 " PAIR ", -1\n\t"                                                                              \
    "v_lshl_add_u32 v221, " N ", 3, 4\n\t"                                                                    \
    "s_mov_b64 exec, s[68:69]\n\t"                                                                            \
    "global_load_dword " PAIR ", v220, s[30:31]\n\t"                                                          \
    "global_load_dwordx2 " PP ", v221, s[28:29]\n\t"   /* BOTH children's payloads (adjacent entries): no load behind the decision */ \
    "s_mov_b64 exec, -1\n\t"
// second half (the keys and payloads have arrived): a missing right child reads 0xFFFF; then as A2T_ROUND: mask SQ, child C_, knext K,
// the child's payload P (of the pair PLO, PHI); the walk's deepest lane -> s70
#define A2D_GROUND_DECIDE(N, C_, K, P, PAIR, SQ, PLO, PHI)                                                              \
    "v_add_u32 v221, 1, " C_ "\n\t"                                                                           \
    "v_cmp_gt_u32 vcc, s40, v221\n\t"                                                                         \
    "v_or_b32 v221, 0xffff0000, " PAIR "\n\t"                                                                 \
    "s_nop 0\n\t"                                                                                             \
    "v_cndmask_b32 " PAIR ", v221, " PAIR ", vcc\n\t"                                                         \
    "v_cmp_le_u32_sdwa vcc, " PAIR ", " PAIR " src0_sel:WORD_1 src1_sel:WORD_0\n\t"                           \
    "v_min_u32_sdwa " K ", " PAIR ", " PAIR " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n\t" \
    "s_nop 1\n\t"                                                                                             \
    "v_and_b32 v222, vcc_lo, v182\n\t"                                                                        \
    "v_cndmask_b32 " P ", " PLO ", " PHI ", vcc\n\t"                                                          \
    "v_addc_co_u32 " C_ ", vcc, 0, " C_ ", vcc\n\t"                                                           \
    "v_cmp_eq_u32 vcc, v222, v183\n\t"                                                                        \
    "s_and_b64 " SQ ", vcc, " SQ "\n\t"                                                                       \
    "s_flbit_i32_b64 s70, " SQ "\n\t"                                                                         \
    "s_sub_i32 s70, 63, s70\n\t"                                                                              \
    "s_bitset0_b64 " SQ ", s70\n\t"

// the value (key v193, payload v197) lands on node s71, whichever tier it is in
#define A2D_LAND                                                                                              \
    "s_lshl_b32 s70, s71, 1\n\t"                                                                              \
    "s_add_i32 s70, s70, 2\n\t"                                                                               \
    "v_mov_b32 v220, s70\n\t"                        /* key: slot land + 1 */                                 \
    "s_lshl_b32 s70, s71, 2\n\t"                                                                              \
    "v_mov_b32 v222, s70\n\t"                        /* payload, global offset */                             \
    "v_add_u32 v221, s56, v222\n\t"                  /* payload, LDS address */                               \
    "s_add_i32 s70, s71, 1\n\t"                                                                               \
    "s_cmp_lt_u32 s70, %[kslots]\n\t"                                                                         \
    "s_cselect_b64 s[68:69], 1, 0\n\t"                                                                        \
    "s_mov_b64 exec, s[68:69]\n\t"                                                                            \
    "ds_write_b16 v220, v193\n\t"                                                                             \
    "s_xor_b64 exec, s[68:69], 1\n\t"                                                                         \
    "global_store_short v220, v193, s[30:31]\n\t"                                                             \
    "s_cmp_lt_u32 s71, %[pln]\n\t"                                                                            \
    "s_cselect_b64 s[68:69], 1, 0\n\t"                                                                        \
    "s_mov_b64 exec, s[68:69]\n\t"                                                                            \
    "ds_write_b32 v221, v197\n\t"                                                                             \
    "s_xor_b64 exec, s[68:69], 1\n\t"                                                                         \
    "global_store_dword v222, v197, s[28:29]\n\t"                                                             \
    "s_mov_b64 exec, -1\n\t"

// stores of the first round in global memory under mask M: lane 0's node sits on level LEV (an LDS key slot), the others' keys
// and all payloads are global
#define A2D_STORE3(M)                                                                                         \
    "s_mov_b64 exec, -1\n\t"                                                                                  \
    "v_lshl_add_u32 v220, v155, 1, 2\n\t"                                                                     \
    "v_lshlrev_b32 v221, 2, v155\n\t"                                                                         \
    "s_and_b64 exec, " M ", 1\n\t"                                                                            \
    "ds_write_b16 v220, v157\n\t"                                                                             \
    "s_andn2_b64 exec, " M ", 1\n\t"                                                                          \
    "global_store_short v220, v157, s[30:31]\n\t"                                                             \
    "s_mov_b64 exec, " M "\n\t"                                                                               \
    "global_store_dword v221, v158, s[28:29]\n\t"                                                             \
    "s_mov_b64 exec, -1\n\t"
// ... of the second: everything global
#define A2D_STORE4(M)                                                                                         \
    "s_mov_b64 exec, -1\n\t"                                                                                  \
    "v_lshl_add_u32 v220, v171, 1, 2\n\t"                                                                     \
    "v_lshlrev_b32 v221, 2, v171\n\t"                                                                         \
    "s_mov_b64 exec, " M "\n\t"                                                                               \
    "global_store_short v220, v173, s[30:31]\n\t"                                                             \
    "global_store_dword v221, v174, s[28:29]\n\t"                                                             \
    "s_mov_b64 exec, -1\n\t"

// push_back + std::push_heap for a hole beyond the LDS payloads: READ asks for the ancestors' keys and payloads in whichever tier
// (lane a: the (a + 1)-th ancestor; LDS reads at clamped addresses, global loads under masks); REST decides and stores
#define A2D_PUSH_READ                                                                                         \
    "s_add_i32 s78, s40, 1\n\t"                                                                               \
    "v_lshrrev_b32_e64 v230, v184, s78\n\t"          /* ancestor's slot (0: none) */                          \
    "v_lshrrev_b32_e64 v231, v185, s78\n\t"          /* the slot it would drop to */                          \
    "v_min_u32 v232, %[kslotsm1], v230\n\t"                                                                   \
    "v_lshlrev_b32 v232, 1, v232\n\t"                                                                         \
    "ds_read_u16 v234, v232\n\t"                                                                              \
    "v_add_u32 v233, -1, v230\n\t"                   /* ancestor's entry */                                   \
    "v_min_u32 v160, %[pln], v233\n\t"                                                                        \
    "v_lshl_add_u32 v160, v160, 2, s56\n\t"                                                                   \
    "ds_read_b32 v235, v160\n\t"                                                                              \
    "v_cmp_le_u32 vcc, %[kslots], v230\n\t"                                                                   \
    "s_mov_b64 s[34:35], vcc\n\t"                    /* ancestors whose key is global */                      \
    "v_lshlrev_b32 v161, 1, v230\n\t"                                                                         \
    "s_mov_b64 exec, vcc\n\t"                                                                                 \
    "global_load_ushort v162, v161, s[30:31]\n\t"                                                             \
    "s_mov_b64 exec, -1\n\t"                                                                                  \
    "v_cmp_le_u32 vcc, %[pln], v233\n\t"                                                                      \
    "v_cmp_ne_u32_e64 s[68:69], 0, v230\n\t"                                                                  \
    "s_and_b64 s[38:39], vcc, s[68:69]\n\t"          /* ancestors whose payload is global */                  \
    "v_lshlrev_b32 v163, 2, v233\n\t"                                                                         \
    "s_mov_b64 exec, s[38:39]\n\t"                                                                            \
    "global_load_dword v164, v163, s[28:29]\n\t"                                                              \
    "s_mov_b64 exec, -1\n\t"                                                                                  \
    "v_lshlrev_b32 v236, 1, v231\n\t"                /* where it would drop to: key offset (both tiers), */   \
    "v_add_u32 v165, -1, v231\n\t"                                                                            \
    "v_lshlrev_b32 v166, 2, v165\n\t"                /* payload global offset, */                             \
    "v_add_u32 v237, s56, v166\n\t"                  /* payload LDS address */
// (REST in three pieces -- which candidate, how far it rises, the stores -- for bl_astar2_ahead.h, which asks for the first push's
// ancestors before the pop is in and looks at where the pop landed between the second and the third)
#define A2D_PUSH_REST A2D_PUSH_PICK A2D_PUSH_DECIDE A2D_PUSH_STORES
#define A2D_PUSH_PICK                                                                                         \
    "s_ff1_i32_b32 s91, s87\n\t"                                                                              \
    "s_add_i32 s70, s87, -1\n\t"                                                                              \
    "s_and_b32 s87, s87, s70\n\t"                                                                             \
    "v_readlane_b32 s89, v226, s91\n\t"                                                                       \
    "v_readlane_b32 s90, v227, s91\n\t"
#define A2D_PUSH_DECIDE                                                                                       \
    "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"                                                                       \
    "v_cndmask_b32_e64 v234, v234, v162, s[34:35]\n\t"                                                        \
    "v_cndmask_b32_e64 v235, v235, v164, s[38:39]\n\t"                                                        \
    "v_cmp_lt_u32 vcc, s89, v234\n\t"                                                                         \
    "s_not_b64 s[92:93], vcc\n\t"                                                                             \
    "s_ff1_i32_b64 s70, s[92:93]\n\t"                                                                         \
    "s_bfm_b64 s[92:93], s70, 0\n\t"                 /* the ancestors that drop */                            \
    "s_lshr_b32 s71, s78, s70\n\t"                   /* the slot the new entry takes */
#define A2D_PUSH_STORES                                                                                       \
    "v_cmp_gt_u32 vcc, %[kslots], v231\n\t"                                                                   \
    "s_and_b64 exec, s[92:93], vcc\n\t"                                                                       \
    "ds_write_b16 v236, v234\n\t"                                                                             \
    "s_andn2_b64 exec, s[92:93], vcc\n\t"                                                                     \
    "global_store_short v236, v234, s[30:31]\n\t"                                                             \
    "s_mov_b64 exec, -1\n\t"                                                                                  \
    "v_cmp_gt_u32 vcc, %[pln], v165\n\t"                                                                      \
    "s_and_b64 exec, s[92:93], vcc\n\t"                                                                       \
    "ds_write_b32 v237, v235\n\t"                                                                             \
    "s_andn2_b64 exec, s[92:93], vcc\n\t"                                                                     \
    "global_store_dword v166, v235, s[28:29]\n\t"                                                             \
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
    "s_add_i32 s40, s40, 1\n\t"                                                                               \
    "s_add_i32 s42, s42, 1\n\t"

#define A2D_BODY                                                                                              \
    /* ---- entry: state and constants (as bl_astar2_turbo.h) + the global arrays */                          \
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
    "ds_read_b128 v[150:153], v191 offset:4160\n\t"                                                           \
    "ds_read_b32 v154, v191 offset:4176\n\t"                                                                  \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                \
    A2T_RSF("s44", "v192") A2T_RSF("s45", "v193") A2T_RSF("s46", "v194") A2T_RSF("s47", "v195")               \
    A2T_RSF("s48", "v196") A2T_RSF("s49", "v197") A2T_RSF("s50", "v198")                                      \
    A2T_RSF("s52", "v200") A2T_RSF("s53", "v201") A2T_RSF("s54", "v202") A2T_RSF("s55", "v203")               \
    A2T_RSF("s56", "v204") A2T_RSF("s57", "v205") A2T_RSF("s59", "v206") A2T_RSF("s60", "v207")               \
    A2T_RSF("s30", "v150") A2T_RSF("s31", "v151") A2T_RSF("s28", "v152") A2T_RSF("s29", "v153") A2T_RSF("s58", "v154") \
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
    "s_sub_u32 s70, s40, %[dlo]\n\t"                                                                          \
    "s_cmp_gt_u32 s70, s58\n\t"                 /* len < PLN + 2 (wraps) or beyond the four-round depth */    \
    "s_cbranch_scc1 91f\n\t"                                                                                  \
    /* ---- the top (LDS) and the entry at the back of the array: key from LDS or global memory, payload global */ \
    "v_mov_b32 v190, 2\n\t"                                                                                   \
    "ds_read_b32 v196, v177\n\t"                                                                              \
    "ds_read_u16 v192, v190\n\t"                                                                              \
    "s_lshl_b32 s70, s40, 1\n\t"                                                                              \
    "v_mov_b32 v168, s70\n\t"                                                                                 \
    "s_min_u32 s71, s70, %[kmax2]\n\t"                                                                        \
    "v_mov_b32 v191, s71\n\t"                                                                                 \
    "ds_read_u16 v193, v191\n\t"                                                                              \
    "s_cmp_ge_u32 s40, %[kslots]\n\t"                                                                         \
    "s_cselect_b64 s[38:39], -1, 0\n\t"              /* the last entry's key is global */                     \
    "s_mov_b64 exec, s[38:39]\n\t"                                                                            \
    "global_load_ushort v169, v168, s[30:31]\n\t"                                                             \
    "s_mov_b64 exec, -1\n\t"                                                                                  \
    "s_lshl_b32 s71, s40, 2\n\t"                                                                              \
    "s_add_i32 s71, s71, -4\n\t"                                                                              \
    "v_mov_b32 v170, s71\n\t"                                                                                 \
    "global_load_dword v197, v170, s[28:29]\n\t"                                                              \
    "s_waitcnt lgkmcnt(2)\n\t"                                                                                \
    A2T_NBR                                                                                                   \
    A2T_PREFETCH                                                                                              \
    /* the slot the last entry leaves is "behind the heap" from here on (an LDS slot: 0xFFFF) */              \
    "s_andn2_b64 exec, 1, s[38:39]\n\t"                                                                       \
    "ds_write_b16 v191, v176\n\t"                                                                             \
    "s_mov_b64 exec, -1\n\t"                                                                                  \
    "s_add_i32 s40, s40, -1\n\t"                                                                              \
    /* ---- openList.pop(): three rounds in LDS */                                                            \
    A2T_STAMP("1")                                                                                            \
    "s_mov_b32 s78, 1\n\t"                                                                                    \
    A2T_ROUND("v200", "v201", "v202", "v203", "s[72:73]", "s[62:63]", A2T_FILL0)                              \
    A2T_ROUND("v205", "v206", "v207", "v208", "s[74:75]", "s[64:65]", "")                                     \
    A2T_ROUND("v240", "v241", "v242", "v243", "s[76:77]", "s[64:65]", "")                                     \
    A2D_ROUND2_PAYLOAD("v241", "s[76:77]")                                                                    \
    "s_cmp_ge_u32 s40, %[kslots]\n\t"                                                                         \
    "s_cbranch_scc1 30f\n\t"                                                                                  \
    /* -- the walk ends inside LDS: expansion, then the climb from the third round */                         \
    A2T_STAMP("2")                                                                                            \
    "s_waitcnt vmcnt(3)\n\t"                                                                                  \
    A2T_STAMP("3")                                                                                            \
    A2T_EXPAND("41")                                                                                          \
    A2T_STAMP("4")                                                                                            \
    "s_waitcnt vmcnt(1)\n\t"                         /* (only the closed entry's store may be under way) */   \
    "v_cndmask_b32_e64 v193, v193, v169, s[38:39]\n\t"                                                        \
    "v_cndmask_b32_e64 v243, v243, v154, s[84:85]\n\t"                                                        \
    A2T_CLIMB("v241", "v242", "s[76:77]", "35f")                                                              \
    "36:\n\t"                                                                                                 \
    A2T_ADDR("v200", "v204") A2T_ADDR("v205", "v209") A2T_ADDR("v240", "v244")                                \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                \
    A2T_STORE("v200", "v204", "v202", "v203", "s[72:73]") A2T_STORE("v205", "v209", "v207", "v208", "s[74:75]") \
    A2T_STORE("v240", "v244", "v242", "v243", "s[76:77]")                                                     \
    "s_mov_b64 exec, -1\n\t"                                                                                  \
    A2D_LAND                                                                                                  \
    /* ---- pushes */                                                                                         \
    "45:\n\t"                                                                                                 \
    A2T_STAMP("0")                                                                                            \
    A2T_PUSH_CHECK("50f") A2D_PUSH_READ A2D_PUSH_REST                                                         \
    A2T_PUSH_CHECK("50f") A2D_PUSH_READ A2D_PUSH_REST                                                         \
    A2T_PUSH_CHECK("50f") A2D_PUSH_READ A2D_PUSH_REST                                                         \
    "50:\n\t"                                                                                                 \
    A2T_STAMP("5")                                                                                            \
    "s_cmp_lg_u32 s88, 0\n\t"                                                                                 \
    "s_cbranch_scc1 92f\n\t"                                                                                  \
    "s_branch 1b\n\t"                                                                                         \
    /* ================================================================== the walk goes on in global memory */ \
    "30:\n\t"                                                                                                 \
    A2D_GROUND_ASK("v155", "v156", "v159", "s[82:83]", "v[164:165]")                                                        \
    A2T_STAMP("2")                                                                                            \
    "s_waitcnt vmcnt(5)\n\t"                         /* the neighbours' loads (the round's keys and payloads still travel) */ \
    A2T_STAMP("3")                                                                                            \
    A2T_EXPAND("42")                                                                                          \
    A2T_STAMP("4")                                                                                            \
    "s_waitcnt vmcnt(1)\n\t"                                                                                  \
    "v_cndmask_b32_e64 v193, v193, v169, s[38:39]\n\t"                                                        \
    "v_cndmask_b32_e64 v243, v243, v154, s[84:85]\n\t"                                                        \
    A2D_GROUND_DECIDE("v155", "v156", "v157", "v158", "v159", "s[82:83]", "v164", "v165")                                     \
    /* (open lists beyond 2^(LEV + 6) entries: the walk's node on the round's last level has a child -> a second global round) */ \
    "v_readlane_b32 s78, v155, s70\n\t"                                                                       \
    "s_lshl_b32 s71, s78, 1\n\t"                                                                              \
    "s_add_i32 s71, s71, 1\n\t"                                                                               \
    "s_cmp_lt_u32 s71, s40\n\t"                                                                               \
    "s_cselect_b32 s71, s70, 0\n\t"                                                                           \
    "s_cmp_ge_u32 s71, 31\n\t"                                                                                \
    "s_cbranch_scc1 60f\n\t"                                                                                  \
    A2T_CLIMB("v156", "v157", "s[82:83]", "31f")                                                              \
    "32:\n\t"                                                                                                 \
    A2T_ADDR("v200", "v204") A2T_ADDR("v205", "v209") A2T_ADDR("v240", "v244")                                \
    "s_waitcnt lgkmcnt(0)\n\t"                       /* (everything from global memory came with the wait in front of the decision) */ \
    A2T_STORE("v200", "v204", "v202", "v203", "s[72:73]") A2T_STORE("v205", "v209", "v207", "v208", "s[74:75]") \
    A2T_STORE("v240", "v244", "v242", "v243", "s[76:77]")                                                     \
    A2D_STORE3("s[82:83]")                                                                                    \
    A2D_LAND                                                                                                  \
    "s_branch 45b\n\t"                                                                                        \
    /* -- five rounds */                                                                                      \
    "60:\n\t"                                                                                                 \
    "s_add_i32 s78, s78, 1\n\t"                                                                               \
    A2D_GROUND_ASK("v171", "v172", "v175", "s[34:35]", "v[166:167]")                                                        \
    "s_waitcnt vmcnt(0)\n\t"                                                                                  \
    A2D_GROUND_DECIDE("v171", "v172", "v173", "v174", "v175", "s[34:35]", "v166", "v167")                                     \
    A2T_CLIMB("v172", "v173", "s[34:35]", "61f")                                                              \
    "62:\n\t"                                                                                                 \
    A2T_ADDR("v200", "v204") A2T_ADDR("v205", "v209") A2T_ADDR("v240", "v244")                                \
    "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"                                                                       \
    A2T_STORE("v200", "v204", "v202", "v203", "s[72:73]") A2T_STORE("v205", "v209", "v207", "v208", "s[74:75]") \
    A2T_STORE("v240", "v244", "v242", "v243", "s[76:77]")                                                     \
    A2D_STORE3("s[82:83]")                                                                                    \
    A2D_STORE4("s[34:35]")                                                                                    \
    A2D_LAND                                                                                                  \
    "s_branch 45b\n\t"                                                                                        \
    "61:\n\t"                                                                                                 \
    A2T_RARE_UP("s[34:35]", "v156", "v157", "s[82:83]", "62b", "63")                                          \
    A2T_RARE_UP("s[82:83]", "v241", "v242", "s[76:77]", "62b", "64")                                          \
    A2T_RARE_UP("s[76:77]", "v206", "v207", "s[74:75]", "62b", "65")                                          \
    A2T_RARE_UP("s[74:75]", "v201", "v202", "s[72:73]", "62b", "66")                                          \
    A2T_RARE_ROOT("s[72:73]", "62b")                                                                          \
    /* rare climbs */                                                                                         \
    "31:\n\t"                                                                                                 \
    A2T_RARE_UP("s[82:83]", "v241", "v242", "s[76:77]", "32b", "33")                                          \
    A2T_RARE_UP("s[76:77]", "v206", "v207", "s[74:75]", "32b", "34")                                          \
    A2T_RARE_UP("s[74:75]", "v201", "v202", "s[72:73]", "32b", "39")                                          \
    A2T_RARE_ROOT("s[72:73]", "32b")                                                                          \
    "35:\n\t"                                                                                                 \
    A2T_RARE_UP("s[76:77]", "v206", "v207", "s[74:75]", "36b", "37")                                          \
    A2T_RARE_UP("s[74:75]", "v201", "v202", "s[72:73]", "36b", "38")                                          \
    A2T_RARE_ROOT("s[72:73]", "36b")                                                                          \
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
    "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"                                                                       \
    "s_mov_b32 %[len], s40\n\t"                                                                               \
    "s_mov_b32 %[pops], s41\n\t"                                                                              \
    "s_mov_b32 %[pushes], s42\n\t"                                                                            \
    "s_mov_b32 %[gm], s88\n\t"                                                                                \
    "s_mov_b32 %[pt], s80\n\t"

#define A2D_CLOBBERS A2T_CLOBBERS, "s30", "s31", "s28", "s29", "s34", "s35",                                  \
    "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", \
    "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175"

#endif
