// nastar_search_asm3.hip.h -- round-3 form of the hand-scheduled selection/expansion loop (nastar_search_asm.hip.h is round 2's):
// 74 instructions per step instead of 92 (42 VALU instead of ~57), same LDS layout, same semantics, for maps whose costs are all
// >= +0 (every encoder of the reference ends in sigmoid * const, VanillaAstar passes 0/1 maps; the kernel checks while loading and
// takes the round-2 stream otherwise).
//
// Why instruction count: the launch of the 4096-map batch lasts (longest chain) x (step latency), and profiles/r02 put the step at
// max(650 cycles lone-wave latency, 4 x VALU count x active waves of the SIMD): with all 16 maps of a CU resident, a SIMD's four
// waves are VALU-issue bound (4 x 60 x 4 = 960 cycles per step) for the first ~100-200 steps of every chain, and the rand32 launch
// IS its longest chain at lone-wave latency.  Length-aware placement was simulated first (tools/sim_placement.py): it needs a chain
// length predictor, and Chebyshev start-goal distance correlates 0.13 with the step count on mazes -- useless; fewer VALU
// instructions per step help every chain with no predictor.  What went:
//   * coordinates first: r_l = r* + dr, c_l = (c* & colmask) + dcol give the in-map test (max(r_l, c_l) < W, unsigned), the cell
//     index (v_lshl_add) AND the heuristic's inputs, instead of deriving (r, c) back from the cell index;
//   * h0 on integers: |dr|, |dc| by v_sad_u32, Chebyshev = v_max_u32 (== fl(fl(dr+dc) - min(dr,dc)) exactly), dr^2 + dc^2 by
//     v_mul_u32_u24 / v_mad_u32_u24 (exact, like the fp32 sum of two small squares), two converts: 10 instead of 15 instructions;
//   * s* is closed (g = -inf) BEFORE the lanes read their cells, so the chunk lane that holds s* sees a closed cell by itself (no
//     lane compare / select), and lane 8's own cell IS s*, so its chunk-entry offset needs no scalar detour;
//   * "open" for the chunk re-insertion is ONE v_cmpx_class_f32 with a per-lane class mask (finite classes on chunk lanes, nothing
//     elsewhere); the neighbours' "g[n] > g2" runs under EXEC = in-map neighbour lanes, so no value-to-beat select;
//   * keys are the raw bits of q = fl(f / fl32(sqrt(W))): q >= +0 when all costs are, and non-negative floats order like their bits.
// Hazard rules as in nastar_search_asm.hip.h.
#pragma once
#include "nastar_search_asm.hip.h"

namespace nastar {

#define NASTAR_ASM3_SELECT \
 /* ---- select: first cell of the minimal (key, index) chunk entry.  v21 arrives from LDS (no VALU->DPP hazard), so the three      \
    quad permutations all read IT and only the combined quad minimum waits for the next stage: 4 instructions, no s_nop;       \
    the step counter moves into one of the remaining wait states (the exits below undo it) */ \
        "v_min_u32_dpp v22, v21, v21 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t" \
        "v_min_u32_dpp v23, v21, v21 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t" \
        "v_min_u32_dpp v22, v21, v22 quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0xf\n\t" \
        "v_min_u32 v22, v22, v23\n\t" \
        "s_add_u32 %[it], %[it], 1\n\t" \
        "s_nop 0\n\t" \
        "v_min_u32_dpp v22, v22, v22 row_half_mirror row_mask:0xf bank_mask:0xf\n\t" \
        "s_nop 1\n\t" \
        "v_min_u32_dpp v22, v22, v22 row_mirror row_mask:0xf bank_mask:0xf\n\t" \
        "s_nop 1\n\t" \
        "v_min_u32_dpp v22, v22, v22 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t" \
        "s_nop 1\n\t" \
        "v_min_u32_dpp v22, v22, v22 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t" \
        "s_nop 0\n\t" \
        "v_readlane_b32 s40, v22, 63\n\t" /* M = minimal key */ \
        "s_cmp_eq_u32 s40, -1\n\t" \
        "s_cbranch_scc1 .Lempty%=\n\t" /* open list empty */ \
        "v_cmp_eq_u32 vcc, s40, v21\n\t" \
        "s_ff1_i32_b64 s41, vcc\n\t" /* first lane (= first chunks) with the minimum */ \
        "v_readlane_b32 s42, v20, s41\n\t" /* s* (its entry names the chunk's first minimal cell) */ \
        "s_cmp_eq_u32 s42, %[goal]\n\t" \
        "s_cbranch_scc1 .Lgoal%=\n\t"
#define NASTAR_ASM3_LOG \
        "s_lshl_b32 s53, %[it], 2\n\t"                    /* sel_log[iters] = s*; the counter already includes this step */ \
        "s_sub_u32 s53, s53, 4\n\t"                                                                                        \
        "v_mov_b32 v53, s53\n\t"                                                                                           \
        "v_mov_b32 v54, s42\n\t"                                                                                           \
        "s_mov_b64 exec, 1\n\t"                                                                                            \
        "global_store_dword v53, v54, %[logp]\n\t"                                                                         \
        "s_mov_b64 exec, -1\n\t"
#define NASTAR_ASM3_LOOPEND \
        "s_cmp_lt_u32 %[it], %[maxit]\n\t" \
        "s_cbranch_scc1 .Lloop%=\n" \
        ".Lbudget%=:\n\t" \
        "s_waitcnt lgkmcnt(0)\n\t" /* the prefetched chunk minima must have landed before v20/v21 are released */ \
        "s_mov_b32 %[sel], -2\n\t" \
        "s_branch .Lend%=\n" \
        ".Lempty%=:\n\t" \
        "s_sub_u32 %[it], %[it], 1\n\t" /* the selection that found nothing was not a step */ \
        "s_mov_b32 %[sel], -1\n\t" \
        "s_branch .Lend%=\n" \
        ".Lgoal%=:\n\t" \
        "s_sub_u32 %[it], %[it], 1\n\t" /* the goal's own step is counted by the caller */ \
        "s_mov_b32 %[sel], s42\n" \
        ".Lend%=:\n\t"

#define NASTAR_ASM3_X_PREFIX \
        "s_lshr_b32 s43, s42, %[LOGW]\n\t" /* r* */ \
        "s_and_b32 s44, s42, %[WM1]\n\t" /* c* */ \
        "v_add_u32 v32, s43, %[dr]\n\t" /* r_l */ \
        "v_and_b32 v33, s44, %[cmask]\n\t" /* chunk lanes: first column of the chunk of s*; others: c* */ \
        "v_add_u32 v33, v33, %[dcc]\n\t" /* c_l */ \
        "v_mov_b32 v24, s42\n\t" \
        "v_max_u32 v23, v32, v33\n\t" \
        "v_lshl_add_u32 v46, v32, %[LOGW], v33\n\t" /* this lane's cell (garbage outside the map) */ \
        "v_cmp_gt_u32 vcc, %[W], v23\n\t" /* inside the map (conv2d zero padding, :77-93); true for lane 8 and the chunk lanes */ \
        "v_lshlrev_b32 v27, 3, v24\n\t" \
        "ds_read_b64 v[28:29], v27\n\t" /* g[s*], cost[s*] */ \
        "s_and_b64 s[54:55], vcc, %[mnb]\n\t" /* in-map neighbour lanes */ \
        "v_cndmask_b32 v46, v24, v46, vcc\n\t" /* il: s* itself for out-of-map neighbours */ \
        "v_lshlrev_b32 v26, 3, v46\n\t" \
        "v_lshrrev_b32 v50, 4, v46\n\t" \
        "v_lshlrev_b32 v50, 3, v50\n\t" /* byte offset of cmin[chunk of il] */
#define NASTAR_ASM3_X_CLOSE \
 /* lane 8 (il == s*) closes s* (:222-225) and empties its chunk's entry BEFORE the lanes read their cells */ \
        "s_mov_b64 exec, 0x100\n\t" \
        "ds_write_b32 v27, %[vminf]\n\t" \
        "ds_write_b64 v50, v[48:49] offset:%[CMIN]\n\t" \
        "s_mov_b64 exec, -1\n\t"
#define NASTAR_ASM3_X_READCELL \
        "ds_read_b64 v[30:31], v26\n\t" /* g[il], cost[il] */
#define NASTAR_ASM3_X_HEUR \
 /* h0 = get_heuristic at (r_l, c_l) (:26-52) on integers, in the shadow of the LDS round trip */ \
        "v_sad_u32 v35, v32, %[gr], 0\n\t" /* |dr| */ \
        "v_sad_u32 v36, v33, %[gc], 0\n\t" /* |dc| */ \
        "v_max_u32 v37, v35, v36\n\t" /* == fl(fl(|dr| + |dc|) - min(|dr|, |dc|)) */ \
        "v_mul_u32_u24 v34, v35, v35\n\t" \
        "v_mad_u32_u24 v34, v36, v36, v34\n\t" \
        "v_cvt_f32_u32 v34, v34\n\t" \
        "v_sqrt_f32 v34, v34\n\t" \
        "v_cvt_f32_u32 v37, v37\n\t" /* independent: the wait state between the transcendental and its use */ \
        "v_mul_f32 v34, 0x3a83126f, v34\n\t" /* fl32(0.001) * euclid */ \
        "v_add_f32 v34, v37, v34\n\t" /* h0 */
#define NASTAR_ASM3_X_WAIT \
        "s_waitcnt lgkmcnt(0)\n\t"
#define NASTAR_ASM3_X_KEY \
        "v_add_f32 v34, v34, v31\n\t" /* :191-192 h = h0 + cost */ \
        "v_add_f32 v40, v28, v29\n\t" /* :234 g2 = g[s*] + cost[s*] */ \
        "v_mul_f32 v34, %[comg], v34\n\t" /* :206 (1-g_ratio)*h */ \
        "v_cndmask_b32_e64 v41, v30, v40, %[mnb]\n\t" /* neighbour lanes key g2, chunk lanes their own g */ \
        "v_mul_f32 v41, %[cgr], v41\n\t" \
        "v_add_f32 v41, v41, v34\n\t" /* :206 f */ \
        "v_mul_f32 v42, %[crcp], v41\n\t" /* :207 f / sqrt(W), correctly rounded (tools/fastdiv_check.c) */ \
        "v_fma_f32 v43, -v42, %[csq], v41\n\t" \
        "v_fma_f32 v47, v43, %[crcp], v42\n\t" /* q >= +0: its bits are the order-preserving key; [v46:v47] = (cell, key) */
#define NASTAR_ASM3_X_RELAX(SET55) \
 /* the chunk's open cells (finite g; s* reads -inf) re-enter its minimum through the SAME 64-bit atomic instruction as the relaxed \
    neighbours (EXEC = both lane sets): one LDS atomic per step instead of two -- with 16 waves per CU the LDS pipe is the shared resource */ \
        "v_cmp_class_f32_e64 s[58:59], v30, %[cls]\n\t" /* chunk lanes whose cell is open */ \
        "s_mov_b64 exec, s[54:55]\n\t" \
        "v_cmpx_gt_f32 vcc, v30, v40\n\t" /* :229,:235 g[n] > g2 on in-map neighbour lanes */ \
        SET55 \
        "ds_write_b32 v26, v40\n\t" /* :238 g[n] = g2 */ \
        "ds_write_b8 v46, %[pcode] offset:%[PDIR]\n\t" /* :246-249 parent = s* */ \
        "s_or_b64 exec, exec, s[58:59]\n\t" \
        "ds_min_u64 v50, v[46:47] offset:%[CMIN]\n\t" /* :242 relaxed neighbours AND the chunk's open cells enter the chunk minima: ONE atomic */ \
        "s_mov_b64 exec, -1\n\t"
/* the expansion = its sections in program order (the round-4 stream, nastar_search_asm4.hip.h, reuses CLOSE / HEUR / WAIT / RELAX) */
#define NASTAR_ASM3_EXPAND_(INIT55, SET55) \
    NASTAR_ASM3_X_PREFIX NASTAR_ASM3_X_CLOSE NASTAR_ASM3_X_READCELL INIT55 NASTAR_ASM3_X_HEUR NASTAR_ASM3_X_WAIT NASTAR_ASM3_X_KEY \
        NASTAR_ASM3_X_RELAX(SET55)
#define NASTAR_ASM3_EXPAND NASTAR_ASM3_EXPAND_(, )
/* dive form: s[60:61] = the neighbours relaxed in this step whose key is STRICTLY below the key s* was selected with (one VALU
   instruction under the EXEC mask the relaxation already runs with: the whole per-step price of the dive test) */
#define NASTAR_ASM3_EXPAND_DIVE NASTAR_ASM3_EXPAND_(, "v_cmp_lt_u32_e64 s[60:61], v47, s40\n\t")

// ---- the "dive" (64x64 instantiation only) ---------------------------------------------------------------------------------------
// When the best neighbour relaxed in this step has a key STRICTLY below the key s* was selected with, it is the next selection: every
// other open cell was >= (key of s*, s*) in the (key, index) order when s* won, and only the relaxed neighbours changed since.  The
// step then continues with that neighbour without waiting for the chunk-minima read-back, the per-lane minimum of four entries and the
// 64-lane reduction.  The per-step test is ONE v_cmp (relaxed neighbours with a key below the previous minimum -> an SGPR pair) + one
// scalar compare-and-branch; only a hit pays for the 3-stage minimum over lanes 0-7 that picks the best candidate (the neighbour lanes
// are in raster order = increasing cell index, so the first lane holding the minimum is the reference's first-flat-index tie-break).
// (A first version ran the 3-stage minimum on every step: ~12 instructions per step; it paid at 64x64 only.)
// Hit rate on the longest searches of the bench batches (tools/sim_dive.py): 78 % on random-obstacle 64x64 maps (BASELINE config 4),
// 50-73 % on random 32x32, 26-28 % on mazes.  Measured with the one-instruction test, same box, back to back (us per 4096-map launch,
// with / without): rand64 275.2 / 298.9, rand32 75.0 / 77.3, maze32 162.4 / 161.2; the reference's 64x64 block fixture (1169 steps, few
// dives) 263 ns per step either way.  The 64x64 instantiation, whose selection phase is the most expensive (four chunk minima per
// lane), dives; the 32x32 / 16x16 ones keep the plain loop: at a maze's 27 % the dive buys nothing and the headline batch is mazes.
// Loop layout: both ways round cost exactly one taken branch (the dive's lookup code sits in front of the expansion and falls into it);
// the chunk minima are still prefetched BEFORE the test (reading them only on the way into a full selection measured 2 % slower).
#define NASTAR_ASM3_DIVE_TEST \
        "s_cmp_ge_u32 %[it], %[maxit]\n\t" \
        "s_cbranch_scc1 .Lbudget%=\n\t" \
        "s_cmp_lg_u64 s[60:61], 0\n\t" /* did any relaxed neighbour beat the previous minimum? */ \
        "s_cbranch_scc1 .Ldive%=\n\t"
#define NASTAR_ASM3_DIVE_LOOKUP \
        ".Ldive%=:\n\t" \
        "v_cndmask_b32_e64 v55, v48, v47, s[60:61]\n\t" /* candidates keep their key, everyone else all ones (v48) */ \
        "s_nop 1\n\t" \
        "v_min_u32_dpp v56, v55, v55 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t" \
        "v_min_u32_dpp v57, v55, v55 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t" \
        "v_min_u32_dpp v56, v55, v56 quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0xf\n\t" \
        "v_min_u32 v56, v56, v57\n\t" \
        "s_add_u32 %[it], %[it], 1\n\t" \
        "s_nop 0\n\t" \
        "v_min_u32_dpp v56, v56, v56 row_half_mirror row_mask:0xf bank_mask:0xf\n\t" \
        "s_nop 0\n\t" \
        "v_readfirstlane_b32 s56, v56\n\t" /* the best candidate key (lanes 0-7 all hold it) */ \
        "s_mov_b32 s40, s56\n\t" /* = the key the next s* is selected with */ \
        "s_nop 0\n\t" \
        "v_cmp_eq_u32 vcc, s56, v55\n\t" \
        "s_ff1_i32_b64 s41, vcc\n\t" /* first neighbour lane with that key = smallest cell index among ties */ \
        "v_readlane_b32 s42, v46, s41\n\t" /* the next s* */ \
        "s_cmp_eq_u32 s42, %[goal]\n\t" \
        "s_cbranch_scc1 .Lgoal%=\n" \
        ".Lselected%=:\n\t"
#define NASTAR_ASM3_EXITS \
        ".Lbudget%=:\n\t" \
        "s_waitcnt lgkmcnt(0)\n\t" /* the prefetched chunk minima must have landed before their registers are released */ \
        "s_mov_b32 %[sel], -2\n\t" \
        "s_branch .Lend%=\n" \
        ".Lempty%=:\n\t" \
        "s_sub_u32 %[it], %[it], 1\n\t" \
        "s_mov_b32 %[sel], -1\n\t" \
        "s_branch .Lend%=\n" \
        ".Lgoal%=:\n\t" \
        "s_sub_u32 %[it], %[it], 1\n\t" \
        "s_mov_b32 %[sel], s42\n" \
        ".Lend%=:\n\t"

// ---- measured and dropped on top of this loop (round 3, profiles/r03/INDEX.md) -----------------------------------------------------
// The round-3 ablation probe (deleted in round 4 with the other development kernels; profiles/r03/ablate3_step_sections.txt) timed the step with one
// section removed at a time.  Lone wavefront, ns per step out of ~230: the four row-level reduction stages 22, the whole reduction 26,
// compare + find-first of the pick 17, the address prefix 29, closing s* 15, the relaxation's three LDS instructions 32 (the atomic
// alone 6), the read-back of the chunk minima 18 exposed, the cell read 7 exposed, one taken branch 10, the reduction's wait states 13,
// the compares of the two exits 15.  A lone wavefront pays ~7 cycles per instruction wherever the instruction sits, so only REMOVING
// instructions helps; three re-arrangements that keep the count were built, verified bit-identical on the 4096-map batches, and
// measured against this loop (us per 4096-map launch, same box, same run; `er_*.json`, `v3b_*.json`):
//   * "early reduction": the next minimum KEY from registers (old minima + the keys this step inserts) reduced while the atomic and the
//     read-back are in flight, +3 VALU: maze32 161.5 vs 159.5, rand32 75.8 vs 75.9, lone wave 305 vs 295 ns per step;
//   * exit tests moved behind the cell read (the LDS shadow) + two steps per loop trip: maze32 157.1 vs 157.2, rand32 74.2 vs 73.4;
//   * the same + an all-VALU address prefix (no SALU hop after the v_readlane): 157.9 / 72.9; + the row stage as three independent row
//     rotations instead of two dependent mirrors with wait states (+2 VALU): 162.3 / 73.4 -- slower on the full batch, where a SIMD's
//     four wavefronts are VALU-issue bound during the first ~150 steps.

#define NASTAR_ASM3_OPERANDS \
        : [it] "+s"(it), [sel] "=s"(sel) \
        : [l8] "v"(v_l8), [dr] "v"(v_dr), [cmask] "v"(v_cmask), [dcc] "v"(v_dcc), [pcode] "v"(v_pcode), [cls] "v"(v_cls), \
          [vminf] "v"(v_minf), [goal] "s"(goal_idx), [gr] "s"(goal_r), [gc] "s"(goal_c), \
          [maxit] "s"(max_iters), [cgr] "s"(d.gr), [comg] "s"(d.omg), [csq] "s"(d.sqrtW), [crcp] "s"(rcp_sqrtW), \
          [mnb] "s"(m_nb), [logp] "s"(logp), \
          [CMIN] "i"(L::CMIN), [CMIN16] "i"(L::CMIN + 16), [PDIR] "i"(L::PDIR), [LOGW] "i"(LOGW), [WM1] "i"(L::W - 1), [W] "i"(L::W) \
        : "memory", "vcc", "scc", "v20", "v21", "v22", "v23", "v24", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", \
          "v34", "v35", "v36", "v37", "v40", "v41", "v42", "v43", "v46", "v47", "v48", "v49", "v50", "s40", "s41", "s42", \
          "s43", "s44", "s54", "s55", "s53", "v53", "v54", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "s48", "s49", \
          "s50", "s51", "v55", "v56", "v57", "s56", "s58", "s59", "s60", "s61"

// Same contract as compact_search_loop_asm; precondition: every cost >= +0, g_ratio in [0, 1] (keys are raw float bits).
// Tried on top (measured, dropped): running the expansion's LDS reads under EXEC = lanes 0-8 and 16-31 only: 173.0 vs 169.5 us (maze32),
// 83.8 vs 79.7 (rand32) -- the extra s_mov on the lone-wave path costs more than the LDS passes it saves.
template <int LOGW, bool kLog, bool kDive = true>
__device__ __forceinline__ int compact_search_loop_asm3(const CompactDims& d, int lane, int goal_idx, int goal_r, int goal_c,
                                                        int max_iters, int& iters, float rcp_sqrtW, int* log_row)
{
    using L = AsmLayout<LOGW>;
    static_assert((L::CPL == 1 || L::CPL == 4) && L::HW >= 256, "1 or 4 chunk minima per lane, chunks inside one map row");
    int dr, dc;
    neighbour_delta(lane & 7, dr, dc);
    const bool is_nb = lane < 8, is_chk = (lane & 48) == 16;
    const int v_dr = is_nb ? dr : 0;
    const int v_dcc = is_nb ? dc : (is_chk ? (lane & 15) : 0);
    const uint32_t v_cmask = is_chk ? 0xFFFFFFF0u : 0xFFFFFFFFu;
    const uint32_t v_cls = is_chk ? 0x1F8u : 0u;  // v_cmp_class: -normal | -denormal | -0 | +0 | +denormal | +normal = finite = open
    const uint32_t v_pcode = P_PASS | (uint32_t)(lane & 7);
    const uint32_t v_l8 = (uint32_t)lane * 8u * L::CPL;
    const float v_minf = NASTAR_NEG_INF;
    const unsigned long long m_nb = 0xFFull;
    int it = __builtin_amdgcn_readfirstlane(iters);
    goal_idx = __builtin_amdgcn_readfirstlane(goal_idx);
    goal_r = __builtin_amdgcn_readfirstlane(goal_r);
    goal_c = __builtin_amdgcn_readfirstlane(goal_c);
    max_iters = __builtin_amdgcn_readfirstlane(max_iters);
    int sel;
    unsigned long long logp = reinterpret_cast<unsigned long long>(log_row);
#define NASTAR_ASM3_BODY(N, LOGPART) \
    NASTAR_ASM_ENTRY NASTAR_ASM_READ_##N NASTAR_ASM_LOOPTOP NASTAR_ASM_LOCALMIN_##N NASTAR_ASM3_SELECT LOGPART NASTAR_ASM3_EXPAND \
        NASTAR_ASM_READ_##N NASTAR_ASM3_LOOPEND
    /* entry -> select; [dive lookup ->] selected: log, expand, prefetch, budget + dive test -> dive | select -> selected */
#define NASTAR_ASM3_BODY_DIVE(N, LOGPART) \
    NASTAR_ASM_ENTRY NASTAR_ASM_READ_##N "s_branch .Lsel%=\n" NASTAR_ASM3_DIVE_LOOKUP LOGPART NASTAR_ASM3_EXPAND_DIVE NASTAR_ASM_READ_##N \
        NASTAR_ASM3_DIVE_TEST ".Lsel%=:\n\t" "s_waitcnt lgkmcnt(0)\n\t" NASTAR_ASM_LOCALMIN_##N NASTAR_ASM3_SELECT \
        "s_branch .Lselected%=\n" NASTAR_ASM3_EXITS
    if constexpr (L::CPL == 1) {
        if constexpr (kLog) asm volatile(NASTAR_ASM3_BODY(1, NASTAR_ASM3_LOG) NASTAR_ASM3_OPERANDS);
        else asm volatile(NASTAR_ASM3_BODY(1, ) NASTAR_ASM3_OPERANDS);
    } else if constexpr (kDive) {
        if constexpr (kLog) asm volatile(NASTAR_ASM3_BODY_DIVE(4, NASTAR_ASM3_LOG) NASTAR_ASM3_OPERANDS);
        else asm volatile(NASTAR_ASM3_BODY_DIVE(4, ) NASTAR_ASM3_OPERANDS);
    } else {
        if constexpr (kLog) asm volatile(NASTAR_ASM3_BODY(4, NASTAR_ASM3_LOG) NASTAR_ASM3_OPERANDS);
        else asm volatile(NASTAR_ASM3_BODY(4, ) NASTAR_ASM3_OPERANDS);
    }
#undef NASTAR_ASM3_BODY_DIVE
#undef NASTAR_ASM3_BODY
    iters = it;
    return sel;
}

}  // namespace nastar
