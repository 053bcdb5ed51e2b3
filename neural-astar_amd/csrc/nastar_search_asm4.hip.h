// nastar_search_asm4.hip.h -- round-4 form of the hand-scheduled selection/expansion loop: the round-3 stream (nastar_search_asm3.hip.h,
// whose sections it reuses) with FIVE fewer instructions on the step's critical path, and parametrised over the cell record so that
// the same stream serves a second LDS layout:
//
//   * UNIT-COST layout (kUnit): VanillaAstar hands ONE binary tensor over as cost map AND obstacle map (reference astar.py:93-94), so
//     every cell the search can touch costs exactly 1.0 and the per-cell cost word need not exist: the cell record is g alone (4 B
//     instead of 8), 5.5 B/cell with parents and chunk minima, 29 maps of 32x32 per CU instead of 16.  Why that matters: profiles/r04
//     (tools/ubench/rate.hip, tools/probe_streams.py) show that with several batches in flight the search is bound by LDS CAPACITY --
//     throughput is proportional to the resident maps per CU (16 -> 12 -> 8 maps: 56.9 -> 43.8 -> 34.4 M maps/s) while the VALU and
//     LDS pipes are about half busy -- so the lever is bytes per map, not instructions per map.  In the stream only the two reads
//     shrink (ds_read_b32) and the two adds take the constant 1.0: h = fl(h0 + 1.0) (:191-192), g2 = fl(g[s*] + 1.0) (:234).
//   * g_ratio == 0.5 (kHalf; the reference's default everywhere: astar.py:19,107, scripts/config/*.yaml): gr = omg = 0.5, both
//     products in f = fl(fl(0.5 g) + fl(0.5 h)) are exact, f = 0.5 fl(g + h) and q = fl(f / sqrt(W)) = 0.5 fl(fl(g + h) / sqrt(W)):
//     the key bits of q' = fl(fl(g + h) / sqrt(W)) order and TIE exactly like those of q (a power-of-two factor commutes with every
//     rounding; the only value small enough to underflow is the goal's own h = cost, which no other key can be confused with), so
//     the two multiplies go.
//   * out-of-map neighbour lanes keep their (garbage) cell index instead of being redirected to s*: their LDS read is harmless
//     (out-of-range addresses read 0, in-range ones read a word nobody uses) and every store already runs under EXEC = in-map
//     lanes, so the select and its VCC dependency go.
//   * idle chunk entries carry the GOAL's index next to the all-ones key: an empty open list then selects "the goal" and leaves
//     through the goal exit, where the minimal key (all ones) tells the two apart -- the per-step empty test and its branch go (the
//     two wait states the following v_cmp needs after v_readlane are filled by the step counter and one s_nop).
// Hazard rules as in nastar_search_asm.hip.h.
#pragma once
#include "nastar_search_asm3.hip.h"

namespace nastar {

// LDS byte offsets of the unit-cost layout: g[HW] fp32 | cmin[CPL * 64] u64 | pdir[HW] u8
template <int LOGW>
struct AsmLayoutUnit {
    static constexpr int W = 1 << LOGW;
    static constexpr int HW = W * W;
    static constexpr int CPL = (HW / 16 + 63) / 64;
    static constexpr int CMIN = HW * 4;
    static constexpr int PDIR = CMIN + CPL * 64 * 8;
    static constexpr int BYTES = PDIR + HW;
};

#define NASTAR_ASM4_ENTRY \
        "s_cmp_ge_u32 %[it], %[maxit]\n\t" \
        "s_cbranch_scc1 .Lbudget%=\n\t" \
        "v_mov_b32 v48, %[goal]\n\t" /* [v48:v49] = an idle chunk entry: (cell = goal, key = all ones) */ \
        "v_mov_b32 v49, -1\n\t"

#define NASTAR_ASM4_SELECT \
        "v_min_u32_dpp v22, v21, v21 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t" \
        "v_min_u32_dpp v23, v21, v21 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t" \
        "v_min_u32_dpp v22, v21, v22 quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0xf\n\t" \
        "v_min_u32 v22, v22, v23\n\t" \
        "s_nop 1\n\t" \
        "v_min_u32_dpp v22, v22, v22 row_half_mirror row_mask:0xf bank_mask:0xf\n\t" \
        "s_nop 1\n\t" \
        "v_min_u32_dpp v22, v22, v22 row_mirror row_mask:0xf bank_mask:0xf\n\t" \
        "s_nop 1\n\t" \
        "v_min_u32_dpp v22, v22, v22 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t" \
        "s_nop 1\n\t" \
        "v_min_u32_dpp v22, v22, v22 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t" \
        "s_nop 0\n\t" \
        "v_readlane_b32 s40, v22, 63\n\t" /* M = minimal key (all ones: open list empty -> the pick below yields the goal) */ \
        "s_add_u32 %[it], %[it], 1\n\t" /* two wait states: VALU-written SGPR -> VALU read */ \
        "s_nop 0\n\t" \
        "v_cmp_eq_u32 vcc, s40, v21\n\t" \
        "s_ff1_i32_b64 s41, vcc\n\t" \
        "v_readlane_b32 s42, v20, s41\n\t" \
        "s_cmp_eq_u32 s42, %[goal]\n\t" \
        "s_cbranch_scc1 .Lgoal%=\n\t"

#define NASTAR_ASM4_X_PREFIX(SH, RDS) \
        "s_lshr_b32 s43, s42, %[LOGW]\n\t" /* r* */ \
        "s_and_b32 s44, s42, %[WM1]\n\t" /* c* */ \
        "v_add_u32 v32, s43, %[dr]\n\t" /* r_l */ \
        "v_and_b32 v33, s44, %[cmask]\n\t" \
        "v_add_u32 v33, v33, %[dcc]\n\t" /* c_l */ \
        "v_mov_b32 v24, s42\n\t" \
        "v_max_u32 v23, v32, v33\n\t" \
        "v_lshl_add_u32 v46, v32, %[LOGW], v33\n\t" /* this lane's cell; garbage on out-of-map neighbour lanes, which only ever READ with it */ \
        "v_cmp_gt_u32 vcc, %[W], v23\n\t" /* inside the map (conv2d zero padding, :77-93) */ \
        "v_lshlrev_b32 v27, " SH ", v24\n\t" \
        RDS /* g[s*] (, cost[s*]) */ \
        "v_lshlrev_b32 v26, " SH ", v46\n\t" \
        "s_and_b64 s[54:55], vcc, %[mnb]\n\t" /* in-map neighbour lanes */ \
        "v_lshrrev_b32 v50, 4, v46\n\t" \
        "v_lshlrev_b32 v50, 3, v50\n\t" /* byte offset of cmin[chunk of this lane's cell] */

#define NASTAR_ASM4_X_KEY(ADDH, ADDG, MULH, MULG) \
        ADDH /* :191-192 h = h0 + cost */ \
        ADDG /* :234 g2 = g[s*] + cost[s*] */ \
        MULH /* :206 (1-g_ratio)*h */ \
        "v_cndmask_b32_e64 v41, v30, v40, %[mnb]\n\t" /* neighbour lanes key g2, chunk lanes their own g */ \
        MULG /* :206 g_ratio*g */ \
        "v_add_f32 v41, v41, v34\n\t" /* :206 f */ \
        "v_mul_f32 v42, %[crcp], v41\n\t" /* :207 f / sqrt(W), correctly rounded (tools/fastdiv_check.c) */ \
        "v_fma_f32 v43, -v42, %[csq], v41\n\t" \
        "v_fma_f32 v47, v43, %[crcp], v42\n\t" /* q >= +0: its bits are the key; [v46:v47] = (cell, key) */

/* the dive's candidate lookup with v49 (all ones) as the "not a candidate" key: v48 now holds the goal index */
#define NASTAR_ASM4_DIVE_LOOKUP \
        ".Ldive%=:\n\t" \
        "v_cndmask_b32_e64 v55, v49, v47, s[60:61]\n\t" \
        "s_nop 1\n\t" \
        "v_min_u32_dpp v56, v55, v55 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t" \
        "v_min_u32_dpp v57, v55, v55 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t" \
        "v_min_u32_dpp v56, v55, v56 quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0xf\n\t" \
        "v_min_u32 v56, v56, v57\n\t" \
        "s_add_u32 %[it], %[it], 1\n\t" \
        "s_nop 0\n\t" \
        "v_min_u32_dpp v56, v56, v56 row_half_mirror row_mask:0xf bank_mask:0xf\n\t" \
        "s_nop 0\n\t" \
        "v_readfirstlane_b32 s56, v56\n\t" \
        "s_mov_b32 s40, s56\n\t" \
        "s_nop 0\n\t" \
        "v_cmp_eq_u32 vcc, s56, v55\n\t" \
        "s_ff1_i32_b64 s41, vcc\n\t" \
        "v_readlane_b32 s42, v46, s41\n\t" \
        "s_cmp_eq_u32 s42, %[goal]\n\t" \
        "s_cbranch_scc1 .Lgoal%=\n" \
        ".Lselected%=:\n\t"

/* exits: budget | goal selected or open list empty (the caller tells them apart by the minimal key) */
#define NASTAR_ASM4_EXITS \
        ".Lbudget%=:\n\t" \
        "s_waitcnt lgkmcnt(0)\n\t" /* the prefetched chunk minima must have landed before their registers are released */ \
        "s_mov_b32 %[sel], -2\n\t" \
        "s_branch .Lend%=\n" \
        ".Lgoal%=:\n\t" \
        "s_sub_u32 %[it], %[it], 1\n\t" /* the goal's own step is counted by the caller; a selection that found nothing was not a step */ \
        "s_mov_b32 %[sel], s42\n\t" \
        "s_mov_b32 %[mkey], s40\n" \
        ".Lend%=:\n\t"
#define NASTAR_ASM4_LOOPEND \
        "s_cmp_lt_u32 %[it], %[maxit]\n\t" \
        "s_cbranch_scc1 .Lloop%=\n" \
        NASTAR_ASM4_EXITS

#define NASTAR_ASM4_OPERANDS \
        : [it] "+s"(it), [sel] "=s"(sel), [mkey] "=s"(mkey) \
        : [l8] "v"(v_l8), [dr] "v"(v_dr), [cmask] "v"(v_cmask), [dcc] "v"(v_dcc), [pcode] "v"(v_pcode), [cls] "v"(v_cls), \
          [vminf] "v"(v_minf), [goal] "s"(goal_idx), [gr] "s"(goal_r), [gc] "s"(goal_c), \
          [maxit] "s"(max_iters), [cgr] "s"(cgr), [comg] "s"(comg), [csq] "s"(csq), [crcp] "s"(rcp_sqrtW), \
          [mnb] "s"(m_nb), [logp] "s"(logp), \
          [CMIN] "i"(CMIN), [CMIN16] "i"(CMIN + 16), [PDIR] "i"(PDIR), [LOGW] "i"(LOGW), [WM1] "i"((1 << LOGW) - 1), [W] "i"(1 << LOGW) \
        : "memory", "vcc", "scc", "v20", "v21", "v22", "v23", "v24", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", \
          "v34", "v35", "v36", "v37", "v40", "v41", "v42", "v43", "v46", "v47", "v48", "v49", "v50", "s40", "s41", "s42", \
          "s43", "s44", "s54", "s55", "s53", "v53", "v54", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "s48", "s49", \
          "s50", "s51", "v55", "v56", "v57", "s56", "s58", "s59", "s60", "s61"

// Runs selection steps until the goal is selected, the open list is empty or `max_iters` steps were executed.  Returns the goal index
// (goal selected, its own step not yet counted in iters), -1 (open list empty) or -2 (budget exhausted).
// Preconditions: every cost >= +0, 0 <= g_ratio <= 1 (keys are raw float bits); idle chunk entries hold (key all ones, cell = goal);
// kHalf: g_ratio == 0.5 exactly (the start's key must have been formed the same way, compact_open_start with gr = omg = 1);
// kUnit: the LDS holds the unit-cost layout (AsmLayoutUnit).
// kDive: the "dive" fast path of nastar_search_asm3.hip.h (the next selection is a just-relaxed neighbour whose key beats the previous minimum)
template <int LOGW, bool kLog, bool kDive, bool kHalf, bool kUnit>
__device__ __forceinline__ int search_loop_asm4(float cgr, float comg, float csq, int lane, int goal_idx, int goal_r, int goal_c,
                                                int max_iters, int& iters, float rcp_sqrtW, int* log_row)
{
    using LG = AsmLayout<LOGW>;
    using LU = AsmLayoutUnit<LOGW>;
    constexpr int CPL = LG::CPL;
    constexpr int CMIN = kUnit ? LU::CMIN : LG::CMIN;
    constexpr int PDIR = kUnit ? LU::PDIR : LG::PDIR;
    static_assert((CPL == 1 || CPL == 4) && LG::HW >= 256, "1 or 4 chunk minima per lane, chunks inside one map row");
    int dr, dc;
    neighbour_delta(lane & 7, dr, dc);
    const bool is_nb = lane < 8, is_chk = (lane & 48) == 16;
    const int v_dr = is_nb ? dr : 0;
    const int v_dcc = is_nb ? dc : (is_chk ? (lane & 15) : 0);
    const uint32_t v_cmask = is_chk ? 0xFFFFFFF0u : 0xFFFFFFFFu;
    const uint32_t v_cls = is_chk ? 0x1F8u : 0u;  // finite = open
    const uint32_t v_pcode = P_PASS | (uint32_t)(lane & 7);
    const uint32_t v_l8 = (uint32_t)lane * 8u * CPL;
    const float v_minf = NASTAR_NEG_INF;
    const unsigned long long m_nb = 0xFFull;
    int it = __builtin_amdgcn_readfirstlane(iters);
    goal_idx = __builtin_amdgcn_readfirstlane(goal_idx);
    goal_r = __builtin_amdgcn_readfirstlane(goal_r);
    goal_c = __builtin_amdgcn_readfirstlane(goal_c);
    max_iters = __builtin_amdgcn_readfirstlane(max_iters);
    int sel;
    uint32_t mkey = 0;
    unsigned long long logp = reinterpret_cast<unsigned long long>(log_row);
#define NASTAR_A4_SH_G "3"
#define NASTAR_A4_SH_U "2"
#define NASTAR_A4_RDS_G "ds_read_b64 v[28:29], v27\n\t"
#define NASTAR_A4_RDS_U "ds_read_b32 v28, v27\n\t"
#define NASTAR_A4_RDC_G "ds_read_b64 v[30:31], v26\n\t"
#define NASTAR_A4_RDC_U "ds_read_b32 v30, v26\n\t"
#define NASTAR_A4_ADDH_G "v_add_f32 v34, v34, v31\n\t"
#define NASTAR_A4_ADDH_U "v_add_f32 v34, 1.0, v34\n\t"
#define NASTAR_A4_ADDG_G "v_add_f32 v40, v28, v29\n\t"
#define NASTAR_A4_ADDG_U "v_add_f32 v40, 1.0, v28\n\t"
#define NASTAR_A4_MULH "v_mul_f32 v34, %[comg], v34\n\t"
#define NASTAR_A4_MULG "v_mul_f32 v41, %[cgr], v41\n\t"
#define NASTAR_A4_EXPAND(L, MH, MG, SET55) \
    NASTAR_ASM4_X_PREFIX(NASTAR_A4_SH_##L, NASTAR_A4_RDS_##L) NASTAR_ASM3_X_CLOSE NASTAR_A4_RDC_##L NASTAR_ASM3_X_HEUR NASTAR_ASM3_X_WAIT \
        NASTAR_ASM4_X_KEY(NASTAR_A4_ADDH_##L, NASTAR_A4_ADDG_##L, MH, MG) NASTAR_ASM3_X_RELAX(SET55)
    /* entry -> select -> [log] expand, prefetch -> budget test -> select ... */
#define NASTAR_A4_BODY(N, L, MH, MG, LOGPART) \
    NASTAR_ASM4_ENTRY NASTAR_ASM_READ_##N NASTAR_ASM_LOOPTOP NASTAR_ASM_LOCALMIN_##N NASTAR_ASM4_SELECT LOGPART NASTAR_A4_EXPAND(L, MH, MG, ) \
        NASTAR_ASM_READ_##N NASTAR_ASM4_LOOPEND
    /* entry -> select; [dive lookup ->] selected: log, expand, prefetch, budget + dive test -> dive | select -> selected */
#define NASTAR_A4_BODY_DIVE(N, L, MH, MG, LOGPART) \
    NASTAR_ASM4_ENTRY NASTAR_ASM_READ_##N "s_branch .Lsel%=\n" NASTAR_ASM4_DIVE_LOOKUP LOGPART \
        NASTAR_A4_EXPAND(L, MH, MG, "v_cmp_lt_u32_e64 s[60:61], v47, s40\n\t") NASTAR_ASM_READ_##N NASTAR_ASM3_DIVE_TEST ".Lsel%=:\n\t" \
        "s_waitcnt lgkmcnt(0)\n\t" NASTAR_ASM_LOCALMIN_##N NASTAR_ASM4_SELECT "s_branch .Lselected%=\n" NASTAR_ASM4_EXITS
#define NASTAR_A4_RUN2(BODY, N, L, MH, MG) \
    do { \
        if constexpr (kLog) asm volatile(BODY(N, L, MH, MG, NASTAR_ASM3_LOG) NASTAR_ASM4_OPERANDS); \
        else asm volatile(BODY(N, L, MH, MG, ) NASTAR_ASM4_OPERANDS); \
    } while (0)
#define NASTAR_A4_RUN(BODY, N) \
    do { \
        if constexpr (kUnit && kHalf) NASTAR_A4_RUN2(BODY, N, U, , ); \
        else if constexpr (kUnit) NASTAR_A4_RUN2(BODY, N, U, NASTAR_A4_MULH, NASTAR_A4_MULG); \
        else if constexpr (kHalf) NASTAR_A4_RUN2(BODY, N, G, , ); \
        else NASTAR_A4_RUN2(BODY, N, G, NASTAR_A4_MULH, NASTAR_A4_MULG); \
    } while (0)
    if constexpr (CPL == 1 && kDive) NASTAR_A4_RUN(NASTAR_A4_BODY_DIVE, 1);
    else if constexpr (CPL == 1) NASTAR_A4_RUN(NASTAR_A4_BODY, 1);
    else if constexpr (kDive) NASTAR_A4_RUN(NASTAR_A4_BODY_DIVE, 4);
    else NASTAR_A4_RUN(NASTAR_A4_BODY, 4);
#undef NASTAR_A4_RUN
#undef NASTAR_A4_RUN2
#undef NASTAR_A4_BODY_DIVE
#undef NASTAR_A4_BODY
#undef NASTAR_A4_EXPAND
    iters = it;
    return (sel >= 0 && mkey == 0xFFFFFFFFu) ? -1 : sel;
}

// Measured and dropped (round 4, profiles/r04/dive_switch_*.json): a per-map dive switch for the one-chunk-minimum sizes -- every map ran
// its first 32 steps in the diving loop (a hit counter in the lookup), the rest in the diving or the plain loop according to the hit
// count (>= 12).  Same results by construction (stream-equality test), but maze32 151.3 vs 148.9 us per 4096-map launch (+1.6 %: the
// probe's test instruction and the second loop entry) for rand32 70.5 vs 71.4 us (-1.3 %): the headline batch is mazes.  32x32 and 16x16
// keep the plain loop, 64x64 always dives.

}  // namespace nastar
