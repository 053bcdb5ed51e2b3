// nastar_search_asm3_abl.hip.h -- DEV builds only (make DEV=1, NASTAR_ABLATE=300+V): the round-3 step loop with ONE section removed or
// altered, run for the whole step budget with both exits disabled.  Results are garbage by design; only the time per step means
// anything (tools/probe_ablate3.py): the difference to V = 0 is what the section contributes to the step's critical path.
#pragma once
#include "nastar_search_asm3.hip.h"

namespace nastar {

/* the selection with its two exits replaced by s_nop (the compares stay) */
#define NASTAR_ABL3_QUAD \
        "v_min_u32_dpp v22, v21, v21 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t" \
        "v_min_u32_dpp v23, v21, v21 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t" \
        "v_min_u32_dpp v22, v21, v22 quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0xf\n\t" \
        "v_min_u32 v22, v22, v23\n\t"
#define NASTAR_ABL3_IT "s_add_u32 %[it], %[it], 1\n\t"
#define NASTAR_ABL3_ROWS(NOP0, NOP1) \
        NOP0 \
        "v_min_u32_dpp v22, v22, v22 row_half_mirror row_mask:0xf bank_mask:0xf\n\t" \
        NOP1 \
        "v_min_u32_dpp v22, v22, v22 row_mirror row_mask:0xf bank_mask:0xf\n\t" \
        NOP1 \
        "v_min_u32_dpp v22, v22, v22 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t" \
        NOP1 \
        "v_min_u32_dpp v22, v22, v22 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t" \
        NOP0
#define NASTAR_ABL3_M63 "v_readlane_b32 s40, v22, 63\n\t"
#define NASTAR_ABL3_M0 "v_readlane_b32 s40, v21, 0\n\t"
#define NASTAR_ABL3_EMPTYCHK "s_cmp_eq_u32 s40, -1\n\t" "s_nop 0\n\t"
#define NASTAR_ABL3_PICK \
        "v_cmp_eq_u32 vcc, s40, v21\n\t" \
        "s_ff1_i32_b64 s41, vcc\n\t" \
        "s_and_b32 s41, s41, 63\n\t" \
        "v_readlane_b32 s42, v20, s41\n\t"
#define NASTAR_ABL3_PICK0 "v_readlane_b32 s42, v20, 0\n\t"
#define NASTAR_ABL3_GOALCHK "s_cmp_eq_u32 s42, %[goal]\n\t" "s_nop 0\n\t"
#define NASTAR_ABL3_NOP0 "s_nop 0\n\t"
#define NASTAR_ABL3_NOP1 "s_nop 1\n\t"
#define NASTAR_ABL3_SELECT_FULL NASTAR_ABL3_QUAD NASTAR_ABL3_IT NASTAR_ABL3_ROWS(NASTAR_ABL3_NOP0, NASTAR_ABL3_NOP1) NASTAR_ABL3_M63 NASTAR_ABL3_EMPTYCHK
#define NASTAR_ABL3_END \
        "s_cmp_lt_u32 %[it], %[maxit]\n\t" \
        "s_cbranch_scc1 .Lloop%=\n" \
        ".Lbudget%=:\n\t" \
        "s_waitcnt lgkmcnt(0)\n\t" \
        "s_mov_b32 %[sel], -2\n\t"
/* relaxation without its LDS instructions / without the atomic only */
#define NASTAR_ABL3_RELAX_NOLDS \
        "v_cmp_class_f32_e64 s[58:59], v30, %[cls]\n\t" \
        "s_mov_b64 exec, s[54:55]\n\t" \
        "v_cmpx_gt_f32 vcc, v30, v40\n\t" \
        "s_or_b64 exec, exec, s[58:59]\n\t" \
        "s_mov_b64 exec, -1\n\t"
#define NASTAR_ABL3_RELAX_NOATOMIC \
        "v_cmp_class_f32_e64 s[58:59], v30, %[cls]\n\t" \
        "s_mov_b64 exec, s[54:55]\n\t" \
        "v_cmpx_gt_f32 vcc, v30, v40\n\t" \
        "ds_write_b32 v26, v40\n\t" \
        "ds_write_b8 v46, %[pcode] offset:%[PDIR]\n\t" \
        "s_or_b64 exec, exec, s[58:59]\n\t" \
        "s_mov_b64 exec, -1\n\t"
#define NASTAR_ABL3_RELAX_NOWRITES \
        "v_cmp_class_f32_e64 s[58:59], v30, %[cls]\n\t" \
        "s_mov_b64 exec, s[54:55]\n\t" \
        "v_cmpx_gt_f32 vcc, v30, v40\n\t" \
        "s_or_b64 exec, exec, s[58:59]\n\t" \
        "ds_min_u64 v50, v[46:47] offset:%[CMIN]\n\t" \
        "s_mov_b64 exec, -1\n\t"
#define NASTAR_ABL3_CLOSE_NOEXEC /* the two stores of lane 8 by every lane (same address -> no EXEC switch): what the two s_mov cost */ \
        "ds_write_b32 v27, %[vminf]\n\t" \
        "ds_write_b64 v50, v[48:49] offset:%[CMIN]\n\t"

#define NASTAR_ABL3_BODY(SELECT, PICK, PREFIX, CLOSE, READCELL, HEUR, WAIT, KEY, RELAX, READBACK, EXTRA) \
    NASTAR_ASM_ENTRY NASTAR_ASM_READ_1 NASTAR_ASM_LOOPTOP SELECT PICK NASTAR_ABL3_GOALCHK PREFIX CLOSE READCELL HEUR WAIT KEY RELAX READBACK \
        EXTRA NASTAR_ABL3_END

template <int LOGW, int V>
__device__ __forceinline__ int compact_search_loop_asm3_abl(const CompactDims& d, int lane, int goal_idx, int goal_r, int goal_c,
                                                            int max_iters, int& iters, float rcp_sqrtW)
{
    using L = AsmLayout<LOGW>;
    static_assert(L::CPL == 1, "ablation probe: one chunk minimum per lane");
    int dr, dc;
    neighbour_delta(lane & 7, dr, dc);
    const bool is_nb = lane < 8, is_chk = (lane & 48) == 16;
    const int v_dr = is_nb ? dr : 0;
    const int v_dcc = is_nb ? dc : (is_chk ? (lane & 15) : 0);
    const uint32_t v_cmask = is_chk ? 0xFFFFFFF0u : 0xFFFFFFFFu;
    const uint32_t v_cls = is_chk ? 0x1F8u : 0u;
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
    unsigned long long logp = 0;
#define X_SEL NASTAR_ABL3_SELECT_FULL
#define X_PICK NASTAR_ABL3_PICK
#define X_PRE NASTAR_ASM3_X_PREFIX
#define X_CLOSE NASTAR_ASM3_X_CLOSE
#define X_RC NASTAR_ASM3_X_READCELL
#define X_HEUR NASTAR_ASM3_X_HEUR
#define X_WAIT NASTAR_ASM3_X_WAIT
#define X_KEY NASTAR_ASM3_X_KEY
#define X_RELAX NASTAR_ASM3_X_RELAX()
#define X_RB NASTAR_ASM_READ_1
#define RUN(...) asm volatile(NASTAR_ABL3_BODY(__VA_ARGS__) NASTAR_ASM3_OPERANDS)
    if constexpr (V == 0) RUN(X_SEL, X_PICK, X_PRE, X_CLOSE, X_RC, X_HEUR, X_WAIT, X_KEY, X_RELAX, X_RB, );
    else if constexpr (V == 1)  /* - the four row-level reduction stages */
        RUN(NASTAR_ABL3_QUAD NASTAR_ABL3_IT NASTAR_ABL3_M63 NASTAR_ABL3_EMPTYCHK, X_PICK, X_PRE, X_CLOSE, X_RC, X_HEUR, X_WAIT, X_KEY, X_RELAX, X_RB, );
    else if constexpr (V == 2)  /* - the whole reduction */
        RUN(NASTAR_ABL3_IT NASTAR_ABL3_M0 NASTAR_ABL3_EMPTYCHK, X_PICK, X_PRE, X_CLOSE, X_RC, X_HEUR, X_WAIT, X_KEY, X_RELAX, X_RB, );
    else if constexpr (V == 3)  /* - compare + find-first of the pick */
        RUN(X_SEL, NASTAR_ABL3_PICK0, X_PRE, X_CLOSE, X_RC, X_HEUR, X_WAIT, X_KEY, X_RELAX, X_RB, );
    else if constexpr (V == 4)  /* - heuristic */
        RUN(X_SEL, X_PICK, X_PRE, X_CLOSE, X_RC, , X_WAIT, X_KEY, X_RELAX, X_RB, );
    else if constexpr (V == 5)  /* - key arithmetic */
        RUN(X_SEL, X_PICK, X_PRE, X_CLOSE, X_RC, X_HEUR, X_WAIT, , X_RELAX, X_RB, );
    else if constexpr (V == 6)  /* - the relaxation's three LDS instructions */
        RUN(X_SEL, X_PICK, X_PRE, X_CLOSE, X_RC, X_HEUR, X_WAIT, X_KEY, NASTAR_ABL3_RELAX_NOLDS, X_RB, );
    else if constexpr (V == 7)  /* - the atomic only */
        RUN(X_SEL, X_PICK, X_PRE, X_CLOSE, X_RC, X_HEUR, X_WAIT, X_KEY, NASTAR_ABL3_RELAX_NOATOMIC, X_RB, );
    else if constexpr (V == 8)  /* - closing s* / emptying its chunk entry (2 stores + 2 EXEC switches) */
        RUN(X_SEL, X_PICK, X_PRE, , X_RC, X_HEUR, X_WAIT, X_KEY, X_RELAX, X_RB, );
    else if constexpr (V == 9)  /* - the cell read and its wait */
        RUN(X_SEL, X_PICK, X_PRE, X_CLOSE, , X_HEUR, , X_KEY, X_RELAX, X_RB, );
    else if constexpr (V == 10)  /* - the read-back of the chunk minima */
        RUN(X_SEL, X_PICK, X_PRE, X_CLOSE, X_RC, X_HEUR, X_WAIT, X_KEY, X_RELAX, , );
    else if constexpr (V == 11)  /* + one extra taken branch */
        RUN(X_SEL, X_PICK, X_PRE, X_CLOSE, X_RC, X_HEUR, X_WAIT, X_KEY, X_RELAX, X_RB, "s_branch .Lx%=\n" ".Lx%=:\n\t");
    else if constexpr (V == 12)  /* - the reduction's wait states (illegal, timing only) */
        RUN(NASTAR_ABL3_QUAD NASTAR_ABL3_IT NASTAR_ABL3_ROWS(, ) NASTAR_ABL3_M63 NASTAR_ABL3_EMPTYCHK, X_PICK, X_PRE, X_CLOSE, X_RC, X_HEUR, X_WAIT, X_KEY, X_RELAX, X_RB, );
    else if constexpr (V == 13)  /* - everything between the selection and the loop end */
        RUN(X_SEL, X_PICK, , , , , , , , X_RB, );
    else if constexpr (V == 14)  /* only the skeleton: counter, read-back, loop */
        RUN(NASTAR_ABL3_IT, , , , , , , , , X_RB, );
    else if constexpr (V == 15)  /* - the relaxation's two plain stores (the atomic stays) */
        RUN(X_SEL, X_PICK, X_PRE, X_CLOSE, X_RC, X_HEUR, X_WAIT, X_KEY, NASTAR_ABL3_RELAX_NOWRITES, X_RB, );
    else if constexpr (V == 16)  /* closing stores without the EXEC switches */
        RUN(X_SEL, X_PICK, X_PRE, NASTAR_ABL3_CLOSE_NOEXEC, X_RC, X_HEUR, X_WAIT, X_KEY, X_RELAX, X_RB, );
    else if constexpr (V == 17)  /* - the address prefix (stale registers) */
        RUN(X_SEL, X_PICK, , X_CLOSE, X_RC, X_HEUR, X_WAIT, X_KEY, X_RELAX, X_RB, );
    else if constexpr (V == 18)  /* - both compares of the disabled exits */
        asm volatile(NASTAR_ASM_ENTRY NASTAR_ASM_READ_1 NASTAR_ASM_LOOPTOP NASTAR_ABL3_QUAD NASTAR_ABL3_IT
                     NASTAR_ABL3_ROWS(NASTAR_ABL3_NOP0, NASTAR_ABL3_NOP1) NASTAR_ABL3_M63 X_PICK X_PRE X_CLOSE X_RC X_HEUR X_WAIT X_KEY X_RELAX X_RB
                     NASTAR_ABL3_END NASTAR_ASM3_OPERANDS);
    else static_assert(V < 0, "unknown ablation variant");
#undef RUN
#undef X_SEL
#undef X_PICK
#undef X_PRE
#undef X_CLOSE
#undef X_RC
#undef X_HEUR
#undef X_WAIT
#undef X_KEY
#undef X_RELAX
#undef X_RB
    iters = it;
    return sel;
}

}  // namespace nastar
