// nastar_search_asm.hip.h -- the selection/expansion loop of the compact forward kernel (nastar_search_compact.hip.h) as ONE
// hand-scheduled gfx950 instruction stream, for square power-of-two maps: 16x16, 32x32 (the headline configuration) and 64x64.
//
// Why assembly: a map's search is a serial chain of steps executed by ONE wavefront, and for the 4096-map batch every map is
// resident from t = 0 (16 per CU), so the launch lasts as long as the longest chain (472 steps) x the latency of one step.
// Micro-benchmarks of one gfx950 wavefront (tools/ubench/lat2.hip, profiles/r02/ubench_lat2.txt) give the price list:
//     any VALU / SALU instruction                    ~4.7 cycles of issue (dependent or not)
//     s_nop 1 (what hipcc pads a DPP source with)     8.1
//     v_cmp -> s_and_b64 -> v_cndmask (mask via SALU) 28.5   (vs 8.6 for v_cmp -> v_cndmask through VCC)
//     v_cmp -> s_ff1 -> v_readlane chain             ~11 per instruction
//     taken branch                                   ~27-30
//     dependent LDS read                              60-65;  write or u64 atomic -> read of another address  77-96
// hipcc's code for the same step spends ~1000 cycles: 28 SALU instructions (lane predicates held as 64-bit SGPR masks and
// combined with s_and/s_or between the v_cmp that makes them and the v_cndmask that uses them), 10 s_nop, full-width dummy LDS
// stores.  Here: every predicate is ONE v_cmp consumed through VCC (roles are folded into per-lane constant operands: lanes
// that are not neighbours get offsets that make their cell s* itself and +inf as the value to beat), the hazard wait states
// of the DPP reductions are filled with independent work, stores run under EXEC masks written by v_cmpx / s_mov (<= 9 active
// lanes per LDS store instead of 64), and there is one taken branch per step.
//
// Hazard rules honoured inside the string (LLVM GCNHazardRecognizer, gfx940 family): VALU write VGPR -> DPP read: 2 wait states;
// VALU write SGPR/VCC -> VALU read: 2; VALU write VGPR -> v_readlane read: 1; transcendental result -> VALU use: 1.
// Semantics = compact_select + compact_expand (CPL == 1) exactly; parity is checked by the same golden / oracle tests.
#pragma once
#include "nastar_search_compact.hip.h"

namespace nastar {

// LDS byte offsets of the single-map compact layout with HW cells, 64 chunk minima (carve_compact_lds): gc | cmin | dump | pdir
template <int LOGW>
struct AsmLayout {
    static constexpr int W = 1 << LOGW;
    static constexpr int HW = W * W;
    static constexpr int CPL = (HW / 16 + 63) / 64;  // chunk minima per lane: 1 (<= 1024 cells) or 4 (64x64)
    static constexpr int CMIN = HW * 8;
    static constexpr int PDIR = CMIN + CPL * 64 * 8 + 256;
};

#define NASTAR_ASM_ENTRY \
        "s_cmp_ge_u32 %[it], %[maxit]\n\t" \
        "s_cbranch_scc1 .Lbudget%=\n\t" \
        "v_mov_b32 v48, -1\n\t" \
        "v_mov_b32 v49, -1\n\t"
/* this lane's chunk minima: one entry (<= 1024 cells) or four contiguous entries (4096 cells); .x = cell index, .y = key */ \

#define NASTAR_ASM_READ_1 \
        "ds_read_b64 v[20:21], %[l8] offset:%[CMIN]\n\t"
#define NASTAR_ASM_READ_4 \
        "ds_read_b128 v[12:15], %[l8] offset:%[CMIN]\n\t" \
        "ds_read_b128 v[16:19], %[l8] offset:%[CMIN16]\n\t"
#define NASTAR_ASM_LOOPTOP \
        ".Lloop%=:\n\t" \
        "s_waitcnt lgkmcnt(0)\n\t" /* the read was issued at the end of the previous step */
#define NASTAR_ASM_LOCALMIN_1
/* first minimal entry of the lane's four (the earlier chunk wins ties): v20 = cell index, v21 = key */ \

#define NASTAR_ASM_LOCALMIN_4 \
        "v_min_u32 v21, v13, v15\n\t" \
        "v_min3_u32 v21, v21, v17, v19\n\t" \
        "v_mov_b32 v20, v18\n\t" \
        "v_cmp_eq_u32 vcc, v21, v17\n\t" \
        "v_cmp_eq_u32_e64 s[48:49], v21, v15\n\t" \
        "v_cmp_eq_u32_e64 s[50:51], v21, v13\n\t" \
        "v_cndmask_b32 v20, v20, v16, vcc\n\t" \
        "v_cndmask_b32_e64 v20, v20, v14, s[48:49]\n\t" \
        "v_cndmask_b32_e64 v20, v20, v12, s[50:51]\n\t"
#define NASTAR_ASM_SELECT \
 /* ---- select: first cell of the minimal (key, index) chunk entry ------------------------------------------- */ \
        "v_min_u32_dpp v22, v21, v21 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t" \
        "s_nop 1\n\t" \
        "v_min_u32_dpp v22, v22, v22 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t" \
        "s_nop 1\n\t" \
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
#define NASTAR_ASM_LOG \
        "s_lshl_b32 s53, %[it], 2\n\t"                    /* sel_log[iters] = s* (store_intermediate_results side channel) */ \
        "v_mov_b32 v53, s53\n\t"                                                                                           \
        "v_mov_b32 v54, s42\n\t"                                                                                           \
        "s_mov_b64 exec, 1\n\t"                                                                                            \
        "global_store_dword v53, v54, %[logp]\n\t"                                                                         \
        "s_mov_b64 exec, -1\n\t"
#define NASTAR_ASM_EXPAND \
        "s_add_u32 %[it], %[it], 1\n\t" \
 /* ---- expand ----------------------------------------------------------------------------------------- */ \
        "s_lshr_b32 s43, s42, %[LOGW]\n\t" /* r */ \
        "s_and_b32 s44, s42, %[WM1]\n\t" /* c */ \
        "s_and_b32 s45, s42, 0xfffffff0\n\t" /* first cell of the chunk of s* */ \
        "v_add_u32 v23, s43, %[dr]\n\t" \
        "v_add_u32 v24, s44, %[dc]\n\t" \
        "v_max_u32 v23, v23, v24\n\t" \
        "v_cmp_gt_u32 vcc, %[W], v23\n\t" /* neighbour inside the map (conv2d zero padding); true for dr = dc = 0 */ \
        "v_mov_b32 v24, s42\n\t" \
        "v_mov_b32 v46, s45\n\t" \
        "v_cndmask_b32_e64 v46, v24, v46, %[mchk]\n\t" /* chunk lanes start from the chunk base, all others from s* */ \
        "v_add_u32 v46, v46, %[off]\n\t" \
        "s_and_b64 s[54:55], vcc, %[mnb]\n\t" /* in-bounds neighbour lanes */ \
        "v_cndmask_b32 v46, v24, v46, vcc\n\t" /* il: this lane's cell (s* itself for out-of-map neighbours / idle lanes) */ \
        "v_lshlrev_b32 v27, 3, v24\n\t" \
        "v_lshlrev_b32 v26, 3, v46\n\t" \
        "ds_read_b64 v[28:29], v27\n\t" /* g[s*], cost[s*] */ \
        "ds_read_b64 v[30:31], v26\n\t" /* g[il], cost[il] */ \
 /* h0 = get_heuristic at il (:26-52), in the shadow of the LDS round trip */ \
        "v_lshrrev_b32 v32, %[LOGW], v46\n\t" \
        "v_and_b32 v33, %[WM1], v46\n\t" \
        "v_subrev_u32 v32, %[gr], v32\n\t" \
        "v_subrev_u32 v33, %[gc], v33\n\t" \
        "v_cvt_f32_i32 v32, v32\n\t" \
        "v_cvt_f32_i32 v33, v33\n\t" \
        "v_mul_f32 v34, v32, v32\n\t" \
        "v_mul_f32 v35, v33, v33\n\t" \
        "v_add_f32 v34, v34, v35\n\t" \
        "v_sqrt_f32 v34, v34\n\t" \
        "v_add_f32_e64 v35, |v32|, |v33|\n\t" \
        "v_min_f32_e64 v32, |v32|, |v33|\n\t" \
        "v_sub_f32 v35, v35, v32\n\t" /* chebyshev */ \
        "v_mul_f32 v34, 0x3a83126f, v34\n\t" /* fl32(0.001) * euclid */ \
        "v_add_f32 v34, v35, v34\n\t" /* h0 */ \
        "s_lshr_b32 s47, s42, 4\n\t" \
        "s_lshl_b32 s47, s47, 3\n\t" /* byte offset of cmin[chunk of s*] */ \
        "s_and_b32 s46, s42, 15\n\t" \
        "s_add_u32 s46, s46, 16\n\t" /* the chunk lane that holds s* itself */ \
        "s_waitcnt lgkmcnt(0)\n\t" \
        "v_add_f32 v34, v34, v31\n\t" /* :191-192 h = h0 + cost */ \
        "v_mul_f32 v34, %[comg], v34\n\t" /* :206 (1-g_ratio)*h */ \
        "v_add_f32 v40, v28, v29\n\t" /* :234 g2 = g[s*] + cost[s*] */ \
        "v_cndmask_b32_e64 v41, v30, v40, %[mnb]\n\t" /* neighbour lanes key g2, chunk lanes their own g */ \
        "v_mul_f32 v41, %[cgr], v41\n\t" \
        "v_add_f32 v41, v41, v34\n\t" /* :206 f */ \
        "v_mul_f32 v42, %[crcp], v41\n\t" /* :207 f / sqrt(W), correctly rounded (tools/fastdiv_check.c) */ \
        "v_fma_f32 v43, -v42, %[csq], v41\n\t" \
        "v_fma_f32 v42, v43, %[crcp], v42\n\t" \
        "v_ashrrev_i32 v43, 31, v42\n\t" \
        "v_cmp_eq_u32 vcc, s46, %[lane]\n\t" \
        "v_bitop3_b32 v47, v43, v42, %[msb] bitop3:0x36\n\t" /* order-preserving u32 key: [v46:v47] = (cell, key) */ \
        "v_cndmask_b32_e64 v45, %[vinf], v40, s[54:55]\n\t" /* value to beat: g2 for in-map neighbour lanes, +inf elsewhere */ \
        "v_cndmask_b32 v44, %[loc], %[vinf], vcc\n\t" /* chunk lanes: -inf (any finite g is open), +inf for s* itself and other lanes */ \
        "v_lshrrev_b32 v50, 4, v46\n\t" \
        "v_lshlrev_b32 v50, 3, v50\n\t" /* byte offset of cmin[chunk of il] */ \
        "v_mov_b32 v52, s47\n\t" \
 /* lane 8 closes s* (:222-225) and empties its chunk's entry; the chunk's open cells (without s*) then re-enter it through \
    the same 64-bit atomic minimum the relaxed neighbours use: (key, cell) lexicographic = first flat index on ties */ \
        "s_mov_b64 exec, 0x100\n\t" \
        "ds_write_b32 v27, %[vminf]\n\t" \
        "ds_write_b64 v52, v[48:49] offset:%[CMIN]\n\t" \
        "s_mov_b64 exec, -1\n\t" \
        "v_cmpx_gt_f32 vcc, v30, v44\n\t" /* chunk lanes other than s* whose g > -inf ... */ \
        "v_cmpx_ge_f32 vcc, 0x7f7fffff, v30\n\t" /* ... and finite: open */ \
        "ds_min_u64 v50, v[46:47] offset:%[CMIN]\n\t" \
        "s_mov_b64 exec, -1\n\t" \
        "v_cmpx_gt_f32 vcc, v30, v45\n\t" /* :229,:235 g[n] > g2 on in-map neighbour lanes */ \
        "ds_write_b32 v26, v40\n\t" /* :238 g[n] = g2 */ \
        "ds_write_b8 v46, %[pcode] offset:%[PDIR]\n\t" /* :246-249 parent = s* */ \
        "ds_min_u64 v50, v[46:47] offset:%[CMIN]\n\t" /* :242 (key, n) enters its chunk's minimum */ \
        "s_mov_b64 exec, -1\n\t"
#define NASTAR_ASM_LOOPEND \
        "s_cmp_lt_u32 %[it], %[maxit]\n\t" \
        "s_cbranch_scc1 .Lloop%=\n" \
        ".Lbudget%=:\n\t" \
        "s_waitcnt lgkmcnt(0)\n\t" /* the prefetched chunk minima must have landed before v20/v21 are released */ \
        "s_mov_b32 %[sel], -2\n\t" \
        "s_branch .Lend%=\n" \
        ".Lempty%=:\n\t" \
        "s_mov_b32 %[sel], -1\n\t" \
        "s_branch .Lend%=\n" \
        ".Lgoal%=:\n\t" \
        "s_mov_b32 %[sel], s42\n" \
        ".Lend%=:\n\t"
#define NASTAR_ASM_OPERANDS \
        : [it] "+s"(it), [sel] "=s"(sel) \
        : [l8] "v"(v_l8), [lane] "v"(lane), [dr] "v"(v_dr), [dc] "v"(v_dc), [off] "v"(v_off), [pcode] "v"(v_pcode), \
          [vinf] "v"(v_inf), [vminf] "v"(v_minf), [loc] "v"(v_loc), [goal] "s"(goal_idx), [gr] "s"(goal_r), [gc] "s"(goal_c), \
          [maxit] "s"(max_iters), [cgr] "s"(d.gr), [comg] "s"(d.omg), [csq] "s"(d.sqrtW), [crcp] "s"(rcp_sqrtW), \
          [mnb] "s"(m_nb), [mchk] "s"(m_chk), [msb] "s"(msb), [logp] "s"(logp), \
          [CMIN] "i"(L::CMIN), [CMIN16] "i"(L::CMIN + 16), [PDIR] "i"(L::PDIR), [LOGW] "i"(LOGW), [WM1] "i"(L::W - 1), [W] "i"(L::W) \
        : "memory", "vcc", "scc", "v20", "v21", "v22", "v23", "v24", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", \
          "v34", "v35", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v52", "s40", "s41", "s42", \
          "s43", "s44", "s45", "s46", "s47", "s50", "s51", "s52", "s54", "s55", "s53", "v53", "v54", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "s48", "s49"

// Runs selection steps until the goal is selected, the open list is empty or `max_iters` steps were executed.
// Returns the goal index (goal selected, not yet counted in iters), -1 (open list empty) or -2 (budget exhausted).
template <int LOGW, bool kLog>
__device__ __forceinline__ int compact_search_loop_asm(const CompactDims& d, int lane, int goal_idx, int goal_r, int goal_c,
                                                       int max_iters, int& iters, float rcp_sqrtW, int* log_row)
{
    using L = AsmLayout<LOGW>;
    static_assert((L::CPL == 1 || L::CPL == 4) && L::HW >= 256, "1 or 4 chunk minima per lane, chunks inside one map row");
    int dr, dc;
    neighbour_delta(lane & 7, dr, dc);
    const bool is_nb = lane < 8, is_chk = (lane & 48) == 16;
    const int v_dr = is_nb ? dr : 0, v_dc = is_nb ? dc : 0;
    const int v_off = is_nb ? dr * L::W + dc : (is_chk ? (lane & 15) : 0);
    const uint32_t v_pcode = P_PASS | (uint32_t)(lane & 7);
    const uint32_t v_l8 = (uint32_t)lane * 8u * L::CPL;
    const float v_inf = NASTAR_POS_INF, v_minf = NASTAR_NEG_INF;
    const float v_loc = is_chk ? NASTAR_NEG_INF : NASTAR_POS_INF;  // lower bound of "open" for the chunk re-insertion
    const unsigned long long m_nb = 0xFFull, m_chk = 0xFFFF0000ull;
    const uint32_t msb = 0x80000000u;
    int it = __builtin_amdgcn_readfirstlane(iters);
    goal_idx = __builtin_amdgcn_readfirstlane(goal_idx);  // wave-uniform by construction; the "s" constraints need it provable
    goal_r = __builtin_amdgcn_readfirstlane(goal_r);
    goal_c = __builtin_amdgcn_readfirstlane(goal_c);
    max_iters = __builtin_amdgcn_readfirstlane(max_iters);
    int sel;
    unsigned long long logp = reinterpret_cast<unsigned long long>(log_row);
#define NASTAR_ASM_BODY(N, LOGPART) \
    NASTAR_ASM_ENTRY NASTAR_ASM_READ_##N NASTAR_ASM_LOOPTOP NASTAR_ASM_LOCALMIN_##N NASTAR_ASM_SELECT LOGPART NASTAR_ASM_EXPAND \
        NASTAR_ASM_READ_##N NASTAR_ASM_LOOPEND
    if constexpr (L::CPL == 1) {
        if constexpr (kLog) asm volatile(NASTAR_ASM_BODY(1, NASTAR_ASM_LOG) NASTAR_ASM_OPERANDS);
        else asm volatile(NASTAR_ASM_BODY(1, ) NASTAR_ASM_OPERANDS);
    } else {
        if constexpr (kLog) asm volatile(NASTAR_ASM_BODY(4, NASTAR_ASM_LOG) NASTAR_ASM_OPERANDS);
        else asm volatile(NASTAR_ASM_BODY(4, ) NASTAR_ASM_OPERANDS);
    }
#undef NASTAR_ASM_BODY
    iters = it;
    return sel;
}

}  // namespace nastar
