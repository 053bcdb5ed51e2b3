// nastar_backward_replay_asm.hip.h -- nastar_backward_replay.hip.h for square power-of-two maps of <= 1024 cells (32x32: the
// reference's training configuration) with the replay loop as one hand-scheduled gfx950 instruction stream.
//
// Same algorithm and arithmetic as nastar_backward_replay_kernel (event accounting of the softmax gradient while replaying the
// forward's selection log); what changes is the cost of a step for the lone wavefront that owns a map (price list: see
// nastar_search_asm.hip.h): hipcc's loop is ~230 instructions with the lane predicates combined on the scalar unit and
// data-dependent branches around every store; here a step is ~110 instructions, two LDS reads wide:
//   * per-cell record {g, cost, G, stamp} = 16 B: ONE ds_read_b128 per lane fetches everything a relaxation needs; g and stamp
//     are written back with one ds_write2_b32;
//   * the (A, B) history sits behind the records, 16 B per step: 16 KiB + 4 KiB = 20,480 B for 32x32 at the training budget
//     (256 steps) -> 8 maps per CU.  S, D (fp64) alias the LAST history slot: that slot is only written by step max_steps-1,
//     i.e. when the budget ran out, after which S and D are not needed any more, and it is never read (cells opened by the last
//     executed step take their stamp from registers);
//   * "opened" lanes add (+v_new, +G v_new), "left the open list / re-keyed" lanes add (-v_old, -G v_old) with ds_add_f64 under
//     EXEC masks (no select-with-zero arithmetic), the interval of a leaving cell is closed one step later (its history entry is
//     fetched by a ds_read_b128 issued in the step that closes it);
//   * the goal step and the reference's batch-coupled fixed-point steps (rare) and the final sweep stay in C++.
// History indexing: entry e = (A, B) after step e-1, stored at HIST + (e-1)*16; the start cell carries stamp 1 (written in step 0
// before it is read: at step 0 the open list is {start}, whose softmax gradient is exactly zero).
#pragma once
#include "nastar_backward_replay.hip.h"

namespace nastar {

struct BwdRec {
    float g, cost, G;
    uint32_t stamp;
};

template <int LOGW>
struct BwdAsmLayout {
    static constexpr int W = 1 << LOGW;
    static constexpr int HW = W * W;
    static constexpr int HIST = HW * 16;
};
__host__ __device__ inline size_t bwdr_asm_lds_bytes(int HW, int max_steps) { return (size_t)HW * 16 + (size_t)max_steps * 16; }

#define NASTAR_BWD_ASM_LOOP \
        "v_mov_b32 v60, 0\n\tv_mov_b32 v61, 0\n\tv_mov_b32 v62, 0\n\tv_mov_b32 v63, 0\n\t" /* A = B = 0 */ \
        "s_mov_b64 s[56:57], 0\n\t" /* no pending interval */ \
        "ds_read_b128 v[80:83], %[sda]\n\t" /* S, D */ \
        "s_cmp_ge_u32 %[t], %[nloop]\n\t" \
        "s_cbranch_scc1 .Lexit%=\n" \
        ".Lstep%=:\n\t" \
        "s_and_b32 s61, %[t], 63\n\t" \
        "s_cmp_eq_u32 s61, 0\n\t" \
        "s_cbranch_scc1 .Lloadlog%=\n" /* every 64th step: next block of the selection log */ \
        ".Lhavelog%=:\n\t" \
        "s_waitcnt lgkmcnt(0)\n\t" /* S, D and the pending intervals' history entries have landed */ \
        "v_readlane_b32 s42, v59, s61\n\t" /* s* of this step */ \
        /* ---- close the intervals that ended in the previous step: kfac v (G dA - dB) -> grad_cost ---- */ \
        "s_mov_b64 exec, s[56:57]\n\t" \
        "v_add_f64 v[68:69], v[70:71], -v[64:65]\n\t" \
        "v_add_f64 v[72:73], v[74:75], -v[66:67]\n\t" \
        "v_cvt_f32_f64 v76, v[68:69]\n\t" \
        "v_cvt_f32_f64 v77, v[72:73]\n\t" \
        "v_mul_f32 v76, v57, v76\n\t" \
        "v_sub_f32 v76, v76, v77\n\t" \
        "v_mul_f32 v76, v56, v76\n\t" \
        "global_atomic_add_f32 v58, v76, %[gbase]\n\t" \
        "s_mov_b64 exec, -1\n\t" \
        /* ---- softmax of this step over the open list: A += 1/S, B += D/S^2 ---- */ \
        "v_cvt_f32_f64 v84, v[80:81]\n\t" \
        "v_cvt_f32_f64 v85, v[82:83]\n\t" \
        "v_rcp_f32 v84, v84\n\t" \
        "s_lshr_b32 s43, s42, %[LOGW]\n\t" /* r */ \
        "s_and_b32 s44, s42, %[WM1]\n\t" /* c */ \
        "v_mul_f32 v85, v85, v84\n\t" \
        "v_cvt_f64_f32 v[86:87], v84\n\t" \
        "v_mul_f32 v85, v85, v84\n\t" \
        "v_add_f64 v[60:61], v[60:61], v[86:87]\n\t" \
        "v_cvt_f64_f32 v[86:87], v85\n\t" \
        "v_add_f64 v[62:63], v[62:63], v[86:87]\n\t" \
        /* ---- expansion of s* (:222-249): lanes 0..7 relax the neighbours, lane 8 closes s* ---- */ \
        "v_add_u32 v23, s43, %[dr]\n\t" \
        "v_add_u32 v24, s44, %[dc]\n\t" \
        "v_max_u32 v23, v23, v24\n\t" \
        "v_cmp_gt_u32 vcc, %[W], v23\n\t" \
        "v_mov_b32 v24, s42\n\t" \
        "v_add_u32 v46, v24, %[off]\n\t" \
        "s_and_b64 s[54:55], vcc, %[mnb]\n\t" /* in-map neighbour lanes */ \
        "v_cndmask_b32 v46, v24, v46, vcc\n\t" /* il: this lane's cell (s* for lane 8, idle lanes, out-of-map neighbours) */ \
        "v_lshlrev_b32 v27, 4, v24\n\t" \
        "v_lshlrev_b32 v26, 4, v46\n\t" \
        "ds_read_b64 v[28:29], v27\n\t" /* g[s*], cost[s*] */ \
        "ds_read_b128 v[30:33], v26\n\t" /* g, cost, G, stamp of il */ \
        "v_lshrrev_b32 v34, %[LOGW], v46\n\t" /* h0 = get_heuristic at il (:26-52) in the shadow of the reads */ \
        "v_and_b32 v35, %[WM1], v46\n\t" \
        "v_subrev_u32 v34, %[gr], v34\n\t" \
        "v_subrev_u32 v35, %[gc], v35\n\t" \
        "v_cvt_f32_i32 v34, v34\n\t" \
        "v_cvt_f32_i32 v35, v35\n\t" \
        "v_mul_f32 v36, v34, v34\n\t" \
        "v_mul_f32 v37, v35, v35\n\t" \
        "v_add_f32 v36, v36, v37\n\t" \
        "v_sqrt_f32 v36, v36\n\t" \
        "v_add_f32_e64 v37, |v34|, |v35|\n\t" \
        "v_min_f32_e64 v34, |v34|, |v35|\n\t" \
        "v_sub_f32 v37, v37, v34\n\t" \
        "v_mul_f32 v36, 0x3a83126f, v36\n\t" \
        "v_add_f32 v34, v37, v36\n\t" /* h0 */ \
        "s_add_u32 s62, %[t], 1\n\t" \
        "s_lshl_b32 s63, %[t], 4\n\t" /* history entry t+1 lives at HIST + t*16 */ \
        "v_mov_b32 v47, s62\n\t" \
        "v_mov_b32 v52, s63\n\t" \
        "s_mov_b64 exec, 1\n\t" \
        "ds_write_b128 v52, v[60:63] offset:%[HIST]\n\t" /* (A, B) after this step, before any stamp of this step is read */ \
        "s_mov_b64 exec, -1\n\t" \
        "s_waitcnt lgkmcnt(1)\n\t" /* the two reads (the history write is still in flight) */ \
        "v_add_f32 v34, v34, v31\n\t" /* :191-192 h = h0 + cost */ \
        "v_mul_f32 v34, %[comg], v34\n\t" /* :206 */ \
        "v_add_f32 v40, v28, v29\n\t" /* :234 g2 */ \
        "v_mul_f32 v41, %[cgr], v40\n\t" \
        "v_mul_f32 v43, %[cgr], v30\n\t" \
        "v_add_f32 v41, v41, v34\n\t" /* f of the relaxed neighbour */ \
        "v_add_f32 v43, v43, v34\n\t" /* f this cell had on the open list */ \
        "v_mul_f32 v42, %[crcp], v41\n\t" \
        "v_mul_f32 v44, %[crcp], v43\n\t" \
        "v_fma_f32 v41, -v42, %[csq], v41\n\t" \
        "v_fma_f32 v43, -v44, %[csq], v43\n\t" \
        "v_fma_f32 v42, v41, %[crcp], v42\n\t" /* :207 q_new */ \
        "v_fma_f32 v44, v43, %[crcp], v44\n\t" /* q_old */ \
        "v_mul_f32 v42, 0xbfb8aa3b, v42\n\t" \
        "v_mul_f32 v44, 0xbfb8aa3b, v44\n\t" \
        "v_exp_f32 v42, v42\n\t" /* v_new = exp(-q_new) */ \
        "v_exp_f32 v44, v44\n\t" /* v_old */ \
        "v_cndmask_b32_e64 v45, %[vinf], v40, s[54:55]\n\t" /* value to beat: g2 on in-map neighbour lanes, +inf elsewhere */ \
        "v_cmp_lt_f32_e64 s[58:59], |v30|, %[vinf]\n\t" /* was on the open list <=> finite g */ \
        "v_cmp_gt_f32 vcc, v30, v45\n\t" /* :229,:235 relaxed */ \
        "v_mul_f32 v36, v32, v42\n\t" /* G v_new */ \
        "v_cvt_f64_f32 v[48:49], v42\n\t" \
        "s_mov_b64 s[46:47], vcc\n\t" \
        "s_and_b64 s[58:59], s[58:59], vcc\n\t" /* re-keyed */ \
        "s_or_b64 s[58:59], s[58:59], 0x100\n\t" /* ... or s* itself (lane 8): these cells' intervals end now */ \
        "v_cvt_f64_f32 v[50:51], v36\n\t" \
        "s_mov_b64 exec, s[46:47]\n\t" \
        "ds_write2_b32 v26, v40, v47 offset1:3\n\t" /* :238 g = g2 ; stamp = t+1 */ \
        "ds_add_f64 %[sda], v[48:49]\n\t" /* S += v_new */ \
        "ds_add_f64 %[sda], v[50:51] offset:8\n\t" /* D += G v_new */ \
        "s_mov_b64 exec, s[58:59]\n\t" \
        "v_mul_f32 v36, v32, v44\n\t" /* G v_old */ \
        "v_cvt_f64_f32_e64 v[48:49], -v44\n\t" \
        "v_lshlrev_b32 v53, 4, v33\n\t" /* stamp*16 */ \
        "v_cvt_f64_f32_e64 v[50:51], -v36\n\t" \
        "ds_add_f64 %[sda], v[48:49]\n\t" /* S -= v_old */ \
        "ds_add_f64 %[sda], v[50:51] offset:8\n\t" /* D -= G v_old */ \
        "ds_read_b128 v[64:67], v53 offset:%[HISTM16]\n\t" /* (A, B) when the cell was opened, consumed next step */ \
        "v_mul_f32 v56, %[ckfac], v44\n\t" \
        "v_mov_b32 v57, v32\n\t" \
        "v_lshlrev_b32 v58, 2, v46\n\t" \
        "v_mov_b64 v[70:71], v[60:61]\n\t" \
        "v_mov_b64 v[74:75], v[62:63]\n\t" \
        "s_mov_b64 s[56:57], exec\n\t" \
        "s_mov_b64 exec, 0x100\n\t" \
        "ds_write_b32 v27, %[vminf]\n\t" /* :222-225 s* joins the closed list */ \
        "s_mov_b64 exec, -1\n\t" \
        "ds_read_b128 v[80:83], %[sda]\n\t" /* S, D for the next step */ \
        "s_add_u32 %[t], %[t], 1\n\t" \
        "s_cmp_lt_u32 %[t], %[nloop]\n\t" \
        "s_cbranch_scc1 .Lstep%=\n\t" \
        "s_branch .Lexit%=\n" \
        ".Lloadlog%=:\n\t" \
        "s_lshl_b32 s62, %[t], 2\n\t" \
        "v_add_u32 v53, s62, %[l4]\n\t" \
        "v_min_u32 v53, %[logmax], v53\n\t" \
        "global_load_dword v59, v53, %[logp]\n\t" \
        "s_waitcnt vmcnt(0)\n\t" \
        "s_branch .Lhavelog%=\n" \
        ".Lexit%=:\n\t" \
        "s_waitcnt lgkmcnt(0)\n\t" \
        "v_mov_b64 %[oA], v[60:61]\n\t" \
        "v_mov_b64 %[oB], v[62:63]\n\t" \
        "v_mov_b64 %[oS], v[80:81]\n\t" \
        "v_mov_b64 %[oD], v[82:83]\n\t" \
        "v_mov_b64 %[opA], v[70:71]\n\t" \
        "v_mov_b64 %[opB], v[74:75]\n\t" \
        "v_mov_b64 %[opA0], v[64:65]\n\t" \
        "v_mov_b64 %[opB0], v[66:67]\n\t" \
        "v_mov_b32 %[opkv], v56\n\t" \
        "v_mov_b32 %[opG], v57\n\t" \
        "v_mov_b32 %[opaddr], v58\n\t" \
        "s_mov_b64 %[opm], s[56:57]\n\t"

template <int LOGW>
__global__ __launch_bounds__(64) void nastar_backward_replay_asm_kernel(const BwdRArgs a, const float rcp_sqrtW)
{
    using L = BwdAsmLayout<LOGW>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = (a.order == nullptr || (a.order_bad != nullptr && *a.order_bad != 0)) ? (int)blockIdx.x : a.order[blockIdx.x];
    if ((unsigned)b >= (unsigned)a.B_total) return;  // not a permutation: never touch memory outside the batch
    const int lane = threadIdx.x;
    CompactDims d = a.d;
    d.H = d.W = L::W;
    d.HW = L::HW;
    BwdRec* rec = reinterpret_cast<BwdRec*>(smem);
    const int max_steps = a.hist_len - 2;                      // min(max_iters, HW + 1)
    double* hist = reinterpret_cast<double*>(smem + L::HIST);  // entry e at hist[2*(e-1)]
    double* sd = hist + 2 * (max_steps - 1);                   // S, D alias the last history slot (see the header)
    const size_t off = (size_t)b * (size_t)L::HW;
    float* gout = a.grad_cost + off;

    int sidx = -1, gidx = -1;
    {
        const float4* s4 = reinterpret_cast<const float4*>(a.start + off);
        const float4* g4 = reinterpret_cast<const float4*>(a.goal + off);
        const float4* c4 = reinterpret_cast<const float4*>(a.cost + off);
        const float4* p4 = reinterpret_cast<const float4*>(a.passable + off);
        float4* o4 = reinterpret_cast<float4*>(gout);
        for (int q = lane; q < L::HW / 4; q += 64) {
            const float4 sv = s4[q], gv = g4[q], cv = c4[q], pv = p4[q];
            const int i = q << 2;
            if (sv.x != 0.f) sidx = i;
            if (sv.y != 0.f) sidx = i + 1;
            if (sv.z != 0.f) sidx = i + 2;
            if (sv.w != 0.f) sidx = i + 3;
            if (gv.x != 0.f) gidx = i;
            if (gv.y != 0.f) gidx = i + 1;
            if (gv.z != 0.f) gidx = i + 2;
            if (gv.w != 0.f) gidx = i + 3;
            rec[i + 0] = {pv.x != 0.f ? NASTAR_POS_INF : NASTAR_NEG_INF, cv.x, bwdr_upstream(a, off + i + 0), 0u};
            rec[i + 1] = {pv.y != 0.f ? NASTAR_POS_INF : NASTAR_NEG_INF, cv.y, bwdr_upstream(a, off + i + 1), 0u};
            rec[i + 2] = {pv.z != 0.f ? NASTAR_POS_INF : NASTAR_NEG_INF, cv.z, bwdr_upstream(a, off + i + 2), 0u};
            rec[i + 3] = {pv.w != 0.f ? NASTAR_POS_INF : NASTAR_NEG_INF, cv.w, bwdr_upstream(a, off + i + 3), 0u};
            o4[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    sidx = wave_max_i32(sidx);
    gidx = wave_max_i32(gidx);
    global_step_fence();  // the zeroed gradient is in L2 before any atomic touches it
    wave_sync();
    if (sidx < 0 || gidx < 0) return;

    const int goal_r = gidx >> LOGW, goal_c = gidx & (L::W - 1);
    const int n_steps = a.iters[b];
    int extra = 0;  // the reference's fixed-point steps after this map's goal step (:251) and the clamp-backward mask (:223)
    if (a.t_batch != nullptr) extra = *a.t_batch - (n_steps - 1);
    const int* log = a.sel_log + (size_t)b * (size_t)a.max_iters;
    const bool solved = n_steps > 0 && log[n_steps - 1] == gidx;  // the last logged selection is the goal unless the budget ran out
    if (lane == 0) {
        if (extra > 0) rec[gidx].G = 0.f;
        const float hh = d.omg * (heuristic0_fast(sidx >> LOGW, sidx & (L::W - 1), goal_r, goal_c) + rec[sidx].cost);
        const float v = bwdr_v<true>(d, 0.0f, hh, rcp_sqrtW);
        rec[sidx].g = 0.0f;     // open list = {start} (:187), g[start] = 0 (:193)
        rec[sidx].stamp = 1u;   // history entry 1 is written in step 0 before it is read
        sd[0] = (double)v;
        sd[1] = (double)(rec[sidx].G * v);
    }
    wave_sync();

    int dr, dc;
    neighbour_delta(lane & 7, dr, dc);
    const bool is_nb = lane < 8;
    const int v_dr = is_nb ? dr : 0, v_dc = is_nb ? dc : 0, v_off = is_nb ? dr * L::W + dc : 0;
    const float v_inf = NASTAR_POS_INF, v_minf = NASTAR_NEG_INF;
    const uint32_t v_l4 = (uint32_t)lane * 4u, v_sda = (uint32_t)(L::HIST + (max_steps - 1) * 16);
    const uint32_t v_logmax = (uint32_t)(a.max_iters - 1) * 4u;
    const unsigned long long m_nb = 0xFFull;
    int t = 0;
    const int nloop = __builtin_amdgcn_readfirstlane(solved ? n_steps - 1 : n_steps);  // expansion steps (the goal step is below)
    const int s_gr = __builtin_amdgcn_readfirstlane(goal_r), s_gc = __builtin_amdgcn_readfirstlane(goal_c);
    double A, B, S, D, pA, pB, pA0, pB0;
    float pkv, pG;
    uint32_t paddr;
    unsigned long long pm;
    const unsigned long long logp = reinterpret_cast<unsigned long long>(log), gbase = reinterpret_cast<unsigned long long>(gout);
    asm volatile(NASTAR_BWD_ASM_LOOP
                 : [t] "+s"(t), [oA] "=v"(A), [oB] "=v"(B), [oS] "=v"(S), [oD] "=v"(D), [opA] "=v"(pA), [opB] "=v"(pB),
                   [opA0] "=v"(pA0), [opB0] "=v"(pB0), [opkv] "=v"(pkv), [opG] "=v"(pG), [opaddr] "=v"(paddr), [opm] "=s"(pm)
                 : [nloop] "s"(nloop), [sda] "v"(v_sda), [dr] "v"(v_dr), [dc] "v"(v_dc), [off] "v"(v_off), [vinf] "v"(v_inf),
                   [vminf] "v"(v_minf), [l4] "v"(v_l4), [logmax] "v"(v_logmax), [gr] "s"(s_gr), [gc] "s"(s_gc), [cgr] "s"(d.gr),
                   [comg] "s"(d.omg), [csq] "s"(d.sqrtW), [crcp] "s"(rcp_sqrtW), [ckfac] "s"(a.kfac), [mnb] "s"(m_nb),
                   [logp] "s"(logp), [gbase] "s"(gbase), [LOGW] "i"(LOGW), [WM1] "i"(L::W - 1), [W] "i"(L::W),
                   [HIST] "i"(L::HIST), [HISTM16] "i"(L::HIST - 16)
                 : "memory", "vcc", "scc", "v23", "v24", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36",
                   "v37", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v56",
                   "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72",
                   "v73", "v74", "v75", "v76", "v77", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "s42", "s43", "s44",
                   "s46", "s47", "s54", "s55", "s56", "s57", "s58", "s59", "s61", "s62", "s63");
    // intervals closed by the last expansion step
    if ((pm >> lane) & 1ull) {
        const float dA = (float)(pA - pA0), dB = (float)(pB - pB0);
        unsafeAtomicAdd(reinterpret_cast<float*>(reinterpret_cast<char*>(gout) + paddr), pkv * (pG * dA - dB));
    }
    double A_last = A, B_last = B;  // (A, B) after the last executed step = what history entry n_steps would hold
    if (solved) {
        // the goal's own step: its softmax counts, the goal is not expanded unless the batch keeps stepping (extra > 0)
        float rS = __builtin_amdgcn_rcpf((float)S);
        A += (double)rS;
        B += (double)((float)D * rS * rS);
        A_last = A;
        B_last = B;
        if (extra > 0) {
            // expansion of the goal with the goal left open (:224), then `extra` identical fixed-point steps
            const int s = gidx;
            const int r = s >> LOGW, c = s & (L::W - 1);
            const int nr = r + dr, nc = c + dc;
            const bool inb = is_nb & ((unsigned)nr < (unsigned)L::W) & ((unsigned)nc < (unsigned)L::W);
            const int il = inb ? s + dr * L::W + dc : s;
            const BwdRec me = rec[il];
            const float g2 = rec[s].g + rec[s].cost;
            const float hh = d.omg * (heuristic0_fast(il >> LOGW, il & (L::W - 1), goal_r, goal_c) + me.cost);
            const bool upd = inb & (me.g > g2);
            const bool rekey = upd & (fabsf(me.g) < NASTAR_POS_INF);
            const float v_old = bwdr_v<true>(d, me.g, hh, rcp_sqrtW), v_new = bwdr_v<true>(d, g2, hh, rcp_sqrtW);
            wave_sync();
            if (upd) {
                const double dS = (double)v_new - (rekey ? (double)v_old : 0.0);
                const double dD = (double)(me.G * v_new) - (rekey ? (double)(me.G * v_old) : 0.0);
                asm volatile("ds_add_f64 %0, %1\n\tds_add_f64 %0, %2 offset:8" ::"v"(v_sda), "v"(dS), "v"(dD) : "memory");
                if (rekey) {
                    const double A0 = me.stamp >= (uint32_t)n_steps ? A_last : hist[2 * (me.stamp - 1)];
                    const double B0 = me.stamp >= (uint32_t)n_steps ? B_last : hist[2 * (me.stamp - 1) + 1];
                    unsafeAtomicAdd(&gout[il], (a.kfac * v_old) * (me.G * (float)(A - A0) - (float)(B - B0)));
                }
                rec[il].g = g2;
                rec[il].stamp = (uint32_t)n_steps;  // opened by the last executed step: stamp = (A_last, B_last)
            }
            wave_sync();
            const double S2 = sd[0], D2 = sd[1];
            rS = __builtin_amdgcn_rcpf((float)S2);
            A += (double)extra * (double)rS;
            B += (double)extra * (double)((float)D2 * rS * rS);
        }
    }
    wave_sync();
    // cells still on the open list: close their intervals at the final (A, B)
    for (int i = lane; i < L::HW; i += 64) {
        const BwdRec me = rec[i];
        if (fabsf(me.g) < NASTAR_POS_INF) {
            const bool last = me.stamp >= (uint32_t)n_steps;
            const double A0 = last ? A_last : hist[2 * (me.stamp - 1)];
            const double B0 = last ? B_last : hist[2 * (me.stamp - 1) + 1];
            const float hh = d.omg * (heuristic0_fast(i >> LOGW, i & (L::W - 1), goal_r, goal_c) + me.cost);
            const float v = bwdr_v<true>(d, me.g, hh, rcp_sqrtW);
            unsafeAtomicAdd(&gout[i], (a.kfac * v) * (me.G * (float)(A - A0) - (float)(B - B0)));
        }
    }
}

}  // namespace nastar
