// nastar_search_duo.hip.h -- TWO maps per 64-lane wavefront: each 32-lane half owns one map (compact 9 B/cell state,
// nastar_search_compact.hip.h).
//
// Why: with 16 single-map wavefronts per CU (4 per SIMD) the compact kernel is VALU-issue bound -- SQ counters of the
// 4096-map maze batch (profiles/r02/sq_compact_single.json): 80 VALU instructions per selection step, ~3.7 SIMD cycles each,
// every step using 8 (neighbours) + 1 (close) + 16 (chunk re-minimisation) of 64 lanes.  A step's instruction stream is the
// same for every map, so two maps share it: lanes 0-7 / 32-39 relax the Moore neighbours, lane 8 / 40 closes s*, lanes
// 16-31 / 48-63 re-minimise the chunk of s*, and the chunk-minimum scan gives each lane 2 entries (one ds_read_b128) with a
// 5-stage reduction (4 DPP row steps + v_permlane16_swap) per half.  Half the VALU issue per map-step, 8 wavefronts per CU
// for the same 16 resident 32x32 maps.  A half whose search has ended keeps executing the other half's steps with every
// store redirected to a dump word (no divergence, no exec-mask regions in the step).
#pragma once
#include "nastar_search_compact.hip.h"

namespace nastar {

// per-half LDS region: gc[HWp] | cmin[NCp] | pdir[HWp]   (NCp = CPL * 32 entries, CPL = chunk minima per lane)
__host__ __device__ inline size_t duo_region_bytes(int HWp, int NCp) { return (size_t)HWp * 9 + (size_t)NCp * 8; }
__host__ __device__ inline size_t duo_lds_bytes(int HWp, int NCp) { return 2 * duo_region_bytes(HWp, NCp) + 256; }

__device__ __forceinline__ CompactLds carve_duo_lds(unsigned char* smem, const CompactDims& d, int half)
{
    unsigned char* base = smem + (size_t)half * duo_region_bytes(d.HWp, d.NCp);
    CompactLds l;
    l.gc = reinterpret_cast<float2*>(base);
    l.cmin = reinterpret_cast<unsigned long long*>(l.gc + d.HWp);
    l.pdir = reinterpret_cast<uint8_t*>(l.cmin + d.NCp);
    l.dump = reinterpret_cast<uint32_t*>(smem + 2 * duo_region_bytes(d.HWp, d.NCp));
    return l;
}

// min / max over each 32-lane half, result in every lane of the half
__device__ __forceinline__ uint32_t half_min_all_u32(uint32_t v)
{
    v = row_min16_u32(v);
    auto a = __builtin_amdgcn_permlane16_swap(v, v, false, false);  // rows 0<->1, 2<->3
    return min(a[0], a[1]);
}
__device__ __forceinline__ int half_max_all_i32(int v)
{
    v = max(v, (int)dpp_mov<DPP_QUAD_XOR1>((uint32_t)v));
    v = max(v, (int)dpp_mov<DPP_QUAD_XOR2>((uint32_t)v));
    v = max(v, (int)dpp_mov<DPP_ROW_HALF_MIRROR>((uint32_t)v));
    v = max(v, (int)dpp_mov<DPP_ROW_MIRROR>((uint32_t)v));
    auto a = __builtin_amdgcn_permlane16_swap((uint32_t)v, (uint32_t)v, false, false);
    return max((int)a[0], (int)a[1]);
}

template <bool kVec4>
__device__ __forceinline__ void duo_load_map(const CompactDims& d, const CompactLds& l, const float* __restrict__ cost,
                                             const float* __restrict__ start, const float* __restrict__ goal,
                                             const float* __restrict__ passable, int hl, int& start_idx, int& goal_idx)
{
    int sidx = -1, gidx = -1;
    if constexpr (kVec4) {
        const float4* s4 = reinterpret_cast<const float4*>(start);
        const float4* g4 = reinterpret_cast<const float4*>(goal);
        const float4* c4 = reinterpret_cast<const float4*>(cost);
        const float4* p4 = reinterpret_cast<const float4*>(passable);
        const int n4 = d.HW >> 2;
        for (int q = hl; q < n4; q += 32) {
            const float4 sv = s4[q];
            const float4 gv = g4[q];
            const float4 cv = c4[q];
            const float4 pv = p4[q];
            const int i = q << 2;
            if (sv.x != 0.f) sidx = i;
            if (sv.y != 0.f) sidx = i + 1;
            if (sv.z != 0.f) sidx = i + 2;
            if (sv.w != 0.f) sidx = i + 3;
            if (gv.x != 0.f) gidx = i;
            if (gv.y != 0.f) gidx = i + 1;
            if (gv.z != 0.f) gidx = i + 2;
            if (gv.w != 0.f) gidx = i + 3;
            float4 lo, hi;
            lo.x = pv.x != 0.f ? NASTAR_POS_INF : NASTAR_NEG_INF;
            lo.y = cv.x;
            lo.z = pv.y != 0.f ? NASTAR_POS_INF : NASTAR_NEG_INF;
            lo.w = cv.y;
            hi.x = pv.z != 0.f ? NASTAR_POS_INF : NASTAR_NEG_INF;
            hi.y = cv.z;
            hi.z = pv.w != 0.f ? NASTAR_POS_INF : NASTAR_NEG_INF;
            hi.w = cv.w;
            *reinterpret_cast<float4*>(l.gc + i) = lo;
            *reinterpret_cast<float4*>(l.gc + i + 2) = hi;
            const uint32_t m = (PARENT_UNSET | (pv.x != 0.f ? P_PASS : 0u)) | ((PARENT_UNSET | (pv.y != 0.f ? P_PASS : 0u)) << 8) |
                               ((PARENT_UNSET | (pv.z != 0.f ? P_PASS : 0u)) << 16) |
                               ((PARENT_UNSET | (pv.w != 0.f ? P_PASS : 0u)) << 24);
            *reinterpret_cast<uint32_t*>(l.pdir + i) = m;
        }
    } else {
        for (int i = hl; i < d.HW; i += 32) {
            if (start[i] != 0.f) sidx = i;
            if (goal[i] != 0.f) gidx = i;
            const float pv = passable[i];
            l.gc[i] = make_float2(pv != 0.f ? NASTAR_POS_INF : NASTAR_NEG_INF, cost[i]);
            l.pdir[i] = (uint8_t)(PARENT_UNSET | (pv != 0.f ? P_PASS : 0u));
        }
    }
    for (int i = d.HW + hl; i < d.HWp; i += 32) l.gc[i] = make_float2(NASTAR_NEG_INF, 0.f);  // tail of the last chunk: never open
    for (int c = hl; c < d.NCp; c += 32) l.cmin[c] = ~0ull;
    start_idx = half_max_all_i32(sidx);
    goal_idx = half_max_all_i32(gidx);
    wave_sync();
}

// ---- selection per half: first flat index of the minimal key (s), `empty` when the half's open list is empty -----------
// Lane hl owns the CONTIGUOUS entries [hl*CPL, (hl+1)*CPL).  e0/e1: the lane's own entries as read (CPL_T <= 2).
template <int CPL_T>
__device__ __forceinline__ int duo_select(const CompactDims& d, const CompactLds& l, int half, int hl, bool& empty, uint2& e0, uint2& e1)
{
    uint2 best;  // .x = cell index, .y = key
    if constexpr (CPL_T == 1) {
        const unsigned long long e = l.cmin[hl];
        e0.x = (uint32_t)e;
        e0.y = (uint32_t)(e >> 32);
        e1 = e0;
        best = e0;
    } else if constexpr (CPL_T == 2) {
        const uint4 q = *reinterpret_cast<const uint4*>(l.cmin + 2 * hl);  // one ds_read_b128
        e0 = make_uint2(q.x, q.y);
        e1 = make_uint2(q.z, q.w);
        best = (e1.y < e0.y) ? e1 : e0;  // strict: the earlier chunk wins ties
    } else {
        const int cpl = CPL_T > 0 ? CPL_T : d.CPL;
        const uint2* p = reinterpret_cast<const uint2*>(l.cmin) + hl * cpl;
        best = p[0];
        if constexpr (CPL_T > 0) {
#pragma unroll
            for (int c = 1; c < CPL_T; ++c) {
                const uint2 e = p[c];
                if (e.y < best.y) best = e;
            }
        } else {
            for (int c = 1; c < cpl; ++c) {
                const uint2 e = p[c];
                if (e.y < best.y) best = e;
            }
        }
        e0 = e1 = best;
    }
    const uint32_t M = half_min_all_u32(best.y);
    const unsigned long long hit = __ballot(best.y == M);  // both halves are never empty: M is one of the half's keys
    const int L0 = __builtin_ctz((uint32_t)hit);
    const int L1 = 32 + __builtin_ctz((uint32_t)(hit >> 32));
    const int s0 = __builtin_amdgcn_readlane((int)best.x, L0);
    const int s1 = __builtin_amdgcn_readlane((int)best.x, L1);
    empty = M == KEY_INF;
    return half ? s1 : s0;
}

struct DuoLane {
    int dr, dc, off;
    bool is_nb;   // hl 0..7: Moore neighbour j of s*
    bool is_chk;  // hl 16..31: cell (hl - 16) of the chunk that holds s*
    bool closer;  // hl 8
    uint32_t pcode;
};

__device__ __forceinline__ DuoLane make_duo_lane(const CompactDims& d, int hl)
{
    DuoLane lc;
    neighbour_delta(hl & 7, lc.dr, lc.dc);
    lc.is_nb = hl < 8;
    lc.is_chk = (hl & 16) != 0;
    lc.closer = hl == 8;
    lc.off = lc.dr * d.W + lc.dc;
    lc.pcode = P_PASS | (uint32_t)(hl & 7);
    return lc;
}

// ---- close s (:222-225), relax its <= 8 Moore neighbours (:228-249), re-minimise the chunk of s without it -- per half ----
// act: this half executes a real step (otherwise all its stores go to the dump words / no-op atomics).
template <int LOGW, bool kFastDiv, int CPL_T>
__device__ __forceinline__ void duo_expand(const CompactDims& d, const CompactLds& l, const DuoLane& lc, int lane, int half, int hl,
                                           int s, bool act, int goal_r, int goal_c, float rcp_sqrtW, const uint2 e0, const uint2 e1)
{
    int r, c;
    if constexpr (LOGW) {
        r = s >> LOGW;
        c = s & ((1 << LOGW) - 1);
    } else {
        r = (int)div_magic((uint32_t)s, d.magicW);
        c = s - r * d.W;
    }
    const int cbase = s & ~(CCSZ - 1);
    const int nr = r + lc.dr, nc = c + lc.dc;
    const bool inb = lc.is_nb & ((unsigned)nr < (unsigned)d.H) & ((unsigned)nc < (unsigned)d.W);  // conv2d zero padding
    const int il = inb ? s + lc.off : (lc.is_chk ? cbase + (hl & (CCSZ - 1)) : s);
    const float2 gs = l.gc[s];
    const float2 gl = l.gc[il];
    int rl, cl;
    if constexpr (LOGW) {
        rl = il >> LOGW;
        cl = il & ((1 << LOGW) - 1);
    } else {
        rl = (int)div_magic((uint32_t)il, d.magicW);
        cl = il - rl * d.W;
    }
    const float h0 = heuristic0_fast(rl, cl, goal_r, goal_c);
    const float hh = d.omg * (h0 + gl.y);  // :191-192 h = h0 + cost ; :206 (1-g_ratio)*h
    const float g2 = gs.x + gs.y;          // :234 step cost of the node being LEFT
    const bool upd = act & inb & (gl.x > g2);  // :229,:235
    const uint32_t k = compact_key<kFastDiv>(d, lc.is_nb ? g2 : gl.x, hh, rcp_sqrtW);
    const bool open_l = lc.is_chk & (fabsf(gl.x) < NASTAR_POS_INF) & (il != s);
    const uint32_t kk = open_l ? k : KEY_INF;
    const uint32_t mc = row_min16_u32(kk);
    const unsigned long long fm = __ballot(lc.is_chk & (kk == mc));  // bits 16..31 and 48..63, never empty
    const uint32_t Mc0 = (uint32_t)__builtin_amdgcn_readlane((int)mc, 16);
    const uint32_t Mc1 = (uint32_t)__builtin_amdgcn_readlane((int)mc, 48);
    const int c0 = __builtin_ctz((uint32_t)fm >> 16);
    const int c1 = __builtin_ctz((uint32_t)(fm >> 48));
    const uint32_t Mc = half ? Mc1 : Mc0;
    const uint32_t ci = (uint32_t)(cbase + (half ? c1 : c0));
    uint32_t* const dmp = l.dump + lane;
    float* const g_dst = upd ? &l.gc[il].x : ((lc.closer & act) ? &l.gc[s].x : reinterpret_cast<float*>(dmp));
    uint8_t* const p_dst = upd ? &l.pdir[il] : reinterpret_cast<uint8_t*>(dmp);
    *g_dst = upd ? g2 : NASTAR_NEG_INF;  // :238 g update          | :222-225 s* joins the closed list, leaves the open list
    *p_dst = (uint8_t)lc.pcode;          // :246-249 parent = s*
    // exact minimum of the chunk without s* (lands before the atomics: LDS executes a wave's operations in order)
    const int C = s >> CCL;
    if constexpr (CPL_T == 1) {
        const bool own = act & (C == hl);
        uint2 e;
        e.x = own ? ci : e0.x;
        e.y = own ? Mc : e0.y;
        reinterpret_cast<uint2*>(l.cmin)[hl] = e;
    } else if constexpr (CPL_T == 2) {
        // every lane rewrites its own two entries -- unchanged, except the owner of chunk C: one unmasked ds_write_b128
        const bool own0 = act & (C == 2 * hl);
        const bool own1 = act & (C == 2 * hl + 1);
        uint4 q;
        q.x = own0 ? ci : e0.x;
        q.y = own0 ? Mc : e0.y;
        q.z = own1 ? ci : e1.x;
        q.w = own1 ? Mc : e1.y;
        *reinterpret_cast<uint4*>(l.cmin + 2 * hl) = q;
    } else {
        if (act & (hl == 16)) l.cmin[C] = cmin_entry(Mc, ci);
    }
    // :242 (re)opened neighbours enter their chunk's minimum; other lanes issue min(x, ~0) on an entry of their own: a no-op
    const int cpl = CPL_T > 0 ? CPL_T : d.CPL;
    atomicMin(upd ? &l.cmin[il >> CCL] : &l.cmin[hl * cpl], upd ? cmin_entry(k, (uint32_t)il) : ~0ull);
    wave_order();
}

// backtrack (differentiable_astar.py:96-125) of both halves at once: lane 0 / 32 each walk their own map
__device__ __forceinline__ void duo_backtrack(const CompactDims& d, const CompactLds& l, int hl, int start_idx, int goal_idx, int cap)
{
    if (hl == 0 && goal_idx >= 0) {
        uint32_t m = l.pdir[goal_idx];
        l.pdir[goal_idx] = (uint8_t)(m | P_PATH);
        uint32_t code = m & P_DIRMASK;
        if (code != PARENT_UNSET) {
            int loc = compact_parent_of(d, goal_idx, code);
            for (int k = 0; k < cap; ++k) {
                uint32_t ml = l.pdir[loc];
                l.pdir[loc] = (uint8_t)(ml | P_PATH);
                if (loc == start_idx) break;
                uint32_t cd = ml & P_DIRMASK;
                if (cd == PARENT_UNSET) break;
                loc = compact_parent_of(d, loc, cd);
            }
        }
    }
    wave_sync();
}

template <bool kVec4>
__device__ __forceinline__ void duo_store_outputs(const CompactDims& d, const CompactLds& l, int hl, float* __restrict__ hist,
                                                  long long* __restrict__ paths, uint8_t* __restrict__ packed)
{
    if constexpr (kVec4) {
        const int n4 = d.HW >> 2;
        float4* h4 = reinterpret_cast<float4*>(hist);
        for (int q = hl; q < n4; q += 32) {
            const uint32_t m = *reinterpret_cast<const uint32_t*>(l.pdir + (q << 2));
            const float4 lo = *reinterpret_cast<const float4*>(l.gc + (q << 2));
            const float4 hi = *reinterpret_cast<const float4*>(l.gc + (q << 2) + 2);
            const bool c0 = (m & P_PASS) && lo.x == NASTAR_NEG_INF;
            const bool c1 = (m & (P_PASS << 8)) && lo.z == NASTAR_NEG_INF;
            const bool c2 = (m & (P_PASS << 16)) && hi.x == NASTAR_NEG_INF;
            const bool c3 = (m & (P_PASS << 24)) && hi.z == NASTAR_NEG_INF;
            float4 v;
            v.x = c0 ? 1.0f : 0.0f;
            v.y = c1 ? 1.0f : 0.0f;
            v.z = c2 ? 1.0f : 0.0f;
            v.w = c3 ? 1.0f : 0.0f;
            h4[q] = v;
            if (packed != nullptr) {
                const uint32_t nh = (c0 ? 8u : 0u) | (c1 ? 4u : 0u) | (c2 ? 2u : 0u) | (c3 ? 1u : 0u);
                const uint32_t np = ((m & P_PATH) ? 8u : 0u) | ((m & (P_PATH << 8)) ? 4u : 0u) |
                                    ((m & (P_PATH << 16)) ? 2u : 0u) | ((m & (P_PATH << 24)) ? 1u : 0u);
                const uint32_t both = nh | (np << 8);
                const uint32_t other = dpp_mov<DPP_QUAD_XOR1>(both);  // the odd lane's quad = low nibble of the byte
                if ((hl & 1) == 0) {
                    const int nb = d.HW >> 3;
                    packed[q >> 1] = (uint8_t)((nh << 4) | (other & 0xFu));
                    packed[nb + (q >> 1)] = (uint8_t)((np << 4) | ((other >> 8) & 0xFu));
                }
            }
        }
        const int n2 = d.HW >> 1;
        longlong2* p2 = reinterpret_cast<longlong2*>(paths);
        for (int q = hl; q < n2; q += 32) {
            const uint32_t m = *reinterpret_cast<const uint16_t*>(l.pdir + (q << 1));
            longlong2 v;
            v.x = (m & P_PATH) ? 1 : 0;
            v.y = (m & (P_PATH << 8)) ? 1 : 0;
            p2[q] = v;
        }
    } else {
        for (int i = hl; i < d.HW; i += 32) {
            const uint32_t m = l.pdir[i];
            hist[i] = ((m & P_PASS) && l.gc[i].x == NASTAR_NEG_INF) ? 1.0f : 0.0f;
            paths[i] = (m & P_PATH) ? 1 : 0;
        }
    }
}

}  // namespace nastar
