// nastar_search_unit.hip.h -- load / start / epilogue of the UNIT-COST LDS layout (NASTAR_FLAG_UNIT_COST; the step loop itself is
// search_loop_asm4<.., kUnit = true> in nastar_search_asm4.hip.h).
//
// VanillaAstar.forward passes ONE tensor as cost map and as obstacle map (reference astar.py:93-94): on the binary maps of the
// reference's datasets every passable cell then costs exactly 1.0 and every obstacle 0.0, and since only passable cells are ever
// expanded or relaxed (:228-229), the search never needs a per-cell cost: h = fl(h0 + 1.0), g2 = fl(g[s*] + 1.0).  The one exception
// is a START placed on an obstacle (the reference expands it all the same, :187): its cost is 0, i.e. g2 of the first step is
// fl(0 + 0) = 0 -- reproduced by opening such a start with g = -1.0 (fl(-1.0 + 1.0) = 0; its own key is irrelevant, it is the only
// open cell when it is selected, and it is closed before anything reads its g again).
//
//   g[]     4 B per cell: +inf passable & never opened, -inf closed or obstacle, finite = open
//   cmin[]  8 B per 16-cell chunk, as in the compact layout (idle entries: key all ones, cell = goal)
//   pdir[]  1 B per cell: parent direction, passable bit, on-path bit
// 32x32: 5,632 B per map -> 29 maps per CU (compact layout: 9,984 B, 16 per CU); 64x64: 22,528 B -> 7 per CU (4).
//
// The promise "cost == passable, every value exactly 0.0 or 1.0" is CHECKED while the map is loaded: a map that breaks it gets
// status NASTAR_ERR_NOT_UNIT_COST and empty outputs, never a wrong search (the Python shim then re-runs the batch on the general kernel).
#pragma once
#include "nastar_search_asm4.hip.h"

namespace nastar {

struct UnitLds {
    float* g;
    unsigned long long* cmin;
    uint8_t* pdir;
};

template <int LOGW>
__device__ __forceinline__ UnitLds carve_unit_lds(unsigned char* smem)
{
    using L = AsmLayoutUnit<LOGW>;
    UnitLds l;
    l.g = reinterpret_cast<float*>(smem);
    l.cmin = reinterpret_cast<unsigned long long*>(smem + L::CMIN);
    l.pdir = smem + L::PDIR;
    return l;
}

// One map, HW = 1 << (2 LOGW) cells, 16-byte loads, all of a group's loads in flight before the first is consumed.
// `bad` (wave-uniform): some value of the map is neither 0.0 nor 1.0.
template <int LOGW>
__device__ __forceinline__ void unit_load_map(const UnitLds& l, const float* __restrict__ map, const float* __restrict__ start,
                                              const float* __restrict__ goal, int lane, int& start_idx, int& goal_idx, bool& bad)
{
    constexpr int HW = 1 << (2 * LOGW);
    constexpr int ITER = HW / 256;  // float4 per lane
    constexpr int G = ITER < 4 ? ITER : 4;
    static_assert(ITER >= 1 && ITER % G == 0, "whole groups of up to four 16-byte loads per lane");
    const float4* s4 = reinterpret_cast<const float4*>(start);
    const float4* g4 = reinterpret_cast<const float4*>(goal);
    const float4* m4 = reinterpret_cast<const float4*>(map);
    int sidx = -1, gidx = -1;
    bool b = false;
    for (int base = 0; base < ITER; base += G) {
        float4 sv[G], gv[G], mv[G];
#pragma unroll
        for (int k = 0; k < G; ++k) {
            const int q = lane + (base + k) * 64;
            sv[k] = s4[q];
            gv[k] = g4[q];
            mv[k] = m4[q];
        }
#pragma unroll
        for (int k = 0; k < G; ++k) {
            const int q = lane + (base + k) * 64;
            const int i = q << 2;
            if (sv[k].x != 0.f) sidx = i;
            if (sv[k].y != 0.f) sidx = i + 1;
            if (sv[k].z != 0.f) sidx = i + 2;
            if (sv[k].w != 0.f) sidx = i + 3;
            if (gv[k].x != 0.f) gidx = i;
            if (gv[k].y != 0.f) gidx = i + 1;
            if (gv[k].z != 0.f) gidx = i + 2;
            if (gv[k].w != 0.f) gidx = i + 3;
            const float4 v = mv[k];
            b |= ((v.x != 0.f) & (v.x != 1.f)) | ((v.y != 0.f) & (v.y != 1.f)) | ((v.z != 0.f) & (v.z != 1.f)) | ((v.w != 0.f) & (v.w != 1.f));
            float4 gg;
            gg.x = v.x != 0.f ? NASTAR_POS_INF : NASTAR_NEG_INF;
            gg.y = v.y != 0.f ? NASTAR_POS_INF : NASTAR_NEG_INF;
            gg.z = v.z != 0.f ? NASTAR_POS_INF : NASTAR_NEG_INF;
            gg.w = v.w != 0.f ? NASTAR_POS_INF : NASTAR_NEG_INF;
            *reinterpret_cast<float4*>(l.g + i) = gg;
            const uint32_t m = (PARENT_UNSET | (v.x != 0.f ? P_PASS : 0u)) | ((PARENT_UNSET | (v.y != 0.f ? P_PASS : 0u)) << 8) |
                               ((PARENT_UNSET | (v.z != 0.f ? P_PASS : 0u)) << 16) | ((PARENT_UNSET | (v.w != 0.f ? P_PASS : 0u)) << 24);
            *reinterpret_cast<uint32_t*>(l.pdir + i) = m;
        }
    }
    start_idx = wave_max_i32(sidx);
    goal_idx = wave_max_i32(gidx);
    bad = __ballot(b) != 0ull;
    // idle chunk entries: key all ones, cell = the goal (nastar_search_asm4.hip.h: an empty open list leaves through the goal exit)
    const unsigned long long idle = cmin_entry(0xFFFFFFFFu, (uint32_t)(goal_idx < 0 ? 0 : goal_idx));
    constexpr int NCP = AsmLayoutUnit<LOGW>::CPL * 64;
    for (int c = lane; c < NCP; c += 64) l.cmin[c] = idle;
    wave_sync();
}

// open list = {start} (:187), g[start] = 0 (:193); kHalf: key of q' = fl(fl(g + h) / sqrt(W)) (g_ratio == 0.5, see asm4)
template <int LOGW, bool kHalf>
__device__ __forceinline__ void unit_open_start(const CompactDims& d, const UnitLds& l, int lane, int sidx, int goal_r, int goal_c,
                                                float rcp_sqrtW)
{
    if (lane == 0) {
        const int r = sidx >> LOGW, c = sidx & ((1 << LOGW) - 1);
        const bool pass = (l.pdir[sidx] & P_PASS) != 0;
        const float cost_s = pass ? 1.0f : 0.0f;
        const float omg = kHalf ? 1.0f : d.omg;
        const float hh = omg * (heuristic0_fast(r, c, goal_r, goal_c) + cost_s);  // :191-192 ; :206
        CompactDims dd = d;
        if (kHalf) dd.gr = 1.0f;
        const uint32_t k0 = __float_as_uint(ord_to_f32(compact_key<true>(dd, 0.0f, hh, rcp_sqrtW)));  // raw bits of q >= +0
        l.g[sidx] = pass ? 0.0f : -1.0f;  // an obstacle start costs 0: fl(-1 + 1) = fl(0 + 0) is what its first step hands on
        l.cmin[sidx >> CCL] = cmin_entry(k0, (uint32_t)sidx);
        l.pdir[sidx] = (uint8_t)(PARENT_UNSET | P_PASS);  // the start is expanded even if it sits on an obstacle (:187)
    }
    wave_sync();
}

// AstarOutput.histories (fp32 0/1): closed = passable & g == -inf; written BEFORE the serial backtrack (the stores drain under it)
template <int LOGW>
__device__ __forceinline__ void unit_store_hist(const UnitLds& l, int lane, float* __restrict__ hist, bool zero)
{
    constexpr int N4 = (1 << (2 * LOGW)) >> 2;
    float4* h4 = reinterpret_cast<float4*>(hist);
    for (int q = lane; q < N4; q += 64) {
        const uint32_t m = *reinterpret_cast<const uint32_t*>(l.pdir + (q << 2));
        const float4 gg = *reinterpret_cast<const float4*>(l.g + (q << 2));
        float4 v;
        v.x = (!zero && (m & P_PASS) && gg.x == NASTAR_NEG_INF) ? 1.0f : 0.0f;
        v.y = (!zero && (m & (P_PASS << 8)) && gg.y == NASTAR_NEG_INF) ? 1.0f : 0.0f;
        v.z = (!zero && (m & (P_PASS << 16)) && gg.z == NASTAR_NEG_INF) ? 1.0f : 0.0f;
        v.w = (!zero && (m & (P_PASS << 24)) && gg.w == NASTAR_NEG_INF) ? 1.0f : 0.0f;
        h4[q] = v;
    }
}

// AstarOutput.paths (int64 0/1) and optionally the 2-bit-per-cell packed masks (layout of compact_store_outputs)
template <int LOGW>
__device__ __forceinline__ void unit_store_paths(const UnitLds& l, int lane, long long* __restrict__ paths, uint8_t* __restrict__ packed,
                                                 bool zero)
{
    constexpr int HW = 1 << (2 * LOGW);
    if (packed != nullptr) {  // wave-uniform
        for (int q = lane; q < (HW >> 2); q += 64) {
            const uint32_t m = *reinterpret_cast<const uint32_t*>(l.pdir + (q << 2));
            const float4 gg = *reinterpret_cast<const float4*>(l.g + (q << 2));
            const bool c0 = !zero && (m & P_PASS) && gg.x == NASTAR_NEG_INF;
            const bool c1 = !zero && (m & (P_PASS << 8)) && gg.y == NASTAR_NEG_INF;
            const bool c2 = !zero && (m & (P_PASS << 16)) && gg.z == NASTAR_NEG_INF;
            const bool c3 = !zero && (m & (P_PASS << 24)) && gg.w == NASTAR_NEG_INF;
            const uint32_t nh = (c0 ? 8u : 0u) | (c1 ? 4u : 0u) | (c2 ? 2u : 0u) | (c3 ? 1u : 0u);
            const uint32_t np = zero ? 0u : (((m & P_PATH) ? 8u : 0u) | ((m & (P_PATH << 8)) ? 4u : 0u) | ((m & (P_PATH << 16)) ? 2u : 0u) |
                                             ((m & (P_PATH << 24)) ? 1u : 0u));
            const uint32_t both = nh | (np << 8);
            const uint32_t other = dpp_mov<DPP_QUAD_XOR1>(both);  // the odd lane's quad = low nibble of the byte
            if ((lane & 1) == 0) {
                packed[q >> 1] = (uint8_t)((nh << 4) | (other & 0xFu));
                packed[(HW >> 3) + (q >> 1)] = (uint8_t)((np << 4) | ((other >> 8) & 0xFu));
            }
        }
    }
    longlong2* p2 = reinterpret_cast<longlong2*>(paths);
    for (int q = lane; q < (HW >> 1); q += 64) {
        const uint32_t m = *reinterpret_cast<const uint16_t*>(l.pdir + (q << 1));
        longlong2 v;
        v.x = (!zero && (m & P_PATH)) ? 1 : 0;
        v.y = (!zero && (m & (P_PATH << 8))) ? 1 : 0;
        p2[q] = v;
    }
}

}  // namespace nastar
