// nastar_search_reg.hip.h -- register-resident A* search for maps of up to 1024 cells (the 32x32 headline case).
//
// One 64-lane wavefront owns one map and keeps the WHOLE search state in VGPRs: cell i lives in lane (i & 63),
// slot (i >> 6); the per-cell arrays key/g/hh/cost are 16-element register vectors indexed by the slot.  Every
// access in the search loop uses a wave-UNIFORM slot, which gfx950 serves with VGPR index mode
// (s_set_gpr_idx_on + v_mov), so an iteration contains no LDS or memory round trip at all:
//
//   cm          lane l < 16 holds the minimum key of slot l (one DPP row)           -> 4 DPP steps give the global min
//   C, cl       first slot holding the min (ballot over cm), first lane inside it (ballot over key[C])
//   expansion   the 3x3 neighbourhood of s* spans <= 2..3 slots; for each, the OWNER lanes test/relax their own cell
//               ( g[n] > g2 with g = +inf unopened / -inf closed-or-obstacle, see nastar_search.hip.h ) and the
//               slot minimum is recomputed with one DPP wave reduction.
//
// A lone wavefront issues roughly one instruction every ~4 cycles, so the latency of one search step is set by its
// instruction count (measured: the LDS-resident kernel spent 1124 cycles/step, half of it in 4-5 LDS waits).  With
// the state in registers the occupancy limit is VGPRs, not LDS: every map of a 4096-map batch is resident at once
// (16 waves per CU), which removes the second dispatch round and leaves the longest search as the only tail.
//
// Reference semantics (differentiable_astar.py:203-252) are identical to nastar_search.hip.h; only the storage differs.
#pragma once
#include "nastar_search.hip.h"

namespace nastar {

typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int REG_SLOTS = 16;
constexpr int REG_MAX_CELLS = REG_SLOTS * 64;

__device__ __forceinline__ uint32_t row_min_u32(uint32_t v) { return row_min16_u32(v); }

struct RegState {
    u32x16 key;
    f32x16 g, hh, cost;
    uint32_t cm;        // lane l < 16: min key of slot l ; other lanes KEY_INF
    uint32_t passbits;  // bit k: cell k*64+lane counts as passable for the histories output
};

// ---- load one map (coalesced dword loads: slot k <-> cells [64k, 64k+64)) -------------------------------------
__device__ __forceinline__ void reg_load_map(const MapDims& d, RegState& st, uint8_t* pdir, const float* __restrict__ cost,
                                             const float* __restrict__ start, const float* __restrict__ goal,
                                             const float* __restrict__ passable, int lane, int& start_idx, int& goal_idx)
{
    float sv[REG_SLOTS], gv[REG_SLOTS], cv[REG_SLOTS], pv[REG_SLOTS];
#pragma unroll
    for (int k = 0; k < REG_SLOTS; ++k) {
        const int i = k * 64 + lane;
        const bool valid = i < d.HW;
        sv[k] = valid ? start[i] : 0.f;
        gv[k] = valid ? goal[i] : 0.f;
        cv[k] = valid ? cost[i] : 0.f;
        pv[k] = valid ? passable[i] : 0.f;
    }
    int sidx = -1, gidx = -1;
#pragma unroll
    for (int k = 0; k < REG_SLOTS; ++k) {
        if (sv[k] != 0.f) sidx = k * 64 + lane;
        if (gv[k] != 0.f) gidx = k * 64 + lane;
    }
    sidx = wave_max_i32(sidx);
    gidx = wave_max_i32(gidx);
    start_idx = sidx;
    goal_idx = gidx;
    const int gi = gidx < 0 ? 0 : gidx;
    const int goal_r = (int)div_magic((uint32_t)gi, d.magicW);
    const int goal_c = gi - goal_r * d.W;
    uint32_t pb = 0;
#pragma unroll
    for (int k = 0; k < REG_SLOTS; ++k) {
        const int i = k * 64 + lane;
        const int r = (int)div_magic((uint32_t)i, d.magicW);
        const int c = i - r * d.W;
        const bool pass = pv[k] != 0.f;  // cells >= HW read 0 -> -inf -> never relaxed
        st.cost[k] = cv[k];
        st.hh[k] = d.omg * (heuristic0(r, c, goal_r, goal_c) + cv[k]);  // :191-192 h = h0 + cost ; :206 (1-g_ratio)*h
        st.g[k] = pass ? NASTAR_POS_INF : NASTAR_NEG_INF;
        st.key[k] = KEY_INF;
        pb |= pass ? (1u << k) : 0u;
    }
    // parent direction codes live in LDS (written fire-and-forget in the loop, read only by the backtrack)
    *reinterpret_cast<uint4*>(pdir + lane * 16) = make_uint4(0x0F0F0F0Fu, 0x0F0F0F0Fu, 0x0F0F0F0Fu, 0x0F0F0F0Fu);
    st.cm = KEY_INF;
    if (sidx >= 0) {  // open list = {start} (:187), g[start] = 0 (:193)
        const int S0 = sidx >> 6, l0 = sidx & 63;
        const uint32_t k0 = make_key(d, 0.0f, st.hh[S0]);  // each lane evaluates its own cell; lane l0's is the one used
        const uint32_t kk = st.key[S0];
        const float gg = st.g[S0];
        st.key[S0] = (lane == l0) ? k0 : kk;
        st.g[S0] = (lane == l0) ? 0.0f : gg;
        const uint32_t k0s = (uint32_t)__builtin_amdgcn_readlane((int)k0, l0);
        st.cm = (lane == S0) ? k0s : KEY_INF;
        if (lane == l0) pb |= 1u << S0;  // the start is expanded even if it sits on an obstacle
    }
    st.passbits = pb;
}

// ---- selection (:206-209): first flat index of the minimal key; -1 if the open list is empty.  Reads only. ------
__device__ __forceinline__ int reg_select(const RegState& st, int& C, int& cl)
{
    // slots are ordered by flat index, so "first slot holding the min, then first lane inside it" is the first index
    const uint32_t rm = row_min_u32(st.cm);
    const uint32_t M = (uint32_t)__builtin_amdgcn_readlane((int)rm, 0);
    if (M == KEY_INF) return -1;
    C = __builtin_ctzll(__ballot(st.cm == M));
    const uint32_t kv = st.key[C];
    cl = __builtin_ctzll(__ballot(kv == M));
    return C * 64 + cl;
}

// the goal joins the closed list (:222-223) without being expanded (every later reference step is a fixed point)
__device__ __forceinline__ void reg_close_only(RegState& st, int lane, int C, int cl)
{
    const float gC = st.g[C];
    st.g[C] = (lane == cl) ? NASTAR_NEG_INF : gC;
}

// ---- close s = (C, cl) (:222-225) and relax its Moore neighbours (:228-249) ------------------------------------
// NSTEP = compile-time bound on the number of 64-cell slots a 3x3 neighbourhood can span for this map width
// (2 for W <= 32, 3 up to W = 64); the steps are straight-line and branch-free so that the register vectors are
// updated in place.  keep_open: s is the goal being stepped at its fixed point (backward only, :224).
// LOGW > 0: W == 1 << LOGW is a compile-time power of two <= 64 (index arithmetic becomes shifts); 0: run-time W.
template <int NSTEP, bool kFastDiv, int LOGW>
__device__ __forceinline__ void reg_expand(const MapDims& d, RegState& st, uint8_t* pdir, int lane, int s, int C, int cl,
                                           bool keep_open, float rcp_sqrtW)
{
    constexpr int CW = LOGW ? (1 << LOGW) : 0;
    // g2 = g[s*] + cost[s*]  (:234) -- every lane adds its own pair, lane cl's result is broadcast
    const float g2 = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(st.g[C] + st.cost[C]), cl));
    int r, c, S_lo, S_hi;
    if constexpr (LOGW) {
        r = s >> LOGW;
        c = s & (CW - 1);
        const int r_lo = r > 0 ? r - 1 : 0, r_hi = r < d.H - 1 ? r + 1 : r;
        S_lo = r_lo >> (6 - LOGW);  // a slot holds 64/W whole rows
        S_hi = r_hi >> (6 - LOGW);
    } else {
        r = (int)div_magic((uint32_t)s, d.magicW);
        c = s - r * d.W;
        const int r_lo = r > 0 ? r - 1 : 0, r_hi = r < d.H - 1 ? r + 1 : r;
        const int c_lo = c > 0 ? c - 1 : 0, c_hi = c < d.W - 1 ? c + 1 : c;
        S_lo = (r_lo * d.W + c_lo) >> 6;
        S_hi = (r_hi * d.W + c_hi) >> 6;
    }
    const float gg2 = d.gr * g2;                              // :206 first product
    const bool lane_is_cl = (lane == cl);
#pragma unroll
    for (int t = 0; t < NSTEP; ++t) {
        const bool act = (S_lo + t) <= S_hi;                  // wave-uniform; an inactive step rewrites slot S_hi unchanged
        const int S = act ? S_lo + t : S_hi;
        const uint32_t kS = st.key[S];
        const float gS = st.g[S];
        const float hS = st.hh[S];
        const int i = S * 64 + lane;
        int ri, ci;
        if constexpr (LOGW) {
            ri = (S << (6 - LOGW)) + (lane >> LOGW);
            ci = lane & (CW - 1);
        } else {
            ri = (int)div_magic((uint32_t)i, d.magicW);
            ci = i - ri * d.W;
        }
        const unsigned drr = (unsigned)(ri - r + 1), dcc = (unsigned)(ci - c + 1);
        const bool is_sel = lane_is_cl && (S == C);
        // :228-229,:235  Moore neighbour (ri, ci are real coordinates so nothing wraps = conv2d zero padding), passable,
        // not closed, and (unopened or open with g > g2)   <=>   g[n] > g2
        const bool upd = act && (drr < 3u) && (dcc < 3u) && !is_sel && (gS > g2);
        const float f = gg2 + hS;                             // :206
        float q;
        if constexpr (kFastDiv) {
            // correctly rounded f / sqrt(W) for f >= 2^-100 (exhaustively verified per W, tools/fastdiv_check.c):
            // q0 = RN(f*y), rem = f - q0*b exactly (FMA), q = RN(q0 + rem*y)
            const float q0 = f * rcp_sqrtW;
            const float rem = __builtin_fmaf(-q0, d.sqrtW, f);
            q = __builtin_fmaf(rem, rcp_sqrtW, q0);
        } else {
            q = f / d.sqrtW;                                  // :207 IEEE fp32 division
        }
        const uint32_t nk = f32_to_ord(q);
        const bool close_it = act && is_sel && !keep_open;
        const uint32_t k_new = upd ? nk : (close_it ? KEY_INF : kS);
        const float g_new = upd ? g2 : (close_it ? NASTAR_NEG_INF : gS);
        st.key[S] = k_new;
        st.g[S] = g_new;
        const unsigned k9 = drr * 3u + dcc;                   // raster position in the 3x3 stencil (4 = centre, unused)
        const int pa = upd ? i : REG_MAX_CELLS + lane;        // non-updating lanes write to a dump byte (no branch)
        pdir[pa] = (uint8_t)k9;                               // :246-249 parent = s*
        const uint32_t sm = wave_min_u32(k_new);              // new minimum of slot S
        st.cm = (lane == S) ? sm : st.cm;
    }
}

// parent of cell n from its raster code k9 in the 3x3 stencil (n = parent + (k9/3-1)*W + (k9%3-1))
__device__ __forceinline__ int reg_parent_of(const MapDims& d, int n, uint32_t k9)
{
    const int q = (k9 >= 6u) ? 2 : ((k9 >= 3u) ? 1 : 0);
    return n - ((q - 1) * d.W + ((int)k9 - 3 * q - 1));
}

// ---- backtrack over the LDS parent codes; returns this lane's path bits (bit k: cell k*64+lane is on the path) ----
__device__ __forceinline__ uint32_t reg_backtrack(const MapDims& d, const uint8_t* pdir, int lane, int start_idx,
                                                  int goal_idx, int cap)
{
    uint32_t pathbits = (lane == (goal_idx & 63)) ? (1u << (goal_idx >> 6)) : 0u;
    constexpr uint32_t UNSET = 0x0Fu;  // raster codes are 0..8
    uint32_t code = (uint32_t)__builtin_amdgcn_readfirstlane((int)pdir[goal_idx]) & P_DIRMASK;
    if (code != UNSET) {
        int loc = reg_parent_of(d, goal_idx, code);
        for (int k = 0; k < cap; ++k) {
            pathbits |= (lane == (loc & 63)) ? (1u << (loc >> 6)) : 0u;
            if (loc == start_idx) break;
            code = (uint32_t)__builtin_amdgcn_readfirstlane((int)pdir[loc]) & P_DIRMASK;
            if (code == UNSET) break;
            loc = reg_parent_of(d, loc, code);
        }
    }
    return pathbits;
}

__device__ __forceinline__ void reg_store_outputs(const MapDims& d, const RegState& st, uint32_t pathbits, int lane,
                                                  float* __restrict__ hist, long long* __restrict__ paths)
{
#pragma unroll
    for (int k = 0; k < REG_SLOTS; ++k) {
        const int i = k * 64 + lane;
        if (i < d.HW) {
            const bool closed = ((st.passbits >> k) & 1u) && (st.g[k] == NASTAR_NEG_INF);
            hist[i] = closed ? 1.0f : 0.0f;
            paths[i] = (pathbits >> k) & 1u;
        }
    }
}

}  // namespace nastar
