// nastar_search_compact.hip.h -- the LDS-resident A* search state machine, compact form (round 2): 9 bytes per cell.
//
// One 64-lane wavefront owns one map (as nastar_search.hip.h), but the per-map LDS state is cut from 17 to 9 B/cell so that
// SIXTEEN 32x32 maps are resident per CU (all 4096 maps of the headline batch start at t = 0: no second dispatch round
// behind 9 maps per CU), and a step needs TWO dependent LDS round trips instead of three:
//
//   gc[]    float2 per cell: .x = g-value with the node state in the sign of infinity (+inf passable & never opened,
//           -inf closed or obstacle, finite = open), .y = cost (differentiable_astar.py:191-193,:222-243)
//   pdir[]  1 byte per cell: parent direction code, passable bit, on-path bit
//   cmin[]  one 64-bit word per 16-cell chunk: (key << 32) | cell index of the chunk's first minimal OPEN cell, maintained
//           with ds_min_u64.  The priority key q = fl(f / fl32(sqrt(W))) (order-preserving u32 image, see nastar_search.hip.h)
//           and the heuristic are NOT stored: h = h0 + cost is re-derived per step for the <= 8 relaxed neighbours and for
//           the 16 cells of the selected chunk, in the shadow of the LDS round trip that fetches their (g, cost) pairs,
//           by ONE instruction stream (lanes 0-7: neighbours, lane 8: closes s*, lanes 16-31: the chunk's cells).
//
//   select  = read cmin (1 word per lane for <= 1024 cells) -> 6-stage DPP min of the keys -> ballot/ff1/readlane gives
//             the cell index directly: the (key, index) pair makes the second "which cell of the chunk" round trip of the
//             round-1 kernel unnecessary.  First chunk, then first cell inside the chunk (lexicographic u64 min) = the
//             reference's first-flat-index tie-break of torch.max (:69).
//   expand  = as before (lanes 0..7 own the Moore neighbours, :228-249), plus the exact re-minimisation of chunk C without
//             s* from recomputed keys (a 4-step DPP row reduction over lanes 16..31).
// Everything is fp32 with one rounding per reference op (TU compiled with -ffp-contract=off).
#pragma once
#include "nastar_search.hip.h"

namespace nastar {

constexpr int CCL = 4;          // log2(cells per chunk)
constexpr int CCSZ = 1 << CCL;  // 16 cells per chunk == one DPP row

struct CompactDims {
    int H, W, HW;
    int nchunks;  // ceil(HW / 16)
    int HWp;      // nchunks * 16
    int CPL;      // chunk minima per lane = ceil(nchunks / 64)
    int NCp;      // CPL * 64 entries in cmin[]
    uint32_t magicW;
    float gr, omg, sqrtW;
};

struct CompactLds {
    float2* gc;
    unsigned long long* cmin;
    uint8_t* pdir;
    uint32_t* dump;  // 64 private scratch words: lanes with nothing to store write here instead of branching
};

__host__ __device__ inline size_t compact_lds_bytes(int HWp, int NCp) { return (size_t)HWp * 9 + (size_t)NCp * 8 + 256; }

__device__ __forceinline__ CompactLds carve_compact_lds(unsigned char* smem, const CompactDims& d)
{
    CompactLds l;
    l.gc = reinterpret_cast<float2*>(smem);
    l.cmin = reinterpret_cast<unsigned long long*>(l.gc + d.HWp);
    l.dump = reinterpret_cast<uint32_t*>(l.cmin + d.NCp);
    l.pdir = reinterpret_cast<uint8_t*>(l.dump + 64);
    return l;
}

__device__ __forceinline__ unsigned long long cmin_entry(uint32_t key, uint32_t idx)
{
    return ((unsigned long long)key << 32) | (unsigned long long)idx;
}

// fl(f / fl32(sqrt(W))) (:207) as the order-preserving u32 key
template <bool kFastDiv>
__device__ __forceinline__ uint32_t compact_key(const CompactDims& d, float G, float hh, float rcp_sqrtW)
{
    const float f = d.gr * G + hh;  // :206  f = g_ratio*g + (1-g_ratio)*h   (two roundings, no FMA)
    float q;
    if constexpr (kFastDiv) {
        // correctly rounded f / sqrt(W) for f >= 2^-100 (exhaustively verified per W, tools/fastdiv_check.c)
        const float q0 = f * rcp_sqrtW;
        const float rem = __builtin_fmaf(-q0, d.sqrtW, f);
        q = __builtin_fmaf(rem, rcp_sqrtW, q0);
    } else {
        q = f / d.sqrtW;
    }
    return f32_to_ord(q);
}

// get_heuristic (:26-52) for one cell of the COMPILED step loops (any H x W that fits LDS, e.g. a 5 x 200 strip): the square root is the
// corrected one (nastar_device.hip.h: sqrt_rn_int; the bare v_sqrt_f32 of the hand-scheduled 16 / 32 / 64 streams is exact for sides < 140)
__device__ __forceinline__ float heuristic0_fast(int r, int c, int goal_r, int goal_c)
{
    const float a = (float)(r - goal_r);
    const float b = (float)(c - goal_c);
    const float dr = fabsf(a), dc = fabsf(b);
    const float cheb = (dr + dc) - fminf(dr, dc);
    const float euc = sqrt_rn_int(a * a + b * b);
    return cheb + 0.001f * euc;
}

// ITER > 0: the map has exactly ITER * 256 cells (compile-time size): the loads of up to 4 iterations (16 x 16 B per lane) are all
// issued before the first is consumed, so a map costs ~one HBM latency to load instead of one per iteration.
// any_signed (optional): set when some cost is < 0 or NaN (wave-uniform) -- the round-3 instruction stream keys on raw float bits
template <bool kVec4, int ITER = 0>
__device__ __forceinline__ void compact_load_map(const CompactDims& d, const CompactLds& l, const float* __restrict__ cost,
                                                 const float* __restrict__ start, const float* __restrict__ goal,
                                                 const float* __restrict__ passable, int lane, int& start_idx, int& goal_idx,
                                                 bool* any_signed = nullptr)
{
    int sidx = -1, gidx = -1;
    bool sgn = false;
    if constexpr (kVec4) {
        const float4* s4 = reinterpret_cast<const float4*>(start);
        const float4* g4 = reinterpret_cast<const float4*>(goal);
        const float4* c4 = reinterpret_cast<const float4*>(cost);
        const float4* p4 = reinterpret_cast<const float4*>(passable);
        auto place = [&](int q, const float4& sv, const float4& gv, const float4& cv, const float4& pv) {
            const int i = q << 2;
            if (sv.x != 0.f) sidx = i;
            if (sv.y != 0.f) sidx = i + 1;
            if (sv.z != 0.f) sidx = i + 2;
            if (sv.w != 0.f) sidx = i + 3;
            if (gv.x != 0.f) gidx = i;
            if (gv.y != 0.f) gidx = i + 1;
            if (gv.z != 0.f) gidx = i + 2;
            if (gv.w != 0.f) gidx = i + 3;
            sgn |= !(cv.x >= 0.f) | !(cv.y >= 0.f) | !(cv.z >= 0.f) | !(cv.w >= 0.f);
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
        };
        if constexpr (ITER > 0) {
            constexpr int G = ITER < 4 ? ITER : 4;
            static_assert(ITER % G == 0, "groups of up to 4 iterations");
            // VanillaAstar hands ONE tensor over as cost map and obstacle map (astar.py:93-94): read it once (wave-uniform test)
            const bool same_cp = passable == cost;
            for (int base = 0; base < ITER; base += G) {
                float4 sv[G], gv[G], cv[G], pv[G];
#pragma unroll
                for (int k = 0; k < G; ++k) {
                    const int q = lane + (base + k) * 64;
                    sv[k] = s4[q];
                    gv[k] = g4[q];
                    cv[k] = c4[q];
                }
                if (same_cp) {
#pragma unroll
                    for (int k = 0; k < G; ++k) pv[k] = cv[k];
                } else {
#pragma unroll
                    for (int k = 0; k < G; ++k) pv[k] = p4[lane + (base + k) * 64];
                }
#pragma unroll
                for (int k = 0; k < G; ++k) place(lane + (base + k) * 64, sv[k], gv[k], cv[k], pv[k]);
            }
        } else {
            const int n4 = d.HW >> 2;
            for (int q = lane; q < n4; q += 64) place(q, s4[q], g4[q], c4[q], p4[q]);
        }
    } else {
        for (int i = lane; i < d.HW; i += 64) {
            if (start[i] != 0.f) sidx = i;
            if (goal[i] != 0.f) gidx = i;
            const float pv = passable[i];
            sgn |= !(cost[i] >= 0.f);
            l.gc[i] = make_float2(pv != 0.f ? NASTAR_POS_INF : NASTAR_NEG_INF, cost[i]);
            l.pdir[i] = (uint8_t)(PARENT_UNSET | (pv != 0.f ? P_PASS : 0u));
        }
    }
    for (int i = d.HW + lane; i < d.HWp; i += 64) l.gc[i] = make_float2(NASTAR_NEG_INF, 0.f);  // tail of the last chunk: never open
    start_idx = wave_max_i32(sidx);
    goal_idx = wave_max_i32(gidx);
    // idle chunk entries: key all ones; the cell field names the GOAL (round-4 stream, nastar_search_asm4.hip.h: an empty open list
    // then leaves through the goal exit -- every other selection path tests the key first and never looks at an idle entry's cell)
    const unsigned long long idle = cmin_entry(0xFFFFFFFFu, (uint32_t)(goal_idx < 0 ? 0 : goal_idx));
    for (int c = lane; c < d.NCp; c += 64) l.cmin[c] = idle;
    if (any_signed != nullptr) *any_signed = __ballot(sgn) != 0ull;
    wave_sync();
}

// open list = {start} (:187), g[start] = 0 (:193).  raw_key: the key is the bit pattern of q itself (nastar_search_asm3.hip.h, q >= +0)
// half_key: g_ratio == 0.5 form of the round-4 stream -- the key of q' = fl(fl(g + h) / sqrt(W)) = 2 q (nastar_search_asm4.hip.h)
template <bool kFastDiv>
__device__ __forceinline__ void compact_open_start(const CompactDims& d, const CompactLds& l, int lane, int sidx, int goal_r,
                                                   int goal_c, float rcp_sqrtW, bool raw_key = false, bool half_key = false)
{
    if (lane == 0) {
        const int r = (int)div_magic((uint32_t)sidx, d.magicW);
        const int c = sidx - r * d.W;
        const float hh = (half_key ? 1.0f : d.omg) * (heuristic0_fast(r, c, goal_r, goal_c) + l.gc[sidx].y);  // :191-192 h = h0 + cost ; :206
        uint32_t k0 = compact_key<kFastDiv>(d, 0.0f, hh, rcp_sqrtW);
        if (raw_key) k0 = __float_as_uint(ord_to_f32(k0));
        l.gc[sidx].x = 0.0f;
        l.cmin[sidx >> CCL] = cmin_entry(k0, (uint32_t)sidx);
        l.pdir[sidx] = (uint8_t)(PARENT_UNSET | P_PASS);  // the start is expanded even if it sits on an obstacle (:187)
    }
    wave_sync();
}

// ---- selection: first flat index of the minimal key; returns -1 when the open list is empty --------------------------
// CPL_T > 0: chunk minima per lane known at compile time (1: <= 1024 cells, 4: <= 4096 cells); 0: runtime d.CPL.
// Lane l owns the CONTIGUOUS entries [l*CPL, (l+1)*CPL): first lane == first chunk == first cell.
template <int CPL_T>
__device__ __forceinline__ int compact_select(const CompactDims& d, const CompactLds& l, int lane, uint2& mine)
{
    uint2 best;  // .x = cell index, .y = key
    if constexpr (CPL_T == 1) {
        const unsigned long long e = l.cmin[lane];  // one ds_read_b64: key and index arrive together
        best.x = (uint32_t)e;
        best.y = (uint32_t)(e >> 32);
    } else {
        const int cpl = CPL_T > 0 ? CPL_T : d.CPL;
        const uint2* p = reinterpret_cast<const uint2*>(l.cmin) + lane * cpl;
        best = p[0];
        if constexpr (CPL_T > 0) {
#pragma unroll
            for (int c = 1; c < CPL_T; ++c) {
                const uint2 e = p[c];
                if (e.y < best.y) best = e;  // strict: the earlier chunk wins ties
            }
        } else {
            for (int c = 1; c < cpl; ++c) {
                const uint2 e = p[c];
                if (e.y < best.y) best = e;
            }
        }
    }
    mine = best;
    const uint32_t M = wave_min_scalar_u32(best.y);
    const int L = __builtin_ctzll(__ballot(best.y == M));  // never empty: M is one of the lanes' keys
    const int s = __builtin_amdgcn_readlane((int)best.x, L);
    return M == KEY_INF ? -1 : s;  // open list empty
}

struct CompactLane {
    int dr, dc, off;
    bool is_nb;   // lanes 0..7: Moore neighbour j of s*
    bool is_chk;  // lanes 16..31: cell (lane - 16) of the chunk that holds s*
    uint32_t pcode;
};

__device__ __forceinline__ CompactLane make_compact_lane(const CompactDims& d, int lane)
{
    CompactLane lc;
    neighbour_delta(lane & 7, lc.dr, lc.dc);
    lc.is_nb = lane < 8;
    lc.is_chk = (lane & 48) == 16;
    lc.off = lc.dr * d.W + lc.dc;
    lc.pcode = P_PASS | (uint32_t)(lane & 7);
    return lc;
}

// ---- close s (:222-225) and relax its <= 8 Moore neighbours (:228-249); re-minimise the chunk of s without it ----------
// CPL_T == 1: `mine` is this lane's own cmin entry as read by compact_select (nothing has touched it since).
// keep_open (lock-step mode, NASTAR_FLAG_LOCKSTEP): s* is the goal -- the reference expands it like any cell but leaves it on the open list
// (:224 open_maps - is_unsolved * selected) while histories records it (:222-223); the caller keeps that flag
template <int LOGW, bool kFastDiv, int CPL_T>
__device__ __forceinline__ void compact_expand(const CompactDims& d, const CompactLds& l, const CompactLane& lc, int lane, int s,
                                               int goal_r, int goal_c, float rcp_sqrtW, const uint2 mine, const bool keep_open = false)
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
    // the cell this lane looks at: its neighbour of s*, its cell of the chunk, or s* itself (idle lanes)
    const int il = inb ? s + lc.off : (lc.is_chk ? cbase + (lane & (CCSZ - 1)) : s);
    // one batch of LDS reads: (g, cost) of s* (broadcast) and of this lane's cell
    const float2 gs = l.gc[s];
    const float2 gl = l.gc[il];
    // position of il (independent of the reads: overlaps their latency)
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
    // g2 = g[s*] + cost[s*]  (:234: expand((g + cost_maps) * selected)) -- step cost of the node being LEFT
    const float g2 = gs.x + gs.y;
    // :229,:235  neighbour is passable, not closed, and (not open, or open with g > g2)   <=>   g[n] > g2
    const bool upd = inb & (gl.x > g2);
    // neighbour lanes: key of the relaxed neighbour (g2); chunk lanes: current key of their cell (its own g)
    const uint32_t k = compact_key<kFastDiv>(d, lc.is_nb ? g2 : gl.x, hh, rcp_sqrtW);
    // chunk minimum without s*: open <=> finite g
    const bool open_l = lc.is_chk & (fabsf(gl.x) < NASTAR_POS_INF) & ((il != s) | keep_open);
    const uint32_t kk = open_l ? k : KEY_INF;
    const uint32_t mc = row_min16_u32(kk);
    const unsigned long long firstm = __ballot(lc.is_chk & (kk == mc));  // bits 16..31; never empty
    const uint32_t Mc = (uint32_t)__builtin_amdgcn_readlane((int)mc, 16);
    const uint32_t ci = (uint32_t)(cbase + __builtin_ctzll(firstm) - 16);
    // All stores are unconditional: a lane with nothing to write targets its private dump word / a no-op atomic.
    uint32_t* const dmp = l.dump + lane;
    float* const g_dst = upd ? &l.gc[il].x : ((lane == 8 && !keep_open) ? &l.gc[s].x : reinterpret_cast<float*>(dmp));
    uint8_t* const p_dst = upd ? &l.pdir[il] : reinterpret_cast<uint8_t*>(dmp);
    *g_dst = upd ? g2 : NASTAR_NEG_INF;  // :238 g update          | :222-225 s* joins the closed list, leaves the open list
    *p_dst = (uint8_t)lc.pcode;          // :246-249 parent = s*
    // exact minimum of the chunk without s* (must land before the atomics below; LDS executes a wave's ops in order)
    if constexpr (CPL_T == 1) {
        // every lane rewrites its own entry -- unchanged, except the owner of chunk C: one unmasked ds_write_b64
        const bool owner = lane == (s >> CCL);
        uint2 e;
        e.x = owner ? ci : mine.x;
        e.y = owner ? Mc : mine.y;
        reinterpret_cast<uint2*>(l.cmin)[lane] = e;
    } else {
        if (lane == 16) l.cmin[s >> CCL] = cmin_entry(Mc, ci);
    }
    // :242 (re)opened neighbours enter their chunk's minimum; idle lanes issue min(x, ~0) on their own entry: a no-op
    atomicMin(upd ? &l.cmin[il >> CCL] : &l.cmin[lane], upd ? cmin_entry(k, (uint32_t)il) : ~0ull);
    wave_order();
}

__device__ __forceinline__ int compact_parent_of(const CompactDims& d, int n, uint32_t code)
{
    int dr, dc;
    neighbour_delta((int)code, dr, dc);
    return n - (dr * d.W + dc);
}

// backtrack (differentiable_astar.py:96-125), see nastar_search.hip.h::backtrack for the equivalence argument.
// The walk is a serial pointer chase by lane 0 on the launch's critical path (the longest search usually has a long path): one LDS
// round trip per hop + the parent decode.  LOGW > 0 (W = 2^LOGW <= 64): the eight "cell - parent" offsets dr*W + dc sit as signed
// bytes in one 64-bit constant, so the decode is shift / sign-extend / subtract instead of neighbour_delta's compare chain.
template <int LOGW = 0>
__device__ __forceinline__ void compact_backtrack(const CompactDims& d, const CompactLds& l, int lane, int start_idx, int goal_idx,
                                                  int cap)
{
    if (lane == 0) {
        auto parent_of = [&](int n, uint32_t code) -> int {
            if constexpr (LOGW > 0 && LOGW <= 6) {
                constexpr int W = 1 << LOGW;
                constexpr unsigned long long LUT =
                    ((unsigned long long)(uint8_t)(int8_t)(-W - 1)) | ((unsigned long long)(uint8_t)(int8_t)(-W) << 8) |
                    ((unsigned long long)(uint8_t)(int8_t)(-W + 1) << 16) | ((unsigned long long)(uint8_t)(int8_t)(-1) << 24) |
                    ((unsigned long long)(uint8_t)(int8_t)(1) << 32) | ((unsigned long long)(uint8_t)(int8_t)(W - 1) << 40) |
                    ((unsigned long long)(uint8_t)(int8_t)(W) << 48) | ((unsigned long long)(uint8_t)(int8_t)(W + 1) << 56);
                return n - (int)(int8_t)(LUT >> (code * 8u));
            } else {
                return compact_parent_of(d, n, code);
            }
        };
        uint32_t m = l.pdir[goal_idx];
        l.pdir[goal_idx] = (uint8_t)(m | P_PATH);
        uint32_t code = m & P_DIRMASK;
        if (code != PARENT_UNSET) {
            int loc = parent_of(goal_idx, code);
            for (int k = 0; k < cap; ++k) {
                uint32_t ml = l.pdir[loc];
                l.pdir[loc] = (uint8_t)(ml | P_PATH);
                if (loc == start_idx) break;
                uint32_t cd = ml & P_DIRMASK;
                if (cd == PARENT_UNSET) break;  // cannot happen for an opened non-start node
                loc = parent_of(loc, cd);
            }
        }
    }
    wave_sync();
}

// AstarOutput.histories alone (fp32 0/1): depends on the closed list only, so it can be written BEFORE the serial backtrack -- the
// stores drain while lane 0 walks the parent chain (compact_store_outputs<.., kHist = false> then writes the paths)
template <bool kVec4>
__device__ __forceinline__ void compact_store_hist(const CompactDims& d, const CompactLds& l, int lane, float* __restrict__ hist)
{
    if constexpr (kVec4) {
        const int n4 = d.HW >> 2;
        float4* h4 = reinterpret_cast<float4*>(hist);
        for (int q = lane; q < n4; q += 64) {
            const uint32_t m = *reinterpret_cast<const uint32_t*>(l.pdir + (q << 2));
            const float4 lo = *reinterpret_cast<const float4*>(l.gc + (q << 2));
            const float4 hi = *reinterpret_cast<const float4*>(l.gc + (q << 2) + 2);
            float4 v;
            v.x = ((m & P_PASS) && lo.x == NASTAR_NEG_INF) ? 1.0f : 0.0f;
            v.y = ((m & (P_PASS << 8)) && lo.z == NASTAR_NEG_INF) ? 1.0f : 0.0f;
            v.z = ((m & (P_PASS << 16)) && hi.x == NASTAR_NEG_INF) ? 1.0f : 0.0f;
            v.w = ((m & (P_PASS << 24)) && hi.z == NASTAR_NEG_INF) ? 1.0f : 0.0f;
            h4[q] = v;
        }
    } else {
        for (int i = lane; i < d.HW; i += 64) hist[i] = ((l.pdir[i] & P_PASS) && l.gc[i].x == NASTAR_NEG_INF) ? 1.0f : 0.0f;
    }
}

// AstarOutput.histories (fp32 0/1), .paths (int64 0/1) and optionally the 2-bit-per-cell packed masks (see store_outputs)
template <bool kVec4, bool kHist = true>
__device__ __forceinline__ void compact_store_outputs(const CompactDims& d, const CompactLds& l, int lane, float* __restrict__ hist,
                                                      long long* __restrict__ paths, uint8_t* __restrict__ packed = nullptr)
{
    if constexpr (kVec4) {
        const int n4 = d.HW >> 2;
        float4* h4 = reinterpret_cast<float4*>(hist);
        if (kHist || packed != nullptr)
        for (int q = lane; q < n4; q += 64) {
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
            if (kHist) h4[q] = v;
            if (packed != nullptr) {  // wave-uniform
                const uint32_t nh = (c0 ? 8u : 0u) | (c1 ? 4u : 0u) | (c2 ? 2u : 0u) | (c3 ? 1u : 0u);
                const uint32_t np = ((m & P_PATH) ? 8u : 0u) | ((m & (P_PATH << 8)) ? 4u : 0u) |
                                    ((m & (P_PATH << 16)) ? 2u : 0u) | ((m & (P_PATH << 24)) ? 1u : 0u);
                const uint32_t both = nh | (np << 8);
                const uint32_t other = dpp_mov<DPP_QUAD_XOR1>(both);  // the odd lane's quad = low nibble of the byte
                if ((lane & 1) == 0) {
                    const int nb = d.HW >> 3;
                    packed[q >> 1] = (uint8_t)((nh << 4) | (other & 0xFu));
                    packed[nb + (q >> 1)] = (uint8_t)((np << 4) | ((other >> 8) & 0xFu));
                }
            }
        }
        const int n2 = d.HW >> 1;
        longlong2* p2 = reinterpret_cast<longlong2*>(paths);
        for (int q = lane; q < n2; q += 64) {
            const uint32_t m = *reinterpret_cast<const uint16_t*>(l.pdir + (q << 1));
            longlong2 v;
            v.x = (m & P_PATH) ? 1 : 0;
            v.y = (m & (P_PATH << 8)) ? 1 : 0;
            p2[q] = v;
        }
    } else {
        for (int i = lane; i < d.HW; i += 64) {
            const uint32_t m = l.pdir[i];
            if (kHist) hist[i] = ((m & P_PASS) && l.gc[i].x == NASTAR_NEG_INF) ? 1.0f : 0.0f;
            paths[i] = (m & P_PATH) ? 1 : 0;
        }
    }
}

}  // namespace nastar
