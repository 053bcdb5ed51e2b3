// nastar_search.hip.h -- the on-chip A* search state machine: one 64-lane wavefront owns one map.
//
// Replaces the reference's per-iteration tensor program (differentiable_astar.py:203-252; ~45 ATen ops over
// [B,H,W] fp32 maps to move ONE node per map) by:
//   * LDS-resident state per map:
//       key[]   q = fl(f / fl32(sqrt(W))) of every OPEN cell as an order-preserving u32, KEY_INF otherwise
//       g[]     fp32 g-value; the sign of infinity doubles as the node state so that ONE comparison decides a
//               relaxation:  +inf = passable & never opened,  -inf = closed or obstacle,  finite = open.
//               "update neighbour n" (differentiable_astar.py:235: (1-open)(1-hist) + open*(g > g2), times the
//               obstacle mask :229) is exactly  g[n] > g2.
//       cost[], hh[] = fl((1-g_ratio) * fl(h0 + cost))     (:191-192,:206)
//       pdir[]  bits 0-3 parent direction code (8 = unset), bit 6 passable, bit 7 on-path
//       chunkmin[] min key of every 64-cell chunk, maintained with ds_min_u32
//   * selection  = first-index arg-min over the open list of q -- the reference's first arg-max of the masked
//     softmax exp(-f/sqrt(W))/sum (:55-74,:206-209) orders cells by exactly this quotient: the IEEE division merges
//     f values one ulp apart into exact ties that then resolve by flat index, so the key must be q, not f.
//     wave-min over chunkmin[] -> first chunk -> wave ballot inside the chunk: two DPP reductions + two ballots
//     per step instead of a full-map softmax;
//   * expansion  = lanes 0..7 each own one Moore neighbour (expand() of a one-hot == "touch <=8 cells", :77-93,
//     :228-249); all LDS reads of a step are issued as one batch, no read-modify-write, no data-dependent branches.
// Everything is fp32 with one rounding per reference op (TU compiled with -ffp-contract=off).
#pragma once
#include "nastar_device.hip.h"

namespace nastar {

struct MapDims {
    int H, W, HW;
    int nchunks;  // ceil(HW / 64)
    int HWp;      // nchunks * 64  (key[] is padded with KEY_INF up to here)
    int NCp;      // chunkmin[] length: nchunks rounded up to a multiple of 64 (>= 64)
    uint32_t magicW;
    float gr, omg;  // fl32(g_ratio), fl32(1 - g_ratio)  (differentiable_astar.py:206)
    float sqrtW;    // fl32(math.sqrt(W))                 (differentiable_astar.py:207)
};

struct MapLds {
    uint32_t* key;
    float* g;
    float* cost;
    float* hh;
    uint32_t* chunkmin;
    uint8_t* pdir;
    uint32_t* dump;  // 64 scratch words: lanes with nothing to write store here instead of branching around the store
};

constexpr uint32_t P_DIRMASK = 0x0Fu, P_PASS = 0x40u, P_PATH = 0x80u;
#define NASTAR_POS_INF (__uint_as_float(0x7f800000u))
#define NASTAR_NEG_INF (__uint_as_float(0xff800000u))

__host__ __device__ inline size_t map_lds_bytes(int HWp, int NCp) { return (size_t)HWp * 17 + (size_t)NCp * 4 + 256; }

__device__ __forceinline__ MapLds carve_map_lds(unsigned char* smem, const MapDims& d)
{
    MapLds l;
    l.key = reinterpret_cast<uint32_t*>(smem);
    l.g = reinterpret_cast<float*>(l.key + d.HWp);
    l.cost = l.g + d.HWp;
    l.hh = l.cost + d.HWp;
    l.chunkmin = reinterpret_cast<uint32_t*>(l.hh + d.HWp);
    l.dump = l.chunkmin + d.NCp;
    l.pdir = reinterpret_cast<uint8_t*>(l.dump + 64);
    return l;
}

// single-wave workgroup: this is a scheduling + LDS-visibility point, not an s_barrier
__device__ __forceinline__ void wave_sync() { __syncthreads(); }

__device__ __forceinline__ uint32_t make_key(const MapDims& d, float g2, float hh)
{
    const float f = d.gr * g2 + hh;       // :206  f = g_ratio*g + (1-g_ratio)*h   (two roundings, no FMA)
    return f32_to_ord(f / d.sqrtW);       // :207  -1*f / sqrt(W): IEEE fp32 division (negation is exact)
}

// ---- load one map from HBM into LDS; returns start / goal flat indices (wave-uniform) -------------------------
// CL = log2(cells per chunk): 6 (one chunk = one wavefront-wide row read) or 4 (one chunk = one 16-lane DPP row, used
// for 32x32 where it still gives <= 64 chunks: the "chunk minimum without s*" is then a 4-step row reduction).
template <bool kVec4, int CL = 6>
__device__ __forceinline__ void load_map(const MapDims& d, const MapLds& l, const float* __restrict__ cost,
                                         const float* __restrict__ start, const float* __restrict__ goal,
                                         const float* __restrict__ passable, int lane, int& start_idx,
                                         int& goal_idx)
{
    int sidx = -1, gidx = -1;
    if constexpr (kVec4) {
        const float4* s4 = reinterpret_cast<const float4*>(start);
        const float4* g4 = reinterpret_cast<const float4*>(goal);
        const int n4 = d.HW >> 2;
        for (int q = lane; q < n4; q += 64) {
            float4 sv = s4[q];
            float4 gv = g4[q];
            int i = q << 2;
            if (sv.x != 0.f) sidx = i;
            if (sv.y != 0.f) sidx = i + 1;
            if (sv.z != 0.f) sidx = i + 2;
            if (sv.w != 0.f) sidx = i + 3;
            if (gv.x != 0.f) gidx = i;
            if (gv.y != 0.f) gidx = i + 1;
            if (gv.z != 0.f) gidx = i + 2;
            if (gv.w != 0.f) gidx = i + 3;
        }
    } else {
        for (int i = lane; i < d.HW; i += 64) {
            if (start[i] != 0.f) sidx = i;
            if (goal[i] != 0.f) gidx = i;
        }
    }
    sidx = wave_max_i32(sidx);
    gidx = wave_max_i32(gidx);
    start_idx = sidx;
    goal_idx = gidx;
    const int gi = gidx < 0 ? 0 : gidx;
    const int goal_r = (int)div_magic((uint32_t)gi, d.magicW);
    const int goal_c = gi - goal_r * d.W;

    if constexpr (kVec4) {
        const float4* c4 = reinterpret_cast<const float4*>(cost);
        const float4* p4 = reinterpret_cast<const float4*>(passable);
        const int n4 = d.HW >> 2;
        for (int q = lane; q < n4; q += 64) {
            float4 cv = c4[q];
            float4 pv = p4[q];
            int i = q << 2;
            int r = (int)div_magic((uint32_t)i, d.magicW);
            int c = i - r * d.W;  // W % 4 == 0: the four cells share a row
            float4 hv, gv;
            hv.x = d.omg * (heuristic0(r, c, goal_r, goal_c) + cv.x);  // :191-192 h = h0 + cost ; :206 (1-g_ratio)*h
            hv.y = d.omg * (heuristic0(r, c + 1, goal_r, goal_c) + cv.y);
            hv.z = d.omg * (heuristic0(r, c + 2, goal_r, goal_c) + cv.z);
            hv.w = d.omg * (heuristic0(r, c + 3, goal_r, goal_c) + cv.w);
            gv.x = pv.x != 0.f ? NASTAR_POS_INF : NASTAR_NEG_INF;
            gv.y = pv.y != 0.f ? NASTAR_POS_INF : NASTAR_NEG_INF;
            gv.z = pv.z != 0.f ? NASTAR_POS_INF : NASTAR_NEG_INF;
            gv.w = pv.w != 0.f ? NASTAR_POS_INF : NASTAR_NEG_INF;
            *reinterpret_cast<float4*>(l.cost + i) = cv;
            *reinterpret_cast<float4*>(l.hh + i) = hv;
            *reinterpret_cast<float4*>(l.g + i) = gv;
            *reinterpret_cast<uint4*>(l.key + i) = make_uint4(KEY_INF, KEY_INF, KEY_INF, KEY_INF);
            uint32_t m = (PARENT_UNSET | (pv.x != 0.f ? P_PASS : 0u)) | ((PARENT_UNSET | (pv.y != 0.f ? P_PASS : 0u)) << 8) |
                         ((PARENT_UNSET | (pv.z != 0.f ? P_PASS : 0u)) << 16) |
                         ((PARENT_UNSET | (pv.w != 0.f ? P_PASS : 0u)) << 24);
            *reinterpret_cast<uint32_t*>(l.pdir + i) = m;
        }
    } else {
        for (int i = lane; i < d.HW; i += 64) {
            float cv = cost[i];
            float pv = passable[i];
            int r = (int)div_magic((uint32_t)i, d.magicW);
            int c = i - r * d.W;
            l.cost[i] = cv;
            l.hh[i] = d.omg * (heuristic0(r, c, goal_r, goal_c) + cv);
            l.g[i] = pv != 0.f ? NASTAR_POS_INF : NASTAR_NEG_INF;
            l.key[i] = KEY_INF;
            l.pdir[i] = (uint8_t)(PARENT_UNSET | (pv != 0.f ? P_PASS : 0u));
        }
    }
    for (int i = d.HW + lane; i < d.HWp; i += 64) l.key[i] = KEY_INF;  // tail padding of the last chunk
    for (int c = lane; c < d.NCp; c += 64) l.chunkmin[c] = KEY_INF;
    wave_sync();
    // open list = {start} (:187), g[start] = 0 (:193)
    if (lane == 0 && sidx >= 0) {
        const uint32_t k0 = make_key(d, 0.0f, l.hh[sidx]);
        l.g[sidx] = 0.0f;
        l.key[sidx] = k0;
        l.chunkmin[sidx >> CL] = k0;
        l.pdir[sidx] = (uint8_t)(PARENT_UNSET | P_PASS);  // the start is expanded even if it sits on an obstacle (:187)
    }
    wave_sync();
}

// per-lane constants of the expansion: lane j < 8 owns Moore neighbour j
struct LaneConst {
    int dr, dc;   // neighbour offset of this lane (lanes >= 8: 0,0)
    int off;      // dr * W + dc
    bool is_nb;   // lane < 8
    uint32_t pcode;  // pdir byte this lane writes when it relaxes its neighbour
};

__device__ __forceinline__ LaneConst make_lane_const(const MapDims& d, int lane)
{
    LaneConst lc;
    neighbour_delta(lane & 7, lc.dr, lc.dc);
    lc.is_nb = lane < 8;
    lc.off = lc.dr * d.W + lc.dc;
    lc.pcode = P_PASS | (uint32_t)(lane & 7);
    return lc;
}

// LDS accesses of one wavefront execute in program order, so the hand-off between the lanes of a step needs no
// s_waitcnt -- only a compiler-level ordering point.
__device__ __forceinline__ void wave_order() { __builtin_amdgcn_wave_barrier(); }

// ---- selection: first flat index of the minimal key; returns -1 when the open list is empty ------------------
// On return kv is the key this lane read from the selected chunk C (lane cl holds the selected cell).
template <bool kMultiChunk, int CL = 6>
__device__ __forceinline__ int select_min(const MapDims& d, const MapLds& l, int lane, int& C, int& cl, uint32_t& kv)
{
    constexpr int CSZ = 1 << CL;
    uint32_t Mv;  // the minimum, replicated in every lane
    unsigned long long any_open;
    if constexpr (!kMultiChunk) {
        const uint32_t cm = l.chunkmin[lane];
        any_open = __ballot(cm != KEY_INF);
        Mv = wave_min_scalar_u32(cm);
        C = __builtin_ctzll(__ballot(cm == Mv) | (1ull << 63));
    } else {
        uint32_t best = KEY_INF;
        int bestc = 0x7fffffff;
        for (int c = lane; c < d.nchunks; c += 64) {
            uint32_t v = l.chunkmin[c];
            if (v < best) { best = v; bestc = c; }
        }
        any_open = __ballot(best != KEY_INF);
        Mv = wave_min_all_u32(best);
        C = (int)__builtin_amdgcn_readfirstlane((int)wave_min_all_u32(best == Mv ? (uint32_t)bestc : 0x7fffffffu));
        if (C >= d.nchunks) C = 0;
    }
    kv = l.key[C * CSZ + (lane & (CSZ - 1))];  // CL == 4: the four 16-lane rows read the same 16 keys
    if (any_open == 0) return -1;  // open list empty (every chunk minimum is KEY_INF)
    cl = __builtin_ctzll(__ballot(kv == Mv) | (1ull << 63));
    return C * CSZ + cl;
}

// ---- close s (:222-225) and relax its <=8 Moore neighbours (:228-249) --------------------------------------
// keep_open: the selected node is the goal being stepped at its fixed point (backward only, :224).
// LOGW > 0: W == 1 << LOGW at compile time.
template <int LOGW, bool kFastDiv, int CL = 6>
__device__ __forceinline__ void close_and_expand(const MapDims& d, const MapLds& l, const LaneConst& lc, int lane, int s,
                                                 int C, int cl, uint32_t kv, bool keep_open, float rcp_sqrtW)
{
    int r, c;
    if constexpr (LOGW) {
        r = s >> LOGW;
        c = s & ((1 << LOGW) - 1);
    } else {
        r = (int)div_magic((uint32_t)s, d.magicW);
        c = s - r * d.W;
    }
    // one batch of LDS reads: g[s*], cost[s*] (broadcast) and each neighbour lane's g[n], hh[n]
    const int nr = r + lc.dr, nc = c + lc.dc;
    const bool inb = lc.is_nb & ((unsigned)nr < (unsigned)d.H) & ((unsigned)nc < (unsigned)d.W);  // conv2d zero padding
    const int n = inb ? s + lc.off : s;
    const float gs = l.g[s];
    const float cs = l.cost[s];
    const float gn = l.g[n];
    const float hn = l.hh[n];
    // chunk minimum without s (independent of the reads above, overlaps their latency)
    uint32_t nm;
    if constexpr (CL == 4) nm = row_min16_u32((lane & 15) == cl ? KEY_INF : kv);
    else nm = wave_min_all_u32(lane == cl ? KEY_INF : kv);
    // g2 = g[s*] + cost[s*]  (:234: expand((g + cost_maps) * selected)) -- step cost of the node being LEFT
    const float g2 = gs + cs;
    // :229,:235  neighbour is passable, not closed, and (not open, or open with g > g2)   <=>   g[n] > g2
    const bool upd = inb & (gn > g2);
    const float f = d.gr * g2 + hn;   // :206  f = g_ratio*g + (1-g_ratio)*h   (two roundings, no FMA)
    float q;
    if constexpr (kFastDiv) {
        // correctly rounded f / sqrt(W) for f >= 2^-100 (exhaustively verified per W, tools/fastdiv_check.c):
        // q0 = RN(f*y), rem = f - q0*b exactly (FMA), q = RN(q0 + rem*y)
        const float q0 = f * rcp_sqrtW;
        const float rem = __builtin_fmaf(-q0, d.sqrtW, f);
        q = __builtin_fmaf(rem, rcp_sqrtW, q0);
    } else {
        q = f / d.sqrtW;               // :207  -1*f / sqrt(W): IEEE fp32 division (negation is exact)
    }
    const uint32_t k = f32_to_ord(q);
    // All stores are unconditional: a lane with nothing to write targets its private dump word, so the step has no
    // exec-mask regions / skip branches.  Lane 8 closes s*, lane 9 publishes the chunk minimum without s*.
    const bool closer = (lane == 8) && !keep_open;
    uint32_t* const dmp = l.dump + lane;
    float* const g_dst = upd ? &l.g[n] : (closer ? &l.g[s] : reinterpret_cast<float*>(dmp));
    uint32_t* const k_dst = upd ? &l.key[n] : (closer ? &l.key[s] : dmp);
    uint8_t* const p_dst = upd ? &l.pdir[n] : reinterpret_cast<uint8_t*>(dmp);
    uint32_t* const c_dst = ((lane == 9) && !keep_open) ? &l.chunkmin[C] : dmp;
    uint32_t* const a_dst = upd ? &l.chunkmin[n >> CL] : dmp;
    *g_dst = upd ? g2 : NASTAR_NEG_INF;      // :238 g update          | :222-223 s* joins the closed list
    *k_dst = upd ? k : KEY_INF;              // :242 (re)opened        | :224 s* leaves the open list
    *p_dst = (uint8_t)lc.pcode;              // :246-249 parent = s*
    *c_dst = nm;                             // chunk minimum of C without s* (before the atomics below)
    atomicMin(a_dst, k);
    wave_order();
}

// parent of cell n from its direction code (code j means "n is neighbour j of its parent")
__device__ __forceinline__ int parent_of(const MapDims& d, int n, uint32_t code)
{
    int dr, dc;
    neighbour_delta((int)code, dr, dc);
    return n - (dr * d.W + dc);
}

// ---- backtrack (differentiable_astar.py:96-125): mark P_PATH from the goal towards the start ------------------
// The reference walks exactly t_batch steps; once the start is reached the walk re-enters the same cycle
// (parents[start] keeps its initial value goal_idx), so stopping at the start is equivalent as long as at most
// `cap` steps are taken (cap = own step count - 1 matters only when the Tmax budget ran out).
__device__ __forceinline__ void backtrack(const MapDims& d, const MapLds& l, int lane, int start_idx, int goal_idx,
                                          int cap)
{
    if (lane == 0) {
        uint32_t m = l.pdir[goal_idx];
        l.pdir[goal_idx] = (uint8_t)(m | P_PATH);
        uint32_t code = m & P_DIRMASK;
        if (code != PARENT_UNSET) {
            int loc = parent_of(d, goal_idx, code);
            for (int k = 0; k < cap; ++k) {
                uint32_t ml = l.pdir[loc];
                l.pdir[loc] = (uint8_t)(ml | P_PATH);
                if (loc == start_idx) break;
                uint32_t cd = ml & P_DIRMASK;
                if (cd == PARENT_UNSET) break;  // cannot happen for an opened non-start node
                loc = parent_of(d, loc, cd);
            }
        }
    }
    wave_sync();
}

// ---- write AstarOutput.histories (fp32 0/1) and .paths (int64 0/1) with 16-byte coalesced stores --------------
// closed <=> g == -inf on a passable cell.
// packed (optional, kVec4 and HW % 8 == 0 only): the same masks as 2 bits per cell -- [HW/8 bytes histories | HW/8 bytes
// paths], MSB = first cell -- the payload of the multi-GPU all-gather, emitted here so that no second kernel has to
// re-read 12 bytes per cell.
template <bool kVec4>
__device__ __forceinline__ void store_outputs(const MapDims& d, const MapLds& l, int lane, float* __restrict__ hist,
                                              long long* __restrict__ paths, uint8_t* __restrict__ packed = nullptr)
{
    if constexpr (kVec4) {
        const int n4 = d.HW >> 2;
        float4* h4 = reinterpret_cast<float4*>(hist);
        for (int q = lane; q < n4; q += 64) {
            const uint32_t m = *reinterpret_cast<const uint32_t*>(l.pdir + (q << 2));
            const float4 gv = *reinterpret_cast<const float4*>(l.g + (q << 2));
            const bool c0 = (m & P_PASS) && gv.x == NASTAR_NEG_INF;
            const bool c1 = (m & (P_PASS << 8)) && gv.y == NASTAR_NEG_INF;
            const bool c2 = (m & (P_PASS << 16)) && gv.z == NASTAR_NEG_INF;
            const bool c3 = (m & (P_PASS << 24)) && gv.w == NASTAR_NEG_INF;
            float4 v;
            v.x = c0 ? 1.0f : 0.0f;
            v.y = c1 ? 1.0f : 0.0f;
            v.z = c2 ? 1.0f : 0.0f;
            v.w = c3 ? 1.0f : 0.0f;
            h4[q] = v;
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
            hist[i] = ((m & P_PASS) && l.g[i] == NASTAR_NEG_INF) ? 1.0f : 0.0f;
            paths[i] = (m & P_PATH) ? 1 : 0;
        }
    }
}

}  // namespace nastar
