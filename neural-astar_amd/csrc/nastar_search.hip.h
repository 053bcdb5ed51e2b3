// nastar_search.hip.h -- the on-chip A* search state machine: one 64-lane wavefront owns one map.
//
// Replaces the reference's per-iteration tensor program (differentiable_astar.py:203-252; ~45 ATen ops over
// [B,H,W] fp32 maps to move ONE node per map) by:
//   * LDS-resident state per map:  key[] (q = f/sqrt(W) of open cells as order-preserving u32, KEY_INF otherwise),
//     g[], cost[], hh[] = fl((1-g_ratio)*fl(h0+cost)), meta[] (flags + parent direction), chunkmin[] (min key of
//     every 64-cell chunk, maintained with ds_min_u32);
//   * selection  = first-index arg-min over the open list of q = fl(f / fl32(sqrt(W))) -- the reference's first
//     arg-max of the masked softmax exp(-f/sqrt(W))/sum (differentiable_astar.py:55-74,206-209) orders cells by
//     exactly this quotient: the IEEE division merges f values one ulp apart into exact ties that then resolve by
//     flat index, so the key must be q, not f.  Wave-min over chunkmin[] -> first chunk -> wave ballot inside the
//     chunk: two DPP reductions + two ballots per step instead of a full-map softmax;
//   * expansion  = lanes 0..7 each own one Moore neighbour (expand() of a one-hot == "touch <=8 cells",
//     differentiable_astar.py:77-93,228-249).
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
    uint8_t* meta;
};

__host__ __device__ inline size_t map_lds_bytes(int HWp, int NCp) { return (size_t)HWp * 17 + (size_t)NCp * 4; }

__device__ __forceinline__ MapLds carve_map_lds(unsigned char* smem, const MapDims& d)
{
    MapLds l;
    l.key = reinterpret_cast<uint32_t*>(smem);
    l.g = reinterpret_cast<float*>(l.key + d.HWp);
    l.cost = l.g + d.HWp;
    l.hh = l.cost + d.HWp;
    l.chunkmin = reinterpret_cast<uint32_t*>(l.hh + d.HWp);
    l.meta = reinterpret_cast<uint8_t*>(l.chunkmin + d.NCp);
    return l;
}

// single-wave workgroup: this is a scheduling + LDS-visibility point, not an s_barrier
__device__ __forceinline__ void wave_sync() { __syncthreads(); }

// ---- load one map from HBM into LDS; returns start / goal flat indices (wave-uniform) -------------------------
template <bool kVec4>
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
            float4 hv;
            hv.x = d.omg * (heuristic0(r, c, goal_r, goal_c) + cv.x);  // :191-192 h = h0 + cost ; :206 (1-g_ratio)*h
            hv.y = d.omg * (heuristic0(r, c + 1, goal_r, goal_c) + cv.y);
            hv.z = d.omg * (heuristic0(r, c + 2, goal_r, goal_c) + cv.z);
            hv.w = d.omg * (heuristic0(r, c + 3, goal_r, goal_c) + cv.w);
            *reinterpret_cast<float4*>(l.cost + i) = cv;
            *reinterpret_cast<float4*>(l.hh + i) = hv;
            *reinterpret_cast<uint4*>(l.key + i) = make_uint4(KEY_INF, KEY_INF, KEY_INF, KEY_INF);
            const uint32_t un = PARENT_UNSET << 4;
            uint32_t m = (un | (pv.x != 0.f ? M_PASS : 0u)) | ((un | (pv.y != 0.f ? M_PASS : 0u)) << 8) |
                         ((un | (pv.z != 0.f ? M_PASS : 0u)) << 16) | ((un | (pv.w != 0.f ? M_PASS : 0u)) << 24);
            *reinterpret_cast<uint32_t*>(l.meta + i) = m;
        }
    } else {
        for (int i = lane; i < d.HW; i += 64) {
            float cv = cost[i];
            float pv = passable[i];
            int r = (int)div_magic((uint32_t)i, d.magicW);
            int c = i - r * d.W;
            l.cost[i] = cv;
            l.hh[i] = d.omg * (heuristic0(r, c, goal_r, goal_c) + cv);
            l.key[i] = KEY_INF;
            l.meta[i] = (uint8_t)((PARENT_UNSET << 4) | (pv != 0.f ? M_PASS : 0u));
        }
    }
    for (int i = d.HW + lane; i < d.HWp; i += 64) l.key[i] = KEY_INF;  // tail padding of the last chunk
    for (int c = lane; c < d.NCp; c += 64) l.chunkmin[c] = KEY_INF;
    wave_sync();
    // open list = {start} (:187), g[start] = 0 (:193)
    if (lane == 0 && sidx >= 0) {
        float f0 = d.gr * 0.0f + l.hh[sidx];
        uint32_t k0 = f32_to_ord(f0 / d.sqrtW);
        l.g[sidx] = 0.0f;
        l.key[sidx] = k0;
        l.meta[sidx] = (uint8_t)(l.meta[sidx] | M_OPEN);
        l.chunkmin[sidx >> 6] = k0;
    }
    wave_sync();
}

// ---- selection: first flat index of the minimal key; returns -1 when the open list is empty ------------------
// On return kv is the key this lane read from the selected chunk C (lane cl holds the selected cell).
template <bool kMultiChunk>
__device__ __forceinline__ int select_min(const MapDims& d, const MapLds& l, int lane, int& C, int& cl, uint32_t& kv,
                                          uint32_t& M)
{
    if constexpr (!kMultiChunk) {
        uint32_t cm = l.chunkmin[lane];
        M = wave_min_u32(cm);
        if (M == KEY_INF) return -1;
        unsigned long long bal = __ballot(cm == M);
        C = __builtin_ctzll(bal);
    } else {
        uint32_t best = KEY_INF;
        int bestc = 0x7fffffff;
        for (int c = lane; c < d.nchunks; c += 64) {
            uint32_t v = l.chunkmin[c];
            if (v < best) { best = v; bestc = c; }
        }
        M = wave_min_u32(best);
        if (M == KEY_INF) return -1;
        C = (int)wave_min_u32(best == M ? (uint32_t)bestc : 0x7fffffffu);
    }
    kv = l.key[C * CHUNK + lane];
    unsigned long long bal2 = __ballot(kv == M);
    cl = __builtin_ctzll(bal2);
    return C * CHUNK + cl;
}

// ---- close s (:222-225) and relax its <=8 Moore neighbours (:228-249) --------------------------------------
__device__ __forceinline__ void close_and_expand(const MapDims& d, const MapLds& l, int lane, int s, int C, int cl,
                                                 uint32_t kv, bool keep_open)
{
    // g2 = g[s*] + cost[s*]  (:234: expand((g + cost_maps) * selected)) -- step cost of the node being LEFT
    const float g2 = l.g[s] + l.cost[s];
    if (!keep_open) {
        const uint32_t nm = wave_min_u32(lane == cl ? KEY_INF : kv);  // chunk minimum without s
        if (lane == 0) {
            l.key[s] = KEY_INF;
            l.chunkmin[C] = nm;
            l.meta[s] = (uint8_t)((l.meta[s] & ~M_OPEN) | M_CLOSED);
        }
    } else if (lane == 0) {
        l.meta[s] = (uint8_t)(l.meta[s] | M_CLOSED);  // a reached goal stays on the open list (:224)
    }
    const int r = (int)div_magic((uint32_t)s, d.magicW);
    const int c = s - r * d.W;
    if (lane < 8) {
        int dr, dc;
        neighbour_delta(lane, dr, dc);
        const int nr = r + dr, nc = c + dc;
        if ((unsigned)nr < (unsigned)d.H && (unsigned)nc < (unsigned)d.W) {  // zero padding of conv2d, no wrap
            const int n = s + dr * d.W + dc;
            const uint32_t m = l.meta[n];
            if ((m & M_PASS) && !(m & M_CLOSED)) {                            // :229 * obstacles ; (1 - histories)
                const bool is_open = (m & M_OPEN) != 0;
                if (!is_open || l.g[n] > g2) {                                 // :235 idx
                    const float f = d.gr * g2 + l.hh[n];                       // :206 for the next selection
                    const uint32_t k = f32_to_ord(f / d.sqrtW);                // :207 IEEE fp32 division
                    l.g[n] = g2;                                               // :238
                    l.key[n] = k;
                    l.meta[n] = (uint8_t)(M_PASS | M_OPEN | ((uint32_t)lane << 4));  // :242 open ; :246-249 parent = s*
                    atomicMin(&l.chunkmin[n >> 6], k);
                }
            }
        }
    }
    wave_sync();
}

// parent of cell n from its direction code (code j means "n is neighbour j of its parent")
__device__ __forceinline__ int parent_of(const MapDims& d, int n, uint32_t code)
{
    int dr, dc;
    neighbour_delta((int)code, dr, dc);
    return n - (dr * d.W + dc);
}

// ---- backtrack (differentiable_astar.py:96-125): mark M_PATH from the goal towards the start ------------------
// The reference walks exactly t_batch steps; once the start is reached the walk re-enters the same cycle
// (parents[start] keeps its initial value goal_idx), so stopping at the start is equivalent as long as at most
// `cap` steps are taken (cap = own step count - 1 matters only when the Tmax budget ran out).
__device__ __forceinline__ void backtrack(const MapDims& d, const MapLds& l, int lane, int start_idx, int goal_idx,
                                          int cap)
{
    if (lane == 0) {
        uint32_t m = l.meta[goal_idx];
        l.meta[goal_idx] = (uint8_t)(m | M_PATH);
        uint32_t code = m >> 4;
        if (code != PARENT_UNSET) {
            int loc = parent_of(d, goal_idx, code);
            for (int k = 0; k < cap; ++k) {
                uint32_t ml = l.meta[loc];
                l.meta[loc] = (uint8_t)(ml | M_PATH);
                if (loc == start_idx) break;
                uint32_t cd = ml >> 4;
                if (cd == PARENT_UNSET) break;  // cannot happen for an opened non-start node
                loc = parent_of(d, loc, cd);
            }
        }
    }
    wave_sync();
}

// ---- write AstarOutput.histories (fp32 0/1) and .paths (int64 0/1) with 16-byte coalesced stores --------------
template <bool kVec4>
__device__ __forceinline__ void store_outputs(const MapDims& d, const MapLds& l, int lane, float* __restrict__ hist,
                                              long long* __restrict__ paths)
{
    if constexpr (kVec4) {
        const int n4 = d.HW >> 2;
        float4* h4 = reinterpret_cast<float4*>(hist);
        for (int q = lane; q < n4; q += 64) {
            uint32_t m = *reinterpret_cast<const uint32_t*>(l.meta + (q << 2));
            float4 v;
            v.x = (m & M_CLOSED) ? 1.0f : 0.0f;
            v.y = (m & (M_CLOSED << 8)) ? 1.0f : 0.0f;
            v.z = (m & (M_CLOSED << 16)) ? 1.0f : 0.0f;
            v.w = (m & (M_CLOSED << 24)) ? 1.0f : 0.0f;
            h4[q] = v;
        }
        const int n2 = d.HW >> 1;
        longlong2* p2 = reinterpret_cast<longlong2*>(paths);
        for (int q = lane; q < n2; q += 64) {
            uint32_t m = *reinterpret_cast<const uint16_t*>(l.meta + (q << 1));
            longlong2 v;
            v.x = (m & M_PATH) ? 1 : 0;
            v.y = (m & (M_PATH << 8)) ? 1 : 0;
            p2[q] = v;
        }
    } else {
        for (int i = lane; i < d.HW; i += 64) {
            uint32_t m = l.meta[i];
            hist[i] = (m & M_CLOSED) ? 1.0f : 0.0f;
            paths[i] = (m & M_PATH) ? 1 : 0;
        }
    }
}

}  // namespace nastar
