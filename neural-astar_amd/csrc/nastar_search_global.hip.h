// nastar_search_global.hip.h -- forward search for maps too large for LDS (> ~9.4 k cells, e.g. 100x100 ... 512x512).
//
// Same state machine and the same arrays as nastar_search.hip.h, but the per-map state lives in a caller-provided HBM
// workspace slab (L2-resident in practice) and the open list gets a THIRD level: key[] -> chunkmin[] (64 cells) ->
// supermin[] (64 chunks = 4096 cells), so a step reads 3 x 64 words instead of scanning thousands of chunk minima.
// One wavefront per map.  Every workspace access is an agent-scope relaxed atomic (sc1: served by L2, never by the
// CU's L1, which is not kept coherent with the L2 atomics used for the minima), and a step ends with a full vmcnt
// drain, so lanes hand data to each other exactly as the LDS version does -- only ~10x slower per step.  This is the
// replacement for the reference's advice to fall back to the CPU pq_astar on large maps (astar.py:36-37).
#pragma once
#include "nastar_search.hip.h"

namespace nastar {

struct GlobalSlab {
    uint32_t* key;
    float* g;
    float* cost;
    float* hh;
    uint32_t* chunkmin;  // [NC64]  (nchunks rounded up to a multiple of 64, padded with KEY_INF)
    uint32_t* supermin;  // [64]
    uint8_t* pdir;
};

struct GlobalDims {
    int H, W, HW;
    int nchunks, NC64, nsuper;
    int HWp;
    float gr, omg, sqrtW;
};

__host__ __device__ inline size_t global_slab_bytes(int HW)
{
    const size_t nchunks = ((size_t)HW + 63) / 64;
    const size_t HWp = nchunks * 64;
    const size_t NC64 = ((nchunks + 63) / 64) * 64;
    size_t b = HWp * 17 + NC64 * 4 + 64 * 4;
    return (b + 255) & ~(size_t)255;
}

__device__ __forceinline__ GlobalSlab carve_global_slab(unsigned char* base, const GlobalDims& d)
{
    GlobalSlab l;
    l.key = reinterpret_cast<uint32_t*>(base);
    l.g = reinterpret_cast<float*>(l.key + d.HWp);
    l.cost = l.g + d.HWp;
    l.hh = l.cost + d.HWp;
    l.chunkmin = reinterpret_cast<uint32_t*>(l.hh + d.HWp);
    l.supermin = l.chunkmin + d.NC64;
    l.pdir = reinterpret_cast<uint8_t*>(l.supermin + 64);
    return l;
}

template <typename T>
__device__ __forceinline__ T gld(const T* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename T>
__device__ __forceinline__ void gst(T* p, T v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void gmin(uint32_t* p, uint32_t v)
{
    __hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// all of this wave's workspace stores/atomics have reached L2 before any later load is issued
__device__ __forceinline__ void global_step_fence()
{
    // sc1 stores are write-through: once vmcnt drains they are in L2, where the sc1 loads of the other lanes read them.
    // (An agent-scope release fence would add a buffer_wbl2 of ~1.7 us per call for nothing: nothing here is cached dirty.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

struct FwdGlobalArgs {
    const float* cost;
    const float* start;
    const float* goal;
    const float* passable;
    float* hist;
    long long* paths;
    int* sel_log;
    int* iters;
    int* status;
    int* summary;  // optional [NASTAR_SUMMARY_WORDS]: summary[c] = 1 when some map ends with status c != 0
    unsigned char* workspace;
    size_t slab_bytes;
    int max_iters;
    GlobalDims d;
};

__global__ __launch_bounds__(64) void nastar_forward_global_kernel(const FwdGlobalArgs a)
{
    const int b = blockIdx.x;
    const int lane = threadIdx.x;
    const GlobalDims d = a.d;
    const GlobalSlab l = carve_global_slab(a.workspace + (size_t)b * a.slab_bytes, d);
    const size_t off = (size_t)b * (size_t)d.HW;
    const float* cost = a.cost + off;
    const float* start = a.start + off;
    const float* goal = a.goal + off;
    const float* passable = a.passable + off;

    // ---- load: find start / goal, fill the slab -------------------------------------------------------------------
    int sidx = -1, gidx = -1;
    for (int i = lane; i < d.HW; i += 64) {
        if (start[i] != 0.f) sidx = i;
        if (goal[i] != 0.f) gidx = i;
    }
    sidx = wave_max_i32(sidx);
    gidx = wave_max_i32(gidx);
    const int gi = gidx < 0 ? 0 : gidx;
    const int goal_r = gi / d.W, goal_c = gi - goal_r * d.W;
    for (int i = lane; i < d.HWp; i += 64) {
        const bool valid = i < d.HW;
        const float cv = valid ? cost[i] : 0.f;
        const float pv = valid ? passable[i] : 0.f;
        const int r = i / d.W, c = i - r * d.W;
        gst(&l.cost[i], cv);
        gst(&l.hh[i], d.omg * (heuristic0(r, c, goal_r, goal_c) + cv));   // :191-192, :206
        gst(&l.g[i], pv != 0.f ? NASTAR_POS_INF : NASTAR_NEG_INF);
        gst(&l.key[i], KEY_INF);
        gst(&l.pdir[i], (uint8_t)(PARENT_UNSET | (pv != 0.f ? P_PASS : 0u)));
    }
    for (int c = lane; c < d.NC64; c += 64) gst(&l.chunkmin[c], KEY_INF);
    gst(&l.supermin[lane], KEY_INF);
    global_step_fence();
    if (lane == 0 && sidx >= 0) {  // open list = {start} (:187), g[start] = 0 (:193)
        const uint32_t k0 = f32_to_ord((d.gr * 0.0f + gld(&l.hh[sidx])) / d.sqrtW);
        gst(&l.g[sidx], 0.0f);
        gst(&l.key[sidx], k0);
        gst(&l.chunkmin[sidx >> 6], k0);
        gst(&l.supermin[sidx >> 12], k0);
        gst(&l.pdir[sidx], (uint8_t)(PARENT_UNSET | P_PASS));
    }
    global_step_fence();

    int dr, dc;
    neighbour_delta(lane & 7, dr, dc);
    int status = NASTAR_OK;
    int iters = 0;
    bool solved = false;
    if (sidx < 0 || gidx < 0) {
        status = NASTAR_ERR_UNSOLVABLE;
    } else {
        while (iters < a.max_iters) {  // :203
            // ---- selection: supermin -> chunkmin -> key, first index on every level ------------------------------
            const uint32_t sm = gld(&l.supermin[lane]);
            const uint32_t Mv = wave_min_all_u32(sm);
            if (__ballot(sm != KEY_INF) == 0) {  // open list empty (:68 would divide by zero)
                status = NASTAR_ERR_UNSOLVABLE;
                break;
            }
            const int S = __builtin_ctzll(__ballot(sm == Mv) | (1ull << 63));
            const uint32_t cmv = gld(&l.chunkmin[S * 64 + lane]);
            const int C = S * 64 + __builtin_ctzll(__ballot(cmv == Mv) | (1ull << 63));
            const uint32_t kv = gld(&l.key[C * CHUNK + lane]);
            const int cl = __builtin_ctzll(__ballot(kv == Mv) | (1ull << 63));
            const int s = C * CHUNK + cl;
            if (a.sel_log != nullptr && lane == 0) a.sel_log[(size_t)b * (size_t)a.max_iters + iters] = s;
            ++iters;
            if (s == gidx) {  // :219-220,:251
                if (lane == 0) gst(&l.g[s], NASTAR_NEG_INF);
                solved = true;
                break;
            }
            // ---- close s, relax its neighbours (:222-249) ----------------------------------------------------------
            const int r = s / d.W, c = s - r * d.W;
            const int nr = r + dr, nc = c + dc;
            const bool inb = (lane < 8) & ((unsigned)nr < (unsigned)d.H) & ((unsigned)nc < (unsigned)d.W);
            const int n = inb ? s + dr * d.W + dc : s;
            const float g2 = gld(&l.g[s]) + gld(&l.cost[s]);             // :234
            const float gn = gld(&l.g[n]);
            const float hn = gld(&l.hh[n]);
            const bool upd = inb & (gn > g2);                             // :229,:235
            const float f = d.gr * g2 + hn;                               // :206
            const uint32_t k = f32_to_ord(f / d.sqrtW);                   // :207
            // chunk / super minima without s (exact recomputation of the two levels s belongs to)
            const uint32_t nm = wave_min_all_u32(lane == cl ? KEY_INF : kv);
            const uint32_t nsm = wave_min_all_u32(lane == (C & 63) ? nm : cmv);
            if (lane == 0) {
                gst(&l.key[s], KEY_INF);          // :224
                gst(&l.g[s], NASTAR_NEG_INF);     // :222-223
                gst(&l.chunkmin[C], nm);
                gst(&l.supermin[S], nsm);
            }
            global_step_fence();                  // the plain minima above must land before the atomic mins below
            if (upd) {
                gst(&l.g[n], g2);                                          // :238
                gst(&l.key[n], k);                                         // :242
                gst(&l.pdir[n], (uint8_t)(P_PASS | (uint32_t)lane));       // :246-249
                gmin(&l.chunkmin[n >> 6], k);
                gmin(&l.supermin[n >> 12], k);
            }
            global_step_fence();
        }
    }
    global_step_fence();

    // ---- backtrack (:96-125), see nastar_search.hip.h::backtrack for the equivalence argument ----------------------
    if (gidx >= 0 && lane == 0) {
        const int cap = solved ? d.HW : iters - 1;
        uint32_t m = gld(&l.pdir[gidx]);
        gst(&l.pdir[gidx], (uint8_t)(m | P_PATH));
        uint32_t code = m & P_DIRMASK;
        if (code != PARENT_UNSET) {
            int pdr, pdc;
            neighbour_delta((int)code, pdr, pdc);
            int loc = gidx - (pdr * d.W + pdc);
            for (int k2 = 0; k2 < cap; ++k2) {
                const uint32_t ml = gld(&l.pdir[loc]);
                gst(&l.pdir[loc], (uint8_t)(ml | P_PATH));
                if (loc == sidx) break;
                const uint32_t cd = ml & P_DIRMASK;
                if (cd == PARENT_UNSET) break;
                neighbour_delta((int)cd, pdr, pdc);
                loc -= pdr * d.W + pdc;
            }
        }
    }
    global_step_fence();
    for (int i = lane; i < d.HW; i += 64) {
        const uint32_t m = gld(&l.pdir[i]);
        a.hist[off + i] = ((m & P_PASS) && gld(&l.g[i]) == NASTAR_NEG_INF) ? 1.0f : 0.0f;
        a.paths[off + i] = (m & P_PATH) ? 1 : 0;
    }
    if (lane == 0) {
        a.iters[b] = iters;
        a.status[b] = status;
        if (status != NASTAR_OK && a.summary) a.summary[status] = 1;
    }
}

}  // namespace nastar
