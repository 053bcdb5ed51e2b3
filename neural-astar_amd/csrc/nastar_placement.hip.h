// nastar_placement.hip.h -- a placement for a batch that has never been searched: which maps to start first.
//
// One search launch lasts as long as its longest search, and the hardware starts and arbitrates workgroups in index order: started
// FIRST, a long search is the oldest wavefront of its SIMD and runs at lone-wavefront speed through the crowded phase
// (nastar_forward_ordered, include/nastar.h).  A batch that recurs gets its order from its previous visit.  For a fresh batch a crude
// length predictor is enough -- what matters is that the long searches are early and spread, not their exact rank
// (profiles/r04/order_placements_probe.jsonl: placing 4096 mazes by the predictor below 150 -> 115.5 us per search launch, by their
// exact step counts 114.8; 64x64 random maps 273 -> 229 / 229; correlation with the step counts only 0.77 / 0.47):
//
//   L = level at which a unit-cost breadth-first wave from the start, over passable cells, 8-connected, reaches the goal
//
// i.e. the number of moves of the shortest route (reference: what VanillaAstar's search would return as path length on a binary map,
// differentiable_astar.py:203-252 with cost == 1).  On open maps it degenerates to the Chebyshev distance, in mazes it is the corridor
// length.  The wave is bit-parallel: lane r holds row r of the map as a bit mask, one level = shifts + one DPP move up and down.
//
//   nastar_bfs_level_kernel   one wavefront per map (32x32: lanes 0-31 = rows, 32-bit masks; 64x64: 64 lanes, 64-bit masks)
//   nastar_rank_levels_kernel ONE workgroup: counting sort of the B levels, longest first (ties in arbitrary order) -> order[B]
//
// 17-44 us on 4096 maps -- worth it only hidden under other work (bench.py: fresh_batches_pipelined runs them on a side stream beside the
// previous batch's search).  A batch whose samples carry their optimal distance needs neither: nastar_placement_from_levels ranks
// |opt_dist[start]| with nastar_rank_levels_kernel alone (ops.order_from_levels; what DeviceMazeBatches attaches to every batch).
#pragma once
#include "nastar_device.hip.h"

namespace nastar {

constexpr int PLC_MAX_LEVEL = 4095;   // levels are clamped here (64x64: <= 2047 by construction); unreachable goal = 0
constexpr int PLC_RANK_THREADS = 1024;

// lane i receives the value of lane i-1 / i+1 of the whole wavefront (0 at the ends): DPP wave_shr:1 / wave_shl:1
__device__ __forceinline__ uint32_t wave_from_below(uint32_t v)  // lane i <- lane i - 1
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xF, 0xF, true);
}
__device__ __forceinline__ uint32_t wave_from_above(uint32_t v)  // lane i <- lane i + 1
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xF, 0xF, true);
}

// row masks of one [W x W] fp32 map: coalesced 16-byte loads (lane q of iteration k owns cells 4(64k + q) ..), the four bits of a lane
// are OR-ed into its row's word in LDS.  `rows` must be zeroed.  nonzero = set.
template <int LOGW>
__device__ __forceinline__ void plc_row_masks(const float* __restrict__ src, int lane, uint32_t* rows /* LDS [W * (W/32)] */)
{
    constexpr int W = 1 << LOGW, HW = W * W;
    const float4* s4 = reinterpret_cast<const float4*>(src);
#pragma unroll
    for (int k = 0; k < HW / 256; ++k) {
        const int q = k * 64 + lane;
        const float4 v = s4[q];
        const int cell = q * 4, r = cell >> LOGW, c = cell & (W - 1);
        const uint32_t bits = (v.x != 0.f ? 1u : 0u) | (v.y != 0.f ? 2u : 0u) | (v.z != 0.f ? 4u : 0u) | (v.w != 0.f ? 8u : 0u);
        if (bits) atomicOr(&rows[r * (W / 32) + (c >> 5)], bits << (c & 31));
    }
}

template <int LOGW>
__global__ __launch_bounds__(64) void nastar_bfs_level_kernel(const float* __restrict__ passable, const float* __restrict__ start,
                                                              const float* __restrict__ goal, int B, int* __restrict__ level)
{
    constexpr int W = 1 << LOGW, HW = W * W, WPR = W / 32;  // words per row
    __shared__ uint32_t rows[3][W * WPR];
    const int b = blockIdx.x, lane = threadIdx.x;
    const size_t off = (size_t)b * HW;
    for (int i = lane; i < 3 * W * WPR; i += 64) (&rows[0][0])[i] = 0u;
    __syncthreads();
    plc_row_masks<LOGW>(passable + off, lane, rows[0]);
    plc_row_masks<LOGW>(start + off, lane, rows[1]);
    plc_row_masks<LOGW>(goal + off, lane, rows[2]);
    __syncthreads();
    int L = 0;
    if constexpr (WPR == 1) {
        const bool on = lane < W;
        const uint32_t pass = on ? rows[0][lane] : 0u, gbit = on ? rows[2][lane] : 0u;
        uint32_t vis = on ? rows[1][lane] : 0u;
        if (__ballot((vis & gbit) != 0u) == 0ull) {
            for (int lvl = 1; lvl <= PLC_MAX_LEVEL; ++lvl) {
                const uint32_t h = vis | (vis << 1) | (vis >> 1);
                const uint32_t nv = ((h | wave_from_below(h) | wave_from_above(h)) & pass) | vis;
                if (__ballot((nv & gbit) != 0u) != 0ull) {
                    L = lvl;
                    break;
                }
                if ((lvl & 15) == 0 && __ballot(nv != vis) == 0ull) break;  // the wave has stopped: goal unreachable, L = 0
                vis = nv;
            }
        }
    } else {
        const uint64_t pass = (uint64_t)rows[0][2 * lane] | ((uint64_t)rows[0][2 * lane + 1] << 32);
        const uint64_t gbit = (uint64_t)rows[2][2 * lane] | ((uint64_t)rows[2][2 * lane + 1] << 32);
        uint64_t vis = (uint64_t)rows[1][2 * lane] | ((uint64_t)rows[1][2 * lane + 1] << 32);
        if (__ballot((vis & gbit) != 0ull) == 0ull) {
            for (int lvl = 1; lvl <= PLC_MAX_LEVEL; ++lvl) {
                const uint64_t h = vis | (vis << 1) | (vis >> 1);
                const uint32_t hl = (uint32_t)h, hh = (uint32_t)(h >> 32);
                const uint64_t up = (uint64_t)wave_from_below(hl) | ((uint64_t)wave_from_below(hh) << 32);
                const uint64_t dn = (uint64_t)wave_from_above(hl) | ((uint64_t)wave_from_above(hh) << 32);
                const uint64_t nv = ((h | up | dn) & pass) | vis;
                if (__ballot((nv & gbit) != 0ull) != 0ull) {
                    L = lvl;
                    break;
                }
                if ((lvl & 15) == 0 && __ballot(nv != vis) == 0ull) break;
                vis = nv;
            }
        }
    }
    if (lane == 0) level[b] = L;
}

// counting sort by level, longest first; ONE workgroup.  order[rank] = map.  Maps of one level come in arbitrary order (LDS atomics).
__global__ __launch_bounds__(PLC_RANK_THREADS) void nastar_rank_levels_kernel(const int* __restrict__ level, int B, int* __restrict__ order)
{
    __shared__ int hist[PLC_MAX_LEVEL + 1];
    __shared__ int wsum[PLC_RANK_THREADS / 64];
    const int tid = threadIdx.x;
    for (int i = tid; i <= PLC_MAX_LEVEL; i += PLC_RANK_THREADS) hist[i] = 0;
    __syncthreads();
    // pass 1: position of every map inside its level's bucket
    constexpr int PER = 8;  // maps per thread per sweep; larger batches loop
    for (int base = 0; base < B; base += PLC_RANK_THREADS * PER) {
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int b = base + k * PLC_RANK_THREADS + tid;
            if (b < B) {
                const int l = min(max(level[b], 0), PLC_MAX_LEVEL);
                atomicAdd(&hist[l], 1);
            }
        }
    }
    __syncthreads();
    // exclusive prefix over DESCENDING levels: thread t owns levels [4t, 4t+3] counted from the top
    constexpr int LPT = (PLC_MAX_LEVEL + 1) / PLC_RANK_THREADS;  // 4
    int loc[LPT], tot = 0;
#pragma unroll
    for (int k = 0; k < LPT; ++k) {
        loc[k] = hist[PLC_MAX_LEVEL - (tid * LPT + k)];
        tot += loc[k];
    }
    // inclusive scan of `tot` over the workgroup: wave scan by DPP-free shuffles, then the 16 wave totals
    int inc = tot;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(inc, d);
        if ((tid & 63) >= d) inc += o;
    }
    if ((tid & 63) == 63) wsum[tid >> 6] = inc;
    __syncthreads();
    int wbase = 0;
    for (int w = 0; w < (tid >> 6); ++w) wbase += wsum[w];
    int run = wbase + inc - tot;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < LPT; ++k) {
        hist[PLC_MAX_LEVEL - (tid * LPT + k)] = run;  // first rank of this level
        run += loc[k];
    }
    __syncthreads();
    // pass 2: scatter
    for (int base = 0; base < B; base += PLC_RANK_THREADS * PER) {
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int b = base + k * PLC_RANK_THREADS + tid;
            if (b < B) {
                const int l = min(max(level[b], 0), PLC_MAX_LEVEL);
                order[atomicAdd(&hist[l], 1)] = b;
            }
        }
    }
}

}  // namespace nastar
