// nastar_capi.hip -- HIP kernels + the C ABI declared in include/nastar.h (libnastar_hip.so).
// gfx950 only.  Build: see neural-astar_amd/csrc/Makefile (hipcc --offload-arch=gfx950 -ffp-contract=off).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <time.h>

#include "nastar_host.hip.h"
#include "nastar_search.hip.h"
#include "nastar_search_hybrid.hip.h"
#include "nastar_search_compact.hip.h"
#include "nastar_search_asm.hip.h"
#ifdef NASTAR_DEV
#include "nastar_dev_flags.h"  // A/B switches of the development build (make dev): round-3 / round-2 streams, no dive, compiled step
#endif
#include "nastar_search_asm4.hip.h"  // (reuses the instruction sections of nastar_search_asm3.hip.h; the round-3 LOOP itself is only instantiated by the development build)
#include "nastar_search_unit.hip.h"
#include "nastar_placement.hip.h"
#include "nastar_backward_replay.hip.h"
#include "nastar_backward_replay_asm.hip.h"

namespace nastar {


// ---- forward, compact LDS state (nastar_search_compact.hip.h): the default for every map that fits LDS ------------------
struct FwdCArgs {
    const float* cost;
    const float* start;
    const float* goal;
    const float* passable;
    float* hist;
    long long* paths;
    int* sel_log;
    int* iters;
    int* status;
    uint8_t* packed;
    const int* order;  // optional placement: workgroup i runs map order[i] (a permutation of 0..B-1), nullptr = identity
    int* order_out;    // optional [B + 1]: the maps in REVERSE order of search completion (the placement for the next visit); [B] = counter
    int* summary;      // optional [NASTAR_SUMMARY_WORDS]: summary[c] = 1 when some map of this launch ends with per-map status c != 0 (device or host-mapped)
    const int* order_bad;  // optional: *order_bad != 0 (written by nastar_order_check_kernel earlier on the stream) = `order` is not a permutation, ignore it
    int* done_counter;     // optional device cell (0 on entry, 0 again at the end): the workgroup whose search finishes LAST sets summary[0] = 1
    int* marks_out;        // early-exit launch, optional [B] (NASTAR_FLAG_MARK_COUPLED): 1 = this map reached its goal but is not at a fixed point of the reference's batch loop
    const int* marks;      // lock-step launches, optional [B]: search only the maps marked 1 (the others return at once: their outputs stand)
    const int* t_end;      // lock-step FINAL launch, optional device cell: the budget is *t_end + 1 steps
    uint32_t* bitmap;      // lock-step PROBE launch: [B][bitmap_words], bit t = the goal was selected at step t; no outputs are written
    int bitmap_words;
    int max_iters;
    int B;
    int flags;
    CompactDims d;
};

// which map this workgroup searches: `order[blockIdx.x]`, unless the launch was asked to check `order` (NASTAR_FLAG_CHECK_ORDER) and the
// check kernel, earlier on the same stream, found that it is not a permutation of 0..B-1 -- then the natural order (every map is searched)
__device__ __forceinline__ int placed_map(const int* order, const int* order_bad, int B)
{
    if (order == nullptr || (order_bad != nullptr && *order_bad != 0)) return (int)blockIdx.x;
    return order[blockIdx.x];
}

// order_out: this search's rank by completion time, counted from the end -- the longest searches come first next time.  The counter cell
// order_out[B] wraps at B (atomicInc): B completions bring it back to where it started (0 for a zeroed buffer) and every rank in [0, B) is
// handed out exactly once WHATEVER the cell held on entry -- a buffer that was not zeroed gets a rotated, still complete order.
__device__ __forceinline__ void note_completion(int* order_out, int B, int b)
{
    unsigned pos = atomicInc(reinterpret_cast<unsigned*>(order_out + B), (unsigned)B - 1u);
    if (pos >= (unsigned)B) pos = (unsigned)B - 1u;  // a cell that held garbage >= B: only the first increment sees it, and rank B-1 is the one nobody else gets
    order_out[B - 1 - (int)pos] = b;
}

// completion_counter of nastar_forward_ex: every search counts itself when it ENDS (before its backtrack and output stores); the last one
// publishes summary[0] = 1.  A workgroup that wrote a summary cell makes it visible system-wide BEFORE it counts (release), the last one
// orders its flag store behind the count it observed (acquire): a host that sees summary[0] != 0 in pinned memory sees every cell of the launch.
__device__ __forceinline__ void note_done(int* counter, int* summary, int B, bool wrote_summary)
{
    if (wrote_summary) __threadfence_system();
    const unsigned pos = atomicInc(reinterpret_cast<unsigned*>(counter), (unsigned)B - 1u);
    if (pos == (unsigned)B - 1u && summary) {
        __threadfence_system();
        __hip_atomic_store(summary, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// NASTAR_FLAG_CHECK_ORDER: is `order` a permutation of 0..B-1?  ONE workgroup, a bitmap of B bits in LDS; writes *bad = 0 / 1 (always)
// and, when it is not, summary[NASTAR_SUMMARY_BAD_ORDER] = 1.  The search / replay kernels that follow on the stream read *bad.
__global__ __launch_bounds__(1024) void nastar_order_check_kernel(const int* __restrict__ order, int B, int* __restrict__ bad, int* summary)
{
    extern __shared__ unsigned bitmap[];
    __shared__ int any_bad;
    const int tid = threadIdx.x, nw = (B + 31) >> 5;
    if (tid == 0) any_bad = 0;
    for (int i = tid; i < nw; i += 1024) bitmap[i] = 0u;
    __syncthreads();
    bool mine = false;
    for (int i = tid; i < B; i += 1024) {
        const int v = order[i];
        if ((unsigned)v >= (unsigned)B) mine = true;
        else if (atomicOr(&bitmap[v >> 5], 1u << (v & 31)) & (1u << (v & 31))) mine = true;  // named twice: some other map is never named
    }
    if (mine) any_bad = 1;
    __syncthreads();
    if (tid == 0) {
        *bad = any_bad;
        if (any_bad && summary) summary[NASTAR_SUMMARY_BAD_ORDER] = 1;
    }
}

// LOGH > 0 && LOGW > 0: the map is exactly (1<<LOGH) x (1<<LOGW) (compile-time sizes, immediate ds offsets).
// CPL_T: chunk minima per lane (1 or 4) when known at compile time, 0 = runtime.
// kAsm: the selection/expansion loop is a hand-scheduled instruction stream (nastar_search_asm4 / _asm3 / _asm.hip.h; 16x16, 32x32, 64x64)
template <bool kVec4, int LOGW, int LOGH, int CPL_T, bool kFastDiv, bool kLog, bool kAsm = false>
__global__ __launch_bounds__(64) void nastar_forward_compact_kernel(const FwdCArgs a, const float rcp_sqrtW)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = placed_map(a.order, a.order_bad, a.B);
    if ((unsigned)b >= (unsigned)a.B) return;  // not a permutation (and not checked: NASTAR_FLAG_CHECK_ORDER): never read or write outside the batch
    const bool lockstep = !kAsm && (a.flags & NASTAR_FLAG_LOCKSTEP);  // (forward_impl picks a compiled instantiation for it)
    if (lockstep && a.marks != nullptr && a.marks[b] == 0) return;    // not in the batch-coupled class: the early-exit launch's outputs stand
    const int lane = threadIdx.x;
    CompactDims d = a.d;
    if constexpr (LOGH > 0 && LOGW > 0) {
        d.H = 1 << LOGH;
        d.W = 1 << LOGW;
        d.HW = 1 << (LOGH + LOGW);
        d.nchunks = d.HW >> CCL;
        d.HWp = d.HW;
        d.CPL = (d.nchunks + 63) / 64;
        d.NCp = d.CPL * 64;
        d.magicW = (uint32_t)((1ull << 32) >> LOGW) + 1u;
    }
    const CompactLds l = carve_compact_lds(smem, d);
    const size_t off = (size_t)b * (size_t)d.HW;

    int start_idx, goal_idx;
    constexpr int kLoadIter = (kVec4 && LOGH > 0 && LOGW > 0 && LOGH + LOGW >= 8) ? (1 << (LOGH + LOGW - 8)) : 0;
    bool any_signed = true;
    // round-3 instruction stream (raw-bit keys): every cost >= +0 and 0 <= g_ratio <= 1 so that every priority is >= +0
    compact_load_map<kVec4, kLoadIter>(d, l, a.cost + off, a.start + off, a.goal + off, a.passable + off, lane, start_idx, goal_idx,
                                       kAsm ? &any_signed : nullptr);
#ifdef NASTAR_DEV
    const bool raw = kAsm && !any_signed && !(a.flags & NASTAR_FLAG_ASM_V2) && d.gr >= 0.f && d.omg >= 0.f;
#else
    const bool raw = kAsm && !any_signed && d.gr >= 0.f && d.omg >= 0.f;
#endif
    const int gi = goal_idx < 0 ? 0 : goal_idx;
    const int goal_r = (int)div_magic((uint32_t)gi, d.magicW);
    const int goal_c = gi - goal_r * d.W;

    const CompactLane lc = make_compact_lane(d, lane);
    int status = NASTAR_OK;
    int iters = 0;
    bool solved = false, coupled = false;
    bool goal_hit = false;
    const bool probe = lockstep && a.bitmap != nullptr;
    const int budget = (lockstep && a.t_end != nullptr) ? __builtin_amdgcn_readfirstlane(*a.t_end + 1) : a.max_iters;
    uint32_t bits = 0u;  // probe: goal selections of the current 32 steps
    uint32_t* const bm = probe ? a.bitmap + (size_t)b * (size_t)a.bitmap_words : nullptr;
    if (start_idx < 0 || goal_idx < 0) {
        status = NASTAR_ERR_UNSOLVABLE;  // not a one-hot start/goal map
    } else {
        // round-4 stream (nastar_search_asm4.hip.h) wherever raw-bit keys apply (costs >= +0, 0 <= g_ratio <= 1: every priority is >= +0); a map with a
        // negative / NaN cost takes the round-2 stream with its order-preserving key transform (nastar_search_asm.hip.h).  The development build
        // (make dev) can also select the round-3 stream and switch the 64x64 dive off: csrc/nastar_dev_flags.h, stream-equality tests.
        // half: g_ratio == 0.5 -- the two products of f = g_ratio g + (1 - g_ratio) h are exact and drop out of the key
#ifdef NASTAR_DEV
        const bool asm4 = raw && !(a.flags & NASTAR_FLAG_ASM_V3);
        const bool no_dive = (a.flags & NASTAR_FLAG_NO_DIVE) != 0;
#else
        const bool asm4 = raw;
#endif
        const bool half = asm4 && d.gr == 0.5f && d.omg == 0.5f;
        compact_open_start<kFastDiv>(d, l, lane, start_idx, goal_r, goal_c, rcp_sqrtW, raw, half);
        int s = 0;
        if constexpr (kAsm) {
            static_assert(LOGW > 0 && LOGW == LOGH && (CPL_T == 1 || CPL_T == 4) && kFastDiv, "asm loop: 16x16, 32x32, 64x64");
            int* const log_row = kLog ? a.sel_log + (size_t)b * (size_t)a.max_iters : nullptr;
            // searching wavefronts issue ahead of the ones still loading their map or already storing their result (the launch waits for the
            // longest SEARCH): maze32 159.4 -> 157.7 us, rand32 75.6 -> 75.0 us per 4096 maps, same box, two runs each; 3-4 batches in flight unchanged at 57 M maps/s (profiles/r03/prio_*.json, prio_streams.txt)
            __builtin_amdgcn_s_setprio(3);
            constexpr bool kD = CPL_T == 4;  // only the 64x64 instantiation dives (nastar_search_asm4.hip.h)
#ifdef NASTAR_DEV
            if (asm4 && kD && no_dive) {
                if (half) s = search_loop_asm4<LOGW, kLog, false, true, false>(d.gr, d.omg, d.sqrtW, lane, goal_idx, goal_r, goal_c, a.max_iters, iters, rcp_sqrtW, log_row);
                else s = search_loop_asm4<LOGW, kLog, false, false, false>(d.gr, d.omg, d.sqrtW, lane, goal_idx, goal_r, goal_c, a.max_iters, iters, rcp_sqrtW, log_row);
            } else if (raw && !asm4 && kD && no_dive) {
                s = compact_search_loop_asm3<LOGW, kLog, false>(d, lane, goal_idx, goal_r, goal_c, a.max_iters, iters, rcp_sqrtW, log_row);
            } else if (raw && !asm4) {
                s = compact_search_loop_asm3<LOGW, kLog>(d, lane, goal_idx, goal_r, goal_c, a.max_iters, iters, rcp_sqrtW, log_row);
            } else
#endif
            if (asm4) {
                s = half ? search_loop_asm4<LOGW, kLog, kD, true, false>(d.gr, d.omg, d.sqrtW, lane, goal_idx, goal_r, goal_c, a.max_iters, iters, rcp_sqrtW, log_row)
                         : search_loop_asm4<LOGW, kLog, kD, false, false>(d.gr, d.omg, d.sqrtW, lane, goal_idx, goal_r, goal_c, a.max_iters, iters, rcp_sqrtW, log_row);
            } else {
                s = compact_search_loop_asm<LOGW, kLog>(d, lane, goal_idx, goal_r, goal_c, a.max_iters, iters, rcp_sqrtW, log_row);
            }
            __builtin_amdgcn_s_setprio(0);
        } else
        while (iters < budget) {  // :203 for t in range(Tmax)
            uint2 mine;
            s = compact_select<CPL_T>(d, l, lane, mine);
            if (s < 0 || (s == goal_idx && !lockstep)) break;  // single exit test: open list empty (:68 would divide by zero) or goal
            if constexpr (kLog) {
                if (lane == 0) a.sel_log[(size_t)b * (size_t)a.max_iters + iters] = s;
            }
            if (probe) {
                if (s == goal_idx) bits |= 1u << (iters & 31);
                if ((iters & 31) == 31) {
                    if (lane == 0) bm[iters >> 5] = bits;
                    bits = 0u;
                }
            }
            ++iters;
            // lock-step mode: the reference's loop to the letter -- a selected goal is expanded like any cell, stays open, and the map is stepped
            // on (it may wander: NASTAR_SUMMARY_COUPLED) until the caller's step count is reached
            goal_hit |= s == goal_idx;
            compact_expand<LOGW, kFastDiv, CPL_T>(d, l, lc, lane, s, goal_r, goal_c, rcp_sqrtW, mine, s == goal_idx);
        }
        if (probe) {  // the words this map's search did not reach say "no goal selection"; nothing else is written
            if (lane == 0) {
                if (iters & 31) bm[iters >> 5] = bits;
                for (int w = (iters + 31) >> 5; w < a.bitmap_words; ++w) bm[w] = 0u;
            }
            return;
        }
        if (lockstep && goal_hit && lane == 0) l.gc[goal_idx].x = NASTAR_NEG_INF;  // histories holds the goal (:222-223); nothing reads its g any more
        if (kAsm ? (s != -2) : (iters < budget)) {
            if (s < 0) {
                status = NASTAR_ERR_UNSOLVABLE;
            } else {  // :219-220,:251 reached the goal: every later step of the reference is a fixed point
                if constexpr (kLog) {
                    if (lane == 0) a.sel_log[(size_t)b * (size_t)a.max_iters + iters] = s;
                }
                ++iters;
                solved = true;
                if (a.summary != nullptr || a.marks_out != nullptr) {
                    // Is this map now at a FIXED POINT of the reference's batch loop?  The reference keeps stepping a finished map until every
                    // map of the batch selects its goal in the same step (:224 the goal stays open, :251); this kernel stops here.  The two
                    // agree iff the goal's own expansion opens nothing that beats the goal: true for every g_ratio in [0.5, 1) with costs >= 0
                    // (f(n) - f(goal) = (2 g_ratio - 1) c_goal + (1 - g_ratio)(h0(n) + c_n) > 0), not for g_ratio < 0.5 with an expensive goal
                    // cell, g_ratio = 1 with a zero-cost one, or negative costs.  Detected here, reported as summary[NASTAR_SUMMARY_COUPLED].
                    const int nr = goal_r + lc.dr, nc = goal_c + lc.dc;
                    const bool inb = lc.is_nb & ((unsigned)nr < (unsigned)d.H) & ((unsigned)nc < (unsigned)d.W);
                    const int n = inb ? s + lc.off : s;
                    const float2 gg = l.gc[s], gn = l.gc[n];
                    const float g2 = gg.x + gg.y;
                    const uint32_t kn = compact_key<kFastDiv>(d, g2, d.omg * (heuristic0_fast(nr, nc, goal_r, goal_c) + gn.y), rcp_sqrtW);
                    const uint32_t kg = compact_key<kFastDiv>(d, gg.x, d.omg * (heuristic0_fast(goal_r, goal_c, goal_r, goal_c) + gg.y), rcp_sqrtW);
                    const bool beats = inb & (gn.x > g2) & ((kn < kg) | ((kn == kg) & (n < s)));
                    coupled = __ballot(beats) != 0ull;
                    if (coupled && a.summary != nullptr && lane == 0) a.summary[NASTAR_SUMMARY_COUPLED] = 1;
                    wave_order();
                }
                if (lane == 0) l.gc[s].x = NASTAR_NEG_INF;  // :222-223 the goal joins the closed list
            }
        }
    }
    wave_sync();
    // histories depend on the closed list only: their stores are issued first and drain under the serial backtrack
    compact_store_hist<kVec4>(d, l, lane, a.hist + off);
    if (probe) return;  // (a map without a one-hot start / goal: the early-exit launch reported it, it is never marked)
    if (lane == 0) {
        a.iters[b] = iters;
        a.status[b] = status;
        if (a.marks_out != nullptr) a.marks_out[b] = coupled ? 1 : 0;
        if (status != NASTAR_OK && a.summary) a.summary[status] = 1;  // plain idempotent store: the word may be host-mapped (no atomics over PCIe)
        if (a.order_out) note_completion(a.order_out, a.B, b);
        // (the COUPLED note above is a summary cell like any other: it must be visible before this search counts itself -- ADVICE r5)
        if (a.done_counter) note_done(a.done_counter, a.summary, a.B, status != NASTAR_OK || coupled);
    }
    if (goal_idx >= 0) compact_backtrack<(LOGW == LOGH ? LOGW : 0)>(d, l, lane, start_idx, goal_idx, solved ? d.HW : iters - 1);
    compact_store_outputs<kVec4, false>(d, l, lane, a.hist + off, a.paths + off,
                                        a.packed ? a.packed + (size_t)b * (size_t)(d.HW >> 2) : nullptr);
}

// ---- forward, UNIT-COST layout (nastar_search_unit.hip.h; NASTAR_FLAG_UNIT_COST): cost map == obstacle map, every value 0.0 or 1.0 ----
// 5.5 B/cell: 29 maps of 32x32 per CU instead of 16 -- with several batches in flight throughput follows the resident maps per CU.
template <int LOGW, bool kDive>
__global__ __launch_bounds__(64) void nastar_forward_unit_kernel(const FwdCArgs a, const float rcp_sqrtW)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int W = 1 << LOGW, HW = W * W;
    const int b = placed_map(a.order, a.order_bad, a.B);
    if ((unsigned)b >= (unsigned)a.B) return;  // not a permutation (and not checked: NASTAR_FLAG_CHECK_ORDER): never read or write outside the batch
    const int lane = threadIdx.x;
    const CompactDims& d = a.d;
    const UnitLds l = carve_unit_lds<LOGW>(smem);
    const size_t off = (size_t)b * (size_t)HW;
    int start_idx, goal_idx;
    bool bad;
    unit_load_map<LOGW>(l, a.cost + off, a.start + off, a.goal + off, lane, start_idx, goal_idx, bad);
    const int gi = goal_idx < 0 ? 0 : goal_idx;
    const int goal_r = gi >> LOGW, goal_c = gi & (W - 1);
    int status = NASTAR_OK;
    int iters = 0;
    bool solved = false;
    if (bad) {
        status = NASTAR_ERR_NOT_UNIT_COST;  // the caller's promise does not hold for this map: empty outputs, never a wrong search
    } else if (start_idx < 0 || goal_idx < 0) {
        status = NASTAR_ERR_UNSOLVABLE;  // not a one-hot start/goal map
    } else {
        const bool half = d.gr == 0.5f && d.omg == 0.5f;
        if (half) unit_open_start<LOGW, true>(d, l, lane, start_idx, goal_r, goal_c, rcp_sqrtW);
        else unit_open_start<LOGW, false>(d, l, lane, start_idx, goal_r, goal_c, rcp_sqrtW);
        __builtin_amdgcn_s_setprio(3);
        int s;
        if (half) s = search_loop_asm4<LOGW, false, kDive, true, true>(d.gr, d.omg, d.sqrtW, lane, goal_idx, goal_r, goal_c, a.max_iters, iters, rcp_sqrtW, nullptr);
        else s = search_loop_asm4<LOGW, false, kDive, false, true>(d.gr, d.omg, d.sqrtW, lane, goal_idx, goal_r, goal_c, a.max_iters, iters, rcp_sqrtW, nullptr);
        __builtin_amdgcn_s_setprio(0);
        if (s != -2) {
            if (s < 0) {
                status = NASTAR_ERR_UNSOLVABLE;
            } else {  // :219-220,:251 reached the goal: every later step of the reference is a fixed point
                ++iters;
                solved = true;
                if (lane == 0) l.g[s] = NASTAR_NEG_INF;  // :222-223 the goal joins the closed list
            }
        }
    }
    wave_sync();
    unit_store_hist<LOGW>(l, lane, a.hist + off, bad);
    if (lane == 0) {
        a.iters[b] = iters;
        a.status[b] = status;
        if (status != NASTAR_OK && a.summary) a.summary[status] = 1;  // plain idempotent store: the word may be host-mapped (no atomics over PCIe)
        if (a.order_out) note_completion(a.order_out, a.B, b);
        if (a.done_counter) note_done(a.done_counter, a.summary, a.B, status != NASTAR_OK);
    }
    if (goal_idx >= 0 && !bad) {
        CompactLds cl;  // the backtrack reads and marks parents only
        cl.gc = nullptr; cl.cmin = l.cmin; cl.dump = nullptr; cl.pdir = l.pdir;
        compact_backtrack<LOGW>(d, cl, lane, start_idx, goal_idx, solved ? HW : iters - 1);
    }
    unit_store_paths<LOGW>(l, lane, a.paths + off, a.packed ? a.packed + (size_t)b * (size_t)(HW >> 2) : nullptr, bad);
}



// ---- get_heuristic standalone (parity/debug) ------------------------------------------------------------
__global__ __launch_bounds__(64) void nastar_heuristic_kernel(const float* goal, float* out, int H, int W, uint32_t magicW)
{
    const int b = blockIdx.x, lane = threadIdx.x, HW = H * W;
    const float* gm = goal + (size_t)b * HW;
    int gidx = -1;
    for (int i = lane; i < HW; i += 64)
        if (gm[i] != 0.f) gidx = i;
    gidx = wave_max_i32(gidx);
    if (gidx < 0) gidx = 0;
    const int gr = (int)div_magic((uint32_t)gidx, magicW), gc = gidx - gr * W;
    for (int i = lane; i < HW; i += 64) {
        int r = (int)div_magic((uint32_t)i, magicW), c = i - r * W;
        out[(size_t)b * HW + i] = heuristic0(r, c, gr, gc);
    }
}

// ---- AstarOutput <-> bit-packed masks (what the multi-GPU all-gather moves: 2 bits per cell instead of 12 bytes) ----
// packed row layout: [ceil(HW/8) bytes of histories bits | ceil(HW/8) bytes of path bits], MSB = first cell (numpy.packbits)
__global__ __launch_bounds__(256) void nastar_pack_kernel(const float* __restrict__ hist, const long long* __restrict__ paths,
                                                          uint8_t* __restrict__ packed, int B, int HW, int nb)
{
    const long long total = (long long)B * nb;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(t / nb), j = (int)(t - (long long)b * nb);
        const size_t base = (size_t)b * HW + (size_t)j * 8;
        uint32_t hb = 0, pb = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (j * 8 + k < HW) {
                hb |= (hist[base + k] != 0.f ? 1u : 0u) << (7 - k);
                pb |= (paths[base + k] != 0 ? 1u : 0u) << (7 - k);
            }
        }
        packed[(size_t)b * 2 * nb + j] = (uint8_t)hb;
        packed[(size_t)b * 2 * nb + nb + j] = (uint8_t)pb;
    }
}

__global__ __launch_bounds__(256) void nastar_unpack_kernel(const uint8_t* __restrict__ packed, float* __restrict__ hist,
                                                            long long* __restrict__ paths, int B, int HW, int nb)
{
    const long long total = (long long)B * nb;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(t / nb), j = (int)(t - (long long)b * nb);
        const uint32_t hb = packed[(size_t)b * 2 * nb + j], pb = packed[(size_t)b * 2 * nb + nb + j];
        const size_t base = (size_t)b * HW + (size_t)j * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (j * 8 + k < HW) {
                hist[base + k] = ((hb >> (7 - k)) & 1u) ? 1.0f : 0.0f;
                paths[base + k] = (pb >> (7 - k)) & 1u;
            }
        }
    }
}


// ---- the reference's stopping step for a batch with maps in the batch-coupled class (differentiable_astar.py:251-252) -------------------
// t_end = the first step at which EVERY map selects its goal: a map that is not marked selects it at every step from its own goal step on
// (fixed point), a marked one at the steps its PROBE bitmap names; without such a step the budget ends the loop (t_end = max_iters - 1).
// tcell = {t_end (-1: no map is marked -- nothing to re-run), number of marked maps, t_max = max over the solved maps of their goal step}.
// ONE workgroup: the bitmaps of the marked maps are AND-ed word by word in LDS (words from t_max on).
constexpr int kTendThreads = 1024;
__global__ __launch_bounds__(kTendThreads) void nastar_batchloop_tend_kernel(const int* __restrict__ iters, const int* __restrict__ status,
                                                                            const int* __restrict__ marks, const uint32_t* __restrict__ bitmap,
                                                                            int words, int B, int max_iters, int* __restrict__ tcell, int lds_words)
{
    extern __shared__ uint32_t s_and[];
    __shared__ int s_tmax, s_nm;
    __shared__ unsigned s_first;
    const int tid = threadIdx.x;
    if (tid == 0) { s_tmax = -1; s_nm = 0; s_first = 0xFFFFFFFFu; }
    __syncthreads();
    int tmax = -1, nm = 0;
    for (int b = tid; b < B; b += kTendThreads) {
        if (status[b] != NASTAR_OK) continue;  // (the reference crashes on an unsolvable map; here it is reported and takes no part)
        tmax = max(tmax, iters[b] - 1);
        nm += marks[b] != 0;
    }
    if (tmax >= 0) atomicMax(&s_tmax, tmax);
    if (nm) atomicAdd(&s_nm, nm);
    __syncthreads();
    tmax = s_tmax;
    nm = s_nm;
    if (nm == 0 || tmax < 0) {
        if (tid == 0) { tcell[0] = -1; tcell[1] = 0; tcell[2] = tmax; }
        return;
    }
    const int w0 = tmax >> 5, nw = words - w0;
    unsigned first = 0xFFFFFFFFu;
    if (nw <= lds_words) {
        for (int w = tid; w < nw; w += kTendThreads) s_and[w] = 0xFFFFFFFFu;
        __syncthreads();
        const long long total = (long long)B * nw;
        for (long long i = tid; i < total; i += kTendThreads) {
            const int b = (int)(i / nw), w = (int)(i - (long long)b * nw);
            if (marks[b] != 0 && status[b] == NASTAR_OK) atomicAnd(&s_and[w], bitmap[(size_t)b * words + w0 + w]);
        }
        __syncthreads();
        for (int w = tid; w < nw; w += kTendThreads) {
            uint32_t v = s_and[w];
            if (w == 0) v &= 0xFFFFFFFFu << (tmax & 31);
            if (v) first = min(first, (unsigned)((w0 + w) * 32 + __builtin_ctz(v)));
        }
    } else {  // (a budget too long for LDS: each thread ANDs whole columns)
        for (int w = tid; w < nw; w += kTendThreads) {
            uint32_t v = 0xFFFFFFFFu;
            for (int b = 0; b < B; ++b)
                if (marks[b] != 0 && status[b] == NASTAR_OK) v &= bitmap[(size_t)b * words + w0 + w];
            if (w == 0) v &= 0xFFFFFFFFu << (tmax & 31);
            if (v) first = min(first, (unsigned)((w0 + w) * 32 + __builtin_ctz(v)));
        }
    }
    if (first != 0xFFFFFFFFu) atomicMin(&s_first, first);
    __syncthreads();
    if (tid == 0) {
        const unsigned f = s_first;
        tcell[0] = (f < (unsigned)(max_iters - 1)) ? (int)f : max_iters - 1;
        tcell[1] = nm;
        tcell[2] = tmax;
    }
}

thread_local char g_last_error[256] = "";

// maps whose state lives in HBM: three levels (cell -> chunk minimum per 64 cells -> super-chunk minimum per 64 chunks), the two minima arrays in
// LDS: 8 B per 64 cells must fit one CU -- 1,179,648 cells (1024 x 1152; 1024 x 1024 takes 130 KiB)
constexpr long long kMaxGlobalCells = 1179648;
// from 6400 cells (80 x 80) on the large-map kernel is the faster one although the compact state would still fit LDS up to ~17 k cells: the
// compiled LDS loop scans HW / 1024 chunk entries per lane and step and keeps 1-2 maps resident per CU; the hybrid step (0.66-0.68 us since its
// next selection travels in scalar registers, nastar_search_hybrid.hip.h) does not grow with the map and keeps 32 maps resident per CU.
// Measured (tools/probe_large.py mid, profiles/r06/probe_mid.jsonl; hybrid / LDS launch time, one map .. 1024 maps): 72x72 0.9-1.33,
// 80x80 0.76-1.01, 96x96 0.76-1.05, 112x112 0.50-0.82, 128x128 0.41-0.58.
constexpr long long kHybridFromCells = 6400;
constexpr size_t kOrderCheckBytes = 16;                // NASTAR_FLAG_CHECK_ORDER: verdict word at the end of the workspace

// maps one launch keeps resident at once: LDS bytes per map against 160 KiB per CU (and 32 wavefront slots), times the CUs of the device
static long long resident_capacity(size_t lds_per_map)
{
    // CU count of the CURRENT device, looked up once per device (a launch on device 3 must not size itself by device 0's answer)
    static int cus_of[64] = {0};
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
        int n = __atomic_load_n(&cus_of[dev], __ATOMIC_RELAXED);
        if (n == 0) {
            if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
            __atomic_store_n(&cus_of[dev], n, __ATOMIC_RELAXED);  // (racing threads store the same value)
        }
        cus = n;
    }
    long long per_cu = (long long)(kMaxLdsBytes / (lds_per_map ? lds_per_map : 1));
    if (per_cu > 32) per_cu = 32;
    if (per_cu < 1) per_cu = 1;
    return per_cu * cus;
}

static int make_cdims(int B, int H, int W, int max_iters, double g_ratio, CompactDims& d)
{
    if (B <= 0 || H <= 0 || W <= 0 || max_iters <= 0) return NASTAR_ERR_BAD_SHAPE;
    if (H > 65535 || W > 65535 || (long long)H * W > 65535 - CCSZ) return NASTAR_ERR_UNSUPPORTED;
    d.H = H;
    d.W = W;
    d.HW = H * W;
    d.nchunks = (d.HW + CCSZ - 1) / CCSZ;
    d.HWp = d.nchunks * CCSZ;
    d.CPL = (d.nchunks + 63) / 64;
    d.NCp = d.CPL * 64;
    d.magicW = (uint32_t)((1ull << 32) / (unsigned)W) + 1u;
    d.gr = (float)g_ratio;
    d.omg = (float)(1.0 - g_ratio);  // python evaluates (1 - g_ratio) in double, ATen casts the scalar to fp32
    d.sqrtW = (float)sqrt((double)W);  // math.sqrt(W) in double, then the fp32 scalar of the division (:207)
    return NASTAR_OK;
}

// maps whose compact state (9 B/cell, nastar_search_compact.hip.h) does not fit the 160 KiB of one CU keep it in HBM
// cells from which a map takes the large-map kernel although its compact state would still fit LDS (the compiled LDS loop scans CPL = HW / 1024
// chunk entries per lane and step; the hybrid kernel's step does not grow with the map).  NASTAR_HYBRID_FROM_CELLS overrides it (probes).
static long long hybrid_from_cells()
{
    static long long v = -1;
    if (v < 0) {
        const char* e = getenv("NASTAR_HYBRID_FROM_CELLS");
        v = (e && *e) ? atoll(e) : kHybridFromCells;
        if (v < 4097) v = 4097;  // (the hand-scheduled 64x64 stream and everything below it stay LDS-resident)
    }
    return v;
}

static bool needs_global_state(int H, int W)
{
    const long long HW = (long long)H * W;
    if (HW >= hybrid_from_cells()) return true;
    if (HW > 65535 - CCSZ) return true;
    const long long nchunks = (HW + CCSZ - 1) / CCSZ;
    return compact_lds_bytes((int)(nchunks * CCSZ), (int)(((nchunks + 63) / 64) * 64)) > kMaxLdsBytes;
}




// Workspace of a forward launch: [hybrid slabs (maps larger than LDS)] [marks: B int32 + the t_end cell, NASTAR_FLAG_MARK_COUPLED]
// [verdict word of NASTAR_FLAG_CHECK_ORDER].  The probe bitmaps of nastar_forward_batchloop_finish follow at a FIXED offset (as if both flags
// had been given), so that the finish call finds the marks wherever the first launch put them.
struct WsLayout {
    size_t slabs, marks_off, tcell_off, chk_off, total;
};
static WsLayout ws_layout(int B, int H, int W, int flags)
{
    WsLayout l{};
    l.slabs = needs_global_state(H, W) ? (size_t)B * hybrid_slab_bytes(H * W) : 0;
    size_t off = l.slabs;
    if (flags & NASTAR_FLAG_MARK_COUPLED) {
        l.marks_off = off;
        off += ((size_t)B * 4 + 15) & ~(size_t)15;
        l.tcell_off = off;
        off += 16;
    }
    if (flags & NASTAR_FLAG_CHECK_ORDER) {
        l.chk_off = off;
        off += kOrderCheckBytes;
    }
    l.total = off;
    return l;
}
static int bitmap_words_for(int max_iters) { return (max_iters + 31) / 32; }

// flag bits this build understands (the A/B switches exist in the development build only: csrc/nastar_dev_flags.h)
#ifdef NASTAR_DEV
constexpr int kKnownFlags = NASTAR_FLAG_UNIT_COST | NASTAR_FLAG_CHECK_ORDER | NASTAR_FLAG_LOCKSTEP | NASTAR_FLAG_MARK_COUPLED | NASTAR_FLAG_NO_ASM |
                            NASTAR_FLAG_ASM_V2 | NASTAR_FLAG_ASM_V3 | NASTAR_FLAG_NO_DIVE;
#else
constexpr int kKnownFlags = NASTAR_FLAG_UNIT_COST | NASTAR_FLAG_CHECK_ORDER | NASTAR_FLAG_LOCKSTEP | NASTAR_FLAG_MARK_COUPLED;
constexpr int NASTAR_FLAG_NO_ASM = 0, NASTAR_FLAG_ASM_V2 = 0, NASTAR_FLAG_ASM_V3 = 0, NASTAR_FLAG_NO_DIVE = 0;  // (host-side tests below fold away)
#endif

// map widths for which the FMA-based division by fl32(sqrt(W)) was verified bit-exact against IEEE division for
// every fp32 f in [2^-100, FLT_MAX] (tools/fastdiv_check.c); widths whose sqrt is a power of two divide exactly.
static bool fastdiv_verified(int W)
{
    static const int ok[] = {2, 8, 32, 128, 512, 10, 12, 20, 24, 28, 40, 45, 48, 50, 60, 96, 100,  // exhaustively checked
                             1, 4, 16, 64, 256, 1024};                                             // sqrt(W) is a power of two
    for (int w : ok)
        if (w == W) return true;
    return false;
}

}  // namespace nastar

using namespace nastar;

extern "C" {

int nastar_version(void) { return NASTAR_VERSION; }

const char* nastar_last_error(void) { return g_last_error; }

size_t nastar_workspace_bytes(int B, int H, int W, int flags)
{
    if (B <= 0 || H <= 0 || W <= 0 || (long long)H * W > kMaxGlobalCells) return 0;
    return ws_layout(B, H, W, flags).total;  // 0: the whole search state lives in LDS and neither marks nor an order check were asked for
}

size_t nastar_batchloop_workspace_bytes(int B, int H, int W, int max_iters)
{
    if (B <= 0 || H <= 0 || W <= 0 || max_iters <= 0 || (long long)H * W > kMaxGlobalCells) return 0;
    return ws_layout(B, H, W, NASTAR_FLAG_MARK_COUPLED | NASTAR_FLAG_CHECK_ORDER).total + (size_t)B * (size_t)bitmap_words_for(max_iters) * 4;
}

// NASTAR_FLAG_CHECK_ORDER: one small launch that decides whether `order` is a permutation of 0..B-1; its verdict word is the LAST
// kOrderCheckBytes of the workspace and is read by the search / replay launch that follows on the same stream
static int check_order(const int32_t* order, int B, void* workspace, size_t workspace_bytes, size_t need, int32_t* summary, hipStream_t s,
                       const int** order_bad)
{
    // `need` = end of the verdict word inside the workspace (the word is the 16 bytes before it)
    *order_bad = nullptr;
    if (!workspace) return NASTAR_ERR_NULL;
    if (need < kOrderCheckBytes || workspace_bytes < need) return NASTAR_ERR_WORKSPACE;
    const size_t lds = (size_t)((B + 31) / 32) * 4;
    if (lds > kMaxLdsBytes - 64) return NASTAR_ERR_UNSUPPORTED;  // > 1.3 M maps in one launch: check the order on the caller's side
    int* bad = reinterpret_cast<int*>(static_cast<unsigned char*>(workspace) + need - kOrderCheckBytes);
    int rc = ensure_lds(nastar_order_check_kernel, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(nastar_order_check_kernel, dim3(1), dim3(1024), lds, s, order, B, bad, summary);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    *order_bad = bad;
    return NASTAR_OK;
}

// order_out for a multi-round launch: maps sorted by their step counts, longest first (nastar_placement.hip.h: counting sort, one
// workgroup, same stream); the trailing counter cell is not used and stays 0
static int rank_order_after(const int32_t* iters, int B, int32_t* order_out, hipStream_t s)
{
    hipLaunchKernelGGL(nastar_rank_levels_kernel, dim3(1), dim3(PLC_RANK_THREADS), 0, s, iters, B, order_out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    return NASTAR_OK;
}

// the lock-step launches of nastar_forward_batchloop_finish hand these through forward_impl
struct LockArgs {
    const int* marks = nullptr;
    const int* t_end = nullptr;
    uint32_t* bitmap = nullptr;
    int bitmap_words = 0;
};

static int forward_impl(const float* cost, const float* start, const float* goal, const float* passable, int B, int H,
                        int W, double g_ratio, int max_iters, float* histories_out, int64_t* paths_out,
                        int32_t* sel_log_out, int32_t* iters_out, int32_t* status_out, void* workspace,
                        size_t workspace_bytes, int flags, void* stream, uint8_t* packed_out, bool* packed_done,
                        const int32_t* order = nullptr, int32_t* order_out = nullptr, int32_t* summary = nullptr, int32_t* done_counter = nullptr,
                        const LockArgs* lock = nullptr)
{
    *packed_done = false;
    if (!cost || !start || !goal || !passable || !histories_out || !paths_out || !iters_out || !status_out)
        return NASTAR_ERR_NULL;
    if (flags & ~kKnownFlags) return NASTAR_ERR_UNSUPPORTED;  // (A/B switches of the development build: make dev, csrc/nastar_dev_flags.h)
    if (B <= 0 || H <= 0 || W <= 0 || max_iters <= 0) return NASTAR_ERR_BAD_SHAPE;
    if ((long long)H * W > kMaxGlobalCells) return NASTAR_ERR_UNSUPPORTED;
    const bool lockstep = (flags & NASTAR_FLAG_LOCKSTEP) != 0;
    const WsLayout wl = ws_layout(B, H, W, flags);
    if (wl.total > 0 && !lock) {  // (the lock-step launches of the finish call were checked there)
        if (!workspace) return NASTAR_ERR_NULL;
        if (workspace_bytes < wl.total) return NASTAR_ERR_WORKSPACE;
    }
    int* marks_out = (!lockstep && (flags & NASTAR_FLAG_MARK_COUPLED)) ? reinterpret_cast<int*>(static_cast<unsigned char*>(workspace) + wl.marks_off) : nullptr;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (needs_global_state(H, W)) {
        // large map: cells in the caller's HBM workspace, open list in LDS (nastar_search_hybrid.hip.h)
        if (!workspace) return NASTAR_ERR_NULL;
        const size_t slab = hybrid_slab_bytes(H * W);
        if (workspace_bytes < (size_t)B * slab) return NASTAR_ERR_WORKSPACE;
        FwdHybridArgs ha;
        ha.cost = cost; ha.start = start; ha.goal = goal; ha.passable = passable;
        ha.hist = histories_out; ha.paths = reinterpret_cast<long long*>(paths_out);
        ha.sel_log = sel_log_out; ha.iters = iters_out; ha.status = status_out; ha.summary = summary;
        ha.workspace = static_cast<unsigned char*>(workspace); ha.slab_bytes = slab; ha.max_iters = max_iters;
        ha.marks_out = marks_out;
        ha.marks = lock ? lock->marks : nullptr; ha.t_end = lock ? lock->t_end : nullptr;
        ha.bitmap = lock ? lock->bitmap : nullptr; ha.bitmap_words = lock ? lock->bitmap_words : 0;
        HybridDims& hd = ha.d;
        hd.H = H; hd.W = W; hd.HW = H * W;
        hd.nchunks = (hd.HW + 63) / 64; hd.nsuper = (hd.nchunks + 63) / 64; hd.spl = (hd.nsuper + 63) / 64;
        hd.gr = (float)g_ratio; hd.omg = (float)(1.0 - g_ratio); hd.sqrtW = (float)sqrt((double)W);
        hd.rcp_sqrtW = 1.0f / hd.sqrtW;
        hd.inv_W = 1.0f / (float)W;
        // headers (start / goal cell per map) to -1: the fill launch raises them with atomicMax
        hipLaunchKernelGGL(nastar_hybrid_header_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, s, ha.workspace, slab,
                           hybrid_header_offset(hd.HW), B);
        const unsigned per_map = (unsigned)((hd.nchunks * 64 + 255) / 256);
        const dim3 grid2(per_map < 64u ? per_map : 64u, (unsigned)B);
        hipLaunchKernelGGL(nastar_hybrid_fill_kernel, grid2, dim3(256), 0, s, ha);
        const bool fd = fastdiv_verified(W);
        const size_t hl = hybrid_lds_bytes(hd.HW);
        if (hl > kMaxLdsBytes) return NASTAR_ERR_UNSUPPORTED;
        int rc2;
        if (lockstep) rc2 = fd ? launch(nastar_forward_hybrid_kernel<true, true>, B, hl, s, ha) : launch(nastar_forward_hybrid_kernel<false, true>, B, hl, s, ha);
        else rc2 = fd ? launch(nastar_forward_hybrid_kernel<true, false>, B, hl, s, ha) : launch(nastar_forward_hybrid_kernel<false, false>, B, hl, s, ha);
        if (rc2) return rc2;
        if (!ha.bitmap) hipLaunchKernelGGL(nastar_hybrid_store_kernel, grid2, dim3(256), 0, s, ha);  // (a probe launch has no outputs)
        hipError_t he = hipGetLastError();
        if (he != hipSuccess) return hip_fail(he, "kernel launch");
        return NASTAR_OK;
    }
    {
        FwdCArgs c;
        int rc = make_cdims(B, H, W, max_iters, g_ratio, c.d);
        if (rc) return rc;
        const size_t lds = compact_lds_bytes(c.d.HWp, c.d.NCp);
        if (lds > kMaxLdsBytes) return NASTAR_ERR_UNSUPPORTED;
        c.cost = cost; c.start = start; c.goal = goal; c.passable = passable;
        c.hist = histories_out; c.paths = reinterpret_cast<long long*>(paths_out);
        c.sel_log = sel_log_out; c.iters = iters_out; c.status = status_out; c.max_iters = max_iters;
        c.packed = nullptr;
        c.order = order;
        c.order_out = order_out;  // (decided below: in-kernel completion order, or a rank of the step counts after the launch)
        c.summary = summary;
        c.done_counter = summary ? done_counter : nullptr;
        c.order_bad = nullptr;
        c.marks_out = marks_out;
        c.marks = lock ? lock->marks : nullptr; c.t_end = lock ? lock->t_end : nullptr;
        c.bitmap = lock ? lock->bitmap : nullptr; c.bitmap_words = lock ? lock->bitmap_words : 0;
        if (order && (flags & NASTAR_FLAG_CHECK_ORDER)) {
            rc = check_order(order, B, workspace, workspace_bytes, wl.chk_off + kOrderCheckBytes, summary, s, &c.order_bad);
            if (rc) return rc;
        }
        c.flags = flags;
        const bool vec4 = (W % 4 == 0) && aligned16(cost) && aligned16(start) && aligned16(goal) && aligned16(passable) &&
                          aligned16(histories_out) && aligned16(paths_out);
        if (packed_out && vec4 && (c.d.HW % 8 == 0)) {  // fused emission of the bit-packed masks
            c.packed = packed_out;
            *packed_done = true;
        }
        const float rcp = 1.0f / c.d.sqrtW;
        const bool fast = fastdiv_verified(W);
        const bool lg = sel_log_out != nullptr;
        void (*kern)(const FwdCArgs, const float) = nullptr;
        c.B = B;
#define NASTAR_CPICK(V4, LW, LH, CPL, FD) \
    kern = lg ? &nastar_forward_compact_kernel<V4, LW, LH, CPL, FD, true> : &nastar_forward_compact_kernel<V4, LW, LH, CPL, FD, false>
        const bool use_asm = !(flags & (NASTAR_FLAG_NO_ASM | NASTAR_FLAG_LOCKSTEP));  // (lock-step mode lives in the compiled step loops)
        // unit-cost layout: the caller promises cost == passable with values in {0, 1} (checked per map by the kernel); taken when the
        // promise can hold at all (ONE tensor), no selection log is wanted and the hand-scheduled stream exists for the size
        if ((flags & NASTAR_FLAG_UNIT_COST) && cost == passable && use_asm && !(flags & (NASTAR_FLAG_ASM_V2 | NASTAR_FLAG_ASM_V3)) && !lg && vec4 && fast &&
            g_ratio >= 0.0 && g_ratio <= 1.0 && H == W && (W == 32 || W == 64)) {
            const size_t ulds = W == 32 ? (size_t)AsmLayoutUnit<5>::BYTES : (size_t)AsmLayoutUnit<6>::BYTES;
            const bool rank_after = order_out && (long long)B > resident_capacity(ulds);
            if (rank_after) c.order_out = nullptr;
            if (marks_out) {  // unit costs are never in the batch-coupled class (f(n) - f(goal) >= 1.001 - 0.001 g_ratio > 0 with every cost 1)
                hipError_t me = hipMemsetAsync(marks_out, 0, (size_t)B * 4, s);
                if (me != hipSuccess) return hip_fail(me, "hipMemsetAsync");
            }
            int urc;
            if (W == 32) urc = launch(&nastar_forward_unit_kernel<5, false>, B, ulds, s, c, rcp);
            else if (flags & NASTAR_FLAG_NO_DIVE) urc = launch(&nastar_forward_unit_kernel<6, false>, B, ulds, s, c, rcp);
            else urc = launch(&nastar_forward_unit_kernel<6, true>, B, ulds, s, c, rcp);
            return (urc == NASTAR_OK && rank_after) ? rank_order_after(iters_out, B, order_out, s) : urc;
        }
        if (use_asm && vec4 && fast && H == 32 && W == 32)
            kern = lg ? &nastar_forward_compact_kernel<true, 5, 5, 1, true, true, true> : &nastar_forward_compact_kernel<true, 5, 5, 1, true, false, true>;
        else if (use_asm && vec4 && fast && H == 16 && W == 16)
            kern = lg ? &nastar_forward_compact_kernel<true, 4, 4, 1, true, true, true> : &nastar_forward_compact_kernel<true, 4, 4, 1, true, false, true>;
        else if (use_asm && vec4 && fast && H == 64 && W == 64)
            kern = lg ? &nastar_forward_compact_kernel<true, 6, 6, 4, true, true, true> : &nastar_forward_compact_kernel<true, 6, 6, 4, true, false, true>;
        else if (vec4 && fast && H == 32 && W == 32) { NASTAR_CPICK(true, 5, 5, 1, true); }
        else if (vec4 && fast && H == 64 && W == 64) { NASTAR_CPICK(true, 6, 6, 4, true); }
        else if (vec4 && fast && H == 16 && W == 16) { NASTAR_CPICK(true, 4, 4, 1, true); }
        else if (vec4 && fast && c.d.CPL == 1) { NASTAR_CPICK(true, 0, 0, 1, true); }
        else if (vec4 && fast) { NASTAR_CPICK(true, 0, 0, 0, true); }
        else if (vec4) { NASTAR_CPICK(true, 0, 0, 0, false); }
        else if (fast) { NASTAR_CPICK(false, 0, 0, 0, true); }
        else { NASTAR_CPICK(false, 0, 0, 0, false); }
#undef NASTAR_CPICK
        // order_out: a launch whose maps are all resident at once ranks them by completion (one atomic per map, in the kernel); with
        // several rounds of workgroups completion time says when a map was STARTED, not how long its search was -- rank the step counts
        const bool rank_after = order_out && (long long)B > resident_capacity(lds);
        if (rank_after) c.order_out = nullptr;
        const int krc = launch(kern, B, lds, s, c, rcp);
        return (krc == NASTAR_OK && rank_after) ? rank_order_after(iters_out, B, order_out, s) : krc;
    }
}

int nastar_forward(const float* cost, const float* start, const float* goal, const float* passable, int B, int H,
                   int W, double g_ratio, int max_iters, float* histories_out, int64_t* paths_out,
                   int32_t* sel_log_out, int32_t* iters_out, int32_t* status_out, void* workspace,
                   size_t workspace_bytes, int flags, void* stream)
{
    bool done;
    return forward_impl(cost, start, goal, passable, B, H, W, g_ratio, max_iters, histories_out, paths_out, sel_log_out,
                        iters_out, status_out, workspace, workspace_bytes, flags, stream, nullptr, &done);
}

int nastar_forward_ordered(const float* cost, const float* start, const float* goal, const float* passable, int B, int H,
                           int W, double g_ratio, int max_iters, float* histories_out, int64_t* paths_out,
                           int32_t* sel_log_out, int32_t* iters_out, int32_t* status_out, uint8_t* packed_out, void* workspace,
                           size_t workspace_bytes, int flags, const int32_t* order, int32_t* order_out, void* stream)
{
    return nastar_forward_ex(cost, start, goal, passable, B, H, W, g_ratio, max_iters, histories_out, paths_out, sel_log_out, iters_out,
                             status_out, packed_out, workspace, workspace_bytes, flags, order, order_out, nullptr, nullptr, stream);
}

int nastar_forward_ex(const float* cost, const float* start, const float* goal, const float* passable, int B, int H, int W, double g_ratio,
                      int max_iters, float* histories_out, int64_t* paths_out, int32_t* sel_log_out, int32_t* iters_out, int32_t* status_out,
                      uint8_t* packed_out, void* workspace, size_t workspace_bytes, int flags, const int32_t* order, int32_t* order_out,
                      int32_t* status_summary, int32_t* completion_counter, void* stream)
{
    if ((order || order_out) && B > 0 && H > 0 && W > 0 && needs_global_state(H, W)) return NASTAR_ERR_UNSUPPORTED;  // LDS-resident searches only
    bool done = false;
    int rc = forward_impl(cost, start, goal, passable, B, H, W, g_ratio, max_iters, histories_out, paths_out, sel_log_out,
                          iters_out, status_out, workspace, workspace_bytes, flags, stream, packed_out, &done, order, order_out, status_summary,
                          completion_counter);
    if (rc != NASTAR_OK || done || !packed_out) return rc;
    return nastar_pack_outputs(histories_out, paths_out, B, H, W, packed_out, stream);
}

int nastar_forward_batchloop_finish(const float* cost, const float* start, const float* goal, const float* passable, int B, int H, int W,
                                    double g_ratio, int max_iters, float* histories_out, int64_t* paths_out, int32_t* sel_log_out,
                                    int32_t* iters_out, int32_t* status_out, void* workspace, size_t workspace_bytes, void* stream)
{
    if (!cost || !start || !goal || !passable || !histories_out || !paths_out || !iters_out || !status_out || !workspace) return NASTAR_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || max_iters <= 0) return NASTAR_ERR_BAD_SHAPE;
    if ((long long)H * W > kMaxGlobalCells) return NASTAR_ERR_UNSUPPORTED;
    if (workspace_bytes < nastar_batchloop_workspace_bytes(B, H, W, max_iters)) return NASTAR_ERR_WORKSPACE;
    const WsLayout wl = ws_layout(B, H, W, NASTAR_FLAG_MARK_COUPLED | NASTAR_FLAG_CHECK_ORDER);
    unsigned char* ws = static_cast<unsigned char*>(workspace);
    LockArgs la;
    la.marks = reinterpret_cast<const int*>(ws + wl.marks_off);
    int* tcell = reinterpret_cast<int*>(ws + wl.tcell_off);
    la.bitmap_words = bitmap_words_for(max_iters);
    uint32_t* bitmap = reinterpret_cast<uint32_t*>(ws + wl.total);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    bool done;
    // 1. PROBE: the marked maps in lock-step mode over the whole budget; which steps select the goal?  (no outputs)
    la.bitmap = bitmap;
    int rc = forward_impl(cost, start, goal, passable, B, H, W, g_ratio, max_iters, histories_out, paths_out, nullptr, iters_out, status_out, workspace,
                          workspace_bytes, NASTAR_FLAG_LOCKSTEP, stream, nullptr, &done, nullptr, nullptr, nullptr, nullptr, &la);
    if (rc) return rc;
    // 2. the first step at which EVERY map of the batch selects its goal
    const int words = la.bitmap_words;
    const int lds_words = words < 16384 ? words : 16384;
    hipLaunchKernelGGL(nastar_batchloop_tend_kernel, dim3(1), dim3(kTendThreads), (size_t)lds_words * 4, s, iters_out, status_out, la.marks, bitmap, words,
                       B, max_iters, tcell, lds_words);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    // 3. FINAL: the marked maps again, for exactly t_end + 1 steps, with outputs (and their rows of the selection log)
    la.bitmap = nullptr;
    la.t_end = tcell;
    return forward_impl(cost, start, goal, passable, B, H, W, g_ratio, max_iters, histories_out, paths_out, sel_log_out, iters_out, status_out, workspace,
                        workspace_bytes, NASTAR_FLAG_LOCKSTEP, stream, nullptr, &done, nullptr, nullptr, nullptr, nullptr, &la);
}

int nastar_completion_supported(int H, int W)
{
    return (H > 0 && W > 0 && (long long)H * W <= kMaxGlobalCells && !needs_global_state(H, W)) ? 1 : 0;
}

int nastar_host_wait_nonzero(const volatile int32_t* word_host, int timeout_us)
{
    if (!word_host) return 0;
    timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (unsigned spin = 0;; ++spin) {
        if (*word_host != 0) return 1;
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#elif defined(__aarch64__)
        asm volatile("yield");
#endif
        if ((spin & 255u) == 255u) {
            timespec t1;
            clock_gettime(CLOCK_MONOTONIC, &t1);
            const long long us = (long long)(t1.tv_sec - t0.tv_sec) * 1000000ll + (t1.tv_nsec - t0.tv_nsec) / 1000;
            if (us >= timeout_us) return *word_host != 0 ? 1 : 0;
        }
    }
}

int nastar_placement_from_levels(const int32_t* levels, int B, int32_t* order_out, void* stream)
{
    if (!levels || !order_out) return NASTAR_ERR_NULL;
    if (B <= 0) return NASTAR_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(nastar_rank_levels_kernel, dim3(1), dim3(PLC_RANK_THREADS), 0, reinterpret_cast<hipStream_t>(stream), levels, B, order_out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    return NASTAR_OK;
}

int nastar_placement_predict(const float* passable, const float* start, const float* goal, int B, int H, int W, int32_t* order_out,
                             void* workspace, size_t workspace_bytes, void* stream)
{
    if (!passable || !start || !goal || !order_out || !workspace) return NASTAR_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0) return NASTAR_ERR_BAD_SHAPE;
    if (H != W || (W != 32 && W != 64) || !aligned16(passable) || !aligned16(start) || !aligned16(goal)) return NASTAR_ERR_UNSUPPORTED;
    if (workspace_bytes < (size_t)B * sizeof(int)) return NASTAR_ERR_WORKSPACE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    int* level = static_cast<int*>(workspace);
    if (W == 32) hipLaunchKernelGGL(nastar_bfs_level_kernel<5>, dim3((unsigned)B), dim3(64), 0, s, passable, start, goal, B, level);
    else hipLaunchKernelGGL(nastar_bfs_level_kernel<6>, dim3((unsigned)B), dim3(64), 0, s, passable, start, goal, B, level);
    hipLaunchKernelGGL(nastar_rank_levels_kernel, dim3(1), dim3(PLC_RANK_THREADS), 0, s, level, B, order_out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    return NASTAR_OK;
}

int nastar_forward_packed(const float* cost, const float* start, const float* goal, const float* passable, int B, int H,
                          int W, double g_ratio, int max_iters, float* histories_out, int64_t* paths_out,
                          int32_t* sel_log_out, int32_t* iters_out, int32_t* status_out, uint8_t* packed_out,
                          void* workspace, size_t workspace_bytes, int flags, void* stream)
{
    if (!packed_out) return NASTAR_ERR_NULL;
    bool done = false;
    int rc = forward_impl(cost, start, goal, passable, B, H, W, g_ratio, max_iters, histories_out, paths_out, sel_log_out,
                          iters_out, status_out, workspace, workspace_bytes, flags, stream, packed_out, &done);
    if (rc != NASTAR_OK || done) return rc;
    return nastar_pack_outputs(histories_out, paths_out, B, H, W, packed_out, stream);  // shapes the fused path skips
}


// ---- backward by replay of the forward's selection log (nastar_backward_replay.hip.h) ------------------------------------
// history entries per map: one per executed step.  An early-exit search executes at most HW + 1 selections whatever the budget; in LOCK-STEP mode
// a goal selection does not close a cell, but between two of them a map of the batch-coupled class closes at least one, and once it closes none it
// selects its goal at every step (the loop then ends as soon as every map does): at most 2 HW steps.  (Square maps: max_iters = W W = HW either way.)
static int bwdr_hist_len(int HW, int max_iters) { return (max_iters < 2 * HW + 2 ? max_iters : 2 * HW + 2) + 2; }
static bool bwdr_fits_lds(int HW) { return bwdr_state_bytes(((HW + 63) / 64) * 64) <= kMaxLdsBytes; }

// 32-bit history stamps (2 B more per cell of the HBM state): maps above 65,519 cells, or a history that can outrun 16 bits
static bool bwdr_wide(int HW, int max_iters) { return HW > 65535 - CCSZ || bwdr_hist_len(HW, max_iters) > 65535; }

size_t nastar_backward_workspace_bytes(int B, int H, int W, int max_iters)
{
    if (B <= 0 || H <= 0 || W <= 0 || max_iters <= 0 || (long long)H * W > kMaxGlobalCells) return 0;
    const int HW = H * W, HWp = ((HW + 63) / 64) * 64;
    size_t n = (size_t)B * (size_t)bwdr_hist_len(HW, max_iters) * 16;
    // (a wide history on an LDS-sized map -- a lock-step log beyond 65535 entries -- takes the HBM state too)
    const bool wide = bwdr_wide(HW, max_iters);
    if (wide || !bwdr_fits_lds(HW)) n += (size_t)B * ((bwdr_state_bytes(HWp, wide) + 255) & ~(size_t)255);
    return n + kOrderCheckBytes;  // + the verdict word of NASTAR_FLAG_CHECK_ORDER (nastar_backward_replay_ordered)
}

static int backward_replay_impl(BwdRArgs& a, const float* cost, const float* start, const float* goal, const float* passable,
                                const int32_t* sel_log, int B, int H, int W, double g_ratio, int max_iters, const int32_t* iters,
                                const int32_t* t_batch_dev, float* grad_cost_out, void* workspace, size_t workspace_bytes, void* stream,
                                int flags = 0)
{
    if (!cost || !start || !goal || !passable || !sel_log || !iters || !grad_cost_out || !workspace) return NASTAR_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || max_iters <= 0) return NASTAR_ERR_BAD_SHAPE;
    if ((long long)H * W > kMaxGlobalCells) return NASTAR_ERR_UNSUPPORTED;
    if ((long long)H * W > 65535 - CCSZ) {
        // above the compact layouts' 16-bit cell indices: the replay needs the geometry and the priority's constants only
        a.d = CompactDims{};
        a.d.H = H; a.d.W = W; a.d.HW = H * W;
        a.d.gr = (float)g_ratio;
        a.d.omg = (float)(1.0 - g_ratio);
        a.d.sqrtW = (float)sqrt((double)W);
    } else {
        int rc = make_cdims(B, H, W, max_iters, g_ratio, a.d);
        if (rc) return rc;
    }
    if (workspace_bytes < nastar_backward_workspace_bytes(B, H, W, max_iters)) return NASTAR_ERR_WORKSPACE;
    a.d.HWp = ((a.d.HW + 63) / 64) * 64;
    a.cost = cost; a.start = start; a.goal = goal; a.passable = passable; a.sel_log = sel_log; a.iters = iters;
    a.B_total = B;
    a.t_batch = t_batch_dev; a.grad_cost = grad_cost_out; a.max_iters = max_iters;
    a.kfac = a.d.omg * (-1.0f / a.d.sqrtW);
    a.hist = static_cast<double*>(workspace);
    // the kernel indexes the history by step: a search executes at most HW + 1 selections whatever the budget
    const int hlen = bwdr_hist_len(a.d.HW, max_iters);
    a.max_iters = max_iters;
    a.state = nullptr;
    a.state_stride = 0;
    a.hist_len = hlen;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const float rcp = 1.0f / a.d.sqrtW;
    const bool fast = fastdiv_verified(W);
    const int max_steps = hlen - 2;
    if ((flags & ~(kKnownFlags)) != 0) return NASTAR_ERR_UNSUPPORTED;
    const bool wide = bwdr_wide(a.d.HW, max_iters);  // (history stamps are 16-bit otherwise)
    // (the hand-scheduled loop closes every selected cell: a lock-step log, whose goal selections leave the goal open, takes the general loop)
    if ((flags & (NASTAR_FLAG_NO_ASM | NASTAR_FLAG_LOCKSTEP)) == 0 && fast && H == W && (W == 32 || W == 16) && aligned16(cost) && aligned16(start) &&
        aligned16(goal) && aligned16(passable) && aligned16(grad_cost_out) && bwdr_asm_lds_bytes(a.d.HW, max_steps) <= kMaxLdsBytes) {
        // hand-scheduled replay loop (nastar_backward_replay_asm.hip.h): the reference's training sizes
        const size_t lds = bwdr_asm_lds_bytes(a.d.HW, max_steps);
        return W == 32 ? launch(nastar_backward_replay_asm_kernel<5>, B, lds, s, a, rcp)
                       : launch(nastar_backward_replay_asm_kernel<4>, B, lds, s, a, rcp);
    }
    if (!wide && bwdr_fits_lds(a.d.HW)) {
        // history in LDS as long as at least 2 maps (or what the state alone allows) stay resident per CU
        const size_t st = bwdr_state_bytes(a.d.HWp), with_hist = st + (size_t)hlen * 16;
        const bool hist_lds = with_hist <= kMaxLdsBytes && (kMaxLdsBytes / with_hist >= 2 || kMaxLdsBytes / st < 2);
        if (hist_lds) return fast ? launch(nastar_backward_replay_kernel<false, true, true>, B, with_hist, s, a, rcp)
                                  : launch(nastar_backward_replay_kernel<false, true, false>, B, with_hist, s, a, rcp);
        return fast ? launch(nastar_backward_replay_kernel<false, false, true>, B, st, s, a, rcp)
                    : launch(nastar_backward_replay_kernel<false, false, false>, B, st, s, a, rcp);
    }
    // state in the HBM workspace: FILL (all CUs: slab, zeroed gradient, start / goal cells into the slab's header), REPLAY (one wavefront per map:
    // O(steps)), SWEEP (all CUs: the cells still open at the end) -- nastar_backward_replay.hip.h
    a.state_stride = (bwdr_state_bytes(a.d.HWp, wide) + 255) & ~(size_t)255;
    a.state = static_cast<unsigned char*>(workspace) + (size_t)B * (size_t)hlen * 16;
    // headers (start / goal cell per map) to -1: the fill launch raises them with atomicMax (the forward's header kernel: same layout of two ints)
    hipLaunchKernelGGL(nastar_hybrid_header_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, s, a.state, a.state_stride,
                       bwdr_header_offset(a.d.HWp, wide), B);
    hipError_t he;
    const unsigned per_map = (unsigned)((a.d.HW + 255) / 256);
    const dim3 grid2(per_map < 64u ? per_map : 64u, (unsigned)B);
    if (wide) hipLaunchKernelGGL(nastar_bwdr_fill_kernel<true>, grid2, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(nastar_bwdr_fill_kernel<false>, grid2, dim3(256), 0, s, a);
    int rc2;
    if (wide) rc2 = fast ? launch(nastar_backward_replay_kernel<true, false, true, true>, B, 64, s, a, rcp)
                         : launch(nastar_backward_replay_kernel<true, false, false, true>, B, 64, s, a, rcp);
    else rc2 = fast ? launch(nastar_backward_replay_kernel<true, false, true>, B, 64, s, a, rcp)
                    : launch(nastar_backward_replay_kernel<true, false, false>, B, 64, s, a, rcp);
    if (rc2 != NASTAR_OK) return rc2;
    if (wide) {
        if (fast) hipLaunchKernelGGL((nastar_bwdr_sweep_kernel<true, true>), grid2, dim3(256), 0, s, a, rcp);
        else hipLaunchKernelGGL((nastar_bwdr_sweep_kernel<true, false>), grid2, dim3(256), 0, s, a, rcp);
    } else {
        if (fast) hipLaunchKernelGGL((nastar_bwdr_sweep_kernel<false, true>), grid2, dim3(256), 0, s, a, rcp);
        else hipLaunchKernelGGL((nastar_bwdr_sweep_kernel<false, false>), grid2, dim3(256), 0, s, a, rcp);
    }
    he = hipGetLastError();
    if (he != hipSuccess) return hip_fail(he, "kernel launch");
    return NASTAR_OK;
}

int nastar_backward_replay(const float* grad_histories, const float* cost, const float* start, const float* goal,
                           const float* passable, const int32_t* sel_log, int B, int H, int W, double g_ratio, int max_iters,
                           const int32_t* iters, const int32_t* t_batch_dev, float* grad_cost_out, void* workspace,
                           size_t workspace_bytes, int flags, void* stream)
{
    if (!grad_histories) return NASTAR_ERR_NULL;
    BwdRArgs a;
    a.grad_hist = grad_histories; a.l1_hist = nullptr; a.l1_traj = nullptr; a.l1_up = nullptr; a.l1_scale = 0.f;
    a.order = nullptr;
    a.order_bad = nullptr;
    return backward_replay_impl(a, cost, start, goal, passable, sel_log, B, H, W, g_ratio, max_iters, iters, t_batch_dev,
                                grad_cost_out, workspace, workspace_bytes, stream, flags);
}

int nastar_backward_l1_replay(const float* histories, const float* opt_trajs, const float* grad_loss_dev, const float* cost,
                              const float* start, const float* goal, const float* passable, const int32_t* sel_log, int B, int H,
                              int W, double g_ratio, int max_iters, const int32_t* iters, const int32_t* t_batch_dev,
                              float* grad_cost_out, void* workspace, size_t workspace_bytes, void* stream)
{
    if (!histories || !opt_trajs) return NASTAR_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0) return NASTAR_ERR_BAD_SHAPE;
    BwdRArgs a;
    a.grad_hist = nullptr; a.l1_hist = histories; a.l1_traj = opt_trajs; a.l1_up = grad_loss_dev;
    a.l1_scale = (float)(1.0 / ((double)B * H * W));
    a.order = nullptr;
    a.order_bad = nullptr;
    return backward_replay_impl(a, cost, start, goal, passable, sel_log, B, H, W, g_ratio, max_iters, iters, t_batch_dev,
                                grad_cost_out, workspace, workspace_bytes, stream);
}

int nastar_backward_replay_ordered(const float* grad_histories, const float* histories, const float* opt_trajs, const float* grad_loss_dev,
                                   const float* cost, const float* start, const float* goal, const float* passable, const int32_t* sel_log,
                                   int B, int H, int W, double g_ratio, int max_iters, const int32_t* iters, const int32_t* t_batch_dev,
                                   float* grad_cost_out, void* workspace, size_t workspace_bytes, int flags, const int32_t* order, void* stream)
{
    if (!grad_histories && (!histories || !opt_trajs)) return NASTAR_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0) return NASTAR_ERR_BAD_SHAPE;
    BwdRArgs a;
    a.grad_hist = grad_histories;
    a.l1_hist = grad_histories ? nullptr : histories; a.l1_traj = grad_histories ? nullptr : opt_trajs;
    a.l1_up = grad_histories ? nullptr : grad_loss_dev;
    a.l1_scale = grad_histories ? 0.f : (float)(1.0 / ((double)B * H * W));
    a.order = order;
    a.order_bad = nullptr;
    if (order && (flags & NASTAR_FLAG_CHECK_ORDER)) {
        // the verdict word sits behind the replay's own workspace (nastar_backward_workspace_bytes already includes it)
        const size_t need = nastar_backward_workspace_bytes(B, H, W, max_iters);
        if (need == 0) return NASTAR_ERR_UNSUPPORTED;
        int rc = check_order(order, B, workspace, workspace_bytes, need, nullptr, reinterpret_cast<hipStream_t>(stream), &a.order_bad);
        if (rc) return rc;
    }
    return backward_replay_impl(a, cost, start, goal, passable, sel_log, B, H, W, g_ratio, max_iters, iters, t_batch_dev,
                                grad_cost_out, workspace, workspace_bytes, stream, flags);
}

int nastar_pack_outputs(const float* histories, const int64_t* paths, int B, int H, int W, uint8_t* packed_out, void* stream)
{
    if (!histories || !paths || !packed_out) return NASTAR_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0) return NASTAR_ERR_BAD_SHAPE;
    const int HW = H * W, nb = (HW + 7) / 8;
    const long long total = (long long)B * nb;
    const unsigned grid = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(nastar_pack_kernel, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), histories,
                       reinterpret_cast<const long long*>(paths), packed_out, B, HW, nb);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    return NASTAR_OK;
}

int nastar_unpack_outputs(const uint8_t* packed, int B, int H, int W, float* histories_out, int64_t* paths_out, void* stream)
{
    if (!histories_out || !paths_out || !packed) return NASTAR_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0) return NASTAR_ERR_BAD_SHAPE;
    const int HW = H * W, nb = (HW + 7) / 8;
    const long long total = (long long)B * nb;
    const unsigned grid = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(nastar_unpack_kernel, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), packed,
                       histories_out, reinterpret_cast<long long*>(paths_out), B, HW, nb);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    return NASTAR_OK;
}

int nastar_debug_occupancy(int H, int W, int* lds_bytes_out)
{
    CompactDims d;
    if (make_cdims(1, H, W, 1, 0.5, d)) return -1;
    const size_t lds = compact_lds_bytes(d.HWp, d.NCp);
    if (lds_bytes_out) *lds_bytes_out = (int)lds;
    if (lds > kMaxLdsBytes) return 0;
    void (*kern)(const FwdCArgs, const float) = &nastar_forward_compact_kernel<true, 0, 0, 0, false, false>;
    if (H == 32 && W == 32) kern = &nastar_forward_compact_kernel<true, 5, 5, 1, true, false>;
    if (H == 64 && W == 64) kern = &nastar_forward_compact_kernel<true, 6, 6, 4, true, false>;
    if (ensure_lds(kern, lds)) return -1;
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, 64, lds) != hipSuccess) return -1;
    return nb;
}

int nastar_heuristic(const float* goal, int B, int H, int W, float* h0_out, void* stream)
{
    if (!goal || !h0_out) return NASTAR_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0) return NASTAR_ERR_BAD_SHAPE;
    if ((long long)H * W > 65535) return NASTAR_ERR_UNSUPPORTED;
    const uint32_t magicW = (uint32_t)((1ull << 32) / (unsigned)W) + 1u;
    hipLaunchKernelGGL(nastar_heuristic_kernel, dim3((unsigned)B), dim3(64), 0, reinterpret_cast<hipStream_t>(stream),
                       goal, h0_out, H, W, magicW);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    return NASTAR_OK;
}

}  // extern "C"
