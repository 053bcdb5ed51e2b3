// nastar_backward_replay.hip.h -- backward of DifferentiableAstar.forward (differentiable_astar.py:203-252 under autograd),
// round-2 algorithm: REPLAY the search from the forward's selection log and account the softmax gradient per EVENT.
//
//     dL/dcost = sum_t kfac * y_t * (G - <G, y_t>),   y_t = v / S_t over the open list,  v_i = exp(-q_i),  kfac = (1-g_ratio)(-1/sqrt(W))
//
// The round-1 kernels re-evaluated the whole softmax (two wave sums, one exp per open cell) before EVERY selection: O(open list)
// per step, 3.7x the cost of a forward step.  Between two re-keyings a cell's v_i is constant, so its gradient over an interval
// [t0, t1] of steps on the open list is   kfac * v_i * (G_i * (A(t1) - A(t0)) - (B(t1) - B(t0)))   with the running sums
//     A(t) = sum_{tau <= t} w_tau / S_tau,      B(t) = sum_{tau <= t} w_tau * D_tau / S_tau^2,      D_t = sum_open v * G
// and S, D change by at most 9 cells per step (the closed node and its <= 8 relaxed neighbours): O(9) per step.
//   * S, D live in LDS as doubles and are updated with ds_add_f64 by the <= 9 lanes that change them; a cell's v enters and
//     leaves with the identical fp32 value, and fp64 adds fp32 terms exactly, so S and D carry no drift (the reference sums in fp32).
//   * A, B are doubles in registers (wave-uniform).  Every step appends (A, B) to a per-map history in the HBM workspace; a cell
//     stores the step that opened it (u16 in LDS) and its interval is closed with one 16-byte history load, consumed one step later
//     (software-pipelined: never on the replay's critical path).
//   * contributions go to grad_cost with global_atomic_add_f32 (fire and forget); cells still open at the end are flushed by a sweep.
//   * no selection at all: s_t comes from the forward's sel_log, so a step is one LDS round trip.
// w_tau = 1 except for the reference's batch-coupled fixed-point steps (`extra`, see nastar_backward in include/nastar.h).
// Prototype with the same arithmetic, checked against the reference's autograd: tools/proto_backward_events.py.
#pragma once
#include <type_traits>

#include "nastar_search_compact.hip.h"

namespace nastar {

struct BwdRArgs {
    const float* grad_hist;  // upstream dL/dhistories, or nullptr: L1 loss fused (l1_* below)
    const float* l1_hist;
    const float* l1_traj;
    const float* l1_up;
    float l1_scale;
    const float* cost;
    const float* start;
    const float* goal;
    const float* passable;
    const int* sel_log;   // [B, max_iters] selections of the forward
    const int* iters;     // [B]
    const int* t_batch;   // device scalar or nullptr
    const int* order;     // optional placement (workgroup i replays map order[i]); nullptr = identity
    const int* order_bad; // optional verdict of nastar_order_check_kernel (NASTAR_FLAG_CHECK_ORDER): != 0 = ignore `order`
    int B_total;          // maps in the batch (bounds the placement)
    float* grad_cost;     // [B,H,W], fully written by this kernel
    double* hist;         // workspace: [B][hist_len][2]  (A, B) after each step; entry 0 = (0, 0)
    unsigned char* state; // workspace for maps whose state does not fit LDS (kGlobal), else nullptr
    size_t state_stride;
    int max_iters;
    int hist_len;  // history entries per map = min(max_iters, HW + 1) + 2
    float kfac;
    CompactDims d;
};

// gc 8 + G 4 + t0 2 per cell (wide: t0 4 -- history stamps beyond 65535: the HBM state of maps above 65,519 cells), + (S, D)
__host__ __device__ inline size_t bwdr_state_bytes(int HWp, bool wide = false) { return (size_t)HWp * (wide ? 16 : 14) + 64; }

__device__ __forceinline__ float bwdr_upstream(const BwdRArgs& a, size_t i)
{
    if (a.grad_hist != nullptr) return a.grad_hist[i];
    const float dlt = a.l1_hist[i] - a.l1_traj[i];  // fused L1 (training.py:58): sign(histories - opt_trajs) * grad / numel
    const float sg = dlt > 0.f ? 1.f : (dlt < 0.f ? -1.f : 0.f);
    return sg * (a.l1_scale * (a.l1_up != nullptr ? *a.l1_up : 1.f));
}

// state accessors: LDS, or the HBM workspace through the CU's vector L1.  The slab and the history of a map are touched by ONE wavefront
// between the fill and the sweep launch, and the lanes of a wavefront are coherent through their L1 without further action (write-through,
// accesses processed in order; global_step_fence per step) -- the neighbourhood of s* is mostly that of the previous step, i.e. L1 hits.
// (Rounds 2-5 used agent-scope relaxed atomics = sc1, served by L2: the forward's large-map kernel made the same move in round 5, 1275 -> 986 ns.)
template <bool kGlobal, typename T>
__device__ __forceinline__ T st_ld(const T* p)
{
    return *p;
}
template <bool kGlobal, typename T>
__device__ __forceinline__ void st_st(T* p, T v)
{
    *p = v;
}

template <bool kLds>
__device__ __forceinline__ double hist_ld(const double* p)
{
    return *p;
}
template <bool kLds>
__device__ __forceinline__ void hist_st(double* p, double v)
{
    *p = v;
}

// q = fl(f / fl32(sqrt(W))) of a cell with g-value G and hh = (1-g_ratio)(h0 + cost)   (:206-207); exp(-q) by v_exp_f32
template <bool kFastDiv>
__device__ __forceinline__ float bwdr_v(const CompactDims& d, float G, float hh, float rcp_sqrtW)
{
    const float f = d.gr * G + hh;
    float q;
    if constexpr (kFastDiv) {  // correctly rounded f / sqrt(W) for the verified widths (tools/fastdiv_check.c)
        const float q0 = f * rcp_sqrtW;
        const float rem = __builtin_fmaf(-q0, d.sqrtW, f);
        q = __builtin_fmaf(rem, rcp_sqrtW, q0);
    } else {
        q = f / d.sqrtW;
    }
    return __builtin_amdgcn_exp2f(q * -1.4426950408889634f);
}

// ---- maps whose state lives in the HBM workspace (kGlobal): the per-cell passes are launches of their own --------------------------------
// One wavefront initialising and, at the end, sweeping 262144 cells took ~10 ms per 512x512 map (its loop body is five agent-scope stores);
// as in the forward (nastar_search_hybrid.hip.h: fill / search / store) the two O(cells) passes run on ALL CUs and the one-wavefront-per-map
// launch between them does the O(steps) replay only.  Slab of a map: g | cost | G | t0 (bwdr_state_bytes) and, in its last 64 bytes, a header
// {int start, int goal, double A, double B}: start / goal cells found by the fill launch (atomicMax on -1), the final running sums left by
// the replay for the sweep.
__host__ __device__ inline size_t bwdr_header_offset(int HWp, bool wide) { return bwdr_state_bytes(HWp, wide) - 64; }

template <bool kWide>
__global__ __launch_bounds__(256) void nastar_bwdr_fill_kernel(const BwdRArgs a)
{
    using stamp_t = typename std::conditional<kWide, uint32_t, unsigned short>::type;
    const int b = blockIdx.y;
    const CompactDims& d = a.d;
    unsigned char* base = a.state + (size_t)b * a.state_stride;
    float* g = reinterpret_cast<float*>(base);
    float* cst = g + d.HWp;
    float* G = cst + d.HWp;
    stamp_t* t0 = reinterpret_cast<stamp_t*>(G + d.HWp);
    int* hdr = reinterpret_cast<int*>(base + bwdr_header_offset(d.HWp, kWide));  // {-1, -1, ..} on entry (nastar_hybrid_header_kernel)
    const size_t off = (size_t)b * (size_t)d.HW;
    float* gout = a.grad_cost + off;
    int sidx = -1, gidx = -1;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < d.HW; i += gridDim.x * 256) {
        if (a.start[off + i] != 0.f) sidx = i;
        if (a.goal[off + i] != 0.f) gidx = i;
        g[i] = a.passable[off + i] != 0.f ? NASTAR_POS_INF : NASTAR_NEG_INF;
        cst[i] = a.cost[off + i];
        G[i] = bwdr_upstream(a, off + i);
        t0[i] = (stamp_t)0;
        gout[i] = 0.f;
    }
    if (sidx >= 0) atomicMax(&hdr[0], sidx);
    if (gidx >= 0) atomicMax(&hdr[1], gidx);
}

// cells still on the open list when the replay ended: their intervals close at the final (A, B) the replay left in the header
template <bool kWide, bool kFastDiv>
__global__ __launch_bounds__(256) void nastar_bwdr_sweep_kernel(const BwdRArgs a, const float rcp_sqrtW)
{
    using stamp_t = typename std::conditional<kWide, uint32_t, unsigned short>::type;
    const int b = blockIdx.y;
    const CompactDims& d = a.d;
    const unsigned char* base = a.state + (size_t)b * a.state_stride;
    const float* g = reinterpret_cast<const float*>(base);
    const float* cst = g + d.HWp;
    const float* G = cst + d.HWp;
    const stamp_t* t0 = reinterpret_cast<const stamp_t*>(G + d.HWp);
    const int* hdr = reinterpret_cast<const int*>(base + bwdr_header_offset(d.HWp, kWide));
    const int sidx = hdr[0], gidx = hdr[1];
    if (sidx < 0 || gidx < 0) return;  // not a one-hot start / goal map: no replay ran, the gradient stays zero
    const double A = *reinterpret_cast<const double*>(hdr + 2), B = *reinterpret_cast<const double*>(hdr + 4);
    const double* hist = a.hist + (size_t)b * (size_t)a.hist_len * 2;
    float* gout = a.grad_cost + (size_t)b * (size_t)d.HW;
    const int goal_r = gidx / d.W, goal_c = gidx - goal_r * d.W;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < d.HW; i += gridDim.x * 256) {
        const float gi = g[i];
        if (fabsf(gi) < NASTAR_POS_INF) {
            const int ti = (int)t0[i];
            const double A0 = hist[2 * ti], B0 = hist[2 * ti + 1];
            const int ri = i / d.W, ci = i - ri * d.W;
            const float h0v = kWide ? heuristic0(ri, ci, goal_r, goal_c) : heuristic0_fast(ri, ci, goal_r, goal_c);
            const float v = bwdr_v<kFastDiv>(d, gi, d.omg * (h0v + cst[i]), rcp_sqrtW);
            const float dA = (float)(A - A0), dB = (float)(B - B0);
            gout[i] += (a.kfac * v) * (G[i] * dA - dB);  // (the replay's atomics are over: one thread per cell)
        }
    }
}

// kHistLds: the (A, B) history lives in LDS behind the state (16 B per executed step): no global round trip inside the loop
// (an HBM history costs a write-through store + a vmcnt(0) drain per step, ~1 us).  kFastDiv: see compact_key.
// kWide (with kGlobal): maps above 65,519 cells / histories beyond 65535 entries -- 32-bit history stamps, and the heuristic with the
// correctly rounded square root (coordinates differences reach past 140: nastar_device.hip.h, sqrt_rn_int)
template <bool kGlobal, bool kHistLds, bool kFastDiv, bool kWide = false>
__global__ __launch_bounds__(64) void nastar_backward_replay_kernel(const BwdRArgs a, const float rcp_sqrtW)
{
    static_assert(!kWide || kGlobal, "wide stamps exist for the HBM state only");
    using stamp_t = typename std::conditional<kWide, uint32_t, unsigned short>::type;
    auto h0 = [](int r, int c, int gr, int gc) { return kWide ? heuristic0(r, c, gr, gc) : heuristic0_fast(r, c, gr, gc); };
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = (a.order == nullptr || (a.order_bad != nullptr && *a.order_bad != 0)) ? (int)blockIdx.x : a.order[blockIdx.x];
    if ((unsigned)b >= (unsigned)a.B_total) return;  // not a permutation: never touch memory outside the batch
    const int lane = threadIdx.x;
    const CompactDims d = a.d;
    unsigned char* base = kGlobal ? a.state + (size_t)b * a.state_stride : smem;
    float* g = reinterpret_cast<float*>(base);                 // [HWp] g-value / node state (sign of infinity, as the forward)
    float* cst = g + d.HWp;                                     // [HWp] cost
    float* G = cst + d.HWp;                                     // [HWp] upstream gradient
    stamp_t* t0 = reinterpret_cast<stamp_t*>(G + d.HWp);  // [HWp] history index at which the cell was (re)opened
    double* sd = reinterpret_cast<double*>(smem + (kGlobal ? 0 : (size_t)d.HWp * 14));  // S, D: always in LDS
    const size_t off = (size_t)b * (size_t)d.HW;
    static_assert(!(kGlobal && kHistLds), "a map too large for LDS keeps its history in the workspace as well");
    double* hist = kHistLds ? reinterpret_cast<double*>(smem + (size_t)d.HWp * 14 + 16) : a.hist + (size_t)b * (size_t)a.hist_len * 2;
    float* gout = a.grad_cost + off;

    int sidx = -1, gidx = -1;
    int* const hdr = kGlobal ? reinterpret_cast<int*>(base + bwdr_header_offset(d.HWp, kWide)) : nullptr;
    if constexpr (kGlobal) {  // the fill launch initialised the slab and found the start / goal cells
        sidx = __builtin_amdgcn_readfirstlane(hdr[0]);
        gidx = __builtin_amdgcn_readfirstlane(hdr[1]);
    } else {
        for (int i = lane; i < d.HW; i += 64) {
            if (a.start[off + i] != 0.f) sidx = i;
            if (a.goal[off + i] != 0.f) gidx = i;
            st_st<kGlobal>(&g[i], a.passable[off + i] != 0.f ? NASTAR_POS_INF : NASTAR_NEG_INF);
            st_st<kGlobal>(&cst[i], a.cost[off + i]);
            st_st<kGlobal>(&G[i], bwdr_upstream(a, off + i));
            st_st<kGlobal>(&t0[i], (stamp_t)0);
            gout[i] = 0.f;
        }
        sidx = wave_max_i32(sidx);
        gidx = wave_max_i32(gidx);
    }
    if (lane == 0) {
        sd[0] = 0.0;
        sd[1] = 0.0;
        hist_st<kHistLds>(&hist[0], 0.0);
        hist_st<kHistLds>(&hist[1], 0.0);
    }
    global_step_fence();  // the zeroed gradient and history entry 0 are in L2 before any atomic / load touches them
    wave_sync();
    if (sidx < 0 || gidx < 0) return;

    const int goal_r = gidx / d.W, goal_c = gidx - goal_r * d.W;
    const int n_steps = a.iters[b];
    // The reference keeps stepping a finished map at its fixed point until the slowest map of the batch is done (:251):
    // extra = t_batch - tau such steps; the goal cell is then re-selected while closed and torch.clamp's backward (:223) zeroes
    // its upstream gradient.
    int extra = 0;
    if (a.t_batch != nullptr) extra = *a.t_batch - (n_steps - 1);
    const int* log = a.sel_log + (size_t)b * (size_t)a.max_iters;
    // A log written in LOCK-STEP mode (nastar_forward_batchloop_finish: a map of the batch-coupled class) may select the goal BEFORE its last
    // entry: the goal is then expanded like any cell and stays open (:224), and every RE-selection finds it in histories already -- clamp's
    // backward (:223) zeroes the goal's upstream gradient for that step and all earlier ones.  An ordinary log selects the goal once, last.
    int n_goal = 0, t_last_goal = -1;
    for (int t = lane; t < n_steps; t += 64)
        if (log[t] == gidx) {
            ++n_goal;
            t_last_goal = t;
        }
    n_goal = (int)wave_sum_f32((float)n_goal);  // (exact: < 2^24 selections)
    t_last_goal = wave_max_i32(t_last_goal);
    const bool goal_zeroed = n_goal + (extra > 0 ? extra : 0) >= 2;
    // ... and from the step after its last re-selection on (budget-truncated runs only) the goal's gradient counts again
    const int t_restore = (goal_zeroed && extra <= 0 && t_last_goal < n_steps - 1) ? t_last_goal : -1;
    const float goal_up = bwdr_upstream(a, off + (size_t)gidx);
    if (lane == 0) {
        if (goal_zeroed) st_st<kGlobal>(&G[gidx], 0.f);
        // open list = {start} (:187), g[start] = 0 (:193): the start is open from history index 0
        const int r = sidx / d.W, c = sidx - r * d.W;
        const float hh = d.omg * (h0(r, c, goal_r, goal_c) + st_ld<kGlobal>(&cst[sidx]));
        const float v = bwdr_v<kFastDiv>(d, 0.0f, hh, rcp_sqrtW);
        st_st<kGlobal>(&g[sidx], 0.0f);
        sd[0] = (double)v;
        sd[1] = (double)(st_ld<kGlobal>(&G[sidx]) * v);
    }
    if constexpr (kGlobal) global_step_fence();
    wave_sync();

    int dr, dc;
    neighbour_delta(lane & 7, dr, dc);
    const bool is_nb = lane < 8;
    const int noff = dr * d.W + dc;
    double A = 0.0, B = 0.0;
    // pending interval (closed in the previous step, its history entry still in flight)
    bool pend = false;
    float pv = 0.f, pG = 0.f;
    int pcell = 0;
    double pA0 = 0.0, pB0 = 0.0, pA = 0.0, pB = 0.0;
    int logv = 0;
    bool goal_fixed_point = false;
    for (int t = 0; t < n_steps; ++t) {
        if ((t & 63) == 0) logv = (t + lane < n_steps) ? log[t + lane] : 0;
        const int s = __builtin_amdgcn_readlane(logv, t & 63);
        // HBM history / state: everything issued so far has landed -- the history entry of the previous step, the loads of the
        // pending intervals, the state stores of the previous step.  (LDS executes a wave's operations in order: nothing to do.)
        if constexpr (kGlobal || !kHistLds) global_step_fence();
        if (pend) {
            const float dA = (float)(pA - pA0), dB = (float)(pB - pB0);
            unsafeAtomicAdd(&gout[pcell], (a.kfac * pv) * (pG * dA - dB));
        }
        // softmax of step t over the current open list: A += 1/S, B += D/S^2   (y_t = v/S, <G,y_t> = D/S)
        const double S = sd[0], D = sd[1];
        const float rS = __builtin_amdgcn_rcpf((float)S);
        A += (double)rS;
        B += (double)((float)D * rS * rS);
        const bool goal_step = s == gidx;
        const bool last_step = t == n_steps - 1;
        if (goal_step && last_step && extra <= 0) {  // the step that ends the batch loop: its softmax counted, nothing follows
            pend = false;
            break;
        }
        // expansion of s (:222-249): lanes 0..7 relax the neighbours, lane 8 closes s (the goal stays open, :224)
        const int r = s / d.W, c = s - r * d.W;
        const int nr = r + dr, nc = c + dc;
        const bool inb = is_nb & ((unsigned)nr < (unsigned)d.H) & ((unsigned)nc < (unsigned)d.W);
        const int il = inb ? s + noff : s;
        const float gs = st_ld<kGlobal>(&g[s]), cs = st_ld<kGlobal>(&cst[s]);
        const float gl = st_ld<kGlobal>(&g[il]), cl = st_ld<kGlobal>(&cst[il]);
        const float Gl = st_ld<kGlobal>(&G[il]);
        const int tl = (int)st_ld<kGlobal>(&t0[il]);
        const int rl = il / d.W, cc = il - rl * d.W;
        const float hh = d.omg * (h0(rl, cc, goal_r, goal_c) + cl);
        const float g2 = gs + cs;
        const bool upd = inb & (gl > g2);
        const bool was_open = fabsf(gl) < NASTAR_POS_INF;
        // lane 8 at the goal's last re-selection of a budget-truncated lock-step log: the goal stays open with the same v, but its interval with
        // G = 0 ends here and one with the upstream value begins (D gains v * G_up)
        const bool restore = (lane == 8) & goal_step & (t == t_restore);
        const bool flushing = (upd & was_open) | ((lane == 8) & !goal_step) | restore;
        const float v_old = bwdr_v<kFastDiv>(d, gl, hh, rcp_sqrtW);
        const float v_new = bwdr_v<kFastDiv>(d, g2, hh, rcp_sqrtW);
        if (upd | flushing) {
            const double dS = (upd ? (double)v_new : 0.0) - ((flushing & !restore) ? (double)v_old : 0.0);
            const double dD = (upd ? (double)(Gl * v_new) : 0.0) - (flushing ? (double)(Gl * v_old) : 0.0) + (restore ? (double)(goal_up * v_old) : 0.0);
            // ds_add_f64 issued directly: for a wave-uniform address hipcc's atomic optimizer would first reduce the lanes in a
            // scalar loop (one iteration per active lane); the LDS unit serialises the <= 9 same-address adds much faster
            const uint32_t sda = (uint32_t)(uintptr_t)sd;
            asm volatile("ds_add_f64 %0, %1\n\tds_add_f64 %0, %2 offset:8" ::"v"(sda), "v"(dS), "v"(dD) : "memory");
        }
        if (upd) {
            st_st<kGlobal>(&g[il], g2);
            st_st<kGlobal>(&t0[il], (stamp_t)(t + 1));
        }
        if ((lane == 8) & !goal_step) st_st<kGlobal>(&g[s], NASTAR_NEG_INF);
        if (restore) {
            st_st<kGlobal>(&G[s], goal_up);
            st_st<kGlobal>(&t0[s], (stamp_t)(t + 1));
        }
        if (lane == 0) {  // history entry t+1 = (A, B) after step t
            hist_st<kHistLds>(&hist[2 * (t + 1)], A);
            hist_st<kHistLds>(&hist[2 * (t + 1) + 1], B);
        }
        // close the interval of every cell that left the open list or was re-keyed: load its opening stamp, consume next step
        pend = flushing;
        pv = v_old;
        pG = Gl;
        pcell = il;
        pA = A;
        pB = B;
        if (flushing) {
            pA0 = hist_ld<kHistLds>(&hist[2 * tl]);
            pB0 = hist_ld<kHistLds>(&hist[2 * tl + 1]);
        }
        wave_order();
        if (goal_step && last_step) {  // extra > 0: `extra` more identical steps on the open list left by the goal's own expansion
            goal_fixed_point = true;
            break;
        }
    }
    global_step_fence();
    if (pend) {
        const float dA = (float)(pA - pA0), dB = (float)(pB - pB0);
        unsafeAtomicAdd(&gout[pcell], (a.kfac * pv) * (pG * dA - dB));
    }
    wave_sync();
    if (goal_fixed_point) {
        const double S = sd[0], D = sd[1];
        const float rS = __builtin_amdgcn_rcpf((float)S);
        A += (double)extra * (double)rS;
        B += (double)extra * (double)((float)D * rS * rS);
    }
    if constexpr (kGlobal) {  // the sweep launch closes the intervals of the cells still on the open list
        if (lane == 0) {
            *reinterpret_cast<double*>(hdr + 2) = A;
            *reinterpret_cast<double*>(hdr + 4) = B;
        }
        return;
    }
    // cells still on the open list: close their intervals at the final (A, B)
    for (int i = lane; i < d.HW; i += 64) {
        const float gi = st_ld<kGlobal>(&g[i]);
        if (fabsf(gi) < NASTAR_POS_INF) {
            const int ti = (int)st_ld<kGlobal>(&t0[i]);
            const double A0 = hist_ld<kHistLds>(&hist[2 * ti]);
            const double B0 = hist_ld<kHistLds>(&hist[2 * ti + 1]);
            const int ri = i / d.W, ci = i - ri * d.W;
            const float hh = d.omg * (h0(ri, ci, goal_r, goal_c) + st_ld<kGlobal>(&cst[i]));
            const float v = bwdr_v<kFastDiv>(d, gi, hh, rcp_sqrtW);
            const float dA = (float)(A - A0), dB = (float)(B - B0);
            unsafeAtomicAdd(&gout[i], (a.kfac * v) * (st_ld<kGlobal>(&G[i]) * dA - dB));
        }
    }
}

}  // namespace nastar
