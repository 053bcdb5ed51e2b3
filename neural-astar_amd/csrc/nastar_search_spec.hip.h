// nastar_search_spec.hip.h -- TWO selections per step where the sequential algorithm allows it ("speculative pair").
//
// The reference selects ONE node per iteration (differentiable_astar.py:203-252), and one launch of the 4096-map batch lasts as long
// as its longest chain of such steps.  On mazes the next selection is, more often than not, simply the second
// best entry of the open list -- not a node the current expansion produces (tools/sim_speculative.py: taking the two best chunk minima
// per step and committing the second only when it is provably the sequential algorithm's next selection shortens the chain 1.55x on
// the bench mazes, 1.1x on random-obstacle maps).
//
// A step here: the two best chunk minima c0 < c1 in (key, cell) order.  Lanes 0-7 / 8-15 own the Moore neighbours of c0 / c1, lanes
// 16-31 / 32-47 the cells of their chunks (c0 and c1 are minima of DIFFERENT chunks), lanes 48 / 49 close them.  c0 is expanded as
// always.  c1 COMMITS in the same step iff it is what the next sequential iteration would select:
//     every entry c0's expansion creates or lowers   AND   every other open cell of c0's chunk   is  >  (key(c1), c1)
// in the 64-bit (key, cell) order (all remaining chunk minima are >= (key(c1), c1) by construction of the pair; "<=" also catches c0
// lowering g[c1] itself, which would change what c1's expansion has to use).  A committed c1 is expanded against the state c0 leaves:
//   * a neighbour n of c1 that is also a neighbour of c0 sees g = min(g_old[n], g2(c0)) (what c0's relaxation leaves, :235-238), c0
//     itself reads as closed;
//   * where both relax the same n, c1's value is the smaller one and is stored by a LATER instruction (LDS executes a wavefront's
//     operations in order), parent code likewise (:246-249);
//   * chunk minima: both chunks are reset, then every relaxed neighbour and every open cell of the two chunks enters through the
//     same ds_min_u64 -- order-free.
// The state after the step is bit-for-bit the state after two sequential iterations; the selection log gets both entries in order.
// Nothing is committed speculatively: a c1 that fails the test is simply selected again by the next step.
#pragma once
#include "nastar_search_compact.hip.h"

namespace nastar {

constexpr uint32_t SPEC_ONES = 0xFFFFFFFFu;

// the raw-bit key of the hand-scheduled streams (nastar_search_asm4.hip.h): q >= +0, its bit pattern orders like q
template <bool kHalf>
__device__ __forceinline__ uint32_t spec_key(float gr, float omg, float sqrtW, float rcp_sqrtW, float G, float h)
{
    const float a = kHalf ? G : gr * G;  // :206 (kHalf: g_ratio == 0.5, both products exact -- the factor drops out of the order)
    const float b = kHalf ? h : omg * h;
    const float f = a + b;
    const float q0 = f * rcp_sqrtW;  // :207 f / sqrt(W), correctly rounded (tools/fastdiv_check.c)
    const float rem = __builtin_fmaf(-q0, sqrtW, f);
    return __float_as_uint(__builtin_fmaf(rem, rcp_sqrtW, q0));
}

// Compiler-generated form (32x32, one chunk minimum per lane).  Same contract as search_loop_asm4: returns the goal index (goal
// selected, its own step not yet counted in iters), -1 (open list empty) or -2 (budget exhausted); `pairs` counts the committed pairs.
template <int LOGW, bool kLog, bool kHalf>
__device__ __forceinline__ int search_loop_spec2(const CompactDims& d, const CompactLds& l, int lane, int goal_idx, int goal_r, int goal_c,
                                                 int max_iters, int& iters, float rcp_sqrtW, int* log_row, int& pairs)
{
    constexpr int W = 1 << LOGW;
    static_assert(W * W / 16 == 64, "one chunk minimum per lane");
    int dr, dc;
    neighbour_delta(lane & 7, dr, dc);
    const bool is_nb = lane < 16, is_chk = lane >= 16 && lane < 48;
    const bool grp1 = is_nb ? lane >= 8 : (is_chk ? lane >= 32 : lane == 49);
    const uint32_t pcode = P_PASS | (uint32_t)(lane & 7);
    const unsigned long long idle = cmin_entry(SPEC_ONES, (uint32_t)goal_idx);
    int it = iters;
    int np = 0;
    while (true) {
        if (it >= max_iters) {
            iters = it;
            pairs = np;
            return -2;
        }
        const unsigned long long e = l.cmin[lane];
        const uint32_t key = (uint32_t)(e >> 32), cell = (uint32_t)e;
        const uint32_t M1 = wave_min_scalar_u32(key);
        const int L1 = __builtin_ctzll(__ballot(key == M1));
        const int c0 = __builtin_amdgcn_readlane((int)cell, L1);
        if (c0 == goal_idx) {  // the goal, or an empty open list (idle entries name the goal under the all-ones key)
            iters = it;
            pairs = np;
            return M1 == SPEC_ONES ? -1 : c0;
        }
        const uint32_t key2 = lane == L1 ? SPEC_ONES : key;
        const uint32_t M2 = wave_min_scalar_u32(key2);
        const int L2 = __builtin_ctzll(__ballot(key2 == M2));
        const int c1 = __builtin_amdgcn_readlane((int)cell, L2);
        const bool spec = M2 != SPEC_ONES && c1 != goal_idx && it + 2 <= max_iters;
        const unsigned long long ent1 = cmin_entry(M2, (uint32_t)c1);

        const int my = grp1 ? c1 : c0;
        const int r = my >> LOGW, c = my & (W - 1);
        const int nr = r + dr, nc = c + dc;
        const bool inb = is_nb && (unsigned)nr < (unsigned)W && (unsigned)nc < (unsigned)W;
        const int il = inb ? (nr << LOGW) + nc : (is_chk ? (my & ~(CCSZ - 1)) + (lane & (CCSZ - 1)) : my);
        const float2 gs = l.gc[my];
        if (lane == 48) {  // :222-225 c0 joins the closed list; its chunk's entry is rebuilt by the atomics below
            l.gc[c0].x = NASTAR_NEG_INF;
            l.cmin[c0 >> CCL] = idle;
        }
        wave_order();  // other lanes read what lane 48 wrote (a compiler-level ordering point: LDS runs a wavefront's operations in order)
        const float2 gl = l.gc[il];
        const int rl = il >> LOGW, cl = il & (W - 1);
        const float h = heuristic0_fast(rl, cl, goal_r, goal_c) + gl.y;  // :191-192
        const float g2 = gs.x + gs.y;                                      // :234
        const float g2_0 = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(g2), 0));
        // what a neighbour of c1 holds once c0 has been expanded
        const int r0 = c0 >> LOGW, q0 = c0 & (W - 1);
        const int ar = rl - r0, ac = cl - q0;
        const int ch = max(ar < 0 ? -ar : ar, ac < 0 ? -ac : ac);
        float gseen = gl.x;
        if (grp1 && ch == 1) gseen = fminf(gseen, g2_0);
        if (grp1 && ch == 0) gseen = NASTAR_NEG_INF;
        const bool upd = inb && (gseen > g2);  // :229,:235
        const uint32_t k = spec_key<kHalf>(d.gr, d.omg, d.sqrtW, rcp_sqrtW, is_nb ? g2 : gl.x, h);
        const unsigned long long child = cmin_entry(k, (uint32_t)il);
        const bool open_l = is_chk && fabsf(gl.x) < NASTAR_POS_INF && il != c1;  // (c0 already reads -inf)
        const bool ins = upd || open_l;
        const bool bad = __ballot(ins && !grp1 && child <= ent1) != 0ull;
        const bool commit1 = spec && !bad;
        if (!grp1 && upd) {
            l.gc[il].x = g2;             // :238
            l.pdir[il] = (uint8_t)pcode;  // :246-249
        }
        wave_order();  // c1's lanes store AFTER c0's: where both relax one cell, c1's smaller value stays
        if (commit1) {
            if (lane == 49) {
                l.gc[c1].x = NASTAR_NEG_INF;
                l.cmin[c1 >> CCL] = idle;
            }
            if (grp1 && upd) {
                l.gc[il].x = g2;
                l.pdir[il] = (uint8_t)pcode;
            }
        }
        wave_order();  // the atomics land on the reset entries
        if (ins && (!grp1 || commit1)) atomicMin(&l.cmin[il >> CCL], child);  // :242
        if constexpr (kLog) {
            if (lane == 0) {
                log_row[it] = c0;
                if (commit1) log_row[it + 1] = c1;
            }
        }
        it += commit1 ? 2 : 1;
        np += commit1 ? 1 : 0;
        wave_order();
    }
}


// ---- LOOKAHEAD: the selection leaves the step's critical path -------------------------------------------------------------------------
// Let E_j be the chunk minima after step t-1, c_t = their minimum (lane L) and R_t = min over j != L of E_j (the runner-up).  Step t
// closes c_t, rebuilds chunk L from its other open cells and lowers some entries to relaxed neighbours' keys, so
//     c_{t+1} = min( R_t,  the <= 8 relaxed neighbours,  the <= 15 other open cells of chunk L )
// exactly: the 64-entry read-back + reduction that selects c_{t+1} today is not needed to KNOW c_{t+1}.  It is still run -- with the
// lane of c_{t+1}'s chunk masked out it yields R_{t+1} -- but its result is only due one step later, i.e. it runs in the shadow of
// step t+1's LDS round trip.  The chain per step shrinks from (expand -> atomics -> read-back -> 64-lane reduction -> lookup) to
// (expand -> 24-lane reduction over values already in registers -> compare with R).  Same selections by construction.
template <int LOGW, bool kLog, bool kHalf>
__device__ __forceinline__ int search_loop_lookahead(const CompactDims& d, const CompactLds& l, int lane, int goal_idx, int goal_r, int goal_c,
                                                     int max_iters, int& iters, float rcp_sqrtW, int* log_row)
{
    constexpr int W = 1 << LOGW;
    static_assert(W * W / 16 == 64, "one chunk minimum per lane");
    int dr, dc;
    neighbour_delta(lane & 7, dr, dc);
    const bool is_nb = lane < 8, is_chk = (lane & 48) == 16;
    const uint32_t pcode = P_PASS | (uint32_t)(lane & 7);
    const unsigned long long idle = cmin_entry(SPEC_ONES, (uint32_t)goal_idx);
    int it = iters;
    // first selection the ordinary way
    unsigned long long e = l.cmin[lane];
    uint32_t ckey = wave_min_scalar_u32((uint32_t)(e >> 32));
    int c = __builtin_amdgcn_readlane((int)(uint32_t)e, __builtin_ctzll(__ballot((uint32_t)(e >> 32) == ckey)));
    while (true) {
        if (it >= max_iters) {
            iters = it;
            return -2;
        }
        if (c == goal_idx) {
            iters = it;
            return ckey == SPEC_ONES ? -1 : c;
        }
        if constexpr (kLog) {
            if (lane == 0) log_row[it] = c;
        }
        ++it;
        // runner-up: minimum of the read-back entries without the lane that owns c's chunk
        const uint32_t kb = lane == (c >> CCL) ? SPEC_ONES : (uint32_t)(e >> 32);
        const uint32_t Rk = wave_min_scalar_u32(kb);
        const int Rc = __builtin_amdgcn_readlane((int)(uint32_t)e, __builtin_ctzll(__ballot(kb == Rk)));
        // expansion of c (as compact_expand / the hand-scheduled streams)
        const int r = c >> LOGW, q = c & (W - 1);
        const int nr = r + dr, nc = q + dc;
        const bool inb = is_nb && (unsigned)nr < (unsigned)W && (unsigned)nc < (unsigned)W;
        const int il = inb ? (nr << LOGW) + nc : (is_chk ? (c & ~(CCSZ - 1)) + (lane & (CCSZ - 1)) : c);
        const float2 gs = l.gc[c];
        wave_order();
        if (lane == 8) {
            l.gc[c].x = NASTAR_NEG_INF;
            l.cmin[c >> CCL] = idle;
        }
        wave_order();
        const float2 gl = l.gc[il];
        const int rl = il >> LOGW, cl = il & (W - 1);
        const float h = heuristic0_fast(rl, cl, goal_r, goal_c) + gl.y;
        const float g2 = gs.x + gs.y;
        const bool upd = inb && (gl.x > g2);
        const uint32_t k = spec_key<kHalf>(d.gr, d.omg, d.sqrtW, rcp_sqrtW, is_nb ? g2 : gl.x, h);
        const bool ins = upd || (is_chk && fabsf(gl.x) < NASTAR_POS_INF);
        if (upd) {
            l.gc[il].x = g2;
            l.pdir[il] = (uint8_t)pcode;
        }
        wave_order();
        if (ins) atomicMin(&l.cmin[il >> CCL], cmin_entry(k, (uint32_t)il));
        wave_order();
        e = l.cmin[lane];  // read back for the NEXT step's runner-up
        // the candidates this step produced
        const uint32_t ck = ins ? k : SPEC_ONES;
        const uint32_t Mc = wave_min_scalar_u32(ck);
        const unsigned long long m = __ballot(ins && ck == Mc);
        const uint32_t ma = (uint32_t)m & 0xFFu, mb = (uint32_t)m & 0xFFFF0000u;  // neighbour lanes / chunk lanes: each in cell order
        const uint32_t ca = ma ? (uint32_t)__builtin_amdgcn_readlane(il, __builtin_ctz(ma)) : SPEC_ONES;
        const uint32_t cb = mb ? (uint32_t)__builtin_amdgcn_readlane(il, __builtin_ctz(mb)) : SPEC_ONES;
        const uint32_t cc = min(ca, cb);
        const bool take = Mc < Rk || (Mc == Rk && cc < (uint32_t)Rc);
        c = take ? (int)cc : Rc;
        ckey = take ? Mc : Rk;
    }
}

}  // namespace nastar
