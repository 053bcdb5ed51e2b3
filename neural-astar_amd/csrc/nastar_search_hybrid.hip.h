// nastar_search_hybrid.hip.h -- forward search for maps too large for LDS (129x129 ... 1024x1024: as long as the open list's chunk minima fit
// the 160 KiB of one CU, 8 B per 64 cells): the OPEN LIST lives in LDS, the cells in HBM.
//
// The reference's answer to large maps is "use the CPU pq_astar" (astar.py:36-37) because its loop touches every cell of every map
// per step (differentiable_astar.py:203-252).  Here a step touches 64 + 8 + 1 cells.  State of one map (one 64-lane wavefront):
//
//   HBM slab (5 B/cell, caller's workspace; L2-resident in practice)
//     g[]     fp32, the node state in the sign of infinity exactly as in the LDS kernels (nastar_search.hip.h): +inf passable & never
//             opened, -inf closed or obstacle, finite = open.  "relax neighbour n" (:229,:235) is the one comparison g[n] > g2.
//     pdir[]  parent direction | passable | on-path bits (1 B)
//     cost is NOT copied: it is read from the caller's tensor when a cell is touched; h0 is recomputed from the coordinates.
//   LDS (8 B per 64 cells + 512 B: 33 KB at 512x512, 130 KB at 1024x1024)
//     cmin[c] per 64-cell chunk: (key << 32 | cell) of its first minimal open cell, ~0 when it holds none   (u64 order = first-index tie-break)
//     smin[s] per 64 chunks: the minimum of their cmin entries
//
// Three launches per call (fill / search / store, see below).  A step of the search (round 4's kernel kept all three levels in HBM and paid
// SEVEN dependent L2 round trips per step; round 5's read the open list back from LDS twice per step and ran five wave minima in series):
//   select   nothing to do: (key, cell) of s* is in two scalar registers, left there by the previous step
//   load     g / cost of s*, of its 8 neighbours and of the 64 cells of its chunk: issued together, ONE round trip               (HBM)
//   shadow   while that round trip is in flight: the open list WITHOUT s* 's chunk and super-chunk -- restS = minimum of the other 63 chunk
//            entries of its super-chunk, restE = minimum of the other super-chunk entries (two wave minima side by side, LDS reads issued
//            before the loads) -- and both heuristics (the chunk's cells, the neighbours)
//   update   newC = the chunk's minimum without s*, nbr = minimum of the relaxed neighbours (one 64-lane and one 8-lane minimum side by
//            side); g / pdir stores; cmin[C] = newC and smin[S] = min(restS, newC) by plain writes, the neighbours enter both levels by
//            ds_min_u64 -- nothing is read back                                                                                     (LDS)
//   next     s* of the next step = min(restE, restS, newC, nbr): every open cell is in exactly one of the four sets          (scalar)
// All four are first-index minima of (key << 32 | cell): entries ascend with the lane in each set, so "the first lane holding the minimal
// key" (ballot + s_ff1 + v_readlane) replaces a second wave minimum over the cells.
// Keys are never stored: q = fl(f / fl32(sqrt(W))) is re-derived from (g, cost, coordinates), with the IEEE division (no reciprocal).
#pragma once
#include "nastar_search.hip.h"

namespace nastar {

struct HybridDims {
    int H, W, HW;
    int nchunks;   // ceil(HW / 64)
    int nsuper;    // ceil(nchunks / 64)
    int spl;       // super-chunk entries per lane = ceil(nsuper / 64): 1 up to 512x512, 4 at 1024x1024 (lane l owns entries [l spl, (l + 1) spl))
    float gr, omg, sqrtW, rcp_sqrtW;
    float inv_W;   // 1 / W: row of a flat index by one multiply + one correction step (see hybrid_row)
};

// per map: g[HWp] fp32 | pdir[HWp] u8 | (256-byte aligned) header {start cell, goal cell} written by the fill kernel
__host__ __device__ inline size_t hybrid_header_offset(int HW)
{
    const size_t HWp = (((size_t)HW + 63) / 64) * 64;
    return (HWp * 5 + 255) & ~(size_t)255;
}
__host__ __device__ inline size_t hybrid_slab_bytes(int HW) { return hybrid_header_offset(HW) + 256; }
// cmin: one entry per chunk, padded to whole super-chunks; smin: one entry per super-chunk, padded to `spl` entries for each of the 64 lanes
__host__ __device__ inline size_t hybrid_lds_bytes(int HW)
{
    const size_t nchunks = ((size_t)HW + 63) / 64;
    const size_t nsuper = (nchunks + 63) / 64;
    const size_t spl = (nsuper + 63) / 64;
    return nsuper * 64 * 8 + spl * 64 * 8;
}

struct FwdHybridArgs {
    const float* cost;
    const float* start;
    const float* goal;
    const float* passable;
    float* hist;
    long long* paths;
    int* sel_log;
    int* iters;
    int* status;
    int* summary;
    unsigned char* workspace;
    size_t slab_bytes;
    int max_iters;
    int* marks_out;        // early-exit launch, optional [B]: 1 = this map reached its goal but is not at a fixed point of the reference's batch loop
    const int* marks;      // lock-step launches, optional [B]: search only the maps marked 1
    const int* t_end;      // lock-step FINAL launch, optional device cell: the budget is *t_end + 1 steps
    uint32_t* bitmap;      // lock-step PROBE launch: [B][bitmap_words], bit t = the goal was selected at step t
    int bitmap_words;
    HybridDims d;
};

// row of flat index i (< 2^21): (i + 0.5) / W in fp32 lands within one row of the true quotient (the product's error is ~2^-23 of a row
// index below 2^11); one correction step either way makes it exact
__device__ __forceinline__ int hybrid_row(int i, const HybridDims& d, int& c)
{
    int r = (int)(((float)i + 0.5f) * d.inv_W);
    c = i - r * d.W;
    if (c < 0) { --r; c += d.W; }
    else if (c >= d.W) { ++r; c -= d.W; }
    return r;
}

// the same without branches (selects): the if / else-if form above compiles to two divergent regions, i.e. two VALU -> SALU -> VALU hops
__device__ __forceinline__ int hybrid_row_nb(int i, const HybridDims& d, int& c)
{
    const int r0 = (int)(((float)i + 0.5f) * d.inv_W);
    const int c0 = i - r0 * d.W;
    const bool lo = c0 < 0, hi = c0 >= d.W;
    c = lo ? c0 + d.W : (hi ? c0 - d.W : c0);
    return lo ? r0 - 1 : (hi ? r0 + 1 : r0);
}

// first-index minimum of per-lane (key, cell) entries = the u64 minimum of (key << 32 | cell), in every lane and without leaving the vector
// registers: the key minimum, then the cell minimum among the lanes that hold it.  (A scalar form -- v_readlane of the minimum, ballot, s_ff1,
// v_readlane of the cell -- is three instructions shorter and was 28 % SLOWER per step: each VALU -> SALU -> VALU hop costs a lone wavefront
// ~25 cycles, profiles/r05/probe_large_scalar_reductions.jsonl.)
__device__ __forceinline__ unsigned long long first_min_entry(uint32_t key, uint32_t cell)
{
    const uint32_t m = wave_min_all_u32(key);
    const uint32_t c = wave_min_all_u32(key == m ? cell : 0xFFFFFFFFu);
    return m == KEY_INF ? ~0ull : (((unsigned long long)m << 32) | c);
}

template <bool kFastDiv>
__device__ __forceinline__ uint32_t hybrid_key(const HybridDims& d, float g, float h)
{
    const float f = d.gr * g + d.omg * h;   // :206  f = g_ratio * g + (1 - g_ratio) * h
    float q;
    if constexpr (kFastDiv) {
        // correctly rounded f / sqrt(W) (exhaustively verified per W, tools/fastdiv_check.c; the LDS kernels use the same three instructions)
        const float q0 = f * d.rcp_sqrtW;
        const float rem = __builtin_fmaf(-q0, d.sqrtW, f);
        q = __builtin_fmaf(rem, d.rcp_sqrtW, q0);
    } else {
        q = f / d.sqrtW;                    // :207  the quotient the reference's softmax orders by (IEEE division)
    }
    return f32_to_ord(q);
}

// Three launches on the caller's stream: FILL (all CUs: node states from the passable map, start / goal cells into the slab header), SEARCH
// (one wavefront per map: as many maps resident per CU as wave slots allow -- a step is an L2 round trip, residency is what hides it), STORE
// (all CUs: histories / paths from the slab).  One wavefront filling and storing 262144 cells took 5 ms per map.
__global__ __launch_bounds__(256) void nastar_hybrid_fill_kernel(const FwdHybridArgs a)
{
    const int b = blockIdx.y;
    if (a.marks != nullptr && a.marks[b] == 0) return;  // lock-step launches touch only the marked maps
    const HybridDims d = a.d;
    const int HWp = d.nchunks * 64;
    unsigned char* const slab = a.workspace + (size_t)b * a.slab_bytes;
    float* const g = reinterpret_cast<float*>(slab);
    uint8_t* const pdir = reinterpret_cast<uint8_t*>(g + HWp);
    int* const hdr = reinterpret_cast<int*>(slab + hybrid_header_offset(d.HW));  // {-1, -1} on entry (hipMemsetAsync 0xFF)
    const size_t off = (size_t)b * (size_t)d.HW;
    int sidx = -1, gidx = -1;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HWp; i += gridDim.x * 256) {
        const bool valid = i < d.HW;
        if (valid && a.start[off + i] != 0.f) sidx = i;
        if (valid && a.goal[off + i] != 0.f) gidx = i;
        const bool pass = valid && a.passable[off + i] != 0.f;
        g[i] = pass ? NASTAR_POS_INF : NASTAR_NEG_INF;
        pdir[i] = (uint8_t)(PARENT_UNSET | (pass ? P_PASS : 0u));
    }
    if (sidx >= 0) atomicMax(&hdr[0], sidx);  // (the LAST non-zero cell, like the LDS kernels)
    if (gidx >= 0) atomicMax(&hdr[1], gidx);
}

__global__ __launch_bounds__(256) void nastar_hybrid_header_kernel(unsigned char* workspace, size_t slab_bytes, size_t header_off, int B)
{
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b < B) {  // (also for maps a lock-step launch skips: their headers are not read again)
        int* hdr = reinterpret_cast<int*>(workspace + (size_t)b * slab_bytes + header_off);
        hdr[0] = -1;
        hdr[1] = -1;
    }
}

__global__ __launch_bounds__(256) void nastar_hybrid_store_kernel(const FwdHybridArgs a)
{
    const int b = blockIdx.y;
    if (a.marks != nullptr && a.marks[b] == 0) return;
    const HybridDims d = a.d;
    const int HWp = d.nchunks * 64;
    const unsigned char* const slab = a.workspace + (size_t)b * a.slab_bytes;
    const float* const g = reinterpret_cast<const float*>(slab);
    const uint8_t* const pdir = reinterpret_cast<const uint8_t*>(g + HWp);
    const size_t off = (size_t)b * (size_t)d.HW;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < d.HW; i += gridDim.x * 256) {
        const uint32_t m = pdir[i];
        a.hist[off + i] = ((m & P_PASS) && g[i] == NASTAR_NEG_INF) ? 1.0f : 0.0f;  // closed list (:222-223)
        a.paths[off + i] = (m & P_PATH) ? 1 : 0;
    }
}

// How the search reaches its slab -- decided by measurement in round 5 (profiles/r05/probe_large_variants_*.jsonl; the A/B switches lived in
// the library until round 6):
//   * plain accesses through the CU's vector L1 instead of agent-scope (sc1) ones served by L2 (986 instead of 1275 ns per step).  The slab of a
//     map is touched by ONE wavefront between the fill and the store launch, and the lanes of a wavefront are coherent through their L1 without
//     further action (it is write-through and processes a wavefront's accesses in order): the neighbourhood of s* is mostly the neighbourhood
//     of the previous one, i.e. L1 hits.
//   * s* travels in a SCALAR register (v_readlane of the reduction's result): the loop's exits (open list empty, goal selected, budget) are
//     scalar branches and the step counter a scalar -- the compiler otherwise treats the wave-uniform s* as divergent and wraps every exit in
//     exec-mask bookkeeping; rows / columns by selects instead of branches (hybrid_row_nb): -7 %.
//   * the selection's tie-break by ballot + first set lane + v_readlane instead of a second wave minimum (entries ascend with the lane): -1 %.
//   * the wait for the previous step's stores before this step's loads stays (dropping it changed nothing: 938-940 ns).
//
// kLock: the LOCK-STEP modes (the reference's batch loop to the letter, differentiable_astar.py:219-225, :251): a selected goal is expanded
// like any cell and stays on the open list, and the map is stepped on -- PROBE (a.bitmap != nullptr: no outputs; bit t of the map's bitmap
// row = "the goal was selected at step t", over the whole budget) and FINAL (outputs after exactly *a.t_end + 1 steps).  With a.marks only
// the maps the early-exit launch marked as batch-coupled are searched; the others return at once (their outputs stand).
__device__ __forceinline__ float hld(const float* p) { return *p; }
__device__ __forceinline__ uint32_t hld(const uint8_t* p) { return *p; }

// does the expansion of the goal cell s (selected just now) open or lower a neighbour that BEATS the goal?  (wave-uniform; lanes 0..7)
template <bool kFastDiv>
__device__ __forceinline__ bool hybrid_goal_beaten(const HybridDims& d, const float* g, const float* cost, int s, int lane, int dr, int dc,
                                                   int goal_r, int goal_c)
{
    int gc0;
    const int gr0 = hybrid_row_nb(s, d, gc0);
    const int nr = gr0 + dr, nc = gc0 + dc;
    const bool inb = (lane < 8) & ((unsigned)nr < (unsigned)d.H) & ((unsigned)nc < (unsigned)d.W);
    const int n = inb ? s + dr * d.W + dc : s;
    global_step_fence();
    const float gs = g[s], gn = g[n];
    const float cs = cost[s], cn = cost[n];
    const float g2 = gs + cs;
    const uint32_t kn = hybrid_key<kFastDiv>(d, g2, heuristic0(nr, nc, goal_r, goal_c) + cn);
    const uint32_t kg = hybrid_key<kFastDiv>(d, gs, heuristic0(gr0, gc0, goal_r, goal_c) + cs);
    const bool beats = inb & (gn > g2) & ((kn < kg) | ((kn == kg) & (n < s)));
    return __ballot(beats) != 0ull;
}

template <bool kFastDiv, bool kLock = false>
__global__ __launch_bounds__(64) void nastar_forward_hybrid_kernel(const FwdHybridArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.x;
    if constexpr (kLock) {
        if (a.marks != nullptr && a.marks[b] == 0) return;  // not in the batch-coupled class: the early-exit launch's outputs stand
    }
    const int lane = threadIdx.x;
    const HybridDims d = a.d;
    const int HWp = d.nchunks * 64;
    unsigned long long* const cmin = reinterpret_cast<unsigned long long*>(smem);
    unsigned long long* const smin = cmin + d.nsuper * 64;
    unsigned char* const slab = a.workspace + (size_t)b * a.slab_bytes;
    float* const g = reinterpret_cast<float*>(slab);
    uint8_t* const pdir = reinterpret_cast<uint8_t*>(g + HWp);
    const int* const hdr = reinterpret_cast<const int*>(slab + hybrid_header_offset(d.HW));
    const size_t off = (size_t)b * (size_t)d.HW;
    const float* cost = a.cost + off;
    const bool probe = kLock && a.bitmap != nullptr;
    const int budget = (kLock && a.t_end != nullptr) ? __builtin_amdgcn_readfirstlane(*a.t_end + 1) : a.max_iters;

    // ---- start / goal from the fill launch; empty open list -------------------------------------------------------------
    const int sidx = __builtin_amdgcn_readfirstlane(hdr[0]), gidx = __builtin_amdgcn_readfirstlane(hdr[1]);
    for (int c = lane; c < d.nsuper * 64; c += 64) cmin[c] = ~0ull;
    for (int c = lane; c < d.spl * 64; c += 64) smin[c] = ~0ull;
    const int gi = gidx < 0 ? 0 : gidx;
    int goal_c;
    const int goal_r = hybrid_row(gi, d, goal_c);
    __syncthreads();
    float h_start = 0.f;  // :191-192 h = h0 + cost at the start cell (wave-uniform)
    if (sidx >= 0) {
        int sc;
        const int sr = hybrid_row(sidx, d, sc);
        h_start = heuristic0(sr, sc, goal_r, goal_c) + cost[sidx];
    }
    if (lane == 0 && sidx >= 0) {  // open list = {start} (:187), g[start] = 0 (:193); the start is expanded even on an obstacle
        const uint32_t k0 = hybrid_key<kFastDiv>(d, 0.0f, h_start);
        const unsigned long long e = ((unsigned long long)k0 << 32) | (uint32_t)sidx;
        g[sidx] = 0.0f;
        pdir[sidx] = (uint8_t)(PARENT_UNSET | P_PASS);
        cmin[sidx >> 6] = e;
        smin[sidx >> 12] = e;
    }
    __syncthreads();

    int dr, dc;
    neighbour_delta(lane & 7, dr, dc);
    int status = NASTAR_OK;
    int iters = 0;
    bool solved = false, goal_hit = false, coupled = false;
    uint32_t bits = 0u;  // probe: goal selections of the current 32 steps
    uint32_t* const bm = probe ? a.bitmap + (size_t)b * (size_t)a.bitmap_words : nullptr;
    if (sidx < 0 || gidx < 0) {
        status = NASTAR_ERR_UNSOLVABLE;  // not a one-hot start / goal map
    } else {
        // (key << 32 | cell) of the next selection, wave-uniform in scalar registers; ~0 = open list empty
        uint32_t sel_key = hybrid_key<kFastDiv>(d, 0.0f, __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(h_start))));
        uint32_t sel_cell = (uint32_t)sidx;
        while (iters < budget) {  // :203
            // ---- select: the entry the previous step left behind names s* (no LDS read, no reduction on this path) -------------
            const int s = (int)sel_cell;
            if (s < 0) {  // every entry idle (~0ull: key KEY_INF, cell ~0): open list empty (:68 would divide by zero)
                status = NASTAR_ERR_UNSOLVABLE;
                break;
            }
            if (a.sel_log != nullptr && !probe && lane == 0) a.sel_log[(size_t)b * (size_t)a.max_iters + iters] = s;
            const bool at_goal = s == gidx;
            if constexpr (kLock) {
                if (probe) {
                    if (at_goal) bits |= 1u << (iters & 31);
                    if ((iters & 31) == 31) {
                        if (lane == 0) bm[iters >> 5] = bits;
                        bits = 0u;
                    }
                }
            }
            ++iters;
            if (!kLock && at_goal) {
                // :219-220,:251 every later step of the reference is a fixed point -- unless the goal's own expansion would open a cell that beats
                // it (nastar_capi.hip, same test): reported as summary[NASTAR_SUMMARY_COUPLED] and, per map, in marks[]
                if (a.summary != nullptr || a.marks_out != nullptr) coupled = hybrid_goal_beaten<kFastDiv>(d, g, cost, s, lane, dr, dc, goal_r, goal_c);
                if (lane == 0) g[s] = NASTAR_NEG_INF;  // :222-223 the goal joins the closed list
                solved = true;
                break;
            }
            goal_hit |= at_goal;
            const int C = s >> 6, S = s >> 12;
            // ---- the open list WITHOUT the chunk / super-chunk of s*, as the previous step left it (LDS, issued ahead of the HBM loads) ----
            const unsigned long long ev = cmin[S * 64 + lane];
            unsigned long long e0 = (lane * d.spl == S) ? ~0ull : smin[lane * d.spl];
            for (int j = 1; j < d.spl; ++j) {  // (maps above 512x512: several super-chunk entries per lane, contiguous -- the first minimal one wins)
                const unsigned long long ej = (lane * d.spl + j == S) ? ~0ull : smin[lane * d.spl + j];
                e0 = (uint32_t)(ej >> 32) < (uint32_t)(e0 >> 32) ? ej : e0;
            }
            int c;
            const int r = hybrid_row_nb(s, d, c);
            const int nr = r + dr, nc = c + dc;
            const bool inb = (lane < 8) & ((unsigned)nr < (unsigned)d.H) & ((unsigned)nc < (unsigned)d.W);  // conv2d zero padding
            const int n = inb ? s + dr * d.W + dc : s;
            const int ic = C * 64 + lane;
            const bool icv = ic < d.HW;
            global_step_fence();  // the previous step's g / pdir stores have reached L2
            // ---- ONE round trip: everything this step reads from HBM --------------------------------------------------
            const float gs = g[s];
            const float gn = g[n];
            const float gc = g[ic];
            const float cs = cost[s];
            const float cn = cost[n];
            const float cc = cost[icv ? ic : 0];
            // ---- in the shadow of that round trip: nothing below needs a loaded value until `g2` -------------------------
            // rest of the super-chunk of s* (its 64 chunk entries but the one of s*) and rest of the map (every other super-chunk): entries
            // ascend with the lane, so the first lane that holds the minimal key holds the first minimal entry
            const uint32_t kS = lane == (C & 63) ? KEY_INF : (uint32_t)(ev >> 32);
            const uint32_t kE = (uint32_t)(e0 >> 32);
            uint32_t mS, mE;
            wave_min_scalar_u32x2(kS, kE, mS, mE);
            const uint32_t cS = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)ev, __builtin_ctzll(__ballot(kS == mS)));
            const uint32_t cE = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)e0, __builtin_ctzll(__ballot(kE == mE)));
            int icc;
            const int icr = hybrid_row_nb(icv ? ic : 0, d, icc);
            float hn = heuristic0(nr, nc, goal_r, goal_c);
            float hc = heuristic0(icr, icc, goal_r, goal_c);
            // (the compiler otherwise sinks both heuristics below the first use of a loaded value, into regions predicated on `upd` / `open_c`:
            //  a lone wavefront pays for predicated-off lanes anyway, and there they sit behind the round trip instead of inside it)
            asm volatile("" : "+v"(hn), "+v"(hc) : : "memory");
            // ---- the loaded values ------------------------------------------------------------------------------------------
            const float g2 = gs + cs;                                              // :234 step cost of the node being LEFT
            const bool upd = inb & (gn > g2);                                      // :229,:235
            // chunk minimum without s* (lock-step: a selected goal stays on the open list, :224): open <=> finite g
            const bool open_c = icv & (fabsf(gc) < NASTAR_POS_INF) & ((ic != s) | (kLock && at_goal));
            uint32_t kn_all = hybrid_key<kFastDiv>(d, g2, hn + cn);
            uint32_t kc_all = hybrid_key<kFastDiv>(d, gc, hc + cc);                // (of +-inf for cells that are not open: discarded)
            asm volatile("" : "+v"(kn_all), "+v"(kc_all));                         // both keys for all lanes, side by side (no predicated regions)
            const uint32_t kn = upd ? kn_all : KEY_INF;
            const uint32_t kc = open_c ? kc_all : KEY_INF;
            // the chunk of s* without s* (64 lanes, ascending cells) and, beside it, the relaxed neighbours (lanes 0..7 in raster order: ascending cells)
            uint32_t mC, mN;
            wave_min_scalar_u32_and8(kc, kn, mC, mN);
            const uint32_t cC = (uint32_t)(C * 64 + __builtin_ctzll(__ballot(kc == mC)));   // (mC == KEY_INF: every lane matches, masked below)
            const uint32_t cN = (uint32_t)__builtin_amdgcn_readlane(n, __builtin_ctzll(__ballot(kn == mN)));
            // ---- stores: closed list, relaxed neighbours (:222-225, :238-249) ----------------------------------------
            if (lane == 0 && !(kLock && at_goal)) g[s] = NASTAR_NEG_INF;
            if (upd) {
                g[n] = g2;
                pdir[n] = (uint8_t)(P_PASS | (uint32_t)lane);
            }
            // ---- open list (LDS executes a wavefront's operations in order; nothing is read back in this step) ------------
            // chunk of s*: its cells without s* ...; super-chunk of s*: its other chunks and that; the neighbours enter both levels by ds_min
            const uint32_t kCS = min(mC, mS);
            const uint32_t cCS = min(mC == kCS ? cC : 0xFFFFFFFFu, mS == kCS ? cS : 0xFFFFFFFFu);
            const unsigned long long en = ((unsigned long long)kn << 32) | (uint32_t)n;
            if (lane == 0) {
                cmin[C] = mC == KEY_INF ? ~0ull : (((unsigned long long)mC << 32) | cC);
                smin[S] = kCS == KEY_INF ? ~0ull : (((unsigned long long)kCS << 32) | cCS);
            }
            wave_order();
            if (upd) {
                atomicMin(&cmin[n >> 6], en);                                      // :242 (re)opened neighbours enter their chunk's minimum
                atomicMin(&smin[n >> 12], en);                                     // ... and their super-chunk's (a neighbour may sit in another one)
            }
            wave_order();
            // ---- the next selection: first-index minimum of {rest of the map, super-chunk of s* without s*, relaxed neighbours} ----
            const uint32_t kX = min(mE, mN);
            const uint32_t cX = min(mE == kX ? cE : 0xFFFFFFFFu, mN == kX ? cN : 0xFFFFFFFFu);
            sel_key = min(kCS, kX);
            sel_cell = min(kCS == sel_key ? cCS : 0xFFFFFFFFu, kX == sel_key ? cX : 0xFFFFFFFFu);
            if (sel_key == KEY_INF) sel_cell = 0xFFFFFFFFu;
        }
    }
    global_step_fence();
    if constexpr (kLock) {
        if (probe) {  // the words this map's search did not reach say "no goal selection"
            if (lane == 0) {
                if (iters & 31) bm[iters >> 5] = bits;
                for (int w = (iters + 31) >> 5; w < a.bitmap_words; ++w) bm[w] = 0u;
            }
            return;
        }
        if (goal_hit && lane == 0) g[gidx] = NASTAR_NEG_INF;  // histories holds the goal (:222-223); nothing reads its g any more
    }
    if (lane == 0) {
        a.iters[b] = iters;
        a.status[b] = status;
        if (a.marks_out != nullptr) a.marks_out[b] = coupled ? 1 : 0;
        if (a.summary) {
            if (status != NASTAR_OK) a.summary[status] = 1;
            if (coupled) a.summary[NASTAR_SUMMARY_COUPLED] = 1;
        }
    }

    // ---- backtrack (:96-125): walk to the start, cap = this map's own step count in the budget-truncated case ----------
    if (gidx >= 0 && lane == 0) {
        const int cap = solved ? d.HW : iters - 1;
        uint32_t m = pdir[gidx];
        pdir[gidx] = (uint8_t)(m | P_PATH);
        uint32_t code = m & P_DIRMASK;
        if (code != PARENT_UNSET) {
            int pdr, pdc;
            neighbour_delta((int)code, pdr, pdc);
            int loc = gidx - (pdr * d.W + pdc);
            for (int k2 = 0; k2 < cap; ++k2) {
                const uint32_t ml = pdir[loc];
                pdir[loc] = (uint8_t)(ml | P_PATH);
                if (loc == sidx) break;
                const uint32_t cd = ml & P_DIRMASK;
                if (cd == PARENT_UNSET) break;
                neighbour_delta((int)cd, pdr, pdc);
                loc -= pdr * d.W + pdc;
            }
        }
    }
    global_step_fence();
}

}  // namespace nastar
