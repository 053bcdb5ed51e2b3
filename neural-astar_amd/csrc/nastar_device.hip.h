// nastar_device.hip.h -- device-side building blocks shared by the forward and backward kernels.
// gfx950 (CDNA4) only: 64-lane wavefronts, DPP cross-lane ops, LDS-resident search state.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nastar {

constexpr uint32_t KEY_INF = 0xFFFFFFFFu;  // "not on the open list"
constexpr uint32_t PARENT_UNSET = 8u;      // parents[] still holds its initial value goal_idx (differentiable_astar.py:195-198)
constexpr int CHUNK = 64;              // cells per chunk == wavefront width

// ---- cross-lane reductions: 4 DPP steps inside each 16-lane row, then 4 readlanes + scalar ops ----------
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_mov(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, false);
}
// `old` = identity of the consuming op lets LLVM's DPP combiner fold the move into the VALU op (v_min_u32_dpp ...)
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_mov_id(uint32_t v, uint32_t identity)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)v, CTRL, 0xF, 0xF, false);
}
constexpr int DPP_QUAD_XOR1 = 0xB1;     // quad_perm:[1,0,3,2]
constexpr int DPP_QUAD_XOR2 = 0x4E;     // quad_perm:[2,3,0,1]
constexpr int DPP_ROW_HALF_MIRROR = 0x141;
constexpr int DPP_ROW_MIRROR = 0x140;

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
{
    v = min(v, dpp_mov_id<DPP_QUAD_XOR1>(v, 0xFFFFFFFFu));
    v = min(v, dpp_mov_id<DPP_QUAD_XOR2>(v, 0xFFFFFFFFu));
    v = min(v, dpp_mov_id<DPP_ROW_HALF_MIRROR>(v, 0xFFFFFFFFu));
    v = min(v, dpp_mov_id<DPP_ROW_MIRROR>(v, 0xFFFFFFFFu));
    uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 0);
    uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
    uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)v, 32);
    uint32_t d = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
    return min(min(a, b), min(c, d));
}

// min over all 64 lanes, result in EVERY lane (stays in a VGPR: no VALU->SALU->VALU round trip, which costs a lone
// wavefront ~25 cycles per hop on gfx950).  4 DPP steps reduce each 16-lane row, v_permlane16_swap / v_permlane32_swap
// (gfx950) then exchange rows / halves.
__device__ __forceinline__ uint32_t wave_min_all_u32(uint32_t v)
{
    v = min(v, dpp_mov_id<DPP_QUAD_XOR1>(v, 0xFFFFFFFFu));
    v = min(v, dpp_mov_id<DPP_QUAD_XOR2>(v, 0xFFFFFFFFu));
    v = min(v, dpp_mov_id<DPP_ROW_HALF_MIRROR>(v, 0xFFFFFFFFu));
    v = min(v, dpp_mov_id<DPP_ROW_MIRROR>(v, 0xFFFFFFFFu));
    auto a = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    v = min(a[0], a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return min(b[0], b[1]);
}

// min over all 64 lanes as a wave-uniform SCALAR: 4 DPP steps per 16-lane row, then row_bcast:15 (rows 1,3 absorb the
// row before them) and row_bcast:31 (rows 2,3 absorb lane 31) leave the total in lane 63; one v_readlane moves it to an
// SGPR that later VALU compares take directly as an operand.  Three instructions shorter than the permlane-swap form.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_mov_rows(uint32_t v, uint32_t identity)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)v, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ uint32_t wave_min_scalar_u32(uint32_t v)
{
    v = min(v, dpp_mov_id<DPP_QUAD_XOR1>(v, 0xFFFFFFFFu));
    v = min(v, dpp_mov_id<DPP_QUAD_XOR2>(v, 0xFFFFFFFFu));
    v = min(v, dpp_mov_id<DPP_ROW_HALF_MIRROR>(v, 0xFFFFFFFFu));
    v = min(v, dpp_mov_id<DPP_ROW_MIRROR>(v, 0xFFFFFFFFu));
    v = min(v, dpp_mov_rows<0x142, 0xA>(v, 0xFFFFFFFFu));  // row_bcast:15
    v = min(v, dpp_mov_rows<0x143, 0xC>(v, 0xFFFFFFFFu));  // row_bcast:31
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// two independent minima side by side: stage k of one sits in the two wait states a VALU write -> DPP read of the other needs
__device__ __forceinline__ void wave_min_scalar_u32x2(uint32_t a, uint32_t b, uint32_t& ma, uint32_t& mb)
{
    a = min(a, dpp_mov_id<DPP_QUAD_XOR1>(a, 0xFFFFFFFFu));
    b = min(b, dpp_mov_id<DPP_QUAD_XOR1>(b, 0xFFFFFFFFu));
    a = min(a, dpp_mov_id<DPP_QUAD_XOR2>(a, 0xFFFFFFFFu));
    b = min(b, dpp_mov_id<DPP_QUAD_XOR2>(b, 0xFFFFFFFFu));
    a = min(a, dpp_mov_id<DPP_ROW_HALF_MIRROR>(a, 0xFFFFFFFFu));
    b = min(b, dpp_mov_id<DPP_ROW_HALF_MIRROR>(b, 0xFFFFFFFFu));
    a = min(a, dpp_mov_id<DPP_ROW_MIRROR>(a, 0xFFFFFFFFu));
    b = min(b, dpp_mov_id<DPP_ROW_MIRROR>(b, 0xFFFFFFFFu));
    a = min(a, dpp_mov_rows<0x142, 0xA>(a, 0xFFFFFFFFu));  // row_bcast:15
    b = min(b, dpp_mov_rows<0x142, 0xA>(b, 0xFFFFFFFFu));
    a = min(a, dpp_mov_rows<0x143, 0xC>(a, 0xFFFFFFFFu));  // row_bcast:31
    b = min(b, dpp_mov_rows<0x143, 0xC>(b, 0xFFFFFFFFu));
    ma = (uint32_t)__builtin_amdgcn_readlane((int)a, 63);
    mb = (uint32_t)__builtin_amdgcn_readlane((int)b, 63);
}

// the minimum of all 64 lanes of `a` and, beside it, the minimum of lanes 0..7 of `b` (its first three stages)
__device__ __forceinline__ void wave_min_scalar_u32_and8(uint32_t a, uint32_t b, uint32_t& ma, uint32_t& mb8)
{
    a = min(a, dpp_mov_id<DPP_QUAD_XOR1>(a, 0xFFFFFFFFu));
    b = min(b, dpp_mov_id<DPP_QUAD_XOR1>(b, 0xFFFFFFFFu));
    a = min(a, dpp_mov_id<DPP_QUAD_XOR2>(a, 0xFFFFFFFFu));
    b = min(b, dpp_mov_id<DPP_QUAD_XOR2>(b, 0xFFFFFFFFu));
    a = min(a, dpp_mov_id<DPP_ROW_HALF_MIRROR>(a, 0xFFFFFFFFu));
    b = min(b, dpp_mov_id<DPP_ROW_HALF_MIRROR>(b, 0xFFFFFFFFu));
    a = min(a, dpp_mov_id<DPP_ROW_MIRROR>(a, 0xFFFFFFFFu));
    mb8 = (uint32_t)__builtin_amdgcn_readlane((int)b, 0);
    a = min(a, dpp_mov_rows<0x142, 0xA>(a, 0xFFFFFFFFu));  // row_bcast:15
    a = min(a, dpp_mov_rows<0x143, 0xC>(a, 0xFFFFFFFFu));  // row_bcast:31
    ma = (uint32_t)__builtin_amdgcn_readlane((int)a, 63);
}

// min over each 16-lane DPP row, result in every lane of the row
__device__ __forceinline__ uint32_t row_min16_u32(uint32_t v)
{
    v = min(v, dpp_mov_id<DPP_QUAD_XOR1>(v, 0xFFFFFFFFu));
    v = min(v, dpp_mov_id<DPP_QUAD_XOR2>(v, 0xFFFFFFFFu));
    v = min(v, dpp_mov_id<DPP_ROW_HALF_MIRROR>(v, 0xFFFFFFFFu));
    v = min(v, dpp_mov_id<DPP_ROW_MIRROR>(v, 0xFFFFFFFFu));
    return v;
}

__device__ __forceinline__ int wave_max_i32(int v)
{
    v = max(v, (int)dpp_mov<DPP_QUAD_XOR1>((uint32_t)v));
    v = max(v, (int)dpp_mov<DPP_QUAD_XOR2>((uint32_t)v));
    v = max(v, (int)dpp_mov<DPP_ROW_HALF_MIRROR>((uint32_t)v));
    v = max(v, (int)dpp_mov<DPP_ROW_MIRROR>((uint32_t)v));
    int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    int c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    return max(max(a, b), max(c, d));
}

__device__ __forceinline__ float wave_sum_f32(float v)
{
    v += __uint_as_float(dpp_mov<DPP_QUAD_XOR1>(__float_as_uint(v)));
    v += __uint_as_float(dpp_mov<DPP_QUAD_XOR2>(__float_as_uint(v)));
    v += __uint_as_float(dpp_mov<DPP_ROW_HALF_MIRROR>(__float_as_uint(v)));
    v += __uint_as_float(dpp_mov<DPP_ROW_MIRROR>(__float_as_uint(v)));
    float a = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), 0));
    float b = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), 16));
    float c = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), 32));
    float d = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), 48));
    return (a + b) + (c + d);
}

// ---- fp32 -> order-preserving uint32 (so LDS ds_min_u32 and integer DPP mins order f correctly) ----------
__device__ __forceinline__ uint32_t f32_to_ord(float f)
{
    uint32_t u = __float_as_uint(f);
    uint32_t mask = (uint32_t)((int32_t)u >> 31) | 0x80000000u;
    return u ^ mask;
}
__device__ __forceinline__ float ord_to_f32(uint32_t k)
{
    uint32_t mask = (k & 0x80000000u) ? 0x80000000u : 0xFFFFFFFFu;
    return __uint_as_float(k ^ mask);
}

// exact n / d for n, d < 2^16 with magic = floor(2^32 / d) + 1
__device__ __forceinline__ uint32_t div_magic(uint32_t n, uint32_t magic) { return __umulhi(n, magic); }

// ---- correctly rounded square root of an exactly representable integer >= 0 ------------------------------------------------------------
// v_sqrt_f32 is a 1-ulp instruction: NOT correctly rounded for ~15 % of the integers (first miss at 6; tools/ubench/sqrt_check.hip,
// profiles/r05/sqrt_check.txt) -- and neither is __fsqrt_rn, which compiles to it.  get_heuristic's h0 = cheb + 0.001 * sqrt(dr^2 + dc^2)
// absorbs the error while max(|dr|, |dc|) < 140 (exhaustive: h0 bit-identical to the host for every pair below that) -- the
// hand-scheduled 16 / 32 / 64 streams may keep the bare instruction -- but 16 pairs differ in [128, 256), 62 in [256, 512): on maps with a
// side above 140 a search could leave the reference's (found by tools/fuzz_parity.py: 12 of 1067 large-map cases, rounds 1-4 latent).
// One FMA residual test against each neighbour (LLVM's own IEEE lowering of sqrt) fixes every integer below 2^24.
__device__ __forceinline__ float sqrt_rn_int(float x)
{
    // (no guard for x = 0: s = 0, its lower neighbour is a NaN pattern and fails `<=`, its upper neighbour gives a residual of 0 -- s stays 0;
    //  a guard would cost every caller a divergent region)
    const float s = __builtin_amdgcn_sqrtf(x);
    const float sd = __uint_as_float(__float_as_uint(s) - 1u), su = __uint_as_float(__float_as_uint(s) + 1u);
    const float vp = __builtin_fmaf(-sd, s, x), vs = __builtin_fmaf(-su, s, x);
    const float t = vp <= 0.f ? sd : s;
    return vs > 0.f ? su : t;
}

// ---- get_heuristic (differentiable_astar.py:26-52), one cell ------------------------------------------
// h0 = fl( fl(|dr|+|dc| - min(|dr|,|dc|)) + fl( fl32(0.001) * fl(sqrt(dr^2+dc^2)) ) ); every op one fp32
// rounding (the TU is compiled with -ffp-contract=off), sqrt correctly rounded (sqrt_rn_int).
__device__ __forceinline__ float heuristic0(int r, int c, int goal_r, int goal_c)
{
    float a = (float)r - (float)goal_r;
    float b = (float)c - (float)goal_c;
    float dr = fabsf(a), dc = fabsf(b);
    float cheb = (dr + dc) - fminf(dr, dc);
    float euc = sqrt_rn_int(a * a + b * b);
    return cheb + 0.001f * euc;
}

// neighbour code j in [0,8) -> (dr, dc), raster order of the 3x3 stencil without its centre
__device__ __forceinline__ void neighbour_delta(int j, int& dr, int& dc)
{
    int k = j + (j >= 4);
    int q = (k >= 6) ? 2 : ((k >= 3) ? 1 : 0);
    dr = q - 1;
    dc = (k - 3 * q) - 1;
}

}  // namespace nastar
