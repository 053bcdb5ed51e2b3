// nastar_conv_wgrad.hip.h -- weight gradient of a 3x3 convolution (padding 1) on the fp16 MFMA, straight from NHWC activations:
//
//   dW[tap][ci][co] = out_scale * sum_p dz[p][co] * a[p + tap][ci]              (autograd of reference planner/encoder.py:60-78's convs)
//
// As a GEMM the reduction index is the PIXEL, but NHWC keeps channels contiguous: an MFMA operand wants 8 consecutive k (pixels) of one
// row (channel) per lane, i.e. the transpose of what a coalesced load delivers.  gfx950's LDS transpose read does that on the fly:
// ds_read_b64_tr_b16 -- measured semantics (tools/ubench/tr16.hip): in every 16-lane group, lane i receives element (i & 3) of the
// 8-byte rows addressed by lanes (i >> 2) + 4 j, j = 0..3.  With lane r addressing [pixel P0 + (r >> 2)][channels c0 + 4 (r & 3) ..+3]
// of a pixel-major LDS tile, lane i ends up with channel c0 + i of pixels P0 .. P0+3: two such reads are the 8-pixel MFMA fragment of
// "its" channel for BOTH operands (same pixel <-> k assignment on the dz and on the activation side), no transposed copies in HBM.
//
//   * a workgroup owns a (COB x 32) x (CIB x 32) channel tile of dW for all 9 taps and a strided set of pixel chunks; wavefront
//     (wco, wci) holds the 9 accumulators (one per tap) of its 32 x 32 block: 144 VGPRs;
//   * a chunk = RC whole image rows (RC * W = 64 pixels: 4 MFMA k-steps of 16 pixels): dz rows pixel-major in LDS, the activation rows
//     with a one-pixel ZERO frame around them ((RC+2) x (W+2) slots, rows outside the image zero): a tap is a constant LDS offset,
//     conv2d's zero padding costs nothing in the loop, and every fragment address is a per-lane constant computed once;
//   * pixel rows in LDS are padded by 64 bytes (row stride = 64 mod 256) so that the 4 pixels x 64 bytes a 32-lane pass of the
//     transpose read touches fall into 4 different bank quarters;
//   * kSplit ("f16x3"): operands are [hi | lo] fp16 pairs; per tap dz_hi*a_hi + dz_lo*a_hi + dz_hi*a_lo (fp32 accumulation);
//   * epilogue: acc * out_scale (undoes the power-of-two gradient scale) -> fp32 atomic adds into dW [9][CI][CO] (the caller zeroes
//     it; pixel splits and all workgroups of a tile add into the same words).
// Shapes: W in {2,4,...,64} dividing 64, H a multiple of RC = 64 / W; CO, CI multiples of 32 (zero padded).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nastar_encoder.hip.h"

namespace nastar {

struct WgradArgs {
    const uint16_t* dz;  // [P][CO (x2)] fp16 NHWC
    const uint16_t* a;   // [P][CI (x2)] fp16 NHWC (the layer's input activations)
    float* dw;           // [9][CI][CO] fp32, accumulated with atomics
    float out_scale;
    int B, H, W, CO, CI;
    int nchunk;          // B*H*W / 64
    int nsplit;          // pixel splits: gridDim.x = nsplit * (CO/(32 COB)) * (CI/(32 CIB))
};

typedef __fp16 nastar_h4 __attribute__((__vector_size__(4 * sizeof(__fp16))));

__device__ __forceinline__ bf16x8 wg_tr_read2(uint32_t addr0, uint32_t addr1)
{
    // two transpose reads = pixels +0..3 and +4..7 of this lane's channel
    const nastar_h4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16(reinterpret_cast<__attribute__((address_space(3))) nastar_h4*>(addr0));
    const nastar_h4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16(reinterpret_cast<__attribute__((address_space(3))) nastar_h4*>(addr1));
    union { struct { nastar_h4 a, b; } h; bf16x8 v; } u;
    u.h.a = lo;
    u.h.b = hi;
    return u.v;
}

template <int COB, int CIB, bool kSplit>
__global__ __launch_bounds__(64 * COB * CIB, 2) void nastar_conv3x3_wgrad_kernel(const WgradArgs g)
{
    constexpr int NTHR = 64 * COB * CIB;
    constexpr int M = kSplit ? 2 : 1;
    constexpr int RDZ = COB * 64 * M + 64;  // bytes per pixel row of the dz tile (32 channels = 64 B per block and precision half) + pad
    constexpr int RA = CIB * 64 * M + 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int W = g.W, RC = 64 / W, PW = W + 2;
    unsigned char* dzt = smem;                    // [64][RDZ]
    unsigned char* at = smem + 64 * RDZ;          // [(RC+2)*(W+2)][RA]
    const int nslot_a = (RC + 2) * PW;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wco = wave % COB, wci = wave / COB;
    const int ntco = g.CO / (32 * COB), ntci = g.CI / (32 * CIB);
    int t = blockIdx.x;
    const int split = t % g.nsplit; t /= g.nsplit;
    const int tco = t % ntco;
    const int tci = t / ntco;
    if (tci >= ntci) return;
    const int co0 = tco * 32 * COB, ci0 = tci * 32 * CIB;
    const int sdz = M * g.CO, sa = M * g.CI;      // fp16 elements per pixel in HBM

    // ---- per-lane fragment addresses (constants): k-step ks, read half tt: pixel = 16 ks + 8 kh + 4 tt + (r16 >> 2) ----
    const int r16 = lane & 15, grp = (lane >> 4) & 1, kh = lane >> 5;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;  // LDS byte address of the dynamic region
    uint32_t adz[4][2], aa[4][2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int pix = 16 * ks + 8 * kh + 4 * tt + (r16 >> 2);
            const int chan = 16 * grp + 4 * (r16 & 3);
            adz[ks][tt] = lds0 + pix * RDZ + wco * 64 + chan * 2;
            const int row = pix / W, col = pix - row * W;
            aa[ks][tt] = lds0 + 64 * RDZ + ((row + 1) * PW + col + 1) * RA + wci * 64 + chan * 2;
        }

    f32x16 acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;

    const int rows_per_img = g.H / RC;  // chunks per image
    for (int ch = split; ch < g.nchunk; ch += g.nsplit) {
        const int b = ch / rows_per_img, y0 = (ch - b * rows_per_img) * RC;
        const size_t p0 = ((size_t)b * g.H + y0) * W;
        __syncthreads();  // the previous chunk's fragments are consumed
        // ---- stage dz: 64 pixels x (M * COB * 64) bytes, 16-byte chunks ----
        {
            constexpr int CPP = M * COB * 4;  // chunks per pixel
            for (int q = tid; q < 64 * CPP; q += NTHR) {
                const int pix = q / CPP, c = q - pix * CPP;
                const int half = c / (COB * 4), cc = c - half * (COB * 4);
                const uint4 v = *reinterpret_cast<const uint4*>(g.dz + (p0 + pix) * sdz + half * g.CO + co0 + cc * 8);
                *reinterpret_cast<uint4*>(dzt + pix * RDZ + half * (COB * 64) + cc * 16) = v;
            }
        }
        // ---- stage a: (RC+2) x (W+2) slots, zero outside the image ----
        {
            constexpr int CPP = M * CIB * 4;
            for (int q = tid; q < nslot_a * CPP; q += NTHR) {
                const int slot = q / CPP, c = q - slot * CPP;
                const int half = c / (CIB * 4), cc = c - half * (CIB * 4);
                const int sr = slot / PW, sc = slot - sr * PW;
                const int y = y0 + sr - 1, x = sc - 1;
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if ((unsigned)y < (unsigned)g.H && (unsigned)x < (unsigned)W)
                    v = *reinterpret_cast<const uint4*>(g.a + (((size_t)b * g.H + y) * W + x) * sa + half * g.CI + ci0 + cc * 8);
                *reinterpret_cast<uint4*>(at + slot * RA + half * (CIB * 64) + cc * 16) = v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 zh = wg_tr_read2(adz[ks][0], adz[ks][1]);
            bf16x8 zl = zh;
            if constexpr (kSplit) zl = wg_tr_read2(adz[ks][0] + COB * 64, adz[ks][1] + COB * 64);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int toff = ((tap / 3 - 1) * PW + (tap % 3 - 1)) * RA;
                const bf16x8 xh = wg_tr_read2(aa[ks][0] + toff, aa[ks][1] + toff);
                acc[tap] = mfma16<true>(zh, xh, acc[tap]);
                if constexpr (kSplit) {
                    const bf16x8 xl = wg_tr_read2(aa[ks][0] + toff + CIB * 64, aa[ks][1] + toff + CIB * 64);
                    acc[tap] = mfma16<true>(zl, xh, acc[tap]);
                    acc[tap] = mfma16<true>(zh, xl, acc[tap]);
                }
            }
        }
    }
    // ---- epilogue: D[row = co][col = ci]; row = (reg & 3) + 8 (reg >> 2) + 4 kh, col = lane & 31 ----
    const int ci = ci0 + wci * 32 + (lane & 31);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        float* dst = g.dw + ((size_t)tap * g.CI + ci) * g.CO + co0 + wco * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = (r & 3) + 8 * (r >> 2) + 4 * kh;
            unsafeAtomicAdd(dst + co, acc[tap][r] * g.out_scale);
        }
    }
}

}  // namespace nastar
