// nastar_conv_wgrad.hip.h -- weight gradient of a 3x3 convolution (padding 1) on the fp16 MFMA, straight from NHWC activations:
//
//   dW[co][ci][tap] = out_scale * sum_p dz[p][co] * a[p + tap][ci]              (autograd of reference planner/encoder.py:60-78's convs)
//
// As a GEMM the reduction index is the PIXEL, but NHWC keeps channels contiguous: an MFMA operand wants 8 consecutive k (pixels) of one
// row (channel) per lane, i.e. the transpose of what a coalesced load delivers.  gfx950's LDS transpose read does that on the fly:
// ds_read_b64_tr_b16 -- measured semantics (tools/ubench/tr16.hip): in every 16-lane group, lane i receives element (i & 3) of the
// 8-byte rows addressed by lanes (i >> 2) + 4 j, j = 0..3.  With lane r addressing [pixel P0 + (r >> 2)][channels c0 + 4 (r & 3) ..+3]
// of a pixel-major LDS tile, lane i ends up with channel c0 + i of pixels P0 .. P0+3: two such reads are the 8-pixel MFMA fragment of
// "its" channel for BOTH operands (same pixel <-> k assignment on the dz and on the activation side), no transposed copies in HBM.
//
//   * a workgroup owns a (COB x 32) x (CIB x 32) channel tile of dW for all 9 taps and a strided set of pixel chunks; it runs
//     3 x COB x CIB wavefronts: wavefront (wco, wci, wdy) holds the 3 accumulators (dx = -1, 0, 1) of its 32 x 32 block and tap row
//     dy: 48 accumulator registers, which leaves room to PREFETCH the next chunk into registers while the matrix pipe works;
//   * tiny images (H*W <= 48: the deep levels of a U-Net) are taken G at a time, each with its own zero frame; otherwise
//   * a chunk = R whole image rows (NP = R * W <= 96 pixels = up to 6 MFMA k-steps of 16 pixels; 64 pixels for the power-of-two map
//     widths, 96 / 72 / 90 for 96-, 12-, 45-wide maps; the dz rows of a partial last k-step are zero): dz rows pixel-major in LDS, the
//     activation rows with a one-pixel ZERO frame around them ((R+2) x (W+2) slots, rows outside the image zero): a tap is a constant LDS
//     offset, conv2d's zero padding costs nothing in the loop, and every staging / fragment address is a per-thread constant computed once;
//   * pixel rows in LDS are padded to a stride of 64 or 192 (mod 256) bytes so that the 4 pixels x 64 bytes a 32-lane pass of the
//     transpose read touches fall into 4 different bank quarters;
//   * kSplit ("f16x3"): operands are [hi | lo] fp16 pairs; per tap dz_hi*a_hi + dz_lo*a_hi + dz_hi*a_lo (fp32 accumulation);
//   * no atomics: every workgroup stores its partial tile to part[split][tap][ci][co]; nastar_wgrad_reduce_kernel sums the splits in
//     a fixed order (bitwise reproducible), applies out_scale (undoes the power-of-two gradient scale read from device memory) and
//     writes torch's [co][ci][3][3] layout cropped to the real channel counts.
//   * images wider than 96 pixels (round 6): a chunk row is a SEGMENT of an image row -- the widest divisor of the image width that is <= 96
//     pixels (nastar_wgrad_segment) -- framed exactly like a whole row, except that a frame COLUMN now holds the neighbouring segment's pixel
//     when there is one (in-image test per chunk: x0 + column); everything inside the matrix loop is unchanged.
// Shapes: W >= 2 with a divisor in [2, 96] (any width <= 96; 128 -> 64, 160 -> 80, ...), H a multiple of the chunk's row count R
// (nastar_wgrad_chunk_rows); CO, CI multiples of 32 (zero padded).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nastar_encoder.hip.h"

namespace nastar {

struct WgradArgs {
    const uint16_t* dz;  // [P][CO (x2)] fp16 NHWC
    const uint16_t* a;   // [P][CI (x2)] fp16 NHWC (the layer's input activations)
    float* part;         // [nsplit][9][CI][CO] fp32 partial sums
    int B, H, W, CO, CI; // W = width of a CHUNK row: the image width, or a segment of it for images wider than 96 pixels (the last segment of a row may reach beyond the image)
    int Wimg, nseg;      // image width; segments per image row (1: W == Wimg)
    int R, G, NP, KS;    // chunk: G blocks of R image rows (G > 1: whole tiny images, R == H) = NP pixels = KS k-steps of 16 (KS*16 >= NP)
    int nchunk;          // B*H / R  (G == 1)  or  ceil(B / G)
    int nsplit;          // pixel splits: gridDim.x = nsplit * (CO/(32 COB)) * (CI/(32 CIB))
};

constexpr int wg_row_bytes(int r) { return r % 128 == 0 ? r + 64 : r; }
constexpr int WG_MAX_SLOTS = 200;       // (R+2)*(W+2) for W <= 64: 198 for W = 64, 200 for W = 48 / 2, 136 for W = 32
constexpr int WG_MAX_SLOTS_WIDE = 294;  // 64 < W <= 96: 3 x 98
constexpr int WG_MAX_PIX = 96, WG_MAX_KS = 6;

// width of a chunk row: the image width up to 96 pixels; otherwise its widest divisor <= 96 when that is at least 64 pixels wide, else
// equal RAGGED segments -- ceil(W / ceil(W / 96)) pixels, the last one of an image row shorter (a prime width, or one whose divisors are all
// small: 194 = 2 x 97): the columns of a segment that lie beyond the image are staged as zeros on both sides of the product
inline int nastar_wgrad_segment(int W)
{
    if (W <= WG_MAX_PIX) return W;
    for (int d = WG_MAX_PIX; d >= 64; --d)
        if (W % d == 0) return d;
    const int nseg = (W + WG_MAX_PIX - 1) / WG_MAX_PIX;
    return (W + nseg - 1) / nseg;
}
inline int nastar_wgrad_segments(int W)
{
    const int seg = nastar_wgrad_segment(W);
    return seg > 0 ? (W + seg - 1) / seg : 0;
}
// rows per chunk for an H x W map, 0 = unsupported: 64-pixel chunks where the chunk row divides 64, otherwise the most rows with R*W <= 96
inline int nastar_wgrad_chunk_rows(int H, int Wimg)
{
    const int W = nastar_wgrad_segment(Wimg);
    if (W < 2 || W > WG_MAX_PIX || H <= 0) return 0;
    if (64 % W == 0 && H % (64 / W) == 0) return 64 / W;
    for (int r = WG_MAX_PIX / W; r >= 1; --r)
        if (H % r == 0) return r;
    return 0;
}
// images per chunk: tiny images (H*W <= 48: the 4x4 and 2x2 levels of a U-Net) are taken several at a time, each with its own zero
// frame in LDS, as many as fit 96 pixels and the 200 frame slots; 1 for everything else (then a chunk is R rows of ONE image)
inline int nastar_wgrad_chunk_images(int H, int W)
{
    if (H * W > 48 || W > WG_MAX_PIX || nastar_wgrad_chunk_rows(H, W) < H) return 1;
    int g = WG_MAX_PIX / (H * W);
    const int by_slots = WG_MAX_SLOTS / ((H + 2) * (W + 2));
    if (g > by_slots) g = by_slots;
    return g < 1 ? 1 : g;
}

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

template <int COB, int CIB, bool kSplit, bool kWide>
__global__ __launch_bounds__(192 * COB * CIB) void nastar_conv3x3_wgrad_kernel(const WgradArgs g)
{
    constexpr int NTHR = 192 * COB * CIB;
    constexpr int M = kSplit ? 2 : 1;
    // bytes per pixel row of the tiles (32 channels = 64 B per block and precision half), padded to 64 or 192 (mod 256): the 4 pixel
    // rows a 32-lane pass of the transpose read touches then fall into 4 different bank quarters
    constexpr int RDZ = wg_row_bytes(COB * 64 * M);
    constexpr int RA = wg_row_bytes(CIB * 64 * M);
    constexpr int CPZ = M * COB * 4, CPA = M * CIB * 4;            // 16-byte chunks per pixel
    constexpr int NZ = (WG_MAX_PIX * CPZ + NTHR - 1) / NTHR;       // staged chunks per thread
    constexpr int NA = ((kWide ? WG_MAX_SLOTS_WIDE : WG_MAX_SLOTS) * CPA + NTHR - 1) / NTHR;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int W = g.W, RC = g.R, PW = W + 2, NP = g.NP, KS = g.KS;
    const int BP = RC * W, BS = (RC + 2) * PW;    // pixels / frame slots per block (one block per image when G > 1)
    const int ZROWS = KS * 16;                    // pixel rows of the dz tile (rows >= NP stay zero)
    unsigned char* dzt = smem;                    // [ZROWS][RDZ]
    // activation tile [G * (RC+2)*(W+2)][RA] follows at smem + ZROWS * RDZ
    const int nslot_a = g.G * BS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wdy = wave % 3, wco = (wave / 3) % COB, wci = wave / (3 * COB);
    const int ntco = g.CO / (32 * COB), ntci = g.CI / (32 * CIB);
    int t = blockIdx.x;
    const int split = t % g.nsplit; t /= g.nsplit;
    const int tco = t % ntco;
    const int tci = t / ntco;
    if (tci >= ntci) return;
    const int co0 = tco * 32 * COB, ci0 = tci * 32 * CIB;
    const int sdz = M * g.CO, sa = M * g.CI;      // fp16 elements per pixel in HBM

    // ---- staging plan (constants per thread): global element offset relative to the chunk's first pixel, LDS byte offset ----
    const int Wi = g.Wimg, BPi = RC * Wi;  // image row pitch / pixels per block in HBM (== W, BP unless the chunk rows are segments)
    int zsrc[NZ], zdst[NZ], zblk[NZ], zcol[NZ], asrc[NA], adst[NA], arow[NA], ablk[NA], acol[NA];
#pragma unroll
    for (int i = 0; i < NZ; ++i) {
        const int q = tid + i * NTHR;
        const int pix = q / CPZ, c = q - pix * CPZ;
        const int half = c / (COB * 4), cc = c - half * (COB * 4);
        const bool ok = q < NP * CPZ;
        const int zr = (pix % BP) / W, zc = pix - (pix / W) * W;  // row / column inside the block (a block = R chunk rows of W pixels)
        zsrc[i] = ok ? ((pix / BP) * BPi + zr * Wi + zc) * sdz + half * g.CO + co0 + cc * 8 : -1;
        zdst[i] = pix * RDZ + half * (COB * 64) + cc * 16;
        zblk[i] = pix / BP;  // which image of the chunk (0 when G == 1)
        zcol[i] = zc;        // column inside the chunk row: beyond the image in the ragged last segment of a row
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int q = tid + i * NTHR;
        const int slot = q / CPA, c = q - slot * CPA;
        const int half = c / (CIB * 4), cc = c - half * (CIB * 4);
        const int gi = slot / BS, sb = slot - gi * BS;
        const int sr = sb / PW, sc = sb - sr * PW;
        const bool ok = q < nslot_a * CPA;
        // relative to the chunk's first pixel (row y0, column x0): block gi, (sr - 1) rows up/down, column sc - 1
        asrc[i] = (gi * BPi + (sr - 1) * Wi + (sc - 1)) * sa + half * g.CI + ci0 + cc * 8;
        ablk[i] = gi;
        adst[i] = ok ? ZROWS * RDZ + slot * RA + half * (CIB * 64) + cc * 16 : -1;
        arow[i] = ok ? sr - 1 : -(1 << 28);  // image row offset of the slot; hugely negative = always zero (unused)
        acol[i] = sc - 1;                    // image column offset of the slot: a frame column is zero unless a neighbouring segment holds it
    }

    // ---- per-lane fragment addresses (constants): k-step ks, read half tt: pixel = 16 ks + 8 kh + 4 tt + (r16 >> 2) ----
    const int r16 = lane & 15, grp = (lane >> 4) & 1, kh = lane >> 5;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;  // LDS byte address of the dynamic region
    // dz side: linear in (ks, tt): adz0 + (16 ks + 4 tt) * RDZ;  activation side: one address per (ks, tt) (rows wrap at W)
    const int chan = 16 * grp + 4 * (r16 & 3);
    const uint32_t adz0 = lds0 + (8 * kh + (r16 >> 2)) * RDZ + wco * 64 + chan * 2;
    uint32_t aa[WG_MAX_KS][2];
#pragma unroll
    for (int ks = 0; ks < WG_MAX_KS; ++ks)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int pix = 16 * ks + 8 * kh + 4 * tt + (r16 >> 2);
            const int pc = pix < NP ? pix : NP - 1;  // pixels of a partial last k-step: their dz rows are zero, any in-range address does
            const int gi = pc / BP, pb = pc - gi * BP;
            const int row = pb / W, col = pb - row * W;
            // tap row dy = wdy - 1 is folded in here; the three dx taps are +-RA around it
            aa[ks][tt] = lds0 + ZROWS * RDZ + (gi * BS + (row + 1 + (wdy - 1)) * PW + col + 1) * RA + wci * 64 + chan * 2;
        }
    // zero rows [NP, ZROWS) of the dz tile once (never staged)
    for (int q = tid; q < (ZROWS - NP) * (RDZ / 16); q += NTHR)
        *reinterpret_cast<uint4*>(dzt + NP * RDZ + q * 16) = make_uint4(0u, 0u, 0u, 0u);

    f32x16 acc[3];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;

    const int rows_per_img = g.H / RC;  // chunk rows per image (x nseg chunks each)
    uint4 zq[NZ], aq[NA];
    auto load_chunk = [&](int ch) {
        // G == 1: rows [y0, y0 + R) of image b;  G > 1: whole images [b, b + G), the last chunk possibly fewer (gcount)
        const int cr = g.G > 1 ? 0 : ch / g.nseg;          // chunk row (over all images), segment of it
        const int x0 = g.G > 1 ? 0 : (ch - cr * g.nseg) * W;
        const int b = g.G > 1 ? ch * g.G : cr / rows_per_img;
        const int y0 = g.G > 1 ? 0 : (cr - b * rows_per_img) * RC;
        const int gcount = g.G > 1 ? (g.B - b < g.G ? g.B - b : g.G) : 1;
        const size_t p0 = ((size_t)b * g.H + y0) * Wi + x0;
        const uint16_t* zb = g.dz + p0 * sdz;
        const uint16_t* ab = g.a + (ptrdiff_t)p0 * sa;
#pragma unroll
        for (int i = 0; i < NZ; ++i) {
            zq[i] = make_uint4(0u, 0u, 0u, 0u);
            if (zsrc[i] >= 0 && zblk[i] < gcount && x0 + zcol[i] < Wi) zq[i] = *reinterpret_cast<const uint4*>(zb + zsrc[i]);
        }
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            aq[i] = make_uint4(0u, 0u, 0u, 0u);
            if ((unsigned)(y0 + arow[i]) < (unsigned)g.H && (unsigned)(x0 + acol[i]) < (unsigned)Wi && ablk[i] < gcount)
                aq[i] = *reinterpret_cast<const uint4*>(ab + asrc[i]);
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int i = 0; i < NZ; ++i)
            if (zsrc[i] >= 0) *reinterpret_cast<uint4*>(dzt + zdst[i]) = zq[i];
#pragma unroll
        for (int i = 0; i < NA; ++i)
            if (adst[i] >= 0) *reinterpret_cast<uint4*>(smem + adst[i]) = aq[i];
    };

    if (split < g.nchunk) load_chunk(split);
    for (int ch = split; ch < g.nchunk; ch += g.nsplit) {
        __syncthreads();  // the previous chunk's fragments are consumed
        store_chunk();
        __syncthreads();
        if (ch + g.nsplit < g.nchunk) load_chunk(ch + g.nsplit);  // in flight during the MFMAs below
#pragma unroll
        for (int ks = 0; ks < WG_MAX_KS; ++ks) {
            if (ks >= KS) break;
            const uint32_t z0 = adz0 + 16 * ks * RDZ, z1 = z0 + 4 * RDZ;
            const bf16x8 zh = wg_tr_read2(z0, z1);
            bf16x8 zl = zh;
            if constexpr (kSplit) zl = wg_tr_read2(z0 + COB * 64, z1 + COB * 64);
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int toff = (dx - 1) * RA;
                const bf16x8 xh = wg_tr_read2(aa[ks][0] + toff, aa[ks][1] + toff);
                acc[dx] = mfma16<true>(zh, xh, acc[dx]);
                if constexpr (kSplit) {
                    const bf16x8 xl = wg_tr_read2(aa[ks][0] + toff + CIB * 64, aa[ks][1] + toff + CIB * 64);
                    acc[dx] = mfma16<true>(zl, xh, acc[dx]);
                    acc[dx] = mfma16<true>(zh, xl, acc[dx]);
                }
            }
        }
    }
    // ---- epilogue: D[row = co][col = ci]; row = (reg & 3) + 8 (reg >> 2) + 4 kh, col = lane & 31; 4 consecutive co per store ----
    const int ci = ci0 + wci * 32 + (lane & 31);
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        const int tap = wdy * 3 + dx;
        float* dst = g.part + (((size_t)split * 9 + tap) * g.CI + ci) * g.CO + co0 + wco * 32;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(dst + 8 * q + 4 * kh) = make_float4(acc[dx][4 * q], acc[dx][4 * q + 1], acc[dx][4 * q + 2], acc[dx][4 * q + 3]);
    }
}

// dw[co][ci][tap] (torch layout, real channel counts) = scale * sum_split part[split][tap][ci_p][co_p];  scale = out_scale / *unscale_dev
__global__ __launch_bounds__(256) void nastar_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int nsplit,
                                                                  int CO, int CI, int co_real, int ci_real, float out_scale,
                                                                  const float* __restrict__ grad_scale_dev)
{
    const int total = 9 * CI * CO;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int co = i % CO;
    int r = i / CO;
    const int ci = r % CI;
    const int tap = r / CI;
    if (co >= co_real || ci >= ci_real) return;
    // fixed summation order (bitwise reproducible); four independent partial sums keep four loads in flight
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = 0;
    for (; k + 4 <= nsplit; k += 4) {
        s0 += part[(size_t)k * total + i];
        s1 += part[(size_t)(k + 1) * total + i];
        s2 += part[(size_t)(k + 2) * total + i];
        s3 += part[(size_t)(k + 3) * total + i];
    }
    for (; k < nsplit; ++k) s0 += part[(size_t)k * total + i];
    const float s = (s0 + s1) + (s2 + s3);
    const float sc = grad_scale_dev ? out_scale / *grad_scale_dev : out_scale;
    dw[((size_t)co * ci_real + ci) * 9 + tap] = s * sc;
}

}  // namespace nastar
