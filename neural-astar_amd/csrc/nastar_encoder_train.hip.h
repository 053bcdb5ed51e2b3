// nastar_encoder_train.hip.h -- the streaming (HBM-bound) kernels of the encoder's TRAINING step: batch-statistics BatchNorm + ReLU
// forward and backward (reference planner/encoder.py:60-78 under autograd, utils/training.py:55-61) on NHWC fp16 activations, plain or
// split ([hi(C) | lo(C)] per pixel; a value is hi + lo, exact in fp32).  The convolutions around them are nastar_conv_flat.hip.h
// (forward, input gradient) and nastar_conv_wgrad.hip.h (weight gradient).
//
//   nastar_chan_stats_kernel    per-channel sums over all pixels, double accumulation:
//                                 forward:   (sum z, sum z^2)                                  -> batch mean / variance
//                                 backward:  (sum dy, sum dy*z),  dy = da * [ms*z + mt > 0]    -> dgamma, dbeta and the BN-backward means
//   nastar_chan_affine_kernel   forward:   a  = relu(k2*z + k3)                                (BatchNorm folded to scale/shift, ReLU)
//                               backward:  dz = k1*da*[ms*z + mt > 0] + k2*z + k3               (ReLU mask + BatchNorm backward, closed form)
// Per-channel coefficient vectors (k1, k2, k3, ms, mt: fp32 [C]) are computed by the host side from the sums (a handful of [C]-sized
// device ops, no host sync).  Every tensor is read once per kernel with 16-byte accesses.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nastar_device.hip.h"
#include "nastar_encoder.hip.h"
#include "nastar_encoder_co1.hip.h"

namespace nastar {

template <bool kSplit>
__device__ __forceinline__ void load8(const uint16_t* base, size_t pix, int stride, int C, int c8, float (&v)[8])
{
    const nastar_f16x8 hi = *reinterpret_cast<const nastar_f16x8*>(base + pix * stride + c8 * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (float)hi[e];
    if constexpr (kSplit) {
        const nastar_f16x8 lo = *reinterpret_cast<const nastar_f16x8*>(base + pix * stride + C + c8 * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += (float)lo[e];
    }
}

template <bool kSplit>
__device__ __forceinline__ void store8(uint16_t* base, size_t pix, int stride, int C, int c8, const float (&v)[8])
{
    nastar_f16x8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = f16_clamp(v[e]);
        hi[e] = (_Float16)x;
        lo[e] = (_Float16)(x - (float)hi[e]);
    }
    *reinterpret_cast<nastar_f16x8*>(base + pix * stride + c8 * 8) = hi;
    if constexpr (kSplit) *reinterpret_cast<nastar_f16x8*>(base + pix * stride + C + c8 * 8) = lo;
}

// sums[c][0], sums[c][1] (double, accumulated with atomics: the caller zeroes them); amax_bits (optional): max |u*m| as float bits.  u == nullptr: (sum v, sum v^2);
// otherwise (sum u*m, sum u*m*v) with the ReLU mask m = [ms[c]*v + mt[c] > 0].  256 threads = (256 / (C/8)) pixel lanes x C/8 channel groups.
// part != nullptr (two-stage, deterministic form): instead of the atomics every workgroup writes its partial sums to part[blockIdx.x][C][2]
// and its partial maximum to amax_part[blockIdx.x]; nastar_chan_stats_finish_kernel adds them in a fixed order.  No zero-fill launches, no
// contended fp64 atomics (512 workgroups x 512 addresses at 4096 maps), and any number of workgroups: small batches get enough of them.
//
// Tried in round 3 and dropped: ONE launch for statistics + finish + BatchNorm coefficients (the workgroups that arrive last, counted with
// device atomics over two levels, add the partial rows and compute the coefficients).  With agent-scope fences every workgroup's
// release / acquire wrote back and invalidated its XCD's L2 under the workgroups still streaming z (CNN step 2.0 -> 2.65 ms); with
// fence-free agent-scope atomics for the few values that cross workgroups the tail's serial memory round trips cost more than the two
// small launches they replaced (per BatchNorm pass at 100 maps: 27.7 / 38.2 us forward / backward against 31.9 / 32.5 in three launches;
// U-Net 19.0 / 24.4 against 17.2 / 16.9; profiles/r03/census_one_launch_bn_rejected.txt): back-to-back launches in one stream have no gap
// on this GPU, so merging launches only pays when it removes work.
__device__ __forceinline__ float block_max_256(float v, float* red);
__device__ __forceinline__ float pow2_scale(float amax, float target, int lo, int hi);

// kU1: u is not a tensor but the input gradient of the 1-channel closing convolution, formed on the fly (nastar_encoder_co1.hip.h)
template <bool kSplit, bool kU1 = false>
__global__ __launch_bounds__(256) void nastar_chan_stats_kernel(const uint16_t* __restrict__ u, const uint16_t* __restrict__ v,
                                                                const float* __restrict__ ms, const float* __restrict__ mt,
                                                                double* __restrict__ sums, unsigned int* __restrict__ amax_bits,
                                                                long long npix, int C, double* __restrict__ part = nullptr,
                                                                float* __restrict__ amax_part = nullptr, const U1Src u1 = U1Src())
{
    __shared__ double red[256][16];
    const int stride = kSplit ? 2 * C : C;
    const int CG = C >> 3;                 // 8-channel groups (C/8 <= 256 and divides 256: C in {8, 16, 32, 64, ... 2048})
    const int c8 = threadIdx.x % CG, pl = threadIdx.x / CG, NPL = 256 / CG;
    double s0[8], s1[8];
    float amax = 0.f;  // max |u * mask| seen by this thread (backward form only)
#pragma unroll
    for (int e = 0; e < 8; ++e) s0[e] = s1[e] = 0.0;
    float fs[8], ft[8];
    if (u || kU1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            fs[e] = ms[c8 * 8 + e];
            ft[e] = mt[c8 * 8 + e];
        }
    }
    float wr[kU1 ? 9 : 1][8];
    if constexpr (kU1) co1_weights(u1.w, c8, u1.gscale[0], wr);
    // two pixels per iteration: their loads are independent (2-4 sixteen-byte loads in flight per thread)
    const long long step = (long long)gridDim.x * NPL;
    for (long long p = (long long)blockIdx.x * NPL + pl; p < npix; p += 2 * step) {
        const bool two = p + step < npix;
        const size_t p2 = (size_t)(two ? p + step : p);
        float x[8], y[8];
        load8<kSplit>(v, (size_t)p, stride, C, c8, x);
        load8<kSplit>(v, p2, stride, C, c8, y);
        if (u || kU1) {
            float d[8], f[8];
            if constexpr (kU1) {
                u1_value(u1, p, wr, d);
                u1_value(u1, (long long)p2, wr, f);
            } else {
                load8<kSplit>(u, (size_t)p, stride, C, c8, d);
                load8<kSplit>(u, p2, stride, C, c8, f);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float dy = (fs[e] * x[e] + ft[e] > 0.f) ? d[e] : 0.f;
                const float fy = (two && fs[e] * y[e] + ft[e] > 0.f) ? f[e] : 0.f;
                amax = fmaxf(amax, fmaxf(fabsf(dy), fabsf(fy)));
                s0[e] += (double)dy;
                s1[e] += (double)dy * (double)x[e];
                s0[e] += (double)fy;
                s1[e] += (double)fy * (double)y[e];
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float yy = two ? y[e] : 0.f;
                s0[e] += (double)x[e];
                s1[e] += (double)x[e] * (double)x[e];
                s0[e] += (double)yy;
                s1[e] += (double)yy * (double)yy;
            }
        }
    }
    __shared__ float wmax[4];
    if (amax_bits || amax_part) {  // non-negative floats order like their bit patterns
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
        if (amax_part) {
            if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = amax;
        } else if ((threadIdx.x & 63) == 0) {
            atomicMax(amax_bits, __float_as_uint(amax));
        }
    }
    // reduce over the pixel lanes of the workgroup (fixed order): all 16 partial sums of a thread go to LDS at once, then thread
    // (c8, e) adds the NPL pixel lanes of "its" channel; one atomic pair per channel and workgroup
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        red[threadIdx.x][2 * e] = s0[e];
        red[threadIdx.x][2 * e + 1] = s1[e];
    }
    __syncthreads();
    for (int o = threadIdx.x; o < CG * 16; o += 256) {  // o = c8 * 16 + (2 e + which)
        const int oc8 = o >> 4, w = o & 15;
        double acc = 0.0;
        for (int k = 0; k < NPL; ++k) acc += red[k * CG + oc8][w];
        const size_t o2 = (size_t)(oc8 * 8 + (w >> 1)) * 2 + (w & 1);
        if (part) part[(size_t)blockIdx.x * (size_t)(2 * C) + o2] = acc;
        else unsafeAtomicAdd(&sums[o2], acc);
    }
    if (amax_part && threadIdx.x == 0) amax_part[blockIdx.x] = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
}

// second stage of the two-stage statistics: sums[o] = sum over workgroups of part[b][o], 32 lanes per output in a fixed order
// (bitwise reproducible); workgroup 0 also reduces the partial maxima.  grid = ceil(2C / 8), 256 threads.
__global__ __launch_bounds__(256) void nastar_chan_stats_finish_kernel(const double* __restrict__ part, const float* __restrict__ amax_part,
                                                                       int nblk, int C2, double* __restrict__ sums, float* __restrict__ amax_out)
{
    const int o = blockIdx.x * 8 + (threadIdx.x >> 5), l = threadIdx.x & 31;
    double acc = 0.0;
    if (o < C2)
        for (int b = l; b < nblk; b += 32) acc += part[(size_t)b * (size_t)C2 + o];
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 32);
    if (o < C2 && l == 0) sums[o] = acc;
    if (amax_out && blockIdx.x == 0) {
        __shared__ float red[256];
        float m = 0.f;
        for (int b = threadIdx.x; b < nblk; b += 256) m = fmaxf(m, amax_part[b]);
        red[threadIdx.x] = m;
        __syncthreads();
        for (int sft = 128; sft > 0; sft >>= 1) {
            if ((int)threadIdx.x < sft) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + sft]);
            __syncthreads();
        }
        if (threadIdx.x == 0) amax_out[0] = red[0];
    }
}

// finish + BatchNorm coefficients in ONE launch (the per-channel arithmetic of nastar_bn_coef_{fwd,bwd}_kernel needs only that channel's two
// sums): grid = ceil(C / 4) workgroups of 256 = 8 outputs x 32 lanes as in nastar_chan_stats_finish_kernel, then lanes 0..3 turn the four
// channels' sums into coefficients.  kBwd: every workgroup reduces the partial maxima and max|gamma invstd| itself (a few KB), workgroup 0
// writes the re-centred gradient scale.  Same sums (same order) as the two-launch form; sums_out (optional) receives them.
template <bool kBwd>
__global__ __launch_bounds__(256) void nastar_bn_finish_coef_kernel(const double* __restrict__ part, const float* __restrict__ amax_part, int nblk,
                                                                    int C, double* __restrict__ sums_out, const float* __restrict__ gamma,
                                                                    const float* __restrict__ beta, double eps, double npix, double momentum,
                                                                    float* __restrict__ running_mean, float* __restrict__ running_var,
                                                                    float* __restrict__ k2, float* __restrict__ k3, double* __restrict__ mean_out,
                                                                    double* __restrict__ invstd_out, const double* __restrict__ mean,
                                                                    const double* __restrict__ invstd, const float* __restrict__ gscale_in,
                                                                    float* __restrict__ gscale_out, float* __restrict__ dgamma,
                                                                    float* __restrict__ dbeta, float* __restrict__ c1, float* __restrict__ c2,
                                                                    float* __restrict__ c3)
{
    __shared__ double fs[8];
    __shared__ float red4[4];
    const int C2 = 2 * C;
    const int o = blockIdx.x * 8 + (threadIdx.x >> 5), l = threadIdx.x & 31;
    double acc = 0.0;
    if (o < C2)
        for (int b = l; b < nblk; b += 32) acc += part[(size_t)b * (size_t)C2 + o];
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 32);
    if (l == 0) {
        fs[threadIdx.x >> 5] = acc;
        if (o < C2 && sums_out) sums_out[o] = acc;
    }
    float r = 1.f;
    double S = 1.0;
    if constexpr (kBwd) {
        float m = 0.f, kmax = 0.f;
        for (int b = threadIdx.x; b < nblk; b += 256) m = fmaxf(m, amax_part[b]);
        for (int c = threadIdx.x; c < C; c += 256) kmax = fmaxf(kmax, fabsf((float)((double)gamma[c] * invstd[c])));
        m = block_max_256(m, red4);
        __syncthreads();
        kmax = block_max_256(kmax, red4);
        S = (double)gscale_in[0];
        r = pow2_scale(2.f * kmax * m, 1024.f, -40, 40);
    }
    __syncthreads();
    const int c = blockIdx.x * 4 + (int)threadIdx.x;
    if (threadIdx.x < 4 && c < C) {
        const double s0 = fs[2 * threadIdx.x], s1 = fs[2 * threadIdx.x + 1];
        if constexpr (!kBwd) {
            const double mu = s0 / npix;
            double var = s1 / npix - mu * mu;
            var = var < 0.0 ? 0.0 : var;
            const double is = 1.0 / sqrt(var + eps);
            const double g = (double)gamma[c];
            k2[c] = (float)(g * is);
            k3[c] = (float)((double)beta[c] - mu * g * is);
            mean_out[c] = mu;
            invstd_out[c] = is;
            if (running_mean) {
                running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * mu);
                running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * var * (npix / (npix > 1.0 ? npix - 1.0 : 1.0)));
            }
        } else {
            const double sdy = s0, sdyz = s1;
            const double sdyx = (sdyz - mean[c] * sdy) * invstd[c];
            dgamma[c] = (float)(sdyx / S);
            dbeta[c] = (float)(sdy / S);
            const double k1 = (double)gamma[c] * invstd[c];
            const double m1 = sdy / npix, m2 = sdyx / npix;
            c1[c] = (float)(k1 * r);
            c2[c] = (float)(-k1 * m2 * invstd[c] * r);
            c3[c] = (float)((-k1 * m1 + k1 * m2 * mean[c] * invstd[c]) * r);
        }
    }
    if constexpr (kBwd) {
        if (blockIdx.x == 0 && threadIdx.x == 0) gscale_out[0] = (float)(S * (double)r);
    }
}

// out = k1*u*[ms*v + mt > 0] + k2*v + k3, optionally ReLU'd.  u == nullptr drops the first term (forward: a = relu(k2*z + k3)).
// A thread keeps ONE 8-channel group (its coefficients live in registers) and walks pixels: 256 threads = (256 / (C/8)) pixel lanes.
template <bool kSplit, bool kU1 = false>
__global__ __launch_bounds__(256) void nastar_chan_affine_kernel(const uint16_t* __restrict__ u, const uint16_t* __restrict__ v,
                                                                 const float* __restrict__ k1, const float* __restrict__ k2,
                                                                 const float* __restrict__ k3, const float* __restrict__ ms,
                                                                 const float* __restrict__ mt, uint16_t* __restrict__ out, long long npix,
                                                                 int C, int relu, const U1Src u1 = U1Src())
{
    const int stride = kSplit ? 2 * C : C;
    const int CG = C >> 3;
    const int c8 = threadIdx.x % CG, pl = threadIdx.x / CG, NPL = 256 / CG;
    float f1[8], f2[8], f3[8], fs[8], ft[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        f2[e] = k2[c8 * 8 + e];
        f3[e] = k3[c8 * 8 + e];
        f1[e] = (u || kU1) ? k1[c8 * 8 + e] : 0.f;
        fs[e] = (u || kU1) ? ms[c8 * 8 + e] : 0.f;
        ft[e] = (u || kU1) ? mt[c8 * 8 + e] : 0.f;
    }
    float wr[kU1 ? 9 : 1][8];
    if constexpr (kU1) co1_weights(u1.w, c8, u1.gscale[0], wr);
    for (long long p = (long long)blockIdx.x * NPL + pl; p < npix; p += (long long)gridDim.x * NPL) {
        float x[8], r[8];
        load8<kSplit>(v, (size_t)p, stride, C, c8, x);
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] = f2[e] * x[e] + f3[e];
        if (u || kU1) {
            float d[8];
            if constexpr (kU1) u1_value(u1, p, wr, d);
            else load8<kSplit>(u, (size_t)p, stride, C, c8, d);
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (fs[e] * x[e] + ft[e] > 0.f) r[e] += f1[e] * d[e];
        }
        if (relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) r[e] = fmaxf(r[e], 0.f);
        }
        store8<kSplit>(out, (size_t)p, stride, C, c8, r);
    }
}

// 2x2 max-pool backward (CNNDownSize blocks, reference encoder.py:91-95 under autograd): r [B,H,W,C] = the pool's input (post-ReLU
// activations), dp [B,H/2,W/2,C] the gradient w.r.t. its output; dr[pixel] = dp[window] where the pixel is the window's FIRST maximum
// (torch's tie rule), 0 elsewhere.  Split form: values are hi + lo, gradients are moved as (hi, lo) pairs.
template <bool kSplit>
__global__ __launch_bounds__(256) void nastar_maxpool2x2_bwd_kernel(const uint16_t* __restrict__ r, const uint16_t* __restrict__ dp,
                                                                    uint16_t* __restrict__ dr, int B, int H, int W, int C)
{
    const int Ho = H >> 1, Wo = W >> 1, CG = C >> 3;
    const int stride = kSplit ? 2 * C : C;
    const long long total = (long long)B * Ho * Wo * CG;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % CG);
        long long t = i / CG;
        const int xo = (int)(t % Wo); t /= Wo;
        const int yo = (int)(t % Ho);
        const int b = (int)(t / Ho);
        float v[4][8];
        size_t pin[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            pin[k] = ((size_t)b * H + 2 * yo + (k >> 1)) * W + 2 * xo + (k & 1);
            load8<kSplit>(r, pin[k], stride, C, c8, v[k]);
        }
        const size_t po = ((size_t)b * Ho + yo) * Wo + xo;
        const nastar_f16x8 ghi = *reinterpret_cast<const nastar_f16x8*>(dp + po * stride + c8 * 8);
        nastar_f16x8 glo = ghi;
        if constexpr (kSplit) glo = *reinterpret_cast<const nastar_f16x8*>(dp + po * stride + C + c8 * 8);
        int best[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            best[e] = 0;
#pragma unroll
            for (int k = 1; k < 4; ++k)
                if (v[k][e] > v[best[e]][e]) best[e] = k;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            nastar_f16x8 ohi, olo;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                ohi[e] = best[e] == k ? ghi[e] : (_Float16)0.f;
                olo[e] = best[e] == k ? glo[e] : (_Float16)0.f;
            }
            *reinterpret_cast<nastar_f16x8*>(dr + pin[k] * stride + c8 * 8) = ohi;
            if constexpr (kSplit) *reinterpret_cast<nastar_f16x8*>(dr + pin[k] * stride + C + c8 * 8) = olo;
        }
    }
}

// ---- U-Net decoder plumbing under autograd (reference encoder.py:37-57: nearest x2 upsampling + skip concatenation) -------------------
// forward: cat[b,y,x] = (x[b,y/2,x/2,:C1], skip[b,y,x,:C2]) materialised once for the weight gradient (the convolution itself gathers)
template <bool kSplit>
__global__ __launch_bounds__(256) void nastar_upcat_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ skip,
                                                           uint16_t* __restrict__ out, int B, int H, int W, int C1, int C2)
{
    const int C = C1 + C2, CG = C >> 3, h = H >> 1, w = W >> 1;
    const int so = kSplit ? 2 * C : C, sx = kSplit ? 2 * C1 : C1, ss = kSplit ? 2 * C2 : C2;
    const long long total = (long long)B * H * W * CG * (kSplit ? 2 : 1);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % CG);
        long long t = i / CG;
        const int half = kSplit ? (int)(t & 1) : 0;
        if (kSplit) t >>= 1;
        const int xx = (int)(t % W); t /= W;
        const int yy = (int)(t % H);
        const int b = (int)(t / H);
        const size_t po = ((size_t)b * H + yy) * W + xx;
        uint4 v;
        if (c8 * 8 < C1) v = *reinterpret_cast<const uint4*>(x + (((size_t)b * h + (yy >> 1)) * w + (xx >> 1)) * sx + half * C1 + c8 * 8);
        else v = *reinterpret_cast<const uint4*>(skip + po * ss + half * C2 + (c8 * 8 - C1));
        *reinterpret_cast<uint4*>(out + po * so + half * C + c8 * 8) = v;
    }
}

// backward: d_x[b,y,x,:] = sum over the 2x2 block of d_cat[..., :C1] (nearest-upsampling backward), d_skip = d_cat[..., C1:]
template <bool kSplit>
__global__ __launch_bounds__(256) void nastar_upcat_bwd_kernel(const uint16_t* __restrict__ dcat, uint16_t* __restrict__ dx,
                                                               uint16_t* __restrict__ dskip, int B, int H, int W, int C1, int C2)
{
    const int C = C1 + C2, h = H >> 1, w = W >> 1;
    const int so = kSplit ? 2 * C : C, sx = kSplit ? 2 * C1 : C1, ss = kSplit ? 2 * C2 : C2;
    const int G1 = C1 >> 3, G2 = C2 >> 3;
    const long long n1 = (long long)B * h * w * G1, n2 = (long long)B * H * W * G2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n1 + n2; i += (long long)gridDim.x * blockDim.x) {
        if (i < n1) {
            const int c8 = (int)(i % G1);
            long long t = i / G1;
            const int xo = (int)(t % w); t /= w;
            const int yo = (int)(t % h);
            const int b = (int)(t / h);
            float acc[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float v[8];
                load8<kSplit>(dcat, ((size_t)b * H + 2 * yo + (k >> 1)) * W + 2 * xo + (k & 1), so, C, c8, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += v[e];
            }
            store8<kSplit>(dx, ((size_t)b * h + yo) * w + xo, sx, C1, c8, acc);
        } else {
            const long long j = i - n1;
            const int c8 = (int)(j % G2);
            const size_t p = (size_t)(j / G2);
            *reinterpret_cast<uint4*>(dskip + p * ss + c8 * 8) = *reinterpret_cast<const uint4*>(dcat + p * so + C1 + c8 * 8);
            if constexpr (kSplit)
                *reinterpret_cast<uint4*>(dskip + p * ss + C2 + c8 * 8) = *reinterpret_cast<const uint4*>(dcat + p * so + C + C1 + c8 * 8);
        }
    }
}

// two gradients of the same tensor that travelled with different power-of-two scales: out = a * (So/Sa) + b * (So/Sb), So = min(Sa, Sb)
template <bool kSplit>
__global__ __launch_bounds__(256) void nastar_grad_add_kernel(const uint16_t* __restrict__ a, const float* __restrict__ sa_dev,
                                                              const uint16_t* __restrict__ b, const float* __restrict__ sb_dev,
                                                              uint16_t* __restrict__ out, float* __restrict__ so_dev, long long npix, int C)
{
    const float Sa = sa_dev[0], Sb = sb_dev[0];
    const float So = fminf(Sa, Sb);
    const float fa = So / Sa, fb = So / Sb;
    const int stride = kSplit ? 2 * C : C, CG = C >> 3;
    const long long total = npix * CG;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % CG);
        const size_t p = (size_t)(i / CG);
        float x[8], y[8];
        load8<kSplit>(a, p, stride, C, c8, x);
        load8<kSplit>(b, p, stride, C, c8, y);
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = x[e] * fa + y[e] * fb;
        store8<kSplit>(out, p, stride, C, c8, x);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) so_dev[0] = So;
}

__global__ void nastar_absmax_kernel(const float* __restrict__ d, long long n, unsigned int* __restrict__ amax_bits);

// ---- small host-replacing kernels: everything a training step needs between the big launches runs on the device, in ONE launch each,
// so that a step is ~100 launches instead of ~600 tiny framework ops (at the reference's batch of 100 maps the step is launch-bound) ------

__device__ __forceinline__ float block_max_256(float v, float* red)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float m = red[0];
    for (int k = 1; k < (int)(blockDim.x >> 6); ++k) m = fmaxf(m, red[k]);
    return m;
}

// power of two 2^floor(log2(target / amax)), clamped to [2^lo, 2^hi]; amax == 0 -> 1
__device__ __forceinline__ float pow2_scale(float amax, float target, int lo, int hi)
{
    if (!(amax > 0.f)) return 1.f;
    int e = (int)floorf(log2f(target / amax));
    e = e < lo ? lo : (e > hi ? hi : e);
    return ldexpf(1.f, e);
}

// Weight pack for nastar_conv3x3_f16 from torch's [co][ci][3][3] fp32 weight: a grid pass for max|w| (nastar_absmax_kernel, float bits in
// scal[2]) + a grid pass for the pack.  transpose_flip: the input-gradient form W'[ci][co][ky][kx] = W[co][ci][2-ky][2-kx] (logical
// cout = ci, cin = co).  scal[0] = 2^-s (folded into the conv's epilogue scale), scal[1] = 2^s; block 0 also fills scale_out / shift_out.
__global__ __launch_bounds__(256) void nastar_pack_weight_kernel(const float* __restrict__ w, int co, int ci, int transpose_flip, int split,
                                                                 float* __restrict__ scal, uint16_t* __restrict__ wpack,
                                                                 float* __restrict__ scale_out, const float* __restrict__ bias,
                                                                 float* __restrict__ shift_out)
{
    const int cout_l = transpose_flip ? ci : co, cin_l = transpose_flip ? co : ci;
    const int cin_p = (cin_l + 31) & ~31, cout_p = (cout_l + 31) & ~31;
    const int cinv = split ? 3 * cin_p : cin_p;
    const int total = 9 * cinv * cout_p;  // elements of [tap][cinv/8][cout_p][8]
    const float sc = split ? pow2_scale(scal[2], 16384.f, 0, 24) : 1.f;
    if (blockIdx.x == 0) {
        if (threadIdx.x == 0) {
            scal[0] = 1.f / sc;
            scal[1] = sc;
        }
        for (int c = threadIdx.x; c < cout_p; c += 256) {
            scale_out[c] = 1.f / sc;
            shift_out[c] = (bias && c < cout_l) ? bias[c] : 0.f;
        }
    }
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int e = i & 7;
        int r = i >> 3;
        const int n = r % cout_p; r /= cout_p;
        const int cv8 = r % (cinv >> 3);
        const int tap = r / (cinv >> 3);
        const int v = cv8 * 8 + e;
        const int seg = v / cin_p, c = v - seg * cin_p;
        float x = 0.f;
        if (n < cout_l && c < cin_l) {
            const int ky = tap / 3, kx = tap - ky * 3;
            x = transpose_flip ? w[((size_t)c * ci + n) * 9 + (2 - ky) * 3 + (2 - kx)] : w[((size_t)n * ci + c) * 9 + tap];
            x *= sc;
        }
        const _Float16 hi = (_Float16)x;
        const _Float16 val = (seg == 2) ? (_Float16)(x - (float)hi) : hi;
        wpack[i] = *reinterpret_cast<const uint16_t*>(&val);
    }
}

// Every weight pack of a training step in ONE launch (the U-Net step had 47 of them, 12 us each): table row t = {w, bias or 0, co, ci,
// transpose_flip, offset into flat16 (fp16 elements), offset into flatf (floats: scale[cout_p] then shift[cout_p]), row of `scal`}.
// grid = (tiles, tensors); scal rows hold max|w| in [2] on entry (nastar_absmax_multi_f32) and get 2^-s, 2^s in [0], [1].
// A workgroup moves one 32 x 32 x 9 tile of w through LDS: it is READ as 32 runs of 288 contiguous floats (w[row][col..col+31][0..8]) and
// WRITTEN as 16-byte chunks of 8 consecutive input channels, 32 consecutive output channels = 512 contiguous bytes per (tap, segment,
// chunk) -- the per-element form of nastar_pack_weight_kernel reads with a stride of 9 (or 9 ci) floats and stores 2 bytes per lane.
__global__ __launch_bounds__(256) void nastar_pack_weight_multi_kernel(const long long* __restrict__ table, int split, float* __restrict__ scal_all,
                                                                       uint16_t* __restrict__ flat16, float* __restrict__ flatf)
{
    __shared__ float tile[32 * 289];  // [row (co)][col (ci)][tap], rows padded to 289 words: lanes that walk the rows hit 32 different banks
    const long long* row = table + 8 * (size_t)blockIdx.y;
    const float* w = reinterpret_cast<const float*>(row[0]);
    const float* bias = reinterpret_cast<const float*>(row[1]);
    const int co = (int)row[2], ci = (int)row[3], transpose_flip = (int)row[4];
    uint16_t* wpack = flat16 + row[5];
    float* scale_out = flatf + row[6];
    float* scal = scal_all + 3 * row[7];
    const int cout_l = transpose_flip ? ci : co, cin_l = transpose_flip ? co : ci;
    const int cin_p = (cin_l + 31) & ~31, cout_p = (cout_l + 31) & ~31;
    float* shift_out = scale_out + cout_p;
    const int cinv = split ? 3 * cin_p : cin_p;
    const int rb_n = (co + 31) >> 5, cb_n = (ci + 31) >> 5;  // tiles over the SOURCE rows (co) and columns (ci)
    const float sc = split ? pow2_scale(scal[2], 16384.f, 0, 24) : 1.f;
    if (blockIdx.x == 0) {
        if (threadIdx.x == 0) {  // the forward and the input-gradient pack of one weight write the same two values
            scal[0] = 1.f / sc;
            scal[1] = sc;
        }
        for (int c = threadIdx.x; c < cout_p; c += 256) {
            scale_out[c] = 1.f / sc;
            shift_out[c] = (bias && c < cout_l) ? bias[c] : 0.f;
        }
    }
    if ((int)blockIdx.x >= rb_n * cb_n) return;
    const int r0 = ((int)blockIdx.x / cb_n) * 32, c0 = ((int)blockIdx.x % cb_n) * 32;
    for (int q = threadIdx.x; q < 32 * 288; q += 256) {
        const int r = q / 288, k = q - r * 288;  // k = col_local * 9 + tap
        const int cl = k / 9;
        float x = 0.f;
        if (r0 + r < co && c0 + cl < ci) x = w[((size_t)(r0 + r) * ci + c0) * 9 + k] * sc;
        tile[r * 289 + k] = x;
    }
    __syncthreads();
    // logical output channel n / input channel c of a tile element: forward (n, c) = (row, col); input-gradient form (n, c) = (col, row), tap flipped
    const int nseg = split ? 3 : 1;
    const int n0 = transpose_flip ? c0 : r0, cc0 = transpose_flip ? r0 : c0;
    for (int q = threadIdx.x; q < 9 * nseg * 4 * 32; q += 256) {
        const int n = q & 31;
        int t2 = q >> 5;
        const int ch = t2 & 3; t2 >>= 2;
        const int seg = t2 % nseg, tap = t2 / nseg;
        const int stap = transpose_flip ? 8 - tap : tap;
        union { uint4 v; _Float16 h[8]; } o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = ch * 8 + e;
            const float x = transpose_flip ? tile[c * 289 + n * 9 + stap] : tile[n * 289 + c * 9 + stap];
            const _Float16 hi = (_Float16)x;
            o.h[e] = (seg == 2) ? (_Float16)(x - (float)hi) : hi;
        }
        const size_t cv8 = (size_t)(seg * cin_p + cc0) / 8 + ch;
        *reinterpret_cast<uint4*>(wpack + (((size_t)tap * (cinv >> 3) + cv8) * cout_p + n0 + n) * 8) = o.v;
    }
}

// torch.optim.RMSprop's plain step (alpha, eps; no momentum, not centered, no weight decay: the reference's optimiser, utils/training.py:52-53)
// for EVERY parameter in one launch: table row t = {param, grad, square_avg, element count}; grid = (blocks per tensor, tensors).
//   square_avg = alpha * square_avg + (1 - alpha) * g * g;  param -= lr * g / (sqrt(square_avg) + eps)
__global__ __launch_bounds__(256) void nastar_rmsprop_multi_kernel(const long long* __restrict__ table, float lr, float alpha, float eps)
{
    const long long* row = table + 4 * (size_t)blockIdx.y;
    float* p = reinterpret_cast<float*>(row[0]);
    const float* g = reinterpret_cast<const float*>(row[1]);
    float* sq = reinterpret_cast<float*>(row[2]);
    const long long n = row[3];
    const float oma = 1.f - alpha;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float gi = g[i];
        const float s2 = alpha * sq[i] + oma * gi * gi;
        sq[i] = s2;
        p[i] -= lr * (gi / (sqrtf(s2) + eps));
    }
}

// forward BatchNorm coefficients from the batch sums (one workgroup): k2 = gamma * invstd, k3 = beta - mean * k2; mean / invstd kept in
// double for the backward; running statistics updated like nn.BatchNorm2d in training mode (unbiased variance, momentum).
__global__ __launch_bounds__(256) void nastar_bn_coef_fwd_kernel(const double* __restrict__ sums, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, double eps, double npix, double momentum,
                                                                 float* __restrict__ running_mean, float* __restrict__ running_var,
                                                                 float* __restrict__ k2, float* __restrict__ k3,
                                                                 double* __restrict__ mean_out, double* __restrict__ invstd_out, int C)
{
    for (int c = threadIdx.x; c < C; c += 256) {
        const double mean = sums[2 * c] / npix;
        double var = sums[2 * c + 1] / npix - mean * mean;
        var = var < 0.0 ? 0.0 : var;
        const double invstd = 1.0 / sqrt(var + eps);
        const double g = (double)gamma[c];
        k2[c] = (float)(g * invstd);
        k3[c] = (float)((double)beta[c] - mean * g * invstd);
        mean_out[c] = mean;
        invstd_out[c] = invstd;
        if (running_mean) {
            running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * mean);
            running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * var * (npix / (npix > 1.0 ? npix - 1.0 : 1.0)));
        }
    }
}

// backward BatchNorm coefficients (one workgroup): from (sum dy, sum dy z) * S_in, the forward's mean / invstd and gamma:
//   dgamma = sum dy xhat / S_in, dbeta = sum dy / S_in,  dz = c1 dy + c2 z + c3  (closed-form BatchNorm backward) times a NEW power of
// two r chosen so that |dz| <~ 2 max|k1| max|dy| lands near 2^10; gscale[0] (S) is updated to S_in * r for the next block.
__global__ __launch_bounds__(256) void nastar_bn_coef_bwd_kernel(const double* __restrict__ sums, const float* __restrict__ amax_dy,
                                                                 const double* __restrict__ mean, const double* __restrict__ invstd,
                                                                 const float* __restrict__ gamma, double npix, float* __restrict__ gscale,
                                                                 float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                 float* __restrict__ c1, float* __restrict__ c2, float* __restrict__ c3, int C,
                                                                 const float* __restrict__ gscale_in = nullptr)
{
    __shared__ float red[4];
    const double S = (double)(gscale_in ? gscale_in[0] : gscale[0]);  // gscale_in: read the scale there, leave it alone, write the new one to gscale
    float kmax = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) kmax = fmaxf(kmax, fabsf((float)((double)gamma[c] * invstd[c])));
    kmax = block_max_256(kmax, red);
    const float r = pow2_scale(2.f * kmax * amax_dy[0], 1024.f, -40, 40);
    for (int c = threadIdx.x; c < C; c += 256) {
        const double sdy = sums[2 * c], sdyz = sums[2 * c + 1];
        const double sdyx = (sdyz - mean[c] * sdy) * invstd[c];
        dgamma[c] = (float)(sdyx / S);
        dbeta[c] = (float)(sdy / S);
        const double k1 = (double)gamma[c] * invstd[c];
        const double m1 = sdy / npix, m2 = sdyx / npix;
        c1[c] = (float)(k1 * r);
        c2[c] = (float)(-k1 * m2 * invstd[c] * r);
        c3[c] = (float)((-k1 * m1 + k1 * m2 * mean[c] * invstd[c]) * r);
    }
    __syncthreads();
    if (threadIdx.x == 0) gscale[0] = (float)(S * (double)r);
}

// gradient seed: d [P] fp32 (dL/dz of the 1-channel last block) -> dzb [P][32 (x2)] fp16 with channel 0 = d * S, the rest zero;
// S = 2^floor(log2(1024 / max|d|)) is written to gscale[0].  Two launches: maximum (atomicMax on float bits), then the scatter.
__global__ __launch_bounds__(256) void nastar_absmax_kernel(const float* __restrict__ d, long long n, unsigned int* __restrict__ amax_bits)
{
    float m = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) m = fmaxf(m, fabsf(d[i]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) atomicMax(amax_bits, __float_as_uint(m));
}

// max |w| of SEVERAL tensors in one launch: table[t] = (pointer, element count); grid = (8, tensors); scal[t][2] (pre-zeroed) receives the
// maximum as float bits (non-negative floats order like their bit patterns).  One launch + one memset per training step instead of a
// memset + a reduction per convolution (5 in the CNN, 26 in the U-Net).
__global__ __launch_bounds__(256) void nastar_absmax_multi_kernel(const long long* __restrict__ table, float* __restrict__ scal)
{
    const int t = blockIdx.y;
    const float* w = reinterpret_cast<const float*>(table[2 * t]);
    const long long n = table[2 * t + 1];
    float m = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) m = fmaxf(m, fabsf(w[i]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned int*>(scal + 3 * t + 2), __float_as_uint(m));
}

template <bool kSplit>
__global__ __launch_bounds__(256) void nastar_grad_seed_kernel(const float* __restrict__ d, long long npix, const float* __restrict__ amax,
                                                               float* __restrict__ gscale, uint16_t* __restrict__ dzb)
{
    const float S = pow2_scale(amax[0], 1024.f, -60, 60);
    if (blockIdx.x == 0 && threadIdx.x == 0) gscale[0] = S;
    constexpr int CH = kSplit ? 8 : 4;  // 16-byte chunks per pixel row (32 or 64 fp16)
    const long long total = npix * CH;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long p = i / CH;
        const int c = (int)(i - p * CH);
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (c == 0 || (kSplit && c == 4)) {
            const float x = d[p] * S;
            const _Float16 hi = (_Float16)x;
            const _Float16 h = (c == 0) ? hi : (_Float16)(x - (float)hi);
            v.x = (uint32_t)(*reinterpret_cast<const uint16_t*>(&h));
        }
        *reinterpret_cast<uint4*>(dzb + i * 8) = v;
    }
}


// ---- the closing block of the CNN / CNNDownSize encoders: 1-channel BatchNorm (batch statistics) + sigmoid * const ---------------------
// (reference encoder.py:60-97 last block + :32-34).  z [n] fp32 is the raw output of the last convolution; the plain-tensor form of this
// block was ~35 framework launches per training step (var_mean, 7 elementwise passes, and autograd's ~25 for the backward).  Two launches
// each way: per-workgroup double partial sums, then a kernel in which EVERY workgroup re-adds the (<= 256) partials in a fixed order and
// applies.  Data-parallel training swaps the partials for the all-reduced global sums (nparts = 1, n_total = global element count).
constexpr int BN1_MAX_PARTS = 256;

__device__ __forceinline__ double bn1_block_sum(double v, double* red)  // 256 threads, fixed tree: bitwise reproducible
{
    red[threadIdx.x] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
}

// K = 2: part[b] = (sum z, sum z^2).   K = 3 (backward; stat = mean, invstd): with xhat = (z - mean) invstd, s = sigmoid(gamma xhat + beta),
// dy = dcost * cmul * s (1 - s):  part[b] = (sum dy, sum dy xhat, sum dcost s).
template <int K>
__global__ __launch_bounds__(256) void nastar_bn1_partial_kernel(const float* __restrict__ z, const float* __restrict__ dcost, long long n,
                                                                 const double* __restrict__ stat, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, const float* __restrict__ cmul,
                                                                 double* __restrict__ part)
{
    __shared__ double red[256];
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    float mean = 0.f, invstd = 0.f, g = 0.f, b = 0.f, c = 1.f;
    if constexpr (K == 3) {
        mean = (float)stat[0];
        invstd = (float)stat[1];
        g = gamma[0];
        b = beta[0];
        c = cmul ? cmul[0] : 1.f;
    }
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float x = z[i];
        if constexpr (K == 2) {
            a0 += (double)x;
            a1 += (double)x * (double)x;
        } else {
            const float xh = (x - mean) * invstd;
            const float sg = 1.0f / (1.0f + __expf(-(g * xh + b)));
            const float dc = dcost[i];
            const float dy = dc * c * sg * (1.0f - sg);
            a0 += (double)dy;
            a1 += (double)dy * (double)xh;
            a2 += (double)dc * (double)sg;
        }
    }
    a0 = bn1_block_sum(a0, red);
    a1 = bn1_block_sum(a1, red);
    if constexpr (K == 3) a2 = bn1_block_sum(a2, red);
    if (threadIdx.x == 0) {
        part[(size_t)blockIdx.x * K] = a0;
        part[(size_t)blockIdx.x * K + 1] = a1;
        if constexpr (K == 3) part[(size_t)blockIdx.x * K + 2] = a2;
    }
}

template <int K>
__device__ __forceinline__ void bn1_total(const double* __restrict__ part, int nparts, double* red, double (&tot)[K])
{
#pragma unroll
    for (int k = 0; k < K; ++k) tot[k] = bn1_block_sum((int)threadIdx.x < nparts ? part[(size_t)threadIdx.x * K + k] : 0.0, red);
}

// cost = cmul * sigmoid(gamma (z - mean) invstd + beta); workgroup 0 stores (mean, invstd) for the backward and updates the running statistics
__global__ __launch_bounds__(256) void nastar_bn1_sigmoid_fwd_kernel(const float* __restrict__ z, long long n, const double* __restrict__ part,
                                                                     int nparts, double n_total, const float* __restrict__ gamma,
                                                                     const float* __restrict__ beta, double eps, const float* __restrict__ cmul,
                                                                     double momentum, float* __restrict__ running_mean,
                                                                     float* __restrict__ running_var, float* __restrict__ cost,
                                                                     double* __restrict__ stat)
{
    __shared__ double red[256];
    double tot[2];
    bn1_total<2>(part, nparts, red, tot);
    const double mean = tot[0] / n_total;
    double var = tot[1] / n_total - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    const double invstd = 1.0 / sqrt(var + eps);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        stat[0] = mean;
        stat[1] = invstd;
        if (running_mean) {
            running_mean[0] = (float)((1.0 - momentum) * (double)running_mean[0] + momentum * mean);
            running_var[0] = (float)((1.0 - momentum) * (double)running_var[0] + momentum * var * (n_total / (n_total > 1.0 ? n_total - 1.0 : 1.0)));
        }
    }
    const float fm = (float)mean, fi = (float)invstd, g = gamma[0], b = beta[0], c = cmul ? cmul[0] : 1.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        cost[i] = c / (1.0f + __expf(-(g * ((z[i] - fm) * fi) + b)));
}

// dz = gamma invstd (dy - sum dy / N - xhat sum dy xhat / N); workgroup 0 writes dgamma = sum dy xhat, dbeta = sum dy, dconst = sum dcost s
__global__ __launch_bounds__(256) void nastar_bn1_sigmoid_bwd_kernel(const float* __restrict__ z, const float* __restrict__ dcost, long long n,
                                                                     const double* __restrict__ stat, const float* __restrict__ gamma,
                                                                     const float* __restrict__ beta, const float* __restrict__ cmul,
                                                                     const double* __restrict__ part, int nparts, double n_total,
                                                                     float* __restrict__ dz, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                     float* __restrict__ dconst)
{
    __shared__ double red[256];
    double tot[3];
    bn1_total<3>(part, nparts, red, tot);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        dbeta[0] = (float)tot[0];
        dgamma[0] = (float)tot[1];
        if (dconst) dconst[0] = (float)tot[2];
    }
    const float mean = (float)stat[0], invstd = (float)stat[1], g = gamma[0], b = beta[0], c = cmul ? cmul[0] : 1.f;
    const float m1 = (float)(tot[0] / n_total), m2 = (float)(tot[1] / n_total), k1 = g * invstd;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float xh = (z[i] - mean) * invstd;
        const float sg = 1.0f / (1.0f + __expf(-(g * xh + b)));
        const float dy = dcost[i] * c * sg * (1.0f - sg);
        dz[i] = k1 * (dy - m1 - xh * m2);
    }
}


// ---- the 1-channel closing convolution as streams (nastar_encoder_co1.hip.h) ------------------------------------------------------------
// projection: P[p][tap] = sum_c w[c][tap] a[p][c] for every pixel; threads (pixel lane, 8-channel group), the C/8 groups of a pixel sit in
// consecutive lanes (C/8 a power of two <= 64) and are summed by xor shuffles; lane `tap` of the group stores P[p][tap]
template <bool kSplit>
__global__ __launch_bounds__(256) void nastar_co1_proj_kernel(const uint16_t* __restrict__ a, const float* __restrict__ w, float* __restrict__ P,
                                                              long long npix, int C, const float* __restrict__ k2 = nullptr,
                                                              const float* __restrict__ k3 = nullptr)
{
    const int stride = kSplit ? 2 * C : C;
    const int CG = C >> 3;
    const int c8 = threadIdx.x % CG, pl = threadIdx.x / CG, NPL = 256 / CG;
    float wr[9][8];
    co1_weights(w, c8, 1.0f, wr);
    // k2 != nullptr: `a` holds the PRE-activation z of the block in front and the layer's input relu(k2 z + k3) is formed while loading
    float f2[8], f3[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        f2[e] = k2 ? k2[c8 * 8 + e] : 1.f;
        f3[e] = k2 ? k3[c8 * 8 + e] : 0.f;
    }
    const long long step = (long long)gridDim.x * NPL;
    const long long rounds = (npix + step - 1) / step;  // every lane runs every round: the shuffles below need the whole group
    for (long long k = 0; k < rounds; ++k) {
        const long long p = k * step + (long long)blockIdx.x * NPL + pl;
        const bool ok = p < npix;
        float x[8], s[9];
        if (ok) {
            load8<kSplit>(a, (size_t)p, stride, C, c8, x);
            if (k2) {
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = fmaxf(f2[e] * x[e] + f3[e], 0.f);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = 0.f;
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            float acc = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) acc = __builtin_fmaf(x[e], wr[t][e], acc);
            s[t] = acc;
        }
        if (CG == 32) {  // the 32 lanes of a pixel: four DPP steps inside each 16-lane row (no LDS crossbar), one exchange across the rows
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                float v = s[t];
                v += __uint_as_float(dpp_mov<DPP_QUAD_XOR1>(__float_as_uint(v)));
                v += __uint_as_float(dpp_mov<DPP_QUAD_XOR2>(__float_as_uint(v)));
                v += __uint_as_float(dpp_mov<DPP_ROW_HALF_MIRROR>(__float_as_uint(v)));
                v += __uint_as_float(dpp_mov<DPP_ROW_MIRROR>(__float_as_uint(v)));
                s[t] = v + __shfl_xor(v, 16);
            }
        } else {
            for (int off = CG >> 1; off > 0; off >>= 1) {
#pragma unroll
                for (int t = 0; t < 9; ++t) s[t] += __shfl_xor(s[t], off);
            }
        }
        if (ok) {
            if (CG >= 16) {
                float mine = s[0];
#pragma unroll
                for (int t = 1; t < 9; ++t) mine = (c8 == t) ? s[t] : mine;
                if (c8 < 9) P[p * 9 + c8] = mine;
            } else if (c8 == 0) {
#pragma unroll
                for (int t = 0; t < 9; ++t) P[p * 9 + t] = s[t];
            }
        }
    }
}

// z[q] = bias + sum_tap P[q + off(tap)][tap], zero padding
__global__ __launch_bounds__(256) void nastar_co1_shift_kernel(const float* __restrict__ P, const float* __restrict__ bias, float* __restrict__ z,
                                                               long long npix, int H, int W)
{
    const float b0 = bias ? bias[0] : 0.f;
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < npix; q += (long long)gridDim.x * 256) {
        int y, x;
        co1_yx(q, H, W, y, x);
        float acc = b0;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int yy = y + ky - 1, xx = x + kx - 1;
                if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) acc += P[(q + (long long)(ky - 1) * W + (kx - 1)) * 9 + ky * 3 + kx];
            }
        z[q] = acc;
    }
}

// weight gradient: dW[c][tap] = sum_p d[p - off(tap)] a[p][c]; per-workgroup partial rows part[blockIdx.x][C*9] (fp32), summed in a fixed
// order by nastar_co1_wgrad_finish_kernel (double accumulation) -> dw [C][9] = torch's [1][C][3][3]
template <bool kSplit>
__global__ __launch_bounds__(256) void nastar_co1_wgrad_kernel(const float* __restrict__ d, const uint16_t* __restrict__ a, float* __restrict__ part,
                                                               long long npix, int C, int H, int W, const float* __restrict__ k2 = nullptr,
                                                               const float* __restrict__ k3 = nullptr)
{
    __shared__ float co1_red[2048];  // [NPL][C]: NPL * C = 2048 whatever C
    const int stride = kSplit ? 2 * C : C;
    const int CG = C >> 3;
    const int c8 = threadIdx.x % CG, pl = threadIdx.x / CG, NPL = 256 / CG;
    float acc[9][8];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[t][e] = 0.f;
    float f2[8], f3[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        f2[e] = k2 ? k2[c8 * 8 + e] : 1.f;
        f3[e] = k2 ? k3[c8 * 8 + e] : 0.f;
    }
    for (long long p = (long long)blockIdx.x * NPL + pl; p < npix; p += (long long)gridDim.x * NPL) {
        float x[8], s[9];
        load8<kSplit>(a, (size_t)p, stride, C, c8, x);
        if (k2) {  // `a` = the pre-activation z: the layer's input is relu(k2 z + k3)
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = fmaxf(f2[e] * x[e] + f3[e], 0.f);
        }
        co1_taps_minus(d, p, H, W, s);
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[t][e] = __builtin_fmaf(s[t], x[e], acc[t][e]);
    }
    // reduce over the NPL pixel lanes, one tap at a time: red[pl][c8*8 + e]
    float* out = part + (size_t)blockIdx.x * (size_t)(C * 9);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) co1_red[pl * C + c8 * 8 + e] = acc[t][e];
        __syncthreads();
        for (int c = threadIdx.x; c < C; c += 256) {
            float sum = 0.f;
            for (int k = 0; k < NPL; ++k) sum += co1_red[k * C + c];
            out[c * 9 + t] = sum;
        }
    }
}

__global__ __launch_bounds__(256) void nastar_co1_wgrad_finish_kernel(const float* __restrict__ part, int nblk, int n, float* __restrict__ dw)
{
    const int o = blockIdx.x * 8 + (threadIdx.x >> 5), l = threadIdx.x & 31;
    double acc = 0.0;
    if (o < n)
        for (int b = l; b < nblk; b += 32) acc += (double)part[(size_t)b * (size_t)n + o];
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 32);
    if (o < n && l == 0) dw[o] = (float)acc;
}

// gradient scale of the step from max|d| alone (nastar_grad_seed_kernel without the padded fp16 tensor)
__global__ void nastar_grad_scale_kernel(const float* __restrict__ amax, float* __restrict__ gscale)
{
    if (threadIdx.x == 0) gscale[0] = pow2_scale(amax[0], 1024.f, -60, 60);
}

}  // namespace nastar
