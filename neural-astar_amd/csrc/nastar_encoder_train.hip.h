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

#include "nastar_encoder.hip.h"

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

// sums[c][0], sums[c][1] (double, accumulated with atomics: the caller zeroes them).  u == nullptr: (sum v, sum v^2);
// otherwise (sum u*m, sum u*m*v) with the ReLU mask m = [ms[c]*v + mt[c] > 0].  256 threads = (256 / (C/8)) pixel lanes x C/8 channel groups.
template <bool kSplit>
__global__ __launch_bounds__(256) void nastar_chan_stats_kernel(const uint16_t* __restrict__ u, const uint16_t* __restrict__ v,
                                                                const float* __restrict__ ms, const float* __restrict__ mt,
                                                                double* __restrict__ sums, long long npix, int C)
{
    __shared__ double red[256][2];
    const int stride = kSplit ? 2 * C : C;
    const int CG = C >> 3;                 // 8-channel groups (C/8 <= 256 and divides 256: C in {8, 16, 32, 64, ... 2048})
    const int c8 = threadIdx.x % CG, pl = threadIdx.x / CG, NPL = 256 / CG;
    double s0[8], s1[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s0[e] = s1[e] = 0.0;
    float fs[8], ft[8];
    if (u) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            fs[e] = ms[c8 * 8 + e];
            ft[e] = mt[c8 * 8 + e];
        }
    }
    for (long long p = (long long)blockIdx.x * NPL + pl; p < npix; p += (long long)gridDim.x * NPL) {
        float x[8];
        load8<kSplit>(v, (size_t)p, stride, C, c8, x);
        if (u) {
            float d[8];
            load8<kSplit>(u, (size_t)p, stride, C, c8, d);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float dy = (fs[e] * x[e] + ft[e] > 0.f) ? d[e] : 0.f;
                s0[e] += (double)dy;
                s1[e] += (double)dy * (double)x[e];
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                s0[e] += (double)x[e];
                s1[e] += (double)x[e] * (double)x[e];
            }
        }
    }
    // reduce over the pixel lanes of the workgroup (fixed order), then one atomic pair per channel and workgroup
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        __syncthreads();
        red[threadIdx.x][0] = s0[e];
        red[threadIdx.x][1] = s1[e];
        __syncthreads();
        if (pl == 0) {
            double a0 = 0.0, a1 = 0.0;
            for (int k = 0; k < NPL; ++k) {
                a0 += red[k * CG + c8][0];
                a1 += red[k * CG + c8][1];
            }
            unsafeAtomicAdd(&sums[(size_t)(c8 * 8 + e) * 2 + 0], a0);
            unsafeAtomicAdd(&sums[(size_t)(c8 * 8 + e) * 2 + 1], a1);
        }
    }
}

// out = k1*u*[ms*v + mt > 0] + k2*v + k3, optionally ReLU'd.  u == nullptr drops the first term (forward: a = relu(k2*z + k3)).
template <bool kSplit>
__global__ __launch_bounds__(256) void nastar_chan_affine_kernel(const uint16_t* __restrict__ u, const uint16_t* __restrict__ v,
                                                                 const float* __restrict__ k1, const float* __restrict__ k2,
                                                                 const float* __restrict__ k3, const float* __restrict__ ms,
                                                                 const float* __restrict__ mt, uint16_t* __restrict__ out, long long npix,
                                                                 int C, int relu)
{
    const int stride = kSplit ? 2 * C : C;
    const int CG = C >> 3;
    const long long total = npix * CG;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % CG);
        const size_t p = (size_t)(i / CG);
        float x[8], r[8];
        load8<kSplit>(v, p, stride, C, c8, x);
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] = k2[c8 * 8 + e] * x[e] + k3[c8 * 8 + e];
        if (u) {
            float d[8];
            load8<kSplit>(u, p, stride, C, c8, d);
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (ms[c8 * 8 + e] * x[e] + mt[c8 * 8 + e] > 0.f) r[e] += k1[c8 * 8 + e] * d[e];
        }
        if (relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) r[e] = fmaxf(r[e], 0.f);
        }
        store8<kSplit>(out, p, stride, C, c8, r);
    }
}

}  // namespace nastar
