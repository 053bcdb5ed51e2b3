// nastar_encoder_co1.hip.h -- the ONE-channel closing convolution of the CNN encoders in the training step, without the matrix cores.
//
// Reference: planner/encoder.py:77 (CNN: Conv2d(256, 1, 3, padding=1)) under autograd (utils/training.py:55-61).  On the MFMA kernels a
// 1-channel output is padded to 32: the forward, the weight gradient and the input gradient of this layer each spent 31/32 of their
// matrix work on zeros (4096 maps: 1.9 + 2.2 + 2.4 ms of a 47 ms step).  All three are streams over the layer's 1 KB-per-pixel input:
//
//   forward    z[q] = bias + sum_{c,ky,kx} W[c][ky][kx] a[q + (ky-1, kx-1)][c]
//              = a per-pixel projection P[p][tap] = sum_c W[c][tap] a[p][c] (every pixel of `a` read ONCE, 9 x C multiply-adds) followed
//                by the shifted sum z[q] = bias + sum_tap P[q + off(tap)][tap] over a tensor 9/C the size (the "tap trick" of the
//                inference kernel nastar_conv3x3_final_kernel, here in fp32 on the vector ALU: exact products of fp32 weights)
//   dW         dW[c][tap] = sum_p d[p - off(tap)] a[p][c]: one pass over `a`, 9 scalars of d per pixel
//   da         da[p][c] = sum_tap d[p - off(tap)] W[c][tap]: 9 multiply-adds per element from a 4 B-per-pixel map -- never stored:
//              the BatchNorm-backward passes that consume da (nastar_chan_stats_kernel / nastar_chan_affine_kernel, kU1) form it on
//              the fly (struct U1Src below), which also removes their 4 B/element read of it.
// Thread layout as in nastar_encoder_train.hip.h: 256 threads = (256 / (C/8)) pixel lanes x C/8 eight-channel groups, 16-byte loads.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nastar {

// the 1-channel gradient map behind `da`: u[p][c] = gscale * sum_tap d[p - off(tap)] * w[c*9 + tap], zero outside the image
struct U1Src {
    const float* d;       // [B*H*W] fp32: dL/dz of the closing convolution's output
    const float* w;       // [C][3][3] fp32: its weight (torch layout [1][C][3][3])
    const float* gscale;  // device scalar S: the power of two the encoder's gradients travel multiplied by
    int H, W;
};

__device__ __forceinline__ void co1_yx(long long p, int H, int W, int& y, int& x)
{
    const unsigned q = (unsigned)(p % ((long long)H * W));
    y = (int)(q / (unsigned)W);
    x = (int)(q - (unsigned)y * (unsigned)W);
}

// the nine values d[p - off(tap)] (0 outside the image), tap = ky*3 + kx, off = ((ky-1), (kx-1))
__device__ __forceinline__ void co1_taps_minus(const float* __restrict__ d, long long p, int H, int W, float (&s)[9])
{
    int y, x;
    co1_yx(p, H, W, y, x);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int yy = y - (ky - 1), xx = x - (kx - 1);
            const bool ok = (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
            s[ky * 3 + kx] = ok ? d[p - (long long)(ky - 1) * W - (kx - 1)] : 0.f;
        }
}

// this thread's 8 channels of the weight, tap-major: wr[tap][e] = scale * w[(c8*8 + e)*9 + tap]
__device__ __forceinline__ void co1_weights(const float* __restrict__ w, int c8, float scale, float (&wr)[9][8])
{
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int t = 0; t < 9; ++t) wr[t][e] = scale * w[(c8 * 8 + e) * 9 + t];
}

__device__ __forceinline__ void u1_value(const U1Src& q, long long p, const float (&wr)[9][8], float (&u)[8])
{
    float s[9];
    co1_taps_minus(q.d, p, q.H, q.W, s);
#pragma unroll
    for (int e = 0; e < 8; ++e) u[e] = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) u[e] = __builtin_fmaf(s[t], wr[t][e], u[e]);
}

}  // namespace nastar
