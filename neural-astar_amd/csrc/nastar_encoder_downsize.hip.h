// nastar_encoder_downsize.hip.h -- the reference's CNNDownSize encoder (planner/encoder.py:81-97; WarCraft: 96x96 RGB -> 12x12 cost,
// scripts/config/train_warcraft.yaml:6-10) in eval mode on the matrix cores, at fp32 accuracy.
//
//   x = cat(image, upsample_nearest(start + goal))                       astar.py:171-177
//   for every hidden block:  x = maxpool2x2(relu(bn(conv3x3(x))))        encoder.py:91-95
//   cost = sigmoid(bn(conv3x3(x))) * const                               encoder.py:32-34, :97
//
// The network is small (191 MFLOP per 96x96 image, 19 GFLOP per 100-image batch): it is launch- and latency-bound, not
// matrix-bound, so instead of the bf16 / split-fp16 machinery of nastar_encoder.hip.h it uses the f32-input MFMA
// v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulation: within accumulation-order noise, ~1e-6, of the fp32 torch
// module -- the north-star tolerance for float outputs -- at the fp32 vector rate).
// One kernel template, implicit GEMM with M = pixels, N = output channels, K = 9 taps x CIN:
//   workgroup = one 2-row x 16-column tile of the convolution output of one image, wavefront w = output channels [32w, 32w+32);
//   the (2+2) x (16+2) x CIN input patch sits in LDS (zero outside the image = conv2d padding); per (tap, channel pair) one MFMA:
//   A[m][k] = patch[pixel m shifted by the tap][c0 + k],  B[k][n] = W[tap][c0 + k][n]   (lane l: m = n = l % 32, k = l / 32);
//   D: lane l holds channel n = l % 32 of the pixels m = 8*(v/4) + 4*(l/32) + v%4, v = 0..15, i.e. both rows of 2 column pairs:
//   BatchNorm (folded scale/shift), ReLU and the 2x2 max-pool happen in registers, the pooled NHWC fp32 tensor is what is stored.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nastar {

typedef float floatx16 __attribute__((ext_vector_type(16)));

struct DsConvArgs {
    const float* in;     // [B, H, W, CIN] fp32 NHWC
    const float* w;      // [9][CIN][COUTP] fp32
    const float* scale;  // [COUTP] folded BatchNorm
    const float* shift;
    float* out;          // kPool: [B, H/2, W/2, COUTP] NHWC ; kFinal: [B, H, W] (channel 0, sigmoid * final_mul)
    float final_mul;
    int B, H, W;
};

constexpr int DS_TR = 2, DS_TC = 16;  // output tile: 2 rows x 16 columns = the 32 rows of one MFMA

template <int CIN, int COUTP, bool kPool, bool kFinal>
__global__ __launch_bounds__(64 * (COUTP / 32)) void nastar_conv3x3_f32mfma_kernel(const DsConvArgs a)
{
    static_assert(CIN % 2 == 0 && COUTP % 32 == 0, "K steps of 2 channels, N tiles of 32 channels");
    extern __shared__ __attribute__((aligned(16))) float patch[];  // [(DS_TR+2)][(DS_TC+2)][CIN]
    constexpr int PW = DS_TC + 2, PH = DS_TR + 2;
    const int tiles_x = (a.W + DS_TC - 1) / DS_TC, tiles_y = (a.H + DS_TR - 1) / DS_TR;
    int t = blockIdx.x;
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int y0 = ty * DS_TR, x0 = tx * DS_TC;
    const float* img = a.in + (size_t)b * a.H * a.W * CIN;
    for (int i = threadIdx.x; i < PH * PW * CIN; i += blockDim.x) {
        const int c = i % CIN, p = i / CIN;
        const int px = p % PW, py = p / PW;
        const int y = y0 + py - 1, x = x0 + px - 1;
        patch[i] = ((unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W) ? img[((size_t)y * a.W + x) * CIN + c] : 0.f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int n = wv * 32 + (lane & 31), kh = lane >> 5;
    const int m = lane & 31, my = m >> 4, mx = m & 15;
    floatx16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const float* prow = patch + ((my + tap / 3) * PW + (mx + tap % 3)) * CIN + kh;
        const float* wrow = a.w + ((size_t)tap * CIN + kh) * COUTP + n;
#pragma unroll 4
        for (int c0 = 0; c0 < CIN; c0 += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(prow[c0], wrow[(size_t)c0 * COUTP], acc, 0, 0, 0);
    }
    const float sc = a.scale[n], sh = a.shift[n];
    if constexpr (kFinal) {
        // last block: BatchNorm, no ReLU / pooling (encoder.py:97), then sigmoid * const (:32-34); channel 0 is the cost map
        if (n == 0) {
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int mm = 8 * (v / 4) + 4 * kh + (v % 4);
                const int y = y0 + (mm >> 4), x = x0 + (mm & 15);
                if (y < a.H && x < a.W) {
                    const float z = acc[v] * sc + sh;
                    a.out[((size_t)b * a.H + y) * a.W + x] = a.final_mul / (1.0f + __expf(-z));
                }
            }
        }
    } else {
        static_assert(kPool, "hidden blocks of CNNDownSize pool");
        const int Ho = a.H >> 1, Wo = a.W >> 1;
        const int yo = y0 >> 1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {  // v = 0, 2, 4, 6: columns (m, m+1) of both tile rows
            const int v = 2 * q;
            const float r0 = fmaxf(acc[v] * sc + sh, 0.f), r1 = fmaxf(acc[v + 1] * sc + sh, 0.f);
            const float r2 = fmaxf(acc[v + 8] * sc + sh, 0.f), r3 = fmaxf(acc[v + 9] * sc + sh, 0.f);
            const int mm = 8 * (v / 4) + 4 * kh + (v % 4);  // even column of tile row 0
            const int xo = (x0 + mm) >> 1;
            if (yo < Ho && xo < Wo) a.out[(((size_t)b * Ho + yo) * Wo + xo) * COUTP + n] = fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
        }
    }
}

// cat(image [B,C,H,W] NCHW, nearest-upsampled (start + goal) [B,h,w]) -> NHWC fp32 [B,H,W,CP] (astar.py:171-177); CP >= C + plus
__global__ __launch_bounds__(256) void nastar_downsize_prep_kernel(const float* __restrict__ image, const float* __restrict__ start,
                                                                   const float* __restrict__ goal, float* __restrict__ out, int B, int C,
                                                                   int H, int W, int h, int w, int plus, int CP)
{
    const long long total = (long long)B * H * W;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W), y = (int)((i / W) % H), b = (int)(i / ((long long)W * H));
        float* o = out + i * CP;
        for (int c = 0; c < CP; ++c) {
            float v = 0.f;
            if (c < C) v = image[(((size_t)b * C + c) * H + y) * W + x];
            else if (plus && c == C) {
                const int ys = (int)(((long long)y * h) / H), xs = (int)(((long long)x * w) / W);  // F.interpolate(mode="nearest")
                const size_t j = ((size_t)b * h + ys) * w + xs;
                v = start[j] + goal[j];
            }
            o[c] = v;
        }
    }
}

}  // namespace nastar
