// Dev micro-benchmark: sustained v_mfma_f32_32x32x16_bf16 rate of the whole chip (what is the real ceiling of the encoder kernels?).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip && ./mfma_peak
// Variants: operand data (zeros vs random bf16), waves per SIMD (1 or 2), with/without one ds_read_b128 per MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <bool kLds>
__global__ __launch_bounds__(512) void mfma_loop(const uint4* in, float* out, int iters)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    for (int q = tid; q < 4096; q += blockDim.x) reinterpret_cast<uint4*>(smem)[q] = in[q & 1023];
    __syncthreads();
    uint4 r0 = in[tid & 1023], r1 = in[(tid + 64) & 1023], r2 = in[(tid + 128) & 1023], r3 = in[(tid + 192) & 1023];
    bf16x8 a0 = *reinterpret_cast<bf16x8*>(&r0), a1 = *reinterpret_cast<bf16x8*>(&r1);
    bf16x8 b0 = *reinterpret_cast<bf16x8*>(&r2), b1 = *reinterpret_cast<bf16x8*>(&r3);
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const uint32_t la = (uint32_t)(uintptr_t)smem + (tid & 63) * 16;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16((i & 1) ? a1 : a0, (i & 2) ? b1 : b0, acc[i], 0, 0, 0);
            if (kLds) {
                if (i == 3) asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(b0) : "v"(la));
                if (i == 7) asm volatile("ds_read_b128 %[d], %[a] offset:2048\n\ts_waitcnt lgkmcnt(0)" : [d] "=v"(b1), "+v"(b0) : [a] "v"(la));
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) out[tid] = s;
}

int main()
{
    std::vector<uint16_t> h(1024 * 8);
    uint4* din; float* dout;
    hipMalloc(&din, 1024 * 16); hipMalloc(&dout, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    for (int data = 0; data < 2; ++data) {
        for (auto& v : h) v = data ? (uint16_t)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15)) : 0;  // ~+-[0.008,0.03) or zeros
        hipMemcpy(din, h.data(), 1024 * 16, hipMemcpyHostToDevice);
        for (int threads = 256; threads <= 512; threads += 256)
            for (int lds = 0; lds < 2; ++lds) {
                auto k = lds ? mfma_loop<true> : mfma_loop<false>;
                for (int rep = 0; rep < 2; ++rep) {
                    hipEventRecord(e0);
                    hipLaunchKernelGGL(k, dim3(256), dim3(threads), 65536, 0, din, dout, iters);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                }
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double flops = 256.0 * (threads / 64) * iters * 8 * 32768.0;
                printf("data=%s waves/SIMD=%d lds_reads=%d: %.3f ms  %.0f TFLOP/s\n", data ? "random" : "zeros", threads / 256, lds, ms, flops / ms / 1e9);
            }
    }
    return 0;
}
