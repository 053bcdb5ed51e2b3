// Aggregate issue RATES of one CU (round 4, dev tool): lat2.hip reports wave 0's own time, and wave 0 is the oldest wave of its SIMD,
// i.e. the arbitration winner -- it says what one chain costs, not what a CU sustains.  Here every wave of ONE workgroup (k = 4, 8,
// 16, 32 wavefronts -> 1, 2, 4, 8 per SIMD) runs the same unrolled pattern and records its own s_memtime window; the host reports
// (last end - first start) and from it instructions per cycle per SIMD (VALU / SALU) or per CU (LDS).
// hipcc --offload-arch=gfx950 -O3 rate.hip -o rate && ./rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#define N 400
#define REP10(x) x x x x x x x x x x

#define KERNEL(name, ...)                                                                                       \
    __global__ void name(uint64_t* t, uint32_t* o)                                                              \
    {                                                                                                           \
        __shared__ unsigned long long sm[8192];                                                                 \
        for (int i = threadIdx.x; i < 8192; i += blockDim.x) sm[i] = ~0ull - (unsigned long long)i;             \
        __syncthreads();                                                                                        \
        uint32_t v = threadIdx.x & 63, w = v * 3 + 1, x = v ^ 5, y = 7;                                         \
        float f0 = (float)v, f1 = 1.5f, f2 = 0.25f, f3 = 3.0f;                                                  \
        uint32_t la = (threadIdx.x * 8) & 65535, lb = ((threadIdx.x & 63) * 8) & 65535;                         \
        uint32_t q0 = 0, q1 = 0, q2 = 0, q3 = 0;                                                                \
        unsigned long long e = ((unsigned long long)v << 32) | 77u, e2 = 5;                                     \
        __syncthreads();                                                                                        \
        uint64_t a = __builtin_readcyclecounter();                                                              \
        for (int i = 0; i < N; ++i) { __VA_ARGS__ }                                                             \
        uint64_t b = __builtin_readcyclecounter();                                                              \
        if ((threadIdx.x & 63) == 0) { t[2 * (threadIdx.x >> 6)] = a; t[2 * (threadIdx.x >> 6) + 1] = b; }      \
        o[threadIdx.x] = v + w + x + y + q0 + q1 + q2 + q3 + (uint32_t)e + (uint32_t)e2 + (uint32_t)(f0 + f1 + f2 + f3) + la + lb; \
    }

// 40 VALU per iteration, 4 independent chains
KERNEL(k_vadd, asm volatile(REP10("v_add_u32 %0, %0, 3\n\tv_add_u32 %1, %1, 3\n\tv_add_u32 %2, %2, 3\n\tv_add_u32 %3, %3, 3\n\t") : "+v"(v), "+v"(w), "+v"(x), "+v"(y));)
KERNEL(k_vfma, asm volatile(REP10("v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tv_fma_f32 %2, %2, %2, %2\n\tv_fma_f32 %3, %3, %3, %3\n\t") : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));)
KERNEL(k_vdpp, asm volatile(REP10("v_min_u32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_min_u32_dpp %1, %2, %2 row_mirror row_mask:0xf bank_mask:0xf\n\tv_min_u32_dpp %2, %3, %3 row_half_mirror row_mask:0xf bank_mask:0xf\n\tv_min_u32_dpp %3, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t") : "+v"(v), "+v"(w), "+v"(x), "+v"(y));)
KERNEL(k_vsqrt, asm volatile(REP10("v_sqrt_f32 %0, %0\n\tv_sqrt_f32 %1, %1\n\tv_sqrt_f32 %2, %2\n\tv_sqrt_f32 %3, %3\n\t") : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));)
KERNEL(k_vcmp, asm volatile(REP10("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc\n\tv_cmp_lt_u32 vcc, %2, %3\n\tv_cndmask_b32 %2, %2, %3, vcc\n\t") : "+v"(v), "+v"(w), "+v"(x), "+v"(y) :: "vcc");)
// 20 SALU + 20 VALU per iteration
KERNEL(k_mix, uint32_t s = 1, s2 = 2; asm volatile(REP10("s_add_u32 %4, %4, 3\n\tv_add_u32 %0, %0, 3\n\ts_add_u32 %5, %5, 3\n\tv_add_u32 %1, %1, 3\n\t") : "+v"(v), "+v"(w), "+v"(x), "+v"(y), "+s"(s), "+s"(s2) :: "scc"); q0 += s + s2;)
// LDS: 20 full-wave ds_read_b64 per iteration (waitcnt once per 4)
KERNEL(k_ldsr64, asm volatile(REP10("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:512\n\ts_waitcnt lgkmcnt(0)\n\t") : "=&v"(e), "=&v"(e2) : "v"(la) : "memory");)
// LDS: every lane reads the SAME address (broadcast)
KERNEL(k_ldsbc, uint32_t za = 64; asm volatile(REP10("ds_read_b64 %0, %1\n\tds_read_b64 %0, %1 offset:512\n\ts_waitcnt lgkmcnt(0)\n\t") : "=&v"(e) : "v"(za) : "memory");)
// LDS: 9 lanes active only
KERNEL(k_lds9, asm volatile("s_mov_b64 exec, 0x1ff\n\t" REP10("ds_read_b64 %0, %1\n\tds_read_b64 %0, %1 offset:512\n\ts_waitcnt lgkmcnt(0)\n\t") "s_mov_b64 exec, -1\n\t" : "=&v"(e) : "v"(la) : "memory");)
// LDS: 64-bit atomic min, distinct words
KERNEL(k_ldsmin, asm volatile(REP10("ds_min_u64 %0, %1\n\tds_min_u64 %0, %1 offset:512\n\ts_waitcnt lgkmcnt(0)\n\t") :: "v"(la), "v"(e) : "memory");)
// LDS: 64-bit atomic min, 9 lanes
KERNEL(k_ldsmin9, asm volatile("s_mov_b64 exec, 0x1ff\n\t" REP10("ds_min_u64 %0, %1\n\tds_min_u64 %0, %1 offset:512\n\ts_waitcnt lgkmcnt(0)\n\t") "s_mov_b64 exec, -1\n\t" :: "v"(la), "v"(e) : "memory");)
// LDS byte store, 8 lanes
KERNEL(k_ldsw8, asm volatile("s_mov_b64 exec, 0xff\n\t" REP10("ds_write_b8 %0, %1\n\tds_write_b32 %0, %1 offset:512\n\ts_waitcnt lgkmcnt(0)\n\t") "s_mov_b64 exec, -1\n\t" :: "v"(la), "v"(v) : "memory");)

int main()
{
    uint64_t* t;
    uint32_t* o;
    hipMalloc(&t, 8 * 2 * 64);
    hipMalloc(&o, 4 * 2048);
    const int waves[5] = {1, 4, 8, 16, 32};
    auto run = [&](const char* name, auto kern, double per_iter, int div) {
        printf("%-34s", name);
        for (int k : waves) {
            if (k * 64 > 1024) {  // 32 waves: two workgroups of 16 cannot be pinned to one CU; skip
                continue;
            }
            uint64_t h[128];
            hipLaunchKernelGGL(kern, 1, 64 * k, 0, 0, t, o);
            hipLaunchKernelGGL(kern, 1, 64 * k, 0, 0, t, o);
            hipDeviceSynchronize();
            hipMemcpy(h, t, 8 * 2 * k, hipMemcpyDeviceToHost);
            uint64_t a = ~0ull, b = 0;
            for (int i = 0; i < k; ++i) { a = std::min(a, h[2 * i]); b = std::max(b, h[2 * i + 1]); }
            const double cyc = (double)(b - a);
            // instructions per cycle per unit (div = 4: per SIMD, waves spread over 4 SIMDs; div = 1: per CU)
            const int units = div == 4 ? std::min(k, 4) : 1;
            printf("  %2dw %6.3f/cyc (%5.2f cyc each)", k, k * N * per_iter / cyc / units, cyc * units / (k * N * per_iter));
        }
        printf("\n");
    };
    printf("aggregate rate of one CU, k wavefronts in one workgroup; VALU/SALU: per SIMD, LDS: per CU\n");
    run("v_add_u32 x4 chains", k_vadd, 40, 4);
    run("v_fma_f32 x4 chains", k_vfma, 40, 4);
    run("v_min_u32_dpp x4", k_vdpp, 40, 4);
    run("v_sqrt_f32 x4", k_vsqrt, 40, 4);
    run("v_cmp + v_cndmask x2", k_vcmp, 40, 4);
    run("s_add + v_add interleaved (all)", k_mix, 40, 4);
    run("ds_read_b64 64 lanes", k_ldsr64, 20, 1);
    run("ds_read_b64 broadcast", k_ldsbc, 20, 1);
    run("ds_read_b64 9 lanes", k_lds9, 20, 1);
    run("ds_min_u64 64 lanes distinct", k_ldsmin, 20, 1);
    run("ds_min_u64 9 lanes", k_ldsmin9, 20, 1);
    run("ds_write_b8/b32 8 lanes", k_ldsw8, 20, 1);
    return 0;
}
