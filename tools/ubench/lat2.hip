// Issue / latency micro-benchmarks, round 2 (dev tool): cost of the instruction patterns of the A* step for ONE wavefront and for
// k wavefronts sharing a CU (k = 1, 4, 8, 16 -> 0.25 .. 4 per SIMD).  Every kernel runs the same unrolled pattern in all waves of
// one workgroup; wave 0 reports s_memtime ticks per pattern instance.
// hipcc --offload-arch=gfx950 -O3 lat2.hip -o lat2 && ./lat2
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define N 1000
#define REP10(x) x x x x x x x x x x

#define KERNEL(name, ...)                                                                                       \
    __global__ void name(uint64_t* t, uint32_t* o)                                                              \
    {                                                                                                           \
        __shared__ unsigned long long sm[2048];                                                                 \
        for (int i = threadIdx.x; i < 2048; i += blockDim.x) sm[i] = (unsigned long long)i * 0x100000001ull;    \
        __syncthreads();                                                                                        \
        uint32_t v = threadIdx.x & 63, w = v * 3 + 1, x = v ^ 5, y = 7;                                           \
        uint32_t s = __builtin_amdgcn_readfirstlane(v) + 3, s2 = 9;                                              \
        uint32_t la = (threadIdx.x * 8) & 16383;                                                                 \
        uint64_t a = __builtin_readcyclecounter();                                                               \
        for (int i = 0; i < N; ++i) { __VA_ARGS__ }                                                                  \
        uint64_t b = __builtin_readcyclecounter();                                                               \
        if (threadIdx.x == 0) t[0] = b - a;                                                                      \
        o[threadIdx.x] = v + w + x + y + s + s2 + (uint32_t)sm[threadIdx.x];                                      \
    }

KERNEL(k_valu_dep, asm volatile(REP10("v_add_u32 %0, %0, 3\n\t") : "+v"(v));)
KERNEL(k_valu_indep4, asm volatile(REP10("v_add_u32 %0, %0, 3\n\tv_add_u32 %1, %1, 3\n\tv_add_u32 %2, %2, 3\n\tv_add_u32 %3, %3, 3\n\t") : "+v"(v), "+v"(w), "+v"(x), "+v"(y));)
KERNEL(k_salu_dep, asm volatile(REP10("s_add_u32 %0, %0, 3\n\t") : "+s"(s) :: "scc");)
KERNEL(k_salu_indep2, asm volatile(REP10("s_add_u32 %0, %0, 3\n\ts_add_u32 %1, %1, 5\n\t") : "+s"(s), "+s"(s2) :: "scc");)
KERNEL(k_salu_cmp_csel, asm volatile(REP10("s_cmp_lg_u32 %0, 77\n\ts_cselect_b32 %0, %1, %0\n\t") : "+s"(s), "+s"(s2) :: "scc");)
KERNEL(k_salu_valu_mix, asm volatile(REP10("s_add_u32 %0, %0, 3\n\tv_add_u32 %1, %1, 3\n\t") : "+s"(s), "+v"(v) :: "scc");)
KERNEL(k_snop, asm volatile(REP10("s_nop 1\n\t"));)
KERNEL(k_dpp, asm volatile(REP10("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t") : "+v"(v));)
KERNEL(k_dpp2, asm volatile(REP10("v_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_min_u32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t") : "+v"(v), "+v"(w));)
KERNEL(k_vcmp_cnd, asm volatile(REP10("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc\n\t") : "+v"(v), "+v"(w) :: "vcc");)
KERNEL(k_vcmp_sand_cnd, asm volatile(REP10("v_cmp_lt_u32 vcc, %0, %1\n\ts_and_b64 vcc, vcc, exec\n\tv_cndmask_b32 %0, %0, %1, vcc\n\t") : "+v"(v), "+v"(w) :: "vcc", "scc");)
KERNEL(k_cmp_ff1_readlane, asm volatile(REP10("v_cmp_eq_u32 vcc, %2, %0\n\ts_ff1_i32_b64 %2, vcc\n\ts_and_b32 %2, %2, 63\n\tv_readlane_b32 %2, %1, %2\n\ts_and_b32 %2, %2, 63\n\t") : "+v"(v), "+v"(w), "+s"(s) :: "vcc", "scc");)
KERNEL(k_readlane_vadd, asm volatile(REP10("v_readlane_b32 %1, %0, 5\n\tv_add_u32 %0, %0, %1\n\t") : "+v"(v), "+s"(s));)
KERNEL(k_sqrt, asm volatile(REP10("v_sqrt_f32 %0, %0\n\t") : "+v"(v));)
KERNEL(k_branch, asm volatile(REP10("s_cmp_eq_u32 %0, 0x12345\n\ts_cbranch_scc1 1f\n\ts_add_u32 %0, %0, 1\n\ts_branch 2f\n\t1:\n\ts_add_u32 %0, %0, 2\n\t2:\n\t") : "+s"(s) :: "scc");)
KERNEL(k_lds_rd, asm volatile(REP10("ds_read_b32 %0, %0\n\ts_waitcnt lgkmcnt(0)\n\tv_and_b32 %0, 0x1ff8, %0\n\t") : "+v"(la));)
KERNEL(k_lds_rd64x2, uint32_t q0, q1; asm volatile(REP10("ds_read_b32 %1, %0\n\tds_read_b32 %2, %0 offset:512\n\ts_waitcnt lgkmcnt(0)\n\tv_xor_b32 %0, %0, %1\n\tv_and_b32 %0, 0x1ff8, %0\n\t") : "+v"(la), "=&v"(q0), "=&v"(q1)); y += q1;)
__global__ void k_lds_min64_rd(uint64_t* t, uint32_t* o)
{
    __shared__ unsigned long long sm[2048];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) sm[i] = ~0ull;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned long long* base = sm + wv * 64;
    unsigned long long e = ((unsigned long long)lane << 32) | 12345u;
    uint64_t a = __builtin_readcyclecounter();
    for (int i = 0; i < N * 10; ++i) {
        atomicMin(&base[lane], e);
        e = base[(lane + 1) & 63] - 1;
        __builtin_amdgcn_wave_barrier();
    }
    uint64_t b = __builtin_readcyclecounter();
    if (threadIdx.x == 0) t[0] = b - a;
    o[threadIdx.x] = (uint32_t)e;
}
__global__ void k_lds_wr_rd64(uint64_t* t, uint32_t* o)
{
    __shared__ unsigned long long sm[2048];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) sm[i] = ~0ull;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned long long* base = sm + wv * 64;
    unsigned long long e = ((unsigned long long)lane << 32) | 12345u;
    uint64_t a = __builtin_readcyclecounter();
    for (int i = 0; i < N * 10; ++i) {
        base[lane] = e;
        e = base[(lane + 1) & 63] - 1;
        __builtin_amdgcn_wave_barrier();
    }
    uint64_t b = __builtin_readcyclecounter();
    if (threadIdx.x == 0) t[0] = b - a;
    o[threadIdx.x] = (uint32_t)e;
}

// same-address LDS atomics: `nsame` lanes of row 1 hit one word, the other lanes are inactive; followed by a dependent read
template <int NSAME>
__global__ void k_lds_min64_same(uint64_t* t, uint32_t* o)
{
    __shared__ unsigned long long sm[2048];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) sm[i] = ~0ull;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned long long* base = sm + wv * 64;
    unsigned long long e = ((unsigned long long)(lane * 7919u) << 32) | (unsigned)lane;
    uint64_t a = __builtin_readcyclecounter();
    for (int i = 0; i < N * 10; ++i) {
        if (lane >= 16 && lane < 16 + NSAME) atomicMin(&base[5], e);
        e += base[(lane + 1) & 63] & 1;
        __builtin_amdgcn_wave_barrier();
    }
    uint64_t b = __builtin_readcyclecounter();
    if (threadIdx.x == 0) t[0] = b - a;
    o[threadIdx.x] = (uint32_t)e;
}
// loop with a taken backward branch per iteration and a one-instruction body
__global__ void k_loop(uint64_t* t, uint32_t* o, int n)
{
    uint32_t s = 0;
    uint64_t a = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) asm volatile("s_add_u32 %0, %0, 3\n\t" : "+s"(s) :: "scc");
    uint64_t b = __builtin_readcyclecounter();
    if (threadIdx.x == 0) t[0] = b - a;
    o[threadIdx.x] = s;
}

int main()
{
    uint64_t* t;
    uint32_t* o;
    hipMalloc(&t, 8);
    hipMalloc(&o, 8192);
    const int waves[4] = {1, 4, 8, 16};
    auto run = [&](const char* name, auto kern, double units) {
        printf("%-30s", name);
        fflush(stdout);
        for (int k : waves) {
            uint64_t h = 0;
            hipLaunchKernelGGL(kern, 1, 64 * k, 0, 0, t, o);
            hipLaunchKernelGGL(kern, 1, 64 * k, 0, 0, t, o);
            hipDeviceSynchronize();
            hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
            printf("  %2dw %7.2f", k, (double)h / units);
            fflush(stdout);
        }
        printf("\n");
    };
    const double U = N * 10.0;
    printf("ticks per pattern instance (one workgroup of k wavefronts on one CU)\n");
    run("valu dep add", k_valu_dep, U);
    run("valu indep x4", k_valu_indep4, U);
    run("salu dep add", k_salu_dep, U);
    run("salu indep x2", k_salu_indep2, U);
    run("s_cmp+s_cselect dep", k_salu_cmp_csel, U);
    run("s_add + v_add (indep)", k_salu_valu_mix, U);
    run("s_nop 1", k_snop, U);
    run("s_nop1+v_min_dpp dep", k_dpp, U);
    run("2 indep v_min_dpp", k_dpp2, U);
    run("v_cmp->v_cndmask(vcc) dep", k_vcmp_cnd, U);
    run("v_cmp->s_and->v_cndmask", k_vcmp_sand_cnd, U);
    run("v_cmp,s_ff1,s_and,readlane,s_and", k_cmp_ff1_readlane, U);
    run("v_readlane->v_add dep", k_readlane_vadd, U);
    run("v_sqrt dep", k_sqrt, U);
    run("cmp+branch(not taken)+jump", k_branch, U);
    run("ds_read_b32 dep (+and)", k_lds_rd, U);
    run("2x ds_read_b64 + 2 valu", k_lds_rd64x2, U);
    run("ds_min_u64 -> ds_read_b64", k_lds_min64_rd, U);
    run("ds_write_b64 -> ds_read_b64", k_lds_wr_rd64, U);
    run("ds_min_u64 1 lane  -> read", k_lds_min64_same<1>, U);
    run("ds_min_u64 4 same  -> read", k_lds_min64_same<4>, U);
    run("ds_min_u64 8 same  -> read", k_lds_min64_same<8>, U);
    run("ds_min_u64 16 same -> read", k_lds_min64_same<16>, U);
    printf("%-30s", "loop: s_add + taken branch");
    for (int k : waves) {
        uint64_t h = 0;
        hipLaunchKernelGGL(k_loop, 1, 64 * k, 0, 0, t, o, 10000);
        hipLaunchKernelGGL(k_loop, 1, 64 * k, 0, 0, t, o, 10000);
        hipDeviceSynchronize();
        hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
        printf("  %2dw %7.2f", k, (double)h / 10000.0);
    }
    printf("\n");
    return 0;
}
