// sqrt_check.hip -- is v_sqrt_f32 correctly rounded on exactly representable INTEGER arguments (what get_heuristic feeds it: dr^2 + dc^2)?
// Exhaustive over n = 0 .. 2^24 against the host's sqrtf (correctly rounded); prints the first mismatches and the count per power-of-two range,
// then the same for the corrected form (one FMA residual test against each neighbour, as LLVM's IEEE lowering does).
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 -o sqrt_check sqrt_check.hip && ./sqrt_check
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>

__device__ __forceinline__ float sqrt_rn_fixup(float x)
{
    float s = __builtin_amdgcn_sqrtf(x);
    if (x > 0.f) {
        const float sd = __uint_as_float(__float_as_uint(s) - 1u), su = __uint_as_float(__float_as_uint(s) + 1u);
        const float vp = __builtin_fmaf(-sd, s, x), vs = __builtin_fmaf(-su, s, x);
        if (vp <= 0.f) s = sd;
        if (vs > 0.f) s = su;
    }
    return s;
}

__global__ void k(uint32_t* raw, uint32_t* fixed, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        raw[i] = __float_as_uint(__builtin_amdgcn_sqrtf((float)i));
        fixed[i] = __float_as_uint(sqrt_rn_fixup((float)i));
    }
}

__global__ void kh(uint32_t* out_raw, uint32_t* out_fix, int R)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < R * R) {
        const float a = (float)(i / R), b = (float)(i % R);
        const float cheb = (a + b) - fminf(a, b);
        const float n = a * a + b * b;
        out_raw[i] = __float_as_uint(cheb + 0.001f * __builtin_amdgcn_sqrtf(n));
        out_fix[i] = __float_as_uint(cheb + 0.001f * sqrt_rn_fixup(n));
    }
}

static void heuristic_table()
{
    // get_heuristic's h0 = cheb + 0.001 * euc for every (|dr|, |dc|) in [0, R)^2 with the raw and the corrected square root: the smallest
    // max(|dr|, |dc|) at which the two differ = the largest map side whose searches cannot be affected
    const int R = 1024;
    uint32_t *a, *b;
    hipMalloc(&a, (size_t)R * R * 4);
    hipMalloc(&b, (size_t)R * R * 4);
    hipLaunchKernelGGL(kh, dim3((R * R + 255) / 256), dim3(256), 0, 0, a, b, R);
    std::vector<uint32_t> ra((size_t)R * R), rb((size_t)R * R);
    hipMemcpy(ra.data(), a, (size_t)R * R * 4, hipMemcpyDeviceToHost);
    hipMemcpy(rb.data(), b, (size_t)R * R * 4, hipMemcpyDeviceToHost);
    int first = -1;
    long long cnt[11] = {0};
    for (int i = 0; i < R * R; ++i) {
        const int dr = i / R, dc = i % R, m = dr > dc ? dr : dc;
        // the reference value, on the host: every op one fp32 rounding
        const float fa = (float)dr, fb = (float)dc;
        volatile float cheb = (fa + fb) - (fa < fb ? fa : fb);
        volatile float euc = sqrtf(fa * fa + fb * fb);
        volatile float t = 0.001f * euc;
        volatile float h = cheb + t;
        float hh = h;
        uint32_t ref;
        memcpy(&ref, &hh, 4);
        if (rb[i] != ref) { printf("CORRECTED h0 differs from the host at (%d, %d)\n", dr, dc); return; }
        if (ra[i] != ref) {
            if (first < 0 || m < first) first = m;
            int lg = 0;
            while ((1 << (lg + 1)) <= m) ++lg;
            cnt[lg]++;
        }
    }
    printf("h0 with raw v_sqrt_f32: smallest max(|dr|, |dc|) with a wrong h0 = %d\n", first);
    for (int lg = 0; lg < 11; ++lg)
        if (cnt[lg]) printf("  max(|dr|,|dc|) in [%d, %d): %lld of the (dr, dc) pairs give a different h0\n", 1 << lg, 1 << (lg + 1), cnt[lg]);
    printf("h0 with the corrected square root: identical to the host for all (dr, dc) in [0, %d)^2\n", R);
}

int main()
{
    heuristic_table();
    const uint32_t N = 1u << 24;
    uint32_t *d_raw, *d_fix;
    hipMalloc(&d_raw, N * 4);
    hipMalloc(&d_fix, N * 4);
    hipLaunchKernelGGL(k, dim3((N + 255) / 256), dim3(256), 0, 0, d_raw, d_fix, N);
    std::vector<uint32_t> raw(N), fix(N);
    hipMemcpy(raw.data(), d_raw, N * 4, hipMemcpyDeviceToHost);
    hipMemcpy(fix.data(), d_fix, N * 4, hipMemcpyDeviceToHost);
    long long bad_raw[25] = {0}, bad_fix = 0;
    uint32_t first_raw = 0, shown = 0;
    for (uint32_t i = 0; i < N; ++i) {
        float ref = sqrtf((float)i);
        uint32_t r;
        memcpy(&r, &ref, 4);
        if (raw[i] != r) {
            int lg = 0;
            while ((1u << (lg + 1)) <= i) ++lg;
            bad_raw[lg]++;
            if (!first_raw) first_raw = i;
            if (shown < 8) { printf("v_sqrt_f32(%u) = 0x%08x, correctly rounded 0x%08x\n", i, raw[i], r); ++shown; }
        }
        if (fix[i] != r) ++bad_fix;
    }
    printf("first integer where v_sqrt_f32 is not correctly rounded: %u\n", first_raw);
    for (int lg = 0; lg < 24; ++lg)
        if (bad_raw[lg]) printf("  [2^%d, 2^%d): %lld mismatches\n", lg, lg + 1, bad_raw[lg]);
    printf("corrected form (FMA residual test against both neighbours): %lld mismatches in [0, 2^24)\n", bad_fix);
    return 0;
}
