// glds.hip -- semantics probe of __builtin_amdgcn_global_load_lds (gfx950, 16-byte form): where do the lanes' 16 bytes land in LDS,
// what do masked-off lanes do, does the immediate offset add?  Build: hipcc --offload-arch=gfx950 -O2 glds.hip -o glds
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void k(const uint32_t* g, uint32_t* out)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[2048];
    for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = 0xdeadbeefu;
    __syncthreads();
    const int lane = threadIdx.x;
    // lane i fetches the 16-byte chunk (63 - i) of g; lanes 8..15 are masked off
    auto lp = (__attribute__((address_space(3))) void*)(lds + 256);  // wave-uniform base: dword 256
    if (lane < 8 || lane >= 16)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + (63 - lane) * 4), lp, 16, 0, 0);
    // second wave-instruction with immediate offset 2048 bytes: chunk lane of g + 1024 dwords
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + 1024 + lane * 4), lp, 16, 2048, 0);
    __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0)
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += 64) out[i] = lds[i];
}
int main()
{
    uint32_t *g, *o;
    hipMalloc(&g, 8192 * 4);
    hipMalloc(&o, 2048 * 4);
    uint32_t h[8192];
    for (int i = 0; i < 8192; ++i) h[i] = i;
    hipMemcpy(g, h, sizeof(h), hipMemcpyHostToDevice);
    k<<<1, 64>>>(g, o);
    uint32_t r[2048];
    hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    for (int c = 0; c < 512; ++c) {
        const uint32_t* p = r + c * 4;
        if (p[0] == 0xdeadbeefu && p[1] == 0xdeadbeefu) continue;
        printf("lds chunk %3d (dword %4d): %u %u %u %u\n", c, c * 4, p[0], p[1], p[2], p[3]);
    }
    return 0;
}
