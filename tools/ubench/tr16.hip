// tr16.hip -- what does ds_read_b64_tr_b16 (gfx950) return?  Every lane passes the address of "its" 8-byte row (4 x 16-bit); the LDS
// holds element ids, so the output shows which (row, element) each lane receives.  Build: hipcc --offload-arch=gfx950 -O2 tr16.hip -o tr16
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((__vector_size__(4 * sizeof(__fp16)))) __fp16 h4;
__global__ void k(uint16_t* out, int stride_bytes)
{
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int lane = threadIdx.x;
    auto p = reinterpret_cast<__attribute__((address_space(3))) h4*>(reinterpret_cast<uintptr_t>(lds) + lane * stride_bytes);
    h4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16(p);
    const uint16_t* u = reinterpret_cast<const uint16_t*>(&v);
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = u[j];
}
int main()
{
    uint16_t* d;
    hipMalloc(&d, 64 * 4 * 2);
    for (int stride : {8, 32, 64}) {
        k<<<1, 64>>>(d, stride);
        uint16_t h[256];
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("stride %d bytes per lane (element ids = byte offset / 2):\n", stride);
        for (int l = 0; l < 64; ++l) {
            printf("lane %2d:", l);
            for (int j = 0; j < 4; ++j) {
                const int id = h[l * 4 + j];
                const int src_lane = (id * 2) / stride, e = (id * 2 % stride) / 2;
                printf("  (row of lane %2d, elem %d)", src_lane, e);
            }
            printf("\n");
        }
    }
    return 0;
}
