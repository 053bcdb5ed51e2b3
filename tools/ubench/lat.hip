// Single-wave latency micro-benchmarks for the instruction patterns of the A* search step (dev tool, not shipped).
// hipcc --offload-arch=gfx950 -O3 lat.hip -o lat && ./lat
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
#define N 2000
#define REP10(x) x x x x x x x x x x

__global__ void k_valu_dep(uint64_t* t, uint32_t* o) { uint32_t v = threadIdx.x; uint64_t a = __builtin_readcyclecounter();
  for (int i = 0; i < N; ++i) { asm volatile(REP10("v_add_u32 %0, %0, 3\n\t") : "+v"(v)); }
  uint64_t b = __builtin_readcyclecounter(); if (threadIdx.x == 0) t[0] = b - a; o[threadIdx.x] = v; }

__global__ void k_dpp(uint64_t* t, uint32_t* o) { uint32_t v = threadIdx.x * 77u; uint64_t a = __builtin_readcyclecounter();
  for (int i = 0; i < N; ++i) { asm volatile(REP10("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t") : "+v"(v)); }
  uint64_t b = __builtin_readcyclecounter(); if (threadIdx.x == 0) t[0] = b - a; o[threadIdx.x] = v; }

__global__ void k_readlane_salu(uint64_t* t, uint32_t* o) { uint32_t v = threadIdx.x; uint32_t s = 0; uint64_t a = __builtin_readcyclecounter();
  for (int i = 0; i < N; ++i) { asm volatile(REP10("v_readlane_b32 %1, %0, 5\n\ts_add_u32 %1, %1, 1\n\tv_add_u32 %0, %0, %1\n\t") : "+v"(v), "+s"(s)); }
  uint64_t b = __builtin_readcyclecounter(); if (threadIdx.x == 0) t[0] = b - a; o[threadIdx.x] = v + s; }

__global__ void k_cmp_ff1(uint64_t* t, uint32_t* o) { uint32_t v = threadIdx.x; uint32_t s = 7; uint64_t a = __builtin_readcyclecounter();
  for (int i = 0; i < N; ++i) { asm volatile(REP10("v_cmp_eq_u32 vcc, %1, %0\n\ts_ff1_i32_b64 %1, vcc\n\t") : "+v"(v), "+s"(s) :: "vcc"); }
  uint64_t b = __builtin_readcyclecounter(); if (threadIdx.x == 0) t[0] = b - a; o[threadIdx.x] = v + s; }

__global__ void k_gpridx(uint64_t* t, uint32_t* o, int idx) { u32x16 vec; for (int i = 0; i < 16; ++i) vec[i] = threadIdx.x + i; uint32_t v = 0;
  int s = __builtin_amdgcn_readfirstlane(idx); uint64_t a = __builtin_readcyclecounter();
  for (int i = 0; i < N; ++i) {
#pragma unroll
    for (int j = 0; j < 10; ++j) { uint32_t x = vec[s]; v += x; s = (s + 1) & 15; asm volatile("" : "+v"(v), "+s"(s)); }
  }
  uint64_t b = __builtin_readcyclecounter(); if (threadIdx.x == 0) t[0] = b - a; o[threadIdx.x] = v; }

__global__ void k_gpridx_w(uint64_t* t, uint32_t* o, int idx) { u32x16 vec; for (int i = 0; i < 16; ++i) vec[i] = threadIdx.x + i; uint32_t v = 1;
  int s = __builtin_amdgcn_readfirstlane(idx); uint64_t a = __builtin_readcyclecounter();
  for (int i = 0; i < N; ++i) {
#pragma unroll
    for (int j = 0; j < 10; ++j) { vec[s] = v; v += 3; s = (s + 1) & 15; asm volatile("" : "+v"(v), "+s"(s)); }
  }
  uint64_t b = __builtin_readcyclecounter(); if (threadIdx.x == 0) t[0] = b - a; uint32_t acc = 0; for (int i = 0; i < 16; ++i) acc += vec[i]; o[threadIdx.x] = acc; }

__global__ void k_lds_dep(uint64_t* t, uint32_t* o) { __shared__ uint32_t sm[1024]; for (int i = threadIdx.x; i < 1024; i += 64) sm[i] = (i * 37 + 11) & 1023; __syncthreads();
  uint32_t v = threadIdx.x; uint64_t a = __builtin_readcyclecounter();
  for (int i = 0; i < N; ++i) {
#pragma unroll
    for (int j = 0; j < 10; ++j) v = sm[v];
  }
  uint64_t b = __builtin_readcyclecounter(); if (threadIdx.x == 0) t[0] = b - a; o[threadIdx.x] = v; }

__global__ void k_lds_wr(uint64_t* t, uint32_t* o) { __shared__ uint32_t sm[1024]; for (int i = threadIdx.x; i < 1024; i += 64) sm[i] = (i * 37 + 11) & 1023; __syncthreads();
  uint32_t v = threadIdx.x; uint64_t a = __builtin_readcyclecounter();
  for (int i = 0; i < N; ++i) {
#pragma unroll
    for (int j = 0; j < 10; ++j) { sm[(v + 64) & 1023] = v; v = sm[v]; }
  }
  uint64_t b = __builtin_readcyclecounter(); if (threadIdx.x == 0) t[0] = b - a; o[threadIdx.x] = v; }

__global__ void k_div(uint64_t* t, float* o, float d) { float v = 1000.0f + threadIdx.x; uint64_t a = __builtin_readcyclecounter();
  for (int i = 0; i < N; ++i) {
#pragma unroll
    for (int j = 0; j < 10; ++j) { v = v / d + 7.0f; }
  }
  uint64_t b = __builtin_readcyclecounter(); if (threadIdx.x == 0) t[0] = b - a; o[threadIdx.x] = v; }

__global__ void k_salu_dep(uint64_t* t, uint32_t* o, uint32_t x) { uint32_t s = __builtin_amdgcn_readfirstlane(x); uint64_t a = __builtin_readcyclecounter();
  for (int i = 0; i < N; ++i) { asm volatile(REP10("s_add_u32 %0, %0, 3\n\t") : "+s"(s)); }
  uint64_t b = __builtin_readcyclecounter(); if (threadIdx.x == 0) t[0] = b - a; o[threadIdx.x] = s; }

__global__ void k_valu_indep(uint64_t* t, uint32_t* o) { uint32_t v0 = threadIdx.x, v1 = 1, v2 = 2, v3 = 3, v4 = 4; uint64_t a = __builtin_readcyclecounter();
  for (int i = 0; i < N; ++i) { asm volatile("v_add_u32 %0, %0, 3\n\tv_add_u32 %1, %1, 3\n\tv_add_u32 %2, %2, 3\n\tv_add_u32 %3, %3, 3\n\tv_add_u32 %4, %4, 3\n\t"
     "v_add_u32 %0, %0, 3\n\tv_add_u32 %1, %1, 3\n\tv_add_u32 %2, %2, 3\n\tv_add_u32 %3, %3, 3\n\tv_add_u32 %4, %4, 3\n\t" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4)); }
  uint64_t b = __builtin_readcyclecounter(); if (threadIdx.x == 0) t[0] = b - a; o[threadIdx.x] = v0 + v1 + v2 + v3 + v4; }

__global__ void k_vcmp_sbranch(uint64_t* t, uint32_t* o) { uint32_t v = threadIdx.x; uint32_t acc = 0; uint64_t a = __builtin_readcyclecounter();
  for (int i = 0; i < N * 10; ++i) { if (__builtin_amdgcn_readfirstlane(v) == 0xffffffffu) break; v += 1; acc += v; asm volatile("" : "+v"(v)); }
  uint64_t b = __builtin_readcyclecounter(); if (threadIdx.x == 0) t[0] = b - a; o[threadIdx.x] = acc; }

int main() {
  uint64_t* t; uint32_t* o; hipMalloc(&t, 8); hipMalloc(&o, 4096);
  auto run = [&](const char* name, auto launch, double per) { uint64_t h = 0; launch(); launch(); hipDeviceSynchronize(); hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
    printf("%-28s %8.2f cycles per unit (%s)\n", name, (double)h / (N * 10.0), "s_memtime ticks"); (void)per; };
  run("valu dependent add", [&] { hipLaunchKernelGGL(k_valu_dep, 1, 64, 0, 0, t, o); }, 1);
  run("valu independent add x5", [&] { hipLaunchKernelGGL(k_valu_indep, 1, 64, 0, 0, t, o); }, 1);
  run("salu dependent add", [&] { hipLaunchKernelGGL(k_salu_dep, 1, 64, 0, 0, t, o, 5u); }, 1);
  run("s_nop1 + v_min_dpp dep", [&] { hipLaunchKernelGGL(k_dpp, 1, 64, 0, 0, t, o); }, 1);
  run("readlane+s_add+v_add dep", [&] { hipLaunchKernelGGL(k_readlane_salu, 1, 64, 0, 0, t, o); }, 1);
  run("v_cmp + s_ff1 dep", [&] { hipLaunchKernelGGL(k_cmp_ff1, 1, 64, 0, 0, t, o); }, 1);
  run("gpr_idx read (+2 alu)", [&] { hipLaunchKernelGGL(k_gpridx, 1, 64, 0, 0, t, o, 3); }, 1);
  run("gpr_idx write (+2 alu)", [&] { hipLaunchKernelGGL(k_gpridx_w, 1, 64, 0, 0, t, o, 3); }, 1);
  run("lds read dependent", [&] { hipLaunchKernelGGL(k_lds_dep, 1, 64, 0, 0, t, o); }, 1);
  run("lds write+read dependent", [&] { hipLaunchKernelGGL(k_lds_wr, 1, 64, 0, 0, t, o); }, 1);
  run("ieee div + add dep", [&] { hipLaunchKernelGGL(k_div, 1, 64, 0, 0, t, (float*)o, 5.656854f); }, 1);
  run("readfirstlane+cmp+branch loop", [&] { hipLaunchKernelGGL(k_vcmp_sbranch, 1, 64, 0, 0, t, o); }, 1);
  return 0;
}
