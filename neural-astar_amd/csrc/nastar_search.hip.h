// nastar_search.hip.h -- what every LDS-resident search kernel shares: the node-state encoding, the parent / flag byte, and the two
// ordering points of a single-wavefront workgroup.  (Rounds 1-3 kept the round-1 17 B/cell state machine here; it was superseded by
// nastar_search_compact.hip.h in round 2 and deleted in round 4 -- git history, NOTES.md.)
//
//   g[]     fp32 g-value; the sign of infinity doubles as the node state so that ONE comparison decides a relaxation:
//           +inf = passable & never opened,  -inf = closed or obstacle,  finite = open.  "update neighbour n"
//           (differentiable_astar.py:235: (1-open)(1-hist) + open*(g > g2), times the obstacle mask :229) is exactly  g[n] > g2.
//   pdir[]  bits 0-3 parent direction code (8 = unset), bit 6 passable, bit 7 on-path
//   selection = first-index arg-min over the open list of q = fl(f / fl32(sqrt(W))) -- the reference's first arg-max of the masked
//           softmax exp(-f/sqrt(W))/sum (:55-74,:206-209) orders cells by exactly this quotient: the IEEE division merges f values one
//           ulp apart into exact ties that then resolve by flat index, so the key must be q, not f.
// Everything is fp32 with one rounding per reference op (TU compiled with -ffp-contract=off).
#pragma once
#include "nastar_device.hip.h"

namespace nastar {

constexpr uint32_t P_DIRMASK = 0x0Fu, P_PASS = 0x40u, P_PATH = 0x80u;
#define NASTAR_POS_INF (__uint_as_float(0x7f800000u))
#define NASTAR_NEG_INF (__uint_as_float(0xff800000u))

// single-wave workgroup: this is a scheduling + LDS-visibility point, not an s_barrier
__device__ __forceinline__ void wave_sync() { __syncthreads(); }

// LDS accesses of one wavefront execute in program order, so the hand-off between the lanes of a step needs no
// s_waitcnt -- only a compiler-level ordering point.
__device__ __forceinline__ void wave_order() { __builtin_amdgcn_wave_barrier(); }

// state in the HBM workspace (maps larger than LDS): all of this wave's stores / atomics have left the CU before any later load is issued.
// The stores are write-through: once vmcnt drains they are in L2.  (An agent-scope release fence would add a buffer_wbl2 of ~1.7 us per
// call for nothing: nothing here is cached dirty.)
__device__ __forceinline__ void global_step_fence()
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

}  // namespace nastar
