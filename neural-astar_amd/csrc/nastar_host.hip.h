// nastar_host.hip.h -- host-side helpers shared by the translation units of libnastar_hip.so (status codes, launch plumbing).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/nastar.h"

namespace nastar {

constexpr size_t kMaxLdsBytes = 160 * 1024;  // MI355X: 160 KiB LDS per CU, one workgroup may own all of it

// message of the last HIP failure on this thread (nastar_last_error); defined in nastar_capi.hip
extern thread_local char g_last_error[256];

inline int hip_fail(hipError_t e, const char* what)
{
    snprintf(g_last_error, sizeof(g_last_error), "%s: %s", what, hipGetErrorString(e));
    return NASTAR_ERR_HIP;
}

template <typename K>
inline int ensure_lds(K kernel, size_t bytes)
{
    if (bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLdsBytes);
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
    }
    return NASTAR_OK;
}

template <typename K, typename... A>
inline int launch(K kernel, int B, size_t lds, hipStream_t stream, const A&... args)
{
    int rc = ensure_lds(kernel, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(kernel, dim3((unsigned)B), dim3(64), lds, stream, args...);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    return NASTAR_OK;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }


}  // namespace nastar
