// nastar_encoder_train_capi.hip -- C-ABI entry points of the encoder training kernels (include/nastar.h): weight gradient of a 3x3
// convolution on the fp16 MFMA (nastar_conv_wgrad.hip.h) and the streaming BatchNorm / ReLU forward + backward kernels
// (nastar_encoder_train.hip.h).  The input-gradient convolution is nastar_conv3x3_f16 with transposed, flipped weights.
#include <hip/hip_runtime.h>

#include "nastar_host.hip.h"
#include "nastar_conv_wgrad.hip.h"
#include "nastar_encoder_train.hip.h"

namespace nastar {

template <int COB, int CIB, bool kSplit>
static int launch_wgrad(WgradArgs g, hipStream_t s)
{
    auto kern = &nastar_conv3x3_wgrad_kernel<COB, CIB, kSplit>;
    constexpr int M = kSplit ? 2 : 1;
    const int RC = 64 / g.W;
    const size_t lds = (size_t)64 * (COB * 64 * M + 64) + (size_t)(RC + 2) * (g.W + 2) * (CIB * 64 * M + 64);
    int rc = ensure_lds(kern, lds);
    if (rc) return rc;
    const int tiles = (g.CO / (32 * COB)) * (g.CI / (32 * CIB));
    int nsplit = 1024 / tiles;
    if (nsplit < 1) nsplit = 1;
    if (nsplit > g.nchunk) nsplit = g.nchunk;
    g.nsplit = nsplit;
    hipLaunchKernelGGL(kern, dim3((unsigned)(nsplit * tiles)), dim3(64 * COB * CIB), lds, s, g);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    return NASTAR_OK;
}

}  // namespace nastar

using namespace nastar;

extern "C" {

int nastar_conv3x3_wgrad_f16(const uint16_t* dz, const uint16_t* a, float* dw, int B, int H, int W, int co, int ci, int split,
                             float out_scale, void* stream)
{
    if (!dz || !a || !dw) return NASTAR_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || co <= 0 || ci <= 0) return NASTAR_ERR_BAD_SHAPE;
    if (co % 32 || ci % 32 || W < 2 || W > 64 || 64 % W || H % (64 / W)) return NASTAR_ERR_UNSUPPORTED;
    if (!aligned16(dz) || !aligned16(a)) return NASTAR_ERR_BAD_SHAPE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(dw, 0, (size_t)9 * ci * co * sizeof(float), s);
    if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(dw)");
    WgradArgs g;
    g.dz = dz; g.a = a; g.dw = dw; g.out_scale = out_scale; g.B = B; g.H = H; g.W = W; g.CO = co; g.CI = ci;
    g.nchunk = (int)(((long long)B * H * W) / 64); g.nsplit = 1;
    const bool co2 = co % 64 == 0, ci2 = ci % 64 == 0;
    if (split) {
        if (co2 && ci2) return launch_wgrad<2, 2, true>(g, s);
        if (co2) return launch_wgrad<2, 1, true>(g, s);
        if (ci2) return launch_wgrad<1, 2, true>(g, s);
        return launch_wgrad<1, 1, true>(g, s);
    }
    if (co2 && ci2) return launch_wgrad<2, 2, false>(g, s);
    if (co2) return launch_wgrad<2, 1, false>(g, s);
    if (ci2) return launch_wgrad<1, 2, false>(g, s);
    return launch_wgrad<1, 1, false>(g, s);
}

int nastar_chan_stats_f16(const uint16_t* u, const uint16_t* v, const float* ms, const float* mt, double* sums, long long npix, int C,
                          int split, void* stream)
{
    if (!v || !sums || (u && (!ms || !mt))) return NASTAR_ERR_NULL;
    if (npix <= 0 || C <= 0) return NASTAR_ERR_BAD_SHAPE;
    if (C % 8 || C > 2048 || 256 % (C / 8)) return NASTAR_ERR_UNSUPPORTED;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(sums, 0, (size_t)C * 2 * sizeof(double), s);
    if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(sums)");
    const long long per = 256 / (C / 8);
    long long grid = (npix + per * 64 - 1) / (per * 64);  // ~64 pixels per pixel lane
    if (grid > 2048) grid = 2048;
    if (grid < 1) grid = 1;
    if (split) hipLaunchKernelGGL(nastar_chan_stats_kernel<true>, dim3((unsigned)grid), dim3(256), 0, s, u, v, ms, mt, sums, npix, C);
    else hipLaunchKernelGGL(nastar_chan_stats_kernel<false>, dim3((unsigned)grid), dim3(256), 0, s, u, v, ms, mt, sums, npix, C);
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    return NASTAR_OK;
}

int nastar_chan_affine_f16(const uint16_t* u, const uint16_t* v, const float* k1, const float* k2, const float* k3, const float* ms,
                           const float* mt, uint16_t* out, long long npix, int C, int relu, int split, void* stream)
{
    if (!v || !out || !k2 || !k3 || (u && (!k1 || !ms || !mt))) return NASTAR_ERR_NULL;
    if (npix <= 0 || C <= 0) return NASTAR_ERR_BAD_SHAPE;
    if (C % 8) return NASTAR_ERR_UNSUPPORTED;
    const long long total = npix * (C / 8);
    const unsigned grid = (unsigned)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (split) hipLaunchKernelGGL(nastar_chan_affine_kernel<true>, dim3(grid), dim3(256), 0, s, u, v, k1, k2, k3, ms, mt, out, npix, C, relu);
    else hipLaunchKernelGGL(nastar_chan_affine_kernel<false>, dim3(grid), dim3(256), 0, s, u, v, k1, k2, k3, ms, mt, out, npix, C, relu);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    return NASTAR_OK;
}

}  // extern "C"
