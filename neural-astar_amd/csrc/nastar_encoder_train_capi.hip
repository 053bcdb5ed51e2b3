// nastar_encoder_train_capi.hip -- C-ABI entry points of the encoder training kernels (include/nastar.h): weight gradient of a 3x3
// convolution on the fp16 MFMA (nastar_conv_wgrad.hip.h) and the streaming BatchNorm / ReLU forward + backward kernels
// (nastar_encoder_train.hip.h).  The input-gradient convolution is nastar_conv3x3_f16 with transposed, flipped weights.
#include <hip/hip_runtime.h>

#include "nastar_host.hip.h"
#include "nastar_conv_wgrad.hip.h"
#include "nastar_encoder_train.hip.h"

namespace nastar {

static int wgrad_nsplit(int nchunk, int co, int ci)
{
    const int cob = co % 64 == 0 ? 2 : 1, cib = ci % 64 == 0 ? 2 : 1;
    const int tiles = (co / (32 * cob)) * (ci / (32 * cib));
    int nsplit = 256 / tiles;                      // one 6..12-wave workgroup per CU
    if (nsplit > nchunk / 4) nsplit = nchunk / 4;  // ... but at least 4 chunks of 64 pixels per workgroup
    if (nsplit < 1) nsplit = 1;
    return nsplit;
}

template <int COB, int CIB, bool kSplit>
static int launch_wgrad(WgradArgs g, hipStream_t s)
{
    void (*kern)(const WgradArgs) = &nastar_conv3x3_wgrad_kernel<COB, CIB, kSplit, false>;
    if (g.W > 64) kern = &nastar_conv3x3_wgrad_kernel<COB, CIB, kSplit, true>;
    constexpr int M = kSplit ? 2 : 1;
    const size_t lds = (size_t)g.KS * 16 * wg_row_bytes(COB * 64 * M) + (size_t)g.G * (g.R + 2) * (g.W + 2) * wg_row_bytes(CIB * 64 * M);
    if (lds > kMaxLdsBytes) return NASTAR_ERR_UNSUPPORTED;
    int rc = ensure_lds(kern, lds);
    if (rc) return rc;
    const int tiles = (g.CO / (32 * COB)) * (g.CI / (32 * CIB));
    hipLaunchKernelGGL(kern, dim3((unsigned)(g.nsplit * tiles)), dim3(192 * COB * CIB), lds, s, g);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    return NASTAR_OK;
}

}  // namespace nastar

using namespace nastar;

extern "C" {

size_t nastar_conv3x3_wgrad_workspace_bytes(int B, int H, int W, int co, int ci)
{
    if (B <= 0 || H <= 0 || W <= 0 || co <= 0 || ci <= 0 || co % 32 || ci % 32) return 0;
    const int R = nastar_wgrad_chunk_rows(H, W);
    if (R == 0) return 0;
    const int G = nastar_wgrad_chunk_images(H, W);
    const int nseg = nastar_wgrad_segments(W);
    const int nchunk = G > 1 ? (B + G - 1) / G : (int)(((long long)B * H) / R) * nseg;
    return (size_t)wgrad_nsplit(nchunk, co, ci) * 9 * ci * co * sizeof(float);
}

int nastar_conv3x3_wgrad_f16(const uint16_t* dz, const uint16_t* a, float* dw, int B, int H, int W, int co, int ci, int co_real,
                             int ci_real, int split, float out_scale, const float* grad_scale_dev, void* workspace,
                             size_t workspace_bytes, void* stream)
{
    if (!dz || !a || !dw || !workspace) return NASTAR_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || co <= 0 || ci <= 0 || co_real <= 0 || ci_real <= 0 || co_real > co || ci_real > ci) return NASTAR_ERR_BAD_SHAPE;
    const int R = nastar_wgrad_chunk_rows(H, W);
    if (co % 32 || ci % 32 || R == 0) return NASTAR_ERR_UNSUPPORTED;
    if (!aligned16(dz) || !aligned16(a) || !aligned16(workspace)) return NASTAR_ERR_BAD_SHAPE;
    if (workspace_bytes < nastar_conv3x3_wgrad_workspace_bytes(B, H, W, co, ci)) return NASTAR_ERR_WORKSPACE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    WgradArgs g;
    g.dz = dz; g.a = a; g.part = static_cast<float*>(workspace); g.B = B; g.H = H; g.CO = co; g.CI = ci;
    g.Wimg = W; g.W = nastar_wgrad_segment(W); g.nseg = nastar_wgrad_segments(W);  // images wider than 96 pixels: chunk rows are segments of an image row
    g.G = nastar_wgrad_chunk_images(H, W);
    g.R = R; g.NP = g.G * R * g.W; g.KS = (g.NP + 15) / 16;
    g.nchunk = g.G > 1 ? (B + g.G - 1) / g.G : (int)(((long long)B * H) / R) * g.nseg;
    g.nsplit = wgrad_nsplit(g.nchunk, co, ci);
    const bool co2 = co % 64 == 0, ci2 = ci % 64 == 0;
    int rc;
    if (split) {
        if (co2 && ci2) rc = launch_wgrad<2, 2, true>(g, s);
        else if (co2) rc = launch_wgrad<2, 1, true>(g, s);
        else if (ci2) rc = launch_wgrad<1, 2, true>(g, s);
        else rc = launch_wgrad<1, 1, true>(g, s);
    } else {
        if (co2 && ci2) rc = launch_wgrad<2, 2, false>(g, s);
        else if (co2) rc = launch_wgrad<2, 1, false>(g, s);
        else if (ci2) rc = launch_wgrad<1, 2, false>(g, s);
        else rc = launch_wgrad<1, 1, false>(g, s);
    }
    if (rc) return rc;
    const int total = 9 * ci * co;
    hipLaunchKernelGGL(nastar_wgrad_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, g.part, dw, g.nsplit, co, ci,
                       co_real, ci_real, out_scale, grad_scale_dev);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    return NASTAR_OK;
}

int nastar_chan_stats_f16(const uint16_t* u, const uint16_t* v, const float* ms, const float* mt, double* sums, float* amax_out,
                          long long npix, int C, int split, void* stream)
{
    if (!v || !sums || (u && (!ms || !mt))) return NASTAR_ERR_NULL;
    if (npix <= 0 || C <= 0) return NASTAR_ERR_BAD_SHAPE;
    if (C % 8 || C > 2048 || 256 % (C / 8)) return NASTAR_ERR_UNSUPPORTED;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(sums, 0, (size_t)C * 2 * sizeof(double), s);
    if (e == hipSuccess && amax_out) e = hipMemsetAsync(amax_out, 0, sizeof(float), s);
    if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(sums)");
    const long long per = 256 / (C / 8);
    long long grid = (npix + per * 16 - 1) / (per * 16);  // >= 16 pixels per pixel lane, at most 512 workgroups of double atomics
    if (grid > 512) grid = 512;
    if (grid < 1) grid = 1;
    unsigned int* ab = reinterpret_cast<unsigned int*>(amax_out);
    if (split) hipLaunchKernelGGL(nastar_chan_stats_kernel<true>, dim3((unsigned)grid), dim3(256), 0, s, u, v, ms, mt, sums, ab, npix, C);
    else hipLaunchKernelGGL(nastar_chan_stats_kernel<false>, dim3((unsigned)grid), dim3(256), 0, s, u, v, ms, mt, sums, ab, npix, C);
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    return NASTAR_OK;
}

// two-stage form: partial sums per workgroup in the caller's workspace, then a fixed-order finishing kernel (no memsets, no atomics)
static long long chan_stats_grid(long long npix, int C)
{
    const long long per = 256 / (C / 8);
    long long grid = (npix + per * 4 - 1) / (per * 4);  // >= 4 pixels (two iterations) per pixel lane ...
    if (grid > 1024) grid = 1024;                       // ... and at most 4 workgroups per CU
    if (grid < 1) grid = 1;
    return grid;
}

size_t nastar_chan_stats_workspace_bytes(long long npix, int C)
{
    if (npix <= 0 || C <= 0 || C % 8 || C > 2048 || 256 % (C / 8)) return 0;
    const long long grid = chan_stats_grid(npix, C);
    return (size_t)grid * (size_t)(2 * C) * sizeof(double) + (size_t)grid * sizeof(float);
}

int nastar_chan_stats_f16_ws(const uint16_t* u, const uint16_t* v, const float* ms, const float* mt, double* sums, float* amax_out,
                             long long npix, int C, int split, void* workspace, size_t workspace_bytes, void* stream)
{
    if (!v || !sums || !workspace || (u && (!ms || !mt))) return NASTAR_ERR_NULL;
    if (npix <= 0 || C <= 0) return NASTAR_ERR_BAD_SHAPE;
    if (C % 8 || C > 2048 || 256 % (C / 8)) return NASTAR_ERR_UNSUPPORTED;
    if (workspace_bytes < nastar_chan_stats_workspace_bytes(npix, C)) return NASTAR_ERR_WORKSPACE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const long long grid = chan_stats_grid(npix, C);
    double* part = static_cast<double*>(workspace);
    float* amax_part = reinterpret_cast<float*>(part + (size_t)grid * (size_t)(2 * C));
    if (split) hipLaunchKernelGGL(nastar_chan_stats_kernel<true>, dim3((unsigned)grid), dim3(256), 0, s, u, v, ms, mt, sums, nullptr, npix, C, part, amax_part);
    else hipLaunchKernelGGL(nastar_chan_stats_kernel<false>, dim3((unsigned)grid), dim3(256), 0, s, u, v, ms, mt, sums, nullptr, npix, C, part, amax_part);
    hipLaunchKernelGGL(nastar_chan_stats_finish_kernel, dim3((unsigned)((2 * C + 7) / 8)), dim3(256), 0, s, part, amax_part, (int)grid, 2 * C, sums,
                       amax_out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    return NASTAR_OK;
}

// statistics (partial rows) + [finish + coefficients]: two launches instead of three per BatchNorm pass
int nastar_bn_stats_coef_fwd_f16(const uint16_t* z, long long npix, int C, int split, const float* gamma, const float* beta, double eps,
                                 double momentum, float* running_mean, float* running_var, float* k2, float* k3, double* mean_out,
                                 double* invstd_out, double* sums_out, void* workspace, size_t workspace_bytes, void* stream)
{
    if (!z || !gamma || !beta || !k2 || !k3 || !mean_out || !invstd_out || !workspace || (running_mean && !running_var)) return NASTAR_ERR_NULL;
    if (npix <= 0 || C <= 0) return NASTAR_ERR_BAD_SHAPE;
    if (C % 8 || C > 2048 || 256 % (C / 8)) return NASTAR_ERR_UNSUPPORTED;
    if (workspace_bytes < nastar_chan_stats_workspace_bytes(npix, C)) return NASTAR_ERR_WORKSPACE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const long long grid = chan_stats_grid(npix, C);
    double* part = static_cast<double*>(workspace);
    if (split) hipLaunchKernelGGL(nastar_chan_stats_kernel<true>, dim3((unsigned)grid), dim3(256), 0, s, nullptr, z, nullptr, nullptr, nullptr, nullptr, npix, C, part, nullptr);
    else hipLaunchKernelGGL(nastar_chan_stats_kernel<false>, dim3((unsigned)grid), dim3(256), 0, s, nullptr, z, nullptr, nullptr, nullptr, nullptr, npix, C, part, nullptr);
    hipLaunchKernelGGL(nastar_bn_finish_coef_kernel<false>, dim3((unsigned)((C + 3) / 4)), dim3(256), 0, s, part, nullptr, (int)grid, C, sums_out, gamma,
                       beta, eps, (double)npix, momentum, running_mean, running_var, k2, k3, mean_out, invstd_out, nullptr, nullptr, nullptr, nullptr,
                       nullptr, nullptr, nullptr, nullptr, nullptr);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    return NASTAR_OK;
}

int nastar_bn_stats_coef_bwd_f16(const uint16_t* da, const uint16_t* z, const float* ms, const float* mt, long long npix, int C, int split,
                                 const double* mean, const double* invstd, const float* gamma, const float* gscale_in, float* gscale_out,
                                 float* dgamma, float* dbeta, float* c1, float* c2, float* c3, double* sums_out, void* workspace,
                                 size_t workspace_bytes, void* stream)
{
    if (!da || !z || !ms || !mt || !mean || !invstd || !gamma || !gscale_in || !gscale_out || !dgamma || !dbeta || !c1 || !c2 || !c3 || !workspace)
        return NASTAR_ERR_NULL;
    if (npix <= 0 || C <= 0 || gscale_in == gscale_out) return NASTAR_ERR_BAD_SHAPE;  // every workgroup of the finishing kernel reads gscale_in
    if (C % 8 || C > 2048 || 256 % (C / 8)) return NASTAR_ERR_UNSUPPORTED;
    if (workspace_bytes < nastar_chan_stats_workspace_bytes(npix, C)) return NASTAR_ERR_WORKSPACE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const long long grid = chan_stats_grid(npix, C);
    double* part = static_cast<double*>(workspace);
    float* amax_part = reinterpret_cast<float*>(part + (size_t)grid * (size_t)(2 * C));
    if (split) hipLaunchKernelGGL(nastar_chan_stats_kernel<true>, dim3((unsigned)grid), dim3(256), 0, s, da, z, ms, mt, nullptr, nullptr, npix, C, part, amax_part);
    else hipLaunchKernelGGL(nastar_chan_stats_kernel<false>, dim3((unsigned)grid), dim3(256), 0, s, da, z, ms, mt, nullptr, nullptr, npix, C, part, amax_part);
    hipLaunchKernelGGL(nastar_bn_finish_coef_kernel<true>, dim3((unsigned)((C + 3) / 4)), dim3(256), 0, s, part, amax_part, (int)grid, C, sums_out, gamma,
                       nullptr, 0.0, (double)npix, 0.0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, mean, invstd, gscale_in, gscale_out, dgamma,
                       dbeta, c1, c2, c3);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    return NASTAR_OK;
}

int nastar_chan_affine_f16(const uint16_t* u, const uint16_t* v, const float* k1, const float* k2, const float* k3, const float* ms,
                           const float* mt, uint16_t* out, long long npix, int C, int relu, int split, void* stream)
{
    if (!v || !out || !k2 || !k3 || (u && (!k1 || !ms || !mt))) return NASTAR_ERR_NULL;
    if (npix <= 0 || C <= 0) return NASTAR_ERR_BAD_SHAPE;
    if (C % 8 || C > 2048 || 256 % (C / 8)) return NASTAR_ERR_UNSUPPORTED;
    const long long per = 256 / (C / 8);
    long long grid = (npix + per * 16 - 1) / (per * 16);  // ~16 pixels per pixel lane ...
    if (grid < 1024) grid = (npix + per * 2 - 1) / (per * 2) < 1024 ? (npix + per * 2 - 1) / (per * 2) : 1024;  // ... but fill the chip at small batches
    if (grid > 8192) grid = 8192;
    if (grid < 1) grid = 1;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (split) hipLaunchKernelGGL(nastar_chan_affine_kernel<true>, dim3((unsigned)grid), dim3(256), 0, s, u, v, k1, k2, k3, ms, mt, out, npix, C, relu);
    else hipLaunchKernelGGL(nastar_chan_affine_kernel<false>, dim3((unsigned)grid), dim3(256), 0, s, u, v, k1, k2, k3, ms, mt, out, npix, C, relu);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    return NASTAR_OK;
}

int nastar_pack_conv_weight_f16(const float* w, int co, int ci, int transpose_flip, int split, const float* bias, uint16_t* wpack,
                                float* scale_out, float* shift_out, float* scal_out, int reuse_max, void* stream)
{
    if (!w || !wpack || !scale_out || !shift_out || !scal_out) return NASTAR_ERR_NULL;
    if (co <= 0 || ci <= 0) return NASTAR_ERR_BAD_SHAPE;
    const int cout_l = transpose_flip ? ci : co, cin_l = transpose_flip ? co : ci;
    const int cin_p = (cin_l + 31) & ~31, cout_p = (cout_l + 31) & ~31;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipError_t e = hipSuccess;
    const int n = co * ci * 9;
    if (split && !reuse_max) {  // reuse_max: scal_out[2] already holds max|w| (the forward pack of the same weight computed it)
        e = hipMemsetAsync(scal_out + 2, 0, sizeof(float), s);
        if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync");
    }
    if (split && !reuse_max)
        hipLaunchKernelGGL(nastar_absmax_kernel, dim3((unsigned)((n + 255) / 256 < 256 ? (n + 255) / 256 : 256)), dim3(256), 0, s, w, (long long)n,
                           reinterpret_cast<unsigned int*>(scal_out + 2));
    const int total = 9 * (split ? 3 : 1) * cin_p * cout_p;
    hipLaunchKernelGGL(nastar_pack_weight_kernel, dim3((unsigned)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024)), dim3(256), 0, s, w,
                       co, ci, transpose_flip, split, scal_out, wpack, scale_out, bias, shift_out);
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    return NASTAR_OK;
}

int nastar_absmax_multi_f32(const long long* table, int n, float* scal, void* stream)
{
    if (!table || !scal) return NASTAR_ERR_NULL;
    if (n <= 0 || n > 65535) return NASTAR_ERR_BAD_SHAPE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(scal, 0, (size_t)n * 3 * sizeof(float), s);
    if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync");
    hipLaunchKernelGGL(nastar_absmax_multi_kernel, dim3(64, (unsigned)n), dim3(256), 0, s, table, scal);  // 64 x 256 lanes per tensor (the U-Net's 2.4 M-element weights)
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    return NASTAR_OK;
}

int nastar_pack_conv_weights_multi_f16(const long long* table, int n, int max_tiles, int split, float* scal, uint16_t* flat16, float* flatf,
                                       void* stream)
{
    if (!table || !scal || !flat16 || !flatf) return NASTAR_ERR_NULL;
    if (n <= 0 || n > 65535) return NASTAR_ERR_BAD_SHAPE;
    if (max_tiles <= 0 || max_tiles > 65535) return NASTAR_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(nastar_pack_weight_multi_kernel, dim3((unsigned)max_tiles, (unsigned)n), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       table, split, scal, flat16, flatf);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    return NASTAR_OK;
}

int nastar_rmsprop_multi_f32(const long long* table, int n, float lr, float alpha, float eps, void* stream)
{
    if (!table) return NASTAR_ERR_NULL;
    if (n <= 0 || n > 65535) return NASTAR_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(nastar_rmsprop_multi_kernel, dim3(512, (unsigned)n), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), table, lr, alpha, eps);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    return NASTAR_OK;
}

int nastar_bn_coef_fwd(const double* sums, const float* gamma, const float* beta, double eps, long long npix, double momentum,
                       float* running_mean, float* running_var, float* k2, float* k3, double* mean_out, double* invstd_out, int C,
                       void* stream)
{
    if (!sums || !gamma || !beta || !k2 || !k3 || !mean_out || !invstd_out || (running_mean && !running_var)) return NASTAR_ERR_NULL;
    if (C <= 0 || npix <= 0) return NASTAR_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(nastar_bn_coef_fwd_kernel, dim3(1), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), sums, gamma, beta, eps,
                       (double)npix, momentum, running_mean, running_var, k2, k3, mean_out, invstd_out, C);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    return NASTAR_OK;
}

int nastar_bn_coef_bwd_io(const double* sums, const float* amax_dy, const double* mean, const double* invstd, const float* gamma,
                          long long npix, const float* gscale_in, float* gscale_out, float* dgamma, float* dbeta, float* c1, float* c2,
                          float* c3, int C, void* stream)
{
    if (!sums || !amax_dy || !mean || !invstd || !gamma || !gscale_in || !gscale_out || !dgamma || !dbeta || !c1 || !c2 || !c3) return NASTAR_ERR_NULL;
    if (C <= 0 || npix <= 0) return NASTAR_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(nastar_bn_coef_bwd_kernel, dim3(1), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), sums, amax_dy, mean, invstd,
                       gamma, (double)npix, gscale_out, dgamma, dbeta, c1, c2, c3, C, gscale_in);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    return NASTAR_OK;
}

int nastar_bn_coef_bwd(const double* sums, const float* amax_dy, const double* mean, const double* invstd, const float* gamma,
                       long long npix, float* gscale, float* dgamma, float* dbeta, float* c1, float* c2, float* c3, int C, void* stream)
{
    if (!sums || !amax_dy || !mean || !invstd || !gamma || !gscale || !dgamma || !dbeta || !c1 || !c2 || !c3) return NASTAR_ERR_NULL;
    if (C <= 0 || npix <= 0) return NASTAR_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(nastar_bn_coef_bwd_kernel, dim3(1), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), sums, amax_dy, mean, invstd,
                       gamma, (double)npix, gscale, dgamma, dbeta, c1, c2, c3, C);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    return NASTAR_OK;
}

int nastar_grad_seed_f16(const float* d, long long npix, int split, uint16_t* dzb, float* gscale, float* amax_scratch, void* stream)
{
    if (!d || !dzb || !gscale || !amax_scratch) return NASTAR_ERR_NULL;
    if (npix <= 0) return NASTAR_ERR_BAD_SHAPE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(amax_scratch, 0, sizeof(float), s);
    if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync");
    const unsigned g1 = (unsigned)((npix + 255) / 256 < 1024 ? (npix + 255) / 256 : 1024);
    hipLaunchKernelGGL(nastar_absmax_kernel, dim3(g1), dim3(256), 0, s, d, npix, reinterpret_cast<unsigned int*>(amax_scratch));
    const long long total = npix * (split ? 8 : 4);
    const unsigned g2 = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (split) hipLaunchKernelGGL(nastar_grad_seed_kernel<true>, dim3(g2), dim3(256), 0, s, d, npix, amax_scratch, gscale, dzb);
    else hipLaunchKernelGGL(nastar_grad_seed_kernel<false>, dim3(g2), dim3(256), 0, s, d, npix, amax_scratch, gscale, dzb);
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    return NASTAR_OK;
}

int nastar_maxpool2x2_bwd_f16(const uint16_t* r, const uint16_t* dp, uint16_t* dr, int B, int H, int W, int C, int split, void* stream)
{
    if (!r || !dp || !dr) return NASTAR_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return NASTAR_ERR_BAD_SHAPE;
    if ((H | W) & 1 || C % 8) return NASTAR_ERR_UNSUPPORTED;
    const long long total = (long long)B * (H / 2) * (W / 2) * (C / 8);
    const unsigned grid = (unsigned)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (split) hipLaunchKernelGGL(nastar_maxpool2x2_bwd_kernel<true>, dim3(grid), dim3(256), 0, s, r, dp, dr, B, H, W, C);
    else hipLaunchKernelGGL(nastar_maxpool2x2_bwd_kernel<false>, dim3(grid), dim3(256), 0, s, r, dp, dr, B, H, W, C);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    return NASTAR_OK;
}

static unsigned stream_grid(long long total)
{
    const long long g = (total + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > 65536 ? 65536 : g));
}

int nastar_upcat_f16(const uint16_t* x, const uint16_t* skip, uint16_t* out, int B, int H, int W, int c1, int c2, int split, void* stream)
{
    if (!x || !out || (c2 > 0 && !skip)) return NASTAR_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || c1 <= 0 || c2 < 0) return NASTAR_ERR_BAD_SHAPE;
    if ((H | W) & 1 || c1 % 8 || c2 % 8) return NASTAR_ERR_UNSUPPORTED;
    const long long total = (long long)B * H * W * ((c1 + c2) / 8) * (split ? 2 : 1);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (split) hipLaunchKernelGGL(nastar_upcat_kernel<true>, dim3(stream_grid(total)), dim3(256), 0, s, x, skip, out, B, H, W, c1, c2);
    else hipLaunchKernelGGL(nastar_upcat_kernel<false>, dim3(stream_grid(total)), dim3(256), 0, s, x, skip, out, B, H, W, c1, c2);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    return NASTAR_OK;
}

int nastar_upcat_bwd_f16(const uint16_t* dcat, uint16_t* dx, uint16_t* dskip, int B, int H, int W, int c1, int c2, int split, void* stream)
{
    if (!dcat || !dx || (c2 > 0 && !dskip)) return NASTAR_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || c1 <= 0 || c2 < 0) return NASTAR_ERR_BAD_SHAPE;
    if ((H | W) & 1 || c1 % 8 || c2 % 8) return NASTAR_ERR_UNSUPPORTED;
    const long long total = (long long)B * (H / 2) * (W / 2) * (c1 / 8) + (long long)B * H * W * (c2 / 8);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (split) hipLaunchKernelGGL(nastar_upcat_bwd_kernel<true>, dim3(stream_grid(total)), dim3(256), 0, s, dcat, dx, dskip, B, H, W, c1, c2);
    else hipLaunchKernelGGL(nastar_upcat_bwd_kernel<false>, dim3(stream_grid(total)), dim3(256), 0, s, dcat, dx, dskip, B, H, W, c1, c2);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    return NASTAR_OK;
}

int nastar_grad_add_f16(const uint16_t* a, const float* scale_a, const uint16_t* b, const float* scale_b, uint16_t* out, float* scale_out,
                        long long npix, int C, int split, void* stream)
{
    if (!a || !b || !out || !scale_a || !scale_b || !scale_out) return NASTAR_ERR_NULL;
    if (npix <= 0 || C <= 0) return NASTAR_ERR_BAD_SHAPE;
    if (C % 8) return NASTAR_ERR_UNSUPPORTED;
    const long long total = npix * (C / 8);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (split) hipLaunchKernelGGL(nastar_grad_add_kernel<true>, dim3(stream_grid(total)), dim3(256), 0, s, a, scale_a, b, scale_b, out, scale_out, npix, C);
    else hipLaunchKernelGGL(nastar_grad_add_kernel<false>, dim3(stream_grid(total)), dim3(256), 0, s, a, scale_a, b, scale_b, out, scale_out, npix, C);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    return NASTAR_OK;
}


// ---- closing 1-channel BatchNorm + sigmoid * const block (nastar_encoder_train.hip.h) ---------------------------------------------------
int nastar_bn1_parts(long long n)
{
    if (n <= 0) return 0;
    long long g = (n + 2047) / 2048;
    return (int)(g > BN1_MAX_PARTS ? BN1_MAX_PARTS : g);
}

int nastar_bn1_fwd_partial(const float* z, long long n, double* part, void* stream)
{
    if (!z || !part) return NASTAR_ERR_NULL;
    if (n <= 0) return NASTAR_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(nastar_bn1_partial_kernel<2>, dim3((unsigned)nastar_bn1_parts(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), z,
                       nullptr, n, nullptr, nullptr, nullptr, nullptr, part);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    return NASTAR_OK;
}

int nastar_bn1_sigmoid_fwd(const float* z, long long n, const double* part, int nparts, double n_total, const float* gamma, const float* beta,
                           double eps, const float* cmul, double momentum, float* running_mean, float* running_var, float* cost_out,
                           double* stat_out, void* stream)
{
    if (!z || !part || !gamma || !beta || !cost_out || !stat_out || (running_mean && !running_var)) return NASTAR_ERR_NULL;
    if (n <= 0 || nparts <= 0 || nparts > BN1_MAX_PARTS || n_total < (double)n) return NASTAR_ERR_BAD_SHAPE;
    long long g = (n + 1023) / 1024;
    g = g > 2048 ? 2048 : g;
    hipLaunchKernelGGL(nastar_bn1_sigmoid_fwd_kernel, dim3((unsigned)g), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), z, n, part, nparts,
                       n_total, gamma, beta, eps, cmul, momentum, running_mean, running_var, cost_out, stat_out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    return NASTAR_OK;
}

int nastar_bn1_sigmoid_bwd_partial(const float* z, const float* dcost, long long n, const double* stat, const float* gamma, const float* beta,
                                   const float* cmul, double* part, void* stream)
{
    if (!z || !dcost || !stat || !gamma || !beta || !part) return NASTAR_ERR_NULL;
    if (n <= 0) return NASTAR_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(nastar_bn1_partial_kernel<3>, dim3((unsigned)nastar_bn1_parts(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), z,
                       dcost, n, stat, gamma, beta, cmul, part);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    return NASTAR_OK;
}

int nastar_bn1_sigmoid_bwd(const float* z, const float* dcost, long long n, const double* stat, const float* gamma, const float* beta,
                           const float* cmul, const double* part, int nparts, double n_total, float* dz_out, float* dgamma_out,
                           float* dbeta_out, float* dconst_out, void* stream)
{
    if (!z || !dcost || !stat || !gamma || !beta || !part || !dz_out || !dgamma_out || !dbeta_out) return NASTAR_ERR_NULL;
    if (n <= 0 || nparts <= 0 || nparts > BN1_MAX_PARTS || n_total < (double)n) return NASTAR_ERR_BAD_SHAPE;
    long long g = (n + 1023) / 1024;
    g = g > 2048 ? 2048 : g;
    hipLaunchKernelGGL(nastar_bn1_sigmoid_bwd_kernel, dim3((unsigned)g), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), z, dcost, n, stat,
                       gamma, beta, cmul, part, nparts, n_total, dz_out, dgamma_out, dbeta_out, dconst_out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    return NASTAR_OK;
}

// ---- the 1-channel closing convolution of the CNN encoders as streams (nastar_encoder_co1.hip.h) ----------------------------------------
static bool co1_shape_ok(int B, int H, int W, int C) { return B > 0 && H > 0 && W > 0 && C >= 8 && C <= 512 && (C & (C - 1)) == 0; }

static long long co1_wgrad_grid(long long npix, int C)
{
    const long long per = 256 / (C / 8);
    long long grid = (npix + per * 8 - 1) / (per * 8);
    if (grid > 1024) grid = 1024;
    if (grid < 1) grid = 1;
    return grid;
}

size_t nastar_conv3x3_co1_workspace_bytes(int B, int H, int W, int C)
{
    if (!co1_shape_ok(B, H, W, C)) return 0;
    const long long npix = (long long)B * H * W;
    const size_t proj = (size_t)npix * 9 * sizeof(float), wg = (size_t)co1_wgrad_grid(npix, C) * (size_t)C * 9 * sizeof(float);
    return proj > wg ? proj : wg;
}

int nastar_conv3x3_co1_f16(const uint16_t* a, const float* w, const float* bias, int B, int H, int W, int C, int split, const float* k2,
                           const float* k3, float* z_out, void* workspace, size_t workspace_bytes, void* stream)
{
    if (!a || !w || !z_out || !workspace || (k2 && !k3)) return NASTAR_ERR_NULL;
    if (!co1_shape_ok(B, H, W, C)) return (B <= 0 || H <= 0 || W <= 0 || C <= 0) ? NASTAR_ERR_BAD_SHAPE : NASTAR_ERR_UNSUPPORTED;
    const long long npix = (long long)B * H * W;
    if (workspace_bytes < (size_t)npix * 9 * sizeof(float)) return NASTAR_ERR_WORKSPACE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    float* P = static_cast<float*>(workspace);
    const long long per = 256 / (C / 8);
    long long grid = (npix + per * 4 - 1) / (per * 4);
    if (grid > 2048) grid = 2048;
    if (split) hipLaunchKernelGGL(nastar_co1_proj_kernel<true>, dim3((unsigned)grid), dim3(256), 0, s, a, w, P, npix, C, k2, k3);
    else hipLaunchKernelGGL(nastar_co1_proj_kernel<false>, dim3((unsigned)grid), dim3(256), 0, s, a, w, P, npix, C, k2, k3);
    long long g2 = (npix + 255) / 256;
    if (g2 > 4096) g2 = 4096;
    hipLaunchKernelGGL(nastar_co1_shift_kernel, dim3((unsigned)g2), dim3(256), 0, s, P, bias, z_out, npix, H, W);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    return NASTAR_OK;
}

int nastar_conv3x3_co1_wgrad_f16(const float* d, const uint16_t* a, int B, int H, int W, int C, int split, const float* k2, const float* k3,
                                 float* dw_out, void* workspace, size_t workspace_bytes, void* stream)
{
    if (!d || !a || !dw_out || !workspace || (k2 && !k3)) return NASTAR_ERR_NULL;
    if (!co1_shape_ok(B, H, W, C)) return (B <= 0 || H <= 0 || W <= 0 || C <= 0) ? NASTAR_ERR_BAD_SHAPE : NASTAR_ERR_UNSUPPORTED;
    const long long npix = (long long)B * H * W;
    const long long grid = co1_wgrad_grid(npix, C);
    if (workspace_bytes < (size_t)grid * (size_t)C * 9 * sizeof(float)) return NASTAR_ERR_WORKSPACE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    float* part = static_cast<float*>(workspace);
    if (split) hipLaunchKernelGGL(nastar_co1_wgrad_kernel<true>, dim3((unsigned)grid), dim3(256), 0, s, d, a, part, npix, C, H, W, k2, k3);
    else hipLaunchKernelGGL(nastar_co1_wgrad_kernel<false>, dim3((unsigned)grid), dim3(256), 0, s, d, a, part, npix, C, H, W, k2, k3);
    hipLaunchKernelGGL(nastar_co1_wgrad_finish_kernel, dim3((unsigned)((C * 9 + 7) / 8)), dim3(256), 0, s, part, (int)grid, C * 9, dw_out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    return NASTAR_OK;
}

int nastar_grad_scale_f32(const float* d, long long npix, float* gscale, float* amax_scratch, void* stream)
{
    if (!d || !gscale || !amax_scratch) return NASTAR_ERR_NULL;
    if (npix <= 0) return NASTAR_ERR_BAD_SHAPE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(amax_scratch, 0, sizeof(float), s);
    if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync");
    const unsigned g1 = (unsigned)((npix + 255) / 256 < 1024 ? (npix + 255) / 256 : 1024);
    hipLaunchKernelGGL(nastar_absmax_kernel, dim3(g1), dim3(256), 0, s, d, npix, reinterpret_cast<unsigned int*>(amax_scratch));
    hipLaunchKernelGGL(nastar_grad_scale_kernel, dim3(1), dim3(64), 0, s, amax_scratch, gscale);
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    return NASTAR_OK;
}

// nastar_chan_stats_f16_ws (backward form) with u = gscale * (the closing convolution's input gradient of d), formed on the fly: the sums a
// data-parallel step all-reduces between the statistics and the coefficients
int nastar_chan_stats_u1_f16_ws(const float* d, const float* wlast, const float* gscale, int B, int H, int W, const uint16_t* v, const float* ms,
                                const float* mt, double* sums, float* amax_out, int C, int split, void* workspace, size_t workspace_bytes,
                                void* stream)
{
    if (!d || !wlast || !gscale || !v || !ms || !mt || !sums || !workspace) return NASTAR_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return NASTAR_ERR_BAD_SHAPE;
    if (C % 8 || C > 2048 || 256 % (C / 8)) return NASTAR_ERR_UNSUPPORTED;
    const long long npix = (long long)B * H * W;
    if (workspace_bytes < nastar_chan_stats_workspace_bytes(npix, C)) return NASTAR_ERR_WORKSPACE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const long long grid = chan_stats_grid(npix, C);
    double* part = static_cast<double*>(workspace);
    float* amax_part = reinterpret_cast<float*>(part + (size_t)grid * (size_t)(2 * C));
    U1Src u1;
    u1.d = d; u1.w = wlast; u1.gscale = gscale; u1.H = H; u1.W = W;
    if (split) hipLaunchKernelGGL((nastar_chan_stats_kernel<true, true>), dim3((unsigned)grid), dim3(256), 0, s, nullptr, v, ms, mt, sums, nullptr, npix, C, part, amax_part, u1);
    else hipLaunchKernelGGL((nastar_chan_stats_kernel<false, true>), dim3((unsigned)grid), dim3(256), 0, s, nullptr, v, ms, mt, sums, nullptr, npix, C, part, amax_part, u1);
    hipLaunchKernelGGL(nastar_chan_stats_finish_kernel, dim3((unsigned)((2 * C + 7) / 8)), dim3(256), 0, s, part, amax_part, (int)grid, 2 * C, sums,
                       amax_out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    return NASTAR_OK;
}

// nastar_bn_stats_coef_bwd_f16 / nastar_chan_affine_f16 for the block in FRONT of the closing convolution: `da` is not read, it is
// gscale_in * (the closing convolution's input gradient of d), formed on the fly
int nastar_bn_stats_coef_bwd_u1_f16(const float* d, const float* wlast, int B, int H, int W, const uint16_t* z, const float* ms, const float* mt,
                                    int C, int split, const double* mean, const double* invstd, const float* gamma, const float* gscale_in,
                                    float* gscale_out, float* dgamma, float* dbeta, float* c1, float* c2, float* c3, double* sums_out,
                                    void* workspace, size_t workspace_bytes, void* stream)
{
    if (!d || !wlast || !z || !ms || !mt || !mean || !invstd || !gamma || !gscale_in || !gscale_out || !dgamma || !dbeta || !c1 || !c2 || !c3 || !workspace)
        return NASTAR_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || gscale_in == gscale_out) return NASTAR_ERR_BAD_SHAPE;
    if (C % 8 || C > 2048 || 256 % (C / 8)) return NASTAR_ERR_UNSUPPORTED;
    const long long npix = (long long)B * H * W;
    if (workspace_bytes < nastar_chan_stats_workspace_bytes(npix, C)) return NASTAR_ERR_WORKSPACE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const long long grid = chan_stats_grid(npix, C);
    double* part = static_cast<double*>(workspace);
    float* amax_part = reinterpret_cast<float*>(part + (size_t)grid * (size_t)(2 * C));
    U1Src u1;
    u1.d = d; u1.w = wlast; u1.gscale = gscale_in; u1.H = H; u1.W = W;
    if (split) hipLaunchKernelGGL((nastar_chan_stats_kernel<true, true>), dim3((unsigned)grid), dim3(256), 0, s, nullptr, z, ms, mt, nullptr, nullptr, npix, C, part, amax_part, u1);
    else hipLaunchKernelGGL((nastar_chan_stats_kernel<false, true>), dim3((unsigned)grid), dim3(256), 0, s, nullptr, z, ms, mt, nullptr, nullptr, npix, C, part, amax_part, u1);
    hipLaunchKernelGGL(nastar_bn_finish_coef_kernel<true>, dim3((unsigned)((C + 3) / 4)), dim3(256), 0, s, part, amax_part, (int)grid, C, sums_out, gamma,
                       nullptr, 0.0, (double)npix, 0.0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, mean, invstd, gscale_in, gscale_out, dgamma,
                       dbeta, c1, c2, c3);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    return NASTAR_OK;
}

int nastar_chan_affine_u1_f16(const float* d, const float* wlast, const float* gscale, int B, int H, int W, const uint16_t* z, const float* k1,
                              const float* k2, const float* k3, const float* ms, const float* mt, uint16_t* out, int C, int split, void* stream)
{
    if (!d || !wlast || !gscale || !z || !out || !k1 || !k2 || !k3 || !ms || !mt) return NASTAR_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return NASTAR_ERR_BAD_SHAPE;
    if (C % 8 || C > 2048 || 256 % (C / 8)) return NASTAR_ERR_UNSUPPORTED;
    const long long npix = (long long)B * H * W;
    const long long per = 256 / (C / 8);
    long long grid = (npix + per * 16 - 1) / (per * 16);
    if (grid < 1024) grid = (npix + per * 2 - 1) / (per * 2) < 1024 ? (npix + per * 2 - 1) / (per * 2) : 1024;
    if (grid > 8192) grid = 8192;
    if (grid < 1) grid = 1;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    U1Src u1;
    u1.d = d; u1.w = wlast; u1.gscale = gscale; u1.H = H; u1.W = W;
    if (split) hipLaunchKernelGGL((nastar_chan_affine_kernel<true, true>), dim3((unsigned)grid), dim3(256), 0, s, nullptr, z, k1, k2, k3, ms, mt, out, npix, C, 0, u1);
    else hipLaunchKernelGGL((nastar_chan_affine_kernel<false, true>), dim3((unsigned)grid), dim3(256), 0, s, nullptr, z, k1, k2, k3, ms, mt, out, npix, C, 0, u1);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    return NASTAR_OK;
}


}  // extern "C"
