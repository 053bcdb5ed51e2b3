// nastar_encoder_capi.hip -- C-ABI entry points of the bf16-MFMA CNN encoder (include/nastar.h) and their launch logic.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "nastar_host.hip.h"
#include "nastar_encoder.hip.h"
#include "nastar_encoder_downsize.hip.h"
#include "nastar_conv_flat.hip.h"

namespace nastar {

// bit 0: force the tiled conv kernel even for 32x32 images (parity tests of both kernels); read per call
static int enc_flags()
{
    const char* e = getenv("NASTAR_ENCODER_FLAGS");
    return e ? atoi(e) : 0;
}

template <int CIN, int COUT, int NT, bool kRelu, bool kFinal>
static int launch_conv(const ConvArgs& ca, hipStream_t stream)
{
    constexpr int KS = (CIN < ENC_KS) ? CIN : ENC_KS;
    constexpr size_t lds = (size_t)(ENC_TH + 2) * (ENC_TW + 2) * ENC_PIX_B + (size_t)9 * (KS / 16) * 2 * NT * 16 + (size_t)NT * 8;
    auto kern = &nastar_conv3x3_kernel<CIN, COUT, NT, kRelu, kFinal>;
    int rc = ensure_lds(kern, lds);
    if (rc) return rc;
    const unsigned grid = (unsigned)((size_t)ca.B * (ca.H / ENC_TH) * (ca.W / ENC_TW) * (COUT / NT));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(ENC_THREADS), lds, stream, ca);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    return NASTAR_OK;
}

// 32x32 images, CIN >= 32 and COUT >= 64: whole-image workgroups (nastar_conv3x3_img32_kernel), otherwise the tiled kernel
static int conv_cu_count(int* out)
{
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hip_fail(hipGetLastError(), "device query");
        n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    *out = n_cu;
    return NASTAR_OK;
}

// CIN >= 32 and COUT >= 64, H and W multiples of 32: persistent 32x32-tile workgroups (nastar_conv3x3_img32_kernel; 32x32 images
// take its whole-image form), otherwise the generic tiled kernel
template <int CIN, int COUT, bool kRelu>
static int launch_conv_auto(const ConvArgs& ca, hipStream_t stream)
{
    if (ca.H % 32 != 0 || ca.W % 32 != 0 || (enc_flags() & 1)) return launch_conv<CIN, COUT, (COUT >= 64 ? 64 : 32), kRelu, false>(ca, stream);
    const bool whole = ca.H == 32 && ca.W == 32;
    void (*kern)(const ConvArgs) = &nastar_conv3x3_img32_kernel<CIN, COUT, kRelu>;
    if (!whole) kern = &nastar_conv3x3_img32_kernel<CIN, COUT, kRelu, false, 0, true>;
    int rc = ensure_lds(kern, I32_LDS_BYTES);
    if (rc) return rc;
    int n_cu = 0;
    if ((rc = conv_cu_count(&n_cu))) return rc;
    const long long items = (long long)ca.B * (ca.H / 32) * (ca.W / 32) * (COUT / I32_NT);
    const unsigned grid = (unsigned)(items < n_cu ? items : n_cu);  // persistent: one workgroup per CU
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), I32_LDS_BYTES, stream, ca);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    return NASTAR_OK;
}

// input assembly + 2 -> 32 + 32 -> 64 channels for 32x32 maps in one persistent kernel
static int launch_conv_stem32(const StemArgs& sa, hipStream_t stream, bool f16 = false)
{
    void (*kern)(const StemArgs) = f16 ? &nastar_conv_stem32_kernel<true> : &nastar_conv_stem32_kernel<false>;
    int rc = ensure_lds(kern, STEM_LDS_BYTES);
    if (rc) return rc;
    int n_cu = 0;
    if ((rc = conv_cu_count(&n_cu))) return rc;
    const unsigned grid = (unsigned)(sa.B < n_cu ? sa.B : n_cu);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), STEM_LDS_BYTES, stream, sa);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    return NASTAR_OK;
}

// 128 -> 256 channels + the fused 256 -> 1 layer + sigmoid * const: writes the cost map, the 256-channel tensor never exists
static int launch_conv_fused_final(const ConvArgs& ca, hipStream_t stream, bool f16 = false, bool split = false)
{
    void (*kern)(const ConvArgs) = &nastar_conv3x3_img32_kernel<128, 256, true, true>;
    if (f16) kern = &nastar_conv3x3_img32_kernel<128, 256, true, true, 0, false, true, false>;
    if (split) kern = &nastar_conv3x3_img32_kernel<128, 256, true, true, 0, false, true, true>;  // f16x3: the last layer's three split products chained in the epilogue
    if (enc_flags() & 512) kern = &nastar_conv3x3_img32_kernel<128, 256, true, true, 2>;  // dev: every workgroup reads image 0 (L2 hits)
    if (enc_flags() & 256) kern = &nastar_conv3x3_img32_kernel<128, 256, true, true, 1>;  // dev: cycle totals into the (unused) output slab
    int rc = ensure_lds(kern, I32_LDS_BYTES);
    if (rc) return rc;
    int n_cu = 0;
    if ((rc = conv_cu_count(&n_cu))) return rc;
    const unsigned grid = (unsigned)(ca.B < n_cu ? ca.B : n_cu);  // a workgroup owns whole images
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), I32_LDS_BYTES, stream, ca);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    return NASTAR_OK;
}

// f16x3 split precision: persistent 32x32-tile kernel over 3 CIN virtual channels; H, W multiples of 32
template <int CIN, int COUT, bool kSplit>
static int launch_conv_split(const ConvArgs& ca, hipStream_t stream)
{
    const bool whole = ca.H == 32 && ca.W == 32;
    void (*kern)(const ConvArgs) = &nastar_conv3x3_img32_kernel<CIN, COUT, true, false, 0, false, true, kSplit>;
    if (!whole) kern = &nastar_conv3x3_img32_kernel<CIN, COUT, true, false, 0, true, true, kSplit>;
    int rc = ensure_lds(kern, I32_LDS_BYTES);
    if (rc) return rc;
    int n_cu = 0;
    if ((rc = conv_cu_count(&n_cu))) return rc;
    const long long items = (long long)ca.B * (ca.H / 32) * (ca.W / 32) * (COUT / I32_NT);
    hipLaunchKernelGGL(kern, dim3((unsigned)(items < n_cu ? items : n_cu)), dim3(512), I32_LDS_BYTES, stream, ca);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    return NASTAR_OK;
}

static int launch_conv_final_split_pass(const ConvArgs& ca, hipStream_t stream)
{
    constexpr size_t lds = (size_t)(ENC_TH + 2) * (ENC_TW + 2) * ENC_PIX_B + (size_t)(((ENC_TH + 2) * (ENC_TW + 2) + 31) / 32) * 32 * 9 * 4;
    auto kern = &nastar_conv3x3_final_kernel<256, true>;
    int rc = ensure_lds(kern, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)((size_t)ca.B * (ca.H / ENC_TH) * (ca.W / ENC_TW))), dim3(ENC_THREADS), lds, stream, ca);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    return NASTAR_OK;
}

static int launch_conv_final(const ConvArgs& ca, hipStream_t stream)
{
    constexpr size_t lds = (size_t)(ENC_TH + 2) * (ENC_TW + 2) * ENC_PIX_B + (size_t)(((ENC_TH + 2) * (ENC_TW + 2) + 31) / 32) * 32 * 9 * 4;
    auto kern = &nastar_conv3x3_final_kernel<256>;
    int rc = ensure_lds(kern, lds);
    if (rc) return rc;
    const unsigned grid = (unsigned)((size_t)ca.B * (ca.H / ENC_TH) * (ca.W / ENC_TW));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(ENC_THREADS), lds, stream, ca);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    return NASTAR_OK;
}
// ---- fp16 forms of the same encoder ------------------------------------------------------------------------------------------------------
// f16x3 ("fp32-grade"): every conv is hi*hi + lo*hi + hi*lo on the fp16 MFMA, activations [hi | lo];  f16: plain fp16 operands.
static size_t fp16_bytes_per_pixel(bool split) { return (size_t)(split ? 2 : 1) * (32 + 64 + 128 + 256) * 2 + 4; }  // layer outputs + fp32 partial sum

static size_t fp16_workspace_bytes(int B, int H, int W, bool split)
{
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    long long cap = (1ll << 20) / ((long long)H * W);
    if (cap < 1) cap = 1;
    return (size_t)(B < cap ? B : cap) * H * W * fp16_bytes_per_pixel(split);
}

template <bool kSplit>
static int encoder_fp16_impl(const float* map, const float* start, const float* goal, int plus, int B, int H, int W, const float* w1_f32,
                             const uint16_t* const* wts, const float* const* scale, const float* const* shift, float final_mul,
                             float* cost_out, void* workspace, size_t workspace_bytes, void* stream)
{
    if (!map || !cost_out || !w1_f32 || !wts || !scale || !shift || !workspace || (plus && (!start || !goal))) return NASTAR_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0) return NASTAR_ERR_BAD_SHAPE;
    if (H % 32 != 0 || W % 32 != 0) return NASTAR_ERR_UNSUPPORTED;
    constexpr int M = kSplit ? 2 : 1;  // fp16 terms per activation
    const size_t per_img = (size_t)H * W * fp16_bytes_per_pixel(kSplit);
    int chunk = (int)(workspace_bytes / per_img);
    if (chunk <= 0) return NASTAR_ERR_WORKSPACE;
    if (chunk > B) chunk = B;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const size_t npc = (size_t)chunk * H * W;
    uint16_t* a1 = static_cast<uint16_t*>(workspace);   // [.., M*32]
    uint16_t* a2 = a1 + npc * 32 * M;                   // [.., M*64]
    uint16_t* a3 = a2 + npc * 64 * M;                   // [.., M*128]
    uint16_t* a4 = a3 + npc * 128 * M;                  // [.., M*256]
    float* zacc = reinterpret_cast<float*>(a4 + npc * 256 * M);
    // weight pointers: split form = {layer 2, 3, 4 over 3*cin virtual channels, last layer hi, lo}; plain fp16 = the five standard packs
    const uint16_t* w2 = kSplit ? wts[0] : wts[1];
    const uint16_t* w3 = kSplit ? wts[1] : wts[2];
    const uint16_t* w4 = kSplit ? wts[2] : wts[3];
    const uint16_t* w5 = kSplit ? wts[3] : wts[4];
    for (int b0 = 0; b0 < B; b0 += chunk) {
        const int nb = (B - b0 < chunk) ? B - b0 : chunk;
        const size_t off = (size_t)b0 * H * W;
        const long long npix = (long long)nb * H * W;
        const unsigned pg = (unsigned)((npix + 255) / 256 < 16384 ? (npix + 255) / 256 : 16384);
        if constexpr (!kSplit) {
            if (H == 32 && W == 32 && !(enc_flags() & 25)) {  // plain fp16 on 32x32 maps: stem + 64->128 + fused 128->256(+1), as the bf16 form
                StemArgs sa;
                sa.map = map + off; sa.start = plus ? start + off : nullptr; sa.goal = plus ? goal + off : nullptr; sa.plus = plus; sa.B = nb;
                sa.w1 = wts[0]; sa.scale1 = scale[0]; sa.shift1 = shift[0]; sa.w2 = w2; sa.scale2 = scale[1]; sa.shift2 = shift[1];
                sa.out = a2;
                int rc;
                if ((rc = launch_conv_stem32(sa, s, true))) return rc;
                ConvArgs ca;
                ca.B = nb; ca.H = H; ca.W = W; ca.out_f32 = nullptr; ca.final_mul = final_mul;
                ca.wfin = nullptr; ca.wfin_lo = nullptr; ca.fscale = nullptr; ca.fshift = nullptr; ca.in_stride = 0; ca.pass_flags = 0; ca.zacc = nullptr;
                ca.in = a2; ca.out = a3; ca.wpack = w3; ca.scale = scale[2]; ca.shift = shift[2];
                if ((rc = launch_conv_split<64, 128, false>(ca, s))) return rc;
                ca.in = a3; ca.out = a4; ca.wpack = w4; ca.scale = scale[3]; ca.shift = shift[3];
                ca.wfin = w5; ca.fscale = scale[4]; ca.fshift = shift[4]; ca.out_f32 = cost_out + off;
                if ((rc = launch_conv_fused_final(ca, s, true))) return rc;
                continue;
            }
        }
        if (plus)
            hipLaunchKernelGGL((nastar_conv_first_f32_kernel<2, kSplit>), dim3(pg), dim3(256), 0, s, map + off, start + off, goal + off,
                               w1_f32, scale[0], shift[0], a1, nb, H, W);
        else
            hipLaunchKernelGGL((nastar_conv_first_f32_kernel<1, kSplit>), dim3(pg), dim3(256), 0, s, map + off, map, map, w1_f32, scale[0],
                               shift[0], a1, nb, H, W);
        ConvArgs ca;
        ca.B = nb; ca.H = H; ca.W = W; ca.out_f32 = nullptr; ca.final_mul = final_mul;
        ca.wfin = nullptr; ca.wfin_lo = nullptr; ca.fscale = nullptr; ca.fshift = nullptr; ca.in_stride = 0; ca.pass_flags = 0; ca.zacc = nullptr;
        int rc;
        ca.in = a1; ca.out = a2; ca.wpack = w2; ca.scale = scale[1]; ca.shift = shift[1];
        if ((rc = launch_conv_split<32, 64, kSplit>(ca, s))) return rc;
        ca.in = a2; ca.out = a3; ca.wpack = w3; ca.scale = scale[2]; ca.shift = shift[2];
        if ((rc = launch_conv_split<64, 128, kSplit>(ca, s))) return rc;
        if constexpr (kSplit) {
            if (H == 32 && W == 32 && !(enc_flags() & 8)) {
                // 32x32 maps: 128 -> 256 with the 1-channel last layer, its BatchNorm and sigmoid * const fused into the epilogue (as the
                // bf16 / fp16 routes): the 1 KB-per-pixel split activation of the widest layer is never written, nor read three times
                ca.in = a3; ca.out = a4; ca.wpack = w4; ca.scale = scale[3]; ca.shift = shift[3];
                ca.wfin = wts[3]; ca.wfin_lo = wts[4]; ca.fscale = scale[4]; ca.fshift = shift[4]; ca.out_f32 = cost_out + off;
                if ((rc = launch_conv_fused_final(ca, s, true, true))) return rc;
                continue;
            }
        }
        ca.in = a3; ca.out = a4; ca.wpack = w4; ca.scale = scale[3]; ca.shift = shift[3];
        if ((rc = launch_conv_split<128, 256, kSplit>(ca, s))) return rc;
        ca.out = nullptr; ca.out_f32 = cost_out + off; ca.scale = scale[4]; ca.shift = shift[4]; ca.in_stride = 256 * M; ca.zacc = zacc;
        if constexpr (kSplit) {  // last layer: three accumulating passes of the tap-major kernel: x_hi*W_hi, x_lo*W_hi, x_hi*W_lo
            ca.in = a4; ca.wpack = wts[3]; ca.pass_flags = 2;
            if ((rc = launch_conv_final_split_pass(ca, s))) return rc;
            ca.in = a4 + 256; ca.wpack = wts[3]; ca.pass_flags = 3;
            if ((rc = launch_conv_final_split_pass(ca, s))) return rc;
            ca.in = a4; ca.wpack = wts[4]; ca.pass_flags = 1;
            if ((rc = launch_conv_final_split_pass(ca, s))) return rc;
        } else {
            ca.in = a4; ca.wpack = w5; ca.pass_flags = 0;
            if ((rc = launch_conv_final_split_pass(ca, s))) return rc;
        }
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    return NASTAR_OK;
}

}  // namespace nastar

using namespace nastar;

extern "C" {

// ---- CNN encoder (eval mode, bf16 MFMA) -------------------------------------------------------------------------------
// padded channels per layer: in 16, 32, 64, 128, 256 (layer 1: 2 real + 14 zero); out 32, 64, 128, 256, 32 (layer 5: 1 real)
// Images per pass: the activation slabs of a pass (800 B per pixel) are far larger than the 256 MB MALL either way, so the pass is
// sized for few launches and even work per persistent workgroup: 4 Mi pixels (4096 maps of 32x32, 3.3 GB of the 288 GB HBM).
// NASTAR_ENCODER_CHUNK overrides it (dev).
static int enc_chunk_images(int H, int W)
{
    const char* e = getenv("NASTAR_ENCODER_CHUNK");
    if (e && atoi(e) > 0) return atoi(e);
    const long long px = (long long)H * W;
    const long long n = (4ll << 20) / px;
    return n < 1 ? 1 : (int)n;
}
constexpr size_t kEncBytesPerPixel = (16 + 128 + 256) * 2;  // x0 + ping (<=128 ch) + pong (<=256 ch), bf16

size_t nastar_encoder_workspace_bytes(int B, int H, int W)
{
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    const int cap = enc_chunk_images(H, W);
    const int chunk = B < cap ? B : cap;  // images processed per pass
    return (size_t)chunk * H * W * kEncBytesPerPixel;
}

int nastar_encoder_cnn_forward(const float* map, const float* start, const float* goal, int plus, int B, int H, int W,
                               const uint16_t* const* wpack, const float* const* scale, const float* const* shift,
                               float final_mul, float* cost_out, void* workspace, size_t workspace_bytes, void* stream)
{
    if (!map || !cost_out || !wpack || !scale || !shift || !workspace || (plus && (!start || !goal))) return NASTAR_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0) return NASTAR_ERR_BAD_SHAPE;
    if (H % ENC_TH != 0 || W % ENC_TW != 0) return NASTAR_ERR_UNSUPPORTED;
    const size_t per_img = (size_t)H * W * kEncBytesPerPixel;
    int chunk = (int)(workspace_bytes / per_img);
    if (chunk <= 0) return NASTAR_ERR_WORKSPACE;
    if (chunk > B) chunk = B;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    uint16_t* x0 = static_cast<uint16_t*>(workspace);
    uint16_t* ping = x0 + (size_t)chunk * H * W * 16;
    uint16_t* pong = ping + (size_t)chunk * H * W * 128;
    for (int b0 = 0; b0 < B; b0 += chunk) {
        const int nb = (B - b0 < chunk) ? B - b0 : chunk;
        const size_t off = (size_t)b0 * H * W;
        const long long npix = (long long)nb * H * W;
        const unsigned pg = (unsigned)((npix + 255) / 256 < 16384 ? (npix + 255) / 256 : 16384);
        ConvArgs ca;
        ca.B = nb; ca.H = H; ca.W = W; ca.out_f32 = nullptr; ca.final_mul = final_mul;
        ca.wfin = nullptr; ca.wfin_lo = nullptr; ca.fscale = nullptr; ca.fshift = nullptr; ca.in_stride = 0; ca.pass_flags = 0; ca.zacc = nullptr;
        int rc;
        if (H == 32 && W == 32 && !(enc_flags() & 17)) {  // bit 4: keep input assembly and the first two layers separate launches
            StemArgs sa;
            sa.map = map + off; sa.start = plus ? start + off : nullptr; sa.goal = plus ? goal + off : nullptr; sa.plus = plus; sa.B = nb;
            sa.w1 = wpack[0]; sa.scale1 = scale[0]; sa.shift1 = shift[0]; sa.w2 = wpack[1]; sa.scale2 = scale[1]; sa.shift2 = shift[1];
            sa.out = pong;
            if ((rc = launch_conv_stem32(sa, s))) return rc;
        } else {
            hipLaunchKernelGGL(nastar_encoder_prep_kernel, dim3(pg), dim3(256), 0, s, map + off, plus ? start + off : map,
                               plus ? goal + off : map, x0, npix, plus);
            ca.in = x0; ca.out = ping; ca.wpack = wpack[0]; ca.scale = scale[0]; ca.shift = shift[0];
            if ((rc = launch_conv<16, 32, 32, true, false>(ca, s))) return rc;
            ca.in = ping; ca.out = pong; ca.wpack = wpack[1]; ca.scale = scale[1]; ca.shift = shift[1];
            if ((rc = launch_conv_auto<32, 64, true>(ca, s))) return rc;
        }
        ca.in = pong; ca.out = ping; ca.wpack = wpack[2]; ca.scale = scale[2]; ca.shift = shift[2];
        if ((rc = launch_conv_auto<64, 128, true>(ca, s))) return rc;
        ca.in = ping; ca.out = pong; ca.wpack = wpack[3]; ca.scale = scale[3]; ca.shift = shift[3];
        if (H == 32 && W == 32 && !(enc_flags() & 9)) {  // bit 3: keep the last layer a separate launch
            ca.wfin = wpack[4]; ca.fscale = scale[4]; ca.fshift = shift[4]; ca.out_f32 = cost_out + off;
            if ((rc = launch_conv_fused_final(ca, s))) return rc;
            continue;
        }
        if ((rc = launch_conv_auto<128, 256, true>(ca, s))) return rc;
        ca.in = pong; ca.out = nullptr; ca.out_f32 = cost_out + off; ca.wpack = wpack[4]; ca.scale = scale[4]; ca.shift = shift[4];
        if ((rc = launch_conv_final(ca, s))) return rc;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    return NASTAR_OK;
}

// ---- fp16 forms of the same encoder (implementation: encoder_fp16_impl above) ----------------------------------------------------------
size_t nastar_encoder_workspace_bytes_f16x3(int B, int H, int W) { return fp16_workspace_bytes(B, H, W, true); }
size_t nastar_encoder_workspace_bytes_f16(int B, int H, int W) { return fp16_workspace_bytes(B, H, W, false); }

int nastar_encoder_cnn_forward_f16x3(const float* map, const float* start, const float* goal, int plus, int B, int H, int W,
                                     const float* w1_f32, const uint16_t* const* wsplit, const float* const* scale,
                                     const float* const* shift, float final_mul, float* cost_out, void* workspace,
                                     size_t workspace_bytes, void* stream)
{
    return encoder_fp16_impl<true>(map, start, goal, plus, B, H, W, w1_f32, wsplit, scale, shift, final_mul, cost_out, workspace,
                                   workspace_bytes, stream);
}

int nastar_encoder_cnn_forward_f16(const float* map, const float* start, const float* goal, int plus, int B, int H, int W,
                                   const float* w1_f32, const uint16_t* const* wpack16, const float* const* scale,
                                   const float* const* shift, float final_mul, float* cost_out, void* workspace,
                                   size_t workspace_bytes, void* stream)
{
    return encoder_fp16_impl<false>(map, start, goal, plus, B, H, W, w1_f32, wpack16, scale, shift, final_mul, cost_out, workspace,
                                    workspace_bytes, stream);
}

// One 3x3 convolution layer on its own (unit tests): in [B,H,W,CIN] bf16 -> out [B,H,W,COUT] bf16, y = relu?(acc*scale+shift).
int nastar_conv3x3_bf16(const uint16_t* in, const uint16_t* wpack, const float* scale, const float* shift, uint16_t* out,
                        int B, int H, int W, int cin, int cout, int relu, void* stream)
{
    if (!in || !wpack || !scale || !shift || !out) return NASTAR_ERR_NULL;
    if (B <= 0 || H % ENC_TH != 0 || W % ENC_TW != 0) return NASTAR_ERR_BAD_SHAPE;
    ConvArgs ca;
    ca.in = in; ca.wpack = wpack; ca.scale = scale; ca.shift = shift; ca.out = out; ca.out_f32 = nullptr; ca.final_mul = 1.f;
    ca.wfin = nullptr; ca.wfin_lo = nullptr; ca.fscale = nullptr; ca.fshift = nullptr; ca.in_stride = 0; ca.pass_flags = 0; ca.zacc = nullptr;
    ca.B = B; ca.H = H; ca.W = W;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (cin == 16 && cout == 32) return relu ? launch_conv<16, 32, 32, true, false>(ca, s) : launch_conv<16, 32, 32, false, false>(ca, s);
    if (cin == 32 && cout == 64) return relu ? launch_conv_auto<32, 64, true>(ca, s) : launch_conv_auto<32, 64, false>(ca, s);
    if (cin == 64 && cout == 128) return relu ? launch_conv_auto<64, 128, true>(ca, s) : launch_conv_auto<64, 128, false>(ca, s);
    if (cin == 128 && cout == 256) return relu ? launch_conv_auto<128, 256, true>(ca, s) : launch_conv_auto<128, 256, false>(ca, s);
    return NASTAR_ERR_UNSUPPORTED;
}

}  // extern "C"

// ---- CNNDownSize (WarCraft) on the f32-input MFMA: nastar_encoder_downsize.hip.h -----------------------------------------------
namespace nastar {
template <int CIN, int COUTP, bool kPool, bool kFinal>
static int launch_ds_conv(const DsConvArgs& a, hipStream_t s)
{
    auto kern = &nastar_conv3x3_f32mfma_kernel<CIN, COUTP, kPool, kFinal>;
    const size_t lds = (size_t)(DS_TR + 2) * (DS_TC + 2) * CIN * sizeof(float);
    int rc = ensure_lds(kern, lds);
    if (rc) return rc;
    const unsigned grid = (unsigned)((size_t)a.B * ((a.H + DS_TR - 1) / DS_TR) * ((a.W + DS_TC - 1) / DS_TC));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * (COUTP / 32)), lds, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    return NASTAR_OK;
}
static int ds_cin_pad(int c) { return c <= 2 ? 2 : 4; }
}  // namespace nastar

extern "C" {

size_t nastar_encoder_downsize_workspace_bytes(int B, int C, int H, int W, int depth)
{
    if (B <= 0 || C <= 0 || C > 4 || H <= 0 || W <= 0 || depth < 1 || depth > 4) return 0;
    size_t n = (size_t)B * H * W * nastar::ds_cin_pad(C);
    int h = H, w = W, ch = 32;
    for (int l = 0; l < depth; ++l) {
        h /= 2; w /= 2;
        n += (size_t)B * h * w * ch;
        ch *= 2;
    }
    return n * sizeof(float);
}

int nastar_encoder_cnn_downsize_forward(const float* image, const float* start, const float* goal, int plus, int B, int C, int H,
                                        int W, int h, int w, int depth, const float* const* wts, const float* const* scale,
                                        const float* const* shift, float final_mul, float* cost_out, void* workspace,
                                        size_t workspace_bytes, void* stream)
{
    using namespace nastar;
    if (!image || !cost_out || !wts || !scale || !shift || !workspace || (plus && (!start || !goal))) return NASTAR_ERR_NULL;
    const int Cin = C + (plus ? 1 : 0);
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || depth < 1 || depth > 4 || Cin > 4) return NASTAR_ERR_BAD_SHAPE;
    if (H % (1 << depth) != 0 || W % (1 << depth) != 0 || (plus && (h <= 0 || w <= 0))) return NASTAR_ERR_UNSUPPORTED;
    if (workspace_bytes < nastar_encoder_downsize_workspace_bytes(B, Cin, H, W, depth)) return NASTAR_ERR_WORKSPACE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int cp = ds_cin_pad(Cin);
    float* x = static_cast<float*>(workspace);
    const long long npix = (long long)B * H * W;
    const unsigned pg = (unsigned)((npix + 255) / 256 < 16384 ? (npix + 255) / 256 : 16384);
    hipLaunchKernelGGL(nastar_downsize_prep_kernel, dim3(pg), dim3(256), 0, s, image, plus ? start : image, plus ? goal : image, x, B, C,
                       H, W, plus ? h : 1, plus ? w : 1, plus, cp);
    DsConvArgs a;
    a.B = B; a.H = H; a.W = W; a.final_mul = final_mul;
    a.in = x;
    float* nxt = x + (size_t)npix * cp;
    int rc = NASTAR_OK;
    for (int l = 0; l < depth; ++l) {  // hidden blocks: conv + BN + ReLU + 2x2 max-pool (encoder.py:91-95)
        a.w = wts[l]; a.scale = scale[l]; a.shift = shift[l]; a.out = nxt;
        if (l == 0) rc = cp == 2 ? launch_ds_conv<2, 32, true, false>(a, s) : launch_ds_conv<4, 32, true, false>(a, s);
        else if (l == 1) rc = launch_ds_conv<32, 64, true, false>(a, s);
        else if (l == 2) rc = launch_ds_conv<64, 128, true, false>(a, s);
        else rc = launch_ds_conv<128, 256, true, false>(a, s);
        if (rc) return rc;
        a.H /= 2; a.W /= 2;
        a.in = nxt;
        nxt += (size_t)B * a.H * a.W * (32 << l);
    }
    a.w = wts[depth]; a.scale = scale[depth]; a.shift = shift[depth]; a.out = cost_out;  // last block + sigmoid * const
    if (depth == 1) rc = launch_ds_conv<32, 32, false, true>(a, s);
    else if (depth == 2) rc = launch_ds_conv<64, 32, false, true>(a, s);
    else if (depth == 3) rc = launch_ds_conv<128, 32, false, true>(a, s);
    else rc = launch_ds_conv<256, 32, false, true>(a, s);
    return rc;
}

}  // extern "C"

// ---- the persistent whole-image kernel of the CNN encoder as a stand-alone layer for 32x32 maps (training forward / input gradient) ----
namespace nastar {

template <int CIN, int COUT, bool kRelu, bool kSplit>
static int launch_img32_layer(const ConvArgs& ca, hipStream_t stream)
{
    void (*kern)(const ConvArgs) = &nastar_conv3x3_img32_kernel<CIN, COUT, kRelu, false, 0, false, true, kSplit>;
    int rc = ensure_lds(kern, I32_LDS_BYTES);
    if (rc) return rc;
    int n_cu = 0;
    if ((rc = conv_cu_count(&n_cu))) return rc;
    const long long items = (long long)ca.B * (COUT / I32_NT);
    hipLaunchKernelGGL(kern, dim3((unsigned)(items < n_cu ? items : n_cu)), dim3(512), I32_LDS_BYTES, stream, ca);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    return NASTAR_OK;
}

template <int CIN, int COUT>
static int launch_img32_pick(const ConvArgs& ca, bool relu, bool split, hipStream_t s)
{
    if (split) return relu ? launch_img32_layer<CIN, COUT, true, true>(ca, s) : launch_img32_layer<CIN, COUT, false, true>(ca, s);
    return relu ? launch_img32_layer<CIN, COUT, true, false>(ca, s) : launch_img32_layer<CIN, COUT, false, false>(ca, s);
}

}  // namespace nastar

extern "C" {

int nastar_conv3x3_img32_f16(const uint16_t* in, const uint16_t* wpack, const float* scale, const float* shift, uint16_t* out, int B,
                             int cin, int cout, int flags, void* stream)
{
    if (!in || !wpack || !scale || !shift || !out) return NASTAR_ERR_NULL;
    if (B <= 0) return NASTAR_ERR_BAD_SHAPE;
    if (!aligned16(in) || !aligned16(wpack) || !aligned16(out)) return NASTAR_ERR_BAD_SHAPE;
    const bool relu = flags & NASTAR_CONV_RELU, split = flags & NASTAR_CONV_SPLIT;
    if (flags & ~(NASTAR_CONV_RELU | NASTAR_CONV_SPLIT)) return NASTAR_ERR_UNSUPPORTED;
    ConvArgs ca;
    ca.in = in; ca.wpack = wpack; ca.scale = scale; ca.shift = shift; ca.out = out; ca.out_f32 = nullptr; ca.wfin = nullptr; ca.wfin_lo = nullptr;
    ca.fscale = nullptr; ca.fshift = nullptr; ca.in_stride = 0; ca.pass_flags = 0; ca.zacc = nullptr; ca.final_mul = 1.0f;
    ca.B = B; ca.H = 32; ca.W = 32;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (cin == 32 && cout == 64) return launch_img32_pick<32, 64>(ca, relu, split, s);
    if (cin == 64 && cout == 128) return launch_img32_pick<64, 128>(ca, relu, split, s);
    if (cin == 128 && cout == 256) return launch_img32_pick<128, 256>(ca, relu, split, s);
    if (cin == 256 && cout == 128) return launch_img32_pick<256, 128>(ca, relu, split, s);
    if (cin == 128 && cout == 64) return launch_img32_pick<128, 64>(ca, relu, split, s);
    return NASTAR_ERR_UNSUPPORTED;
}

}  // extern "C"

// ---- generic fp16 / f16x3 building blocks (nastar_conv_flat.hip.h): any image size, any channel count -------------------------------
namespace nastar {


template <int NT, bool kFinal, bool kSplit>
static int launch_flat(const FlatConvArgs& fa, hipStream_t s)
{
    auto kern = &nastar_conv3x3_flat_kernel<NT, kFinal, kSplit>;
    const size_t lds = (size_t)FC_PIXB + (size_t)(fa.wide ? FC_WSLOTS : FC_TP + 2 * (fa.W + 1)) * FC_PIXB + (size_t)9 * 4 * NT * 16 + (size_t)NT * 8;
    int rc = ensure_lds(kern, lds);
    if (rc) return rc;
    const unsigned grid = (unsigned)(((fa.ntiles + 7) / 8) * 8 * (fa.COUT / NT));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(FC_THREADS), lds, s, fa);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    return NASTAR_OK;
}

}  // namespace nastar

extern "C" {

int nastar_conv3x3_f16(const uint16_t* in, const uint16_t* in2, const uint16_t* wpack, const float* scale, const float* shift,
                       uint16_t* out, float* out_f32, int B, int H, int W, int c1, int c2, int cout, int flags, float final_mul,
                       void* stream)
{
    const bool relu = flags & NASTAR_CONV_RELU, fin = flags & NASTAR_CONV_FINAL, ups = flags & NASTAR_CONV_UPSAMPLE,
               split = flags & NASTAR_CONV_SPLIT;
    if (!in || !wpack || !scale || !shift || (fin ? !out_f32 : !out) || (c2 > 0 && !in2)) return NASTAR_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || c1 <= 0 || c2 < 0 || cout <= 0) return NASTAR_ERR_BAD_SHAPE;
    if (c1 % FC_KS || c2 % FC_KS || cout % 32 || (fin && cout != 32) || (ups && ((H | W) & 1))) return NASTAR_ERR_UNSUPPORTED;
    const long long npix = (long long)B * H * W;
    const long long widest = (long long)(split ? 2 : 1) * (c1 > c2 ? (c1 > cout ? c1 : cout) : (c2 > cout ? c2 : cout));
    if (npix >= (1ll << 31) || npix * widest >= (1ll << 34)) return NASTAR_ERR_UNSUPPORTED;  // 32-bit offsets in 16-byte units
    if (!aligned16(in) || (in2 && !aligned16(in2)) || !aligned16(wpack) || (out && !aligned16(out))) return NASTAR_ERR_BAD_SHAPE;
    FlatConvArgs fa;
    fa.in = in; fa.in2 = in2; fa.wpack = wpack; fa.scale = scale; fa.shift = shift; fa.out = out; fa.out_f32 = out_f32;
    fa.final_mul = final_mul; fa.B = B; fa.H = H; fa.W = W; fa.C1 = c1; fa.C2 = c2; fa.COUT = cout; fa.npix = (int)npix;
    fa.ups = ups ? 1 : 0; fa.relu = relu ? 1 : 0; fa.raw = (flags & NASTAR_CONV_RAW) ? 1 : 0; fa.ntiles = (int)((npix + FC_TP - 1) / FC_TP);
    fa.wide = W > FC_MAXW ? 1 : 0;  // images wider than the flat tiles' halo allows: 2-D tiles (64 x 4 pixels of one image, framed)
    if (fa.wide) fa.ntiles = B * ((H + FC_WTH - 1) / FC_WTH) * ((W + FC_WTW - 1) / FC_WTW);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (fin) return split ? launch_flat<32, true, true>(fa, s) : launch_flat<32, true, false>(fa, s);
    if (cout % 64 == 0) return split ? launch_flat<64, false, true>(fa, s) : launch_flat<64, false, false>(fa, s);
    return split ? launch_flat<32, false, true>(fa, s) : launch_flat<32, false, false>(fa, s);
}

int nastar_maxpool2x2_f16(const uint16_t* in, uint16_t* out, int B, int H, int W, int C, int split, void* stream)
{
    if (!in || !out) return NASTAR_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return NASTAR_ERR_BAD_SHAPE;
    if ((H | W) & 1 || C % 8) return NASTAR_ERR_UNSUPPORTED;
    const long long total = (long long)B * (H / 2) * (W / 2) * (C / 8);
    const unsigned grid = (unsigned)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (split) hipLaunchKernelGGL(nastar_maxpool2x2_f16_kernel<true>, dim3(grid), dim3(256), 0, s, in, out, B, H, W, C);
    else hipLaunchKernelGGL(nastar_maxpool2x2_f16_kernel<false>, dim3(grid), dim3(256), 0, s, in, out, B, H, W, C);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    return NASTAR_OK;
}

int nastar_encoder_prep_f16(const float* map, const float* start, const float* goal, int plus, long long npix, int cp, int split,
                            uint16_t* out, void* stream)
{
    if (!map || !out || (plus && (!start || !goal))) return NASTAR_ERR_NULL;
    if (npix <= 0 || cp < 2 || cp % 8) return NASTAR_ERR_BAD_SHAPE;
    const unsigned grid = (unsigned)((npix + 255) / 256 < 65536 ? (npix + 255) / 256 : 65536);
    hipLaunchKernelGGL(nastar_encoder_prep_f16_kernel, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), map, start, goal,
                       out, npix, plus, cp, split);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    return NASTAR_OK;
}

}  // extern "C"
