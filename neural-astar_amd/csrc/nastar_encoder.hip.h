// nastar_encoder.hip.h -- the CNN cost-map encoder of NeuralAstar on MFMA (SURVEY.md section 8f "next #1").
//
// Reference: planner/encoder.py:60-78 (CNN: 3x3 convs input -> 32 -> 64 -> 128 -> 256 -> 1 with BatchNorm + ReLU between
// them) and :32-34 (sigmoid(model(x)) * const); input assembly astar.py:171-177 (cat(map, start + goal)).
// Inference (eval-mode BatchNorm folded into a per-channel scale/shift) in bf16 with fp32 accumulation:
//
//   * activations are NHWC bf16, channels padded to a multiple of 16, so the reduction index k = (tap, channel) of the
//     implicit GEMM is contiguous per pixel;
//   * one 256-thread workgroup computes 16 image rows x 32 columns x NT output channels; per 32-channel input slice it
//     stages the (16+2) x (32+2) pixel halo tile and the slice's 9 x 32 x NT weights in LDS, then every wavefront runs
//     v_mfma_f32_32x32x16_bf16 with   A = weights [32 out-channels x 16 k],  B = pixels [16 k x 32 columns of one row],
//     so that C/D = [channel][pixel]: a lane ends up with 4 consecutive channels of its pixel per register quad, i.e.
//     8-byte NHWC stores;
//   * the pixel tile uses a 64-byte pixel stride with the 16-byte chunk index rotated by (column >> 2), which makes every
//     16-lane group of a ds_read_b128 hit 16 distinct 4-bank slots (no padding, no conflicts);
//   * epilogue: y = relu(acc * scale[c] + shift[c]) -> bf16, or for the last layer sigmoid(acc + bias) * const -> fp32
//     [B,1,H,W] (only output channel 0 of the zero-padded 32-channel block is real).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nastar {

typedef __attribute__((ext_vector_type(8))) short bf16x8;   // 8 bf16 = one MFMA A/B operand (4 VGPRs)
typedef __attribute__((ext_vector_type(16))) float f32x16;  // 32x32 accumulator fragment
typedef _Float16 nastar_h8 __attribute__((ext_vector_type(8)));
// one 32x32x16 MFMA; the operand registers hold 8 bf16 or 8 fp16 per lane
template <bool kF16>
__device__ __forceinline__ f32x16 mfma16(bf16x8 a, bf16x8 b, f32x16 c)
{
    if constexpr (kF16)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<nastar_h8*>(&a), *reinterpret_cast<nastar_h8*>(&b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ uint16_t f32_to_bf16_rn(float f)
{
    uint32_t u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);  // round to nearest even (inputs are finite)
    return (uint16_t)(u >> 16);
}
// two floats -> packed bf16 pair, round to nearest even: one v_cvt_pk_bf16_f32 on gfx950 (the integer sequence above costs ~10)
typedef __bf16 nastar_bf16x2 __attribute__((ext_vector_type(2)));
typedef float nastar_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi)
{
    const nastar_f32x2 v = {lo, hi};
    const nastar_bf16x2 b = __builtin_convertvector(v, nastar_bf16x2);
    return *reinterpret_cast<const uint32_t*>(&b);
}
// ---- "f16x3" precision (north-star tolerance 1e-5 on float outputs): every operand is split into two fp16 terms x = hi + lo
// (22 significant bits) and a product is hi*hi + lo*hi + hi*lo on the fp16 MFMA with fp32 accumulation -- 3x the matrix work of the bf16
// path for fp32-grade cost maps.  Activations are stored [hi(C) | lo(C)] per pixel, weights are packed over 3C "virtual" input channels
// [W_hi | W_hi | W_lo] that meet the activation blocks [x_hi | x_lo | x_hi].
typedef _Float16 nastar_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 nastar_f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_f16x2(float lo, float hi)  // round to nearest even
{
    const nastar_f16x2 b = {(_Float16)lo, (_Float16)hi};
    return *reinterpret_cast<const uint32_t*>(&b);
}
__device__ __forceinline__ float f16_residual(float v) { return v - (float)(_Float16)v; }
__device__ __forceinline__ float f16_clamp(float v) { return fminf(fmaxf(v, -65504.f), 65504.f); }
template <bool kF16>
__device__ __forceinline__ float f16_sat(float v) { return kF16 ? f16_clamp(v) : v; }  // fp16 range guard (no-op for bf16)
template <bool kF16>
__device__ __forceinline__ uint32_t pack_pair(float lo, float hi) { return kF16 ? pack_f16x2(lo, hi) : pack_bf16x2(lo, hi); }
__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }

constexpr int ENC_TW = 32;        // tile width  = image width handled per workgroup column block
constexpr int ENC_TH = 16;        // tile height (4 rows per wavefront)
constexpr int ENC_KS = 32;        // input channels per LDS slice
constexpr int ENC_PIX_B = 64;     // bytes per pixel in the LDS tile (32 bf16)
#ifndef ENC_FRAG_BUFS
#define ENC_FRAG_BUFS 1           // register sets for the MFMA operand fragments (2 = explicit double buffering)
#endif
#ifndef ENC_WAVES
#define ENC_WAVES 8               // wavefronts per workgroup: 8 -> two per SIMD, each owning 2 of the tile's 16 rows
#endif
constexpr int ENC_THREADS = ENC_WAVES * 64;
constexpr int ENC_RPW = ENC_TH / ENC_WAVES;  // image rows per wavefront

struct ConvArgs {
    const uint16_t* in;     // [B,H,W,CIN] bf16
    const uint16_t* wpack;  // [9][CIN/8][COUT][8] bf16   (tap = (dy+1)*3 + (dx+1))
    const float* scale;     // [COUT]
    const float* shift;     // [COUT]
    uint16_t* out;          // [B,H,W,COUT] bf16        (kFinal == false)
    float* out_f32;         // [B,H,W] fp32             (kFinal == true: sigmoid(acc*scale+shift) * mul)
    const uint16_t* wfin;   // fused last layer (img32 kernel, kFuse): its packed weights [9][COUT/8][32][8] ...
    const uint16_t* wfin_lo;  // ... split precision (kFuse && kSplit): wfin = the fp16 hi halves, wfin_lo = the lo halves
    const float* fscale;    // ... and its folded BatchNorm scale / shift (1 channel)
    const float* fshift;
    // tap-major last-layer kernel only: channels per input pixel (0 = CIN) and multi-pass accumulation for the f16x3 form
    int in_stride;
    int pass_flags;         // bit 0: add zacc[pixel] to the tap sum; bit 1: store the raw sum to zacc and stop (not the last pass)
    float* zacc;            // [B,H,W] fp32 partial sums
    float final_mul;
    int B, H, W;
};

// LDS pixel tile address (bytes): pixel (ty, tx) of the (TH+2) x (TW+2) halo tile, 16-byte chunk c (0..3), rotated.
__device__ __forceinline__ int enc_tile_off(int ty, int tx, int c)
{
    return (ty * (ENC_TW + 2) + tx) * ENC_PIX_B + (((c + (tx >> 2)) & 3) << 4);
}

// CIN, COUT: padded channel counts (CIN % 16 == 0, COUT % 32 == 0); NT: output channels per workgroup (32 or 64).
template <int CIN, int COUT, int NT, bool kRelu, bool kFinal>
__global__ __launch_bounds__(ENC_THREADS) void nastar_conv3x3_kernel(const ConvArgs a)
{
    constexpr int KS = (CIN < ENC_KS) ? CIN : ENC_KS;      // channels per slice (16 or 32)
    constexpr int KSTEPS = KS / 16;                          // MFMA k-steps per tap and slice
    constexpr int NSLICE = CIN / KS;
    constexpr int NB = NT / 32;                              // 32-channel output blocks per workgroup
    constexpr int TILE_BYTES = (ENC_TH + 2) * (ENC_TW + 2) * ENC_PIX_B;
    // weights in LDS: [tap][kk][khalf][n][8 bf16] = 9 * KSTEPS * 2 * NT * 16 bytes after the pixel tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* tile = smem;
    unsigned char* wl = smem + TILE_BYTES;
    float* ss = reinterpret_cast<float*>(wl + 9 * KSTEPS * 2 * NT * 16);  // scale[NT] | shift[NT] of this channel block

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int tiles_x = a.W / ENC_TW, tiles_y = a.H / ENC_TH;
    int t = blockIdx.x;
    const int nblk = t % (COUT / NT); t /= (COUT / NT);
    const int txb = t % tiles_x; t /= tiles_x;
    const int tyb = t % tiles_y; t /= tiles_y;
    const int b = t;
    const int y0 = tyb * ENC_TH, x0 = txb * ENC_TW;
    const int n0 = nblk * NT;

    f32x16 acc[ENC_RPW][NB];
#pragma unroll
    for (int m = 0; m < ENC_RPW; ++m)
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    const int px = lane & 31;      // pixel column inside the tile (MFMA B column / C column)
    const int kh = lane >> 5;      // which 8-element half of a 16-wide k-step this lane feeds

    // ---- staging: a slice = the (TH+2) x (TW+2) pixel halo tile x KS channels + the 9 x KS x NT weights.  The global loads
    // of slice s+1 are issued into registers BEFORE the MFMAs of slice s and written to LDS after them, so HBM/L2 latency
    // hides behind the matrix pipe.  All per-chunk offsets are computed once (the index arithmetic would otherwise cost
    // ~1000 VALU instructions per slice and wave).
    constexpr int CH16 = KS / 8;                                               // 16-byte chunks per pixel (2 or 4)
    constexpr int NTC = (ENC_TH + 2) * (ENC_TW + 2) * CH16;                    // tile chunks per slice
    constexpr int NWC = 9 * KSTEPS * 2 * NT;                                   // weight chunks per slice
    constexpr int NTQ = (NTC + ENC_THREADS - 1) / ENC_THREADS, NWQ = (NWC + ENC_THREADS - 1) / ENC_THREADS;  // ... per thread
    int t_src[NTQ], t_dst[NTQ], w_src[NWQ];  // element offsets into a.in / a.wpack (slice 0), byte offset into the tile
#pragma unroll
    for (int i = 0; i < NTQ; ++i) {
        const int q = tid + i * ENC_THREADS;
        const int c = q % CH16;
        const int p = q / CH16;
        const int tx = p % (ENC_TW + 2), ty = p / (ENC_TW + 2);
        const int gy = y0 + ty - 1, gx = x0 + tx - 1;
        const bool ok = q < NTC && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
        t_src[i] = ok ? (int)(((gy * a.W + gx) * CIN) + c * 8) : -1;
        t_dst[i] = q < NTC ? enc_tile_off(ty, tx, c) : -1;
    }
#pragma unroll
    for (int i = 0; i < NWQ; ++i) {
        const int q = tid + i * ENC_THREADS;
        const int n = q % NT;
        int r = q / NT;
        const int h = r % 2; r /= 2;
        const int kk = r % KSTEPS; r /= KSTEPS;  // r = tap
        w_src[i] = q < NWC ? (int)((((r * (CIN / 8)) + kk * 2 + h) * COUT + n0 + n) * 8) : -1;
    }
    const uint16_t* in_img = a.in + (size_t)b * a.H * a.W * CIN;
    uint4 tq[NTQ], wq[NWQ];
    auto load_slice = [&](int s) {
#pragma unroll
        for (int i = 0; i < NTQ; ++i) {
            tq[i] = make_uint4(0u, 0u, 0u, 0u);
            if (t_src[i] >= 0) tq[i] = *reinterpret_cast<const uint4*>(in_img + t_src[i] + s * KS);
        }
#pragma unroll
        for (int i = 0; i < NWQ; ++i) {
            wq[i] = make_uint4(0u, 0u, 0u, 0u);
            if (w_src[i] >= 0) wq[i] = *reinterpret_cast<const uint4*>(a.wpack + w_src[i] + (size_t)s * (KS / 8) * COUT * 8);
        }
    };
    auto store_slice = [&]() {
#pragma unroll
        for (int i = 0; i < NTQ; ++i)
            if (t_dst[i] >= 0) *reinterpret_cast<uint4*>(tile + t_dst[i]) = tq[i];
#pragma unroll
        for (int i = 0; i < NWQ; ++i)
            if (w_src[i] >= 0) *reinterpret_cast<uint4*>(wl + (size_t)(tid + i * ENC_THREADS) * 16) = wq[i];  // [tap][kk][khalf][n][8]
    };
    // ---- compute: a stage = (k-step kk, column offset dx).  It needs the RPW+2 tile rows of this wave (shared by the three
    // row offsets dy) and the weights of the 3 taps (dy, dx): RPW+2 + 3*NB ds_read_b128 feed RPW x 3 x NB MFMAs.  The fragments of stage
    // i+1 are read while the MFMAs of stage i run (two register sets), because with one wavefront per SIMD nothing else
    // hides the LDS latency.
    constexpr int NST = 3 * KSTEPS;
    constexpr int NBUF = ENC_FRAG_BUFS;
    bf16x8 xb[NBUF][ENC_RPW + 2], wa[NBUF][3][NB];
    auto load_stage = [&](int st, int buf) {
        const int kk = st / 3, dx = st % 3;
#pragma unroll
        for (int r = 0; r < ENC_RPW + 2; ++r)
            xb[buf][r] = *reinterpret_cast<const bf16x8*>(tile + enc_tile_off(wave * ENC_RPW + r, px + dx, kk * 2 + kh));
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int n = 0; n < NB; ++n)
                wa[buf][dy][n] = *reinterpret_cast<const bf16x8*>(
                    wl + (((((dy * 3 + dx) * KSTEPS + kk) * 2 + kh) * NT) + n * 32 + px) * 16);
    };

    load_slice(0);
    if (tid < NT) {  // epilogue constants: fetched now, read from LDS after the last slice (no exposed global latency there)
        ss[tid] = a.scale[n0 + tid];
        ss[NT + tid] = a.shift[n0 + tid];
    }
    store_slice();
    __syncthreads();
    for (int s = 0; s < NSLICE; ++s) {
        if (s + 1 < NSLICE) load_slice(s + 1);  // in flight during the MFMAs below
        if constexpr (NBUF == 2) load_stage(0, 0);
#pragma unroll
        for (int st = 0; st < NST; ++st) {
            if constexpr (NBUF == 2) {
                if (st + 1 < NST) load_stage(st + 1, (st + 1) & 1);
            } else {
                load_stage(st, 0);
            }
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int m = 0; m < ENC_RPW; ++m)
#pragma unroll
                    for (int n = 0; n < NB; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[st & (NBUF - 1)][dy][n], xb[st & (NBUF - 1)][m + dy],
                                                                            acc[m][n], 0, 0, 0);
        }
        if (s + 1 < NSLICE) {
            __syncthreads();  // every wave is done reading this slice
            store_slice();
            __syncthreads();
        }
    }

    // ---- epilogue: C/D layout col = lane&31 = pixel, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) = channel in block ------
#pragma unroll
    for (int m = 0; m < ENC_RPW; ++m) {
        const int gy = y0 + wave * ENC_RPW + m, gx = x0 + px;
        const size_t pix = ((size_t)b * a.H + gy) * a.W + gx;
        if constexpr (kFinal) {
            if (kh == 0 && nblk == 0) {  // channel 0 = reg 0 of the lanes with lane>>5 == 0
                const float z = acc[m][0][0] * ss[0] + ss[NT];
                a.out_f32[pix] = a.final_mul / (1.0f + __expf(-z));
            }
        } else {
#pragma unroll
            for (int n = 0; n < NB; ++n) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int cl = n * 32 + 8 * g + 4 * kh;  // channel inside this workgroup's block
                    const int c = n0 + cl;
                    const float4 sc = *reinterpret_cast<const float4*>(ss + cl);
                    const float4 sh = *reinterpret_cast<const float4*>(ss + NT + cl);
                    float v0 = acc[m][n][4 * g + 0] * sc.x + sh.x;
                    float v1 = acc[m][n][4 * g + 1] * sc.y + sh.y;
                    float v2 = acc[m][n][4 * g + 2] * sc.z + sh.z;
                    float v3 = acc[m][n][4 * g + 3] * sc.w + sh.w;
                    if constexpr (kRelu) {
                        v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f);
                    }
                    uint2 o;
                    o.x = pack_bf16x2(v0, v1);
                    o.y = pack_bf16x2(v2, v3);
                    *reinterpret_cast<uint2*>(a.out + pix * COUT + c) = o;
                }
            }
        }
    }
}

// ---- 32x32 images: one workgroup = the whole image x 64 output channels -------------------------------------------------
// The generic kernel above reads 0.83 LDS operand fragments per MFMA and drains the matrix pipe at two barriers per slice.
// For 32x32 images (the reference's maze / MPD / TMPD datasets) the tile is the image itself, which buys:
//   * a 4 row x 64 channel register tile per wavefront (8 accumulators): 6 pixel-row + 6 weight fragments feed 24 MFMAs,
//     0.5 ds_read_b128 per MFMA;
//   * 16-channel slices (32 bytes per pixel in LDS, chunk index XORed with bit 3 of the column: any 16 consecutive columns hit
//     16 distinct 16-byte bank slots), small enough for TWO LDS buffers: slice s+1 is written while slice s is being
//     multiplied, one barrier per slice;
//   * the halo is the zero padding, written once; staging copies exactly 4 (pixels) + 2.25 (weights) 16-byte chunks per
//     thread and slice at constant strides (no offset tables).
constexpr int I32_KS = 16, I32_NT = 64, I32_RPW = 4, I32_NB = 2, I32_PIX_B = 32;
constexpr int I32_TILE_BYTES = 34 * 34 * I32_PIX_B;               // 36992
constexpr int I32_W_BYTES = 9 * 2 * I32_NT * 16;                  // 18432: [tap][khalf][n][8 bf16]
constexpr int I32_BUF_BYTES = I32_TILE_BYTES + I32_W_BYTES;       // 55424
constexpr int I32_P_BYTES = 1024 * 9 * 4;                         // fused last layer: per-pixel tap sums P[pixel][9] fp32
constexpr int I32_OB_BYTES = I32_P_BYTES + 8192;                 // epilogue scratch: 8 transpose patches of 4 KB | P + last-layer fragments (hi | lo)
constexpr size_t I32_LDS_BYTES = 2 * (size_t)I32_BUF_BYTES + I32_OB_BYTES + 2 * 256 * 4;

__device__ __forceinline__ int i32_tile_off(int ty, int tx, int c)
{
    return (ty * 34 + tx) * I32_PIX_B + ((c ^ ((tx >> 3) & 1)) << 4);
}

// ---- the MFMA inner loop of the 32x32-tile kernels: one 16-channel slice = 9 steps (dx, dy) of 8 MFMAs ------------------------------
// A step needs pixel rows dy..dy+3 at column offset dx and the 2 weight fragments of tap (dy, dx); the fragments of step t+1 are
// requested BEFORE the MFMAs of step t (one new row + 2 weights inside a column offset, 4 rows + 2 weights when dx advances), so
// LDS latency hides behind 256 matrix-pipe cycles.  The reads are inline asm slotted one per MFMA gap and pinned with
// sched_barriers: left to itself the register-starved compiler sinks every ds_read next to its use.
#define DSR(dst_, addr_, off_) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst_) : "v"(addr_), "n"(off_))
#define WOFF(dx_, dy_, n_) (((((dy_) * 3 + (dx_)) * 2) * I32_NT + (n_) * 32) * 16)
#define ROFF(r_) ((r_) * 34 * I32_PIX_B)
#define SB __builtin_amdgcn_sched_barrier(0)
#define LGKM3(a_, b_, c_) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a_), "+v"(b_), "+v"(c_))
#define LGKM6(a_, b_, c_, d_, e_, f_) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a_), "+v"(b_), "+v"(c_), "+v"(d_), "+v"(e_), "+v"(f_))
// one MFMA of the 4 x 2 register tile (row m of the step's window, channel block n), order pinned
#define MF(m_, n_, W_, R_) acc[m_][n_] = mfma16<kF16>(W_##n_, R_, acc[m_][n_]); SB   /* kF16: constant in the using kernel */
// a step: 8 MFMAs on weights W_{0,1} x rows R0_..R3_, with the next step's fragment reads slotted one per MFMA gap
#define STEP0(W_, R0_, R1_, R2_, R3_) \
    SB; MF(0, 0, W_, R0_); MF(0, 1, W_, R0_); MF(1, 0, W_, R1_); MF(1, 1, W_, R1_);  \
    MF(2, 0, W_, R2_); MF(2, 1, W_, R2_); MF(3, 0, W_, R3_); MF(3, 1, W_, R3_)
#define STEP3(W_, R0_, R1_, R2_, R3_, RD0_, RD1_, RD2_)       \
    SB; MF(0, 0, W_, R0_); RD0_; SB; MF(0, 1, W_, R0_); RD1_; SB; MF(1, 0, W_, R1_); RD2_; SB; MF(1, 1, W_, R1_);        \
    MF(2, 0, W_, R2_); MF(2, 1, W_, R2_); MF(3, 0, W_, R3_); MF(3, 1, W_, R3_)
#define STEP6(W_, R0_, R1_, R2_, R3_, RD0_, RD1_, RD2_, RD3_, RD4_, RD5_) \
    SB; MF(0, 0, W_, R0_); RD0_; SB; MF(0, 1, W_, R0_); RD1_; SB; MF(1, 0, W_, R1_); RD2_; SB; MF(1, 1, W_, R1_); RD3_; SB; \
    MF(2, 0, W_, R2_); RD4_; SB; MF(2, 1, W_, R2_); RD5_; SB; MF(3, 0, W_, R3_); MF(3, 1, W_, R3_)
// the whole slice; cb_ = LDS byte address of the operand buffer, rowb*/wgtb = this lane's fragment offsets, acc = the 4x2 tile
#define I32_SLICE_MFMAS(cb_)                                                                                              \
    do {                                                                                                                  \
        const uint32_t ra0 = (cb_) + rowb0, ra1 = (cb_) + rowb1, ra2 = (cb_) + rowb2, wa_ = (cb_) + wgtb;                 \
        bf16x8 A0, A1, A2, A3, A4, A5, B0, B1, B2, B3, B4, B5, WA0, WA1, WB0, WB1;                                        \
        DSR(A0, ra0, ROFF(0)); DSR(A1, ra0, ROFF(1)); DSR(A2, ra0, ROFF(2)); DSR(A3, ra0, ROFF(3)); \
        DSR(WA0, wa_, WOFF(0, 0, 0)); DSR(WA1, wa_, WOFF(0, 0, 1)); \
        LGKM6(A0, A1, A2, A3, WA0, WA1); \
        STEP3(WA, A0, A1, A2, A3, DSR(A4, ra0, ROFF(4)), DSR(WB0, wa_, WOFF(0, 1, 0)), DSR(WB1, wa_, WOFF(0, 1, 1))); \
        LGKM3(A4, WB0, WB1); \
        STEP3(WB, A1, A2, A3, A4, DSR(A5, ra0, ROFF(5)), DSR(WA0, wa_, WOFF(0, 2, 0)), DSR(WA1, wa_, WOFF(0, 2, 1))); \
        LGKM3(A5, WA0, WA1); \
        STEP6(WA, A2, A3, A4, A5, DSR(B0, ra1, ROFF(0)), DSR(B1, ra1, ROFF(1)), DSR(B2, ra1, ROFF(2)), DSR(B3, ra1, ROFF(3)), DSR(WB0, wa_, WOFF(1, 0, 0)), DSR(WB1, wa_, WOFF(1, 0, 1))); \
        LGKM6(B0, B1, B2, B3, WB0, WB1); \
        STEP3(WB, B0, B1, B2, B3, DSR(B4, ra1, ROFF(4)), DSR(WA0, wa_, WOFF(1, 1, 0)), DSR(WA1, wa_, WOFF(1, 1, 1))); \
        LGKM3(B4, WA0, WA1); \
        STEP3(WA, B1, B2, B3, B4, DSR(B5, ra1, ROFF(5)), DSR(WB0, wa_, WOFF(1, 2, 0)), DSR(WB1, wa_, WOFF(1, 2, 1))); \
        LGKM3(B5, WB0, WB1); \
        STEP6(WB, B2, B3, B4, B5, DSR(A0, ra2, ROFF(0)), DSR(A1, ra2, ROFF(1)), DSR(A2, ra2, ROFF(2)), DSR(A3, ra2, ROFF(3)), DSR(WA0, wa_, WOFF(2, 0, 0)), DSR(WA1, wa_, WOFF(2, 0, 1))); \
        LGKM6(A0, A1, A2, A3, WA0, WA1); \
        STEP3(WA, A0, A1, A2, A3, DSR(A4, ra2, ROFF(4)), DSR(WB0, wa_, WOFF(2, 1, 0)), DSR(WB1, wa_, WOFF(2, 1, 1))); \
        LGKM3(A4, WB0, WB1); \
        STEP3(WB, A1, A2, A3, A4, DSR(A5, ra2, ROFF(5)), DSR(WA0, wa_, WOFF(2, 2, 0)), DSR(WA1, wa_, WOFF(2, 2, 1))); \
        LGKM3(A5, WA0, WA1); \
        STEP0(WA, A2, A3, A4, A5); \
    } while (0)

// kProbe (dev builds only): 1 = per-wave cycle totals written to the output slab, 2 = every workgroup reads image 0 (L2-hit ablation)
// kTiled: the image is H x W with H, W multiples of 32 and a work item is one of its 32x32 tiles: the halo ring (132 pixels) is
// then real data, loaded as a fifth chunk by threads 0..263 (zero outside the image); kTiled = false is the whole-image case above.
// kF16: the f16x3 split-precision form (see pack_f16x2): in [.., 2 CIN] = [hi | lo], weights packed over 3 CIN virtual channels,
// out [.., 2 COUT] = [hi | lo]; 3 CIN / 16 slices per item.
// kF16 alone (kSplit = false): plain fp16 operands, same data flow as bf16 (11 instead of 8 significant bits).
template <int CIN, int COUT, bool kRelu, bool kFuse = false, int kProbe = 0, bool kTiled = false, bool kF16 = false, bool kSplit = kF16>
__global__ __launch_bounds__(512) void nastar_conv3x3_img32_kernel(const ConvArgs a)
{
    static_assert(!(kFuse && kTiled), "the fused last layer needs the whole image in one workgroup");
    static_assert(kF16 || !kSplit, "split precision is an fp16 form");
    constexpr int NS1 = CIN / I32_KS;                 // slices per precision block
    constexpr int NSLICE = (kSplit ? 3 : 1) * NS1;
    constexpr int CINV = (kSplit ? 3 : 1) * CIN;      // virtual input channels of the packed weights
    constexpr int IN_STRIDE = (kSplit ? 2 : 1) * CIN, OUT_STRIDE = (kSplit ? 2 : 1) * COUT;  // channels per pixel in HBM
    constexpr int NGRP = COUT / I32_NT;
    static_assert(NSLICE % 2 == 0, "buffer parity must repeat per work item");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* obase = smem + 2 * I32_BUF_BYTES;                               // epilogue transpose: 8 waves x 4 KB
    float* P = reinterpret_cast<float*>(obase);                                    // (kFuse) tap sums of the last layer
    unsigned char* wf = obase + I32_P_BYTES;                                       // (kFuse) its A fragments for this channel group
    float* ss = reinterpret_cast<float*>(smem + 2 * I32_BUF_BYTES + I32_OB_BYTES);  // scale[COUT] | shift[COUT]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int px = lane & 31, kh = lane >> 5;
    // PERSISTENT workgroups (one per CU: the kernel needs 146 KB of LDS): a workgroup walks work items (image, 64-channel
    // group), so its fixed costs -- launch, zero fill, the exposed first load, the store drain -- are paid once, the loads
    // of the next item's first slice fly during this item's epilogue and its stores drain under the next item's MFMAs.
    // Workgroups are dealt round-robin to the 8 XCDs (each with its own L2): number them so that the channel groups of one
    // image -- which read the same input -- are worked on at the same time on the SAME XCD.
    int wg = blockIdx.x;
    if ((gridDim.x & 7) == 0) wg = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int tiles_x = kTiled ? a.W / 32 : 1, ntile = kTiled ? tiles_x * (a.H / 32) : 1, img_w = kTiled ? a.W : 32;
    const int nitems = a.B * ntile * NGRP;

    // zero both pixel tiles (the halo stays zero = the convolution's padding)
    for (int q = tid; q < I32_TILE_BYTES / 16; q += 512) {
        *reinterpret_cast<uint4*>(smem + q * 16) = make_uint4(0u, 0u, 0u, 0u);
        *reinterpret_cast<uint4*>(smem + I32_BUF_BYTES + q * 16) = make_uint4(0u, 0u, 0u, 0u);
    }
    if (tid < COUT) {
        ss[tid] = a.scale[tid];
        ss[COUT + tid] = a.shift[tid];
    }
    // staging: chunk q = tid + 512 i.  pixels: p = q >> 1 = (tid >> 1) + 256 i -> row (tid >> 6) + 8 i, column (tid >> 1) & 31
    const int t_dst = i32_tile_off((tid >> 6) + 1, ((tid >> 1) & 31) + 1, tid & 1);
    const size_t t_lane = ((size_t)(tid >> 6) * img_w + ((tid >> 1) & 31)) * IN_STRIDE + (tid & 1) * 8;
    // weights: q -> n = q & 63, khalf = (q >> 6) & 1, tap = (q >> 7) = (tid >> 7) + 4 i
    const size_t w_lane = ((size_t)((tid >> 7) * (CINV / 8) + ((tid >> 6) & 1)) * COUT + (tid & 63)) * 8;
    // (named scalars, not arrays: the arrays were left in scratch memory by the compiler)
    uint4 t0, t1, t2, t3, w0, w1, w2 = make_uint4(0u, 0u, 0u, 0u);
    const bool w2on = tid < 128;
    constexpr size_t WSTR = (size_t)4 * (CINV / 8) * COUT * 8, WSL = (size_t)2 * COUT * 8;
    const size_t TSTR = (size_t)8 * img_w * IN_STRIDE;  // 8 image rows
    // halo ring (kTiled): chunk tid < 264 -> ring pixel tid >> 1: top row, bottom row, left column, right column of the 34x34 tile
    uint4 hq = make_uint4(0u, 0u, 0u, 0u);
    int h_dst = 0, h_y = 0, h_x = 0;
    if constexpr (kTiled) {
        const int r = tid >> 1;
        h_y = r < 34 ? 0 : (r < 68 ? 33 : (r < 100 ? r - 67 : r - 99));
        h_x = r < 34 ? r : (r < 68 ? r - 34 : (r < 100 ? 0 : 33));
        h_dst = i32_tile_off(h_y, h_x, tid & 1);
    }
    const bool h_on = kTiled && tid < 264;
    // item -> (image b, tile origin y0/x0, channel group)
#define I32_DECODE(item_, b_, y0_, x0_, g_)                                                              \
    const int g_ = (item_) % NGRP, tl_##b_ = ((item_) / NGRP) % ntile, b_ = (kProbe == 2) ? 0 : (item_) / (NGRP * ntile), \
              y0_ = (tl_##b_ / tiles_x) * 32, x0_ = (tl_##b_ % tiles_x) * 32
#define I32_LOAD_SLICE(item_, s_)                                                                        \
    do {                                                                                                 \
        I32_DECODE(item_, ib_, iy_, ix_, ig_);                                                           \
        /* activation channels of virtual slice s: blocks [x_hi | x_lo | x_hi] */                         \
        const int cs_ = (kSplit && (s_) >= 2 * NS1) ? (s_) - 2 * NS1 : (s_);                             \
        const uint16_t* ip_ = a.in + (((size_t)ib_ * (kTiled ? a.H : 32) + iy_) * img_w + ix_) * IN_STRIDE + cs_ * I32_KS; \
        const uint16_t* tp_ = ip_ + t_lane;                                                              \
        const uint16_t* wp_ = a.wpack + w_lane + ig_ * I32_NT * 8 + (size_t)(s_) * WSL;                  \
        if constexpr (kTiled) {                                                                          \
            const int gy_ = iy_ + h_y - 1, gx_ = ix_ + h_x - 1;                                          \
            hq = make_uint4(0u, 0u, 0u, 0u);                                                             \
            if (h_on && (unsigned)gy_ < (unsigned)a.H && (unsigned)gx_ < (unsigned)a.W)                  \
                hq = *reinterpret_cast<const uint4*>(ip_ + ((ptrdiff_t)(h_y - 1) * img_w + (h_x - 1)) * IN_STRIDE + (tid & 1) * 8); \
        }                                                                                                \
        t0 = *reinterpret_cast<const uint4*>(tp_);                                                       \
        t1 = *reinterpret_cast<const uint4*>(tp_ + TSTR);                                                \
        t2 = *reinterpret_cast<const uint4*>(tp_ + 2 * TSTR);                                            \
        t3 = *reinterpret_cast<const uint4*>(tp_ + 3 * TSTR);                                            \
        w0 = *reinterpret_cast<const uint4*>(wp_);                                                       \
        w1 = *reinterpret_cast<const uint4*>(wp_ + WSTR);                                                \
        if (w2on) w2 = *reinterpret_cast<const uint4*>(wp_ + 2 * WSTR);                                  \
    } while (0)
#define I32_STORE_SLICE(buf_)                                                                       \
    do {                                                                                            \
        unsigned char* b_ = (buf_);                                                                 \
        *reinterpret_cast<uint4*>(b_ + t_dst) = t0;                                                 \
        *reinterpret_cast<uint4*>(b_ + t_dst + 8 * 34 * I32_PIX_B) = t1;                            \
        *reinterpret_cast<uint4*>(b_ + t_dst + 16 * 34 * I32_PIX_B) = t2;                           \
        *reinterpret_cast<uint4*>(b_ + t_dst + 24 * 34 * I32_PIX_B) = t3;                           \
        *reinterpret_cast<uint4*>(b_ + I32_TILE_BYTES + tid * 16) = w0;                             \
        *reinterpret_cast<uint4*>(b_ + I32_TILE_BYTES + (tid + 512) * 16) = w1;                     \
        if (w2on) *reinterpret_cast<uint4*>(b_ + I32_TILE_BYTES + (tid + 1024) * 16) = w2;          \
        if (h_on) *reinterpret_cast<uint4*>(b_ + h_dst) = hq;                                       \
    } while (0)

    // LDS byte offsets of this lane's operand fragments (buffer 0): pixel rows at the three column offsets, weights
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    const uint32_t rowb0 = i32_tile_off(wave * I32_RPW, px + 0, kh), rowb1 = i32_tile_off(wave * I32_RPW, px + 1, kh),
                   rowb2 = i32_tile_off(wave * I32_RPW, px + 2, kh), wgtb = I32_TILE_BYTES + (kh * I32_NT + px) * 16;
    // work item = image * NGRP + group.  Plain: items dealt cyclically.  Fused: a workgroup owns whole images (all groups in turn)
    int item = kFuse ? blockIdx.x * NGRP : wg;
    if (item < nitems) I32_LOAD_SLICE(item, 0);
    __syncthreads();  // zero fill and ss visible
    if (item < nitems) {
        I32_STORE_SLICE(smem);
        __syncthreads();
    }
    // kProbe == 1: per-wave cycle totals of the barrier waits, the slice loops and the epilogues
    long long tk_bar = 0, tk_main = 0, tk_epi = 0, tk0 = 0, tk_start = 0;
    if constexpr (kProbe == 1) tk_start = clock64();
    while (item < nitems) {
        I32_DECODE(item, b, ty0, tx0, grp);
        const int n0 = grp * I32_NT;
        const int next = !kFuse ? item + (int)gridDim.x : (grp == NGRP - 1 ? item + ((int)gridDim.x - 1) * NGRP + 1 : item + 1);
        uint4 wfq = make_uint4(0u, 0u, 0u, 0u), wfq_lo = make_uint4(0u, 0u, 0u, 0u);
        if constexpr (kFuse) {
            // last layer's weights for this channel group as MFMA A fragments: row = tap (lanes 0..8 of each half), k = the 8
            // channels this lane's accumulator registers 8j..8j+7 hold: n0 + 32 n + 16 j + 4 kh + {0..3} and the same + 8
            if (tid < 256) {
                const int f = tid >> 6, l = tid & 63, tap = l & 31, base = n0 + (f >> 1) * 32 + (f & 1) * 16 + 4 * (l >> 5);
                if (tap < 9) {
                    const uint2 lo = *reinterpret_cast<const uint2*>(a.wfin + ((size_t)(tap * (COUT / 8) + (base >> 3)) * 32) * 8 + (base & 7));
                    const uint2 hi = *reinterpret_cast<const uint2*>(a.wfin + ((size_t)(tap * (COUT / 8) + ((base + 8) >> 3)) * 32) * 8 + (base & 7));
                    wfq = make_uint4(lo.x, lo.y, hi.x, hi.y);
                    if constexpr (kSplit) {  // the fp16 lo halves of the same weights (x = hi + lo): third product of the split form
                        const uint2 l2 = *reinterpret_cast<const uint2*>(a.wfin_lo + ((size_t)(tap * (COUT / 8) + (base >> 3)) * 32) * 8 + (base & 7));
                        const uint2 h2 = *reinterpret_cast<const uint2*>(a.wfin_lo + ((size_t)(tap * (COUT / 8) + ((base + 8) >> 3)) * 32) * 8 + (base & 7));
                        wfq_lo = make_uint4(l2.x, l2.y, h2.x, h2.y);
                    }
                }
            }
        }
        f32x16 acc[I32_RPW][I32_NB];
#pragma unroll
        for (int m = 0; m < I32_RPW; ++m)
#pragma unroll
            for (int n = 0; n < I32_NB; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
        if constexpr (kProbe == 1) tk0 = clock64();
#pragma unroll 1
        for (int s = 0; s < NSLICE; ++s) {
            if (s + 1 < NSLICE) I32_LOAD_SLICE(item, s + 1);
            else if (next < nitems) I32_LOAD_SLICE(next, 0);
            I32_SLICE_MFMAS(lds0 + (s & 1) * I32_BUF_BYTES);
            if constexpr (kFuse) {
                if (s == 0 && tid < 256) {  // the previous item's epilogue copied its fragments before the hand-over barrier
                    *reinterpret_cast<uint4*>(wf + tid * 16) = wfq;
                    if constexpr (kSplit) *reinterpret_cast<uint4*>(wf + 4096 + tid * 16) = wfq_lo;
                }
            }
            if (s + 1 < NSLICE) {
                I32_STORE_SLICE(smem + ((s + 1) & 1) * I32_BUF_BYTES);  // last read in iteration s-1, before the previous barrier
                long long tb = 0;
                if constexpr (kProbe == 1) tb = clock64();
                __syncthreads();
                if constexpr (kProbe == 1) tk_bar += clock64() - tb;
            }
        }
        // Hand both operand buffers to the NEXT item before this item's epilogue: its first slice goes to buffer 0 (last read for
        // slice NSLICE-2, i.e. before that slice's barrier) and the barrier that publishes it is the last synchronisation of this
        // item.  The epilogue then runs unsynchronised: the SIMD's older wavefront finishes it first and starts the next item's
        // MFMAs while the younger one is still converting and storing -- half of the epilogue disappears behind the matrix pipe.
        bf16x8 fa[I32_NB][2], fal[I32_NB][2];
        if constexpr (kFuse) {  // (read before the barrier: a fast wave rewrites wf at the end of the next item's first slice)
#pragma unroll
            for (int n = 0; n < I32_NB; ++n)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    fa[n][j] = *reinterpret_cast<const bf16x8*>(wf + ((n * 2 + j) * 64 + lane) * 16);
                    if constexpr (kSplit) fal[n][j] = *reinterpret_cast<const bf16x8*>(wf + 4096 + ((n * 2 + j) * 64 + lane) * 16);
                }
        }
        if (next < nitems) {
            I32_STORE_SLICE(smem);
            long long tb = 0;
            if constexpr (kProbe == 1) tb = clock64();
            __syncthreads();
            if constexpr (kProbe == 1) tk_bar += clock64() - tb;
        }
        if constexpr (kProbe == 1) { const long long t = clock64(); tk_main += t - tk0; tk0 = t; }
        if constexpr (kFuse) {
            // ---- fused last layer (encoder.py:77 conv 256 -> 1, BatchNorm, :32-34 sigmoid * const).  Its 9 taps are the rows of a 1x1
            // convolution P[pixel][tap] += sum_c w[tap][c] y[c][pixel] whose B operand is exactly this lane's freshly rounded bf16
            // outputs (D rows 8j..8j+7 of a 32-channel block = one 16-wide k-step under a fixed channel permutation, which the A
            // fragments in wf follow).  P accumulates over the 4 channel groups in LDS; the 3x3 shifted sum runs once per image.
#pragma unroll
            for (int m = 0; m < I32_RPW; ++m) {
                f32x16 pa;
#pragma unroll
                for (int r = 0; r < 16; ++r) pa[r] = 0.f;
#pragma unroll
                for (int n = 0; n < I32_NB; ++n)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        uint32_t yw[4], yl[4];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int g = 2 * j + h, cl = n0 + n * 32 + 8 * g + 4 * kh;
                            const float4 sc = *reinterpret_cast<const float4*>(ss + cl);
                            const float4 sh = *reinterpret_cast<const float4*>(ss + COUT + cl);
                            float v0 = acc[m][n][4 * g + 0] * sc.x + sh.x;
                            float v1 = acc[m][n][4 * g + 1] * sc.y + sh.y;
                            float v2 = acc[m][n][4 * g + 2] * sc.z + sh.z;
                            float v3 = acc[m][n][4 * g + 3] * sc.w + sh.w;
                            if constexpr (kRelu) {
                                v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f);
                            }
                            if constexpr (kSplit) {  // the same (hi, lo) pair the unfused epilogue would have stored for the next layer
                                v0 = f16_clamp(v0); v1 = f16_clamp(v1); v2 = f16_clamp(v2); v3 = f16_clamp(v3);
                                yl[2 * h + 0] = pack_pair<kF16>(f16_residual(v0), f16_residual(v1));
                                yl[2 * h + 1] = pack_pair<kF16>(f16_residual(v2), f16_residual(v3));
                            }
                            yw[2 * h + 0] = pack_pair<kF16>(f16_sat<kF16>(v0), f16_sat<kF16>(v1));
                            yw[2 * h + 1] = pack_pair<kF16>(f16_sat<kF16>(v2), f16_sat<kF16>(v3));
                        }
                        const uint4 yq = make_uint4(yw[0], yw[1], yw[2], yw[3]);
                        pa = mfma16<kF16>(fa[n][j], *reinterpret_cast<const bf16x8*>(&yq), pa);
                        if constexpr (kSplit) {  // x_lo * W_hi + x_hi * W_lo: the two cross terms of the split product
                            const uint4 ylq = make_uint4(yl[0], yl[1], yl[2], yl[3]);
                            pa = mfma16<kF16>(fa[n][j], *reinterpret_cast<const bf16x8*>(&ylq), pa);
                            pa = mfma16<kF16>(fal[n][j], *reinterpret_cast<const bf16x8*>(&yq), pa);
                        }
                    }
                // D row = tap = (reg & 3) + 8 (reg >> 2) + 4 kh: taps 0-3 / 8 in the lower half-wave, 4-7 in the upper
                float* pp = P + ((wave * I32_RPW + m) * 32 + px) * 9 + 4 * kh;
                if (grp == 0) {
                    pp[0] = pa[0]; pp[1] = pa[1]; pp[2] = pa[2]; pp[3] = pa[3];
                    if (kh == 0) pp[8] = pa[4];
                } else {
                    pp[0] += pa[0]; pp[1] += pa[1]; pp[2] += pa[2]; pp[3] += pa[3];
                    if (kh == 0) pp[8] += pa[4];
                }
            }
            if (grp == NGRP - 1) {
                __syncthreads();  // the shifted sum reads the rows of neighbouring waves
                const float fs = a.fscale[0], fb = a.fshift[0];
                for (int o = tid; o < 1024; o += 512) {
                    const int oy = o >> 5, ox = o & 31;
                    float z = 0.f;
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                        for (int dx = 0; dx < 3; ++dx) {
                            const int yy = oy + dy - 1, xx = ox + dx - 1;
                            if ((unsigned)yy < 32u && (unsigned)xx < 32u) z += P[(yy * 32 + xx) * 9 + dy * 3 + dx];
                        }
                    z = z * fs + fb;
                    a.out_f32[(size_t)b * 1024 + o] = a.final_mul / (1.0f + __expf(-z));
                }
            }
        } else {
            // ---- epilogue: scale/shift/ReLU -> bf16, transposed through a wave-private LDS patch so that 8 lanes write one pixel's
            // 128 contiguous bytes.  D layout: column = lane & 31 = pixel, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) = channel.
            unsigned char* ob = obase + wave * 4096;  // one image row: 32 pixels x 64 channels
#pragma unroll
            for (int m = 0; m < I32_RPW; ++m) {
#pragma unroll
                for (int part = 0; part < (kSplit ? 2 : 1); ++part) {  // f16x3: the hi halves, then the lo halves (v - fp16(v))
#pragma unroll
                    for (int n = 0; n < I32_NB; ++n)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int cl = n0 + n * 32 + 8 * g + 4 * kh;
                            const float4 sc = *reinterpret_cast<const float4*>(ss + cl);
                            const float4 sh = *reinterpret_cast<const float4*>(ss + COUT + cl);
                            float v0 = acc[m][n][4 * g + 0] * sc.x + sh.x;
                            float v1 = acc[m][n][4 * g + 1] * sc.y + sh.y;
                            float v2 = acc[m][n][4 * g + 2] * sc.z + sh.z;
                            float v3 = acc[m][n][4 * g + 3] * sc.w + sh.w;
                            if constexpr (kRelu) {
                                v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f);
                            }
                            if constexpr (kF16) {  // fp16 range: saturate instead of producing inf (BatchNorm-ed activations are O(1..10))
                                v0 = f16_clamp(v0); v1 = f16_clamp(v1); v2 = f16_clamp(v2); v3 = f16_clamp(v3);
                            }
                            if (part == 1) {
                                v0 = f16_residual(v0); v1 = f16_residual(v1); v2 = f16_residual(v2); v3 = f16_residual(v3);
                            }
                            uint2 o;
                            o.x = pack_pair<kF16>(v0, v1);
                            o.y = pack_pair<kF16>(v2, v3);
                            const int chunk = n * 4 + g;  // 16-byte chunk of the pixel's 128 bytes, swizzled by the column
                            *reinterpret_cast<uint2*>(ob + px * 128 + ((chunk ^ ((px >> 1) & 7)) << 4) + kh * 8) = o;
                        }
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int p = j * 8 + (lane >> 3), chunk = lane & 7;
                        const uint4 v = *reinterpret_cast<const uint4*>(ob + p * 128 + ((chunk ^ ((p >> 1) & 7)) << 4));
                        const size_t pix = ((size_t)b * (kTiled ? a.H : 32) + ty0 + wave * I32_RPW + m) * img_w + tx0 + p;
                        *reinterpret_cast<uint4*>(a.out + pix * OUT_STRIDE + part * COUT + n0 + chunk * 8) = v;
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
        if constexpr (kProbe == 1) tk_epi += clock64() - tk0;
        item = next;
    }
    if constexpr (kProbe == 1) {
        if (lane == 0) {
            long long* dbg = reinterpret_cast<long long*>(a.out) + ((size_t)blockIdx.x * 8 + wave) * 4;
            dbg[0] = clock64() - tk_start; dbg[1] = tk_bar; dbg[2] = tk_main; dbg[3] = tk_epi;
        }
    }
}

#undef I32_DECODE
#undef I32_LOAD_SLICE
#undef I32_STORE_SLICE

// ---- last layer (256 -> 1 channel) + sigmoid * const -------------------------------------------------------------------
// With ONE output channel an implicit GEMM wastes 31 of the 32 MFMA rows.  Instead the 9 taps become the "output channels"
// of a 1x1 convolution:  P[p][t] = sum_c x[p][c] * w[t][c]  for every pixel p of the halo tile (MFMA rows = taps), and the
// 3x3 convolution is the shifted sum  out[y][x] = sum_t P[(y+dy, x+dx)][t].  9x fewer MFMAs than the padded GEMM; the layer
// becomes a pure stream over its 512-byte-per-pixel input.
template <int CIN, bool kF16 = false>
__global__ __launch_bounds__(ENC_THREADS) void nastar_conv3x3_final_kernel(const ConvArgs a)
{
    const int istr = a.in_stride > 0 ? a.in_stride : CIN;
    constexpr int KS = ENC_KS, KSTEPS = KS / 16, NSLICE = CIN / KS, CH16 = KS / 8;
    constexpr int HP = (ENC_TH + 2) * (ENC_TW + 2);       // halo pixels (612)
    constexpr int NBLK = (HP + 31) / 32;                  // 32-pixel MFMA column blocks (20)
    constexpr int BPW = (NBLK + ENC_WAVES - 1) / ENC_WAVES;
    constexpr int TILE_BYTES = HP * ENC_PIX_B;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* tile = smem;
    float* part = reinterpret_cast<float*>(smem + TILE_BYTES);  // [NBLK*32][9] partial sums

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_x = a.W / ENC_TW, tiles_y = a.H / ENC_TH;
    int t = blockIdx.x;
    const int txb = t % tiles_x; t /= tiles_x;
    const int tyb = t % tiles_y; t /= tiles_y;
    const int b = t;
    const int y0 = tyb * ENC_TH, x0 = txb * ENC_TW;
    const int px = lane & 31, kh = lane >> 5;

    // A operand: row = tap (lanes with px < 9), all CIN channels of output channel 0, kept in registers
    bf16x8 wa[NSLICE * KSTEPS];
#pragma unroll
    for (int ks = 0; ks < NSLICE * KSTEPS; ++ks) {
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (px < 9) v = *reinterpret_cast<const uint4*>(a.wpack + (((size_t)px * (CIN / 8) + ks * 2 + kh) * 32 + 0) * 8);
        wa[ks] = *reinterpret_cast<bf16x8*>(&v);
    }
    // staging offsets (see nastar_conv3x3_kernel)
    constexpr int NTC = HP * CH16, NTQ = (NTC + ENC_THREADS - 1) / ENC_THREADS;
    int t_src[NTQ], t_dst[NTQ];
#pragma unroll
    for (int i = 0; i < NTQ; ++i) {
        const int q = tid + i * ENC_THREADS;
        const int c = q % CH16, p = q / CH16;
        const int tx = p % (ENC_TW + 2), ty = p / (ENC_TW + 2);
        const int gy = y0 + ty - 1, gx = x0 + tx - 1;
        const bool ok = q < NTC && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
        t_src[i] = ok ? (int)(((gy * a.W + gx) * istr) + c * 8) : -1;
        t_dst[i] = q < NTC ? enc_tile_off(ty, tx, c) : -1;
    }
    const uint16_t* in_img = a.in + (size_t)b * a.H * a.W * istr;
    uint4 tq[NTQ];
    auto load_slice = [&](int s) {
#pragma unroll
        for (int i = 0; i < NTQ; ++i) {
            tq[i] = make_uint4(0u, 0u, 0u, 0u);
            if (t_src[i] >= 0) tq[i] = *reinterpret_cast<const uint4*>(in_img + t_src[i] + s * KS);
        }
    };
    auto store_slice = [&]() {
#pragma unroll
        for (int i = 0; i < NTQ; ++i)
            if (t_dst[i] >= 0) *reinterpret_cast<uint4*>(tile + t_dst[i]) = tq[i];
    };
    // this lane's pixel in each of the wave's column blocks
    int boff[BPW];
#pragma unroll
    for (int j = 0; j < BPW; ++j) {
        int hp = (wave + j * ENC_WAVES) * 32 + px;
        hp = hp < HP ? hp : HP - 1;  // padding columns of the last block recompute the last pixel (never read back)
        boff[j] = hp;
    }
    f32x16 acc[BPW];
#pragma unroll
    for (int j = 0; j < BPW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    load_slice(0);
    store_slice();
    __syncthreads();
    for (int s = 0; s < NSLICE; ++s) {
        if (s + 1 < NSLICE) load_slice(s + 1);
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk)
#pragma unroll
            for (int j = 0; j < BPW; ++j) {
                if ((wave + j * ENC_WAVES) < NBLK) {  // wave-uniform
                    const int ty = boff[j] / (ENC_TW + 2), tx = boff[j] % (ENC_TW + 2);
                    const bf16x8 xb = *reinterpret_cast<const bf16x8*>(tile + enc_tile_off(ty, tx, kk * 2 + kh));
                    acc[j] = mfma16<kF16>(wa[s * KSTEPS + kk], xb, acc[j]);
                }
            }
        if (s + 1 < NSLICE) {
            __syncthreads();
            store_slice();
            __syncthreads();
        }
    }
    // partial sums to LDS: D row = tap = (reg&3) + 8*(reg>>2) + 4*kh, column = pixel
#pragma unroll
    for (int j = 0; j < BPW; ++j) {
        if ((wave + j * ENC_WAVES) < NBLK) {
            float* dst = part + ((wave + j * ENC_WAVES) * 32 + px) * 9;
            if (kh == 0) {
                dst[0] = acc[j][0]; dst[1] = acc[j][1]; dst[2] = acc[j][2]; dst[3] = acc[j][3];
                dst[8] = acc[j][4];
            } else {
                dst[4] = acc[j][0]; dst[5] = acc[j][1]; dst[6] = acc[j][2]; dst[7] = acc[j][3];
            }
        }
    }
    __syncthreads();
    // shifted sum + BatchNorm(1 channel) + sigmoid * const  (encoder.py:32-34)
    for (int o = tid; o < ENC_TH * ENC_TW; o += ENC_THREADS) {
        const int oy = o / ENC_TW, ox = o % ENC_TW;
        float z = 0.f;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) z += part[((oy + dy) * (ENC_TW + 2) + ox + dx) * 9 + dy * 3 + dx];
        const size_t pi = ((size_t)b * a.H + y0 + oy) * a.W + x0 + ox;
        if (a.pass_flags & 1) z += a.zacc[pi];
        if (a.pass_flags & 2) {
            a.zacc[pi] = z;
            continue;
        }
        z = z * a.scale[0] + a.shift[0];
        a.out_f32[pi] = a.final_mul / (1.0f + (kF16 ? expf(-z) : __expf(-z)));
    }
}

// ---- stem for 32x32 maps: input assembly + conv 2->32 + conv 32->64 in ONE persistent kernel -------------------------------------
// The first two layers are latency-, not matrix-bound (a 2-slice item cannot hide its own operand loads, and the 2->32 layer is a
// 100 MB round trip through HBM for 0.3 % of the FLOPs).  Here a workgroup walks images: it packs (map, start+goal) into a 34x34
// bf16x2 patch in LDS, computes the 32 first-layer channels with 2 MFMAs per image row (K = 9 taps x 2 channels = 18 of 32) and
// writes them -- scaled, ReLUed, rounded to bf16 -- straight into the two 16-channel operand tiles of the second layer, whose
// weights (both slices, 37 KB) were staged once per workgroup.  The only global reads per image are its 12 KB of fp32 maps,
// prefetched one image ahead; the first-layer activations never exist in HBM.
struct StemArgs {
    const float* map; const float* start; const float* goal;   // [B,32,32] fp32 (start/goal unused when plus == 0)
    const uint16_t* w1; const float* scale1; const float* shift1;   // layer 1: packed [9][2][32][8] (cin padded to 16), 32 channels
    const uint16_t* w2; const float* scale2; const float* shift2;   // layer 2: packed [9][4][64][8], 64 channels
    uint16_t* out;                                                  // [B,32,32,64] bf16
    int B, plus;
};
constexpr int STEM_RAW_BYTES = 34 * 34 * 4;
constexpr size_t STEM_LDS_BYTES = 2 * (size_t)I32_BUF_BYTES + 8 * 4096 + STEM_RAW_BYTES + (64 + 128) * 4;

template <bool kF16 = false>   // operands bf16 or plain fp16 (weights packed accordingly by the host)
__global__ __launch_bounds__(512) void nastar_conv_stem32_kernel(const StemArgs a)
{
    constexpr int CIN = 32, COUT = 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* obase = smem + 2 * I32_BUF_BYTES;                       // epilogue transpose patches, 8 x 4 KB
    uint32_t* raw = reinterpret_cast<uint32_t*>(obase + 8 * 4096);         // [34][34] bf16x2 (map, start+goal), zero halo
    float* ss1 = reinterpret_cast<float*>(obase + 8 * 4096 + STEM_RAW_BYTES);  // scale1[32] | shift1[32]
    float* ss2 = ss1 + 64;                                                 // scale2[64] | shift2[64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int px = lane & 31, kh = lane >> 5;

    for (int q = tid; q < I32_TILE_BYTES / 16; q += 512) {
        *reinterpret_cast<uint4*>(smem + q * 16) = make_uint4(0u, 0u, 0u, 0u);
        *reinterpret_cast<uint4*>(smem + I32_BUF_BYTES + q * 16) = make_uint4(0u, 0u, 0u, 0u);
    }
    for (int q = tid; q < 34 * 34; q += 512) raw[q] = 0u;
    if (tid < 32) { ss1[tid] = a.scale1[tid]; ss1[32 + tid] = a.shift1[tid]; }
    if (tid < 64) { ss2[tid] = a.scale2[tid]; ss2[64 + tid] = a.shift2[tid]; }
    // second-layer weights, both slices, once: chunk q = tid + 512 i -> n = q & 63, khalf = (q >> 6) & 1, tap = q >> 7
    {
        const size_t w_lane = ((size_t)((tid >> 7) * (CIN / 8) + ((tid >> 6) & 1)) * COUT + (tid & 63)) * 8;
        constexpr size_t WSTR = (size_t)4 * (CIN / 8) * COUT * 8, WSL = (size_t)2 * COUT * 8;
#pragma unroll
        for (int sl = 0; sl < 2; ++sl)
#pragma unroll
            for (int i = 0; i < 3; ++i)
                if (i < 2 || tid < 128)
                    *reinterpret_cast<uint4*>(smem + sl * I32_BUF_BYTES + I32_TILE_BYTES + (tid + i * 512) * 16) =
                        *reinterpret_cast<const uint4*>(a.w2 + w_lane + i * WSTR + sl * WSL);
    }
    // first-layer A fragments (row = output channel px, k = (tap, channel)): k-step 0 holds taps 0..7, k-step 1 tap 8
    bf16x8 w1a, w1b;
    {
        uint16_t e0[8], e1[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int tap = kh * 4 + (i >> 1), ch = i & 1;
            e0[i] = a.w1[((size_t)(tap * 2) * 32 + px) * 8 + ch];
            e1[i] = (kh == 0 && i < 2) ? a.w1[((size_t)(8 * 2) * 32 + px) * 8 + ch] : (uint16_t)0;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) { w1a[i] = (short)e0[i]; w1b[i] = (short)e1[i]; }
    }
    // this lane's patch offsets of the 4 taps it feeds (k-step 0) and of tap 8, relative to (image row, column px)
    int toff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int t = kh * 4 + j;
        toff[j] = (t / 3) * 34 + (t % 3) + px;
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    const uint32_t rowb0 = i32_tile_off(wave * I32_RPW, px + 0, kh), rowb1 = i32_tile_off(wave * I32_RPW, px + 1, kh),
                   rowb2 = i32_tile_off(wave * I32_RPW, px + 2, kh), wgtb = I32_TILE_BYTES + (kh * I32_NT + px) * 16;

    // raw input prefetch: this thread owns pixels tid and tid + 512 of the image
    float m0 = 0.f, m1 = 0.f, g0 = 0.f, g1 = 0.f;
    auto fetch = [&](int b) {
        const size_t o = (size_t)b * 1024 + tid;
        m0 = a.map[o]; m1 = a.map[o + 512];
        if (a.plus) { g0 = a.start[o] + a.goal[o]; g1 = a.start[o + 512] + a.goal[o + 512]; }
    };
    auto put_raw = [&]() {
        raw[((tid >> 5) + 1) * 34 + (tid & 31) + 1] = pack_pair<kF16>(m0, g0);
        raw[((tid >> 5) + 17) * 34 + (tid & 31) + 1] = pack_pair<kF16>(m1, g1);
    };
    int b = blockIdx.x;
    if (b < a.B) fetch(b);
    __syncthreads();   // zero fill, weights, constants
    if (b < a.B) put_raw();
    for (; b < a.B; b += gridDim.x) {
        const int next = b + gridDim.x;
        if (next < a.B) fetch(next);
        __syncthreads();   // patch visible; every wave is past the previous image's MFMAs, the tiles may be rewritten
        // ---- layer 1 for this wave's 4 rows -> the two operand tiles of layer 2 ---------------------------------------------------
#pragma unroll
        for (int m = 0; m < I32_RPW; ++m) {
            const int y = wave * I32_RPW + m;
            const uint32_t* rp = raw + y * 34;
            const uint4 q0 = make_uint4(rp[toff[0]], rp[toff[1]], rp[toff[2]], rp[toff[3]]);
            const uint4 q1 = make_uint4(kh == 0 ? rp[2 * 34 + 2 + px] : 0u, 0u, 0u, 0u);
            f32x16 a1;
#pragma unroll
            for (int r = 0; r < 16; ++r) a1[r] = 0.f;
            a1 = mfma16<kF16>(w1a, *reinterpret_cast<const bf16x8*>(&q0), a1);
            a1 = mfma16<kF16>(w1b, *reinterpret_cast<const bf16x8*>(&q1), a1);
#pragma unroll
            for (int g = 0; g < 4; ++g) {  // D rows 8g + 4kh + {0..3} = first-layer channels; g >> 1 = second-layer slice
                const int c = 8 * g + 4 * kh;
                const float4 sc = *reinterpret_cast<const float4*>(ss1 + c);
                const float4 sh = *reinterpret_cast<const float4*>(ss1 + 32 + c);
                const float v0 = fmaxf(a1[4 * g + 0] * sc.x + sh.x, 0.f), v1 = fmaxf(a1[4 * g + 1] * sc.y + sh.y, 0.f);
                const float v2 = fmaxf(a1[4 * g + 2] * sc.z + sh.z, 0.f), v3 = fmaxf(a1[4 * g + 3] * sc.w + sh.w, 0.f);
                uint2 o;
                o.x = pack_pair<kF16>(f16_sat<kF16>(v0), f16_sat<kF16>(v1));
                o.y = pack_pair<kF16>(f16_sat<kF16>(v2), f16_sat<kF16>(v3));
                *reinterpret_cast<uint2*>(smem + (g >> 1) * I32_BUF_BYTES + i32_tile_off(y + 1, px + 1, g & 1) + kh * 8) = o;
            }
        }
        __syncthreads();   // tiles complete; the patch is free again
        if (next < a.B) put_raw();
        // ---- layer 2: both slices are resident, no synchronisation in between -------------------------------------------------------
        f32x16 acc[I32_RPW][I32_NB];
#pragma unroll
        for (int m = 0; m < I32_RPW; ++m)
#pragma unroll
            for (int n = 0; n < I32_NB; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
        I32_SLICE_MFMAS(lds0);
        I32_SLICE_MFMAS(lds0 + I32_BUF_BYTES);
        // ---- epilogue (as in nastar_conv3x3_img32_kernel) ------------------------------------------------------------------------------
        unsigned char* ob = obase + wave * 4096;
#pragma unroll
        for (int m = 0; m < I32_RPW; ++m) {
#pragma unroll
            for (int n = 0; n < I32_NB; ++n)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int cl = n * 32 + 8 * g + 4 * kh;
                    const float4 sc = *reinterpret_cast<const float4*>(ss2 + cl);
                    const float4 sh = *reinterpret_cast<const float4*>(ss2 + COUT + cl);
                    const float v0 = fmaxf(acc[m][n][4 * g + 0] * sc.x + sh.x, 0.f), v1 = fmaxf(acc[m][n][4 * g + 1] * sc.y + sh.y, 0.f);
                    const float v2 = fmaxf(acc[m][n][4 * g + 2] * sc.z + sh.z, 0.f), v3 = fmaxf(acc[m][n][4 * g + 3] * sc.w + sh.w, 0.f);
                    uint2 o;
                    o.x = pack_pair<kF16>(f16_sat<kF16>(v0), f16_sat<kF16>(v1));
                    o.y = pack_pair<kF16>(f16_sat<kF16>(v2), f16_sat<kF16>(v3));
                    const int chunk = n * 4 + g;
                    *reinterpret_cast<uint2*>(ob + px * 128 + ((chunk ^ ((px >> 1) & 7)) << 4) + kh * 8) = o;
                }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int p = j * 8 + (lane >> 3), chunk = lane & 7;
                const uint4 v = *reinterpret_cast<const uint4*>(ob + p * 128 + ((chunk ^ ((p >> 1) & 7)) << 4));
                const size_t pix = (size_t)b * 1024 + (wave * I32_RPW + m) * 32 + p;
                *reinterpret_cast<uint4*>(a.out + pix * COUT + chunk * 8) = v;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// ---- f16x3 form of the first layer: input assembly + conv (1|2) -> 32 + BatchNorm + ReLU in plain fp32 on the vector ALU ---------------
// 18 multiply-adds per output: not worth a matrix instruction, and exact.  One thread per pixel; output [B,H,W,64] fp16 = [hi(32) | lo(32)]
// (or [B,H,W,32] = hi only for the plain fp16 form).
template <int CINR, bool kLo>  // CINR real input channels: 2 ("m+": map, start+goal) or 1 ("m"); kLo: also emit the lo halves
__global__ __launch_bounds__(256) void nastar_conv_first_f32_kernel(const float* __restrict__ map, const float* __restrict__ start,
                                                                   const float* __restrict__ goal, const float* __restrict__ w,
                                                                   const float* __restrict__ scale, const float* __restrict__ shift,
                                                                   uint16_t* __restrict__ out, int B, int H, int W)
{
    // weights / scale / shift are indexed with compile-time constants only: wave-uniform scalar loads, multiply-adds straight from SGPRs.
    // One thread computes one pixel (all 32 channels), but a pixel's 64 / 128 output bytes written by ONE lane make every store
    // instruction touch 64 different lines with 16 bytes each (measured 222 us per 1024 maps for 146 MB of traffic).  The results
    // therefore go through a wave-private LDS patch (144-byte pixel stride: conflict-free both ways) and leave as full-line stores:
    // each store instruction writes 1 KB of consecutive bytes.
    constexpr int OPB = kLo ? 128 : 64;          // output bytes per pixel
    constexpr int LSTR = 144;                    // LDS bytes per pixel
    __shared__ __attribute__((aligned(16))) unsigned char patch[4][64 * LSTR];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned char* mine = patch[wave];
    const long long npix = (long long)B * H * W;
    const long long nround = (npix + 255) / 256;
    for (long long rd = blockIdx.x; rd < nround; rd += gridDim.x) {
        const long long w0 = rd * 256 + wave * 64;  // first pixel of this wavefront
        const long long i = w0 + lane;
        if (i < npix) {
            const int x = (int)(i % W), y = (int)((i / W) % H);
            float in[CINR][9];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
                const bool ok = (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
                const long long j = i + (long long)(t / 3 - 1) * W + (t % 3 - 1);
                in[0][t] = ok ? map[j] : 0.f;
                if constexpr (CINR == 2) in[1][t] = ok ? start[j] + goal[j] : 0.f;
            }
            uint32_t hi[16], lo[16];
#pragma unroll
            for (int c = 0; c < 32; c += 2) {
                float v[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    float z = 0.f;
#pragma unroll
                    for (int ci = 0; ci < CINR; ++ci)
#pragma unroll
                        for (int t = 0; t < 9; ++t) z += w[((c + e) * CINR + ci) * 9 + t] * in[ci][t];  // [32][cin][3][3], the torch layout
                    v[e] = f16_clamp(fmaxf(z * scale[c + e] + shift[c + e], 0.f));
                }
                hi[c >> 1] = pack_f16x2(v[0], v[1]);
                lo[c >> 1] = pack_f16x2(f16_residual(v[0]), f16_residual(v[1]));
            }
            uint4* dst = reinterpret_cast<uint4*>(mine + lane * LSTR);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                dst[q] = make_uint4(hi[4 * q], hi[4 * q + 1], hi[4 * q + 2], hi[4 * q + 3]);
                if constexpr (kLo) dst[4 + q] = make_uint4(lo[4 * q], lo[4 * q + 1], lo[4 * q + 2], lo[4 * q + 3]);
            }
        }
        __builtin_amdgcn_wave_barrier();  // LDS accesses of one wavefront execute in order: a compiler-level ordering point is enough
        constexpr int CPP = OPB / 16;     // 16-byte chunks per pixel (4 or 8)
        unsigned char* gout = reinterpret_cast<unsigned char*>(out) + (size_t)w0 * OPB;
#pragma unroll
        for (int r = 0; r < CPP; ++r) {
            const int k = r * 64 + lane;  // chunk index inside the wavefront's 64-pixel block
            const int p = k / CPP, c = k % CPP;
            if (w0 + p < npix) *reinterpret_cast<uint4*>(gout + (size_t)k * 16) = *reinterpret_cast<const uint4*>(mine + p * LSTR + c * 16);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// input assembly (astar.py:171-177): x0[b][y][x][0] = map, [1] = start + goal, channels 2..15 = 0   (bf16 NHWC, 16 ch)
__attribute__((unused)) static __global__ __launch_bounds__(256) void nastar_encoder_prep_kernel(const float* __restrict__ map, const float* __restrict__ start,
                                                                  const float* __restrict__ goal, uint16_t* __restrict__ x0,
                                                                  long long npix, int plus)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long long)gridDim.x * blockDim.x) {
        const uint32_t c0 = f32_to_bf16_rn(map[i]);
        const uint32_t c1 = plus ? f32_to_bf16_rn(start[i] + goal[i]) : 0u;
        uint4* dst = reinterpret_cast<uint4*>(x0 + i * 16);
        dst[0] = make_uint4(c0 | (c1 << 16), 0u, 0u, 0u);
        dst[1] = make_uint4(0u, 0u, 0u, 0u);
    }
}

}  // namespace nastar
