// nastar_conv_flat.hip.h -- a 3x3 convolution layer on the fp16 MFMA for ANY image size and channel count: the building block of the
// U-Net cost-map encoder (reference planner/encoder.py:37-57, Unet(vgg16_bn): conv stages at 32x32 ... 2x2 pixels, 64 ... 1024 input
// channels, nearest x2 upsampling + skip concatenation in the decoder) and of everything else the fixed-shape kernels of
// nastar_encoder.hip.h do not cover.
//
//   out[p][n] = act( scale[n] * sum_{tap, c} w[tap][c][n] * x[p + tap][c] + shift[n] )          (conv2d, padding 1, folded BatchNorm)
//
// Design (gfx950, wave64):
//   * pixels are a FLAT index p = (b*H + y)*W + x over the whole batch, activations NHWC fp16; a workgroup (4 wavefronts) owns 256
//     consecutive pixels x NT output channels -- 8 rows of a 32x32 map, 16 whole 4x4 maps or 64 whole 2x2 maps alike, so the deep,
//     tiny levels of the U-Net fill the 32-wide MFMA columns exactly like the shallow ones;
//   * a neighbour (y+dy, x+dx) of pixel p is flat pixel p + dy*W + dx when it lies inside the image: the workgroup stages the flat
//     range [p0 - W - 1, p0 + 256 + W + 1) of the current 32-channel slice in LDS (64 B per pixel, 16-byte chunk index rotated by
//     slot/4: conflict-free ds_read_b128 for 32 consecutive pixels) and a lane whose neighbour falls outside the image reads a
//     64-byte all-zero slot instead (conv2d zero padding = ONE address select per tap, computed once per workgroup);
//   * v_mfma_f32_32x32x16_f16 with A = weights [32 out-channels x 16 k], B = pixels [16 k x 32 pixels]; per wavefront a 2 x NT/32
//     register tile (64 pixels x NT channels); the slice's 9 x 32 x NT weights sit in LDS next to the pixels; the global loads of slice
//     s+1 are issued before the MFMAs of slice s (registers) and written to LDS after them; two workgroups per CU overlap the rest;
//   * the input gather is where the decoder's upsample + concat happens: channels [0, C1) come from `in` -- read at (y/2, x/2) of the
//     half-resolution tensor when `ups` is set (F.interpolate(scale_factor=2, mode="nearest")) -- and channels [C1, C1+C2) from the
//     skip tensor `in2` at (y, x): torch.cat never materialises;
//   * kSplit ("f16x3", fp32-grade): activations are [hi(C) | lo(C)] fp16 pairs per pixel, the weights are packed over 3C virtual
//     channels [W_hi | W_hi | W_lo] that meet the activation segments [x_hi | x_lo | x_hi] (nastar_encoder.hip.h); the epilogue
//     emits both halves;
//   * images wider than 126 pixels (round 6; the reference has no size limit, encoder.py:60-97, and its own test drives 64x128): the same
//     kernel on 2-D tiles -- 64 columns x 4 rows of ONE image, one tile row per wavefront, staged with a one-pixel frame at pitch 66 (396
//     slots); frame slots outside the image are staged as zeros, so a tap is again a constant slot offset (dy * 66 + dx) and the matrix
//     loop is untouched: only the staging plan, the read plan and the epilogue's pixel index know about the mode (`wide`, wave-uniform);
//   * epilogue: y = acc*scale + shift (folded BatchNorm / bias), optional ReLU, fp16 NHWC stores of 4 channels per lane (the MFMA D
//     layout); kFinal: channel 0 only, sigmoid(y) * final_mul as fp32 [B,H,W] (encoder.py:32-34).
// Workgroup ids are remapped so that the NT-channel blocks of one pixel tile run on the same XCD back to back (they re-read the
// same activations: one L2).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nastar_encoder.hip.h"

namespace nastar {

constexpr int FC_TP = 256;       // flat pixels per workgroup
constexpr int FC_KS = 32;        // channels per LDS slice (two MFMA k-steps)
constexpr int FC_PIXB = 64;      // bytes per pixel slot in LDS
constexpr int FC_THREADS = 256;
constexpr int FC_MAXW = 126;     // widest image row the FLAT tiles' staging registers cover (tile slots = 256 + 2W + 2 <= 8*64); wider images: 2-D tiles
constexpr int FC_WTW = 64, FC_WTH = 4;              // 2-D tile of the wide mode: 64 columns x 4 rows = the same 256 pixels, one tile row per wavefront
constexpr int FC_WPITCH = FC_WTW + 2;               // its LDS pitch: a one-pixel frame around the tile ...
constexpr int FC_WSLOTS = (FC_WTH + 2) * FC_WPITCH; // ... 6 x 66 = 396 slots (<= 512)
constexpr int FC_NTQ = 8;        // 16-byte pixel chunks staged per thread and slice

struct FlatConvArgs {
    const uint16_t* in;    // [B, H(/2), W(/2), C1 (x2 when split)] fp16
    const uint16_t* in2;   // [B, H, W, C2 (x2)] fp16 skip tensor, or null (C2 = 0)
    const uint16_t* wpack; // [9][CINV/8][COUT][8] fp16, CINV = (C1+C2) or 3*(C1+C2) virtual channels (split)
    const float* scale;    // [COUT]
    const float* shift;    // [COUT]
    uint16_t* out;         // [B, H, W, COUT (x2)] fp16                      (!kFinal)
    float* out_f32;        // [B, H, W] fp32 = sigmoid(y[channel 0]) * mul   (kFinal)
    float final_mul;
    int B, H, W;           // OUTPUT geometry (= input geometry of `in2`; `in` is half of it when ups)
    int C1, C2, COUT;
    int npix;              // B*H*W
    int ups;               // 1: `in` is [B, H/2, W/2, C1], nearest-upsampled x2 on the fly
    int relu;
    int raw;               // kFinal: store y[channel 0] itself instead of sigmoid(y) * final_mul (training: BatchNorm follows)
    int ntiles;            // ceil(npix / FC_TP)  (wide: B * ceil(H / 4) * ceil(W / 64))
    int wide;              // 1: W > FC_MAXW -- 2-D tiles (FC_WTW x FC_WTH pixels of ONE image with a staged one-pixel frame) instead of flat pixel ranges
};

__device__ __forceinline__ int fc_slot_off(int slot, int c) { return FC_PIXB + slot * FC_PIXB + (((c + (slot >> 2)) & 3) << 4); }

// Tried in round 3 and dropped: NT = 128 (a 2 x 4 accumulator tile per wavefront: 6 fragment reads per 8 MFMAs instead of 4 per 4, 240 VGPRs +
// 128 AGPRs, ONE workgroup per CU): 15-50 % SLOWER on every layer it applies to (U-Net, 4096 maps: 64->128 @16x16 0.230 -> 0.289 ms,
// 256->256 @8x8 0.310 -> 0.408 ms, 512->512 @4x4 0.302 -> 0.314 ms): with a single workgroup per CU nothing covers the slice barriers and
// the staging latency; the second resident workgroup is worth more than the saved fragment reads.
template <int NT, bool kFinal, bool kSplit>
__global__ __launch_bounds__(FC_THREADS, 2) void nastar_conv3x3_flat_kernel(const FlatConvArgs a)
{
    constexpr int NB = NT / 32;                               // 32-channel output blocks per workgroup
    constexpr int NWC = 9 * 2 * 2 * NT;                       // 16-byte weight chunks per slice: [tap][kk][khalf][n]
    constexpr int NWQ = (NWC + FC_THREADS - 1) / FC_THREADS;  // ... per thread
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // LDS: zero slot (64 B) | pixel slots | weights | scale, shift
    const bool wide = a.wide != 0;
    const int halo = a.W + 1;
    const int nslot = wide ? FC_WSLOTS : FC_TP + 2 * halo;
    unsigned char* wl = smem + FC_PIXB + (size_t)nslot * FC_PIXB;
    float* ss = reinterpret_cast<float*>(wl + (size_t)NWC * 16);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware item order: dispatch id -> (xcd, j); the channel blocks of one pixel tile are consecutive j of the same xcd
    const int nblk_total = a.COUT / NT;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int tile = (j / nblk_total) * 8 + xcd;
    const int nblk = j % nblk_total;
    if (tile >= a.ntiles) return;
    const int p0 = tile * FC_TP, n0 = nblk * NT;
    const int q0 = p0 - halo;  // flat pixel held by slot 0
    // wide mode: tile -> (image, first row, first column)
    int wb = 0, wy0 = 0, wx0 = 0;
    if (wide) {
        const int tx = (a.W + FC_WTW - 1) / FC_WTW, ty = (a.H + FC_WTH - 1) / FC_WTH;
        wb = tile / (tx * ty);
        const int r = tile - wb * (tx * ty);
        wy0 = (r / tx) * FC_WTH;
        wx0 = (r - (r / tx) * tx) * FC_WTW;
    }

    const int CIN = a.C1 + a.C2;
    const int NSL = CIN / FC_KS;                 // slices per precision segment
    const int NSLICE = kSplit ? 3 * NSL : NSL;
    const int CINV = kSplit ? 3 * CIN : CIN;     // virtual input channels of the weight pack
    const int st1 = kSplit ? 2 * a.C1 : a.C1, st2 = kSplit ? 2 * a.C2 : a.C2;  // fp16 elements per pixel of in / in2
    const int HW = a.H * a.W;

    if (tid < 16) reinterpret_cast<uint32_t*>(smem)[tid] = 0u;  // the zero slot

    // ---- staging plan (once): thread t moves chunks idx = t + i*256, idx = slot*4 + c ----
    // offsets into in / in2 of (pixel, chunk) at channel 0 of the source, in 16-BYTE UNITS (pixel strides are multiples of 32 fp16, so
    // 32-bit offsets reach 32 GiB tensors); -1 = zero fill
    int src1[FC_NTQ], src2[FC_NTQ];
#pragma unroll
    for (int i = 0; i < FC_NTQ; ++i) {
        const int idx = tid + i * FC_THREADS;
        const int c = idx & 3, slot = idx >> 2;
        int q = q0 + slot;
        bool ok = slot < nslot && q >= 0 && q < a.npix;
        if (wide) {  // slot -> (row, column) of the framed 2-D tile; frame slots outside the image are zero fill
            const int sr = slot / FC_WPITCH, sc = slot - sr * FC_WPITCH;
            const int y = wy0 + sr - 1, x = wx0 + sc - 1;
            ok = slot < nslot && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
            q = (wb * a.H + y) * a.W + x;
        }
        int o1 = -1, o2 = -1;
        if (ok) {
            if (a.ups) {
                const int b = q / HW, r = q - b * HW;
                const int y = r / a.W, x = r - y * a.W;
                o1 = ((b * (a.H >> 1) + (y >> 1)) * (a.W >> 1) + (x >> 1)) * (st1 >> 3) + c;
            } else {
                o1 = q * (st1 >> 3) + c;
            }
            o2 = q * (st2 >> 3) + c;
        }
        src1[i] = o1;
        src2[i] = o2;
    }
    int wsrc[NWQ];  // element offset into wpack for slice 0
#pragma unroll
    for (int i = 0; i < NWQ; ++i) {
        const int q = tid + i * FC_THREADS;
        const int n = q % NT;
        int r = q / NT;
        const int h = r & 1; r >>= 1;
        const int kk = r & 1; r >>= 1;  // r = tap
        wsrc[i] = q < NWC ? ((r * (CINV >> 3) + kk * 2 + h) * a.COUT + n0 + n) * 8 : -1;
    }
    uint4 tq[FC_NTQ], wq[NWQ];
    auto load_slice = [&](int s) {
        // virtual slice s -> precision segment and physical channel base; segment 1 reads the lo halves
        const int seg = kSplit ? s / NSL : 0;
        const int ch = (kSplit ? s - seg * NSL : s) * FC_KS;
        const bool first = ch < a.C1;
        const uint16_t* base = first ? a.in + ch + (seg == 1 ? a.C1 : 0) : a.in2 + (ch - a.C1) + (seg == 1 ? a.C2 : 0);
#pragma unroll
        for (int i = 0; i < FC_NTQ; ++i) {
            const int o = first ? src1[i] : src2[i];
            tq[i] = make_uint4(0u, 0u, 0u, 0u);
            if (o >= 0) tq[i] = *reinterpret_cast<const uint4*>(base + (size_t)o * 8);
        }
        const uint16_t* wb = a.wpack + (size_t)s * (FC_KS / 8) * a.COUT * 8;
#pragma unroll
        for (int i = 0; i < NWQ; ++i) {
            wq[i] = make_uint4(0u, 0u, 0u, 0u);
            if (wsrc[i] >= 0) wq[i] = *reinterpret_cast<const uint4*>(wb + wsrc[i]);
        }
    };
    auto store_slice = [&]() {
#pragma unroll
        for (int i = 0; i < FC_NTQ; ++i) {
            const int idx = tid + i * FC_THREADS;
            const int c = idx & 3, slot = idx >> 2;
            if (slot < nslot) *reinterpret_cast<uint4*>(smem + fc_slot_off(slot, c)) = tq[i];
        }
#pragma unroll
        for (int i = 0; i < NWQ; ++i) {
            const int q = tid + i * FC_THREADS;
            if (q < NWC) *reinterpret_cast<uint4*>(wl + (size_t)q * 16) = wq[i];
        }
    };

    // ---- read plan (once): this lane's B-fragment address per (tap, pixel block); the k-step kk = 1 chunk is the address ^ 32 ----
    const int px = lane & 31, kh = lane >> 5;
    int baddr[9][2];
#pragma unroll
    for (int pb = 0; pb < 2; ++pb) {
        const int lp = wave * 64 + pb * 32 + px;
        const int p = p0 + lp;
        const int r = p % HW;
        const int y = r / a.W, x = r - y * a.W;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3 - 1, dx = tap % 3 - 1;
            const bool ok = p < a.npix && (unsigned)(y + dy) < (unsigned)a.H && (unsigned)(x + dx) < (unsigned)a.W;
            const int slot = lp + halo + dy * a.W + dx;
            baddr[tap][pb] = ok ? fc_slot_off(slot, kh) : (kh << 4);
            // wide: wavefront = tile row, (pb, px) = tile column; the frame is staged (zeros outside the image): no select
            if (wide) baddr[tap][pb] = fc_slot_off((wave + 1 + dy) * FC_WPITCH + pb * 32 + px + 1 + dx, kh);
        }
    }

    f32x16 acc[2][NB];
#pragma unroll
    for (int pb = 0; pb < 2; ++pb)
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[pb][n][r] = 0.f;

    load_slice(0);
    if (tid < NT) {
        ss[tid] = a.scale[n0 + tid];
        ss[NT + tid] = a.shift[n0 + tid];
    }
    store_slice();
    __syncthreads();
    for (int s = 0; s < NSLICE; ++s) {
        if (s + 1 < NSLICE) load_slice(s + 1);  // in flight during the MFMAs below
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                bf16x8 xb[2], wa[NB];
#pragma unroll
                for (int pb = 0; pb < 2; ++pb) xb[pb] = *reinterpret_cast<const bf16x8*>(smem + (baddr[tap][pb] ^ (kk << 5)));
#pragma unroll
                for (int n = 0; n < NB; ++n)
                    wa[n] = *reinterpret_cast<const bf16x8*>(wl + ((((tap * 2 + kk) * 2 + kh) * NT) + n * 32 + px) * 16);
#pragma unroll
                for (int pb = 0; pb < 2; ++pb)
#pragma unroll
                    for (int n = 0; n < NB; ++n) acc[pb][n] = mfma16<true>(wa[n], xb[pb], acc[pb][n]);
            }
        }
        if (s + 1 < NSLICE) {
            __syncthreads();  // every wave is done reading this slice
            store_slice();
            __syncthreads();
        }
    }

    // ---- epilogue: D layout col = lane&31 = pixel, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) = channel in the 32-block ----
    if constexpr (kFinal) {
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
            int p = p0 + wave * 64 + pb * 32 + px;
            bool pv = p < a.npix;
            if (wide) {
                const int y = wy0 + wave, x = wx0 + pb * 32 + px;
                pv = y < a.H && x < a.W;
                p = (wb * a.H + y) * a.W + x;
            }
            if (pv && kh == 0 && nblk == 0) {
                const float z = acc[pb][0][0] * ss[0] + ss[NT];
                a.out_f32[p] = a.raw ? z : a.final_mul / (1.0f + __expf(-z));
            }
        }
    } else {
        // A lane holds 4 consecutive channels of ITS pixel per register quad: stored directly, every store instruction would scatter 8
        // bytes into 64 different lines.  The values go through a wave-private LDS patch instead (the staging area is free now; 16-byte
        // chunk index XORed with the pixel pair: as in the 32x32 kernel) and leave as whole pixel rows: 64 / CPP pixels x NT channels
        // = 1 KB of full 128- (or 64-) byte pieces per store instruction.
        __syncthreads();  // every wave is done reading the last slice
        constexpr int CPP = NT / 8;        // 16-byte chunks per pixel and precision half
        constexpr int PPR = 64 / CPP;      // pixels per store round
        unsigned char* ob = smem + FC_PIXB + wave * (32 * NT * 2);
        const int ostride = kSplit ? 2 * a.COUT : a.COUT;
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
#pragma unroll
            for (int part = 0; part < (kSplit ? 2 : 1); ++part) {  // the hi halves, then the lo halves (v - fp16(v))
#pragma unroll
                for (int n = 0; n < NB; ++n) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int cl = n * 32 + 8 * g + 4 * kh;
                        const float4 sc = *reinterpret_cast<const float4*>(ss + cl);
                        const float4 sh = *reinterpret_cast<const float4*>(ss + NT + cl);
                        float v0 = acc[pb][n][4 * g + 0] * sc.x + sh.x;
                        float v1 = acc[pb][n][4 * g + 1] * sc.y + sh.y;
                        float v2 = acc[pb][n][4 * g + 2] * sc.z + sh.z;
                        float v3 = acc[pb][n][4 * g + 3] * sc.w + sh.w;
                        if (a.relu) {
                            v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f);
                        }
                        v0 = f16_clamp(v0); v1 = f16_clamp(v1); v2 = f16_clamp(v2); v3 = f16_clamp(v3);
                        if (part == 1) {
                            v0 = f16_residual(v0); v1 = f16_residual(v1); v2 = f16_residual(v2); v3 = f16_residual(v3);
                        }
                        uint2 o;
                        o.x = pack_f16x2(v0, v1);
                        o.y = pack_f16x2(v2, v3);
                        const int chunk = n * 4 + g;
                        *reinterpret_cast<uint2*>(ob + px * (NT * 2) + ((chunk ^ ((px >> 1) & (CPP - 1))) << 4) + kh * 8) = o;
                    }
                }
                __builtin_amdgcn_wave_barrier();  // a wavefront's LDS accesses execute in order: a compiler-level ordering point is enough
#pragma unroll
                for (int j = 0; j < 32 / PPR; ++j) {
                    const int q = j * PPR + lane / CPP, chunk = lane % CPP;
                    const uint4 v = *reinterpret_cast<const uint4*>(ob + q * (NT * 2) + ((chunk ^ ((q >> 1) & (CPP - 1))) << 4));
                    int p = p0 + wave * 64 + pb * 32 + q;
                    bool pv = p < a.npix;
                    if (wide) {
                        const int y = wy0 + wave, x = wx0 + pb * 32 + q;
                        pv = y < a.H && x < a.W;
                        p = (wb * a.H + y) * a.W + x;
                    }
                    if (pv) *reinterpret_cast<uint4*>(a.out + (size_t)p * ostride + part * a.COUT + n0 + chunk * 8) = v;
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
}

// (A second form of this kernel -- LDS filled by gfx950's global_load_lds into two buffers, 512-pixel workgroups -- was measured in round 2
// and never beat the one above; it was deleted in round 4 with the other development kernels: NOTES.md section 4.9, git history.)

// ---- 2x2 max-pool, NHWC fp16 (VGG stages of the U-Net encoder).  Split form: a value is the pair (hi, lo); hi + lo is exact in fp32
// (11 + 11 significant bits inside 24), so the pair with the larger sum is the larger value. -------------------------------------------
template <bool kSplit>
__global__ __launch_bounds__(256) void nastar_maxpool2x2_f16_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, int B,
                                                                    int H, int W, int C)
{
    const int Ho = H >> 1, Wo = W >> 1, CC = C >> 3;  // 8-channel chunks
    const long long total = (long long)B * Ho * Wo * CC;
    const int stride = kSplit ? 2 * C : C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int cc = (int)(i % CC);
        long long t = i / CC;
        const int xo = (int)(t % Wo); t /= Wo;
        const int yo = (int)(t % Ho);
        const int b = (int)(t / Ho);
        nastar_f16x8 best_hi, best_lo;
        float best[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const size_t pin = ((size_t)b * H + 2 * yo + (k >> 1)) * W + 2 * xo + (k & 1);
            const nastar_f16x8 hi = *reinterpret_cast<const nastar_f16x8*>(in + pin * stride + cc * 8);
            nastar_f16x8 lo = hi;
            if constexpr (kSplit) lo = *reinterpret_cast<const nastar_f16x8*>(in + pin * stride + C + cc * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = kSplit ? (float)hi[e] + (float)lo[e] : (float)hi[e];
                if (k == 0 || v > best[e]) {
                    best[e] = v;
                    best_hi[e] = hi[e];
                    best_lo[e] = lo[e];
                }
            }
        }
        const size_t po = ((size_t)b * Ho + yo) * Wo + xo;
        *reinterpret_cast<nastar_f16x8*>(out + po * stride + cc * 8) = best_hi;
        if constexpr (kSplit) *reinterpret_cast<nastar_f16x8*>(out + po * stride + C + cc * 8) = best_lo;
    }
}

// ---- input assembly (astar.py:171-177): x0[p] = (map, start + goal, 0 ...) as CP-channel fp16 NHWC; split form appends CP zero lo
// halves (the inputs are 0 / 1 / 2: exact in fp16) ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nastar_encoder_prep_f16_kernel(const float* __restrict__ map, const float* __restrict__ start,
                                                                      const float* __restrict__ goal, uint16_t* __restrict__ out,
                                                                      long long npix, int plus, int CP, int split)
{
    const int stride = split ? 2 * CP : CP;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long long)gridDim.x * blockDim.x) {
        uint16_t* o = out + i * stride;
        const float sg = plus ? start[i] + goal[i] : 0.f;
        const uint32_t w0 = pack_f16x2(map[i], sg);
        for (int c = 0; c < stride; c += 2) *reinterpret_cast<uint32_t*>(o + c) = (c == 0) ? w0 : 0u;
    }
}

}  // namespace nastar
