// SuperPoint convolution kernels for gfx950 (see conv.h for the design summary and reference map).
#include "conv.h"

namespace omni {

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------------------------
// Weight packing.  For every (cout tile ct, cin chunk ch, tap) the 64x64 weight block is stored in the exact order
// the MFMA A-operand fragments are read: [k-group kg][m-frag m][lane l][e]
//   fp16 (v_mfma_f32_32x32x16_f16): 4 k-groups of 16 channels, lane holds 8 halfs:
//        cout = ct*64 + m*32 + (l & 31),  cin = ch*64 + kg*16 + (l >> 5)*8 + e
//   fp32 (v_mfma_f32_32x32x2_f32):  8 k-groups of 8 channels, lane holds 4 floats, float e feeds the e-th of four
//        consecutive MFMAs (the K index inside a group is permuted identically for A and B, which a dot product
//        does not see):  cin = ch*64 + kg*8 + (l >> 5)*4 + e
// so staging a block into LDS is a straight 8/16 KB copy and every fragment read is a conflict-free, lane-linear
// ds_read_b128.
// ---------------------------------------------------------------------------------------------------------------
template <typename T, int KG, int EPL>
static void pack_weights(const float* w, int cin, int cout, int ks, T* out) {
    const int taps = ks * ks, n_ct = cout / 64, n_ch = cin / 64;
    size_t o = 0;
    for (int ct = 0; ct < n_ct; ++ct)
        for (int ch = 0; ch < n_ch; ++ch)
            for (int tap = 0; tap < taps; ++tap)
                for (int kg = 0; kg < KG; ++kg)
                    for (int m = 0; m < 2; ++m)
                        for (int l = 0; l < 64; ++l)
                            for (int e = 0; e < EPL; ++e) {
                                const int co = ct * 64 + m * 32 + (l & 31);
                                const int ci = ch * 64 + kg * (64 / KG) + (l >> 5) * EPL + e;
                                const float v = w[((size_t)co * cin + ci) * taps + tap];
                                if constexpr (sizeof(T) == 2) out[o++] = __float2half_rn(v);
                                else out[o++] = v;
                            }
}
void conv_pack_weights_f16(const float* w, int cin, int cout, int ks, __half* out) { pack_weights<__half, 4, 8>(w, cin, cout, ks, out); }
void conv_pack_weights_f32(const float* w, int cin, int cout, int ks, float* out) { pack_weights<float, 8, 4>(w, cin, cout, ks, out); }

template <typename T> struct ConvTraits;
template <> struct ConvTraits<_Float16> { static constexpr int PIX_STRIDE = 72; static constexpr int KG = 4; static constexpr int EPL = 8; };
template <> struct ConvTraits<float> { static constexpr int PIX_STRIDE = 68; static constexpr int KG = 8; static constexpr int EPL = 4; };

template <typename T, int KS>
static constexpr size_t conv_smem_bytes() {
    return ((size_t)(CONV_TH + KS - 1) * (CONV_TW + KS - 1) * ConvTraits<T>::PIX_STRIDE + 2 * 4096) * sizeof(T);
}

// ---------------------------------------------------------------------------------------------------------------
// Implicit-GEMM conv (KS = 1 or 3, stride 1, pad KS/2), fused bias + ReLU (+ 2x2 max-pool).
// grid = (tiles, cout/64, batch), 256 threads = 4 waves; wave w owns output rows 2w, 2w+1 of the 8x32 tile as two
// N-fragments of (2 rows x 16 cols) and all 64 output channels as two M-fragments: 4 accumulators (64 VGPRs).
// ---------------------------------------------------------------------------------------------------------------
template <typename T, int KS, bool POOL>
__global__ void __launch_bounds__(256)
conv_mfma_kernel(const T* __restrict__ in, void* __restrict__ out_v, const T* __restrict__ wp, const float* __restrict__ bias,
                 int H, int W, int cin, int cout, int relu, int out_f32, int in_cstride) {
    using TR = ConvTraits<T>;
    constexpr int HALO = KS / 2;
    constexpr int ITH = CONV_TH + KS - 1, ITW = CONV_TW + KS - 1;
    constexpr int PS = TR::PIX_STRIDE;
    constexpr int TAPS = KS * KS;
    constexpr int PIECE = 16 / sizeof(T);                 // elements per 16-byte piece
    constexpr int PPP = CONV_CIN_CHUNK / PIECE;           // pieces per pixel
    constexpr int WPIECES = 4096 / PIECE;                 // 16-byte pieces per weight block
    constexpr int WPT = WPIECES / 256;                    // weight pieces per thread (2 or 4)
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* in_tile = reinterpret_cast<T*>(smem_raw);
    T* wbuf = in_tile + ITH * ITW * PS;                   // [2][4096]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, hh = lane >> 5;
    const int dy = n >> 4, xx = n & 15;
    const int tiles_x = (W + CONV_TW - 1) / CONV_TW;
    const int tile_y0 = (blockIdx.x / tiles_x) * CONV_TH, tile_x0 = (blockIdx.x % tiles_x) * CONV_TW;
    const int ct = blockIdx.y, b = blockIdx.z;
    const int n_ch = cin / CONV_CIN_CHUNK;
    const T* in_b = in + (int64_t)b * H * W * in_cstride;
    const T* wp_ct = wp + (int64_t)ct * n_ch * TAPS * 4096;

    floatx16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][f][i] = 0.f;

    for (int ch = 0; ch < n_ch; ++ch) {
        // ---- stage the input halo tile for this channel chunk (previous chunk's last barrier protects the buffer).
        // All global loads are issued before the first LDS store so their latencies overlap (a load-store-per-iteration
        // loop serialises ~11 HBM round trips per tile).
        {
            constexpr int NPIECES = ITH * ITW * PPP;
            constexpr int NIT = (NPIECES + 255) / 256;
            uint4 stage[NIT];
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int idx = tid + it * 256;
                const int pix = idx / PPP, piece = idx - pix * PPP;
                const int iy = pix / ITW, ixx = pix - iy * ITW;
                const int gy = tile_y0 - HALO + iy, gx = tile_x0 - HALO + ixx;
                stage[it] = make_uint4(0u, 0u, 0u, 0u);
                if (idx < NPIECES && gy >= 0 && gy < H && gx >= 0 && gx < W)
                    stage[it] = *reinterpret_cast<const uint4*>(in_b + ((int64_t)gy * W + gx) * in_cstride + ch * CONV_CIN_CHUNK + piece * PIECE);
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int idx = tid + it * 256;
                const int pix = idx / PPP, piece = idx - pix * PPP;
                if (idx < NPIECES) *reinterpret_cast<uint4*>(in_tile + pix * PS + piece * PIECE) = stage[it];
            }
        }
        const T* wp_ch = wp_ct + (int64_t)ch * TAPS * 4096;
#pragma unroll
        for (int i = 0; i < WPT; ++i)
            *reinterpret_cast<uint4*>(wbuf + (tid + i * 256) * PIECE) = *reinterpret_cast<const uint4*>(wp_ch + (tid + i * 256) * PIECE);
        __syncthreads();

        for (int tap = 0; tap < TAPS; ++tap) {
            uint4 wnext[WPT];
            if (tap + 1 < TAPS) {
#pragma unroll
                for (int i = 0; i < WPT; ++i)
                    wnext[i] = *reinterpret_cast<const uint4*>(wp_ch + (int64_t)(tap + 1) * 4096 + (tid + i * 256) * PIECE);
            }
            const T* wcur = wbuf + (tap & 1) * 4096;
            const int ky = tap / KS, kx = tap - ky * KS;
            // LDS pixel index of this lane's pixel for fragment f at this tap (tile origin = -HALO)
            const int prow = 2 * wave + dy + ky;
            const T* bp0 = in_tile + (prow * ITW + xx + kx) * PS + hh * TR::EPL;
            const T* bp1 = bp0 + 16 * PS;
#pragma unroll
            for (int kg = 0; kg < TR::KG; ++kg) {
                if constexpr (sizeof(T) == 2) {
                    const half8_t a0 = *reinterpret_cast<const half8_t*>(wcur + ((kg * 2 + 0) * 64 + lane) * 8);
                    const half8_t a1 = *reinterpret_cast<const half8_t*>(wcur + ((kg * 2 + 1) * 64 + lane) * 8);
                    const half8_t b0 = *reinterpret_cast<const half8_t*>(bp0 + kg * 16);
                    const half8_t b1 = *reinterpret_cast<const half8_t*>(bp1 + kg * 16);
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[0][0], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc[1][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc[0][1], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc[1][1], 0, 0, 0);
                } else {
                    const floatx4 a0 = *reinterpret_cast<const floatx4*>(wcur + ((kg * 2 + 0) * 64 + lane) * 4);
                    const floatx4 a1 = *reinterpret_cast<const floatx4*>(wcur + ((kg * 2 + 1) * 64 + lane) * 4);
                    const floatx4 b0 = *reinterpret_cast<const floatx4*>(bp0 + kg * 8);
                    const floatx4 b1 = *reinterpret_cast<const floatx4*>(bp1 + kg * 8);
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], b0[s], acc[0][0], 0, 0, 0);
                        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], b0[s], acc[1][0], 0, 0, 0);
                        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], b1[s], acc[0][1], 0, 0, 0);
                        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], b1[s], acc[1][1], 0, 0, 0);
                    }
                }
            }
            if (tap + 1 < TAPS) {
                T* wnxt = wbuf + ((tap + 1) & 1) * 4096;
#pragma unroll
                for (int i = 0; i < WPT; ++i) *reinterpret_cast<uint4*>(wnxt + (tid + i * 256) * PIECE) = wnext[i];
            }
            __syncthreads();
        }
    }

    // ---- epilogue: (2x2 max-pool) + bias + ReLU, NHWC stores of 4 consecutive channels per register quad
    const int Ho = POOL ? (H >> 1) : H, Wo = POOL ? (W >> 1) : W;
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        const int oy = tile_y0 + 2 * wave + dy, ox = tile_x0 + 16 * f + xx;
        bool writer = (oy < H) && (ox < W);
        int py = oy, px = ox;
        if constexpr (POOL) { writer = writer && ((n & 1) == 0) && ((n & 16) == 0); py = oy >> 1; px = ox >> 1; }
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            floatx16 v = acc[m][f];
            if constexpr (POOL) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    float t = v[i];
                    t = fmaxf(t, __shfl_xor(t, 1, 64));
                    t = fmaxf(t, __shfl_xor(t, 16, 64));
                    v[i] = t;
                }
            }
            if (writer) {
                const int64_t obase = (((int64_t)b * Ho + py) * Wo + px) * cout + ct * 64 + m * 32 + 4 * hh;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c = ct * 64 + m * 32 + 8 * g + 4 * hh;
                    const float4 bs = *reinterpret_cast<const float4*>(bias + c);
                    float r0 = v[4 * g + 0] + bs.x, r1 = v[4 * g + 1] + bs.y, r2 = v[4 * g + 2] + bs.z, r3 = v[4 * g + 3] + bs.w;
                    if (relu) { r0 = fmaxf(r0, 0.f); r1 = fmaxf(r1, 0.f); r2 = fmaxf(r2, 0.f); r3 = fmaxf(r3, 0.f); }
                    if (sizeof(T) == 4 || out_f32) {
                        *reinterpret_cast<float4*>(reinterpret_cast<float*>(out_v) + obase + 8 * g) = make_float4(r0, r1, r2, r3);
                    } else {
                        half4_t h4;
                        h4[0] = (_Float16)r0; h4[1] = (_Float16)r1; h4[2] = (_Float16)r2; h4[3] = (_Float16)r3;
                        *reinterpret_cast<half4_t*>(reinterpret_cast<_Float16*>(out_v) + obase + 8 * g) = h4;
                    }
                }
            }
        }
    }
}

template <typename T, int KS, bool POOL>
static int launch_conv(hipStream_t st, const ConvArgs& a) {
    const size_t smem = conv_smem_bytes<T, KS>();
    auto kfn = conv_mfma_kernel<T, KS, POOL>;
    OMNI_HIP_TRY(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    dim3 grid(cdiv(a.W, CONV_TW) * cdiv(a.H, CONV_TH), a.cout / CONV_COUT_TILE, a.batch);
    hipLaunchKernelGGL(kfn, grid, dim3(256), smem, st, reinterpret_cast<const T*>(a.in), a.out, reinterpret_cast<const T*>(a.w_packed),
                       a.bias, a.H, a.W, a.cin, a.cout, a.relu ? 1 : 0, a.out_f32 ? 1 : 0, a.in_cstride > 0 ? a.in_cstride : a.cin);
    OMNI_LAUNCH_CHECK();
    return OMNI_OK;
}

int conv_mfma(hipStream_t st, int precision, const ConvArgs& a) {
    OMNI_REQUIRE(a.cin % 64 == 0 && a.cout % 64 == 0, OMNI_ERR_INVALID, "conv_mfma: cin=%d cout=%d must be multiples of 64", a.cin, a.cout);
    OMNI_REQUIRE(a.ksize == 1 || a.ksize == 3, OMNI_ERR_INVALID, "conv_mfma: ksize=%d", a.ksize);
    OMNI_REQUIRE(!a.pool || (a.H % 2 == 0 && a.W % 2 == 0), OMNI_ERR_INVALID, "pooling needs even H, W");
    OMNI_REQUIRE(!(a.pool && a.ksize == 1), OMNI_ERR_INVALID, "1x1 + pool not instantiated");
    if (precision == OMNI_PREC_F16) {
        if (a.ksize == 3) return a.pool ? launch_conv<_Float16, 3, true>(st, a) : launch_conv<_Float16, 3, false>(st, a);
        return launch_conv<_Float16, 1, false>(st, a);
    } else {
        if (a.ksize == 3) return a.pool ? launch_conv<float, 3, true>(st, a) : launch_conv<float, 3, false>(st, a);
        return launch_conv<float, 1, false>(st, a);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// conv1a (Cin = 1): direct fp32 VALU conv from the u8 image; lane = (pixel, group of 8 output channels) so a wave
// writes 8 pixels x 64 channels = 1 KiB (fp16) of contiguous NHWC.  0.7 % of the network's FLOPs.
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256)
conv1a_kernel(const uint8_t* __restrict__ gray, int stride, int H, int W, int mask_row0, int mask_row1,
              const float* __restrict__ w, const float* __restrict__ bias, const float* __restrict__ lut, T* __restrict__ out) {
    __shared__ float tile[10][36];
    __shared__ float wsm[9][64];
    __shared__ float bsm[64];
    __shared__ float lsm[256];
    const int tid = threadIdx.x;
    const int tiles_x = (W + 31) / 32;
    const int ty0 = (blockIdx.x / tiles_x) * 8, tx0 = (blockIdx.x % tiles_x) * 32;
    const int b = blockIdx.y;
    const uint8_t* g = gray + (int64_t)b * stride * H;
    lsm[tid] = lut[tid];
    for (int i = tid; i < 576; i += 256) { const int co = i / 9, tap = i - co * 9; wsm[tap][co] = w[i]; }
    if (tid < 64) bsm[tid] = bias[tid];
    __syncthreads();
    for (int i = tid; i < 340; i += 256) {
        const int iy = i / 34, ix = i - iy * 34;
        const int gy = ty0 - 1 + iy, gx = tx0 - 1 + ix;
        float v = 0.f;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W && !(gy >= mask_row0 && gy < mask_row1)) v = lsm[g[(int64_t)gy * stride + gx]];
        tile[iy][ix] = v;
    }
    __syncthreads();
#pragma unroll 1
    for (int it = 0; it < 8; ++it) {
        const int wi = tid + it * 256;
        const int pix = wi >> 3, cg = wi & 7;
        const int py = pix >> 5, px = pix & 31;
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = bsm[cg * 8 + j];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const float v = tile[py + tap / 3][px + tap % 3];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = fmaf(v, wsm[tap][cg * 8 + j], acc[j]);
        }
        const int gy = ty0 + py, gx = tx0 + px;
        if (gy < H && gx < W) {
            T* o = out + (((int64_t)b * H + gy) * W + gx) * 64 + cg * 8;
            if constexpr (sizeof(T) == 4) {
                *reinterpret_cast<float4*>(o) = make_float4(fmaxf(acc[0], 0.f), fmaxf(acc[1], 0.f), fmaxf(acc[2], 0.f), fmaxf(acc[3], 0.f));
                *reinterpret_cast<float4*>(o + 4) = make_float4(fmaxf(acc[4], 0.f), fmaxf(acc[5], 0.f), fmaxf(acc[6], 0.f), fmaxf(acc[7], 0.f));
            } else {
                half8_t h;
#pragma unroll
                for (int j = 0; j < 8; ++j) h[j] = (_Float16)fmaxf(acc[j], 0.f);
                *reinterpret_cast<half8_t*>(o) = h;
            }
        }
    }
}

int conv1a_direct(hipStream_t st, int precision, const uint8_t* gray, int stride, int batch, int H, int W, int fisheye_mask,
                  const float* w, const float* bias, const float* lut, void* out) {
    const int r0 = fisheye_mask ? H * 3 / 4 : H, r1 = fisheye_mask ? H * 3 / 4 + H / 4 : H;   // cv::Rect(0, rows*3/4, cols, rows/4)
    dim3 grid(cdiv(W, 32) * cdiv(H, 8), batch);
    if (precision == OMNI_PREC_F16)
        hipLaunchKernelGGL(conv1a_kernel<_Float16>, grid, dim3(256), 0, st, gray, stride, H, W, r0, r1, w, bias, lut, (_Float16*)out);
    else
        hipLaunchKernelGGL(conv1a_kernel<float>, grid, dim3(256), 0, st, gray, stride, H, W, r0, r1, w, bias, lut, (float*)out);
    OMNI_LAUNCH_CHECK();
    return OMNI_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Detector head tail: 1x1 conv 256 -> 65, softmax(65), drop the dustbin, depth-to-space 8x8.  fp32 throughout.
// One wave per coarse cell: lane c < 64 owns logit c, the dustbin logit is a wave reduction.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

#define DET_THREADS 512
#define DET_WAVES (DET_THREADS / 64)
template <typename T>
__global__ void __launch_bounds__(DET_THREADS)
detector_head_kernel(const T* __restrict__ in, int in_stride, int in_off, int n_cells, int Hc, int Wc,
                     const float* __restrict__ wT, const float* __restrict__ bias, float* __restrict__ semi) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* wTs = reinterpret_cast<float*>(smem_raw);      // [256][65]
    float* xs = wTs + 256 * 65;                            // [DET_WAVES][256]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 256 * 65; i += DET_THREADS) wTs[i] = wT[i];
    const float my_bias = bias[lane];
    const float dust_bias = bias[64];
    __syncthreads();
    float* x = xs + wave * 256;
    for (int base = blockIdx.x * DET_WAVES; base < n_cells; base += gridDim.x * DET_WAVES) {
        const int cell = base + wave;
        const bool valid = cell < n_cells;
        if (valid) {
            const T* ip = in + (int64_t)cell * in_stride + in_off + lane * 4;
            if constexpr (sizeof(T) == 4) {
                *reinterpret_cast<float4*>(x + lane * 4) = *reinterpret_cast<const float4*>(ip);
            } else {
                const half4_t h = *reinterpret_cast<const half4_t*>(ip);
                *reinterpret_cast<float4*>(x + lane * 4) = make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
            }
        }
        __syncthreads();
        if (valid) {
            float acc = my_bias;
            for (int k = 0; k < 256; k += 4) {
                const float4 xv = *reinterpret_cast<const float4*>(x + k);
                acc = fmaf(xv.x, wTs[(k + 0) * 65 + lane], acc);
                acc = fmaf(xv.y, wTs[(k + 1) * 65 + lane], acc);
                acc = fmaf(xv.z, wTs[(k + 2) * 65 + lane], acc);
                acc = fmaf(xv.w, wTs[(k + 3) * 65 + lane], acc);
            }
            float dpart = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) dpart = fmaf(x[lane * 4 + j], wTs[(lane * 4 + j) * 65 + 64], dpart);
            const float dust = wave_sum_f(dpart) + dust_bias;
            const float mx = fmaxf(wave_max_f(acc), dust);
            const float e = expf(acc - mx);
            const float ed = expf(dust - mx);
            const float s = wave_sum_f(e) + ed;
            const float p = e / s;
            const int wx = cell % Wc;
            const int hy = (cell / Wc) % Hc;
            const int b = cell / (Wc * Hc);
            semi[((int64_t)b * Hc * 8 + hy * 8 + (lane >> 3)) * (Wc * 8) + wx * 8 + (lane & 7)] = p;
        }
        __syncthreads();
    }
}

int detector_head(hipStream_t st, int precision, const void* in, int in_stride, int in_off, int batch, int Hc, int Wc,
                  const float* wT, const float* bias, float* semi) {
    const int n_cells = batch * Hc * Wc;
    const size_t smem = (size_t)(256 * 65 + DET_WAVES * 256) * 4;
    int grid = cdiv(n_cells, DET_WAVES);
    if (grid > 512) grid = 512;                            // 2 workgroups/CU (75 KB LDS each): weights staged once per workgroup
    if (precision == OMNI_PREC_F16) {
        OMNI_HIP_TRY(hipFuncSetAttribute((const void*)detector_head_kernel<_Float16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        hipLaunchKernelGGL(detector_head_kernel<_Float16>, dim3(grid), dim3(DET_THREADS), smem, st, (const _Float16*)in, in_stride, in_off,
                           n_cells, Hc, Wc, wT, bias, semi);
    } else {
        OMNI_HIP_TRY(hipFuncSetAttribute((const void*)detector_head_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        hipLaunchKernelGGL(detector_head_kernel<float>, dim3(grid), dim3(DET_THREADS), smem, st, (const float*)in, in_stride, in_off, n_cells,
                           Hc, Wc, wT, bias, semi);
    }
    OMNI_LAUNCH_CHECK();
    return OMNI_OK;
}

// desc / ||desc||_2 per coarse cell: one wave per cell, lane holds 4 channels
__global__ void __launch_bounds__(256)
l2norm_kernel(float* __restrict__ d, int64_t n_cells) {
    const int lane = threadIdx.x & 63;
    const int64_t cell = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (cell >= n_cells) return;
    float4* p = reinterpret_cast<float4*>(d + cell * 256 + lane * 4);
    float4 v = *p;
    float ss = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    ss = wave_sum_f(ss);
    const float nrm = sqrtf(ss);
    v.x /= nrm; v.y /= nrm; v.z /= nrm; v.w /= nrm;
    *p = v;
}

int l2norm_channels(hipStream_t st, float* desc, int64_t n_cells) {
    hipLaunchKernelGGL(l2norm_kernel, dim3((unsigned)cdiv64(n_cells, 4)), dim3(256), 0, st, desc, n_cells);
    OMNI_LAUNCH_CHECK();
    return OMNI_OK;
}

template <typename T>
__global__ void nhwc_to_nchw_f32_kernel(const T* __restrict__ in, float* __restrict__ out, int C, int HW, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // output index: ((b*C + c)*HW + p)
    if (i >= total) return;
    const int p = (int)(i % HW);
    const int c = (int)((i / HW) % C);
    const int64_t b = i / ((int64_t)HW * C);
    out[i] = (float)in[(b * HW + p) * C + c];
}

int nhwc_any_to_nchw_f32(hipStream_t st, int precision_of_in, const void* in, float* out, int batch, int C, int HW) {
    const int64_t total = (int64_t)batch * C * HW;
    const unsigned grid = (unsigned)cdiv64(total, 256);
    if (precision_of_in == OMNI_PREC_F16)
        hipLaunchKernelGGL(nhwc_to_nchw_f32_kernel<_Float16>, dim3(grid), dim3(256), 0, st, (const _Float16*)in, out, C, HW, total);
    else
        hipLaunchKernelGGL(nhwc_to_nchw_f32_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)in, out, C, HW, total);
    OMNI_LAUNCH_CHECK();
    return OMNI_OK;
}

}  // namespace omni
