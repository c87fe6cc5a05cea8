// SuperPoint convolution kernels for gfx950 (see conv.h for the design summary and reference map).
#include "config.h"
#include "conv.h"
#include <type_traits>

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
// Shared epilogue.  Fragment geometry (both conv kernels): wave w owns output rows 2w and 2w+1 of the 8x32 tile; N-fragment
// f is ROW 2w+f, its 32 lanes-columns are the 32 consecutive pixels of that row (a single-row fragment keeps every
// ds_read_b128 of the B operand bank-conflict free: 16 consecutive pixels -> 16 distinct 16-byte slots).  Lane holds pixel
// n = lane & 31 and output channels (reg&3) + 8*(reg>>2) + 4*(lane>>5) of each 32-channel M-fragment.
// 2x2 max-pool = elementwise max of the two row accumulators (in registers) + one cross-lane exchange with lane^1.
// ---------------------------------------------------------------------------------------------------------------
template <typename OutT, bool POOL, typename BiasFn>
__device__ __forceinline__ void conv_epilogue(floatx16 (&acc)[2][2], OutT* __restrict__ out, int b, int H, int W, int cout, int ct,
                                              int tile_y0, int tile_x0, int wave, int lane, int relu, BiasFn bias_of) {
    const int n = lane & 31, hh = lane >> 5;
    const int Ho = POOL ? (H >> 1) : H, Wo = POOL ? (W >> 1) : W;
    const int ox = tile_x0 + n;
    auto store4 = [&](OutT* o, float r0, float r1, float r2, float r3) {
        if (relu) { r0 = fmaxf(r0, 0.f); r1 = fmaxf(r1, 0.f); r2 = fmaxf(r2, 0.f); r3 = fmaxf(r3, 0.f); }
        if constexpr (sizeof(OutT) == 4) {
            *reinterpret_cast<float4*>(o) = make_float4(r0, r1, r2, r3);
        } else {
            half4_t h4;
            h4[0] = (_Float16)r0; h4[1] = (_Float16)r1; h4[2] = (_Float16)r2; h4[3] = (_Float16)r3;
            *reinterpret_cast<half4_t*>(o) = h4;
        }
    };
    if constexpr (POOL) {
        const int oy = tile_y0 + 2 * wave;
        const bool writer = (oy < H) && (ox < W) && ((n & 1) == 0);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            floatx16 v;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float q = fmaxf(acc[m][0][i], acc[m][1][i]);
                q = fmaxf(q, __shfl_xor(q, 1, 64));
                v[i] = q;
            }
            if (writer) {
                OutT* o = out + (((int64_t)b * Ho + (oy >> 1)) * Wo + (ox >> 1)) * cout + ct * 64 + m * 32 + 4 * hh;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 bs = bias_of(m, g);
                    store4(o + 8 * g, v[4 * g + 0] + bs.x, v[4 * g + 1] + bs.y, v[4 * g + 2] + bs.z, v[4 * g + 3] + bs.w);
                }
            }
        }
    } else {
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            const int oy = tile_y0 + 2 * wave + f;
            if ((oy < H) && (ox < W)) {
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    OutT* o = out + (((int64_t)b * Ho + oy) * Wo + ox) * cout + ct * 64 + m * 32 + 4 * hh;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float4 bs = bias_of(m, g);
                        store4(o + 8 * g, acc[m][f][4 * g + 0] + bs.x, acc[m][f][4 * g + 1] + bs.y, acc[m][f][4 * g + 2] + bs.z,
                               acc[m][f][4 * g + 3] + bs.w);
                    }
                }
            }
        }
    }
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
            const T* bp0 = in_tile + ((2 * wave + ky) * ITW + n + kx) * PS + hh * TR::EPL;   // fragment 0 = row 2w
            const T* bp1 = bp0 + ITW * PS;                                                  // fragment 1 = row 2w+1
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
    auto bias_of = [&](int m, int g) { return *reinterpret_cast<const float4*>(bias + ct * 64 + m * 32 + 8 * g + 4 * hh); };
    if (sizeof(T) == 4 || out_f32)
        conv_epilogue<float, POOL>(acc, reinterpret_cast<float*>(out_v), b, H, W, cout, ct, tile_y0, tile_x0, wave, lane, relu, bias_of);
    else
        conv_epilogue<_Float16, POOL>(acc, reinterpret_cast<_Float16*>(out_v), b, H, W, cout, ct, tile_y0, tile_x0, wave, lane, relu, bias_of);
}

// ---------------------------------------------------------------------------------------------------------------
// v2 kernel for the fp16 3x3 layers with 64 INPUT channels (conv1b, conv2a, conv2b, conv3a: 65 % of the FLOPs).
//   * persistent workgroups (one per CU, 4 waves = one per SIMD): all 9 taps of the 64x64 weight block (73.7 KB) are
//     staged into LDS ONCE per workgroup -- no per-tap staging, no per-tap barriers;
//   * the 10x34-pixel input halo tile (43.5 KB) is double buffered and filled by LDS-DMA (global_load_lds_dwordx4,
//     1 KiB per wave instruction, no VGPR round trip) for tile t+1 while the MFMAs of tile t run: ONE barrier per tile;
//   * LDS image is lane-linear as DMA requires; bank conflicts are removed by an XOR swizzle applied to the per-lane
//     SOURCE address and to the fragment reads (16-byte chunk c of pixel p lives at chunk c ^ ((p >> 1) & 7)), so 16
//     consecutive pixels hit 16 distinct 16-byte slots of the 256-byte bank row;
//   * out-of-image halo pixels are DMA'd from a clamped (valid) address and then overwritten with zeros.
// LDS: 73 728 + 2 x 44 032 = 161 792 B of the CU's 163 840.
// ---------------------------------------------------------------------------------------------------------------
#define C64_ITW 34
#define C64_ITH 10
#define C64_PIX (C64_ITW * C64_ITH)            // 340 pixels
#define C64_CHUNKS (C64_PIX * 8)               // 2720 16-byte chunks
#define C64_BUF_BYTES (43 * 1024)              // 43 wave-instructions of 1 KiB
#define C64_W_BYTES (9 * 4096 * 2)
#define C64_SMEM (C64_W_BYTES + 2 * C64_BUF_BYTES)
#define PP_SMEM (C64_SMEM + 256)                // + the cout tile's bias
// FUSE1A: halo buffers without the DMA tail padding, + conv1a's bias, the u8 -> (hi, lo) half table and four 240-byte image patches
#define PP_SMEM_FUSED (C64_W_BYTES + 2 * C64_PIX * 128 + 256 + 256 + 1024 + 960)

#ifdef OMNI_TEST_VARIANTS          // the v2 kernel: a bit-identity reference of the test build (lib_test/), not part of the shipped library
template <bool POOL>
__global__ void __launch_bounds__(256, 1)
conv3x3_c64_f16_kernel(const _Float16* __restrict__ in, _Float16* __restrict__ out, const _Float16* __restrict__ wp,
                       const float* __restrict__ bias, int H, int W, int cout, int n_ct, int tiles_x, int tiles_y, int batch, int relu) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    _Float16* wts = reinterpret_cast<_Float16*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, hh = lane >> 5;
    const int ct = blockIdx.x % n_ct, wg = blockIdx.x / n_ct, nwg = gridDim.x / n_ct;
    const int tiles_per_img = tiles_x * tiles_y;
    const int total = batch * tiles_per_img;

    {   // weights for this cout tile: 9 taps x 8 KB, fragment order (see pack_weights)
        const uint4* src = reinterpret_cast<const uint4*>(wp + (int64_t)ct * 9 * 4096);
        uint4* dst = reinterpret_cast<uint4*>(wts);
        for (int i = tid; i < C64_W_BYTES / 16; i += 256) dst[i] = src[i];
    }
    float4 bias_r[2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int g = 0; g < 4; ++g) bias_r[m][g] = *reinterpret_cast<const float4*>(bias + ct * 64 + m * 32 + 8 * g + 4 * hh);

    auto buf_ptr = [&](int which) -> char* { return smem_raw + C64_W_BYTES + which * C64_BUF_BYTES; };

    // issue the LDS-DMA of tile t into buffer `which` (async; completion = vmcnt)
    auto issue = [&](int t, int which) {
        const int b = t / tiles_per_img, r = t - b * tiles_per_img;
        const int y0 = (r / tiles_x) * CONV_TH - 1, x0 = (r % tiles_x) * CONV_TW - 1;
        const _Float16* img = in + (int64_t)b * H * W * 64;
        char* base = buf_ptr(which);
#pragma unroll
        for (int j = 0; j < 11; ++j) {
            const int wi = wave * 11 + j;                       // wave-uniform instruction index, 1 KiB each
            if (wi < 43) {
                int idx = wi * 64 + lane;
                idx = idx < C64_CHUNKS ? idx : C64_CHUNKS - 1;   // tail lanes of instruction 42: harmless duplicate
                const int pix = idx >> 3, phys = idx & 7;
                const int iy = pix / C64_ITW, ix = pix - iy * C64_ITW;
                int gy = y0 + iy, gx = x0 + ix;
                gy = gy < 0 ? 0 : (gy >= H ? H - 1 : gy);        // clamped: zero-fixed below
                gx = gx < 0 ? 0 : (gx >= W ? W - 1 : gx);
                const int logical = phys ^ ((pix >> 1) & 7);
                const _Float16* g = img + ((int64_t)gy * W + gx) * 64 + logical * 8;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                 (__attribute__((address_space(3))) void*)(base + wi * 1024), 16, 0, 0);
            }
        }
    };
    auto zero_fix = [&](int t, int which) {
        const int b = t / tiles_per_img, r = t - b * tiles_per_img;
        const int y0 = (r / tiles_x) * CONV_TH - 1, x0 = (r % tiles_x) * CONV_TW - 1;
        if (y0 >= 0 && y0 + C64_ITH <= H && x0 >= 0 && x0 + C64_ITW <= W) return;     // interior tile (workgroup-uniform)
        char* base = buf_ptr(which);
        for (int pix = tid; pix < C64_PIX; pix += 256) {
            const int iy = pix / C64_ITW, ix = pix - iy * C64_ITW;
            const int gy = y0 + iy, gx = x0 + ix;
            if (gy < 0 || gy >= H || gx < 0 || gx >= W) {
                uint4* p = reinterpret_cast<uint4*>(base + pix * 128);
#pragma unroll
                for (int c = 0; c < 8; ++c) p[c] = make_uint4(0u, 0u, 0u, 0u);
            }
        }
    };

    int t = wg;
    if (t < total) issue(t, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t < total) zero_fix(t, 0);
    __syncthreads();

    int cur = 0;
    for (; t < total; t += nwg, cur ^= 1) {
        const int tn = t + nwg;
        if (tn < total) issue(tn, cur ^ 1);

        floatx16 acc[2][2];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[m][f][i] = 0.f;
        const _Float16* tile = reinterpret_cast<const _Float16*>(buf_ptr(cur));
        // 36 k-steps (9 taps x 4 groups of 16 channels), software pipelined by hand: the four fragment reads of step s+1
        // are issued before the four MFMAs of step s (one wave per SIMD has no other wave to hide LDS latency behind).
        half8_t fa0[2], fa1[2], fb0[2], fb1[2];
        auto load_step = [&](int step, int slot) {
            const int tap = step >> 2, kg = step & 3;
            const int ky = tap / 3, kx = tap - ky * 3;
            const int p0 = (2 * wave + ky) * C64_ITW + n + kx;     // fragment 0 = row 2w, 32 consecutive pixels
            const int p1 = p0 + C64_ITW;                            // fragment 1 = row 2w+1
            const int c = kg * 2 + hh;
            // issue order = order of first use by the next step's MFMAs: (a0,b0) (a1,b0) (a0,b1) (a1,b1)
            fb0[slot] = *reinterpret_cast<const half8_t*>(tile + p0 * 64 + ((c ^ ((p0 >> 1) & 7)) << 3));
            fa0[slot] = *reinterpret_cast<const half8_t*>(wts + (step * 2 + 0) * 512 + lane * 8);
            fa1[slot] = *reinterpret_cast<const half8_t*>(wts + (step * 2 + 1) * 512 + lane * 8);
            fb1[slot] = *reinterpret_cast<const half8_t*>(tile + p1 * 64 + ((c ^ ((p1 >> 1) & 7)) << 3));
        };
        load_step(0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int step = 0; step < 36; ++step) {
            const int sl = step & 1;
            if (step + 1 < 36) load_step(step + 1, sl ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa0[sl], fb0[sl], acc[0][0], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa1[sl], fb0[sl], acc[1][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa0[sl], fb1[sl], acc[0][1], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa1[sl], fb1[sl], acc[1][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);   // keep "reads of step s+1, then MFMAs of step s" exactly as written
        }

        {   // epilogue
            const int b = t / tiles_per_img, r = t - b * tiles_per_img;
            auto bias_of = [&](int m, int g) { return bias_r[m][g]; };
            conv_epilogue<_Float16, POOL>(acc, out, b, H, W, cout, ct, (r / tiles_x) * CONV_TH, (r % tiles_x) * CONV_TW, wave, lane, relu, bias_of);
        }
        // next tile's DMA must have landed before anyone reads it; everyone must be done with `cur` before it is refilled
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tn < total) zero_fix(tn, cur ^ 1);
        __syncthreads();
    }
}

template <bool POOL>
static int launch_conv_c64(hipStream_t st, const ConvArgs& a, int n_cu) {
    auto kfn = conv3x3_c64_f16_kernel<POOL>;
    OMNI_HIP_TRY(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C64_SMEM));
    const int tiles_x = cdiv(a.W, CONV_TW), tiles_y = cdiv(a.H, CONV_TH), n_ct = a.cout / 64;
    const int total = a.batch * tiles_x * tiles_y;
    int per_ct = n_cu / n_ct;
    if (per_ct < 1) per_ct = 1;
    if (per_ct > total) per_ct = total;
    hipLaunchKernelGGL(kfn, dim3(per_ct * n_ct), dim3(256), C64_SMEM, st, reinterpret_cast<const _Float16*>(a.in),
                       reinterpret_cast<_Float16*>(a.out), reinterpret_cast<const _Float16*>(a.w_packed), a.bias, a.H, a.W, a.cout, n_ct,
                       tiles_x, tiles_y, a.batch, a.relu ? 1 : 0);
    OMNI_LAUNCH_CHECK();
    return OMNI_OK;
}
#endif  // OMNI_TEST_VARIANTS

// ---------------------------------------------------------------------------------------------------------------
// v3 "ping-pong" kernel for the same layers (fp16, 3x3, 64 input channels).  Why: in v2 one wave per SIMD does everything
// in sequence -- DMA issue, 144 MFMAs, max-pool/bias/ReLU/stores, barriers -- so the matrix pipe idles ~60 % of a tile.
//   * 8 waves = two GROUPS of four (waves 0-3 / 4-7; wave i and i+4 share a SIMD).  Phases are separated by ONE workgroup
//     barrier; in every phase one group runs the 144-MFMA loop of its tile while the other group SERVICES: issues the
//     LDS-DMA of its next tile into its own buffer (free since its MFMA phase ended), runs the epilogue of the tile it
//     just computed (accumulators stay in registers across the barrier), waits for its DMA and zero-fixes the halo.  Next
//     phase the roles swap, so each SIMD's matrix pipe always has a wave feeding it.
//   * LDS: 73 728 B resident weights (shared by both groups) + one 44 032 B halo buffer per group = 161 792 B.
//   * fragment reads are inline-asm ds_read_b128 prefetched TWO k-steps ahead behind counted s_waitcnt lgkmcnt(8/4/0):
//     hipcc's own schedule of this loop waited lgkmcnt(0) every other step, exposing a full LDS round trip each time.
//   * B-fragment addresses: 12 per-lane bases (4 halo rows x 3 kx) computed once per kernel; the 16-channel group kg is
//     one XOR with kg << 5 (the swizzle touches bits 4-6 only), A-fragment addresses are immediates on two bases.
// K order (tap-major, 16-channel groups inner) is that of the other two kernels: results are bit-identical.
// ---------------------------------------------------------------------------------------------------------------
#define PP_THREADS 512

typedef float float2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

// fmaxf() on MFMA results costs three instructions under plain -O3 (hipcc canonicalises both inputs with a v_max x, x first);
// the epilogues below are instruction-count bound next to the partner wave's MFMA stream, so they use the bare instruction
// (only on accumulators that were written a whole phase earlier: hipcc pads no MFMA hazard in front of inline asm)
__device__ __forceinline__ float vmax(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// max(a, b) = median(a, b, +inf): one v_med3_f32, no canonicalisation, and visible to hipcc's hazard recogniser (usable right
// behind the MFMAs that produced a and b)
__device__ __forceinline__ float vmax_med3(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, __builtin_inff()); }
// ReLU as an integer max on the bit pattern (negative floats and -0 are negative integers): one compiler-visible instruction,
// so the MFMA -> VALU wait states in front of it are still inserted by hipcc (they are NOT for an inline-asm reader)
__device__ __forceinline__ float vrelu(float a) { const int b = __builtin_bit_cast(int, a); return __builtin_bit_cast(float, b > 0 ? b : 0); }
// max(v, value of lane ^ 1) in ONE instruction (v_max_f32 with a DPP source).  Inline asm: the 2 wait states a DPP read needs after
// the VALU write of its source are NOT inserted by hipcc here -- callers must produce v at least two instructions earlier
// (the pooling code computes all 32 vertical maxima first, then the 32 horizontal ones).
__device__ __forceinline__ float vmax_swap_pairs(float v) {
    float r;
    asm("v_max_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v));
    return r;
}
__device__ __forceinline__ float dpp_swap_pairs(float v) {      // value of lane ^ 1 (quad_perm [1,0,3,2]): a VALU modifier, no LDS
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
}
// two floats -> packed halfs (RNE) -> ReLU.  relu(cvt(x)) == cvt(relu(x)) exactly (rounding is monotone and keeps the sign), and
// on the packed bit patterns ReLU is one v_pk_max_i16 (negative halfs and -0 are negative 16-bit integers)
typedef short short2_t __attribute__((ext_vector_type(2)));
// PK = false: ReLU on the floats first (two v_max_i32).  Measured A/B on MI355X (tools/ab.sh): the packed form wins in the pooled
// conv1b kernel (0.146 -> 0.139 ms), the unpacked one in the non-pooled layers (conv2a 0.054 -> 0.050 ms) -- scheduling next to
// the partner wave's MFMA stream, not instruction count, decides.
template <bool PK = true>
__device__ __forceinline__ uint32_t pack_relu_f16(float a, float b, int relu) {
    if constexpr (!PK) {
        if (relu) { a = vrelu(a); b = vrelu(b); }
        float2_t f0; f0[0] = a; f0[1] = b;
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(f0, half2_t));
    }
    float2_t f; f[0] = a; f[1] = b;
    const half2_t h = __builtin_convertvector(f, half2_t);
    if (!relu) return __builtin_bit_cast(uint32_t, h);
    const short2_t z = {0, 0};
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(short2_t, h), z));
}
// 16 accumulator values of one 32-channel M-fragment (+ bias) -> two 16-byte NHWC stores per lane: the lower half-wave
// (hh = 0) ends up with channels [16 gp, 16 gp + 8) and the upper one with [16 gp + 8, 16 gp + 16) of pair gp after one
// v_permlane32_swap per dword (the half-waves hold interleaved 4-channel runs of the same pixel).
template <bool PK>
__device__ __forceinline__ void store_frag16(const float (&v)[16], const float4 (&bs)[4], _Float16* __restrict__ pix_base /* + m*32 */, int hh,
                                             int relu, bool pred) {
    uint32_t d[4][2];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        d[g][0] = pack_relu_f16<PK>(v[4 * g + 0] + bs[g].x, v[4 * g + 1] + bs[g].y, relu);
        d[g][1] = pack_relu_f16<PK>(v[4 * g + 2] + bs[g].z, v[4 * g + 3] + bs[g].w, relu);
    }
#pragma unroll
    for (int gp = 0; gp < 2; ++gp) {
        uint32_t x[2], y[2];
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            auto r = __builtin_amdgcn_permlane32_swap(d[2 * gp][w], d[2 * gp + 1][w], false, false);
            x[w] = r[0]; y[w] = r[1];
        }
        if (pred) *reinterpret_cast<uint4*>(pix_base + 16 * gp + 8 * hh) = make_uint4(x[0], x[1], y[0], y[1]);
    }
}


// one fragment read of k-step STEP: WHICH = 0: B row f=0, 1: A m=0, 2: A m=1, 3: B row f=1 (= order of first use)
template <int STEP, int WHICH>
__device__ __forceinline__ void pp_read(uint32_t a_base0, uint32_t a_base1, const uint32_t (&bb)[4][3], half8_t& dst) {
    constexpr int tap = STEP >> 2, kg = STEP & 3, ky = tap / 3, kx = tap - ky * 3;
    if constexpr (WHICH == 0 || WHICH == 3) {
        const uint32_t addr = bb[ky + (WHICH == 3 ? 1 : 0)][kx] ^ (kg << 5);
        asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr));
    } else {
        constexpr int aoff = (STEP % 18) * 2048 + (WHICH == 2 ? 1024 : 0);
        const uint32_t ab = STEP < 18 ? a_base0 : a_base1;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(ab), "i"(aoff));
    }
}
template <int STEP>
__device__ __forceinline__ void pp_load_step(uint32_t a_base0, uint32_t a_base1, const uint32_t (&bb)[4][3], half8_t& fa0, half8_t& fa1,
                                             half8_t& fb0, half8_t& fb1) {
    pp_read<STEP, 0>(a_base0, a_base1, bb, fb0);
    pp_read<STEP, 1>(a_base0, a_base1, bb, fa0);
    pp_read<STEP, 2>(a_base0, a_base1, bb, fa1);
    pp_read<STEP, 3>(a_base0, a_base1, bb, fb1);
}

template <int N>
__device__ __forceinline__ void pp_wait(half8_t& fa0, half8_t& fa1, half8_t& fb0, half8_t& fb1) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(fa0), "+v"(fa1), "+v"(fb0), "+v"(fb1) : "i"(N));
}

// k-step STEP: its four MFMAs with the four fragment reads of step STEP + 2 interleaved one per MFMA shadow (an in-order
// wave issues ~8 instructions per 32-cycle MFMA; bunching the reads behind the fourth MFMA overflows that gap and leaves
// the other three empty).  On entry the reads of STEP and STEP + 1 are in flight: lgkmcnt(4) retires those of STEP.
template <int STEP, int ABL>
__device__ __forceinline__ void pp_mfma_steps(uint32_t a_base0, uint32_t a_base1, const uint32_t (&bb)[4][3], floatx16 (&acc)[2][2],
                                              half8_t (&fa0)[3], half8_t (&fa1)[3], half8_t (&fb0)[3], half8_t (&fb1)[3]) {
    if constexpr (STEP < 36) {
        constexpr int sl = STEP % 3, nx = (STEP + 2) % 3;
        constexpr bool pre = STEP + 2 < 36;
        constexpr bool rdB = pre && ABL != 1 && ABL != 2, rdA = pre && ABL != 1 && ABL != 3;
        pp_wait<ABL ? 0 : ((STEP + 1 < 36) ? 4 : 0)>(fa0[sl], fa1[sl], fb0[sl], fb1[sl]);
        __builtin_amdgcn_sched_barrier(0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa0[sl], fb0[sl], acc[0][0], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (rdB) pp_read<STEP + 2, 0>(a_base0, a_base1, bb, fb0[nx]);
        __builtin_amdgcn_sched_barrier(0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa1[sl], fb0[sl], acc[1][0], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (rdA) pp_read<STEP + 2, 1>(a_base0, a_base1, bb, fa0[nx]);
        __builtin_amdgcn_sched_barrier(0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa0[sl], fb1[sl], acc[0][1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (rdA) pp_read<STEP + 2, 2>(a_base0, a_base1, bb, fa1[nx]);
        __builtin_amdgcn_sched_barrier(0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa1[sl], fb1[sl], acc[1][1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (rdB) pp_read<STEP + 2, 3>(a_base0, a_base1, bb, fb1[nx]);
        __builtin_amdgcn_sched_barrier(0);
        pp_mfma_steps<STEP + 1, ABL>(a_base0, a_base1, bb, acc, fa0, fa1, fb0, fb1);
    }
}

// conv1a fused into conv1b (FUSE1A): the service phase BUILDS the group's next 10x34x64 halo tile from the u8 image instead of
// DMA-ing it from HBM -- conv1a (1 -> 64 channels, 3x3, K = 9) as two v_mfma_f32_32x32x16_f16 per 32 pixels with split
// operands, x = xh + xl (the 256-entry u8 -> f32 table pre-split into two halfs) and w = wh + wl:
//     K slots 0-8: xh*wh, 9-17: xl*wh, 18-26: xh*wl (xl*wl ~ 2^-24 relative is dropped): fp32-class accuracy on the matrix cores,
// bias + ReLU + the zero padding of conv1b in the epilogue, written straight into the swizzled LDS image.  The 36.9 MB/image
// conv1a activation tensor never exists: conv1b's input traffic drops from 397 MB to 2.3 MB per 8 images, which was what
// held the kernel at ~3 TB/s of DMA next to its MFMA work.
struct Fuse1aArgs {
    const uint8_t* gray; int gstride; int mask_r0, mask_r1;
    const _Float16* w1a_frag;     // [2 k-halves][2 m][64 lanes][8 halfs] A fragments of the split conv1a weights
    const float* bias1a;          // [64]
    const uint32_t* lut_hl;       // [256] half(x) | half(x - half(x)) << 16,  x = float(i) * float(1 / 255.0)
    unsigned long long* trace;    // OMNI_PP_TRACE=1: s_memtime stamps of workgroup 0 (debug only), else nullptr
    const char* zero_page;        // OMNI_ZERO_PAGE_BYTES of zeros: DMA source of the halo pixels outside the image (set by the launcher)
    uint32_t magic_tpi, magic_tx; // ceil(2^32 / act_per_img), ceil(2^32 / tiles_x) (set by the launcher)
    // ConvArgs::skip_*: the tile walk of an image = the tile rows above the rectangle (n_above tiles), the tiles left and right of it in its
    // rows (skip_bw per row, up to n_upto), the tile rows below it; no rectangle: n_above = n_upto = act_per_img = tiles_per_img
    int act_per_img, n_above, n_upto, skip_y0, skip_y1, skip_x0, skip_w, skip_bw;
    uint32_t magic_bw;            // ceil(2^32 / skip_bw)
    int xcd;                      // OMNI_CONV_XCD: xcd_block_id() (set by the launcher)
};

template <bool POOL, int ABL, bool FUSE1A>
__global__ void __launch_bounds__(PP_THREADS)
conv3x3_c64_pp_kernel(const _Float16* __restrict__ in, _Float16* __restrict__ out, const _Float16* __restrict__ wp,
                      const float* __restrict__ bias, int H, int W, int cout, int n_ct, int tiles_x, int tiles_y, int batch, int relu, int dbg,
                      Fuse1aArgs fz) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem_raw;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wl = wave & 3;
    const int n = lane & 31, hh = lane >> 5;
    const int bid = xcd_block_id(fz.xcd);
    const int ct = bid % n_ct, wg = bid / n_ct, nwg = gridDim.x / n_ct;
    const int total = batch * fz.act_per_img;
    const int n_mine = wg < total ? (total - wg + nwg - 1) / nwg : 0;      // tiles of this workgroup: t_k = wg + k * nwg

    {   // weights for this cout tile: 9 taps x 8 KB, fragment order (see pack_weights)
        const uint4* src = reinterpret_cast<const uint4*>(wp + (int64_t)ct * 9 * 4096);
        uint4* dst = reinterpret_cast<uint4*>(smem_raw);
        for (int i = tid; i < C64_W_BYTES / 16; i += PP_THREADS) dst[i] = src[i];
    }
    constexpr int BUFB = FUSE1A ? C64_PIX * 128 : C64_BUF_BYTES;           // no DMA tail padding when the tile is built in place
    char* const buf = smem_raw + C64_W_BYTES + grp * BUFB;                 // this group's halo buffer
    const uint32_t buf_lds = lds0 + C64_W_BYTES + grp * BUFB;
    uint32_t bb[4][3];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int p = (2 * wl + r) * C64_ITW + n + kx;
            bb[r][kx] = buf_lds + p * 128 + ((hh ^ ((p >> 1) & 7)) << 4);
        }
    const uint32_t a_base0 = lds0 + lane * 16, a_base1 = a_base0 + 18 * 2048;

    // t / tiles_per_img and r / tiles_x by multiply-high with ceil(2^32 / d) (exact while t * d < 2^32: the launcher checks): the service phase
    // calls this three times per tile, next to a partner wave that leaves it few issue slots -- two hardware-less integer divisions each were
    // ~200 instructions per phase
    auto tile_origin = [&](int t, int& b, int& ty0, int& tx0) {
        b = fz.magic_tpi ? (int)__umulhi((uint32_t)t, fz.magic_tpi) : t;             // magic 0 = divisor 1 (2^32 does not fit)
        int r = t - b * fz.act_per_img, ry, rx;
        if (r < fz.n_above || r >= fz.n_upto) {                                      // full tile rows above / below the skipped rectangle
            int base = 0;
            if (r >= fz.n_upto) { r -= fz.n_upto; base = fz.skip_y1; }
            ry = fz.magic_tx ? (int)__umulhi((uint32_t)r, fz.magic_tx) : r;
            rx = r - ry * tiles_x; ry += base;
        } else {                                                                     // its rows: the tiles left and right of it
            r -= fz.n_above;
            const int q = fz.magic_bw ? (int)__umulhi((uint32_t)r, fz.magic_bw) : r;
            const int c = r - q * fz.skip_bw;
            ry = fz.skip_y0 + q; rx = c < fz.skip_x0 ? c : c + fz.skip_w;
        }
        ty0 = ry * CONV_TH; tx0 = rx * CONV_TW;
    };
    // per-lane source offsets (bytes, relative to the halo origin) of this wave's 11 DMA instructions, valid for interior tiles
    uint32_t goff[11];
#pragma unroll
    for (int j = 0; j < 11; ++j) {
        int idx = (wl * 11 + j) * 64 + lane;
        idx = idx < C64_CHUNKS ? idx : C64_CHUNKS - 1;        // tail lanes of instruction 42: harmless duplicate
        const int pix = idx >> 3, phys = idx & 7;
        const int iy = pix / C64_ITW, ix = pix - iy * C64_ITW;
        goff[j] = (uint32_t)((iy * W + ix) * 128 + ((phys ^ ((pix >> 1) & 7)) << 4));
    }
    // LDS-DMA of tile t into this group's buffer: 43 wave-instructions of 1 KiB, 11 per wave (10 for the last)
    // (every lambda below takes the tile's origin, decoded ONCE per tile by the service phase: next to an MFMA-streaming partner wave every instruction of
    // the service role costs ~7 cycles, and a decode is ~40 of them)
    auto issue = [&](int b, int ty0, int tx0) {
        const int y0 = ty0 - 1, x0 = tx0 - 1;
        const _Float16* img = in + (int64_t)b * H * W * 64;
        if (y0 >= 0 && y0 + C64_ITH <= H && x0 >= 0 && x0 + C64_ITW <= W) {       // interior (wave-uniform): base + 32-bit lane offset
            const char* org = reinterpret_cast<const char*>(img + ((int64_t)y0 * W + x0) * 64);
#pragma unroll
            for (int j = 0; j < 11; ++j)
                if (wl * 11 + j < 43)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(org + goff[j]),
                                                     (__attribute__((address_space(3))) void*)(buf + (wl * 11 + j) * 1024), 16, 0, 0);
            return;
        }
#pragma unroll
        for (int j = 0; j < 11; ++j) {
            const int wi = wl * 11 + j;
            if (wi < 43) {
                int idx = wi * 64 + lane;
                idx = idx < C64_CHUNKS ? idx : C64_CHUNKS - 1;
                const int pix = idx >> 3, phys = idx & 7;
                const int iy = pix / C64_ITW, ix = pix - iy * C64_ITW;
                const int gy = y0 + iy, gx = x0 + ix;
                const int logical = phys ^ ((pix >> 1) & 7);
                // halo pixels outside the image come from a page of zeros: the zero padding lands with the data, no fix-up pass
                const char* g = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? reinterpret_cast<const char*>(img + ((int64_t)gy * W + gx) * 64 + logical * 8)
                                                                         : fz.zero_page + ((uint32_t)(idx * 16) & (OMNI_ZERO_PAGE_BYTES - 16));
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                 (__attribute__((address_space(3))) void*)(buf + wi * 1024), 16, 0, 0);
            }
        }
    };
    // bias lives in LDS behind the halo buffers (256 B): holding it in registers (32 VGPRs) spills next to 64 accumulators,
    // 48 fragment registers and the DMA / fragment address tables
    float* const bias_lds = reinterpret_cast<float*>(smem_raw + C64_W_BYTES + 2 * BUFB);
    if (tid < 64) bias_lds[tid] = bias[ct * 64 + tid];
    float* const bias1a_lds = bias_lds + 64;                              // FUSE1A only
    uint32_t* const lut_lds = reinterpret_cast<uint32_t*>(bias_lds + 128);
    if constexpr (FUSE1A) {
        if (tid < 64) bias1a_lds[tid] = fz.bias1a[tid];
        if (tid < 256 && fz.lut_hl) lut_lds[tid] = fz.lut_hl[tid];
    }
    auto bias4 = [&](int m, float4 (&bs)[4]) {
#pragma unroll
        for (int g = 0; g < 4; ++g) bs[g] = *reinterpret_cast<const float4*>(bias_lds + m * 32 + 8 * g + 4 * hh);
    };

    // ---- FUSE1A: wave wl builds halo pixels [96 wl, 96 wl + 96) = fragments 3 wl + fi (p = 32 (3 wl + fi) + n, row-major over the
    // 10 x 34 tile): at most 4 halo rows, i.e. a 6-row x 40-byte patch of the u8 image (4-byte aligned columns).  ONE
    // buffer_load_dword per wave fetches the patch (the descriptor's bounds check makes every address safe; rows/columns
    // outside the image are clamped and masked later), it is parked in a wave-private 240-byte LDS slot (shared by the two
    // groups, which never service at the same time) and the 27 taps of a lane are ds_read_u8 at immediate offsets.
    const int fr_r0 = (96 * wl) / C64_ITW;
    int fr_iy[3], fr_ix[3];                  // tile-invariant halo coordinates of this lane's pixel in its three fragments
#pragma unroll
    for (int fi = 0; fi < 3; ++fi) {
        const int p = (3 * wl + fi) * 32 + n;
        fr_iy[fi] = p / C64_ITW; fr_ix[fi] = p - fr_iy[fi] * C64_ITW;
    }
    unsigned char* const patch = reinterpret_cast<unsigned char*>(lut_lds + 256) + wl * 240;
    int tap_off[5];                          // patch offsets of this half-wave's taps: 0-4 (lanes 0-31) or 5-8 (+ a dummy) (lanes 32-63)
#pragma unroll
    for (int k = 0; k < 5; ++k) { const int tp = hh ? (k < 4 ? 5 + k : 8) : k; tap_off[k] = (tp / 3) * 40 + tp % 3; }
    auto build_issue = [&](int b, int ty0, int tx0, uint32_t& pv) {
        const int j = lane / 10, d = lane - j * 10;
        int yy = ty0 - 2 + fr_r0 + j, xo = ((tx0 - 2) & ~3) + 4 * d;
        yy = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);
        xo = xo < 0 ? 0 : (xo > fz.gstride - 4 ? fz.gstride - 4 : xo);
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(fz.gray), 0, batch * H * fz.gstride, 0x00020000);
        pv = __builtin_amdgcn_raw_buffer_load_b32(rsrc, (b * H + yy) * fz.gstride + xo, 0, 0);
    };
    // table reads and operand packing of all three fragments first (their latencies overlap), then MFMAs + stores per fragment
    auto build_finish = [&](int b, int ty0, int tx0, uint32_t pv, unsigned long long* stamp = nullptr) {
        (void)b;
        const int cx0 = (tx0 - 2) & ~3, xsh = (tx0 - 2) - cx0;
        {   // park the patch in LDS with everything that must read as x = 0 already zeroed: rows outside the image or inside the
            // fisheye mask, columns outside the image (the loads were clamped to valid addresses) -- no per-tap checks later
            const int j = lane / 10, d = lane - j * 10;
            const int yy = ty0 - 2 + fr_r0 + j, x0c = cx0 + 4 * d;
            const bool rok = yy >= 0 && yy < H && !(yy >= fz.mask_r0 && yy < fz.mask_r1);
            uint32_t bm = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) bm |= (x0c + k >= 0 && x0c + k < W) ? (0xFFu << (8 * k)) : 0u;
            if (lane < 60) reinterpret_cast<uint32_t*>(patch)[lane] = rok ? (pv & bm) : 0u;
        }
        half8_t wa[2][2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int m = 0; m < 2; ++m) wa[j][m] = *reinterpret_cast<const half8_t*>(fz.w1a_frag + ((j * 2 + m) * 64 + lane) * 8);
        const int nf = (3 * wl + 2) * 32 < C64_PIX ? 3 : 2;                   // wave-uniform: fragments 0..10 over 4 waves
        // K slots are split between the half-waves so that a lane touches only its own taps: lanes 0-31 own taps 0-4, lanes 32-63
        // taps 5-8 and the bias.  Per MFMA a lane supplies 8 slots = 4 dwords:
        //   hh = 0:  [xh0 xl0][xh1 xl1][xh2 xl2][xh3 xl3] | [xh4 xl4][xh0 xh1][xh2 xh3][xh4 0]     (weights wh,wh per tap | wh4 wh4 wl0..wl4 0)
        //   hh = 1:  [xh5 xl5][xh6 xl6][xh7 xl7][xh8 xl8] | [xh5 xh6][xh7 xh8][1 1][0 0]           (weights wh,wh per tap | wl5..wl8 bias_hi bias_lo 0 0)
        // and the table entry (xh | xl << 16) IS the first kind of dword.
        half8_t B0[3], B1[3];
#pragma unroll
        for (int fi = 0; fi < 3; ++fi) {
            const int gy = ty0 - 1 + fr_iy[fi], gx = tx0 - 1 + fr_ix[fi];
            // halo pixels outside the image are conv1b's zero padding: every tap and the bias slots read as 0
            const uint32_t pinm = (fr_iy[fi] < C64_ITH && gy >= 0 && gy < H && gx >= 0 && gx < W) ? 0xFFFFFFFFu : 0u;
            int iyc = fr_iy[fi] - fr_r0;                                      // tail lanes of fragment 10 (p >= 340): stay inside the patch
            iyc = iyc > 3 ? 3 : iyc;
            const unsigned char* pb = patch + iyc * 40 + fr_ix[fi] + xsh;
            uint32_t T[5];
            const uint32_t one2 = 0x3C003C00u & pinm;
            uint32_t b1[4];
            if (!fz.lut_hl) {
                // no table (conv1a_pack_u8_weights): the byte p IS the operand -- 0x4400 | p = half(4 + p / 256) exactly, in both halves of the dword;
                // the weights carry 256 / 255 and the bias slot takes the 4 sum(w) back.  One VALU operation per tap instead of a gather whose 64 lanes
                // hit the table's banks at random
#pragma unroll
                for (int k = 0; k < 5; ++k) T[k] = ((uint32_t)pb[tap_off[k]] * 0x00010001u + 0x44004400u) & pinm;
                b1[0] = hh ? one2 : T[4]; b1[1] = 0u; b1[2] = 0u; b1[3] = 0u;
            } else {
#pragma unroll
            for (int k = 0; k < 5; ++k) T[k] = lut_lds[pb[tap_off[k]]] & pinm;
            const uint32_t h01 = (T[0] & 0xFFFFu) | (T[1] << 16), h23 = (T[2] & 0xFFFFu) | (T[3] << 16);
            b1[0] = hh ? h01 : T[4];
            b1[1] = hh ? h23 : h01;
            b1[2] = hh ? one2 : h23;
            b1[3] = hh ? 0u : (T[4] & 0xFFFFu);
            }
            B0[fi] = __builtin_bit_cast(half8_t, make_uint4(T[0], T[1], T[2], T[3]));
            B1[fi] = __builtin_bit_cast(half8_t, make_uint4(b1[0], b1[1], b1[2], b1[3]));
        }
        if (stamp) stamp[0] = __builtin_amdgcn_s_memtime();                  // operands of all three fragments packed (patch landed, table read)
        // all 12 MFMAs of the wave's three fragments first -- the six of K step 0, then the six of K step 1, each depending on one issued six
        // instructions earlier -- and only then the conversions: one fragment at a time left every ReLU / pack waiting for the matrix pipe
        // (~250 cycles per fragment; the accumulators of the tile just written back are free registers here).
        // The partner wave streams MFMAs back to back at priority 2: without outranking it for these twelve, each of them waits for a gap in a
        // matrix pipe that has none
        floatx16 a1[3][2];
        __builtin_amdgcn_s_setprio(3);
#pragma unroll
        for (int fi = 0; fi < 3; ++fi) {
            if (fi >= nf) continue;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
#pragma unroll
                for (int i = 0; i < 16; ++i) a1[fi][m][i] = 0.f;
                a1[fi][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[0][m], B0[fi], a1[fi][m], 0, 0, 0);
            }
        }
#pragma unroll
        for (int fi = 0; fi < 3; ++fi) {
            if (fi >= nf) continue;
#pragma unroll
            for (int m = 0; m < 2; ++m) a1[fi][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[1][m], B1[fi], a1[fi][m], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        if (stamp) stamp[1] = __builtin_amdgcn_s_memtime();                  // the twelve MFMAs issued
#pragma unroll
        for (int fi = 0; fi < 3; ++fi) {
            if (fi >= nf) continue;
            const int p = (3 * wl + fi) * 32 + n;
            const uint32_t sw = (uint32_t)((p >> 1) & 7);
            char* const prow = buf + p * 128 + 8 * hh;
            if (p < C64_PIX) {
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq)
                        *reinterpret_cast<uint2*>(prow + (((uint32_t)(m * 4 + gq) ^ sw) << 4)) =
                            make_uint2(pack_relu_f16(a1[fi][m][4 * gq + 0], a1[fi][m][4 * gq + 1], 1), pack_relu_f16(a1[fi][m][4 * gq + 2], a1[fi][m][4 * gq + 3], 1));
            }
        }
    };

    // group g computes tiles k = g, g+2, ... in phases p = k; services (epilogue of k, DMA of k+2) in phase k+1.
    // group 0 loads its first tile here, group 1 during phase 0.
    int k_load = grp;                       // next tile index this group will DMA
    int ob = 0, oty0 = 0, otx0 = 0;          // origin of the tile this group built last = the tile whose epilogue comes next
    __syncthreads();                         // weights / bias / table staged (FUSE1A reads them in the prologue already)
    if (grp == 0 && k_load < n_mine) {
        tile_origin(wg + k_load * nwg, ob, oty0, otx0);
        if constexpr (FUSE1A) {
            uint32_t pv;
            build_issue(ob, oty0, otx0, pv);
            build_finish(ob, oty0, otx0, pv);
        } else {
            issue(ob, oty0, otx0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        k_load += 2;
    }
    __syncthreads();

    floatx16 acc[2][2];
    int t_pending = -1;
    for (int p = 0; p <= n_mine; ++p) {
        if ((p & 1) == grp) {
            if (p < n_mine) {                // compute role: tile k = p
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int f = 0; f < 2; ++f)
#pragma unroll
                        for (int i = 0; i < 16; ++i) acc[m][f][i] = 0.f;
                // keep the 54 XOR-ed B addresses out of loop-invariant hoisting (they would pin 54 VGPRs and spill)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) asm volatile("" : "+v"(bb[r][kx]));
                half8_t fa0[3], fa1[3], fb0[3], fb1[3];
                pp_load_step<0>(a_base0, a_base1, bb, fa0[0], fa1[0], fb0[0], fb1[0]);
                pp_load_step<1>(a_base0, a_base1, bb, fa0[1], fa1[1], fb0[1], fb1[1]);
                __builtin_amdgcn_sched_barrier(0);
                if (!(dbg & 16)) __builtin_amdgcn_s_setprio(2);       // the matrix-pipe wave outranks its SIMD partner's service work
                const bool trc = fz.trace && blockIdx.x == 0 && wl == 0 && lane == 0 && p >= 2 && p < 6;
                if (trc) fz.trace[p * 8 + 4] = __builtin_amdgcn_s_memtime();
                pp_mfma_steps<0, ABL>(a_base0, a_base1, bb, acc, fa0, fa1, fb0, fb1);
                if (trc) fz.trace[p * 8 + 5] = __builtin_amdgcn_s_memtime();
                __builtin_amdgcn_s_setprio(0);
                t_pending = wg + p * nwg;
            }
        } else {                             // service role
            // bias BEFORE the DMA is issued: hipcc orders any ds_read behind an in-flight LDS-DMA with s_waitcnt vmcnt(0), which
            // would expose the whole DMA latency in front of the epilogue
            float4 bs[POOL ? 1 : 2][4];
            if constexpr (POOL) bias4(n & 1, bs[0]);
            else { bias4(0, bs[0]); bias4(1, bs[1]); }
            __builtin_amdgcn_sched_barrier(0);
            const bool load = k_load < n_mine && !(dbg & 1);
            if (dbg & 32) __builtin_amdgcn_s_setprio(3);
            uint32_t pv = 0;
            const bool tr = fz.trace && blockIdx.x == 0 && wl == 0 && lane == 0 && p >= 2 && p < 6;
            if (tr) fz.trace[p * 8 + 0] = __builtin_amdgcn_s_memtime();
            int nb = 0, nty0 = 0, ntx0 = 0;   // origin of the tile built in this phase
            if (load) {
                tile_origin(wg + k_load * nwg, nb, nty0, ntx0);
                if constexpr (FUSE1A) build_issue(nb, nty0, ntx0, pv);   // the patch load flies under the epilogue
                else issue(nb, nty0, ntx0);
            }
            if (tr) fz.trace[p * 8 + 1] = __builtin_amdgcn_s_memtime();
            bool full = false;               // every lane of every store instruction of the epilogue is active
            if (t_pending >= 0 && !(dbg & 2)) {
                const int b = ob, ty0 = oty0, tx0 = otx0;        // (= tile_origin(t_pending): no build of this group in between)
                full = (ty0 + CONV_TH <= H) && (tx0 + CONV_TW <= W);
                const int ox = tx0 + n;
                if constexpr (POOL) {
                    // 2x2 max-pool: rows 2wl / 2wl+1 are the two accumulators of a lane, columns n / n^1 one DPP exchange.
                    // Even lanes then store M-fragment 0 and odd lanes M-fragment 1 of pooled pixel n >> 1.
                    const int oy = ty0 + 2 * wl;
                    const bool odd = n & 1;
                    float v[16], q0[16], q1[16];
                    // (sending only the fragment the partner keeps -- one exchange instead of two -- measured slower: 0.158 vs 0.139 ms)
#pragma unroll
                    for (int i = 0; i < 16; ++i) { q0[i] = vmax(acc[0][0][i], acc[0][1][i]); q1[i] = vmax(acc[1][0][i], acc[1][1][i]); }
                    __builtin_amdgcn_sched_barrier(0);                // all 32 vertical maxima issue before the first DPP read (wait states)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const float h0 = vmax_swap_pairs(q0[i]), h1 = vmax_swap_pairs(q1[i]);
                        v[i] = odd ? h1 : h0;
                    }
                    _Float16* o = out + (((int64_t)b * (H >> 1) + (oy >> 1)) * (W >> 1) + (ox >> 1)) * cout + ct * 64 + (odd ? 32 : 0);
                    store_frag16<true>(v, bs[0], o, hh, relu, (oy < H) && (ox < W));
                } else {
#pragma unroll
                    for (int f = 0; f < 2; ++f) {
                        const int oy = ty0 + 2 * wl + f;
#pragma unroll
                        for (int m = 0; m < 2; ++m) {
                            float v[16];
#pragma unroll
                            for (int i = 0; i < 16; ++i) v[i] = acc[m][f][i];
                            _Float16* o = out + (((int64_t)b * H + oy) * W + ox) * cout + ct * 64 + m * 32;
                            store_frag16<false>(v, bs[m], o, hh, relu, (oy < H) && (ox < W));
                        }
                    }
                }
                t_pending = -1;
            }
            if (tr) fz.trace[p * 8 + 2] = __builtin_amdgcn_s_memtime();
            if (load && FUSE1A) {
                if constexpr (FUSE1A) build_finish(nb, nty0, ntx0, pv, tr ? fz.trace + p * 8 + 6 : nullptr);
                if (tr) fz.trace[p * 8 + 3] = __builtin_amdgcn_s_memtime();
                k_load += 2;
                ob = nb; oty0 = nty0; otx0 = ntx0;
            } else if (load) {
                // vmcnt retires in issue order: the DMA loads were issued before the epilogue's stores, so waiting down to
                // the number of store instructions leaves the stores in flight across the barrier (only when every store
                // instruction was certainly issued, i.e. the output tile is interior; otherwise wait for everything)
                if (full) {
                    if constexpr (POOL) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                if (tr) fz.trace[p * 8 + 3] = __builtin_amdgcn_s_memtime();
                k_load += 2;
                ob = nb; oty0 = nty0; otx0 = ntx0;
            }
        }
        if (dbg & 32) __builtin_amdgcn_s_setprio(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // own LDS traffic done; global stores may stay in flight
        __builtin_amdgcn_s_barrier();
    }
}

template <bool POOL, int ABL, bool FUSE1A = false>
static int launch_conv_pp_abl(hipStream_t st, const ConvArgs& a, int n_cu, int dbg, const Fuse1aArgs& fz = Fuse1aArgs{}) {
    auto kfn = conv3x3_c64_pp_kernel<POOL, ABL, FUSE1A>;
    constexpr int smem_bytes = FUSE1A ? PP_SMEM_FUSED : PP_SMEM;
    static_assert(smem_bytes <= 160 * 1024, "LDS budget");
    OMNI_HIP_TRY(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes));
    const int tiles_x = cdiv(a.W, CONV_TW), tiles_y = cdiv(a.H, CONV_TH), n_ct = a.cout / 64;
    Fuse1aArgs fzz = fz;
    // the tile walk of one image, without the rectangle the caller already holds (ConvArgs::skip_*)
    const bool skip = a.skip_ty1 > a.skip_ty0 && a.skip_tx1 > a.skip_tx0;
    OMNI_REQUIRE(!skip || (a.skip_ty0 >= 0 && a.skip_ty1 <= tiles_y && a.skip_tx0 >= 0 && a.skip_tx1 <= tiles_x), OMNI_ERR_INVALID, "conv: skip rectangle outside the tile grid");
    fzz.skip_y0 = skip ? a.skip_ty0 : 0; fzz.skip_y1 = skip ? a.skip_ty1 : 0; fzz.skip_x0 = skip ? a.skip_tx0 : 0; fzz.skip_w = skip ? a.skip_tx1 - a.skip_tx0 : 0;
    fzz.skip_bw = tiles_x - fzz.skip_w;
    fzz.act_per_img = tiles_x * tiles_y - (fzz.skip_y1 - fzz.skip_y0) * fzz.skip_w;
    fzz.n_above = skip ? fzz.skip_y0 * tiles_x : fzz.act_per_img;
    fzz.n_upto = fzz.n_above + (fzz.skip_y1 - fzz.skip_y0) * fzz.skip_bw;
    OMNI_REQUIRE(fzz.act_per_img > 0, OMNI_ERR_INVALID, "conv: the skip rectangle covers the whole image");
    const int total = a.batch * fzz.act_per_img;
    int per_ct = n_cu / n_ct;
    if (per_ct < 1) per_ct = 1;
    if (per_ct > cdiv(total, 2)) per_ct = cdiv(total, 2);      // at least two tiles per workgroup: one per wave group
    OMNI_REQUIRE(a.zero_page, OMNI_ERR_INVALID, "conv: ConvArgs.zero_page is not set");
    fzz.zero_page = reinterpret_cast<const char*>(a.zero_page);
    OMNI_REQUIRE((int64_t)total * (tiles_x * tiles_y) < (1ll << 32), OMNI_ERR_INVALID, "conv: too many tiles for the multiply-high division");
    auto magic = [](int d) { return d > 1 ? (uint32_t)(((1ull << 32) + (uint64_t)d - 1) / (uint64_t)d) : 0u; };      // 0 = divisor 1 (2^32 does not fit)
    fzz.magic_tpi = magic(fzz.act_per_img);
    fzz.magic_tx = magic(tiles_x);
    fzz.magic_bw = magic(fzz.skip_bw);
    fzz.xcd = config_process()[CFG_CONV_XCD];
    // OMNI_PP_TRACE=1 (debug): s_memtime stamps of workgroup 0's phases 2-5 for the layers without the conv1a fusion (conv1ab_fused prints its own)
    static const bool want_trace = config_process()[CFG_PP_TRACE] != 0;
    static unsigned long long* trace_dev = nullptr;
    if (want_trace && !FUSE1A) {
        if (!trace_dev) OMNI_HIP_TRY(hipMalloc((void**)&trace_dev, 64 * 8));
        OMNI_HIP_TRY(hipMemsetAsync(trace_dev, 0, 64 * 8, st));
        fzz.trace = trace_dev;
    }
    hipLaunchKernelGGL(kfn, dim3(per_ct * n_ct), dim3(PP_THREADS), smem_bytes, st, reinterpret_cast<const _Float16*>(a.in),
                       reinterpret_cast<_Float16*>(a.out), reinterpret_cast<const _Float16*>(a.w_packed), a.bias, a.H, a.W, a.cout, n_ct,
                       tiles_x, tiles_y, a.batch, a.relu ? 1 : 0, dbg, fzz);
    OMNI_LAUNCH_CHECK();
    if (want_trace && !FUSE1A) {
        unsigned long long h[64];
        OMNI_HIP_TRY(hipMemcpyAsync(h, trace_dev, sizeof(h), hipMemcpyDeviceToHost, st));
        OMNI_HIP_TRY(hipStreamSynchronize(st));
        static int launches = 0;
        if (total >= 16 * per_ct && launches++ < 2)
            for (int p = 2; p < 6; ++p)
                if (h[p * 8])
                    fprintf(stderr, "pp trace pool=%d H=%d W=%d cout=%d phase %d: service: issue %llu epilogue %llu dma wait %llu | partner mfma loop %llu\n", (int)POOL, a.H, a.W,
                            a.cout, p, h[p * 8 + 1] - h[p * 8], h[p * 8 + 2] - h[p * 8 + 1], h[p * 8 + 3] - h[p * 8 + 2], h[(p ^ 1) * 8 + 5] - h[(p ^ 1) * 8 + 4]);
    }
    return OMNI_OK;
}
template <bool POOL>
static int launch_conv_pp(hipStream_t st, const ConvArgs& a, int n_cu) {
    // OMNI_PP_DBG: timing ablations only (WRONG results): bit 0 no DMA, bit 1 no epilogue, bits 2-3: 1 no fragment reads, 2 no B reads, 3 no A reads
    static const int dbg = config_process()[CFG_PP_DBG];
    switch (dbg >> 2) {
        case 1: return launch_conv_pp_abl<POOL, 1>(st, a, n_cu, dbg & 3);
        case 2: return launch_conv_pp_abl<POOL, 2>(st, a, n_cu, dbg & 3);
        case 3: return launch_conv_pp_abl<POOL, 3>(st, a, n_cu, dbg & 3);
        default: return launch_conv_pp_abl<POOL, 0>(st, a, n_cu, dbg & 3);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// v4 "register-stationary" kernel for the fp16 3x3 layers with 128 INPUT channels (conv3b, conv4a, conv4b, convPa|convDa:
// 28 % of the FLOPs).  With 128 channels a halo pixel is 256 bytes: resident weights (147 KB per 64 output channels) and two
// halo buffers no longer fit LDS together, and a 32-channel weight tile leaves too little register tiling for the LDS read
// bandwidth.  So the weights leave LDS altogether:
//   * 256 threads = 4 waves, ONE per SIMD with the whole 512-register file; wave w keeps the A fragments of ITS 32 output
//     channels for all K = 9 x 128 in registers (72 fragments = 288 VGPRs, loaded once per workgroup) -- the four waves of a
//     workgroup cover 128 output channels;
//   * LDS holds only the input: two 8 x 34-pixel halo tiles (6 x 32 outputs, 69.6 KB each, 16-byte chunks XOR-swizzled with
//     pixel & 15), filled by LDS-DMA for tile t+1 while tile t computes: one barrier per tile;
//   * every wave sweeps the whole tile: per (kx, 16-channel group) the 8 halo rows are read ONCE (192 ds_read_b128 per tile) and
//     each feeds up to three MFMAs (the three ky that map the row onto an output row): 432 MFMAs per tile per wave,
//     0.44 LDS reads per MFMA.
// ---------------------------------------------------------------------------------------------------------------
template <int J, int N, typename F>
__device__ __forceinline__ void spl2_for_each(F&& f) {
    if constexpr (J < N) { f(std::integral_constant<int, J>{}); spl2_for_each<J + 1, N>(f); }
}
#define CSP_KP 8                                  // key points per tile of the sparse descriptor kernels (32 corner cells)
bool conv_rs_transposed(int H, int W);
// (RS_TH x RS_TW = 6 x 32: conv.h)
#define RS_ITH (RS_TH + 2)
#define RS_ITW (RS_TW + 2)
#define RS_PIX (RS_ITH * RS_ITW)                 // 272 halo pixels
#define RS_CHUNKS (RS_PIX * 16)                  // 4352 16-byte chunks = 68 wave-instructions of 1 KiB
#define RS_BUF_BYTES (RS_PIX * 256)              // 69 632
#define RS_SMEM (2 * RS_BUF_BYTES + 512)         // + this workgroup's 128 biases

// read L = (kx * 8 + r) * 8 + kg: the 16-channel group kg is innermost so that eight consecutive reads share one address base
// (computed on the fly: 24 resident bases would not fit next to 288 weight and 96 accumulator registers)
template <int L>
__device__ __forceinline__ void rs_read(uint32_t row_base /* lds + (n * 256) */, int n, int hh, half8_t& dst) {
    constexpr int kx = L / 64, r = (L / 8) % 8, kg = L % 8;
    constexpr int pc = r * RS_ITW + kx;
    const uint32_t addr = (row_base + pc * 256 + ((((pc + n) & 15) ^ hh) << 4)) ^ (kg << 5);
    asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr));
}
template <int N>
__device__ __forceinline__ void rs_wait(half8_t& f) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "i"(N)); }

template <int L, typename IssueFn>
__device__ __forceinline__ void rs_steps(uint32_t row_base, int n, int hh, const half8_t (&wreg)[72], floatx16 (&acc)[RS_TH], half8_t (&fb)[3],
                                         IssueFn& issue_piece) {
    if constexpr (L < 192) {
        constexpr int kx = L / 64, r = (L / 8) % 8, kg = L % 8;
        if constexpr (L + 2 < 192) rs_read<L + 2>(row_base, n, hh, fb[(L + 2) % 3]);
        rs_wait<(L + 2 < 192) ? 2 : (L + 1 < 192 ? 1 : 0)>(fb[L % 3]);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (r >= 0 && r < RS_TH) acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[(0 * 3 + kx) * 8 + kg], fb[L % 3], acc[r], 0, 0, 0);
        if constexpr (r - 1 >= 0 && r - 1 < RS_TH) acc[r - 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[(1 * 3 + kx) * 8 + kg], fb[L % 3], acc[r - 1], 0, 0, 0);
        if constexpr (r - 2 >= 0 && r - 2 < RS_TH) acc[r - 2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[(2 * 3 + kx) * 8 + kg], fb[L % 3], acc[r - 2], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        rs_steps<L + 1>(row_base, n, hh, wreg, acc, fb, issue_piece);
    }
}

// TRN: transposed tiles -- the 32-pixel fragments run along y, the six fragment rows along x (the LDS image, the k order and every MFMA are
// those of the plain kernel; only the pixel <-> address maps differ).  A 60x75 layer is 2 x 13 tiles of 32x6 instead of 3 x 10 of 6x32:
// 0.90 instead of 0.78 of the computed pixels are real (the 75-pixel rows wasted 22 % of every 32-pixel fragment).  Same products, summed
// tap-column-major instead of tap-row-major (fp32 accumulation order): equal to the plain kernel to fp32 rounding, batch-independent as before.
template <bool POOL, bool TRN = false>
__global__ void __launch_bounds__(256, 1)
conv3x3_c128_rs_kernel(const _Float16* __restrict__ in, _Float16* __restrict__ out, const _Float16* __restrict__ wp,
                       const float* __restrict__ bias, int H, int W, int cout, int n_cg, int tiles_x, int tiles_y, int batch, int relu,
                       const char* __restrict__ zero_page /* OMNI_ZERO_PAGE_BYTES of zeros: DMA source of the halo pixels outside the image */,
                       unsigned long long* trace /* OMNI_RS_TRACE=1: s_memtime stamps of workgroup 0 (debug only), else nullptr */,
                       RsSkip sk /* ConvArgs::skip_* in this kernel's tile grid (plain tiles only): the tiles that run */) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem_raw;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, hh = lane >> 5;
    const bool tr = trace != nullptr && blockIdx.x == 0 && tid == 0;
    int tk = 0;
    const int bid = xcd_block_id(sk.xcd);
    const int cg = bid % n_cg, wg = bid / n_cg, nwg = gridDim.x / n_cg;
    const int tiles_per_img = sk.act;              // (the tiles that run: all of them unless a rectangle is left out)
    const int total = batch * tiles_per_img;
    const int g32 = cg * 4 + wave;                 // this wave's group of 32 output channels

    // A fragments: the generic packed layout [cout tile 64][cin chunk 64][tap][kg4][m][lane][8]; k-step s = tap * 8 + chunk * 4 + kg4
    half8_t wreg[72];
    {
        const _Float16* wbase = wp + (int64_t)(g32 >> 1) * 2 * 9 * 4096 + (g32 & 1) * 512 + lane * 8;
#pragma unroll
        for (int s2 = 0; s2 < 72; ++s2) {
            const int tk = s2 >> 3, ch = (s2 >> 2) & 1, kg4 = s2 & 3;
            const int tap = TRN ? (tk % 3) * 3 + tk / 3 : tk;          // transposed tiles: the kernel's (row, column) shifts are the image's (column, row)
            wreg[s2] = *reinterpret_cast<const half8_t*>(wbase + ((ch * 9 + tap) * 4 + kg4) * 1024);
        }
    }
    float* const bias_lds = reinterpret_cast<float*>(smem_raw + 2 * RS_BUF_BYTES);
    if (tid < 128) bias_lds[tid] = bias[cg * 128 + tid];

    // tile t = (image, index r among the image's tiles that run): the tile rows above the rectangle, the tiles left and right of it in its own rows,
    // the tile rows below (tests/test_mask_skip_cpu.py walks this arithmetic over every rectangle of several grids)
    auto tile_origin = [&](int t, int& b, int& ty0, int& tx0) {
        b = t / tiles_per_img;
        int r = t - b * tiles_per_img, ry, rx;
        if (r < sk.n_above || r >= sk.n_upto) {
            int base = 0;
            if (r >= sk.n_upto) { r -= sk.n_upto; base = sk.y1; }
            ry = r / tiles_x; rx = r - ry * tiles_x; ry += base;
        } else {
            r -= sk.n_above;
            const int q = r / sk.bw, c = r - q * sk.bw;
            ry = sk.y0 + q; rx = c < sk.x0 ? c : c + sk.w;
        }
        ty0 = ry * (TRN ? RS_TW : RS_TH); tx0 = rx * (TRN ? RS_TH : RS_TW);
    };
    // DMA: 68 wave-instructions of 1 KiB per tile, wave w issues pieces 17 w .. 17 w + 16 into buffer `which`.  Buffer-addressed: the
    // descriptor is the image, so halo rows above / below it are out of range (the offset wraps negative or passes the image's bytes) and
    // arrive as zeros; the lanes of halo columns left / right of it are sent out of range by one v_cndmask.  Every tile costs the same:
    // the former border path (per-lane coordinates + a page of zeros as DMA source) was 3.5 k cycles per border tile against 14.6 k for
    // the tile's MFMAs, and at 60 x 75 every tile is a border tile.
    uint32_t goff[17];                        // byte offset of this lane's chunk of piece j relative to the halo origin
    uint32_t hx[4] = {0, 0, 0, 0};            // ... and the halo column (image x) of its pixel, 6 bits per piece
#pragma unroll
    for (int j = 0; j < 17; ++j) {
        const int idx = (wave * 17 + j) * 64 + lane;
        const int pix = idx >> 4, phys = idx & 15;
        const int iv = pix / RS_ITW, iu = pix - iv * RS_ITW;
        const int iy = TRN ? iu : iv, ix = TRN ? iv : iu;
        goff[j] = (uint32_t)((iy * W + ix) * 256 + ((phys ^ (pix & 15)) << 4));
        hx[j / 5] |= (uint32_t)ix << (6 * (j % 5));
    }
    const uint32_t in_img_bytes = (uint32_t)H * W * 256u;
    auto issue = [&](int t, int which) {
        int b, ty0, tx0;
        tile_origin(t, b, ty0, tx0);
        const int y0 = ty0 - 1, x0 = tx0 - 1;
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(in) + (int64_t)b * H * W * 128, 0, in_img_bytes, 0x00020000);
        const uint32_t sorg = (uint32_t)((y0 * W + x0) * 256);
        char* base = smem_raw + which * RS_BUF_BYTES + wave * 17 * 1024;
#pragma unroll
        for (int j = 0; j < 17; ++j) {
            const uint32_t x = (uint32_t)(x0 + (int)((hx[j / 5] >> (6 * (j % 5))) & 63u));
            const uint32_t voff = x < (uint32_t)W ? goff[j] + sorg : 0x80000000u;
#if __HIP_DEVICE_COMPILE__          // hipcc's host pass has no target for this builtin and silently drops the kernel's launch stub when it meets it
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(base + j * 1024), 16, voff, 0, 0, 0);
#else
            (void)rsrc; (void)base; (void)voff;
#endif
        }
    };
    int t = wg;
    if (t < total) issue(t, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int cur = 0;
    for (; t < total; t += nwg, cur ^= 1) {
        const int tn = t + nwg;
        if (tr && tk < 8) { trace[tk * 8 + 0] = __builtin_amdgcn_s_memtime(); trace[tk * 8 + 6] = (unsigned long long)t; }
        // (issuing these 17 DMA instructions from inside the MFMA loop, one every 10 reads, measured SLOWER: conv3b 0.055 -> 0.063 ms --
        // an in-order wave pays for every instruction placed between its MFMAs)
        if (tn < total) issue(tn, cur ^ 1);
        if (tr && tk < 8) trace[tk * 8 + 1] = __builtin_amdgcn_s_memtime();
        auto issue_piece = [&](int) {};
        const uint32_t row_base = lds0 + cur * RS_BUF_BYTES + n * 256;
        floatx16 acc[RS_TH];
#pragma unroll
        for (int f = 0; f < RS_TH; ++f)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[f][i] = 0.f;
        half8_t fb[3];
        rs_read<0>(row_base, n, hh, fb[0]);
        rs_read<1>(row_base, n, hh, fb[1]);
        __builtin_amdgcn_sched_barrier(0);
        rs_steps<0>(row_base, n, hh, wreg, acc, fb, issue_piece);
        if (tr && tk < 8) trace[tk * 8 + 2] = __builtin_amdgcn_s_memtime();

        {   // epilogue: (2x2 max-pool) + bias + ReLU, 16-byte NHWC stores
            int b, ty0, tx0;
            tile_origin(t, b, ty0, tx0);
            const int ox = tx0 + n;
            float4 bs[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) bs[g] = *reinterpret_cast<const float4*>(bias_lds + wave * 32 + 8 * g + 4 * hh);
            if constexpr (POOL) {
#pragma unroll
                for (int f2 = 0; f2 < RS_TH / 2; ++f2) {
                    const int oy = ty0 + 2 * f2;
                    float v[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const float q = vmax_med3(acc[2 * f2][i], acc[2 * f2 + 1][i]);
                        v[i] = vmax_med3(q, dpp_swap_pairs(q));
                    }
                    _Float16* o = out + (((int64_t)b * (H >> 1) + (oy >> 1)) * (W >> 1) + (ox >> 1)) * cout + g32 * 32;
                    store_frag16<false>(v, bs, o, hh, relu, (oy < H) && (ox < W) && !(n & 1));
                }
            } else {
#pragma unroll
                for (int f = 0; f < RS_TH; ++f) {
                    const int oy = TRN ? ty0 + n : ty0 + f, oxx = TRN ? tx0 + f : ox;
                    float v[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) v[i] = acc[f][i];
                    _Float16* o = out + (((int64_t)b * H + oy) * W + oxx) * cout + g32 * 32;
                    store_frag16<false>(v, bs, o, hh, relu, (oy < H) && (oxx < W));
                }
            }
        }
        if (tr && tk < 8) trace[tk * 8 + 3] = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // next tile landed (and this tile's stores retired)
        if (tr && tk < 8) trace[tk * 8 + 4] = __builtin_amdgcn_s_memtime();
        __syncthreads();
        if (tr && tk < 8) { trace[tk * 8 + 5] = __builtin_amdgcn_s_memtime(); ++tk; }
    }
}

template <bool POOL, bool TRN = false>
static int launch_conv_rs(hipStream_t st, const ConvArgs& a, int n_cu) {
    if constexpr (!POOL && !TRN) {
        // the tile orientation with fewer tiles (OMNI_RS_TRN=0/1 forces one: A/B hook; results do not depend on it)
        if (conv_rs_transposed(a.H, a.W)) return launch_conv_rs<false, true>(st, a, n_cu);
    }
    auto kfn = conv3x3_c128_rs_kernel<POOL, TRN>;
    OMNI_HIP_TRY(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RS_SMEM));
    const int tiles_x = cdiv(a.W, TRN ? RS_TH : RS_TW), tiles_y = cdiv(a.H, TRN ? RS_TW : RS_TH), n_cg = a.cout / 128;
    // the tiles of an image that run: all of them, or all but the rectangle the caller already holds (ConvArgs::skip_*, in THIS kernel's 6 x 32 tile grid)
    const bool skip = !TRN && a.skip_ty1 > a.skip_ty0 && a.skip_tx1 > a.skip_tx0;
    OMNI_REQUIRE(!skip || (a.skip_ty0 >= 0 && a.skip_ty1 <= tiles_y && a.skip_tx0 >= 0 && a.skip_tx1 <= tiles_x), OMNI_ERR_INVALID, "conv_rs: skip rectangle outside the tile grid");
    RsSkip sk;
    sk.y0 = skip ? a.skip_ty0 : 0; sk.y1 = skip ? a.skip_ty1 : 0; sk.x0 = skip ? a.skip_tx0 : 0; sk.w = skip ? a.skip_tx1 - a.skip_tx0 : 0;
    sk.bw = tiles_x - sk.w;
    sk.act = tiles_x * tiles_y - (sk.y1 - sk.y0) * sk.w;
    sk.n_above = skip ? sk.y0 * tiles_x : sk.act;
    sk.n_upto = sk.n_above + (sk.y1 - sk.y0) * sk.bw;
    OMNI_REQUIRE(sk.act > 0 && sk.bw > 0, OMNI_ERR_INVALID, "conv_rs: the skip rectangle covers whole tile rows");
    sk.xcd = config_process()[CFG_CONV_XCD];
    const int total = a.batch * sk.act;
    int per_cg = n_cu / n_cg;
    if (per_cg < 1) per_cg = 1;
    if (per_cg > total) per_cg = total;
    OMNI_REQUIRE(a.zero_page, OMNI_ERR_INVALID, "conv: ConvArgs.zero_page is not set");
    const char* zero_page = reinterpret_cast<const char*>(a.zero_page);
    static const bool want_trace = config_process()[CFG_RS_TRACE] != 0;
    static unsigned long long* trace_dev = nullptr;
    if (want_trace) {
        if (!trace_dev) OMNI_HIP_TRY(hipMalloc((void**)&trace_dev, 64 * 8));
        OMNI_HIP_TRY(hipMemsetAsync(trace_dev, 0, 64 * 8, st));
    }
    hipLaunchKernelGGL(kfn, dim3(per_cg * n_cg), dim3(256), RS_SMEM, st, reinterpret_cast<const _Float16*>(a.in),
                       reinterpret_cast<_Float16*>(a.out), reinterpret_cast<const _Float16*>(a.w_packed), a.bias, a.H, a.W, a.cout, n_cg,
                       tiles_x, tiles_y, a.batch, a.relu ? 1 : 0, zero_page, want_trace ? trace_dev : nullptr, sk);
    OMNI_LAUNCH_CHECK();
    if (want_trace) {
        unsigned long long h[64];
        OMNI_HIP_TRY(hipMemcpyAsync(h, trace_dev, sizeof(h), hipMemcpyDeviceToHost, st));
        OMNI_HIP_TRY(hipStreamSynchronize(st));
        static int launches = 0;
        if (launches++ / 4 == 5)                                 // the sixth forward pass: conv3b, conv4a, conv4b, heads
            for (int k = 0; k < 8; ++k)
                fprintf(stderr, "rs trace H=%d W=%d cout=%d pool=%d tile %llu (ty %llu tx %llu): issue %llu mfma %llu epilogue %llu wait %llu zerofix+barrier %llu | total %llu\n",
                        a.H, a.W, a.cout, (int)POOL, h[k * 8 + 6], (h[k * 8 + 6] % (tiles_x * tiles_y)) / tiles_x, (h[k * 8 + 6] % (tiles_x * tiles_y)) % tiles_x,
                        h[k * 8 + 1] - h[k * 8], h[k * 8 + 2] - h[k * 8 + 1], h[k * 8 + 3] - h[k * 8 + 2], h[k * 8 + 4] - h[k * 8 + 3],
                        h[k * 8 + 5] - h[k * 8 + 4], h[k * 8 + 5] - h[k * 8]);
    }
    return OMNI_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// v5 (round 6): the register-stationary kernel with NOTHING outside its MFMA stream -- the scheme conv_split.hip's cin = 128 kernel has used
// since round 3.  The v4 kernel above spends 19.9 k cycles per 6 x 32 tile for 13.8 k of MFMA: the 17-instruction DMA burst in front of the
// stream (1.5 k), the epilogue behind it (3.1 k: 96 accumulator values per wave -> bias, ReLU, halfs, swaps, stores) and the closing wait.
// One wave per SIMD: whatever is not issued between two MFMAs idles the matrix cores.  Here
//   * the tile is 3 rows (POOL: 4) x 32 pixels, so that the 48 (POOL: 64 -> 32 after the vertical maximum) raw values of the PREVIOUS tile fit
//     next to this tile's accumulators and the 288 weight registers: its epilogue runs inside this tile's stream, one PART (four values of a
//     row: bias, ReLU, halfs; every second part the v_permlane32_swap and the 16-byte store) per "dense" step (a halo row that feeds three
//     MFMAs), pinned one MFMA / five VALU by sched_group_barrier;
//   * the next tile's LDS-DMA goes out one piece per two-MFMA step;
//   * stores are raw-buffer stores whose offset is out of range for lanes with nothing to store (no exec branch in the stream); issued in the
//     first half of the stream, they have retired when the tile's closing s_waitcnt vmcnt(0) comes.
// Same weights, same LDS image, same order of summation per output pixel (tap column outer, tap row, 16-channel group inner) as v4: bit-identical
// results (tests/test_gpu_superpoint.py::test_f16_persistent_kernels_are_bit_identical_to_generic_kernel).  OMNI_CONV_RS=1 keeps v4.
// ---------------------------------------------------------------------------------------------------------------
#define RS2_RING 3
template <int TH> struct Rs2Cfg {
    static constexpr int ITH = TH + 2, PIX = ITH * RS_ITW, NPIECES = (PIX * 16 + 63) / 64, PPW = (NPIECES + 3) / 4, BUF = NPIECES * 1024, NL = 3 * ITH * 8;
    static constexpr size_t smem() { return 2 * (size_t)BUF + 512; }
};
template <int L, int ITH>
__device__ __forceinline__ void rs2_read(uint32_t row_base /* lds + (n * 256) */, int n, int hh, half8_t& dst) {
    constexpr int kx = L / (ITH * 8), r = (L / 8) % ITH, kg = L % 8;
    constexpr int pc = r * RS_ITW + kx;
    const uint32_t addr = (row_base + pc * 256 + ((((pc + n) & 15) ^ hh) << 4)) ^ (kg << 5);
    asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr));
}
template <int TH> constexpr int rs2_nmfma(int L) { const int r = (L / 8) % (TH + 2); int c = 0; for (int ky = 0; ky < 3; ++ky) if (r - ky >= 0 && r - ky < TH) ++c; return c; }
template <int TH> constexpr int rs2_count_before(int L, int want) { int c = 0; for (int l = 0; l < L; ++l) c += rs2_nmfma<TH>(l) == want ? 1 : 0; return c; }
template <int L, int TH, int NPARTS, int NDMA, typename Epi, typename Dma>
__device__ __forceinline__ void rs2_steps(uint32_t row_base, int n, int hh, const half8_t (&wreg)[72], floatx16 (&acc)[TH], half8_t (&fb)[RS2_RING], Epi&& epi, Dma&& dma) {
    constexpr int ITH = TH + 2, NL = 3 * ITH * 8;
    if constexpr (L < NL) {
        constexpr int kx = L / (ITH * 8), r = (L / 8) % ITH, kg = L % 8;
        constexpr int D = RS2_RING - 1;                                               // fragments read ahead
        if constexpr (L + D < NL) rs2_read<L + D, ITH>(row_base, n, hh, fb[(L + D) % RS2_RING]);
        asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(fb[L % RS2_RING]) : "i"((L + D < NL) ? D : (NL - 1 - L)));
        __builtin_amdgcn_sched_barrier(0);
        constexpr int nm = rs2_nmfma<TH>(L);
        if constexpr (r >= 0 && r < TH) acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[(0 * 3 + kx) * 8 + kg], fb[L % RS2_RING], acc[r], 0, 0, 0);
        if constexpr (r - 1 >= 0 && r - 1 < TH) acc[r - 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[(1 * 3 + kx) * 8 + kg], fb[L % RS2_RING], acc[r - 1], 0, 0, 0);
        if constexpr (r - 2 >= 0 && r - 2 < TH) acc[r - 2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[(2 * 3 + kx) * 8 + kg], fb[L % RS2_RING], acc[r - 2], 0, 0, 0);
        if constexpr (nm == 3) {
            constexpr int d = rs2_count_before<TH>(L, 3);
            if constexpr (d < NPARTS) {
                epi(std::integral_constant<int, d>{});
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
            }
        } else if constexpr (nm == 2) {
            constexpr int d = rs2_count_before<TH>(L, 2);
            if constexpr (d < NDMA) dma(std::integral_constant<int, d>{});
        }
        __builtin_amdgcn_sched_barrier(0);
        rs2_steps<L + 1, TH, NPARTS, NDMA>(row_base, n, hh, wreg, acc, fb, epi, dma);
    }
}

template <bool POOL, bool TRN = false>
__global__ void __launch_bounds__(256, 1)
conv3x3_c128_rs2_kernel(const _Float16* __restrict__ in, _Float16* __restrict__ out, const _Float16* __restrict__ wp,
                        const float* __restrict__ bias, int H, int W, int cout, int n_cg, int tiles_x, int tiles_y, int batch, int relu, RsSkip sk) {
    constexpr int TH = POOL ? 4 : 3;
    using C = Rs2Cfg<TH>;
    constexpr int ITH = C::ITH, PPW = C::PPW, NPIECES = C::NPIECES;
    static_assert(!TRN || !POOL, "transposed tiles: no pooling");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem_raw;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, hh = lane >> 5;
    const int bid = xcd_block_id(sk.xcd);
    const int cg = bid % n_cg, wg = bid / n_cg, nwg = gridDim.x / n_cg;
    const int tiles_per_img = sk.act;
    const int total = batch * tiles_per_img;
    const int g32 = cg * 4 + wave;

    half8_t wreg[72];
    {
        const _Float16* wbase = wp + (int64_t)(g32 >> 1) * 2 * 9 * 4096 + (g32 & 1) * 512 + lane * 8;
#pragma unroll
        for (int s2 = 0; s2 < 72; ++s2) {
            const int tk = s2 >> 3, ch = (s2 >> 2) & 1, kg4 = s2 & 3;
            const int tap = TRN ? (tk % 3) * 3 + tk / 3 : tk;
            wreg[s2] = *reinterpret_cast<const half8_t*>(wbase + ((ch * 9 + tap) * 4 + kg4) * 1024);
        }
    }
    float* const bias_lds = reinterpret_cast<float*>(smem_raw + 2 * C::BUF);
    if (tid < 128) bias_lds[tid] = bias[cg * 128 + tid];

    auto tile_origin = [&](int t, int& b, int& ty0, int& tx0) {
        b = t / tiles_per_img;
        int r = t - b * tiles_per_img, ry, rx;
        if (r < sk.n_above || r >= sk.n_upto) {
            int base = 0;
            if (r >= sk.n_upto) { r -= sk.n_upto; base = sk.y1; }
            ry = r / tiles_x; rx = r - ry * tiles_x; ry += base;
        } else {
            r -= sk.n_above;
            const int q = r / sk.bw, c = r - q * sk.bw;
            ry = sk.y0 + q; rx = c < sk.x0 ? c : c + sk.w;
        }
        ty0 = ry * (TRN ? RS_TW : TH); tx0 = rx * (TRN ? TH : RS_TW);
    };
    uint32_t goff[PPW];                       // byte offset of this lane's chunk of piece j relative to the halo origin
    uint32_t hx[(PPW + 4) / 5];               // ... and the halo column (image x) of its pixel, 6 bits per piece
#pragma unroll
    for (int i = 0; i < (PPW + 4) / 5; ++i) hx[i] = 0;
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        int piece = wave * PPW + j;
        piece = piece < NPIECES ? piece : NPIECES - 1;         // (the last wave repeats its last piece)
        const int idx = piece * 64 + lane;
        const int pix = idx >> 4, phys = idx & 15;
        const int iv = pix / RS_ITW, iu = pix - iv * RS_ITW;
        const int iy = TRN ? iu : iv, ix = TRN ? iv : iu;
        goff[j] = (uint32_t)((iy * W + ix) * 256 + ((phys ^ (pix & 15)) << 4));
        hx[j / 5] |= (uint32_t)(ix & 63) << (6 * (j % 5));
    }
    const uint32_t in_img_bytes = (uint32_t)H * W * 256u;
    struct Src { __amdgpu_buffer_rsrc_t r; uint32_t sorg; int x0; };
    auto source = [&](int t) -> Src {
        int b, ty0, tx0;
        tile_origin(t, b, ty0, tx0);
        Src s;
        s.r = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(in) + (int64_t)b * H * W * 128, 0, in_img_bytes, 0x00020000);
        s.x0 = tx0 - 1;
        s.sorg = (uint32_t)(((ty0 - 1) * W + s.x0) * 256);
        return s;
    };
    auto dma_piece = [&](const Src& s, int which, auto JC) {
        constexpr int j = decltype(JC)::value;
        int piece = wave * PPW + j;
        piece = piece < NPIECES ? piece : NPIECES - 1;
        const uint32_t x = (uint32_t)(s.x0 + (int)((hx[j / 5] >> (6 * (j % 5))) & 63u));
        const uint32_t voff = x < (uint32_t)W ? goff[j] + s.sorg : 0x80000000u;
#if __HIP_DEVICE_COMPILE__
        __builtin_amdgcn_raw_ptr_buffer_load_lds(s.r, (__attribute__((address_space(3))) void*)(smem_raw + which * C::BUF + piece * 1024), 16, voff, 0, 0, 0);
#else
        (void)s; (void)which; (void)piece; (void)voff;
#endif
    };
    int t = wg;
    if (t < total) {
        const Src s0 = source(t);
        spl2_for_each<0, PPW>([&](auto JC) { dma_piece(s0, 0, JC); });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float4 bs[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bs[g] = *reinterpret_cast<const float4*>(bias_lds + wave * 32 + 8 * g + 4 * hh);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int g = 0; g < 4; ++g) asm volatile("" : "+v"(bs[g].x), "+v"(bs[g].y), "+v"(bs[g].z), "+v"(bs[g].w));

    // ---- the pending epilogue: the previous tile's raw values (POOL: the vertical maxima of its row pairs), its image and the lane's offsets ------
    constexpr int NPF = POOL ? TH / 2 : TH;                    // pending fragment rows
    constexpr int NPARTS = NPF * 4;
    float pv[NPF][16];
    uint32_t poff[NPF];
#pragma unroll
    for (int f = 0; f < NPF; ++f) {
        poff[f] = 0x80000000u;
#pragma unroll
        for (int i = 0; i < 16; ++i) pv[f][i] = 0.f;
    }
    const uint32_t out_img_bytes = (uint32_t)((int64_t)(POOL ? (H >> 1) * (W >> 1) : H * W) * cout * 2);
    __amdgpu_buffer_rsrc_t oimg = __builtin_amdgcn_make_buffer_rsrc(out, 0, out_img_bytes, 0x00020000);
    uint32_t dq[2] = {0u, 0u};                                 // the first group's two packed dwords of the pair being built
    // part P = (fragment row P / 4, register group g = P % 4): four values -> bias, ReLU, two packed dwords; g odd: swap with the group before it
    // across the half-waves (the lower one then owns channels [16 gp, +8), the upper one [16 gp + 8, +8)) and one 16-byte store
    auto epi_part = [&](auto PC) {
        constexpr int P = decltype(PC)::value, f = P / 4, g = P % 4;
        float x[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = pv[f][4 * g + e];
            if constexpr (POOL) v = vmax_med3(v, dpp_swap_pairs(v));
            x[e] = v;
        }
        const uint32_t d0 = pack_relu_f16<false>(x[0] + bs[g].x, x[1] + bs[g].y, 1), d1 = pack_relu_f16<false>(x[2] + bs[g].z, x[3] + bs[g].w, 1);      // (ReLU: the launcher insists)
        if constexpr ((g & 1) == 0) { dq[0] = d0; dq[1] = d1; }
        else {
            const auto r0 = __builtin_amdgcn_permlane32_swap(dq[0], d0, false, false);
            const auto r1 = __builtin_amdgcn_permlane32_swap(dq[1], d1, false, false);
            typedef uint32_t u4_t __attribute__((ext_vector_type(4)));
            const u4_t dd = {r0[0], r1[0], r0[1], r1[1]};
            __builtin_amdgcn_raw_buffer_store_b128(dd, oimg, poff[f] + (uint32_t)((16 * (g >> 1) + 8 * hh) * 2), 0, 0);
        }
    };

    int cur = 0;
    for (; t < total; t += nwg, cur ^= 1) {
        const int tn = t + nwg;
        const Src sn = source(tn < total ? tn : t);               // (the last tile loads itself again: no branch in the stream; nobody reads that buffer)
        const uint32_t row_base = lds0 + cur * C::BUF + n * 256;
        floatx16 acc[TH];
#pragma unroll
        for (int f = 0; f < TH; ++f)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[f][i] = 0.f;
        half8_t fb[RS2_RING];
        rs2_read<0, ITH>(row_base, n, hh, fb[0]);
        rs2_read<1, ITH>(row_base, n, hh, fb[1]);
        __builtin_amdgcn_sched_barrier(0);
        rs2_steps<0, TH, NPARTS, PPW>(row_base, n, hh, wreg, acc, fb, epi_part, [&](auto JC) { dma_piece(sn, cur ^ 1, JC); });

        // this tile's raw values and addresses become the pending epilogue
        int b, ty0, tx0;
        tile_origin(t, b, ty0, tx0);
        oimg = __builtin_amdgcn_make_buffer_rsrc(out + (int64_t)b * (out_img_bytes / 2), 0, out_img_bytes, 0x00020000);
        const int ox = tx0 + n;
        if constexpr (POOL) {
#pragma unroll
            for (int f2 = 0; f2 < TH / 2; ++f2) {
                const int oy = ty0 + 2 * f2;
#pragma unroll
                for (int i = 0; i < 16; ++i) pv[f2][i] = vmax_med3(acc[2 * f2][i], acc[2 * f2 + 1][i]);
                poff[f2] = ((oy < H) && (ox < W) && !(n & 1)) ? (uint32_t)((((oy >> 1) * (W >> 1) + (ox >> 1)) * cout + g32 * 32) * 2) : 0x80000000u;
            }
        } else {
#pragma unroll
            for (int f = 0; f < TH; ++f) {
                const int oy = TRN ? ty0 + n : ty0 + f, oxx = TRN ? tx0 + f : ox;
#pragma unroll
                for (int i = 0; i < 16; ++i) pv[f][i] = acc[f][i];
                poff[f] = ((oy < H) && (oxx < W)) ? (uint32_t)(((oy * W + oxx) * cout + g32 * 32) * 2) : 0x80000000u;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // next tile landed (and the previous tile's stores, issued early in the stream, retired)
        __syncthreads();
    }
    spl2_for_each<0, NPARTS>(epi_part);                           // the last tile's
}

template <bool POOL, bool TRN = false>
static int launch_conv_rs2(hipStream_t st, const ConvArgs& a, int n_cu) {
    constexpr int TH = POOL ? 4 : 3;
    if constexpr (!POOL && !TRN) {
        static const int force = config_process()[CFG_RS_TRN];
        const int plain = cdiv(a.W, RS_TW) * cdiv(a.H, TH), trn = cdiv(a.W, TH) * cdiv(a.H, RS_TW);
        if (force == 1 || (force < 0 && trn < plain)) return launch_conv_rs2<false, true>(st, a, n_cu);
    }
    OMNI_REQUIRE(a.relu, OMNI_ERR_INVALID, "conv_rs2: instantiated with the ReLU in its epilogue (every cin = 128 layer of the graph has one)");
    auto kfn = conv3x3_c128_rs2_kernel<POOL, TRN>;
    static DynSmemState smem_state;
    OMNI_HIP_TRY(ensure_dyn_smem(smem_state, (const void*)kfn, Rs2Cfg<TH>::smem()));
    const int tiles_x = cdiv(a.W, TRN ? TH : RS_TW), tiles_y = cdiv(a.H, TRN ? RS_TW : TH), n_cg = a.cout / 128;
    const bool skip = !TRN && a.skip_ty1 > a.skip_ty0 && a.skip_tx1 > a.skip_tx0;
    OMNI_REQUIRE(!skip || (a.skip_ty0 >= 0 && a.skip_ty1 <= tiles_y && a.skip_tx0 >= 0 && a.skip_tx1 <= tiles_x), OMNI_ERR_INVALID, "conv_rs2: skip rectangle outside the tile grid");
    OMNI_REQUIRE((int64_t)a.H * a.W * 256 < (1ll << 31) && (int64_t)a.H * a.W * a.cout * 2 < (1ll << 31), OMNI_ERR_INVALID, "conv_rs2: image too large for 32-bit offsets");
    RsSkip sk;
    sk.y0 = skip ? a.skip_ty0 : 0; sk.y1 = skip ? a.skip_ty1 : 0; sk.x0 = skip ? a.skip_tx0 : 0; sk.w = skip ? a.skip_tx1 - a.skip_tx0 : 0;
    sk.bw = tiles_x - sk.w;
    sk.act = tiles_x * tiles_y - (sk.y1 - sk.y0) * sk.w;
    sk.n_above = skip ? sk.y0 * tiles_x : sk.act;
    sk.n_upto = sk.n_above + (sk.y1 - sk.y0) * sk.bw;
    OMNI_REQUIRE(sk.act > 0 && sk.bw > 0, OMNI_ERR_INVALID, "conv_rs2: the skip rectangle covers whole tile rows");
    sk.xcd = config_process()[CFG_CONV_XCD];
    const int total = a.batch * sk.act;
    int per_cg = n_cu / n_cg;
    if (per_cg < 1) per_cg = 1;
    if (per_cg > total) per_cg = total;
    hipLaunchKernelGGL(kfn, dim3(per_cg * n_cg), dim3(256), Rs2Cfg<TH>::smem(), st, reinterpret_cast<const _Float16*>(a.in), reinterpret_cast<_Float16*>(a.out),
                       reinterpret_cast<const _Float16*>(a.w_packed), a.bias, a.H, a.W, a.cout, n_cg, tiles_x, tiles_y, a.batch, a.relu ? 1 : 0, sk);
    OMNI_LAUNCH_CHECK();
    return OMNI_OK;
}
// rows of the pooled cin = 128 fp16 layer's tile (the grid ConvArgs::skip_* is given in)
int conv_rs_pool_tile_rows() { return RS_TH; }                  // (the pooled layer runs on v4 in every configuration)

// the tile orientation launch_conv_rs picks for a non-pooled H x W layer (it fixes the order the taps are summed in)
bool conv_rs_transposed(int H, int W) {
    static const int force = config_process()[CFG_RS_TRN];
    const int plain = cdiv(W, RS_TW) * cdiv(H, RS_TH), trn = cdiv(W, RS_TH) * cdiv(H, RS_TW);
    return force == 1 || (force < 0 && trn < plain);
}

// ---------------------------------------------------------------------------------------------------------------
// 3x3 conv, 128 input channels, ONLY at the coarse cells around the key points (convDa, superpoint.ipynb:183: its output is read by
// computeDescriptors at the four cells around each key point and nowhere else -- <= 800 of 4 500 cells).  Weights as in the register-stationary
// kernel above (wave = 32 output channels, all K = 9 x 128 in 288 registers, the same packed array), but a "pixel tile" is the 32 corner
// cells of 8 key points: their 3x3 x 128-channel neighbourhoods are gathered into LDS (thread = (cell, eighth): 18 16-byte pieces, fetched
// into registers one tile ahead), one accumulator fragment per wave, 72 MFMAs per tile in EXACTLY the order the dense kernel sums the taps
// in for this layer shape (TRN), bias + ReLU + fp16 as store_frag16: the values are bit-identical to the dense layer's at those cells.
// out: compact [image][key point][corner][256] fp16 (+ cout_off); cells outside the map / beyond n_kps are not written.
// ---------------------------------------------------------------------------------------------------------------
// k-step S (dense order) of the sparse kernel: B fragment of step S + 2 issued, step S waited for, one MFMA (the rs_steps pattern: a single
// wave per SIMD has nobody to hide an LDS round trip behind, and left to itself hipcc waited lgkmcnt(0) in front of every MFMA)
template <int S, bool TRN>
__device__ __forceinline__ void spc_steps(uint32_t base, const half8_t (&wreg)[72], floatx16& acc, half8_t (&fb)[3]) {
    if constexpr (S < 72) {
        constexpr int o = S / 8, kg = S % 8, tap = TRN ? o : (o % 3) * 3 + o / 3;
        if constexpr (S + 2 < 72) {
            constexpr int o2 = (S + 2) / 8, kg2 = (S + 2) % 8, tap2 = TRN ? o2 : (o2 % 3) * 3 + o2 / 3;
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[(S + 2) % 3]) : "v"(base), "i"(tap2 * 256 + kg2 * 32));
        }
        asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(fb[S % 3]) : "i"((S + 2 < 72) ? 2 : (S + 1 < 72 ? 1 : 0)));
        __builtin_amdgcn_sched_barrier(0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[tap * 8 + kg], fb[S % 3], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        spc_steps<S + 1, TRN>(base, wreg, acc, fb);
    }
}

#define SPC_CELL_BYTES 2320                       // 9 taps x 256 B + 16: an odd number of 16-byte slots, conflict-free B-fragment reads
#define SPC_BUF_BYTES (32 * SPC_CELL_BYTES)       // 74 240
#define SPC_SMEM (2 * SPC_BUF_BYTES)
template <bool TRN>
__global__ void __launch_bounds__(256, 1)
conv3x3_c128_sparse_kernel(const _Float16* __restrict__ in /*[B][Hc][Wc][128]*/, const _Float16* __restrict__ wp, const float* __restrict__ bias,
                           int Hc, int Wc, int g32_first, int W, int H, int max_num, const float* __restrict__ kps_xy,
                           const int* __restrict__ n_kps, _Float16* __restrict__ out, int out_cstride, int tiles_per_img, int n_tiles,
                           const char* __restrict__ zero_page /* omni_ctx::zero_page: what out-of-map taps and missing cells read */) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem_raw;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, hh = lane >> 5;
    const int half = blockIdx.x & 1, wg = blockIdx.x >> 1, nwg = gridDim.x >> 1;      // 128 output channels per workgroup, two halves
    const int g32 = g32_first + half * 4 + wave;
    half8_t wreg[72];                              // wreg[tap * 8 + kg], tap = ky * 3 + kx of the IMAGE
    {
        const _Float16* wbase = wp + (int64_t)(g32 >> 1) * 2 * 9 * 4096 + (g32 & 1) * 512 + lane * 8;
#pragma unroll
        for (int s2 = 0; s2 < 72; ++s2) {
            const int tap = s2 >> 3, ch = (s2 >> 2) & 1, kg4 = s2 & 3;
            wreg[s2] = *reinterpret_cast<const half8_t*>(wbase + ((ch * 9 + tap) * 4 + kg4) * 1024);
        }
    }
    float4 bs[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bs[g] = *reinterpret_cast<const float4*>(bias + g32 * 32 + 8 * g + 4 * hh);
    const float fW = (float)W, fH = (float)H, fWc = (float)Wc, fHc = (float)Hc;
    // corner cell `slot` of tile t -> (image, key point, cell x, cell y); false: no such key point (sp_sample_kernel's arithmetic)
    auto cell_of = [&](int t, int slot, int& b, int& kp, int& cx, int& cy) __attribute__((always_inline)) -> bool {
        b = t / tiles_per_img;
        kp = (t - b * tiles_per_img) * CSP_KP + (slot >> 2);
        cx = cy = -4;
        if (kp >= n_kps[b]) return false;
        const float kx = kps_xy[((int64_t)b * max_num + kp) * 2 + 0], ky = kps_xy[((int64_t)b * max_num + kp) * 2 + 1];
        const float gx = __fsub_rn(__fdiv_rn(__fmul_rn(2.0f, kx), fW), 1.0f);
        const float gy = __fsub_rn(__fdiv_rn(__fmul_rn(2.0f, ky), fH), 1.0f);
        const float ix = __fdiv_rn(__fsub_rn(__fmul_rn(__fadd_rn(gx, 1.0f), fWc), 1.0f), 2.0f);
        const float iy = __fdiv_rn(__fsub_rn(__fmul_rn(__fadd_rn(gy, 1.0f), fHc), 1.0f), 2.0f);
        cx = (int)floorf(ix) + (slot & 1); cy = (int)floorf(iy) + ((slot >> 1) & 1);
        return cx >= 0 && cx < Wc && cy >= 0 && cy < Hc;
    };
    // gather: thread = (cell tid >> 3, eighth tid & 7): pieces j * 8 + eighth, j < 18, of the cell's 144 16-byte pieces (tap = j >> 1).  The 18
    // staging registers are named scalars and the nine taps are spelled out: as an array indexed inside unrolled loops hipcc kept them in scratch
    // memory (a store per load, the gather serialised: 214 us per launch instead of 20)
    const int gcell = tid >> 3, gpart = tid & 7;
    uint4 sa0, sb0, sa1, sb1, sa2, sb2, sa3, sb3, sa4, sb4, sa5, sb5, sa6, sb6, sa7, sb7, sa8, sb8;
    int fb, fkp, fcx, fcy; bool fok; const char* fimg;
#define SPC_FETCH_TAP(T)                                                                                                     \
    {                                                                                                                        \
        const int y = fcy + (T) / 3 - 1, x = fcx + (T) % 3 - 1;                                                              \
        const bool inside = fok && y >= 0 && y < Hc && x >= 0 && x < Wc;                                                     \
        const char* src = inside ? fimg + ((int64_t)y * Wc + x) * 256 + gpart * 16 : zero_page + (tid & 255) * 32;           \
        sa##T = *reinterpret_cast<const uint4*>(src);                                                                        \
        sb##T = *reinterpret_cast<const uint4*>(inside ? src + 128 : src + 16);                                              \
    }
#define SPC_FETCH(TILE)                                                                                                      \
    {                                                                                                                        \
        fok = cell_of((TILE), gcell, fb, fkp, fcx, fcy);                                                                     \
        fimg = reinterpret_cast<const char*>(in + (int64_t)fb * Hc * Wc * 128);                                              \
        SPC_FETCH_TAP(0) SPC_FETCH_TAP(1) SPC_FETCH_TAP(2) SPC_FETCH_TAP(3) SPC_FETCH_TAP(4) SPC_FETCH_TAP(5) SPC_FETCH_TAP(6) SPC_FETCH_TAP(7) SPC_FETCH_TAP(8) \
    }
#define SPC_PARK_TAP(T) *reinterpret_cast<uint4*>(pdst + (T) * 256) = sa##T; *reinterpret_cast<uint4*>(pdst + (T) * 256 + 128) = sb##T;
#define SPC_PARK(WHICH)                                                                                                      \
    {                                                                                                                        \
        char* pdst = smem_raw + (WHICH) * SPC_BUF_BYTES + gcell * SPC_CELL_BYTES + gpart * 16;                               \
        SPC_PARK_TAP(0) SPC_PARK_TAP(1) SPC_PARK_TAP(2) SPC_PARK_TAP(3) SPC_PARK_TAP(4) SPC_PARK_TAP(5) SPC_PARK_TAP(6) SPC_PARK_TAP(7) SPC_PARK_TAP(8) \
    }
    int t = wg;
    if (t < n_tiles) { SPC_FETCH(t) SPC_PARK(0) }
    __syncthreads();
    int cur = 0;
    for (; t < n_tiles; t += nwg, cur ^= 1) {
        const int tn = t + nwg;
        if (tn < n_tiles) SPC_FETCH(tn)                                // in flight behind the MFMA loop
        floatx16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        {   // the dense kernel's order: plain tiles sum kx outer / ky inner, transposed tiles ky outer / kx inner (spc_steps)
            const uint32_t base = lds0 + cur * SPC_BUF_BYTES + n * SPC_CELL_BYTES + hh * 16;
            half8_t fb[3];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // nothing of ours in flight on the LDS counter but the reads below
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[0]) : "v"(base), "i"(0));
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[1]) : "v"(base), "i"(32));
            __builtin_amdgcn_sched_barrier(0);
            spc_steps<0, TRN>(base, wreg, acc, fb);
        }
        {   // bias + ReLU + fp16 (store_frag16<false>): lane = cell n, channels 32 g32 + 8 g + 4 hh + (0..3)
            int b, kp, cx, cy;
            const bool ok = cell_of(t, n, b, kp, cx, cy);
            if (ok) {
                _Float16* op = out + (((int64_t)b * max_num + kp) * 4 + (n & 3)) * out_cstride + (g32 - g32_first) * 32 + 4 * hh;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const uint32_t d0 = pack_relu_f16<false>(acc[4 * g + 0] + bs[g].x, acc[4 * g + 1] + bs[g].y, 1);
                    const uint32_t d1 = pack_relu_f16<false>(acc[4 * g + 2] + bs[g].z, acc[4 * g + 3] + bs[g].w, 1);
                    *reinterpret_cast<uint2*>(op + 8 * g) = make_uint2(d0, d1);
                }
            }
        }
        if (tn < n_tiles) SPC_PARK(cur ^ 1)
        __syncthreads();
    }
#undef SPC_FETCH_TAP
#undef SPC_FETCH
#undef SPC_PARK_TAP
#undef SPC_PARK
}

int conv_c128_sparse(hipStream_t st, const omni_ctx* ctx, const void* in_f16, const void* w_packed, const float* bias, int Hc, int Wc, int g32_first,
                     int W, int H, int max_num, const float* kps_xy, const int* n_kps, void* out_f16, int out_cstride, int batch) {
    const int tiles_per_img = cdiv(max_num, CSP_KP), n_tiles = tiles_per_img * batch;
    const int cus = ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256;
    int pairs = cus / 2;                                                 // two workgroups (channel halves) per tile stream
    if (pairs > n_tiles) pairs = n_tiles;
    if (pairs < 1) pairs = 1;
    const bool trn = conv_rs_transposed(Hc, Wc);
    OMNI_REQUIRE(ctx->zero_page, OMNI_ERR_INVALID, "conv_c128_sparse: the context has no zero block");
    auto launch = [&](auto kfn) -> int {
        static DynSmemState attr;
        OMNI_HIP_TRY(ensure_dyn_smem(attr, (const void*)kfn, SPC_SMEM));
        hipLaunchKernelGGL(kfn, dim3(2 * pairs), dim3(256), SPC_SMEM, st, (const _Float16*)in_f16, (const _Float16*)w_packed, bias, Hc, Wc, g32_first, W, H,
                           max_num, kps_xy, n_kps, (_Float16*)out_f16, out_cstride, tiles_per_img, n_tiles, (const char*)ctx->zero_page);
        OMNI_LAUNCH_CHECK();
        return OMNI_OK;
    };
    return trn ? launch(conv3x3_c128_sparse_kernel<true>) : launch(conv3x3_c128_sparse_kernel<false>);
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

// conv1a (from the u8 image) + conv1b + ReLU + 2x2 max-pool in one launch (fp16 path); a.in is unused
int conv1ab_fused(hipStream_t st, const ConvArgs& a, const uint8_t* gray, int gstride, int fisheye_mask, const _Float16* w1a_frag,
                  const float* bias1a, const uint32_t* lut_hl) {
    OMNI_REQUIRE(a.cin == 64 && a.cout % 64 == 0 && a.ksize == 3 && a.pool && a.n_cu > 0, OMNI_ERR_INVALID, "conv1ab_fused: bad layer shape");
    OMNI_REQUIRE(gstride % 4 == 0 && ((uintptr_t)gray & 3) == 0 && (int64_t)a.batch * a.H * gstride < (1ll << 31), OMNI_ERR_INVALID,
                 "conv1ab_fused: image rows must be 4-byte aligned (stride %d)", gstride);
    Fuse1aArgs fz;
    fz.gray = gray; fz.gstride = gstride;
    omni_fisheye_mask_rows(a.H, fisheye_mask, &fz.mask_r0, &fz.mask_r1);   // cv::Rect(0, rows*3/4, cols, rows/4)
    fz.w1a_frag = w1a_frag; fz.bias1a = bias1a; fz.lut_hl = lut_hl; fz.trace = nullptr;
    static const bool want_trace = config_process()[CFG_PP_TRACE] != 0;
    static unsigned long long* trace_dev = nullptr;
    if (want_trace) {
        if (!trace_dev) OMNI_HIP_TRY(hipMalloc((void**)&trace_dev, 64 * 8));
        OMNI_HIP_TRY(hipMemsetAsync(trace_dev, 0, 64 * 8, st));
        fz.trace = trace_dev;
    }
    static const int dbg = config_process()[CFG_PP_DBG];   // timing ablations only
    const int rc = launch_conv_pp_abl<true, 0, true>(st, a, a.n_cu, dbg, fz);
    if (want_trace && rc == OMNI_OK) {
        unsigned long long h[64];
        OMNI_HIP_TRY(hipMemcpyAsync(h, trace_dev, sizeof(h), hipMemcpyDeviceToHost, st));
        OMNI_HIP_TRY(hipStreamSynchronize(st));
        static int printed = 0;
        if (printed++ == 20)
            for (int p = 2; p < 6; ++p)
                fprintf(stderr, "pp trace phase %d: service start %llu issue +%llu epilogue +%llu build +%llu (operands ready +%llu, first fragment +%llu) | mfma loop %llu cycles (start +%lld vs service start)\n", p,
                        h[p * 8], h[p * 8 + 1] - h[p * 8], h[p * 8 + 2] - h[p * 8 + 1], h[p * 8 + 3] - h[p * 8 + 2], h[p * 8 + 6] - h[p * 8 + 2], h[p * 8 + 7] - h[p * 8 + 6],
                        h[p * 8 + 5] - h[p * 8 + 4], (long long)(h[p * 8 + 4] - h[p * 8]));
    }
    return rc;
}

static inline uint16_t f2h_bits(float v) { const __half h = __float2half_rn(v); uint16_t u; memcpy(&u, &h, 2); return u; }
static inline float h2f(uint16_t u) { __half h; memcpy(&h, &u, 2); return __half2float(h); }
// w [64][9] fp32, bias [64] -> A fragments [j][m][lane = hh*32 + i][e] of the split weights (wh = half(w), wl = half(w - wh)); the K
// slot (j, hh, e) pairs with the B operand built in the fused kernel:
//   hh = 0:  j = 0: wh0 wh0 wh1 wh1 wh2 wh2 wh3 wh3      j = 1: wh4 wh4 wl0 wl1 wl2 wl3 wl4 0
//   hh = 1:  j = 0: wh5 wh5 wh6 wh6 wh7 wh7 wh8 wh8      j = 1: wl5 wl6 wl7 wl8 bias_hi bias_lo 0 0
void conv1a_pack_split_weights(const float* w, const float* bias, uint16_t* frag /*2*2*64*8*/) {
    for (int j = 0; j < 2; ++j)
        for (int m = 0; m < 2; ++m)
            for (int l = 0; l < 64; ++l)
                for (int e = 0; e < 8; ++e) {
                    const int co = m * 32 + (l & 31), hh = l >> 5;
                    auto hi = [&](int t) { return f2h_bits(w[co * 9 + t]); };
                    auto lo = [&](int t) { return f2h_bits(w[co * 9 + t] - h2f(f2h_bits(w[co * 9 + t]))); };
                    uint16_t v = 0;
                    if (j == 0) v = hi((hh ? 5 : 0) + e / 2);
                    else if (hh == 0) v = e < 2 ? hi(4) : (e < 7 ? lo(e - 2) : 0);
                    else v = e < 4 ? lo(5 + e) : (e == 4 ? f2h_bits(bias[co]) : (e == 5 ? f2h_bits(bias[co] - h2f(f2h_bits(bias[co]))) : 0));
                    frag[((j * 2 + m) * 64 + l) * 8 + e] = v;
                }
}
// The table-free form (Fuse1aArgs::lut_hl = nullptr): the B operand of tap t is the pair (P_t, P_t), P_t = half(4 + p_t / 256) = 0x4400 | p_t exactly, against
// (Wh_t, Wl_t) = the split of W_t = fl32(w_t * 256 / 255); the bias slot (1, 1) carries the split of bias - 4 sum_t (Wh_t + Wl_t) (in double):
//   sum_t W_t (4 + p_t / 256) + bias - 4 sum_t W_t = sum_t w_t p_t / 255 + bias     (W_t to 22 bits; the products are exact in fp32)
//   hh = 0:  j = 0: Wh0 Wl0 Wh1 Wl1 Wh2 Wl2 Wh3 Wl3      j = 1: Wh4 Wl4 0 0 0 0 0 0
//   hh = 1:  j = 0: Wh5 Wl5 Wh6 Wl6 Wh7 Wl7 Wh8 Wl8      j = 1: bias_hi bias_lo 0 0 0 0 0 0
void conv1a_pack_u8_weights(const float* w, const float* bias, uint16_t* frag /*2*2*64*8*/) {
    for (int m = 0; m < 2; ++m)
        for (int l = 0; l < 64; ++l) {
            const int co = m * 32 + (l & 31), hh = l >> 5;
            uint16_t Wh[9], Wl[9];
            double sum = 0.0;
            for (int t = 0; t < 9; ++t) {
                const float W = (float)((double)w[co * 9 + t] * 256.0 / 255.0);
                Wh[t] = f2h_bits(W); Wl[t] = f2h_bits(W - h2f(Wh[t]));
                sum += (double)h2f(Wh[t]) + (double)h2f(Wl[t]);
            }
            const float b = (float)((double)bias[co] - 4.0 * sum);
            const uint16_t bh = f2h_bits(b), bl = f2h_bits(b - h2f(bh));
            for (int j = 0; j < 2; ++j)
                for (int e = 0; e < 8; ++e) {
                    uint16_t v = 0;
                    if (j == 0) { const int t = (hh ? 5 : 0) + e / 2; v = (e & 1) ? Wl[t] : Wh[t]; }
                    else if (hh == 0) v = e == 0 ? Wh[4] : (e == 1 ? Wl[4] : 0);
                    else v = e == 0 ? bh : (e == 1 ? bl : 0);
                    frag[((j * 2 + m) * 64 + l) * 8 + e] = v;
                }
        }
}
void conv1a_make_split_lut(uint32_t* lut /*256*/) {
    for (int i = 0; i < 256; ++i) {
        const volatile float alpha = (float)(1.0 / 255.0);
        const float x = (float)i * alpha;                                       // cv::Mat::convertTo(CV_32F, 1/255.0): OpenCV 3.4 scales 8-bit sources in float
        const uint16_t hi = f2h_bits(x);
        lut[i] = (uint32_t)hi | ((uint32_t)f2h_bits(x - h2f(hi)) << 16);
    }
}

int conv_mfma(hipStream_t st, int precision, const ConvArgs& a) {
    OMNI_REQUIRE(a.cin % 64 == 0 && a.cout % 64 == 0, OMNI_ERR_INVALID, "conv_mfma: cin=%d cout=%d must be multiples of 64", a.cin, a.cout);
    OMNI_REQUIRE(a.ksize == 1 || a.ksize == 3, OMNI_ERR_INVALID, "conv_mfma: ksize=%d", a.ksize);
    OMNI_REQUIRE(!a.pool || (a.H % 2 == 0 && a.W % 2 == 0), OMNI_ERR_INVALID, "pooling needs even H, W");
    OMNI_REQUIRE(!(a.pool && a.ksize == 1), OMNI_ERR_INVALID, "1x1 + pool not instantiated");
    if (precision == OMNI_PREC_F16 && a.ksize == 3 && a.cin == 64 && !a.out_f32 && (a.in_cstride == 0 || a.in_cstride == 64) && a.n_cu > 0 && a.zero_page &&
        a.variant != 1) {
#ifdef OMNI_TEST_VARIANTS
        if (a.variant == 2) return a.pool ? launch_conv_c64<true>(st, a, a.n_cu) : launch_conv_c64<false>(st, a, a.n_cu);
#endif
        return a.pool ? launch_conv_pp<true>(st, a, a.n_cu) : launch_conv_pp<false>(st, a, a.n_cu);
    }
    static const bool no_rs = config_process()[CFG_CONV_RS] == 0;     // A/B hook
    if (precision == OMNI_PREC_F16 && a.ksize == 3 && a.cin == 128 && a.cout % 128 == 0 && !a.out_f32 && (a.in_cstride == 0 || a.in_cstride == 128) &&
        a.n_cu > 0 && a.zero_page && a.variant == 0 && !no_rs && (!a.pool || a.H % 2 == 0))
    {
        static const int rs_ver = config_process()[CFG_CONV_RS];
        // v5 for the unpooled layers (conv4a, conv4b, convPa|convDa); the pooled conv3b stays on v4: its v5 form (4-row tiles) measured the same time and sits at
        // the register limit (13 registers of scratch)
        if (rs_ver >= 2 && a.relu && !a.pool) return launch_conv_rs2<false>(st, a, a.n_cu);
        return a.pool ? launch_conv_rs<true>(st, a, a.n_cu) : launch_conv_rs<false>(st, a, a.n_cu);
    }
    if (precision == OMNI_PREC_F16) {
#ifdef OMNI_TEST_VARIANTS              // the generic kernel on the fp16 3x3 layers (OMNI_CONV_V1=1): the other bit-identity reference of the test build
        if (a.ksize == 3) return a.pool ? launch_conv<_Float16, 3, true>(st, a) : launch_conv<_Float16, 3, false>(st, a);
#else
        OMNI_REQUIRE(a.ksize != 3, OMNI_ERR_INVALID, "conv_mfma: no production fp16 kernel for a 3x3 layer with cin=%d cout=%d in_cstride=%d (the generic one is only built into the test library)",
                     a.cin, a.cout, a.in_cstride);
#endif
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
    int r0, r1;
    omni_fisheye_mask_rows(H, fisheye_mask, &r0, &r1);   // cv::Rect(0, rows*3/4, cols, rows/4)
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
#define DET_CPW 4                                            // cells per wave per pass: each weight read feeds 4 FMAs
template <typename T>
__global__ void __launch_bounds__(DET_THREADS)
detector_head_kernel(const T* __restrict__ in, int in_stride, int in_off, int n_cells, int Hc, int Wc,
                     const float* __restrict__ wT, const float* __restrict__ bias, float* __restrict__ semi) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* wTs = reinterpret_cast<float*>(smem_raw);      // [256][65]
    float* xs = wTs + 256 * 65;                            // [DET_WAVES][DET_CPW][256]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 256 * 65; i += DET_THREADS) wTs[i] = wT[i];
    const float my_bias = bias[lane];
    const float dust_bias = bias[64];
    __syncthreads();
    float* x = xs + wave * DET_CPW * 256;
    for (int base = blockIdx.x * DET_WAVES * DET_CPW; base < n_cells; base += gridDim.x * DET_WAVES * DET_CPW) {
        const int cell0 = base + wave * DET_CPW;
#pragma unroll
        for (int c = 0; c < DET_CPW; ++c) {
            const int cell = cell0 + c;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (cell < n_cells) {
                const T* ip = in + (int64_t)cell * in_stride + in_off + lane * 4;
                if constexpr (sizeof(T) == 4) {
                    v = *reinterpret_cast<const float4*>(ip);
                } else {
                    const half4_t h = *reinterpret_cast<const half4_t*>(ip);
                    v = make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
                }
            }
            *reinterpret_cast<float4*>(x + c * 256 + lane * 4) = v;
        }
        __syncthreads();
        float acc[DET_CPW], dpart[DET_CPW];
#pragma unroll
        for (int c = 0; c < DET_CPW; ++c) { acc[c] = my_bias; dpart[c] = 0.f; }
        for (int k = 0; k < 256; k += 4) {
            const float w0 = wTs[(k + 0) * 65 + lane], w1 = wTs[(k + 1) * 65 + lane];
            const float w2 = wTs[(k + 2) * 65 + lane], w3 = wTs[(k + 3) * 65 + lane];
#pragma unroll
            for (int c = 0; c < DET_CPW; ++c) {
                const float4 xv = *reinterpret_cast<const float4*>(x + c * 256 + k);
                acc[c] = fmaf(xv.x, w0, acc[c]);
                acc[c] = fmaf(xv.y, w1, acc[c]);
                acc[c] = fmaf(xv.z, w2, acc[c]);
                acc[c] = fmaf(xv.w, w3, acc[c]);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float wd = wTs[(lane * 4 + j) * 65 + 64];
#pragma unroll
            for (int c = 0; c < DET_CPW; ++c) dpart[c] = fmaf(x[c * 256 + lane * 4 + j], wd, dpart[c]);
        }
#pragma unroll
        for (int c = 0; c < DET_CPW; ++c) {
            const int cell = cell0 + c;
            const float dust = wave_sum_f(dpart[c]) + dust_bias;
            const float mx = fmaxf(wave_max_f(acc[c]), dust);
            const float e = expf(acc[c] - mx);
            const float ed = expf(dust - mx);
            const float sum = wave_sum_f(e) + ed;
            const float p = e / sum;
            if (cell < n_cells) {
                const int wx = cell % Wc;
                const int hy = (cell / Wc) % Hc;
                const int b = cell / (Wc * Hc);
                semi[((int64_t)b * Hc * 8 + hy * 8 + (lane >> 3)) * (Wc * 8) + wx * 8 + (lane & 7)] = p;
            }
        }
        __syncthreads();
    }
}

int detector_head(hipStream_t st, int precision, const void* in, int in_stride, int in_off, int batch, int Hc, int Wc,
                  const float* wT, const float* bias, float* semi) {
    const int n_cells = batch * Hc * Wc;
    const size_t smem = (size_t)(256 * 65 + DET_WAVES * DET_CPW * 256) * 4;
    int grid = cdiv(n_cells, DET_WAVES * DET_CPW);
    if (grid > 256) grid = 256;                            // one 98.5 KB workgroup per CU: weights staged once per workgroup
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

// ---------------------------------------------------------------------------------------------------------------
// Detector head tail on the matrix cores: the 256 -> 64 part of convPb as v_mfma_f32_32x32x2_f32 (exact f32, an fmaf chain),
// M = output channels (A = weights from LDS), N = 32 coarse cells per wave (B = the cell's 256 activations straight from
// HBM, converted to f32), K permuted so that half-wave kk owns input channels [128 kk, 128 kk + 128); the dustbin logit
// (channel 64) is a VALU dot product over the same registers.  A lane then holds 32 of its cell's 64 logits (the other
// half-wave the other 32): softmax needs one cross-half exchange, and the 8x8 depth-to-space turns each accumulator quad
// into one 16-byte store (channel c = 8 ry + rx: 4 consecutive rx).
// ---------------------------------------------------------------------------------------------------------------
#define DETM_THREADS 256
// getKeyPoints' threshold (superpoint_tensorrt.cpp:167-173: mask = prob > thres) inside the head's epilogue, where the probabilities of a cell's 64
// pixels sit in registers: a lane holds 32 of them -- l0 / l1 = rows g / 4 + g of the cell, columns 4 hh + e -- and stores the 32 comparisons as one
// word of the image's bitmap (bit i = row i >> 2, column i & 3 of its half cell).  The values compared are the ones stored into the heat map, so the
// bitmap equals thresholding that map.  (A first version appended to the candidate lists right here, one atomic per wave and image: the wave waited
// ~4 us per fragment for the atomic's return -- the head 71 -> 89 us; the bitmap costs nothing measurable and sp_mask_kernel compacts it.)
__device__ __forceinline__ void det_emit_candidates(const float (&p0)[16], const float (&p1)[16], bool valid, int cell, int hh, const DetCand& dc) {
    if (!valid) return;
    uint32_t cm = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) { cm |= (p0[r] > dc.thres ? 1u : 0u) << r; cm |= (p1[r] > dc.thres ? 1u : 0u) << (16 + r); }
    dc.bits[(int64_t)cell * 2 + hh] = cm;
}

template <typename T>
__global__ void __launch_bounds__(DETM_THREADS)
detector_head_mfma_kernel(const T* __restrict__ in, int in_stride, int in_off, int n_cells, int Hc, int Wc,
                          const float* __restrict__ wA /*[2][32][2][32][4] fragment order*/, const float* __restrict__ wdust /*[256]*/,
                          const float* __restrict__ bias /*[65]*/, float* __restrict__ semi, DetCand dc) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* wl = reinterpret_cast<float*>(smem_raw);           // [2][32][2][32][4] = 16384 floats
    float* wd = wl + 16384;                                    // [256]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, hh = lane >> 5;
    for (int i = tid; i < 16384 / 4; i += DETM_THREADS) reinterpret_cast<float4*>(wl)[i] = reinterpret_cast<const float4*>(wA)[i];
    wd[tid] = wdust[tid];
    float* bl = wd + 256;                                      // [65] bias (LDS: 32 registers otherwise)
    if (tid < 65) bl[tid] = bias[tid];
    __syncthreads();
    const float dust_bias = bl[64];
    const int n_frag = (n_cells + 31) >> 5;
    constexpr int EPV = 16 / sizeof(T);                       // elements per 16-byte load: 8 halfs or 4 floats
    constexpr int XPC = 32 / EPV;                             // 16-byte loads per 32-channel chunk
    for (int f = blockIdx.x * 4 + wave; f < n_frag; f += gridDim.x * 4) {
        const int cell = f * 32 + n;
        const bool valid = cell < n_cells;
        const T* ip = in + (int64_t)(valid ? cell : n_cells - 1) * in_stride + in_off + hh * 128;
        floatx16 acc0, acc1;
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
        float dust = 0.f;
        uint4 xr[XPC], xn[XPC];
#pragma unroll
        for (int i = 0; i < XPC; ++i) xr[i] = *reinterpret_cast<const uint4*>(ip + i * EPV);
#pragma unroll 1
        for (int ch = 0; ch < 4; ++ch) {                      // 4 chunks of 32 input channels per half-wave; next chunk prefetched
            if (ch + 1 < 4) {
#pragma unroll
                for (int i = 0; i < XPC; ++i) xn[i] = *reinterpret_cast<const uint4*>(ip + (ch + 1) * 32 + i * EPV);
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int t4 = ch * 8 + t;
                float x4[4];
                if constexpr (sizeof(T) == 2) {
                    const uint4 u = xr[t >> 1];
                    const uint32_t lo = (t & 1) ? u.z : u.x, hi = (t & 1) ? u.w : u.y;
                    const half2_t a = __builtin_bit_cast(half2_t, lo), b = __builtin_bit_cast(half2_t, hi);
                    x4[0] = (float)a[0]; x4[1] = (float)a[1]; x4[2] = (float)b[0]; x4[3] = (float)b[1];
                } else {
                    const uint4 u = xr[t];
                    x4[0] = __uint_as_float(u.x); x4[1] = __uint_as_float(u.y); x4[2] = __uint_as_float(u.z); x4[3] = __uint_as_float(u.w);
                }
                const floatx4 a0 = *reinterpret_cast<const floatx4*>(wl + (((0 * 32 + t4) * 2 + hh) * 32 + n) * 4);
                const floatx4 a1 = *reinterpret_cast<const floatx4*>(wl + (((1 * 32 + t4) * 2 + hh) * 32 + n) * 4);
                const floatx4 d4 = *reinterpret_cast<const floatx4*>(wd + hh * 128 + t4 * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], x4[e], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], x4[e], acc1, 0, 0, 0);
                    dust = fmaf(x4[e], d4[e], dust);
                }
            }
#pragma unroll
            for (int i = 0; i < XPC; ++i) xr[i] = xn[i];
        }
        dust = dust + __shfl_xor(dust, 32, 64) + dust_bias;
        float l0[16], l1[16], mx = dust;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = (r & 3) + 8 * (r >> 2) + 4 * hh;
            l0[r] = acc0[r] + bl[c]; l1[r] = acc1[r] + bl[32 + c];
            mx = fmaxf(mx, fmaxf(l0[r], l1[r]));
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { l0[r] = expf(l0[r] - mx); l1[r] = expf(l1[r] - mx); sum += l0[r] + l1[r]; }
        sum = sum + __shfl_xor(sum, 32, 64) + expf(dust - mx);
#pragma unroll
        for (int r = 0; r < 16; ++r) { l0[r] = l0[r] / sum; l1[r] = l1[r] / sum; }
        const int wx = cell % Wc;
        const int hy = (cell / Wc) % Hc;
        const int b = cell / (Wc * Hc);
        if (valid) {
            float* o = semi + ((int64_t)b * Hc * 8 + hy * 8) * (Wc * 8) + wx * 8 + 4 * hh;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                // channel c = 32 m + 8 g + 4 hh + (r & 3)  ->  row ry = 4 m + g, columns rx = 4 hh + (r & 3)
                *reinterpret_cast<float4*>(o + (int64_t)g * (Wc * 8)) = make_float4(l0[4 * g + 0], l0[4 * g + 1], l0[4 * g + 2], l0[4 * g + 3]);
                *reinterpret_cast<float4*>(o + (int64_t)(4 + g) * (Wc * 8)) = make_float4(l1[4 * g + 0], l1[4 * g + 1], l1[4 * g + 2], l1[4 * g + 3]);
            }
        }
        if (dc.bits) det_emit_candidates(l0, l1, valid, cell, hh, dc);
    }
}

// host: [256][65] transposed weights -> MFMA A-fragment order [m][t4][kk][i][e] = W[32 m + i][128 kk + 4 t4 + e], plus the dustbin row
void detector_pack_weights(const float* wT /*[256][65]*/, float* wA /*16384*/, float* wdust /*256*/) {
    for (int m = 0; m < 2; ++m)
        for (int t4 = 0; t4 < 32; ++t4)
            for (int kk = 0; kk < 2; ++kk)
                for (int i = 0; i < 32; ++i)
                    for (int e = 0; e < 4; ++e)
                        wA[((((size_t)m * 32 + t4) * 2 + kk) * 32 + i) * 4 + e] = wT[(size_t)(128 * kk + 4 * t4 + e) * 65 + 32 * m + i];
    for (int k = 0; k < 256; ++k) wdust[k] = wT[(size_t)k * 65 + 64];
}

int detector_head_mfma(hipStream_t st, int precision, const void* in, int in_stride, int in_off, int batch, int Hc, int Wc,
                       const float* wA, const float* wdust, const float* bias, float* semi, int n_cu, const DetCand& dc) {
    const int n_cells = batch * Hc * Wc;
    const size_t smem = (size_t)(16384 + 256 + 80) * 4;
    int grid = cdiv(cdiv(n_cells, 32), 4);
    if (n_cu > 0 && grid > n_cu) grid = n_cu;
    if (precision == OMNI_PREC_F16) {
        OMNI_HIP_TRY(hipFuncSetAttribute((const void*)detector_head_mfma_kernel<_Float16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        hipLaunchKernelGGL(detector_head_mfma_kernel<_Float16>, dim3(grid), dim3(DETM_THREADS), smem, st, (const _Float16*)in, in_stride, in_off,
                           n_cells, Hc, Wc, wA, wdust, bias, semi, dc);
    } else {
        OMNI_HIP_TRY(hipFuncSetAttribute((const void*)detector_head_mfma_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        hipLaunchKernelGGL(detector_head_mfma_kernel<float>, dim3(grid), dim3(DETM_THREADS), smem, st, (const float*)in, in_stride, in_off,
                           n_cells, Hc, Wc, wA, wdust, bias, semi, dc);
    }
    OMNI_LAUNCH_CHECK();
    return OMNI_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// The same head for fp16 activations on the fp16 matrix pipe (v_mfma_f32_32x32x16_f16, 16x the K rate of the f32 form): the activations ARE
// fp16 (exact operands); only the f32 weights are split, w = hi + lo with hi = fp16(w), lo = fp16(w - hi), and both products are
// accumulated in f32 (a fp16 x fp16 product is exact in f32): fp32-class logits -- the weight is represented to 2^-22 -- at 4 MFMAs of 32
// cycles per 16 input channels instead of 16 MFMAs of 64 cycles.  B operand = 16 bytes of the cell's channels straight from HBM (no
// conversion); the dustbin logit stays a VALU dot product over the same registers.  Epilogue (softmax over 65, depth-to-space) unchanged.
// ---------------------------------------------------------------------------------------------------------------------------
// T = float (OMNI_PREC_SPLIT: the heads layer's fp32 output): the activations are split as well, x = hi + lo in registers right behind the load, and the
// product is the three terms of the split convolutions, w_lo x_hi + w_hi x_hi + w_hi x_lo (6 MFMAs of 32 cycles per 16 input channels where the exact-f32
// kernel issues 16 of 64); the dustbin logit stays an fmaf chain over the fp32 values.
template <typename T>
__global__ void __launch_bounds__(DETM_THREADS, 2)      // two waves per SIMD: the launcher puts two workgroups on a CU (66 KB of LDS each); unbounded, hipcc took 290 registers = one
detector_head_mfma16_kernel(const T* __restrict__ in, int in_stride, int in_off, int n_cells, int Hc, int Wc,
                            const uint4* __restrict__ wA16 /*[hl 2][m 2][s 16][lane 64] x 8 halfs*/, const float* __restrict__ wdust /*[256]*/,
                            const float* __restrict__ bias /*[65]*/, float* __restrict__ semi, DetCand dc) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    uint4* wl = reinterpret_cast<uint4*>(smem_raw);            // 4096 x 16 B = 64 KB
    float* wd = reinterpret_cast<float*>(wl + 4096);           // [256]
    float* bl = wd + 256;                                      // [65]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, hh = lane >> 5;
    for (int i = tid; i < 4096; i += DETM_THREADS) wl[i] = wA16[i];
    wd[tid] = wdust[tid];
    if (tid < 65) bl[tid] = bias[tid];
    __syncthreads();
    const float dust_bias = bl[64];
    const int n_frag = (n_cells + 31) >> 5;
    for (int f = blockIdx.x * 4 + wave; f < n_frag; f += gridDim.x * 4) {
        const int cell = f * 32 + n;
        const bool valid = cell < n_cells;
        const T* ip = in + (int64_t)(valid ? cell : n_cells - 1) * in_stride + in_off + hh * 8;     // k-step s: channels [16 s + 8 hh, + 8)
        floatx16 acc0, acc1;
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
        float dust = 0.f;
        typedef T xvec_t __attribute__((ext_vector_type(8)));               // a lane's 8 channels of one k-step: 16 bytes of halfs or 32 bytes of floats
        constexpr int KH = sizeof(T) == 2 ? 8 : 4, NCH = 16 / KH;     // k-steps per prefetched chunk: 64 registers of activations in flight either way
        xvec_t xr[KH], xn[KH];
#pragma unroll
        for (int i = 0; i < KH; ++i) xr[i] = *reinterpret_cast<const xvec_t*>(ip + i * 16);
#pragma unroll 1
        for (int hf = 0; hf < NCH; ++hf) {                    // chunks of KH k-steps; the next chunk's loads fly during the current one
            if (hf + 1 < NCH) {
#pragma unroll
                for (int i = 0; i < KH; ++i) xn[i] = *reinterpret_cast<const xvec_t*>(ip + ((hf + 1) * KH + i) * 16);
            }
#pragma unroll
            for (int t = 0; t < KH; ++t) {
                const int sidx = hf * KH + t;
                const half8_t a0h = __builtin_bit_cast(half8_t, wl[((0 * 2 + 0) * 16 + sidx) * 64 + lane]);
                const half8_t a1h = __builtin_bit_cast(half8_t, wl[((0 * 2 + 1) * 16 + sidx) * 64 + lane]);
                const half8_t a0l = __builtin_bit_cast(half8_t, wl[((1 * 2 + 0) * 16 + sidx) * 64 + lane]);
                const half8_t a1l = __builtin_bit_cast(half8_t, wl[((1 * 2 + 1) * 16 + sidx) * 64 + lane]);
                half8_t xh;
                if constexpr (sizeof(T) == 2) xh = xr[t]; else xh = __builtin_convertvector(xr[t], half8_t);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0l, xh, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1l, xh, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, xh, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, xh, acc1, 0, 0, 0);
                if constexpr (sizeof(T) == 4) {               // the activation's lo half: x - half(x), exactly representable differences rounded once
                    typedef float f8_t __attribute__((ext_vector_type(8)));
                    const half8_t xl = __builtin_convertvector(xr[t] - __builtin_convertvector(xh, f8_t), half8_t);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, xl, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, xl, acc1, 0, 0, 0);
                }
                const float4 d0 = *reinterpret_cast<const float4*>(wd + sidx * 16 + hh * 8), d1 = *reinterpret_cast<const float4*>(wd + sidx * 16 + hh * 8 + 4);
                dust = fmaf((float)xr[t][0], d0.x, dust); dust = fmaf((float)xr[t][1], d0.y, dust);
                dust = fmaf((float)xr[t][2], d0.z, dust); dust = fmaf((float)xr[t][3], d0.w, dust);
                dust = fmaf((float)xr[t][4], d1.x, dust); dust = fmaf((float)xr[t][5], d1.y, dust);
                dust = fmaf((float)xr[t][6], d1.z, dust); dust = fmaf((float)xr[t][7], d1.w, dust);
            }
#pragma unroll
            for (int i = 0; i < KH; ++i) xr[i] = xn[i];
        }
        dust = dust + __shfl_xor(dust, 32, 64) + dust_bias;
        float l0[16], l1[16], mx = dust;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = (r & 3) + 8 * (r >> 2) + 4 * hh;
            l0[r] = acc0[r] + bl[c]; l1[r] = acc1[r] + bl[32 + c];
            mx = fmaxf(mx, fmaxf(l0[r], l1[r]));
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { l0[r] = expf(l0[r] - mx); l1[r] = expf(l1[r] - mx); sum += l0[r] + l1[r]; }
        sum = sum + __shfl_xor(sum, 32, 64) + expf(dust - mx);
#pragma unroll
        for (int r = 0; r < 16; ++r) { l0[r] = l0[r] / sum; l1[r] = l1[r] / sum; }
        const int wx = cell % Wc;
        const int hy = (cell / Wc) % Hc;
        const int b = cell / (Wc * Hc);
        if (valid) {
            float* o = semi + ((int64_t)b * Hc * 8 + hy * 8) * (Wc * 8) + wx * 8 + 4 * hh;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                // channel c = 32 m + 8 g + 4 hh + (r & 3)  ->  row ry = 4 m + g, columns rx = 4 hh + (r & 3)
                *reinterpret_cast<float4*>(o + (int64_t)g * (Wc * 8)) = make_float4(l0[4 * g + 0], l0[4 * g + 1], l0[4 * g + 2], l0[4 * g + 3]);
                *reinterpret_cast<float4*>(o + (int64_t)(4 + g) * (Wc * 8)) = make_float4(l1[4 * g + 0], l1[4 * g + 1], l1[4 * g + 2], l1[4 * g + 3]);
            }
        }
        if (dc.bits) det_emit_candidates(l0, l1, valid, cell, hh, dc);
    }
}

// host: [256][65] transposed weights -> split-fp16 A fragments [hl][m][s][lane][e] = split(W[32 m + (lane & 31)][16 s + 8 (lane >> 5) + e])
void detector_pack_weights16(const float* wT /*[256][65]*/, uint16_t* wA16 /*2*2*16*64*8*/) {
    for (int m = 0; m < 2; ++m)
        for (int sx = 0; sx < 16; ++sx)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    const float w = wT[(size_t)(16 * sx + 8 * (lane >> 5) + e) * 65 + 32 * m + (lane & 31)];
                    const uint16_t hi = f2h_bits(w);
                    const uint16_t lo = f2h_bits(w - h2f(hi));
                    wA16[((((size_t)0 * 2 + m) * 16 + sx) * 64 + lane) * 8 + e] = hi;
                    wA16[((((size_t)1 * 2 + m) * 16 + sx) * 64 + lane) * 8 + e] = lo;
                }
}

int detector_head_mfma16(hipStream_t st, int in_precision, const void* in, int in_stride, int in_off, int batch, int Hc, int Wc, const void* wA16, const float* wdust,
                         const float* bias, float* semi, int n_cu, const DetCand& dc) {
    const int n_cells = batch * Hc * Wc;
    const size_t smem = (size_t)4096 * 16 + (256 + 80) * 4;
    int grid = cdiv(cdiv(n_cells, 32), 4);
    if (n_cu > 0 && grid > 2 * n_cu) grid = 2 * n_cu;         // 66 KB of LDS: two workgroups per CU
    if (in_precision == OMNI_PREC_F16) {
        OMNI_HIP_TRY(hipFuncSetAttribute((const void*)detector_head_mfma16_kernel<_Float16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        hipLaunchKernelGGL(detector_head_mfma16_kernel<_Float16>, dim3(grid), dim3(DETM_THREADS), smem, st, (const _Float16*)in, in_stride, in_off, n_cells, Hc, Wc,
                           (const uint4*)wA16, wdust, bias, semi, dc);
    } else {                                                  // fp32 activations, split on the fly (OMNI_PREC_SPLIT)
        OMNI_HIP_TRY(hipFuncSetAttribute((const void*)detector_head_mfma16_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        hipLaunchKernelGGL(detector_head_mfma16_kernel<float>, dim3(grid), dim3(DETM_THREADS), smem, st, (const float*)in, in_stride, in_off, n_cells, Hc, Wc,
                           (const uint4*)wA16, wdust, bias, semi, dc);
    }
    OMNI_LAUNCH_CHECK();
    return OMNI_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// convDb (1x1, 256 -> 256, f32 out) + the descriptor L2 normalisation over the 256 channels in ONE pass (fp16 path).  The pair is
// HBM bound (0.59 GFLOP but 2.3 MB read + 4.6 MB written per image); run separately the f32 map made a second round trip
// (write 4.6, read 4.6, write 4.6 MB).  A workgroup = 8 waves, wave w owns output channels 32w..32w+31 with their A fragments
// RESIDENT IN REGISTERS (16 k-steps x 4 VGPRs); it walks 32-pixel tiles: the 32 x 256-channel fp16 tile (16 KB)
// goes global -> registers (prefetched two tiles ahead) -> LDS (chunks XOR-swizzled with pixel & 15 so the B-operand ds_read_b128 of
// 16 consecutive pixels hit 16 distinct slots), 16 MFMAs per wave, bias, per-pixel sum of squares (lane pair by shuffle, the 8 waves
// through LDS), scale, 16-byte stores.  No LDS-DMA here: plain loads keep hipcc's counted waits exact.
//   algorithmic bytes per image: 4500 px x (512 B read + 1024 B written) = 6.9 MB
#define CDB_PX 32
#define CDB_WAVES 8
void convdb_pack_weights(const float* w /*[256 cout][256 cin]*/, uint16_t* frag /*[8][16][64][8]*/) {
    for (int wv = 0; wv < CDB_WAVES; ++wv)
        for (int ks = 0; ks < 16; ++ks)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int co = 32 * wv + (lane & 31), k = 16 * ks + 8 * (lane >> 5) + j;
                    frag[(((size_t)wv * 16 + ks) * 64 + lane) * 8 + j] = f2h_bits(w[(size_t)co * 256 + k]);
                }
}

__global__ void __launch_bounds__(64 * CDB_WAVES)
convdb_l2norm_kernel(const _Float16* __restrict__ in, int in_cstride, const _Float16* __restrict__ wfrag, const float* __restrict__ bias,
                     float* __restrict__ out, int64_t n_pixels) {
    __shared__ __attribute__((aligned(16))) char tile[2][CDB_PX * 512];
    __shared__ float part[2][CDB_WAVES][CDB_PX];
    __shared__ float sbias[256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, kg = lane >> 5;
    half8_t wa[16];                                                       // this wave's 32 output channels, all K = 256
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) wa[ks] = *reinterpret_cast<const half8_t*>(wfrag + (((size_t)wave * 16 + ks) * 64 + lane) * 8);
    if (tid < 256) sbias[tid] = bias[tid];
    const int64_t n_tiles = (n_pixels + CDB_PX - 1) / CDB_PX;
    const int64_t G = gridDim.x;
    // staging: thread -> 2 of the tile's 1024 16-byte chunks (q = tid + 512 j: pixel q >> 5, chunk q & 31), coalesced 512 B per pixel
    const int spx0 = tid >> 5, spx1 = spx0 + 16, sc = tid & 31;
    auto load_chunk = [&](int64_t t, int spx) -> uint4 {
        int64_t px = t * CDB_PX + spx;
        px = px < n_pixels ? px : n_pixels - 1;                           // ragged last tile: duplicates, never stored
        return *reinterpret_cast<const uint4*>(in + px * in_cstride + sc * 8);
    };
    const int soff0 = spx0 * 512 + ((sc ^ (spx0 & 15)) << 4), soff1 = spx1 * 512 + ((sc ^ (spx1 & 15)) << 4);
    const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
    int64_t t = blockIdx.x;
    uint4 ra0 = t < n_tiles ? load_chunk(t, spx0) : zero4, ra1 = t < n_tiles ? load_chunk(t, spx1) : zero4;
    uint4 rb0 = t + G < n_tiles ? load_chunk(t + G, spx0) : zero4, rb1 = t + G < n_tiles ? load_chunk(t + G, spx1) : zero4;
    int buf = 0;
    for (; t < n_tiles; t += G) {
        *reinterpret_cast<uint4*>(tile[buf] + soff0) = ra0;
        *reinterpret_cast<uint4*>(tile[buf] + soff1) = ra1;
        ra0 = rb0; ra1 = rb1;
        if (t + 2 * G < n_tiles) { rb0 = load_chunk(t + 2 * G, spx0); rb1 = load_chunk(t + 2 * G, spx1); }
        __syncthreads();              // tile visible (the buffer's previous readers are two barriers behind)
        floatx16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const char* bp = tile[buf] + n * 512;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const half8_t b = *reinterpret_cast<const half8_t*>(bp + (((2 * ks + kg) ^ (n & 15)) << 4));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[ks], b, acc, 0, 0, 0);
        }
        // C layout: channel (r & 3) + 8 (r >> 2) + 4 kg of the wave's 32, pixel n
        float ss = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = acc[r] + sbias[32 * wave + (r & 3) + 8 * (r >> 2) + 4 * kg];
            acc[r] = v;
            ss = fmaf(v, v, ss);
        }
        ss += __shfl_xor(ss, 32, 64);
        if (lane < 32) part[buf][wave][n] = ss;
        __syncthreads();
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < CDB_WAVES; ++w) tot += part[buf][w][n];
        // (rows of key-point slots that do not exist hold zeros or stale data and are never read back -- sp_sample_compact_kernel stops at n_kps -- but a zero
        // norm must not turn them into NaN / Inf either: such a row is written as zeros)
        const float nrm = sqrtf(tot);
        const float den = nrm > 0.f ? nrm : __builtin_inff();
        const int64_t px = t * CDB_PX + n;
        if (px < n_pixels) {
            float* op = out + px * 256 + 32 * wave + 4 * kg;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 v;
                v.x = acc[4 * g] / den; v.y = acc[4 * g + 1] / den; v.z = acc[4 * g + 2] / den; v.w = acc[4 * g + 3] / den;
                *reinterpret_cast<float4*>(op + 8 * g) = v;
            }
        }
        buf ^= 1;
    }
}

int convdb_l2norm(hipStream_t st, const omni_ctx* ctx, const void* in_f16, int in_cstride, const void* wfrag, const float* bias, float* out,
                  int64_t n_pixels) {
    const int64_t tiles = cdiv64(n_pixels, CDB_PX);
    const int cus = ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256;
    const int64_t grid = tiles < cus ? tiles : cus;                     // one persistent workgroup per CU (152 VGPRs x 8 waves)
    hipLaunchKernelGGL(convdb_l2norm_kernel, dim3((unsigned)grid), dim3(64 * CDB_WAVES), 0, st, (const _Float16*)in_f16, in_cstride,
                       (const _Float16*)wfrag, bias, out, n_pixels);
    OMNI_LAUNCH_CHECK();
    return OMNI_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// OMNI_PREC_SPLIT: convDb (1 x 1, 256 -> 256) + the per-row L2 norm over fp32 rows -- the compact [key point][corner] rows conv_split_c128_sparse leaves --
// on the fp16 matrix cores with split operands, in ONE pass.  Structure of convdb_l2norm_kernel (8 waves x 32 output channels with register-resident A
// fragments, 32-row tiles global -> registers two tiles ahead -> swizzled LDS, sum of squares across the waves through LDS); the weights are (hi, lo)
// fragment pairs (128 registers), a row's fp32 values are split into a hi and a lo tile as they are parked in LDS, and a k-step is the three terms
// w_lo x_hi + w_hi x_hi + w_hi x_lo.  Replaces an exact-f32 MFMA convolution (16 MFMAs of 64 cycles per 16 channels, 126 us per 51 200 rows) and a separate
// normalisation pass; descriptors only (north_star: 1e-3): the key points do not see it.
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64 * CDB_WAVES)
convdb_l2norm_split_kernel(const float* __restrict__ in, int in_cstride, const _Float16* __restrict__ wfrag_hi, const _Float16* __restrict__ wfrag_lo,
                           const float* __restrict__ bias, float* __restrict__ out, int64_t n_pixels) {
    __shared__ __attribute__((aligned(16))) char tile[2][2][CDB_PX * 512];           // [buffer][hi | lo]
    __shared__ float part[2][CDB_WAVES][CDB_PX];
    __shared__ float sbias[256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, kg = lane >> 5;
    half8_t wh[16], wl[16];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        wh[ks] = *reinterpret_cast<const half8_t*>(wfrag_hi + (((size_t)wave * 16 + ks) * 64 + lane) * 8);
        wl[ks] = *reinterpret_cast<const half8_t*>(wfrag_lo + (((size_t)wave * 16 + ks) * 64 + lane) * 8);
    }
    if (tid < 256) sbias[tid] = bias[tid];
    const int64_t n_tiles = (n_pixels + CDB_PX - 1) / CDB_PX;
    const int64_t G = gridDim.x;
    // staging: thread -> 4 of the tile's 2048 float4 (q = tid + 512 j: row q >> 6, float4 q & 63 = channels 4 (q & 63) ..), 1 KiB per row coalesced
    const int f4 = tid & 63, chunk = f4 >> 1, hb = (f4 & 1) * 8;                       // its 8 bytes inside the 16-byte chunk of 8 halfs
    auto load4 = [&](int64_t t, int j) -> float4 {
        int64_t px = t * CDB_PX + (tid >> 6) + 8 * j;
        px = px < n_pixels ? px : n_pixels - 1;                                        // ragged last tile: duplicates, never stored
        return *reinterpret_cast<const float4*>(in + px * in_cstride + f4 * 4);
    };
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    int64_t t = blockIdx.x;
    float4 ra[4], rb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { ra[j] = t < n_tiles ? load4(t, j) : zero4; rb[j] = t + G < n_tiles ? load4(t + G, j) : zero4; }
    int buf = 0;
    for (; t < n_tiles; t += G) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int spx = (tid >> 6) + 8 * j;
            const int off = spx * 512 + ((chunk ^ (spx & 15)) << 4) + hb;
            typedef float f4_t __attribute__((ext_vector_type(4)));
            typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
            const f4_t v = {ra[j].x, ra[j].y, ra[j].z, ra[j].w};
            const h4_t hi = __builtin_convertvector(v, h4_t);
            const h4_t lo = __builtin_convertvector(v - __builtin_convertvector(hi, f4_t), h4_t);
            *reinterpret_cast<h4_t*>(tile[buf][0] + off) = hi;
            *reinterpret_cast<h4_t*>(tile[buf][1] + off) = lo;
            ra[j] = rb[j];
        }
        if (t + 2 * G < n_tiles) {
#pragma unroll
            for (int j = 0; j < 4; ++j) rb[j] = load4(t + 2 * G, j);
        }
        __syncthreads();              // tile visible (the buffer's previous readers are two barriers behind)
        floatx16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const char* bph = tile[buf][0] + n * 512;
        const char* bpl = tile[buf][1] + n * 512;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const int o = ((2 * ks + kg) ^ (n & 15)) << 4;
            const half8_t xh = *reinterpret_cast<const half8_t*>(bph + o), xl = *reinterpret_cast<const half8_t*>(bpl + o);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[ks], xh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ks], xh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ks], xl, acc, 0, 0, 0);
        }
        float ss = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = acc[r] + sbias[32 * wave + (r & 3) + 8 * (r >> 2) + 4 * kg];
            acc[r] = v;
            ss = fmaf(v, v, ss);
        }
        ss += __shfl_xor(ss, 32, 64);
        if (lane < 32) part[buf][wave][n] = ss;
        __syncthreads();
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < CDB_WAVES; ++w) tot += part[buf][w][n];
        const float nrm = sqrtf(tot);
        const int64_t px = t * CDB_PX + n;
        if (px < n_pixels) {
            float* op = out + px * 256 + 32 * wave + 4 * kg;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 v;
                v.x = acc[4 * g] / nrm; v.y = acc[4 * g + 1] / nrm; v.z = acc[4 * g + 2] / nrm; v.w = acc[4 * g + 3] / nrm;
                *reinterpret_cast<float4*>(op + 8 * g) = v;
            }
        }
        buf ^= 1;
    }
}

// w (fp32) -> the two fragment arrays of convdb_l2norm_split: hi = half(w), lo = half(w - hi), each in convdb_pack_weights' order
void convdb_pack_weights_split(const float* w /*[256][256]*/, uint16_t* frag_hi /*[65536]*/, uint16_t* frag_lo /*[65536]*/) {
    std::vector<float> lo(65536);
    for (int i = 0; i < 65536; ++i) lo[i] = w[i] - h2f(f2h_bits(w[i]));
    convdb_pack_weights(w, frag_hi);
    convdb_pack_weights(lo.data(), frag_lo);
}

int convdb_l2norm_split(hipStream_t st, const omni_ctx* ctx, const float* in_f32, int in_cstride, const void* wfrag_hi, const void* wfrag_lo, const float* bias,
                        float* out, int64_t n_pixels) {
    const int64_t tiles = cdiv64(n_pixels, CDB_PX);
    const int cus = ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256;
    const int64_t grid = tiles < cus ? tiles : cus;
    hipLaunchKernelGGL(convdb_l2norm_split_kernel, dim3((unsigned)grid), dim3(64 * CDB_WAVES), 0, st, in_f32, in_cstride, (const _Float16*)wfrag_hi,
                       (const _Float16*)wfrag_lo, bias, out, n_pixels);
    OMNI_LAUNCH_CHECK();
    return OMNI_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// convDb + L2 norm + bilinear sampling ONLY where key points are (computeDescriptors, superpoint_tensorrt.cpp:192-230, needs the coarse
// descriptor map at the four cells around each key point: <= 800 of the 4 500 cells of a 600x480 image).  Same structure and the SAME
// arithmetic per cell as convdb_l2norm_kernel above (MFMA k order, bias add, sum-of-squares order over the 8 waves, division) -- a "pixel
// tile" is the 32 corner cells of 8 key points, gathered through the key-point list instead of walked in raster order -- followed by the
// sampling kernel's fmaf chain (corner order x0y0, x1y0, x0y1, x1y1) in the quad's first lane: the sampled descriptor is bit-identical to
// sp_sample_kernel on the dense map, the 4.6 MB/image fp32 map is never written, 5.6x fewer cells are computed.
// raw_desc[b][i][256] for i < n_kps[b]; runs after sp_nms_kernel on the same stream.
// ---------------------------------------------------------------------------------------------------------------
// value of lane (lane & ~3) + J in every lane of the quad (v_mov_b32 with a quad_perm DPP control: no LDS)
template <int J>
__device__ __forceinline__ int quad_bcast_i(int v) { return __builtin_amdgcn_update_dpp(0, v, J * 85, 0xF, 0xF, false); }
template <int J>
__device__ __forceinline__ float quad_bcast(float v) { return __builtin_bit_cast(float, quad_bcast_i<J>(__builtin_bit_cast(int, v))); }

__global__ void __launch_bounds__(64 * CDB_WAVES)
convdb_sparse_kernel(const _Float16* __restrict__ in, int in_cstride, const _Float16* __restrict__ wfrag, const float* __restrict__ bias,
                     int W, int H, int max_num, const float* __restrict__ kps_xy, const int* __restrict__ n_kps, float* __restrict__ raw_desc,
                     int tiles_per_img, int n_tiles, int compact /* in = [image][key point][corner][in_cstride] instead of the coarse map */) {
    __shared__ __attribute__((aligned(16))) char tile[2][CDB_PX * 512];
    __shared__ float part[2][CDB_WAVES][CDB_PX];
    __shared__ float sbias[256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, kg = lane >> 5;
    const int Wc = W >> 3, Hc = H >> 3;
    const float fW = (float)W, fH = (float)H, fWc = (float)Wc, fHc = (float)Hc;
    half8_t wa[16];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) wa[ks] = *reinterpret_cast<const half8_t*>(wfrag + (((size_t)wave * 16 + ks) * 64 + lane) * 8);
    if (tid < 256) sbias[tid] = bias[tid];
    const int G = gridDim.x;
    // corner cell `slot` (key point slot >> 2 of the tile, corner slot & 3) of tile t: coarse-map pixel index (-1: outside the map or no such key
    // point) and bilinear weight.  The float expressions are sp_sample_kernel's, operation by operation (grid_sample, align_corners = false).
    auto corner = [&](int t, int slot, float& wgt, int& kp, int& b) -> int64_t {
        b = t / tiles_per_img;
        kp = (t - b * tiles_per_img) * CSP_KP + (slot >> 2);
        wgt = 0.f;
        if (kp >= n_kps[b]) return -1;
        const float kx = kps_xy[((int64_t)b * max_num + kp) * 2 + 0], ky = kps_xy[((int64_t)b * max_num + kp) * 2 + 1];
        const float gx = __fsub_rn(__fdiv_rn(__fmul_rn(2.0f, kx), fW), 1.0f);
        const float gy = __fsub_rn(__fdiv_rn(__fmul_rn(2.0f, ky), fH), 1.0f);
        const float ix = __fdiv_rn(__fsub_rn(__fmul_rn(__fadd_rn(gx, 1.0f), fWc), 1.0f), 2.0f);
        const float iy = __fdiv_rn(__fsub_rn(__fmul_rn(__fadd_rn(gy, 1.0f), fHc), 1.0f), 2.0f);
        const float fx0 = floorf(ix), fy0 = floorf(iy);
        const int cx = (int)fx0 + (slot & 1), cy = (int)fy0 + ((slot >> 1) & 1);
        const float wx = (slot & 1) ? ix - fx0 : (fx0 + 1.f) - ix, wy = (slot & 2) ? iy - fy0 : (fy0 + 1.f) - iy;
        wgt = __fmul_rn(wx, wy);
        if (cx < 0 || cx >= Wc || cy < 0 || cy >= Hc) return -1;
        return compact ? ((int64_t)b * max_num + kp) * 4 + (slot & 3) : ((int64_t)b * Hc + cy) * Wc + cx;
    };
    // staging: thread -> 2 of the tile's 1024 16-byte chunks (cell tid >> 5 and + 16, chunk tid & 31)
    const int spx0 = tid >> 5, spx1 = spx0 + 16, sc = tid & 31;
    const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
    auto load_chunk = [&](int t, int spx) -> uint4 {
        float wgt; int kp, b;
        const int64_t px = corner(t, spx, wgt, kp, b);
        return px >= 0 ? *reinterpret_cast<const uint4*>(in + px * in_cstride + sc * 8) : zero4;
    };
    const int soff0 = spx0 * 512 + ((sc ^ (spx0 & 15)) << 4), soff1 = spx1 * 512 + ((sc ^ (spx1 & 15)) << 4);
    int t = blockIdx.x;
    uint4 ra0 = t < n_tiles ? load_chunk(t, spx0) : zero4, ra1 = t < n_tiles ? load_chunk(t, spx1) : zero4;
    uint4 rb0 = t + G < n_tiles ? load_chunk(t + G, spx0) : zero4, rb1 = t + G < n_tiles ? load_chunk(t + G, spx1) : zero4;
    int buf = 0;
    for (; t < n_tiles; t += G) {
        *reinterpret_cast<uint4*>(tile[buf] + soff0) = ra0;
        *reinterpret_cast<uint4*>(tile[buf] + soff1) = ra1;
        ra0 = rb0; ra1 = rb1;
        if (t + 2 * G < n_tiles) { rb0 = load_chunk(t + 2 * G, spx0); rb1 = load_chunk(t + 2 * G, spx1); }
        __syncthreads();
        floatx16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const char* bp = tile[buf] + n * 512;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const half8_t b8 = *reinterpret_cast<const half8_t*>(bp + (((2 * ks + kg) ^ (n & 15)) << 4));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[ks], b8, acc, 0, 0, 0);
        }
        float ss = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = acc[r] + sbias[32 * wave + (r & 3) + 8 * (r >> 2) + 4 * kg];
            acc[r] = v;
            ss = fmaf(v, v, ss);
        }
        ss += __shfl_xor(ss, 32, 64);
        if (lane < 32) part[buf][wave][n] = ss;
        __syncthreads();
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < CDB_WAVES; ++w) tot += part[buf][w][n];
        const float nrm = sqrtf(tot);
        // this lane's corner: weight, in-map flag; the quad's four (weight, flag) pairs to every lane of the quad
        float wgt; int kp, b;
        const bool inmap = corner(t, n, wgt, kp, b) >= 0;
        const float wq[4] = {quad_bcast<0>(wgt), quad_bcast<1>(wgt), quad_bcast<2>(wgt), quad_bcast<3>(wgt)};
        const int im = (int)inmap;
        const bool vq[4] = {quad_bcast_i<0>(im) != 0, quad_bcast_i<1>(im) != 0, quad_bcast_i<2>(im) != 0, quad_bcast_i<3>(im) != 0};
        const bool store = (n & 3) == 0 && kp < n_kps[b];
        float* op = raw_desc + ((int64_t)b * max_num + kp) * 256 + 32 * wave + 4 * kg;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float dv = acc[4 * g + e] / nrm;
                const float dq[4] = {quad_bcast<0>(dv), quad_bcast<1>(dv), quad_bcast<2>(dv), quad_bcast<3>(dv)};
                float v = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (vq[j]) v = fmaf(dq[j], wq[j], v);
                o[e] = v;
            }
            if (store) *reinterpret_cast<float4*>(op + 8 * g) = make_float4(o[0], o[1], o[2], o[3]);
        }
        buf ^= 1;
    }
}

int convdb_sparse_sample(hipStream_t st, const omni_ctx* ctx, const void* in_f16, int in_cstride, const void* wfrag, const float* bias, int W, int H,
                         int max_num, const float* kps_xy, const int* n_kps, float* raw_desc, int batch, bool compact) {
    const int tiles_per_img = cdiv(max_num, CSP_KP), n_tiles = tiles_per_img * batch;
    const int cus = ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256;
    const int grid = n_tiles < cus ? n_tiles : cus;
    hipLaunchKernelGGL(convdb_sparse_kernel, dim3((unsigned)grid), dim3(64 * CDB_WAVES), 0, st, (const _Float16*)in_f16, in_cstride,
                       (const _Float16*)wfrag, bias, W, H, max_num, kps_xy, n_kps, raw_desc, tiles_per_img, n_tiles, compact ? 1 : 0);
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

// ---- the constant region of the fisheye mask (superpoint.hip): one pixel of a map read out, a rectangle of every image filled with it ----------
__global__ void read_pixel_f16_kernel(const uint4* __restrict__ map, int64_t chunk0, int chunks, uint4* __restrict__ vec) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < chunks) vec[i] = map[chunk0 + i];
}
int conv_read_pixel_f16(hipStream_t st, const void* map, int Ho, int Wo, int C, int y, int x, void* vec) {
    OMNI_REQUIRE(C % 8 == 0 && y >= 0 && y < Ho && x >= 0 && x < Wo, OMNI_ERR_INVALID, "conv_read_pixel_f16: bad pixel");
    const int chunks = C / 8;
    hipLaunchKernelGGL(read_pixel_f16_kernel, dim3(cdiv(chunks, 64)), dim3(64), 0, st, (const uint4*)map, ((int64_t)y * Wo + x) * chunks, chunks, (uint4*)vec);
    OMNI_LAUNCH_CHECK();
    return OMNI_OK;
}
// one thread per 16-byte chunk of the rectangle (a pixel's chunks are consecutive: coalesced row segments)
__global__ void fill_rect_f16_kernel(uint4* __restrict__ map, int Ho, int Wo, int chunks, int y0, int x0, int rh, int rw, const uint4* __restrict__ vec, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // ((b * rh + y) * rw + x) * chunks + c
    if (i >= total) return;
    const int c = (int)(i % chunks);
    const int64_t px = i / chunks;
    const int x = (int)(px % rw);
    const int64_t by = px / rw;
    const int y = (int)(by % rh);
    const int64_t b = by / rh;
    map[((b * Ho + y0 + y) * Wo + x0 + x) * chunks + c] = vec[c];
}
int conv_fill_rect_f16(hipStream_t st, void* map, int batch, int Ho, int Wo, int C, int y0, int y1, int x0, int x1, const void* vec) {
    OMNI_REQUIRE(C % 8 == 0 && y0 >= 0 && y1 <= Ho && x0 >= 0 && x1 <= Wo, OMNI_ERR_INVALID, "conv_fill_rect_f16: rectangle outside the map");
    if (y1 <= y0 || x1 <= x0 || batch <= 0) return OMNI_OK;
    const int chunks = C / 8;
    const int64_t total = (int64_t)batch * (y1 - y0) * (x1 - x0) * chunks;
    hipLaunchKernelGGL(fill_rect_f16_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, st, (uint4*)map, Ho, Wo, chunks, y0, x0, y1 - y0, x1 - x0, (const uint4*)vec, total);
    OMNI_LAUNCH_CHECK();
    return OMNI_OK;
}

__global__ void fill_rect_bytes_kernel(char* __restrict__ map, int64_t img_bytes, int64_t row_bytes, int64_t org_bytes, int chunks, int y0, int x0, int rh, int rw,
                                       const uint4* __restrict__ vec, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // ((b * rh + y) * rw + x) * chunks + c
    if (i >= total) return;
    const int c = (int)(i % chunks);
    const int64_t px = i / chunks;
    const int x = (int)(px % rw);
    const int64_t by = px / rw;
    const int y = (int)(by % rh);
    const int64_t b = by / rh;
    *reinterpret_cast<uint4*>(map + b * img_bytes + org_bytes + (y0 + y) * row_bytes + ((int64_t)(x0 + x) * chunks + c) * 16) = vec[c];
}
int conv_read_pixel_bytes(hipStream_t st, const void* map, int64_t row_bytes, int64_t org_bytes, int pix_bytes, int y, int x, void* vec) {
    OMNI_REQUIRE(pix_bytes % 16 == 0 && y >= 0 && x >= 0 && (org_bytes + y * row_bytes + (int64_t)x * pix_bytes) % 16 == 0, OMNI_ERR_INVALID, "conv_read_pixel_bytes: bad pixel");
    const int chunks = pix_bytes / 16;
    hipLaunchKernelGGL(read_pixel_f16_kernel, dim3(cdiv(chunks, 64)), dim3(64), 0, st, (const uint4*)map, (org_bytes + y * row_bytes + (int64_t)x * pix_bytes) / 16, chunks, (uint4*)vec);
    OMNI_LAUNCH_CHECK();
    return OMNI_OK;
}
int conv_fill_rect_bytes(hipStream_t st, void* map, int batch, int64_t img_bytes, int64_t row_bytes, int64_t org_bytes, int pix_bytes, int y0, int y1, int x0, int x1,
                         const void* vec) {
    OMNI_REQUIRE(pix_bytes % 16 == 0 && img_bytes % 16 == 0 && row_bytes % 16 == 0 && org_bytes % 16 == 0 && y0 >= 0 && x0 >= 0, OMNI_ERR_INVALID, "conv_fill_rect_bytes: bad layout");
    if (y1 <= y0 || x1 <= x0 || batch <= 0) return OMNI_OK;
    const int chunks = pix_bytes / 16;
    const int64_t total = (int64_t)batch * (y1 - y0) * (x1 - x0) * chunks;
    hipLaunchKernelGGL(fill_rect_bytes_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, st, (char*)map, img_bytes, row_bytes, org_bytes, chunks, y0, x0, y1 - y0,
                       x1 - x0, (const uint4*)vec, total);
    OMNI_LAUNCH_CHECK();
    return OMNI_OK;
}

}  // namespace omni
