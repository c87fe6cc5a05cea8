// OMNI_PREC_SPLIT: the 3x3 convolutions of the SuperPoint graph (swarm_loop/superpoint.ipynb:143-181) at fp32-class accuracy on
// the fp16 matrix cores.  Every fp32 operand is carried as a pair of halfs, v = hi + lo with |v - hi - lo| <= 2^-22 |v|, and a
// product is three v_mfma_f32_32x32x16_f16 terms
//     x.w ~ xh.wh + xh.wl + xl.wh        (the dropped xl.wl is 2^-22 relative; accumulation is fp32 inside the MFMA)
// -- the scheme of conv1a inside the fp16 path's fused kernel and of MobileNetVLAD's blocks (vlad_s.hip), here for the eight
// layers that hold 99 % of the network's FLOPs.  north_star's tolerance (key points identical to the fp32 graph, descriptors
// 1e-3) needs fp32-class activations end to end; the exact-f32 MFMA (v_mfma_f32_32x32x2_f32) runs at 1/16 of the fp16 rate,
// three fp16 terms at 1/3.
//
// Activation layout in HBM ("split-64", NHWC): per pixel, per block of 64 channels, 256 bytes = [hi c0..c63 | lo c0..c63] halfs,
// all values pre-multiplied by SPL_ACT_SCALE (a power of two: exact) so that the lo halves of small activations stay normal fp16
// numbers.  Weights are multiplied by a per-layer power of two before they are split (same reason); the epilogue undoes both.
//
// Kernel = the register-stationary design of conv3x3_c128_rs_kernel (conv.hip): 4 waves, ONE per SIMD with the whole register
// file; a wave keeps the split A fragments of ITS 32 output channels x 64 input channels x 9 taps in 288 registers (36 wh + 36 wl
// fragments); LDS holds only two halo tiles of 272 "virtual pixels" x 256 B filled by LDS-DMA (16-byte chunks XOR-swizzled with
// the pixel index), one barrier per tile.  A B fragment read (32 pixels x 16 channels of hi or lo) feeds the taps of up to two
// output rows, and a hi fragment both wh and wl: 96 ds_read_b128 and 216 MFMAs per wave per tile.
//   cin = 64  (conv1b, conv2a, conv2b, conv3a): tile = 4 rows x 32 pixels x 64 output channels; wave = (32 output channels,
//             row pair), halo 6 x 34 pixels;
//   cin = 128 (conv3b, conv4a, conv4b, convPa|convDa): tile = 2 rows x 32 pixels x 64 output channels; wave = (32 output
//             channels, 64-channel block of the input): K is split over the two waves of a pair, whose partial accumulators
//             meet through 16 KB of LDS (each wave finishes half of the pair's registers); halo 4 x 34 pixels x 2 blocks.
#include "conv.h"
#include <type_traits>

namespace omni {

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef _Float16 half2v_t __attribute__((ext_vector_type(2)));
typedef float float2v_t __attribute__((ext_vector_type(2)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

#define SPL_ITW 34
#define SPL_VPIX 272                                  // virtual pixels per halo buffer
#define SPL_BUF_BYTES (SPL_VPIX * 256)                // 69 632
#define SPL_XCH_BYTES 16384                           // partial accumulators of the K-split wave pairs (cin = 128)
#define SPL_SMEM (2 * SPL_BUF_BYTES + SPL_XCH_BYTES + 256)   // + the 64 biases of this workgroup's output channels
#define SPL_ACT_SCALE 32.0f                           // activations are stored x 32: fp16 holds |v| < 2047, lo halves are normal numbers down to |v| = 0.004

static inline uint16_t f2h_bits(float v) { const __half h = __float2half_rn(v); uint16_t u; memcpy(&u, &h, 2); return u; }
static inline float h2f(uint16_t u) { __half h; memcpy(&h, &u, 2); return __half2float(h); }

// OIHW fp32 (3x3) -> split A fragments [g32 = cout / 32][cb = cin / 64][wh | wl][tap][kg4][lane][8 halfs]:
//   cout = g32 * 32 + (lane & 31), cin = cb * 64 + kg4 * 16 + (lane >> 5) * 8 + e  (the operand order of v_mfma_f32_32x32x16_f16)
// of w * 2^k, k chosen so that max |w| * 2^k is in [256, 512).  Returns 2^-k (the epilogue's factor).
float conv_pack_weights_split(const float* w, int cin, int cout, uint16_t* out) {
    float mx = 0.f;
    const size_t n = (size_t)cin * cout * 9;
    for (size_t i = 0; i < n; ++i) mx = fmaxf(mx, fabsf(w[i]));
    int ex = 0;
    if (mx > 0.f) (void)frexpf(mx, &ex);              // mx = m * 2^ex, m in [0.5, 1)
    const int k = 9 - ex;
    size_t o = 0;
    for (int g32 = 0; g32 < cout / 32; ++g32)
        for (int cb = 0; cb < cin / 64; ++cb)
            for (int hl = 0; hl < 2; ++hl)
                for (int tap = 0; tap < 9; ++tap)
                    for (int kg4 = 0; kg4 < 4; ++kg4)
                        for (int l = 0; l < 64; ++l)
                            for (int e = 0; e < 8; ++e) {
                                const int co = g32 * 32 + (l & 31), ci = cb * 64 + kg4 * 16 + (l >> 5) * 8 + e;
                                const float v = ldexpf(w[((size_t)co * cin + ci) * 9 + tap], k);
                                const uint16_t hi = f2h_bits(v);
                                out[o++] = hl ? f2h_bits(v - h2f(hi)) : hi;
                            }
    return ldexpf(1.f, -k);
}

__device__ __forceinline__ float spl_max(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, __builtin_inff()); }
__device__ __forceinline__ float spl_swap_pairs(float v) {      // value of lane ^ 1
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
}

// B fragment read L = (kx * 4 + r) * 8 + kg of a wave's 4 halo rows: kg 0-3 = the hi halves of 16-channel groups 0-3, kg 4-7 the lo halves
template <int L>
__device__ __forceinline__ void spl_read(uint32_t row_base /* lds + n_eff * 256 */, int n_eff, int hh, half8_t& dst) {
    constexpr int kx = L / 32, r = (L / 8) % 4, kg = L % 8;
    constexpr int pc = r * SPL_ITW + kx;
    const uint32_t addr = (row_base + pc * 256 + ((((pc + n_eff) & 15) ^ hh) << 4)) ^ (kg << 5);
    asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr));
}
template <int N>
__device__ __forceinline__ void spl_wait(half8_t& f) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "i"(N)); }

// halo row r feeds output row 0 through tap row ky = r and output row 1 through ky = r - 1
template <int L>
__device__ __forceinline__ void spl_steps(uint32_t row_base, int n_eff, int hh, const half8_t (&wreg)[72], floatx16 (&acc)[2], half8_t (&fb)[3]) {
    if constexpr (L < 96) {
        constexpr int kx = L / 32, r = (L / 8) % 4, kg = L % 8, kq = kg & 3;
        constexpr bool row0 = r <= 2, row1 = r >= 1;
        constexpr int t0 = (r * 3 + kx) * 4 + kq, t1 = ((r - 1) * 3 + kx) * 4 + kq;
        if constexpr (L + 2 < 96) spl_read<L + 2>(row_base, n_eff, hh, fb[(L + 2) % 3]);
        spl_wait<(L + 2 < 96) ? 2 : (L + 1 < 96 ? 1 : 0)>(fb[L % 3]);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (row0) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[t0], fb[L % 3], acc[0], 0, 0, 0);
        if constexpr (row1) acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[t1], fb[L % 3], acc[1], 0, 0, 0);
        if constexpr (kg < 4) {                                                     // a hi fragment also meets the lo halves of the weights
            if constexpr (row0) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[36 + t0], fb[L % 3], acc[0], 0, 0, 0);
            if constexpr (row1) acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[36 + t1], fb[L % 3], acc[1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        spl_steps<L + 1>(row_base, n_eff, hh, wreg, acc, fb);
    }
}

// 8 accumulator values of one pixel = register groups 2 gp, 2 gp + 1 of a 32-channel fragment (channels 16 gp + 4 hh + {0..3} and
// 16 gp + 8 + 4 hh + {0..3}) -> v = a * inv + bias -> ReLU ->
//   OUT_F32: two float4 stores;   else: hi = half(v), lo = half(v - hi), one 16-byte store each after a v_permlane32_swap per
//   dword (the half-waves hold interleaved 4-channel runs of the same pixel: afterwards the lower one owns channels [16 gp, +8)
//   and the upper one [16 gp + 8, +8)).  Every lane must call this (the swap is a cross-lane operation); pred guards the stores.
template <bool OUT_F32>
__device__ __forceinline__ void spl_store_pair(const float (&a)[8], const float4& b0, const float4& b1, float inv, int relu, void* frag_out, int gp,
                                               int hh, bool pred) {
    float v[8];
    v[0] = fmaf(a[0], inv, b0.x); v[1] = fmaf(a[1], inv, b0.y); v[2] = fmaf(a[2], inv, b0.z); v[3] = fmaf(a[3], inv, b0.w);
    v[4] = fmaf(a[4], inv, b1.x); v[5] = fmaf(a[5], inv, b1.y); v[6] = fmaf(a[6], inv, b1.z); v[7] = fmaf(a[7], inv, b1.w);
    if constexpr (OUT_F32) {
        if (relu) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
        }
        if (pred) {
            float* o = reinterpret_cast<float*>(frag_out) + 16 * gp + 4 * hh;
            *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(o + 8) = make_float4(v[4], v[5], v[6], v[7]);
        }
    } else {
        const float lo_lim = relu ? 0.f : -65000.f;
        uint32_t dh[2][2], dl[2][2];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float x0 = __builtin_amdgcn_fmed3f(v[2 * j], lo_lim, 65000.f), x1 = __builtin_amdgcn_fmed3f(v[2 * j + 1], lo_lim, 65000.f);
            float2v_t f; f[0] = x0; f[1] = x1;
            const half2v_t h = __builtin_convertvector(f, half2v_t);
            float2v_t r; r[0] = x0 - (float)h[0]; r[1] = x1 - (float)h[1];
            const half2v_t l = __builtin_convertvector(r, half2v_t);
            dh[j >> 1][j & 1] = __builtin_bit_cast(uint32_t, h);
            dl[j >> 1][j & 1] = __builtin_bit_cast(uint32_t, l);
        }
        uint32_t xh[2], yh[2], xl[2], yl[2];
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            auto r = __builtin_amdgcn_permlane32_swap(dh[0][w], dh[1][w], false, false);
            xh[w] = r[0]; yh[w] = r[1];
            auto q = __builtin_amdgcn_permlane32_swap(dl[0][w], dl[1][w], false, false);
            xl[w] = q[0]; yl[w] = q[1];
        }
        if (pred) {
            _Float16* o = reinterpret_cast<_Float16*>(frag_out) + 16 * gp + 8 * hh;
            *reinterpret_cast<uint4*>(o) = make_uint4(xh[0], xh[1], yh[0], yh[1]);
            *reinterpret_cast<uint4*>(o + 64) = make_uint4(xl[0], xl[1], yl[0], yl[1]);
        }
    }
}

// TRN (cin = 128, no pooling): transposed tiles -- the 32-pixel fragments run along y, the two fragment rows along x (the LDS image, the k order
// and every MFMA are those of the plain kernel; only the pixel <-> address maps and the tap the weights are loaded for differ).  A 60x75 layer
// is 2 x 38 tiles instead of 3 x 30: 75-pixel rows fill 2.3 of 3 fragments, 60-pixel columns 1.9 of 2.
template <bool C128, bool POOL, bool OUT_F32, bool TRN = false>
__global__ void __launch_bounds__(256, 1)
conv3x3_split_kernel(const char* __restrict__ in, void* __restrict__ out, const _Float16* __restrict__ wp, const float* __restrict__ bias,
                     float inv, int H, int W, int cout, int n_cg, int tiles_x, int tiles_y, int batch, int relu, const char* __restrict__ zero_page) {
    extern __shared__ __attribute__((aligned(256))) char smem_raw[];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem_raw;
    static_assert(!TRN || (C128 && !POOL), "transposed tiles: cin = 128 without pooling only");
    constexpr int TH = C128 ? 2 : 4, ITH = TH + 2;
    constexpr int PIXB = C128 ? 512 : 256;                  // bytes per input pixel in HBM
    constexpr int NPIECES = C128 ? 68 : 51, PPW = C128 ? 17 : 13;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int co = wave & 1, part = wave >> 1;
    const int n = lane & 31, hh = lane >> 5;
    const int cg = blockIdx.x % n_cg, wg = blockIdx.x / n_cg, nwg = gridDim.x / n_cg;
    const int tiles_per_img = tiles_x * tiles_y;
    const int total = batch * tiles_per_img;
    const int g32 = cg * 2 + co;

    half8_t wreg[72];
    {
        const int cb = C128 ? part : 0, ncb = C128 ? 2 : 1;
        const _Float16* wbase = wp + ((int64_t)g32 * ncb + cb) * (2 * 36 * 512) + lane * 8;
#pragma unroll
        for (int s = 0; s < 72; ++s) {
            const int hl = s / 36, tk = (s % 36) / 4, kq = s % 4;
            const int tap = TRN ? (tk % 3) * 3 + tk / 3 : tk;          // transposed tiles: the kernel's (row, column) shifts are the image's (column, row)
            wreg[s] = *reinterpret_cast<const half8_t*>(wbase + ((hl * 9 + tap) * 4 + kq) * 512);
        }
    }
    float* const bias_lds = reinterpret_cast<float*>(smem_raw + 2 * SPL_BUF_BYTES + SPL_XCH_BYTES);
    if (tid < 64) bias_lds[tid] = bias[cg * 64 + tid];

    auto tile_origin = [&](int t, int& b, int& ty0, int& tx0) {
        b = t / tiles_per_img;
        const int r = t - b * tiles_per_img;
        ty0 = (r / tiles_x) * (TRN ? 32 : TH); tx0 = (r % tiles_x) * (TRN ? TH : 32);
    };
    // DMA piece p (1 KiB) = virtual pixels [4 p, 4 p + 4): lane -> (virtual pixel vp, 16-byte chunk slot); the chunk stored in slot s of
    // virtual pixel vp is the pixel block's chunk s ^ (vp & 15).  cin = 64: vp = halo pixel (6 x 34); cin = 128: vp = block * 136 + halo pixel
    auto src_of = [&](int vp, int slot, int& iy, int& ix) -> uint32_t {
        int blk = 0, p = vp;
        if constexpr (C128) { blk = vp >= 136 ? 1 : 0; p = vp - 136 * blk; }
        const int iv = p / SPL_ITW, iu = p - iv * SPL_ITW;
        iy = TRN ? iu : iv; ix = TRN ? iv : iu;
        return (uint32_t)(blk * 256 + ((slot ^ (vp & 15)) << 4));
    };
    uint32_t goff[PPW];                       // interior tiles: byte offset of this lane's chunk of piece j relative to the halo origin
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        int piece = wave * PPW + j;
        piece = piece < NPIECES ? piece : NPIECES - 1;
        const int idx = piece * 64 + lane;
        int iy, ix;
        const uint32_t inner = src_of(idx >> 4, idx & 15, iy, ix);
        goff[j] = (uint32_t)(iy * W + ix) * PIXB + inner;
    }
    auto issue = [&](int t, int which) {
        int b, ty0, tx0;
        tile_origin(t, b, ty0, tx0);
        const int y0 = ty0 - 1, x0 = tx0 - 1;
        const char* img = in + (int64_t)b * H * W * PIXB;
        char* base = smem_raw + which * SPL_BUF_BYTES + wave * PPW * 1024;
        if (y0 >= 0 && y0 + (TRN ? SPL_ITW : ITH) <= H && x0 >= 0 && x0 + (TRN ? ITH : SPL_ITW) <= W) {             // interior (wave-uniform)
            const char* org = img + ((int64_t)y0 * W + x0) * PIXB;
#pragma unroll
            for (int j = 0; j < PPW; ++j)
                if (wave * PPW + j < NPIECES)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(org + goff[j]),
                                                     (__attribute__((address_space(3))) void*)(base + j * 1024), 16, 0, 0);
            return;
        }
        // border tile: halo pixels outside the image are DMA'd from a block of zeros (the conv's zero padding lands in LDS with the data)
        int lq = lane >> 4;
        asm volatile("" : "+v"(lq));          // recompute the coordinates per tile rather than hoisting PPW pairs into registers
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            if (wave * PPW + j < NPIECES) {
                const int vp = (wave * PPW + j) * 4 + lq;
                int iy, ix;
                const uint32_t inner = src_of(vp, lane & 15, iy, ix);
                const int gy = y0 + iy, gx = x0 + ix;
                const uint32_t off = (uint32_t)(gy * W + gx) * PIXB + inner;
                const char* src = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? img + off : zero_page + (off & (OMNI_ZERO_PAGE_BYTES - 16));
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(base + j * 1024), 16, 0, 0);
            }
        }
    };

    // this wave's rows of the buffer start at virtual pixel 68 part (cin = 64: output rows 2 part, 2 part + 1 read halo rows 2 part .. + 3)
    // or 136 part (cin = 128: the 4 x 34 pixels of input block `part`)
    const int n_eff = n + part * (C128 ? 136 : 2 * SPL_ITW);
    const int Ho = POOL ? (H >> 1) : H, Wo = POOL ? (W >> 1) : W;
    // bytes per output pixel and this wave's 32-channel fragment inside it
    const int64_t opix = OUT_F32 ? (int64_t)cout * 4 : (int64_t)cout * 4;           // split-64: 2 halfs per channel
    const int64_t ofrag = OUT_F32 ? (int64_t)g32 * 32 * 4 : (int64_t)(g32 >> 1) * 256 + (g32 & 1) * 64;
    char* const outc = reinterpret_cast<char*>(out);

    int t = wg;
    if (t < total) issue(t, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int cur = 0;
    for (; t < total; t += nwg, cur ^= 1) {
        const int tn = t + nwg;
        if (tn < total) issue(tn, cur ^ 1);
        const uint32_t row_base = lds0 + cur * SPL_BUF_BYTES + n_eff * 256;
        floatx16 acc[2];
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[f][i] = 0.f;
        half8_t fb[3];
        spl_read<0>(row_base, n_eff, hh, fb[0]);
        spl_read<1>(row_base, n_eff, hh, fb[1]);
        __builtin_amdgcn_sched_barrier(0);
        spl_steps<0>(row_base, n_eff, hh, wreg, acc, fb);

        int b, ty0, tx0;
        tile_origin(t, b, ty0, tx0);
        const int ox = tx0 + n;
        if constexpr (!C128) {
            // the wave owns rows ty0 + 2 part, + 1 and all 16 registers of its fragment
            float4 bs[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) bs[g] = *reinterpret_cast<const float4*>(bias_lds + co * 32 + 8 * g + 4 * hh);
            const int oy = ty0 + 2 * part;
            if constexpr (POOL) {
                float q[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) q[i] = spl_max(acc[0][i], acc[1][i]);
#pragma unroll
                for (int i = 0; i < 16; ++i) q[i] = spl_max(q[i], spl_swap_pairs(q[i]));
                char* o = outc + (((int64_t)b * Ho + (oy >> 1)) * Wo + (ox >> 1)) * opix + ofrag;
                const bool pred = (oy < H) && (ox < W) && !(n & 1);
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                    float a8[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) a8[j] = q[8 * gp + j];
                    spl_store_pair<OUT_F32>(a8, bs[2 * gp], bs[2 * gp + 1], inv, relu, o, gp, hh, pred);
                }
            } else {
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    char* o = outc + (((int64_t)b * Ho + (oy + f)) * Wo + ox) * opix + ofrag;
                    const bool pred = (oy + f < H) && (ox < W);
#pragma unroll
                    for (int gp = 0; gp < 2; ++gp) {
                        float a8[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) a8[j] = acc[f][8 * gp + j];
                        spl_store_pair<OUT_F32>(a8, bs[2 * gp], bs[2 * gp + 1], inv, relu, o, gp, hh, pred);
                    }
                }
            }
        } else {
            // K split over the wave pair (co, 0) / (co, 1): each wave hands the partner the registers the partner finishes (wave `part`
            // finishes register groups 2 part, 2 part + 1 = channels [16 part, 16 part + 16) of the fragment, both rows)
            float4* const xch = reinterpret_cast<float4*>(smem_raw + 2 * SPL_BUF_BYTES);
            float4* const mine = xch + ((co * 2 + part) * 4) * 64 + lane;
            const float4* const theirs = xch + ((co * 2 + (part ^ 1)) * 4) * 64 + lane;
            float a8[2][8];
            auto exchange = [&](auto PC) {
                constexpr int P = decltype(PC)::value;
#pragma unroll
                for (int f = 0; f < 2; ++f)
#pragma unroll
                    for (int gg = 0; gg < 2; ++gg) {
                        const int r0 = 4 * (2 * (1 - P) + gg);
                        mine[(f * 2 + gg) * 64] = make_float4(acc[f][r0], acc[f][r0 + 1], acc[f][r0 + 2], acc[f][r0 + 3]);
                    }
                __syncthreads();
#pragma unroll
                for (int f = 0; f < 2; ++f)
#pragma unroll
                    for (int gg = 0; gg < 2; ++gg) {
                        const float4 p4 = theirs[(f * 2 + gg) * 64];
                        const int r0 = 4 * (2 * P + gg);
                        a8[f][4 * gg + 0] = acc[f][r0] + p4.x; a8[f][4 * gg + 1] = acc[f][r0 + 1] + p4.y;
                        a8[f][4 * gg + 2] = acc[f][r0 + 2] + p4.z; a8[f][4 * gg + 3] = acc[f][r0 + 3] + p4.w;
                    }
            };
            if (part == 0) exchange(std::integral_constant<int, 0>{}); else exchange(std::integral_constant<int, 1>{});
            const float4 b0 = *reinterpret_cast<const float4*>(bias_lds + co * 32 + 16 * part + 4 * hh);
            const float4 b1 = *reinterpret_cast<const float4*>(bias_lds + co * 32 + 16 * part + 8 + 4 * hh);
            if constexpr (POOL) {
                float q[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) q[j] = spl_max(a8[0][j], a8[1][j]);
#pragma unroll
                for (int j = 0; j < 8; ++j) q[j] = spl_max(q[j], spl_swap_pairs(q[j]));
                char* o = outc + (((int64_t)b * Ho + (ty0 >> 1)) * Wo + (ox >> 1)) * opix + ofrag;
                spl_store_pair<OUT_F32>(q, b0, b1, inv, relu, o, part, hh, (ty0 < H) && (ox < W) && !(n & 1));
            } else {
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    const int oy = TRN ? ty0 + n : ty0 + f, oxx = TRN ? tx0 + f : ox;
                    char* o = outc + (((int64_t)b * Ho + oy) * Wo + oxx) * opix + ofrag;
                    spl_store_pair<OUT_F32>(a8[f], b0, b1, inv, relu, o, part, hh, (oy < H) && (oxx < W));
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // next tile landed (and this tile's stores retired)
        __syncthreads();
    }
}

template <bool C128, bool POOL, bool OUT_F32, bool TRN = false>
static int launch_split(hipStream_t st, const ConvArgs& a) {
    if constexpr (C128 && !POOL && !TRN) {
        // the tile orientation with fewer tiles (OMNI_SPLIT_TRN=0/1 forces one: A/B hook; it fixes the order the taps are summed in)
        static const int force = [] { const char* e = getenv("OMNI_SPLIT_TRN"); return e ? atoi(e) : -1; }();
        const int plain = cdiv(a.W, 32) * cdiv(a.H, 2), trn = cdiv(a.H, 32) * cdiv(a.W, 2);
        if (force == 1 || (force < 0 && trn < plain)) return launch_split<C128, POOL, OUT_F32, true>(st, a);
    }
    auto kfn = conv3x3_split_kernel<C128, POOL, OUT_F32, TRN>;
    static DynSmemState smem_state;
    OMNI_HIP_TRY(ensure_dyn_smem(smem_state, (const void*)kfn, SPL_SMEM));
    constexpr int TH = C128 ? 2 : 4;
    const int tiles_x = cdiv(a.W, TRN ? TH : 32), tiles_y = cdiv(a.H, TRN ? 32 : TH), n_cg = a.cout / 64;
    const int total = a.batch * tiles_x * tiles_y;
    int per_cg = a.n_cu / n_cg;
    if (per_cg < 1) per_cg = 1;
    if (per_cg > total) per_cg = total;
    const float inv = a.out_f32 ? a.split_inv / SPL_ACT_SCALE : a.split_inv;
    hipLaunchKernelGGL(kfn, dim3(per_cg * n_cg), dim3(256), SPL_SMEM, st, reinterpret_cast<const char*>(a.in), a.out,
                       reinterpret_cast<const _Float16*>(a.w_packed), a.bias, inv, a.H, a.W, a.cout, n_cg, tiles_x, tiles_y, a.batch, a.relu ? 1 : 0,
                       reinterpret_cast<const char*>(a.zero_page));
    OMNI_LAUNCH_CHECK();
    return OMNI_OK;
}

// a.in: split-64 activations (x SPL_ACT_SCALE); a.w_packed / a.split_inv from conv_pack_weights_split; a.bias: out_f32 ? the layer's bias
// : SPL_ACT_SCALE * bias; a.out: out_f32 ? NHWC fp32 (true values) : split-64 (x SPL_ACT_SCALE)
int conv_split(hipStream_t st, const ConvArgs& a) {
    OMNI_REQUIRE(a.ksize == 3 && (a.cin == 64 || a.cin == 128) && a.cout % 64 == 0, OMNI_ERR_INVALID, "conv_split: cin=%d cout=%d ksize=%d", a.cin, a.cout, a.ksize);
    OMNI_REQUIRE(!a.pool || (a.H % 2 == 0 && a.W % 2 == 0), OMNI_ERR_INVALID, "pooling needs even H, W");
    OMNI_REQUIRE(a.n_cu > 0 && a.zero_page && a.split_inv > 0.f, OMNI_ERR_INVALID, "conv_split: n_cu / zero_page / split_inv not set");
    OMNI_REQUIRE(!(a.pool && a.out_f32), OMNI_ERR_INVALID, "conv_split: pool + fp32 output not instantiated");
    OMNI_REQUIRE((int64_t)a.H * a.W * (a.cin == 128 ? 512 : 256) < (1ll << 32), OMNI_ERR_INVALID, "conv_split: image too large for 32-bit pixel offsets");
    if (a.cin == 64) {
        if (a.pool) return launch_split<false, true, false>(st, a);
        return a.out_f32 ? launch_split<false, false, true>(st, a) : launch_split<false, false, false>(st, a);
    }
    if (a.pool) return launch_split<true, true, false>(st, a);
    return a.out_f32 ? launch_split<true, false, true>(st, a) : launch_split<true, false, false>(st, a);
}

float conv_split_act_scale() { return SPL_ACT_SCALE; }

// ---------------------------------------------------------------------------------------------------------------
// conv1a (1 -> 64 channels, 3x3, ReLU) from the u8 image, exact fp32 FMAs, written as split-64 activations x SPL_ACT_SCALE
// (the structure of conv1a_kernel in conv.hip: lane = (pixel, group of 8 output channels))
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
conv1a_split_kernel(const uint8_t* __restrict__ gray, int stride, int H, int W, int mask_row0, int mask_row1, const float* __restrict__ w,
                    const float* __restrict__ bias, const float* __restrict__ lut, _Float16* __restrict__ out) {
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
            _Float16* o = out + (((int64_t)b * H + gy) * W + gx) * 128 + cg * 8;
            half8_t hi, lo;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v = __builtin_amdgcn_fmed3f(acc[j] * SPL_ACT_SCALE, 0.f, 65000.f);
                hi[j] = (_Float16)v;
                lo[j] = (_Float16)(v - (float)hi[j]);
            }
            *reinterpret_cast<half8_t*>(o) = hi;
            *reinterpret_cast<half8_t*>(o + 64) = lo;
        }
    }
}

int conv1a_split(hipStream_t st, const uint8_t* gray, int stride, int batch, int H, int W, int fisheye_mask, const float* w, const float* bias,
                 const float* lut, void* out) {
    const int r0 = fisheye_mask ? H * 3 / 4 : H, r1 = fisheye_mask ? H * 3 / 4 + H / 4 : H;   // cv::Rect(0, rows*3/4, cols, rows/4)
    dim3 grid(cdiv(W, 32) * cdiv(H, 8), batch);
    hipLaunchKernelGGL(conv1a_split_kernel, grid, dim3(256), 0, st, gray, stride, H, W, r0, r1, w, bias, lut, (_Float16*)out);
    OMNI_LAUNCH_CHECK();
    return OMNI_OK;
}

// test hook: split-64 NHWC (x SPL_ACT_SCALE) -> NCHW fp32 (true values)
__global__ void split_to_nchw_f32_kernel(const _Float16* __restrict__ in, float* __restrict__ out, int C, int HW, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int64_t b = i / ((int64_t)C * HW);
    const int64_t r = i - b * (int64_t)C * HW;
    const int c = (int)(r / HW);
    const int64_t p = r - (int64_t)c * HW;
    const _Float16* px = in + ((b * HW + p) * C) * 2 + (c >> 6) * 128 + (c & 63);
    out[i] = ((float)px[0] + (float)px[64]) * (1.0f / SPL_ACT_SCALE);
}
int split_to_nchw_f32(hipStream_t st, const void* in, float* out, int batch, int C, int HW) {
    const int64_t total = (int64_t)batch * C * HW;
    hipLaunchKernelGGL(split_to_nchw_f32_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, st, reinterpret_cast<const _Float16*>(in), out, C, HW, total);
    OMNI_LAUNCH_CHECK();
    return OMNI_OK;
}

}  // namespace omni
