// OMNI_PREC_SPLIT, Winograd form: the cin = 64 3x3 convolutions of the SuperPoint graph (conv1b, conv2a, conv2b: swarm_loop/superpoint.ipynb:144-147,
// 65 % of the network's FLOPs) as F(2x2, 3x3) -- 16 products per 2 x 2 outputs and input channel instead of 36 -- still at fp32-class accuracy on the
// fp16 matrix cores:
//     Y = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A        d = 4 x 4 input patch, g = 3 x 3 filter, B^T, G, A^T = the F(2x2,3x3) matrices (entries 0, +-1, +-1/2)
//   * U = G g G^T is computed on the host in double, scaled by a power of two and split into (hi, lo) halfs (conv_pack_weights_wino);
//   * V = B^T d B is computed IN fp32 on the VALU (every entry a sum of four inputs at most), THEN split: hi = half(V), lo = half(V - hi);
//   * a product is the three v_mfma_f32_32x32x16_f16 terms Uh.Vh + Ul.Vh + Uh.Vl of conv_split.hip (fp32 accumulation; the dropped Ul.Vl is 2^-22 relative);
//   * A^T M A, bias, ReLU (and the 2 x 2 max-pool: exactly one Winograd tile) in fp32.
// Measured on the CPU against the torch oracle (tools/round6/wino_numerics.py): 1.2-1.7e-6 per layer, the same as the direct split kernels (gate: 2e-5 of
// the layer's magnitude, tests/test_gpu_superpoint.py).
//
// Activation layout ("raw-32" frames): the zero frame of conv.h (split_frame_h x split_frame_w pixels, pixel (y, x) at row y + 1, column x + 1), per pixel
// 64 fp32 channels = 256 bytes -- the same bytes per pixel as split-64, the values likewise x conv_split_act_scale().  A Winograd layer reads raw-32 (its
// transforms need the fp32 value: a split-64 input would cost two more VALU operations per input) and writes raw-32 (next layer Winograd) or split-64
// (next layer direct).  |activation| must stay below 16 000 / scale = 500 (a transformed entry sums four: fp16's range); the epilogues clamp there.
//
// Kernel: 4 waves, ONE per SIMD with the whole register file (512).  Wave i owns the four Winograd positions (i, j = 0..3) -- row i of the transformed
// patch -- for all 64 output channels: its U fragments (4 j x 4 k-groups x 2 channel fragments x (hi, lo) = 64 fragments) stay in 256 registers, its
// 4 x 2 accumulators in 128.  Tile = 4 output rows x 32 columns = 2 x 16 Winograd tiles; lane (n, hh) = (tile n = 16 trow + tcol, input channels
// 8 hh .. 8 hh + 7 of a 16-channel k-group).  A lane TRANSFORMS ITS OWN B OPERANDS: it reads the two halo rows its position row needs (B^T has two non-zero
// entries per row) x 4 columns x 8 channels from the raw-32 halo in LDS (6 x 34 pixels, filled by LDS-DMA, 16-byte chunks XOR-swizzled with the pixel
// column so that the 16 lanes of a ds_read_b128 phase hit 16 different bank groups), combines them vertically (1 FMA per value) and horizontally (1 add),
// splits (1.5 instructions per value) and feeds 6 MFMAs per position: V never touches LDS.  The output transform along j is lane-local; along i the four
// waves meet through 48 KB of LDS: every wave finishes a quarter of the output channels.
#include "config.h"
#include "conv.h"
#include <type_traits>

namespace omni {

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef _Float16 half2v_t __attribute__((ext_vector_type(2)));
typedef float float2v_t __attribute__((ext_vector_type(2)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float wn_f4 __attribute__((ext_vector_type(4)));
typedef uint32_t wn_u4 __attribute__((ext_vector_type(4)));
typedef uint32_t wn_u2 __attribute__((ext_vector_type(2)));

#define WN_ITW 34
#define WN_HPIX (6 * WN_ITW)                          // 204 halo pixels
#define WN_HALO_BYTES (WN_HPIX * 256)                 // 52 224
#define WN_XCH_OFF (2 * WN_HALO_BYTES)                // 104 448
#define WN_XCH_BYTES (4 * 3 * 4 * 1024)               // [source wave][destination among the other three][slot = 2 b + q][lane] x 16 B
#define WN_BIAS_OFF (WN_XCH_OFF + WN_XCH_BYTES)       // 153 600: the 64 biases of the workgroup's output channels
#define WN_LUT_OFF (WN_BIAS_OFF + 256)                // FUSE1A: the u8 -> (xh | xl << 16) table, 1 KiB
#define WN_W1A_OFF (WN_LUT_OFF + 1024)                // FUSE1A: conv1a's four A fragments, 4 KiB
#define WN_ZERO_OFF (WN_W1A_OFF + 4096)               // 1 KiB of zeros: what a wave reads in place of its own terms of the exchange; FUSE1A: the taps of a halo pixel outside the image
#define WN_PATCH_OFF (WN_ZERO_OFF + 1024)             // FUSE1A: 4 waves x 256 B, the wave's 5-row x 40-byte patch of the u8 image
#define WN_SMEM (WN_PATCH_OFF + 1024)                 // 161 024 of 163 840
#define WN_ACT_SCALE 32.0f                            // = SPL_ACT_SCALE (conv_split.hip): raw-32 and split-64 frames hold activations x 32
#define WN_ACT_CLAMP 16000.0f                         // |stored activation|: four of them must sum inside fp16's range

static inline uint16_t wn_f2h_bits(float v) { const __half h = __float2half_rn(v); uint16_t u; memcpy(&u, &h, 2); return u; }
static inline float wn_h2f(uint16_t u) { __half h; memcpy(&h, &u, 2); return __half2float(h); }

// OIHW fp32 (3x3, cin = 64) -> the transformed, split A fragments [cg = cout / 64][i][hl][j][kg][m][lane][8 halfs] of U(i, j) = (G g G^T)(i, j) * 2^k:
//   cout = 64 cg + (32 m + (lane & 31) + 16 i) mod 64, cin = 16 kg + 8 (lane >> 5) + e (the operand order of v_mfma_f32_32x32x16_f16); k: max |U| 2^k in [256, 512).
// Returns 2^-k.  out: cin * cout * 16 * 2 halfs.
float conv_pack_weights_wino(const float* w, int cin, int cout, uint16_t* out) {
    static const double G[4][3] = {{1, 0, 0}, {.5, .5, .5}, {.5, -.5, .5}, {0, 0, 1}};
    std::vector<float> U((size_t)16 * cout * cin);
    double mx = 0;
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci) {
            const float* g = w + ((size_t)co * cin + ci) * 9;
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) {
                    double s = 0;
                    for (int k = 0; k < 3; ++k)
                        for (int l = 0; l < 3; ++l) s += G[i][k] * (double)g[k * 3 + l] * G[j][l];
                    U[((size_t)(i * 4 + j) * cout + co) * cin + ci] = (float)s;
                    if (fabs(s) > mx) mx = fabs(s);
                }
        }
    int ex = 0;
    if (mx > 0) (void)frexp(mx, &ex);
    const int k = 9 - ex;
    size_t o = 0;
    for (int cg = 0; cg < cout / 64; ++cg)
        for (int i = 0; i < 4; ++i)
            for (int hl = 0; hl < 2; ++hl)
                for (int j = 0; j < 4; ++j)
                    for (int kg = 0; kg < cin / 16; ++kg)
                        for (int m = 0; m < 2; ++m)
                            for (int l = 0; l < 64; ++l)
                                for (int e = 0; e < 8; ++e) {
                                    const int co = cg * 64 + ((m * 32 + (l & 31) + 16 * i) & 63), ci = kg * 16 + (l >> 5) * 8 + e;      // (rotated by wave: see the exchange)
                                    const float v = ldexpf(U[((size_t)(i * 4 + j) * cout + co) * cin + ci], k);
                                    const uint16_t hi = wn_f2h_bits(v);
                                    out[o++] = hl ? wn_f2h_bits(v - wn_h2f(hi)) : hi;
                                }
    return ldexpf(1.f, -k);
}

// the tiles of an image that run (ConvArgs::skip_*), as SplSkip of conv_split.hip: the 4 x 32 tile grid is the direct kernel's
struct WnSkip { int act, n_above, n_upto, y0, y1, x0, w, bw; uint32_t magic_tx, magic_bw; int xcd; };
struct WnTileIx { int b, r, ty, tx; };
struct WnFuse {
    const uint8_t* gray = nullptr; int gstride = 0, mask_r0 = 0, mask_r1 = 0;
    const _Float16* w1a_frag = nullptr;   // conv1a_split_pack_fused: [2 k-halves][2 m][64 lanes][8 halfs], weights and bias x the activation scale
    const uint32_t* lut_hl = nullptr;     // conv1a_make_split_lut
};

__device__ __forceinline__ float wn_max(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, __builtin_inff()); }      // no canonicalisation of the operands
template <int J, int N, typename F>
__device__ __forceinline__ void wn_for_each(F&& f) {
    if constexpr (J < N) { f(std::integral_constant<int, J>{}); wn_for_each<J + 1, N>(f); }
}
// LDS accesses by 32-bit LDS address: the workgroup's dynamic LDS starts at address 0 (no static LDS in this kernel; checked once by the launcher's
// first block through `lds_base_is_zero`), so an offset IS the address and the instruction's immediate offset field takes the constant part
typedef __attribute__((address_space(3))) wn_f4 wn_lds_f4;
__device__ __forceinline__ wn_f4 wn_lds_ld(const char*, uint32_t off) { return *reinterpret_cast<const wn_lds_f4*>((uintptr_t)off); }
__device__ __forceinline__ void wn_lds_st(char*, uint32_t off, wn_f4 v) { *reinterpret_cast<wn_lds_f4*>((uintptr_t)off) = v; }

// hi = half(v), lo = half(v - hi) of eight values: four v_cvt_pk_f16_f32 and eight v_fma_mix{lo,hi}_f16 (x - float(hi) is exact in f32: one rounding)
__device__ __forceinline__ void wn_split8(const wn_f4& a, const wn_f4& b, half8_t& hi, half8_t& lo) {
    uint32_t dh[4], dl[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const float x0 = p < 2 ? a[2 * p] : b[2 * p - 4], x1 = p < 2 ? a[2 * p + 1] : b[2 * p - 3];
        float2v_t fv; fv[0] = x0; fv[1] = x1;
        dh[p] = __builtin_bit_cast(uint32_t, __builtin_convertvector(fv, half2v_t));
        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(dl[p]) : "v"(dh[p]), "v"(x0));
        asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(dl[p]) : "v"(dh[p]), "v"(x1));
    }
    hi = __builtin_bit_cast(half8_t, wn_u4{dh[0], dh[1], dh[2], dh[3]});
    lo = __builtin_bit_cast(half8_t, wn_u4{dl[0], dl[1], dl[2], dl[3]});
}

// POOL: 2 x 2 max-pool behind the ReLU (one Winograd tile = one pooled pixel).  OUT_SPLIT: the output frame is split-64 (next layer: a direct kernel of
// conv_split.hip), else raw-32.  FUSE1A (conv1b): the halo is built from the u8 image -- conv1a (1 -> 64 channels, 3x3, + ReLU) on the matrix cores with
// split operands, the scheme and the packed constants of conv_split.hip's FUSE1A -- instead of being DMA'd from a conv1a tensor.
template <bool POOL, bool OUT_SPLIT, bool FUSE1A>
__global__ void __launch_bounds__(256, 1)
conv3x3_wino_kernel(const char* __restrict__ in, char* __restrict__ out, const _Float16* __restrict__ wp, const float* __restrict__ bias,
                    float inv, int H, int W, int cout, int n_cg, int tiles_x, int tiles_y, int batch, int relu, WnSkip sk, WnFuse fz,
                    unsigned long long* trace /* OMNI_WINO_TRACE=1: s_memtime stamps of workgroup 0, waves 0 and 3 (debug), else nullptr */) {
    extern __shared__ __attribute__((aligned(256))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem != 0u) __builtin_trap();      // (wn_lds_ld / wn_lds_st address LDS from 0)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // = i, the row of the transformed patch this wave owns
    const int n = lane & 31, hh = lane >> 5, trow = n >> 4, tcol = n & 15;
    const int bid = xcd_block_id(sk.xcd);
    const int cg = bid % n_cg, wg = bid / n_cg, nwg = gridDim.x / n_cg;
    const int tiles_per_img = sk.act;
    const int total = batch * tiles_per_img;
    // row i of B^T: W = d[ra] + beta d[rb]
    const int ra = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
    const int rb = wave == 3 ? 3 : (wave == 2 ? 1 : 2);
    const float beta = wave == 1 ? 1.f : -1.f;

    half8_t wreg[64];                                                    // [hl][j][kg][m]
    {
        const _Float16* wbase = wp + ((size_t)(cg * 4 + wave) * 64) * 512 + lane * 8;
#pragma unroll
        for (int s = 0; s < 64; ++s) {
            wreg[s] = *reinterpret_cast<const half8_t*>(wbase + s * 512);
            // ... in the ACCUMULATOR half of the register file, where the MFMAs read them in place: the 256 architectural registers are the accumulators' (128:
            // the output transform reads them with plain VALU instructions) and the transforms' (left to itself hipcc puts the accumulators there, parks the weights
            // in what is left of it and copies four registers in front of every MFMA)
            asm volatile("" : "+a"(wreg[s]));
        }
    }
    if (tid < 64) reinterpret_cast<float*>(smem + WN_BIAS_OFF)[tid] = bias[cg * 64 + tid];
    if (tid < 64) reinterpret_cast<uint4*>(smem + WN_ZERO_OFF)[tid] = make_uint4(0, 0, 0, 0);
    reinterpret_cast<uint32_t*>(smem + WN_PATCH_OFF)[tid] = 0u;
    if constexpr (FUSE1A) {
        reinterpret_cast<uint32_t*>(smem + WN_LUT_OFF)[tid] = fz.lut_hl[tid];
        reinterpret_cast<uint4*>(smem + WN_W1A_OFF)[tid] = reinterpret_cast<const uint4*>(fz.w1a_frag)[tid];
    }

    // ---- the tile walk (conv_split.hip's) ----------------------------------------------------------------------------------------------------
    const int step_b = nwg / tiles_per_img, step_r = nwg - step_b * tiles_per_img;
    auto decode = [&](WnTileIx& q) {
        int r = q.r, ty, tx;
        if (r < sk.n_above || r >= sk.n_upto) {
            int base = 0;
            if (r >= sk.n_upto) { r -= sk.n_upto; base = sk.y1; }
            const int ry = sk.magic_tx ? (int)__umulhi((uint32_t)r, sk.magic_tx) : r;
            tx = r - ry * tiles_x; ty = ry + base;
        } else {
            r -= sk.n_above;
            const int qy = sk.magic_bw ? (int)__umulhi((uint32_t)r, sk.magic_bw) : r;
            const int c = r - qy * sk.bw;
            ty = sk.y0 + qy; tx = c < sk.x0 ? c : c + sk.w;
        }
        q.ty = ty; q.tx = tx;
    };
    auto advance = [&](WnTileIx& q) {
        q.r += step_r;
        if (q.r >= tiles_per_img) { q.r -= tiles_per_img; ++q.b; }
        q.b += step_b;
        decode(q);
    };

    // ---- the halo in LDS: pixel (R, X) at (R * 34 + X) * 256, its 16-byte chunk c (channels 4 c .. 4 c + 3) in slot c ^ ((X >> 1) & 15) -----------
    const int Wf = split_frame_w(W), Hf = split_frame_h(H);
    const uint32_t in_img_bytes = (uint32_t)Hf * Wf * 256u;
    // DMA piece p (1 KiB) = halo pixels [4 p, 4 p + 4): lane -> (pixel 4 p + lane / 16, slot lane % 16)
    constexpr int NPIECES = 51, PPW = 13;
    uint32_t goff[PPW];
    if constexpr (!FUSE1A) {
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            int piece = wave * PPW + j;
            piece = piece < NPIECES ? piece : NPIECES - 1;
            const int hp = piece * 4 + (lane >> 4), slot = lane & 15;
            const int R = hp / WN_ITW, X = hp - R * WN_ITW;
            goff[j] = (uint32_t)(R * Wf + X) * 256u + (uint32_t)((slot ^ ((X >> 1) & 15)) << 4);
        }
    }
    struct Org { __amdgpu_buffer_rsrc_t r; uint32_t soff; };
    auto origin = [&](int qb, int qty, int qtx) -> Org {
        Org o;
        o.r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(in) + (int64_t)qb * in_img_bytes - 4096, 0, in_img_bytes + 8192, 0x00020000);
        o.soff = (uint32_t)(qty * 4 * Wf + qtx * 32) * 256u + 4096u;      // halo origin (4 ty - 1, 32 tx - 1) = frame pixel (4 ty, 32 tx)
        return o;
    };
    auto dma_piece = [&](const Org& o, int which, auto JC) {
        constexpr int j = decltype(JC)::value, g = j / 4;
        [[maybe_unused]] constexpr int k = j % 4;
        int first = wave * PPW + 4 * g;
        first = first < NPIECES ? first : NPIECES - 1;
#if __HIP_DEVICE_COMPILE__
        __builtin_amdgcn_raw_ptr_buffer_load_lds(o.r, (__attribute__((address_space(3))) void*)(smem + which * WN_HALO_BYTES + first * 1024), 16,
                                                 goff[j], o.soff - k * 1024, k * 1024, 0);
#else
        (void)o; (void)which; (void)first;
#endif
    };
    auto dma_tile = [&](int qb, int qty, int qtx, int which) {
        const Org o = origin(qb, qty, qtx);
        wn_for_each<0, PPW>([&](auto JC) { dma_piece(o, which, JC); });
    };

    // FUSE1A: wave w builds halo pixels [64 w, 64 w + 64) (two 32-pixel fragments; pixels >= 204 are nobody's) of a tile into halo buffer `which`: conv1a on
    // the matrix cores with split operands, conv_split.hip's scheme and packed constants.  Those pixels lie in at most 3 halo rows = a 5-row x 40-byte patch
    // of the image: ONE buffer_load_dword per lane, issued a whole tile ahead (patch_issue: the stream of tile t loads the patch of tile t + 2), parked --
    // rows / columns outside the image and the fisheye mask's rows zeroed -- in a wave-private LDS slot (patch_park), from which a lane reads the bytes under
    // ITS taps (lanes 0-31: taps 0-4, lanes 32-63: taps 5-8 and the bias slots) and then their table entries (xh | xl << 16), which ARE the MFMA's operand dwords.
    [[maybe_unused]] const int fz_r0 = (64 * wave) / WN_ITW;                        // first halo row the wave's pixels touch
    [[maybe_unused]] const int fz_pj = lane / 10 + fz_r0 - 2, fz_pd = 4 * (lane % 10) - 4;       // the lane's patch dword: image row 4 ty + fz_pj, columns 32 tx + fz_pd ..
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t fz_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(fz.gray), 0, 0x7fffffff, 0x00020000);
    uint32_t pv = 0;                                                               // the patch dword in flight
    auto patch_issue = [&](int qb, int qty, int qtx) {
        int yy = qty * 4 + fz_pj, xx = qtx * 32 + fz_pd;
        yy = yy < 0 ? 0 : (yy > H - 1 ? H - 1 : yy);
        xx = xx < 0 ? 0 : (xx > fz.gstride - 4 ? fz.gstride - 4 : xx);            // (clamped to valid addresses here, zeroed when parked)
        pv = __builtin_amdgcn_raw_buffer_load_b32(fz_rsrc, (qb * H + yy) * fz.gstride + xx, 0, 0);
    };
    auto patch_park = [&](int qty, int qtx) {
        const int y = qty * 4 + fz_pj, x = qtx * 32 + fz_pd;
        const bool ok = ((unsigned)y < (unsigned)H) & ((unsigned)(y - fz.mask_r0) >= (unsigned)(fz.mask_r1 - fz.mask_r0)) & (x >= 0) & (x < W);
        const int left = W - x;                                                    // bytes of the dword inside the row (W need not be a multiple of 4)
        const uint32_t keep = left >= 4 ? 0xffffffffu : (1u << (8 * (left & 3))) - 1u;
        reinterpret_cast<uint32_t*>(smem + WN_PATCH_OFF + wave * 256)[lane] = ok ? (pv & keep) : 0u;
    };
    // (all four (fragment, channel fragment) chains side by side: bytes -> table entries -> two dependent MFMAs -> ReLU -> four 16-byte LDS stores are
    // dependent LDS / matrix-core round trips; run at the HEAD of a tile's stream, where the accumulators are dead and the registers are there)
    auto build_tile = [&](int qty, int qtx, int which) {
        const uint32_t* lut = reinterpret_cast<const uint32_t*>(smem + WN_LUT_OFF);
        const int ty0 = qty * 4, tx0 = qtx * 32;
        uint32_t U[2][5], dstb[2];
        bool valid[2], inside[2];
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            // MFMA column n <-> pixel 2 (n & 15) + (n >> 4) of the fragment: the eight lanes of a ds_write_b128 phase then store pixels two columns apart, whose
            // swizzles (X >> 1) differ (pixels n, n + 1 share one: a two-way bank conflict on every store)
            const int vp = 64 * wave + 32 * f + 2 * (n & 15) + (n >> 4);
            const int R = vp / WN_ITW, X = vp - R * WN_ITW;
            const int gy = ty0 - 1 + R, gx = tx0 - 1 + X;                           // the image pixel under this halo pixel
            inside[f] = vp < WN_HPIX;
            valid[f] = inside[f] & ((unsigned)gy < (unsigned)H) & ((unsigned)gx < (unsigned)W);
            // the byte under tap (0, 0): patch row R - fz_r0, patch column X + 2 (patch column 0 = image column 32 tx - 4); a pixel outside the image is conv1b's
            // zero padding: its taps read zeros, its bias slots are zero
            const uint32_t pb = valid[f] ? (uint32_t)(WN_PATCH_OFF + wave * 256 + (R - fz_r0) * 40 + X + 2) : (uint32_t)WN_ZERO_OFF;
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const int t1 = k < 4 ? 5 + k : 8;
                const int c0 = (k / 3) * 40 + k % 3, c1 = (t1 / 3) * 40 + t1 % 3;
                U[f][k] = *reinterpret_cast<const uint8_t*>(smem + pb + (hh ? c1 : c0));
            }
            // chunk c = 8 m + 2 g + hh of the pixel goes to slot c ^ swizzle: (base | ((swizzle ^ hh) << 4)) ^ ((8 m + 2 g) << 4), one v_xor per store -- and NOT sixteen
            // loop-invariant addresses hoisted out of the tile loop into registers the stream has no room for (the empty asm pins the base inside the loop)
            dstb[f] = (uint32_t)(which * WN_HALO_BYTES + vp * 256) + (uint32_t)(((((X >> 1) & 15) ^ hh)) << 4);
            asm volatile("" : "+v"(dstb[f]));
        }
        half8_t wa[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) wa[i] = *reinterpret_cast<const half8_t*>(smem + WN_W1A_OFF + i * 1024 + lane * 16);
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int k = 0; k < 5; ++k) U[f][k] = lut[U[f][k]];
        floatx16 a[2][2];
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            const uint32_t h01 = __builtin_amdgcn_perm(U[f][1], U[f][0], 0x05040100u), h23 = __builtin_amdgcn_perm(U[f][3], U[f][2], 0x05040100u);
            const uint32_t x3 = hh ? (valid[f] ? 0x3C003C00u : 0u) : U[f][4];       // lanes 32-63: the bias slots (1.0, 1.0)
            const half8_t B0 = __builtin_bit_cast(half8_t, wn_u4{U[f][0], U[f][1], U[f][2], U[f][3]});
            const half8_t B1 = __builtin_bit_cast(half8_t, wn_u4{U[f][4], h01, h23, x3});
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                floatx16 z;
#pragma unroll
                for (int q = 0; q < 16; ++q) z[q] = 0.f;
                a[f][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[m], B0, z, 0, 0, 0);
                a[f][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[2 + m], B1, a[f][m], 0, 0, 0);
            }
        }
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            if (inside[f]) {
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        wn_f4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = __builtin_amdgcn_fmed3f(a[f][m][4 * g + e], 0.f, WN_ACT_CLAMP);      // (fmaxf: a canonicalising v_max in front of the v_max)
                        wn_lds_st(smem, dstb[f] ^ (uint32_t)((m * 8 + g * 2) << 4), v);    // channels 32 m + 8 g + 4 hh + (0..3)
                    }
            }
        }
    };

    // this lane's read addresses inside a halo buffer: [column pair p][16-byte half q of its 8 channels], rows ra and rb; the k-group toggles bits 6-7
    uint32_t adA[2][2];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int q = 0; q < 2; ++q)
            adA[p][q] = (uint32_t)(((2 * trow + ra) * WN_ITW + 2 * tcol + 2 * p) * 256) + (uint32_t)(((hh * 2 + q) ^ ((tcol + p) & 15)) << 4);
    const int dAB = (rb - ra) * WN_ITW * 256;                                      // row rb's address = row ra's + this (wave-uniform)

    // output addressing
    const int Ho = POOL ? (H >> 1) : H, Wo = POOL ? (W >> 1) : W;
    const int Wof = split_frame_w(Wo);
    const int64_t opix = (int64_t)cout * 4;                                        // raw-32 and split-64: 4 bytes per channel
    const int64_t out_img_bytes = (int64_t)split_frame_h(Ho) * Wof * opix;
    // the channels this wave finishes: 16 wave + 8 q + 4 hh + (0..3) of the workgroup's 64

    // the exchange: where this wave writes the quarter of wave (wave + k) & 3 and where it reads source s's terms for its own quarter (s = itself: zeros)
    uint32_t xw[3], xr[4][4];                                                      // xr[s][slot] (the wave's own s: the 1 KiB of zeros, whatever the slot)
#pragma unroll
    for (int k = 1; k < 4; ++k) { const int d = (wave + k) & 3; xw[k - 1] = (uint32_t)(WN_XCH_OFF + (3 * wave + (d < wave ? d : d - 1)) * 4096); }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) xr[s][sl] = s == wave ? (uint32_t)WN_ZERO_OFF : (uint32_t)(WN_XCH_OFF + (3 * s + (wave < s ? wave : wave - 1)) * 4096 + sl * 1024);
    const float c0 = wave == 3 ? 0.f : 1.f, c1 = wave == 0 ? 0.f : (wave == 1 ? 1.f : -1.f);        // column `wave` of A^T
    int t = wg;
    WnTileIx cur_ix, nxt_ix;
    cur_ix.b = t / tiles_per_img;
    cur_ix.r = t - cur_ix.b * tiles_per_img;
    decode(cur_ix);
    nxt_ix = cur_ix;
    advance(nxt_ix);
    __syncthreads();                                                               // bias (and FUSE1A's table and fragments) in LDS
    if (t < total) {                                                               // the workgroup's first tile: nothing to hide behind
        if constexpr (FUSE1A) {
            patch_issue(cur_ix.b, cur_ix.ty, cur_ix.tx);
            patch_park(cur_ix.ty, cur_ix.tx);
            build_tile(cur_ix.ty, cur_ix.tx, 0);
            const bool n1 = t + nwg < total;                                       // the patch of the tile the first stream builds
            patch_issue(n1 ? nxt_ix.b : cur_ix.b, n1 ? nxt_ix.ty : cur_ix.ty, n1 ? nxt_ix.tx : cur_ix.tx);
        } else {
            dma_tile(cur_ix.b, cur_ix.ty, cur_ix.tx, 0);
        }
    }
    if constexpr (FUSE1A) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();

    const bool tr = trace != nullptr && blockIdx.x == 0 && (tid == 0 || tid == 192);
    int tk = -2;                                   // the trace skips the first two tiles
    auto stamp = [&](int i) {
        if (tr && tk >= 0 && tk < 4) {
            const unsigned long long v = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            trace[(tid ? 32 : 0) + tk * 8 + i] = v;
        }
    };
    // ---- the pending finish of a tile: its own T' terms and where its outputs go ----------------------------------------------------------------
    wn_f4 own[2][2];                                                               // [b][q]
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int q = 0; q < 2; ++q) own[b][q] = wn_f4{0.f, 0.f, 0.f, 0.f};
    int pend = 0, p_b = 0, p_ty = 0, p_tx = 0;
    // y(a, b) = sum_s A^T(a, s) T'_s(b) = (T0 + T1) + T2 and (T1 - T2) - T3 over the OTHER three waves' terms (the wave's own slot reads zeros), its own term
    // added last with its coefficient: a channel is always finished by the same wave, in the same order.  Then bias, ReLU, (pool), store: tile (trow, tcol)
    // of the 2 x 16 = output pixels (4 ty + 2 trow + a, 32 tx + 2 tcol + b)
    wn_f4 xt[2][4];                                                                // [b][s]: the four waves' terms of one quad of the pending tile (the wave's own slot: zeros)
    auto xch_issue = [&](int q) {                                                  // quad 0: at the head of the next tile's stream; quad 1: behind quad 0's finish
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int s = 0; s < 4; ++s) xt[b][s] = wn_lds_ld(smem, xr[s][2 * b + q] + lane * 16);
    };
    auto finish = [&](int q) {
        const int oy = p_ty * 4 + 2 * trow, ox = p_tx * 32 + 2 * tcol;
        // the output image as a raw buffer: a lane with nothing to store carries an offset outside it (the store is dropped: no branch in the stream)
        const __amdgpu_buffer_rsrc_t oimg = __builtin_amdgcn_make_buffer_rsrc(out + (int64_t)p_b * out_img_bytes, 0, (int)out_img_bytes, 0x00020000);
        const float lo_lim = relu ? 0.f : -WN_ACT_CLAMP;
        const int ch0 = 16 * wave + 4 * hh;                                        // + 8 q: the channel inside the workgroup's 64-channel block cg
        const uint32_t chan_off = OUT_SPLIT ? (uint32_t)(cg * 256 + ch0 * 2) : (uint32_t)((cg * 64 + ch0) * 4);
        auto pix_off = [&](int y, int x, bool ok) -> uint32_t {                   // output pixel (y, x) of the (pooled) map
            return (ok & (pend != 0)) ? (uint32_t)((y + 1) * Wof + (x + 1)) * (uint32_t)opix + chan_off : 0x80000000u;
        };
        auto store = [&](uint32_t off, int q, const wn_f4& raw) {
            wn_f4 v;
            const float4 bb = *reinterpret_cast<const float4*>(smem + WN_BIAS_OFF + (16 * wave + 8 * q + 4 * hh) * 4);
            v[0] = __builtin_amdgcn_fmed3f(fmaf(raw[0], inv, bb.x), lo_lim, WN_ACT_CLAMP);
            v[1] = __builtin_amdgcn_fmed3f(fmaf(raw[1], inv, bb.y), lo_lim, WN_ACT_CLAMP);
            v[2] = __builtin_amdgcn_fmed3f(fmaf(raw[2], inv, bb.z), lo_lim, WN_ACT_CLAMP);
            v[3] = __builtin_amdgcn_fmed3f(fmaf(raw[3], inv, bb.w), lo_lim, WN_ACT_CLAMP);
            if constexpr (OUT_SPLIT) {
                float2v_t f0, f1; f0[0] = v[0]; f0[1] = v[1]; f1[0] = v[2]; f1[1] = v[3];
                const uint32_t h0 = __builtin_bit_cast(uint32_t, __builtin_convertvector(f0, half2v_t));
                const uint32_t h1 = __builtin_bit_cast(uint32_t, __builtin_convertvector(f1, half2v_t));
                uint32_t l0, l1;
                asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(h0), "v"(v[0]));
                asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(h1), "v"(v[2]));
                asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l0) : "v"(h0), "v"(v[1]));
                asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l1) : "v"(h1), "v"(v[3]));
                __builtin_amdgcn_raw_buffer_store_b64(wn_u2{h0, h1}, oimg, off + q * 16, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b64(wn_u2{l0, l1}, oimg, off + q * 16 + 128, 0, 0);
            } else {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(wn_u4, v), oimg, off + q * 32, 0, 0);
            }
        };
        uint32_t off[POOL ? 1 : 4];
        if constexpr (POOL) off[0] = pix_off(oy >> 1, ox >> 1, (oy < H) & (ox < W));
        else {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) off[2 * a + b] = pix_off(oy + a, ox + b, (oy + a < H) & (ox + b < W));
        }
        {
            wn_f4 pmax;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                wn_f4 y0, y1;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    y0[e] = fmaf(own[b][q][e], c0, (xt[b][0][e] + xt[b][1][e]) + xt[b][2][e]);
                    y1[e] = fmaf(own[b][q][e], c1, (xt[b][1][e] - xt[b][2][e]) - xt[b][3][e]);
                }
                if constexpr (POOL) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) pmax[e] = b == 0 ? wn_max(y0[e], y1[e]) : wn_max(pmax[e], wn_max(y0[e], y1[e]));
                } else {
                    store(off[b], q, y0);
                    store(off[2 + b], q, y1);
                }
            }
            if constexpr (POOL) store(off[0], q, pmax);
        }
    };

    // ---- the stream: 16 steps x = 4 kg + j of 6 MFMAs (three terms x two channel fragments), software-pipelined by hand: region r issues the MFMAs of step r
    // and, between them, the transform of step r + 1 (column sums, the position's difference, the (hi, lo) split) and the LDS reads of step r + 2 -- plus, in
    // the regions that can afford the registers (accumulators j = 2, 3 are first written in regions 2, 3), the previous tile's finish (region 0) and the next
    // tile's halo (DMA pieces one per region; FUSE1A: byte loads in region 0, the two fragments in regions 1 and 2).  One wave per SIMD and in-order issue:
    // whatever is not placed between two MFMAs idles the matrix cores.
    int cur = 0;
    uint32_t ax[2][2][2];                                                          // [row a / b][column pair][q]: this tile's, this k-group's read addresses
    wn_f4 rA[4][2], rB[4][2];                                                      // landing registers of column cc, rows ra and rb
    wn_f4 Wc[4][2];                                                                // the column sums
    half8_t vh[2], vl[2];                                                          // the B operands of step x in buffer x & 1
    auto loads = [&](auto XC) {
        constexpr int x = decltype(XC)::value, kg = x >> 2, j = x & 3;
        if constexpr (x < 16 && j != 2) {
            if constexpr (j == 0 && kg > 0) {
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int q = 0; q < 2; ++q) { ax[0][p][q] ^= (uint32_t)(((kg - 1) ^ kg) << 6); ax[1][p][q] ^= (uint32_t)(((kg - 1) ^ kg) << 6); }
            }
            auto col = [&](int cc) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    rA[cc][q] = wn_lds_ld(smem, ax[0][cc >> 1][q] + (cc & 1) * 256);
                    rB[cc][q] = wn_lds_ld(smem, ax[1][cc >> 1][q] + (cc & 1) * 256);
                }
            };
            if constexpr (j == 0) { col(0); col(2); } else if constexpr (j == 1) col(1); else col(3);
        }
    };
    auto vsum = [&](int cc) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) Wc[cc][q][e] = fmaf(beta, rB[cc][q][e], rA[cc][q][e]);
    };
    wn_f4 Vt[2];
    uint32_t dh[4], dl[4];
    // the transform of step x in five parts (one per gap between two MFMAs of the region before)
    auto tpart = [&](auto XC, auto GC) {
        constexpr int x = decltype(XC)::value, j = x & 3, G = decltype(GC)::value;
        if constexpr (x < 16) {
            if constexpr (G == 0) { if constexpr (j == 0) vsum(0); else if constexpr (j == 1) vsum(1); else if constexpr (j == 3) vsum(3); }
            else if constexpr (G == 1) { if constexpr (j == 0) vsum(2); }
            else if constexpr (G == 2) {
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        Vt[q][e] = j == 0 ? Wc[0][q][e] - Wc[2][q][e] : j == 1 ? Wc[1][q][e] + Wc[2][q][e] : j == 2 ? Wc[2][q][e] - Wc[1][q][e] : Wc[1][q][e] - Wc[3][q][e];
            } else if constexpr (G == 3) {
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    float2v_t fv; fv[0] = Vt[p >> 1][2 * (p & 1)]; fv[1] = Vt[p >> 1][2 * (p & 1) + 1];
                    dh[p] = __builtin_bit_cast(uint32_t, __builtin_convertvector(fv, half2v_t));
                }
#pragma unroll
                for (int p = 0; p < 4; ++p) asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(dl[p]) : "v"(dh[p]), "v"(Vt[p >> 1][2 * (p & 1)]));
            } else {
#pragma unroll
                for (int p = 0; p < 4; ++p) asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(dl[p]) : "v"(dh[p]), "v"(Vt[p >> 1][2 * (p & 1) + 1]));
                vh[x & 1] = __builtin_bit_cast(half8_t, wn_u4{dh[0], dh[1], dh[2], dh[3]});
                vl[x & 1] = __builtin_bit_cast(half8_t, wn_u4{dl[0], dl[1], dl[2], dl[3]});
            }
        }
    };
    // the exchange writes of T'(b) for the quarter of wave (wave + k) & 3: register groups 2 (k & 1) + q of channel fragment k >> 1 (see the rotation below)
    auto xwrite = [&](floatx16 (&acc)[4][2], auto BC, auto KC) {
        constexpr int b = decltype(BC)::value, k = decltype(KC)::value;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            constexpr int m = k >> 1;
            const int g = 2 * (k & 1) + q;
            wn_lds_st(smem, xw[k - 1] + (uint32_t)((2 * b + q) * 1024) + lane * 16, wn_f4{acc[b][m][4 * g], acc[b][m][4 * g + 1], acc[b][m][4 * g + 2], acc[b][m][4 * g + 3]});
        }
    };
    // a tile's addresses (k-group 0) and the reads of its steps 0 and 1 from halo buffer `which` (valid once the barrier behind its fill has been passed)
    auto head_loads = [&](int which) {
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int q = 0; q < 2; ++q) { ax[0][p][q] = adA[p][q] + (uint32_t)(which * WN_HALO_BYTES); ax[1][p][q] = ax[0][p][q] + (uint32_t)dAB; }
        loads(std::integral_constant<int, 0>{});
        loads(std::integral_constant<int, 1>{});
    };
    WnTileIx fill_ix = cur_ix;                                                     // the tile whose halo this stream fills (the next one; the last stream: its own again)
    head_loads(0);
    for (; t < total; t += nwg, cur ^= 1) {
        stamp(0);
        const bool has_next = t + nwg < total;
        fill_ix.b = has_next ? nxt_ix.b : cur_ix.b; fill_ix.ty = has_next ? nxt_ix.ty : cur_ix.ty; fill_ix.tx = has_next ? nxt_ix.tx : cur_ix.tx;
        [[maybe_unused]] const Org org_n = origin(fill_ix.b, fill_ix.ty, fill_ix.tx);
        [[maybe_unused]] WnTileIx nn_ix = nxt_ix;                                  // FUSE1A: the tile after next, whose patch this stream loads
        if constexpr (FUSE1A) {
            advance(nn_ix);
            if (!(t + 2 * nwg < total)) nn_ix = fill_ix;                           // (none: any valid patch, nobody parks it)
        }
        // the stream's head: the reads of steps 0 and 1 were issued inside the previous tile's last region (head_loads), the exchange's are issued here
        xch_issue(0);
        if constexpr (FUSE1A) { patch_park(fill_ix.ty, fill_ix.tx); patch_issue(nn_ix.b, nn_ix.ty, nn_ix.tx); build_tile(fill_ix.ty, fill_ix.tx, cur ^ 1); }
        wn_for_each<0, 5>([&](auto GC) { tpart(std::integral_constant<int, 0>{}, GC); });
        stamp(1);
        floatx16 acc[4][2];
#ifdef WN_STEP_TRACE
        unsigned long long ts[17];
#endif
        __builtin_amdgcn_sched_barrier(0);
        wn_for_each<0, 16>([&](auto RC) {
            constexpr int r = decltype(RC)::value, kg = r >> 2, j = r & 3, s = (j * 4 + kg) * 2;
            constexpr auto X1 = std::integral_constant<int, r + 1>{};
            floatx16 zero;
#pragma unroll
            for (int q = 0; q < 16; ++q) zero[q] = 0.f;
#ifdef WN_STEP_TRACE      // separate build (tools/round6/steptrace.sh): s_memtime at the top of every region (the values are read after the stream)
            asm volatile("s_memtime %0" : "=s"(ts[r]));
#endif
            if constexpr (r == 15) {
                // the barrier in front of the last region: every wave's share of the next tile's halo has landed (DMA issued by region 7; FUSE1A: built at the head)
                // -- its first reads go out under this region's MFMAs --, and every wave has read the previous tile's exchange (regions 0): it is rewritten below
                if constexpr (FUSE1A) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            acc[j][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[s], vh[r & 1], kg == 0 ? zero : acc[j][0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (r == 0) { finish(0); xch_issue(1); }
            tpart(X1, std::integral_constant<int, 0>{});
            if constexpr (r == 15) head_loads(cur ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            acc[j][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[s + 1], vh[r & 1], kg == 0 ? zero : acc[j][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            tpart(X1, std::integral_constant<int, 1>{});
            loads(std::integral_constant<int, r + 2>{});                           // (into the registers the column sums have just freed; read a region later)
            if constexpr (r == 14) {                                               // the output transform along j starts while the last accumulators fill: M0 + M1
#pragma unroll
                for (int m = 0; m < 2; ++m) acc[0][m] = acc[0][m] + acc[1][m];
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[j][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[32 + s], vh[r & 1], acc[j][0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            tpart(X1, std::integral_constant<int, 2>{});
            if constexpr (!FUSE1A && r >= 1 && 2 * (r - 1) < PPW) dma_piece(org_n, cur ^ 1, std::integral_constant<int, 2 * (r - 1)>{});
            if constexpr (r == 15) {                                               // T'(0) = (M0 + M1) + M2
#pragma unroll
                for (int m = 0; m < 2; ++m) acc[0][m] = acc[0][m] + acc[2][m];
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[j][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[32 + s + 1], vh[r & 1], acc[j][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            tpart(X1, std::integral_constant<int, 3>{});
            if constexpr (r == 15) xwrite(acc, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});      // T'(0) of the quarters of waves + 1 ..
            __builtin_amdgcn_sched_barrier(0);
            acc[j][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[s], vl[r & 1], acc[j][0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            tpart(X1, std::integral_constant<int, 4>{});
            if constexpr (r == 15) xwrite(acc, std::integral_constant<int, 0>{}, std::integral_constant<int, 3>{});                             // .. + 3
            if constexpr (r == 0) finish(1);
            if constexpr (!FUSE1A && r >= 1 && 2 * (r - 1) + 1 < PPW) dma_piece(org_n, cur ^ 1, std::integral_constant<int, 2 * (r - 1) + 1>{});
            if constexpr (r == 15) {                                               // M1 - M2 (T'(1) = (M1 - M2) - M3 behind the last MFMA)
#pragma unroll
                for (int m = 0; m < 2; ++m) acc[1][m] = acc[1][m] - acc[2][m];
                xwrite(acc, std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});                           // .. + 2
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[j][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[s + 1], vl[r & 1], acc[j][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        });
#ifdef WN_STEP_TRACE
        asm volatile("s_memtime %0" : "=s"(ts[16]));
        if (tr && tk >= 0 && tk < 4) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int k = 0; k < 17; ++k) trace[64 + (tid ? 68 : 0) + tk * 17 + k] = ts[k];
        }
#endif
        stamp(2);
        // ---- output transform along j (lane-local): T'(b) = sum_j A^T(b, j) M(i, j): T'(0) = (M0 + M1) + M2, T'(1) = (M1 - M2) - M3, one register group
        // (four channels of the lane's tile) at a time, straight into the exchange (or, the wave's own groups, into 16 registers) ---------------------------
#pragma unroll
        for (int m = 0; m < 2; ++m) acc[1][m] = acc[1][m] - acc[3][m];
        xwrite(acc, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
        xwrite(acc, std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{});
        xwrite(acc, std::integral_constant<int, 1>{}, std::integral_constant<int, 3>{});
        // ---- ... along i through LDS.  The packed weights ROTATE the output channels per wave (conv_pack_weights_wino): in wave i, register group g of channel
        // fragment m holds channels 16 ((i + k) & 3) + 8 (g & 1) + 4 hh + (0..3), k = 2 m + g / 2 -- its OWN quarter (k = 0) always in fragment 0, groups 0-1,
        // whatever i is: the code is the same for the four waves (a switch over the wave with the accumulators live sent 100 registers to scratch).
        // Region (source s, destination d) = 3 s + (d < s ? d : d - 1), slot 2 b + q (q = g & 1).
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int q = 0; q < 2; ++q) own[b][q] = wn_f4{acc[b][0][4 * q], acc[b][0][4 * q + 1], acc[b][0][4 * q + 2], acc[b][0][4 * q + 3]};
        pend = 1; p_b = cur_ix.b; p_ty = cur_ix.ty; p_tx = cur_ix.tx;
        cur_ix = nxt_ix;
        advance(nxt_ix);
        stamp(3);
        // the exchange is written (FUSE1A: the patch load in flight is NOT waited for -- and hipcc must not do it either)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        stamp(4);
        ++tk;
    }
    xch_issue(0); finish(0); xch_issue(1); finish(1);                              // the last tile's (nothing pending: every store is dropped)
}

template <bool POOL, bool OUT_SPLIT, bool FUSE1A>
static int launch_wino(hipStream_t st, const ConvArgs& a, const WnFuse& fz) {
    auto kfn = conv3x3_wino_kernel<POOL, OUT_SPLIT, FUSE1A>;
    static DynSmemState smem_state;
    OMNI_HIP_TRY(ensure_dyn_smem(smem_state, (const void*)kfn, WN_SMEM));
    const int tiles_x = cdiv(a.W, 32), tiles_y = cdiv(a.H, 4), n_cg = a.cout / 64;
    const bool skip = a.skip_ty1 > a.skip_ty0 && a.skip_tx1 > a.skip_tx0;
    OMNI_REQUIRE(!skip || (a.skip_ty0 >= 0 && a.skip_ty1 <= tiles_y && a.skip_tx0 >= 0 && a.skip_tx1 <= tiles_x), OMNI_ERR_INVALID, "conv_wino: skip rectangle outside the tile grid");
    WnSkip sk;
    sk.y0 = skip ? a.skip_ty0 : 0; sk.y1 = skip ? a.skip_ty1 : 0; sk.x0 = skip ? a.skip_tx0 : 0; sk.w = skip ? a.skip_tx1 - a.skip_tx0 : 0;
    sk.bw = tiles_x - sk.w;
    sk.act = tiles_x * tiles_y - (sk.y1 - sk.y0) * sk.w;
    sk.n_above = skip ? sk.y0 * tiles_x : sk.act;
    sk.n_upto = sk.n_above + (sk.y1 - sk.y0) * sk.bw;
    OMNI_REQUIRE(sk.act > 0, OMNI_ERR_INVALID, "conv_wino: the skip rectangle covers the whole image");
    auto magic = [](int d) { return d > 1 ? (uint32_t)(((1ull << 32) + (uint64_t)d - 1) / (uint64_t)d) : 0u; };
    sk.magic_tx = magic(tiles_x); sk.magic_bw = magic(sk.bw);
    sk.xcd = config_process()[CFG_CONV_XCD];
    const int total = a.batch * sk.act;
    int per_cg = a.n_cu / n_cg;
    if (per_cg < 1) per_cg = 1;
    if (per_cg > total) per_cg = total;
    static const bool want_trace = config_process()[CFG_WINO_TRACE] != 0;
    static unsigned long long* trace_dev = nullptr;
    if (want_trace) {
        if (!trace_dev) OMNI_HIP_TRY(hipMalloc((void**)&trace_dev, 256 * 8));
        OMNI_HIP_TRY(hipMemsetAsync(trace_dev, 0, 256 * 8, st));
    }
    hipLaunchKernelGGL(kfn, dim3(per_cg * n_cg), dim3(256), WN_SMEM, st, reinterpret_cast<const char*>(a.in), reinterpret_cast<char*>(a.out),
                       reinterpret_cast<const _Float16*>(a.w_packed), a.bias, a.split_inv, a.H, a.W, a.cout, n_cg, tiles_x, tiles_y, a.batch, a.relu ? 1 : 0, sk, fz,
                       want_trace ? trace_dev : nullptr);
    OMNI_LAUNCH_CHECK();
    if (want_trace) {
        unsigned long long h[256];
        OMNI_HIP_TRY(hipMemcpyAsync(h, trace_dev, sizeof(h), hipMemcpyDeviceToHost, st));
        OMNI_HIP_TRY(hipStreamSynchronize(st));
        static int launches = 0;                   // per instantiation: launches 3 and 4 (the first ones run on cold TLBs and caches)
        if (total >= 8 * per_cg && ++launches >= 3 && launches <= 4)
            for (int w = 0; w < 2; ++w)
                for (int k = 0; k < 4; ++k) {
                    const unsigned long long* q = h + w * 32 + k * 8;
#ifdef WN_STEP_TRACE
                    {
                        const unsigned long long* u = h + 64 + w * 68 + k * 17;
                        fprintf(stderr, "wino step trace pool=%d split=%d fuse1a=%d wave %d tile %d:", (int)POOL, (int)OUT_SPLIT, (int)FUSE1A, w * 3, k + 2);
                        for (int r = 0; r < 16; ++r) fprintf(stderr, " %llu", u[r + 1] - u[r]);
                        fprintf(stderr, "\n");
                    }
#endif
                    fprintf(stderr, "wino trace pool=%d split=%d fuse1a=%d H=%d W=%d wave %d tile %d: head %llu stream %llu j-transform+write %llu wait+barrier %llu | total %llu\n",
                            (int)POOL, (int)OUT_SPLIT, (int)FUSE1A, a.H, a.W, w * 3, k + 2, q[1] - q[0], q[2] - q[1], q[3] - q[2], q[4] - q[3], q[4] - q[0]);
                }
    }
    return OMNI_OK;
}

static int wino_check(const ConvArgs& a) {
    OMNI_REQUIRE(a.ksize == 3 && a.cin == 64 && a.cout % 64 == 0, OMNI_ERR_INVALID, "conv_wino: cin=%d cout=%d ksize=%d", a.cin, a.cout, a.ksize);
    OMNI_REQUIRE(a.H % 2 == 0 && a.W % 2 == 0, OMNI_ERR_INVALID, "conv_wino: F(2x2,3x3) tiles need even H, W (%d x %d)", a.H, a.W);
    OMNI_REQUIRE(a.n_cu > 0 && a.split_inv > 0.f, OMNI_ERR_INVALID, "conv_wino: n_cu / split_inv not set");
    OMNI_REQUIRE((int64_t)split_frame_h(a.H) * split_frame_w(a.W) * 256 < (1ll << 32) - 16384, OMNI_ERR_INVALID, "conv_wino: image too large for 32-bit pixel offsets");
    return OMNI_OK;
}

// a.in: raw-32 frames; a.w_packed / a.split_inv from conv_pack_weights_wino; a.bias = act scale x bias; a.out: raw-32 or (out_split) split-64 frames
int conv_wino(hipStream_t st, const ConvArgs& a, bool out_split) {
    int rc;
    if ((rc = wino_check(a))) return rc;
    const WnFuse fz;
    if (a.pool) return out_split ? launch_wino<true, true, false>(st, a, fz) : launch_wino<true, false, false>(st, a, fz);
    return out_split ? launch_wino<false, true, false>(st, a, fz) : launch_wino<false, false, false>(st, a, fz);
}

// conv1a + conv1b + ReLU + 2x2 max-pool in one launch: a = the conv1b layer (a.in unused), the constants of conv1ab_split_fused
int conv1ab_wino_fused(hipStream_t st, const ConvArgs& a, const uint8_t* gray, int gstride, int fisheye_mask, const void* w1a_frag, const uint32_t* lut_hl, bool out_split) {
    int rc;
    if ((rc = wino_check(a))) return rc;
    OMNI_REQUIRE(a.pool && a.cout == 64, OMNI_ERR_INVALID, "conv1ab_wino_fused: conv1b is 64 -> 64 channels, pooled");
    WnFuse fz;
    fz.gray = gray; fz.gstride = gstride;
    if (fisheye_mask) omni_fisheye_mask_rows(a.H, 1, &fz.mask_r0, &fz.mask_r1);
    fz.w1a_frag = reinterpret_cast<const _Float16*>(w1a_frag); fz.lut_hl = lut_hl;
    return out_split ? launch_wino<true, true, true>(st, a, fz) : launch_wino<true, false, true>(st, a, fz);
}

// ---- format converters (test hooks and the mixed configurations of OMNI_SPLIT_WINO) -----------------------------------------------------------------
// split-64 frame -> raw-32 frame (same geometry, out of place): value = float(hi) + float(lo)
__global__ void split_to_raw32_kernel(const uint16_t* __restrict__ in, float* __restrict__ out, int64_t n_blocks64) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // one thread per (pixel block of 64 channels, channel)
    if (i >= n_blocks64 * 64) return;
    const int64_t blk = i >> 6;
    const int c = (int)(i & 63);
    const __half hi = __ushort_as_half(in[blk * 128 + c]), lo = __ushort_as_half(in[blk * 128 + 64 + c]);
    out[i] = __half2float(hi) + __half2float(lo);
}
int split_to_raw32(hipStream_t st, const void* in_split, void* out_raw, int batch, int C, int H, int W) {
    const int64_t nb = (int64_t)batch * split_frame_h(H) * split_frame_w(W) * (C / 64);
    hipLaunchKernelGGL(split_to_raw32_kernel, dim3((unsigned)cdiv64(nb * 64, 256)), dim3(256), 0, st, reinterpret_cast<const uint16_t*>(in_split),
                       reinterpret_cast<float*>(out_raw), nb);
    OMNI_LAUNCH_CHECK();
    return OMNI_OK;
}
// The constant region of the fisheye mask behind an UNPOOLED Winograd layer is constant per position inside the 2 x 2 output tile, not per pixel (the four
// outputs of a tile go through different rows of A^T): the four vectors of the tile at even (y, x) -> vec[2 py + px][pix_bytes], and the rectangle filled with
// them by the parity of the absolute coordinates (conv_read_pixel_bytes / conv_fill_rect_bytes of conv.hip for one vector)
__global__ void read_pixels2x2_kernel(const uint4* __restrict__ map, int64_t chunk0, int64_t row_chunks, int chunks, uint4* __restrict__ vec) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 4 * chunks) return;
    const int k = i / chunks, c = i - k * chunks;
    vec[i] = map[chunk0 + (k >> 1) * row_chunks + (int64_t)(k & 1) * chunks + c];
}
int conv_read_pixels2x2_bytes(hipStream_t st, const void* map, int64_t row_bytes, int64_t org_bytes, int pix_bytes, int y, int x, void* vec) {
    OMNI_REQUIRE(pix_bytes % 16 == 0 && row_bytes % 16 == 0 && org_bytes % 16 == 0 && y >= 0 && x >= 0 && y % 2 == 0 && x % 2 == 0, OMNI_ERR_INVALID, "conv_read_pixels2x2_bytes: bad pixel");
    const int chunks = pix_bytes / 16;
    hipLaunchKernelGGL(read_pixels2x2_kernel, dim3(cdiv(4 * chunks, 64)), dim3(64), 0, st, (const uint4*)map, (org_bytes + y * row_bytes + (int64_t)x * pix_bytes) / 16, row_bytes / 16, chunks,
                       (uint4*)vec);
    OMNI_LAUNCH_CHECK();
    return OMNI_OK;
}
__global__ void fill_rect2x2_kernel(char* __restrict__ map, int64_t img_bytes, int64_t row_bytes, int64_t org_bytes, int chunks, int y0, int x0, int rh, int rw,
                                    const uint4* __restrict__ vec, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // ((b * rh + y) * rw + x) * chunks + c
    if (i >= total) return;
    const int c = (int)(i % chunks);
    const int64_t px = i / chunks;
    const int x = x0 + (int)(px % rw);
    const int64_t by = px / rw;
    const int y = y0 + (int)(by % rh);
    const int64_t b = by / rh;
    *reinterpret_cast<uint4*>(map + b * img_bytes + org_bytes + y * row_bytes + ((int64_t)x * chunks + c) * 16) = vec[((y & 1) * 2 + (x & 1)) * chunks + c];
}
int conv_fill_rect2x2_bytes(hipStream_t st, void* map, int batch, int64_t img_bytes, int64_t row_bytes, int64_t org_bytes, int pix_bytes, int y0, int y1, int x0, int x1,
                            const void* vec) {
    OMNI_REQUIRE(pix_bytes % 16 == 0 && img_bytes % 16 == 0 && row_bytes % 16 == 0 && org_bytes % 16 == 0 && y0 >= 0 && x0 >= 0, OMNI_ERR_INVALID, "conv_fill_rect2x2_bytes: bad layout");
    if (y1 <= y0 || x1 <= x0 || batch <= 0) return OMNI_OK;
    const int chunks = pix_bytes / 16;
    const int64_t total = (int64_t)batch * (y1 - y0) * (x1 - x0) * chunks;
    hipLaunchKernelGGL(fill_rect2x2_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, st, (char*)map, img_bytes, row_bytes, org_bytes, chunks, y0, x0, y1 - y0, x1 - x0,
                       (const uint4*)vec, total);
    OMNI_LAUNCH_CHECK();
    return OMNI_OK;
}

// raw-32 frames -> NCHW fp32, true values (test hook: omni_sp_debug_layer)
__global__ void raw32_to_nchw_kernel(const float* __restrict__ in, float* __restrict__ out, int batch, int C, int H, int W, int Hf, int Wf, float inv_scale) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)batch * C * H * W;
    if (i >= total) return;
    const int x = (int)(i % W), y = (int)((i / W) % H), c = (int)((i / ((int64_t)W * H)) % C), b = (int)(i / ((int64_t)W * H * C));
    out[i] = in[(((int64_t)b * Hf + y + 1) * Wf + x + 1) * C + c] * inv_scale;
}
int raw32_to_nchw_f32(hipStream_t st, const void* in_raw, float* out, int batch, int C, int H, int W) {
    const int64_t total = (int64_t)batch * C * H * W;
    hipLaunchKernelGGL(raw32_to_nchw_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, st, reinterpret_cast<const float*>(in_raw), out, batch, C, H, W,
                       split_frame_h(H), split_frame_w(W), 1.f / WN_ACT_SCALE);
    OMNI_LAUNCH_CHECK();
    return OMNI_OK;
}

}  // namespace omni
