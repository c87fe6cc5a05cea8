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
// A layer's H x W map sits in a ZERO FRAME of split_frame_h(H) x split_frame_w(W) pixels (pixel (y, x) at row y + 1, column x + 1; the
// frame is zeroed once when the buffers are made and never written): the conv's zero padding and the overhang of the last tiles are
// plain memory, every halo tile is one rectangle of the frame and the LDS-DMA of all tiles is the same 13 / 17 instructions per wave
// with per-lane offsets computed once -- no border path (which cost 4 500 cycles per border tile against 7 200 for its MFMAs, and at
// 60 x 75 most tiles are border tiles).
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
#include "config.h"
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
// (measured in round 4 and not kept: the pixel block as the instruction's immediate offset and a wait without the fragment as an operand (no s_nop
// in front of the MFMA) remove ~120 instructions per tile, but hipcc then keeps two dozen more addresses in registers and splits more weight
// fragments' live ranges (v_accvgpr_mov x 4 in front of their MFMAs): same time)
template <int N>
__device__ __forceinline__ void spl_wait(half8_t& f) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "i"(N)); }

// halo row r feeds output row 0 through tap row ky = r and output row 1 through ky = r - 1.  The fragments go through a ring of SPL_NB
// registers quads, SPL_NB - 1 reads ahead of the matrix cores: a step holds 1 to 4 MFMAs (32 to 128 cycles: rows 0 and 3 feed one output
// row only, a lo fragment meets the hi weights only), so a short look-ahead would fall under the LDS latency in the sparse stretches --
// and with one wave per SIMD nothing else hides it.  Both output rows sum their 9 x 64 (x 2) products in the SAME order (tap row 0, 1, 2 per tap
// column): a pixel's value does not depend on which row of a tile it is -- what the constant region of the fisheye mask relies on.
// (Round 4, measured and not kept: rows 0 and 3 interleaved and issued together, so that no two consecutive MFMAs accumulate into the same
// registers -- same time, and the two rows' sums then differ in the last bit.)
//
// Everything else a tile needs rides in this stream (one wave per SIMD: whatever is not issued between two MFMAs idles the matrix cores):
// the "dense" steps (4 MFMAs = 128 cycles) each carry one PART of the PREVIOUS tile's epilogue (scale + bias + ReLU +
// hi / lo split of 2 values per lane; every second part the cross-lane swap, every fourth the stores), the other steps the LDS-DMA
// instructions of the NEXT tile one at a time (a burst of 17 behind the stores stalls the wave in the VMEM issue queue for 2 000
// cycles; measured with the s_memtime trace below).  The in-stream trace of round 4 (tools/fz_ablate.sh trace) prices it: the steps without
// a part run within 3-8 % of their MFMAs' time, a part costs ~100 cycles beyond its step's four MFMAs (135 before its 27 instructions became 13).
#define SPL_NB 8
constexpr bool spl_dense(int L) { return ((L / 8) % 4 == 1 || (L / 8) % 4 == 2) && L % 8 < 4; }
constexpr int spl_dense_before(int L) { int c = 0; for (int l = 0; l < L; ++l) c += spl_dense(l) ? 1 : 0; return c; }

#define SPL_MID_STEP 20                               // cin = 128 has at most 8 epilogue parts (dense steps 8-11, 16-19): all behind it
// SCH: the wait in front of step L's MFMAs -- how many LDS operations may still be in flight when fragment L must have landed (LDS returns in
// order): the ring's own look-ahead, plus (FUSE1A) the LDS operations of the tile build issued since that fragment's read
struct SplPlainSched { static constexpr int ring_wait(int L) { return (L + SPL_NB - 1 < 96) ? SPL_NB - 1 : 95 - L; } };
template <int L, int NSL, int NDMA, bool MID, typename SCH = SplPlainSched, typename Epi, typename Dma, typename Fz>
__device__ __forceinline__ void spl_steps(uint32_t row_base, int n_eff, int hh, const half8_t (&wreg)[72], floatx16 (&acc)[2], half8_t (&fb)[SPL_NB],
                                          Epi&& epi, Dma&& dma, Fz&& fz) {
    if constexpr (L < 96) {
        constexpr int kx = L / 32, r = (L / 8) % 4, kg = L % 8, kq = kg & 3;
        constexpr bool row0 = r <= 2, row1 = r >= 1;
        constexpr int t0 = (r * 3 + kx) * 4 + kq, t1 = ((r - 1) * 3 + kx) * 4 + kq;
        constexpr int D = SPL_NB - 1;
        constexpr int nd = spl_dense_before(L), nn = L - nd;
        if constexpr (L + D < 96) spl_read<L + D>(row_base, n_eff, hh, fb[(L + D) % SPL_NB]);
        spl_wait<SCH::ring_wait(L)>(fb[L % SPL_NB]);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (row0) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[t0], fb[L % SPL_NB], acc[0], 0, 0, 0);
        if constexpr (row1) acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[t1], fb[L % SPL_NB], acc[1], 0, 0, 0);
        if constexpr (kg < 4) {                                                     // a hi fragment also meets the lo halves of the weights
            if constexpr (row0) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[36 + t0], fb[L % SPL_NB], acc[0], 0, 0, 0);
            if constexpr (row1) acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[36 + t1], fb[L % SPL_NB], acc[1], 0, 0, 0);
        }
        if constexpr (spl_dense(L)) {
            if constexpr (nd < NSL) {
                epi(std::integral_constant<int, nd>{});
                // one MFMA, then a quarter of the part in its shadow
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
            }
        } else {
            if constexpr (nn < NDMA) dma(std::integral_constant<int, nn>{});
        }
        fz(std::integral_constant<int, L>{});                                       // FUSE1A: this step's share of the next tile's build
        if constexpr (MID && L == SPL_MID_STEP) __builtin_amdgcn_s_barrier();       // cin = 128: see the K-split exchange in the kernel
        __builtin_amdgcn_sched_barrier(0);
        spl_steps<L + 1, NSL, NDMA, MID, SCH>(row_base, n_eff, hh, wreg, acc, fb, epi, dma, fz);
    }
}
template <int J, int N, typename F>
__device__ __forceinline__ void spl_for_each(F&& f) {
    if constexpr (J < N) { f(std::integral_constant<int, J>{}); spl_for_each<J + 1, N>(f); }
}
// the first SPL_NB - 1 fragments of a tile
template <int L = 0>
__device__ __forceinline__ void spl_prime(uint32_t row_base, int n_eff, int hh, half8_t (&fb)[SPL_NB]) {
    if constexpr (L < SPL_NB - 1) {
        spl_read<L>(row_base, n_eff, hh, fb[L]);
        spl_prime<L + 1>(row_base, n_eff, hh, fb);
    }
}

// ---- a tile's pending epilogue: the raw accumulator values of one lane and where they go ------------------------------------------------------
//   cin = 64 : v = acc[2][16] (register groups 2 gp, 2 gp + 1 of output row f = "pair" (f, gp); POOL: the pair is the maximum over the 2 x 2 window)
//   cin = 128: v = the 2 x 8 values this wave finishes after the K-split exchange (pair = output row f; POOL: one pair)
// A pair = 8 values of one pixel = channels 16 gp + 4 hh + {0..3} and 16 gp + 8 + 4 hh + {0..3} of the wave's 32-channel fragment:
//   v = a * inv + bias -> ReLU -> OUT_F32: two float4 stores;  else hi = half(v), lo = half(v - hi), one 16-byte store each after a
//   v_permlane32_swap per dword (the half-waves hold interleaved 4-channel runs of the same pixel: afterwards the lower one owns channels
//   [16 gp, +8) and the upper one [16 gp + 8, +8)).
// A SLICE is half a pair (slice h of a pair: values 2h, 2h+1, 4+2h, 5+2h -> dword h of the four 8-byte halves; OUT_F32: values 4h .. 4h+3);
// the pair's stores go with its second slice.  Every lane runs every slice (the swap is a cross-lane operation); pred guards the stores.
struct SplPacked { uint4 a, b; };
struct SplTileIx { int b, r, ty, tx; };                               // image, index among the image's tiles that run, tile row, tile column
// ConvArgs::skip_* for this kernel's tile grid: the tiles of an image that run = the tile rows above the rectangle (n_above tiles), the tiles
// left and right of it in its own rows (bw per row, up to n_upto), the tile rows below; no rectangle: n_above = n_upto = act = tiles_x * tiles_y
struct SplSkip { int act, n_above, n_upto, y0, y1, x0, w, bw; uint32_t magic_tx, magic_bw; int xcd /* OMNI_CONV_XCD: xcd_block_id() */; };
struct SplOrg { __amdgpu_buffer_rsrc_t r; uint32_t soff; };           // a halo tile's DMA source: the image's frame + the scalar offset of the halo origin
typedef float spl_f4 __attribute__((ext_vector_type(4)));
template <bool C128, bool POOL>
struct SplEpi {
    static constexpr int NV = C128 ? 16 : 32;
    static constexpr int NPK = C128 ? (POOL ? 1 : 2) : (POOL ? 2 : 4);
    static constexpr int NPO = POOL ? 1 : 2;
    float v[NV];
    spl_f4 p[C128 ? 4 : 1];       // cin = 128: the partner wave's partial sums of the same values (read from LDS at the top of the next tile)
    __amdgpu_buffer_rsrc_t img;   // the output image (raw buffer: a store beyond its bytes is dropped)
    uint32_t off[NPO];            // byte offset of the wave's 32-channel fragment of the lane's pixel in output row f; SPL_NO_STORE: nothing to store
    SplPacked pk;                 // the pair being built
    uint32_t qh, ql;              // ... and the first two values of the half being built
};
#define SPL_NO_STORE 0x80000000u   // >= the bytes of any image (checked by the launcher), also after the in-fragment offset is added
typedef uint32_t spl_u4 __attribute__((ext_vector_type(4)));

// Part S of a tile's epilogue.  A pair is done in four parts (OUT_F32: two): part q of half h computes the hi / lo dwords of two values
// (q = 0: values 2h, 2h+1; q = 1: values 4+2h, 5+2h); the second part of a half swaps the dwords across the half-waves, the last part of a
// pair stores it.  ~13 VALU instructions each: one part per dense step stays inside the shadow of the step's four MFMAs (a whole half per
// step -- 26 instructions against 128 cycles -- overran it by ~100 cycles per step; measured with the trace).
template <bool OUT_F32> constexpr int spl_parts_per_pair() { return OUT_F32 ? 2 : 4; }
template <int S, bool C128, bool POOL, bool OUT_F32>
__device__ __forceinline__ void spl_epi_part(SplEpi<C128, POOL>& e, const float4 (&bs)[C128 ? 2 : 4], float inv, int relu, int hh, int part) {
    constexpr int PP = spl_parts_per_pair<OUT_F32>();
    constexpr int i = S / PP, h = OUT_F32 ? S % 2 : (S % 4) >> 1, q = OUT_F32 ? 0 : S & 1;
    constexpr int f = POOL ? 0 : (C128 ? i : i >> 1);
    constexpr int gpc = C128 ? 0 : (POOL ? i : i & 1);
    const int gp = C128 ? part : gpc;
    auto val = [&](auto IC) -> float {                      // value IC of the lane: cin = 128 adds the partner's half of the K sum
        constexpr int ix = decltype(IC)::value;
        if constexpr (C128) {
            return e.v[ix] + e.p[ix >> 2][ix & 3];
        } else {
            return e.v[ix];
        }
    };
    auto raw = [&](auto JC) -> float {
        constexpr int j = decltype(JC)::value;
        if constexpr (POOL) {
            constexpr int base = C128 ? 0 : 8 * gpc;
            const float m = spl_max(val(std::integral_constant<int, base + j>{}), val(std::integral_constant<int, (C128 ? 8 : 16) + base + j>{}));
            return spl_max(m, spl_swap_pairs(m));
        } else {
            return val(std::integral_constant<int, (C128 ? 8 * f : 16 * f + 8 * gpc) + j>{});
        }
    };
    const float4 b0 = bs[C128 ? 0 : 2 * gpc], b1 = bs[C128 ? 1 : 2 * gpc + 1];
    if constexpr (OUT_F32) {
        const float4 bb = h ? b1 : b0;
        float x0 = fmaf(raw(std::integral_constant<int, 4 * h>{}), inv, bb.x), x1 = fmaf(raw(std::integral_constant<int, 4 * h + 1>{}), inv, bb.y);
        float x2 = fmaf(raw(std::integral_constant<int, 4 * h + 2>{}), inv, bb.z), x3 = fmaf(raw(std::integral_constant<int, 4 * h + 3>{}), inv, bb.w);
        if (relu) { x0 = fmaxf(x0, 0.f); x1 = fmaxf(x1, 0.f); x2 = fmaxf(x2, 0.f); x3 = fmaxf(x3, 0.f); }
        const spl_u4 d = {__float_as_uint(x0), __float_as_uint(x1), __float_as_uint(x2), __float_as_uint(x3)};
        __builtin_amdgcn_raw_buffer_store_b128(d, e.img, e.off[f] + (16 * gp + 4 * hh + 8 * h) * 4, 0, 0);
    } else {
        const float lo_lim = relu ? 0.f : -65000.f;
        const float4 bb = q ? b1 : b0;
        float r0, r1;
        if constexpr (POOL) {
            // the 2 x 2 maximum of two values in five instructions, one statement: the vertical maxima, then the neighbour lane's through a DPP operand
            // (written as fmed3(a, b, inf) hipcc makes it a maxnum whose operands it canonicalises first -- three instructions per maximum -- and the
            // lane exchange a v_mov 0 + s_nop + v_mov_dpp: 27 instructions per part in the round-3 kernel, next to four MFMAs that hide about sixteen).
            // The s_nop is the second of the two wait states a DPP read needs behind the VALU write of its source (the other maximum is the first).
            constexpr int base = C128 ? 0 : 8 * gpc, up = C128 ? 8 : 16, j0 = 4 * q + 2 * h;
            const float a0 = val(std::integral_constant<int, base + j0>{}), c0 = val(std::integral_constant<int, up + base + j0>{});
            const float a1 = val(std::integral_constant<int, base + j0 + 1>{}), c1 = val(std::integral_constant<int, up + base + j0 + 1>{});
            asm("v_max_f32 %0, %2, %3\n\tv_max_f32 %1, %4, %5\n\ts_nop 0\n\t"
                "v_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                "v_max_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                : "=&v"(r0), "=&v"(r1) : "v"(a0), "v"(c0), "v"(a1), "v"(c1));
        } else {
            r0 = raw(std::integral_constant<int, 4 * q + 2 * h>{}); r1 = raw(std::integral_constant<int, 4 * q + 2 * h + 1>{});
        }
        const float y0 = fmaf(r0, inv, h ? bb.z : bb.x);
        const float y1 = fmaf(r1, inv, h ? bb.w : bb.y);
        const float x0 = __builtin_amdgcn_fmed3f(y0, lo_lim, 65000.f), x1 = __builtin_amdgcn_fmed3f(y1, lo_lim, 65000.f);
        float2v_t fv; fv[0] = x0; fv[1] = x1;
        const half2v_t hv = __builtin_convertvector(fv, half2v_t);
        const uint32_t dh = __builtin_bit_cast(uint32_t, hv);
        // lo = half(x - float(hi)): one v_fma_mix{lo,hi}_f16 per value (half(fma(f16 source, -1.0, f32 source)) into the low / high half of the destination;
        // x - float(hi) is exact in f32, so the single rounding equals convert, subtract, convert -- five instructions for the pair as hipcc writes it)
        uint32_t dl;
        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(dl) : "v"(dh), "v"(x0));
        asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(dl) : "v"(dh), "v"(x1));
        if constexpr (q == 0) { e.qh = dh; e.ql = dl; }
        else {
            const auto rh = __builtin_amdgcn_permlane32_swap(e.qh, dh, false, false);
            const auto rl = __builtin_amdgcn_permlane32_swap(e.ql, dl, false, false);
            if constexpr (h == 0) { e.pk.a.x = rh[0]; e.pk.a.z = rh[1]; e.pk.b.x = rl[0]; e.pk.b.z = rl[1]; }
            else {
                e.pk.a.y = rh[0]; e.pk.a.w = rh[1]; e.pk.b.y = rl[0]; e.pk.b.w = rl[1];
                // no branch in the stream: lanes with nothing to store carry an offset outside the buffer
                const uint32_t vo = e.off[f] + (16 * gp + 8 * hh) * 2;
                const spl_u4 da = {e.pk.a.x, e.pk.a.y, e.pk.a.z, e.pk.a.w}, db = {e.pk.b.x, e.pk.b.y, e.pk.b.z, e.pk.b.w};
                __builtin_amdgcn_raw_buffer_store_b128(da, e.img, vo, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(db, e.img, vo + 128, 0, 0);
            }
        }
    }
}
template <int S, bool C128, bool POOL, bool OUT_F32>
__device__ __forceinline__ void spl_epi_all(SplEpi<C128, POOL>& e, const float4 (&bs)[C128 ? 2 : 4], float inv, int relu, int hh, int part) {
    if constexpr (S < SplEpi<C128, POOL>::NPK * spl_parts_per_pair<OUT_F32>()) {
        spl_epi_part<S, C128, POOL, OUT_F32>(e, bs, inv, relu, hh, part);
        spl_epi_all<S + 1, C128, POOL, OUT_F32>(e, bs, inv, relu, hh, part);
    }
}


// ==================================================================================================================================================
// FUSE1A (conv1b): the NEXT tile's 6 x 34-pixel x 64-channel halo is built from the u8 image inside the stream -- conv1a (swarm_loop/superpoint.ipynb:143,
// 1 -> 64 channels, 3x3, + ReLU) on the matrix cores with split operands, the fp16 path's scheme (conv.hip, FUSE1A there) -- instead of being DMA'd from
// a conv1a tensor in HBM (conv1a_split_kernel wrote 4.7 GB per 64 images and conv1b read it 1.6 times over through the halos).
//   x = float(u8) * float(1/255) = xh + xl (a 256-entry table of half pairs: OpenCV's convertTo value exactly), 32 w = wh + wl, 32 bias = bh + bl:
//   out = xh.wh + xl.wh + xh.wl + bias (the dropped xl.wl is 2^-22 relative), K = 32 = two v_mfma_f32_32x32x16_f16 per 32 pixels x 32 channels.
// A wave builds virtual pixels [64 w, 64 w + 64) of the halo (two 32-pixel fragments; pixels >= 204 land in the unused tail of the buffer), i.e. at
// most 3 halo rows = a 5-row x 40-byte patch of the image: one buffer_load_dword per lane at the top of the tile, parked (rows / columns outside the
// image and the fisheye mask's rows zeroed) in a wave-private LDS slot, from which a lane reads the bytes of ITS taps (the half-waves own disjoint
// taps: lanes 0-31 taps 0-4, lanes 32-63 taps 5-8 and the bias slots) and then their table entries, which ARE the MFMA's operand dwords.  Results:
// ReLU, hi = half(v), lo = half(v - hi), two 8-byte LDS stores per 4 channels straight into the swizzled halo image the stream reads next tile.
// All of it is cut into MICRO-OPS placed in the stream's steps by a compile-time schedule (FzSched) that also counts the LDS operations in flight:
// the ring's waits grow by the build's own operations, a micro-op that consumes LDS reads waits for exactly the operations issued since.
struct SplFuse {
    const uint8_t* gray = nullptr; int gstride = 0, gbytes = 0, mask_r0 = 0, mask_r1 = 0;
    const _Float16* w1a_frag = nullptr;   // conv1a_split_pack_fused: [2 k-halves][2 m][64 lanes][8 halfs]
    const uint32_t* lut_hl = nullptr;     // [256] half(x) | half(x - half(x)) << 16 (conv1a_make_split_lut)
};
#define FZ_LUT_OFF (2 * SPL_BUF_BYTES)               // inside the K-split exchange region (cin = 64 does not use it): the table, 1 KiB
#define FZ_PATCH_OFF (FZ_LUT_OFF + 1024)             // 4 waves x 256 B
#define FZ_W1A_OFF (FZ_LUT_OFF + 2048)               // 4 A fragments x 1 KiB
#define FZ_ZERO_OFF (FZ_LUT_OFF + 6144)              // 128 zero bytes: where the taps of a halo pixel outside the image read from
static_assert(FZ_ZERO_OFF + 128 <= 2 * SPL_BUF_BYTES + SPL_XCH_BYTES, "FUSE1A: LDS carve");

// micro-ops (f = fragment 0 / 1, k = tap slot 0-4, i = MFMA 2 j + m: k-half j, channel fragment m, u = 8 f + 4 m + g: register group g)
constexpr int FZ_PARK = 0;      // patch dword -> LDS (masked)
constexpr int FZ_V = 1;         // + f: is the lane's halo pixel inside the image? -> tap base address, bias slots
constexpr int FZ_U = 3;         // + 5 f + k: ds_read_u8 of tap slot k
constexpr int FZ_T = 13;        // + 5 f + k: table entry of that byte
constexpr int FZ_W = 23;        // + 4 f + i: A fragment i
constexpr int FZ_K = 31;        // + f: pack the B operands
constexpr int FZ_M = 33;        // + 4 f + i: the MFMA
constexpr int FZ_E = 41;        // + 4 u + s: register group u = 8 f + 4 m + g (values 4 g .. 4 g + 3 of channel fragment m) in four sub-units of <= 5 instructions:
                                //   0: values 0, 1 -> ReLU -> hi pair      1: their lo pair; values 2, 3 read out of the accumulators
                                //   2: ReLU, hi pair, lo pair of 2, 3       3: the two LDS stores
constexpr int FZ_NOPS = FZ_E + 64;
constexpr int FZ_PER_STEP = 4;
constexpr int fz_nlds(int id) { return id == FZ_PARK ? 1 : (id >= FZ_U && id < FZ_K) ? 1 : (id >= FZ_E && (id - FZ_E) % 4 == 3 ? 2 : 0); }
constexpr int fz_producer(int id) {      // the micro-op whose LDS-read results `id` consumes (-1: none)
    if (id >= FZ_T && id < FZ_W) return FZ_U + (id - FZ_T);
    if (id >= FZ_K && id < FZ_M) return FZ_T + 5 * (id - FZ_K) + 4;      // the last table read of the fragment covers the other four (LDS returns in order)
    if (id >= FZ_M && id < FZ_E) return FZ_W + (id - FZ_M);
    return -1;
}
struct FzSched {
    int op[96][FZ_PER_STEP];    // micro-ops of stream step L in issue order (-1: none)
    int step[FZ_NOPS];
    int wait[FZ_NOPS];          // lgkmcnt in front of a consuming micro-op (-1: it consumes no LDS read)
    int ring[96];               // lgkmcnt in front of step L's MFMAs
    bool ok;
};
constexpr FzSched fz_make_sched() {
    FzSched s{};
    s.ok = true;
    for (int L = 0; L < 96; ++L) for (int k = 0; k < FZ_PER_STEP; ++k) s.op[L][k] = -1;
    for (int i = 0; i < FZ_NOPS; ++i) { s.step[i] = -1; s.wait[i] = -1; }
    auto put = [&s](int id, int L) {
        for (int k = 0; k < FZ_PER_STEP; ++k) if (s.op[L][k] < 0) { s.op[L][k] = id; s.step[id] = L; return; }
        s.ok = false;
    };
    // The previous tile's epilogue owns the dense steps 8-11 / 16-19 (its stores are the only VMEM operations behind the patch load: the park at
    // step 20 finds them all issued).  The in-stream trace prices the rest: an MFMA hides about five other instructions issued behind it and NOT more --
    // the slots of one gap cannot be borrowed by another (in-order issue: the next MFMA waits behind whatever stands in front of it), and every
    // instruction beyond costs ~5 cycles.  The ring's own address + read + wait (+ s_nop) fill the gap behind a step's LAST MFMA, so a step has
    // 5 x (its MFMAs - 1) free slots: 15 in a dense step, 5 in a two-MFMA step, none in a one-MFMA step.  Splitting the 64 results into hi / lo halves is
    // 20 instructions per register group of four: cut into sub-units of <= 5 and packed, in order, into the free slots behind the fragment's MFMAs.
    // (First versions: a whole group in each of the sixteen free dense steps, or half a group in each of 32 steps: 70 cycles per group either way.)
    // Both fragments' bytes and table entries are read early (steps 20-35), so that the sub-units of fragment 0 have steps 40-63 to themselves
    // (eight dense steps x 3 + twelve two-MFMA steps x 1 = 36 places for 32 sub-units), those of fragment 1 steps 72-91 behind its MFMAs at 64-67.
    put(FZ_PARK, 20); put(FZ_V + 0, 21); put(FZ_V + 1, 22);
    { const int st[10] = {23, 23, 24, 24, 25, 25, 26, 26, 27, 27}; for (int k = 0; k < 10; ++k) put(FZ_U + k, st[k]); }
    { const int st[10] = {28, 29, 30, 31, 32, 32, 33, 33, 34, 34}; for (int k = 0; k < 10; ++k) put(FZ_T + k, st[k]); }
    { const int st[4] = {34, 35, 35, 35}; for (int i = 0; i < 4; ++i) put(FZ_W + i, st[i]); }
    put(FZ_K + 0, 36);
    for (int i = 0; i < 4; ++i) put(FZ_M + i, 36 + i);
    for (int i = 0; i < 4; ++i) put(FZ_W + 4 + i, 60 + i);
    put(FZ_K + 1, 64);
    for (int i = 0; i < 4; ++i) put(FZ_M + 4 + i, 64 + i);
    for (int f = 0; f < 2; ++f) {
        int L = f ? 68 : 40, n = 0;
        for (int i = 0; i < 32; ++i) {
            for (;; ++L, n = 0) {
                if (L >= 96) break;
                const int r = (L / 8) % 4, kg = L % 8;
                const int nm = ((r == 1 || r == 2) ? 2 : 1) * (kg < 4 ? 2 : 1);
                const int places = nm == 4 ? (spl_dense_before(L) < 8 ? 0 : 3) : (nm == 2 ? 1 : 0);
                if (n < places && s.op[L][FZ_PER_STEP - 1] < 0 && (places > 1 || s.op[L][0] < 0)) break;     // (a two-MFMA step that already holds something: full)
            }
            if (L >= 96) { s.ok = false; break; }
            put(FZ_E + 32 * f + i, L); ++n;
        }
    }
    // LDS operations in program order: the ring's reads 0 .. SPL_NB - 2 (spl_prime), then per step the ring's read L + SPL_NB - 1, the wait for
    // fragment L, the step's micro-ops
    int pos = 0, ring_pos[96] = {}, start[FZ_NOPS] = {}, end[FZ_NOPS] = {};
    for (int r = 0; r < SPL_NB - 1; ++r) ring_pos[r] = ++pos;
    for (int L = 0; L < 96; ++L) {
        if (L + SPL_NB - 1 < 96) ring_pos[L + SPL_NB - 1] = ++pos;
        const int younger = pos - ring_pos[L];
        s.ring[L] = younger < 15 ? younger : 15;
        for (int k = 0; k < FZ_PER_STEP; ++k) {
            const int id = s.op[L][k];
            if (id < 0) continue;
            start[id] = pos; pos += fz_nlds(id); end[id] = pos;
        }
    }
    for (int id = 0; id < FZ_NOPS; ++id) {
        if (s.step[id] < 0) s.ok = false;
        const int p = fz_producer(id);
        if (p < 0) continue;
        if (s.step[p] < 0 || start[id] < end[p]) { s.ok = false; continue; }      // the producer must be issued first
        const int younger = start[id] - end[p];
        s.wait[id] = younger < 15 ? younger : 15;
    }
    // program order inside the chains that are NOT tied by an LDS read: V -> U, K -> M (same or later step, later slot), M -> E
    for (int f = 0; f < 2; ++f) {
        for (int k = 0; k < 5; ++k) if (start[FZ_U + 5 * f + k] < 0 || s.step[FZ_V + f] > s.step[FZ_U + 5 * f + k]) s.ok = false;
        if (s.step[FZ_PARK] > s.step[FZ_U + 5 * f]) s.ok = false;
        for (int i = 0; i < 4; ++i) if (s.step[FZ_K + f] > s.step[FZ_M + 4 * f + i]) s.ok = false;
        for (int i = 0; i < 32; ++i) if (s.step[FZ_M + 4 * f + 3] >= s.step[FZ_E + 32 * f + i]) s.ok = false;      // behind the fragment's MFMAs
        if (s.step[FZ_M + 4 * f] > s.step[FZ_M + 4 * f + 2] || s.step[FZ_M + 4 * f + 1] > s.step[FZ_M + 4 * f + 3]) s.ok = false;
    }
    if (s.step[FZ_E + 31] >= s.step[FZ_M + 4]) s.ok = false;                     // fragment 1 takes over fragment 0's accumulators
    return s;
}
inline constexpr FzSched kFzSched = fz_make_sched();
static_assert(kFzSched.ok, "FUSE1A: the build's schedule is inconsistent");
struct SplFuseSched { static constexpr int ring_wait(int L) { return kFzSched.ring[L]; } };

// conv1a's weights and bias (x SPL_ACT_SCALE: conv1b reads scaled activations) as split A fragments [j = k-half][m][lane = 32 hh + i][8 halfs] of channel
// 32 m + i; the K slot (j, hh, e) pairs with the B dwords FZ_K packs (T0..T4 = the table entries (xh | xl << 16) of the half-wave's own taps):
//   hh = 0 (taps 0-4):         j = 0: [T0][T1][T2][T3] x (wh, wh) per tap      j = 1: [T4] x (wh4, wh4), [xh0 xh1] x (wl0, wl1), [xh2 xh3] x (wl2, wl3), [T4] x (wl4, 0)
//   hh = 1 (taps 5-8, T4 idle): j = 0: [T0][T1][T2][T3] x (wh, wh) of taps 5-8   j = 1: [T4] x (0, 0),     [xh5 xh6] x (wl5, wl6), [xh7 xh8] x (wl7, wl8), [1 1] x (bias_hi, bias_lo)
void conv1a_split_pack_fused(const float* w /*[64][9]*/, const float* bias /*[64]*/, uint16_t* frag /*[2][2][64][8]*/) {
    for (int j = 0; j < 2; ++j)
        for (int m = 0; m < 2; ++m)
            for (int l = 0; l < 64; ++l)
                for (int e = 0; e < 8; ++e) {
                    const int co = m * 32 + (l & 31), hh = l >> 5;
                    auto val = [&](int t) { return SPL_ACT_SCALE * w[co * 9 + t]; };
                    auto hi = [&](float v) { return f2h_bits(v); };
                    auto lo = [&](float v) { return f2h_bits(v - h2f(f2h_bits(v))); };
                    const float b = SPL_ACT_SCALE * bias[co];
                    uint16_t v = 0;
                    if (j == 0) v = hi(val((hh ? 5 : 0) + e / 2));
                    else if (hh == 0) v = e < 2 ? hi(val(4)) : (e < 6 ? lo(val(e - 2)) : (e == 6 ? lo(val(4)) : 0));
                    else v = e < 2 ? 0 : (e < 6 ? lo(val(5 + e - 2)) : (e == 6 ? hi(b) : lo(b)));
                    frag[((j * 2 + m) * 64 + l) * 8 + e] = v;
                }
}

// TRN (cin = 128, no pooling): transposed tiles -- the 32-pixel fragments run along y, the two fragment rows along x (the LDS image, the k order
// and every MFMA are those of the plain kernel; only the pixel <-> address maps and the tap the weights are loaded for differ).  A 60x75 layer
// is 2 x 38 tiles instead of 3 x 30: 75-pixel rows fill 2.3 of 3 fragments, 60-pixel columns 1.9 of 2.
template <bool C128, bool POOL, bool OUT_F32, bool TRN = false, bool FUSE1A = false, bool FZMIX = true>
__global__ void __launch_bounds__(256, 1)
conv3x3_split_kernel(const char* __restrict__ in, void* __restrict__ out, const _Float16* __restrict__ wp, const float* __restrict__ bias,
                     float inv, int H, int W, int cout, int n_cg, int tiles_x, int tiles_y, int batch, int relu, SplSkip sk,
                     int dbg /* OMNI_SPLIT_DBG (timing experiments, WRONG results): 1 = no stores, 2 = every DMA reads tile 0 */,
                     unsigned long long* trace /* OMNI_SPLIT_TRACE=1: s_memtime stamps of workgroup 0, waves 0 and 3 (debug only), else nullptr */,
                     SplFuse fz /* FUSE1A: the u8 image the halo tiles are built from (`in` is unused) */) {
    extern __shared__ __attribute__((aligned(256))) char smem_raw[];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem_raw;
    static_assert(!TRN || (C128 && !POOL), "transposed tiles: cin = 128 without pooling only");
    static_assert(!FUSE1A || (!C128 && POOL && !OUT_F32), "FUSE1A: conv1b (cin = 64, pooled, split output) only");
    constexpr int TH = C128 ? 2 : 4;
    constexpr int PIXB = C128 ? 512 : 256;                  // bytes per input pixel in HBM
    constexpr int NPIECES = C128 ? 68 : 51, PPW = C128 ? 17 : 13;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int co = wave & 1, part = wave >> 1;
    const int n = lane & 31, hh = lane >> 5;
    const int bid = xcd_block_id(sk.xcd);
    const int cg = bid % n_cg, wg = bid / n_cg, nwg = gridDim.x / n_cg;
    const int tiles_per_img = sk.act;                       // (the tiles that run)
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
    if constexpr (FUSE1A) {
        reinterpret_cast<uint32_t*>(smem_raw + FZ_LUT_OFF)[tid] = fz.lut_hl[tid];
        reinterpret_cast<uint4*>(smem_raw + FZ_W1A_OFF)[tid] = reinterpret_cast<const uint4*>(fz.w1a_frag)[tid];
        if (tid < 32) reinterpret_cast<uint32_t*>(smem_raw + FZ_ZERO_OFF)[tid] = 0u;
        reinterpret_cast<uint32_t*>(smem_raw + FZ_PATCH_OFF)[tid] = 0u;
    }

    // tile t = (image b, tile r of the image's tiles that run); a workgroup walks t = wg, wg + nwg, ...: (b, r) advance by a carry, (tile row,
    // tile column) come from r by multiply-high divisions (scalar: a dozen SALU instructions per tile)
    // (the transposed cin = 128 tiles -- 32 rows tall -- never fit a rectangle: their walk stays on carries alone, r unused)
    const int step_b = nwg / tiles_per_img, step_r = nwg - step_b * tiles_per_img;
    const int step_y = step_r / tiles_x, step_x = step_r - step_y * tiles_x;
    auto decode = [&](SplTileIx& q) {
        int r = q.r, ty, tx;                                            // (locals, assigned to q once: stores in both branches send the struct to scratch)
        if constexpr (TRN) {
            ty = r / tiles_x; tx = r - ty * tiles_x;
        } else if (r < sk.n_above || r >= sk.n_upto) {                  // full tile rows above / below the rectangle
            int base = 0;
            if (r >= sk.n_upto) { r -= sk.n_upto; base = sk.y1; }
            const int ry = sk.magic_tx ? (int)__umulhi((uint32_t)r, sk.magic_tx) : r;      // magic 0 = divisor 1
            tx = r - ry * tiles_x; ty = ry + base;
        } else {                                                        // its rows: the tiles left and right of it
            r -= sk.n_above;
            const int qy = sk.magic_bw ? (int)__umulhi((uint32_t)r, sk.magic_bw) : r;
            const int c = r - qy * sk.bw;
            ty = sk.y0 + qy; tx = c < sk.x0 ? c : c + sk.w;
        }
        q.ty = ty; q.tx = tx;
    };
    auto advance = [&](SplTileIx& q) {
        if constexpr (TRN) {
            q.tx += step_x;
            if (q.tx >= tiles_x) { q.tx -= tiles_x; ++q.ty; }
            q.ty += step_y;
            if (q.ty >= tiles_y) { q.ty -= tiles_y; ++q.b; }
            q.b += step_b;
        } else {
            q.r += step_r;
            if (q.r >= tiles_per_img) { q.r -= tiles_per_img; ++q.b; }
            q.b += step_b;
            decode(q);
        }
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
    const int Wf = split_frame_w(W), Hf = split_frame_h(H);              // the input's frame
    uint32_t goff[PPW];                       // byte offset of this lane's chunk of piece j relative to the halo origin
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        int piece = wave * PPW + j;
        piece = piece < NPIECES ? piece : NPIECES - 1;
        const int idx = piece * 64 + lane;
        int iy, ix;
        const uint32_t inner = src_of(idx >> 4, idx & 15, iy, ix);
        goff[j] = (uint32_t)(iy * Wf + ix) * PIXB + inner;
    }
    // piece j of this wave (the last wave of a cin = 64 workgroup has 12: its 13th is the 12th again).  Buffer addressing: descriptor = the
    // image's frame, scalar offset = the halo origin, per-lane offset = goff[j]: no vector arithmetic per instruction.  Four consecutive
    // pieces share one M0 (the LDS base) and differ in the instruction's immediate offset, which moves the LDS address AND the memory
    // address by k KiB: the scalar offset takes the k KiB back (the descriptor starts 4 KiB in front of the frame so that it stays positive).
    // A write to M0 waits for the LDS-DMA instructions in flight to have consumed the old value -- 17 writes per tile cost the stream
    // ~600 cycles.
    const uint32_t in_img_bytes = (uint32_t)Hf * Wf * PIXB;
    auto origin = [&](int qb, int qty, int qtx) -> SplOrg {
        SplOrg o;
        o.r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(in) + (int64_t)qb * in_img_bytes - 4096, 0, in_img_bytes + 8192, 0x00020000);
        // halo origin (ty0 - 1, tx0 - 1) = frame pixel (ty0, tx0)
        o.soff = (uint32_t)(qty * (TRN ? 32 : TH) * Wf + qtx * (TRN ? TH : 32)) * PIXB + 4096;
        return o;
    };
    auto dma_piece = [&](const SplOrg& o, int which, auto JC) {
        constexpr int j = decltype(JC)::value, g = j / 4;
        [[maybe_unused]] constexpr int k = j % 4;
        int first = wave * PPW + 4 * g;                                  // the group's first piece
        first = first < NPIECES ? first : NPIECES - 1;
#if __HIP_DEVICE_COMPILE__          // hipcc's host pass has no target for this builtin and silently drops the kernel's launch stub when it meets it
        __builtin_amdgcn_raw_ptr_buffer_load_lds(o.r, (__attribute__((address_space(3))) void*)(smem_raw + which * SPL_BUF_BYTES + first * 1024), 16,
                                                 goff[j], o.soff - k * 1024, k * 1024, 0);
#else
        (void)o; (void)which; (void)first;
#endif
    };

    // this wave's rows of the buffer start at virtual pixel 68 part (cin = 64: output rows 2 part, 2 part + 1 read halo rows 2 part .. + 3)
    // or 136 part (cin = 128: the 4 x 34 pixels of input block `part`)
    const int n_eff = n + part * (C128 ? 136 : 2 * SPL_ITW);
    const int Ho = POOL ? (H >> 1) : H, Wo = POOL ? (W >> 1) : W;
    // output addressing inside image b (a raw buffer of Hof * Wof * opix bytes): pixel (y, x) = (y * Wof + x) * opix + oorg
    const int Hof = OUT_F32 ? Ho : split_frame_h(Ho), Wof = OUT_F32 ? Wo : split_frame_w(Wo);
    // bytes per output pixel and this wave's 32-channel fragment inside it
    const int64_t opix = (int64_t)cout * 4;                 // fp32, or split-64: 2 halfs per channel
    const int64_t ofrag = OUT_F32 ? (int64_t)g32 * 32 * 4 : (int64_t)(g32 >> 1) * 256 + (g32 & 1) * 64;
    const uint32_t oorg = (uint32_t)((OUT_F32 ? 0 : ((int64_t)Wof + 1) * opix) + ofrag);                 // the frame's origin + the wave's fragment

    // ---- FUSE1A: the tile build (see FzSched) ------------------------------------------------------------------------------------------------
    struct FzSt { uint32_t pv, u[2][5], pbv[2], xc[2], wb, hq, lq, dh, dl1; float x0, x1; half8_t wa[4], B0, B1; floatx16 a[2]; } z;
    int fz_b = 0, fz_ty0 = 0, fz_tx0 = 0, fz_which = 0;          // the tile being built: image, origin of its outputs, halo buffer
    [[maybe_unused]] const int fz_r0 = (64 * wave) / SPL_ITW;                         // first halo row the wave's 64 virtual pixels touch
    // the lane's dword of the wave's 5-row x 40-byte patch: row lane / 10, bytes [4 (lane % 10), + 4) = image row ty0 + fz_pj, columns tx0 + fz_pd ..
    [[maybe_unused]] const int fz_pj = lane / 10 + fz_r0 - 2, fz_pd = 4 * (lane % 10) - 4;
    [[maybe_unused]] const uint32_t fz_patch = lds0 + FZ_PATCH_OFF + wave * 256;
    [[maybe_unused]] int fz_iy[2], fz_ix[2];
    [[maybe_unused]] uint32_t fz_pb[2];                                              // LDS address of the byte under tap (0, 0) of the lane's pixel of fragment f
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        const int p = 64 * wave + 32 * f + n;
        fz_iy[f] = p / SPL_ITW; fz_ix[f] = p - fz_iy[f] * SPL_ITW;
        int dy = fz_iy[f] - fz_r0;
        dy = dy > 2 ? 2 : dy;                                                       // (virtual pixels >= 204 are not part of the halo: anything inside the patch)
        fz_pb[f] = fz_patch + dy * 40 + fz_ix[f] + 2;                               // patch column 0 = image column tx0 - 4, halo column ix = tx0 - 1 + ix
    }
    [[maybe_unused]] const uint32_t fz_wl = lds0 + FZ_W1A_OFF + lane * 16;
    // where the wave's results go: pixel vp = 64 wave + 32 f + n, logical 16-byte chunk c (hi halves: c = channel / 8, lo halves: 8 + channel / 8) in
    // slot c ^ (vp & 15); a lane holds channels 4 hh + {0..3} of a group of 8: A0 ^ (c << 4) (+ 8 KiB for fragment 1)
    [[maybe_unused]] const uint32_t fz_a0 = (uint32_t)((64 * wave + n) * 256 + ((n & 15) << 4) + 8 * hh);
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t fz_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(fz.gray), 0, fz.gbytes, 0x00020000);
    // the patch load of the tile (fz_b, fz_ty0, fz_tx0): rows / columns outside the image are clamped to valid addresses here and zeroed when parked
    auto fz_issue = [&]() {
        int yy = fz_ty0 + fz_pj, xx = fz_tx0 + fz_pd;
        yy = yy < 0 ? 0 : (yy > H - 1 ? H - 1 : yy);
        xx = xx < 0 ? 0 : (xx > fz.gstride - 4 ? fz.gstride - 4 : xx);
        z.pv = __builtin_amdgcn_raw_buffer_load_b32(fz_rsrc, (fz_b * H + yy) * fz.gstride + xx, 0, 0);
        z.wb = lds0 + fz_which * SPL_BUF_BYTES + fz_a0;
    };
    // micro-op ID; PRO: outside the stream (the workgroup's first tile): every wait is lgkmcnt(0)
#ifndef FZ_ABL
#define FZ_ABL 0      // timing ablations of the build (WRONG results; separate builds of the library, tools/fz_ablate.sh): 1 = results kept alive but not split /
#endif                // stored, 2 = split but not stored, 4 = nothing of the build inside the stream, 8 = no MFMAs of the build, 16 = no reads / table look-ups
    auto fz_op = [&](auto IDC, auto PROC) {
        constexpr int ID = decltype(IDC)::value;
        constexpr bool PRO = decltype(PROC)::value;
        if constexpr (!PRO && (FZ_ABL & 4)) return;
        if constexpr (!PRO && (FZ_ABL & 1) && ID >= FZ_E) {
            constexpr int uu = (ID - FZ_E) / 4, ss = (ID - FZ_E) % 4;
            if constexpr (ss < 2) asm volatile("" :: "v"(z.a[(uu % 8) / 4][4 * (uu % 4) + 2 * ss]), "v"(z.a[(uu % 8) / 4][4 * (uu % 4) + 2 * ss + 1]));
            return;
        }
        if constexpr (!PRO && (FZ_ABL & 8) && ID >= FZ_M && ID < FZ_E) {
            constexpr int mi = (ID - FZ_M) % 4;
            if constexpr (mi / 2 == 0) {
#pragma unroll
                for (int q = 0; q < 16; ++q) z.a[mi % 2][q] = __builtin_bit_cast(float, z.u[(ID - FZ_M) / 4][q % 5]) + (float)z.wa[mi][q % 8];
            }
            return;
        }
        if constexpr (!PRO && (FZ_ABL & 16) && ID >= FZ_U && ID < FZ_K) return;
        [[maybe_unused]] constexpr int WAITN = PRO ? 0 : (kFzSched.wait[ID] < 0 ? 0 : kFzSched.wait[ID]);
        if constexpr (ID == FZ_PARK) {
            const int y = fz_ty0 + fz_pj, x = fz_tx0 + fz_pd;
            // (no short-circuit evaluation: a branch here would split the step's scheduling region)
            const bool ok = ((unsigned)y < (unsigned)H) & ((unsigned)(y - fz.mask_r0) >= (unsigned)(fz.mask_r1 - fz.mask_r0)) & ((unsigned)x < (unsigned)W);
            const uint32_t v = ok ? z.pv : 0u;
            asm volatile("ds_write_b32 %0, %1" :: "v"(fz_patch + lane * 4), "v"(v) : "memory");
        } else if constexpr (ID < FZ_U) {                                   // V(f)
            constexpr int f = ID - FZ_V;
            const int gy = fz_ty0 - 1 + fz_iy[f], gx = fz_tx0 - 1 + fz_ix[f];
            const bool valid = ((unsigned)gy < (unsigned)H) & ((unsigned)gx < (unsigned)W);
            z.pbv[f] = valid ? fz_pb[f] : lds0 + FZ_ZERO_OFF;               // a pixel outside the image is conv1b's zero padding: taps read zeros, no bias
            z.xc[f] = (valid & (hh != 0)) ? 0x3C003C00u : 0u;              // the two bias slots (lanes 32-63): (1.0, 1.0)
        } else if constexpr (ID < FZ_T) {                                   // U(f, k): the byte under the half-wave's k-th tap
            constexpr int f = (ID - FZ_U) / 5, k = (ID - FZ_U) % 5;
            constexpr int c0 = (k / 3) * 40 + k % 3, tp1 = k < 4 ? 5 + k : 8, c1 = (tp1 / 3) * 40 + tp1 % 3;
            const uint32_t addr = z.pbv[f] + (uint32_t)hh * (uint32_t)(c1 - c0);
            asm volatile("ds_read_u8 %0, %1 offset:%2" : "=v"(z.u[f][k]) : "v"(addr), "i"(c0) : "memory");
        } else if constexpr (ID < FZ_W) {                                   // T(f, k): its table entry (xh | xl << 16)
            constexpr int f = (ID - FZ_T) / 5, k = (ID - FZ_T) % 5;
            asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(z.u[f][k]) : "i"(WAITN));
            const uint32_t addr = lds0 + FZ_LUT_OFF + (z.u[f][k] << 2);
            asm volatile("ds_read_b32 %0, %1" : "=v"(z.u[f][k]) : "v"(addr) : "memory");
        } else if constexpr (ID < FZ_K) {                                   // W(f, i)
            constexpr int i = (ID - FZ_W) % 4;
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(z.wa[i]) : "v"(fz_wl), "i"(i * 1024) : "memory");
        } else if constexpr (ID < FZ_M) {                                   // K(f)
            constexpr int f = ID - FZ_K;
            asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(z.u[f][0]), "+v"(z.u[f][1]), "+v"(z.u[f][2]), "+v"(z.u[f][3]), "+v"(z.u[f][4]) : "i"(WAITN));
            const uint32_t h01 = __builtin_amdgcn_perm(z.u[f][1], z.u[f][0], 0x05040100u), h23 = __builtin_amdgcn_perm(z.u[f][3], z.u[f][2], 0x05040100u);
            const uint32_t x = hh ? z.xc[f] : z.u[f][4];
            z.B0 = __builtin_bit_cast(half8_t, make_uint4(z.u[f][0], z.u[f][1], z.u[f][2], z.u[f][3]));
            z.B1 = __builtin_bit_cast(half8_t, make_uint4(z.u[f][4], h01, h23, x));
        } else if constexpr (ID < FZ_E) {                                   // M(f, i)
            constexpr int i = (ID - FZ_M) % 4, j = i / 2, m = i % 2;
            asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(z.wa[i]) : "i"(WAITN));
            if constexpr (j == 0) {
                floatx16 zero;
#pragma unroll
                for (int q = 0; q < 16; ++q) zero[q] = 0.f;
                z.a[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(z.wa[i], z.B0, zero, 0, 0, 0);
            } else {
                z.a[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(z.wa[i], z.B1, z.a[m], 0, 0, 0);
            }
            // (measured and not kept: inline-asm MFMAs writing architectural registers, so that the 64 results need no v_accvgpr_read: hipcc then parks as
            // many other values in the accumulator half and reads those back -- the same instruction count, the same time)
        } else {                                                            // E(u, sub)
            constexpr int u = (ID - FZ_E) / 4, sub = (ID - FZ_E) % 4, f = u / 8, m = (u % 8) / 4, g = u % 4;
            auto lo_pair = [&](uint32_t dh, float x0, float x1) -> uint32_t {
                uint32_t dl;
                if constexpr (FZMIX) {
                    // lo = half(x - float(hi)) in one instruction per value: v_fma_mix{lo,hi}_f16 = half(fma(f16 source, -1.0, f32 source)) into the low / high
                    // half of the destination (x - float(hi) is exact in f32, so the single rounding is the same as convert, subtract, convert)
                    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(dl) : "v"(dh), "v"(x0));
                    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(dl) : "v"(dh), "v"(x1));
                } else {
                    const half2v_t hv = __builtin_bit_cast(half2v_t, dh);
                    float2v_t rv; rv[0] = x0 - (float)hv[0]; rv[1] = x1 - (float)hv[1];
                    dl = __builtin_bit_cast(uint32_t, __builtin_convertvector(rv, half2v_t));
                }
                return dl;
            };
            if constexpr (sub == 0) {                                       // values 0, 1: ReLU (+ the fp16 range), the hi pair
                z.x0 = __builtin_amdgcn_fmed3f(z.a[m][4 * g], 0.f, 65000.f);
                z.x1 = __builtin_amdgcn_fmed3f(z.a[m][4 * g + 1], 0.f, 65000.f);
                float2v_t fv; fv[0] = z.x0; fv[1] = z.x1;
                z.hq = __builtin_bit_cast(uint32_t, __builtin_convertvector(fv, half2v_t));
            } else if constexpr (sub == 1) {                                // their lo pair; values 2, 3 leave the accumulators (v_accvgpr_read here, not in sub-unit 2)
                z.lq = lo_pair(z.hq, z.x0, z.x1);
                z.x0 = z.a[m][4 * g + 2]; z.x1 = z.a[m][4 * g + 3];
                asm volatile("" : "+v"(z.x0), "+v"(z.x1));
            } else if constexpr (sub == 2) {                                // values 2, 3: ReLU, hi pair, lo pair
                z.x0 = __builtin_amdgcn_fmed3f(z.x0, 0.f, 65000.f);
                z.x1 = __builtin_amdgcn_fmed3f(z.x1, 0.f, 65000.f);
                float2v_t fv; fv[0] = z.x0; fv[1] = z.x1;
                z.dh = __builtin_bit_cast(uint32_t, __builtin_convertvector(fv, half2v_t));
                z.dl1 = lo_pair(z.dh, z.x0, z.x1);
            } else {                                                        // the group's two 8-byte stores
                const uint32_t ah = z.wb ^ (uint32_t)((m * 4 + g) << 4), al = z.wb ^ (uint32_t)((8 + m * 4 + g) << 4);
                const uint2 vh = make_uint2(z.hq, z.dh), vl = make_uint2(z.lq, z.dl1);
                if constexpr (!PRO && (FZ_ABL & 2)) { asm volatile("" :: "v"(ah), "v"(vh), "v"(al), "v"(vl)); }
                else {
                    asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(ah), "v"(vh), "i"(f * 8192) : "memory");
                    asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(al), "v"(vl), "i"(f * 8192) : "memory");
                }
            }
        }
    };
    // the micro-ops of stream step L
#ifdef SPL_STEP_TRACE      // separate build (tools/fz_ablate.sh trace): s_memtime at the end of positions 7, 15, 19, 23, 31, 63 (the values are read after the stream)
    unsigned long long ts[6] = {0, 0, 0, 0, 0, 0};
#endif
    auto fz_step = [&](auto LC) {
#ifdef SPL_STEP_TRACE
        {
            constexpr int P = decltype(LC)::value;
            constexpr int k = P == 7 ? 0 : P == 15 ? 1 : P == 19 ? 2 : P == 23 ? 3 : P == 31 ? 4 : P == 63 ? 5 : -1;
            if constexpr (k >= 0) asm volatile("s_memtime %0" : "=s"(ts[k]));
        }
#endif
        if constexpr (FUSE1A) {
            constexpr int L = decltype(LC)::value;
            spl_for_each<0, FZ_PER_STEP>([&](auto KC) {
                constexpr int id = kFzSched.op[L][decltype(KC)::value];
                if constexpr (id >= 0) fz_op(std::integral_constant<int, id>{}, std::false_type{});
            });
            constexpr bool has_e = kFzSched.op[L][0] >= FZ_E || kFzSched.op[L][1] >= FZ_E || kFzSched.op[L][2] >= FZ_E || kFzSched.op[L][3] >= FZ_E;
            if constexpr (has_e && spl_dense(L)) {
                // behind each of the step's first three MFMAs five of the sub-units' instructions (the v_fma_mix pairs and the stores are inline asm,
                // which the scheduler leaves behind their operands)
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            } else if constexpr (has_e) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            }
        }
    };

    int t = wg;
    SplTileIx cur_ix, nxt_ix;
    {
        cur_ix.b = t / tiles_per_img;
        cur_ix.r = t - cur_ix.b * tiles_per_img;
        decode(cur_ix);
        nxt_ix = cur_ix;
        advance(nxt_ix);
    }
    if constexpr (FUSE1A) {
        __syncthreads();                                   // the table, conv1a's fragments, the zeros
        if (t < total) {                                   // the workgroup's first tile, outside any stream: every micro-op in order, every wait lgkmcnt(0)
            fz_b = cur_ix.b; fz_ty0 = cur_ix.ty * TH; fz_tx0 = cur_ix.tx * 32; fz_which = 0;
            fz_issue();
            spl_for_each<0, FZ_NOPS>([&](auto IC) {
                constexpr int pos = decltype(IC)::value;
                // program order of the chains: PARK, then per fragment V, U x 5, T x 5, W x 4, K, M x 4, E x 32
                constexpr int f = pos < 53 ? 0 : 1, q = pos < 53 ? pos : pos - 52;      // fragment 0: positions 0-52 (with PARK), fragment 1: 53-104
                constexpr int id = pos == 0 ? FZ_PARK :
                                   q == 1 ? FZ_V + f :
                                   q < 7 ? FZ_U + 5 * f + (q - 2) :
                                   q < 12 ? FZ_T + 5 * f + (q - 7) :
                                   q < 16 ? FZ_W + 4 * f + (q - 12) :
                                   q == 16 ? FZ_K + f :
                                   q < 21 ? FZ_M + 4 * f + (q - 17) : FZ_E + 32 * f + (q - 21);
                fz_op(std::integral_constant<int, id>{}, std::true_type{});
                if constexpr (id < FZ_E) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            });
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    } else {
        if (t < total) {
            const SplOrg o = origin(cur_ix.b, cur_ix.ty, cur_ix.tx);
            spl_for_each<0, PPW>([&](auto JC) { dma_piece(o, 0, JC); });
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();

    // the wave's biases: cin = 64: the four register groups of its 32 channels; cin = 128: the two groups it finishes
    float4 bs[C128 ? 2 : 4];
#pragma unroll
    for (int g = 0; g < (C128 ? 2 : 4); ++g) bs[g] = *reinterpret_cast<const float4*>(bias_lds + co * 32 + (C128 ? 16 * part : 0) + 8 * g + 4 * hh);
    // the biases have landed before the first stream starts -- and hipcc must know it: behind the loop's back edge it otherwise puts an s_waitcnt lgkmcnt(2..3)
    // for these reads in front of the first two epilogue parts of EVERY tile, which drains the fragment ring there (seven reads in flight -> three)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int g = 0; g < (C128 ? 2 : 4); ++g) asm volatile("" : "+v"(bs[g].x), "+v"(bs[g].y), "+v"(bs[g].z), "+v"(bs[g].w));

    using Epi = SplEpi<C128, POOL>;
    Epi e;
#pragma unroll
    for (int i = 0; i < Epi::NV; ++i) e.v[i] = 0.f;
#pragma unroll
    for (int i = 0; i < (C128 ? 4 : 1); ++i) e.p[i] = spl_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < Epi::NPO; ++i) e.off[i] = SPL_NO_STORE;                    // nothing pending before the first tile
    const uint32_t out_img_bytes = (uint32_t)((int64_t)Hof * Wof * opix);
    e.img = __builtin_amdgcn_make_buffer_rsrc(out, 0, out_img_bytes, 0x00020000);
    e.pk.a = make_uint4(0, 0, 0, 0); e.pk.b = e.pk.a; e.qh = e.ql = 0;

    // cin = 128, K split over the wave pair (co, 0) / (co, 1): wave `part` finishes register groups 2 part, 2 part + 1 (channels [16 part, +16) of
    // the fragment, both rows) and hands the partner the other two.  The hand-over rides on the barrier that ends the tile: partial sums ->
    // LDS -> [vmcnt(0), barrier] -> four ds_reads issued in front of the next tile's fragment reads (LDS returns in order: the ring's first
    // wait covers them) -> added inside the slices.  The s_barrier at step SPL_MID_STEP orders those reads before the next tile's writes.
    float4* const xch_mine = reinterpret_cast<float4*>(smem_raw + 2 * SPL_BUF_BYTES) + ((co * 2 + part) * 4) * 64 + lane;
    const uint32_t xch_theirs = lds0 + 2 * SPL_BUF_BYTES + (((co * 2 + (part ^ 1)) * 4) * 64 + lane) * 16;

    auto read_partner = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(e.p[i]) : "v"(xch_theirs), "i"(i * 1024) : "memory");
    };
    const bool tr = trace != nullptr && blockIdx.x == 0 && (tid == 0 || tid == 192);
    int tk = -2;                                   // the trace skips the first two tiles
    auto stamp = [&](int i) {
        if (tr && tk >= 0 && tk < 4) {
            const unsigned long long v = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            trace[(tid ? 32 : 0) + tk * 8 + i] = v;
        }
    };

    int cur = 0;
    for (; t < total; t += nwg, cur ^= 1) {
        stamp(0);
        // the next tile's DMA (the last tile of this workgroup loads its own tile again: no branch in the stream; nobody reads that buffer)
        const bool has_next = t + nwg < total;                          // (field by field: a select between the two structs sends them to scratch)
        const SplOrg org_n = (dbg & 2) ? origin(0, 0, 0) : origin(has_next ? nxt_ix.b : cur_ix.b, has_next ? nxt_ix.ty : cur_ix.ty, has_next ? nxt_ix.tx : cur_ix.tx);
        if constexpr (FUSE1A) {                                          // the next tile's patch load flies under the first fifth of the stream
            fz_b = has_next ? nxt_ix.b : cur_ix.b; fz_ty0 = (has_next ? nxt_ix.ty : cur_ix.ty) * TH; fz_tx0 = (has_next ? nxt_ix.tx : cur_ix.tx) * 32;
            fz_which = cur ^ 1;
            fz_issue();
        }
        const uint32_t row_base = lds0 + cur * SPL_BUF_BYTES + n_eff * 256;
        floatx16 acc[2];
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[f][i] = 0.f;
        half8_t fb[SPL_NB];
        // cin = 128: the partner's partial sums of the PREVIOUS tile (written before the barrier that ended it), in front of this tile's fragment reads and
        // consumed in the slices (steps >= 8: the ring's waits cover them; LDS returns in order).  Read HERE, in the block that consumes them -- not behind the
        // barrier at the end of the previous iteration: registers written by an asm read must not cross the loop's back edge before their data has landed (a
        // copy there is legal for hipcc and wrong for the kernel; profiles/r05q_*).  The first tile reads what nobody wrote: its epilogue stores nothing.
        if constexpr (C128) read_partner();
        spl_prime(row_base, n_eff, hh, fb);
        stamp(1);
        __builtin_amdgcn_sched_barrier(0);
        spl_steps<0, Epi::NPK * spl_parts_per_pair<OUT_F32>(), FUSE1A ? 0 : PPW, C128, std::conditional_t<FUSE1A, SplFuseSched, SplPlainSched>>(
            row_base, n_eff, hh, wreg, acc, fb,
            [&](auto SC) {
                // the first part sits behind step 8's ring wait: the partner's sums have landed (LDS returns in order) -- and hipcc must not use them any
                // earlier: to the compiler the asm reads above DEFINED e.p, nothing stops it from adding e.v + e.p at the top of the block.  An empty asm
                // that redefines them here (no instruction) pins every use behind this point of the stream
                if constexpr (C128 && decltype(SC)::value == 0) asm volatile("" : "+v"(e.p[0]), "+v"(e.p[1]), "+v"(e.p[2]), "+v"(e.p[3]));
                spl_epi_part<decltype(SC)::value, C128, POOL, OUT_F32>(e, bs, inv, relu, hh, part);
            },
            [&](auto JC) { dma_piece(org_n, cur ^ 1, JC); }, fz_step);
        stamp(2);
#ifdef SPL_STEP_TRACE
        if (tr && tk >= 0 && tk < 4) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int k = 0; k < 6; ++k) trace[64 + (tid ? 32 : 0) + tk * 8 + k] = ts[k];
        }
#endif

        // this tile's raw values and addresses become the pending epilogue
        const int b = cur_ix.b, ty0 = cur_ix.ty * (TRN ? 32 : TH), tx0 = cur_ix.tx * (TRN ? TH : 32);
        const int ox = tx0 + n;
        e.img = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(out) + (int64_t)b * out_img_bytes, 0, out_img_bytes, 0x00020000);
        if constexpr (!C128) {
            // the wave owns rows ty0 + 2 part, + 1 and all 16 registers of its fragment
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int i = 0; i < 16; ++i) e.v[16 * f + i] = acc[f][i];
            const int oy = ty0 + 2 * part;
            if constexpr (POOL) {
                e.off[0] = ((oy < H) && (ox < W) && !(n & 1)) ? (uint32_t)((oy >> 1) * Wof + (ox >> 1)) * (uint32_t)opix + oorg : SPL_NO_STORE;
            } else {
#pragma unroll
                for (int f = 0; f < 2; ++f)
                    e.off[f] = ((oy + f < H) && (ox < W)) ? (uint32_t)((oy + f) * Wof + ox) * (uint32_t)opix + oorg : SPL_NO_STORE;
            }
        } else {
            auto put = [&](auto PC, auto FC, auto GC) {
                constexpr int P = decltype(PC)::value, f = decltype(FC)::value, gg = decltype(GC)::value;
                constexpr int r0 = 4 * (2 * (1 - P) + gg), q0 = 4 * (2 * P + gg);
                const float4 w4 = make_float4(acc[f][r0], acc[f][r0 + 1], acc[f][r0 + 2], acc[f][r0 + 3]);
                xch_mine[(f * 2 + gg) * 64] = w4;
                e.v[8 * f + 4 * gg + 0] = acc[f][q0]; e.v[8 * f + 4 * gg + 1] = acc[f][q0 + 1];
                e.v[8 * f + 4 * gg + 2] = acc[f][q0 + 2]; e.v[8 * f + 4 * gg + 3] = acc[f][q0 + 3];
            };
            auto put_all = [&](auto PC) {
                put(PC, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
                put(PC, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
                put(PC, std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
                put(PC, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
            };
            if (part == 0) put_all(std::integral_constant<int, 0>{}); else put_all(std::integral_constant<int, 1>{});
            if constexpr (POOL) {
                e.off[0] = ((ty0 < H) && (ox < W) && !(n & 1)) ? (uint32_t)((ty0 >> 1) * Wof + (ox >> 1)) * (uint32_t)opix + oorg : SPL_NO_STORE;
            } else {
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    const int oy = TRN ? ty0 + n : ty0 + f, oxx = TRN ? tx0 + f : ox;
                    e.off[f] = ((oy < H) && (oxx < W)) ? (uint32_t)(oy * Wof + oxx) * (uint32_t)opix + oorg : SPL_NO_STORE;
                }
            }
        }
        if (dbg & 1) {
#pragma unroll
            for (int i = 0; i < Epi::NPO; ++i) e.off[i] = SPL_NO_STORE;
        }
        cur_ix = nxt_ix;
        advance(nxt_ix);
        stamp(3);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");       // next tile landed (and the previous tile's stores, issued early in the stream, retired)
        stamp(4);
        __syncthreads();
        stamp(5);
        ++tk;
    }
    if constexpr (C128) {
        read_partner();
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(e.p[0]), "+v"(e.p[1]), "+v"(e.p[2]), "+v"(e.p[3]) :: "memory");
    }
    spl_epi_all<0, C128, POOL, OUT_F32>(e, bs, inv, relu, hh, part);       // the last tile's
}

template <bool C128, bool POOL, bool OUT_F32, bool TRN = false, bool FUSE1A = false, bool FZMIX = true>
static int launch_split(hipStream_t st, const ConvArgs& a, const SplFuse& fz = SplFuse{}) {
    if constexpr (C128 && !POOL && !TRN) {
        // the tile orientation with fewer tiles (OMNI_SPLIT_TRN=0/1 forces one: A/B hook; it fixes the order the taps are summed in)
        static const int force = config_process()[CFG_SPLIT_TRN];
        const int plain = cdiv(a.W, 32) * cdiv(a.H, 2), trn = cdiv(a.H, 32) * cdiv(a.W, 2);
        if (force == 1 || (force < 0 && trn < plain)) return launch_split<C128, POOL, OUT_F32, true>(st, a);
    }
    auto kfn = conv3x3_split_kernel<C128, POOL, OUT_F32, TRN, FUSE1A, FZMIX>;
    static DynSmemState smem_state;
    OMNI_HIP_TRY(ensure_dyn_smem(smem_state, (const void*)kfn, SPL_SMEM));
    constexpr int TH = C128 ? 2 : 4;
    const int tiles_x = cdiv(a.W, TRN ? TH : 32), tiles_y = cdiv(a.H, TRN ? 32 : TH), n_cg = a.cout / 64;
    // the tiles of an image that run: all of them, or all but the rectangle the caller already holds (ConvArgs::skip_*, in THIS kernel's tile grid)
    const bool skip = !TRN && a.skip_ty1 > a.skip_ty0 && a.skip_tx1 > a.skip_tx0;       // (a transposed-tile launch recomputes a rectangle it is offered: same values)
    OMNI_REQUIRE(!skip || (a.skip_ty0 >= 0 && a.skip_ty1 <= tiles_y && a.skip_tx0 >= 0 && a.skip_tx1 <= tiles_x), OMNI_ERR_INVALID, "conv_split: skip rectangle outside the tile grid");
    SplSkip sk;
    sk.y0 = skip ? a.skip_ty0 : 0; sk.y1 = skip ? a.skip_ty1 : 0; sk.x0 = skip ? a.skip_tx0 : 0; sk.w = skip ? a.skip_tx1 - a.skip_tx0 : 0;
    sk.bw = tiles_x - sk.w;
    sk.act = tiles_x * tiles_y - (sk.y1 - sk.y0) * sk.w;
    sk.n_above = skip ? sk.y0 * tiles_x : sk.act;
    sk.n_upto = sk.n_above + (sk.y1 - sk.y0) * sk.bw;
    OMNI_REQUIRE(sk.act > 0, OMNI_ERR_INVALID, "conv_split: the skip rectangle covers the whole image");
    auto magic = [](int d) { return d > 1 ? (uint32_t)(((1ull << 32) + (uint64_t)d - 1) / (uint64_t)d) : 0u; };      // exact for n * d < 2^32; 0 = divisor 1
    sk.magic_tx = magic(tiles_x); sk.magic_bw = magic(sk.bw);
    sk.xcd = config_process()[CFG_CONV_XCD];
    const int total = a.batch * sk.act;
    int per_cg = a.n_cu / n_cg;
    if (per_cg < 1) per_cg = 1;
    if (per_cg > total) per_cg = total;
    const float inv = a.out_f32 ? a.split_inv / SPL_ACT_SCALE : a.split_inv;
    static const bool want_trace = config_process()[CFG_SPLIT_TRACE] != 0;
    static const int dbg = config_process()[CFG_SPLIT_DBG];
    static unsigned long long* trace_dev = nullptr;
    if (want_trace) {
        if (!trace_dev) OMNI_HIP_TRY(hipMalloc((void**)&trace_dev, 128 * 8));
        OMNI_HIP_TRY(hipMemsetAsync(trace_dev, 0, 128 * 8, st));
    }
    hipLaunchKernelGGL(kfn, dim3(per_cg * n_cg), dim3(256), SPL_SMEM, st, reinterpret_cast<const char*>(a.in), a.out,
                       reinterpret_cast<const _Float16*>(a.w_packed), a.bias, inv, a.H, a.W, a.cout, n_cg, tiles_x, tiles_y, a.batch, a.relu ? 1 : 0,
                       sk, dbg, want_trace ? trace_dev : nullptr, fz);
    OMNI_LAUNCH_CHECK();
    if (want_trace) {
        unsigned long long h[128];
        OMNI_HIP_TRY(hipMemcpyAsync(h, trace_dev, sizeof(h), hipMemcpyDeviceToHost, st));
        OMNI_HIP_TRY(hipStreamSynchronize(st));
        static int launches = 0;
        if (total >= 8 * per_cg && launches++ < 2)  // per instantiation: the first launches with at least eight tiles per workgroup
            for (int w = 0; w < 2; ++w)
                for (int k = 0; k < 4; ++k) {
                    const unsigned long long* q = h + w * 32 + k * 8;
#ifdef SPL_STEP_TRACE
                    {
                        const unsigned long long* u = h + 64 + w * 32 + k * 8;
                        fprintf(stderr, "split step trace c128=%d pool=%d f32=%d fuse1a=%d wave %d tile %d: positions 0-7 %llu | 8-15 %llu | 16-19 %llu | 20-23 %llu | 24-31 %llu | 32-63 %llu | 64-95 %llu\n",
                                (int)C128, (int)POOL, (int)OUT_F32, (int)FUSE1A, w * 3, k + 2, u[0] - q[1], u[1] - u[0], u[2] - u[1], u[3] - u[2], u[4] - u[3], u[5] - u[4], q[2] - u[5]);
                    }
#endif
                    fprintf(stderr, "split trace c128=%d pool=%d f32=%d trn=%d fuse1a=%d H=%d W=%d cout=%d wave %d tile %d: origin+prime %llu stream %llu hand-over %llu vmcnt %llu barrier %llu | total %llu\n",
                            (int)C128, (int)POOL, (int)OUT_F32, (int)TRN, (int)FUSE1A, a.H, a.W, a.cout, w * 3, k + 2, q[1] - q[0], q[2] - q[1], q[3] - q[2], q[4] - q[3],
                            q[5] - q[4], q[5] - q[0]);
                }
    }
    return OMNI_OK;
}

// a.in: split-64 activations (x SPL_ACT_SCALE); a.w_packed / a.split_inv from conv_pack_weights_split; a.bias: out_f32 ? the layer's bias
// : SPL_ACT_SCALE * bias; a.out: out_f32 ? NHWC fp32 (true values) : split-64 (x SPL_ACT_SCALE)
int conv_split(hipStream_t st, const ConvArgs& a) {
    OMNI_REQUIRE(a.ksize == 3 && (a.cin == 64 || a.cin == 128) && a.cout % 64 == 0, OMNI_ERR_INVALID, "conv_split: cin=%d cout=%d ksize=%d", a.cin, a.cout, a.ksize);
    OMNI_REQUIRE(!a.pool || (a.H % 2 == 0 && a.W % 2 == 0), OMNI_ERR_INVALID, "pooling needs even H, W");
    OMNI_REQUIRE(a.n_cu > 0 && a.split_inv > 0.f, OMNI_ERR_INVALID, "conv_split: n_cu / split_inv not set");
    OMNI_REQUIRE(!(a.pool && a.out_f32), OMNI_ERR_INVALID, "conv_split: pool + fp32 output not instantiated");
    OMNI_REQUIRE((int64_t)split_frame_h(a.H) * split_frame_w(a.W) * (a.cin == 128 ? 512 : 256) < (1ll << 32) &&
                     (int64_t)split_frame_h(a.H) * split_frame_w(a.W) * a.cout * 4 < (1ll << 31), OMNI_ERR_INVALID,
                 "conv_split: image too large for 32-bit pixel offsets");
    if (a.cin == 64) {
        if (a.pool) return launch_split<false, true, false>(st, a);
        return a.out_f32 ? launch_split<false, false, true>(st, a) : launch_split<false, false, false>(st, a);
    }
    if (a.pool) return launch_split<true, true, false>(st, a);
    return a.out_f32 ? launch_split<true, false, true>(st, a) : launch_split<true, false, false>(st, a);
}

float conv_split_act_scale() { return SPL_ACT_SCALE; }

// ==================================================================================================================================================
// convDa in OMNI_PREC_SPLIT ONLY at the coarse cells around the key points (superpoint.ipynb:183: computeDescriptors reads cDa at the four cells
// around each key point and nowhere else -- <= 800 of 4 500 cells; the fp16 path's conv3x3_c128_sparse_kernel, conv.hip, with split operands).
// Same weights, same wave roles and the SAME order of summation as the dense cin = 128 kernel above, so the values are bit-identical to the
// dense layer's at those cells (tests/test_gpu_superpoint.py::test_fp32_sparse_descriptor_head_is_bit_identical_to_the_dense_map_path):
//   workgroup = 64 output channels (a quarter of convDa's 256; blockIdx & 3) x a stream of tiles; tile = the 32 corner cells of 8 key points;
//   wave (co, part) = 32 output channels x input block `part` (64 channels), its 36 wh + 36 wl fragments in 288 registers; per cell the dense
//   kernel sums (tap column outer, tap row inner [transposed tiles: row outer, column inner], per tap: hi fragments kq = 0..3 against wh then wl,
//   lo fragments against wh) into a zero accumulator, the pair's two halves of K meet through LDS, v = fmaf(own + partner, inv, bias), ReLU.
// A cell's 3 x 3 x 128-channel neighbourhood is 4.6 KB of split-64 activations (the zero frame makes every tap a plain read): 147 KB per tile
// does not fit twice, so the taps stream through LDS in three CHUNKS of one outer index (3 taps x 512 B per cell, 48.5 KB per buffer, two
// buffers): thread = (cell, eighth) fetches 12 16-byte pieces of the next chunk into registers behind the 36 MFMAs of the current one.
// out: compact rows [image][key point][corner][256] fp32 (what sp_gather_cells_kernel made from the dense map); cells outside the map / beyond
// n_kps are not written, tiles without key points are not walked.
#define SSP_CELL_BYTES 1552                       // 3 taps x 512 B + 16: an odd number of 16-byte slots, conflict-free fragment reads
#define SSP_BUF_BYTES (32 * SSP_CELL_BYTES)       // 49 664
#define SSP_XCH_OFF (2 * SSP_BUF_BYTES)
#define SSP_SMEM (2 * SSP_BUF_BYTES + 8192)
#define SSP_KP 8                                  // key points per tile (CSP_KP of conv.hip)

// step S of a chunk (tap slot i = S / 8, fragment kg = S % 8): fragment S + 2 issued, S waited for, its MFMAs (the rs_steps pattern)
template <int S, int O>
__device__ __forceinline__ void ssp_steps(uint32_t base, const half8_t (&wreg)[72], floatx16& acc, half8_t (&fb)[3]) {
    if constexpr (S < 24) {
        constexpr int i = S / 8, kg = S % 8, kq = kg & 3, t = (i * 3 + O) * 4 + kq;
        if constexpr (S + 2 < 24) {
            constexpr int i2 = (S + 2) / 8, kg2 = (S + 2) % 8;
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[(S + 2) % 3]) : "v"(base), "i"(i2 * 512 + kg2 * 32));
        }
        asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(fb[S % 3]) : "i"((S + 2 < 24) ? 2 : (S + 1 < 24 ? 1 : 0)));
        __builtin_amdgcn_sched_barrier(0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[t], fb[S % 3], acc, 0, 0, 0);
        if constexpr (kg < 4) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[36 + t], fb[S % 3], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        ssp_steps<S + 1, O>(base, wreg, acc, fb);
    }
}

template <bool TRN>
__global__ void __launch_bounds__(256, 1)
conv3x3_split_c128_sparse_kernel(const char* __restrict__ in /* split-64 frames [B][Hf][Wf][2][256 B] */, const _Float16* __restrict__ wp, const float* __restrict__ bias,
                                 float inv, int Hc, int Wc, int g32_first, int W, int H, int max_num, const float* __restrict__ kps_xy,
                                 const int* __restrict__ n_kps, float* __restrict__ out, int tiles_per_img, int n_tiles) {
    extern __shared__ __attribute__((aligned(256))) char smem_raw[];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem_raw;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int co = wave & 1, part = wave >> 1;
    const int n = lane & 31, hh = lane >> 5;
    const int quarter = blockIdx.x & 3, wg = blockIdx.x >> 2, nwg = gridDim.x >> 2;
    const int g32 = g32_first + quarter * 2 + co;
    half8_t wreg[72];
    {
        const _Float16* wbase = wp + ((int64_t)g32 * 2 + part) * (2 * 36 * 512) + lane * 8;
#pragma unroll
        for (int s = 0; s < 72; ++s) {
            const int hl = s / 36, tk = (s % 36) / 4, kq = s % 4;
            const int tap = TRN ? (tk % 3) * 3 + tk / 3 : tk;
            wreg[s] = *reinterpret_cast<const half8_t*>(wbase + ((hl * 9 + tap) * 4 + kq) * 512);
        }
    }
    float4 bs[2];                                   // the two register groups this wave finishes: channels 32 g32 + 16 part + 8 gg + 4 hh + (0..3)
#pragma unroll
    for (int gg = 0; gg < 2; ++gg) bs[gg] = *reinterpret_cast<const float4*>(bias + g32 * 32 + 16 * part + 8 * gg + 4 * hh);
    const int Wf = split_frame_w(Wc), Hf = split_frame_h(Hc);
    const int64_t img_bytes = (int64_t)Hf * Wf * 512;
    const int64_t row_bytes = (int64_t)Wf * 512;
    const float fW = (float)W, fH = (float)H, fWc = (float)Wc, fHc = (float)Hc;
    // corner cell `slot` of tile t -> (image, key point, cell x, cell y); false: no such key point / outside the map (sample_corner's arithmetic)
    auto cell_of = [&](int t, int slot, int& b, int& kp, int& cx, int& cy) __attribute__((always_inline)) -> bool {
        b = t / tiles_per_img;
        kp = (t - b * tiles_per_img) * SSP_KP + (slot >> 2);
        cx = cy = 0;
        if (kp >= n_kps[b]) return false;
        const float kx = kps_xy[((int64_t)b * max_num + kp) * 2 + 0], ky = kps_xy[((int64_t)b * max_num + kp) * 2 + 1];
        const float gx = __fsub_rn(__fdiv_rn(__fmul_rn(2.0f, kx), fW), 1.0f);
        const float gy = __fsub_rn(__fdiv_rn(__fmul_rn(2.0f, ky), fH), 1.0f);
        const float ix = __fdiv_rn(__fsub_rn(__fmul_rn(__fadd_rn(gx, 1.0f), fWc), 1.0f), 2.0f);
        const float iy = __fdiv_rn(__fsub_rn(__fmul_rn(__fadd_rn(gy, 1.0f), fHc), 1.0f), 2.0f);
        cx = (int)floorf(ix) + (slot & 1); cy = (int)floorf(iy) + ((slot >> 1) & 1);
        return cx >= 0 && cx < Wc && cy >= 0 && cy < Hc;
    };
    // the workgroup's tiles: t = wg, wg + nwg, ... without the tiles that hold no key point (uniform: n_kps is read by every thread alike)
    auto next_tile = [&](int t) __attribute__((always_inline)) -> int {
        while (t < n_tiles) {
            const int b = t / tiles_per_img;
            if ((t - b * tiles_per_img) * SSP_KP < n_kps[b]) break;
            t += nwg;
        }
        return t;
    };
    // gather: thread = (cell tid >> 3, eighth tid & 7): pieces j * 8 + eighth, j < 12, of the cell's 96 16-byte pieces of a chunk (tap slot j >> 2,
    // bytes ((j & 3) * 8 + eighth) * 16 of the pixel's 512).  Named registers (as an array indexed in unrolled loops hipcc keeps them in scratch)
    const int gcell = tid >> 3, gpart = tid & 7;
    uint4 s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11;
    const char* fsrc = in;                          // the cell's own pixel in the frame (a cell that does not exist: pixel (1, 1) of image 0 -- read, never used)
#define SSP_SRC(O, J) (fsrc + (TRN ? ((O) - 1) * row_bytes + (((J) >> 2) - 1) * 512 : (((J) >> 2) - 1) * row_bytes + ((O) - 1) * 512) + (((J) & 3) * 8 + gpart) * 16)
#define SSP_FETCH(O)                                                                                                          \
    {                                                                                                                          \
        s0 = *reinterpret_cast<const uint4*>(SSP_SRC(O, 0)); s1 = *reinterpret_cast<const uint4*>(SSP_SRC(O, 1));             \
        s2 = *reinterpret_cast<const uint4*>(SSP_SRC(O, 2)); s3 = *reinterpret_cast<const uint4*>(SSP_SRC(O, 3));             \
        s4 = *reinterpret_cast<const uint4*>(SSP_SRC(O, 4)); s5 = *reinterpret_cast<const uint4*>(SSP_SRC(O, 5));             \
        s6 = *reinterpret_cast<const uint4*>(SSP_SRC(O, 6)); s7 = *reinterpret_cast<const uint4*>(SSP_SRC(O, 7));             \
        s8 = *reinterpret_cast<const uint4*>(SSP_SRC(O, 8)); s9 = *reinterpret_cast<const uint4*>(SSP_SRC(O, 9));             \
        s10 = *reinterpret_cast<const uint4*>(SSP_SRC(O, 10)); s11 = *reinterpret_cast<const uint4*>(SSP_SRC(O, 11));         \
    }
#define SSP_DST(J) (pdst + ((J) >> 2) * 512 + ((J) & 3) * 128)
#define SSP_PARK(WHICH)                                                                                                       \
    {                                                                                                                          \
        char* pdst = smem_raw + (WHICH) * SSP_BUF_BYTES + gcell * SSP_CELL_BYTES + gpart * 16;                                 \
        *reinterpret_cast<uint4*>(SSP_DST(0)) = s0; *reinterpret_cast<uint4*>(SSP_DST(1)) = s1; *reinterpret_cast<uint4*>(SSP_DST(2)) = s2;    \
        *reinterpret_cast<uint4*>(SSP_DST(3)) = s3; *reinterpret_cast<uint4*>(SSP_DST(4)) = s4; *reinterpret_cast<uint4*>(SSP_DST(5)) = s5;    \
        *reinterpret_cast<uint4*>(SSP_DST(6)) = s6; *reinterpret_cast<uint4*>(SSP_DST(7)) = s7; *reinterpret_cast<uint4*>(SSP_DST(8)) = s8;    \
        *reinterpret_cast<uint4*>(SSP_DST(9)) = s9; *reinterpret_cast<uint4*>(SSP_DST(10)) = s10; *reinterpret_cast<uint4*>(SSP_DST(11)) = s11; \
    }
    auto aim = [&](int t) __attribute__((always_inline)) {          // fsrc := the gather cell of tile t
        int b, kp, cx, cy;
        const bool ok = cell_of(t, gcell, b, kp, cx, cy);
        fsrc = ok ? in + b * img_bytes + (cy + 1) * row_bytes + (int64_t)(cx + 1) * 512 : in + row_bytes + 512;
    };
    float4* const xch_mine = reinterpret_cast<float4*>(smem_raw + SSP_XCH_OFF) + ((co * 2 + part) * 2) * 64 + lane;
    const float4* const xch_theirs = reinterpret_cast<const float4*>(smem_raw + SSP_XCH_OFF) + ((co * 2 + (part ^ 1)) * 2) * 64 + lane;

    int t = next_tile(wg);
    if (t < n_tiles) { aim(t); SSP_FETCH(0) SSP_PARK(0) }
    __syncthreads();
    int cur = 0;
    while (t < n_tiles) {
        const int tn = next_tile(t + nwg);
        floatx16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        // one chunk: the next chunk's pieces into registers, this chunk's 36 MFMAs, the pieces into the other buffer, barrier
#define SSP_CHUNK(O, FETCH_NEXT)                                                                                              \
        {                                                                                                                      \
            FETCH_NEXT                                                                                                         \
            const uint32_t base = lds0 + cur * SSP_BUF_BYTES + n * SSP_CELL_BYTES + part * 256 + hh * 16;                      \
            half8_t fb[3];                                                                                                     \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                 \
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[0]) : "v"(base), "i"(0));                                   \
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[1]) : "v"(base), "i"(32));                                  \
            __builtin_amdgcn_sched_barrier(0);                                                                                 \
            ssp_steps<0, O>(base, wreg, acc, fb);                                                                              \
        }
        SSP_CHUNK(0, SSP_FETCH(1))
        SSP_PARK(cur ^ 1)
        __syncthreads();
        cur ^= 1;
        SSP_CHUNK(1, SSP_FETCH(2))
        SSP_PARK(cur ^ 1)
        __syncthreads();
        cur ^= 1;
        SSP_CHUNK(2, if (tn < n_tiles) { aim(tn); SSP_FETCH(0) })
        // the pair's hand-over: this wave finishes register groups 2 part, 2 part + 1 and gives the partner the other two
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
            const int r0 = 4 * (2 * (1 - part) + gg);
            xch_mine[gg * 64] = part == 0 ? make_float4(acc[8 + 4 * gg], acc[9 + 4 * gg], acc[10 + 4 * gg], acc[11 + 4 * gg])
                                          : make_float4(acc[4 * gg], acc[1 + 4 * gg], acc[2 + 4 * gg], acc[3 + 4 * gg]);
            (void)r0;
        }
        if (tn < n_tiles) SSP_PARK(cur ^ 1)
        __syncthreads();
        cur ^= 1;
        {
            int b, kp, cx, cy;
            const bool ok = cell_of(t, n, b, kp, cx, cy);
            if (ok) {
                float* op = out + (((int64_t)b * max_num + kp) * 4 + (n & 3)) * 256 + (g32 - g32_first) * 32 + 16 * part + 4 * hh;
#pragma unroll
                for (int gg = 0; gg < 2; ++gg) {
                    const float4 p4 = xch_theirs[gg * 64];
                    const float a0 = part == 0 ? acc[4 * gg] : acc[8 + 4 * gg], a1 = part == 0 ? acc[4 * gg + 1] : acc[9 + 4 * gg];
                    const float a2 = part == 0 ? acc[4 * gg + 2] : acc[10 + 4 * gg], a3 = part == 0 ? acc[4 * gg + 3] : acc[11 + 4 * gg];
                    float4 x;
                    x.x = fmaxf(fmaf(a0 + p4.x, inv, bs[gg].x), 0.f); x.y = fmaxf(fmaf(a1 + p4.y, inv, bs[gg].y), 0.f);
                    x.z = fmaxf(fmaf(a2 + p4.z, inv, bs[gg].z), 0.f); x.w = fmaxf(fmaf(a3 + p4.w, inv, bs[gg].w), 0.f);
                    *reinterpret_cast<float4*>(op + 8 * gg) = x;
                }
            }
        }
        t = tn;
    }
#undef SSP_SRC
#undef SSP_FETCH
#undef SSP_DST
#undef SSP_PARK
#undef SSP_CHUNK
}

// a4b: split-64 frames of the Hc x Wc x 128 map; w_packed / split_inv / bias: the FUSED heads layer's (convPa | convDa, cout 512: convDa = g32 8..15);
// out: [batch][max_num][4][256] fp32
int conv_split_c128_sparse(hipStream_t st, const omni_ctx* ctx, const void* a4b, const void* w_packed, const float* bias, float split_inv, int Hc, int Wc,
                           int g32_first, int W, int H, int max_num, const float* kps_xy, const int* n_kps, float* out, int batch) {
    const int tiles_per_img = cdiv(max_num, SSP_KP), n_tiles = tiles_per_img * batch;
    const int cus = ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256;
    int groups = cus / 4;                                                // four workgroups (channel quarters) per tile stream
    if (groups > n_tiles) groups = n_tiles;
    if (groups < 1) groups = 1;
    OMNI_REQUIRE((int64_t)batch * split_frame_h(Hc) * split_frame_w(Wc) * 512 < (1ll << 40), OMNI_ERR_INVALID, "conv_split_c128_sparse: map too large");
    // the orientation launch_split picks for the dense layer of this shape (it fixes the order the taps are summed in)
    static const int force = config_process()[CFG_SPLIT_TRN];
    const int plain = cdiv(Wc, 32) * cdiv(Hc, 2), trn = cdiv(Hc, 32) * cdiv(Wc, 2);
    const bool use_trn = force == 1 || (force < 0 && trn < plain);
    const float inv = split_inv / SPL_ACT_SCALE;
    auto launch = [&](auto kfn) -> int {
        static DynSmemState attr;
        OMNI_HIP_TRY(ensure_dyn_smem(attr, (const void*)kfn, SSP_SMEM));
        hipLaunchKernelGGL(kfn, dim3(4 * groups), dim3(256), SSP_SMEM, st, (const char*)a4b, (const _Float16*)w_packed, bias, inv, Hc, Wc, g32_first, W, H, max_num,
                           kps_xy, n_kps, out, tiles_per_img, n_tiles);
        OMNI_LAUNCH_CHECK();
        return OMNI_OK;
    };
    return use_trn ? launch(conv3x3_split_c128_sparse_kernel<true>) : launch(conv3x3_split_c128_sparse_kernel<false>);
}

// conv1a (from the u8 image, built tile by tile inside the kernel) + conv1b + ReLU + 2x2 max-pool in one launch; a = the conv1b layer (a.in unused)
int conv1ab_split_fused(hipStream_t st, const ConvArgs& a, const uint8_t* gray, int gstride, int fisheye_mask, const void* w1a_frag, const uint32_t* lut_hl) {
    OMNI_REQUIRE(a.ksize == 3 && a.cin == 64 && a.cout % 64 == 0 && a.pool && !a.out_f32 && a.H % 2 == 0 && a.W % 8 == 0, OMNI_ERR_INVALID, "conv1ab_split_fused: bad layer shape");
    OMNI_REQUIRE(a.n_cu > 0 && a.split_inv > 0.f, OMNI_ERR_INVALID, "conv1ab_split_fused: n_cu / split_inv not set");
    OMNI_REQUIRE(gstride % 4 == 0 && gstride >= a.W && ((uintptr_t)gray & 3) == 0 && (int64_t)a.batch * a.H * gstride < (1ll << 31), OMNI_ERR_INVALID,
                 "conv1ab_split_fused: image rows must be 4-byte aligned (stride %d)", gstride);
    OMNI_REQUIRE((int64_t)split_frame_h(a.H / 2) * split_frame_w(a.W / 2) * a.cout * 4 < (1ll << 31), OMNI_ERR_INVALID, "conv1ab_split_fused: image too large for 32-bit pixel offsets");
    SplFuse fz;
    fz.gray = gray; fz.gstride = gstride; fz.gbytes = a.batch * a.H * gstride;
    omni_fisheye_mask_rows(a.H, fisheye_mask, &fz.mask_r0, &fz.mask_r1);   // cv::Rect(0, rows*3/4, cols, rows/4)
    fz.w1a_frag = reinterpret_cast<const _Float16*>(w1a_frag); fz.lut_hl = lut_hl;
    return launch_split<false, true, false, false, true>(st, a, fz);
}

// ---------------------------------------------------------------------------------------------------------------
// conv1a (1 -> 64 channels, 3x3, ReLU) from the u8 image, exact fp32 FMAs, written as split-64 activations x SPL_ACT_SCALE
// (the structure of conv1a_kernel in conv.hip: lane = (pixel, group of 8 output channels))
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
conv1a_split_kernel(const uint8_t* __restrict__ gray, int stride, int H, int W, int mask_row0, int mask_row1, const float* __restrict__ w,
                    const float* __restrict__ bias, const float* __restrict__ lut, _Float16* __restrict__ out, int skip_tr0, int skip_ntr) {
    __shared__ float tile[10][36];
    __shared__ float wsm[9][64];
    __shared__ float bsm[64];
    __shared__ float lsm[256];
    const int tid = threadIdx.x;
    const int tiles_x = (W + 31) / 32;
    int tr = blockIdx.x / tiles_x;
    if (tr >= skip_tr0) tr += skip_ntr;                    // the tile rows [skip_tr0, skip_tr0 + skip_ntr) already stand in `out` (the mask's constant band)
    const int ty0 = tr * 8, tx0 = (blockIdx.x % tiles_x) * 32;
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
            _Float16* o = out + (((int64_t)b * split_frame_h(H) + gy + 1) * split_frame_w(W) + gx + 1) * 128 + cg * 8;
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
                 const float* lut, void* out, int skip_tr0, int skip_tr1) {
    int r0, r1;
    omni_fisheye_mask_rows(H, fisheye_mask, &r0, &r1);   // cv::Rect(0, rows*3/4, cols, rows/4)
    const int tiles_y = cdiv(H, 8), skip_n = skip_tr1 > skip_tr0 ? skip_tr1 - skip_tr0 : 0;
    OMNI_REQUIRE(skip_n == 0 || (skip_tr0 >= 0 && skip_tr1 <= tiles_y && skip_n < tiles_y), OMNI_ERR_INVALID, "conv1a_split: skipped tile rows outside the image");
    dim3 grid(cdiv(W, 32) * (tiles_y - skip_n), batch);
    hipLaunchKernelGGL(conv1a_split_kernel, grid, dim3(256), 0, st, gray, stride, H, W, r0, r1, w, bias, lut, (_Float16*)out, skip_n ? skip_tr0 : tiles_y, skip_n);
    OMNI_LAUNCH_CHECK();
    return OMNI_OK;
}

// test hook: framed split-64 NHWC (x SPL_ACT_SCALE) -> NCHW fp32 (true values)
__global__ void split_to_nchw_f32_kernel(const _Float16* __restrict__ in, float* __restrict__ out, int C, int H, int W, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int HW = H * W, Hf = split_frame_h(H), Wf = split_frame_w(W);
    const int64_t b = i / ((int64_t)C * HW);
    const int64_t r = i - b * (int64_t)C * HW;
    const int c = (int)(r / HW);
    const int p = (int)(r - (int64_t)c * HW);
    const int y = p / W, x = p - y * W;
    const _Float16* px = in + (((b * Hf + y + 1) * Wf + x + 1) * C) * 2 + (c >> 6) * 128 + (c & 63);
    out[i] = ((float)px[0] + (float)px[64]) * (1.0f / SPL_ACT_SCALE);
}
int split_to_nchw_f32(hipStream_t st, const void* in, float* out, int batch, int C, int H, int W) {
    const int64_t total = (int64_t)batch * C * H * W;
    hipLaunchKernelGGL(split_to_nchw_f32_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, st, reinterpret_cast<const _Float16*>(in), out, C, H, W, total);
    OMNI_LAUNCH_CHECK();
    return OMNI_OK;
}

}  // namespace omni
