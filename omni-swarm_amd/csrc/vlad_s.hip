// MobileNetVLAD inverted-residual blocks on the fp16 matrix cores with SPLIT operands: fp32-class results at v_mfma_f32_32x32x16_f16 rate.
// Replaces, for the shapes instantiated below, the exact-f32 kernels of vlad.hip (v_mfma_f32_32x32x2_f32 / fp32 VALU) behind
// MobileNetVLADTensorRT::inference (swarm_loop/src/mobilenetvlad_tensorrt.cpp:4-14).
//
// Every fp32 operand of the two pointwise convolutions is carried as a pair of halfs, v = hi + lo (hi = half(v), lo = half(v - hi), |v - hi - lo|
// <= 2^-22 |v|), and a product is three matrix-core terms, a.b ~ a_hi.b_hi + a_lo.b_hi + a_hi.b_lo (the dropped lo.lo term is 2^-22 relative),
// accumulated in fp32 -- the scheme conv1a of SuperPoint uses (conv.hip).  Nothing else is rounded: the hidden activations live in LDS as fp32,
// the depthwise conv is fp32 VALU arithmetic with fp32 weights, bias and residual are fp32, tensors in HBM are fp32.
//
// One launch per block; one workgroup per 8x8 (stride 1) / 8x4 (stride 2) output tile; the hidden layer is walked in chunks of 48 channels
// (every 6x-expanded width is a multiple of 48: no padded channels anywhere in the VALU phases) that never leave LDS:
//     expand    h[region px][48] = ReLU6([x_hi | x_lo | 1 1] . [We_hi ; We_hi ; be_hi be_lo] + x_hi . We_lo)      wave = one 32-pixel tile
//     depthwise d[out px][48]    = ReLU6(dw3x3(h) + bd), split into d_hi / d_lo                                   thread = (channel pair, 4x2 / 2x2 px)
//     project   acc[cout][px]   += Wp_hi . d_hi + Wp_hi . d_lo + Wp_lo . d_hi                                     fp32 accumulators across the chunks
//   * the expand bias rides in two K slots that are 1 at in-image pixels: out-of-image region pixels are all-zero rows, so h = ReLU6(0) = 0
//     there -- the zero padding of the depthwise conv -- without mask code;
//   * weights are pre-packed in MFMA fragment order and fetched straight into registers one phase ahead of their use;
//   * LDS rows are odd multiples of 16 bytes (conflict-free ds_read_b128 fragment reads); two barriers per chunk.
#include "config.h"
#include "common.h"
#include "vlad_h.h"

namespace omni {

typedef _Float16 sh8 __attribute__((ext_vector_type(8)));
typedef _Float16 sh4 __attribute__((ext_vector_type(4)));
typedef _Float16 sh2 __attribute__((ext_vector_type(2)));
typedef float sf16 __attribute__((ext_vector_type(16)));
typedef float sf2 __attribute__((ext_vector_type(2)));
typedef float sf4 __attribute__((ext_vector_type(4)));

#define SB_CH 48                      // hidden channels per chunk
#define SB_HS 208                     // bytes per pixel row of h: 48 floats + 16
#define SB_DS 112                     // bytes per pixel row of d_hi / d_lo: 48 halfs + 16
#define SB_WD (10 * SB_CH)            // floats of depthwise taps + bias per chunk

// n / d through the launcher's m = ceil(2^32 / d) (0 encodes d == 1): exact for n < 2^20, d < 2^12
__device__ __forceinline__ int sb_div(int n, unsigned m) { return m ? (int)__umulhi((unsigned)n, m) : n; }

template <int STRIDE, int CIN, int NT>
struct SBlockCfg {
    static constexpr int S1 = (2 * CIN + 2 + 15) / 16, S2 = (CIN + 15) / 16;        // k-steps: [x_hi | x_lo | 1 1] and x_hi again (for We_lo)
    static constexpr int XS = S1 * 32 + 16;
    static constexpr int TH = STRIDE == 1 ? 8 : 4, OPX = 8 * TH, NTN = OPX / 32;
    static constexpr int RWX = 7 * STRIDE + 3, RHY = (TH - 1) * STRIDE + 3, R = RWX * RHY, RT = (R + 31) / 32, RP = RT * 32;
    // one wave per 32-pixel region tile -- but never more than FOUR waves: at the ~210-250 registers these kernels need, a SIMD holds two waves, a CU
    // eight; a five-wave workgroup (the stride-2 region: 17 x 9 = 153 pixels = 5 tiles) then fits ONCE per CU where a four-wave one fits twice
    // (round 5: b1 / b3 / b6 / b13 ran at half the residency of the other blocks).  The fifth region tile is a second expand pass of wave 0, region
    // pixels 128.. a second x -> LDS pass of the first threads.
    static constexpr int NW = RT < 4 ? RT : 4, THREADS = NW * 64, XP = (R + THREADS / 2 - 1) / (THREADS / 2);
    static constexpr int CB = (S1 + S2) * 2048 + NT * 6144;                           // bytes of matrix fragments per chunk
    static constexpr int MQ = STRIDE == 1 ? (NT + 1) / 2 : 1;                         // projection m-tiles per wave
    // LDS: x rows | d_hi, d_lo | h rows | depthwise taps.  Only the R real region rows are kept (the MFMA tiles of the last wave read past the x
    // rows: garbage in lanes whose results are dropped)
    static constexpr size_t smem(int hid) { return (size_t)R * XS + 2 * (size_t)OPX * SB_DS + (size_t)R * SB_HS + (size_t)(hid / SB_CH) * SB_WD * 4; }
};

// waves per SIMD the register allocation aims at: 2 = up to 256 registers (what these kernels take when they may: 210-256, no spills); SB_WPE_SMALL (A/B
// build switch for the cin <= 16 shapes) 3 = 168 registers, three workgroups per CU, at the price of 150-230 bytes of scratch per lane
#ifndef SB_WPE_SMALL
#define SB_WPE_SMALL 2
#endif
template <int STRIDE, int CIN, int NT>
__global__ void __launch_bounds__((SBlockCfg<STRIDE, CIN, NT>::THREADS), (CIN > 32 ? 1 : (CIN <= 16 ? SB_WPE_SMALL : 2)))
vlad_sblock_kernel(VladSBlockArgs a) {
    using C = SBlockCfg<STRIDE, CIN, NT>;
    constexpr int S1 = C::S1, S2 = C::S2, XS = C::XS, TH = C::TH, OPX = C::OPX, RWX = C::RWX, R = C::R, CB = C::CB, MQ = C::MQ, NW = C::NW, RT = C::RT, XP = C::XP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int n_chunks = a.hid / SB_CH;
    char* xin = smem;                                        // [R][XS]  halfs: x_hi | x_lo | 1 1 0...
    char* dhi = smem + R * XS;                               // [OPX][112] halfs
    char* dlo = dhi + OPX * SB_DS;
    char* h = dlo + OPX * SB_DS;                             // [R][208] floats
    float* wdl = reinterpret_cast<float*>(h + R * SB_HS);    // [chunk][10][48] depthwise taps + bias
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, kk = lane >> 5;
    const int tiles_x = (a.Wo + 7) >> 3, tiles_y = (a.Ho + TH - 1) / TH, tiles_img = tiles_x * tiles_y;
    const char* blob = reinterpret_cast<const char*>(a.blob);
    const char* frag = blob + (size_t)n_chunks * SB_WD * 4;  // per-chunk matrix fragments

    // expand fragments of one chunk -> registers
    sh8 we1[S1][2], we2[S2][2];
    auto load_we = [&](int c) {
        const char* cb = frag + (int64_t)c * CB + lane * 16;
#pragma unroll
        for (int ks = 0; ks < S1; ++ks)
#pragma unroll
            for (int m = 0; m < 2; ++m) we1[ks][m] = *reinterpret_cast<const sh8*>(cb + (ks * 2 + m) * 1024);
#pragma unroll
        for (int ks = 0; ks < S2; ++ks)
#pragma unroll
            for (int m = 0; m < 2; ++m) we2[ks][m] = *reinterpret_cast<const sh8*>(cb + ((S1 + ks) * 2 + m) * 1024);
    };
    load_we(0);
    {   // depthwise taps of every chunk: once per workgroup
        const sf4* ws = reinterpret_cast<const sf4*>(blob);
        for (int e = tid; e < n_chunks * (SB_WD / 4); e += C::THREADS) reinterpret_cast<sf4*>(wdl)[e] = ws[e];
    }
    const int pair = tid % 24, blk = tid / 24;               // depthwise work item (threads 0..191): channel pair x block of output pixels
    const int nt = STRIDE == 1 ? (wave & 1) : 0;             // projection tiles of this wave: pixel tile nt, m-tiles mt0 + MSTEP * q
    const int mt0 = STRIDE == 1 ? (wave >> 1) : wave;
    constexpr int MSTEP = STRIDE == 1 ? 2 : NW;
    const bool pwave = mt0 < NT;
    sf4 pbias[MQ][4];
#pragma unroll
    for (int q = 0; q < MQ; ++q)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ch = (mt0 + MSTEP * q) * 32 + 8 * g + 4 * kk;
            const sf4 z = {0.f, 0.f, 0.f, 0.f};
            pbias[q][g] = ch < a.cout ? *reinterpret_cast<const sf4*>(a.bp + ch) : z;
        }

    // tile -> (image, first output row, first output column): multiply-high with the launcher's reciprocals, no integer divisions
    auto tile_origin = [&](int tile, int& tb, int& toy, int& tox) {
        tb = sb_div(tile, a.m_img);
        const int ttr = tile - tb * tiles_img, tty = sb_div(ttr, a.m_tx);
        toy = tty * TH; tox = (ttr - tty * tiles_x) * 8;
    };
    // the t-th tile that RUNS -> its number in the full grid (the band of the fisheye mask's constant region is not walked: VladSBlockArgs::sk_*)
    const int act_total = a.sk_act * a.batch;
    auto full_tile = [&](int t) -> int {
        if (a.sk_y1 <= a.sk_y0) return t;
        const int tb = sb_div(t, a.m_act), r = t - tb * a.sk_act;
        int ttr;
        if (r < a.sk_above) ttr = r;
        else if (r < a.sk_upto) {
            const int q = r - a.sk_above, ry = sb_div(q, a.m_bw), c = q - ry * a.sk_bw;
            ttr = (a.sk_y0 + ry) * tiles_x + (c < a.sk_x0 ? c : c + a.sk_w);
        } else ttr = r - a.sk_upto + a.sk_y1 * tiles_x;
        return tb * tiles_img + ttr;
    };
    // input region of a tile: thread = (region pixel, half of its channels); fetched into registers one tile ahead
    const int xhf = tid & 1;
    sf4 xpre[XP][CIN / 8];                                   // (XP = 1 unless the region has more pixels than half the threads: stride 2)
    bool xin_img[XP];
    auto fetch_x = [&](int tile) {
        int tb, toy, tox;
        tile_origin(tile, tb, toy, tox);
#pragma unroll
        for (int ps = 0; ps < XP; ++ps) {
            const int xr = (tid >> 1) + ps * (C::THREADS / 2), xry = xr / RWX, xrx = xr - xry * RWX;
            const int gy = toy * STRIDE - 1 + xry, gx = tox * STRIDE - 1 + xrx;
            xin_img[ps] = xr < R && gy >= 0 && gy < a.Hi && gx >= 0 && gx < a.Wi;
            const unsigned off = (unsigned)((tb * a.Hi + gy) * a.Wi + gx) * CIN + xhf * (CIN / 2);      // 32-bit element offset (checked by the launcher)
#pragma unroll
            for (int q = 0; q < CIN / 8; ++q) {
                const sf4 z = {0.f, 0.f, 0.f, 0.f};
                xpre[ps][q] = xin_img[ps] ? *reinterpret_cast<const sf4*>(a.in + off + q * 4) : z;
            }
        }
    };
    // this lane's output pixel of a tile, its residual (stride 1, cin == cout: the block input at the same pixel) and the final store.  The
    // store of tile t is issued at the top of tile t + 1, BEFORE that iteration's global loads: the s_waitcnt vmcnt(0) in front of the next
    // x -> LDS conversion then only sees operations that are a whole tile old (loads and stores share the counter on this part)
    sf16 acc[MQ];
    sf4 resv[MQ][4];
    unsigned out_off = 0;
    bool out_ok = false;
    auto out_pixel = [&](int tile, unsigned& off_out, unsigned& off_res) {
        int tb, toy, tox;
        tile_origin(tile, tb, toy, tox);
        const int o = nt * 32 + n, oy = toy + (o >> 3), ox = tox + (o & 7);
        off_out = (unsigned)((tb * a.Ho + oy) * a.Wo + ox) * (unsigned)a.cout;
        off_res = (unsigned)((tb * a.Hi + oy) * a.Wi + ox) * (unsigned)CIN;
        return pwave && oy < a.Ho && ox < a.Wo;
    };
    auto fetch_res = [&](bool ok, unsigned off_res) {
        if (!a.res) return;
#pragma unroll
        for (int q = 0; q < MQ; ++q)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ch = (mt0 + MSTEP * q) * 32 + 8 * g + 4 * kk;
                const sf4 z = {0.f, 0.f, 0.f, 0.f};
                resv[q][g] = (ok && ch < a.cout) ? *reinterpret_cast<const sf4*>(a.in + off_res + ch) : z;
            }
    };
    auto store_out = [&]() {
        if (!out_ok || (a.dbg & 1)) return;
#pragma unroll
        for (int q = 0; q < MQ; ++q)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ch = (mt0 + MSTEP * q) * 32 + 8 * g + 4 * kk;
                if (ch >= a.cout) continue;
                sf4 v = pbias[q][g];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += acc[q][4 * g + j];
                if (a.res) v += resv[q][g];
                *reinterpret_cast<sf4*>(a.out + out_off + ch) = v;
            }
    };

    // expand: this wave's 32 region pixels x the chunk's 48 channels (m-tile 0: channels 0-31, m-tile 1: 32-47 + 16 zero rows)
    auto expand = [&]() {
#pragma unroll
      for (int rt = 0; rt < (RT + NW - 1) / NW; ++rt) {
        const int rw = wave + rt * NW;                       // this pass's region tile (wave-uniform)
        if (rw >= RT) break;
        const char* xb = xin + (rw * 32 + n) * XS + kk * 16;
        sf16 e0, e1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { e0[r] = 0.f; e1[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < S1; ++ks) {
            const sh8 xb8 = *reinterpret_cast<const sh8*>(xb + ks * 32);
            e0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(we1[ks][0], xb8, e0, 0, 0, 0);
            e1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(we1[ks][1], xb8, e1, 0, 0, 0);
        }
#pragma unroll
        for (int ks = 0; ks < S2; ++ks) {
            const sh8 xb8 = *reinterpret_cast<const sh8*>(xb + ks * 32);
            e0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(we2[ks][0], xb8, e0, 0, 0, 0);
            e1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(we2[ks][1], xb8, e1, 0, 0, 0);
        }
        if (rw * 32 + n < R) {
            char* hp = h + (rw * 32 + n) * SB_HS + kk * 16;      // channels 8 g + 4 kk + (0..3) of pixel n
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                sf4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = __builtin_amdgcn_fmed3f(e0[4 * g + j], 0.f, 6.f);
                *reinterpret_cast<sf4*>(hp + g * 32) = v;
            }
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                sf4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = __builtin_amdgcn_fmed3f(e1[4 * g + j], 0.f, 6.f);
                *reinterpret_cast<sf4*>(hp + 128 + g * 32) = v;
            }
        }
      }
    };

    if ((int)blockIdx.x < act_total) fetch_x(full_tile(blockIdx.x));
#define TR(k) do { if (trw) a.trace[wave * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
    for (int t = blockIdx.x; t < act_total; t += gridDim.x) {
        const int tile = full_tile(t);
        const bool trw = a.trace && blockIdx.x == 0 && (tid & 63) == 0 && t == (int)blockIdx.x + 2 * (int)gridDim.x;
        TR(8);
#pragma unroll
        for (int ps = 0; ps < XP; ++ps) {
            const int xr = (tid >> 1) + ps * (C::THREADS / 2);
            if (xr < R) {   // x_hi | x_lo, the two bias slots, zero pad -> LDS
                char* row = xin + xr * XS;
#pragma unroll
                for (int q = 0; q < CIN / 8; ++q) {
                    const sf4 v = xpre[ps][q];
                    const sh4 hi = __builtin_convertvector(v, sh4);
                    const sh4 lo = __builtin_convertvector(v - __builtin_convertvector(hi, sf4), sh4);
                    *reinterpret_cast<sh4*>(row + (xhf * (CIN / 2) + q * 4) * 2) = hi;
                    *reinterpret_cast<sh4*>(row + CIN * 2 + (xhf * (CIN / 2) + q * 4) * 2) = lo;
                }
                sh8 t = {0, 0, 0, 0, 0, 0, 0, 0};
                if (xhf == 0 && xin_img[ps]) { t[0] = (_Float16)1.f; t[1] = (_Float16)1.f; }
                *reinterpret_cast<sh8*>(row + CIN * 4 + xhf * 16) = t;             // hf 0: the bias slots, hf 1: the 16 pad bytes behind them
            }
        }
        // global traffic of this iteration, oldest first: the previous tile's result, this tile's residual, the next tile's input
        store_out();
        {
            unsigned off_res;
            out_ok = out_pixel(tile, out_off, off_res);
            fetch_res(out_ok, off_res);
        }
        if (t + (int)gridDim.x < act_total && !(a.dbg & 2)) fetch_x(full_tile(t + gridDim.x));
#pragma unroll
        for (int p = 0; p < MQ; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;

        __syncthreads();                                     // xin complete (and every wave is past the previous tile's projection)
        TR(0);
        expand();
        TR(1);
        __syncthreads();
        TR(2);

        for (int c = 0; c < n_chunks; ++c) {
            const char* cb = frag + (int64_t)c * CB + (S1 + S2) * 2048 + lane * 16;
            sh8 wph[MQ][3], wpl[MQ][3];
#pragma unroll
            for (int q = 0; q < MQ; ++q) {
                const int m = mt0 + MSTEP * q;
#pragma unroll
                for (int ks = 0; ks < 3; ++ks) {
                    const sh8 z = {0, 0, 0, 0, 0, 0, 0, 0};
                    wph[q][ks] = m < NT ? *reinterpret_cast<const sh8*>(cb + ((m * 3 + ks) * 2 + 0) * 1024) : z;
                    wpl[q][ks] = m < NT ? *reinterpret_cast<const sh8*>(cb + ((m * 3 + ks) * 2 + 1) * 1024) : z;
                }
            }
            if (n_chunks > 1) load_we(c + 1 < n_chunks ? c + 1 : 0);      // next chunk's (after the last chunk: the next tile's first) expand fragments
            // ---- depthwise 3x3 + ReLU6 in fp32, result split into (hi, lo) halfs
            if (tid < 192) {
                const float* wd = wdl + c * SB_WD + pair * 2;
                sf2 w9[9];
#pragma unroll
                for (int t = 0; t < 9; ++t) w9[t] = *reinterpret_cast<const sf2*>(wd + t * SB_CH);
                const sf2 bias = *reinterpret_cast<const sf2*>(wd + 9 * SB_CH);
                constexpr int BW = STRIDE == 1 ? 4 : 2, BH = 2;                       // outputs per work item
                constexpr int IW = (BW - 1) * STRIDE + 3, IH = (BH - 1) * STRIDE + 3; // 6x4 (stride 1) / 5x5 (stride 2) inputs
                const int by = STRIDE == 1 ? (blk >> 1) : (blk >> 2), bx = STRIDE == 1 ? (blk & 1) * 4 : (blk & 3) * 2;
                const char* hp = h + ((by * BH * STRIDE) * RWX + bx * STRIDE) * SB_HS + pair * 8;
                sf2 v[IH][IW];
#pragma unroll
                for (int y = 0; y < IH; ++y)
#pragma unroll
                    for (int x = 0; x < IW; ++x) v[y][x] = *reinterpret_cast<const sf2*>(hp + (y * RWX + x) * SB_HS);
#pragma unroll
                for (int oy = 0; oy < BH; ++oy)
#pragma unroll
                    for (int ox = 0; ox < BW; ++ox) {
                        sf2 o = bias;
#pragma unroll
                        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                            for (int dx = 0; dx < 3; ++dx) o = __builtin_elementwise_fma(v[oy * STRIDE + dy][ox * STRIDE + dx], w9[dy * 3 + dx], o);
                        sf2 r; r[0] = __builtin_amdgcn_fmed3f(o[0], 0.f, 6.f); r[1] = __builtin_amdgcn_fmed3f(o[1], 0.f, 6.f);
                        const sh2 hi = __builtin_convertvector(r, sh2);
                        const sh2 lo = __builtin_convertvector(r - __builtin_convertvector(hi, sf2), sh2);
                        const int off = ((by * BH + oy) * 8 + bx + ox) * SB_DS + pair * 4;
                        *reinterpret_cast<sh2*>(dhi + off) = hi;
                        *reinterpret_cast<sh2*>(dlo + off) = lo;
                    }
            }
            TR(3);
            __syncthreads();                                 // d complete; every reader is done with h
            TR(4);
            // ---- projection of this chunk (reads d), then expand of the next one (writes h)
            if (pwave) {
                const int doff = (nt * 32 + n) * SB_DS + kk * 16;
                sh8 bh[3], bl[3];
#pragma unroll
                for (int ks = 0; ks < 3; ++ks) {
                    bh[ks] = *reinterpret_cast<const sh8*>(dhi + doff + ks * 32);
                    bl[ks] = *reinterpret_cast<const sh8*>(dlo + doff + ks * 32);
                }
#pragma unroll
                for (int q = 0; q < MQ; ++q) {
                    if (mt0 + MSTEP * q < NT) {
#pragma unroll
                        for (int ks = 0; ks < 3; ++ks) {
                            acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wph[q][ks], bh[ks], acc[q], 0, 0, 0);
                            acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wph[q][ks], bl[ks], acc[q], 0, 0, 0);
                            acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wpl[q][ks], bh[ks], acc[q], 0, 0, 0);
                        }
                    }
                }
            }
            TR(5);
            if (c + 1 < n_chunks) {
                expand();
                __syncthreads();                             // h of the next chunk complete; every reader is done with d
            }
            // (after the last chunk no barrier: the other waves start on the next tile's x while the projection waves finish; d is next
            // written behind two more barriers)
            TR(6);
        }
        TR(7);
    }
#undef TR
    store_out();
}

static inline uint16_t sb_f2h(float v) { const __half hv = __float2half_rn(v); uint16_t u; memcpy(&u, &hv, 2); return u; }
static inline float sb_h2f(uint16_t u) { __half hv; memcpy(&hv, &u, 2); return __half2float(hv); }
static inline float sb_hi(float v) { return sb_h2f(sb_f2h(v)); }
static inline float sb_lo(float v) { return v - sb_hi(v); }

// instantiated shapes: (stride, cin, projection m-tiles)
#define SB_SHAPES(X) X(2, 8, 1) X(1, 8, 1) X(1, 16, 1) X(2, 16, 1) X(1, 24, 1) X(1, 32, 1) X(2, 32, 2) X(1, 56, 2) X(1, 56, 4)

bool vlad_sblock_supported(int cin, int hid, int cout, int stride) {
    if (hid < SB_CH || hid % SB_CH || cout % 4 || cout < 4) return false;
    const int nt0 = (cout + 31) / 32, nt = nt0 == 3 ? 4 : nt0;
#define X(S, CI, N) if (stride == S && cin == CI && nt == N) return true;
    SB_SHAPES(X)
#undef X
    return false;
}

static inline int sb_nt(int cout) { const int nt0 = (cout + 31) / 32; return nt0 == 3 ? 4 : nt0; }

size_t vlad_sblock_blob_bytes(int cin, int hid, int cout) {
    const int S1 = (2 * cin + 2 + 15) / 16, S2 = (cin + 15) / 16, NT = sb_nt(cout);
    return (size_t)(hid / SB_CH) * (SB_WD * 4 + (S1 + S2) * 2048 + NT * 6144);
}

// we [hid][cin], be [hid], wd [hid][9], bd [hid], wp [cout][hid] (the layer table's OIHW weights)
void vlad_sblock_pack(int cin, int hid, int cout, const float* we, const float* be, const float* wd, const float* bd, const float* wp, void* out) {
    const int S1 = (2 * cin + 2 + 15) / 16, S2 = (cin + 15) / 16, NT = sb_nt(cout), CB = (S1 + S2) * 2048 + NT * 6144;
    const int n_chunks = hid / SB_CH;
    memset(out, 0, vlad_sblock_blob_bytes(cin, hid, cout));
    float* wdo = reinterpret_cast<float*>(out);
    for (int ch = 0; ch < hid; ++ch) {
        float* q = wdo + (size_t)(ch / SB_CH) * SB_WD + ch % SB_CH;
        for (int t = 0; t < 9; ++t) q[t * SB_CH] = wd[(size_t)ch * 9 + t];
        q[9 * SB_CH] = bd[ch];
    }
    char* frag = reinterpret_cast<char*>(out) + (size_t)n_chunks * SB_WD * 4;
    for (int c = 0; c < n_chunks; ++c) {
        char* cb = frag + (size_t)c * CB;
        // expand A fragments: row = hidden channel (m-tile 0: chunk channels 0-31, m-tile 1: 32-47), k as in the LDS pixel row
        for (int pass = 0; pass < 2; ++pass)
            for (int ks = 0; ks < (pass ? S2 : S1); ++ks)
                for (int m = 0; m < 2; ++m)
                    for (int l = 0; l < 64; ++l)
                        for (int e = 0; e < 8; ++e) {
                            const int row = m * 32 + (l & 31), k = ks * 16 + (l >> 5) * 8 + e;
                            float v = 0.f;
                            if (row < SB_CH) {
                                const int ch = c * SB_CH + row;
                                if (!pass) {
                                    if (k < cin) v = sb_hi(we[(size_t)ch * cin + k]);                       // x_hi . We_hi
                                    else if (k < 2 * cin) v = sb_hi(we[(size_t)ch * cin + k - cin]);        // x_lo . We_hi
                                    else if (k == 2 * cin) v = sb_hi(be[ch]);
                                    else if (k == 2 * cin + 1) v = sb_lo(be[ch]);
                                } else if (k < cin) v = sb_lo(we[(size_t)ch * cin + k]);                    // x_hi . We_lo
                            }
                            reinterpret_cast<uint16_t*>(cb + (((pass ? S1 : 0) + ks) * 2 + m) * 1024)[l * 8 + e] = sb_f2h(v);
                        }
        // projection A fragments: row = output channel, k = hidden channel of the chunk, (hi, lo) per k-step
        for (int m = 0; m < NT; ++m)
            for (int ks = 0; ks < 3; ++ks)
                for (int part = 0; part < 2; ++part)
                    for (int l = 0; l < 64; ++l)
                        for (int e = 0; e < 8; ++e) {
                            const int co = m * 32 + (l & 31), hc = c * SB_CH + ks * 16 + (l >> 5) * 8 + e;
                            float v = 0.f;
                            if (co < cout) v = part ? sb_lo(wp[(size_t)co * hid + hc]) : sb_hi(wp[(size_t)co * hid + hc]);
                            reinterpret_cast<uint16_t*>(cb + (S1 + S2) * 2048 + ((m * 3 + ks) * 2 + part) * 1024)[l * 8 + e] = sb_f2h(v);
                        }
    }
}

template <int STRIDE, int CIN, int NT>
static int launch_sb(hipStream_t st, const VladSBlockArgs& a) {
    using C = SBlockCfg<STRIDE, CIN, NT>;
    static const size_t pad = (size_t)config_process()[CFG_VLAD_SB_LDSPAD];   // A/B hook: occupancy
    const size_t smem = C::smem(a.hid) + pad;
    OMNI_REQUIRE(smem <= 160 * 1024, OMNI_ERR_CAPACITY, "vlad_sblock: %zu B of LDS for hid=%d", smem, a.hid);
    auto kfn = vlad_sblock_kernel<STRIDE, CIN, NT>;
    static DynSmemState attr;
    OMNI_HIP_TRY(ensure_dyn_smem(attr, (const void*)kfn, smem));
    // persistent workgroups: as many as fit the CUs' LDS at once, each walking tiles blockIdx.x, + gridDim.x, ... with the next tile's input in flight
    const int tiles_x = cdiv(a.Wo, 8), tiles_y = cdiv(a.Ho, C::TH);
    const bool skip = a.sk_y1 > a.sk_y0 && a.sk_w > 0;
    OMNI_REQUIRE(!skip || (a.sk_y0 >= 0 && a.sk_y1 <= tiles_y && a.sk_x0 >= 0 && a.sk_x0 + a.sk_w <= tiles_x), OMNI_ERR_INVALID, "vlad_sblock: skip rectangle outside the tile grid");
    const int act_img = tiles_x * tiles_y - (skip ? (a.sk_y1 - a.sk_y0) * a.sk_w : 0);
    OMNI_REQUIRE(act_img > 0, OMNI_ERR_INVALID, "vlad_sblock: the skip rectangle covers the whole map");
    const int tiles = act_img * a.batch;                          // the tiles that run
    OMNI_REQUIRE((int64_t)a.batch * a.Hi * a.Wi * a.cin < (1ll << 31) && (int64_t)a.batch * a.Ho * a.Wo * a.cout < (1ll << 31) && tiles < (1 << 20),
                 OMNI_ERR_CAPACITY, "vlad_sblock: tensor beyond 32-bit element offsets");
    // how many of these workgroups a CU really holds: the runtime knows (registers AND LDS AND wave slots).  Rounds 2-4 estimated it from the LDS alone
    // (3 for b1 / b2 / b4 / b5) while the registers allow two waves per SIMD: a third of the persistent workgroups only started when the first ones had
    // finished their share -- 1.5 rounds of work in the time of 2
    static int per_cu_cached = 0;
    static size_t per_cu_smem = 0;
    if (!per_cu_cached || per_cu_smem != smem) {
        int nb = 0;
        OMNI_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)kfn, C::THREADS, smem));
        per_cu_cached = nb < 1 ? 1 : nb;
        per_cu_smem = smem;
    }
    const int per_cu = per_cu_cached;
    // OMNI_VLAD_SB_PERSIST (A/B hook): 0 = one tile per workgroup, N >= 1 = N x (CUs x resident workgroups per CU) workgroups
    const int persist = a.persist;                               // (the handle's snapshot of the table)
    const int64_t cap = (int64_t)a.n_cu * per_cu * (persist > 0 ? persist : 1);
    const int grid = (!persist || tiles < cap) ? tiles : (int)cap;
    static const bool want_trace = config_process()[CFG_VLAD_SB_TRACE] != 0;
    static unsigned long long* trace_dev = nullptr;
    VladSBlockArgs at = a;
    at.trace = nullptr;
    static const int dbg = config_process()[CFG_VLAD_SB_DBG];
    at.dbg = dbg;
    {   // sb_div()'s reciprocals
        const unsigned d_img = (unsigned)(cdiv(a.Wo, 8) * cdiv(a.Ho, C::TH)), d_tx = (unsigned)cdiv(a.Wo, 8);
        at.m_img = d_img > 1 ? (unsigned)(((1ull << 32) + d_img - 1) / d_img) : 0u;
        at.m_tx = d_tx > 1 ? (unsigned)(((1ull << 32) + d_tx - 1) / d_tx) : 0u;
        if (!skip) { at.sk_y0 = at.sk_y1 = at.sk_x0 = at.sk_w = 0; }
        at.sk_bw = tiles_x - at.sk_w; at.sk_act = act_img;
        at.sk_above = skip ? at.sk_y0 * tiles_x : act_img;
        at.sk_upto = at.sk_above + (at.sk_y1 - at.sk_y0) * at.sk_bw;
        at.m_act = act_img > 1 ? (unsigned)(((1ull << 32) + (unsigned)act_img - 1) / (unsigned)act_img) : 0u;
        at.m_bw = at.sk_bw > 1 ? (unsigned)(((1ull << 32) + (unsigned)at.sk_bw - 1) / (unsigned)at.sk_bw) : 0u;
    }
    if (want_trace) {
        if (!trace_dev) OMNI_HIP_TRY(hipMalloc((void**)&trace_dev, 8 * 16 * 8));
        OMNI_HIP_TRY(hipMemsetAsync(trace_dev, 0, 8 * 16 * 8, st));
        at.trace = trace_dev;
    }
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(C::THREADS), smem, st, at);
    OMNI_LAUNCH_CHECK();
    if (want_trace) {
        unsigned long long hbuf[8 * 16];
        OMNI_HIP_TRY(hipMemcpyAsync(hbuf, trace_dev, sizeof(hbuf), hipMemcpyDeviceToHost, st));
        OMNI_HIP_TRY(hipStreamSynchronize(st));
        static int launches = 0;
        if (launches++ == 8)                            // a warmed-up launch
            for (int w = 0; w < C::THREADS / 64; ++w) {
                const unsigned long long* t = hbuf + w * 16;
                if (!t[8]) continue;
                fprintf(stderr, "sb trace s%d cin%d grid %d wave %d (last chunk): x->lds+store+bar %llu expand %llu bar %llu | dw %llu bar %llu proj %llu | tile %llu\n",
                        STRIDE, CIN, grid, w, t[0] - t[8], t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4], t[7] - t[8]);
            }
    }
    return OMNI_OK;
}

int launch_vlad_sblock(hipStream_t st, const VladSBlockArgs& a, int stride) {
    const int nt = sb_nt(a.cout);
#define X(S, CI, N) if (stride == S && a.cin == CI && nt == N) return launch_sb<S, CI, N>(st, a);
    SB_SHAPES(X)
#undef X
    set_error("vlad_sblock: no instantiation for cin=%d cout=%d stride=%d", a.cin, a.cout, stride);
    return OMNI_ERR_INVALID;
}

}  // namespace omni
