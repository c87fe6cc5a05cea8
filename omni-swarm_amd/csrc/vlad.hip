// omni_vlad_*: drop-in for MobileNetVLADTensorRT (swarm_loop/include/swarm_loop/mobilenetvlad_tensorrt.h:6-22,
// swarm_loop/src/mobilenetvlad_tensorrt.cpp:4-14).
//
// !! ASSUMED ARCHITECTURE, PARITY UNPINNED !!  The reference ships only the I/O contract (u8 image -> float32 [0,255]
// with no scaling -> 4096 floats); the network body is the un-vendored HF-Net mobilenetvlad saved-model.  The layer
// table arrives through omni_vlad_weights so the real graph can be swapped in; oracle/mobilenetvlad_ref.py states the
// assumed one: (x-128)/128 tiled to 3 ch, MobileNetV2(0.35) to 112 ch at stride 32, NetVLAD K=32, FC->4096, L2.
//
// The net is ~0.34 GMAC/image (1.4 % of SuperPoint) so it is written as plain fp32 NHWC kernels: an LDS-tiled
// pointwise (1x1) conv, a depthwise 3x3, a stem conv, the NetVLAD aggregation and an HBM-bound FC that reads the
// 58.7 MB weight matrix once per batch.
#include "config.h"
#include "common.h"
#include "vlad_h.h"
#include "conv.h"          // conv_read_pixel_bytes / conv_fill_rect_bytes (the constant region of the fisheye mask)

struct VladLayerDev { int kind, cin, cout, stride, hin, win, hout, wout; float* w; float* b; };

struct VladFusedBlock {      // one inverted-residual block = [expand] + depthwise + project, weights point into `layers`
    int cin, hid, cout, stride, expand, res, hin, win, hout, wout;
    const float* bp;
    float* blob;             // device: packed per-chunk weights (see VladBlockArgs)
    float* mblob;            // device: the same per chunk with the projection padded to cop columns (VladMBlockArgs); null = not available
    void* hblob;             // device: fp16 fragment-order weights of vlad_hblock_kernel (vlad_h.hip); null = not available
    void* sblob;             // device: split-fp16 fragment-order weights of vlad_sblock_kernel (vlad_s.hip); null = not available
    int cop;
    const float *we_t, *be, *wd_t, *bd, *wp_t;   // the layers' own device weights ([cin][hid], [9][hid], [hid][cout]) for the MFMA path
};

struct omni_vlad {
    omni_ctx* ctx = nullptr;
    omni::Config cfg;                         // the switches as they stood when the handle was created (config.h)
    bool sblock = true;                       // blocks with an sblob run on vlad_sblock_kernel (OMNI_VLAD_SBLOCK=0 disables: A/B and parity tests)
    int prec = OMNI_PREC_F32;                 // OMNI_PREC_F16: blocks with an hblob run on vlad_hblock_kernel (omni_vlad_set_precision)
    bool fused = false;                       // every block has a fused kernel (OMNI_VLAD_UNFUSED=1 forces the layer-by-layer path)
    bool mfma_late = true;                    // low-resolution blocks on the f32-MFMA pointwise path (OMNI_VLAD_MFMA=0 disables)
    int mblock_max_px = 0;                    // blocks whose INPUT has at most this many pixels per image run on vlad_mblock_kernel (OMNI_VLAD_MBLOCK_PX)
    int mfma_max_px = 2048;                   // ... = blocks whose input has at most this many pixels per image (OMNI_VLAD_MFMA_PX)
    std::vector<VladFusedBlock> blocks;
    int W = 0, H = 0, max_batch = 0, K = 0, Dm = 0, out_dim = 0, hf = 0, wf = 0;
    std::vector<VladLayerDev> layers;
    float *assign_wT = nullptr, *assign_b = nullptr, *clusters = nullptr, *fc_w = nullptr, *fc_b = nullptr;
    float* mb_partial = nullptr; size_t mb_partial_bytes = 0; int mb_cpw = 1;   // hidden-layer split of vlad_mblock_kernel
    float *fc_wp = nullptr, *fc_part = nullptr;   // FC weights in MFMA B-operand order + K-split partial tiles (vlad_fc_mfma_kernel); null: VALU path
    bool fc_mfma = false;
    float *buf[3] = {nullptr, nullptr, nullptr};   // rotating activation buffers
    size_t buf_elems = 0;
    float *assign = nullptr, *vlad = nullptr, *out = nullptr;
    uint8_t* gray_stage = nullptr;
    omni::HostBuf hstage;
    // The constant region of the fisheye mask (round 6; the SuperPoint side: superpoint.hip, omni_sp::MaskSkip).  LoopCam blanks the bottom quarter of the
    // frame BEFORE both networks run (loop_cam.cpp:536-539, 556-558): inside the band, one 3x3 tap in from its borders per convolution, the output of the
    // stem and of every block is one constant vector.  A planned layer writes into a buffer of its own (mskip[k].buf: the rotating buffers are shared between
    // layers) whose rectangle is filled once, from a dense pass over a blank masked frame (vlad_calibrate_mask_skip), and the tiles inside the rectangle are
    // left out of the tile walk: bit-identical to the dense pass (tests/test_gpu_vlad_detector.py).  mskip[0] = stem + block 0, mskip[k] = block k.
    struct MaskSkip {
        int ty0 = 0, ty1 = 0, tx0 = 0, tx1 = 0;       // tile rectangle in the layer's output tile grid
        int oy0 = 0, oy1 = 0, ox0 = 0, ox1 = 0;       // the same in output pixels
        int oh = 0, ow = 0, oc = 0;
        float* buf = nullptr;                         // [max_batch][oh][ow][oc]
        void* vec = nullptr;                          // [oc] the constant
        double frac = 0.0;                            // the rectangle's share of the layer's tiles
    };
    std::vector<MaskSkip> mskip;
    bool mask_skip_ready = false;
    uint8_t* zero_gray = nullptr;
    std::mutex mu;
};

namespace omni {

__device__ __forceinline__ float relu6f(float v) { return fminf(fmaxf(v, 0.f), 6.f); }

// stem: u8 -> (x-128)/128, 3x3 stride-2 pad-1 conv with the 3 identical input channels folded into one, ReLU6
__global__ void __launch_bounds__(256)
vlad_stem_kernel(const uint8_t* __restrict__ gray, int stride, int H, int W, int mask0, int mask1, int Ho, int Wo, int cout,
                 int cstride, const float* __restrict__ w /*[cout][9]*/, const float* __restrict__ bias, float* __restrict__ out) {
    const int b = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= Ho * Wo) return;
    const int oy = p / Wo, ox = p - oy * Wo;
    const uint8_t* g = gray + (int64_t)b * stride * H;
    float v[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int y = oy * cstride - 1 + t / 3, x = ox * cstride - 1 + t % 3;
        float px = 0.f;                                    // zero padding is applied AFTER normalisation
        if (y >= 0 && y < H && x >= 0 && x < W) {
            const float raw = (y >= mask0 && y < mask1) ? 0.f : (float)g[(int64_t)y * stride + x];
            px = (raw - 128.0f) / 128.0f;
        }
        v[t] = px;
    }
    float* o = out + ((int64_t)b * Ho * Wo + p) * cout;
    for (int c = 0; c < cout; ++c) {
        float acc = bias[c];
#pragma unroll
        for (int t = 0; t < 9; ++t) acc = fmaf(v[t], w[c * 9 + t], acc);
        o[c] = relu6f(acc);
    }
}

// depthwise 3x3 (pad 1, stride 1|2) + ReLU6; thread = (pixel, 4 channels)
__global__ void __launch_bounds__(256)
vlad_dw_kernel(const float* __restrict__ in, int Hi, int Wi, int C, int Ho, int Wo, int cstride,
               const float* __restrict__ w /*[9][C]*/, const float* __restrict__ bias, float* __restrict__ out, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c4 = C >> 2;
    const int cg = (int)(i % c4);
    const int64_t p = i / c4;
    const int ox = (int)(p % Wo);
    const int oy = (int)((p / Wo) % Ho);
    const int64_t b = p / ((int64_t)Wo * Ho);
    float4 acc = *reinterpret_cast<const float4*>(bias + cg * 4);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int y = oy * cstride - 1 + t / 3, x = ox * cstride - 1 + t % 3;
        if (y < 0 || y >= Hi || x < 0 || x >= Wi) continue;
        const float4 v = *reinterpret_cast<const float4*>(in + ((b * Hi + y) * Wi + x) * C + cg * 4);
        const float4 ww = *reinterpret_cast<const float4*>(w + t * C + cg * 4);
        acc.x = fmaf(v.x, ww.x, acc.x); acc.y = fmaf(v.y, ww.y, acc.y); acc.z = fmaf(v.z, ww.z, acc.z); acc.w = fmaf(v.w, ww.w, acc.w);
    }
    acc.x = relu6f(acc.x); acc.y = relu6f(acc.y); acc.z = relu6f(acc.z); acc.w = relu6f(acc.w);
    *reinterpret_cast<float4*>(out + p * C + cg * 4) = acc;
}

// pointwise conv: [P x cin] x [cin x cout], 64-pixel x 32-channel tile per workgroup, K staged 64 at a time.
// act: 0 linear, 1 relu6; res != nullptr adds the residual.
__global__ void __launch_bounds__(256)
vlad_pw_kernel(const float* __restrict__ in, int64_t P, int cin, int cout, const float* __restrict__ wT /*[cin][cout]*/,
               const float* __restrict__ bias, const float* __restrict__ res, int act, float* __restrict__ out) {
    __shared__ float xs[64][65];
    __shared__ __attribute__((aligned(16))) float ws[64][32];
    const int tid = threadIdx.x;
    const int64_t p0 = (int64_t)blockIdx.x * 64;
    const int c0 = blockIdx.y * 32;
    const int tp = tid >> 3, tc = tid & 7;                 // 2 pixels x 4 channels per thread
    float acc[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    for (int k0 = 0; k0 < cin; k0 += 64) {
        const int kc = cin - k0 < 64 ? cin - k0 : 64;
        for (int i = tid; i < 64 * 16; i += 256) {         // 64 pixels x 16 float4
            const int r = i >> 4, q = i & 15;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p0 + r < P && q * 4 < kc) v = *reinterpret_cast<const float4*>(in + (p0 + r) * cin + k0 + q * 4);
            xs[r][q * 4 + 0] = v.x; xs[r][q * 4 + 1] = v.y; xs[r][q * 4 + 2] = v.z; xs[r][q * 4 + 3] = v.w;
        }
        for (int i = tid; i < 64 * 8; i += 256) {          // 64 k x 8 float4
            const int r = i >> 3, q = i & 7;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < kc && c0 + q * 4 < cout) v = *reinterpret_cast<const float4*>(wT + (int64_t)(k0 + r) * cout + c0 + q * 4);
            *reinterpret_cast<float4*>(&ws[r][q * 4]) = v;
        }
        __syncthreads();
        const int kend = (kc + 3) & ~3;                    // cin of the early layers is 8..48: do not run the padded tail
#pragma unroll 4
        for (int k = 0; k < kend; ++k) {
            const float4 wv = *reinterpret_cast<const float4*>(&ws[k][tc * 4]);
            const float x0 = xs[tp * 2][k], x1 = xs[tp * 2 + 1][k];
            acc[0][0] = fmaf(x0, wv.x, acc[0][0]); acc[0][1] = fmaf(x0, wv.y, acc[0][1]);
            acc[0][2] = fmaf(x0, wv.z, acc[0][2]); acc[0][3] = fmaf(x0, wv.w, acc[0][3]);
            acc[1][0] = fmaf(x1, wv.x, acc[1][0]); acc[1][1] = fmaf(x1, wv.y, acc[1][1]);
            acc[1][2] = fmaf(x1, wv.z, acc[1][2]); acc[1][3] = fmaf(x1, wv.w, acc[1][3]);
        }
        __syncthreads();
    }
    const int c = c0 + tc * 4;
    if (c >= cout) return;
    const float4 bs = *reinterpret_cast<const float4*>(bias + c);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int64_t p = p0 + tp * 2 + r;
        if (p >= P) continue;
        float4 v = make_float4(acc[r][0] + bs.x, acc[r][1] + bs.y, acc[r][2] + bs.z, acc[r][3] + bs.w);
        if (res) { const float4 rr = *reinterpret_cast<const float4*>(res + p * cout + c); v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w; }
        if (act == 1) { v.x = relu6f(v.x); v.y = relu6f(v.y); v.z = relu6f(v.z); v.w = relu6f(v.w); }
        *reinterpret_cast<float4*>(out + p * cout + c) = v;
    }
}

// NetVLAD soft-assignment: a[p][k] = softmax_k(f[p] . Aw[k] + ab[k]); one wave per position, lane k < K (K <= 64)
__global__ void __launch_bounds__(256)
vlad_assign_kernel(const float* __restrict__ feat, int64_t n_pos, int Dm, int K, const float* __restrict__ awT /*[Dm][K]*/,
                   const float* __restrict__ ab, float* __restrict__ assign) {
    const int lane = threadIdx.x & 63;
    const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= n_pos) return;
    const float* f = feat + p * Dm;
    float logit = -3.0e38f;
    if (lane < K) {
        float acc = ab[lane];
        for (int d = 0; d < Dm; ++d) acc = fmaf(f[d], awT[d * K + lane], acc);
        logit = acc;
    }
    float mx = logit;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    const float e = (lane < K) ? expf(logit - mx) : 0.f;
    float s = e;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane < K) assign[p * K + lane] = e / s;
}

// V[b][k][d] = sum_p a[p][k] * (c[k][d] - f[p][d]); then intra-normalise over d; one workgroup per (image, cluster)
__global__ void __launch_bounds__(128)
vlad_aggregate_kernel(const float* __restrict__ feat, const float* __restrict__ assign, int n_pos, int Dm, int K,
                      const float* __restrict__ clusters, float* __restrict__ vlad) {
    __shared__ float red[128];
    const int b = blockIdx.y, k = blockIdx.x, d = threadIdx.x;
    const float* f = feat + (int64_t)b * n_pos * Dm;
    const float* a = assign + (int64_t)b * n_pos * K;
    float v = 0.f;
    if (d < Dm) {
        const float c = clusters[k * Dm + d];
        for (int p = 0; p < n_pos; ++p) v = fmaf(a[p * K + k], c - f[(int64_t)p * Dm + d], v);
    }
    red[d] = (d < Dm) ? v * v : 0.f;
    __syncthreads();
    for (int s = 64; s > 0; s >>= 1) { if (d < s) red[d] += red[d + s]; __syncthreads(); }
    const float nrm = sqrtf(red[0]);
    if (d < Dm) vlad[((int64_t)b * K + k) * Dm + d] = v / nrm;
}

// x[b][:] /= ||x[b][:]||_2 ; one workgroup per row
__global__ void __launch_bounds__(256)
l2norm_rows_kernel(float* __restrict__ x, int n) {
    __shared__ float red[256];
    float* r = x + (int64_t)blockIdx.x * n;
    float ss = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) ss = fmaf(r[i], r[i], ss);
    red[threadIdx.x] = ss;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
    const float nrm = sqrtf(red[0]);
    for (int i = threadIdx.x; i < n; i += 256) r[i] = r[i] / nrm;
}

// FC: out[b][j] = v[b] . W[j] + bias[j]; one wave per output row j, the weight row (n_in floats) is streamed once and
// dotted with up to 8 batch vectors held in LDS -> HBM-bound on the 58.7 MB matrix.
#define FC_MAXB 8
__global__ void __launch_bounds__(256)
vlad_fc_kernel(const float* __restrict__ v, int nb, int n_in, const float* __restrict__ W, const float* __restrict__ bias,
               int n_out, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* vs = reinterpret_cast<float*>(smem_raw);        // [nb][n_in]
    for (int i = threadIdx.x * 4; i < nb * n_in; i += 256 * 4) *reinterpret_cast<float4*>(vs + i) = *reinterpret_cast<const float4*>(v + i);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= n_out) return;
    const float* wr = W + (int64_t)j * n_in;
    float acc[FC_MAXB];
#pragma unroll
    for (int b = 0; b < FC_MAXB; ++b) acc[b] = 0.f;
    for (int i = lane * 4; i < n_in; i += 256) {
        const float4 w4 = *reinterpret_cast<const float4*>(wr + i);
#pragma unroll
        for (int b = 0; b < FC_MAXB; ++b) {
            if (b < nb) {
                const float4 x = *reinterpret_cast<const float4*>(vs + b * n_in + i);
                acc[b] = fmaf(w4.x, x.x, acc[b]); acc[b] = fmaf(w4.y, x.y, acc[b]);
                acc[b] = fmaf(w4.z, x.z, acc[b]); acc[b] = fmaf(w4.w, x.w, acc[b]);
            }
        }
    }
#pragma unroll
    for (int b = 0; b < FC_MAXB; ++b) {
        if (b < nb) {
            float s = acc[b];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
            if (lane == 0) out[(int64_t)b * n_out + j] = s + bias[j];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Fused inverted-residual block: expand 1x1 + ReLU6 -> depthwise 3x3 (stride 1|2) + ReLU6 -> project 1x1 (+ residual)
// in ONE kernel per block.  The 6x-expanded tensor never leaves the CU: per output tile of TILE pixels the expanded halo
// region lives in LDS, processed in chunks of 32 hidden channels:
//     xin [R][CIN]   input region (R = (TH*s+2) x (TW*s+2) pixels), loaded once, also the residual source
//     h   [R][32]    expanded chunk (0 outside the image: the depthwise conv pads the EXPANDED tensor with zeros)
//     d   [TILE][32] depthwise output chunk
// and the projection accumulates over chunks in registers.  Lane = hidden channel in the first two steps (weights of the
// lane's channel sit in registers, LDS reads are broadcasts or unit-stride), lane = output channel in the third.
// 57 launches per inference become 23, and the 55 MB expanded tensors of the 300x240 layers are never written.
// ---------------------------------------------------------------------------------------------------------------
struct VladBlockArgs {
    const float* in; float* out;
    const float* blob;         // per 32-channel chunk of the hidden layer: [We cin x 32 | be 32 | Wd 9 x 32 | bd 32 | Wp 32 x cout], zero padded
    const float* bp;           // [cout]
    int Hi, Wi, Ho, Wo, hid, cout, stride, expand, res, batch;
};

// Weights reach the lanes through LDS: the next chunk's blob is fetched into registers while the current chunk computes and
// written to the alternate LDS buffer afterwards, so only the first chunk pays a global-memory round trip (a workgroup per
// CU with nothing else resident cannot hide one per weight matrix per chunk).
// STRIDE is a template parameter so that the region geometry (RW, RH, R) is a compile-time constant: with a run-time stride every
// "pixel -> (row, column)" of the in-image test was a 32-bit integer division by a run-time value (~40 VALU instructions each), executed
// per region pixel per lane per chunk -- 1900 of the ~2300 instructions a wave issued per tile of the 300x240 block (the kernels are
// VALU-issue bound: PMC SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES 0.38-0.46 with 4 waves per SIMD).  The test itself is now done once per
// tile into an LDS mask that every chunk reuses.
template <int CIN, int CP, int TILE, int STRIDE>
__global__ void __launch_bounds__(256)
vlad_block_kernel(VladBlockArgs a) {
    constexpr int TW = TILE >= 128 ? 16 : 8, TH = TILE / TW;
    constexpr int NACC = TILE * CP / 256;
    constexpr int NLD = (32 * (CIN + 11 + CP) / 4 + 255) / 256;      // float4 per thread that cover the largest blob
    static_assert(NACC >= 1, "tile too small for this cout");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int s = STRIDE;
    constexpr int RW = TW * s + 2, RH = TH * s + 2, R = RW * RH;
    // stride 1: region rows padded to the expand loop's step (no bounds tests inside it).  Stride 2 keeps the exact row count and the tests:
    // measured at 32 images, padded vs exact -- b2 110 -> 104 us, b4/b5 67 -> 61 us, but b1 198 -> 239 us, b6 43 -> 62 us
    constexpr bool PAD = STRIDE == 1;
    constexpr int RP = PAD ? (R + 31) / 32 * 32 : R;
    const int blob = 32 * (CIN + 11 + a.cout), blob4 = blob / 4;     // cout % 4 == 0
    float* xin = reinterpret_cast<float*>(smem_raw);       // [RP][CIN], rows >= R zero
    float* h = xin + RP * CIN;                              // [RP][32]
    float* d = h + RP * 32;                                 // [TILE][32]
    float* wbuf = d + TILE * 32;                            // [2][blob]
    float* msk = wbuf + 2 * blob;                           // [RP] 1 inside the image, 0 outside (the depthwise conv pads the EXPANDED tensor)
    const int tid = threadIdx.x;
    const int tiles_x = (a.Wo + TW - 1) / TW, tiles_y = (a.Ho + TH - 1) / TH;
    const int b = blockIdx.x / (tiles_x * tiles_y), tr = blockIdx.x - b * tiles_x * tiles_y;
    const int oy0 = (tr / tiles_x) * TH, ox0 = (tr % tiles_x) * TW;
    const int iy0 = oy0 * s - 1, ix0 = ox0 * s - 1;        // region origin in the input
    const float* inb = a.in + (int64_t)b * a.Hi * a.Wi * CIN;
    const int n_chunks = (a.hid + 31) / 32;

    for (int i = tid; i < blob4; i += 256) reinterpret_cast<float4*>(wbuf)[i] = reinterpret_cast<const float4*>(a.blob)[i];
    for (int i = tid; i < RP * (CIN / 4); i += 256) {
        const int r = i / (CIN / 4), q = i - r * (CIN / 4);
        const int gy = iy0 + r / RW, gx = ix0 + r % RW;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < R && gy >= 0 && gy < a.Hi && gx >= 0 && gx < a.Wi) v = *reinterpret_cast<const float4*>(inb + ((int64_t)gy * a.Wi + gx) * CIN + q * 4);
        *reinterpret_cast<float4*>(xin + r * CIN + q * 4) = v;
    }
    for (int r = tid; r < RP; r += 256) {
        const int gy = iy0 + r / RW, gx = ix0 + r % RW;
        msk[r] = (r < R && gy >= 0 && gy < a.Hi && gx >= 0 && gx < a.Wi) ? 1.f : 0.f;
    }
    float acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = 0.f;
    const int c = tid & 31, rg = tid >> 5;                  // steps 1-2: lane = hidden channel of the chunk
    const int co = tid & (CP - 1), og = tid / CP;           // step 3: lane = output channel
    __syncthreads();

    for (int ci = 0; ci < n_chunks; ++ci) {
        const float* wb = wbuf + (ci & 1) * blob;
        const float* w_we = wb;
        const float* w_be = wb + CIN * 32;
        const float* w_wd = w_be + 32;
        const float* w_bd = w_wd + 9 * 32;
        const float* w_wp = w_bd + 32;
        float4 st[NLD];
        const bool more = ci + 1 < n_chunks;
        if (more) {
            const float4* src = reinterpret_cast<const float4*>(a.blob + (int64_t)(ci + 1) * blob);
#pragma unroll
            for (int j = 0; j < NLD; ++j) { const int i = tid + j * 256; if (i < blob4) st[j] = src[i]; }
        }
        // 1. expand (or copy, for the t = 1 block)
        if (a.expand) {
            float we[CIN];
#pragma unroll
            for (int k = 0; k < CIN; ++k) we[k] = w_we[k * 32 + c];
            const float be = w_be[c];
            // U region pixels at a time: U independent FMA chains (a 256-thread workgroup per CU has no other wave to hide
            // the latency of one dependent chain of CIN FMAs behind)
            constexpr int U = CIN >= 48 ? 2 : 4;
            static_assert(!PAD || RP % (8 * U) == 0, "padded region rows must be a multiple of the expand step");
            for (int r0 = rg; r0 < RP; r0 += 8 * U) {
                float t[U];
#pragma unroll
                for (int u = 0; u < U; ++u) t[u] = be;
#pragma unroll
                for (int k4 = 0; k4 < CIN / 4; ++k4) {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int r = (PAD || (r0 + 8 * u) < RP) ? (r0 + 8 * u) : rg;
                        const float4 x = *reinterpret_cast<const float4*>(xin + r * CIN + k4 * 4);
                        t[u] = fmaf(x.x, we[k4 * 4 + 0], t[u]); t[u] = fmaf(x.y, we[k4 * 4 + 1], t[u]);
                        t[u] = fmaf(x.z, we[k4 * 4 + 2], t[u]); t[u] = fmaf(x.w, we[k4 * 4 + 3], t[u]);
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int r = r0 + 8 * u;
                    if (PAD || r < RP) h[r * 32 + c] = msk[r] != 0.f ? relu6f(t[u]) : 0.f;      // padded rows: mask 0
                }
            }
        } else {
            const int cc = ci * 32 + c;
            for (int r = rg; r < RP; r += 8) h[r * 32 + c] = (cc < CIN) ? xin[r * CIN + cc] : 0.f;
        }
        __syncthreads();
        // 2. depthwise 3x3 + ReLU6
        {
            float wd[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) wd[t] = w_wd[t * 32 + c];
            const float bd = w_bd[c];
#pragma unroll
            for (int i = 0; i < TILE / 8; ++i) {
                const int o = rg + 8 * i;
                const int oy = o / TW, ox = o - oy * TW;
                const float* hp = h + ((oy * s) * RW + ox * s) * 32 + c;
                float t = bd;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) t = fmaf(hp[(dy * RW + dx) * 32], wd[dy * 3 + dx], t);
                d[o * 32 + c] = relu6f(t);
            }
        }
        __syncthreads();
        // 3. projection, accumulated over the chunks (padded hidden channels carry zero weights)
        if (co < a.cout) {
#pragma unroll 8
            for (int k = 0; k < 32; ++k) {
                const float w = w_wp[k * a.cout + co];
#pragma unroll
                for (int i = 0; i < NACC; ++i) acc[i] = fmaf(d[(og + (256 / CP) * i) * 32 + k], w, acc[i]);
            }
        }
        if (more) {
            float4* dst = reinterpret_cast<float4*>(wbuf + ((ci + 1) & 1) * blob);
#pragma unroll
            for (int j = 0; j < NLD; ++j) { const int i = tid + j * 256; if (i < blob4) dst[i] = st[j]; }
        }
        __syncthreads();
    }
    if (co < a.cout) {
        const float bp = a.bp[co];
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            const int o = og + (256 / CP) * i;
            const int oy = o / TW, ox = o - oy * TW;
            if (oy0 + oy < a.Ho && ox0 + ox < a.Wo) {
                float v = acc[i] + bp;
                if (a.res) v += xin[((oy + 1) * RW + ox + 1) * CIN + co];       // stride 1, cin == cout
                a.out[(((int64_t)b * a.Ho + oy0 + oy) * a.Wo + ox0 + ox) * a.cout + co] = v;
            }
        }
    }
}

template <int CIN, int CP, int TILE, int STRIDE>
static int launch_vlad_block_s(hipStream_t st, const VladBlockArgs& a) {
    constexpr int TW = TILE >= 128 ? 16 : 8, TH = TILE / TW;
    constexpr int RW = TW * STRIDE + 2, RH = TH * STRIDE + 2, RP = STRIDE == 1 ? (RW * RH + 31) / 32 * 32 : RW * RH;
    const size_t smem = ((size_t)RP * (CIN + 32 + 1) + TILE * 32 + 2 * 32 * (CIN + 11 + a.cout)) * 4;
    auto kfn = vlad_block_kernel<CIN, CP, TILE, STRIDE>;
    OMNI_HIP_TRY(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    const int tiles = cdiv(a.Wo, TW) * cdiv(a.Ho, TH);
    hipLaunchKernelGGL(kfn, dim3(tiles * a.batch), dim3(256), smem, st, a);
    OMNI_LAUNCH_CHECK();
    return OMNI_OK;
}
template <int CIN, int CP, int TILE>
static int launch_vlad_block(hipStream_t st, const VladBlockArgs& a) {
    return a.stride == 2 ? launch_vlad_block_s<CIN, CP, TILE, 2>(st, a) : launch_vlad_block_s<CIN, CP, TILE, 1>(st, a);
}

// ---------------------------------------------------------------------------------------------------------------
// Pointwise (1x1) convolution on the matrix cores, exact f32: out[p][co] = act(sum_k x[p][k] W[k][co] + b[co]) (+ res).
// v_mfma_f32_32x32x2_f32 (an fmaf chain, bit-exact f32): one 32-pixel x 32-channel tile per workgroup, M = pixels (A = the
// activations), N = output channels (B = weights [cin][cout]); the K range is split over the workgroup's SPLITK waves, inside a
// wave the two half-waves take alternate k; partial tiles are summed through LDS in a fixed order.  No LDS staging of operands:
// the low-resolution MobileNetV2 blocks (19x15 ... 38x30 pixels per image) are a few hundred such tiles, the 6x-expanded
// tensors are a few MB and stay in L2 between the three launches of a block.
// ---------------------------------------------------------------------------------------------------------------
typedef float floatx16v __attribute__((ext_vector_type(16)));
template <int SPLITK>
__global__ void __launch_bounds__(64 * SPLITK)
vlad_pw_mfma_kernel(const float* __restrict__ in, int64_t P, int cin, int cout, const float* __restrict__ wT /*[cin][cout]*/,
                    const float* __restrict__ bias, const float* __restrict__ res, int act, float* __restrict__ out) {
    __shared__ float part[SPLITK > 1 ? SPLITK - 1 : 1][16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, kk = lane >> 5;
    const int64_t p0 = (int64_t)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32;
    const int64_t pr = (p0 + i < P) ? (p0 + i) : (P - 1);          // clamped row for the A loads
    const int cc = c0 + i;                                           // this lane's output channel (B column)
    const bool cval = cc < cout;
    // K split: wave w takes [ks, ke) (a multiple of 8 wide), inside it half-wave kk the contiguous half [ks + kk nk, ks + (kk+1) nk)
    // -- the A operand is then 16-byte loads, and the loads of the next four k are issued before the four MFMAs of the current
    // ones (a single accumulator chain: MFMAs are dependent, so the loads are all that can overlap)
    const int kq = ((cin + SPLITK * 8 - 1) / (SPLITK * 8)) * 8;
    const int ks = wave * kq < cin ? wave * kq : cin, ke = (ks + kq < cin) ? ks + kq : cin;
    const int nk = (ke - ks) >> 1;                                   // k per half-wave: a multiple of 4 (cin % 8 == 0)
    const float* xa = in + pr * cin + ks + kk * nk;
    const float* wb = wT + (cval ? cc : 0) + (int64_t)(ks + kk * nk) * cout;
    floatx16v acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f), b4 = a4;
    auto load4 = [&](int t, float4& a, float4& b) {
        a = *reinterpret_cast<const float4*>(xa + t);
        const float* w = wb + (int64_t)t * cout;
        b = cval ? make_float4(w[0], w[cout], w[2 * (int64_t)cout], w[3 * (int64_t)cout]) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    if (nk > 0) load4(0, a4, b4);
    for (int t = 0; t < nk; t += 4) {
        float4 an = a4, bn = b4;
        if (t + 4 < nk) load4(t + 4, an, bn);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc, 0, 0, 0);
        a4 = an; b4 = bn;
    }
    if constexpr (SPLITK > 1) {
        if (wave > 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) part[wave - 1][r][lane] = acc[r];
        }
        __syncthreads();
        if (wave > 0) return;
#pragma unroll
        for (int w = 0; w < SPLITK - 1; ++w)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += part[w][r][lane];
    }
    if (!cval) return;
    const float bs = bias[cc];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int64_t p = p0 + (r & 3) + 8 * (r >> 2) + 4 * kk;     // C/D layout: row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5), col = lane & 31
        if (p < P) {
            float v = acc[r] + bs;
            if (res) v += res[p * cout + cc];
            if (act == 1) v = relu6f(v);
            out[p * cout + cc] = v;
        }
    }
}

static int vlad_pw_mfma(hipStream_t st, const float* in, int64_t P, int cin, int cout, const float* wT, const float* bias, const float* res,
                        int act, float* out) {
    dim3 grid((unsigned)cdiv64(P, 32), cdiv(cout, 32));
    if (cin >= 128)
        hipLaunchKernelGGL(vlad_pw_mfma_kernel<4>, grid, dim3(256), 0, st, in, P, cin, cout, wT, bias, res, act, out);
    else
        hipLaunchKernelGGL(vlad_pw_mfma_kernel<1>, grid, dim3(64), 0, st, in, P, cin, cout, wT, bias, res, act, out);
    OMNI_LAUNCH_CHECK();
    return OMNI_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Fused inverted-residual block on the matrix cores (exact f32, v_mfma_f32_32x32x2_f32): expand 1x1 + ReLU6 -> depthwise 3x3 (stride 1|2)
// + ReLU6 -> project 1x1 (+ residual) in ONE launch for ANY block shape of the layer table.  A workgroup = one 8x8 tile of output pixels of
// one image; per 32-channel chunk of the hidden layer
//     expand   h[R][32]  = relu6(xin[R][cin] . We[cin][32] + be)  on the halo region R = ((8-1) s + 3)^2 pixels, M = region pixels (32 per
//                          MFMA tile, tiles spread over the 4 waves), zero outside the image (the depthwise conv pads the EXPANDED tensor)
//     depthwise d[64][32] on the VALU (lane = channel)
//     project  acc[64][cout] += d[64][32] . Wp[32][cout]          (tile pairs (m, n) spread over the waves, accumulators in registers)
// Operands come from LDS with odd row strides (conflict-free MFMA fragment reads); the chunk's weights are staged per chunk.  The three
// launches per block of the low-resolution path (pointwise MFMA / depthwise / pointwise MFMA, 10-18 us each at the launch-latency floor,
// expanded tensor through L2/HBM twice) become one, and the fp32-VALU fused kernel of the high-resolution blocks runs at matrix rate.
// ---------------------------------------------------------------------------------------------------------------
struct VladMBlockArgs {
    const float* in; float* out;
    const float* blob;         // per 32-channel chunk: [We cin x 32 | be 32 | Wd 9 x 32 | bd 32 | Wp 32 x cop], zero padded
    const float* bp;           // [cout]
    int Hi, Wi, Ho, Wo, cin, hid, cout, cop, stride, res, batch;
    // hidden-layer split: workgroup (tile, g) handles chunks [g * cpw, (g + 1) * cpw) and writes its partial projection tile; a second, tiny
    // launch (vlad_mblock_reduce_kernel) adds the partials in group order (deterministic), bias and residual.  A block is a chain of short
    // dependent phases behind barriers: one workgroup walking all 5-11 chunks lives 30-90 us with the CU mostly idle (PMC: waves wait 50 % of
    // their life, profiles/r02_*); splitting the chain over workgroups turns it into parallel work.  (Reducing inside the kernel -- last
    // arrival of a tile behind __threadfence + a device-scope counter -- measured 8x SLOWER: an agent-scope release on this part writes the
    // XCD's L2 back, once per workgroup.)
    int cpw, n_groups;         // chunks per workgroup, groups per tile (1 = no split: the output is written directly)
    float* partial;            // [tile][group][64][cop]
};
__global__ void __launch_bounds__(256)
vlad_mblock_kernel(VladMBlockArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int s = a.stride, cin = a.cin, cinp = cin + 1, cop = a.cop;
    const int RW = 7 * s + 3, R = RW * RW, R32 = (R + 31) >> 5, RP = R32 * 32;
    const int NT = cop >> 5;
    float* xin = reinterpret_cast<float*>(smem_raw);       // [RP][cinp]
    float* h = xin + RP * cinp;                             // [RP][33]
    float* d = h + RP * 33;                                 // [64][33]
    float* we = d + 64 * 33;                                // [cin][32]
    float* be = we + cin * 32;                              // [32]
    float* wd = be + 32;                                    // [9][32]
    float* bd = wd + 9 * 32;                                // [32]
    float* wp = bd + 32;                                    // [32][cop]
    float* msk = wp + 32 * cop;                             // [RP] 1 inside the image, 0 outside
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, kk = lane >> 5;
    const int tiles_x = (a.Wo + 7) >> 3, tiles_y = (a.Ho + 7) >> 3;
    const int b = blockIdx.x / (tiles_x * tiles_y), tr = blockIdx.x - b * tiles_x * tiles_y;
    const int oy0 = (tr / tiles_x) * 8, ox0 = (tr % tiles_x) * 8;
    const int iy0 = oy0 * s - 1, ix0 = ox0 * s - 1;        // region origin in the input
    const float* inb = a.in + (int64_t)b * a.Hi * a.Wi * cin;
    const int blob = 32 * (cin + 11 + cop), n_chunks_all = (a.hid + 31) >> 5;
    const int chunk0 = blockIdx.y * a.cpw, n_chunks = chunk0 + a.cpw < n_chunks_all ? chunk0 + a.cpw : n_chunks_all;

    {   // input region (also the residual source) + in-image mask
        const int q4 = cin >> 2;
        for (int e = tid; e < RP * q4; e += 256) {
            const int r = e / q4, q = e - r * q4;
            const int gy = iy0 + r / RW, gx = ix0 + r % RW;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < R && gy >= 0 && gy < a.Hi && gx >= 0 && gx < a.Wi) v = *reinterpret_cast<const float4*>(inb + ((int64_t)gy * a.Wi + gx) * cin + q * 4);
            float* x = xin + r * cinp + q * 4;
            x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
        }
        for (int r = tid; r < RP; r += 256) {
            const int gy = iy0 + r / RW, gx = ix0 + r % RW;
            msk[r] = (r < R && gy >= 0 && gy < a.Hi && gx >= 0 && gx < a.Wi) ? 1.f : 0.f;
        }
    }
    floatx16v acc[2];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;

    for (int ci = chunk0; ci < n_chunks; ++ci) {
        __syncthreads();                                    // previous chunk's readers are done with we/wp/h/d (first pass: xin is complete)
        {   // this chunk's weights.  (Fetching the NEXT chunk's blob into registers meanwhile and parking it after the projection measured
            // slower -- 49 vs 45 us per block: the extra registers and barriers cost more than the exposed L2 round trip.)
            const float4* src = reinterpret_cast<const float4*>(a.blob + (int64_t)ci * blob);
            float4* dst = reinterpret_cast<float4*>(we);
            for (int e = tid; e < blob / 4; e += 256) dst[e] = src[e];
        }
        __syncthreads();
        // 1. expand on the matrix cores: region tile m (32 pixels) x 32 hidden channels, K = cin
        for (int m = wave; m < R32; m += 4) {
            floatx16v e;
#pragma unroll
            for (int r = 0; r < 16; ++r) e[r] = 0.f;
            const float* xa = xin + (m * 32 + i) * cinp + kk;
            const float* wb = we + kk * 32 + i;
            for (int t = 0; t < cin; t += 2) e = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[t], wb[t * 32], e, 0, 0, 0);
            const float bias = be[i];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int px = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;       // C layout: row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5), column = lane & 31
                h[px * 33 + i] = msk[px] != 0.f ? relu6f(e[r] + bias) : 0.f;
            }
        }
        __syncthreads();
        // 2. depthwise 3x3 + ReLU6: thread = (channel, 8 output pixels)
        {
            const int c = tid & 31, g = tid >> 5;
            float w9[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) w9[t] = wd[t * 32 + c];
            const float bias = bd[c];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int o = g + 8 * q, oy = o >> 3, ox = o & 7;
                const float* hp = h + ((oy * s) * RW + ox * s) * 33 + c;
                float t = bias;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) t = fmaf(hp[(dy * RW + dx) * 33], w9[dy * 3 + dx], t);
                d[o * 33 + c] = relu6f(t);
            }
        }
        __syncthreads();
        // 3. projection, accumulated over the chunks: tile pairs (m, n) = (output-pixel half, 32 output channels)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int pr = wave + 4 * p;
            if (pr < 2 * NT) {
                const int m = pr / NT, n = pr - m * NT;
                const float* da = d + (m * 32 + i) * 33 + kk;
                const float* wb = wp + kk * cop + n * 32 + i;
#pragma unroll
                for (int t = 0; t < 32; t += 2) acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(da[t], wb[t * cop], acc[p], 0, 0, 0);
            }
        }
    }
    if (a.n_groups > 1) {
        // partial projection tile of this group -> global scratch (L2), then the last arrival of the tile reduces in group order
        float* part = a.partial + ((int64_t)blockIdx.x * a.n_groups + blockIdx.y) * 64 * cop;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int pr = wave + 4 * p;
            if (pr >= 2 * NT) continue;
            const int m = pr / NT, n = pr - m * NT;
#pragma unroll
            for (int r = 0; r < 16; ++r) part[(m * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk) * cop + n * 32 + i] = acc[p][r];
        }
        return;                                             // vlad_mblock_reduce_kernel adds the groups in order (+ bias, residual)
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int pr = wave + 4 * p;
        if (pr >= 2 * NT) continue;
        const int m = pr / NT, n = pr - m * NT;
        const int ch = n * 32 + i;
        if (ch >= a.cout) continue;
        const float bias = a.bp[ch];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk, oy = o >> 3, ox = o & 7;
            if (oy0 + oy < a.Ho && ox0 + ox < a.Wo) {
                float v = acc[p][r] + bias;
                if (a.res) v += xin[((oy + 1) * RW + ox + 1) * cinp + ch];      // stride 1, cin == cout
                a.out[(((int64_t)b * a.Ho + oy0 + oy) * a.Wo + ox0 + ox) * a.cout + ch] = v;
            }
        }
    }
}

// out[tile pixel][ch] = sum_g partial[tile][g][o][ch] (g ascending) + bias (+ the block input at the same pixel); thread = (pixel, 4 channels)
__global__ void __launch_bounds__(256)
vlad_mblock_reduce_kernel(VladMBlockArgs a) {
    const int cop = a.cop, q4 = cop >> 2;
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int tiles_x = (a.Wo + 7) >> 3, tiles_y = (a.Ho + 7) >> 3;
    const int64_t total = (int64_t)tiles_x * tiles_y * a.batch * 64 * q4;
    if (e >= total) return;
    const int c4 = (int)(e % q4) * 4;
    const int o = (int)((e / q4) & 63);
    const int64_t tile = e / ((int64_t)q4 * 64);
    const int b = (int)(tile / (tiles_x * tiles_y)), tr = (int)(tile - (int64_t)b * tiles_x * tiles_y);
    const int oy = (tr / tiles_x) * 8 + (o >> 3), ox = (tr % tiles_x) * 8 + (o & 7);
    if (c4 >= a.cout || oy >= a.Ho || ox >= a.Wo) return;
    const float* p = a.partial + (tile * a.n_groups * 64 + o) * cop + c4;
    float4 v = *reinterpret_cast<const float4*>(p);
    for (int g = 1; g < a.n_groups; ++g) {
        const float4 t = *reinterpret_cast<const float4*>(p + (int64_t)g * 64 * cop);
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    const float4 bs = *reinterpret_cast<const float4*>(a.bp + c4);          // cout % 4 == 0
    v.x += bs.x; v.y += bs.y; v.z += bs.z; v.w += bs.w;
    if (a.res) {                                                             // stride 1, cin == cout
        const float4 r = *reinterpret_cast<const float4*>(a.in + (((int64_t)b * a.Hi + oy) * a.Wi + ox) * a.cin + c4);
        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
    }
    *reinterpret_cast<float4*>(a.out + (((int64_t)b * a.Ho + oy) * a.Wo + ox) * a.cout + c4) = v;
}

static size_t vlad_mblock_smem(int cin, int cop, int stride) {
    const int RW = 7 * stride + 3, RP = ((RW * RW + 31) / 32) * 32;
    return ((size_t)RP * (cin + 1) + (size_t)RP * 33 + 64 * 33 + 32 * (cin + 11 + cop) + RP) * 4;
}
static int launch_vlad_mblock(hipStream_t st, const VladMBlockArgs& a) {
    const size_t smem = vlad_mblock_smem(a.cin, a.cop, a.stride);
    static DynSmemState attr;
    OMNI_HIP_TRY(ensure_dyn_smem(attr, (const void*)vlad_mblock_kernel, smem));
    const int tiles = cdiv(a.Wo, 8) * cdiv(a.Ho, 8);
    hipLaunchKernelGGL(vlad_mblock_kernel, dim3(tiles * a.batch, a.n_groups), dim3(256), smem, st, a);
    OMNI_LAUNCH_CHECK();
    if (a.n_groups > 1) {
        const int64_t total = (int64_t)tiles * a.batch * 64 * (a.cop / 4);
        hipLaunchKernelGGL(vlad_mblock_reduce_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, st, a);
        OMNI_LAUNCH_CHECK();
    }
    return OMNI_OK;
}

// returns OMNI_ERR_INVALID (without setting an error) when no instantiation covers the block's shape
static bool vlad_block_supported(int cin, int cout) {
    const int cp = cout <= 8 ? 8 : cout <= 16 ? 16 : cout <= 32 ? 32 : cout <= 64 ? 64 : cout <= 128 ? 128 : 0;
    if (!cp) return false;
    return (cin == 8 && (cp == 8 || cp == 16)) || (cin == 16 && (cp == 8 || cp == 16 || cp == 32)) || (cin == 24 && cp == 32) ||
           (cin == 32 && (cp == 32 || cp == 64)) || (cin == 56 && (cp == 64 || cp == 128));
}
static int vlad_block(hipStream_t st, int cin, const VladBlockArgs& a) {
    const int cp = a.cout <= 8 ? 8 : a.cout <= 16 ? 16 : a.cout <= 32 ? 32 : a.cout <= 64 ? 64 : 128;
    // tile size by output resolution: every workgroup pays two global round trips (input region, first weight blob) before it
    // computes, so the high-resolution blocks use 128- / 64-pixel tiles (still > 1000 workgroups per 4 images), the rest 32 / 16
    const int64_t px = (int64_t)a.batch * a.Ho * a.Wo;
    // (measured: 128- / 64-pixel tiles are SLOWER for the 300x240 / 150x120 blocks -- 54 -> 70 us, 38 -> 46 us: their LDS footprint
    // halves the resident workgroups and with them the latency hiding; kept instantiated for other image sizes only)
    // OMNI_VLAD_BIG=64|128 re-enables them for A/B at other batch sizes (numerics do not depend on the tile size)
    const int big = config_process()[CFG_VLAD_BIG]; (void)px;
#define VB(CI, CPV, T) if (cin == CI && cp == CPV) return launch_vlad_block<CI, CPV, T>(st, a)
    if (big == 128) { VB(16, 8, 128); VB(8, 8, 128); VB(16, 16, 128); }
    if (big >= 64) { VB(16, 8, 64); VB(8, 8, 64); VB(8, 16, 64); VB(16, 16, 64); }
    VB(8, 8, 32); VB(8, 16, 32); VB(16, 8, 32); VB(16, 16, 32); VB(16, 32, 32); VB(24, 32, 16); VB(32, 32, 16); VB(32, 64, 16);
    VB(56, 64, 16); VB(56, 128, 16);
#undef VB
    set_error("vlad_block: no kernel for cin=%d cout=%d", cin, a.cout);
    return OMNI_ERR_INVALID;
}

// stem, coalesced: thread = (pixel, 4 output channels), float4 NHWC stores
__global__ void __launch_bounds__(256)
vlad_stem4_kernel(const uint8_t* __restrict__ gray, int stride, int H, int W, int mask0, int mask1, int Ho, int Wo, int cout,
                  int cstride, const float* __restrict__ w /*[cout][9]*/, const float* __restrict__ bias, float* __restrict__ out) {
    const int b = blockIdx.y;
    const int cq = cout >> 2;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Ho * Wo * cq) return;
    const int p = i / cq, c4 = (i - p * cq) * 4;
    const int oy = p / Wo, ox = p - oy * Wo;
    const uint8_t* g = gray + (int64_t)b * stride * H;
    float v[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int y = oy * cstride - 1 + t / 3, x = ox * cstride - 1 + t % 3;
        float px = 0.f;                                    // zero padding is applied AFTER normalisation
        if (y >= 0 && y < H && x >= 0 && x < W) {
            const float raw = (y >= mask0 && y < mask1) ? 0.f : (float)g[(int64_t)y * stride + x];
            px = (raw - 128.0f) / 128.0f;
        }
        v[t] = px;
    }
    float r[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float acc = bias[c4 + j];
#pragma unroll
        for (int t = 0; t < 9; ++t) acc = fmaf(v[t], w[(c4 + j) * 9 + t], acc);
        r[j] = relu6f(acc);
    }
    *reinterpret_cast<float4*>(out + ((int64_t)b * Ho * Wo + p) * cout + c4) = make_float4(r[0], r[1], r[2], r[3]);
}

// NetVLAD aggregation, parallel over positions: workgroup = (cluster k, image), thread = (d, part) with 8 position parts;
// partial sums are combined in a fixed order (deterministic), then the intra-normalisation over d.
#define AGG_PARTS 8
__global__ void __launch_bounds__(1024)
vlad_aggregate8_kernel(const float* __restrict__ feat, const float* __restrict__ assign, int n_pos, int Dm, int K,
                       const float* __restrict__ clusters, float* __restrict__ vlad) {
    __shared__ float part[AGG_PARTS][128];
    __shared__ float red[128];
    const int b = blockIdx.y, k = blockIdx.x;
    const int dd = threadIdx.x & 127, pt = threadIdx.x >> 7;
    const float* f = feat + (int64_t)b * n_pos * Dm;
    const float* a = assign + (int64_t)b * n_pos * K;
    float v = 0.f;
    if (dd < Dm) {
        const float c = clusters[k * Dm + dd];
        // same fmaf chain, loads of nine positions in flight (36 dependent L2 round trips per thread were most of this kernel's 26 us)
#pragma unroll 9
        for (int p = pt; p < n_pos; p += AGG_PARTS) v = fmaf(a[p * K + k], c - f[(int64_t)p * Dm + dd], v);
    }
    part[pt][dd] = v;
    __syncthreads();
    if (pt == 0) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < AGG_PARTS; ++q) t += part[q][dd];
        part[0][dd] = t;
        red[dd] = (dd < Dm) ? t * t : 0.f;
    }
    __syncthreads();
    for (int s2 = 64; s2 > 0; s2 >>= 1) { if (threadIdx.x < s2) red[threadIdx.x] += red[threadIdx.x + s2]; __syncthreads(); }
    const float nrm = sqrtf(red[0]);
    if (pt == 0 && dd < Dm) vlad[((int64_t)b * K + k) * Dm + dd] = part[0][dd] / nrm;
}

// NetVLAD soft-assignment with the assignment matrix staged in LDS: workgroup = 8 positions, wave = 2 positions x 32 clusters
__global__ void __launch_bounds__(256)
vlad_assign2_kernel(const float* __restrict__ feat, int64_t n_pos, int Dm, int K, const float* __restrict__ awT /*[Dm][K]*/,
                    const float* __restrict__ ab, float* __restrict__ assign) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* aw = reinterpret_cast<float*>(smem_raw);            // [Dm][K]
    float* fs = aw + Dm * K;                                    // [8][Dm]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t p0 = (int64_t)blockIdx.x * 8;
    for (int i = tid; i < Dm * K; i += 256) aw[i] = awT[i];
    for (int i = tid; i < 8 * Dm; i += 256) { const int64_t p = p0 + i / Dm; fs[i] = (p < n_pos) ? feat[p * Dm + i % Dm] : 0.f; }
    __syncthreads();
    const int k = lane & 31, sub = lane >> 5;                  // K <= 32 on this path
    const int pl = wave * 2 + sub;
    const int64_t p = p0 + pl;
    float logit = -3.0e38f;
    if (k < K) {
        float acc = ab[k];
        for (int dd = 0; dd < Dm; ++dd) acc = fmaf(fs[pl * Dm + dd], aw[dd * K + k], acc);
        logit = acc;
    }
    float mx = logit;
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    const float e = (k < K) ? expf(logit - mx) : 0.f;
    float sum = e;
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
    if (k < K && p < n_pos) assign[p * K + k] = e / sum;
}

// FC, 4 output rows per wave: every LDS read of the batch vectors feeds 4 weight rows (the single-row version is LDS-bound)
__global__ void __launch_bounds__(256)
vlad_fc4_kernel(const float* __restrict__ v, int nb, int n_in, const float* __restrict__ W, const float* __restrict__ bias,
                int n_out, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* vs = reinterpret_cast<float*>(smem_raw);        // [nb][n_in]
    for (int i = threadIdx.x * 4; i < nb * n_in; i += 256 * 4) *reinterpret_cast<float4*>(vs + i) = *reinterpret_cast<const float4*>(v + i);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int j0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
    if (j0 >= n_out) return;
    float acc[4][FC_MAXB];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int b = 0; b < FC_MAXB; ++b) acc[r][b] = 0.f;
    for (int i = lane * 4; i < n_in; i += 256) {
        float4 w4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) w4[r] = (j0 + r < n_out) ? *reinterpret_cast<const float4*>(W + (int64_t)(j0 + r) * n_in + i) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int b = 0; b < FC_MAXB; ++b) {
            if (b < nb) {
                const float4 x = *reinterpret_cast<const float4*>(vs + b * n_in + i);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc[r][b] = fmaf(w4[r].x, x.x, acc[r][b]); acc[r][b] = fmaf(w4[r].y, x.y, acc[r][b]);
                    acc[r][b] = fmaf(w4[r].z, x.z, acc[r][b]); acc[r][b] = fmaf(w4[r].w, x.w, acc[r][b]);
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int b = 0; b < FC_MAXB; ++b) {
            if (b < nb && j0 + r < n_out) {
                float sacc = acc[r][b];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) sacc += __shfl_xor(sacc, off, 64);
                if (lane == 0) out[(int64_t)b * n_out + j0 + r] = sacc + bias[j0 + r];
            }
        }
}

// ---------------------------------------------------------------------------------------------------------------
// FC 3584 -> 4096 on the matrix cores, exact f32 (v_mfma_f32_32x32x2_f32 = an fmaf chain): out[b][j] = v[b] . W[j] + bias[j] for up to 32
// images per pass, the 58.7 MB matrix streamed ONCE per pass.  M = images (A = the NetVLAD vectors, L2-resident), N = 32 output rows j
// (B = W), K split over FCM_KY workgroups x 4 waves; inside a wave the two half-waves take the two halves of its K range, so both operands
// are 16-byte loads.  W is re-packed at create time as [j tile][k group of 4][32 rows][4]: a half-wave's B load is one contiguous 512 B.
// The waves of a workgroup are summed through LDS in a fixed order, the FCM_KY partial tiles by vlad_fc_finish_kernel (+ bias + the final
// L2 normalisation): deterministic.  Replaces 4 passes of vlad_fc4_kernel (33 us each at 8 images per pass) + l2norm_rows_kernel.
// ---------------------------------------------------------------------------------------------------------------
#define FCM_KY 8
#define FCM_DEPTH 14
__global__ void __launch_bounds__(256)
vlad_fc_mfma_kernel(const float* __restrict__ v, int nb, int n_in, const float* __restrict__ Wp, int n_out, float* __restrict__ part /*[FCM_KY][32][n_out]*/) {
    __shared__ float red[3][16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, kk = lane >> 5;
    const int jt = blockIdx.x, ky = blockIdx.y;
    const int groups = n_in >> 2;                                  // k groups of 4
    const int per_wave = groups / (FCM_KY * 4), half = per_wave >> 1;       // create() checks divisibility
    const int g0 = (ky * 4 + wave) * per_wave + kk * half;
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    const f32x4v* wp = reinterpret_cast<const f32x4v*>(Wp) + ((int64_t)jt * groups + g0) * 32 + i;
    const float4* vp = reinterpret_cast<const float4*>(v + (int64_t)(i < nb ? i : nb - 1) * n_in) + g0;     // rows past nb repeat the last image (not written)
    floatx16v acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // the wave's whole K range goes in flight at once (<= FCM_DEPTH 16-byte loads of W and of v per lane, groups of FCM_DEPTH): with two
    // loads ahead the 14 dependent 1 KiB loads of a wave were the bound (1.8 TB/s), not the matrix pipe
    for (int t0 = 0; t0 < half; t0 += FCM_DEPTH) {
        float4 av[FCM_DEPTH];
        f32x4v bv[FCM_DEPTH];
#pragma unroll
        for (int u = 0; u < FCM_DEPTH; ++u) {
            const int t = t0 + u < half ? t0 + u : half - 1;            // tail: harmless re-read, not used
            av[u] = vp[t];
            bv[u] = __builtin_nontemporal_load(wp + (int64_t)t * 32);
        }
#pragma unroll
        for (int u = 0; u < FCM_DEPTH; ++u) {
            if (t0 + u < half) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].x, bv[u][0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].y, bv[u][1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].z, bv[u][2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].w, bv[u][3], acc, 0, 0, 0);
            }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave - 1][r][lane] = acc[r];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int w = 0; w < 3; ++w)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += red[w][r][lane];
    const int j = jt * 32 + i;                                      // C layout: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int b = (r & 3) + 8 * (r >> 2) + 4 * kk;
        if (b < nb) part[((int64_t)ky * 32 + b) * n_out + j] = acc[r];
    }
}

// out[b][:] = normalise(sum_ky part[ky][b][:] + bias); one workgroup per image
__global__ void __launch_bounds__(256)
vlad_fc_finish_kernel(const float* __restrict__ part, const float* __restrict__ bias, int n_out, float* __restrict__ out) {
    __shared__ float red[256];
    const int b = blockIdx.x;
    float ss = 0.f;
    for (int j = threadIdx.x; j < n_out; j += 256) {
        float t = part[(int64_t)b * n_out + j];
#pragma unroll
        for (int ky = 1; ky < FCM_KY; ++ky) t += part[((int64_t)ky * 32 + b) * n_out + j];
        t += bias[j];
        out[(int64_t)b * n_out + j] = t;
        ss = fmaf(t, t, ss);
    }
    red[threadIdx.x] = ss;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
    const float nrm = sqrtf(red[0]);
    for (int j = threadIdx.x; j < n_out; j += 256) out[(int64_t)b * n_out + j] = out[(int64_t)b * n_out + j] / nrm;
}

static int upload(float** dst, const float* src, size_t n, hipStream_t st) {
    OMNI_HIP_TRY(hipMalloc((void**)dst, n * 4));
    OMNI_HIP_TRY(hipMemcpyAsync(*dst, src, n * 4, hipMemcpyHostToDevice, st));
    OMNI_HIP_TRY(hipStreamSynchronize(st));
    return OMNI_OK;
}

static int vlad_backbone_unfused(omni_vlad* v, const uint8_t* gray_dev, int stride, int batch, int fisheye_mask, int* cur_out) {
    hipStream_t st = v->ctx->stream;
    const int H = v->H, W = v->W;
    int m0, m1;
    omni_fisheye_mask_rows(H, fisheye_mask, &m0, &m1);
    int cur = -1;             // index of the buffer holding the current activation
    int block_in = -1;        // buffer holding the inverted-residual block input (for pw_linear_res)
    auto pick = [&](int a, int b2) { for (int i = 0; i < 3; ++i) if (i != a && i != b2) return i; return 0; };
    for (size_t li = 0; li < v->layers.size(); ++li) {
        const VladLayerDev& L = v->layers[li];
        const int64_t Pout = (int64_t)batch * L.hout * L.wout;
        if (L.kind == OMNI_VLAD_CONV3X3_RELU6) {
            const int dst = 0;
            hipLaunchKernelGGL(vlad_stem_kernel, dim3(cdiv(L.hout * L.wout, 256), batch), dim3(256), 0, st, gray_dev, stride, H, W, m0, m1,
                               L.hout, L.wout, L.cout, L.stride, L.w, L.b, v->buf[dst]);
            cur = dst; block_in = dst;
        } else if (L.kind == OMNI_VLAD_DW3X3_RELU6) {
            const int dst = pick(cur, block_in);
            const int64_t total = Pout * (L.cout / 4);
            hipLaunchKernelGGL(vlad_dw_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, st, v->buf[cur], L.hin, L.win, L.cin, L.hout,
                               L.wout, L.stride, L.w, L.b, v->buf[dst], total);
            cur = dst;
        } else {
            const bool expand = (L.kind == OMNI_VLAD_PW_RELU6);
            if (expand) block_in = cur;          // input of the inverted-residual block
            const int dst = pick(cur, block_in);
            const float* res = (L.kind == OMNI_VLAD_PW_LINEAR_RES) ? v->buf[block_in] : nullptr;
            hipLaunchKernelGGL(vlad_pw_kernel, dim3((unsigned)cdiv64(Pout, 64), cdiv(L.cout, 32)), dim3(256), 0, st, v->buf[cur], Pout, L.cin,
                               L.cout, L.w, L.b, res, expand ? 1 : 0, v->buf[dst]);
            cur = dst;
            if (!expand) block_in = cur;         // a projection ends the block; the next block starts from here
        }
        OMNI_LAUNCH_CHECK();
    }
    *cur_out = cur;
    return OMNI_OK;
}

// Stem conv3x3 stride 2 (u8 -> 16 channels, ReLU6) + block 0 (t = 1: depthwise 3x3 + ReLU6, projection 16 -> 8) in ONE pass.  Run
// separately the 16-channel stem map makes an HBM round trip at the network's highest resolution (4.6 MB per image written, read back
// by block 0, whose 32-pixel tiles are mostly fixed cost): 163 + 223 us per 32 images, the two most expensive MobileNetVLAD launches.
// A workgroup = 16x8 output pixels: the 37x21 input patch is normalised into LDS once ((x - 128) / 128, zero padding applied AFTER
// normalisation, fisheye rows read as 0), the stem is evaluated on the 18x10 halo region (zero outside the stem map = the depthwise
// conv's padding), then depthwise and projection -- the same FMA order as vlad_stem4_kernel / vlad_block_kernel: bit-identical.
#define SB_TW 16
#define SB_TH 8
__global__ void __launch_bounds__(256)
vlad_stem_b0_kernel(const uint8_t* __restrict__ gray, int stride, int H, int W, int mask0, int mask1, int Ho, int Wo,
                    const float* __restrict__ ws /*[16][9]*/, const float* __restrict__ bs, const float* __restrict__ wd_t /*[9][16]*/,
                    const float* __restrict__ bd, const float* __restrict__ wp_t /*[16][8]*/, const float* __restrict__ bp,
                    float* __restrict__ out /*[B][Ho][Wo][8]*/, int sk_y0, int sk_y1, int sk_x0, int sk_x1) {
    {   // tiles inside the constant region of the fisheye mask hold their values already (omni_vlad::MaskSkip)
        const int tx_n = (Wo + SB_TW - 1) / SB_TW, tyy = (int)blockIdx.x / tx_n, txx = (int)blockIdx.x - tyy * tx_n;
        if (tyy >= sk_y0 && tyy < sk_y1 && txx >= sk_x0 && txx < sk_x1) return;
    }
    constexpr int RW = SB_TW + 2, RH = SB_TH + 2, R = RW * RH;           // 18 x 10 stem pixels
    constexpr int PW = 2 * RW + 1, PH = 2 * RH + 1;                       // 37 x 21 input pixels
    __shared__ float patch[PH * PW];
    __shared__ __attribute__((aligned(16))) float h[R * 16];
    __shared__ __attribute__((aligned(16))) float d[SB_TW * SB_TH * 16];
    __shared__ float s_ws[16 * 9], s_bs[16], s_wd[9 * 16], s_bd[16];
    const int tid = threadIdx.x, b = blockIdx.y;
    const int tiles_x = (Wo + SB_TW - 1) / SB_TW;
    const int oy0 = (blockIdx.x / tiles_x) * SB_TH, ox0 = (blockIdx.x % tiles_x) * SB_TW;
    const int sy0 = oy0 - 1, sx0 = ox0 - 1;                              // halo region origin in the stem map
    const int iy0 = 2 * sy0 - 1, ix0 = 2 * sx0 - 1;                      // patch origin in the image (stride 2, pad 1)
    if (tid < 144) { s_ws[tid] = ws[tid]; s_wd[tid] = wd_t[tid]; }
    if (tid < 16) { s_bs[tid] = bs[tid]; s_bd[tid] = bd[tid]; }
    const uint8_t* g = gray + (int64_t)b * stride * H;
    for (int i = tid; i < PH * PW; i += 256) {
        const int y = iy0 + i / PW, x = ix0 + i % PW;
        float px = 0.f;
        if (y >= 0 && y < H && x >= 0 && x < W) {
            const float raw = (y >= mask0 && y < mask1) ? 0.f : (float)g[(int64_t)y * stride + x];
            px = (raw - 128.0f) / 128.0f;
        }
        patch[i] = px;
    }
    __syncthreads();
    for (int i = tid; i < R * 4; i += 256) {                              // stem: (region pixel, 4 channels)
        const int r = i >> 2, c4 = (i & 3) * 4;
        const int ry = r / RW, rx = r - ry * RW;
        const int sy = sy0 + ry, sx = sx0 + rx;
        float4 res = make_float4(0.f, 0.f, 0.f, 0.f);
        if (sy >= 0 && sy < Ho && sx >= 0 && sx < Wo) {
            float v[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) v[t] = patch[(2 * ry + t / 3) * PW + 2 * rx + t % 3];
            float rr[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float acc = s_bs[c4 + j];
#pragma unroll
                for (int t = 0; t < 9; ++t) acc = fmaf(v[t], s_ws[(c4 + j) * 9 + t], acc);
                rr[j] = relu6f(acc);
            }
            res = make_float4(rr[0], rr[1], rr[2], rr[3]);
        }
        *reinterpret_cast<float4*>(h + r * 16 + c4) = res;
    }
    __syncthreads();
    for (int i = tid; i < SB_TW * SB_TH * 4; i += 256) {                  // depthwise 3x3 + ReLU6: (output pixel, 4 channels)
        const int o = i >> 2, c4 = (i & 3) * 4;
        const int oy = o / SB_TW, ox = o - oy * SB_TW;
        float t[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = s_bd[c4 + j];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const float4 x = *reinterpret_cast<const float4*>(h + ((oy + dy) * RW + ox + dx) * 16 + c4);
                const float* wv = s_wd + (dy * 3 + dx) * 16 + c4;
                t[0] = fmaf(x.x, wv[0], t[0]); t[1] = fmaf(x.y, wv[1], t[1]); t[2] = fmaf(x.z, wv[2], t[2]); t[3] = fmaf(x.w, wv[3], t[3]);
            }
        *reinterpret_cast<float4*>(d + o * 16 + c4) = make_float4(relu6f(t[0]), relu6f(t[1]), relu6f(t[2]), relu6f(t[3]));
    }
    __syncthreads();
    {                                                                     // projection 16 -> 8: thread = (output channel, 4 pixels)
        const int co = tid & 7, og = tid >> 3;
        float wp[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) wp[k] = wp_t[k * 8 + co];
        const float bpv = bp[co];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int o = og + 32 * i;
            float acc = 0.f;
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) {
                const float4 x = *reinterpret_cast<const float4*>(d + o * 16 + k4 * 4);
                acc = fmaf(x.x, wp[k4 * 4 + 0], acc); acc = fmaf(x.y, wp[k4 * 4 + 1], acc);
                acc = fmaf(x.z, wp[k4 * 4 + 2], acc); acc = fmaf(x.w, wp[k4 * 4 + 3], acc);
            }
            const int oy = oy0 + o / SB_TW, ox = ox0 + o % SB_TW;
            if (oy < Ho && ox < Wo) out[(((int64_t)b * Ho + oy) * Wo + ox) * 8 + co] = acc + bpv;
        }
    }
}

// stem + one fused kernel per inverted-residual block (v->blocks, built at create time)
// skip_mode: 0 = the rotating buffers, every tile; 1 = the planned layers write into their own buffers, every tile (the calibration pass); 2 = ... and leave
// the tiles of their constant rectangles out.  *out_ptr = the backbone's output.
static int vlad_backbone_fused(omni_vlad* v, const uint8_t* gray_dev, int stride, int batch, int fisheye_mask, int skip_mode, const float** out_ptr) {
    hipStream_t st = v->ctx->stream;
    const int H = v->H, W = v->W;
    const float* in_over = nullptr;          // the previous layer wrote into a buffer of its own: this layer's input (instead of buf[cur])
    int m0, m1;
    omni_fisheye_mask_rows(H, fisheye_mask, &m0, &m1);
    const VladLayerDev& S = v->layers[0];
    int cur = 0, rc;
    size_t first = 0;
    const bool stem_fuse = v->cfg[omni::CFG_VLAD_STEM_FUSE] != 0;           // 0: stem and block 0 as two kernels (A/B and parity tests)
    const VladFusedBlock* B0 = v->blocks.empty() ? nullptr : &v->blocks[0];
    if (stem_fuse && B0 && S.cout == 16 && S.stride == 2 && !B0->expand && !B0->res && B0->cin == 16 && B0->hid == 16 && B0->cout == 8 &&
        B0->stride == 1) {
        // stem + block 0 in one kernel: the 16-channel stem map never reaches HBM
        hipLaunchKernelGGL(vlad_stem_b0_kernel, dim3(cdiv(B0->wout, SB_TW) * cdiv(B0->hout, SB_TH), batch), dim3(256), 0, st, gray_dev, stride, H,
                           W, m0, m1, S.hout, S.wout, S.w, S.b, B0->wd_t, B0->bd, B0->wp_t, B0->bp, skip_mode ? v->mskip[0].buf : v->buf[1],
                           skip_mode == 2 ? v->mskip[0].ty0 : 0, skip_mode == 2 ? v->mskip[0].ty1 : 0, v->mskip.empty() ? 0 : v->mskip[0].tx0, v->mskip.empty() ? 0 : v->mskip[0].tx1);
        OMNI_LAUNCH_CHECK();
        cur = 1; first = 1;
        if (skip_mode) in_over = v->mskip[0].buf;
    } else {
        hipLaunchKernelGGL(vlad_stem4_kernel, dim3(cdiv(S.hout * S.wout * (S.cout / 4), 256), batch), dim3(256), 0, st, gray_dev, stride, H, W, m0,
                           m1, S.hout, S.wout, S.cout, S.stride, S.w, S.b, v->buf[0]);
        OMNI_LAUNCH_CHECK();
    }
    for (size_t bi = first; bi < v->blocks.size(); ++bi) {
        const VladFusedBlock& B = v->blocks[bi];
        const int64_t Pin = (int64_t)batch * B.hin * B.win, Pout = (int64_t)batch * B.hout * B.wout;
        const float* const in = in_over ? in_over : v->buf[cur];
        in_over = nullptr;
        if (v->prec == OMNI_PREC_F16 && B.hblob) {
            VladHBlockArgs ha;
            ha.in = in; ha.out = v->buf[(cur + 1) % 3]; ha.blob = B.hblob; ha.bp = B.bp;
            ha.Hi = B.hin; ha.Wi = B.win; ha.Ho = B.hout; ha.Wo = B.wout; ha.cin = B.cin; ha.hid = B.hid; ha.cout = B.cout; ha.res = B.res; ha.batch = batch;
            if ((rc = launch_vlad_hblock(st, ha, B.stride))) return rc;
            cur = (cur + 1) % 3;
            continue;
        }
        if (v->sblock && B.sblob) {
            VladSBlockArgs sa;
            const bool own = skip_mode && bi < v->mskip.size();          // a planned layer: its own output buffer (the ring is not advanced)
            sa.in = in; sa.out = own ? v->mskip[bi].buf : v->buf[(cur + 1) % 3]; sa.blob = B.sblob; sa.bp = B.bp;
            if (own && skip_mode == 2) { sa.sk_y0 = v->mskip[bi].ty0; sa.sk_y1 = v->mskip[bi].ty1; sa.sk_x0 = v->mskip[bi].tx0; sa.sk_w = v->mskip[bi].tx1 - v->mskip[bi].tx0; }
            sa.Hi = B.hin; sa.Wi = B.win; sa.Ho = B.hout; sa.Wo = B.wout; sa.cin = B.cin; sa.hid = B.hid; sa.cout = B.cout; sa.res = B.res; sa.batch = batch;
            sa.n_cu = v->ctx->prop.multiProcessorCount > 0 ? v->ctx->prop.multiProcessorCount : 256; sa.trace = nullptr; sa.dbg = 0;
            sa.persist = v->cfg[omni::CFG_VLAD_SB_PERSIST];
            if ((rc = launch_vlad_sblock(st, sa, B.stride))) return rc;
            if (own) in_over = v->mskip[bi].buf;
            else cur = (cur + 1) % 3;
            continue;
        }
        if (B.mblob && B.hin * B.win <= v->mblock_max_px) {                  // per-image size: batch-independent numerics
            VladMBlockArgs m;
            m.in = in; m.out = v->buf[(cur + 1) % 3]; m.blob = B.mblob; m.bp = B.bp;
            m.Hi = B.hin; m.Wi = B.win; m.Ho = B.hout; m.Wo = B.wout; m.cin = B.cin; m.hid = B.hid; m.cout = B.cout; m.cop = B.cop;
            m.stride = B.stride; m.res = B.res; m.batch = batch;
            const int n_chunks = (B.hid + 31) / 32, tiles = cdiv(B.wout, 8) * cdiv(B.hout, 8) * batch;
            m.cpw = v->mb_cpw > 0 ? v->mb_cpw : n_chunks; m.n_groups = cdiv(n_chunks, m.cpw);
            if ((size_t)tiles * m.n_groups * 64 * B.cop * 4 > v->mb_partial_bytes) { m.cpw = n_chunks; m.n_groups = 1; }      // scratch too small: no split
            m.partial = v->mb_partial;
            if ((rc = launch_vlad_mblock(st, m))) return rc;
            cur = (cur + 1) % 3;
            continue;
        }
        if (B.expand && B.cin % 8 == 0 && B.hid % 8 == 0 && B.hin * B.win <= v->mfma_max_px && v->mfma_late) {     // per-image size: batch-independent numerics
            // low-resolution block: expand / project as f32-MFMA pointwise GEMMs, depthwise in between (three launches; the fused
            // VALU kernel has too few workgroups at these sizes and is latency-bound)
            const int e = (cur + 1) % 3, d = (cur + 2) % 3;
            if ((rc = vlad_pw_mfma(st, in, Pin, B.cin, B.hid, B.we_t, B.be, nullptr, 1, v->buf[e]))) return rc;
            const int64_t total = Pout * (B.hid / 4);
            hipLaunchKernelGGL(vlad_dw_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, st, v->buf[e], B.hin, B.win, B.hid, B.hout, B.wout,
                               B.stride, B.wd_t, B.bd, v->buf[d], total);
            OMNI_LAUNCH_CHECK();
            if ((rc = vlad_pw_mfma(st, v->buf[d], Pout, B.hid, B.cout, B.wp_t, B.bp, B.res ? in : nullptr, 0, v->buf[e]))) return rc;
            cur = e;
            continue;
        }
        VladBlockArgs a;
        a.in = in; a.out = v->buf[(cur + 1) % 3];
        a.blob = B.blob; a.bp = B.bp;
        a.Hi = B.hin; a.Wi = B.win; a.Ho = B.hout; a.Wo = B.wout; a.hid = B.hid; a.cout = B.cout; a.stride = B.stride;
        a.expand = B.expand; a.res = B.res; a.batch = batch;
        if ((rc = vlad_block(st, B.cin, a))) return rc;
        cur = (cur + 1) % 3;
    }
    *out_ptr = in_over ? in_over : v->buf[cur];
    return OMNI_OK;
}

// One dense pass over a blank masked frame with the planned layers writing into their own buffers; each layer's constant is read from the middle of its
// rectangle and written into that rectangle of every image slot.  Passes without the mask use the rotating buffers: the rectangles stay valid.
static int vlad_calibrate_mask_skip(omni_vlad* v) {
    hipStream_t st = v->ctx->stream;
    if (!v->zero_gray) {
        OMNI_HIP_TRY(hipMalloc((void**)&v->zero_gray, (size_t)v->W * v->H));
        OMNI_HIP_TRY(hipMemsetAsync(v->zero_gray, 0, (size_t)v->W * v->H, st));
    }
    const float* unused;
    int rc;
    if ((rc = vlad_backbone_fused(v, v->zero_gray, v->W, 1, 1, 1, &unused))) return rc;
    for (const omni_vlad::MaskSkip& k : v->mskip) {
        const int pix = k.oc * 4;
        const int64_t row = (int64_t)k.ow * pix, img = row * k.oh;
        if ((rc = conv_read_pixel_bytes(st, k.buf, row, 0, pix, (k.oy0 + k.oy1) / 2, (k.ox0 + k.ox1) / 2, k.vec))) return rc;
        if ((rc = conv_fill_rect_bytes(st, k.buf, v->max_batch, img, row, 0, pix, k.oy0, k.oy1, k.ox0, k.ox1, k.vec))) return rc;
    }
    v->mask_skip_ready = true;
    return OMNI_OK;
}

// Where the stem's / the blocks' outputs are constant under the fisheye mask, and the tile rectangles inside (omni_vlad::MaskSkip): integer arithmetic on
// (H, W), the layers' strides and the kernels' tile shapes (vlad_stem_b0_kernel: SB_TH x SB_TW; vlad_sblock_kernel: 8 x 8 at stride 1, 4 rows x 8 at stride 2).
// A 3x3 convolution with padding 1 at stride s reads input rows s r - 1 .. s r + 1: the zero padding is not the constant.
static void vlad_plan_mask_skip(omni_vlad* v) {
    v->mskip.clear();
    if (!v->fused || !v->sblock || !v->cfg[omni::CFG_VLAD_MASK_SKIP] || !v->cfg[omni::CFG_VLAD_STEM_FUSE] || v->blocks.empty()) return;
    const VladLayerDev& S = v->layers[0];
    const VladFusedBlock& B0 = v->blocks[0];
    if (!(S.cout == 16 && S.stride == 2 && !B0.expand && !B0.res && B0.cin == 16 && B0.hid == 16 && B0.cout == 8 && B0.stride == 1)) return;      // (vlad_stem_b0_kernel's shape)
    int m0, m1;
    omni_fisheye_mask_rows(v->H, 1, &m0, &m1);
    int a = m0, b = m1 - 1, c = 0, d = v->W - 1;                 // constant rows [a, b] x columns [c, d] (inclusive) of the current map
    auto conv3 = [&](int stride) {                               // through a 3x3 convolution, padding 1
        if (stride == 2) { a = (a + 2) / 2; b = (b - 1) >> 1; c = (c + 2) / 2; d = (d - 1) >> 1; }      // rows 2r - 1 .. 2r + 1 inside [a, b]
        else { a += 1; b -= 1; c += 1; d -= 1; }
    };
    auto plan = [&](int th, int tw, int oh, int ow, int oc) -> bool {
        omni_vlad::MaskSkip k;
        if (b < a || d < c) return false;
        k.ty0 = (a + th - 1) / th; k.ty1 = (b + 1) / th; k.tx0 = (c + tw - 1) / tw; k.tx1 = (d + 1) / tw;
        if (k.ty1 <= k.ty0 || k.tx1 <= k.tx0) return false;
        k.oy0 = k.ty0 * th; k.oy1 = k.ty1 * th; k.ox0 = k.tx0 * tw; k.ox1 = k.tx1 * tw;
        k.oh = oh; k.ow = ow; k.oc = oc;
        k.frac = (double)(k.ty1 - k.ty0) * (k.tx1 - k.tx0) / ((double)cdiv(oh, th) * cdiv(ow, tw));
        v->mskip.push_back(k);
        return true;
    };
    conv3(2);                                                    // the stem
    conv3(1);                                                    // block 0's depthwise convolution (its projection is 1x1)
    if (!plan(SB_TH, SB_TW, B0.hout, B0.wout, B0.cout)) return;
    for (size_t bi = 1; bi < v->blocks.size(); ++bi) {
        const VladFusedBlock& B = v->blocks[bi];
        if (!B.sblob || (B.cout * 4) % 16 != 0) break;
        conv3(B.stride);                                         // (expand and projection are 1x1; the residual adds two constants)
        if (!plan(B.stride == 1 ? 8 : 4, 8, B.hout, B.wout, B.cout)) break;
    }
}

static int vlad_forward(omni_vlad* v, const uint8_t* gray_dev, int stride, int batch, int fisheye_mask) {
    hipStream_t st = v->ctx->stream;
    int cur = 0, rc;
    const float* feat = nullptr;             // the backbone's output map
    if (v->fused) {
        // the mask's constant region (omni_vlad::MaskSkip): only the split block kernels know the shortened tile walk
        const bool skip = fisheye_mask && !v->mskip.empty() && v->sblock && v->prec != OMNI_PREC_F16;
        if (skip && !v->mask_skip_ready && (rc = vlad_calibrate_mask_skip(v))) return rc;
        if ((rc = vlad_backbone_fused(v, gray_dev, stride, batch, fisheye_mask, skip ? 2 : 0, &feat))) return rc;
    } else {
        if ((rc = vlad_backbone_unfused(v, gray_dev, stride, batch, fisheye_mask, &cur))) return rc;
        feat = v->buf[cur];
    }
    const int n_pos = v->hf * v->wf;
    const int64_t n_all = (int64_t)batch * n_pos;
    if (v->fused && v->K <= 32) {
        const size_t smem = ((size_t)v->Dm * v->K + 8 * v->Dm) * 4;
        hipLaunchKernelGGL(vlad_assign2_kernel, dim3((unsigned)cdiv64(n_all, 8)), dim3(256), smem, st, feat, n_all, v->Dm, v->K,
                           v->assign_wT, v->assign_b, v->assign);
        OMNI_LAUNCH_CHECK();
        hipLaunchKernelGGL(vlad_aggregate8_kernel, dim3(v->K, batch), dim3(1024), 0, st, feat, v->assign, n_pos, v->Dm, v->K,
                           v->clusters, v->vlad);
    } else {
        hipLaunchKernelGGL(vlad_assign_kernel, dim3((unsigned)cdiv64(n_all, 4)), dim3(256), 0, st, feat, n_all, v->Dm, v->K, v->assign_wT,
                           v->assign_b, v->assign);
        OMNI_LAUNCH_CHECK();
        hipLaunchKernelGGL(vlad_aggregate_kernel, dim3(v->K, batch), dim3(128), 0, st, feat, v->assign, n_pos, v->Dm, v->K, v->clusters,
                           v->vlad);
    }
    OMNI_LAUNCH_CHECK();
    const int n_in = v->K * v->Dm;
    hipLaunchKernelGGL(l2norm_rows_kernel, dim3(batch), dim3(256), 0, st, v->vlad, n_in);
    OMNI_LAUNCH_CHECK();
    if (v->fc_mfma) {
        for (int b0 = 0; b0 < batch; b0 += 32) {
            const int nb = batch - b0 < 32 ? batch - b0 : 32;
            hipLaunchKernelGGL(vlad_fc_mfma_kernel, dim3(v->out_dim / 32, FCM_KY), dim3(256), 0, st, v->vlad + (int64_t)b0 * n_in, nb, n_in, v->fc_wp,
                               v->out_dim, v->fc_part);
            OMNI_LAUNCH_CHECK();
            hipLaunchKernelGGL(vlad_fc_finish_kernel, dim3(nb), dim3(256), 0, st, v->fc_part, v->fc_b, v->out_dim, v->out + (int64_t)b0 * v->out_dim);
            OMNI_LAUNCH_CHECK();
        }
        return OMNI_OK;
    }
    for (int b0 = 0; b0 < batch; b0 += FC_MAXB) {
        const int nb = batch - b0 < FC_MAXB ? batch - b0 : FC_MAXB;
        const size_t smem = (size_t)nb * n_in * 4;
        if (v->fused) {
            OMNI_HIP_TRY(hipFuncSetAttribute((const void*)vlad_fc4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            hipLaunchKernelGGL(vlad_fc4_kernel, dim3(cdiv(v->out_dim, 16)), dim3(256), smem, st, v->vlad + (int64_t)b0 * n_in, nb, n_in, v->fc_w,
                               v->fc_b, v->out_dim, v->out + (int64_t)b0 * v->out_dim);
        } else {
            OMNI_HIP_TRY(hipFuncSetAttribute((const void*)vlad_fc_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            hipLaunchKernelGGL(vlad_fc_kernel, dim3(cdiv(v->out_dim, 4)), dim3(256), smem, st, v->vlad + (int64_t)b0 * n_in, nb, n_in, v->fc_w,
                               v->fc_b, v->out_dim, v->out + (int64_t)b0 * v->out_dim);
        }
        OMNI_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(l2norm_rows_kernel, dim3(batch), dim3(256), 0, st, v->out, v->out_dim);
    OMNI_LAUNCH_CHECK();
    return OMNI_OK;
}

}  // namespace omni

extern "C" {

omni_vlad* omni_vlad_create(omni_ctx* ctx, const omni_vlad_weights* w, int width, int height, int max_batch) {
    if (!ctx || !w || !w->layers || w->n_layers < 1) { omni::set_error("null ctx/weights"); return nullptr; }
    if (width < 32 || height < 32 || max_batch < 1 || max_batch > 256) { omni::set_error("bad size/batch"); return nullptr; }
    if (w->n_clusters < 1 || w->n_clusters > 64 || w->feat_dim < 4 || w->feat_dim > 128 || w->feat_dim % 4 || (w->n_clusters * w->feat_dim) % 4 ||
        (size_t)w->n_clusters * w->feat_dim * 4 * FC_MAXB > 150 * 1024) {
        omni::set_error("NetVLAD shape K=%d D=%d unsupported (K<=64, D<=128, D%%4==0, K*D*32 B <= 150 KB)", w->n_clusters, w->feat_dim);
        return nullptr;
    }
    (void)hipSetDevice(ctx->device);
    hipStream_t st = ctx->stream;
    omni_vlad* v = new omni_vlad();
    if (omni::config_resolve(&v->cfg) != OMNI_OK) { delete v; return nullptr; }
    v->ctx = ctx; v->W = width; v->H = height; v->max_batch = max_batch; v->K = w->n_clusters; v->Dm = w->feat_dim; v->out_dim = w->out_dim;
    int h = height, wd = width, c = 0;
    size_t max_elems = 0;
    bool ok = true;
    for (int i = 0; i < w->n_layers && ok; ++i) {
        const omni_vlad_layer& L = w->layers[i];
        VladLayerDev d{};
        d.kind = L.kind; d.cin = L.cin; d.cout = L.cout; d.stride = L.stride; d.hin = h; d.win = wd;
        if (L.kind == OMNI_VLAD_CONV3X3_RELU6 || L.kind == OMNI_VLAD_DW3X3_RELU6) { d.hout = (h - 1) / L.stride + 1; d.wout = (wd - 1) / L.stride + 1; }
        else { d.hout = h; d.wout = wd; }
        if (i == 0 && L.kind != OMNI_VLAD_CONV3X3_RELU6) { omni::set_error("layer 0 must be the stem conv"); ok = false; break; }
        if (i > 0 && (L.cin != c || L.cin % 4 || L.cout % 4 || L.kind == OMNI_VLAD_CONV3X3_RELU6)) {
            omni::set_error("layer %d: cin=%d (prev cout %d) / cout=%d must chain and be multiples of 4", i, L.cin, c, L.cout); ok = false; break;
        }
        std::vector<float> tmp;
        if (L.kind == OMNI_VLAD_CONV3X3_RELU6) {           // fold the 3 identical input channels: [cout][cin][3][3] -> [cout][9]
            tmp.assign((size_t)L.cout * 9, 0.f);
            for (int co = 0; co < L.cout; ++co) for (int ci = 0; ci < L.cin; ++ci) for (int t = 0; t < 9; ++t) tmp[co * 9 + t] += L.weight[((size_t)co * L.cin + ci) * 9 + t];
        } else if (L.kind == OMNI_VLAD_DW3X3_RELU6) {      // [C][1][3][3] -> [9][C]
            tmp.resize((size_t)L.cout * 9);
            for (int ch = 0; ch < L.cout; ++ch) for (int t = 0; t < 9; ++t) tmp[(size_t)t * L.cout + ch] = L.weight[(size_t)ch * 9 + t];
        } else {                                           // [cout][cin] -> [cin][cout]
            tmp.resize((size_t)L.cout * L.cin);
            for (int co = 0; co < L.cout; ++co) for (int ci = 0; ci < L.cin; ++ci) tmp[(size_t)ci * L.cout + co] = L.weight[(size_t)co * L.cin + ci];
        }
        if (omni::upload(&d.w, tmp.data(), tmp.size(), st) || omni::upload(&d.b, L.bias, L.cout, st)) { ok = false; break; }
        v->layers.push_back(d);
        h = d.hout; wd = d.wout; c = L.cout;
        const size_t e = (size_t)h * wd * c;
        if (e > max_elems) max_elems = e;
    }
    if (ok && c != w->feat_dim) { omni::set_error("backbone ends with %d channels, NetVLAD expects %d", c, w->feat_dim); ok = false; }
    if (ok) {   // group the layer table into inverted-residual blocks for the fused kernels
        bool fusable = v->layers.size() > 1 && v->layers[0].cout % 4 == 0;
        size_t i = 1;
        while (fusable && i < v->layers.size()) {
            VladFusedBlock B{};
            const VladLayerDev* e = nullptr;
            if (v->layers[i].kind == OMNI_VLAD_PW_RELU6) { e = &v->layers[i]; ++i; }
            if (i + 1 >= v->layers.size()) { fusable = false; break; }
            const VladLayerDev& dwl = v->layers[i];
            const VladLayerDev& pl = v->layers[i + 1];
            if (dwl.kind != OMNI_VLAD_DW3X3_RELU6 || (pl.kind != OMNI_VLAD_PW_LINEAR && pl.kind != OMNI_VLAD_PW_LINEAR_RES) ||
                (dwl.stride != 1 && dwl.stride != 2)) { fusable = false; break; }
            B.cin = e ? e->cin : dwl.cin; B.hid = dwl.cin; B.cout = pl.cout; B.stride = dwl.stride; B.expand = e ? 1 : 0;
            B.res = pl.kind == OMNI_VLAD_PW_LINEAR_RES; B.hin = dwl.hin; B.win = dwl.win; B.hout = dwl.hout; B.wout = dwl.wout;
            B.bp = pl.b; B.blob = nullptr;
            B.we_t = e ? e->w : nullptr; B.be = e ? e->b : nullptr; B.wd_t = dwl.w; B.bd = dwl.b; B.wp_t = pl.w;
            if (!omni::vlad_block_supported(B.cin, B.cout) || B.cout % 4 || (B.res && (B.stride != 1 || B.cin != B.cout)) || (!e && B.hid > 32)) { fusable = false; break; }
            {   // pack the block's weights per 32-channel chunk of the hidden layer (host copies of the layer weights, OIHW)
                const omni_vlad_layer* Le = e ? &w->layers[i - 1] : nullptr;
                const omni_vlad_layer& Ld = w->layers[i];
                const omni_vlad_layer& Lp = w->layers[i + 1];
                const int n_chunks = (B.hid + 31) / 32, blob = 32 * (B.cin + 11 + B.cout);
                std::vector<float> pk((size_t)n_chunks * blob, 0.f);
                for (int ch = 0; ch < B.hid; ++ch) {
                    float* q = pk.data() + (size_t)(ch / 32) * blob;
                    const int c = ch % 32;
                    if (Le) { for (int k = 0; k < B.cin; ++k) q[k * 32 + c] = Le->weight[(size_t)ch * B.cin + k]; q[B.cin * 32 + c] = Le->bias[ch]; }
                    for (int t = 0; t < 9; ++t) q[(B.cin + 1) * 32 + t * 32 + c] = Ld.weight[(size_t)ch * 9 + t];
                    q[(B.cin + 10) * 32 + c] = Ld.bias[ch];
                    for (int co = 0; co < B.cout; ++co) q[(B.cin + 11) * 32 + c * B.cout + co] = Lp.weight[(size_t)co * B.hid + ch];
                }
                if (omni::upload(&B.blob, pk.data(), pk.size(), st)) { fusable = false; ok = false; break; }
                B.mblob = nullptr; B.cop = ((B.cout + 31) / 32) * 32;
                if (Le && B.cin % 4 == 0 && B.cop <= 128 && omni::vlad_mblock_smem(B.cin, B.cop, B.stride) <= 160 * 1024) {
                    const int mb = 32 * (B.cin + 11 + B.cop);
                    std::vector<float> mk((size_t)n_chunks * mb, 0.f);
                    for (int ch = 0; ch < B.hid; ++ch) {
                        float* q = mk.data() + (size_t)(ch / 32) * mb;
                        const int c = ch % 32;
                        for (int k = 0; k < B.cin; ++k) q[k * 32 + c] = Le->weight[(size_t)ch * B.cin + k];
                        q[B.cin * 32 + c] = Le->bias[ch];
                        for (int t = 0; t < 9; ++t) q[(B.cin + 1) * 32 + t * 32 + c] = Ld.weight[(size_t)ch * 9 + t];
                        q[(B.cin + 10) * 32 + c] = Ld.bias[ch];
                        for (int co = 0; co < B.cout; ++co) q[(B.cin + 11) * 32 + c * B.cop + co] = Lp.weight[(size_t)co * B.hid + ch];
                    }
                    if (omni::upload(&B.mblob, mk.data(), mk.size(), st)) { fusable = false; ok = false; break; }
                }
                B.sblob = nullptr;
                if (Le && omni::vlad_sblock_supported(B.cin, B.hid, B.cout, B.stride)) {
                    std::vector<char> sk(omni::vlad_sblock_blob_bytes(B.cin, B.hid, B.cout));
                    omni::vlad_sblock_pack(B.cin, B.hid, B.cout, Le->weight, Le->bias, Ld.weight, Ld.bias, Lp.weight, sk.data());
                    if (hipMalloc(&B.sblob, sk.size()) != hipSuccess || hipMemcpy(B.sblob, sk.data(), sk.size(), hipMemcpyHostToDevice) != hipSuccess) {
                        omni::set_error("device allocation failed"); fusable = false; ok = false; break;
                    }
                }
                B.hblob = nullptr;
                if (Le && omni::vlad_hblock_supported(B.cin, B.hid, B.cout, B.stride)) {
                    std::vector<char> hk(omni::vlad_hblock_blob_bytes(B.cin, B.hid, B.cout));
                    omni::vlad_hblock_pack(B.cin, B.hid, B.cout, Le->weight, Le->bias, Ld.weight, Ld.bias, Lp.weight, hk.data());
                    if (hipMalloc(&B.hblob, hk.size()) != hipSuccess || hipMemcpy(B.hblob, hk.data(), hk.size(), hipMemcpyHostToDevice) != hipSuccess) {
                        omni::set_error("device allocation failed"); fusable = false; ok = false; break;
                    }
                }
            }
            v->blocks.push_back(B);
            i += 2;
        }
        v->fused = fusable && !v->cfg[omni::CFG_VLAD_UNFUSED];
        v->mfma_late = v->cfg[omni::CFG_VLAD_MFMA] != 0;
        // fused matrix-core block kernel for blocks whose input has at most this many pixels per image (0 disables).  Measured at 32 images
        // (profiles/r02_vlad32_*): the ten 38x30 / 19x15 blocks take 451 us on it vs 502 us as three launches each; on the 75x60 ... 300x240
        // blocks it is slower than the fp32-VALU fused kernel (one 8x8 tile per workgroup keeps 47-108 KB of LDS: 1-2 workgroups per CU
        // and every phase of a chunk is a dependent chain behind a barrier -- waves wait 50 % of their life, MFMA-busy 14-18 %).  A
        // split-fp16 variant (v_mfma_f32_32x32x16_f16, hi/lo operands: 5x less matrix time) measured SLOWER still (60 us per block): the
        // matrix pipe is not what bounds these blocks, the per-workgroup latency chain is.
        v->sblock = v->cfg[omni::CFG_VLAD_SBLOCK] != 0;
        v->mblock_max_px = v->cfg[omni::CFG_VLAD_MBLOCK_PX];
        if (v->cfg[omni::CFG_VLAD_MFMA_PX] > 0) v->mfma_max_px = v->cfg[omni::CFG_VLAD_MFMA_PX];
    }
    if (ok) {
        v->hf = h; v->wf = wd; v->buf_elems = max_elems * max_batch;
        std::vector<float> awT((size_t)v->Dm * v->K);
        for (int k = 0; k < v->K; ++k) for (int d = 0; d < v->Dm; ++d) awT[(size_t)d * v->K + k] = w->assign_w[(size_t)k * v->Dm + d];
        const size_t n_in = (size_t)v->K * v->Dm;
        ok = !omni::upload(&v->assign_wT, awT.data(), awT.size(), st) && !omni::upload(&v->assign_b, w->assign_b, v->K, st) &&
             !omni::upload(&v->clusters, w->clusters, n_in, st) && !omni::upload(&v->fc_w, w->fc_w, n_in * v->out_dim, st) &&
             !omni::upload(&v->fc_b, w->fc_b, v->out_dim, st);
        {   // FC on the matrix cores: W [out][n_in] -> [out / 32][n_in / 4][32 rows][4] (vlad_fc_mfma_kernel); OMNI_VLAD_FC_MFMA=0 keeps the VALU kernel
            const bool want = v->cfg[omni::CFG_VLAD_FC_MFMA] != 0 && v->fused;
            if (ok && want && v->out_dim % 32 == 0 && n_in % (4 * 2 * FCM_KY * 4) == 0) {
                const size_t groups = n_in / 4;
                std::vector<float> pk(n_in * (size_t)v->out_dim);
                for (int j = 0; j < v->out_dim; ++j)
                    for (size_t g = 0; g < groups; ++g)
                        memcpy(&pk[(((size_t)(j / 32) * groups + g) * 32 + (j % 32)) * 4], &w->fc_w[(size_t)j * n_in + g * 4], 16);
                ok = !omni::upload(&v->fc_wp, pk.data(), pk.size(), st) && hipMalloc((void**)&v->fc_part, (size_t)FCM_KY * 32 * v->out_dim * 4) == hipSuccess;
                v->fc_mfma = ok;
            }
        }
        {   // scratch of the hidden-layer split (OMNI_VLAD_MBLOCK_CPW = chunks per workgroup; 0 = no split)
            v->mb_cpw = v->cfg[omni::CFG_VLAD_MBLOCK_CPW];             // measured: splitting does not pay (same total issue-bound work + a reduce launch per block)
            size_t need = 0, max_tiles = 0;
            for (auto& B : v->blocks) {
                if (!B.mblob || B.hin * B.win > v->mblock_max_px) continue;
                const size_t tiles = (size_t)omni::cdiv(B.wout, 8) * omni::cdiv(B.hout, 8) * max_batch;
                const int n_chunks = (B.hid + 31) / 32, cpw = v->mb_cpw > 0 ? v->mb_cpw : n_chunks;
                need = std::max(need, tiles * omni::cdiv(n_chunks, cpw) * 64 * B.cop * 4);
                max_tiles = std::max(max_tiles, tiles);
            }
            if (ok && need && need <= ((size_t)1 << 30)) {
                (void)max_tiles;
                ok = hipMalloc((void**)&v->mb_partial, need) == hipSuccess;
                v->mb_partial_bytes = ok ? need : 0;
            }
        }
        for (int i = 0; i < 3 && ok; ++i) ok = hipMalloc((void**)&v->buf[i], v->buf_elems * 4) == hipSuccess;
        if (ok) {
            omni::vlad_plan_mask_skip(v);
            for (auto& k : v->mskip)
                ok = ok && hipMalloc((void**)&k.buf, (size_t)max_batch * k.oh * k.ow * k.oc * 4) == hipSuccess && hipMalloc(&k.vec, (size_t)k.oc * 4) == hipSuccess;
        }
        ok = ok && hipMalloc((void**)&v->assign, (size_t)max_batch * h * wd * v->K * 4) == hipSuccess &&
             hipMalloc((void**)&v->vlad, (size_t)max_batch * n_in * 4) == hipSuccess &&
             hipMalloc((void**)&v->out, (size_t)max_batch * v->out_dim * 4) == hipSuccess &&
             hipMalloc((void**)&v->gray_stage, (size_t)max_batch * width * height) == hipSuccess;
        if (!ok && !*omni_last_error()) omni::set_error("device allocation failed");
    }
    if (!ok) { omni_vlad_destroy(v); return nullptr; }
    return v;
}

void omni_vlad_destroy(omni_vlad* v) {
    if (!v) return;
    (void)hipSetDevice(v->ctx->device);
    (void)hipStreamSynchronize(v->ctx->stream);
    for (auto& L : v->layers) { if (L.w) (void)hipFree(L.w); if (L.b) (void)hipFree(L.b); }
    for (auto& B : v->blocks) { if (B.blob) (void)hipFree(B.blob); if (B.mblob) (void)hipFree(B.mblob); if (B.hblob) (void)hipFree(B.hblob); if (B.sblob) (void)hipFree(B.sblob); }
    void* ptrs[] = {v->mb_partial, v->fc_wp, v->fc_part, v->assign_wT, v->assign_b, v->clusters, v->fc_w, v->fc_b, v->buf[0], v->buf[1], v->buf[2], v->assign, v->vlad, v->out, v->gray_stage};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (auto& k : v->mskip) { if (k.buf) (void)hipFree(k.buf); if (k.vec) (void)hipFree(k.vec); }
    if (v->zero_gray) (void)hipFree(v->zero_gray);
    v->hstage.release();
    delete v;
}

int64_t omni_vlad_pack_block(int cin, int hid, int cout, int stride, const float* we, const float* be, const float* wd, const float* bd,
                             const float* wp, void* out, int64_t out_bytes) {
    if (!omni::vlad_sblock_supported(cin, hid, cout, stride)) return -1;
    const int64_t need = (int64_t)omni::vlad_sblock_blob_bytes(cin, hid, cout);
    if (!out) return need;
    if (!we || !be || !wd || !bd || !wp || out_bytes < need) { omni::set_error("omni_vlad_pack_block: null weights or %lld < %lld bytes", (long long)out_bytes, (long long)need); return -2; }
    omni::vlad_sblock_pack(cin, hid, cout, we, be, wd, bd, wp, out);
    return need;
}

int omni_vlad_set_precision(omni_vlad* v, int precision) {
    OMNI_REQUIRE(v, OMNI_ERR_INVALID, "null handle");
    OMNI_REQUIRE(precision == OMNI_PREC_F32 || precision == OMNI_PREC_F16, OMNI_ERR_INVALID, "precision %d", precision);
    if (precision == OMNI_PREC_F16) {
        OMNI_REQUIRE(v->fused, OMNI_ERR_INVALID, "OMNI_PREC_F16 needs the fused block path (layer table not groupable into inverted-residual blocks)");
        int n = 0;
        for (auto& B : v->blocks) n += B.hblob != nullptr;
        OMNI_REQUIRE(n > 0, OMNI_ERR_INVALID, "OMNI_PREC_F16: no block of this layer table has an fp16 kernel");
    }
    std::lock_guard<std::mutex> lk(v->mu);
    v->prec = precision;
    return OMNI_OK;
}

int omni_vlad_enqueue_dev(omni_vlad* v, const uint8_t* gray_dev, int stride, int batch, int fisheye_mask) {
    omni::TraceRange trace_range("MobileNetVLAD enqueue");
    OMNI_REQUIRE(v && gray_dev, OMNI_ERR_INVALID, "null argument");
    OMNI_REQUIRE(batch >= 1 && batch <= v->max_batch, OMNI_ERR_CAPACITY, "batch=%d outside [1,%d]", batch, v->max_batch);
    OMNI_REQUIRE(stride >= v->W, OMNI_ERR_INVALID, "stride=%d < width=%d", stride, v->W);
    std::lock_guard<std::mutex> lk(v->mu);
    (void)hipSetDevice(v->ctx->device);
    return omni::vlad_forward(v, gray_dev, stride, batch, fisheye_mask);
}

int omni_vlad_fetch(omni_vlad* v, int batch, float* out) {
    OMNI_REQUIRE(v && out, OMNI_ERR_INVALID, "null argument");
    OMNI_REQUIRE(batch >= 1 && batch <= v->max_batch, OMNI_ERR_CAPACITY, "batch=%d outside [1,%d]", batch, v->max_batch);
    std::lock_guard<std::mutex> lk(v->mu);
    (void)hipSetDevice(v->ctx->device);
    const size_t bytes = (size_t)batch * v->out_dim * 4;
    int rc;
    if ((rc = v->hstage.ensure(bytes))) return rc;
    OMNI_HIP_TRY(hipMemcpyAsync(v->hstage.p, v->out, bytes, hipMemcpyDeviceToHost, v->ctx->stream));
    OMNI_HIP_TRY(hipStreamSynchronize(v->ctx->stream));
    memcpy(out, v->hstage.p, bytes);
    return OMNI_OK;
}

int omni_vlad_mask_skip_layers(const omni_vlad* v, double* frac, int max_layers) {
    if (!v) return 0;
    const int n = (int)v->mskip.size();
    for (int i = 0; i < n && i < max_layers && frac; ++i) frac[i] = v->mskip[i].frac;
    return n;
}

int omni_vlad_dev_output(omni_vlad* v, const float** out_dev) {
    OMNI_REQUIRE(v && out_dev, OMNI_ERR_INVALID, "null argument");
    *out_dev = v->out;
    return OMNI_OK;
}

int omni_vlad_infer(omni_vlad* v, const uint8_t* gray_host, int stride, int batch, int fisheye_mask, float* out) {
    OMNI_REQUIRE(v && gray_host && out, OMNI_ERR_INVALID, "null argument");
    OMNI_REQUIRE(batch >= 1 && batch <= v->max_batch, OMNI_ERR_CAPACITY, "batch=%d outside [1,%d]", batch, v->max_batch);
    OMNI_REQUIRE(stride >= v->W, OMNI_ERR_INVALID, "stride=%d < width=%d", stride, v->W);
    {
        std::lock_guard<std::mutex> lk(v->mu);
        (void)hipSetDevice(v->ctx->device);
        const size_t n = (size_t)batch * v->H * v->W;
        int rc;
        if ((rc = v->hstage.ensure(n))) return rc;
        uint8_t* h = v->hstage.as<uint8_t>();
        for (int b = 0; b < batch; ++b)
            for (int y = 0; y < v->H; ++y) memcpy(h + ((size_t)b * v->H + y) * v->W, gray_host + ((size_t)b * v->H + y) * stride, v->W);
        OMNI_HIP_TRY(hipMemcpyAsync(v->gray_stage, h, n, hipMemcpyHostToDevice, v->ctx->stream));
        OMNI_HIP_TRY(hipStreamSynchronize(v->ctx->stream));
        if ((rc = omni::vlad_forward(v, v->gray_stage, v->W, batch, fisheye_mask))) return rc;
    }
    return omni_vlad_fetch(v, batch, out);
}

}  // extern "C"
