// MobileNetVLAD inverted-residual blocks at fp16 operand precision (OMNI_PREC_F16): the reference's engine is an fp16 TensorRT plan
// (swarm_loop/launch/realsense.launch:10-11, mobilenetvlad_tensorrt.cpp:4-14).
//
// One launch per block, one workgroup per 8x8 output tile, the hidden layer walked in chunks of 32 channels that never leave LDS:
//     expand   h[region px][32]  = ReLU6(x[px][cin] . We + be)     v_mfma_f32_32x32x16_f16, A = weights (registers), B = pixels (LDS)
//     depthwise d[64 px][32]     = ReLU6(dw3x3(h) + bd)            v_pk_fma_f32 on f16 -> f32 converted taps, f32 weights
//     project  acc[cout][64 px] += Wp . d                          v_mfma_f32_32x32x16_f16, fp32 accumulators across the chunks
// What is fp16: the expand / project OPERANDS (block input, both weight matrices, hidden and depthwise activations as stored in LDS).
// What stays fp32: every accumulation, the depthwise weights and biases, the projection bias, the residual add and the block's input and
// output tensors in HBM -- the residual stream is never rounded, so the error does not compound over the 16 blocks.
//   * the expand bias rides in two spare K slots of the padded input (x = 1 at in-image pixels, weights bias_hi / bias_lo): out-of-image
//     region pixels are all-zero rows, so h = ReLU6(0) = 0 there -- exactly the zero padding the depthwise conv needs -- with no mask code;
//   * weights are pre-packed per chunk in MFMA fragment order and fetched straight into registers (1 KiB per fragment, L2 resident);
//   * LDS rows are odd multiples of 16 bytes: the ds_read_b128 B-fragment reads are conflict-free;
//   * two barriers per chunk: [depthwise c] | [project c, expand c+1] |
// LDS per workgroup: 33 KB (stride 1) / 46-67 KB (stride 2) against 108 KB of the fp32 kernel: 3-4 workgroups per CU instead of 1.
#include "common.h"
#include "vlad_h.h"

namespace omni {

typedef _Float16 half8v __attribute__((ext_vector_type(8)));
typedef _Float16 half4v __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float floatx16h __attribute__((ext_vector_type(16)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float relu6h(float v) { return __builtin_amdgcn_fmed3f(v, 0.f, 6.f); }

#define HB_HS 80          // bytes per pixel row of h / d: 32 halfs + 16

template <int STRIDE, int KS, int NT>
__global__ void __launch_bounds__(256)
vlad_hblock_kernel(VladHBlockArgs a) {
    constexpr int RW = 7 * STRIDE + 3, R = RW * RW, RT = (R + 31) / 32, RP = RT * 32;
    constexpr int XS = KS * 32 + 16;                         // bytes per pixel row of xin
    constexpr int CB = KS * 1024 + NT * 2048 + 1280;         // bytes per chunk of the weight blob
    constexpr int PQ = (2 * NT + 3) / 4;                     // projection tile pairs per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* xin = smem;                                        // [RP][XS]
    char* h = xin + RP * XS;                                 // [RP][80]
    char* d = h + RP * HB_HS;                                // [64][80]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, kk = lane >> 5;
    const int cin = a.cin;
    const int tiles_x = (a.Wo + 7) >> 3, tiles_y = (a.Ho + 7) >> 3;
    const int b = blockIdx.x / (tiles_x * tiles_y), tr = blockIdx.x - b * tiles_x * tiles_y;
    const int oy0 = (tr / tiles_x) * 8, ox0 = (tr % tiles_x) * 8;
    const int iy0 = oy0 * STRIDE - 1, ix0 = ox0 * STRIDE - 1;
    const float* inb = a.in + (int64_t)b * a.Hi * a.Wi * cin;
    const int n_chunks = (a.hid + 31) >> 5;
    const char* blob = reinterpret_cast<const char*>(a.blob);

    {   // input region -> fp16, K padded to KS*16: [cin channels | 1 1 (in-image) | 0 ...]
        const int q4 = cin >> 2;
        for (int e = tid; e < RP * q4; e += 256) {
            const int r = e / q4, q = e - r * q4;
            const int ry = r / RW, gy = iy0 + ry, gx = ix0 + r - ry * RW;
            float4v v = {0.f, 0.f, 0.f, 0.f};
            if (r < R && gy >= 0 && gy < a.Hi && gx >= 0 && gx < a.Wi) v = *reinterpret_cast<const float4v*>(inb + ((int64_t)gy * a.Wi + gx) * cin + q * 4);
            *reinterpret_cast<half4v*>(xin + r * XS + q * 8) = __builtin_convertvector(v, half4v);
        }
        const int tail0 = cin * 2;                           // byte offset of the first pad slot (cin % 8 == 0: 16-byte aligned)
        for (int r = tid; r < RP; r += 256) {
            const int ry = r / RW, gy = iy0 + ry, gx = ix0 + r - ry * RW;
            const bool inside = r < R && gy >= 0 && gy < a.Hi && gx >= 0 && gx < a.Wi;
            half8v t = {0, 0, 0, 0, 0, 0, 0, 0};
            if (inside) { t[0] = (_Float16)1.f; t[1] = (_Float16)1.f; }
            *reinterpret_cast<half8v*>(xin + r * XS + tail0) = t;
            const half8v z = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int o = tail0 + 16; o < KS * 32; o += 16) *reinterpret_cast<half8v*>(xin + r * XS + o) = z;
        }
    }

    floatx16h acc[PQ];
#pragma unroll
    for (int p = 0; p < PQ; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;

    // expand of chunk c: region tiles t = wave, wave + 4, ...
    auto expand = [&](int c) {
        half8v we[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) we[ks] = *reinterpret_cast<const half8v*>(blob + (int64_t)c * CB + ks * 1024 + lane * 16);
        for (int t = wave; t < RT; t += 4) {
            const char* xb = xin + (t * 32 + n) * XS + kk * 16;
            floatx16h e;
#pragma unroll
            for (int r = 0; r < 16; ++r) e[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                e = __builtin_amdgcn_mfma_f32_32x32x16_f16(we[ks], *reinterpret_cast<const half8v*>(xb + ks * 32), e, 0, 0, 0);
            char* hp = h + (t * 32 + n) * HB_HS + kk * 8;      // channels 8 g + 4 kk + (0..3) of pixel n
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4v v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = relu6h(e[4 * g + j]);
                *reinterpret_cast<half4v*>(hp + g * 16) = __builtin_convertvector(v, half4v);
            }
        }
    };

    __syncthreads();                                         // xin complete
    expand(0);
    __syncthreads();

    const int c2 = tid & 15, strip = tid >> 4, orow = strip >> 1, oxs = (strip & 1) * 4;
    for (int c = 0; c < n_chunks; ++c) {
        const char* cb = blob + (int64_t)c * CB;
        // this chunk's projection fragments and depthwise taps (consumed after the LDS phases below: the loads fly meanwhile)
        half8v wp[PQ][2];
#pragma unroll
        for (int p = 0; p < PQ; ++p) {
            const int pr = wave + 4 * p, m = pr >> 1;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                wp[p][ks] = pr < 2 * NT ? *reinterpret_cast<const half8v*>(cb + KS * 1024 + (m * 2 + ks) * 1024 + lane * 16) : half8v{0, 0, 0, 0, 0, 0, 0, 0};
        }
        float2v wd[10];
        {
            const float2v* wsrc = reinterpret_cast<const float2v*>(cb + KS * 1024 + NT * 2048) + c2;
#pragma unroll
            for (int t = 0; t < 10; ++t) wd[t] = wsrc[t * 16];
        }
        // ---- depthwise 3x3 + ReLU6: thread = (channel pair c2, four consecutive output pixels of one row)
        {
            constexpr int NCOL = 3 * STRIDE + 3;             // 6 (stride 1) / 9 (stride 2) input columns feed 4 outputs
            const char* hp = h + ((orow * STRIDE) * RW + oxs * STRIDE) * HB_HS + c2 * 4;
            float2v o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = wd[9];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                float2v v[NCOL];
#pragma unroll
                for (int x = 0; x < NCOL; ++x) v[x] = __builtin_convertvector(*reinterpret_cast<const half2v*>(hp + (dy * RW + x) * HB_HS), float2v);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) o[j] = __builtin_elementwise_fma(v[j * STRIDE + dx], wd[dy * 3 + dx], o[j]);
            }
            char* dp = d + (orow * 8 + oxs) * HB_HS + c2 * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float2v r; r[0] = relu6h(o[j][0]); r[1] = relu6h(o[j][1]);
                *reinterpret_cast<half2v*>(dp + j * HB_HS) = __builtin_convertvector(r, half2v);
            }
        }
        __syncthreads();                                     // d complete; every reader is done with h
        // ---- projection of this chunk (reads d) and expand of the next one (writes h)
#pragma unroll
        for (int p = 0; p < PQ; ++p) {
            const int pr = wave + 4 * p;
            if (pr < 2 * NT) {
                const char* db = d + ((pr & 1) * 32 + n) * HB_HS + kk * 16;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wp[p][ks], *reinterpret_cast<const half8v*>(db + ks * 32), acc[p], 0, 0, 0);
            }
        }
        if (c + 1 < n_chunks) expand(c + 1);
        __syncthreads();                                     // h of the next chunk complete; every reader is done with d
    }

    // ---- epilogue: + bias (+ the fp32 block input), fp32 NHWC stores of 4 consecutive channels
#pragma unroll
    for (int p = 0; p < PQ; ++p) {
        const int pr = wave + 4 * p;
        if (pr >= 2 * NT) continue;
        const int m = pr >> 1, o = (pr & 1) * 32 + n, oy = oy0 + (o >> 3), ox = ox0 + (o & 7);
        if (oy >= a.Ho || ox >= a.Wo) continue;
        float* op = a.out + (((int64_t)b * a.Ho + oy) * a.Wo + ox) * a.cout;
        const float* rp = a.in + (((int64_t)b * a.Hi + oy) * a.Wi + ox) * cin;       // residual: stride 1, cin == cout
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ch = m * 32 + 8 * g + 4 * kk;
            if (ch >= a.cout) continue;
            float4v v = *reinterpret_cast<const float4v*>(a.bp + ch);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += acc[p][4 * g + j];
            if (a.res) v += *reinterpret_cast<const float4v*>(rp + ch);
            *reinterpret_cast<float4v*>(op + ch) = v;
        }
    }
}

static inline uint16_t hb_f2h(float v) { const __half hv = __float2half_rn(v); uint16_t u; memcpy(&u, &hv, 2); return u; }
static inline float hb_h2f(uint16_t u) { __half hv; memcpy(&hv, &u, 2); return __half2float(hv); }

bool vlad_hblock_supported(int cin, int hid, int cout, int stride) {
    const int ks = (cin + 2 + 15) / 16, nt = (cout + 31) / 32;
    return cin % 8 == 0 && cout % 4 == 0 && hid >= 1 && ks >= 1 && ks <= 4 && (nt == 1 || nt == 2 || nt == 4 || nt == 3) && (stride == 1 || stride == 2);
}

size_t vlad_hblock_blob_bytes(int cin, int hid, int cout) {
    const int ks = (cin + 2 + 15) / 16, nt = (cout + 31) / 32 == 3 ? 4 : (cout + 31) / 32;
    return (size_t)((hid + 31) / 32) * (ks * 1024 + nt * 2048 + 1280);
}

// we [hid][cin], be [hid], wd [hid][9], bd [hid], wp [cout][hid] (the layer table's OIHW weights) -> per-chunk fragment-order blob
void vlad_hblock_pack(int cin, int hid, int cout, const float* we, const float* be, const float* wd, const float* bd, const float* wp, void* out) {
    const int KS = (cin + 2 + 15) / 16, NT = (cout + 31) / 32 == 3 ? 4 : (cout + 31) / 32, CB = KS * 1024 + NT * 2048 + 1280;
    const int n_chunks = (hid + 31) / 32;
    memset(out, 0, (size_t)n_chunks * CB);
    for (int c = 0; c < n_chunks; ++c) {
        char* cb = reinterpret_cast<char*>(out) + (size_t)c * CB;
        for (int ks = 0; ks < KS; ++ks)                                   // A fragments of We^T: row = hidden channel, k = input channel
            for (int l = 0; l < 64; ++l)
                for (int e = 0; e < 8; ++e) {
                    const int ch = c * 32 + (l & 31), k = ks * 16 + (l >> 5) * 8 + e;
                    float v = 0.f;
                    if (ch < hid) {
                        if (k < cin) v = we[(size_t)ch * cin + k];
                        else if (k == cin) v = hb_h2f(hb_f2h(be[ch]));
                        else if (k == cin + 1) v = be[ch] - hb_h2f(hb_f2h(be[ch]));
                    }
                    reinterpret_cast<uint16_t*>(cb + ks * 1024)[l * 8 + e] = hb_f2h(v);
                }
        for (int m = 0; m < NT; ++m)                                      // A fragments of Wp: row = output channel, k = hidden channel of the chunk
            for (int ks = 0; ks < 2; ++ks)
                for (int l = 0; l < 64; ++l)
                    for (int e = 0; e < 8; ++e) {
                        const int co = m * 32 + (l & 31), hc = c * 32 + ks * 16 + (l >> 5) * 8 + e;
                        const float v = (co < cout && hc < hid) ? wp[(size_t)co * hid + hc] : 0.f;
                        reinterpret_cast<uint16_t*>(cb + KS * 1024 + (m * 2 + ks) * 1024)[l * 8 + e] = hb_f2h(v);
                    }
        float* wdo = reinterpret_cast<float*>(cb + KS * 1024 + NT * 2048);       // [10][32]: nine taps + bias
        for (int i = 0; i < 32; ++i) {
            const int ch = c * 32 + i;
            if (ch >= hid) continue;
            for (int t = 0; t < 9; ++t) wdo[t * 32 + i] = wd[(size_t)ch * 9 + t];
            wdo[9 * 32 + i] = bd[ch];
        }
    }
}

template <int STRIDE, int KS, int NT>
static int launch_hb(hipStream_t st, const VladHBlockArgs& a) {
    constexpr int RW = 7 * STRIDE + 3, RP = ((RW * RW + 31) / 32) * 32;
    constexpr size_t smem = (size_t)RP * (KS * 32 + 16) + (size_t)RP * HB_HS + 64 * HB_HS;
    auto kfn = vlad_hblock_kernel<STRIDE, KS, NT>;
    static DynSmemState attr;
    OMNI_HIP_TRY(ensure_dyn_smem(attr, (const void*)kfn, smem));
    hipLaunchKernelGGL(kfn, dim3(cdiv(a.Wo, 8) * cdiv(a.Ho, 8) * a.batch), dim3(256), smem, st, a);
    OMNI_LAUNCH_CHECK();
    return OMNI_OK;
}

int launch_vlad_hblock(hipStream_t st, const VladHBlockArgs& a, int stride) {
    const int ks = (a.cin + 2 + 15) / 16, nt0 = (a.cout + 31) / 32, nt = nt0 == 3 ? 4 : nt0;
#define HB(S, K, N) if (stride == S && ks == K && nt == N) return launch_hb<S, K, N>(st, a)
#define HB_K(S, K) HB(S, K, 1); HB(S, K, 2); HB(S, K, 4)
    HB_K(1, 1); HB_K(1, 2); HB_K(1, 3); HB_K(1, 4);
    HB_K(2, 1); HB_K(2, 2); HB_K(2, 3); HB_K(2, 4);
#undef HB_K
#undef HB
    set_error("vlad_hblock: no instantiation for cin=%d cout=%d stride=%d", a.cin, a.cout, stride);
    return OMNI_ERR_INVALID;
}

}  // namespace omni
