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
#include "common.h"

struct VladLayerDev { int kind, cin, cout, stride, hin, win, hout, wout; float* w; float* b; };

struct omni_vlad {
    omni_ctx* ctx = nullptr;
    int W = 0, H = 0, max_batch = 0, K = 0, Dm = 0, out_dim = 0, hf = 0, wf = 0;
    std::vector<VladLayerDev> layers;
    float *assign_wT = nullptr, *assign_b = nullptr, *clusters = nullptr, *fc_w = nullptr, *fc_b = nullptr;
    float *buf[3] = {nullptr, nullptr, nullptr};   // rotating activation buffers
    size_t buf_elems = 0;
    float *assign = nullptr, *vlad = nullptr, *out = nullptr;
    uint8_t* gray_stage = nullptr;
    omni::HostBuf hstage;
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

static int upload(float** dst, const float* src, size_t n, hipStream_t st) {
    OMNI_HIP_TRY(hipMalloc((void**)dst, n * 4));
    OMNI_HIP_TRY(hipMemcpyAsync(*dst, src, n * 4, hipMemcpyHostToDevice, st));
    OMNI_HIP_TRY(hipStreamSynchronize(st));
    return OMNI_OK;
}

static int vlad_forward(omni_vlad* v, const uint8_t* gray_dev, int stride, int batch, int fisheye_mask) {
    hipStream_t st = v->ctx->stream;
    const int H = v->H, W = v->W;
    const int m0 = fisheye_mask ? H * 3 / 4 : H, m1 = fisheye_mask ? H * 3 / 4 + H / 4 : H;
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
    const int n_pos = v->hf * v->wf;
    const int64_t n_all = (int64_t)batch * n_pos;
    hipLaunchKernelGGL(vlad_assign_kernel, dim3((unsigned)cdiv64(n_all, 4)), dim3(256), 0, st, v->buf[cur], n_all, v->Dm, v->K, v->assign_wT,
                       v->assign_b, v->assign);
    OMNI_LAUNCH_CHECK();
    hipLaunchKernelGGL(vlad_aggregate_kernel, dim3(v->K, batch), dim3(128), 0, st, v->buf[cur], v->assign, n_pos, v->Dm, v->K, v->clusters,
                       v->vlad);
    OMNI_LAUNCH_CHECK();
    const int n_in = v->K * v->Dm;
    hipLaunchKernelGGL(l2norm_rows_kernel, dim3(batch), dim3(256), 0, st, v->vlad, n_in);
    OMNI_LAUNCH_CHECK();
    for (int b0 = 0; b0 < batch; b0 += FC_MAXB) {
        const int nb = batch - b0 < FC_MAXB ? batch - b0 : FC_MAXB;
        const size_t smem = (size_t)nb * n_in * 4;
        OMNI_HIP_TRY(hipFuncSetAttribute((const void*)vlad_fc_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        hipLaunchKernelGGL(vlad_fc_kernel, dim3(cdiv(v->out_dim, 4)), dim3(256), smem, st, v->vlad + (int64_t)b0 * n_in, nb, n_in, v->fc_w,
                           v->fc_b, v->out_dim, v->out + (int64_t)b0 * v->out_dim);
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
    if (ok) {
        v->hf = h; v->wf = wd; v->buf_elems = max_elems * max_batch;
        std::vector<float> awT((size_t)v->Dm * v->K);
        for (int k = 0; k < v->K; ++k) for (int d = 0; d < v->Dm; ++d) awT[(size_t)d * v->K + k] = w->assign_w[(size_t)k * v->Dm + d];
        const size_t n_in = (size_t)v->K * v->Dm;
        ok = !omni::upload(&v->assign_wT, awT.data(), awT.size(), st) && !omni::upload(&v->assign_b, w->assign_b, v->K, st) &&
             !omni::upload(&v->clusters, w->clusters, n_in, st) && !omni::upload(&v->fc_w, w->fc_w, n_in * v->out_dim, st) &&
             !omni::upload(&v->fc_b, w->fc_b, v->out_dim, st);
        for (int i = 0; i < 3 && ok; ++i) ok = hipMalloc((void**)&v->buf[i], v->buf_elems * 4) == hipSuccess;
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
    void* ptrs[] = {v->assign_wT, v->assign_b, v->clusters, v->fc_w, v->fc_b, v->buf[0], v->buf[1], v->buf[2], v->assign, v->vlad, v->out, v->gray_stage};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    v->hstage.release();
    delete v;
}

int omni_vlad_enqueue_dev(omni_vlad* v, const uint8_t* gray_dev, int stride, int batch, int fisheye_mask) {
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
