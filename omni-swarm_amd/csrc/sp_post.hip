// SuperPoint post-processing kernels (see sp_post.h for the reference map).
//
// NMS2 is order dependent (a serial pass over candidates in row-major order, superpoint_tensorrt.cpp:265-283).
// Its outcome has the closed form (checked against a literal simulation in tests/):
//     alive(p)   <=> cand(p) and no EARLIER (row-major) alive q in the 9x9 window with conf(q) > conf(p)
//     survive(p) <=> alive(p) and no alive q anywhere in the window with conf(q) > conf(p)
// alive() only depends on strictly-higher-confidence earlier neighbours, so it is a DAG and a chaotic relaxation
// (unknown -> alive/dead as soon as all relevant neighbours are decided) converges to the unique serial answer.
// One 1024-thread workgroup per image runs the relaxation with the 2-bit pixel state plane held in LDS
// (600x480 -> 72 KB of the CU's 160 KB), then selects the top max_num survivors with an in-LDS bitonic sort on
// 64-bit (confidence, row-major index) keys: final order = confidence desc, index asc (the fixed spec; the
// reference's std::sort leaves ties unspecified).
#include "sp_post.h"
#include "topk.h"

namespace omni {

#define NMS_THREADS 1024
#define NMS_SORT_CAP 8192          // keys held in LDS while sorting (64 KB)
#define ST_NONE 0u
#define ST_UNKNOWN 1u
#define ST_ALIVE 2u
#define ST_DEAD 3u

// ---- getKeyPoints: mask = prob > thres; findNonZero  (:167-173) -------------------------------------------------
// 2048 pixels per workgroup, ONE global atomic per workgroup (per-wave atomics on a single counter serialise in L2).
// The list order is irrelevant: NMS2's scan order is the row-major PIXEL order, which the state plane encodes.
#define CAND_PX_PER_BLOCK 2048
__global__ void __launch_bounds__(256)
sp_cand_kernel(const float* __restrict__ semi, int hw, float thres, int* __restrict__ cand, int* __restrict__ counters) {
    __shared__ int s_wave_cnt[4];
    __shared__ int s_base;
    const int b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int p0 = blockIdx.x * CAND_PX_PER_BLOCK + tid * 8;          // 8 consecutive pixels per thread
    const float* sm = semi + (int64_t)b * hw;
    float v[8];
    if (p0 + 8 <= hw) {
        const float4 a = *reinterpret_cast<const float4*>(sm + p0), c = *reinterpret_cast<const float4*>(sm + p0 + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = c.x; v[5] = c.y; v[6] = c.z; v[7] = c.w;
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (p0 + j < hw) ? sm[p0 + j] : -3.0e38f;
    }
    unsigned mask = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) mask |= (v[j] > thres) ? (1u << j) : 0u;
    const int mine = __popc(mask);
    int incl = mine;                                                   // inclusive prefix over the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off, 64); if (lane >= off) incl += t; }
    if (lane == 63) s_wave_cnt[wave] = incl;
    __syncthreads();
    if (tid == 0) {
        const int total = s_wave_cnt[0] + s_wave_cnt[1] + s_wave_cnt[2] + s_wave_cnt[3];
        s_base = total ? atomicAdd(&counters[b * 4 + 0], total) : 0;
    }
    __syncthreads();
    int pos = s_base + incl - mine;
    for (int w = 0; w < wave; ++w) pos += s_wave_cnt[w];
    int* out = cand + (int64_t)b * hw;
#pragma unroll
    for (int j = 0; j < 8; ++j) if (mask & (1u << j)) out[pos++] = p0 + j;
}

__device__ __forceinline__ unsigned st_get(const unsigned* st, int p) { return (st[p >> 4] >> ((p & 15) * 2)) & 3u; }

// ---- NMS2 (:237-310) ----------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(NMS_THREADS)
sp_nms_kernel(const float* __restrict__ semi, int W, int H, int /*r: fixed at 4*/, int max_num, const int* __restrict__ cand,
              int* __restrict__ counters, uint64_t* __restrict__ surv_keys, float* __restrict__ kps_xy,
              float* __restrict__ scores, int* __restrict__ n_kps, int state_words, int smem_main_bytes) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];   // all LDS is dynamic: keeps the base 16-B aligned
    unsigned* st = reinterpret_cast<unsigned*>(smem_raw);
    uint64_t* keys = reinterpret_cast<uint64_t*>(smem_raw);     // reused after the relaxation
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int hw = W * H;
    const float* sm = semi + (int64_t)b * hw;
    const int* cd = cand + (int64_t)b * hw;
    uint64_t* sk = surv_keys + (int64_t)b * hw;
    const int n_cand = counters[b * 4 + 0];

    for (int i = tid; i < state_words; i += NMS_THREADS) st[i] = 0u;
    __syncthreads();
    for (int ci = tid; ci < n_cand; ci += NMS_THREADS) {
        const int p = cd[ci];
        atomicOr(&st[p >> 4], ST_UNKNOWN << ((p & 15) * 2));      // grid(vv,uu) = 1  (:259)
    }
    __syncthreads();

    // relaxation of alive().  The neighbour scan is branch-free and fully unrolled (R = 4): all LDS state reads and all
    // (predicated) heat-map loads of a candidate are independent, so their latencies overlap instead of chaining.
    constexpr int R = 4;
    int iters = 0;
    for (;;) {
        int changed = 0;
        for (int ci = tid; ci < n_cand; ci += NMS_THREADS) {
            const int p = cd[ci];
            if (st_get(st, p) != ST_UNKNOWN) continue;
            const int y = p / W, x = p - y * W;
            const float c0 = sm[p];
            bool any_alive = false, any_unknown = false;
#pragma unroll
            for (int k = -R; k <= 0; ++k) {
#pragma unroll
                for (int j = -R; j <= R; ++j) {
                    if (k == 0 && j >= 0) continue;                   // earlier in row-major order only (compile time)
                    const int v = y + k, u = x + j;
                    const bool inb = (v >= 0) && (u >= 0) && (u < W); // fixed spec: out-of-image neighbours ignored
                    const int q = inb ? v * W + u : p;
                    const unsigned sq = inb ? st_get(st, q) : ST_NONE;
                    const bool live = (sq == ST_UNKNOWN) || (sq == ST_ALIVE);
                    const float cq = live ? sm[q] : 0.f;
                    const bool higher = live && (cq > c0);
                    any_alive |= higher && (sq == ST_ALIVE);
                    any_unknown |= higher && (sq == ST_UNKNOWN);
                }
            }
            if (any_alive) { atomicOr(&st[p >> 4], 2u << ((p & 15) * 2)); changed = 1; }            // 01 -> 11 dead
            else if (!any_unknown) { atomicXor(&st[p >> 4], 3u << ((p & 15) * 2)); changed = 1; }   // 01 -> 10 alive
        }
        ++iters;
        if (!__syncthreads_or(changed)) break;
        if (iters > (1 << 20)) break;   // bound every spin: a chain can never be longer than the candidate count
    }

    // survive(): alive and not beaten by any alive neighbour (an earlier one cannot exist; a later one can)
    int& s_nsurv = *reinterpret_cast<int*>(smem_raw + smem_main_bytes);
    if (tid == 0) s_nsurv = 0;
    __syncthreads();
    for (int ci = tid; ci < n_cand; ci += NMS_THREADS) {
        const int p = cd[ci];
        if (st_get(st, p) != ST_ALIVE) continue;
        const int y = p / W, x = p - y * W;
        const float c0 = sm[p];
        bool beaten = false;
#pragma unroll
        for (int k = 0; k <= R; ++k) {                                 // only LATER neighbours can beat an alive point
#pragma unroll
            for (int j = -R; j <= R; ++j) {
                if (k == 0 && j <= 0) continue;
                const int v = y + k, u = x + j;
                const bool inb = (v < H) && (u >= 0) && (u < W);
                const int q = inb ? v * W + u : p;
                const bool al = inb && (st_get(st, q) == ST_ALIVE);
                const float cq = al ? sm[q] : 0.f;
                beaten |= al && (cq > c0);
            }
        }
        if (!beaten) sk[atomicAdd(&s_nsurv, 1)] = omni_make_key(c0, (uint32_t)p);
    }
    __syncthreads();
    const int n_surv = s_nsurv;
    __syncthreads();   // everyone has read the state plane; LDS is reused for sorting from here on

    // top max_num by (conf desc, index asc): running best in keys[0, max_num), batches appended behind it
    const int M = max_num;
    for (int i = tid; i < M; i += NMS_THREADS) keys[i] = OMNI_KEY_EMPTY;
    const int batch_cap = NMS_SORT_CAP - M;
    int off = 0;
    do {
        const int cnt = (n_surv - off) < batch_cap ? (n_surv - off) : batch_cap;
        int n_pow2 = 2;
        while (n_pow2 < M + cnt) n_pow2 <<= 1;
        __syncthreads();
        for (int i = tid; i < n_pow2 - M; i += NMS_THREADS) keys[M + i] = (i < cnt) ? sk[off + i] : OMNI_KEY_EMPTY;
        __syncthreads();
        bitonic_sort_desc(keys, n_pow2, tid, NMS_THREADS);
        off += cnt;
    } while (off < n_surv);
    __syncthreads();

    const int n_out = n_surv < M ? n_surv : M;
    for (int i = tid; i < M; i += NMS_THREADS) {
        float x = 0.f, y = 0.f, c = 0.f;
        if (i < n_out) {
            const uint64_t key = keys[i];
            const uint32_t p = 0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFull);
            c = omni_orderable_f32((uint32_t)(key >> 32));
            y = (float)(p / (uint32_t)W);
            x = (float)(p % (uint32_t)W);
        }
        kps_xy[((int64_t)b * M + i) * 2 + 0] = x;
        kps_xy[((int64_t)b * M + i) * 2 + 1] = y;
        scores[(int64_t)b * M + i] = c;
    }
    if (tid == 0) { n_kps[b] = n_out; counters[b * 4 + 1] = n_surv; counters[b * 4 + 2] = iters; }
}

// ---- computeDescriptors (:192-230) --------------------------------------------------------------------------------
// Pass 1: torch::grid_sampler (bilinear, zeros, align_corners=false) of every key point; 8 key points per workgroup,
// thread = channel (NHWC: the 4 taps are coalesced 1 KiB rows).
#define SAMPLE_KPB 8
__global__ void __launch_bounds__(256)
sp_sample_kernel(const float* __restrict__ desc_nhwc, int W, int H, int max_num, const float* __restrict__ kps_xy,
                 const int* __restrict__ n_kps, float* __restrict__ raw_desc) {
    const int b = blockIdx.y;
    const int c = threadIdx.x;
    const int Wc = W >> 3, Hc = H >> 3;
    const int n = n_kps[b];
    const int i0 = blockIdx.x * SAMPLE_KPB;
    if (i0 >= n) return;
    const float* dm = desc_nhwc + (int64_t)b * Hc * Wc * 256;
    float* raw = raw_desc + (int64_t)b * max_num * 256;
    const float fW = (float)W, fH = (float)H, fWc = (float)Wc, fHc = (float)Hc;
#pragma unroll
    for (int kk = 0; kk < SAMPLE_KPB; ++kk) {
        const int i = i0 + kk;
        if (i >= n) break;
        const float kx = kps_xy[((int64_t)b * max_num + i) * 2 + 0];
        const float ky = kps_xy[((int64_t)b * max_num + i) * 2 + 1];
        // grid = 2*k/size - 1 (:204-205); unnormalise with align_corners=false: ((g+1)*size_c - 1)/2
        const float gx = __fsub_rn(__fdiv_rn(__fmul_rn(2.0f, kx), fW), 1.0f);
        const float gy = __fsub_rn(__fdiv_rn(__fmul_rn(2.0f, ky), fH), 1.0f);
        const float ix = __fdiv_rn(__fsub_rn(__fmul_rn(__fadd_rn(gx, 1.0f), fWc), 1.0f), 2.0f);
        const float iy = __fdiv_rn(__fsub_rn(__fmul_rn(__fadd_rn(gy, 1.0f), fHc), 1.0f), 2.0f);
        const float fx0 = floorf(ix), fy0 = floorf(iy);
        const int x0 = (int)fx0, y0 = (int)fy0, x1 = x0 + 1, y1 = y0 + 1;
        const float wx1 = ix - fx0, wy1 = iy - fy0, wx0 = (fx0 + 1.f) - ix, wy0 = (fy0 + 1.f) - iy;
        float v = 0.f;
        if (y0 >= 0 && y0 < Hc) {
            if (x0 >= 0 && x0 < Wc) v += dm[((int64_t)y0 * Wc + x0) * 256 + c] * (wx0 * wy0);
            if (x1 >= 0 && x1 < Wc) v += dm[((int64_t)y0 * Wc + x1) * 256 + c] * (wx1 * wy0);
        }
        if (y1 >= 0 && y1 < Hc) {
            if (x0 >= 0 && x0 < Wc) v += dm[((int64_t)y1 * Wc + x0) * 256 + c] * (wx0 * wy1);
            if (x1 >= 0 && x1 < Wc) v += dm[((int64_t)y1 * Wc + x1) * 256 + c] * (wx1 * wy1);
        }
        raw[(int64_t)i * 256 + c] = v;
    }
}

// Pass 1b: divide every CHANNEL by its L2 norm ACROSS the image's key points -- torch::norm(desc, 2, /*dim=*/1) on the
// [256, n] tensor (:214): the reference normalises channels across key points, not descriptors across channels.
// One workgroup per image, thread = channel, sequential (deterministic) sum over the key points.
__global__ void __launch_bounds__(256)
sp_chan_norm_kernel(int max_num, const int* __restrict__ n_kps, float* __restrict__ raw_desc) {
    const int b = blockIdx.x, c = threadIdx.x;
    const int n = n_kps[b];
    float* raw = raw_desc + (int64_t)b * max_num * 256;
    float ss = 0.f;
#pragma unroll 8
    for (int i = 0; i < n; ++i) { const float v = raw[(int64_t)i * 256 + c]; ss = fmaf(v, v, ss); }
    const float dn = sqrtf(ss);
#pragma unroll 8
    for (int i = 0; i < n; ++i) raw[(int64_t)i * 256 + c] = raw[(int64_t)i * 256 + c] / dn;   // 0/0 -> NaN as in the reference
}

// Pass 2: (d - mean) * comp^T  (:221) -- 4 key points per workgroup, thread = (64-channel part, output dim)
__global__ void __launch_bounds__(256)
sp_pca_kernel(const float* __restrict__ raw_desc, int max_num, const int* __restrict__ n_kps, int pca_dim,
              const float* __restrict__ compT, const float* __restrict__ mean, float* __restrict__ out) {
    __shared__ float sx[4][256];
    __shared__ float sp[4][4][64];
    const int b = blockIdx.y;
    const int n = n_kps[b];
    const int i0 = blockIdx.x * 4;
    if (i0 >= n) return;
    const int tid = threadIdx.x;
    for (int kp = 0; kp < 4; ++kp) {
        const int i = i0 + kp;
        sx[kp][tid] = (i < n) ? (raw_desc[((int64_t)b * max_num + i) * 256 + tid] - mean[tid]) : 0.f;
    }
    __syncthreads();
    const int part = tid >> 6;
    for (int j0 = 0; j0 < pca_dim; j0 += 64) {
        const int j = j0 + (tid & 63);
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        if (j < pca_dim) {
            for (int cc = 0; cc < 64; ++cc) {
                const int c = part * 64 + cc;
                const float w = compT[(int64_t)c * pca_dim + j];
#pragma unroll
                for (int kp = 0; kp < 4; ++kp) acc[kp] = fmaf(sx[kp][c], w, acc[kp]);
            }
        }
#pragma unroll
        for (int kp = 0; kp < 4; ++kp) sp[kp][part][tid & 63] = acc[kp];
        __syncthreads();
        if (tid < 64 && j0 + tid < pca_dim) {
            for (int kp = 0; kp < 4; ++kp) {
                const int i = i0 + kp;
                if (i < n)
                    out[((int64_t)b * max_num + i) * pca_dim + j0 + tid] =
                        (sp[kp][0][tid] + sp[kp][1][tid]) + (sp[kp][2][tid] + sp[kp][3][tid]);
            }
        }
        __syncthreads();
    }
}

__global__ void copy_raw_desc_kernel(const float* __restrict__ raw, int max_num, const int* __restrict__ n_kps,
                                     float* __restrict__ out) {
    const int b = blockIdx.y;
    const int i = blockIdx.x;
    if (i >= n_kps[b]) return;
    out[((int64_t)b * max_num + i) * 256 + threadIdx.x] = raw[((int64_t)b * max_num + i) * 256 + threadIdx.x];
}

// ---- layout helpers -----------------------------------------------------------------------------------------------
__global__ void transpose_cl_kernel(const float* __restrict__ in, float* __restrict__ out, int C, int HW, int to_nhwc) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 256 threads: 32 x 8
    const float* ib = in + (int64_t)b * C * HW;
    float* ob = out + (int64_t)b * C * HW;
    if (to_nhwc) {            // in [C][HW] -> out [HW][C]
        for (int r = ty; r < 32; r += 8) { int c = c0 + r, p = p0 + tx; tile[r][tx] = (c < C && p < HW) ? ib[(int64_t)c * HW + p] : 0.f; }
        __syncthreads();
        for (int r = ty; r < 32; r += 8) { int p = p0 + r, c = c0 + tx; if (c < C && p < HW) ob[(int64_t)p * C + c] = tile[tx][r]; }
    } else {                  // in [HW][C] -> out [C][HW]
        for (int r = ty; r < 32; r += 8) { int p = p0 + r, c = c0 + tx; tile[r][tx] = (c < C && p < HW) ? ib[(int64_t)p * C + c] : 0.f; }
        __syncthreads();
        for (int r = ty; r < 32; r += 8) { int c = c0 + r, p = p0 + tx; if (c < C && p < HW) ob[(int64_t)c * HW + p] = tile[tx][r]; }
    }
}

int nchw_to_nhwc(hipStream_t stream, const float* in, float* out, int batch, int C, int HW) {
    hipLaunchKernelGGL(transpose_cl_kernel, dim3(cdiv(HW, 32), cdiv(C, 32), batch), dim3(256), 0, stream, in, out, C, HW, 1);
    OMNI_LAUNCH_CHECK();
    return OMNI_OK;
}
int nhwc_to_nchw(hipStream_t stream, const float* in, float* out, int batch, int C, int HW) {
    hipLaunchKernelGGL(transpose_cl_kernel, dim3(cdiv(HW, 32), cdiv(C, 32), batch), dim3(256), 0, stream, in, out, C, HW, 0);
    OMNI_LAUNCH_CHECK();
    return OMNI_OK;
}

int sp_postprocess(hipStream_t stream, const SpPostParams& p, const SpPostBuffers& b, const float* semi,
                   const float* desc_nhwc, int batch) {
    const int hw = p.width * p.height;
    const int state_words = cdiv(hw, 16);
    size_t smem = (size_t)state_words * 4;
    if (smem < (size_t)NMS_SORT_CAP * 8) smem = (size_t)NMS_SORT_CAP * 8;
    const int smem_main = (int)smem;
    smem += 16;
    OMNI_REQUIRE(smem <= 160 * 1024, OMNI_ERR_CAPACITY, "image %dx%d too large for the in-LDS NMS state plane", p.width, p.height);
    OMNI_REQUIRE(p.max_num >= 1 && p.max_num <= 1024, OMNI_ERR_CAPACITY, "max_num=%d outside [1,1024]", p.max_num);
    OMNI_HIP_TRY(hipMemsetAsync(b.counters, 0, (size_t)batch * 4 * sizeof(int), stream));
    hipLaunchKernelGGL(sp_cand_kernel, dim3(cdiv(hw, CAND_PX_PER_BLOCK), batch), dim3(256), 0, stream, semi, hw, p.thres, b.cand, b.counters);
    OMNI_LAUNCH_CHECK();
    OMNI_HIP_TRY(hipFuncSetAttribute((const void*)sp_nms_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(sp_nms_kernel, dim3(batch), dim3(NMS_THREADS), smem, stream, semi, p.width, p.height, p.dist_thresh,
                       p.max_num, b.cand, b.counters, b.surv_keys, b.kps_xy, b.scores, b.n_kps, state_words, smem_main);
    OMNI_LAUNCH_CHECK();
    hipLaunchKernelGGL(sp_sample_kernel, dim3(cdiv(p.max_num, SAMPLE_KPB), batch), dim3(256), 0, stream, desc_nhwc, p.width, p.height,
                       p.max_num, b.kps_xy, b.n_kps, b.raw_desc);
    OMNI_LAUNCH_CHECK();
    hipLaunchKernelGGL(sp_chan_norm_kernel, dim3(batch), dim3(256), 0, stream, p.max_num, b.n_kps, b.raw_desc);
    OMNI_LAUNCH_CHECK();
    if (p.pca_dim > 0) {
        hipLaunchKernelGGL(sp_pca_kernel, dim3(cdiv(p.max_num, 4), batch), dim3(256), 0, stream, b.raw_desc, p.max_num, b.n_kps,
                           p.pca_dim, b.pca_compT, b.pca_mean, b.desc_out);
    } else {
        hipLaunchKernelGGL(copy_raw_desc_kernel, dim3(p.max_num, batch), dim3(256), 0, stream, b.raw_desc, p.max_num, b.n_kps,
                           b.desc_out);
    }
    OMNI_LAUNCH_CHECK();
    return OMNI_OK;
}

}  // namespace omni
