// Local-descriptor matcher: drop-in for cv::BFMatcher(cv::NORM_L2, crossCheck=true).match(query, train)
//   call sites: swarm_loop/src/loop_cam.cpp:147-150 (up <-> down), swarm_loop/src/loop_detector.cpp:564-567 (new <-> old)
//
// OpenCV 3.4 semantics (core/src/batch_distance.cpp, K=1, crosscheck): every TRAIN row picks its nearest QUERY row
// (first minimum wins); every query keeps, among the train rows that picked it, the nearest (lowest train index on
// ties); matches are emitted in ascending query order.  OMNI_BF_MUTUAL gives the strict mutual-NN variant.
//
// One workgroup per 64x64 tile of every pair's distance matrix (4x4 register tile per thread); "first minimum wins" is a
// 64-bit atomicMin on (distance bits << 32 | index) into the pair's row / column best arrays; a second small kernel per
// pair does the cross-check and the ordered compaction.
// Distances: d = sqrtf(sum_k (a_k-b_k)^2), sequential in k, mul and add NOT contracted -- bit-identical to the
// scalar C oracle (oracle/csrc/oracle.c:l2_dist).
#include "common.h"

// The distance arithmetic below must not be contracted into FMAs: the sum of squares has to round exactly like the
// scalar reference loop.  Plain operators under this pragma carry no 'contract' flag (HIP's __fmul_rn/__fadd_rn are
// header inlines parsed under the default -ffp-contract=fast and DO get fused).
#pragma clang fp contract(off)

namespace omni {

#define BF_THREADS 256
#define BF_TILE 64
#define BF_MAX_N 1024
#define BF_MAX_DIM 256

__device__ __forceinline__ unsigned long long bf_key(float d, int idx) {
    return ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)idx;
}

// Kernel 1: one workgroup per (pair, 64-query tile, 64-train tile): the 64x64 distance tile with a 4x4 register tile per
// thread; row / column minima go to the pair's global best arrays with 64-bit atomicMin on (distance bits << 32 | index)
// -- order independent, so "first minimum wins" holds however the tiles are scheduled.
__global__ void __launch_bounds__(BF_THREADS)
bf_tile_kernel(int max_n, int dim, const float* __restrict__ q_base, int64_t q_stride, const int* __restrict__ nq_arr,
               const float* __restrict__ t_base, int64_t t_stride, const int* __restrict__ nt_arr,
               unsigned long long* __restrict__ best /*[pairs][2][max_n]*/) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int p = blockIdx.z;
    const int tid = threadIdx.x;
    int nq = nq_arr[p], nt = nt_arr[p];
    nq = nq < 0 ? 0 : (nq > max_n ? max_n : nq);
    nt = nt < 0 ? 0 : (nt > max_n ? max_n : nt);
    const int i0 = blockIdx.y * BF_TILE, j0 = blockIdx.x * BF_TILE;
    if (i0 >= nq || j0 >= nt) return;
    const float* q = q_base + (int64_t)p * q_stride;
    const float* t = t_base + (int64_t)p * t_stride;
    const int ld = dim + 4;
    float* as = reinterpret_cast<float*>(smem_raw);                                    // [64][ld] query tile
    float* bs = as + BF_TILE * ld;                                                     // [64][ld] train tile
    unsigned long long* rowbest = best + (int64_t)p * 2 * max_n;
    unsigned long long* colbest = rowbest + max_n;
    for (int idx = tid; idx < BF_TILE * (dim >> 2); idx += BF_THREADS) {
        const int r = idx / (dim >> 2), c4 = idx % (dim >> 2);
        float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vb = va;
        if (i0 + r < nq) va = *reinterpret_cast<const float4*>(q + (int64_t)(i0 + r) * dim + c4 * 4);
        if (j0 + r < nt) vb = *reinterpret_cast<const float4*>(t + (int64_t)(j0 + r) * dim + c4 * 4);
        *reinterpret_cast<float4*>(as + r * ld + c4 * 4) = va;
        *reinterpret_cast<float4*>(bs + r * ld + c4 * 4) = vb;
    }
    __syncthreads();
    const int ty = tid >> 4, tx = tid & 15;
    float acc[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[r][c] = 0.f;
    for (int k = 0; k < dim; k += 4) {
        float4 a[4], b[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) a[r] = *reinterpret_cast<const float4*>(as + (ty * 4 + r) * ld + k);
#pragma unroll
        for (int c = 0; c < 4; ++c) b[c] = *reinterpret_cast<const float4*>(bs + (tx * 4 + c) * ld + k);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                // (a-b)^2 accumulated sequentially in k, no fma contraction (matches the C oracle bit for bit;
                // a-b == -(b-a) exactly, so both match directions see the same value)
                const float d0 = a[r].x - b[c].x, d1 = a[r].y - b[c].y, d2 = a[r].z - b[c].z, d3 = a[r].w - b[c].w;
                float sacc = acc[r][c];
                sacc = sacc + d0 * d0;
                sacc = sacc + d1 * d1;
                sacc = sacc + d2 * d2;
                sacc = sacc + d3 * d3;
                acc[r][c] = sacc;
            }
    }
    // tile-local minima first (LDS-free: per-thread over its 4x4, then one atomic per row / column per thread)
    unsigned long long rmin[4], cmin[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) rmin[r] = ~0ull;
#pragma unroll
    for (int c = 0; c < 4; ++c) cmin[c] = ~0ull;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = i0 + ty * 4 + r;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int j = j0 + tx * 4 + c;
            if (i < nq && j < nt) {
                const float d = (float)sqrt((double)acc[r][c]);   // correctly rounded fp32 sqrt (53 >= 2*24+2: double rounding is innocuous)
                const unsigned long long kr = bf_key(d, j), kc = bf_key(d, i);
                rmin[r] = kr < rmin[r] ? kr : rmin[r];
                cmin[c] = kc < cmin[c] ? kc : cmin[c];
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) if (rmin[r] != ~0ull) atomicMin(&rowbest[i0 + ty * 4 + r], rmin[r]);
#pragma unroll
    for (int c = 0; c < 4; ++c) if (cmin[c] != ~0ull) atomicMin(&colbest[j0 + tx * 4 + c], cmin[c]);
}

// Kernel 2: one workgroup per pair: cross-check + ordered compaction
__global__ void __launch_bounds__(BF_THREADS)
bf_cross_kernel(int max_n, int mode, const int* __restrict__ nq_arr, const int* __restrict__ nt_arr,
                const unsigned long long* __restrict__ best, int* __restrict__ out_qidx, int* __restrict__ out_tidx,
                float* __restrict__ out_dist, int* __restrict__ out_n) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    unsigned long long* qbest = reinterpret_cast<unsigned long long*>(smem_raw);      // [max_n] cross-check result
    const int p = blockIdx.x;
    const int tid = threadIdx.x;
    int nq = nq_arr[p], nt = nt_arr[p];
    nq = nq < 0 ? 0 : (nq > max_n ? max_n : nq);
    nt = nt < 0 ? 0 : (nt > max_n ? max_n : nt);
    const unsigned long long* rowbest = best + (int64_t)p * 2 * max_n;
    const unsigned long long* colbest = rowbest + max_n;
    for (int i = tid; i < max_n; i += BF_THREADS) qbest[i] = ~0ull;
    __syncthreads();
    if (mode == OMNI_BF_OPENCV) {
        // for train j ascending: idx = tidx[j]; if (tdist[j] < dist[idx]) {dist[idx] = tdist[j]; nidx[idx] = j;}
        // == per query, min over (d, j) of the trains that picked it
        for (int j = tid; j < nt; j += BF_THREADS) {
            const unsigned long long cb = colbest[j];
            if (cb == ~0ull) continue;                       // nq == 0
            const int i = (int)(cb & 0xFFFFFFFFull);
            atomicMin(&qbest[i], (cb & 0xFFFFFFFF00000000ull) | (unsigned)j);
        }
    } else {
        for (int i = tid; i < nq; i += BF_THREADS) {
            const unsigned long long rb = rowbest[i];
            if (rb == ~0ull) continue;                       // nt == 0
            const int j = (int)(rb & 0xFFFFFFFFull);
            if ((int)(colbest[j] & 0xFFFFFFFFull) == i) qbest[i] = rb;
        }
    }
    __syncthreads();
    // ordered compaction by query index: wave 0 walks the queries 64 at a time
    if (tid < 64) {
        int base = 0;
        int* oq = out_qidx + (int64_t)p * max_n;
        int* ot = out_tidx + (int64_t)p * max_n;
        float* od = out_dist + (int64_t)p * max_n;
        for (int i0 = 0; i0 < nq; i0 += 64) {
            const int i = i0 + tid;
            const unsigned long long bq = (i < nq) ? qbest[i] : ~0ull;
            const bool has = (bq != ~0ull);
            const unsigned long long m = __ballot(has);
            if (has) {
                const int pos = base + __popcll(m & ((1ull << tid) - 1ull));
                oq[pos] = i;
                ot[pos] = (int)(bq & 0xFFFFFFFFull);
                od[pos] = __uint_as_float((unsigned)(bq >> 32));
            }
            base += __popcll(m);
        }
        if (tid == 0) out_n[p] = base;
    }
}

static int bf_launch(omni_ctx* ctx, int n_pairs, int max_n, int dim, int mode, const float* q, int64_t qs, const int* nq,
                     const float* t, int64_t ts, const int* nt, int* oq, int* ot, float* od, int* on) {
    OMNI_REQUIRE(dim >= 4 && dim % 4 == 0 && dim <= BF_MAX_DIM, OMNI_ERR_INVALID, "dim=%d must be a multiple of 4 in [4,%d]", dim, BF_MAX_DIM);
    OMNI_REQUIRE(max_n >= 1 && max_n <= BF_MAX_N, OMNI_ERR_CAPACITY, "max_n=%d outside [1,%d]", max_n, BF_MAX_N);
    OMNI_REQUIRE(mode == OMNI_BF_OPENCV || mode == OMNI_BF_MUTUAL, OMNI_ERR_INVALID, "bad mode %d", mode);
    int rc;
    const size_t best_bytes = (size_t)n_pairs * 2 * max_n * 8;
    if ((rc = ctx->scratch2.ensure(best_bytes))) return rc;          // stream-ordered reuse: every user of scratch2 runs on ctx->stream
    unsigned long long* best = ctx->scratch2.as<unsigned long long>();
    OMNI_HIP_TRY(hipMemsetAsync(best, 0xFF, best_bytes, ctx->stream));
    const size_t smem1 = (size_t)2 * BF_TILE * (dim + 4) * 4;
    OMNI_HIP_TRY(hipFuncSetAttribute((const void*)bf_tile_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem1));
    const int tiles = cdiv(max_n, BF_TILE);
    hipLaunchKernelGGL(bf_tile_kernel, dim3(tiles, tiles, n_pairs), dim3(BF_THREADS), smem1, ctx->stream, max_n, dim, q, qs, nq, t, ts, nt, best);
    OMNI_LAUNCH_CHECK();
    hipLaunchKernelGGL(bf_cross_kernel, dim3(n_pairs), dim3(BF_THREADS), (size_t)max_n * 8, ctx->stream, max_n, mode, nq, nt, best, oq, ot,
                       od, on);
    OMNI_LAUNCH_CHECK();
    return OMNI_OK;
}

}  // namespace omni

extern "C" {

int omni_bf_match_batched_dev(omni_ctx* ctx, int n_pairs, int max_n, int dim, int mode, const float* q_dev, int64_t q_stride,
                              const int* nq_dev, const float* t_dev, int64_t t_stride, const int* nt_dev, int* q_idx_dev,
                              int* t_idx_dev, float* dist_dev, int* n_matches_dev) {
    omni::TraceRange trace_range("BF match (batched)");
    OMNI_REQUIRE(ctx && q_dev && t_dev && nq_dev && nt_dev && q_idx_dev && t_idx_dev && dist_dev && n_matches_dev,
                 OMNI_ERR_INVALID, "null argument");
    OMNI_REQUIRE(n_pairs >= 1, OMNI_ERR_INVALID, "n_pairs=%d", n_pairs);
    std::lock_guard<std::mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    return omni::bf_launch(ctx, n_pairs, max_n, dim, mode, q_dev, q_stride, nq_dev, t_dev, t_stride, nt_dev, q_idx_dev,
                           t_idx_dev, dist_dev, n_matches_dev);
}

int omni_bf_match(omni_ctx* ctx, const float* q_host, int nq, const float* t_host, int nt, int dim, int mode, int* q_idx,
                  int* t_idx, float* dist, int* n_matches) {
    OMNI_REQUIRE(ctx && q_idx && t_idx && dist && n_matches, OMNI_ERR_INVALID, "null argument");
    OMNI_REQUIRE(nq >= 0 && nt >= 0 && nq <= BF_MAX_N && nt <= BF_MAX_N, OMNI_ERR_CAPACITY, "nq=%d nt=%d outside [0,%d]", nq, nt, BF_MAX_N);
    *n_matches = 0;
    if (nq == 0 || nt == 0) return OMNI_OK;   // BFMatcher on an empty set returns no matches
    OMNI_REQUIRE(q_host && t_host, OMNI_ERR_INVALID, "null descriptors");
    std::lock_guard<std::mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    const int max_n = nq > nt ? nq : nt;
    const size_t fq = (size_t)nq * dim * 4, ft = (size_t)nt * dim * 4;
    // device scratch layout: q | t | nq,nt | oq | ot | od | on
    const size_t off_t = (fq + 255) & ~(size_t)255;
    const size_t off_n = off_t + ((ft + 255) & ~(size_t)255);
    const size_t off_oq = off_n + 256;
    const size_t off_ot = off_oq + (((size_t)max_n * 4 + 255) & ~(size_t)255);
    const size_t off_od = off_ot + (((size_t)max_n * 4 + 255) & ~(size_t)255);
    const size_t off_on = off_od + (((size_t)max_n * 4 + 255) & ~(size_t)255);
    const size_t total = off_on + 256;
    int rc;
    if ((rc = ctx->scratch.ensure(total))) return rc;
    if ((rc = ctx->hstage.ensure(total))) return rc;
    char* d = ctx->scratch.as<char>();
    char* h = ctx->hstage.as<char>();
    memcpy(h, q_host, fq);
    memcpy(h + off_t, t_host, ft);
    ((int*)(h + off_n))[0] = nq;
    ((int*)(h + off_n))[1] = nt;
    OMNI_HIP_TRY(hipMemcpyAsync(d, h, off_n + 256, hipMemcpyHostToDevice, ctx->stream));
    rc = omni::bf_launch(ctx, 1, max_n, dim, mode, (const float*)d, 0, (const int*)(d + off_n), (const float*)(d + off_t), 0,
                         (const int*)(d + off_n) + 1, (int*)(d + off_oq), (int*)(d + off_ot), (float*)(d + off_od),
                         (int*)(d + off_on));
    if (rc) return rc;
    OMNI_HIP_TRY(hipMemcpyAsync(h + off_oq, d + off_oq, total - off_oq, hipMemcpyDeviceToHost, ctx->stream));
    OMNI_HIP_TRY(hipStreamSynchronize(ctx->stream));
    const int n = *(int*)(h + off_on);
    memcpy(q_idx, h + off_oq, (size_t)n * 4);
    memcpy(t_idx, h + off_ot, (size_t)n * 4);
    memcpy(dist, h + off_od, (size_t)n * 4);
    *n_matches = n;
    return OMNI_OK;
}

int omni_bf_match_multi(omni_ctx* ctx, int n_pairs, const float* const* q_host, const int* nq, const float* const* t_host, const int* nt, int dim,
                        int mode, int max_n, int* q_idx, int* t_idx, float* dist, int* n_matches) {
    OMNI_REQUIRE(ctx && q_host && t_host && nq && nt && q_idx && t_idx && dist && n_matches, OMNI_ERR_INVALID, "null argument");
    OMNI_REQUIRE(n_pairs >= 1 && n_pairs <= 64, OMNI_ERR_CAPACITY, "n_pairs=%d outside [1,64]", n_pairs);
    OMNI_REQUIRE(max_n >= 1 && max_n <= BF_MAX_N && dim >= 4 && dim <= BF_MAX_DIM, OMNI_ERR_CAPACITY, "max_n=%d dim=%d", max_n, dim);
    for (int p = 0; p < n_pairs; ++p) {
        OMNI_REQUIRE(nq[p] >= 0 && nt[p] >= 0 && nq[p] <= max_n && nt[p] <= max_n, OMNI_ERR_CAPACITY, "pair %d: nq=%d nt=%d outside [0,%d]", p, nq[p], nt[p], max_n);
        OMNI_REQUIRE((nq[p] == 0 || q_host[p]) && (nt[p] == 0 || t_host[p]), OMNI_ERR_INVALID, "pair %d: null descriptors", p);
        n_matches[p] = 0;
    }
    std::lock_guard<std::mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    // device scratch layout: q [P][max_n][dim] | t [P][max_n][dim] | nq [P] nt [P] | oq | ot | od [P][max_n] | on [P]
    const size_t slab = (size_t)max_n * dim * 4, fq = (((size_t)n_pairs * slab) + 255) & ~(size_t)255;
    const size_t off_t = fq, off_n = off_t + fq, off_oq = off_n + 512;
    const size_t fo = (((size_t)n_pairs * max_n * 4) + 255) & ~(size_t)255;
    const size_t off_ot = off_oq + fo, off_od = off_ot + fo, off_on = off_od + fo, total = off_on + 256;
    int rc;
    if ((rc = ctx->scratch.ensure(total))) return rc;
    if ((rc = ctx->hstage.ensure(total))) return rc;
    char* d = ctx->scratch.as<char>();
    char* h = ctx->hstage.as<char>();
    for (int p = 0; p < n_pairs; ++p) {
        if (nq[p]) memcpy(h + (size_t)p * slab, q_host[p], (size_t)nq[p] * dim * 4);
        if (nt[p]) memcpy(h + off_t + (size_t)p * slab, t_host[p], (size_t)nt[p] * dim * 4);
        ((int*)(h + off_n))[p] = nq[p];
        ((int*)(h + off_n))[64 + p] = nt[p];
    }
    OMNI_HIP_TRY(hipMemcpyAsync(d, h, off_n + 512, hipMemcpyHostToDevice, ctx->stream));
    rc = omni::bf_launch(ctx, n_pairs, max_n, dim, mode, (const float*)d, (int64_t)max_n * dim, (const int*)(d + off_n), (const float*)(d + off_t),
                         (int64_t)max_n * dim, (const int*)(d + off_n) + 64, (int*)(d + off_oq), (int*)(d + off_ot), (float*)(d + off_od), (int*)(d + off_on));
    if (rc) return rc;
    OMNI_HIP_TRY(hipMemcpyAsync(h + off_oq, d + off_oq, total - off_oq, hipMemcpyDeviceToHost, ctx->stream));
    OMNI_HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (int p = 0; p < n_pairs; ++p) {
        const int n = (nq[p] && nt[p]) ? ((int*)(h + off_on))[p] : 0;     // BFMatcher on an empty set returns no matches
        memcpy(q_idx + (size_t)p * max_n, h + off_oq + (size_t)p * max_n * 4, (size_t)n * 4);
        memcpy(t_idx + (size_t)p * max_n, h + off_ot + (size_t)p * max_n * 4, (size_t)n * 4);
        memcpy(dist + (size_t)p * max_n, h + off_od + (size_t)p * max_n * 4, (size_t)n * 4);
        n_matches[p] = n;
    }
    return OMNI_OK;
}

}  // extern "C"
