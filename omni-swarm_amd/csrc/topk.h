// Exact top-k over 64-bit sortable keys (common.h: score-major, id-minor) -- shared by the
// global-descriptor search (faiss::IndexFlatIP::search, loop_detector.cpp:213) and the key-point
// selection of NMS2 (std::sort by confidence + keep max_num, superpoint_tensorrt.cpp:304-308).
//
// Hierarchical: every workgroup bitonic-sorts one chunk of TOPK_CHUNK keys in LDS (descending) and keeps the
// first k; levels repeat until one chunk is left.  Exact for any n, k <= TOPK_MAX_K.
#pragma once
#include "common.h"

#define TOPK_CHUNK 2048
#define TOPK_THREADS 1024
#define TOPK_MAX_K 1024

namespace omni {

// In-LDS bitonic sort, descending, n_pow2 keys, all `nthreads` threads of the block participate.
__device__ __forceinline__ void bitonic_sort_desc(uint64_t* keys, int n_pow2, int tid, int nthreads) {
    for (int k = 2; k <= n_pow2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < n_pow2; i += nthreads) {
                int ixj = i ^ j;
                if (ixj > i) {
                    uint64_t a = keys[i], b = keys[ixj];
                    bool desc_block = ((i & k) == 0);
                    if (desc_block ? (a < b) : (a > b)) { keys[i] = b; keys[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }
}

// keys_in : [nq][n_in]   (row stride in_stride)
// keys_out: [nq][n_chunks][k]  (row stride n_chunks*k)
static __global__ void __launch_bounds__(TOPK_THREADS)
topk_chunk_kernel(const uint64_t* __restrict__ keys_in, int64_t n_in, int64_t in_stride,
                  uint64_t* __restrict__ keys_out, int k) {
    __shared__ uint64_t s[TOPK_CHUNK];
    const int q = blockIdx.y;
    const int64_t base = (int64_t)blockIdx.x * TOPK_CHUNK;
    const uint64_t* in = keys_in + (int64_t)q * in_stride;
    for (int i = threadIdx.x; i < TOPK_CHUNK; i += TOPK_THREADS) {
        int64_t g = base + i;
        s[i] = (g < n_in) ? in[g] : OMNI_KEY_EMPTY;
    }
    __syncthreads();
    bitonic_sort_desc(s, TOPK_CHUNK, threadIdx.x, TOPK_THREADS);
    uint64_t* out = keys_out + ((int64_t)q * gridDim.x + blockIdx.x) * k;
    for (int i = threadIdx.x; i < k; i += TOPK_THREADS) out[i] = s[i];
}

// Small-k variant (k <= TOPK_SEL_MAX_K): instead of sorting the whole chunk, radix-select the k-th largest score (4 passes of
// 8 bits over the keys' high words, 256-bin LDS histograms), gather the keys at or above it (k plus ties) and rank those by
// counting.  Exact: ties at the cut-off are resolved on the full 64-bit key (lower id first); if more than TOPK_SEL_CAP keys tie
// the chunk falls back to the bitonic sort in place.  ~4x less work than the full sort for the k = 5 + max_index of the reference.
#define TOPK_SEL_MAX_K 64
#define TOPK_SEL_CAP 512
#define TOPK_SEL_THREADS 256
static __global__ void __launch_bounds__(TOPK_SEL_THREADS)
topk_select_kernel(const uint64_t* __restrict__ keys_in, int64_t n_in, int64_t in_stride, uint64_t* __restrict__ keys_out, int k) {
    __shared__ uint64_t s[TOPK_CHUNK];           // gathered keys first, the whole chunk in the fallback
    __shared__ int hist[256];
    __shared__ int sv[4];                        // [0] gathered [1] bucket [2] need [3] valid keys
    const int tid = threadIdx.x;
    const int q = blockIdx.y;
    const int64_t base = (int64_t)blockIdx.x * TOPK_CHUNK;
    const uint64_t* in = keys_in + (int64_t)q * in_stride;
    constexpr int PER = TOPK_CHUNK / TOPK_SEL_THREADS;
    uint64_t key[PER];
    int nvalid = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int64_t g = base + tid + j * TOPK_SEL_THREADS;
        key[j] = (g < n_in) ? in[g] : OMNI_KEY_EMPTY;
        nvalid += key[j] != OMNI_KEY_EMPTY;
    }
    if (tid < 4) sv[tid] = 0;
    __syncthreads();
    if (nvalid) atomicAdd(&sv[3], nvalid);
    __syncthreads();
    const int V = sv[3];
    uint32_t cutoff = 0;
    if (V > k) {
        int need = k;
        for (int pass = 0; pass < 4; ++pass) {
            const int shift = 24 - 8 * pass;
            hist[tid] = 0;
            __syncthreads();
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const uint32_t h = (uint32_t)(key[j] >> 32);
                if (key[j] != OMNI_KEY_EMPTY && (pass == 0 || (h >> (shift + 8)) == (cutoff >> (shift + 8)))) atomicAdd(&hist[(h >> shift) & 255u], 1);
            }
            __syncthreads();
            if (tid < 64) {                                            // one wave: lane l owns the 4 buckets 255-4l .. 252-4l
                int c[4], tot = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) { c[e] = hist[255 - 4 * tid - e]; tot += c[e]; }
                int incl = tot;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off, 64); if (tid >= off) incl += t; }
                int above = incl - tot;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (above < need && need <= above + c[e]) { sv[1] = 255 - 4 * tid - e; sv[2] = need - above; }
                    above += c[e];
                }
            }
            __syncthreads();
            cutoff |= (uint32_t)sv[1] << shift;
            need = sv[2];
            __syncthreads();
        }
    }
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        if (key[j] != OMNI_KEY_EMPTY && (uint32_t)(key[j] >> 32) >= cutoff) {
            const int slot = atomicAdd(&sv[0], 1);
            if (slot < TOPK_SEL_CAP) s[slot] = key[j];
        }
    }
    __syncthreads();
    const int T = sv[0];
    uint64_t* out = keys_out + ((int64_t)q * gridDim.x + blockIdx.x) * k;
    if (T <= TOPK_SEL_CAP) {
        for (int i = tid; i < T; i += TOPK_SEL_THREADS) {
            const uint64_t mine = s[i];
            int rank = 0;
            for (int j = 0; j < T; ++j) rank += (s[j] > mine);
            if (rank < k) out[rank] = mine;
        }
        for (int i = T + tid; i < k; i += TOPK_SEL_THREADS) out[i] = OMNI_KEY_EMPTY;      // fewer valid keys than k
    } else {                                                         // massive ties at the cut-off: sort the chunk
        __syncthreads();
#pragma unroll
        for (int j = 0; j < PER; ++j) s[tid + j * TOPK_SEL_THREADS] = key[j];
        __syncthreads();
        bitonic_sort_desc(s, TOPK_CHUNK, tid, TOPK_SEL_THREADS);
        for (int i = tid; i < k; i += TOPK_SEL_THREADS) out[i] = s[i];
    }
}

// Runs the hierarchy on `stream`.  keys [nq][n] (stride n) in buf_a; buf_a/buf_b are ping-pong scratch, each
// >= nq * max(n, cdiv(n,TOPK_CHUNK)*k) keys.  Returns the device pointer holding the final [nq][k] sorted keys
// (row stride k) in *result.
static inline int topk_keys(hipStream_t stream, uint64_t* buf_a, uint64_t* buf_b, int nq, int64_t n, int k,
                            uint64_t** result) {
    OMNI_REQUIRE(k >= 1 && k <= TOPK_MAX_K, OMNI_ERR_CAPACITY, "k=%d outside [1,%d]", k, TOPK_MAX_K);
    uint64_t* in = buf_a;
    uint64_t* out = buf_b;
    int64_t cur = n, stride = n;
    for (;;) {
        int64_t chunks = cdiv64(cur > 0 ? cur : 1, TOPK_CHUNK);
        dim3 grid((unsigned)chunks, (unsigned)nq);
        if (k <= TOPK_SEL_MAX_K)
            hipLaunchKernelGGL(topk_select_kernel, grid, dim3(TOPK_SEL_THREADS), 0, stream, in, cur, stride, out, k);
        else
            hipLaunchKernelGGL(topk_chunk_kernel, grid, dim3(TOPK_THREADS), 0, stream, in, cur, stride, out, k);
        OMNI_LAUNCH_CHECK();
        cur = chunks * k;
        stride = cur;
        uint64_t* t = in; in = out; out = t;
        if (chunks == 1) break;
    }
    *result = in;
    return OMNI_OK;
}

}  // namespace omni
