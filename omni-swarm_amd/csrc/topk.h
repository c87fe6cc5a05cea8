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
