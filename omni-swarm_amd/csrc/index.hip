// Global-descriptor index: drop-in for faiss::IndexFlatIP as LoopDetector uses it
//   add    : swarm_loop/src/loop_detector.cpp:166,169   (IndexFlatIP::add(1, x))
//   search : swarm_loop/src/loop_detector.cpp:213       (IndexFlatIP::search(1, q, k, D, I))
//   ntotal : swarm_loop/src/loop_detector.cpp:167,170,232,291
//
// HBM layout: one row-major [capacity][dim] matrix (fp32 or fp16), rows appended in insertion order
// (the recency rule `label <= ntotal - max_index`, loop_detector.cpp:232, needs insertion order).
// Search = one streaming scan kernel (HBM-bound: dim*sizeof(elem) bytes per row, read exactly once)
// that writes one 64-bit sortable key per (query,row), then the exact hierarchical top-k of topk.h.
#include "common.h"
#include "topk.h"

struct omni_index {
    omni_ctx* ctx = nullptr;
    int dim = 0;
    int storage = OMNI_STORE_F32;
    int64_t capacity = 0;
    int64_t ntotal = 0;
    int rank = 0, world = 1;
    void* db = nullptr;
    omni::DevBuf qbuf, keys_a, keys_b, out_d, out_i, stage;
    omni::HostBuf hq, hout;
    hipEvent_t scan0 = nullptr, scan1 = nullptr;
    bool scan_timed = false;
    std::mutex mu;
    size_t elem() const { return storage == OMNI_STORE_F16 ? 2 : 4; }
};

namespace omni {

#ifdef OMNI_SCAN_NO_NT
#define NT_LOAD(p) (*(p))
#else
#define NT_LOAD(p) __builtin_nontemporal_load(p)
#endif
#define SCAN_THREADS 256
#define SCAN_WAVES (SCAN_THREADS / 64)
#define SCAN_MAX_QB 8

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// One wave per row at a time, QB queries resident in LDS.  Each lane streams 16 B per load (1 KiB per wave
// instruction, fully coalesced).  keys[q][row] = make_key(dot(q, row), row).
template <typename T, int QB>
__global__ void __launch_bounds__(SCAN_THREADS)
ip_scan_kernel(const T* __restrict__ db, int64_t n_rows, int dim, const float* __restrict__ queries,
               uint64_t* __restrict__ keys, int64_t key_stride) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* sq = reinterpret_cast<float*>(smem_raw);             // [QB][dim]
    for (int i = threadIdx.x * 4; i < QB * dim; i += SCAN_THREADS * 4)
        *reinterpret_cast<float4*>(sq + i) = *reinterpret_cast<const float4*>(queries + i);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    constexpr int EPL = 16 / sizeof(T);                          // elements per lane per load (4 or 8)
    const int64_t wave_global = (int64_t)blockIdx.x * SCAN_WAVES + wave;
    const int64_t wave_stride = (int64_t)gridDim.x * SCAN_WAVES;
    for (int64_t row = wave_global; row < n_rows; row += wave_stride) {
        const T* rp = db + row * dim;
        float acc[QB];
#pragma unroll
        for (int q = 0; q < QB; ++q) acc[q] = 0.f;
#pragma unroll 4
        for (int c = lane * EPL; c < dim; c += 64 * EPL) {
            float v[EPL];
            // every DB byte is read exactly once per scan: non-temporal loads keep the stream out of the way of the LDS-resident
            // queries' neighbours in L2 (MI355X guide: streamed-once operands measure 6.5-6.8 vs 6.4 TB/s with nt)
            typedef float f32x4_t __attribute__((ext_vector_type(4)));
            typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
            if constexpr (sizeof(T) == 4) {
                const f32x4_t x = NT_LOAD(reinterpret_cast<const f32x4_t*>(rp + c));
                v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = x[3];
            } else {
                const u32x4_t xx = NT_LOAD(reinterpret_cast<const u32x4_t*>(rp + c));
                const uint4 x = make_uint4(xx[0], xx[1], xx[2], xx[3]);
                const __half2* h = reinterpret_cast<const __half2*>(&x);
#pragma unroll
                for (int e = 0; e < 4; ++e) { float2 f = __half22float2(h[e]); v[2 * e] = f.x; v[2 * e + 1] = f.y; }
            }
#pragma unroll
            for (int q = 0; q < QB; ++q) {
                const float* qp = sq + q * dim + c;
#pragma unroll
                for (int e = 0; e < EPL; e += 4) {
                    const float4 qq = *reinterpret_cast<const float4*>(qp + e);
                    acc[q] = fmaf(v[e], qq.x, acc[q]);
                    acc[q] = fmaf(v[e + 1], qq.y, acc[q]);
                    acc[q] = fmaf(v[e + 2], qq.z, acc[q]);
                    acc[q] = fmaf(v[e + 3], qq.w, acc[q]);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < QB; ++q) {
            float s = wave_sum(acc[q]);
            if (lane == 0) keys[(int64_t)q * key_stride + row] = omni_make_key(s, (uint32_t)row);
        }
    }
}

__global__ void decode_topk_kernel(const uint64_t* __restrict__ keys, int nq, int k, int rank, int world,
                                   float* __restrict__ D, int64_t* __restrict__ I) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq * k) return;
    uint64_t key = keys[i];
    uint32_t inv = (uint32_t)(key & 0xFFFFFFFFull);
    if (key == OMNI_KEY_EMPTY) { D[i] = -3.402823466e+38f; I[i] = -1; return; }
    uint32_t local = 0xFFFFFFFFu - inv;
    D[i] = omni_orderable_f32((uint32_t)(key >> 32));
    I[i] = (int64_t)local * world + rank;
}

__global__ void f32_to_f16_kernel(const float* __restrict__ in, __half* __restrict__ out, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = __float2half_rn(in[i]);
}

template <typename T>
static int launch_scan(hipStream_t st, const omni_index* ix, int qb, const float* q_dev, uint64_t* keys, int64_t key_stride) {
    const int64_t n = ix->ntotal;
    int cus = ix->ctx->prop.multiProcessorCount > 0 ? ix->ctx->prop.multiProcessorCount : 256;
    int64_t want = cdiv64(n, SCAN_WAVES);
    int grid = (int)(want < (int64_t)cus * 8 ? want : (int64_t)cus * 8);
    if (grid < 1) grid = 1;
    size_t smem = (size_t)qb * ix->dim * sizeof(float);
    const T* db = reinterpret_cast<const T*>(ix->db);
#define OMNI_SCAN_CASE(QB)                                                                                   \
    case QB:                                                                                                 \
        hipLaunchKernelGGL((ip_scan_kernel<T, QB>), dim3(grid), dim3(SCAN_THREADS), smem, st, db, n, ix->dim, \
                           q_dev, keys, key_stride);                                                         \
        break;
    switch (qb) {
        OMNI_SCAN_CASE(1) OMNI_SCAN_CASE(2) OMNI_SCAN_CASE(3) OMNI_SCAN_CASE(4)
        OMNI_SCAN_CASE(5) OMNI_SCAN_CASE(6) OMNI_SCAN_CASE(7) OMNI_SCAN_CASE(8)
        default: set_error("internal: bad query block %d", qb); return OMNI_ERR_INVALID;
    }
#undef OMNI_SCAN_CASE
    OMNI_LAUNCH_CHECK();
    return OMNI_OK;
}

static int ensure_capacity(omni_index* ix, int64_t rows) {
    if (rows <= ix->capacity) return OMNI_OK;
    int64_t cap = ix->capacity > 0 ? ix->capacity : 1024;
    while (cap < rows) cap *= 2;
    void* nd = nullptr;
    OMNI_HIP_TRY(hipMalloc(&nd, (size_t)cap * ix->dim * ix->elem()));
    if (ix->db && ix->ntotal > 0)
        OMNI_HIP_TRY(hipMemcpyAsync(nd, ix->db, (size_t)ix->ntotal * ix->dim * ix->elem(), hipMemcpyDeviceToDevice,
                                    ix->ctx->stream));
    OMNI_HIP_TRY(hipStreamSynchronize(ix->ctx->stream));
    if (ix->db) (void)hipFree(ix->db);
    ix->db = nd;
    ix->capacity = cap;
    return OMNI_OK;
}

// x_dev: n x dim fp32 in HBM -> appended (converted when storage is fp16)
static int append_dev(omni_index* ix, int64_t n, const float* x_dev) {
    int rc = ensure_capacity(ix, ix->ntotal + n);
    if (rc) return rc;
    hipStream_t st = ix->ctx->stream;
    const int64_t cnt = n * ix->dim;
    if (ix->storage == OMNI_STORE_F32) {
        OMNI_HIP_TRY(hipMemcpyAsync((float*)ix->db + ix->ntotal * ix->dim, x_dev, (size_t)cnt * 4,
                                    hipMemcpyDeviceToDevice, st));
    } else {
        hipLaunchKernelGGL(f32_to_f16_kernel, dim3((unsigned)cdiv64(cnt, 256)), dim3(256), 0, st, x_dev,
                           (__half*)ix->db + ix->ntotal * ix->dim, cnt);
        OMNI_LAUNCH_CHECK();
    }
    ix->ntotal += n;
    return OMNI_OK;
}

static int search_dev(omni_index* ix, int nq, const float* q_dev, int k, float* D_dev, int64_t* I_dev) {
    hipStream_t st = ix->ctx->stream;
    const int64_t n = ix->ntotal;
    const int64_t chunks = cdiv64(n > 0 ? n : 1, TOPK_CHUNK);
    const int64_t per_q = (n > chunks * k ? n : chunks * k);
    int rc;
    if ((rc = ix->keys_a.ensure((size_t)nq * per_q * 8))) return rc;
    if ((rc = ix->keys_b.ensure((size_t)nq * per_q * 8))) return rc;
    uint64_t* ka = ix->keys_a.as<uint64_t>();
    uint64_t* kb = ix->keys_b.as<uint64_t>();
    if (n > 0) {
        OMNI_HIP_TRY(hipEventRecord(ix->scan0, st));
        for (int q0 = 0; q0 < nq; q0 += SCAN_MAX_QB) {
            int qb = nq - q0 < SCAN_MAX_QB ? nq - q0 : SCAN_MAX_QB;
            if (ix->storage == OMNI_STORE_F32)
                rc = launch_scan<float>(st, ix, qb, q_dev + (int64_t)q0 * ix->dim, ka + (int64_t)q0 * n, n);
            else
                rc = launch_scan<__half>(st, ix, qb, q_dev + (int64_t)q0 * ix->dim, ka + (int64_t)q0 * n, n);
            if (rc) return rc;
        }
        OMNI_HIP_TRY(hipEventRecord(ix->scan1, st));
        ix->scan_timed = true;
    }
    uint64_t* res = nullptr;
    if ((rc = topk_keys(st, ka, kb, nq, n, k, &res))) return rc;
    hipLaunchKernelGGL(decode_topk_kernel, dim3(cdiv(nq * k, 256)), dim3(256), 0, st, res, nq, k, ix->rank, ix->world,
                       D_dev, I_dev);
    OMNI_LAUNCH_CHECK();
    return OMNI_OK;
}

}  // namespace omni

extern "C" {

omni_index* omni_index_create(omni_ctx* ctx, int dim, int storage, int64_t initial_capacity_rows) {
    if (!ctx) { omni::set_error("null ctx"); return nullptr; }
    if (dim <= 0 || dim % 512 != 0 || dim > 8192) {
        omni::set_error("dim=%d unsupported: must be a positive multiple of 512, <= 8192 (reference: 4096)", dim);
        return nullptr;
    }
    if (storage != OMNI_STORE_F32 && storage != OMNI_STORE_F16) { omni::set_error("bad storage %d", storage); return nullptr; }
    (void)hipSetDevice(ctx->device);
    omni_index* ix = new omni_index();
    ix->ctx = ctx; ix->dim = dim; ix->storage = storage;
    if (hipEventCreate(&ix->scan0) != hipSuccess || hipEventCreate(&ix->scan1) != hipSuccess) {
        omni::set_error("hipEventCreate failed"); delete ix; return nullptr;
    }
    if (initial_capacity_rows > 0 && omni::ensure_capacity(ix, initial_capacity_rows) != OMNI_OK) { delete ix; return nullptr; }
    return ix;
}

void omni_index_destroy(omni_index* ix) {
    if (!ix) return;
    (void)hipSetDevice(ix->ctx->device);
    (void)hipStreamSynchronize(ix->ctx->stream);
    if (ix->db) (void)hipFree(ix->db);
    ix->qbuf.release(); ix->keys_a.release(); ix->keys_b.release(); ix->out_d.release(); ix->out_i.release();
    ix->stage.release(); ix->hq.release(); ix->hout.release();
    if (ix->scan0) (void)hipEventDestroy(ix->scan0);
    if (ix->scan1) (void)hipEventDestroy(ix->scan1);
    delete ix;
}

int64_t omni_index_ntotal(const omni_index* ix) { return ix ? ix->ntotal : -1; }

int omni_index_reset(omni_index* ix) {
    OMNI_REQUIRE(ix, OMNI_ERR_INVALID, "null index");
    std::lock_guard<std::mutex> lk(ix->mu);
    ix->ntotal = 0;
    return OMNI_OK;
}

int omni_index_set_shard(omni_index* ix, int rank, int world) {
    OMNI_REQUIRE(ix && world >= 1 && rank >= 0 && rank < world, OMNI_ERR_INVALID, "bad shard %d/%d", rank, world);
    std::lock_guard<std::mutex> lk(ix->mu);
    ix->rank = rank; ix->world = world;
    return OMNI_OK;
}

int omni_index_add_dev(omni_index* ix, int64_t n, const float* x_dev) {
    OMNI_REQUIRE(ix && x_dev && n >= 0, OMNI_ERR_INVALID, "bad argument");
    std::lock_guard<std::mutex> lk(ix->mu);
    (void)hipSetDevice(ix->ctx->device);
    if (n == 0) return OMNI_OK;
    return omni::append_dev(ix, n, x_dev);
}

int omni_index_add(omni_index* ix, int64_t n, const float* x_host) {
    OMNI_REQUIRE(ix && x_host && n >= 0, OMNI_ERR_INVALID, "bad argument");
    std::lock_guard<std::mutex> lk(ix->mu);
    (void)hipSetDevice(ix->ctx->device);
    if (n == 0) return OMNI_OK;
    // stream in slabs of <= 4096 rows through a device staging buffer (fp16 storage converts on device)
    const int64_t slab = 4096;
    int rc;
    for (int64_t s = 0; s < n; s += slab) {
        int64_t m = n - s < slab ? n - s : slab;
        size_t bytes = (size_t)m * ix->dim * 4;
        if (ix->storage == OMNI_STORE_F32) {
            if ((rc = omni::ensure_capacity(ix, ix->ntotal + m))) return rc;
            OMNI_HIP_TRY(hipMemcpyAsync((float*)ix->db + ix->ntotal * ix->dim, x_host + s * ix->dim, bytes,
                                        hipMemcpyHostToDevice, ix->ctx->stream));
            ix->ntotal += m;
        } else {
            if ((rc = ix->stage.ensure(bytes))) return rc;
            OMNI_HIP_TRY(hipMemcpyAsync(ix->stage.p, x_host + s * ix->dim, bytes, hipMemcpyHostToDevice, ix->ctx->stream));
            if ((rc = omni::append_dev(ix, m, ix->stage.as<float>()))) return rc;
        }
        OMNI_HIP_TRY(hipStreamSynchronize(ix->ctx->stream));   // x_host may be pageable / reused by the caller
    }
    return OMNI_OK;
}

int omni_index_search_dev(omni_index* ix, int nq, const float* q_dev, int k, float* D_dev, int64_t* I_dev) {
    OMNI_REQUIRE(ix && q_dev && D_dev && I_dev, OMNI_ERR_INVALID, "null argument");
    OMNI_REQUIRE(nq >= 1 && nq <= 4096, OMNI_ERR_CAPACITY, "nq=%d outside [1,4096]", nq);
    OMNI_REQUIRE(k >= 1 && k <= TOPK_MAX_K, OMNI_ERR_CAPACITY, "k=%d outside [1,%d]", k, TOPK_MAX_K);
    std::lock_guard<std::mutex> lk(ix->mu);
    (void)hipSetDevice(ix->ctx->device);
    return omni::search_dev(ix, nq, q_dev, k, D_dev, I_dev);
}

int omni_index_search(omni_index* ix, int nq, const float* q_host, int k, float* D, int64_t* I) {
    OMNI_REQUIRE(ix && q_host && D && I, OMNI_ERR_INVALID, "null argument");
    OMNI_REQUIRE(nq >= 1 && nq <= 4096, OMNI_ERR_CAPACITY, "nq=%d outside [1,4096]", nq);
    OMNI_REQUIRE(k >= 1 && k <= TOPK_MAX_K, OMNI_ERR_CAPACITY, "k=%d outside [1,%d]", k, TOPK_MAX_K);
    std::lock_guard<std::mutex> lk(ix->mu);
    if (ix->ntotal == 0) {      // faiss pads an empty index's result with -1 labels; nothing to launch
        for (size_t i = 0; i < (size_t)nq * k; ++i) { D[i] = -3.402823466e+38f; I[i] = -1; }
        return OMNI_OK;
    }
    (void)hipSetDevice(ix->ctx->device);
    hipStream_t st = ix->ctx->stream;
    int rc;
    const size_t qbytes = (size_t)nq * ix->dim * 4;
    if ((rc = ix->qbuf.ensure(qbytes))) return rc;
    if ((rc = ix->hq.ensure(qbytes))) return rc;
    if ((rc = ix->out_d.ensure((size_t)nq * k * 4))) return rc;
    if ((rc = ix->out_i.ensure((size_t)nq * k * 8))) return rc;
    if ((rc = ix->hout.ensure((size_t)nq * k * 12))) return rc;
    memcpy(ix->hq.p, q_host, qbytes);
    OMNI_HIP_TRY(hipMemcpyAsync(ix->qbuf.p, ix->hq.p, qbytes, hipMemcpyHostToDevice, st));
    if ((rc = omni::search_dev(ix, nq, ix->qbuf.as<float>(), k, ix->out_d.as<float>(), ix->out_i.as<int64_t>()))) return rc;
    char* ho = ix->hout.as<char>();
    OMNI_HIP_TRY(hipMemcpyAsync(ho, ix->out_i.p, (size_t)nq * k * 8, hipMemcpyDeviceToHost, st));
    OMNI_HIP_TRY(hipMemcpyAsync(ho + (size_t)nq * k * 8, ix->out_d.p, (size_t)nq * k * 4, hipMemcpyDeviceToHost, st));
    OMNI_HIP_TRY(hipStreamSynchronize(st));
    memcpy(I, ho, (size_t)nq * k * 8);
    memcpy(D, ho + (size_t)nq * k * 8, (size_t)nq * k * 4);
    return OMNI_OK;
}

namespace {
struct OmnxHeader { char magic[8]; int32_t dim, storage; int64_t ntotal; };
}

int omni_index_save(omni_index* ix, const char* path) {
    OMNI_REQUIRE(ix && path, OMNI_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(ix->mu);
    (void)hipSetDevice(ix->ctx->device);
    FILE* f = fopen(path, "wb");
    OMNI_REQUIRE(f, OMNI_ERR_INVALID, "cannot open %s for writing", path);
    OmnxHeader h{};
    memcpy(h.magic, "OMNX1\0\0", 8); h.dim = ix->dim; h.storage = ix->storage; h.ntotal = ix->ntotal;
    bool ok = fwrite(&h, sizeof(h), 1, f) == 1;
    const size_t row = (size_t)ix->dim * ix->elem(), slab_rows = 4096;
    int rc = ix->hout.ensure(slab_rows * row);
    for (int64_t s = 0; ok && rc == OMNI_OK && s < ix->ntotal; s += slab_rows) {
        const size_t m = (size_t)(ix->ntotal - s < (int64_t)slab_rows ? ix->ntotal - s : slab_rows);
        if (hipMemcpyAsync(ix->hout.p, (const char*)ix->db + (size_t)s * row, m * row, hipMemcpyDeviceToHost, ix->ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ix->ctx->stream) != hipSuccess) { omni::set_error("device read failed while saving"); rc = OMNI_ERR_HIP; break; }
        ok = fwrite(ix->hout.p, row, m, f) == m;
    }
    ok = (fclose(f) == 0) && ok;
    if (rc) return rc;
    OMNI_REQUIRE(ok, OMNI_ERR_INVALID, "short write to %s", path);
    return OMNI_OK;
}

int omni_index_load(omni_index* ix, const char* path) {
    OMNI_REQUIRE(ix && path, OMNI_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(ix->mu);
    (void)hipSetDevice(ix->ctx->device);
    FILE* f = fopen(path, "rb");
    OMNI_REQUIRE(f, OMNI_ERR_INVALID, "cannot open %s", path);
    OmnxHeader h{};
    int rc = OMNI_OK;
    if (fread(&h, sizeof(h), 1, f) != 1 || memcmp(h.magic, "OMNX1\0\0", 8) != 0) { omni::set_error("%s is not an OMNX1 snapshot", path); rc = OMNI_ERR_INVALID; }
    else if (h.dim != ix->dim || h.storage != ix->storage || h.ntotal < 0) {
        omni::set_error("snapshot %s is dim=%d storage=%d, the handle is dim=%d storage=%d", path, h.dim, h.storage, ix->dim, ix->storage);
        rc = OMNI_ERR_INVALID;
    }
    const size_t row = (size_t)ix->dim * ix->elem(), slab_rows = 4096;
    if (!rc) rc = omni::ensure_capacity(ix, h.ntotal > 0 ? h.ntotal : 1);
    if (!rc) rc = ix->hout.ensure(slab_rows * row);
    for (int64_t s = 0; !rc && s < h.ntotal; s += slab_rows) {
        const size_t m = (size_t)(h.ntotal - s < (int64_t)slab_rows ? h.ntotal - s : slab_rows);
        if (fread(ix->hout.p, row, m, f) != m) { omni::set_error("%s is truncated", path); rc = OMNI_ERR_INVALID; break; }
        if (hipMemcpyAsync((char*)ix->db + (size_t)s * row, ix->hout.p, m * row, hipMemcpyHostToDevice, ix->ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ix->ctx->stream) != hipSuccess) { omni::set_error("device write failed while loading"); rc = OMNI_ERR_HIP; }
    }
    fclose(f);
    if (!rc) ix->ntotal = h.ntotal;
    return rc;
}

int omni_index_last_scan_ms(omni_index* ix, float* ms) {
    OMNI_REQUIRE(ix && ms, OMNI_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(ix->mu);
    OMNI_REQUIRE(ix->scan_timed, OMNI_ERR_INVALID, "no timed scan yet");
    OMNI_HIP_TRY(hipEventSynchronize(ix->scan1));
    OMNI_HIP_TRY(hipEventElapsedTime(ms, ix->scan0, ix->scan1));
    return OMNI_OK;
}

int omni_topk_merge(int n_lists, int nq, int k_each, const float* D_lists, const int64_t* I_lists, int k_out,
                    float* D, int64_t* I) {
    OMNI_REQUIRE(n_lists >= 1 && nq >= 1 && k_each >= 1 && k_out >= 1 && D_lists && I_lists && D && I, OMNI_ERR_INVALID,
                 "bad argument");
    // Host-side P*k-way merge (P <= 8 shards, k <= 1024): simple insertion into a k_out list per query.
    for (int q = 0; q < nq; ++q) {
        float* Dq = D + (size_t)q * k_out;
        int64_t* Iq = I + (size_t)q * k_out;
        for (int j = 0; j < k_out; ++j) { Dq[j] = -3.402823466e+38f; Iq[j] = -1; }
        for (int l = 0; l < n_lists; ++l) {
            const float* Dl = D_lists + ((size_t)l * nq + q) * k_each;
            const int64_t* Il = I_lists + ((size_t)l * nq + q) * k_each;
            for (int j = 0; j < k_each; ++j) {
                if (Il[j] < 0) continue;
                const float s = Dl[j];
                const int64_t id = Il[j];
                int pos = k_out;
                // position of first entry that (s,id) precedes: score desc, id asc
                while (pos > 0 && (Iq[pos - 1] < 0 || s > Dq[pos - 1] || (s == Dq[pos - 1] && id < Iq[pos - 1]))) --pos;
                if (pos >= k_out) continue;
                for (int m = k_out - 1; m > pos; --m) { Dq[m] = Dq[m - 1]; Iq[m] = Iq[m - 1]; }
                Dq[pos] = s; Iq[pos] = id;
            }
        }
    }
    return OMNI_OK;
}

}  // extern "C"
