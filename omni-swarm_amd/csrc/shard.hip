// omni_shard_*: the key-frame database row-sharded over the GPUs of one node, one process per GPU, exchanged with RCCL over xGMI
// (new: the reference is single-GPU; SURVEY.md 8e, BASELINE configs[3]/[4]).
//
// Global row g lives on rank g % world at local slot g / world (omni_index_set_shard): insertion order survives per shard, which the
// recency rule `label <= ntotal - max_index` (swarm_loop/src/loop_detector.cpp:232) needs, evaluated on GLOBAL ids after the merge.
// One exchange unit = F consecutive key-frame steps of every rank (a micro-batch):
//     ncclAllGather(rows)            every rank's F*m new global descriptors, device to device (64 KB per row)
//     local append of the rows this rank owns, in global-id order            (no traffic)
//     ONE pass over the shard for the F*world queries, each restricted to the rows its step would have seen
//                                                                            (omni_index_search_batch_prefix_dev: add-before-query, loop_detector.cpp:89-98)
//     ncclAllGather(per-shard top-k)  k*(8+4) B per query per rank: latency-bound on xGMI
//     D2H of the gathered lists + host merge of THIS rank's queries          (omni_topk_merge: score desc, global id asc)
// Everything runs on the shard's stream (the local index's context stream): no host bounce before the final D2H, no torch.
// RCCL is resolved at run time (dlopen of the librccl next to the HIP runtime this library is linked into the process with): the
// single-GPU paths of libomni_hip.so do not depend on it; omni_shard_* fails with a message when it is missing.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <climits>
#include <string>

#include "common.h"

namespace {

struct RcclApi {
    void* so = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
    std::string err;           // why the library could not be used (dlerror() is consumed by the first read: kept here)
};

RcclApi& rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        // The RCCL copy must sit on the SAME ROCm stack as the HIP runtime this library runs on: a process can hold two (PyTorch wheels
        // bundle their own libamdhip64 / libhsa-runtime64 / librccl next to /opt/rocm's), and an RCCL whose HSA runtime is not the initialised
        // one fails ncclCommInitRank with "no ROCm-capable device is detected".  So: first the librccl in the directory the loaded
        // libamdhip64 came from, then the default search.
        // OMNI_RCCL_LIB (the one string option of config.h's table): an explicit library path first (tests point it at tests/stub_rccl, which lets several ranks share one GPU)
        if (const char* forced = getenv("OMNI_RCCL_LIB")) {
            if (forced[0]) {
                api.so = dlopen(forced, RTLD_NOW | RTLD_GLOBAL);
                if (!api.so) { const char* e = dlerror(); api.err = std::string("OMNI_RCCL_LIB=") + forced + ": " + (e ? e : "dlopen failed"); return; }
            }
        }
        Dl_info hip_info;
        if (!api.so && dladdr(reinterpret_cast<const void*>(&hipGetDeviceCount), &hip_info) && hip_info.dli_fname) {
            std::string dir(hip_info.dli_fname);
            const size_t slash = dir.rfind('/');
            if (slash != std::string::npos) {
                dir.resize(slash + 1);
                for (const char* n : {"librccl.so.1", "librccl.so"}) {
                    api.so = dlopen((dir + n).c_str(), RTLD_NOW | RTLD_GLOBAL);
                    if (api.so) break;
                }
            }
        }
        if (!api.so) {
            const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
            for (const char* n : names) { api.so = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (api.so) break; }
        }
        if (!api.so) { const char* e = dlerror(); api.err = e ? e : "librccl.so.1 not found"; return; }
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(api.so, "ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(api.so, "ncclCommInitRank"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(api.so, "ncclCommDestroy"));
        api.AllGather = reinterpret_cast<decltype(api.AllGather)>(dlsym(api.so, "ncclAllGather"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(api.so, "ncclGetErrorString"));
        api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllGather && api.GetErrorString;
        if (!api.ok) api.err = "ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllGather / ncclGetErrorString missing from the library";
    });
    return api;
}

#define OMNI_RCCL_TRY(expr)                                                                                             \
    do {                                                                                                                \
        ncclResult_t _r = (expr);                                                                                       \
        if (_r != ncclSuccess) { ::omni::set_error("%s failed: %s (%s:%d)", #expr, rccl().GetErrorString(_r), __FILE__, __LINE__); return OMNI_ERR_HIP; } \
    } while (0)

// rows owned by `rank` out of the gathered [world][F][m][dim] block, written in global-id order.  Global order of the block is
// (step f, rank r, row j); with ntotal % world == 0 a row is ours iff (r*m + j) % world == rank: exactly m rows per step.
__global__ void shard_pick_rows_kernel(const float* __restrict__ all, int world, int F, int m, int dim, int rank, float* __restrict__ out) {
    const int o = blockIdx.x;                    // owned row o = f*m + i: the i-th owned row of step f
    const int f = o / m, i = o % m;
    const int e = rank + i * world;              // its position r*m + j inside the step
    const int r = e / m, j = e % m;
    const float4* src = reinterpret_cast<const float4*>(all + (((int64_t)r * F + f) * m + j) * dim);
    float4* dst = reinterpret_cast<float4*>(out + (int64_t)o * dim);
    for (int c = threadIdx.x; c < dim / 4; c += blockDim.x) dst[c] = src[c];
}

}  // namespace

struct omni_shard {
    omni_index* local = nullptr;
    omni_ctx* ctx = nullptr;
    int rank = 0, world = 1, dim = 0;
    int64_t ntotal = 0;                          // global row count
    ncclComm_t comm = nullptr;
    omni::DevBuf all_rows, owned, send, recv, qrows;
    omni::HostBuf hrecv, hq;
    // one exchange may be in flight (omni_shard_step_enqueue -> omni_shard_step_wait): its shape, the local row count to restore on failure, and
    // two events on the shard's stream: e_rows = the caller's row buffer has been gathered (it may be overwritten), e_done = the lists are on the host
    hipEvent_t e_rows = nullptr, e_done = nullptr;
    // timing events around the two collectives of the last exchange (omni_shard_last_exchange_us): t[0]..t[1] = the all-gather of the new rows,
    // t[2]..t[3] = the all-gather of the per-shard top-k lists
    hipEvent_t t[4] = {nullptr, nullptr, nullptr, nullptr};
    bool timed = false;
    bool pending = false;
    int pend_F = 0, pend_m = 0, pend_k = 0;
    int64_t pend_local_before = 0;
    std::mutex mu;
};

extern "C" {

int omni_shard_library_path(char* out, int cap) {
    OMNI_REQUIRE(out && cap > 1, OMNI_ERR_INVALID, "bad argument");
    OMNI_REQUIRE(rccl().ok, OMNI_ERR_HIP, "RCCL (librccl.so.1) could not be loaded: %s", rccl().err.c_str());
    Dl_info info;
    const char* name = (dladdr(reinterpret_cast<const void*>(rccl().AllGather), &info) && info.dli_fname) ? info.dli_fname : "?";
    snprintf(out, (size_t)cap, "%s", name);
    return OMNI_OK;
}

int omni_shard_unique_id(char* id_out) {
    OMNI_REQUIRE(id_out, OMNI_ERR_INVALID, "null argument");
    OMNI_REQUIRE(rccl().ok, OMNI_ERR_HIP, "RCCL (librccl.so.1) could not be loaded: %s", rccl().err.c_str());
    ncclUniqueId id;
    OMNI_RCCL_TRY(rccl().GetUniqueId(&id));
    memcpy(id_out, id.internal, NCCL_UNIQUE_ID_BYTES);
    return OMNI_OK;
}

omni_shard* omni_shard_create(omni_ctx* ctx, omni_index* local, int dim, int rank, int world, const char* unique_id) {
    if (!ctx || !local || !unique_id || world < 1 || rank < 0 || rank >= world) { omni::set_error("bad argument"); return nullptr; }
    if (!rccl().ok) { omni::set_error("RCCL (librccl.so.1) could not be loaded: %s", rccl().err.c_str()); return nullptr; }
    if (omni_index_dim(local) != dim) { omni::set_error("dim=%d but the local index holds %d-d rows", dim, omni_index_dim(local)); return nullptr; }
    if (omni_index_ntotal(local) != 0) { omni::set_error("the local shard must be empty"); return nullptr; }
    if (omni_index_set_shard(local, rank, world) != OMNI_OK) return nullptr;
    (void)hipSetDevice(ctx->device);
    omni_shard* s = new omni_shard();
    s->local = local; s->ctx = ctx; s->rank = rank; s->world = world; s->dim = dim;
    ncclUniqueId id;
    memcpy(id.internal, unique_id, NCCL_UNIQUE_ID_BYTES);
    if (hipEventCreateWithFlags(&s->e_rows, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&s->e_done, hipEventDisableTiming) != hipSuccess) {
        omni::set_error("hipEventCreate failed"); delete s; return nullptr;
    }
    for (hipEvent_t& e : s->t)
        if (hipEventCreate(&e) != hipSuccess) { omni::set_error("hipEventCreate failed"); omni_shard_destroy(s); return nullptr; }
    ncclResult_t r = rccl().CommInitRank(&s->comm, world, id, rank);
    if (r != ncclSuccess) { omni::set_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, world, rccl().GetErrorString(r)); s->comm = nullptr; omni_shard_destroy(s); return nullptr; }
    return s;
}

void omni_shard_destroy(omni_shard* s) {
    if (!s) return;
    (void)hipSetDevice(s->ctx->device);
    (void)hipStreamSynchronize(s->ctx->stream);
    if (s->comm) (void)rccl().CommDestroy(s->comm);
    if (s->e_rows) (void)hipEventDestroy(s->e_rows);
    if (s->e_done) (void)hipEventDestroy(s->e_done);
    for (hipEvent_t e : s->t) if (e) (void)hipEventDestroy(e);
    s->all_rows.release(); s->owned.release(); s->send.release(); s->recv.release(); s->qrows.release(); s->hrecv.release(); s->hq.release();
    delete s;
}

int64_t omni_shard_ntotal(const omni_shard* s) { return s ? s->ntotal : -1; }

int omni_shard_preload_local(omni_shard* s, const float* rows_host, int64_t n_local, int64_t ntotal_global) {
    OMNI_REQUIRE(s && (rows_host || n_local == 0), OMNI_ERR_INVALID, "null argument");
    OMNI_REQUIRE(s->ntotal == 0 && ntotal_global % s->world == 0 && n_local * s->world == ntotal_global, OMNI_ERR_INVALID,
                 "preload needs an empty index and ntotal_global = world * n_local");
    std::lock_guard<std::mutex> lk(s->mu);
    int rc = n_local ? omni_index_add(s->local, n_local, rows_host) : OMNI_OK;
    if (rc) return rc;
    s->ntotal = ntotal_global;
    return OMNI_OK;
}

// gathered top-k lists -> this rank's merged result; lists laid out [shard][n_lists_per_shard][k] as I (i64) then D (f32) per shard
static int merge_mine(const omni_shard* s, const char* h, int per_shard, int k, const int* which, int n_out, float* D, int64_t* I) {
    const size_t shard_bytes = (size_t)per_shard * k * 12;
    std::vector<float> Dl((size_t)s->world * k);
    std::vector<int64_t> Il((size_t)s->world * k);
    for (int o = 0; o < n_out; ++o) {
        for (int sh = 0; sh < s->world; ++sh) {
            const char* base = h + sh * shard_bytes;
            memcpy(&Il[(size_t)sh * k], base + (size_t)which[o] * k * 8, (size_t)k * 8);
            memcpy(&Dl[(size_t)sh * k], base + (size_t)per_shard * k * 8 + (size_t)which[o] * k * 4, (size_t)k * 4);
        }
        int rc = omni_topk_merge(s->world, 1, k, Dl.data(), Il.data(), k, D + (size_t)o * k, I + (size_t)o * k);
        if (rc) return rc;
    }
    return OMNI_OK;
}

// One exchange unit, asynchronous half: everything up to the copy of the gathered lists to the host is ENQUEUED on the shard's stream (the
// context stream of the local index: the detector's own stream, not one of the CNN streams) -- no host synchronisation, so the caller can go on
// enqueuing the next micro-batch's CNN work while the collectives, the scan and the copy run.  omni_shard_rows_consumed() / omni_shard_step_wait()
// are the two points where the host (or, through an event, another stream) meets it again.
int omni_shard_step_enqueue(omni_shard* s, int F, int m, const float* rows_dev, int query_row, int k) {
    omni::TraceRange trace_range("exchange step enqueue (all-gather rows, scan, all-gather top-k)");
    OMNI_REQUIRE(s && rows_dev, OMNI_ERR_INVALID, "null argument");
    OMNI_REQUIRE(F >= 1 && m >= 1 && query_row >= 0 && query_row < m && k >= 1, OMNI_ERR_INVALID, "bad F/m/query_row/k");
    OMNI_REQUIRE((int64_t)F * s->world <= 4096, OMNI_ERR_CAPACITY, "F * world = %d queries per exchange is too many", F * s->world);
    std::lock_guard<std::mutex> lk(s->mu);
    OMNI_REQUIRE(!s->pending, OMNI_ERR_INVALID, "an exchange is already in flight: call omni_shard_step_wait first");
    OMNI_REQUIRE(s->ntotal % s->world == 0, OMNI_ERR_INVALID, "global row count %lld is not a multiple of the world size", (long long)s->ntotal);
    (void)hipSetDevice(s->ctx->device);
    hipStream_t st = s->ctx->stream;
    const int W = s->world, dim = s->dim;
    const size_t mine = (size_t)F * m * dim;                      // floats this rank contributes
    int rc;
    if ((rc = s->all_rows.ensure(mine * W * 4))) return rc;
    if ((rc = s->owned.ensure(mine * 4))) return rc;
    const int nq = F * W;
    const size_t list_bytes = (size_t)nq * k * 12;
    if ((rc = s->send.ensure(list_bytes))) return rc;
    if ((rc = s->recv.ensure(list_bytes * W))) return rc;
    if ((rc = s->hrecv.ensure(list_bytes * W))) return rc;
    // 1. every rank's new rows
    s->timed = false;
    OMNI_HIP_TRY(hipEventRecord(s->t[0], st));
    OMNI_RCCL_TRY(rccl().AllGather(rows_dev, s->all_rows.p, mine, ncclFloat32, s->comm, st));
    OMNI_HIP_TRY(hipEventRecord(s->t[1], st));
    OMNI_HIP_TRY(hipEventRecord(s->e_rows, st));                 // rows_dev is free again from here on (stream order)
    // 2. append the rows this rank owns, in global-id order
    hipLaunchKernelGGL(shard_pick_rows_kernel, dim3(F * m), dim3(256), 0, st, s->all_rows.as<float>(), W, F, m, dim, s->rank, s->owned.as<float>());
    OMNI_LAUNCH_CHECK();
    const int64_t local_before = omni_index_ntotal(s->local);
    if ((rc = omni_index_add_dev(s->local, (int64_t)F * m, s->owned.as<float>()))) return rc;
    // from here on a failure must take the appended rows out again: s->ntotal only moves on success, and a shard that kept the rows would
    // hand every later step wrong per-query limits and wrong global ids
    rc = [&]() -> int {
        int rc;
        // 3. the F*world queries (step f of rank r = row query_row of block [r][f]) in one pass, each over the rows of its turn
        std::vector<int64_t> idx(nq), lim(nq);
        for (int f = 0; f < F; ++f)
            for (int r = 0; r < W; ++r) {
                idx[f * W + r] = ((int64_t)r * F + f) * m + query_row;
                lim[f * W + r] = (s->ntotal + (int64_t)(f + 1) * W * m - s->rank + W - 1) / W;      // local rows with global id < ntotal after step f
            }
        char* sb = s->send.as<char>();
        for (int q0 = 0; q0 < nq; q0 += 64) {
            const int n = nq - q0 < 64 ? nq - q0 : 64;
            if ((rc = omni_index_search_batch_prefix_dev(s->local, n, s->all_rows.as<float>(), idx.data() + q0, k, lim.data() + q0,
                                                         reinterpret_cast<float*>(sb + (size_t)nq * k * 8) + (size_t)q0 * k,
                                                         reinterpret_cast<int64_t*>(sb) + (size_t)q0 * k)))
                return rc;
        }
        // 4. the per-shard lists of every query, to everybody, and on to the host
        OMNI_HIP_TRY(hipEventRecord(s->t[2], st));
        OMNI_RCCL_TRY(rccl().AllGather(s->send.p, s->recv.p, list_bytes, ncclInt8, s->comm, st));
        OMNI_HIP_TRY(hipEventRecord(s->t[3], st));
        OMNI_HIP_TRY(hipMemcpyAsync(s->hrecv.p, s->recv.p, list_bytes * W, hipMemcpyDeviceToHost, st));
        OMNI_HIP_TRY(hipEventRecord(s->e_done, st));
        return OMNI_OK;
    }();
    if (rc) { (void)omni_index_truncate(s->local, local_before); return rc; }
    s->pending = true; s->pend_F = F; s->pend_m = m; s->pend_k = k; s->pend_local_before = local_before;
    return OMNI_OK;
}

// blocks until the row buffer handed to the exchange in flight has been gathered (its first collective): the caller may overwrite it
int omni_shard_rows_consumed(omni_shard* s) {
    OMNI_REQUIRE(s, OMNI_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(s->mu);
    if (!s->pending) return OMNI_OK;
    (void)hipSetDevice(s->ctx->device);
    OMNI_HIP_TRY(hipEventSynchronize(s->e_rows));
    return OMNI_OK;
}

// second half: waits for the exchange in flight (an event, not the stream: later work on the stream is not waited for), merges the lists of
// THIS rank's F queries (D_host, I_host: [F][k]) and moves the global row count
int omni_shard_step_wait(omni_shard* s, float* D_host, int64_t* I_host) {
    omni::TraceRange trace_range("exchange step wait + merge");
    OMNI_REQUIRE(s && D_host && I_host, OMNI_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(s->mu);
    OMNI_REQUIRE(s->pending, OMNI_ERR_INVALID, "no exchange in flight");
    (void)hipSetDevice(s->ctx->device);
    const int F = s->pend_F, W = s->world, k = s->pend_k;
    s->pending = false;
    if (hipEventSynchronize(s->e_done) != hipSuccess) {
        (void)omni_index_truncate(s->local, s->pend_local_before);
        omni::set_error("hipEventSynchronize failed while waiting for the exchange");
        return OMNI_ERR_HIP;
    }
    s->ntotal += (int64_t)F * W * s->pend_m;
    s->timed = true;
    // 5. merge the lists of MY queries (query f*W + rank)
    std::vector<int> which(F);
    for (int f = 0; f < F; ++f) which[f] = f * W + s->rank;
    return merge_mine(s, s->hrecv.as<char>(), F * W, k, which.data(), F, D_host, I_host);
}

// device time of the two collectives of the last exchange that omni_shard_step_wait collected (HIP events on the shard's stream, microseconds): the
// all-gather of every rank's new rows (F * m * dim floats per rank) and the all-gather of the per-shard top-k lists (F * world * k * 12 bytes per rank)
int omni_shard_last_exchange_us(omni_shard* s, float* rows_gather_us, float* topk_gather_us) {
    OMNI_REQUIRE(s && rows_gather_us && topk_gather_us, OMNI_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(s->mu);
    OMNI_REQUIRE(s->timed, OMNI_ERR_INVALID, "no finished exchange to report");
    float a = 0, b = 0;
    OMNI_HIP_TRY(hipEventElapsedTime(&a, s->t[0], s->t[1]));
    OMNI_HIP_TRY(hipEventElapsedTime(&b, s->t[2], s->t[3]));
    *rows_gather_us = a * 1e3f; *topk_gather_us = b * 1e3f;
    return OMNI_OK;
}

int omni_shard_step_batch_dev(omni_shard* s, int F, int m, const float* rows_dev, int query_row, int k, float* D_host, int64_t* I_host) {
    OMNI_REQUIRE(s && rows_dev && D_host && I_host, OMNI_ERR_INVALID, "null argument");
    const int rc = omni_shard_step_enqueue(s, F, m, rows_dev, query_row, k);
    return rc ? rc : omni_shard_step_wait(s, D_host, I_host);
}

int omni_shard_search(omni_shard* s, int nq, const float* q_host, int k, float* D, int64_t* I) {
    OMNI_REQUIRE(s && q_host && D && I && nq >= 1 && nq <= 64 && k >= 1, OMNI_ERR_INVALID, "bad argument (1 <= nq <= 64)");
    std::lock_guard<std::mutex> lk(s->mu);
    OMNI_REQUIRE(!s->pending, OMNI_ERR_INVALID, "an exchange is in flight: call omni_shard_step_wait first");
    (void)hipSetDevice(s->ctx->device);
    hipStream_t st = s->ctx->stream;
    const int W = s->world;
    const size_t list_bytes = (size_t)nq * k * 12, qbytes = (size_t)nq * s->dim * 4;
    int rc;
    if ((rc = s->qrows.ensure(qbytes))) return rc;
    if ((rc = s->hq.ensure(qbytes))) return rc;
    if ((rc = s->send.ensure(list_bytes))) return rc;
    if ((rc = s->recv.ensure(list_bytes * W))) return rc;
    if ((rc = s->hrecv.ensure(list_bytes * W))) return rc;
    memcpy(s->hq.p, q_host, qbytes);
    OMNI_HIP_TRY(hipMemcpyAsync(s->qrows.p, s->hq.p, qbytes, hipMemcpyHostToDevice, st));
    char* sb = s->send.as<char>();
    std::vector<int64_t> lim(nq, INT64_MAX);
    if ((rc = omni_index_search_batch_prefix_dev(s->local, nq, s->qrows.as<float>(), nullptr, k, lim.data(), reinterpret_cast<float*>(sb + (size_t)nq * k * 8),
                                                 reinterpret_cast<int64_t*>(sb))))
        return rc;
    OMNI_RCCL_TRY(rccl().AllGather(s->send.p, s->recv.p, list_bytes, ncclInt8, s->comm, st));
    OMNI_HIP_TRY(hipMemcpyAsync(s->hrecv.p, s->recv.p, list_bytes * W, hipMemcpyDeviceToHost, st));
    OMNI_HIP_TRY(hipStreamSynchronize(st));
    std::vector<int> which(nq);
    for (int q = 0; q < nq; ++q) which[q] = q;
    return merge_mine(s, s->hrecv.as<char>(), nq, k, which.data(), nq, D, I);
}

}  // extern "C"
