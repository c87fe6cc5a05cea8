// TEST INFRASTRUCTURE ONLY -- a stand-in for librccl that lets several ranks share ONE GPU.
//
// RCCL refuses two ranks on one device (ncclCommInitRank: invalid usage), and the boxes the tests run on have one MI355X, so
// csrc/shard.hip's multi-rank exchange (omni_shard_step_batch_dev / omni_shard_search with world > 1) could not run before an 8-GPU
// node ran it.  This library implements the five entry points shard.hip resolves with dlsym (ncclGetUniqueId, ncclCommInitRank,
// ncclCommDestroy, ncclAllGather, ncclGetErrorString) as an all-gather through a memory-mapped file plus hipMemcpy, with the
// semantics shard.hip relies on: stream-ordered, every rank receives the ranks' contributions in rank order.  It is loaded only when
// OMNI_RCCL_LIB points at it (tests/test_gpu_shard_rccl.py); the product never links it.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace {
constexpr size_t kSlotBytes = 8u << 20;            // per rank and collective; the tests move < 1 MB
constexpr size_t kHdrBytes = 4096;
struct Hdr { std::atomic<int> arrived, gen; };
struct Comm {
    char* base = nullptr; size_t bytes = 0; int rank = 0, nranks = 1; char path[120] = {0};
    Hdr* hdr() const { return reinterpret_cast<Hdr*>(base); }
    char* slot(int r) const { return base + kHdrBytes + (size_t)r * kSlotBytes; }
};
bool barrier(Comm* c) {
    Hdr* h = c->hdr();
    const int g = h->gen.load(std::memory_order_acquire);
    if (h->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == c->nranks) {
        h->arrived.store(0, std::memory_order_relaxed);
        h->gen.store(g + 1, std::memory_order_release);
        return true;
    }
    const auto t0 = std::chrono::steady_clock::now();
    while (h->gen.load(std::memory_order_acquire) == g) {
        sched_yield();
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) return false;      // a peer died: fail instead of hanging the box
    }
    return true;
}
size_t type_bytes(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: case ncclBfloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
        default: return 0;
    }
}
}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    memset(id->internal, 0, sizeof(id->internal));
    const char* dir = getenv("TMPDIR");
    snprintf(id->internal, sizeof(id->internal), "%s/omni_stub_rccl_%d_%lld", dir && dir[0] ? dir : "/tmp", (int)getpid(),
             (long long)std::chrono::steady_clock::now().time_since_epoch().count());
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    Comm* c = new Comm();
    c->rank = rank; c->nranks = nranks; c->bytes = kHdrBytes + (size_t)nranks * kSlotBytes;
    strncpy(c->path, id.internal, sizeof(c->path) - 1);
    const int fd = open(c->path, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)c->bytes) != 0) { if (fd >= 0) close(fd); delete c; return ncclSystemError; }
    c->base = static_cast<char*>(mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0));   // a fresh file reads as zeros: counters start at 0
    close(fd);
    if (c->base == MAP_FAILED) { delete c; return ncclSystemError; }
    if (!barrier(c)) { munmap(c->base, c->bytes); delete c; return ncclSystemError; }
    if (rank == 0) unlink(c->path);                 // everybody is attached: the mapping outlives the name, nothing is left behind after a crash
    *comm = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (!c) return ncclSuccess;
    munmap(c->base, c->bytes);
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    const size_t bytes = sendcount * type_bytes(datatype);
    if (!c || !sendbuff || !recvbuff || bytes == 0 || bytes > kSlotBytes) return ncclInvalidArgument;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;                  // everything enqueued before the collective
    if (hipMemcpy(c->slot(c->rank), sendbuff, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    if (!barrier(c)) return ncclSystemError;
    for (int r = 0; r < c->nranks; ++r)
        if (hipMemcpy(static_cast<char*>(recvbuff) + (size_t)r * bytes, c->slot(r), bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    if (!barrier(c)) return ncclSystemError;         // nobody overwrites a slot somebody is still reading
    return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "no error";
        case ncclUnhandledCudaError: return "stub rccl: HIP call failed";
        case ncclSystemError: return "stub rccl: shared file / barrier failed (a peer died?)";
        case ncclInvalidArgument: return "stub rccl: invalid argument (or more than 8 MB per rank)";
        default: return "stub rccl: error";
    }
}

}  // extern "C"
