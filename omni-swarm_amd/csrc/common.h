// Internal helpers shared by the HIP translation units of libomni_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

#include "../../include/omni_hip.h"

namespace omni {

void set_error(const char* fmt, ...);

#define OMNI_HIP_TRY(expr)                                                                   \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            ::omni::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return OMNI_ERR_HIP;                                                             \
        }                                                                                    \
    } while (0)

#define OMNI_REQUIRE(cond, code, ...)                                                        \
    do {                                                                                     \
        if (!(cond)) { ::omni::set_error(__VA_ARGS__); return (code); }                      \
    } while (0)

#define OMNI_LAUNCH_CHECK()                                                                  \
    do {                                                                                     \
        hipError_t _e = hipGetLastError();                                                   \
        if (_e != hipSuccess) {                                                              \
            ::omni::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
            return OMNI_ERR_HIP;                                                             \
        }                                                                                    \
    } while (0)

// roctx ranges around the stages of the hot path (ctx.hip; OMNI_ROCTX=1: librocprofiler-sdk-roctx / libroctx64 resolved by dlopen, like RCCL): what the
// reference's per-stage timers print (superpoint_tensorrt.cpp:130-162, loop_cam.cpp:205-207) becomes a `rocprofv3 --marker-trace` timeline.  No-ops when off.
void trace_push(const char* name);
void trace_pop();
struct TraceRange { explicit TraceRange(const char* name) { trace_push(name); } ~TraceRange() { trace_pop(); } TraceRange(const TraceRange&) = delete; };

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of a kernel: remember what was set per (kernel instantiation, device)
// instead of per process, so that a process driving several GPUs through several omni_ctx stays correct.  One static State per call site.
struct DynSmemState { size_t set[16] = {0}; };
static inline hipError_t ensure_dyn_smem(DynSmemState& st, const void* kfn, size_t bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 16 && st.set[dev] >= bytes) return hipSuccess;
    e = hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess && dev >= 0 && dev < 16) st.set[dev] = bytes;
    return e;
}
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Grow-only device scratch buffer.
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int ensure(size_t need) {
        if (need <= bytes) return OMNI_OK;
        // hipFree waits for the WHOLE device to go idle: a buffer whose need creeps up with the database (the key lists of a scan: 8 bytes per row per
        // query) must not be reallocated at every micro-batch -- that made the host wait for the key-frame unit in flight each time it enqueued a search
        // (2.7 ms of a 3.8 ms cycle, round 4).  Grow by half at least: a handful of reallocations over a database's life.
        size_t want = bytes + bytes / 2;
        if (want < need) want = need;
        want = (want + 4095) & ~(size_t)4095;
        if (p) (void)hipFree(p);
        p = nullptr; bytes = 0;
        if (hipMalloc(&p, want) != hipSuccess) {
            p = nullptr;
            (void)hipGetLastError();
            want = need;                                    // no room for the slack: the exact size
            OMNI_HIP_TRY(hipMalloc(&p, want));
        }
        bytes = want;
        return OMNI_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
    template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

// Pinned host staging buffer.
struct HostBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int ensure(size_t need) {
        if (need <= bytes) return OMNI_OK;
        size_t want = bytes + bytes / 2;                    // (as DevBuf: hipHostFree synchronises too)
        if (want < need) want = need;
        if (p) (void)hipHostFree(p);
        p = nullptr; bytes = 0;
        OMNI_HIP_TRY(hipHostMalloc(&p, want, hipHostMallocDefault));
        bytes = want;
        return OMNI_OK;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; bytes = 0; }
    template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

}  // namespace omni

extern "C" hipEvent_t omni_sp_convs_event(omni_sp* s);      // internal (superpoint.hip -> cam.hip): recorded behind a pass's convolution stack

#define OMNI_ZERO_PAGE_BYTES 65536

struct omni_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t ev_order = nullptr;     // omni_ctx_order_after: marks this stream's position for another stream to wait on
    hipDeviceProp_t prop;
    std::mutex mu;
    // 64 KB of zeros in HBM, never written after creation: the LDS-DMA source of convolution halo pixels outside the image.  A lane reads the
    // 16 bytes at (its would-be image offset mod 64 KB), so the requests spread over the L2 channels: with ONE shared line the rate depended
    // on where the allocator happened to put it (7 % of the whole pipeline between two placements)
    void* zero_page = nullptr;
    omni::DevBuf scratch;     // generic per-call scratch (bf match, host-entry staging)
    omni::DevBuf scratch2;
    omni::HostBuf hstage;
    int ensure_zero_page() {
        if (zero_page) return OMNI_OK;
        OMNI_HIP_TRY(hipMalloc(&zero_page, OMNI_ZERO_PAGE_BYTES));
        OMNI_HIP_TRY(hipMemsetAsync(zero_page, 0, OMNI_ZERO_PAGE_BYTES, stream));
        return OMNI_OK;
    }
};

// ---- 64-bit sortable keys -------------------------------------------------------------------------------------
// key = (orderable(score) << 32) | (0xFFFFFFFF - id): descending key order == (score desc, id asc).
__host__ __device__ static inline uint32_t omni_f32_orderable(float f) {
    uint32_t u;
#if defined(__HIP_DEVICE_COMPILE__)
    u = __float_as_uint(f);
#else
    memcpy(&u, &f, 4);
#endif
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ static inline float omni_orderable_f32(uint32_t o) {
    uint32_t u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
    float f;
#if defined(__HIP_DEVICE_COMPILE__)
    f = __uint_as_float(u);
#else
    memcpy(&f, &u, 4);
#endif
    return f;
}
__host__ __device__ static inline uint64_t omni_make_key(float score, uint32_t id) {
    return ((uint64_t)omni_f32_orderable(score) << 32) | (uint64_t)(0xFFFFFFFFu - id);
}
#define OMNI_KEY_EMPTY 0ull   // sorts last; decodes to id 0xFFFFFFFF -> reported as -1
