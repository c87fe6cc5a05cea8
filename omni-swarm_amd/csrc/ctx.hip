// Context, device memory and timer entry points of the C ABI (include/omni_hip.h).
// Replaces the cudaStreamCreate / cudaMalloc / cudaMallocHost plumbing of the reference's
// TensorRTInferenceGeneric (swarm_loop/src/tensorrt_generic.cpp:14-36,99-120).
#include "config.h"
#include "common.h"

#include <cstdlib>
#include <dlfcn.h>

namespace omni {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace omni

// The key-frame pipeline keeps five streams busy (two units in flight x (SuperPoint, MobileNetVLAD) + the detector's); the HIP runtime multiplexes
// streams onto GPU_MAX_HW_QUEUES hardware queues, 4 by default, and two streams on one queue run strictly one after the other: the detector's search
// of micro-batch k then sits behind the MobileNetVLAD kernels of micro-batch k + 1, the host waits for it, and the GPU idles 15 % of the time
// (rocprofv3 kernel trace of round 4).  The runtime reads the variable when it initialises: ask for 8 queues when this library is loaded,
// unless the process already chose (OMNI_HW_QUEUES=0 leaves the runtime's default).
namespace {
struct HwQueues {
    HwQueues() {
        const int n = omni::config_option_now(omni::CFG_HW_QUEUES);      // this option only: the process-wide table stays unresolved until something uses it
        if (n > 0) { char v[16]; snprintf(v, sizeof(v), "%d", n); setenv("GPU_MAX_HW_QUEUES", v, 0); }
    }
} g_hw_queues;
}  // namespace

namespace omni {
namespace {
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        if (!config_process()[CFG_ROCTX]) return;
        void* so = nullptr;
        for (const char* n : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) { so = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (so) break; }
        if (!so) { fprintf(stderr, "libomni_hip: OMNI_ROCTX=1 but no roctx library could be loaded (%s): no ranges\n", dlerror()); return; }
        push = reinterpret_cast<int (*)(const char*)>(dlsym(so, "roctxRangePushA"));
        pop = reinterpret_cast<int (*)()>(dlsym(so, "roctxRangePop"));
        if (!push || !pop) { push = nullptr; pop = nullptr; }
    }
};
Roctx& roctx() { static Roctx r; return r; }
}  // namespace
void trace_push(const char* name) { Roctx& r = roctx(); if (r.push) (void)r.push(name); }
void trace_pop() { Roctx& r = roctx(); if (r.pop) (void)r.pop(); }
}  // namespace omni

static omni_ctx* ctx_create(int device_id, bool high_priority);

extern "C" {

void omni_trace_push(const char* name) { if (name) omni::trace_push(name); }
void omni_trace_pop(void) { omni::trace_pop(); }

int omni_abi_version(void) { return OMNI_ABI_VERSION; }
const char* omni_last_error(void) { return omni::g_err; }

omni_ctx* omni_ctx_create(int device_id) { return ctx_create(device_id, false); }
omni_ctx* omni_ctx_create_priority(int device_id, int high_priority) { return ctx_create(device_id, high_priority != 0); }

}  // extern "C"

static omni_ctx* ctx_create(int device_id, bool high_priority) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        omni::set_error("no HIP device available (%s); this library has no CPU fallback",
                        e == hipSuccess ? "device count 0" : hipGetErrorString(e));
        return nullptr;
    }
    if (device_id < 0 || device_id >= n) { omni::set_error("device_id %d out of range [0,%d)", device_id, n); return nullptr; }
    omni_ctx* c = new omni_ctx();
    c->device = device_id;
    int least = 0, greatest = 0;                 // (numerically lower = higher priority)
    if (hipSetDevice(device_id) != hipSuccess || hipGetDeviceProperties(&c->prop, device_id) != hipSuccess ||
        (high_priority && hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) ||
        (high_priority ? hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, greatest) : hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess ||
        hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess || c->ensure_zero_page() != OMNI_OK ||
        hipStreamSynchronize(c->stream) != hipSuccess) {
        omni::set_error("failed to initialise HIP context on device %d", device_id);
        delete c;
        return nullptr;
    }
    return c;
}

extern "C" {

void omni_ctx_destroy(omni_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    c->scratch.release(); c->scratch2.release(); c->hstage.release();
    if (c->zero_page) (void)hipFree(c->zero_page);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->ev_order) (void)hipEventDestroy(c->ev_order);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int omni_ctx_sync(omni_ctx* c) {
    OMNI_REQUIRE(c, OMNI_ERR_INVALID, "null ctx");
    OMNI_HIP_TRY(hipStreamSynchronize(c->stream));
    return OMNI_OK;
}

void* omni_ctx_stream(omni_ctx* c) { return c ? (void*)c->stream : nullptr; }

int omni_ctx_device_info(omni_ctx* c, char* name, int name_len, int* n_cu, int* clock_mhz, size_t* hbm_bytes) {
    OMNI_REQUIRE(c, OMNI_ERR_INVALID, "null ctx");
    if (name && name_len > 0) { snprintf(name, name_len, "%s (%s)", c->prop.name, c->prop.gcnArchName); }
    if (n_cu) *n_cu = c->prop.multiProcessorCount;
    if (clock_mhz) *clock_mhz = c->prop.clockRate / 1000;
    if (hbm_bytes) *hbm_bytes = c->prop.totalGlobalMem;
    return OMNI_OK;
}

// ---- calibration (measurement support, like omni_sp_profile): what the fp16 matrix cores of THIS board sustain -------------------------------------
// Every SIMD of the device issues v_mfma_f32_32x32x16_f16 back to back on registers (one wave per SIMD, four independent accumulators, no memory
// traffic).  The guide's dense peak is 4096 FLOP per clock and CU at the 2.4 GHz engine clock; with the matrix cores saturated the board's power budget
// decides the clock (tools/probes/mfma_peak_probe.hip, profiles/r05u_*: 1.95 GHz, 2.0 PFLOP/s).  bench.py quotes this next to `roofline.peak`.
namespace omni {
typedef _Float16 cal_half8 __attribute__((ext_vector_type(8)));
typedef float cal_f16 __attribute__((ext_vector_type(16)));
__global__ void __launch_bounds__(256) mfma_ceiling_kernel(int iters, float* sink, unsigned long long* ticks) {
    cal_half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (float)(threadIdx.x + i)); b[i] = (_Float16)(0.002f * (float)((int)threadIdx.x - i)); }
    cal_f16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();       // constant 100 MHz
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();           // shader clock
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    if (s == 12345.678f) sink[0] = s;                                      // keeps the loop alive
    if (threadIdx.x == 0 && blockIdx.x == 0) { ticks[0] = t1 - t0; ticks[1] = r1 - r0; }
}
}  // namespace omni

int omni_ctx_mfma_ceiling(omni_ctx* c, float ms, float* tflops, float* sclk_ghz) {
    OMNI_REQUIRE(c && ms > 0.f && ms <= 2000.f, OMNI_ERR_INVALID, "omni_ctx_mfma_ceiling: null ctx or a duration outside (0, 2000] ms");
    (void)hipSetDevice(c->device);
    const int cus = c->prop.multiProcessorCount > 0 ? c->prop.multiProcessorCount : 256;
    float* sink = nullptr;
    unsigned long long* ticks = nullptr;
    OMNI_HIP_TRY(hipMalloc((void**)&sink, 4));
    if (hipMalloc((void**)&ticks, 16) != hipSuccess) { (void)hipFree(sink); OMNI_REQUIRE(false, OMNI_ERR_HIP, "omni_ctx_mfma_ceiling: hipMalloc"); }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = OMNI_OK;
    float best = 0.f, ghz = 0.f;
    do {
        if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { omni::set_error("omni_ctx_mfma_ceiling: hipEventCreate"); rc = OMNI_ERR_HIP; break; }
        // four MFMAs per iteration at ~17 ns each; a quarter of the time as warm-up (clocks settle), then the measured launch
        const int iters = (int)(ms * 1e6f / (4.f * 17.f)) + 1;
        for (int pass = 0; pass < 2 && rc == OMNI_OK; ++pass) {
            const int it = pass == 0 ? iters / 4 + 1 : iters;
            if (hipEventRecord(e0, c->stream) != hipSuccess) { rc = OMNI_ERR_HIP; break; }
            hipLaunchKernelGGL(omni::mfma_ceiling_kernel, dim3(cus), dim3(256), 0, c->stream, it, sink, ticks);
            if (hipGetLastError() != hipSuccess || hipEventRecord(e1, c->stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess) { rc = OMNI_ERR_HIP; break; }
            float t = 0.f;
            unsigned long long h[2] = {0, 0};
            if (hipEventElapsedTime(&t, e0, e1) != hipSuccess || hipMemcpy(h, ticks, 16, hipMemcpyDeviceToHost) != hipSuccess) { rc = OMNI_ERR_HIP; break; }
            if (pass == 1 && t > 0.f) {
                best = (float)((double)cus * 4.0 * it * 4.0 * 32.0 * 32.0 * 16.0 * 2.0 / (t * 1e-3) / 1e12);
                ghz = h[1] ? (float)((double)h[0] / ((double)h[1] * 10.0)) : 0.f;
            }
        }
        if (rc != OMNI_OK) omni::set_error("omni_ctx_mfma_ceiling: the calibration launch failed");
    } while (0);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(sink); (void)hipFree(ticks);
    if (rc != OMNI_OK) return rc;
    if (tflops) *tflops = best;
    if (sclk_ghz) *sclk_ghz = ghz;
    return OMNI_OK;
}

void* omni_dev_alloc(omni_ctx* c, size_t bytes) {
    if (!c) { omni::set_error("null ctx"); return nullptr; }
    (void)hipSetDevice(c->device);
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
    if (e != hipSuccess) { omni::set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); return nullptr; }
    return p;
}

int omni_dev_free(omni_ctx* c, void* p) {
    OMNI_REQUIRE(c, OMNI_ERR_INVALID, "null ctx");
    if (p) OMNI_HIP_TRY(hipFree(p));
    return OMNI_OK;
}

void* omni_host_alloc(size_t bytes) {
    void* p = nullptr;
    hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) { omni::set_error("hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); return nullptr; }
    return p;
}

int omni_host_free(void* p) {
    if (p) OMNI_HIP_TRY(hipHostFree(p));
    return OMNI_OK;
}

int omni_memcpy_h2d(omni_ctx* c, void* dst, const void* src, size_t bytes) {
    OMNI_REQUIRE(c && dst && src, OMNI_ERR_INVALID, "null argument");
    OMNI_HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    OMNI_HIP_TRY(hipStreamSynchronize(c->stream));
    return OMNI_OK;
}

int omni_memcpy_d2h(omni_ctx* c, void* dst, const void* src, size_t bytes) {
    OMNI_REQUIRE(c && dst && src, OMNI_ERR_INVALID, "null argument");
    OMNI_HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    OMNI_HIP_TRY(hipStreamSynchronize(c->stream));
    return OMNI_OK;
}

int omni_memcpy_d2h_async(omni_ctx* c, void* dst_pinned, const void* src, size_t bytes) {
    OMNI_REQUIRE(c && dst_pinned && src, OMNI_ERR_INVALID, "null argument");
    OMNI_HIP_TRY(hipMemcpyAsync(dst_pinned, src, bytes, hipMemcpyDeviceToHost, c->stream));
    return OMNI_OK;
}

int omni_ctx_order_after(omni_ctx* later, omni_ctx* earlier) {
    OMNI_REQUIRE(later && earlier && later->device == earlier->device, OMNI_ERR_INVALID, "omni_ctx_order_after: two contexts of one device");
    if (later == earlier) return OMNI_OK;
    if (!earlier->ev_order) OMNI_HIP_TRY(hipEventCreateWithFlags(&earlier->ev_order, hipEventDisableTiming));
    OMNI_HIP_TRY(hipEventRecord(earlier->ev_order, earlier->stream));
    OMNI_HIP_TRY(hipStreamWaitEvent(later->stream, earlier->ev_order, 0));
    return OMNI_OK;
}

int omni_timer_start(omni_ctx* c) {
    OMNI_REQUIRE(c, OMNI_ERR_INVALID, "null ctx");
    OMNI_HIP_TRY(hipEventRecord(c->ev0, c->stream));
    return OMNI_OK;
}

int omni_timer_stop(omni_ctx* c, float* ms) {
    OMNI_REQUIRE(c && ms, OMNI_ERR_INVALID, "null argument");
    OMNI_HIP_TRY(hipEventRecord(c->ev1, c->stream));
    OMNI_HIP_TRY(hipEventSynchronize(c->ev1));
    OMNI_HIP_TRY(hipEventElapsedTime(ms, c->ev0, c->ev1));
    return OMNI_OK;
}

}  // extern "C"
