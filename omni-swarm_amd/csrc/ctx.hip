// Context, device memory and timer entry points of the C ABI (include/omni_hip.h).
// Replaces the cudaStreamCreate / cudaMalloc / cudaMallocHost plumbing of the reference's
// TensorRTInferenceGeneric (swarm_loop/src/tensorrt_generic.cpp:14-36,99-120).
#include "common.h"

namespace omni {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace omni

extern "C" {

int omni_abi_version(void) { return OMNI_ABI_VERSION; }
const char* omni_last_error(void) { return omni::g_err; }

omni_ctx* omni_ctx_create(int device_id) {
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
    if (hipSetDevice(device_id) != hipSuccess || hipGetDeviceProperties(&c->prop, device_id) != hipSuccess ||
        hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess || c->ensure_zero_page() != OMNI_OK ||
        hipStreamSynchronize(c->stream) != hipSuccess) {
        omni::set_error("failed to initialise HIP context on device %d", device_id);
        delete c;
        return nullptr;
    }
    return c;
}

void omni_ctx_destroy(omni_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    c->scratch.release(); c->scratch2.release(); c->hstage.release();
    if (c->zero_page) (void)hipFree(c->zero_page);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
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
