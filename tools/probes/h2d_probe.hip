// probe: pinned host -> HBM upload of one micro-batch's images (64 x 480 x 600 u8) as ONE 2-D copy with pitch == width (what omni_cam_enqueue_host
// issues, cam.hip) against a plain 1-D copy of the same bytes.   hipcc --offload-arch=gfx950 -O2 -o /tmp/h2d_probe tools/probes/h2d_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
int main() {
    const size_t W = 600, rows = 64 * 480, bytes = W * rows;
    unsigned char *h, *d;
    hipHostMalloc((void**)&h, bytes, hipHostMallocDefault);
    hipMalloc((void**)&d, bytes);
    memset(h, 7, bytes);
    hipStream_t s;
    hipStreamCreate(&s);
    auto run = [&](int mode, size_t pitch) {
        for (int w = 0; w < 3; ++w) { if (mode == 0) hipMemcpy2DAsync(d, W, h, pitch, W, rows, hipMemcpyHostToDevice, s); else hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s); }
        hipStreamSynchronize(s);
        auto t0 = std::chrono::steady_clock::now();
        const int N = 20;
        for (int i = 0; i < N; ++i) {
            if (mode == 0) hipMemcpy2DAsync(d, W, h, pitch, W, rows, hipMemcpyHostToDevice, s); else hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s);
            hipStreamSynchronize(s);
        }
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / N;
        printf("%s: %.3f ms per %.1f MB = %.1f GB/s\n", mode == 0 ? "hipMemcpy2DAsync (pitch == width)" : "hipMemcpyAsync", ms, bytes / 1e6, bytes / ms / 1e6);
    };
    run(0, W);
    run(1, W);
    run(0, W);
    run(1, W);
    return 0;
}
