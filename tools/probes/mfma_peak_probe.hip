// What the fp16 matrix cores of THIS board deliver when nothing else is in the way: every wave of a full grid issues v_mfma_f32_32x32x16_f16 back to back on
// registers (four independent accumulators per wave, no memory traffic), timed with HIP events and -- inside the kernel -- with s_memtime (shader-clock cycles).
//   cycles / seconds = the clock the kernel actually ran at;  FLOP / seconds = the ceiling a real kernel is measured against (the guide's 2.5 PFLOP/s is
//   256 CUs x 4096 FLOP per clock x 2.4 GHz).
// build + run (GPU box):  hipcc --offload-arch=gfx950 -O3 -o omni-swarm_amd/build/mfma_peak_probe tools/probes/mfma_peak_probe.hip && omni-swarm_amd/build/mfma_peak_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

// DUTY: 4 = back to back; 1 = one MFMA, then ~96 cycles of dependent VALU work (a quarter of the matrix pipe's capacity per wave)
template <int DUTY>
__global__ void __launch_bounds__(512) mfma_loop(int iters, float* sink, unsigned long long* cycles) {
    half8_t a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x - i)); }
    floatx16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    float pad = (float)threadIdx.x;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();   // constant 100 MHz
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
        if constexpr (DUTY == 4) {
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
        } else {
#pragma unroll
            for (int k = 0; k < 24; ++k) pad = __builtin_fmaf(pad, 1.0001f, 0.5f);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = pad;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    if (s == 12345.678f) sink[0] = s;                                  // keeps the loop alive
    if (threadIdx.x == 0 && blockIdx.x == 0) { cycles[0] = t1 - t0; cycles[1] = r1 - r0; }
}

int main(int argc, char** argv) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, 0) != hipSuccess) { fprintf(stderr, "no device\n"); return 1; }
    const int cus = p.multiProcessorCount, waves = 8;                 // one workgroup of 8 waves per CU: two waves per SIMD, as the ping-pong convolution kernel runs
    float* sink; unsigned long long* cyc;
    (void)hipMalloc(&sink, 4); (void)hipMalloc(&cyc, 64);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    printf("%s: %d CUs, clockRate %d kHz\n", p.name, cus, p.clockRate);
    auto run = [&](auto kfn, const char* what, int waves_, int mfma_per_iter, int iters) {
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(kfn, dim3(cus), dim3(64 * waves_), 0, 0, iters, sink, cyc);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
            unsigned long long h[8]; (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
            const double flop = (double)cus * waves_ * iters * (double)mfma_per_iter * 32 * 32 * 16 * 2;
            const double memtime_ghz = (double)h[0] / ((double)h[1] * 10.0);          // s_memrealtime ticks are 10 ns
            printf("%-34s iters %8d: %9.3f ms  %7.1f TFLOP/s (%.3f of 2500)  wave 0: %llu s_memtime ticks = %.3f GHz (against s_memrealtime) for %.1f ms, %.2f ticks per MFMA of its own\n", what, iters, ms,
                   flop / ms / 1e9, flop / ms / 1e9 / 2500.0, h[0], memtime_ghz, (double)h[1] * 1e-5, (double)h[0] / ((double)iters * mfma_per_iter));
        }
    };
    (void)waves;
    run(mfma_loop<4>, "back to back, 2 waves per SIMD", 8, 4, 2000);
    run(mfma_loop<4>, "back to back, 2 waves per SIMD", 8, 4, 200000);
    run(mfma_loop<4>, "back to back, 2 waves per SIMD", 8, 4, 1000000);
    run(mfma_loop<4>, "back to back, 1 wave per SIMD", 4, 4, 400000);
    run(mfma_loop<1>, "1 MFMA + 24 FMA, 2 waves per SIMD", 8, 1, 400000);
    run(mfma_loop<1>, "1 MFMA + 24 FMA, 1 wave per SIMD", 4, 1, 400000);
    return 0;
}
