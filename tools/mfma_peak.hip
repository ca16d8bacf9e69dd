// Calibration: what does a pure register-resident fp32 MFMA loop reach on this box (no LDS, no HBM)?
// Prints TFLOP/s and the shader clock seen under that load (clock64 vs the 100 MHz wall clock).
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o tools/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, long long* clk, int iters) {
    f4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
    long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    long long c1 = clock64(), w1 = wall_clock64();
    f4 s = acc[0];
#pragma unroll
    for (int i = 1; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y + s.z + s.w;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

template <int NACC>
void run(int blocks, int iters, float* out, long long* clk) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(256), 0, 0, out, clk, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(256), 0, 0, out, clk, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[2]; hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    double flop = (double)blocks * 4 /*waves*/ * iters * NACC * 2048.0;
    printf("{\"nacc\": %d, \"blocks\": %d, \"iters\": %d, \"ms\": %.3f, \"tflops\": %.2f, \"shader_mhz\": %.0f}\n", NACC, blocks, iters, ms,
           flop / ms * 1e-9, (double)h[0] / (double)h[1] * 100.0);
}

int main() {
    float* out; long long* clk;
    hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&clk, 16);
    run<16>(256, 20000, out, clk);    // 1 wave per SIMD
    run<16>(512, 20000, out, clk);    // 2 waves per SIMD
    run<8>(1024, 40000, out, clk);    // 4 waves per SIMD
    run<4>(512, 80000, out, clk);     // short dependency distance
    run<16>(2048, 20000, out, clk);   // long run: sustained clock
    return 0;
}
