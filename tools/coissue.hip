// Calibration: do MFMA (matrix core) and VALU instructions of two different waves on one SIMD overlap?
// 512-thread blocks, one per CU: waves 0-3 run `ma` MFMAs per iteration, waves 4-7 run `va` v_fma per iteration.
// Build: hipcc -O3 --offload-arch=gfx950 tools/coissue.hip -o tools/coissue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void k(float* out, int iters, int do_mfma, int do_valu) {
    const int wave = threadIdx.x >> 6;
    float r = 0.f;
    if (wave < 4) {
        if (do_mfma) {
            f4 acc[4] = {f4{0, 0, 0, 0}, f4{0, 0, 0, 0}, f4{0, 0, 0, 0}, f4{0, 0, 0, 0}};
            float a = threadIdx.x * 1e-3f, b = 1e-3f;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
            }
            r = acc[0].x + acc[1].y + acc[2].z + acc[3].w;
        }
    } else if (do_valu) {
        float x[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 1e-3f + i;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 32; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = fmaf(x[i], 0.999f, 0.001f);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) r += x[i];
    }
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

static float run(float* out, int iters, int m, int v) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, iters, m, v);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, iters, m, v);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    const int iters = 4000;   // per iteration: 32 MFMAs (1024 cycles) vs 256 v_fma (1024 cycles)
    printf("{\"mfma_only_ms\": %.3f, \"valu_only_ms\": %.3f, \"both_ms\": %.3f}\n", run(out, iters, 1, 0), run(out, iters, 0, 1), run(out, iters, 1, 1));
    return 0;
}
