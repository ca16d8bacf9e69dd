// Calibration (round 4): how much vector-unit work hides under fp16 MFMAs on one SIMD of this chip?
// Three experiments on 256 workgroups (one per CU), random-ish register operands, no memory traffic:
//   X  cross-wave: waves 0-3 issue only MFMAs, waves 4-7 (their SIMD partners) only vector instructions; each alone, then together.
//      together == max(a, b): the two pipes overlap across waves; together == a + b: they do not.
//   S  same wave: every MFMA followed by k independent v_fma_f32 (k = 0 .. 10), one wave per SIMD: cycles per MFMA slice.
//   P  the same stream on BOTH waves of every SIMD (8 waves): cycles per slice pair.
// for v_mfma_f32_32x32x16_f16 (the attention kernels) and v_mfma_f32_16x16x32_f16 (the split GEMM). It answers whether a second workgroup's
// epilogue (vector work) can run "under" the first one's MFMA loop on the same SIMD — the two-tiles-in-flight question of DESIGN.md §9.
// Build: hipcc -O3 --offload-arch=gfx950 tools/coissue_f16.hip -o tools/coissue_f16     Output: one JSON line per experiment.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int SHAPE> struct Mf;
template <> struct Mf<32> {
    typedef f16v Acc;
    static constexpr int CYC = 32;
    __device__ static Acc mfma(h8 a, h8 b, Acc c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};
template <> struct Mf<16> {
    typedef f4 Acc;
    static constexpr int CYC = 16;
    __device__ static Acc mfma(h8 a, h8 b, Acc c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};

__device__ __forceinline__ h8 mk(float s) {
    h8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (_Float16)(s * (float)(i + 1));
    return v;
}

// mode 0: MFMA waves (wave < 4) only; 1: vector waves (wave >= 4) only; 2: both. TRANS: the vector waves issue v_exp_f32 instead of v_fma_f32.
template <int SHAPE, bool TRANS>
__global__ __launch_bounds__(512) void cross_kernel(float* out, int iters, int mode, unsigned long long* cyc) {
    typedef Mf<SHAPE> M;
    const int wave = threadIdx.x >> 6;
    float r = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (wave < 4) {
        if (mode != 1) {
            typename M::Acc acc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < (int)(sizeof(acc[0]) / 4); ++e) acc[i][e] = 0.f;
            const h8 a = mk(threadIdx.x * 1e-3f), b = mk(1e-3f);
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] = M::mfma(a, b, acc[i]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) r += acc[i][0];
        }
    } else if (mode != 0) {
        float x[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 1e-3f + i;
        const int n = 32 * M::CYC / (TRANS ? 8 : 4) / 8;   // vector instructions worth the same nominal cycles as the partner's 32 MFMAs
        for (int it = 0; it < iters; ++it) {
            for (int u = 0; u < n; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = TRANS ? __builtin_amdgcn_exp2f(x[i]) : fmaf(x[i], 0.999f, 0.001f);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) r += x[i];
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 512 + threadIdx.x] = r;
    if (blockIdx.x == 100 && (threadIdx.x == 0 || threadIdx.x == 256)) cyc[threadIdx.x >> 8] = t1 - t0;
}

// every MFMA followed by K independent v_fma_f32 (or v_exp_f32 every fourth, EXPS per slice); WAVES = 4 (one per SIMD) or 8 (two per SIMD, same stream)
template <int SHAPE, int K, int EXPS>
__global__ __launch_bounds__(512) void slice_kernel(float* out, int iters, unsigned long long* cyc) {
    typedef Mf<SHAPE> M;
    typename M::Acc acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < (int)(sizeof(acc[0]) / 4); ++e) acc[i][e] = 0.f;
    const h8 a = mk(threadIdx.x * 1e-3f), b = mk(1e-3f);
    float x[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) x[i] = threadIdx.x * 1e-3f + i;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            acc[u & 1] = M::mfma(a, b, acc[u & 1]);
#pragma unroll
            for (int i = 0; i < K; ++i) x[i] = i < EXPS ? __builtin_amdgcn_exp2f(x[i]) : fmaf(x[i], 0.999f, 0.001f);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float r = acc[0][0] + acc[1][0];
#pragma unroll
    for (int i = 0; i < 12; ++i) r += x[i];
    out[blockIdx.x * 512 + threadIdx.x] = r;
    if (blockIdx.x == 100 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

static float time_launch(void (*launch)(void*), void* ctx) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(ctx); hipDeviceSynchronize();
    hipEventRecord(e0, 0); launch(ctx); hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

static float* g_out; static unsigned long long* g_cyc; static int g_iters = 2000;

template <int SHAPE, bool TRANS>
static void cross() {
    float ms[3]; unsigned long long c[3][2];
    for (int mode = 0; mode < 3; ++mode) {
        struct C { int mode; } cx{mode};
        ms[mode] = time_launch([](void* p) { hipLaunchKernelGGL((cross_kernel<SHAPE, TRANS>), dim3(256), dim3(512), 0, 0, g_out, g_iters, ((C*)p)->mode, g_cyc); }, &cx);
        hipMemcpy(c[mode], g_cyc, 16, hipMemcpyDeviceToHost);
    }
    printf("{\"exp\": \"X\", \"mfma\": \"%s\", \"vector\": \"%s\", \"mfma_only_ms\": %.3f, \"vector_only_ms\": %.3f, \"both_ms\": %.3f, \"sum_ms\": %.3f, \"cycles_mfma_wave\": [%llu, %llu], \"cycles_vector_wave\": [%llu, %llu]}\n",
           SHAPE == 32 ? "32x32x16_f16" : "16x16x32_f16", TRANS ? "v_exp_f32" : "v_fma_f32", ms[0], ms[1], ms[2], ms[0] + ms[1], c[0][0], c[2][0], c[1][1], c[2][1]);
}

template <int SHAPE, int K, int EXPS>
static void slice() {
    for (int waves = 4; waves <= 8; waves += 4) {
        struct C { int w; } cx{waves};
        const float ms = time_launch([](void* p) { hipLaunchKernelGGL((slice_kernel<SHAPE, K, EXPS>), dim3(256), dim3(64 * ((C*)p)->w), 0, 0, g_out, g_iters, g_cyc); }, &cx);
        unsigned long long c; hipMemcpy(&c, g_cyc, 8, hipMemcpyDeviceToHost);
        const double per = (double)c / ((double)g_iters * 16);
        printf("{\"exp\": \"%s\", \"mfma\": \"%s\", \"fillers_per_mfma\": %d, \"of_them_v_exp\": %d, \"waves_per_simd\": %d, \"ms\": %.3f, \"cycles_per_slice_per_wave\": %.1f, \"mfma_cycles\": %d}\n",
               waves == 4 ? "S" : "P", SHAPE == 32 ? "32x32x16_f16" : "16x16x32_f16", K, EXPS, waves / 4, ms, per, Mf<SHAPE>::CYC);
    }
}

int main() {
    hipMalloc(&g_out, 256 * 512 * 4); hipMalloc(&g_cyc, 64);
    cross<32, false>(); cross<32, true>(); cross<16, false>(); cross<16, true>();
    slice<32, 0, 0>(); slice<32, 2, 0>(); slice<32, 4, 0>(); slice<32, 5, 0>(); slice<32, 6, 0>(); slice<32, 8, 0>(); slice<32, 10, 0>(); slice<32, 6, 2>();
    slice<16, 0, 0>(); slice<16, 1, 0>(); slice<16, 2, 0>(); slice<16, 3, 0>(); slice<16, 4, 0>(); slice<16, 6, 0>();
    return 0;
}
