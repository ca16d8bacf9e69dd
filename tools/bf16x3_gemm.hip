// Prototype for DESIGN.md section 9.1: C = A . B^T with fp32 operands split exactly into three bf16 pieces and the six leading
// cross products accumulated on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16) into fp32. Measures (a) the error against a
// float64 reference next to the error of a plain fp32 k-ordered FMA chain (what the product path's fp32 MFMA computes) and
// (b) the throughput in fp32-equivalent TFLOP/s on the conformer FFN shape.
// Not part of the product; build: hipcc -O3 --offload-arch=gfx950 tools/bf16x3_gemm.hip -o tools/bf16x3_gemm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

constexpr int BM = 256, BN = 256, BK = 16;        // block tile; 8 waves as 4 (m) x 2 (n), each 64 x 128 = 2 x 4 MFMA tiles
constexpr int PIECE_A = BM * BK;                  // bf16 elements of one piece of the A tile
constexpr int STAGE = 3 * (BM + BN) * BK;         // bf16 elements per LDS stage (48 KB)

// fp32 [rows][K] -> three bf16 pieces in the K-blocked layout [piece][K/16][rows][16] (one K tile of all rows is contiguous;
// row-major pieces measured 157-179 instead of 194-197 fp32-equivalent TFLOP/s)
__global__ void split3(const float* __restrict__ x, __bf16* __restrict__ out, long long rows, int K) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * K) return;
    const long long r = i / K;
    const int k = (int)(i - r * K);
    const float a = x[i];
    const __bf16 a1 = (__bf16)a;
    const float r1 = a - (float)a1;
    const __bf16 a2 = (__bf16)r1;
    const float r2 = r1 - (float)a2;
    const __bf16 a3 = (__bf16)r2;
    const long long o = ((long long)(k / 16) * rows + r) * 16 + (k % 16);   // K-blocked [piece][K/16][rows][16], as the product uses
    const long long ps = rows * (long long)K;
    out[o] = a1; out[ps + o] = a2; out[2 * ps + o] = a3;
}

__global__ __launch_bounds__(512, 1) void gemm_bf16x3(const __bf16* __restrict__ A, const __bf16* __restrict__ B, float* __restrict__ C,
                                                       int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) __bf16 lds[];   // [2 stages][A: 3 pieces x 256 x 16 | B: 3 pieces x 256 x 16]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int ntn = N / BN;
    const int m0 = (blockIdx.x / ntn) * BM, n0 = (blockIdx.x % ntn) * BN;
    const long long psA = (long long)M * K, psB = (long long)N * K;
    const int nk = K / BK;
    // staging: per K tile each piece of A (and B) is 256 rows x 32 B = 512 chunks of 16 B: thread tid moves chunk (row tid / 2,
    // half tid % 2) of every piece: 3 + 3 chunks
    u4 st[6];
    auto load = [&](int kt) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            st[p] = *reinterpret_cast<const u4*>(A + p * psA + ((long long)kt * M + m0) * 16 + tid * 8);
            st[3 + p] = *reinterpret_cast<const u4*>(B + p * psB + ((long long)kt * N + n0) * 16 + tid * 8);
        }
    };
    auto store = [&](int buf) {
        __bf16* s = lds + buf * STAGE;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            *reinterpret_cast<u4*>(s + p * PIECE_A + tid * 8) = st[p];
            *reinterpret_cast<u4*>(s + 3 * PIECE_A + p * PIECE_A + tid * 8) = st[3 + p];
        }
    };
    f16v acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    load(0);
    store(0);
    __syncthreads();
    const int frow = lane & 31, fhalf = lane >> 5, fsw = fhalf;   // swapping the 16-B halves of rows with (row >> 3) & 1 removes the 2-way ds_read_b128 conflict: measured no gain
    for (int kt = 0; kt < nk; ++kt) {
        const __bf16* s = lds + (kt & 1) * STAGE;
        bf16x8 a[3][2], b[3][4];
        constexpr int RA[3] = {0, 2, 1}, RB[3] = {2, 0, 1};   // in the order the products consume them
#pragma unroll
        for (int o = 0; o < 3; ++o) {
#pragma unroll
            for (int i = 0; i < 2; ++i) a[RA[o]][i] = *reinterpret_cast<const bf16x8*>(s + RA[o] * PIECE_A + (wm * 64 + i * 32 + frow) * 16 + fsw * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j) b[RB[o]][j] = *reinterpret_cast<const bf16x8*>(s + (3 + RB[o]) * PIECE_A + (wn * 128 + j * 32 + frow) * 16 + fsw * 8);
        }
        if (kt + 1 < nk) load(kt + 1);
        // six leading cross products, smallest first
        constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[PB[t]][j], a[PA[t]][i], acc[i][j], 0, 0, 0);
        if (kt + 1 < nk) store((kt + 1) & 1);
        __syncthreads();
    }
    // swapped operands (B as the MFMA "A" operand): lane holds output row m = frow of its 32-row tile and columns
    // n = 8*(r/4) + 4*fhalf + (r%4) of the 32-column tile -> four float4 stores per tile
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float* dst = C + (long long)(m0 + wm * 64 + i * 32 + frow) * N + n0 + wn * 128 + j * 32 + 4 * fhalf;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                *reinterpret_cast<float4*>(dst + 8 * g) = v;
            }
        }
}


// Variant 1 (measured 4.07 vs 4.18 ms on ffn1: memory latency is not what bounds the product kernel; random-data bf16 MFMA throughput
// here is 1.19 PFLOP/s against 1.3-1.5 for the best published HIP GEMM structure on this chip): operands by LDS-DMA (global_load_lds_dwordx4: no staging registers) into a 3-stage ring, two K tiles in flight across a
// raw s_barrier with a counted vmcnt (cdna_hip_programming.md section 5, "Pipelining across barriers").
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
__global__ __launch_bounds__(512, 1) void gemm_bf16x3_glds(const __bf16* __restrict__ A, const __bf16* __restrict__ B, float* __restrict__ C,
                                                            int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) __bf16 lds[];   // [3 stages][A: 3 pieces x 256 x 16 | B: 3 pieces x 256 x 16]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int ntn = N / BN;
    const int m0 = (blockIdx.x / ntn) * BM, n0 = (blockIdx.x % ntn) * BN;
    const long long psA = (long long)M * K, psB = (long long)N * K;
    const int nk = K / BK;
    // one piece of an operand tile = 256 rows x 32 B = 8 KiB = one 1-KiB DMA per wave: lane l of wave w moves chunk 64 w + l
    auto issue = [&](int kt, int buf) {
        __bf16* s = lds + buf * STAGE + wave * 512;   // wave-uniform destination; the hardware adds lane * 16 B
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            __builtin_amdgcn_global_load_lds((glb_void*)(A + p * psA + ((long long)kt * M + m0) * 16 + tid * 8), (lds_void*)(s + p * PIECE_A), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_void*)(B + p * psB + ((long long)kt * N + n0) * 16 + tid * 8), (lds_void*)(s + (3 + p) * PIECE_A), 16, 0, 0);
        }
    };
    f16v acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    issue(0, 0);
    if (nk > 1) { issue(1, 1); asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); } else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    __builtin_amdgcn_s_barrier();
    const int frow = lane & 31, fhalf = lane >> 5, fsw = fhalf;   // swapping the 16-B halves of rows with (row >> 3) & 1 removes the 2-way ds_read_b128 conflict: measured no gain
    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const __bf16* s = lds + buf * STAGE;
        if (kt + 2 < nk) issue(kt + 2, buf == 0 ? 2 : buf - 1);   // the stage read in iteration kt - 1: every wave is past that barrier
        bf16x8 a[3][2], b[3][4];
        constexpr int RA[3] = {0, 2, 1}, RB[3] = {2, 0, 1};   // in the order the products consume them
#pragma unroll
        for (int o = 0; o < 3; ++o) {
#pragma unroll
            for (int i = 0; i < 2; ++i) a[RA[o]][i] = *reinterpret_cast<const bf16x8*>(s + RA[o] * PIECE_A + (wm * 64 + i * 32 + frow) * 16 + fsw * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j) b[RB[o]][j] = *reinterpret_cast<const bf16x8*>(s + (3 + RB[o]) * PIECE_A + (wn * 128 + j * 32 + frow) * 16 + fsw * 8);
        }
        constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[PB[t]][j], a[PA[t]][i], acc[i][j], 0, 0, 0);
        // K tile kt + 1 (this wave's part) has landed; kt + 2 may still be in flight. The barrier then makes every wave's part
        // visible before anyone reads it, and retires this iteration's reads of `buf` before it is overwritten two iterations on.
        if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        buf = buf == 2 ? 0 : buf + 1;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float* dst = C + (long long)(m0 + wm * 64 + i * 32 + frow) * N + n0 + wn * 128 + j * 32 + 4 * fhalf;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                *reinterpret_cast<float4*>(dst + 8 * g) = v;
            }
        }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

static int g_variant = 0;
static int run(int M, int N, int K, bool check) {
    std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
    unsigned long long s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (float)((s >> 11) * (1.0 / 9007199254740992.0)) * 2.f - 1.f; };
    for (auto& v : hA) v = rnd();
    for (auto& v : hB) v = rnd() * 0.05f;
    float *dA, *dB, *dC;
    __bf16 *sA, *sB;
    CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dB, hB.size() * 4)); CK(hipMalloc(&dC, (size_t)M * N * 4));
    CK(hipMalloc(&sA, hA.size() * 6)); CK(hipMalloc(&sB, hB.size() * 6));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(split3, dim3((unsigned)(((long long)M * K + 255) / 256)), dim3(256), 0, 0, dA, sA, (long long)M, K);
    hipLaunchKernelGGL(split3, dim3((unsigned)(((long long)N * K + 255) / 256)), dim3(256), 0, 0, dB, sB, (long long)N, K);
    const size_t ldsb = (g_variant ? 3 : 2) * STAGE * sizeof(__bf16);
    auto kern = g_variant ? gemm_bf16x3_glds : gemm_bf16x3;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    const dim3 grid((M / BM) * (N / BN));
    hipLaunchKernelGGL(kern, grid, dim3(512), ldsb, 0, sA, sB, dC, M, N, K);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = check ? 1 : 5;
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, grid, dim3(512), ldsb, 0, sA, sB, dC, M, N, K);
    hipEventRecord(e1, 0); CK(hipEventSynchronize(e1));
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    printf("{\"variant\": %d, \"M\": %d, \"N\": %d, \"K\": %d, \"ms\": %.3f, \"fp32_equiv_tflops\": %.1f", g_variant, M, N, K, ms, 2.0 * M * N * K / ms * 1e-9);
    if (check) {
        std::vector<float> hC((size_t)M * N);
        CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
        double e_split = 0, e_chain = 0, ref_rms = 0;
        long long cnt = 0;
        for (int m = 0; m < M; m += 7)
            for (int n = 0; n < N; n += 5) {
                double ref = 0; float chain = 0.f;
                for (int k = 0; k < K; ++k) { ref += (double)hA[(size_t)m * K + k] * hB[(size_t)n * K + k]; chain = fmaf(hA[(size_t)m * K + k], hB[(size_t)n * K + k], chain); }
                e_split = fmax(e_split, fabs(hC[(size_t)m * N + n] - ref));
                e_chain = fmax(e_chain, fabs((double)chain - ref));
                ref_rms += ref * ref; ++cnt;
            }
        printf(", \"max_abs_err_bf16x3\": %.3e, \"max_abs_err_fp32_chain\": %.3e, \"ref_rms\": %.3e", e_split, e_chain, sqrt(ref_rms / cnt));
    }
    printf("}\n");
    hipFree(dA); hipFree(dB); hipFree(dC); hipFree(sA); hipFree(sB);
    return 0;
}

int main(int argc, char** argv) {
  for (g_variant = 0; g_variant < 2; ++g_variant) {
    if (run(512, 512, 1024, true)) return 1;       // accuracy
    if (run(96000, 4096, 1024, false)) return 1;   // conformer ffn1
    if (run(96000, 1024, 4096, false)) return 1;   // conformer ffn2
  }
    return 0;
}
