// SEANet strided conv of encoder stage 1 (64 -> 128 channels, k = 8, stride 4, causal) with the weights stationary in
// registers: out[u] = W . [x[4u-4] | ... | x[4u+3]] + b, x already ELU'd by the producing block.
// As a tiled GEMM this layer has a single 128-wide n-tile, so every workgroup re-fetched the whole 256 KB weight
// matrix per 128 output rows (33 FLOP per byte moved into LDS: 118 TFLOP/s). Here wave w keeps output channels
// 32w..32w+31 x K = 512 in 256 registers (MFMA A-fragment order, one wave per SIMD owns 512 registers per lane) for
// the lifetime of the persistent workgroup; only activations move: a tile of 64 output rows = 260 input rows (65 KB)
// goes through LDS, the next tile is prefetched into registers during the MFMAs, and all four waves read every row
// tile as the B operand (fetched one k-group ahead). Same scheme as lstm_seq.hip / seanet_res128.hip.
// k order (tap-major, channels ascending, the b128 lane trick) and the bias add equal the GEMM path: outputs are
// bit-identical (tests/test_acoustic_gpu.py, option "fused_down64"). Needs L % 4 == 0 (no right padding); otherwise the
// GEMM path is used. Reflect padding at the clip start is an index map while staging (row -r -> row r).
#include "gemm_core.h"
#include "encodec_kernels.h"

namespace at {

constexpr int D64_TU = 64;                 // output rows per tile
constexpr int D64_ROWS = 4 * D64_TU + 4;   // input rows per tile: row i <-> time 4*u0 - 4 + i
constexpr int D64_CHUNKS = D64_ROWS * 16;  // float4 chunks of the input tile
constexpr int D64_PRE = (D64_CHUNKS + 255) / 256;
constexpr int D64_LDS_FLOATS = D64_ROWS * 64;

__global__ __launch_bounds__(256, 1) void seanet_down64_kernel(Down64Args a) {
    extern __shared__ __attribute__((aligned(16))) float Xs[];   // [260 rows][64], 16-B chunk ^= (row >> 2) & 15
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, q = lane >> 4;
    const int L = a.L, Lo = L / 4;
    const int tiles_per_clip = (Lo + D64_TU - 1) / D64_TU;
    const long long total_tiles = (long long)a.B * tiles_per_clip;

    // ---- weights -> registers, once: wr[kg][n] = W[32w + 16n + r16][kg*16 + q*4 .. +3] -----------------------------------
    f4 wr[32][2];
#pragma unroll
    for (int kg = 0; kg < 32; ++kg)
#pragma unroll
        for (int n = 0; n < 2; ++n) wr[kg][n] = *reinterpret_cast<const f4*>(a.w + (wave * 32 + n * 16 + r16) * 512 + kg * 16 + q * 4);
    f4 bias[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) bias[n] = *reinterpret_cast<const f4*>(a.b + wave * 32 + n * 16 + q * 4);

    f4 pre[D64_PRE];
    auto prefetch = [&](long long tile) {
        const long long b = tile / tiles_per_clip;
        const int u0 = (int)(tile - b * tiles_per_clip) * D64_TU;
        const float* xb = a.x + b * (long long)L * 64;
#pragma unroll
        for (int j = 0; j < D64_PRE; ++j) {
            int c = tid + 256 * j;
            c = c < D64_CHUNKS ? c : D64_CHUNKS - 1;
            int tau = 4 * u0 - 4 + (c >> 4);
            tau = tau < 0 ? -tau : tau;            // causal reflect padding at the clip start
            tau = tau > L - 1 ? L - 1 : tau;       // rows past the end only feed outputs that are never stored
            pre[j] = *reinterpret_cast<const f4*>(xb + (long long)tau * 64 + (c & 15) * 4);
        }
    };
    if ((long long)blockIdx.x < total_tiles) prefetch(blockIdx.x);

    // B-fragment addresses: row = 4*(16m + r16) + tap, chunk = c16*4 + q, swizzle (row >> 2) & 15 = (r16 + (tap >> 2)) & 15
    for (long long tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
        const long long b = tile / tiles_per_clip;
        const int u0 = (int)(tile - b * tiles_per_clip) * D64_TU;
        __syncthreads();   // previous tile's readers are done
#pragma unroll
        for (int j = 0; j < D64_PRE; ++j) {
            const int c = tid + 256 * j;
            if (c < D64_CHUNKS) {
                const int row = c >> 4, ch = c & 15;
                *reinterpret_cast<f4*>(Xs + row * 64 + ((ch ^ ((row >> 2) & 15)) << 2)) = pre[j];
            }
        }
        __syncthreads();
        if (tile + gridDim.x < total_tiles) prefetch(tile + gridDim.x);   // flies during the MFMAs below
        __builtin_amdgcn_sched_barrier(0);
        f4 acc[4][2];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) acc[m][n] = f4{0.f, 0.f, 0.f, 0.f};
        auto frag = [&](int kg, int m) -> f4 {
            const int tap = kg >> 2, c16 = kg & 3;
            const int sw = (r16 + (tap >> 2)) & 15;
            return *reinterpret_cast<const f4*>(Xs + (4 * (16 * m + r16) + tap) * 64 + (((c16 * 4 + q) ^ sw) << 2));
        };
        f4 xb[4], xn[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) xb[m] = frag(0, m);
#pragma unroll
        for (int kg = 0; kg < 32; ++kg) {
            if (kg + 1 < 32) {
#pragma unroll
                for (int m = 0; m < 4; ++m) xn[m] = frag(kg + 1, m);
            }
            __builtin_amdgcn_sched_barrier(0);   // keep the prefetch in front of the MFMAs (one wave per SIMD)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[kg][n][e], xb[m][e], acc[m][n], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < 4; ++m) xb[m] = xn[m];
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int u = u0 + m * 16 + r16;
            if (u < Lo) {
                float* dst = a.out + (b * (long long)Lo + u) * 128 + wave * 32 + q * 4;
#pragma unroll
                for (int n = 0; n < 2; ++n) *reinterpret_cast<f4*>(dst + n * 16) = acc[m][n] + bias[n];
            }
        }
    }
}

int launch_seanet_down64(const Down64Args& a, hipStream_t stream) {
    AT_REQUIRE(a.L >= 8 && a.L % 4 == 0 && a.B >= 1, "register-stationary stride-4 conv needs L % 4 == 0");
    const size_t lds = (size_t)D64_LDS_FLOATS * sizeof(float);
    { static LdsAttrFlags lds_attr_0; if (int rc = set_max_dynamic_lds(lds_attr_0, seanet_down64_kernel, lds)) return rc; }
    const long long tiles = (long long)a.B * ((a.L / 4 + D64_TU - 1) / D64_TU);
    const int grid = (int)(tiles < 256 ? tiles : 256);
    hipLaunchKernelGGL(seanet_down64_kernel, dim3(grid), dim3(256), lds, stream, a);
    AT_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace at
