// SEANet residual block at 128 channels (encoder stage 2, 3 kHz) in ONE kernel:
//   out = ELU( Wsc.x + bsc + W1.ELU( W3 * ELU(x) + b3 ) + b1 )            (k3 128->64, k1 64->128, k1 shortcut)
// Replaces two windowed GEMMs whose short K loops (12 and 6 K tiles) spend a quarter of their time in tile
// prologues/epilogues (measured 94 TFLOP/s), and whose hidden activation h made an HBM round trip.
//
// Weights stationary in REGISTERS, split over the OUTPUT channels: 192 KB of weights do not fit the LDS next to the
// activations, but the register file does hold them — one wave per SIMD owns 512 registers per lane:
//   wave w holds W3 rows 16w..16w+15 (k = 384: 96 registers) and [W1|Wsc] rows 32w..32w+31 (k = 192: 96 registers),
//   both in MFMA A-fragment order, for the lifetime of the persistent workgroup.
// Activations stream through LDS in tiles of 64 time rows (+2 halo rows); every wave reads ALL row tiles of the
// activation tile (B operand) and produces its own channel slice, so nothing is exchanged between waves except
// through the two LDS buffers (x tile in, h tile between the convs). B fragments are fetched one k-group ahead
// (a single wave per SIMD has nobody else to hide the LDS latency). The next tile's rows are prefetched into
// registers while the current tile computes.
// MFMA order (k ascending, tap-major for the k3 conv, [h | x] for the tail), bias and ELU placement equal the unfused
// GEMM path: outputs are bit-identical (tests/test_acoustic_gpu.py::test_fused_res128_equals_unfused).
// (EnCodec architecture: SURVEY.md Appendix A.1.)
#include "gemm_core.h"
#include "encodec_kernels.h"

namespace at {

constexpr int R128_TT = 64;          // time rows per tile
constexpr int R128_XROWS = 66;       // row i <-> time t0 - 2 + i
constexpr int R128_LDX = 132;        // 128 + 4 pad: 16 consecutive rows x one 16-B chunk hit distinct bank groups
constexpr int R128_LDH = 68;
constexpr int R128_LDS_FLOATS = 2 * R128_XROWS * R128_LDX + R128_TT * R128_LDH;
constexpr int R128_CHUNKS = R128_XROWS * 32;   // float4 chunks of the input tile
constexpr int R128_PRE = (R128_CHUNKS + 255) / 256;

__global__ __launch_bounds__(256, 1) void seanet_res128_kernel(Res64Args a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xr = smem;                               // raw x rows
    float* Xe = Xr + R128_XROWS * R128_LDX;         // ELU(x) rows
    float* Hs = Xe + R128_XROWS * R128_LDX;         // ELU(conv3 + b3) rows
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, q = lane >> 4;
    const int L = a.L;
    const int tiles_per_clip = (L + R128_TT - 1) / R128_TT;
    const long long total_tiles = (long long)a.B * tiles_per_clip;

    // ---- weights -> registers, once per workgroup ------------------------------------------------------------------
    f4 w3[24], wt[2][12];
#pragma unroll
    for (int kg = 0; kg < 24; ++kg) w3[kg] = *reinterpret_cast<const f4*>(a.w3 + (wave * 16 + r16) * 384 + kg * 16 + q * 4);
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int kg = 0; kg < 12; ++kg) wt[n][kg] = *reinterpret_cast<const f4*>(a.wt + (wave * 32 + n * 16 + r16) * 192 + kg * 16 + q * 4);
    const f4 b3 = *reinterpret_cast<const f4*>(a.b3 + wave * 16 + q * 4);
    f4 bt[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) bt[n] = *reinterpret_cast<const f4*>(a.bt + wave * 32 + n * 16 + q * 4);

    // input staging: chunk c = tid + 256*j -> (row = c / 32, 16-B chunk = c % 32)
    f4 pre[R128_PRE];
    auto prefetch = [&](long long tile) {
        const long long b = tile / tiles_per_clip;
        const int t0 = (int)(tile - b * tiles_per_clip) * R128_TT;
        const float* xb = a.x + b * (long long)L * 128;
#pragma unroll
        for (int j = 0; j < R128_PRE; ++j) {
            int c = tid + 256 * j;
            c = c < R128_CHUNKS ? c : R128_CHUNKS - 1;
            int tau = t0 - 2 + (c >> 5);
            tau = tau < 0 ? -tau : tau;            // causal reflect padding at the clip start
            tau = tau > L - 1 ? L - 1 : tau;       // rows past the end are never stored
            pre[j] = *reinterpret_cast<const f4*>(xb + (long long)tau * 128 + (c & 31) * 4);
        }
    };
    if ((long long)blockIdx.x < total_tiles) prefetch(blockIdx.x);

    for (long long tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
        const long long b = tile / tiles_per_clip;
        const int t0 = (int)(tile - b * tiles_per_clip) * R128_TT;
        __syncthreads();   // previous tile's readers are done
#pragma unroll
        for (int j = 0; j < R128_PRE; ++j) {
            const int c = tid + 256 * j;
            if (c < R128_CHUNKS) {
                const int off = (c >> 5) * R128_LDX + (c & 31) * 4;
                const f4 v = pre[j];
                *reinterpret_cast<f4*>(Xr + off) = v;
                f4 e;
                e.x = elu1(v.x); e.y = elu1(v.y); e.z = elu1(v.z); e.w = elu1(v.w);
                *reinterpret_cast<f4*>(Xe + off) = e;
            }
        }
        __syncthreads();
        if (tile + gridDim.x < total_tiles) prefetch(tile + gridDim.x);   // flies during the MFMAs below
        __builtin_amdgcn_sched_barrier(0);
        // ---- h[:, 16w..16w+15] = ELU(conv3(ELU(x)) + b3) for the 4 row tiles: row j uses x rows j, j+1, j+2 -------------
        {
            f4 acc[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[m] = f4{0.f, 0.f, 0.f, 0.f};
            const float* xe = Xe + r16 * R128_LDX + q * 4;
            f4 xb[4], xn[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) xb[m] = *reinterpret_cast<const f4*>(xe + (m * 16) * R128_LDX);
#pragma unroll
            for (int kg = 0; kg < 24; ++kg) {
                if (kg + 1 < 24) {
                    const int tap = (kg + 1) >> 3, c16 = (kg + 1) & 7;
#pragma unroll
                    for (int m = 0; m < 4; ++m) xn[m] = *reinterpret_cast<const f4*>(xe + (m * 16 + tap) * R128_LDX + c16 * 16);
                }
                __builtin_amdgcn_sched_barrier(0);   // keep the prefetch in front of the MFMAs (one wave per SIMD)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(w3[kg][e], xb[m][e], acc[m], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < 4; ++m) xb[m] = xn[m];
            }
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const f4 v = acc[m] + b3;
                f4 o;
                o.x = elu1(v.x); o.y = elu1(v.y); o.z = elu1(v.z); o.w = elu1(v.w);
                *reinterpret_cast<f4*>(Hs + (m * 16 + r16) * R128_LDH + wave * 16 + q * 4) = o;
            }
        }
        __syncthreads();
        // ---- out[:, 32w..32w+31] = ELU([h | x] . [W1 | Wsc]^T + (b1 + bsc)): row j uses h row j and x row j + 2 ----------
        {
            f4 acc[4][2];
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[m][n] = f4{0.f, 0.f, 0.f, 0.f};
            const float* hs = Hs + r16 * R128_LDH + q * 4;
            const float* xr = Xr + (r16 + 2) * R128_LDX + q * 4;
            f4 xb[4], xn[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) xb[m] = *reinterpret_cast<const f4*>(hs + (m * 16) * R128_LDH);
#pragma unroll
            for (int kg = 0; kg < 12; ++kg) {
                if (kg + 1 < 12) {
#pragma unroll
                    for (int m = 0; m < 4; ++m)
                        xn[m] = (kg + 1) < 4 ? *reinterpret_cast<const f4*>(hs + (m * 16) * R128_LDH + (kg + 1) * 16)
                                             : *reinterpret_cast<const f4*>(xr + (m * 16) * R128_LDX + (kg + 1 - 4) * 16);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int n = 0; n < 2; ++n)
                            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wt[n][kg][e], xb[m][e], acc[m][n], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < 4; ++m) xb[m] = xn[m];
            }
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int t = t0 + m * 16 + r16;
                if (t < L) {
                    float* dst = a.out + (b * (long long)L + t) * 128 + wave * 32 + q * 4;
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        const f4 v = acc[m][n] + bt[n];
                        f4 o;
                        o.x = elu1(v.x); o.y = elu1(v.y); o.z = elu1(v.z); o.w = elu1(v.w);
                        *reinterpret_cast<f4*>(dst + n * 16) = o;
                    }
                }
            }
        }
    }
}

int launch_seanet_res128(const Res64Args& a, hipStream_t stream) {
    AT_REQUIRE(a.L >= 4 && a.B >= 1, "fused resblock needs at least 4 rows");
    const size_t lds = (size_t)R128_LDS_FLOATS * sizeof(float);
    { static LdsAttrFlags lds_attr_0; if (int rc = set_max_dynamic_lds(lds_attr_0, seanet_res128_kernel, lds)) return rc; }
    const long long tiles = (long long)a.B * ((a.L + R128_TT - 1) / R128_TT);
    const int grid = (int)(tiles < 256 ? tiles : 256);
    hipLaunchKernelGGL(seanet_res128_kernel, dim3(grid), dim3(256), lds, stream, a);
    AT_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace at
