// SEANet residual block at 64 channels (encoder stage 1, 12 kHz) in ONE kernel:
//   out = ELU( Wsc.x + bsc + W1.ELU( W3 * ELU(x) + b3 ) + b1 )            (k3 64->32, k1 32->64, k1 shortcut)
// Replaces two windowed GEMMs whose fp32 intermediates made the block memory-bound (63 TFLOP/s):
//   unfused: read x twice, write + read h, write out = 1024 B per row;  fused: 256 B in + 256 B out per row.
// Same scheme as seanet_res128.hip: weights stationary in registers in MFMA A-fragment order, split over the OUTPUT
// channels (conv3: wave = (channel tile w & 1, row half w >> 1), 48 registers; tail: wave = channel tile w, 24
// registers); activations stream through LDS in tiles of 64 time rows (+2 halo rows): the x tile in, the h tile
// between the two convs. 45 KB of LDS and ~130 registers put THREE workgroups on a CU, which cover each other's
// barriers, LDS latency and ELU phases (the fp32 MFMA and the VALU of a SIMD do not overlap, so ELU time is real
// time: one ELU costs 8 issue slots, 10.4 k of them per tile against 768 MFMAs). The next tile's rows are
// prefetched into registers while the current tile computes.
// MFMA order / bias / ELU placement equal the unfused GEMM path: outputs are bit-identical
// (tests/test_acoustic_gpu.py::test_fused_stage0_equals_unfused covers the "fused_res64" option).
// (EnCodec architecture: SURVEY.md Appendix A.1.)
#include "gemm_core.h"
#include "encodec_kernels.h"

namespace at {

constexpr int R64_TT = 64;           // time rows per tile
constexpr int R64_XROWS = 66;        // row i <-> time t0 - 2 + i
constexpr int R64_LDX = 68, R64_LDH = 36;
constexpr int R64_LDS_FLOATS = 2 * R64_XROWS * R64_LDX + R64_TT * R64_LDH;
constexpr int R64_CHUNKS = R64_XROWS * 16;   // float4 chunks of the input tile
constexpr int R64_PRE = (R64_CHUNKS + 255) / 256;

__global__ __launch_bounds__(256, 3) void seanet_res64_kernel(Res64Args a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xr = smem;                             // raw x rows
    float* Xe = Xr + R64_XROWS * R64_LDX;         // ELU(x) rows
    float* Hs = Xe + R64_XROWS * R64_LDX;         // ELU(conv3 + b3) rows
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, q = lane >> 4;
    const int L = a.L;
    const int tiles_per_clip = (L + R64_TT - 1) / R64_TT;
    const long long total_tiles = (long long)a.B * tiles_per_clip;
    const int cn = wave & 1, mh = wave >> 1;      // conv3: channel tile, row half

    // ---- weights -> registers, once per workgroup ------------------------------------------------------------------
    f4 w3[12], wt[6];
#pragma unroll
    for (int kg = 0; kg < 12; ++kg) w3[kg] = *reinterpret_cast<const f4*>(a.w3 + (cn * 16 + r16) * 192 + kg * 16 + q * 4);
#pragma unroll
    for (int kg = 0; kg < 6; ++kg) wt[kg] = *reinterpret_cast<const f4*>(a.wt + (wave * 16 + r16) * 96 + kg * 16 + q * 4);
    const f4 b3 = *reinterpret_cast<const f4*>(a.b3 + cn * 16 + q * 4);
    const f4 bt = *reinterpret_cast<const f4*>(a.bt + wave * 16 + q * 4);

    // input staging: chunk c = tid + 256*j -> (row = c / 16, 16-B chunk = c % 16)
    f4 pre[R64_PRE];
    auto prefetch = [&](long long tile) {
        const long long b = tile / tiles_per_clip;
        const int t0 = (int)(tile - b * tiles_per_clip) * R64_TT;
        const float* xb = a.x + b * (long long)L * 64;
#pragma unroll
        for (int j = 0; j < R64_PRE; ++j) {
            int c = tid + 256 * j;
            c = c < R64_CHUNKS ? c : R64_CHUNKS - 1;
            int tau = t0 - 2 + (c >> 4);
            tau = tau < 0 ? -tau : tau;            // causal reflect padding at the clip start
            tau = tau > L - 1 ? L - 1 : tau;       // rows past the end are never stored
            pre[j] = *reinterpret_cast<const f4*>(xb + (long long)tau * 64 + (c & 15) * 4);
        }
    };
    if ((long long)blockIdx.x < total_tiles) prefetch(blockIdx.x);

    for (long long tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
        const long long b = tile / tiles_per_clip;
        const int t0 = (int)(tile - b * tiles_per_clip) * R64_TT;
        __syncthreads();   // previous tile's readers are done
#pragma unroll
        for (int j = 0; j < R64_PRE; ++j) {
            const int c = tid + 256 * j;
            if (c < R64_CHUNKS) {
                const int off = (c >> 4) * R64_LDX + (c & 15) * 4;
                const f4 v = pre[j];
                *reinterpret_cast<f4*>(Xr + off) = v;
                f4 e;
                e.x = elu1(v.x); e.y = elu1(v.y); e.z = elu1(v.z); e.w = elu1(v.w);
                *reinterpret_cast<f4*>(Xe + off) = e;
            }
        }
        __syncthreads();
        if (tile + gridDim.x < total_tiles) prefetch(tile + gridDim.x);   // flies during the MFMAs below
        // ---- h[rows of half mh][16cn..16cn+15] = ELU(conv3(ELU(x)) + b3): row j uses x rows j, j+1, j+2 -------------------
        {
            f4 acc[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
            const float* xe = Xe + (mh * 32 + r16) * R64_LDX + q * 4;
#pragma unroll
            for (int kg = 0; kg < 12; ++kg) {
                const int tap = kg >> 2, c16 = kg & 3;
                f4 xb[2];
#pragma unroll
                for (int m = 0; m < 2; ++m) xb[m] = *reinterpret_cast<const f4*>(xe + (m * 16 + tap) * R64_LDX + c16 * 16);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int m = 0; m < 2; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(w3[kg][e], xb[m][e], acc[m], 0, 0, 0);
            }
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const f4 v = acc[m] + b3;
                f4 o;
                o.x = elu1(v.x); o.y = elu1(v.y); o.z = elu1(v.z); o.w = elu1(v.w);
                *reinterpret_cast<f4*>(Hs + (mh * 32 + m * 16 + r16) * R64_LDH + cn * 16 + q * 4) = o;
            }
        }
        __syncthreads();
        // ---- out[:, 16w..16w+15] = ELU([h | x] . [W1 | Wsc]^T + (b1 + bsc)): row j uses h row j and x row j + 2 ------------
        {
            f4 acc[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[m] = f4{0.f, 0.f, 0.f, 0.f};
            const float* hs = Hs + r16 * R64_LDH + q * 4;
            const float* xr = Xr + (r16 + 2) * R64_LDX + q * 4;
#pragma unroll
            for (int kg = 0; kg < 6; ++kg) {
                f4 xb[4];
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    xb[m] = kg < 2 ? *reinterpret_cast<const f4*>(hs + (m * 16) * R64_LDH + kg * 16)
                                   : *reinterpret_cast<const f4*>(xr + (m * 16) * R64_LDX + (kg - 2) * 16);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wt[kg][e], xb[m][e], acc[m], 0, 0, 0);
            }
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int t = t0 + m * 16 + r16;
                if (t < L) {
                    const f4 v = acc[m] + bt;
                    f4 o;
                    o.x = elu1(v.x); o.y = elu1(v.y); o.z = elu1(v.z); o.w = elu1(v.w);
                    *reinterpret_cast<f4*>(a.out + (b * (long long)L + t) * 64 + wave * 16 + q * 4) = o;
                }
            }
        }
    }
}

int launch_seanet_res64(const Res64Args& a, hipStream_t stream) {
    AT_REQUIRE(a.L >= 4 && a.B >= 1, "fused resblock needs at least 4 rows");
    const size_t lds = (size_t)R64_LDS_FLOATS * sizeof(float);
    { static LdsAttrFlags lds_attr_0; if (int rc = set_max_dynamic_lds(lds_attr_0, seanet_res64_kernel, lds)) return rc; }
    const long long tiles = (long long)a.B * ((a.L + R64_TT - 1) / R64_TT);
    const int grid = (int)(tiles < 768 ? tiles : 768);   // three resident workgroups per CU
    hipLaunchKernelGGL(seanet_res64_kernel, dim3(grid), dim3(256), lds, stream, a);
    AT_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace at
