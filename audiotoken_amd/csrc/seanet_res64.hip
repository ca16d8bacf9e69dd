// SEANet residual block at 64 channels (encoder stage 1, 12 kHz) in ONE kernel:
//   out = ELU( Wsc.x + bsc + W1.ELU( W3 * ELU(x) + b3 ) + b1 )            (k3 64->32, k1 32->64, k1 shortcut)
// Replaces two windowed GEMMs whose fp32 intermediates made the block memory-bound (63 TFLOP/s):
//   unfused: read x twice, write + read h, write out = 1024 B per row;  fused: 256 B in + 256 B out per row.
// Same structure as seanet_stage0.hip: weights stationary in registers (MFMA A-fragment order), activations through
// padded LDS rows, one 128-row tile per workgroup iteration, persistent grid, next tile's rows prefetched into registers
// while the current tile computes. MFMA order / bias / ELU placement equal the unfused kernels (bit-identical output).
// (EnCodec architecture: SURVEY.md Appendix A.1.)
#include "gemm_core.h"
#include "encodec_kernels.h"

namespace at {

constexpr int R64_TT = 128;
constexpr int R64_XROWS = 132;     // rows i <-> time t0 - 2 + i, 130 used
constexpr int R64_LDX = 68, R64_LDH = 36;
constexpr int R64_LDW = 100;         // tail weight rows in LDS (96 + 4 pad)
constexpr int R64_LDS_FLOATS = 2 * R64_XROWS * R64_LDX + R64_TT * R64_LDH + 64 * R64_LDW;
constexpr int R64_CHUNKS = 130 * 16;   // float4 chunks of the input tile

__global__ __launch_bounds__(512, 1) void seanet_res64_kernel(Res64Args a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xr = smem;
    float* Xe = Xr + R64_XROWS * R64_LDX;
    float* Hs = Xe + R64_XROWS * R64_LDX;
    float* Wts = Hs + R64_TT * R64_LDH;   // [64][100]: [W1 | Wsc] rows (register budget is spent on W3 and the prefetch)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, q = lane >> 4;
    const int L = a.L;
    const int tiles_per_clip = (L + R64_TT - 1) / R64_TT;
    const long long total_tiles = (long long)a.B * tiles_per_clip;

    f4 w3[2][12];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int kg = 0; kg < 12; ++kg) w3[nt][kg] = *reinterpret_cast<const f4*>(a.w3 + (nt * 16 + r16) * 192 + kg * 16 + q * 4);
    for (int c = tid; c < 64 * 24; c += 512)
        *reinterpret_cast<f4*>(Wts + (c / 24) * R64_LDW + (c % 24) * 4) = *reinterpret_cast<const f4*>(a.wt + (c / 24) * 96 + (c % 24) * 4);
    f4 b3[2], bt[4];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) b3[nt] = *reinterpret_cast<const f4*>(a.b3 + nt * 16 + q * 4);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) bt[nt] = *reinterpret_cast<const f4*>(a.bt + nt * 16 + q * 4);

    // input staging: chunk c = tid + 512*j -> (row = c / 16, 16-B chunk = c % 16)
    f4 pre[5];
    auto prefetch = [&](long long tile) {
        const long long b = tile / tiles_per_clip;
        const int t0 = (int)(tile - b * tiles_per_clip) * R64_TT;
        const float* xb = a.x + b * (long long)L * 64;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            int c = tid + 512 * j;
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
        for (int j = 0; j < 5; ++j) {
            const int c = tid + 512 * j;
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
        // ---- h = ELU(conv3(ELU(x)) + b3): output row j uses x rows j, j+1, j+2 (wave = m-tile) ---------------------
        const int row = wave * 16 + r16;
        {
            f4 acc[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
            const float* xr = Xe + row * R64_LDX + q * 4;
#pragma unroll
            for (int kg = 0; kg < 12; ++kg) {
                const f4 xb = *reinterpret_cast<const f4*>(xr + (kg >> 2) * R64_LDX + (kg & 3) * 16);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w3[nt][kg][e], xb[e], acc[nt], 0, 0, 0);
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const f4 v = acc[nt] + b3[nt];
                f4 o;
                o.x = elu1(v.x); o.y = elu1(v.y); o.z = elu1(v.z); o.w = elu1(v.w);
                *reinterpret_cast<f4*>(Hs + row * R64_LDH + nt * 16 + q * 4) = o;
            }
        }
        __syncthreads();
        // ---- out = ELU([ELU(h) | x] . [W1 | Wsc]^T + (b1 + bsc)): row j uses h row j and x row j + 2 -----------------------
        {
            f4 acc[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[nt] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kg = 0; kg < 6; ++kg) {
                const f4 xb = kg < 2 ? *reinterpret_cast<const f4*>(Hs + row * R64_LDH + kg * 16 + q * 4)
                                     : *reinterpret_cast<const f4*>(Xr + (row + 2) * R64_LDX + (kg - 2) * 16 + q * 4);
                f4 wa[4];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) wa[nt] = *reinterpret_cast<const f4*>(Wts + (nt * 16 + r16) * R64_LDW + kg * 16 + q * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[nt][e], xb[e], acc[nt], 0, 0, 0);
            }
            const int t = t0 + row;
            if (t < L) {
                float* dst = a.out + (b * (long long)L + t) * 64 + q * 4;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const f4 v = acc[nt] + bt[nt];
                    f4 o;
                    o.x = elu1(v.x); o.y = elu1(v.y); o.z = elu1(v.z); o.w = elu1(v.w);
                    *reinterpret_cast<f4*>(dst + nt * 16) = o;
                }
            }
        }
    }
}

int launch_seanet_res64(const Res64Args& a, hipStream_t stream) {
    AT_REQUIRE(a.L >= 4 && a.B >= 1, "fused resblock needs at least 4 rows");
    const size_t lds = (size_t)R64_LDS_FLOATS * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        AT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(seanet_res64_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    const long long tiles = (long long)a.B * ((a.L + R64_TT - 1) / R64_TT);
    const int grid = (int)(tiles < 256 ? tiles : 256);
    hipLaunchKernelGGL(seanet_res64_kernel, dim3(grid), dim3(512), lds, stream, a);
    AT_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace at
