// SEANet encoder stage 0 in ONE kernel: waveform -> conv0 (1->32, k7) -> residual block (ELU, k3 32->16, ELU,
// k1 16->32, + k1 shortcut) -> ELU -> strided conv (32->64, k4 s2). Replaces four launches (conv0, two resblock
// GEMMs, down0 GEMM) whose 24 kHz x 32-channel fp32 intermediates made the stage HBM-bound:
//   unfused traffic  : 4 + 128 (conv0) + 128+64 (conv3) + 64+128+128 (tail) + 128+128 (down0) = 900 B per input sample
//   fused traffic    : 4 B in + 128 B out per input sample (the algorithmic minimum for this stage boundary)
// (EnCodec architecture: SURVEY.md Appendix A.1; reference call site audiotoken/encoder.py:48.)
//
// One workgroup (4 waves) owns a tile of 126 input samples (63 outputs) of one clip: the strided conv then needs exactly
// 128 rows of the block output = 8 MFMA row tiles, two per wave, no ragged ninth tile. Every intermediate lives in
// LDS, time-major with padded rows (36 / 20 floats) so the MFMA fragment reads (16 consecutive time rows, same
// 16-byte chunk) spread over the banks. All conv weights stay in REGISTERS in MFMA A-fragment order for the lifetime
// of the (persistent) workgroup: "weights stationary", activations stream through LDS. 68 KB of LDS and 4 waves per
// workgroup put TWO workgroups on a CU: while one is in its VALU phase (conv0 taps, ELUs) or waits at a barrier the
// other one feeds the MFMA pipe (one 8-wave workgroup in lock-step phases measured 47 % MFMA utilisation).
// MFMA accumulation order, bias/ELU placement and the conv0 tap order are identical to the unfused kernels, so fused
// and unfused outputs are bit-identical (tests/test_acoustic_gpu.py::test_fused_stage0_equals_unfused).
// Causal reflect padding at the clip start is handled by evaluating conv0 at |t| and mirroring two rows of the block
// output; requires N % 2 == 0 so the strided conv needs no right "extra" padding (else the unfused path is used).
#include "gemm_core.h"
#include "encodec_kernels.h"
#include <type_traits>

namespace at {

constexpr int S0_ADV = 126;                // input samples per tile (tile advance)
constexpr int S0_UO = 63;                  // outputs per tile
constexpr int S0_XROWS = 144;              // x0 buffer rows (130 used, 9 MFMA row tiles written): row i <-> time t0 - 4 + i
constexpr int S0_ROWS = 128;               // h / r rows: row j <-> time t0 - 2 + j (8 m-tiles)
constexpr int S0_RALLOC = 132;             // r rows allocated (the masked 64th output reads rows 126..129)
constexpr int S0_LDX = 36, S0_LDH = 20, S0_LDR = 36;
constexpr int S0_WAV = 144;                // waveform segment (136 used)
constexpr int S0_LDS_FLOATS = 2 * S0_XROWS * S0_LDX + S0_ROWS * S0_LDH + S0_RALLOC * S0_LDR + S0_WAV + 32 + 112;

__global__ __launch_bounds__(256, 2) void seanet_stage0_kernel(Stage0Args a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* X0r = smem;                               // conv0 output, raw
    float* X0e = X0r + S0_XROWS * S0_LDX;            // ELU(conv0 output)
    float* Hs = X0e + S0_XROWS * S0_LDX;             // ELU(conv3 output)
    float* Rs = Hs + S0_ROWS * S0_LDH;               // ELU(block output)
    float* Wv = Rs + S0_RALLOC * S0_LDR;             // waveform segment: Wv[s] = wav[|t0 - 10 + s|]
    float* B0s = Wv + S0_WAV;                        // conv0 bias [32]
    float* Bs = B0s + 32;                            // biases: b3 [16] | bt [32] | bd [64] (read at epilogue time: registers are full of weights)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, q = lane >> 4;
    const int N = a.N, L1 = N / 2;
    const int tiles_per_clip = (N + S0_ADV - 1) / S0_ADV;
    const long long total_tiles = (long long)a.B * tiles_per_clip;

    // ---- weights -> registers, once per workgroup ------------------------------------------------------------------
    if (tid < 32) B0s[tid] = a.b0[tid];
    // conv0 as a K = 8 MFMA (7 taps + a zero column): A fragment w0f[nt][s] = W0[nt*16 + r16][4s + q]
    float w0f[2][2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) w0f[nt][ks] = (4 * ks + q) < 7 ? a.w0[(nt * 16 + r16) * 7 + 4 * ks + q] : 0.f;
    if (tid < 16) Bs[tid] = a.b3[tid];
    if (tid < 32) Bs[16 + tid] = a.bt[tid];
    if (tid < 64) Bs[48 + tid] = a.bd[tid];
    f4 w3[6], wt[2][3], wd[4][8];
#pragma unroll
    for (int kg = 0; kg < 6; ++kg) w3[kg] = *reinterpret_cast<const f4*>(a.w3 + r16 * 96 + kg * 16 + q * 4);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int kg = 0; kg < 3; ++kg) wt[nt][kg] = *reinterpret_cast<const f4*>(a.wt + (nt * 16 + r16) * 48 + kg * 16 + q * 4);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int kg = 0; kg < 8; ++kg) wd[nt][kg] = *reinterpret_cast<const f4*>(a.wd + (nt * 16 + r16) * 128 + kg * 16 + q * 4);

    // waveform sample of the first tile; later tiles are fetched one tile ahead (the load flies during the MFMA phases)
    auto fetch_wav = [&](long long tile) -> float {
        if (tile >= total_tiles || tid >= S0_WAV) return 0.f;
        const long long b = tile / tiles_per_clip;
        const int t0 = (int)(tile - b * tiles_per_clip) * S0_ADV;
        int w = t0 - 10 + tid;
        w = w < 0 ? -w : w;
        w = w > N - 1 ? N - 1 : w;
        return a.wav[b * N + w];
    };
    float wnext = fetch_wav(blockIdx.x);

    for (long long tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
        const long long b = tile / tiles_per_clip;
        const int t0 = (int)(tile - b * tiles_per_clip) * S0_ADV;
        __syncthreads();   // previous tile's readers are done with every buffer
        // ---- A: waveform segment ----------------------------------------------------------------------------------
        if (tid < S0_WAV) Wv[tid] = wnext;
        wnext = fetch_wav(tile + gridDim.x);
        __syncthreads();
        // ---- B: conv0 at time |t0 - 4 + i| -> raw and ELU copies, on the MFMA (K = 7 taps + a zero column = 2 k-steps;
        //      the k order 0..6 is the tap order of the scalar conv0 kernel). 9 row tiles cover the 130 rows needed. Only
        //      the first tile of a clip needs the reflect index map; the rest reads Wv[i + tap] ----------------------------
        for (int mt = wave; mt < 9; mt += 4) {
            const int i = mt * 16 + r16;
            int a0, a1;
            if (t0 == 0) {
                int tau = i - 4;
                tau = tau < 0 ? -tau : tau;
                a0 = tau + q - 6;
                a0 = (a0 < 0 ? -a0 : a0) + 10;
                a1 = tau + q - 2;
                a1 = (a1 < 0 ? -a1 : a1) + 10;
            } else {
                a0 = i + q;
                a1 = i + 4 + q;
            }
            a0 = a0 < S0_WAV - 1 ? a0 : S0_WAV - 1;   // rows >= 130 and the zero tap stay inside the (finite) segment
            a1 = a1 < S0_WAV - 1 ? a1 : S0_WAV - 1;
            const float x0v = Wv[a0], x1v = Wv[a1];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                f4 acc = {0.f, 0.f, 0.f, 0.f};
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w0f[nt][0], x0v, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w0f[nt][1], x1v, acc, 0, 0, 0);
                const f4 o = acc + *reinterpret_cast<const f4*>(B0s + nt * 16 + q * 4);
                *reinterpret_cast<f4*>(X0r + i * S0_LDX + nt * 16 + q * 4) = o;
                f4 e;
                e.x = elu1(o.x); e.y = elu1(o.y); e.z = elu1(o.z); e.w = elu1(o.w);
                *reinterpret_cast<f4*>(X0e + i * S0_LDX + nt * 16 + q * 4) = e;
            }
        }
        __syncthreads();
        // ---- C + D per row tile (two per wave; the h rows a wave writes are the ones it reads back, so no workgroup
        //      barrier in between): h = ELU(conv3(ELU(x0)) + b3), row j uses x0 rows j..j+2;
        //      r = ELU([h | x0] . [W1 | Wsc]^T + (b1 + bsc)), row j uses h row j and raw x0 row j + 2 ---------------------
        {
            // both row tiles of the wave advance together: two (C) / four (D) independent accumulator chains
            const int row[2] = {wave * 16 + r16, (wave + 4) * 16 + r16};
            f4 acc[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int kg = 0; kg < 6; ++kg) {
                f4 xb[2];
#pragma unroll
                for (int m = 0; m < 2; ++m)
                    xb[m] = *reinterpret_cast<const f4*>(X0e + (row[m] + (kg >> 1)) * S0_LDX + (kg & 1) * 16 + q * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int m = 0; m < 2; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(w3[kg][e], xb[m][e], acc[m], 0, 0, 0);
            }
            const f4 b3 = *reinterpret_cast<const f4*>(Bs + q * 4);
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const f4 v = acc[m] + b3;
                f4 o;
                o.x = elu1(v.x); o.y = elu1(v.y); o.z = elu1(v.z); o.w = elu1(v.w);
                *reinterpret_cast<f4*>(Hs + row[m] * S0_LDH + q * 4) = o;
            }
            f4 acc2[2][2];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc2[m][nt] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kg = 0; kg < 3; ++kg) {
                f4 xb[2];
#pragma unroll
                for (int m = 0; m < 2; ++m)
                    xb[m] = kg == 0 ? *reinterpret_cast<const f4*>(Hs + row[m] * S0_LDH + q * 4)
                                    : *reinterpret_cast<const f4*>(X0r + (row[m] + 2) * S0_LDX + (kg - 1) * 16 + q * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt)
                            acc2[m][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wt[nt][kg][e], xb[m][e], acc2[m][nt], 0, 0, 0);
            }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const f4 v = acc2[m][nt] + *reinterpret_cast<const f4*>(Bs + 16 + nt * 16 + q * 4);
                    f4 o2;
                    o2.x = elu1(v.x); o2.y = elu1(v.y); o2.z = elu1(v.z); o2.w = elu1(v.w);
                    *reinterpret_cast<f4*>(Rs + row[m] * S0_LDR + nt * 16 + q * 4) = o2;
                }
        }
        __syncthreads();
        if (t0 == 0) {   // reflect padding of the strided conv's input at the clip start: r[-1] = r[1], r[-2] = r[2]
            if (tid < 16) {
                const int j = tid >> 3, c = (tid & 7) * 4;   // j = 0 <-> t = -2 (copy of t = 2, row 4); j = 1 <-> t = -1 (row 3)
                *reinterpret_cast<f4*>(Rs + j * S0_LDR + c) = *reinterpret_cast<const f4*>(Rs + (4 - j) * S0_LDR + c);
            }
            __syncthreads();
        }
        // ---- E: x1[u] = down0(ELU(r)): output u uses r rows 2u .. 2u+3 ---------------------------------------------------
        {
            const int u = wave * 16 + r16;
            f4 acc[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[nt] = f4{0.f, 0.f, 0.f, 0.f};
            const float* rr = Rs + (2 * u) * S0_LDR + q * 4;
#pragma unroll
            for (int kg = 0; kg < 8; ++kg) {
                const f4 xb = *reinterpret_cast<const f4*>(rr + (kg >> 1) * S0_LDR + (kg & 1) * 16);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wd[nt][kg][e], xb[e], acc[nt], 0, 0, 0);
            }
            const int tout = t0 / 2 + u;
            if (u < S0_UO && tout < L1) {
                float* dst = a.x1 + (b * L1 + tout) * 64 + q * 4;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) *reinterpret_cast<f4*>(dst + nt * 16) = acc[nt] + *reinterpret_cast<const f4*>(Bs + 48 + nt * 16 + q * 4);
            }
        }
    }
}

int launch_seanet_stage0(const Stage0Args& a, hipStream_t stream) {
    AT_REQUIRE(a.N % 2 == 0 && a.N >= 16 && a.B >= 1, "fused stage 0 needs an even sample count");
    const size_t lds = (size_t)S0_LDS_FLOATS * sizeof(float);
    { static LdsAttrFlags lds_attr_0; if (int rc = set_max_dynamic_lds(lds_attr_0, seanet_stage0_kernel, lds)) return rc; }
    const long long tiles = (long long)a.B * ((a.N + S0_ADV - 1) / S0_ADV);
    const int grid = (int)(tiles < 512 ? tiles : 512);   // two resident workgroups per CU
    hipLaunchKernelGGL(seanet_stage0_kernel, dim3(grid), dim3(256), lds, stream, a);
    AT_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace at
