// SEANet decoder tail in ONE kernel — the mirror image of seanet_stage0.hip:
//   x (64 ch, 12 kHz, already ELU'd) -> ConvTranspose1d(64->32, k4, s2) -> residual block (ELU, k3 32->16, ELU, k1 16->32,
//   + k1 shortcut) -> ELU -> conv k7 32->1 -> waveform (24 kHz).
// As separate launches (transposed-conv GEMM, two block GEMMs, conv_last) the 24 kHz x 32-channel fp32 intermediates made
// this stage HBM-bound: 17.9 ms of a 78.6 ms decode at B = 256. Fused: 256 B in per input row, 8 B out.
// (EnCodec architecture: SURVEY.md Appendix A.1; reference call site audiotoken/decoder.py:66-76.)
//
// One workgroup (4 waves) owns 120 output samples of one clip. The transposed conv is the k = 2 GEMM of the unfused path
// (out[t][p*32 + co] = x[t-1].W[:, co, p+2] + x[t].W[:, co, p], zero left pad): 64 input rows -> 128 rows of u, exactly
// 8 MFMA row tiles, which cover the 120 outputs plus the causal halo of the k3 conv (2 rows) and the k7 conv (6 rows).
// Weights stay in registers in MFMA A-fragment order; x, u, ELU(u), h and ELU(r) live in LDS (67 KB: two workgroups per
// CU). The last conv (32 -> 1 channel) is not MFMA-shaped; it runs on the VALU in the lane/shuffle order of
// conv_last_kernel. Accumulation orders, bias and ELU placement equal the unfused kernels: outputs are bit-identical
// (tests/test_acoustic_gpu.py, option "fused_dectail"). Reflect padding at the clip start (the k3 conv pads u, the k7 conv
// pads ELU(r)) is done by mirroring two / six LDS rows in the first tile of a clip.
#include "gemm_core.h"
#include "encodec_kernels.h"

namespace at {

constexpr int DT_TO = 120;                 // output samples per tile
constexpr int DT_ROWS = 128;               // u / h / r rows per tile: row j <-> time t0 - 8 + j
constexpr int DT_XROWS = 65;               // x rows per tile: row i <-> time t0/2 - 5 + i
constexpr int DT_LDX = 68, DT_LDU = 36, DT_LDH = 20;
constexpr int DT_XR_FLOATS = DT_ROWS * DT_LDU;          // x tile (65 x 68 = 4420) and, later, ELU(r) (128 x 36 = 4608)
constexpr int DT_LDS_FLOATS = DT_XR_FLOATS + DT_ROWS * DT_LDU + (DT_ROWS + 2) * DT_LDU + DT_ROWS * DT_LDH + 7 * 32 + 64 + 16 + 32 + 4;
constexpr int DT_CHUNKS = DT_XROWS * 16;
constexpr int DT_PRE = (DT_CHUNKS + 255) / 256;

__global__ __launch_bounds__(256, 2) void seanet_dectail_kernel(DecTailArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;                                   // x rows; re-used for ELU(r) once the transposed conv is done
    float* Re = smem;
    float* U = smem + DT_XR_FLOATS;                     // u rows, raw (shortcut input)
    float* Ue = U + DT_ROWS * DT_LDU + 2 * DT_LDU;      // ELU(u) rows, two spare rows in front (row -2, -1 of the k3 window)
    float* H = Ue + DT_ROWS * DT_LDU;                   // ELU(conv3 + b3)
    float* Wl = H + DT_ROWS * DT_LDH;                   // last conv [7][32]
    float* Bu = Wl + 7 * 32;                            // biases: bu [64] | b3 [16] | bt [32] | bl [1]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, q = lane >> 4;
    const int L = a.L, Lout = 2 * L;
    const int tiles_per_clip = (Lout + DT_TO - 1) / DT_TO;
    const long long total_tiles = (long long)a.B * tiles_per_clip;

    // ---- weights -> registers / LDS, once per workgroup ------------------------------------------------------------------
    f4 wu[8], w3[6], wt[2][3];
#pragma unroll
    for (int kg = 0; kg < 8; ++kg) wu[kg] = *reinterpret_cast<const f4*>(a.wu + (wave * 16 + r16) * 128 + kg * 16 + q * 4);
#pragma unroll
    for (int kg = 0; kg < 6; ++kg) w3[kg] = *reinterpret_cast<const f4*>(a.w3 + r16 * 96 + kg * 16 + q * 4);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int kg = 0; kg < 3; ++kg) wt[nt][kg] = *reinterpret_cast<const f4*>(a.wt + (nt * 16 + r16) * 48 + kg * 16 + q * 4);
    if (tid < 224) Wl[tid] = a.wl[tid];
    if (tid < 64) Bu[tid] = a.bu[tid];
    if (tid < 16) Bu[64 + tid] = a.b3[tid];
    if (tid < 32) Bu[80 + tid] = a.bt[tid];
    if (tid == 0) Bu[112] = a.bl[0];

    f4 pre[DT_PRE];
    auto prefetch = [&](long long tile) {
        const long long b = tile / tiles_per_clip;
        const int t0 = (int)(tile - b * tiles_per_clip) * DT_TO;
        const float* xb = a.x + b * (long long)L * 64;
#pragma unroll
        for (int j = 0; j < DT_PRE; ++j) {
            int c = tid + 256 * j;
            c = c < DT_CHUNKS ? c : DT_CHUNKS - 1;
            const int tx = t0 / 2 - 5 + (c >> 4);
            const int txc = tx < 0 ? 0 : (tx > L - 1 ? L - 1 : tx);   // rows past the end only feed outputs that are never stored
            f4 v = *reinterpret_cast<const f4*>(xb + (long long)txc * 64 + (c & 15) * 4);
            if (tx < 0) v = f4{0.f, 0.f, 0.f, 0.f};                   // the transposed conv's zero left pad
            pre[j] = v;
        }
    };
    if ((long long)blockIdx.x < total_tiles) prefetch(blockIdx.x);

    for (long long tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
        const long long b = tile / tiles_per_clip;
        const int t0 = (int)(tile - b * tiles_per_clip) * DT_TO;
        __syncthreads();   // previous tile's readers are done with every buffer
#pragma unroll
        for (int j = 0; j < DT_PRE; ++j) {
            const int c = tid + 256 * j;
            if (c < DT_CHUNKS) *reinterpret_cast<f4*>(Xs + (c >> 4) * DT_LDX + (c & 15) * 4) = pre[j];
        }
        __syncthreads();
        if (tile + gridDim.x < total_tiles) prefetch(tile + gridDim.x);   // flies during the MFMAs below
        // ---- transposed conv: wave w owns columns 16w..16w+15 of the [64 t][2 x 32] output = phase w >> 1, channels
        //      16 (w & 1) ..; GEMM row m <-> t = t0/2 - 4 + m uses x rows m (tap 0: x[t-1]) and m + 1 (tap 1: x[t]) ------------
        {
            f4 acc[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[m] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kg = 0; kg < 8; ++kg) {
                const int tap = kg >> 2, c16 = kg & 3;
                f4 xb[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) xb[m] = *reinterpret_cast<const f4*>(Xs + (m * 16 + r16 + tap) * DT_LDX + c16 * 16 + q * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wu[kg][e], xb[m][e], acc[m], 0, 0, 0);
            }
            const f4 bu = *reinterpret_cast<const f4*>(Bu + wave * 16 + q * 4);
            const int ph = wave >> 1, co = (wave & 1) * 16 + q * 4;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int j = 2 * (m * 16 + r16) + ph;
                const f4 v = acc[m] + bu;
                *reinterpret_cast<f4*>(U + j * DT_LDU + co) = v;
                f4 e;
                e.x = elu1(v.x); e.y = elu1(v.y); e.z = elu1(v.z); e.w = elu1(v.w);
                *reinterpret_cast<f4*>(Ue + j * DT_LDU + co) = e;
            }
        }
        __syncthreads();   // U / Ue complete; Xs is dead from here on (Re takes its place)
        if (t0 == 0) {     // the k3 conv's reflect pad: u[-1] = u[1], u[-2] = u[2]  (row j <-> time j - 8)
            if (tid < 16) {
                const int k = 1 + (tid >> 3), c = (tid & 7) * 4;
                *reinterpret_cast<f4*>(Ue + (8 - k) * DT_LDU + c) = *reinterpret_cast<const f4*>(Ue + (8 + k) * DT_LDU + c);
            }
            __syncthreads();
        }
        // ---- block: two row tiles per wave; the h rows a wave writes are the ones it reads back (no barrier in between) ------
        {
            int row[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) row[i] = (2 * wave + i) * 16 + r16;
            f4 acc[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int kg = 0; kg < 6; ++kg) {
                const int tap = kg >> 1, c16 = kg & 1;
                f4 xb[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) xb[i] = *reinterpret_cast<const f4*>(Ue + (row[i] + tap - 2) * DT_LDU + c16 * 16 + q * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(w3[kg][e], xb[i][e], acc[i], 0, 0, 0);
            }
            const f4 b3 = *reinterpret_cast<const f4*>(Bu + 64 + q * 4);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const f4 v = acc[i] + b3;
                f4 o;
                o.x = elu1(v.x); o.y = elu1(v.y); o.z = elu1(v.z); o.w = elu1(v.w);
                *reinterpret_cast<f4*>(H + row[i] * DT_LDH + q * 4) = o;
            }
            f4 acc2[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc2[i][nt] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kg = 0; kg < 3; ++kg) {
                f4 xb[2];
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    xb[i] = kg == 0 ? *reinterpret_cast<const f4*>(H + row[i] * DT_LDH + q * 4)
                                    : *reinterpret_cast<const f4*>(U + row[i] * DT_LDU + (kg - 1) * 16 + q * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt)
                            acc2[i][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wt[nt][kg][e], xb[i][e], acc2[i][nt], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const f4 v = acc2[i][nt] + *reinterpret_cast<const f4*>(Bu + 80 + nt * 16 + q * 4);
                    f4 o;
                    o.x = elu1(v.x); o.y = elu1(v.y); o.z = elu1(v.z); o.w = elu1(v.w);
                    *reinterpret_cast<f4*>(Re + row[i] * DT_LDU + nt * 16 + q * 4) = o;
                }
        }
        __syncthreads();
        if (t0 == 0) {     // the k7 conv's reflect pad on ELU(r): r[-k] = r[k], k = 1..6
            if (tid < 48) {
                const int k = 1 + tid / 8, c = (tid & 7) * 4;
                *reinterpret_cast<f4*>(Re + (8 - k) * DT_LDU + c) = *reinterpret_cast<const f4*>(Re + (8 + k) * DT_LDU + c);
            }
            __syncthreads();
        }
        // ---- last conv (32 -> 1, k7) on the VALU, lane layout and reduction order of conv_last_kernel: 8 lanes per output,
        //      4 channels each over the 7 taps, then a 3-step shuffle reduction -------------------------------------------------
        {
            const int cg = tid & 7;
            const float bl = Bu[112];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int o = it * 32 + (tid >> 3);           // output sample t0 + o  <->  row j = 8 + o
                const int oc = o < DT_TO ? o : DT_TO - 1;
                float acc = 0.f;
#pragma unroll
                for (int tap = 0; tap < 7; ++tap) {
                    const f4 v = *reinterpret_cast<const f4*>(Re + (8 + oc - 6 + tap) * DT_LDU + cg * 4);
                    const f4 ww = *reinterpret_cast<const f4*>(Wl + tap * 32 + cg * 4);
                    acc = fmaf(v.x, ww.x, acc); acc = fmaf(v.y, ww.y, acc);
                    acc = fmaf(v.z, ww.z, acc); acc = fmaf(v.w, ww.w, acc);
                }
                acc += __shfl_xor(acc, 1);
                acc += __shfl_xor(acc, 2);
                acc += __shfl_xor(acc, 4);
                const int tout = t0 + o;
                if (cg == 0 && o < DT_TO && tout < Lout) a.out[b * (long long)Lout + tout] = acc + bl;
            }
        }
    }
}

int launch_seanet_dectail(const DecTailArgs& a, hipStream_t stream) {
    AT_REQUIRE(a.L >= 8 && a.B >= 1, "fused decoder tail needs at least 8 input rows");
    const size_t lds = (size_t)DT_LDS_FLOATS * sizeof(float);
    { static LdsAttrFlags lds_attr_0; if (int rc = set_max_dynamic_lds(lds_attr_0, seanet_dectail_kernel, lds)) return rc; }
    const long long tiles = (long long)a.B * ((2 * a.L + DT_TO - 1) / DT_TO);
    const int grid = (int)(tiles < 512 ? tiles : 512);   // two resident workgroups per CU
    hipLaunchKernelGGL(seanet_dectail_kernel, dim3(grid), dim3(256), lds, stream, a);
    AT_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace at
