// fp32 windowed GEMM on the CDNA4 f32 matrix cores (v_mfma_f32_16x16x4_f32): the dense contraction behind
// every conv1d / transposed conv1d / Linear of the tokenizers (see GemmArgs in at_common.h).
//
// Why fp32 MFMA: token ids must equal the reference's fp32 CPU result; the f32-input MFMA is an exact
// k-ordered fmaf chain (MI355X guide §3 "FP32-input MFMA") at the fp32 vector peak (157 TFLOP/s) and leaves
// the VALU free for the ELU prologue / epilogues.
//
// Tiling (wave64, 4 waves per workgroup): block tile BM x BN, K step 32. Both operand tiles are staged
// global -> registers -> LDS (register staging because the A tile needs the ELU prologue and reflected rows),
// double-buffered so tile k+1's global loads fly during tile k's MFMAs, one barrier per K step.
// LDS image: [row][32 floats], 16-byte chunk index XOR ((row>>1)&7) — conflict-free for the ds_read_b128
// fragment reads (16 rows x same chunk) and for the ds_write_b128 staging writes (8 chunks of one row).
// Fragment trick: one ds_read_b128 gives a lane 4 consecutive k; MFMA step j uses element j, so the four
// lane-quads cover 16 distinct k per four MFMAs — any k permutation is legal as long as A and B agree.
// The MFMA is issued "swapped" (weights as the A operand, activations as B) so each lane ends up with 4
// consecutive output channels of one output row: the epilogue is one float4 load (bias / residual) and one
// float4 store per 16x16 tile.
#include "gemm_core.h"
#include <cstdlib>

namespace at {

__device__ __forceinline__ f4 apply_act4(f4 v, int epi) {
    if (epi == EPI_SWISH) {
        // x * sigmoid(x) with v_exp_f32 / v_rcp_f32 (each ~1 ulp): the FFN epilogue evaluates 64 of these per lane per tile
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = v[k] * sigmoidf_(v[k]);
    } else if (epi == EPI_ELU) {
        v.x = elu1(v.x); v.y = elu1(v.y); v.z = elu1(v.z); v.w = elu1(v.w);
    } else if (epi == EPI_GELU) {
        v.x = gelu_erf(v.x);
        v.y = gelu_erf(v.y);
        v.z = gelu_erf(v.z);
        v.w = gelu_erf(v.w);
    } else if (epi == EPI_LOGFLOOR) {
        const float fl = 1.192092955078125e-07f;
        v.x = logf(fmaxf(v.x, fl)); v.y = logf(fmaxf(v.y, fl)); v.z = logf(fmaxf(v.z, fl)); v.w = logf(fmaxf(v.w, fl));
    }
    return v;
}

template <int BM, int BN, int WM, int WN, int PRO, int BKT = 32, int MINB = 1>
__global__ __launch_bounds__(256, MINB) void gemm_f32_kernel(GemmArgs a) {
    using Tile = GemmTile<BM, BN, WM, WN, PRO, BKT>;
    constexpr int TM = Tile::TM, TN = Tile::TN;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // 1-D grid, n-tile fastest: the blocks in flight cover few m-tiles x all n-tiles, so the activation panel is
    // fetched from HBM about once and, with the round-robin XCD dispatch, each XCD's L2 keeps N/8 of the weights.
    const int nt = (a.N + BN - 1) / BN;
    const int m0 = (blockIdx.x / nt) * BM, n0 = (blockIdx.x % nt) * BN, b = blockIdx.z;
    f4 acc[TM][TN];
    Tile::run(a, smem, m0, n0, b, acc);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int r16 = lane & 15, q = lane >> 4;
    float* Cb = a.C + (long long)b * a.c_bstride;
    const float* Rb = a.R ? a.R + (long long)b * a.r_bstride : nullptr;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm * TM * 16 + i * 16 + r16;
        if (m >= a.M) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * TN * 16 + j * 16 + q * 4;
            if (n >= a.N) continue;
            f4 v = acc[i][j];
            if (a.bias) v += *reinterpret_cast<const f4*>(a.bias + n);
            if (a.epi == EPI_GLU) {
                // interleaved rows: (v.x, v.y) = (a_c, b_c), (v.z, v.w) = (a_{c+1}, b_{c+1})
                float2 o;
                o.x = v.x * sigmoidf_(v.y);
                o.y = v.z * sigmoidf_(v.w);
                if (a.row_mask && a.row_mask[(long long)b * a.M + m] == 0.f) o = float2{0.f, 0.f};
                *reinterpret_cast<float2*>(Cb + (long long)m * a.ldc + (n >> 1)) = o;
                continue;
            }
            v = apply_act4(v, a.epi);
            v *= a.alpha;
            if (Rb) v += *reinterpret_cast<const f4*>(Rb + (long long)m * a.ldr + n);
            if (a.row_mask && a.row_mask[(long long)b * a.M + m] == 0.f) v = f4{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f4*>(Cb + (long long)m * a.ldc + n) = v;
        }
    }
}

template <int BM, int BN, int WM, int WN, int BKT = 32>
static int launch_cfg(const GemmArgs& a, hipStream_t stream) {
    using Tile = GemmTile<BM, BN, WM, WN, PRO_NONE, BKT>;
    dim3 grid(((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN), 1, a.batch);
    if (a.pro == PRO_ELU)
        hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, PRO_ELU, BKT>), grid, dim3(256), Tile::LDS_BYTES, stream, a);
    else if (a.pro == PRO_POWER)
        hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, PRO_POWER, BKT>), grid, dim3(256), Tile::LDS_BYTES, stream, a);
    else
        hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, PRO_NONE, BKT>), grid, dim3(256), Tile::LDS_BYTES, stream, a);
    AT_CHECK_HIP(hipGetLastError());
    return 0;
}

int check_gemm_args(const GemmArgs& a) {
    AT_REQUIRE(a.X2 ? (a.ktaps == 1 && a.K1 == a.Cin && a.K1 % 4 == 0 && a.K > a.K1 && a.ld2 % 4 == 0 && a.ld2 >= a.K - a.K1)
                    : a.K == a.ktaps * a.Cin, "K must equal ktaps*Cin (or K1 + width of the second source)");
    AT_REQUIRE(a.Cin % 4 == 0 && a.N % 4 == 0, "Cin and N must be multiples of 4");
    AT_REQUIRE(a.ldx % 4 == 0 && a.ldx >= 4, "ldx must be a positive multiple of 4");
    AT_REQUIRE((a.ldc % 4 == 0 || (a.epi == EPI_GLU && a.ldc % 2 == 0)) && (a.R == nullptr || a.ldr % 4 == 0), "ldc/ldr must be multiples of 4");
    AT_REQUIRE(a.pro != PRO_POWER || (a.aux_off % 4 == 0 && a.ktaps == 1), "PRO_POWER needs ktaps == 1 and an aligned aux_off");
    AT_REQUIRE(a.batch >= 1 && a.batch <= 65535, "batch out of range");
    AT_REQUIRE(a.pad_mode == 0 || a.Tin > a.pad_left, "reflect padding needs Tin > pad");
    AT_REQUIRE(a.pro != PRO_POWER || a.aux_off + a.Cin <= a.ldx, "PRO_POWER: imaginary half must lie inside the input row");
    return 0;
}

int launch_gemm(const GemmArgs& a, hipStream_t stream) {
    if (int rc = check_gemm_args(a)) return rc;
    if (a.M <= 0 || a.N <= 0) return 0;
    if (a.N <= 16) return launch_cfg<128, 16, 4, 1>(a, stream);
    if (a.N <= 32) return launch_cfg<128, 32, 4, 1>(a, stream);
    if (a.N <= 64) {
        // grouped convs (HuBERT positional conv: 48 channels per group): a K tile of 16 keeps every tile inside one tap (TAP body)
        if (a.ktaps > 1 && a.Cin % 32 != 0 && a.Cin % 16 == 0 && a.K % 16 == 0) return launch_cfg<128, 64, 4, 1, 16>(a, stream);
        return launch_cfg<128, 64, 4, 1>(a, stream);
    }
    // a launch that cannot fill the 512 tile slots (2 per CU) with 128 x 128 tiles — single-clip latency paths such as
    // ffn2 of one 30 s clip: 12 x 8 tiles with K = 4096 — runs 4x as many 64 x 64 tiles instead (same k order, same results)
    const long long tiles128 = (long long)((a.M + 127) / 128) * ((a.N + 127) / 128) * a.batch;
    if ((long long)a.M * a.batch <= 1024 || tiles128 < 256) return launch_cfg<64, 64, 2, 2>(a, stream);
    // K tile of 16 for the big tiles: 32 KB of LDS per workgroup and a shorter barrier-to-barrier section measured
    // 4-5 % faster than K tile 32 on the conformer shapes (ffn2 133 -> 139 TFLOP/s); K % 16 != 0 keeps the 32 path
    static const int bk16 = getenv("AUDIOTOKEN_GEMM_BK16") ? atoi(getenv("AUDIOTOKEN_GEMM_BK16")) : 1;
    if (bk16 && a.pro == PRO_NONE && a.K % 16 == 0) return launch_cfg<128, 128, 2, 2, 16>(a, stream);   // (neutral on the ELU-prologue conv shapes)
    return launch_cfg<128, 128, 2, 2>(a, stream);
}

}  // namespace at
