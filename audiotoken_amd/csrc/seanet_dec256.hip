// Decoder stage 0 (256 channels at 8 T rows per clip) on the split GEMMs, as the encoder's stage-3 block (encodec.hip "chain3"): the block's k3 conv
// (256 -> 128) and its tail ([ELU(h) | u] 384 -> 256) are windowed / plain launches of gemm_f16x2_tg on two-piece fp16 operands instead of two fp32-MFMA
// GEMMs (1.54 ms for 64 clips x 10 s, 99 TFLOP/s). This file holds the one pass that feeds them from the transposed conv's fp32 output u [g][L][256]:
//   ac3  [2][g][16][Lpc][16]   ELU(u) pieces, row t at index t + 2, the two front rows = the causal conv's reflect padding (u[2], u[1])
//   at3  [2][g][24][Mpc][16]   raw u pieces in K-blocks 8..23 (the shortcut half of the tail's operand; blocks 0..7 are written by the k3 GEMM's
//                              ELU -> pieces epilogue)
// and zero-fills the rows past the data that padded GEMM tiles read (so nothing non-finite can reach a range census). A thread owns one (row, K-block):
// 64 B read, 32 contiguous bytes per piece written; a wave's 64 lanes are 64 consecutive rows of one K-block = one contiguous 2 KB run per piece.
// (EnCodec SEANet resblock: SURVEY.md Appendix A.1; reference call site audiotoken/decoder.py:66-76.)
#include "gemm_core.h"
#include "encodec_kernels.h"
#include "split_scheme.h"

namespace at {

__global__ __launch_bounds__(256) void dec_res256_split_kernel(const float* __restrict__ u, int L, _Float16* __restrict__ ac3, int Lpc, _Float16* __restrict__ at3,
                                                               int Mpc, float scale, int* __restrict__ status) {
    typedef SchemeF16x2 SC;
    typedef f16x8 V8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = gridDim.y, clip = blockIdx.y;
    const int t = blockIdx.x * 64 + lane;                        // row of this lane; rows up to max(Lpc - 2, Mpc) are visited
    const long long psA = (long long)g * 16 * Lpc * 16, psT = (long long)g * 24 * Mpc * 16;
    RangeMax over;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int kb = wave + 4 * i;                             // channels 16 kb .. 16 kb + 15
        f4 v[4];
        const bool data = t < L;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = data ? *reinterpret_cast<const f4*>(u + ((long long)clip * L + t) * 256 + kb * 16 + j * 4) : f4{0.f, 0.f, 0.f, 0.f};
        V8 raw[2][2], el[2][2];                                  // [piece][half of the 16 channels]
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            typename SC::V4 p[2];
            over |= split4<SC>(v[j], scale, p);
#pragma unroll
            for (int k = 0; k < 4; ++k) { raw[0][j >> 1][(j & 1) * 4 + k] = p[0][k]; raw[1][j >> 1][(j & 1) * 4 + k] = p[1][k]; }
            const f4 e = {elu1(v[j].x), elu1(v[j].y), elu1(v[j].z), elu1(v[j].w)};
            split4<SchemeNoCheck<SC>>(e, scale, p);             // |ELU(v)| <= max(|v|, 1)
#pragma unroll
            for (int k = 0; k < 4; ++k) { el[0][j >> 1][(j & 1) * 4 + k] = p[0][k]; el[1][j >> 1][(j & 1) * 4 + k] = p[1][k]; }
        }
        auto store = [](_Float16* d, long long ps, const V8 (&x)[2][2]) {
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) {
                *reinterpret_cast<V8*>(d + pc * ps) = x[pc][0];
                *reinterpret_cast<V8*>(d + pc * ps + 8) = x[pc][1];
            }
        };
        if (t < Mpc) store(at3 + (((long long)clip * 24 + 8 + kb) * Mpc + t) * 16, psT, raw);
        if (t + 2 < Lpc) {
            store(ac3 + (((long long)clip * 16 + kb) * Lpc + t + 2) * 16, psA, el);
            if (t == 1 || t == 2) store(ac3 + (((long long)clip * 16 + kb) * Lpc + (2 - t)) * 16, psA, el);   // reflect: index 2 - k holds time -k = u[k]
        }
    }
    range_publish(status, status ? status + 1 : nullptr, over);
}

int launch_dec_res256_split(const float* u, int g, int L, __bf16* ac3, int Lpc, __bf16* at3, int Mpc, float scale, int* status, hipStream_t stream) {
    AT_REQUIRE(u && ac3 && at3 && g >= 1 && L >= 3 && Lpc >= L + 2 && Mpc >= L, "dec_res256_split: bad arguments");
    const int rows = Lpc - 2 > Mpc ? Lpc - 2 : Mpc;
    hipLaunchKernelGGL(dec_res256_split_kernel, dim3((rows + 63) / 64, g), dim3(256), 0, stream, u, L, reinterpret_cast<_Float16*>(ac3), Lpc,
                       reinterpret_cast<_Float16*>(at3), Mpc, scale, status);
    AT_CHECK_HIP(hipGetLastError());
    return 0;
}

// zero rows [row0, row1) of every (piece, clip, K-block) plane of an operand [planes][Lp][16] of 16-bit pieces (front padding rows / rows past the data)
int launch_zero_piece_rows(__bf16* S, long long planes, int Lp, int row0, int row1, hipStream_t stream) {
    if (row1 <= row0 || planes <= 0) return 0;
    AT_CHECK_HIP(hipMemset2DAsync(reinterpret_cast<char*>(S) + (size_t)row0 * 32, (size_t)Lp * 32, 0, (size_t)(row1 - row0) * 32, (size_t)planes, stream));
    return 0;
}

}  // namespace at
