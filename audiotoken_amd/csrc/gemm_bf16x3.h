// Split-bf16 linear layers (gemm_bf16x3.hip): C = A . W^T with fp32-class accuracy on the bf16 matrix cores.
#pragma once
#include "at_common.h"

namespace at {

// Operand schemes. Both write an fp32 operand as a sum of 16-bit pieces and accumulate the leading cross products on the 16-bit
// matrix cores into fp32:
//   XB_SCHEME_BF16X3  a = a1 + a2 + a3 (bf16, 8 + 8 + 8 bits, exact), six products, any magnitude (bf16 has the fp32 exponent range);
//   XB_SCHEME_F16X2   a * s = hi + lo (fp16, 11 + 11 bits + the sign of lo), three products hi.lo + lo.hi + hi.hi; the operand is
//                     pre-scaled by a power of two s so that lo stays in fp16's range (weights: per tensor, max |w s| in [2^14, 2^15);
//                     activations: XB_F16_ACT_SCALE) and the accumulator is multiplied by acc_scale = 1 / (s_a s_w) — all exact.
//                     Measured against float64 (tools/f16x2_gemm.hip): rms error 1.10e-7 (K = 1024) / 1.39e-6 (K = 4096) vs 1.51e-7 /
//                     1.96e-6 for bf16x3 and 1.78e-7 / 2.25e-6 for the k-ordered fp32 FMA chain, at half the MFMAs of bf16x3.
//                     Values with |x s| > 65504 do not fit: the split writers raise bit XB_STATUS_F16_OVERFLOW in *status.
enum { XB_SCHEME_BF16X3 = 0, XB_SCHEME_F16X2 = 1 };
constexpr float XB_F16_ACT_SCALE = 16.0f;
constexpr int XB_STATUS_F16_OVERFLOW = 2;
// a NaN / infinity reached a quantiser (RVQ, VQ, k-means): the range bookkeeping's fmaxf drops NaNs, so a NaN that does not descend from a flagged overflow
// (a NaN in the caller's waveform) is caught where every activation ends up — |x|^2 of the vectors that are quantised
constexpr int XB_STATUS_NONFINITE = 4;
// 16-bit storage of one operand piece (bf16 bits or fp16 bits, by scheme)
typedef __bf16 piece_t;
inline int xb_pieces(int scheme) { return scheme == XB_SCHEME_F16X2 ? 2 : 3; }
// power-of-two scale that puts max |w| into [2^14, 2^15) (1 for an all-zero tensor)
float xb_weight_scale(float max_abs);
// The provable activation scale of a LayerNorm-fed split site (round 5). A normalised row has Euclidean norm <= sqrt(D), so |LN(x)_k| <= sqrt(D) |gamma_k| +
// |beta_k| for ANY input: with gmax = max |gamma|, bmax = max |beta| (recorded at upload; carried by the packed model) the largest power of two s with
// s (sqrt(D) gmax + bmax) <= 65 000 cannot overflow fp16 (65 504) whatever the data. A site keeps XB_F16_ACT_SCALE (16) unless that bound forbids it: the
// scale of a power-of-two split only matters at the two ends of the fp16 range, so ordinary checkpoints compute exactly what they did before. swish(LN(.))
// and gelu(LN(.)) obey the same bound (|act(y)| <= |y|).
float xb_ln_site_scale(float gmax, float bmax, int D);

enum { XB_EPI_LINEAR = 0, XB_EPI_SWISH_SPLIT = 1, XB_EPI_GLU = 2, XB_EPI_GELU_SPLIT = 3, XB_EPI_GELU = 4, XB_EPI_ELU_SPLIT = 5, XB_EPI_RAW_ELU_SPLIT2 = 6,
       XB_EPI_QKV = 7 };

struct Bf16x3Args {
    const __bf16* A = nullptr;   // activations: 3 K-blocked pieces [3][K/16][Mpad][16]
    const __bf16* W = nullptr;   // weights:     3 K-blocked pieces [3][K/16][N][16]
    const float* bias = nullptr;
    int M = 0, N = 0, K = 0, Mpad = 0;
    int epi = XB_EPI_LINEAR;
    int scheme = XB_SCHEME_BF16X3;
    float acc_scale = 1.0f;     // multiplies the accumulator before bias / activation (XB_SCHEME_F16X2: 1 / (s_a s_w))
    float split_scale = 1.0f;   // multiplies values before they are split into S / S2 (XB_SCHEME_F16X2: XB_F16_ACT_SCALE)
    int* status = nullptr;      // device word, OR-ed with XB_STATUS_F16_OVERFLOW when a split output does not fit fp16 (nullable)
    // XB_EPI_LINEAR: C = alpha * (acc + bias) + R, fp32 row-major
    // XB_EPI_GLU: weight rows interleaved (a_c, b_c): C[m][c] = a_c * sigmoid(b_c), N/2 output columns, fp32 row-major
    float* C = nullptr; int ldc = 0;
    const float* R = nullptr; int ldr = 0;
    float alpha = 1.0f;
    // XB_EPI_SWISH_SPLIT / XB_EPI_GELU_SPLIT: S = split3(act(acc + bias)) as 3 K-blocked pieces [3][N/16][Spad][16] (the next layer's A operand)
    __bf16* S = nullptr; int Spad = 0;
    // Windowed (conv1d) mode, batch > 1 or taps > 1: A pieces are [3][batch][C/16][stride][Lp][16] — per clip, K-blocked over the
    // Cin channels and PHASE-MAJOR in time: input row t lives in plane t % stride at index t / stride (+ any front padding the
    // producer added), so output row m of tap j reads plane j % stride, index m + j / stride: consecutive output rows are
    // consecutive 32-byte chunks for every tap (a row-major time axis gave stride-spaced chunks: 20-30 % slower).
    // The K-blocks of a windowed GEMM run in WINDOW ORDER (xb_window_block below): channel block slowest, then phase plane, then row
    // offset, so consecutive K-blocks read the same plane of the same channel block one row apart — the re-read of a tap-major order
    // (every input row once per tap sharing its plane, far apart in time) left L2 and cost k / stride x the activation bytes. The
    // weights are split into the same order (launch_split_blocked with the window description). M, Mpad, Spad are PER CLIP; C / R are [batch][M][ldc];
    // S is [3][batch][N/16][Sphases][Spad][16] with output row m in plane m % Sphases at index m / Sphases + Sfront.
    // Defaults describe a plain linear layer.
    int batch = 1, stride = 1;
    int cblocks = 0;   // Cin / 16 (0: K / 16, i.e. one tap)
    int Lp = 0;        // rows per phase plane of A (0: Mpad)
    int Sphases = 1, Sfront = 0;
    // The split output may be a block range of a wider per-clip buffer (an operand K-concatenated from two producers): Sblocks =
    // 16-channel blocks per clip of that buffer (0: N / 16), Sblock0 = the block output column 0 maps to.
    int Sblocks = 0, Sblock0 = 0;
    // XB_EPI_ELU_SPLIT: S = split3(ELU(acc + bias)). XB_EPI_RAW_ELU_SPLIT2: S = split3(acc + bias) and S2 = split3(ELU(acc + bias)),
    // each with its own phase / padding / block description (SEANet: the residual block's shortcut operand and its conv operand).
    __bf16* S2 = nullptr;
    int S2pad = 0, S2phases = 1, S2front = 0, S2blocks = 0, S2block0 = 0;
    // XB_EPI_QKV (the fused q / k / v projection, N = 3 * qkv_hid, plain linear mode): columns [0, qkv_hid) = q go to C as fp32 (ldc), columns
    // [qkv_hid, 3 qkv_hid) = k, v are written as ROW-MAJOR pieces S[which = 0 k / 1 v][piece][Spad rows][qkv_hid] times split_scale — the operand
    // images the attention kernel stages without splitting (attention_bf16x3.hip, KVP)
    int qkv_hid = 0;
    // filled by launch_gemm_f16x2_tg from Sphases / S2phases (at_common.h, FastDivU): the epilogue's row -> (plane, index) split without the
    // compiler's reciprocal sequence
    FastDivU fdS, fdS2;
};
// Window order of the K-blocks of a conv with `ktaps` taps, `stride` and `cblocks` 16-channel blocks. Tap j lives in phase plane
// j % stride at row offset j / stride; plane p holds n_p = ceil((ktaps - p) / stride) taps. K-block kt = cbk * ktaps + i, where i walks the
// planes in order and the row offsets inside a plane. xb_window_block: kt -> (plane image t = cbk * stride + p, row offset);
// xb_window_dst: tap-major source block (tap * cblocks + cbk) -> kt.
__host__ __device__ inline void xb_window_block(int kt, int stride, int ktaps, int& t, int& off) {
    const int cbk = kt / ktaps, i = kt - cbk * ktaps;
    const int q = ktaps / stride, r = ktaps - q * stride;   // planes p < r hold q + 1 taps, the others q
    int p;
    if (i < r * (q + 1)) { p = i / (q + 1); off = i - p * (q + 1); }
    else { const int i2 = i - r * (q + 1); p = r + i2 / q; off = i2 - (p - r) * q; }
    t = cbk * stride + p;
}
__host__ __device__ inline int xb_window_dst(int src_block, int cblocks, int stride, int ktaps) {
    const int tap = src_block / cblocks, cbk = src_block - tap * cblocks;
    const int off = tap / stride, p = tap - off * stride;
    const int q = ktaps / stride, r = ktaps - q * stride;
    const int i = (p < r ? p * (q + 1) : r * (q + 1) + (p - r) * q) + off;
    return cbk * ktaps + i;
}

// fills the causal reflect padding of windowed-mode pieces [3][B][blocks][phases][Lp][16]: padded rows i < pad (row i lives in plane
// i % phases at index i / phases) become copies of padded row 2 * pad - i
int launch_reflect_front(__bf16* S, int B, int blocks, int phases, int Lp, int pad, hipStream_t stream, int npieces = 3);

// fp32 row-major [rows][ld] (first K columns) * scale -> K-blocked pieces [pieces][K/16][rows_pad][16]; rows >= `rows` are zero-filled.
// win_cblocks > 0: x is a packed conv weight (K index = tap * Cin + c, Cin = 16 * win_cblocks) and its K-blocks are written in the
// window order of a conv with that stride (xb_window_dst).
int launch_split_blocked(const float* x, int ld, long long rows, long long rows_pad, int K, __bf16* out, hipStream_t stream,
                         int scheme = XB_SCHEME_BF16X3, float scale = 1.0f, int* status = nullptr, int win_cblocks = 0, int win_stride = 1);
// fp32 [B][L][C] -> the windowed-mode pieces of a causal strided conv with kernel = 2 * stride (reflect front padding included)
int launch_split_phase_major(const float* x, int B, int L, int C, int stride, int Lp, __bf16* out, hipStream_t stream);
// the same for any front padding (`pad` rows: reflected, or zeros with reflect = 0) and either scheme: fp32 [B][L][C] * scale ->
// [pieces][B][C/16][stride][Lp][16]
int launch_split_windowed(const float* x, int B, int L, int C, int stride, int pad, int Lp, __bf16* out, hipStream_t stream,
                          int scheme = XB_SCHEME_BF16X3, float scale = 1.0f, int* status = nullptr, int reflect = 1);
int launch_gemm_bf16x3(const Bf16x3Args& a, hipStream_t stream);
// Range tables (split_scheme.h, RangeMax / range_publish): every `status` pointer handed to a split writer addresses a {flag word, census word}
// PAIR of its site in a table the model handle owns; this ORs the sites' flag words into the caller's single status word at the end of a call
int launch_range_combine(const int* range_tab, int nsites, int* status_out, hipStream_t stream);
// the fp16 scheme's kernel for launches that fill the chip (gemm_f16x2_tg.hip); launch_gemm_bf16x3 dispatches to it
bool gemm_f16x2_tg_eligible(const Bf16x3Args& a);
int launch_gemm_f16x2_tg(const Bf16x3Args& a, hipStream_t stream);

}  // namespace at
