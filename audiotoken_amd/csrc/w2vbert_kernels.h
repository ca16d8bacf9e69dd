// Launchers of the semantic_m-specific kernels (w2vbert_kernels.hip).
#pragma once
#include "at_common.h"

namespace at {

int launch_frame_prep(const float* wav, const float* smask, const float* window, double* frames, float* fmask, int B, int N,
                      int F, hipStream_t stream, int* status = nullptr);   // status: OR-ed with XB_STATUS_NONFINITE when a frame holds a NaN / infinity
int launch_dft_f64(const double* frames, const double* dft, float* spec, long long M, int N, hipStream_t stream);
int launch_fbank_normalize(const float* logmel, const float* fmask, float* stats, float* feats, float* amask, int B, int F, int Tp,
                           hipStream_t stream);
int launch_layernorm(const float* x, const float* gamma, const float* beta, const float* row_mask, float* y, long long rows, int D,
                     hipStream_t stream);
// arith: 0 = fp32-MFMA kernel, 1 = bf16x3, 2 = f16x2 (attention_bf16x3.hip), -1 = the default (
// $AUDIOTOKEN_SEMANTIC_ARITH, else f16x2); status: device word for the fp16 range check (nullable)
// ctx_pieces != nullptr (split arithmetic only): the context is written as operand pieces [NP][hid/16][rows_pad][16] instead of fp32 ctx
// kv_pieces != nullptr (f16x2 only): k and v are read as the row-major fp16 pieces [which][piece][rows_pad][hid] the q / k / v projection wrote
// (XB_EPI_QKV) instead of being split from the fp32 qkv rows by every query-tile workgroup; qkv then only supplies q
int launch_relpos_attention(const float* qkv, const float* amask, const float* dist_emb, float* ctx, int B, int T,
                            hipStream_t stream, int heads = 16, int arith = -1, int* status = nullptr, __bf16* ctx_pieces = nullptr, long long rows_pad = 0,
                            const __bf16* kv_pieces = nullptr, int w8 = -1, const __bf16* dist_pieces = nullptr, float dist_scale = 1.0f);
// the same attention with both products as operand splits on the 16-bit matrix cores (attention_bf16x3.hip); scheme = XB_SCHEME_*
int launch_relpos_attention_x3(const float* qkv, const float* amask, const float* dist_emb, float* ctx, int B, int T, hipStream_t stream, int heads,
                               int scheme, int* status, __bf16* ctx_pieces = nullptr, long long rows_pad = 0, const __bf16* kv_pieces = nullptr, int w8 = -1,
                               const __bf16* dist_pieces = nullptr, float dist_scale = 1.0f);
// round 4: the f16x2 attention with pre-split k / v as ONE 8-wave workgroup per CU, 64-key tiles and LDS-DMA staging (attention_f16x2_w8.hip);
// launch_relpos_attention_x3 dispatches to it when kv_pieces != nullptr and w8 != 0 (w8 = -1: $AUDIOTOKEN_ATTN_W8, default 1)
bool relpos_attention_w8_eligible(int T, int heads, long long rows_pad, long long B, bool relpos);
// dist_pieces: the distance embeddings as fp16 pieces [2][96][64] * dist_scale (launch_dist_split; 24 576 bytes), NULL = no rel-pos bias (HuBERT)
int launch_relpos_attention_w8(const float* qkv, const float* amask, const __bf16* dist_pieces, float dist_scale, float* ctx, int B, int T, hipStream_t stream,
                               int heads, int* status, __bf16* ctx_pieces, long long rows_pad, const __bf16* kv_pieces);
int launch_dist_split(const float* dist_emb, __bf16* out, float scale, hipStream_t stream);
// fp32 qkv rows [rows][3 hid] -> row-major k / v pieces [which][piece][rows_pad][hid] * XB_F16_ACT_SCALE (what XB_EPI_QKV writes)
int launch_kv_rowmajor_split(const float* qkv, __bf16* out, long long rows, long long rows_pad, int hid, int* status, hipStream_t stream);
// pieces != nullptr: the output is written as the K-blocked operand pieces [NP][64][rows_pad][16] of `scheme` (times `scale`) instead of fp32
// the same op as a streaming kernel: one channel per thread walking along time, 16 waves per CU, every input row read once (dwconv_stream.hip);
// bit-identical to launch_dwconv_ln_swish
int launch_dwconv_stream(const float* g, const float* w, const float* gamma, const float* beta, float* out, int B, int T, hipStream_t stream,
                         __bf16* pieces = nullptr, long long rows_pad = 0, int scheme = 0, float scale = 1.0f, int* status = nullptr);
int launch_dwconv_ln_swish(const float* g, const float* w, const float* gamma, const float* beta, float* out, int B, int T,
                           hipStream_t stream, __bf16* pieces = nullptr, long long rows_pad = 0, int scheme = 0, float scale = 1.0f, int* status = nullptr);
// LayerNorm(1024) written as operand pieces [NP][64][rows_pad][16] of `scheme` (times `scale`); y != nullptr: also as fp32 [rows][1024]
int launch_layernorm2_split(const float* x, const float* gamma, const float* beta, float* y, const float* gamma2, const float* beta2, __bf16* out, long long rows,
                            long long rows_pad, int D, int scheme, float scale, int* status, hipStream_t stream);
int launch_layernorm_split(const float* x, const float* gamma, const float* beta, const float* row_mask, float* y, __bf16* out, long long rows, long long rows_pad,
                           int D, int scheme, float scale, int* status, hipStream_t stream);
// status (nullable): OR-ed with XB_STATUS_NONFINITE when a row of x holds a NaN / infinity
// ld: row stride of `dots` (0 = C; > C when the score GEMM ran against a zero-padded code book)
// codebook (nullable): the fp32 code rows [C][D]; when given, the codes within a relative window of the best approximate distance are re-evaluated exactly
// (sum of squared differences in float64) — see vq_argmax_kernel
int launch_vq_argmax(const float* x, const float* dots, const float* e2, int16_t* out, long long rows, int D, int C,
                     hipStream_t stream, int* status = nullptr, int ld = 0, const float* codebook = nullptr);

}  // namespace at
