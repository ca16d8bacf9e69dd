// Launchers of the EnCodec-specific kernels (encodec_kernels.hip).
#pragma once
#include "at_common.h"

namespace at {

struct LstmStepArgs {
    const float* xg;    // [B][T][4H], gate-interleaved columns (4*j + g), includes b_ih
    const float* b_hh;  // [4H], gate-interleaved
    float* c;           // [B][H] cell state (in/out)
    float* h_out;       // [B][T][H]
    float* y_out;       // optional [B][T][H]: h + skip
    const float* skip;  // [B][T][H] (LSTM stack input) when y_out != null
    int T, t, H, first;
    int y_elu;          // write ELU(h + skip) instead of h + skip (the only consumer applies ELU first)
};

// One LSTM layer over the whole sequence in a single persistent launch (lstm_seq.hip). <= 256 clips per launch.
struct LstmSeqArgs {
    const float* xg;     // [B][T][2048] input-side gates incl. b_ih, gate-interleaved columns
    const float* w_hh;   // [2048][512] gate-interleaved rows
    const float* b_hh;   // [2048]
    float* h_out;        // [B][T][512] (also the exchange buffer between workgroups)
    float* y_out;        // optional [B][T][512]: h + skip
    const float* skip;
    unsigned* sync;      // >= 1024 words: [63] sticky status, [128..639] flags[group][slice] (zeroed per launch)
    int B, T;
    int y_elu;           // y_out = ELU(h + skip)
    int n_groups;        // filled by the launcher
    long long h_bytes;   // filled by the launcher
    unsigned spin_limit = 1u << 18;   // polls of a hand-off flag before a workgroup gives up (~0.1-0.3 s; normal waits are microseconds)
    float w_scale_f16 = 0.f;          // lstm_seq_x3 only: > 0 = two-piece fp16 scheme with W_hh scaled by this power of two; 0 = three bf16 pieces
};
int launch_lstm_seq(const LstmSeqArgs& a, hipStream_t stream);
int lstm_seq_max_clips();
// the same layer with the recurrent product as split-bf16 MFMAs: 16 workgroups of 32 hidden units per 16-clip group (lstm_seq_x3.hip)
int launch_lstm_seq_x3(const LstmSeqArgs& a, hipStream_t stream);
int lstm_seq_x3_max_clips();
// Both layers in ONE persistent launch, layer 2 one step behind layer 1 (lstm_pipe.hip): three roles of workgroups per 16-clip group — layer 1,
// layer 2's input gates W_ih2 h1_t (replaces the projection GEMM between the layers), layer 2. Two-piece fp16 scheme only; <= lstm_pipe_max_clips().
struct LstmPipeArgs {
    const float* xg1;    // [B][T][2048] layer-1 input-side gates incl. b_ih (the projection GEMM's output)
    const float* w_hh1; const float* b_hh1;
    const float* w_ih2; const float* b_ih2;   // fp32 [2048][512] gate-interleaved rows, as the recurrent weights
    const float* w_hh2; const float* b_hh2;
    float* h1;           // [B][T][512] layer-1 output (exchange buffer of roles A and X)
    float* xg2;          // [B][T][2048] layer-2 input-side gates (exchange buffer of roles X and B)
    float* h2;           // [B][T][512] layer-2 output
    float* y_out;        // [B][T][512] h2 + skip
    const float* skip;
    unsigned* sync;      // as LstmSeqArgs
    int B, T;
    int y_elu;
    int n_groups = 0;    // filled by the launcher
    long long h_bytes = 0, xg_bytes = 0;
    unsigned spin_limit = 1u << 18;
    float ws_hh1 = 0.f, ws_ih2 = 0.f, ws_hh2 = 0.f;   // finalize-time power-of-two scales of the three weight matrices
    float act_scale = 0.f;                            // activation scale of the projection GEMM this launch stands in for (XB_F16_ACT_SCALE)
};
int launch_lstm_pipe(const LstmPipeArgs& a, hipStream_t stream);
int lstm_pipe_max_clips();
bool lstm_pipe_eligible(int B, int T);
// RVQ search with split dot products (rvq_encode_x3.hip); cb_s = codebooks as [3][n_cb * 1024][128] bf16 pieces, or with
// scheme = XB_SCHEME_F16X2 as [2][n_cb * 1024][128] fp16 pieces of E * cb_scale (launch_split_plain with the same scheme / scale);
// the fp16 scheme also needs the residual scale and the device status word that receives the range verdict
int launch_split_plain(const float* x, long long n, __bf16* out, hipStream_t stream, int scheme = 0, float scale = 1.f);
int launch_rvq_encode_x3(const float* x, long long rows, int T, const float* codebooks, const __bf16* cb_s, long long cb_piece,
                         const float* e2, int n_q, int16_t* codes, hipStream_t stream, int scheme = 0, float act_scale = 1.f,
                         float cb_scale = 1.f, int* status = nullptr);

// Fused SEANet stage 0 (seanet_stage0.hip): wav -> conv0 -> resblock(32) -> ELU -> conv k4 s2 -> x1 [B][N/2][64]
struct Stage0Args {
    const float* wav;   // [B][N]
    float* x1;          // [B][N/2][64]
    const float *w0, *b0;   // conv0 [32][7], [32]
    const float *w3, *b3;   // resblock conv3 packed [16][3*32], [16]
    const float *wt, *bt;   // resblock tail packed [32][16 + 32] = [W1 | Wsc], summed bias [32]
    const float *wd, *bd;   // strided conv packed [64][4*32], [64]
    int B, N;
    // operand scheme of the three split contractions (gemm_bf16x3.h): XB_SCHEME_F16X2 needs the activation scale, the power-of-two
    // scales of the three weight tensors and the device status word that receives the range verdict
    int scheme = 0;
    float act_scale = 1.f, w3_scale = 1.f, wt_scale = 1.f, wd_scale = 1.f;
    int* status = nullptr;
    // seanet_stage0x3 only: the block's 1x1 shortcut folded into conv0 — Wsc . conv0 is itself a 7-tap conv of the waveform with weights
    // wsc0 = Wsc . W0 [32][7]; bsc0 = Wsc . b0 + (b1 + bsc) [32] (computed in float64 at finalize). The kernel then never splits or stores the
    // raw conv0 output: the tail's split contraction is K = 16 (h only) and the shortcut is two fp32 MFMAs on the waveform segment.
    const float *wsc0 = nullptr, *bsc0 = nullptr;
};
int launch_seanet_stage0(const Stage0Args& a, hipStream_t stream);
int launch_seanet_stage0x3(const Stage0Args& a, hipStream_t stream);   // the same stage with split-bf16 contractions (seanet_stage0x3.hip)

// Fused 64-channel SEANet residual block with ELU epilogue (seanet_res64.hip): x [B][L][64] -> out [B][L][64]
struct Res64Args {
    const float* x;
    float* out;
    const float *w3, *b3;   // conv3 packed [32][3*64], [32]
    const float *wt, *bt;   // tail packed [64][32 + 64] = [W1 | Wsc], summed bias [64]
    int B, L;
    // seanet_res128x3 only: when set, the output is written as the three K-blocked bf16 pieces a stride-5 windowed split-bf16 GEMM
    // reads (gemm_bf16x3.h: [3][B][C/16][5 planes][Lp][16], row t in plane t % 5 at index t / 5 + 1) instead of fp32 rows
    __bf16* S = nullptr;
    int Lp = 0;
    // S_scheme = XB_SCHEME_F16X2 (gemm_bf16x3.h): two fp16 pieces of out * S_scale instead; status = device word for the fp16 range check
    int S_scheme = 0;
    float S_scale = 1.0f;
    int* status = nullptr;
    // operand scheme of the block's own contractions (seanet_res64x3 / seanet_res128x3): XB_SCHEME_BF16X3 (three bf16 pieces, six products)
    // or XB_SCHEME_F16X2 (two fp16 pieces, three products) with the activation scale and the two weight matrices' power-of-two scales
    int scheme = 0;
    float act_scale = 1.0f, w3_scale = 1.0f, wt_scale = 1.0f;
};
// fills the causal reflect padding (5 front rows = index 0 of every plane) of those pieces
int launch_reflect_front5(__bf16* S, int B, int cblocks, int Lp, hipStream_t stream, int npieces = 3);
int launch_seanet_res64(const Res64Args& a, hipStream_t stream);
// the 128-channel block on the bf16 matrix cores with exact 3-way bf16 splits of all operands (seanet_res128x3.hip)
int launch_seanet_res128x3(const Res64Args& a, hipStream_t stream);
// the same block, fp16 scheme only, with role-split waves (conv3 waves / tail waves, one barrier per 32-row tile: seanet_res128rs.hip);
// bit-identical to launch_seanet_res128x3 with scheme = XB_SCHEME_F16X2
int launch_seanet_res128rs(const Res64Args& a, hipStream_t stream);
int launch_seanet_res64x3(const Res64Args& a, hipStream_t stream);   // seanet_res64x3.hip
// Encoder stage-1 strided conv (64 -> 128, k 8, stride 4) with register-stationary weights (seanet_down64.hip)
struct Down64Args {
    const float* x;     // [B][L][64], already ELU'd
    float* out;         // [B][L/4][128]
    const float *w, *b; // packed [128][8*64], [128]
    int B, L;
    // seanet_down64x3: operand scheme (XB_SCHEME_BF16X3 / XB_SCHEME_F16X2 with its power-of-two scales) and the fp16 range status word
    int scheme = 0;
    float act_scale = 1.0f, w_scale = 1.0f;
    int* status = nullptr;
};
int launch_seanet_down64(const Down64Args& a, hipStream_t stream);
// the same conv on the bf16 matrix cores with exact 3-way bf16 splits of both operands (seanet_down64x3.hip)
int launch_seanet_down64x3(const Down64Args& a, hipStream_t stream);
// Stage 1 in one role-split kernel (seanet_res64down.hip): the 64-channel block and the strided conv above, fp16 scheme only; the block output stays
// in LDS. Bit-identical to launch_seanet_res64x3 + launch_seanet_down64x3 with scheme = XB_SCHEME_F16X2.
struct ResDown64Args {
    const float* x;         // [B][L][64]
    float* out;             // [B][L/4][128]
    const float *w3, *b3;   // block conv3 packed [32][3*64], [32]
    const float *wt, *bt;   // block tail packed [64][32 + 64], summed bias [64]
    const float *wd, *bd;   // strided conv packed [128][8*64], [128]
    int B, L;
    float act_scale = 1.0f, w3_scale = 1.0f, wt_scale = 1.0f, wd_scale = 1.0f;
    int* status_res = nullptr;    // range {flag, census} pair of the block's own splits
    int* status_down = nullptr;   // ... of the conv's operand (the block output)
};
int launch_seanet_res64down(const ResDown64Args& a, hipStream_t stream);
// Decoder tail (seanet_dectail.hip): x [B][L][64] (ELU'd) -> transposed conv (64->32, k4 s2) -> resblock(32) -> ELU -> conv k7 -> wav [B][2L]
struct DecTailArgs {
    const float* x;
    float* out;
    const float *wu, *bu;   // transposed conv as k = 2 GEMM: packed [2*32][2*64], bias [64] (per phase)
    const float *w3, *b3;   // block conv3 packed [16][3*32], [16]
    const float *wt, *bt;   // block tail packed [32][16 + 32], summed bias [32]
    const float *wl, *bl;   // last conv packed [7*32], [1]
    int B, L;
    // seanet_dectail_x2 only: the two-piece fp16 scheme's activation scale, the three weight matrices' power-of-two scales, the range status word
    float act_scale = 0.f, wu_scale = 0.f, w3_scale = 0.f, wt_scale = 0.f;
    int* status = nullptr;
};
// decoder stage 0 on the split GEMMs (seanet_dec256.hip): u fp32 [g][L][256] -> ELU(u) pieces with two reflected front rows (the k3 conv's operand) and
// raw u pieces in K-blocks 8..23 of the tail's operand; rows past the data zero-filled
int launch_dec_res256_split(const float* u, int g, int L, __bf16* ac3, int Lpc, __bf16* at3, int Mpc, float scale, int* status, hipStream_t stream);
int launch_zero_piece_rows(__bf16* S, long long planes, int Lp, int row0, int row1, hipStream_t stream);
int launch_seanet_dectail(const DecTailArgs& a, hipStream_t stream);
// the same kernel with its three contractions as two-piece fp16 operand splits (seanet_dectail_x2.hip); the last conv stays fp32 on the VALU
int launch_seanet_dectail_x2(const DecTailArgs& a, hipStream_t stream);
// Same block at 128 channels (seanet_res128.hip): x [B][L][128] -> out [B][L][128]; w3 [64][3*128], wt [128][64 + 128]
int launch_seanet_res128(const Res64Args& a, hipStream_t stream);

int launch_conv0(const float* wav, const float* w, const float* bias, float* out, int B, int N, hipStream_t stream);
int launch_lstm_step(const GemmArgs& a, const LstmStepArgs& s, hipStream_t stream);
int launch_rvq_encode(const float* x, long long rows, int T, const float* codebooks, const float* e2, int n_q,
                      int16_t* codes, hipStream_t stream);
int launch_rvq_decode(const int64_t* codes, int B, int K, int T, const float* codebooks, float* z, hipStream_t stream);
int launch_conv_last(const float* x, const float* w, const float* bias, float* out, int B, int L, hipStream_t stream);

}  // namespace at
