// Launchers of the HuBERT-specific kernels (hubert_kernels.hip).
#pragma once
#include "gemm_bf16x3.h"
#include "at_common.h"

namespace at {
// conv0 + GroupNorm + GELU in one pass (statistics from float64 waveform moments); part needs B * hub_ws_nchunk(T0) * 65 doubles
int hub_ws_nchunk(int T0);
// split != nullptr: instead of out, write the clip-major K-blocked, phase-major bf16 pieces [3][B][512/16][2][Lp][16] the split-bf16
// conv GEMM reads (frame t in plane t & 1 at index t >> 1)
int launch_hub_conv0_gn_gelu(const float* wav, const float* w, const float* gamma, const float* beta, float* part, float* ss, float* out,
                             int B, int N, int T0, hipStream_t stream, __bf16* split = nullptr, int Lp = 0, int scheme = 0, float split_scale = 1.0f,
                             int* status = nullptr);
int launch_hub_frame_mask(const float* smask, float* fmask, int B, int N, int T, hipStream_t stream);
// grouped positional conv on the two-piece fp16 scheme (hubert_posconv.hip): weights pre-arranged by launch_posconv_weight_split from the folded fp32
// weights [16][48 out][128 taps][48 in] * w_scale; pos = x + gelu(conv(x) + bias), x / pos fp32 [B][T][768]
size_t posconv_weight_pieces_bytes();
int launch_posconv_weight_split(const float* w, __bf16* out, float scale, hipStream_t stream);
int launch_hubert_posconv(const float* x, const __bf16* w_pieces, const float* bias, float* pos, int B, int T, float w_scale, int* status, hipStream_t stream);

}  // namespace at
