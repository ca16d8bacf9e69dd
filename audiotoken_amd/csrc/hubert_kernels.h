// Launchers of the HuBERT-specific kernels (hubert_kernels.hip).
#pragma once
#include "at_common.h"

namespace at {
int launch_hub_conv0(const float* wav, const float* w, float* out, int B, int N, int T0, hipStream_t stream);
int hub_gn_nslab(int T0);
int launch_hub_groupnorm_gelu(float* x, const float* gamma, const float* beta, float* part, float* ss, int B, int T0, hipStream_t stream);
int launch_hub_frame_mask(const float* smask, float* fmask, int B, int N, int T, hipStream_t stream);
}  // namespace at
