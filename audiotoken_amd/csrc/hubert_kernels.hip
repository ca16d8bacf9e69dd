// Kernels specific to the semantic_s tokenizer (HuBERT-base): first feature-extractor conv (Cin = 1) and the
// per-channel GroupNorm + GELU that follows it (HF modeling_hubert.py:154-176). Everything else (strided convs,
// grouped positional conv, Linear, attention, LayerNorm, k-means assignment) reuses the shared kernels.
#include "gemm_core.h"
#include "hubert_kernels.h"

namespace at {

// conv0: [B][N] -> [B][T0][512], k = 10, stride 5, valid, no bias. HBM-bound (20 B in, 2 KB out per frame).
// One thread = one frame x 4 channels; 128 consecutive threads write one 2-KB row.
__global__ __launch_bounds__(256) void hub_conv0_kernel(const float* __restrict__ wav, const float* __restrict__ w /*[512][10]*/,
                                                        float* __restrict__ out, int N, int T0, long long total) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total) return;
    const int cg = (int)(gid & 127);
    const long long bt = gid >> 7;
    const long long b = bt / T0;
    const int t = (int)(bt - b * T0);
    const float* x = wav + b * N + (long long)t * 5;
    float xv[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) xv[k] = x[k];
    f4 o;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float* wc = w + (cg * 4 + c) * 10;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 10; ++k) acc = fmaf(wc[k], xv[k], acc);
        o[c] = acc;
    }
    *reinterpret_cast<f4*>(out + bt * 512 + cg * 4) = o;
}

int launch_hub_conv0(const float* wav, const float* w, float* out, int B, int N, int T0, hipStream_t stream) {
    const long long total = (long long)B * T0 * 128;
    hipLaunchKernelGGL(hub_conv0_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, wav, w, out, N, T0, total);
    AT_CHECK_HIP(hipGetLastError());
    return 0;
}

// GroupNorm(512 groups, 512 channels) = per (clip, channel) normalisation over ALL T0 frames (padding included, as HF does),
// eps 1e-5, affine; then exact GELU. Pass 1: per-channel sum / sum of squared deviations in two sweeps over a time slab,
// combined across slabs with Chan's parallel-variance update on the host-free second kernel. Pass 2: apply in place.
// Layout [B][T0][512]: a workgroup = (clip, slab of frames); thread = 2 channels... kept simple: 512 threads, 1 channel each.
constexpr int GN_SLAB = 2048;

__global__ __launch_bounds__(512) void hub_gn_partial_kernel(const float* __restrict__ x, float* __restrict__ part /*[B][nslab][512][2]*/,
                                                             int T0, int nslab) {
    const int b = blockIdx.y, slab = blockIdx.x, c = threadIdx.x;
    const int t0 = slab * GN_SLAB;
    const int t1 = t0 + GN_SLAB < T0 ? t0 + GN_SLAB : T0;
    const float* p = x + ((long long)b * T0) * 512 + c;
    float s = 0.f;
    for (int t = t0; t < t1; ++t) s += p[(long long)t * 512];
    const float n = (float)(t1 - t0);
    const float mean = s / n;
    float q = 0.f;
    for (int t = t0; t < t1; ++t) { const float d = p[(long long)t * 512] - mean; q = fmaf(d, d, q); }
    float* o = part + (((long long)b * nslab + slab) * 512 + c) * 2;
    o[0] = mean;
    o[1] = q;
}

__global__ __launch_bounds__(512) void hub_gn_final_kernel(const float* __restrict__ part, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ ss /*[B][512][2]: scale, shift*/,
                                                           int T0, int nslab) {
    const int b = blockIdx.x, c = threadIdx.x;
    double n = 0.0, mean = 0.0, m2 = 0.0;
    for (int s = 0; s < nslab; ++s) {
        const float* o = part + (((long long)b * nslab + s) * 512 + c) * 2;
        const int t0 = s * GN_SLAB;
        const double nb = (double)((t0 + GN_SLAB < T0 ? t0 + GN_SLAB : T0) - t0);
        const double delta = (double)o[0] - mean;
        const double nn = n + nb;
        mean += delta * nb / nn;
        m2 += (double)o[1] + delta * delta * n * nb / nn;
        n = nn;
    }
    const float var = (float)(m2 / n);
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
    const float scale = rstd * gamma[c];
    ss[((long long)b * 512 + c) * 2 + 0] = scale;
    ss[((long long)b * 512 + c) * 2 + 1] = fmaf(-(float)mean, scale, beta[c]);
}

__global__ __launch_bounds__(256) void hub_gn_apply_gelu_kernel(float* __restrict__ x, const float* __restrict__ ss, int T0, long long total) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;   // one thread = 4 channels of one frame
    if (gid >= total) return;
    const int cg = (int)(gid & 127);
    const long long bt = gid >> 7;
    const long long b = bt / T0;
    f4 v = *reinterpret_cast<const f4*>(x + bt * 512 + cg * 4);
    const float* p = ss + (b * 512 + cg * 4) * 2;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float y = fmaf(v[k], p[2 * k], p[2 * k + 1]);
        v[k] = 0.5f * y * (1.0f + erff(y * 0.70710678118654752440f));
    }
    *reinterpret_cast<f4*>(x + bt * 512 + cg * 4) = v;
}

int hub_gn_nslab(int T0) { return (T0 + GN_SLAB - 1) / GN_SLAB; }

int launch_hub_groupnorm_gelu(float* x, const float* gamma, const float* beta, float* part, float* ss, int B, int T0, hipStream_t stream) {
    const int nslab = hub_gn_nslab(T0);
    hipLaunchKernelGGL(hub_gn_partial_kernel, dim3(nslab, B), dim3(512), 0, stream, x, part, T0, nslab);
    AT_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(hub_gn_final_kernel, dim3(B), dim3(512), 0, stream, part, gamma, beta, ss, T0, nslab);
    AT_CHECK_HIP(hipGetLastError());
    const long long total = (long long)B * T0 * 128;
    hipLaunchKernelGGL(hub_gn_apply_gelu_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, x, ss, T0, total);
    AT_CHECK_HIP(hipGetLastError());
    return 0;
}

// frame-level validity from the sample mask: frames [0, out_len(sum(mask))) are valid (HF modeling_hubert.py:664-693)
__global__ __launch_bounds__(256) void hub_frame_mask_kernel(const float* __restrict__ smask, float* __restrict__ fmask, int N, int T) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    float s = 0.f;
    if (smask) for (int i = threadIdx.x; i < N; i += 256) s += smask[(long long)b * N + i];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    long long len = smask ? (long long)((red[0] + red[1]) + (red[2] + red[3])) : N;
    const int ks[7] = {10, 3, 3, 3, 3, 2, 2}, st[7] = {5, 2, 2, 2, 2, 2, 2};
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const long long num = len - ks[i];
        len = (num >= 0 ? num / st[i] : -((-num + st[i] - 1) / st[i])) + 1;   // floor division
    }
    for (int t = threadIdx.x; t < T; t += 256) fmask[(long long)b * T + t] = t < len ? 1.0f : 0.0f;
}

int launch_hub_frame_mask(const float* smask, float* fmask, int B, int N, int T, hipStream_t stream) {
    hipLaunchKernelGGL(hub_frame_mask_kernel, dim3(B), dim3(256), 0, stream, smask, fmask, N, T);
    AT_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace at
