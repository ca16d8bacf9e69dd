// Kernels specific to the semantic_s tokenizer (HuBERT-base): first feature-extractor conv (Cin = 1) and the
// per-channel GroupNorm + GELU that follows it (HF modeling_hubert.py:154-176). Everything else (strided convs,
// grouped positional conv, Linear, attention, LayerNorm, k-means assignment) reuses the shared kernels.
#include "gemm_core.h"
#include "hubert_kernels.h"
#include "split_scheme.h"

namespace at {

// ---- conv0 + GroupNorm + GELU in ONE pass over the 2-KB frame rows ---------------------------------------------------------
// GroupNorm(512 groups of 1 channel) needs, per (clip, channel), the mean and variance over time of y_c[t] = w_c . x[5t .. 5t+9].
// Both are quadratic forms of the waveform's windowed moments, which are tiny and cheap:
//     mean_c = w_c . m,   E[y_c^2] = w_c^T R w_c,    m[k] = mean_t x[5t+k],   R[k][l] = mean_t x[5t+k] x[5t+l]
// so the statistics come from one float64 sweep over the WAVEFORM (4 B/sample) instead of two sweeps over the conv
// output (2 KB/frame), and conv0, the affine normalisation and the exact GELU are applied while the output row is
// written once: 100 GB of HBM traffic (write, 2 x read, read+write) become 25 GB. float64 moments make the quadratic
// form as accurate as a two-pass variance of the rounded outputs (the reference's own fp32 statistics differ from both
// by ~1e-6 relative).
constexpr int WS_CHUNK = 4096;     // frames per partial block
constexpr int WS_NMOM = 65;        // 10 first moments + 55 second moments (upper triangle)

int hub_ws_nchunk(int T0) { return (T0 + WS_CHUNK - 1) / WS_CHUNK; }

__global__ __launch_bounds__(256) void hub_wavstats_kernel(const float* __restrict__ wav, double* __restrict__ part /*[B][nchunk][65]*/,
                                                           int N, int T0, int nchunk) {
    __shared__ double red[4][WS_NMOM];
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int t0 = chunk * WS_CHUNK;
    const int t1 = t0 + WS_CHUNK < T0 ? t0 + WS_CHUNK : T0;
    const float* x = wav + (long long)b * N;
    double acc[WS_NMOM];
#pragma unroll
    for (int i = 0; i < WS_NMOM; ++i) acc[i] = 0.0;
    for (int t = t0 + threadIdx.x; t < t1; t += 256) {
        double xv[10];
#pragma unroll
        for (int k = 0; k < 10; ++k) xv[k] = (double)x[(long long)t * 5 + k];
        int idx = 10;
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            acc[k] += xv[k];
#pragma unroll
            for (int l = k; l < 10; ++l) { acc[idx] = fma(xv[k], xv[l], acc[idx]); ++idx; }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < WS_NMOM; ++i) {
        double v = acc[i];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
        if (lane == 0) red[wave][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < WS_NMOM)
        part[((long long)b * nchunk + chunk) * WS_NMOM + threadIdx.x] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

__global__ __launch_bounds__(512) void hub_gn_coeff_kernel(const double* __restrict__ part, const float* __restrict__ w /*[512][10]*/,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float* __restrict__ ss /*[B][512][2]: scale, shift*/, int T0, int nchunk) {
    __shared__ double mom[WS_NMOM];
    const int b = blockIdx.x, c = threadIdx.x;
    if (c < WS_NMOM) {
        double s = 0.0;
        for (int j = 0; j < nchunk; ++j) s += part[((long long)b * nchunk + j) * WS_NMOM + c];   // fixed order: deterministic
        mom[c] = s / (double)T0;
    }
    __syncthreads();
    double wv[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) wv[k] = (double)w[c * 10 + k];
    double mean = 0.0, ey2 = 0.0;
    int idx = 10;
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        mean += wv[k] * mom[k];
#pragma unroll
        for (int l = k; l < 10; ++l) {
            const double term = wv[k] * wv[l] * mom[idx++];
            ey2 += (l == k) ? term : 2.0 * term;
        }
    }
    double var = ey2 - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const float rstd = 1.0f / sqrtf((float)var + 1e-5f);
    const float scale = rstd * gamma[c];
    ss[((long long)b * 512 + c) * 2 + 0] = scale;
    ss[((long long)b * 512 + c) * 2 + 1] = fmaf(-(float)mean, scale, beta[c]);
}

// conv0: [B][N] -> [B][T0][512], k = 10, stride 5, valid, no bias (20 B in, 2 KB out per frame), then scale/shift and GELU.
// Workgroup = 64 consecutive frames of one clip; thread = 4 channels, which keeps its 40 taps and 8 affine coefficients in
// registers for all of its 32 frames (one frame x 4 channels per thread re-fetched 48 words per 16 bytes stored and
// ran at 1.3 TB/s). The waveform segment (325 samples) goes through LDS; 128 threads write one 2-KB output row (fp32 form) or stage
// 16 frames of pieces in LDS and write them as 256-byte runs (piece form: see below).
constexpr int C0_FRAMES = 64;

template <class SC>
__global__ __launch_bounds__(256) void hub_conv0_gn_gelu_kernel(const float* __restrict__ wav, const float* __restrict__ w /*[512][10]*/,
                                                                const float* __restrict__ ss, float* __restrict__ out, int N, int T0,
                                                                typename SC::T* __restrict__ split, int Lp, float split_scale, int* __restrict__ status) {
    __shared__ float xs[C0_FRAMES * 5 + 8];
    const int b = blockIdx.y, t0 = blockIdx.x * C0_FRAMES;
    const int cg = threadIdx.x & 127, sub = threadIdx.x >> 7;
    const float* x = wav + (long long)b * N;
    for (int i = threadIdx.x; i < C0_FRAMES * 5 + 5; i += 256) {
        const long long s = (long long)t0 * 5 + i;
        xs[i] = s < N ? x[s] : 0.f;
    }
    // Channel PAIRS on the packed fp32 instructions (round 5): this kernel runs no MFMA — it is vector-issue bound (6.3 G outputs x ~33 instructions at 128 x 30 s:
    // 7.7 of its 8.2 ms) — and v_pk_fma_f32 / v_pk_mul_f32 do two IEEE operations per issue slot. Same operations in the same order per element as the scalar
    // form (10-tap fma chain, affine fma, gelu_erf_scaled): bit-identical outputs.
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 wr[2][10], sc[2], sh[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
#pragma unroll
        for (int k = 0; k < 10; ++k) wr[c][k] = f2{w[(cg * 4 + 2 * c) * 10 + k], w[(cg * 4 + 2 * c + 1) * 10 + k]};
        sc[c] = f2{ss[((long long)b * 512 + cg * 4 + 2 * c) * 2], ss[((long long)b * 512 + cg * 4 + 2 * c + 1) * 2]};
        sh[c] = f2{ss[((long long)b * 512 + cg * 4 + 2 * c) * 2 + 1], ss[((long long)b * 512 + cg * 4 + 2 * c + 1) * 2 + 1]};
    }
    __syncthreads();
    const int nf = T0 - t0 < C0_FRAMES ? T0 - t0 : C0_FRAMES;
    float* orow = out + ((long long)b * T0 + t0) * 512 + cg * 4;
    RangeMax over;
    // fp16 scheme: the site's power-of-two scale rides on GELU's leading 0.5 (bit for bit scale x GELU: gelu_erf_scaled), the split takes it prescaled
    const float half = (SC::NP == 2 && split) ? 0.5f * split_scale : 0.5f;
    // Piece output through LDS (round 5). The next conv's operand layout is [piece][clip][channel block 32][plane t & 1][t >> 1][16 channels]: a thread's 8-byte
    // quad of one frame lands 32 bytes from its neighbours' and 1.5 MB from the next channel block's — written straight from registers a wave store was sixteen
    // separate 32-byte segments (3 TB/s of a kernel that writes 25 GB). Now 16 frames at a time are staged in LDS as [piece][block][plane][8 rows][16]
    // (+ 32 B per block against bank conflicts) and leave as 256-byte runs (8 rows x 32 B of one (piece, block, plane)), as layernorm_split_kernel does.
    constexpr int SUBF = 16, BLK_LD = 2 * (SUBF / 2) * 16 + 16;
    typedef typename SC::T PT;
    __shared__ __attribute__((aligned(16))) PT tile[SC::NP][32][BLK_LD];
    typedef unsigned int u4_ __attribute__((ext_vector_type(4)));
    const long long ps = (long long)gridDim.y * 32 * 2 * Lp * 16;
    for (int f0 = 0; f0 < (split ? C0_FRAMES : nf); f0 += SUBF) {
        if (split && f0 >= nf) break;
#pragma unroll 1
        for (int f = f0 + sub; f < (split ? f0 + SUBF : nf); f += 2) {
            if (f >= nf) break;
            float xv[10];
#pragma unroll
            for (int k = 0; k < 10; ++k) xv[k] = xs[f * 5 + k];
            f4 o;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                f2 acc = f2{0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 10; ++k) acc = __builtin_elementwise_fma(wr[c][k], f2{xv[k], xv[k]}, acc);
                const f2 g = gelu_erf_scaled2(__builtin_elementwise_fma(acc, sc[c], sh[c]), half);
                o[2 * c] = g[0];
                o[2 * c + 1] = g[1];
            }
            if (split) {
                // channels 4cg .. 4cg + 3 = channel block cg / 4, quarter cg % 4; frame t -> plane t & 1, row t >> 1 (phase-major time axis for the stride-2 conv)
                typename SC::V4 p[SC::NP];
                if constexpr (SC::NP == 2) over |= split4_prescaled<SC>(o, p);
                else over |= split4<SC>(o, split_scale, p);
                const int fl = f - f0;
#pragma unroll
                for (int i = 0; i < SC::NP; ++i)
                    *reinterpret_cast<typename SC::V4*>(&tile[i][cg >> 2][((fl & 1) * (SUBF / 2) + (fl >> 1)) * 16 + (cg & 3) * 4]) = p[i];
            } else {
                *reinterpret_cast<f4*>(orow + (long long)f * 512) = o;
            }
        }
        if (!split) break;      // (the fp32 output form wrote all its frames in the one pass above)
        __syncthreads();
        // 16-byte chunks: [piece][block][plane][row][half]; 16 consecutive lanes = one 256-byte run in global memory
        const int ts = t0 + f0;                                   // even
        for (int e = threadIdx.x; e < SC::NP * 32 * 2 * (SUBF / 2) * 2; e += 256) {
            const int hf = e & 1, row = (e >> 1) & (SUBF / 2 - 1), plane = (e >> 4) & 1, blk = (e >> 5) & 31, pi = e >> 10;
            if (ts + 2 * row + plane < T0)
                *reinterpret_cast<u4_*>(split + pi * ps + ((((long long)b * 32 + blk) * 2 + plane) * Lp + (ts >> 1) + row) * 16 + hf * 8) =
                    *reinterpret_cast<const u4_*>(&tile[pi][blk][(plane * (SUBF / 2) + row) * 16 + hf * 8]);
        }
        __syncthreads();
    }
    if constexpr (SC::RANGE_CHECK)
        range_publish(status, status ? status + 1 : nullptr, over);
}

int launch_hub_conv0_gn_gelu(const float* wav, const float* w, const float* gamma, const float* beta, float* part, float* ss, float* out,
                             int B, int N, int T0, hipStream_t stream, __bf16* split, int Lp, int scheme, float split_scale, int* status) {
    const int nchunk = hub_ws_nchunk(T0);
    double* dpart = reinterpret_cast<double*>(part);   // workspace slices are 256-byte aligned
    hipLaunchKernelGGL(hub_wavstats_kernel, dim3(nchunk, B), dim3(256), 0, stream, wav, dpart, N, T0, nchunk);
    AT_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(hub_gn_coeff_kernel, dim3(B), dim3(512), 0, stream, dpart, w, gamma, beta, ss, T0, nchunk);
    AT_CHECK_HIP(hipGetLastError());
    const dim3 grid((T0 + C0_FRAMES - 1) / C0_FRAMES, B);
    if (scheme == XB_SCHEME_F16X2)
        hipLaunchKernelGGL(hub_conv0_gn_gelu_kernel<SchemeF16x2>, grid, dim3(256), 0, stream, wav, w, ss, out, N, T0, reinterpret_cast<_Float16*>(split), Lp, split_scale, status);
    else
        hipLaunchKernelGGL(hub_conv0_gn_gelu_kernel<SchemeBf16x3>, grid, dim3(256), 0, stream, wav, w, ss, out, N, T0, split, Lp, split_scale, status);
    AT_CHECK_HIP(hipGetLastError());
    return 0;
}

// frame-level validity from the sample mask: frames [0, out_len(sum(mask))) are valid (HF modeling_hubert.py:664-693). One workgroup per clip streams the
// clip's mask with 16-byte loads, four in flight per thread (round 5: the scalar-load form took 0.72 ms for 128 x 480 000 samples — 2.7 GB/s per
// workgroup; the sum of 0 / 1 values is exact in fp32 in any order up to 2^24 samples per clip, beyond that the tail is added in double).
__global__ __launch_bounds__(1024) void hub_frame_mask_kernel(const float* __restrict__ smask, float* __restrict__ fmask, int N, int T) {
    __shared__ double red[16];
    const int b = blockIdx.x;
    double s = 0.0;
    if (smask) {
        const float* row = smask + (long long)b * N;
        const int head = (int)((16 - (reinterpret_cast<uintptr_t>(row) & 15)) & 15) / 4;      // floats up to the first 16-byte boundary
        const int h = head < N ? head : N;
        const int n4 = (N - h) / 4;
        const f4* v = reinterpret_cast<const f4*>(row + h);
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        int i = threadIdx.x;
        for (; i + 3 * 1024 < n4; i += 4 * 1024) {
            const f4 a0 = v[i], a1 = v[i + 1024], a2 = v[i + 2048], a3 = v[i + 3072];
            acc[0] += (a0.x + a0.y) + (a0.z + a0.w);
            acc[1] += (a1.x + a1.y) + (a1.z + a1.w);
            acc[2] += (a2.x + a2.y) + (a2.z + a2.w);
            acc[3] += (a3.x + a3.y) + (a3.z + a3.w);
        }
        for (; i < n4; i += 1024) { const f4 a0 = v[i]; acc[0] += (a0.x + a0.y) + (a0.z + a0.w); }
        s = (double)acc[0] + (double)acc[1] + (double)acc[2] + (double)acc[3];
        if ((int)threadIdx.x < h) s += (double)row[threadIdx.x];
        for (int j = h + 4 * n4 + (int)threadIdx.x; j < N; j += 1024) s += (double)row[j];
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    double tot = 0.0;
#pragma unroll
    for (int w = 0; w < 16; ++w) tot += red[w];
    long long len = smask ? (long long)tot : N;
    const int ks[7] = {10, 3, 3, 3, 3, 2, 2}, st[7] = {5, 2, 2, 2, 2, 2, 2};
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const long long num = len - ks[i];
        len = (num >= 0 ? num / st[i] : -((-num + st[i] - 1) / st[i])) + 1;   // floor division
    }
    for (int t = threadIdx.x; t < T; t += 1024) fmask[(long long)b * T + t] = t < len ? 1.0f : 0.0f;
}

int launch_hub_frame_mask(const float* smask, float* fmask, int B, int N, int T, hipStream_t stream) {
    hipLaunchKernelGGL(hub_frame_mask_kernel, dim3(B), dim3(1024), 0, stream, smask, fmask, N, T);
    AT_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace at
