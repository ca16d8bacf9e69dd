// Kernels of the semantic_m tokenizer that are not plain GEMMs: log-mel framing, masked normalisation +
// frame stacking, LayerNorm, relative-position attention, GLU'd depthwise conv + LayerNorm + swish, VQ argmax.
// GEMM-shaped work (DFT, mel projection, all Linear layers) goes through gemm_core.h.
#include "gemm_core.h"
#include "w2vbert_kernels.h"
#include "split_scheme.h"
#include <cstdlib>
#include <type_traits>

namespace at {

// ------------------------------------------------------------------------------------------------------
// Frame preparation (reference audiotoken/processors.py:155-175, one frame per wave):
//   x * 2^15 -> subtract the frame mean -> pre-emphasis with the PRE-update neighbour -> Povey window.
// Evaluated in float64 with the reference's fp32 constants (0.97f, 0.03f, the fp32 window table) and handed to the
// f64 DFT below. Rationale: after the x 2^15 scaling a sample is ~1e4, so every fp32 rounding of the reference's
// frame arithmetic is ~1e-3 absolute; in a mel bin that is 40 dB below its neighbours this noise moves the
// log-mel value by ~1e-3 and depends on the summation order of torch.mean (ISA-dependent), i.e. it cannot be
// reproduced. Carrying the frame in f64 makes this side exact, so the distance to the reference is the
// reference's own rounding noise only (measured: <= 5e-4 on the golden vectors).
// Also emits the frame-validity mask: valid iff all 400 sample-mask values are 1 (avg_pool1d == 1, :102-108).
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void frame_prep_kernel(const float* __restrict__ wav, const float* __restrict__ smask,
                                                         const float* __restrict__ window, double* __restrict__ frames,
                                                         float* __restrict__ fmask, int N, int F, long long total_frames, int* __restrict__ status) {
    const int lane = threadIdx.x & 63;
    const long long fid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (fid >= total_frames) return;
    const long long b = fid / F;
    const int f = (int)(fid - b * F);
    const float* x = wav + b * N + (long long)f * 160;
    double v[7];
    double sum = 0.0;
    bool all_valid = true;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const int n = lane + 64 * j;
        v[j] = n < 400 ? (double)x[n] * 32768.0 : 0.0;
        sum += v[j];
        if (smask && n < 400) all_valid = all_valid && (smask[b * N + (long long)f * 160 + n] == 1.0f);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
    // a NaN / infinity in the waveform: the log(max(., floor)) of the mel stage would launder it (fmaxf drops a NaN) where the reference's clamp keeps it,
    // so it is flagged here, where every sample the model uses passes by (XB_STATUS_NONFINITE)
    if (status && lane == 0 && !(fabs(sum) <= 1.0e300)) atomicOr(status, XB_STATUS_NONFINITE);
    const double mean = sum / 400.0;
    const unsigned long long ok = __ballot(all_valid);
    double* out = frames + fid * 400;
    const double pre = (double)0.97f, first = (double)0.03f;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const int n = lane + 64 * j;
        if (n >= 400) continue;
        const double cur = v[j] - mean;
        double y;
        if (n == 0) {
            y = cur * first;
        } else {
            const double prev = (double)x[n - 1] * 32768.0 - mean;
            y = cur - pre * prev;
        }
        out[n] = y * (double)window[n];
    }
    if (lane == 0) fmask[fid] = (ok == ~0ull) ? 1.0f : 0.0f;
}

int launch_frame_prep(const float* wav, const float* smask, const float* window, double* frames, float* fmask, int B, int N,
                      int F, hipStream_t stream, int* status) {
    const long long total = (long long)B * F;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(frame_prep_kernel, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, stream, wav, smask, window, frames,
                       fmask, N, F, total, status);
    AT_CHECK_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------------
// DFT of the prepared frames in float64 on the f64 matrix cores (v_mfma_f64_16x16x4_f64):
//   spec[m][n] = (float) sum_t frames[m][t] * dft[n][t],  n < 257: cos, 260 <= n < 517: -sin  (K = 400).
// Why f64: the reference takes an fp32 FFT, whose rounding noise in weak bins next to strong ones reaches 1e-3 of
// the log-mel value. Any other fp32 transform adds its own, independent noise on top; an (effectively) exact
// transform adds none, so the distance to the reference is the reference's own noise and nothing more.
// The work is tiny (0.4 MFLOP per frame), so the f64 rate is irrelevant.
// Workgroup 64 frames x 64 bins, 4 waves x (32 x 32), K step 16, operands staged through LDS as doubles.
// ------------------------------------------------------------------------------------------------------
typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int DFT_K = 400, DFT_BK = 16, DFT_LD = 17;

__global__ __launch_bounds__(256) void dft_f64_kernel(const double* __restrict__ frames, const double* __restrict__ dft,
                                                      float* __restrict__ spec, long long M, int N) {
    __shared__ double As[64 * DFT_LD];
    __shared__ double Bs[64 * DFT_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, q = lane >> 4;
    const long long m0 = (long long)blockIdx.x * 64;
    const int n0 = blockIdx.y * 64;
    const int wm = wave >> 1, wn = wave & 1;
    d4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = d4{0., 0., 0., 0.};
    const int srow = tid >> 2, sk = (tid & 3) * 4;   // staging: 64 rows x 16 k, 4 values per thread
    for (int k0 = 0; k0 < DFT_K; k0 += DFT_BK) {
        __syncthreads();
        {
            const long long m = m0 + srow;
#pragma unroll
            for (int e = 0; e < 4; ++e) As[srow * DFT_LD + sk + e] = m < M ? frames[m * DFT_K + k0 + sk + e] : 0.0;
            const int n = n0 + srow;
#pragma unroll
            for (int e = 0; e < 4; ++e) Bs[srow * DFT_LD + sk + e] = n < N ? dft[(long long)n * DFT_K + k0 + sk + e] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = q * 4 + e;   // MFMA step e covers k in {4*quad + e}
            double a[2], bb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = As[(wm * 32 + i * 16 + r16) * DFT_LD + k];
#pragma unroll
            for (int j = 0; j < 2; ++j) bb[j] = Bs[(wn * 32 + j * 16 + r16) * DFT_LD + k];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], bb[j], acc[i][j], 0, 0, 0);
        }
    }
    // f64 C/D map: col = lane & 15 (bin), row = (lane >> 4) + 4 * reg (frame)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const long long m = m0 + wm * 32 + i * 16 + q + 4 * reg;
                const int n = n0 + wn * 32 + j * 16 + r16;
                if (m < M && n < N) spec[m * N + n] = (float)acc[i][j][reg];
            }
}

int launch_dft_f64(const double* frames, const double* dft, float* spec, long long M, int N, hipStream_t stream) {
    if (M <= 0) return 0;
    dim3 grid((unsigned)((M + 63) / 64), (N + 63) / 64);
    hipLaunchKernelGGL(dft_f64_kernel, grid, dim3(256), 0, stream, frames, dft, spec, M, N);
    AT_CHECK_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------------
// Masked per-clip, per-mel-bin mean and POPULATION variance (processors.py:117-135):
//   mean = sum(x*m)/max(cnt,1);  var = sum(((x*m) - mean)^2 * m)/max(cnt,1)
// One workgroup (320 threads = 80 bins x 4 frame groups) per clip. Sums run in float64 and are rounded once: the
// normalisation divides by sqrt(var + 1e-7), so on near-stationary input (e.g. a tone whose period divides the 160-sample
// hop: every frame identical, true variance 0) an fp32 summation-order difference of one ulp in the mean would be amplified
// 3000x; the exact mean makes x - mean vanish there, as it does in the reference.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(320) void fbank_stats_kernel(const float* __restrict__ logmel, const float* __restrict__ fmask,
                                                          float* __restrict__ stats /*[B][2][80]*/, int F) {
    __shared__ double red[4][80];
    __shared__ float mean_s[80];
    __shared__ double cnt_s;
    const int b = blockIdx.x;
    const int bin = threadIdx.x % 80, grp = threadIdx.x / 80;
    const float* x = logmel + (long long)b * F * 80;
    const float* m = fmask + (long long)b * F;
    double s = 0.0, c = 0.0;
    for (int f = grp; f < F; f += 4) {
        const float mk = m[f];
        s += (double)(x[(long long)f * 80 + bin] * mk);
        c += (double)mk;
    }
    red[grp][bin] = s;
    __syncthreads();
    __shared__ double cred[4];  // every bin column sees the same mask: bin 0's four partial counts define cnt
    if (bin == 0) cred[grp] = c;
    __syncthreads();
    if (threadIdx.x == 0) cnt_s = fmax((cred[0] + cred[1]) + (cred[2] + cred[3]), 1.0);
    __syncthreads();
    if (grp == 0) mean_s[bin] = (float)(((red[0][bin] + red[1][bin]) + (red[2][bin] + red[3][bin])) / cnt_s);
    __syncthreads();
    const float mean = mean_s[bin];
    double v = 0.0;
    for (int f = grp; f < F; f += 4) {
        const float mk = m[f];
        const float d = x[(long long)f * 80 + bin] * mk - mean;
        v += (double)(d * d * mk);
    }
    __syncthreads();
    red[grp][bin] = v;
    __syncthreads();
    if (grp == 0) {
        stats[((long long)b * 2 + 0) * 80 + bin] = mean;
        stats[((long long)b * 2 + 1) * 80 + bin] = (float)(((red[0][bin] + red[1][bin]) + (red[2][bin] + red[3][bin])) / cnt_s);
    }
}

// normalise, stack 2 frames -> 160 features, fill invalid with 1.0, pad time (processors.py:242-266, 192-207)
__global__ __launch_bounds__(256) void fbank_stack_kernel(const float* __restrict__ logmel, const float* __restrict__ fmask,
                                                          const float* __restrict__ stats, float* __restrict__ feats,
                                                          float* __restrict__ amask, int F, int Tp, long long total) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;  // one thread = one (b, t', 4 features)
    if (gid >= total) return;
    const int c4 = (int)(gid % 40);
    const long long bt = gid / 40;
    const long long b = bt / Tp;
    const int t = (int)(bt - b * Tp);
    const int half = c4 >= 20;
    const int bin = (c4 - half * 20) * 4;
    const int f = 2 * t + half;
    const int Fs = F / 2;  // stacked frames that exist (odd trailing frame dropped)
    f4 o = {1.f, 1.f, 1.f, 1.f};
    const bool exists = t < Fs;
    if (exists && fmask[b * F + f] != 0.f) {
        const f4 x = *reinterpret_cast<const f4*>(logmel + (b * F + f) * 80 + bin);
        const f4 mu = *reinterpret_cast<const f4*>(stats + (b * 2 + 0) * 80 + bin);
        const f4 var = *reinterpret_cast<const f4*>(stats + (b * 2 + 1) * 80 + bin);
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = (x[k] - mu[k]) / sqrtf(var[k] + 1e-7f);
    }
    *reinterpret_cast<f4*>(feats + bt * 160 + c4 * 4) = o;
    if (c4 == 0) amask[bt] = (exists && fmask[b * F + 2 * t] != 0.f) ? 1.0f : 0.0f;
}

int launch_fbank_normalize(const float* logmel, const float* fmask, float* stats, float* feats, float* amask, int B, int F, int Tp,
                           hipStream_t stream) {
    hipLaunchKernelGGL(fbank_stats_kernel, dim3(B), dim3(320), 0, stream, logmel, fmask, stats, F);
    AT_CHECK_HIP(hipGetLastError());
    const long long total = (long long)B * Tp * 40;
    hipLaunchKernelGGL(fbank_stack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, logmel, fmask, stats,
                       feats, amask, F, Tp, total);
    AT_CHECK_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------------
// LayerNorm over the last dim (eps 1e-5), one wave per row, two-pass moments in registers.
// y = ((x*rstd) + (-rstd*mean)) * gamma + beta  (operation order of torch's CPU LayerNorm kernel);
// gamma/beta null -> non-affine. Optional row mask: masked rows are written as zeros.
// D must be a multiple of 4 and <= 1024. In-place (y == x) is safe: a wave reads its whole row first.
// ------------------------------------------------------------------------------------------------------
template <int MAXV>  // float4 per lane
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const float* __restrict__ row_mask,
                                                        float* __restrict__ y, long long rows, int D) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nv = D >> 2;
    const f4* xr = reinterpret_cast<const f4*>(x + row * D);
    f4 v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int c = lane + 64 * j;
        v[j] = c < nv ? xr[c] : f4{0.f, 0.f, 0.f, 0.f};
        s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    const float mean = s / (float)D;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int c = lane + 64 * j;
        if (c < nv) {
            const f4 d = v[j] - mean;
            q += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) q += __shfl_xor(q, off);
    const float rstd = 1.0f / sqrtf(q / (float)D + 1e-5f);
    const float shift = -rstd * mean;
    const bool zero = row_mask && row_mask[row] == 0.f;
    f4* yr = reinterpret_cast<f4*>(y + row * D);
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int c = lane + 64 * j;
        if (c >= nv) continue;
        f4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = fmaf(v[j][k], rstd, shift);
        if (gamma) {
            const f4 g = reinterpret_cast<const f4*>(gamma)[c];
            const f4 bb = reinterpret_cast<const f4*>(beta)[c];
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = fmaf(o[k], g[k], bb[k]);
        }
        if (zero) o = f4{0.f, 0.f, 0.f, 0.f};
        yr[c] = o;
    }
}

int launch_layernorm(const float* x, const float* gamma, const float* beta, const float* row_mask, float* y, long long rows, int D,
                     hipStream_t stream) {
    AT_REQUIRE(D % 4 == 0 && D <= 1024 && D > 0, "LayerNorm width must be a multiple of 4, <= 1024");
    if (rows <= 0) return 0;
    const unsigned blocks = (unsigned)((rows + 3) / 4);
    if (D <= 256)
        hipLaunchKernelGGL(layernorm_kernel<1>, dim3(blocks), dim3(256), 0, stream, x, gamma, beta, row_mask, y, rows, D);
    else
        hipLaunchKernelGGL(layernorm_kernel<4>, dim3(blocks), dim3(256), 0, stream, x, gamma, beta, row_mask, y, rows, D);
    AT_CHECK_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------------
// LayerNorm(1024) whose output goes straight to a split GEMM: the normalised row is written as the K-blocked operand pieces
// [NP][64][rows_pad][16] (gemm_bf16x3.h) instead of (or, for the post-LN HuBERT layers, besides) fp32 — the separate split pass
// and one fp32 round trip of the activation disappear (16 -> 8 bytes of HBM traffic per element). Same arithmetic as layernorm_kernel.
// Workgroup = 16 consecutive rows (4 waves x 4 rows); the pieces go through LDS so that every (piece, k-block) is written as one
// contiguous 512-byte run (16 rows x 32 B) — a wave writing its own row directly scatters 32-byte segments (measured slower in round 1).
// ------------------------------------------------------------------------------------------------------
constexpr int LNS_ROWS = 8, LNS_D = 1024;    // rows per workgroup: 2 per wave (32 KB of LDS with two pieces: four workgroups per CU)
constexpr int LNS_RW = LNS_ROWS / 4;          // rows per wave

// Which 8-row group workgroup b takes (round 6, an order-only change: every group is independent). The GEMM before a LayerNorm leaves the END of each XCD's
// m-range freshest in that XCD's L2 and in the Infinity Cache (gemm_f16x2_tg.hip: XCD x owns the x-th eighth of the rows and walks it upwards), and the GEMM
// after it starts at the BEGINNING of each eighth. So workgroup b — dispatched to XCD b & 7 — walks the (b & 7)-th eighth DOWNWARDS: it reads what was written
// last first (before its own 8 bytes per element of traffic evict it) and writes last what is read first next.
__device__ __forceinline__ long long lns_group(unsigned b, unsigned nb) {
    const unsigned per = nb >> 3;
    if (b >= (per << 3)) return b;                    // the nb % 8 groups at the end keep their place
    return (long long)((b & 7u) + 1u) * per - 1 - (b >> 3);
}

// DOUBLE: two LayerNorms back to back — y = LN(x; gamma, beta) is written as fp32 rows (WRITE_Y) and the pieces are split(LN(y; gamma2, beta2)). That is the
// conformer's final_layer_norm followed by the next layer's ffn1_layer_norm (w2vbert.hip): one pass over the residual stream instead of two (y is not read
// back). Each LayerNorm reduces exactly as layernorm_kernel / the single form do (same lane mapping, same sums): bit-identical to the two launches.
// D: 1024 (conformer) or 768 (HuBERT, round 5: its post-LN layers write the fp32 residual stream AND the next GEMM's pieces in one pass); D / 256 float4 per lane
template <class SC, bool WRITE_Y, bool DOUBLE = false, int D = LNS_D>
__global__ __launch_bounds__(256) void layernorm_split_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              const float* __restrict__ row_mask, float* __restrict__ y, typename SC::T* __restrict__ out,
                                                              long long rows, long long rows_pad, float scale, int* __restrict__ status,
                                                              const float* __restrict__ gamma2 = nullptr, const float* __restrict__ beta2 = nullptr) {
    typedef typename SC::T PT;
    typedef typename SC::V4 V4;
    constexpr int NP = SC::NP;
    // [piece][k-block][8 rows x 16 + 16 pad]: with the natural 256-byte k-block stride all 16 k-blocks a wave writes at once fell on the same banks
    // (PMC, round 4: 47 % of this kernel's LDS cycles were conflict replays); + 32 bytes per k-block spreads them over all 64 banks
    constexpr int KB_LD = LNS_ROWS * 16 + 16;
    constexpr int NJ = D / 256;
    static_assert(D % 256 == 0 && NJ >= 1 && NJ <= 4, "layernorm_split: D = 256 .. 1024 in steps of 256");
    __shared__ __attribute__((aligned(16))) PT tile[NP][D / 16][KB_LD];   // 36 KB (two pieces) / 54 KB (three)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long r0 = lns_group(blockIdx.x, gridDim.x) * LNS_ROWS;
    RangeMax over;
    // all rows of the wave are loaded before the first is reduced (one row at a time left 4 KB per wave in flight: 3.5 TB/s, 70 % of the wave
    // cycles waiting); 8 rows per workgroup instead of 16 puts four workgroups on a CU: 16.0 -> 13.7 ms per semantic_m step (4 rows: no further gain)
    f4 vall[LNS_RW][NJ];
#pragma unroll
    for (int rr = 0; rr < LNS_RW; ++rr) {
        const long long row = r0 + wave * LNS_RW + rr;
        if (row < rows) {
            const f4* xr = reinterpret_cast<const f4*>(x + row * D);
#pragma unroll
            for (int j = 0; j < NJ; ++j) vall[rr][j] = xr[lane + 64 * j];
        } else {
#pragma unroll
            for (int j = 0; j < NJ; ++j) vall[rr][j] = f4{0.f, 0.f, 0.f, 0.f};
        }
    }
#pragma unroll
    for (int rr = 0; rr < LNS_RW; ++rr) {
        const int lr = wave * LNS_RW + rr;
        const long long row = r0 + lr;
        f4 v[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) v[j] = vall[rr][j];
        const bool zero = row >= rows || (row_mask && row_mask[row] == 0.f);
        // v -> LayerNorm(v; g, bb) in place (torch's CPU operation order, as layernorm_kernel)
        auto normalize = [&](const float* g_, const float* b_) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < NJ; ++j) s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
            const float mean = s / (float)D;
            float q = 0.f;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const f4 d = v[j] - mean;
                q += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) q += __shfl_xor(q, off);
            const float rstd = 1.0f / sqrtf(q / (float)D + 1e-5f);
            const float shift = -rstd * mean;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int c = lane + 64 * j;           // float4 index: columns 4c .. 4c + 3 = k-block c / 4, quarter c % 4
                f4 o;
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = fmaf(v[j][k], rstd, shift);
                if (g_) {
                    const f4 g = reinterpret_cast<const f4*>(g_)[c];
                    const f4 bb = reinterpret_cast<const f4*>(b_)[c];
#pragma unroll
                    for (int k = 0; k < 4; ++k) o[k] = fmaf(o[k], g[k], bb[k]);
                }
                if (zero) o = f4{0.f, 0.f, 0.f, 0.f};
                v[j] = o;
            }
        };
        normalize(gamma, beta);
        if constexpr (WRITE_Y)
            if (row < rows) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) reinterpret_cast<f4*>(y + row * D)[lane + 64 * j] = v[j];
            }
        if constexpr (DOUBLE) normalize(gamma2, beta2);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int c = lane + 64 * j;
            V4 p[NP];
            over |= split4<SC>(v[j], scale, p);
#pragma unroll
            for (int i = 0; i < NP; ++i) *reinterpret_cast<V4*>(&tile[i][c >> 2][lr * 16 + (c & 3) * 4]) = p[i];
        }
    }
    __syncthreads();
    // (piece, k-block) = LNS_ROWS rows x 32 B contiguous = 2 LNS_ROWS chunks of 16 B: thread -> chunk
    typedef unsigned int u4_ __attribute__((ext_vector_type(4)));
    const long long ps = rows_pad * (long long)D;
    constexpr int CH = 2 * LNS_ROWS;
    constexpr int KB = D / 16;
    for (int e = threadIdx.x; e < NP * KB * CH; e += 256) {
        const int ch = e % CH, kb = (e / CH) % KB, pi = e / (CH * KB);
        const int lr = ch >> 1;
        if (r0 + lr < rows_pad)
            *reinterpret_cast<u4_*>(out + pi * ps + ((long long)kb * rows_pad + r0 + lr) * 16 + (ch & 1) * 8) =
                *reinterpret_cast<const u4_*>(&tile[pi][kb][lr * 16 + (ch & 1) * 8]);
    }
    if constexpr (SC::RANGE_CHECK)
        range_publish(status, status ? status + 1 : nullptr, over);
}

// y = LN(x; gamma, beta) as fp32 rows (y == x allowed: a wave reads its rows first) and out = split(LN(y; gamma2, beta2)): layernorm_split_kernel<.., DOUBLE>
int launch_layernorm2_split(const float* x, const float* gamma, const float* beta, float* y, const float* gamma2, const float* beta2, __bf16* out, long long rows,
                            long long rows_pad, int D, int scheme, float scale, int* status, hipStream_t stream) {
    AT_REQUIRE(D == LNS_D && rows_pad >= rows && rows_pad % LNS_ROWS == 0 && out && y, "layernorm2_split: D must be 1024, rows_pad a multiple of 8");
    const unsigned blocks = (unsigned)(rows_pad / LNS_ROWS);
    if (scheme == XB_SCHEME_F16X2)
        hipLaunchKernelGGL((layernorm_split_kernel<SchemeF16x2, true, true>), dim3(blocks), dim3(256), 0, stream, x, gamma, beta, (const float*)nullptr, y,
                           reinterpret_cast<_Float16*>(out), rows, rows_pad, scale, status, gamma2, beta2);
    else
        hipLaunchKernelGGL((layernorm_split_kernel<SchemeBf16x3, true, true>), dim3(blocks), dim3(256), 0, stream, x, gamma, beta, (const float*)nullptr, y, out, rows,
                           rows_pad, scale, status, gamma2, beta2);
    AT_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_layernorm_split(const float* x, const float* gamma, const float* beta, const float* row_mask, float* y, __bf16* out, long long rows, long long rows_pad,
                           int D, int scheme, float scale, int* status, hipStream_t stream) {
    AT_REQUIRE((D == LNS_D || D == 768) && rows_pad >= rows && rows_pad % LNS_ROWS == 0 && out, "layernorm_split: D must be 1024 or 768, rows_pad a multiple of 8");
    const unsigned blocks = (unsigned)(rows_pad / LNS_ROWS);
    if (D == 768) {   // HuBERT: always with the fp32 rows (the post-LN residual stream)
        AT_REQUIRE(y != nullptr, "layernorm_split: the 768-wide form writes fp32 rows as well");
        if (scheme == XB_SCHEME_F16X2)
            hipLaunchKernelGGL((layernorm_split_kernel<SchemeF16x2, true, false, 768>), dim3(blocks), dim3(256), 0, stream, x, gamma, beta, row_mask, y,
                               reinterpret_cast<_Float16*>(out), rows, rows_pad, scale, status, (const float*)nullptr, (const float*)nullptr);
        else
            hipLaunchKernelGGL((layernorm_split_kernel<SchemeBf16x3, true, false, 768>), dim3(blocks), dim3(256), 0, stream, x, gamma, beta, row_mask, y, out, rows,
                               rows_pad, scale, status, (const float*)nullptr, (const float*)nullptr);
        AT_CHECK_HIP(hipGetLastError());
        return 0;
    }
    if (scheme == XB_SCHEME_F16X2) {
        _Float16* o = reinterpret_cast<_Float16*>(out);
        if (y) hipLaunchKernelGGL((layernorm_split_kernel<SchemeF16x2, true>), dim3(blocks), dim3(256), 0, stream, x, gamma, beta, row_mask, y, o, rows, rows_pad, scale, status, (const float*)nullptr, (const float*)nullptr);
        else hipLaunchKernelGGL((layernorm_split_kernel<SchemeF16x2, false>), dim3(blocks), dim3(256), 0, stream, x, gamma, beta, row_mask, y, o, rows, rows_pad, scale, status, (const float*)nullptr, (const float*)nullptr);
    } else {
        if (y) hipLaunchKernelGGL((layernorm_split_kernel<SchemeBf16x3, true>), dim3(blocks), dim3(256), 0, stream, x, gamma, beta, row_mask, y, out, rows, rows_pad, scale, status, (const float*)nullptr, (const float*)nullptr);
        else hipLaunchKernelGGL((layernorm_split_kernel<SchemeBf16x3, false>), dim3(blocks), dim3(256), 0, stream, x, gamma, beta, row_mask, y, out, rows, rows_pad, scale, status, (const float*)nullptr, (const float*)nullptr);
    }
    AT_CHECK_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------------
// Relative-position self-attention (reference audiotoken/modeling_wav2vec2_bert.py:46-73), flash style, fp32 MFMA.
//   scores[l][r] = q_l.k_r / 8 + (q_l . E[clamp(r-l,-64,8)+64]) / 8 + (key r padded ? finfo.min : 0);  softmax;  . v
// The reference materialises the [B,16,T,T] bias with an einsum; here q.E^T (73 buckets) is formed once per query
// tile with MFMA into LDS and gathered per score — the T x T tensor never exists.
// Workgroup = 128 queries of one (clip, head); wave = 32 queries. Key/value tiles of 64 go global -> LDS
// (K row-major + XOR swizzle; V transposed so a lane's 4 consecutive keys are one ds_read_b128).
// S^T = K.Q^T is computed with keys on the MFMA rows: a lane then holds, for ONE query, the 4 keys (4*quad+reg) of
// each 16-key tile — exactly the B-operand fragment of the P.V MFMA, so P never leaves registers.
// qkv layout: [B*T][3072] = [q | k | v], head h at columns h*64.
// ------------------------------------------------------------------------------------------------------
constexpr bool kAttnX3Default = true;   // split-bf16 attention kernel (attention_bf16x3.hip)
constexpr int ATT_QB = 128, ATT_KB = 64, ATT_D = 64;
constexpr float ATT_SCALE2 = 0.125f * 1.4426950408889634f;   // 1/sqrt(64) * log2(e): scores live in the exp2 domain (p = v_exp_f32(s - m))
constexpr int ATT_QE_LD = 81;   // 73 buckets padded to an odd stride
constexpr int ATT_VT_LD = 68;
constexpr int ATT_LDS_FLOATS = ATT_KB * ATT_D + ATT_D * ATT_VT_LD + ATT_QB * ATT_QE_LD + ATT_KB + 4;

__global__ __launch_bounds__(256, 2) void relpos_attention_kernel(const float* __restrict__ qkv, const float* __restrict__ amask,
                                                               const float* __restrict__ dist_emb /*[80][64], rows>=73 zero; null = no rel-pos bias*/,
                                                               float* __restrict__ ctx, int T, int hid /*heads*64*/) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;                              // [64 keys][64 d], chunk ^= key&15
    float* Vt = Ks + ATT_KB * ATT_D;               // [64 dv][68]: Vt[dv][key]
    float* QE = Vt + ATT_D * ATT_VT_LD;            // [128 queries][81]: log2(e)/8 * q.E[bucket]
    float* kb = QE + ATT_QB * ATT_QE_LD;           // [64] additive key bias: 0 / finfo.min (padded) / -inf (beyond T)
    int* kb_any = reinterpret_cast<int*>(kb + ATT_KB);   // [1] any non-zero entry in kb
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, qd = lane >> 4;
    const int h = blockIdx.y, b = blockIdx.z;
    const int l0 = blockIdx.x * ATT_QB;
    const long long rowbase = (long long)b * T;
    const int LD = 3 * hid;
    const float* qp = qkv + h * 64;
    const float* kp = qkv + hid + h * 64;
    const float* vp = qkv + 2 * hid + h * 64;
    const bool relpos = dist_emb != nullptr;

    // query fragments (B operand): qf[i][c] = q[l][c*16 + qd*4 .. +3]
    int lq[2];
    f4 qf[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        lq[i] = l0 + wave * 32 + i * 16 + r16;
        const int lc = lq[i] < T ? lq[i] : T - 1;
#pragma unroll
        for (int c = 0; c < 4; ++c) qf[i][c] = *reinterpret_cast<const f4*>(qp + (rowbase + lc) * LD + c * 16 + qd * 4);
    }
    // QE = 0.125 * q . E^T  -> LDS (lane: query r16 of tile i, buckets bt*16 + qd*4 + reg)
    if (relpos)
#pragma unroll
    for (int bt = 0; bt < 5; ++bt) {
        f4 ef[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) ef[c] = *reinterpret_cast<const f4*>(dist_emb + (bt * 16 + r16) * 64 + c * 16 + qd * 4);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ef[c][e], qf[i][c][e], acc, 0, 0, 0);
            float* dst = QE + (wave * 32 + i * 16 + r16) * ATT_QE_LD + bt * 16 + qd * 4;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) dst[reg] = ATT_SCALE2 * acc[reg];
        }
    }
    // far-field constants: bucket 0 (r - l <= -64) and bucket 72 (r - l >= 8); own rows, visible after the first barrier
    f4 oacc[2][4];
    float mrun[2], lrun[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        mrun[i] = -INFINITY;
        lrun[i] = 0.f;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) oacc[i][dt] = f4{0.f, 0.f, 0.f, 0.f};
    }
    // wave-uniform values are passed through readfirstlane so the far/near and masked/unmasked choices below compile to
    // scalar branches instead of exec-mask regions
    const int wl_min = l0 + __builtin_amdgcn_readfirstlane(wave) * 32, wl_max = wl_min + 31;
    const float FMIN = -3.4028234663852886e38f;

    const int nkt = (T + ATT_KB - 1) / ATT_KB;
    // K/V staging goes global -> registers -> LDS, one tile ahead: the loads of tile kt+1 are issued before the MFMAs of
    // tile kt and written to LDS after them. All loads are unconditional (row clamped, value masked at the LDS store):
    // loads under a per-lane branch serialise behind s_waitcnt vmcnt(0).
    // Buffer descriptors over THIS clip's rows only: a key row >= T is out of range and reads as 0 (its score gets -inf from
    // kb, so any finite value would do) — no clamps, no selects; a tile fetch costs one 32-bit add per load
    // instead of clamp + 64-bit multiply-add + select (VALU time is not hidden behind the fp32 MFMA).
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    const int clip_bytes = T * LD * 4;
    const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc((void*)(kp + rowbase * LD), 0, clip_bytes - (hid + h * 64) * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc((void*)(vp + rowbase * LD), 0, clip_bytes - (2 * hid + h * 64) * 4, 0x00020000);
    const int st_key = tid >> 2, st_cg = (tid & 3) * 4;      // K tile: thread -> key, 4 chunks from st_cg
    const int st_dv = tid & 63, st_k0 = (tid >> 6) * 16;     // V tile (transposed): thread -> dv, 16 keys from st_k0
    const int k_voff = (st_key * LD + st_cg * 4) * 4;
    const int v_voff = (st_k0 * LD + st_dv) * 4;
    const int row_bytes = LD * 4;
    u4 kv[4];
    unsigned vv[16];
    float am = 0.f;
    auto prefetch = [&](int kt) {
        // the whole offset goes through the VGPR operand: only that one is range-checked against num_records
        const int toff = kt * ATT_KB * row_bytes;
        const int ko = k_voff + toff;
        int vo = v_voff + toff;
#pragma unroll
        for (int j = 0; j < 4; ++j) kv[j] = __builtin_amdgcn_raw_buffer_load_b128(krs, ko + j * 16, 0, 0);
#pragma unroll
        for (int e = 0; e < 16; ++e) { vv[e] = __builtin_amdgcn_raw_buffer_load_b32(vrs, vo, 0, 0); vo += row_bytes; }
        const int rr_m = kt * ATT_KB + (tid & 63);
        am = amask[rowbase + (rr_m < T ? rr_m : T - 1)];
    };
    prefetch(0);
    for (int kt = 0; kt < nkt; ++kt) {
        const int r0 = kt * ATT_KB;
        __syncthreads();  // previous tile fully consumed (also orders the QE stores before first use)
        {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *reinterpret_cast<u4*>(Ks + st_key * ATT_D + (((st_cg + j) ^ (st_key & 15)) << 2)) = kv[j];
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<u4*>(Vt + st_dv * ATT_VT_LD + st_k0 + g * 4) = u4{vv[g * 4], vv[g * 4 + 1], vv[g * 4 + 2], vv[g * 4 + 3]};
            if (tid < ATT_KB) {   // wave 0: additive key bias and "this tile has one" flag
                const int rr_m = r0 + tid;
                const float kbv = rr_m < T ? (am != 0.f ? 0.f : FMIN) : -INFINITY;
                kb[tid] = kbv;
                const unsigned long long anyb = __builtin_amdgcn_ballot_w64(kbv != 0.f);
                if (tid == 0) kb_any[0] = anyb != 0ull ? 1 : 0;
            }
        }
        __syncthreads();
        if (kt + 1 < nkt) prefetch(kt + 1);
        // S^T tiles: lane holds s[i][j][reg] = q_l . k_r for l = lq[i], r = r0 + j*16 + qd*4 + reg
        f4 s[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) s[i][j] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            f4 kf[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                kf[j] = *reinterpret_cast<const f4*>(Ks + (j * 16 + r16) * ATT_D + ((((c << 2) + qd) ^ r16) << 2));
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) s[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[j][e], qf[i][c][e], s[i][j], 0, 0, 0);
        }
        // bias + mask, online softmax (exp2 domain: the log2(e)/8 factor is folded into the score FMA and into QE)
        const bool far_left = (r0 + ATT_KB - 1) - wl_min <= -64;   // every (l, r) of this wave has r - l <= -64
        const bool far_right = r0 - wl_max >= 8;                   // every (l, r) has r - l >= 8
        const bool plain = far_left || far_right || !relpos;        // one bias constant per query for the whole tile
        const bool masked = __builtin_amdgcn_readfirstlane(kb_any[0]) != 0;   // uniform: some key of the tile is padded / beyond T
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float* qe = QE + (wave * 32 + i * 16 + r16) * ATT_QE_LD;
            const float c_far = relpos ? (far_left ? qe[0] : qe[72]) : 0.f;
            float mx = -INFINITY;
            if (plain && !masked) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const float sc = fmaf(ATT_SCALE2, s[i][j][reg], c_far);
                        s[i][j][reg] = sc;
                        mx = fmaxf(mx, sc);
                    }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f4 kbv = *reinterpret_cast<const f4*>(kb + j * 16 + qd * 4);
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        float bias;
                        if (plain) {
                            bias = c_far;
                        } else {
                            int dd = (r0 + j * 16 + qd * 4 + reg) - lq[i];
                            dd = dd < -64 ? -64 : (dd > 8 ? 8 : dd);
                            bias = qe[dd + 64];
                        }
                        const float t = bias + kbv[reg];
                        const float sc = fmaf(ATT_SCALE2, s[i][j][reg], t);
                        s[i][j][reg] = sc;
                        mx = fmaxf(mx, sc);
                    }
                }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float mnew = fmaxf(mrun[i], mx);
            float rs = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const float p = __builtin_amdgcn_exp2f(s[i][j][reg] - mnew);
                    s[i][j][reg] = p;
                    rs += p;
                }
            rs += __shfl_xor(rs, 16);
            rs += __shfl_xor(rs, 32);
            if (__builtin_amdgcn_ballot_w64(mnew != mrun[i]) != 0ull) {   // some query's running maximum moved: rescale
                const float alpha = __builtin_amdgcn_exp2f(mrun[i] - mnew);   // exp2(-inf) = 0 on the first tile
                lrun[i] = lrun[i] * alpha + rs;
                mrun[i] = mnew;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) oacc[i][dt] *= alpha;
            } else {
                lrun[i] += rs;
            }
        }
        // O^T += V^T . P^T : A = V^T fragment (dv rows), B = P (lane's own registers)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f4 vf[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) vf[dt] = *reinterpret_cast<const f4*>(Vt + (dt * 16 + r16) * ATT_VT_LD + j * 16 + qd * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt)
                        oacc[i][dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[dt][e], s[i][j][e], oacc[i][dt], 0, 0, 0);
        }
    }
    // lane holds O[l = lq[i]][dv = dt*16 + qd*4 + reg]
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (lq[i] >= T) continue;
        const float inv = 1.0f / lrun[i];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
            *reinterpret_cast<f4*>(ctx + (rowbase + lq[i]) * hid + h * 64 + dt * 16 + qd * 4) = oacc[i][dt] * inv;
    }
}

static int default_attention_arith() {
    if (const char* e = std::getenv("AUDIOTOKEN_SEMANTIC_ARITH")) {
        const std::string v(e);
        return v == "f32" ? 0 : v == "bf16x3" ? 1 : 2;
    }
    return kAttnX3Default ? 2 : 0;
}

int launch_relpos_attention(const float* qkv, const float* amask, const float* dist_emb, float* ctx, int B, int T,
                            hipStream_t stream, int heads, int arith, int* status, __bf16* ctx_pieces, long long rows_pad, const __bf16* kv_pieces, int w8,
                            const __bf16* dist_pieces, float dist_scale) {
    static const int dflt = default_attention_arith();
    if (arith < 0) arith = dflt;
    if (arith > 0) return launch_relpos_attention_x3(qkv, amask, dist_emb, ctx, B, T, stream, heads, arith == 2 ? 1 : 0, status, ctx_pieces, rows_pad, kv_pieces, w8, dist_pieces, dist_scale);
    AT_REQUIRE(ctx_pieces == nullptr && kv_pieces == nullptr, "relpos_attention: piece input / output needs the split kernels");
    dim3 grid((T + ATT_QB - 1) / ATT_QB, heads, B);
    const size_t lds = ATT_LDS_FLOATS * sizeof(float);
    { static LdsAttrFlags lds_attr_0; if (int rc = set_max_dynamic_lds(lds_attr_0, relpos_attention_kernel, lds)) return rc; }
    hipLaunchKernelGGL(relpos_attention_kernel, grid, dim3(256), lds, stream, qkv, amask, dist_emb, ctx, T, heads * 64);
    AT_CHECK_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------------
// Conformer conv-module middle: causal depthwise conv k=31 (left zero pad 30 within the clip) -> LayerNorm(1024)
// -> swish (HF modeling_wav2vec2_bert.py:212-222). Input g = GLU output [B*T][1024] (from the EPI_GLU GEMM).
// Workgroup = 8 consecutive time steps of one clip x all 1024 channels; thread = 4 channels. Each input row is
// read once per workgroup (38 rows for 8 outputs) instead of 31 times; tap weights [31][1024] stay in registers.
// ------------------------------------------------------------------------------------------------------
constexpr int DW_K = 31;

// SC = void: fp32 output [B*T][1024]; SC = an operand scheme: the output goes straight to the pointwise-conv-2 GEMM as K-blocked pieces
// [NP][64][rows_pad][16] (thread = 4 channels = one quarter of a k-block row: an 8-byte store per piece)
template <class SC, int DW_TT>
__global__ __launch_bounds__(256) void dwconv_ln_swish_kernel(const float* __restrict__ g, const float* __restrict__ w /*[31][1024]*/,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              float* __restrict__ out, int T, void* __restrict__ pieces, long long rows_pad, float scale,
                                                              int* __restrict__ status) {
    __shared__ float red[2][4][DW_TT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * DW_TT;
    const long long base = (long long)b * T;
    f4 wt[DW_K];
#pragma unroll
    for (int j = 0; j < DW_K; ++j) wt[j] = reinterpret_cast<const f4*>(w + j * 1024)[tid];
    f4 acc[DW_TT];
#pragma unroll
    for (int i = 0; i < DW_TT; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < DW_TT + DW_K - 1; ++r) {   // input row t0 - 30 + r
        const int t = t0 - (DW_K - 1) + r;
        f4 x = {0.f, 0.f, 0.f, 0.f};
        if (t >= 0 && t < T) x = reinterpret_cast<const f4*>(g + (base + t) * 1024)[tid];
#pragma unroll
        for (int i = 0; i < DW_TT; ++i) {
            const int tap = r - i;   // out[t0+i] uses in[t0+i-30+tap]
            if (tap >= 0 && tap < DW_K) acc[i] += wt[tap] * x;
        }
    }
    // LayerNorm over 1024 channels for each of the 8 rows
    float s[DW_TT];
#pragma unroll
    for (int i = 0; i < DW_TT; ++i) {
        s[i] = (acc[i].x + acc[i].y) + (acc[i].z + acc[i].w);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) s[i] += __shfl_xor(s[i], off);
    }
    if (lane == 0)
#pragma unroll
        for (int i = 0; i < DW_TT; ++i) red[0][wave][i] = s[i];
    __syncthreads();
    float mean[DW_TT], q[DW_TT];
#pragma unroll
    for (int i = 0; i < DW_TT; ++i) {
        mean[i] = ((red[0][0][i] + red[0][1][i]) + (red[0][2][i] + red[0][3][i])) * (1.0f / 1024.0f);
        const f4 d = acc[i] - mean[i];
        q[i] = (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) q[i] += __shfl_xor(q[i], off);
    }
    if (lane == 0)
#pragma unroll
        for (int i = 0; i < DW_TT; ++i) red[1][wave][i] = q[i];
    __syncthreads();
    const f4 gm = reinterpret_cast<const f4*>(gamma)[tid];
    const f4 bt = reinterpret_cast<const f4*>(beta)[tid];
    RangeMax over;
    (void)over;
#pragma unroll
    for (int i = 0; i < DW_TT; ++i) {
        const int t = t0 + i;
        if (t >= T) continue;
        const float var = ((red[1][0][i] + red[1][1][i]) + (red[1][2][i] + red[1][3][i])) * (1.0f / 1024.0f);
        const float rstd = 1.0f / sqrtf(var + 1e-5f);
        const float shift = -rstd * mean[i];
        f4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = swishf_(fmaf(fmaf(acc[i][k], rstd, shift), gm[k], bt[k]));
        if constexpr (std::is_void<SC>::value) {
            reinterpret_cast<f4*>(out + (base + t) * 1024)[tid] = o;
        } else {
            over |= store_pieces4<SC>(reinterpret_cast<typename SC::T*>(pieces), rows_pad * 1024, rows_pad, base + t, tid * 4, o, scale);
        }
    }
    if constexpr (!std::is_void<SC>::value)
        if constexpr (SC::RANGE_CHECK)
            range_publish(status, status ? status + 1 : nullptr, over);
}

int launch_dwconv_ln_swish(const float* g, const float* w, const float* gamma, const float* beta, float* out, int B, int T,
                           hipStream_t stream, __bf16* pieces, long long rows_pad, int scheme, float scale, int* status) {
    constexpr int TT = 16;   // output rows per workgroup: 46 input rows per 16 outputs (8 rows: 38 per 8 — 1.4 ms more per semantic_m step, same box)
    dim3 grid((T + TT - 1) / TT, B);
    if (pieces && scheme == XB_SCHEME_F16X2)
        hipLaunchKernelGGL((dwconv_ln_swish_kernel<SchemeF16x2, TT>), grid, dim3(256), 0, stream, g, w, gamma, beta, out, T, (void*)pieces, rows_pad, scale, status);
    else if (pieces)
        hipLaunchKernelGGL((dwconv_ln_swish_kernel<SchemeBf16x3, TT>), grid, dim3(256), 0, stream, g, w, gamma, beta, out, T, (void*)pieces, rows_pad, scale, status);
    else
        hipLaunchKernelGGL((dwconv_ln_swish_kernel<void, TT>), grid, dim3(256), 0, stream, g, w, gamma, beta, out, T, nullptr, 0, 1.0f, nullptr);
    AT_CHECK_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------------
// VQ assignment from the score GEMM: dots[m][n] = x_m . e_n. vector_quantize_pytorch eval:
//   idx = argmax_n -sqrt(max((|x|^2 + |e_n|^2) + (-2 dots), 0)), first maximal index. One wave per row.
// REFINEMENT (round 5, `codebook` non-null). The expanded form cancels: with centres fitted to the data (what a trained quantiser has) |x|^2 + |e|^2 ~ 2048
// against distances of ~1, so the GEMM's fp32 accumulation error of x . e (~2e-4 .. 5e-4 at |x . e| ~ 1000: 32 .. 256 sequential accumulator updates) becomes
// ~1e-3 of the DISTANCE — and on hidden states with massive channels the shipped kernel disagreed with float64 at 29 of 1000 positions at margins >= 1e-3
// although the hidden states themselves were as close to float64 as the reference's (tools/cond_probe.py; torch's blocked sgemm on the CPU loses 3-5 x less).
// So the scan only SHORTLISTS: every code whose approximate squared distance is within (|x|^2 + |e_best|^2) 2^-17 of the best (>= 30 x the GEMM's error) is
// re-evaluated as sum_k (x_k - e_k)^2 — no cancellation, differences and sum in float64, the wave cooperating on one code at a time — and the smallest exact
// distance wins, ties to the lower index. Where the shortlist has one entry (the rule, away from near-ties) nothing changes; at near-ties the id is the one
// exact arithmetic gives, which is the reference's wherever the reference's own fp32 quantiser is decisive (its error is ~1e-4 of the squared distance).
// ------------------------------------------------------------------------------------------------------
template <int MAXQ>   // float4 of the row per lane: D <= 256 * MAXQ
__global__ __launch_bounds__(256) void vq_argmax_kernel(const float* __restrict__ x, const float* __restrict__ dots,
                                                        const float* __restrict__ e2, int16_t* __restrict__ out, long long rows,
                                                        int D, int C, int* __restrict__ status, int ld, const float* __restrict__ codebook) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    f4 xr[MAXQ];
    float x2 = 0.f;
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) {
        const int c = lane + 64 * q;
        xr[q] = c < (D >> 2) ? reinterpret_cast<const f4*>(x + row * D)[c] : f4{0.f, 0.f, 0.f, 0.f};
        x2 += (xr[q].x * xr[q].x + xr[q].y * xr[q].y) + (xr[q].z * xr[q].z + xr[q].w * xr[q].w);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) x2 += __shfl_xor(x2, off);
    if (status && lane == 0 && !(x2 <= 3.0e38f)) atomicOr(status, XB_STATUS_NONFINITE);   // a NaN / infinity anywhere upstream ends up in these rows
    float best = -INFINITY, second = -INFINITY;   // `second`: this lane's runner-up (round 6) — decides whether the refinement's second pass can find anything
    int bidx = 0;
    for (int c = lane; c < (C >> 2); c += 64) {   // increasing n per lane: strict > keeps the first index
        const f4 d = reinterpret_cast<const f4*>(dots + row * ld)[c];
        const f4 y2 = reinterpret_cast<const f4*>(e2)[c];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float d2 = __fadd_rn(__fadd_rn(x2, y2[k]), -2.0f * d[k]);
            const float v = -sqrtf(fmaxf(d2, 0.f));
            if (v > best) { second = best; best = v; bidx = c * 4 + k; }
            else second = fmaxf(second, v);
        }
    }
    const float lane_best = best;
    const int lane_bidx = bidx;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ob = __shfl_xor(best, off);
        const int oi = __shfl_xor(bidx, off);
        if (ob > best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
    }
    if (codebook && best > -INFINITY) {
        // shortlist: approximate d^2 <= best d^2 + window (uniform values: every lane holds the reduced best)
        const float best_d2 = best * best;                                   // (= max(d2, 0) of the winner up to one rounding of the square root)
        const float window = (x2 + e2[bidx]) * (1.0f / 131072.0f);
        // exact squared distance of code n, the wave cooperating on its row (uniform n)
        auto exact = [&](int n) {
            const float* er = codebook + (long long)n * D;
            double acc = 0.0;
#pragma unroll
            for (int q = 0; q < MAXQ; ++q) {
                const int cc = lane + 64 * q;
                if (cc < (D >> 2)) {
                    const f4 ev = reinterpret_cast<const f4*>(er)[cc];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const double df = (double)xr[q][t] - (double)ev[t];
                        acc = fma(df, df, acc);
                    }
                }
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
            return acc;
        };
        // Is there ANY other code inside the window? A lane's candidates are its own best (unless that is the winner) and its runner-up; v^2 equals the clamped
        // d^2 up to the rounding of the square root (<= 2^-23 relative), so the test below is the second pass's own test widened by 2^-19: rows without a
        // near-tie (the rule) skip the second read of their dots row and the float64 work altogether; rows with one run the exact test as before (bit-identical).
        const float lim = (best_d2 + window) * (1.0f + 1.0f / 524288.0f);
        const float cand = lane_bidx == bidx ? second : lane_best;
        if (!__any(cand * cand <= lim)) {
            if (lane == 0) out[row] = (int16_t)bidx;
            return;
        }
        double ex_best = exact(bidx);                                        // the approximate winner first: it is always a candidate
        int ex_idx = bidx;
        // a degenerate code book (hundreds of near-duplicate rows) or a degenerate vector could shortlist every code: at most VQ_REFINE_MAX exact evaluations
        // per row, in index order — beyond that the best of what was evaluated stands (deterministic; never worse than the unrefined choice)
        constexpr int VQ_REFINE_MAX = 96;
        int evaluated = 1;
        for (int c0 = 0; c0 < (C >> 2) && evaluated < VQ_REFINE_MAX; c0 += 64) {
            const int c = c0 + lane;
            unsigned mask4 = 0;
            if (c < (C >> 2)) {
                const f4 d = reinterpret_cast<const f4*>(dots + row * ld)[c];
                const f4 y2 = reinterpret_cast<const f4*>(e2)[c];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float d2 = fmaxf(__fadd_rn(__fadd_rn(x2, y2[k]), -2.0f * d[k]), 0.f);
                    if (d2 <= best_d2 + window && c * 4 + k != bidx) mask4 |= 1u << k;
                }
            }
            unsigned long long pending = __ballot(mask4 != 0);
            while (pending && evaluated < VQ_REFINE_MAX) {                    // one shortlisted code at a time, the whole wave on its row
                const int src = __ffsll((long long)pending) - 1;
                const unsigned m = (unsigned)__shfl((int)mask4, src);
                for (int k = 0; k < 4 && evaluated < VQ_REFINE_MAX; ++k) {
                    if (!((m >> k) & 1u)) continue;
                    const int n = (c0 + src) * 4 + k;
                    const double acc = exact(n);
                    ++evaluated;
                    if (acc < ex_best || (acc == ex_best && n < ex_idx)) { ex_best = acc; ex_idx = n; }
                }
                pending &= pending - 1;
            }
        }
        bidx = ex_idx;
    }
    if (lane == 0) out[row] = (int16_t)bidx;
}

int launch_vq_argmax(const float* x, const float* dots, const float* e2, int16_t* out, long long rows, int D, int C,
                     hipStream_t stream, int* status, int ld, const float* codebook) {
    if (rows <= 0) return 0;
    AT_REQUIRE(D % 4 == 0 && D <= 1024, "vq_argmax: D must be a multiple of 4, <= 1024");
    const dim3 grid((unsigned)((rows + 3) / 4));
    if (D <= 768)
        hipLaunchKernelGGL(vq_argmax_kernel<3>, grid, dim3(256), 0, stream, x, dots, e2, out, rows, D, C, status, ld > 0 ? ld : C, codebook);
    else
        hipLaunchKernelGGL(vq_argmax_kernel<4>, grid, dim3(256), 0, stream, x, dots, e2, out, rows, D, C, status, ld > 0 ? ld : C, codebook);
    AT_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace at
