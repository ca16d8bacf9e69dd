// Device side of the input path of encode_batch_files: raw PCM of the decoded files -> the [B][segment_length] float32 batch + mask the
// encoders take, in ONE kernel — sample format conversion, the per-chunk windowed-sinc resampling of reference audiotoken/utils.py:82-98
// (torchaudio.transforms.Resample defaults: sinc_interp_hann, lowpass_filter_width 6, rolloff 0.99 — restated in audiotoken_amd/audio_io.py, which also
// builds the kernel table this file consumes), the fixed-length segmentation, zero padding and mask of reference audiotoken/datasets.py:75-105.
// The host keeps only what needs no samples: headers, lengths, the chunk / segment index arithmetic and the AudioConfig bookkeeping (feeder.py).
//
// Why on the device: one MI355X encodes 15 k audio-seconds per second (0.25-0.37 G samples/s). On the host the reference's per-chunk Resample is a
// conv1d with a [new/g][1][2 width + orig/g] kernel — 475 taps per output for 44.1 -> 16 kHz, all but ~36 of them exact zeros — i.e. ~10^11 MAC/s to feed
// one GPU, plus three float32 passes over every sample (convert, pad, collate). Here the host moves 2 bytes per sample once (int16 PCM, pinned, H2D) and the
// kernel reads each source sample from L2 and writes each output sample once: HBM-bound by construction (4 B written + 4 B mask per output sample).
//
// Arithmetic. Native rate: x = pcm * scale (exact: scale is a power of two for integer PCM). Resampled: y[f * n + p] = sum_k K[p][k] * xpad[f * o + k],
// xpad = the chunk with `width` zeros in front (and zeros behind), o = orig / g, n = new / g — the sum runs over the NON-ZERO taps of phase p only
// ([lo_p, hi_p), tabulated by the host: a zero weight contributes an exact zero), in ascending k with one fmaf each. torch's conv1d on the host sums the same
// products in its own (blocked) order, so the two differ by fp32 summation order only: tests/test_feeder_gpu.py bounds it (<= 1e-6 on [-1, 1) signals).
// Every streamed chunk is resampled on its own, exactly like the reference (chunk seams are part of "the same tokens").
#include "at_common.h"
#include "../../include/audiotoken_hip.h"

namespace at {

// mirrors `at_segment_desc` (include/audiotoken_hip.h)
struct SegDesc {
    const void* pcm;        // device: the file's samples, channel 0, `fmt` elements
    const float* table;     // device: resampling table [n][kw] followed by int32 [n][2] = {lo, hi} non-zero tap range per phase; null = native rate
    long long chunk_off;    // first sample of this streamed chunk inside pcm
    int chunk_len;          // source samples in the chunk
    int out_start;          // first output sample of this segment inside the RESAMPLED chunk
    int valid_len;          // output samples that exist (the rest of the row is padding)
    int fmt;                // AT_PCM_*
    float scale;            // multiplies integer samples (1 / 32768, 1 / 2^31, 1 / 2^(bits - 1)); ignored for float32
    int o, n, width;        // resampling ratio and half kernel width in source samples (o == n: native)
    int chunk_out_len;      // samples of the whole RESAMPLED chunk (what a per-chunk transform such as hubert_processor takes its moments over)
};

__device__ __forceinline__ float pcm_load(const void* pcm, int fmt, long long i, float scale) {
    switch (fmt) {
        case AT_PCM_S16: return (float)static_cast<const short*>(pcm)[i] * scale;
        case AT_PCM_S32: return (float)static_cast<const int*>(pcm)[i] * scale;
        case AT_PCM_U8: return ((float)static_cast<const unsigned char*>(pcm)[i] - 128.0f) * scale;
        default: return static_cast<const float*>(pcm)[i];
    }
}

// sample j of the (converted, resampled) streamed chunk a descriptor is cut from
__device__ __forceinline__ float chunk_sample(const SegDesc& d, const int* range, int kw, int j) {
    if (!d.table) return pcm_load(d.pcm, d.fmt, d.chunk_off + j, d.scale);
    const int f = j / d.n, p = j - f * d.n;
    const float* w = d.table + (long long)p * kw;
    const int lo = range[2 * p], hi = range[2 * p + 1];
    const long long base = (long long)f * d.o - d.width;      // source index of tap 0
    float acc = 0.f;
    for (int k = lo; k < hi; ++k) {
        const long long s = base + k;
        const float x = (s >= 0 && s < d.chunk_len) ? pcm_load(d.pcm, d.fmt, d.chunk_off + s, d.scale) : 0.f;
        acc = fmaf(w[k], x, acc);
    }
    return acc;
}

// one workgroup = 1024 consecutive output samples of one segment (4 per thread: 16-byte stores). `stats` (nullable): per segment {mean, sqrt(var + eps)} of
// the segment's whole streamed chunk — the reference's per-chunk transform of Tokenizers.semantic_s (hubert_processor, audiotoken/encoder.py:20-26 applied at
// datasets.py:78-79 BEFORE the chunk is cut and padded): valid samples become (x - mean) / sqrt(var + eps), padding stays pad_value.
__global__ __launch_bounds__(256) void pcm_segments_kernel(const SegDesc* __restrict__ descs, int seg_len, float pad_value, float* __restrict__ out,
                                                           float* __restrict__ mask, const float2* __restrict__ stats) {
    const SegDesc d = descs[blockIdx.y];
    const int t0 = ((int)blockIdx.x * 256 + (int)threadIdx.x) * 4;
    if (t0 >= seg_len) return;
    float v[4], m[4];
    const int kw = 2 * d.width + d.o;
    const int* range = d.table ? reinterpret_cast<const int*>(d.table + (long long)d.n * kw) : nullptr;
    const float2 st = stats ? stats[blockIdx.y] : make_float2(0.f, 1.f);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int t = t0 + e;
        if (t < d.valid_len) {
            m[e] = 1.0f;
            v[e] = chunk_sample(d, range, kw, d.out_start + t);
            if (stats) v[e] = (v[e] - st.x) / st.y;
        } else {
            m[e] = 0.f;
            v[e] = pad_value;
        }
    }
    const long long o = (long long)blockIdx.y * seg_len + t0;
    if (t0 + 3 < seg_len && (seg_len & 3) == 0) {
        *reinterpret_cast<float4*>(out + o) = make_float4(v[0], v[1], v[2], v[3]);
        if (mask) *reinterpret_cast<float4*>(mask + o) = make_float4(m[0], m[1], m[2], m[3]);
    } else {
        for (int e = 0; e < 4 && t0 + e < seg_len; ++e) {
            out[o + e] = v[e];
            if (mask) mask[o + e] = m[e];
        }
    }
}

// ---- zero-mean / unit-variance per streamed chunk (hubert_processor) --------------------------------------------------------------------------------
// Pass 1: float64 partial sums {sum x, sum x^2} of 4096 consecutive samples of the segment's WHOLE chunk (chunk_out_len samples: also the part of it that a
// later, possibly dropped, segment holds) per workgroup, written to partials[seg][block] — no atomics: pass 2 adds them in a fixed order, so the
// moments, and with them the tokens, do not depend on scheduling. Exact products and float64 sums: the result is the correctly rounded moment to ~1e-13,
// numpy's float32 pairwise mean / var (what the reference's Wav2Vec2FeatureExtractor computes) sits within float32 summation error of it.
constexpr int MOM_PER_BLOCK = 4096;
__global__ __launch_bounds__(256) void chunk_moments_kernel(const SegDesc* __restrict__ descs, double2* __restrict__ partials, int max_blocks) {
    const SegDesc d = descs[blockIdx.y];
    const int kw = 2 * d.width + d.o;
    const int* range = d.table ? reinterpret_cast<const int*>(d.table + (long long)d.n * kw) : nullptr;
    const int j0 = (int)blockIdx.x * MOM_PER_BLOCK;
    double s = 0.0, q = 0.0;
    if (j0 < d.chunk_out_len) {
        for (int j = j0 + (int)threadIdx.x; j < min(j0 + MOM_PER_BLOCK, d.chunk_out_len); j += 256) {
            const double x = (double)chunk_sample(d, range, kw, j);
            s += x;
            q = fma(x, x, q);
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { s += __shfl_xor(s, off); q += __shfl_xor(q, off); }
    __shared__ double2 w[4];
    if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = make_double2(s, q);
    __syncthreads();
    if (threadIdx.x == 0)
        partials[(long long)blockIdx.y * max_blocks + blockIdx.x] = make_double2((w[0].x + w[1].x) + (w[2].x + w[3].x), (w[0].y + w[1].y) + (w[2].y + w[3].y));
}
// Pass 2: one wave per segment adds the partials in a fixed order; mean and the population variance E[x^2] - mean^2 in float64, rounded once
__global__ __launch_bounds__(64) void chunk_moments_finish_kernel(const SegDesc* __restrict__ descs, const double2* __restrict__ partials, int max_blocks, float eps,
                                                                  float2* __restrict__ stats) {
    const int seg = blockIdx.x, lane = threadIdx.x;
    const int L = descs[seg].chunk_out_len;
    const int nb = (L + MOM_PER_BLOCK - 1) / MOM_PER_BLOCK;
    double s = 0.0, q = 0.0;
    for (int b = lane; b < nb; b += 64) {
        const double2 p = partials[(long long)seg * max_blocks + b];
        s += p.x;
        q += p.y;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { s += __shfl_xor(s, off); q += __shfl_xor(q, off); }
    if (lane == 0) {
        const double n = (double)max(L, 1), mean = s / n;
        const double var = fmax(q / n - mean * mean, 0.0);
        // the denominator as the reference forms it in float32: sqrt(var + 1e-7)
        stats[seg] = make_float2((float)mean, sqrtf((float)var + eps));
    }
}

}  // namespace at

extern "C" {

static_assert(sizeof(at_segment_desc) == sizeof(at::SegDesc), "at_segment_desc layout");

int at_segments_from_pcm(const at_segment_desc* descs_dev, int nseg, int seg_len, float pad_value, float* segments, float* masks, at_stream_t stream) {
    using namespace at;
    AT_REQUIRE(descs_dev && segments && nseg >= 0 && seg_len >= 1, "at_segments_from_pcm: bad arguments");
    if (nseg == 0) return 0;
    AT_REQUIRE(nseg <= 65535, "at_segments_from_pcm: at most 65535 segments per call");
    dim3 grid((unsigned)((seg_len + 1023) / 1024), (unsigned)nseg);
    hipLaunchKernelGGL(pcm_segments_kernel, grid, dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const SegDesc*>(descs_dev), seg_len, pad_value, segments, masks,
                       (const float2*)nullptr);
    AT_CHECK_HIP(hipGetLastError());
    return 0;
}

size_t at_segments_zmuv_workspace_bytes(int nseg, int max_chunk_out_len) {
    if (nseg <= 0 || max_chunk_out_len <= 0) return 0;
    const size_t nb = ((size_t)max_chunk_out_len + at::MOM_PER_BLOCK - 1) / at::MOM_PER_BLOCK;
    return (size_t)nseg * (nb * sizeof(double2) + sizeof(float2)) + 64;
}

int at_segments_from_pcm_zmuv(const at_segment_desc* descs_dev, int nseg, int seg_len, int max_chunk_out_len, float pad_value, float eps, float* segments,
                              float* masks, void* workspace, size_t workspace_bytes, at_stream_t stream_) {
    using namespace at;
    AT_REQUIRE(descs_dev && segments && workspace && nseg >= 0 && seg_len >= 1 && max_chunk_out_len >= 1, "at_segments_from_pcm_zmuv: bad arguments");
    if (nseg == 0) return 0;
    AT_REQUIRE(nseg <= 65535, "at_segments_from_pcm_zmuv: at most 65535 segments per call");
    AT_REQUIRE(workspace_bytes >= at_segments_zmuv_workspace_bytes(nseg, max_chunk_out_len), "at_segments_from_pcm_zmuv: workspace too small");
    hipStream_t stream = (hipStream_t)stream_;
    const int nb = (max_chunk_out_len + MOM_PER_BLOCK - 1) / MOM_PER_BLOCK;
    double2* partials = static_cast<double2*>(workspace);
    float2* stats = reinterpret_cast<float2*>(partials + (size_t)nseg * nb);
    const SegDesc* descs = reinterpret_cast<const SegDesc*>(descs_dev);
    hipLaunchKernelGGL(chunk_moments_kernel, dim3((unsigned)nb, (unsigned)nseg), dim3(256), 0, stream, descs, partials, nb);
    hipLaunchKernelGGL(chunk_moments_finish_kernel, dim3((unsigned)nseg), dim3(64), 0, stream, descs, partials, nb, eps, stats);
    dim3 grid((unsigned)((seg_len + 1023) / 1024), (unsigned)nseg);
    hipLaunchKernelGGL(pcm_segments_kernel, grid, dim3(256), 0, stream, descs, seg_len, pad_value, segments, masks, (const float2*)stats);
    AT_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
