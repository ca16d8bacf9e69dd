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
};

__device__ __forceinline__ float pcm_load(const void* pcm, int fmt, long long i, float scale) {
    switch (fmt) {
        case AT_PCM_S16: return (float)static_cast<const short*>(pcm)[i] * scale;
        case AT_PCM_S32: return (float)static_cast<const int*>(pcm)[i] * scale;
        case AT_PCM_U8: return ((float)static_cast<const unsigned char*>(pcm)[i] - 128.0f) * scale;
        default: return static_cast<const float*>(pcm)[i];
    }
}

// one workgroup = 1024 consecutive output samples of one segment (4 per thread: 16-byte stores)
__global__ __launch_bounds__(256) void pcm_segments_kernel(const SegDesc* __restrict__ descs, int seg_len, float pad_value, float* __restrict__ out,
                                                           float* __restrict__ mask) {
    const SegDesc d = descs[blockIdx.y];
    const int t0 = ((int)blockIdx.x * 256 + (int)threadIdx.x) * 4;
    if (t0 >= seg_len) return;
    float v[4], m[4];
    const int kw = 2 * d.width + d.o;
    const int* range = d.table ? reinterpret_cast<const int*>(d.table + (long long)d.n * kw) : nullptr;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int t = t0 + e;
        if (t < d.valid_len) {
            m[e] = 1.0f;
            const int j = d.out_start + t;
            if (!d.table) {
                v[e] = pcm_load(d.pcm, d.fmt, d.chunk_off + j, d.scale);
            } else {
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
                v[e] = acc;
            }
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

}  // namespace at

extern "C" {

static_assert(sizeof(at_segment_desc) == sizeof(at::SegDesc), "at_segment_desc layout");

int at_segments_from_pcm(const at_segment_desc* descs_dev, int nseg, int seg_len, float pad_value, float* segments, float* masks, at_stream_t stream) {
    using namespace at;
    AT_REQUIRE(descs_dev && segments && nseg >= 0 && seg_len >= 1, "at_segments_from_pcm: bad arguments");
    if (nseg == 0) return 0;
    AT_REQUIRE(nseg <= 65535, "at_segments_from_pcm: at most 65535 segments per call");
    dim3 grid((unsigned)((seg_len + 1023) / 1024), (unsigned)nseg);
    hipLaunchKernelGGL(pcm_segments_kernel, grid, dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const SegDesc*>(descs_dev), seg_len, pad_value, segments, masks);
    AT_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
