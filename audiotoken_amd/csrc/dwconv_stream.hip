// Conformer conv-module middle, streaming form: causal depthwise conv k = 31 -> LayerNorm(1024) -> swish (HF modeling_wav2vec2_bert.py:212-222),
// the arithmetic of dwconv_ln_swish_kernel (w2vbert_kernels.hip) with a different work split.
//
// That kernel gives a thread 4 channels x 16 output rows with all 31 tap weights in registers: 368 registers, ONE wave per SIMD, every input row read
// 2.9 times (46 rows per 16 outputs) — 0.51 ms per launch for 0.79 GB of compulsory traffic (5 x its HBM floor; bound by load latency it cannot hide).
// Here a thread owns ONE channel and walks along time: 31 weights + a 38-row input window + 8 accumulators = ~90 registers, 16 waves per CU, and
// every input row is read once (plus a 30-row warm-up per time segment). A workgroup = all 1024 channels of one (clip, time segment); it produces 8 output
// rows per iteration: 248 dependent-chain FMAs per thread in tap order (the same chain per output element as the register-stationary kernel), then the
// LayerNorm of the 8 rows together.
//
// Bit-identity with dwconv_ln_swish_kernel is kept on purpose (pinned token checksums, tests/test_ops_gpu.py::test_dwconv_stream_is_bit_identical): the
// LayerNorm moments are reduced over the SAME tree — (x + y) + (z + w) inside a channel quad, the xor-butterfly 32, 16, 8, 4, 2, 1 over the 64 quads of a
// 256-channel group, ((g0 + g1) + (g2 + g3)) over the groups — although the quads of a group now live in four waves: the quad sums go through LDS
// and one wave per (row, group) runs the butterfly with lane = quad. The squared deviations use the contraction the compiler chose there:
// fma(dx, dx, dy * dy) + fma(dz, dz, dw * dw).
#include "at_common.h"
#include "w2vbert_kernels.h"
#include "split_scheme.h"

#include <type_traits>

namespace at {

namespace {

constexpr int DS_K = 31;        // taps
constexpr int DS_RB = 8;        // output rows per iteration
constexpr int DS_WIN = DS_RB + DS_K - 1;   // input window: row k <-> time tb - 30 + k
constexpr int DS_C = 1024;

// lane ^ 1 / lane ^ 2 inside a quad on the vector unit (DPP quad_perm), not through the LDS pipe like ds_bpermute
__device__ __forceinline__ float ds_quad_xor1(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true)); }
__device__ __forceinline__ float ds_quad_xor2(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true)); }

// The moment reduction over 1024 channels of DS_RB rows: v[i] = this thread's (channel's) term. sq: LDS scratch [DS_RB][256]. Two workgroup barriers.
// The four (256-channel group) butterflies of a row are done by ONE wave — lane = quad, exactly the register-stationary kernel's wave — which also forms
// the row's statistic once, for the 1024 threads that would otherwise each recompute it (a form in which every thread reduced its own copy made the
// kernel LDS-instruction-bound: 0.63 ms per launch; with per-thread 1 / sqrt it was vector-bound: 0.40 ms):
//   SECOND = false: stat[r][0] = mean = total / 1024
//   SECOND = true:  stat[r][1] = rstd = 1 / sqrt(total / 1024 + 1e-5), stat[r][2] = shift = -rstd * mean
template <bool SECOND>
__device__ __forceinline__ void ds_reduce(float (&v)[DS_RB], float (*sq)[256], float (*stat)[4], int c, int lane) {
#pragma unroll
    for (int i = 0; i < DS_RB; ++i) {
        float p = v[i] + ds_quad_xor1(v[i]);      // (x + y) and (z + w)
        p = p + ds_quad_xor2(p);                  // (x + y) + (z + w), in all four lanes of the quad
        if ((lane & 3) == 0) sq[i][c >> 2] = p;
    }
    __syncthreads();
    const int wave = c >> 6;
    if (wave < DS_RB) {   // wave r reduces row r
        float t[4];
#pragma unroll
        for (int grp = 0; grp < 4; ++grp) t[grp] = sq[wave][grp * 64 + lane];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1)
#pragma unroll
            for (int grp = 0; grp < 4; ++grp) t[grp] += __shfl_xor(t[grp], off);
        if (lane == 0) {
            const float total = (t[0] + t[1]) + (t[2] + t[3]);
            if (!SECOND) {
                stat[wave][0] = total * (1.0f / 1024.0f);
            } else {
                const float vr = total * (1.0f / 1024.0f);
                const float rstd = 1.0f / sqrtf(vr + 1e-5f);
                stat[wave][1] = rstd;
                stat[wave][2] = -rstd * stat[wave][0];
            }
        }
    }
    __syncthreads();
}

}  // namespace

// SC = void: fp32 output [B*T][1024]; SC = an operand scheme: K-blocked pieces [NP][64][rows_pad][16] (as dwconv_ln_swish_kernel)
template <class SC>
__global__ __launch_bounds__(1024, 1) void dwconv_stream_kernel(const float* __restrict__ g, const float* __restrict__ w /*[31][1024]*/,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                float* __restrict__ out, int T, int seg_len, void* __restrict__ pieces, long long rows_pad,
                                                                float scale, int* __restrict__ status) {
    __shared__ float sq[DS_RB][256];
    __shared__ __attribute__((aligned(16))) float stat[DS_RB][4];   // per row: mean, rstd, shift
    __shared__ __attribute__((aligned(16))) float otile[DS_RB][DS_C + 4];   // pieces only: the 8 output rows, re-read as float4 per channel quad (+ 16 B per row: the re-read walks ROWS fastest)
    const int c = threadIdx.x, lane = c & 63;
    const int b = blockIdx.y;
    const int t_begin = blockIdx.x * seg_len;
    if (t_begin >= T) return;
    const int t_end = t_begin + seg_len < T ? t_begin + seg_len : T;
    const long long base = (long long)b * T;
    const float* gc = g + base * DS_C + c;

    float wt[DS_K];
#pragma unroll
    for (int j = 0; j < DS_K; ++j) wt[j] = w[j * DS_C + c];
    float xw[DS_WIN];
#pragma unroll
    for (int k = 0; k < DS_K - 1; ++k) {   // warm-up: the 30 rows before the segment (zeros before the clip)
        const int t = t_begin - (DS_K - 1) + k;
        xw[k] = t >= 0 ? gc[(long long)t * DS_C] : 0.f;
    }
    float pre[DS_RB];
    auto prefetch = [&](int tb) {
#pragma unroll
        for (int i = 0; i < DS_RB; ++i) {
            int t = tb + i;
            t = t < T ? t : T - 1;   // rows past the end only feed outputs that are never stored
            pre[i] = gc[(long long)t * DS_C];
        }
    };
    prefetch(t_begin);
    RangeMax over;
    (void)over;
    for (int tb = t_begin; tb < t_end; tb += DS_RB) {
#pragma unroll
        for (int i = 0; i < DS_RB; ++i) xw[DS_K - 1 + i] = pre[i];
        if (tb + DS_RB < t_end) prefetch(tb + DS_RB);
        float acc[DS_RB];
#pragma unroll
        for (int i = 0; i < DS_RB; ++i) {
            float a = 0.f;
#pragma unroll
            for (int tap = 0; tap < DS_K; ++tap) a = __builtin_fmaf(wt[tap], xw[i + tap], a);   // out[tb + i] uses in[tb + i - 30 + tap], tap ascending
            acc[i] = a;
        }
        // ---- LayerNorm over the 1024 channels of each row ---------------------------------------------------------------------------
        float term[DS_RB];
#pragma unroll
        for (int i = 0; i < DS_RB; ++i) term[i] = acc[i];
        ds_reduce<false>(term, sq, stat, c, lane);
#pragma unroll
        for (int i = 0; i < DS_RB; ++i) {
            const float d = acc[i] - stat[i][0];
            // quad term (dx*dx + dy*dy) + (dz*dz + dw*dw) as fma(dx, dx, dy * dy) + fma(dz, dz, dw * dw): the odd channel supplies the rounded
            // product, the even channel forms the fma. ds_reduce's first step adds lane ^ 1, so the odd lane hands it 0 (f + 0 is exact): the pair sum
            // is f, the quad sum f_xy + f_zw
            const float m = __fmul_rn(d, d);
            const float f = __builtin_fmaf(d, d, ds_quad_xor1(m));   // even lanes: fma(d_even, d_even, d_odd * d_odd)
            term[i] = (lane & 1) ? 0.f : f;
        }
        ds_reduce<true>(term, sq, stat, c, lane);
        // ---- normalise, swish, store -------------------------------------------------------------------------------------------------
        // gamma / beta of this channel are re-read per iteration (opaque pointers: not hoisted) — two registers this 128-register kernel needs elsewhere
        const float* gp = gamma;
        const float* bp = beta;
        asm volatile("" : "+s"(gp), "+s"(bp));
        const float gm = gp[c], bt = bp[c];
#pragma unroll
        for (int i = 0; i < DS_RB; ++i) {
            const int t = tb + i;
            const float o = swishf_(fmaf(fmaf(acc[i], stat[i][1], stat[i][2]), gm, bt));
            if constexpr (std::is_void<SC>::value) {
                if (t < t_end) out[(base + t) * DS_C + c] = o;
            } else {
                otile[i][c] = o;
            }
        }
        if constexpr (!std::is_void<SC>::value) {
            __syncthreads();
            // 8 rows x 256 channel quads = 2048 float4: two per thread -> pieces (8-byte store per piece). Lane order: quad of a k-block fastest, then the
            // ROW, then the k-block — in the K-blocked layout [k-block][row][16] the 8 rows of a k-block are one 256-byte run, so a wave store is two such
            // runs (round 5; channel-quad-fastest order made it sixteen separate 32-byte segments: the pattern that held conv0 of semantic_s at 3 TB/s)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int e = c + 1024 * j, i = (e >> 2) & (DS_RB - 1), qd = ((e >> 5) << 2) | (e & 3);
                const int t = tb + i;
                if (t < t_end) {
                    const f4 o = *reinterpret_cast<const f4*>(&otile[i][qd * 4]);
                    over |= store_pieces4<SC>(reinterpret_cast<typename SC::T*>(pieces), rows_pad * DS_C, rows_pad, base + t, qd * 4, o, scale);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < DS_K - 1; ++k) xw[k] = xw[k + DS_RB];
    }
    if constexpr (!std::is_void<SC>::value)
        if constexpr (SC::RANGE_CHECK)
            range_publish(status, status ? status + 1 : nullptr, over);
}

int launch_dwconv_stream(const float* g, const float* w, const float* gamma, const float* beta, float* out, int B, int T, hipStream_t stream,
                         __bf16* pieces, long long rows_pad, int scheme, float scale, int* status) {
    // time segments: enough workgroups for every CU (one resident workgroup of 16 waves each), segment length a multiple of the 8-row iteration
    const int cus = device_cus();
    int nseg = (cus + B - 1) / B;
    const int max_seg = (T + DS_RB - 1) / DS_RB;
    nseg = nseg < 1 ? 1 : (nseg > max_seg ? max_seg : nseg);
    const int seg_len = ((T + nseg - 1) / nseg + DS_RB - 1) / DS_RB * DS_RB;
    dim3 grid((T + seg_len - 1) / seg_len, B);
    if (pieces && scheme == XB_SCHEME_F16X2)
        hipLaunchKernelGGL((dwconv_stream_kernel<SchemeF16x2>), grid, dim3(1024), 0, stream, g, w, gamma, beta, out, T, seg_len, (void*)pieces, rows_pad, scale, status);
    else if (pieces)
        hipLaunchKernelGGL((dwconv_stream_kernel<SchemeBf16x3>), grid, dim3(1024), 0, stream, g, w, gamma, beta, out, T, seg_len, (void*)pieces, rows_pad, scale, status);
    else
        hipLaunchKernelGGL((dwconv_stream_kernel<void>), grid, dim3(1024), 0, stream, g, w, gamma, beta, out, T, seg_len, nullptr, 0LL, 1.0f, nullptr);
    AT_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace at
