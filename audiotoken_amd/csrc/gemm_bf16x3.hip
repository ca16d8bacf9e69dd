// Linear layers on the bf16 matrix cores with fp32-class accuracy: C = A . W^T where both fp32 operands are split EXACTLY into
// three bf16 pieces (a = a1 + a2 + a3, 8 + 8 + 8 mantissa bits) and the six leading cross products
//     a1 w1,  a1 w2,  a2 w1,  a2 w2,  a1 w3,  a3 w1
// are accumulated by v_mfma_f32_32x32x16_bf16 into fp32. Every product of two bf16 numbers is exact in fp32; what is dropped
// (a2 w3, a3 w2, a3 w3) is below 2^-24 of the leading term. Measured against a float64 reference (tools/bf16x3_gemm.hip, K = 1024):
// max error 1.8e-6 for the split scheme vs 2.4e-6 for the k-ordered fp32 FMA chain of the fp32 MFMA path — it is not the less
// accurate of the two. Six bf16 MFMAs cost 6/16 of the fp32 MFMAs they replace (2.5 PFLOP/s vs 157 TFLOP/s peak).
//
// Data layout: every piece is K-BLOCKED, [K/16][rows_pad][16] bf16, so the 256 rows x 16 k of a workgroup's K tile are one
// contiguous 8 KB run (row-major pieces measured 157-179 instead of 194-196 fp32-equivalent TFLOP/s). Weights are split once
// at finalize(); activations are split by their producer: a standalone pass for LayerNorm outputs (launch_split_blocked) and
// the GEMM epilogue itself for the FFN's hidden activation.
// Tiles: 256 x 128 per workgroup (4 waves, each 64 x 128 = 2 x 4 MFMA tiles of 32 x 32, two workgroups per CU), 256 x 256
// (8 waves) or 128 x 128 for small launches; K tile 16; register-staged double-buffered LDS; operands swapped (weights as MFMA A) so a lane owns 4 consecutive output channels.
// Used for the conformer feed-forward layers (w2vbert.hip); everything that has a bit-identical fused twin stays on the fp32 MFMA.
//
// Round 2: the same kernel also runs the TWO-piece fp16 scheme (XB_SCHEME_F16X2, gemm_bf16x3.h): a s = hi + lo in fp16, three
// products (hi.lo, lo.hi, hi.hi) on v_mfma_f32_32x32x16_f16, operands pre-scaled by powers of two, accumulator rescaled in the
// epilogue. Half the MFMAs and 4 instead of 6 operand bytes per element at a float64 error below the bf16x3 scheme's
// (tools/f16x2_gemm.hip); used where activations are normalised (the two semantic tokenizers); the scheme is a template
// parameter, the tile shapes, staging and epilogues are shared.
#include "at_common.h"
#include "gemm_bf16x3.h"
#include "split_scheme.h"
#include "split_epilogue.h"
#include "../../include/audiotoken_hip.h"
#include <type_traits>
#include <cstdlib>
#include <cmath>

namespace at {

typedef unsigned int u4 __attribute__((ext_vector_type(4)));
// Clamped to [2^-60, 2^60]: a tensor whose max |w| is tiny or denormal would otherwise get a scale near (or at) infinity — w s = inf / NaN and
// acc_scale = 1 / (16 s) = 0, i.e. silent NaN or zero outputs (the weight splits run without a status word). Below ~1e-30 the tensor is
// treated as all-zero (scale 1): its pieces underflow to zero exactly as its fp32 products would against normalised activations.
float xb_weight_scale(float max_abs) {
    if (!(max_abs > 1e-30f) || !std::isfinite(max_abs)) return 1.0f;
    const float e = std::floor(std::log2(32767.0f / max_abs));
    return std::exp2(std::fmin(std::fmax(e, -60.0f), 60.0f));
}
float xb_ln_site_scale(float gmax, float bmax, int D) {
    const float bound = std::sqrt((float)D) * gmax + bmax;
    float s = XB_F16_ACT_SCALE;
    while (s > 1.0f / 1048576.0f && !(s * bound <= 65000.0f)) s *= 0.5f;
    return s;
}

constexpr int XB_K = 16;
// Two tile shapes with IDENTICAL per-element arithmetic (the k order and the order of the six products do not depend on the
// tile), so results do not depend on which one a launch uses: 256 x 256 (8 waves, 4 x 2, each 2 x 4 MFMA tiles) for large
// launches, 128 x 128 (4 waves, 2 x 2, each 2 x 2) when 256 x 256 tiles would leave most of the 256 CUs idle (single clips).
template <int WM, int WN, int TI, int TJ>
struct XbCfg {
    static constexpr int BM = WM * TI * 32, BN = WN * TJ * 32, NT = WM * WN * 64;
    static constexpr int PA = BM * XB_K, PW = BN * XB_K;            // bf16 elements of one piece of the A / W tile
    template <int NP> static constexpr int stage() { return NP * (PA + PW); }   // 16-bit elements per LDS stage
};

__device__ __forceinline__ void split3(float a, __bf16& p1, __bf16& p2, __bf16& p3) {
    p1 = (__bf16)a;
    const float r1 = a - (float)p1;
    p2 = (__bf16)r1;
    const float r2 = r1 - (float)p2;
    p3 = (__bf16)r2;
}

// fp32 row-major [rows][ld] (first K columns) * scale -> NP K-blocked pieces. Workgroup = 64 rows x 64 k through LDS so that both
// the reads (256 B per row) and the writes (64 rows x 32 B = 2 KB per k-block) are contiguous.
template <class SC>
__global__ __launch_bounds__(256) void split_blocked_kernel(const float* __restrict__ x, int ld, long long rows, long long rows_pad, int K,
                                                            typename SC::T* __restrict__ out, float scale, int* __restrict__ status,
                                                            int win_cblocks, int win_stride) {
    __shared__ float tile[64][65];
    const long long r0 = (long long)blockIdx.x * 64;
    const int k0 = blockIdx.y * 64;
    for (int e = threadIdx.x; e < 64 * 16; e += 256) {        // 64 rows x 16 float4
        const int r = e >> 4, c = (e & 15) * 4;
        f4 v = {0.f, 0.f, 0.f, 0.f};
        if (r0 + r < rows) v = *reinterpret_cast<const f4*>(x + (r0 + r) * ld + k0 + c);
        tile[r][c] = v.x; tile[r][c + 1] = v.y; tile[r][c + 2] = v.z; tile[r][c + 3] = v.w;
    }
    __syncthreads();
    const long long ps = rows_pad * (long long)K;
    RangeMax over;
    // thread -> (k-block kb of 4, row r, quarter qd of the 16 k): 4 values = 8 bytes per piece
    for (int e = threadIdx.x; e < 4 * 64 * 4; e += 256) {
        const int qd = e & 3, r = (e >> 2) & 63, kb = e >> 8;
        typename SC::V4 p[SC::NP];
        const f4 v = {tile[r][kb * 16 + qd * 4], tile[r][kb * 16 + qd * 4 + 1], tile[r][kb * 16 + qd * 4 + 2], tile[r][kb * 16 + qd * 4 + 3]};
        over |= split4<SC>(v, scale, p);
        int kdst = k0 / 16 + kb;
        if (win_cblocks > 0) kdst = xb_window_dst(kdst, win_cblocks, win_stride, (K / 16) / win_cblocks);
        const long long o = ((long long)kdst * rows_pad + r0 + r) * 16 + qd * 4;
        if (r0 + r < rows_pad) {
#pragma unroll
            for (int i = 0; i < SC::NP; ++i) *reinterpret_cast<typename SC::V4*>(out + i * ps + o) = p[i];
        }
    }
    if constexpr (SC::RANGE_CHECK)
        range_publish(status, status ? status + 1 : nullptr, over);
}

// fp32 [B][L][C] * scale -> the pieces a causal conv with `pad` reflected front rows reads as a windowed GEMM:
// [NP][B][C/16][stride planes][Lp][16], padded row i = t + pad in plane i % stride at index i / stride; rows i < pad are the
// reflection x[pad - i]. A wave owns one 16-channel block of one (clip, plane): lanes = (4-channel group, 16 consecutive
// indices), so a wave-store covers 512 contiguous bytes per piece and the reads are 64-byte runs.
template <class SC>
__global__ __launch_bounds__(256) void split_phase_major_kernel(const float* __restrict__ x, int L, int C, int stride, int pad, int Lp, int nidx,
                                                                long long piece_stride, typename SC::T* __restrict__ out, float scale, int* __restrict__ status,
                                                                int reflect) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k = lane & 3, ii = lane >> 2;
    const int cb = blockIdx.y * 4 + wave, cblocks = C / 16;
    const int b = blockIdx.z / stride, plane = blockIdx.z - b * stride;
    if (cb >= cblocks) return;
    const float* xb = x + (long long)b * L * C + cb * 16 + k * 4;
    typename SC::T* ob = out + (((long long)b * cblocks + cb) * stride + plane) * Lp * 16 + k * 4;
    RangeMax over;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int idx = blockIdx.x * 64 + j * 16 + ii;
        if (idx >= nidx) continue;
        int t = idx * stride + plane - pad;
        const bool front_zero = t < 0 && !reflect;   // zero front padding (the transposed convs' x[-1]) instead of the reflection
        t = t < 0 ? -t : t;
        f4 v = {0.f, 0.f, 0.f, 0.f};
        if (t < L && !front_zero) v = *reinterpret_cast<const f4*>(xb + (long long)t * C);
        typename SC::V4 pc[SC::NP];
        over |= split4<SC>(v, scale, pc);
        typename SC::T* d = ob + (long long)idx * 16;
#pragma unroll
        for (int i = 0; i < SC::NP; ++i) *reinterpret_cast<typename SC::V4*>(d + i * piece_stride) = pc[i];
    }
    if constexpr (SC::RANGE_CHECK)
        range_publish(status, status ? status + 1 : nullptr, over);
}

__global__ void reflect_front_kernel(__bf16* S, int B, int blocks, int phases, int Lp, int pad, int npieces) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;          // (piece, clip, block) x pad row x 4-channel group
    const long long total = (long long)npieces * B * blocks * pad * 4;
    if (gid >= total) return;
    const int c4 = (int)(gid & 3), i = (int)((gid >> 2) % pad);
    const long long rest = (gid >> 2) / pad;   // (piece * B + clip) * blocks + block
    const int j = 2 * pad - i;
    __bf16* base = S + rest * phases * Lp * 16 + c4 * 4;
    *reinterpret_cast<bf16x4*>(base + ((long long)(i % phases) * Lp + i / phases) * 16) =
        *reinterpret_cast<const bf16x4*>(base + ((long long)(j % phases) * Lp + j / phases) * 16);
}

int launch_reflect_front(__bf16* S, int B, int blocks, int phases, int Lp, int pad, hipStream_t stream, int npieces) {
    const long long total = (long long)npieces * B * blocks * pad * 4;
    hipLaunchKernelGGL(reflect_front_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, S, B, blocks, phases, Lp, pad, npieces);
    AT_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_split_windowed(const float* x, int B, int L, int C, int stride, int pad, int Lp, __bf16* out, hipStream_t stream, int scheme, float scale,
                          int* status, int reflect) {
    AT_REQUIRE(C % 16 == 0 && L > pad && stride >= 1 && pad >= 0, "split_windowed: C % 16, L > pad");
    const int nidx = (L + pad + stride - 1) / stride;
    AT_REQUIRE(nidx <= Lp, "split_windowed: Lp too small");
    dim3 grid((nidx + 63) / 64, (C / 16 + 3) / 4, B * stride);
    const long long ps = (long long)B * (C / 16) * stride * Lp * 16;
    if (scheme == XB_SCHEME_F16X2)
        hipLaunchKernelGGL(split_phase_major_kernel<SchemeF16x2>, grid, dim3(256), 0, stream, x, L, C, stride, pad, Lp, nidx, ps, reinterpret_cast<_Float16*>(out), scale, status, reflect);
    else
        hipLaunchKernelGGL(split_phase_major_kernel<SchemeBf16x3>, grid, dim3(256), 0, stream, x, L, C, stride, pad, Lp, nidx, ps, out, 1.0f, nullptr, reflect);
    AT_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_split_phase_major(const float* x, int B, int L, int C, int stride, int Lp, __bf16* out, hipStream_t stream) {
    return launch_split_windowed(x, B, L, C, stride, stride, Lp, out, stream, XB_SCHEME_BF16X3, 1.0f, nullptr, 1);
}

int launch_split_blocked(const float* x, int ld, long long rows, long long rows_pad, int K, __bf16* out, hipStream_t stream, int scheme, float scale,
                         int* status, int win_cblocks, int win_stride) {
    AT_REQUIRE(K % 64 == 0 && ld % 4 == 0 && rows_pad >= rows && rows_pad % 64 == 0, "split_blocked: K % 64, ld % 4, rows_pad % 64");
    AT_REQUIRE(win_cblocks == 0 || (win_stride >= 1 && (K / 16) % win_cblocks == 0), "split_blocked: bad window description");
    dim3 grid((unsigned)(rows_pad / 64), K / 64);
    if (scheme == XB_SCHEME_F16X2)
        hipLaunchKernelGGL(split_blocked_kernel<SchemeF16x2>, grid, dim3(256), 0, stream, x, ld, rows, rows_pad, K, reinterpret_cast<_Float16*>(out), scale, status, win_cblocks, win_stride);
    else
        hipLaunchKernelGGL(split_blocked_kernel<SchemeBf16x3>, grid, dim3(256), 0, stream, x, ld, rows, rows_pad, K, out, scale, status, win_cblocks, win_stride);
    AT_CHECK_HIP(hipGetLastError());
    return 0;
}

// DUAL: the XB_EPI_RAW_ELU_SPLIT2 epilogue as its own instantiation (in the general kernel its two split writers spilled)
template <class SC, int WM, int WN, int TI, int TJ, bool DUAL = false>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN <= 4) ? 2 : 1) void gemm_bf16x3_kernel(Bf16x3Args a) {
    using Cfg = XbCfg<WM, WN, TI, TJ>;
    typedef typename SC::T PT;
    typedef typename SC::V8 V8;
    constexpr int NP = SC::NP, STAGE = Cfg::template stage<NP>();
    constexpr int XB_M = Cfg::BM, XB_N = Cfg::BN, NT = Cfg::NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];   // [2 stages][A: NP x rows x 16 | W: NP x rows x 16]
    PT* lds = reinterpret_cast<PT*>(lds_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int ntn = a.N / XB_N, ntm = a.Mpad / XB_M;
    const int n0 = (blockIdx.x % ntn) * XB_N;   // n fastest: the activation tile is fetched once per row of blocks
    const int mt = blockIdx.x / ntn;
    const int clip = mt / ntm, m0 = (mt - clip * ntm) * XB_M;
    const int cblocks = a.cblocks > 0 ? a.cblocks : a.K / XB_K;
    const int Lp = a.Lp > 0 ? a.Lp : a.Mpad;
    const long long a_clip = (long long)cblocks * a.stride * Lp * 16;   // elements of one clip of one piece
    const long long psA = a_clip * a.batch, psW = (long long)a.N * a.K;
    const int nk = a.K / XB_K;
    const PT* Ab = reinterpret_cast<const PT*>(a.A) + clip * a_clip + (long long)m0 * 16;
    const PT* Wb = reinterpret_cast<const PT*>(a.W);
    // staging: one piece of an operand tile is rows x 32 B = 2 * rows chunks of 16 B, contiguous in the K-blocked layout
    constexpr int CA = (2 * XB_M) / NT, CW = (2 * XB_N) / NT;   // chunks per thread per piece
    static_assert((2 * XB_M) % NT == 0 && (2 * XB_N) % NT == 0, "tile rows must be a multiple of half the thread count");
    u4 sa[NP][CA], sw[NP][CW];
    auto load = [&](int kt) {
        int tk, offk;   // K tile -> (plane image of a channel block, row offset), window order
        xb_window_block(kt, a.stride, nk / cblocks, tk, offk);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
#pragma unroll
            for (int c = 0; c < CA; ++c)
                sa[p][c] = *reinterpret_cast<const u4*>(Ab + p * psA + ((long long)tk * Lp + offk) * 16 + (tid + c * NT) * 8);
#pragma unroll
            for (int c = 0; c < CW; ++c) sw[p][c] = *reinterpret_cast<const u4*>(Wb + p * psW + ((long long)kt * a.N + n0) * 16 + (tid + c * NT) * 8);
        }
    };
    auto store = [&](int buf) {
        PT* s = lds + buf * STAGE;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
#pragma unroll
            for (int c = 0; c < CA; ++c) *reinterpret_cast<u4*>(s + p * Cfg::PA + (tid + c * NT) * 8) = sa[p][c];
#pragma unroll
            for (int c = 0; c < CW; ++c) *reinterpret_cast<u4*>(s + NP * Cfg::PA + p * Cfg::PW + (tid + c * NT) * 8) = sw[p][c];
        }
    };
    f16v acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    load(0);
    store(0);
    __syncthreads();
    const int frow = lane & 31, fhalf = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        const PT* s = lds + (kt & 1) * STAGE;
        V8 xa[NP][TI], wb[NP][TJ];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
#pragma unroll
            for (int i = 0; i < TI; ++i) xa[p][i] = *reinterpret_cast<const V8*>(s + p * Cfg::PA + (wm * TI * 32 + i * 32 + frow) * 16 + fhalf * 8);
#pragma unroll
            for (int j = 0; j < TJ; ++j) wb[p][j] = *reinterpret_cast<const V8*>(s + NP * Cfg::PA + p * Cfg::PW + (wn * TJ * 32 + j * 32 + frow) * 16 + fhalf * 8);
        }
        if (kt + 1 < nk) load(kt + 1);
        // the leading cross products, smallest first
#pragma unroll
        for (int t = 0; t < SC::NPROD; ++t)
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[i][j] = SC::mfma(wb[SC::prod_w(t)][j], xa[SC::prod_a(t)][i], acc[i][j]);
        if (kt + 1 < nk) store((kt + 1) & 1);
        __syncthreads();
    }
    // lane holds output row m = frow of its 32-row tile and, per register group g, 4 consecutive columns n = 8g + 4 fhalf ..
    // one specialised copy of the epilogue per mode (a uniform switch): with the mode tested per element the compiler evaluated
    // every activation and selected (+5 % on the FFN GEMMs)
    XbEpilogue<SC> ep(a, clip);
    auto epilogue = [&](auto mode) {
        constexpr int E = decltype(mode)::value;
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            const int m = m0 + wm * TI * 32 + i * 32 + frow;
            if (m >= a.M) continue;
#pragma unroll
            for (int j = 0; j < TJ; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = n0 + wn * TJ * 32 + j * 32 + 8 * g + 4 * fhalf;
                    const f4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                    ep.template apply<E>(m, n, v);
                }
        }
    };
    if constexpr (DUAL) {
        epilogue(std::integral_constant<int, XB_EPI_RAW_ELU_SPLIT2>{});
    } else {
        switch (a.epi) {
            case XB_EPI_SWISH_SPLIT: epilogue(std::integral_constant<int, XB_EPI_SWISH_SPLIT>{}); break;
            case XB_EPI_GELU_SPLIT: epilogue(std::integral_constant<int, XB_EPI_GELU_SPLIT>{}); break;
            case XB_EPI_ELU_SPLIT: epilogue(std::integral_constant<int, XB_EPI_ELU_SPLIT>{}); break;
            case XB_EPI_GLU: epilogue(std::integral_constant<int, XB_EPI_GLU>{}); break;
            case XB_EPI_GELU: epilogue(std::integral_constant<int, XB_EPI_GELU>{}); break;
            case XB_EPI_QKV: epilogue(std::integral_constant<int, XB_EPI_QKV>{}); break;
            default: epilogue(std::integral_constant<int, XB_EPI_LINEAR>{}); break;
        }
    }
    ep.finish();
}

template <class SC, int WM, int WN, int TI, int TJ, bool DUAL = false>
static int launch_xb(const Bf16x3Args& a, hipStream_t stream) {
    using Cfg = XbCfg<WM, WN, TI, TJ>;
    const size_t ldsb = 2 * (size_t)Cfg::template stage<SC::NP>() * 2;
    { static LdsAttrFlags lds_attr; if (int rc = set_max_dynamic_lds(lds_attr, gemm_bf16x3_kernel<SC, WM, WN, TI, TJ, DUAL>, ldsb)) return rc; }
    const dim3 grid((unsigned)((long long)a.batch * (a.Mpad / Cfg::BM) * (a.N / Cfg::BN)));
    hipLaunchKernelGGL((gemm_bf16x3_kernel<SC, WM, WN, TI, TJ, DUAL>), grid, dim3(Cfg::NT), ldsb, stream, a);
    AT_CHECK_HIP(hipGetLastError());
    return 0;
}

template <class SC>
static int launch_scheme(const Bf16x3Args& a, hipStream_t stream) {
    const long long tiles256 = (long long)a.batch * (a.Mpad / 256) * ((a.N + 255) / 256);
    if (a.epi == XB_EPI_RAW_ELU_SPLIT2) return launch_xb<SC, 4, 1, 2, 4, true>(a, stream);
    if (tiles256 < 256 || a.N % 256 != 0) return launch_xb<SC, 2, 2, 2, 2>(a, stream);   // 4x as many 128 x 128 tiles: same arithmetic, fills the chip
    // large launches: 256 x 128 tiles, 4 waves, TWO workgroups per CU (one loads while the other multiplies) measured 1-2 % ahead of
    // 256 x 256 with 8 waves and one workgroup per CU (less operand traffic, no overlap).
    return launch_xb<SC, 4, 1, 2, 4>(a, stream);
}

__global__ void range_combine_kernel(const int* tab, int nsites, int* out) {
    int v = 0;
    for (int k = 0; k < nsites; ++k) v |= tab[2 * k] & (XB_STATUS_F16_OVERFLOW | XB_STATUS_NONFINITE);
    if (v) atomicOr(out, v);
}
int launch_range_combine(const int* range_tab, int nsites, int* status_out, hipStream_t stream) {
    if (!range_tab || !status_out) return 0;
    hipLaunchKernelGGL(range_combine_kernel, dim3(1), dim3(1), 0, stream, range_tab, nsites, status_out);
    AT_CHECK_HIP(hipGetLastError());
    return 0;
}
// the operator-level test entry points have no handle: one {flag, census} pair per device, allocated on first use and kept
static int* op_range_pair() {
    static int* pairs[kMaxDevices] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
    if (!pairs[dev] && hipMalloc((void**)&pairs[dev], 2 * sizeof(int)) != hipSuccess) pairs[dev] = nullptr;
    return pairs[dev];
}

int launch_gemm_bf16x3(const Bf16x3Args& a, hipStream_t stream) {
    AT_REQUIRE(a.A && a.W && a.M >= 1 && a.N % 128 == 0 && a.K % XB_K == 0 && a.Mpad % 256 == 0 && a.Mpad >= a.M,
               "gemm_bf16x3: N % 128, K % 16, Mpad % 256");
    const bool split_out = a.epi == XB_EPI_SWISH_SPLIT || a.epi == XB_EPI_GELU_SPLIT || a.epi == XB_EPI_ELU_SPLIT || a.epi == XB_EPI_RAW_ELU_SPLIT2;
    AT_REQUIRE(a.epi != XB_EPI_QKV || (a.S != nullptr && a.C != nullptr && a.qkv_hid > 0 && a.N == 3 * a.qkv_hid && a.qkv_hid % 256 == 0 && a.Spad >= a.M && a.batch == 1),
               "gemm_bf16x3: XB_EPI_QKV needs C, S, N = 3 * qkv_hid, qkv_hid % 256");
    AT_REQUIRE(split_out ? (a.S != nullptr && a.Sphases >= 1 && (long long)a.Spad * a.Sphases >= a.M + (long long)a.Sfront * a.Sphases) : (a.C != nullptr && a.ldc % 2 == 0), "gemm_bf16x3: bad output");
    AT_REQUIRE(a.epi != XB_EPI_RAW_ELU_SPLIT2 || (a.S2 != nullptr && a.S2phases >= 1 && (long long)a.S2pad * a.S2phases >= a.M + (long long)a.S2front * a.S2phases), "gemm_bf16x3: bad second output");
    AT_REQUIRE(a.batch >= 1 && a.stride >= 1 && (a.cblocks == 0 || (a.K / XB_K) % a.cblocks == 0), "gemm_bf16x3: bad window description");
    AT_REQUIRE((a.Lp > 0 ? a.Lp : a.Mpad) >= a.Mpad + ((a.cblocks > 0 ? (a.K / XB_K) / a.cblocks : 1) - 1) / a.stride, "gemm_bf16x3: Lp too small for the last tile");
    AT_REQUIRE(a.scheme == XB_SCHEME_BF16X3 || a.scheme == XB_SCHEME_F16X2, "gemm_bf16x3: unknown scheme");
    if (gemm_f16x2_tg_eligible(a)) return launch_gemm_f16x2_tg(a, stream);   // the register-staged kernel below serves the other shapes (and at_op_gemm_split kernel = 2)
    if (a.scheme == XB_SCHEME_F16X2) return launch_scheme<SchemeF16x2>(a, stream);
    return launch_scheme<SchemeBf16x3>(a, stream);
}

}  // namespace at

// ---- operator-level entry point for the parity tests ------------------------------------------------------------------------------
extern "C" int at_op_gemm_split(const float* X, const float* W, const float* bias, float* C, int M, int N, int K, int scheme, float w_max_abs,
                                int kernel, void* workspace, size_t workspace_bytes, int32_t* status_dev, at_stream_t stream_) {
    using namespace at;
    AT_REQUIRE(X && W && C && workspace && M >= 1 && N % 128 == 0 && K % 64 == 0, "at_op_gemm_split: bad arguments (N % 128, K % 64)");
    AT_REQUIRE(scheme == XB_SCHEME_BF16X3 || scheme == XB_SCHEME_F16X2, "at_op_gemm_split: scheme 0 (bf16x3) or 1 (f16x2)");
    hipStream_t stream = (hipStream_t)stream_;
    const long long Mpad = ((long long)M + 255) / 256 * 256;
    const int np = xb_pieces(scheme);
    const size_t need = ((size_t)Mpad * K + (size_t)N * K) * np * sizeof(piece_t);
    AT_REQUIRE(workspace_bytes >= need, "at_op_gemm_split: workspace too small");
    piece_t* xs = reinterpret_cast<piece_t*>(workspace);
    piece_t* wsp = xs + (size_t)Mpad * K * np;
    const float sa = scheme == XB_SCHEME_F16X2 ? XB_F16_ACT_SCALE : 1.0f, sw = scheme == XB_SCHEME_F16X2 ? xb_weight_scale(w_max_abs) : 1.0f;
    int* pair = status_dev ? op_range_pair() : nullptr;
    if (status_dev) {
        AT_REQUIRE(pair != nullptr, "at_op_gemm_split: cannot allocate the range pair");
        AT_CHECK_HIP(hipMemsetAsync(status_dev, 0, sizeof(int32_t), stream));
        AT_CHECK_HIP(hipMemsetAsync(pair, 0, 2 * sizeof(int), stream));
    }
    if (int rc = launch_split_blocked(X, K, M, Mpad, K, xs, stream, scheme, sa, pair)) return rc;
    if (int rc = launch_split_blocked(W, K, N, N, K, wsp, stream, scheme, sw, pair)) return rc;
    Bf16x3Args a;
    a.A = xs; a.W = wsp; a.bias = bias; a.M = M; a.N = N; a.K = K; a.Mpad = (int)Mpad; a.epi = XB_EPI_LINEAR; a.C = C; a.ldc = N;
    a.scheme = scheme; a.acc_scale = 1.0f / (sa * sw); a.split_scale = sa; a.status = pair;
    int rc;
    if (kernel == 1) {   // force the two-group kernel (gemm_f16x2_tg.hip) whatever the launch size
        AT_REQUIRE(gemm_f16x2_tg_eligible(a), "at_op_gemm_split: the two-group kernel needs f16x2, N % 128, K % 32");
        rc = launch_gemm_f16x2_tg(a, stream);
    } else if (kernel == 2) {   // force the register-staged kernel
        rc = a.scheme == XB_SCHEME_F16X2 ? launch_scheme<SchemeF16x2>(a, stream) : launch_scheme<SchemeBf16x3>(a, stream);
    } else {
        rc = launch_gemm_bf16x3(a, stream);
    }
    if (rc) return rc;
    return launch_range_combine(pair, 1, reinterpret_cast<int*>(status_dev), stream);
}

// A causal conv1d (reflect front padding k - stride, as the SEANet convs) on the WINDOWED two-piece fp16 split GEMM, for the parity / soak tests
// of the two-group kernel's windowed instantiations: X [B][L][Cin] fp32 channels-last, W [Cout][k * Cin] (tap-major), C [B][L / stride][Cout].
extern "C" int at_op_conv_split(const float* X, const float* W, const float* bias, float* C, int B, int L, int Cin, int Cout, int ktaps, int stride,
                                float w_max_abs, void* workspace, size_t workspace_bytes, int32_t* status_dev, at_stream_t stream_) {
    using namespace at;
    AT_REQUIRE(X && W && C && workspace && B >= 1 && Cin % 16 == 0 && Cout % 128 == 0 && ktaps >= stride && stride >= 1 && L % stride == 0 && (ktaps * Cin) % 64 == 0,
               "at_op_conv_split: Cin % 16, Cout % 128, k >= stride, L % stride, (k Cin) % 64");
    hipStream_t stream = (hipStream_t)stream_;
    const int M = L / stride, Mpad = (M + 255) / 256 * 256, pad = ktaps - stride, K = ktaps * Cin;
    const int Lp = Mpad + (ktaps - 1) / stride + 1;
    AT_REQUIRE(L > pad, "at_op_conv_split: L > k - stride");
    const size_t a_el = (size_t)2 * B * (Cin / 16) * stride * Lp * 16, w_el = (size_t)2 * Cout * K;
    AT_REQUIRE(workspace_bytes >= (a_el + w_el) * sizeof(piece_t), "at_op_conv_split: workspace too small");
    piece_t* xs = reinterpret_cast<piece_t*>(workspace);
    piece_t* wsp = xs + a_el;
    const float sw = xb_weight_scale(w_max_abs);
    int* pair = status_dev ? op_range_pair() : nullptr;
    if (status_dev) {
        AT_REQUIRE(pair != nullptr, "at_op_conv_split: cannot allocate the range pair");
        AT_CHECK_HIP(hipMemsetAsync(status_dev, 0, sizeof(int32_t), stream));
        AT_CHECK_HIP(hipMemsetAsync(pair, 0, 2 * sizeof(int), stream));
    }
    AT_CHECK_HIP(hipMemsetAsync(xs, 0, a_el * sizeof(piece_t), stream));   // rows past the data (read by the padded output rows only) stay finite
    if (int rc = launch_split_windowed(X, B, L, Cin, stride, pad, Lp, xs, stream, XB_SCHEME_F16X2, XB_F16_ACT_SCALE, pair, 1)) return rc;
    if (int rc = launch_split_blocked(W, K, Cout, Cout, K, wsp, stream, XB_SCHEME_F16X2, sw, nullptr, Cin / 16, stride)) return rc;
    Bf16x3Args a;
    a.A = xs; a.W = wsp; a.bias = bias; a.M = M; a.Mpad = Mpad; a.N = Cout; a.K = K;
    a.batch = B; a.stride = stride; a.cblocks = Cin / 16; a.Lp = Lp;
    a.scheme = XB_SCHEME_F16X2; a.acc_scale = 1.0f / (XB_F16_ACT_SCALE * sw); a.split_scale = XB_F16_ACT_SCALE; a.status = pair;
    a.epi = XB_EPI_LINEAR; a.C = C; a.ldc = Cout;
    AT_REQUIRE(gemm_f16x2_tg_eligible(a), "at_op_conv_split: shape not eligible for the two-group kernel");
    if (int rc = launch_gemm_f16x2_tg(a, stream)) return rc;
    return launch_range_combine(pair, 1, reinterpret_cast<int*>(status_dev), stream);
}
