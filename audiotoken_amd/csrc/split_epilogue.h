// Epilogues of the split GEMMs (gemm_bf16x3.hip, gemm_f16x2_tg.hip): what happens to four consecutive output columns n .. n + 3 of
// output row m once their accumulators are final. One object per thread; `apply<E>` is instantiated per epilogue mode.
#pragma once
#include "split_scheme.h"

namespace at {

// FASTDIV: the caller filled a.fdS / a.fdS2 (the two-group kernel's launcher does)
template <class SC, bool FASTDIV = false>
struct XbEpilogue {
    typedef typename SC::T PT;
    const Bf16x3Args& a;
    const int clip;
    float* Cb;
    const float* Rb;
    RangeMax over;

    __device__ XbEpilogue(const Bf16x3Args& args, int clip_) : a(args), clip(clip_) {
        Cb = a.C ? a.C + (long long)clip * a.M * a.ldc : nullptr;
        Rb = a.R ? a.R + (long long)clip * a.M * a.ldr : nullptr;
    }

    // split outputs: [pieces][batch][blocks][phases][pad][16]; output row m lives in plane m % phases at index m / phases + front
    // PRE: `v` already carries a.split_scale (the swish / GELU epilogues fold the power of two into their last multiply: bit-identical, one instruction less)
    template <bool FD = false, bool PRE = false>
    __device__ __forceinline__ void write_split(__bf16* S_, int pad, int phases, int front, int blocks, int block0, int m, int n, const f4& v, const FastDivU* fd = nullptr) {
        PT* S = reinterpret_cast<PT*>(S_);
        const int nb = blocks > 0 ? blocks : a.N / 16;
        const long long s_clip = (long long)pad * phases * nb * 16;   // elements of one clip of one piece
        const long long psS = s_clip * a.batch;
        typename SC::V4 p[SC::NP];
        if constexpr (PRE) over |= split4_prescaled<SC>(v, p);
        else over |= split4<SC>(v, a.split_scale, p);
        const int sq = FD ? (int)fd->div((unsigned)m) : m / phases, sp = m - sq * phases;
        PT* d = S + clip * s_clip + (((long long)(block0 + (n >> 4)) * phases + sp) * pad + sq + front) * 16 + (n & 15);
#pragma unroll
        for (int i = 0; i < SC::NP; ++i) *reinterpret_cast<typename SC::V4*>(d + i * psS) = p[i];
    }

    // the bias / residual values of an output quad, for callers that load them AHEAD of the stores: inside apply() every load sits behind
    // the previous quad's store (C, R and the piece buffers may alias as far as the compiler knows — R == C for an in-place residual), i.e. one
    // memory round trip per quad: in-kernel stamps showed 55-83 k cycles for the residual epilogue of a 256 x 256 tile (32 quads per lane)
    __device__ __forceinline__ f4 load_bias(int n) const { return a.bias ? *reinterpret_cast<const f4*>(a.bias + n) : f4{0.f, 0.f, 0.f, 0.f}; }
    __device__ __forceinline__ f4 load_residual(int m, int n) const {
        return Rb ? *reinterpret_cast<const f4*>(Rb + (long long)m * a.ldr + n) : f4{0.f, 0.f, 0.f, 0.f};
    }
    // apply() with the bias quad (and, for the two plain modes, the residual quad) supplied by the caller. PH1: the caller has checked
    // a.Sphases == 1 (the piece output of a plain linear layer): the generic plane / index split of write_split is an integer division per
    // quad by a run-time value — the fast path's address is base + block * pad + row (in the swish epilogue of a 256 x 256 tile the
    // division sequences and 64-bit multiplies were ~500 of 2 000 instructions per lane, a third of its vector-unit time).
    template <int E, bool PH1 = false>
    __device__ __forceinline__ void apply_with(int m, int n, f4 v, const f4& bias4, const f4& res4) {
        if constexpr (SC::RANGE_CHECK) v *= a.acc_scale;
        v += bias4;
        finish_quad<E, true, PH1>(m, n, v, res4);
    }
    // the value of a plain (fp32 row-major) output quad: the statement sequence of apply_with<E> + finish_quad<E, true> for E = LINEAR / GELU, for
    // callers that address C / R themselves (the two-group kernel's row-layout epilogue: tile base in scalar registers + 32-bit lane offsets)
    template <int E>
    __device__ __forceinline__ f4 plain_value(f4 v, const f4& bias4, const f4& res4) const {
        if constexpr (SC::RANGE_CHECK) v *= a.acc_scale;
        v += bias4;
        if constexpr (E == XB_EPI_GELU) { v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w); }
        v *= a.alpha;
        v += res4;
        return v;
    }
    // phases == 1: [pieces][batch][blocks][pad][16]
    template <bool PRE = false>
    __device__ __forceinline__ void write_split_ph1(__bf16* S_, int pad, int front, int blocks, int block0, int m, int n, const f4& v) {
        PT* S = reinterpret_cast<PT*>(S_);
        const int nb = blocks > 0 ? blocks : a.N / 16;
        const long long s_clip = (long long)pad * nb * 16;
        const long long psS = s_clip * a.batch;
        typename SC::V4 p[SC::NP];
        if constexpr (PRE) over |= split4_prescaled<SC>(v, p);
        else over |= split4<SC>(v, a.split_scale, p);
        PT* d = S + clip * s_clip + (long long)(block0 + (n >> 4)) * pad * 16 + (m + front) * 16 + (n & 15);
#pragma unroll
        for (int i = 0; i < SC::NP; ++i) *reinterpret_cast<typename SC::V4*>(d + i * psS) = p[i];
    }

    template <int E>
    __device__ __forceinline__ void apply(int m, int n, f4 v) {
        if constexpr (SC::RANGE_CHECK) v *= a.acc_scale;   // exact: a power of two (1 for the bf16 scheme)
        if (a.bias) v += *reinterpret_cast<const f4*>(a.bias + n);
        finish_quad<E, false>(m, n, v, f4{0.f, 0.f, 0.f, 0.f});
    }

    template <int E, bool HAVE_RES, bool PH1 = false>
    __device__ __forceinline__ void finish_quad(int m, int n, f4 v, const f4& res4) {
        if constexpr (E == XB_EPI_RAW_ELU_SPLIT2) {
            write_split<FASTDIV>(a.S, a.Spad, a.Sphases, a.Sfront, a.Sblocks, a.Sblock0, m, n, v, &a.fdS);
            const f4 e = {elu1(v.x), elu1(v.y), elu1(v.z), elu1(v.w)};
            write_split<FASTDIV>(a.S2, a.S2pad, a.S2phases, a.S2front, a.S2blocks, a.S2block0, m, n, e, &a.fdS2);
        } else if constexpr (E == XB_EPI_SWISH_SPLIT || E == XB_EPI_GELU_SPLIT || E == XB_EPI_ELU_SPLIT) {
            // fp16 scheme, swish / GELU: the site's power-of-two scale s rides on the activation's last multiply — swish: x * rcp((1 + e) / s) with (1 + e) / s
            // formed by ONE fma(e, 1 / s, 1 / s) in place of the add (v_rcp_f32 of a power-of-two multiple is that multiple of the reciprocal: only the
            // exponent moves); GELU: (0.5 s) y (...) — so the values handed to the split are bit for bit s times the unscaled activation and the split's
            // own v_pk_mul_f32 goes (round 5)
            constexpr bool PRE = SC::NP == 2 && (E == XB_EPI_SWISH_SPLIT || E == XB_EPI_GELU_SPLIT);
            const float s_ = PRE ? a.split_scale : 1.0f, inv_s = PRE ? 1.0f / a.split_scale : 1.0f;
            f4 w;
            if constexpr (E == XB_EPI_GELU_SPLIT) {
                // two values per issue slot on the packed fp32 instructions (round 6: 89 -> 62 vector instructions per quad incl. the split; per element the same
                // operations in the same order as gelu_erf_scaled: bit-identical, the pinned checksums cannot move)
                typedef float f2_ __attribute__((ext_vector_type(2)));
                const f2_ g0 = gelu_erf_scaled2(f2_{v[0], v[1]}, 0.5f * s_), g1 = gelu_erf_scaled2(f2_{v[2], v[3]}, 0.5f * s_);
                w = f4{g0[0], g0[1], g1[0], g1[1]};
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    w[k] = E == XB_EPI_ELU_SPLIT ? elu1(v[k])
                                                 : v[k] * __builtin_amdgcn_rcpf(fmaf(__expf(-v[k]), inv_s, inv_s));   // = s x sigmoid(x): v_exp_f32 + v_rcp_f32 (~1 ulp each), as the fp32 GEMM's epilogue
            }
            if constexpr (PH1) write_split_ph1<PRE>(a.S, a.Spad, a.Sfront, a.Sblocks, a.Sblock0, m, n, w);
            else write_split<FASTDIV, PRE>(a.S, a.Spad, a.Sphases, a.Sfront, a.Sblocks, a.Sblock0, m, n, w, &a.fdS);
        } else if constexpr (E == XB_EPI_QKV) {
            if (n < a.qkv_hid) {
                *reinterpret_cast<f4*>(Cb + (long long)m * a.ldc + n) = v;
            } else {
                const int which = n >= 2 * a.qkv_hid ? 1 : 0;
                typename SC::V4 p[SC::NP];
                over |= split4<SC>(v, a.split_scale, p);
                PT* d = reinterpret_cast<PT*>(a.S) + ((long long)which * SC::NP * a.Spad + m) * a.qkv_hid + (n - (1 + which) * a.qkv_hid);
#pragma unroll
                for (int i = 0; i < SC::NP; ++i) *reinterpret_cast<typename SC::V4*>(d + (long long)i * a.Spad * a.qkv_hid) = p[i];
            }
        } else if constexpr (E == XB_EPI_GLU) {
            float2 o;
            o.x = v.x * sigmoidf_(v.y);
            o.y = v.z * sigmoidf_(v.w);
            *reinterpret_cast<float2*>(Cb + (long long)m * a.ldc + (n >> 1)) = o;
        } else {
            if constexpr (E == XB_EPI_GELU) { v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w); }
            v *= a.alpha;
            if constexpr (HAVE_RES) v += res4;
            else if (Rb) v += *reinterpret_cast<const f4*>(Rb + (long long)m * a.ldr + n);
            *reinterpret_cast<f4*>(Cb + (long long)m * a.ldc + n) = v;
        }
    }

    __device__ __forceinline__ void finish() {
        if constexpr (SC::RANGE_CHECK)
            range_publish(a.status, a.status ? a.status + 1 : nullptr, over);
    }
};

}  // namespace at
