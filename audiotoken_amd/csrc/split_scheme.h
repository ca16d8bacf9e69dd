// Device-side helpers of the two operand-split schemes (gemm_bf16x3.h): piece types, the MFMA of each scheme, and the split
// of fp32 values into pieces. Shared by the split GEMM and by every producer that writes its output directly as pieces.
#pragma once
#include "at_common.h"
#include "gemm_bf16x3.h"

namespace at {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// The two operand schemes (gemm_bf16x3.h). `prod_a/prod_w` list the cross products in the order they are accumulated, smallest first.
struct SchemeBf16x3 {
    typedef __bf16 T; typedef bf16x8 V8; typedef bf16x4 V4;
    static constexpr int NP = 3, NPROD = 6;
    static constexpr bool RANGE_CHECK = false;
    __device__ static constexpr int prod_a(int t) { constexpr int v[6] = {0, 2, 1, 0, 1, 0}; return v[t]; }
    __device__ static constexpr int prod_w(int t) { constexpr int v[6] = {2, 0, 1, 1, 0, 0}; return v[t]; }
    __device__ static __forceinline__ f16v mfma(V8 w, V8 a, f16v c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, a, c, 0, 0, 0); }
    __device__ static __forceinline__ f4 mfma16(V8 w, V8 a, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, a, c, 0, 0, 0); }
};
struct SchemeF16x2 {
    typedef _Float16 T; typedef f16x8 V8; typedef f16x4 V4;
    static constexpr int NP = 2, NPROD = 3;
    static constexpr bool RANGE_CHECK = true;
    __device__ static constexpr int prod_a(int t) { constexpr int v[3] = {0, 1, 0}; return v[t]; }
    __device__ static constexpr int prod_w(int t) { constexpr int v[3] = {1, 0, 0}; return v[t]; }
    __device__ static __forceinline__ f16v mfma(V8 w, V8 a, f16v c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(w, a, c, 0, 0, 0); }
    __device__ static __forceinline__ f4 mfma16(V8 w, V8 a, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(w, a, c, 0, 0, 0); }
};

// a scheme with its range check compiled out (operands known to fit: weights scaled on the host, LSTM states in (-1, 1))
template <class SC>
struct SchemeNoCheck : SC { static constexpr bool RANGE_CHECK = false; };

// Range bookkeeping of the fp16 scheme. A split writer keeps the running maximum of |x * scale| over everything it splits (one v_max3 per value
// pair; `over |= split4(...)` reads as before) and publishes it ONCE at the end of the kernel: bit XB_STATUS_F16_OVERFLOW of *status when the maximum
// does not fit fp16 (65504; an infinity fails it — fmaxf drops a NaN, which can only descend from an infinity that was flagged where it arose, or from a NaN in
// the caller's input: XB_STATUS_NONFINITE, raised by the quantisers, covers that), and, when the launch has a
// census word, the maximum itself (atomicMax on the float's bits: non-negative floats order like integers) — the per-site headroom that
// at_*_range_report() returns: how close real activations come to the fp16 range is measured, not assumed.
struct RangeMax {
    float m = 0.f;
    __device__ __forceinline__ RangeMax& operator|=(float quad_max) { m = fmaxf(m, quad_max); return *this; }
    __device__ __forceinline__ RangeMax& operator|=(const RangeMax& o) { m = fmaxf(m, o.m); return *this; }
    __device__ __forceinline__ bool overflow() const { return !(m <= 65504.0f); }
};
// one atomic per WAVE at most: butterfly maximum over the wave's active lanes (an inactive lane contributes 0), then the first active lane
// publishes — and only when it raises the census word (after the first waves of a launch most are already below it)
__device__ __forceinline__ void range_publish(int* status, int* census, const RangeMax& r) {
    float m = r.m;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    if (lane == __builtin_amdgcn_readfirstlane(lane)) {
        if (status && !(m <= 65504.0f)) atomicOr(status, XB_STATUS_F16_OVERFLOW);
        if (census) {
            const int bits = __float_as_int(m);   // m >= 0 (or +inf); a NaN cannot come out of fmaxf unless every operand was one
            // Agent-scope loads (L2-served: this CU's L1 is not coherent). A NEGATIVE word is a LINK: the model handles keep one flag word per (layer, site) — so
            // that the range fallback knows which layer overflowed — but ONE census word per site, shared by all layers; the per-layer rows hold
            // {flag, -(distance in ints to the shared census word)}. (Round 4: with a freshly zeroed census word per layer the first thousands of
            // waves of every launch all saw 0 and all issued the atomic: + 35 us per LayerNorm launch, 2-4 ms per semantic_m step.)
            int cur = __hip_atomic_load(census, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cur < 0) {
                census += cur;
                cur = __hip_atomic_load(census, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (bits > cur) atomicMax(census, bits);
        }
    }
}

// a -> NP pieces: p[0] = round(a), p[1] = round(a - p[0]), ... (every subtraction is exact)
template <class SC>
__device__ __forceinline__ void split_n(float a, typename SC::T (&p)[SC::NP]) {
#pragma unroll
    for (int i = 0; i < SC::NP; ++i) {
        p[i] = (typename SC::T)a;
        a -= (float)p[i];
    }
}
// four already scaled values -> fp16 hi quad and lo quad (see split4)
template <class V4>
__device__ __forceinline__ void split_pair_f16(float a0, float a1, float b0, float b1, V4& hi4, V4& lo4) {
    unsigned ha, hb, la, lb;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(ha) : "v"(a0), "v"(a1));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hb) : "v"(b0), "v"(b1));
    asm("v_fma_mixlo_f16 %0, -%1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(la) : "v"(ha), "v"(a0));
    asm("v_fma_mixhi_f16 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(la) : "v"(ha), "v"(a1));
    asm("v_fma_mixlo_f16 %0, -%1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lb) : "v"(hb), "v"(b0));
    asm("v_fma_mixhi_f16 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lb) : "v"(hb), "v"(b1));
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    static_assert(sizeof(V4) == sizeof(u2), "a quad of 16-bit pieces is two dwords");
    hi4 = __builtin_bit_cast(V4, u2{ha, hb});
    lo4 = __builtin_bit_cast(V4, u2{la, lb});
}

// four values * scale -> NP pieces of four; returns max |v * scale| of the four for the caller's RangeMax (0 for a scheme without a range check)
template <class SC>
__device__ __forceinline__ float split4(const f4& v, float scale, typename SC::V4 (&p)[SC::NP]) {
    if constexpr (SC::NP == 2) {
        // the fp16 scheme on value PAIRS, 2 instructions per value (round 5; 3 before): hi = rne(x s) by v_cvt_pk_f16_f32 on the scaled pair
        // (v_pk_mul_f32: the scale is a power of two, x s exact), lo = rne(x s - hi) by v_fma_mixlo / mixhi_f16 — (-hi as fp16) * 1.0 + x s with the
        // exact difference formed inside the fma and rounded once: the very values of convert, convert back, subtract, convert (x s - hi is exact in
        // fp32), without the two v_cvt_f32_f16 and the packed subtract. (attention_f16x2_w8.hip has split its probabilities this way since round 4.)
        typedef float f2 __attribute__((ext_vector_type(2)));
        const f2 a = f2{v[0], v[1]} * scale, b = f2{v[2], v[3]} * scale;
        float over = 0.f;
        if constexpr (SC::RANGE_CHECK) over = fmaxf(fmaxf(fabsf(a[0]), fabsf(a[1])), fmaxf(fabsf(b[0]), fabsf(b[1])));
        split_pair_f16(a[0], a[1], b[0], b[1], p[0], p[1]);
        return over;
    } else {
        float over = 0.f;
        if constexpr (SC::RANGE_CHECK) over = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))) * fabsf(scale);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float x = v[k] * scale;
            typename SC::T q[SC::NP];
            split_n<SC>(x, q);
#pragma unroll
            for (int i = 0; i < SC::NP; ++i) p[i][k] = q[i];
        }
        return over;
    }
}

// split4 for values the caller has ALREADY multiplied by the site's scale (an epilogue that folds the power of two into its last multiply: exact)
template <class SC>
__device__ __forceinline__ float split4_prescaled(const f4& vs, typename SC::V4 (&p)[SC::NP]) {
    static_assert(SC::NP == 2, "prescaled split: the fp16 scheme");
    float over = 0.f;
    if constexpr (SC::RANGE_CHECK) over = fmaxf(fmaxf(fabsf(vs[0]), fabsf(vs[1])), fmaxf(fabsf(vs[2]), fabsf(vs[3])));
    split_pair_f16(vs[0], vs[1], vs[2], vs[3], p[0], p[1]);
    return over;
}

// pieces of the 4 consecutive columns col .. col + 3 (col % 4 == 0) of row `row`, into K-blocked pieces [NP][K/16][rows_pad][16]
// (piece_stride = rows_pad * K elements); one 8-byte store per piece. Returns max |v * scale| (split4).
template <class SC>
__device__ __forceinline__ float store_pieces4(typename SC::T* out, long long piece_stride, long long rows_pad, long long row, int col, const f4& v, float scale) {
    typename SC::V4 p[SC::NP];
    const float over = split4<SC>(v, scale, p);
    typename SC::T* d = out + ((long long)(col >> 4) * rows_pad + row) * 16 + (col & 15);
#pragma unroll
    for (int i = 0; i < SC::NP; ++i) *reinterpret_cast<typename SC::V4*>(d + i * piece_stride) = p[i];
    return over;
}

}  // namespace at
