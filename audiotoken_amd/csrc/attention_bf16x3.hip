// Relative-position self-attention (reference audiotoken/modeling_wav2vec2_bert.py:46-73) with both matrix products on the
// bf16 matrix cores as exact 3-way operand splits (see gemm_bf16x3.hip for the arithmetic): q, k, v and the probabilities p are
// split in registers / at staging time into three bf16 pieces each, and S^T = K.Q^T and O^T += V^T.P^T take six
// v_mfma_f32_32x32x16_bf16 per fp32-equivalent 32x32x16 step (6/16 of the fp32 MFMA time of relpos_attention_kernel).
// Same flash structure as the fp32 kernel (w2vbert_kernels.hip): workgroup = 128 queries of one (clip, head), wave = 32 queries
// = one MFMA column tile; keys on the MFMA rows so a lane owns, for ONE query, 16 keys of each 32-key tile and P never leaves
// registers; rel-pos bias table q.E^T (73 buckets) in LDS with far-field constants; exp2-domain online softmax.
// K/V tiles of 32 keys: K as [3 pieces][32 keys][64 d], V transposed [3 pieces][64 d][32 keys] with the keys of a 16-group
// permuted so that the 8 keys a lane's P registers hold for one MFMA k-step are contiguous (one ds_read_b128 per fragment).
// 71 KB of LDS: two workgroups per CU. Also serves HuBERT (12 heads, no rel-pos bias).
//
// Round 2: the kernel is templated on the operand scheme (split_scheme.h). With the two-piece fp16 scheme q, k, v are scaled by
// XB_F16_ACT_SCALE and the probabilities by 2^10 before they are split (all exact; undone in the softmax scale and the final 1 / l),
// three products per step instead of six, 61 KB of LDS; values beyond the fp16 range raise XB_STATUS_F16_OVERFLOW.
#include "at_common.h"
#include "w2vbert_kernels.h"
#include "split_scheme.h"

#include <cstdio>

namespace at {

typedef unsigned int u4 __attribute__((ext_vector_type(4)));

// -DAX_DEBUG_STAMPS (tools/ax_stamps.sh; never in the product build): wave 0 of one workgroup sums the cycle counter over the phases of its key tiles
#ifdef AX_DEBUG_STAMPS
__device__ unsigned long long ax_stamps[12];
#define AX_T(i) const unsigned long long ax_t##i = __builtin_readcyclecounter()
#define AX_ACC(k, a_, b_) ax_d[k] += ax_t##b_ - ax_t##a_
#else
#define AX_T(i) do {} while (0)
#define AX_ACC(k, a_, b_) do {} while (0)
#endif

constexpr int AX_QB = 128, AX_KB = 32;
constexpr int AX_KLD = 72;    // bf16 per K row (64 d + 8 pad: 144-byte stride, conflict-free 16-byte fragment reads)
constexpr int AX_VLD = 40;    // bf16 per V^T row (32 keys + 8 pad)
constexpr int AX_QE_LD = 81;
constexpr float AX_SCALE2 = 0.125f * 1.4426950408889634f;
constexpr float AX_P_SCALE = 1024.0f;   // fp16 scheme: probabilities (<= 1) are split as p * 2^10
template <int NP> constexpr int ax_k_elems() { return NP * AX_KB * AX_KLD; }
template <int NP> constexpr int ax_v_elems() { return NP * 64 * AX_VLD; }
template <int NP> constexpr size_t ax_lds_bytes() { return (size_t)(ax_k_elems<NP>() + ax_v_elems<NP>()) * 2 + (size_t)(AX_QB * AX_QE_LD + AX_KB + 4) * 4; }

// KVP (fp16 scheme only): k and v arrive ALREADY split — row-major fp16 pieces [which][piece][rows_pad][hid] written by the q / k / v
// projection's epilogue (XB_EPI_QKV) — so a K / V tile is staged with four 16-byte loads and stores per thread and no vector arithmetic (the
// splits were a third of this kernel's vector instructions, repeated by each of the 12 query-tile workgroups of a (clip, head)); V stays
// row-major [key][64 d + 32 pad] in LDS and P.V takes its transposed fragments with ds_read_b64_tr_b16 (4 keys x 16 d per 16 lanes).
template <class SC, bool KVP>
__global__ __launch_bounds__(256, 2) void relpos_attention_x3_kernel(const float* __restrict__ qkv, const float* __restrict__ amask,
                                                                     const float* __restrict__ dist_emb, float* __restrict__ ctx, int T, int hid,
                                                                     int* __restrict__ status, typename SC::T* __restrict__ ctx_pieces, long long rows_pad,
                                                                     int nheads, int nclips, const typename SC::T* __restrict__ kv_pieces) {
    typedef typename SC::T PT;
    typedef typename SC::V8 V8;
    typedef typename SC::V4 V4;
    constexpr int NP = SC::NP;
    constexpr int AX_VROW = 96;   // KVP: V rows [key][64 d + 32 pad] (192 B: the 4 rows of a transposing read fall on 4 disjoint bank quarters)
    constexpr int AX_K_ELEMS = ax_k_elems<NP>(), AX_V_ELEMS = KVP ? NP * AX_KB * AX_VROW : ax_v_elems<NP>();
    // operand scales (1 for the bf16 scheme): q, k, v * XS; p * PS. S = acc / XS^2, O = acc / (PS XS)
    constexpr float XS = SC::RANGE_CHECK ? XB_F16_ACT_SCALE : 1.0f, PS = SC::RANGE_CHECK ? AX_P_SCALE : 1.0f;
    constexpr float S_SCALE2 = AX_SCALE2 / (XS * XS);
    RangeMax over;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    // KVP: the K / V tile, its key bias and its any-masked flag are DOUBLE-buffered (tile kt + 1 is written while tile kt is consumed: one barrier per
    // tile instead of two — the first one cost ~280 of ~3500 cycles per tile, tools/ax_stamps.sh). Two workgroups per CU must still fit 160 KB, so the
    // rel-pos table keeps only the 73 buckets that are read (row stride 75 instead of the 80 columns the MFMA tiles produce + 1): 81 680 B.
    constexpr int NBUF = KVP ? 2 : 1;
    constexpr int QE_LD = KVP ? 75 : AX_QE_LD;
    constexpr int AX_KV_ELEMS = AX_K_ELEMS + AX_V_ELEMS;
    PT* Ks0 = reinterpret_cast<PT*>(smem_raw);                   // [NBUF][ K [NP][32 keys][72] | V ]
    float* QE = reinterpret_cast<float*>(Ks0 + NBUF * AX_KV_ELEMS);   // [128 queries][QE_LD]: log2(e)/8 * q.E[bucket]
    float* kb0 = QE + AX_QB * QE_LD;                             // [NBUF][32] additive key bias: 0 / finfo.min (padded) / -inf (beyond T)
    int* kb_any0 = reinterpret_cast<int*>(kb0 + NBUF * AX_KB);   // [NBUF]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l32 = lane & 31, hh = lane >> 5;
    // 1-D grid, XCD-aware: workgroups g and g + 8 share an XCD (round-robin dispatch; speed only), so XCD x takes a contiguous range of
    // (clip, head) pairs and runs the query tiles of one pair back to back — its K / V rows (768 KB) are then fetched into ONE L2 and
    // re-used by the other query tiles there, instead of by whichever XCDs the tiles of a 3-D grid land on (PMC: 6.8 GB fetched per launch
    // against 1.2 GB of qkv)
    const int nqt = (T + AX_QB - 1) / AX_QB;
    const int nblk = gridDim.x, per_xcd = (nblk + 7) >> 3;
    const int lid = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (lid >= nqt * nheads * nclips) return;
    const int qt = lid % nqt, pair = lid / nqt;
    const int h = pair % nheads, b = pair / nheads;
    const int l0 = qt * AX_QB;
    const long long rowbase = (long long)b * T;
    const int LD = 3 * hid;
    const float* qp = qkv + h * 64;
    const float* kp = qkv + hid + h * 64;
    const float* vp = qkv + 2 * hid + h * 64;
    const bool relpos = dist_emb != nullptr;

    // ---- rel-pos table QE = log2(e)/8 * q . E^T on the fp32 MFMA, once per workgroup (as relpos_attention_kernel) ---------------
    if (relpos) {
        const int r16 = lane & 15, qd = lane >> 4;
        f4 qf[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int l = l0 + wave * 32 + i * 16 + r16;
            const int lc = l < T ? l : T - 1;
#pragma unroll
            for (int c = 0; c < 4; ++c) qf[i][c] = *reinterpret_cast<const f4*>(qp + (rowbase + lc) * LD + c * 16 + qd * 4);
        }
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
                float* dst = QE + (wave * 32 + i * 16 + r16) * QE_LD + bt * 16 + qd * 4;
#pragma unroll
                for (int reg = 0; reg < 4; ++reg)
                    if (bt * 16 + qd * 4 + reg < QE_LD) dst[reg] = AX_SCALE2 * acc[reg];
            }
        }
    }
    // ---- this lane's query (column l32 of the wave's tile): 3 bf16 pieces of q[lq][dstep*16 + 8*hh .. +7] -------------------
    const int lq = l0 + wave * 32 + l32;
    V8 qpc[NP][4];
    {
        const int lc = lq < T ? lq : T - 1;
#pragma unroll
        for (int ds = 0; ds < 4; ++ds) {
            const f4 a = *reinterpret_cast<const f4*>(qp + (rowbase + lc) * LD + ds * 16 + 8 * hh);
            const f4 c = *reinterpret_cast<const f4*>(qp + (rowbase + lc) * LD + ds * 16 + 8 * hh + 4);
            V4 pa[NP], pc[NP];
            over |= split4<SC>(a, XS, pa);
            over |= split4<SC>(c, XS, pc);
#pragma unroll
            for (int i = 0; i < NP; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) { qpc[i][ds][k] = pa[i][k]; qpc[i][ds][4 + k] = pc[i][k]; }
        }
    }
    f16v oacc[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
    float mrun = -INFINITY, lrun = 0.f;
    const int wl_min = l0 + __builtin_amdgcn_readfirstlane(wave) * 32, wl_max = wl_min + 31;
    const float FMIN = -3.4028234663852886e38f;
    const int nkt = (T + AX_KB - 1) / AX_KB;

    // ---- K/V staging: global (fp32, per-clip buffer descriptors: rows >= T read 0) -> registers -> split -> LDS -----------------
    const int clip_bytes = T * LD * 4;
    const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc((void*)(kp + rowbase * LD), 0, clip_bytes - (hid + h * 64) * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc((void*)(vp + rowbase * LD), 0, clip_bytes - (2 * hid + h * 64) * 4, 0x00020000);
    const int sk_key = tid >> 3, sk_d = (tid & 7) * 8;          // K: thread -> key, 8 consecutive d
    const int sv_d = tid & 63, sv_g = tid >> 6;                 // V: thread -> dv, keys 8*sv_g .. +7
    const int row_bytes = LD * 4;
    const int k_voff = (sk_key * LD + sk_d) * 4, v_voff = (sv_g * 8 * LD + sv_d) * 4;
    u4 kreg[2];
    unsigned vreg[8];
    float am = 0.f;
    // KVP: per-clip buffer descriptors of the four piece planes (k hi, k lo, v hi, v lo), as for the fp32 rows: keys >= T read 0.
    // thread -> (key tid >> 3, 8 consecutive d of this head)
    __amdgpu_buffer_rsrc_t prs[4] = {krs, krs, krs, krs};
    u4 pk[2], pv[2];
    int p_voff = 0;
    if constexpr (KVP) {
        const long long ps = rows_pad * (long long)hid;                       // elements of one piece plane
#pragma unroll
        for (int j = 0; j < 4; ++j)
            prs[j] = __builtin_amdgcn_make_buffer_rsrc((void*)(kv_pieces + j * ps + rowbase * hid + h * 64), 0, (T * hid - h * 64) * 2, 0x00020000);
        p_voff = (sk_key * hid + sk_d) * 2;
    }
    auto prefetch = [&](int kt) {
        if constexpr (KVP) {
            const int o = p_voff + kt * AX_KB * hid * 2;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                pk[i] = __builtin_amdgcn_raw_buffer_load_b128(prs[i], o, 0, 0);
                pv[i] = __builtin_amdgcn_raw_buffer_load_b128(prs[2 + i], o, 0, 0);
            }
        } else {
            const int toff = kt * AX_KB * row_bytes;
            kreg[0] = __builtin_amdgcn_raw_buffer_load_b128(krs, k_voff + toff, 0, 0);
            kreg[1] = __builtin_amdgcn_raw_buffer_load_b128(krs, k_voff + toff + 16, 0, 0);
            int vo = v_voff + toff;
#pragma unroll
            for (int e = 0; e < 8; ++e) { vreg[e] = __builtin_amdgcn_raw_buffer_load_b32(vrs, vo, 0, 0); vo += row_bytes; }
        }
        const int rr = kt * AX_KB + (tid & 31);
        am = amask[rowbase + (rr < T ? rr : T - 1)];
    };
    // registers (tile kt's prefetch) -> LDS buffer `buf`: K / V pieces or fp32 rows split here, the key bias and its any-masked flag
    auto stage = [&](int kt, int buf) {
        const int r0 = kt * AX_KB;
        PT* Ks = Ks0 + buf * AX_KV_ELEMS;
        PT* Vt = Ks + AX_K_ELEMS;
        if constexpr (KVP) {
            PT* kd = Ks + sk_key * AX_KLD + sk_d;
            PT* vd = Vt + sk_key * AX_VROW + sk_d;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                *reinterpret_cast<u4*>(kd + i * AX_KB * AX_KLD) = pk[i];
                *reinterpret_cast<u4*>(vd + i * AX_KB * AX_VROW) = pv[i];
            }
        } else {
            V8 kpc[NP];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const f4 kv = {__uint_as_float(kreg[u][0]), __uint_as_float(kreg[u][1]), __uint_as_float(kreg[u][2]), __uint_as_float(kreg[u][3])};
                V4 p4[NP];
                over |= split4<SC>(kv, XS, p4);
#pragma unroll
                for (int i = 0; i < NP; ++i)
#pragma unroll
                    for (int k = 0; k < 4; ++k) kpc[i][4 * u + k] = p4[i][k];
            }
            PT* kd = Ks + sk_key * AX_KLD + sk_d;
#pragma unroll
            for (int i = 0; i < NP; ++i) *reinterpret_cast<V8*>(kd + i * AX_KB * AX_KLD) = kpc[i];
            // V^T: keys 8*sv_g + e -> 16-group g = sv_g >> 1, position 8*((e >> 2)) + 4*(sv_g & 1) + (e & 3) inside it
            PT* vd = Vt + sv_d * AX_VLD + (sv_g >> 1) * 16 + 4 * (sv_g & 1);
#pragma unroll
            for (int u = 0; u < 2; ++u) {   // u = e >> 2 selects the half hh whose lanes read this key quartet
                const f4 vv = {__uint_as_float(vreg[4 * u]), __uint_as_float(vreg[4 * u + 1]), __uint_as_float(vreg[4 * u + 2]), __uint_as_float(vreg[4 * u + 3])};
                V4 p4[NP];
                over |= split4<SC>(vv, XS, p4);
#pragma unroll
                for (int i = 0; i < NP; ++i) *reinterpret_cast<V4*>(vd + 8 * u + i * 64 * AX_VLD) = p4[i];
            }
        }
        if (tid < AX_KB) {
            const int rr = r0 + tid;
            const float kbv = rr < T ? (am != 0.f ? 0.f : FMIN) : -INFINITY;
            kb0[buf * AX_KB + tid] = kbv;
            const unsigned long long anyb = __builtin_amdgcn_ballot_w64(kbv != 0.f);
            if (tid == 0) kb_any0[buf] = anyb != 0ull ? 1 : 0;
        }
    };
    prefetch(0);
    if constexpr (KVP) {   // prologue of the double-buffered form: tile 0 staged, tile 1 in flight
        stage(0, 0);
        prefetch(1);
        __syncthreads();   // (also orders the QE stores before first use)
    }
#ifdef AX_DEBUG_STAMPS
    unsigned long long ax_d[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long ax_begin = __builtin_readcyclecounter();
#endif
    for (int kt = 0; kt < nkt; ++kt) {
        const int r0 = kt * AX_KB;
        const int cur = KVP ? (kt & 1) : 0;
        const PT* Ks = Ks0 + cur * AX_KV_ELEMS;
        const PT* Vt = Ks + AX_K_ELEMS;
        const float* kb = kb0 + cur * AX_KB;
        AX_T(0);
        if constexpr (!KVP) __syncthreads();   // previous tile fully consumed (also orders the QE stores before first use)
        AX_T(1);
        if constexpr (KVP) {
            // the other buffer was last read before the barrier that ended the previous iteration. Unconditional: past the last tile the stores land in
            // a buffer nobody reads and the loads are clipped by the buffer descriptors (zeros) — no branch, so the LDS stores and the loads of the next
            // tiles are scheduled among this tile's MFMAs instead of in a block of their own
            stage(kt + 1, cur ^ 1);
        } else {
            stage(kt, 0);
        }
        AX_T(2);
        if constexpr (!KVP) __syncthreads();
        AX_T(3);
        if constexpr (KVP) prefetch(kt + 2);
        else if (kt + 1 < nkt) prefetch(kt + 1);
        AX_T(4);
        // ---- S^T = K . Q^T: lane holds s[r] = q_lq . k_(r0 + 8*(r/4) + 4*hh + r%4) ------------------------------------------------
        f16v s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int ds = 0; ds < 4; ++ds) {
            V8 kf[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) kf[p] = *reinterpret_cast<const V8*>(Ks + p * AX_KB * AX_KLD + l32 * AX_KLD + ds * 16 + 8 * hh);
#pragma unroll
            for (int t = 0; t < SC::NPROD; ++t) s = SC::mfma(kf[SC::prod_a(t)], qpc[SC::prod_w(t)][ds], s);
        }
        AX_T(5);
        // ---- bias + mask, online softmax in the exp2 domain ------------------------------------------------------------------------
        const bool far_left = (r0 + AX_KB - 1) - wl_min <= -64;
        const bool far_right = r0 - wl_max >= 8;
        const bool plain = far_left || far_right || !relpos;
        const bool masked = __builtin_amdgcn_readfirstlane(kb_any0[cur]) != 0;
        const float* qe = QE + (wave * 32 + l32) * QE_LD;
        const float c_far = relpos ? (far_left ? qe[0] : qe[72]) : 0.f;
        float mx = -INFINITY;
        if (plain && !masked) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float sc = fmaf(S_SCALE2, s[r], c_far);
                s[r] = sc;
                mx = fmaxf(mx, sc);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = 8 * (r >> 2) + 4 * hh + (r & 3);
                float bias;
                if (plain) {
                    bias = c_far;
                } else {
                    int dd = (r0 + key) - lq;
                    dd = dd < -64 ? -64 : (dd > 8 ? 8 : dd);
                    bias = qe[dd + 64];
                }
                const float sc = fmaf(S_SCALE2, s[r], bias + kb[key]);
                s[r] = sc;
                mx = fmaxf(mx, sc);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float mnew = fmaxf(mrun, mx);
        float rs = 0.f;
        {
            typedef float f2_ __attribute__((ext_vector_type(2)));
            const f2_ m2 = {mnew, mnew};
#pragma unroll
            for (int r = 0; r < 16; r += 2) {   // the subtraction on value pairs (v_pk_add_f32); exp2 and the running sum stay in order
                const f2_ d = f2_{s[r], s[r + 1]} - m2;
                const float p0 = __builtin_amdgcn_exp2f(d[0]), p1 = __builtin_amdgcn_exp2f(d[1]);
                s[r] = p0; s[r + 1] = p1;
                rs += p0;
                rs += p1;
            }
        }
        rs += __shfl_xor(rs, 32);
        {
            // Unconditional rescale (round 3): alpha = exp2(0) = 1 exactly when the row maximum did not move, so the products are exact and the
            // results equal the branchy form's bit for bit; the branch cost a 32-register copy of the output accumulators on BOTH paths (the
            // compiler gave the rescaled values new registers), more vector-unit slots than the 16 packed multiplies — and this kernel is bound by
            // vector-unit issue (tools/ax_stamps.sh: ~450 instructions per 32-key tile against 24 MFMAs).
            const float alpha = __builtin_amdgcn_exp2f(mrun - mnew);   // exp2(-inf) = 0 on the first tile
            lrun = lrun * alpha + rs;
            mrun = mnew;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
        }
        AX_T(6);
        // ---- P pieces (B operand of P.V: k-step ks uses registers 8ks .. 8ks+7 = keys 16ks + {0..3, 8..11} + 4hh) -----------------
        V8 pp[NP][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int jq = 0; jq < 2; ++jq) {   // quads, so that the fp16 scheme's split compiles to the packed forms (split_scheme.h)
                const f4 pv = {s[8 * ks + 4 * jq], s[8 * ks + 4 * jq + 1], s[8 * ks + 4 * jq + 2], s[8 * ks + 4 * jq + 3]};
                typename SC::V4 q4[NP];
                split4<SchemeNoCheck<SC>>(pv, PS, q4);   // p <= 1: p * 2^10 always fits
#pragma unroll
                for (int i = 0; i < NP; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) pp[i][ks][4 * jq + j] = q4[i][j];
            }
        AX_T(7);
        // ---- O^T += V^T . P^T ---------------------------------------------------------------------------------------------------------
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                V8 vf[NP];
                if constexpr (KVP) {
                    // lane (row d = dt * 32 + l32, half hh) needs keys 16 ks + 4 hh + {0..3} and + 8: two transposing reads of a 4-key x 16-d
                    // block each; within its 16-lane group lane 4 q + p supplies the address of key q, d columns 4 p .. 4 p + 3
                    typedef short s4_ __attribute__((__vector_size__(4 * sizeof(short))));
                    const int li = lane & 15, d0 = dt * 32 + (lane & 16);
                    const PT* vb = Vt + (ks * 16 + 4 * hh + (li >> 2)) * AX_VROW + d0 + 4 * (li & 3);
#pragma unroll
                    for (int p = 0; p < NP; ++p) {
                        const s4_ lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4_*)(vb + p * AX_KB * AX_VROW));
                        const s4_ hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4_*)(vb + p * AX_KB * AX_VROW + 8 * AX_VROW));
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            vf[p][k] = __builtin_bit_cast(PT, (short)lo[k]);
                            vf[p][4 + k] = __builtin_bit_cast(PT, (short)hi[k]);
                        }
                    }
                } else {
#pragma unroll
                for (int p = 0; p < NP; ++p) vf[p] = *reinterpret_cast<const V8*>(Vt + p * 64 * AX_VLD + (dt * 32 + l32) * AX_VLD + ks * 16 + 8 * hh);
                }
#pragma unroll
                for (int t = 0; t < SC::NPROD; ++t) oacc[dt] = SC::mfma(vf[SC::prod_a(t)], pp[SC::prod_w(t)][ks], oacc[dt]);
            }
        if constexpr (KVP) __syncthreads();   // the next tile's buffer is complete and this one is free
        AX_T(8);
        AX_ACC(0, 0, 1); AX_ACC(1, 1, 2); AX_ACC(2, 2, 3); AX_ACC(3, 3, 4); AX_ACC(4, 4, 5); AX_ACC(5, 5, 6); AX_ACC(6, 6, 7); AX_ACC(7, 7, 8);
    }
#ifdef AX_DEBUG_STAMPS
    if (blockIdx.x == (gridDim.x / 2) && threadIdx.x == 0) {
        for (int i = 0; i < 8; ++i) ax_stamps[i] = ax_d[i];
        ax_stamps[8] = __builtin_readcyclecounter() - ax_begin;
        ax_stamps[9] = (unsigned long long)nkt;
    }
#endif
    // lane holds O[lq][dv = 32 dt + 8*(r/4) + 4 hh + r%4]
    if (lq < T) {
        const float inv = (1.0f / lrun) * (1.0f / (PS * XS));
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f4 v = {oacc[dt][4 * g] * inv, oacc[dt][4 * g + 1] * inv, oacc[dt][4 * g + 2] * inv, oacc[dt][4 * g + 3] * inv};
                if (ctx_pieces)   // straight to the output projection's GEMM operand (split_scheme.h): an 8-byte store per piece
                    over |= store_pieces4<SC>(ctx_pieces, rows_pad * hid, rows_pad, rowbase + lq, h * 64 + dt * 32 + 8 * g + 4 * hh, v, XS);
                else
                    *reinterpret_cast<f4*>(ctx + (rowbase + lq) * hid + h * 64 + dt * 32 + 8 * g + 4 * hh) = v;
            }
    }
    if constexpr (SC::RANGE_CHECK)
        range_publish(status, status ? status + 1 : nullptr, over);
}

template <class SC, bool KVP>
static int launch_ax(const float* qkv, const float* amask, const float* dist_emb, float* ctx, int B, int T, hipStream_t stream, int heads, int* status,
                     __bf16* ctx_pieces, long long rows_pad, const __bf16* kv_pieces) {
    const long long nblk = (long long)((T + AX_QB - 1) / AX_QB) * heads * B;
    dim3 grid((unsigned)((nblk + 7) / 8 * 8));
    constexpr size_t lds = KVP ? (size_t)2 * (ax_k_elems<SC::NP>() + SC::NP * AX_KB * 96) * 2 + (size_t)(AX_QB * 75 + 2 * AX_KB + 4) * 4 : ax_lds_bytes<SC::NP>();
    { static LdsAttrFlags lds_attr; if (int rc = set_max_dynamic_lds(lds_attr, relpos_attention_x3_kernel<SC, KVP>, lds)) return rc; }
    hipLaunchKernelGGL((relpos_attention_x3_kernel<SC, KVP>), grid, dim3(256), lds, stream, qkv, amask, dist_emb, ctx, T, heads * 64, status,
                       reinterpret_cast<typename SC::T*>(ctx_pieces), rows_pad, heads, B, reinterpret_cast<const typename SC::T*>(kv_pieces));
    AT_CHECK_HIP(hipGetLastError());
#ifdef AX_DEBUG_STAMPS
    {
        static int printed = 0;
        if (printed < 4) {
            ++printed;
            (void)hipStreamSynchronize(stream);
            unsigned long long hb[12];
            (void)hipMemcpyFromSymbol(hb, HIP_SYMBOL(ax_stamps), sizeof(hb));
            const double n = (double)(hb[9] ? hb[9] : 1);
            std::fprintf(stderr, "ax stamps B %d T %d heads %d kvp %d (cycles per 32-key tile, wave 0 of one workgroup): barrier1 %.0f  stage->LDS %.0f  barrier2 %.0f  prefetch issue %.0f  K reads + S mfma %.0f  softmax %.0f  P split %.0f  V reads + PV mfma %.0f | loop total %.0f per tile, %llu tiles\n",
                         B, T, heads, (int)KVP, hb[0] / n, hb[1] / n, hb[2] / n, hb[3] / n, hb[4] / n, hb[5] / n, hb[6] / n, hb[7] / n, hb[8] / n, hb[9]);
        }
    }
#endif
    return 0;
}

int launch_relpos_attention_x3(const float* qkv, const float* amask, const float* dist_emb, float* ctx, int B, int T, hipStream_t stream, int heads,
                               int scheme, int* status, __bf16* ctx_pieces, long long rows_pad, const __bf16* kv_pieces, int w8, const __bf16* dist_pieces, float dist_scale) {
    AT_REQUIRE(B >= 1 && T >= 1 && heads >= 1 && heads <= 64, "relpos_attention_x3: bad shape");
    AT_REQUIRE((long long)T * 3 * heads * 64 * 4 < (1ll << 31), "relpos_attention_x3: one clip's qkv rows exceed the buffer-descriptor range");
    AT_REQUIRE(ctx_pieces == nullptr || rows_pad >= (long long)B * T, "relpos_attention_x3: rows_pad too small");
    AT_REQUIRE(kv_pieces == nullptr || (scheme == XB_SCHEME_F16X2 && rows_pad >= (long long)B * T && (long long)T * heads * 64 * 2 < (1ll << 31)),
               "relpos_attention_x3: pre-split k / v need the fp16 scheme and rows_pad");
    if (scheme == XB_SCHEME_F16X2 && kv_pieces) {
        static const int w8_default = []() { const char* e = std::getenv("AUDIOTOKEN_ATTN_W8"); return (e && e[0] == '0') ? 0 : 1; }();
        // (the round-4 kernel builds its rel-pos table from PRE-SPLIT distance embeddings: with rel-pos it needs them)
        if ((w8 < 0 ? w8_default : w8) != 0 && (dist_emb == nullptr || dist_pieces != nullptr) && relpos_attention_w8_eligible(T, heads, rows_pad, B, dist_emb != nullptr))
            return launch_relpos_attention_w8(qkv, amask, dist_emb ? dist_pieces : nullptr, dist_scale, ctx, B, T, stream, heads, status, ctx_pieces, rows_pad, kv_pieces);
    }
    if (scheme == XB_SCHEME_F16X2 && kv_pieces) return launch_ax<SchemeF16x2, true>(qkv, amask, dist_emb, ctx, B, T, stream, heads, status, ctx_pieces, rows_pad, kv_pieces);
    if (scheme == XB_SCHEME_F16X2) return launch_ax<SchemeF16x2, false>(qkv, amask, dist_emb, ctx, B, T, stream, heads, status, ctx_pieces, rows_pad, nullptr);
    return launch_ax<SchemeBf16x3, false>(qkv, amask, dist_emb, ctx, B, T, stream, heads, status, ctx_pieces, rows_pad, nullptr);
}

}  // namespace at
