// Relative-position self-attention (reference audiotoken/modeling_wav2vec2_bert.py:46-73) on the two-piece fp16 scheme, round 4: the product's
// attention kernel for both semantic tokenizers (the older relpos_attention_x3_kernel of attention_bf16x3.hip stays as the bf16x3 / unsplit-input
// form and as this kernel's A/B twin, option "attn_w8" = 0).
//
// Same arithmetic contract as attention_bf16x3.hip's <SchemeF16x2, KVP> form — q, k, v times XB_F16_ACT_SCALE and the probabilities times 2^10 as
// hi + lo fp16 pieces, three products per multiply-add on v_mfma_f32_32x32x16_f16, S^T = K.Q^T with the keys on the MFMA rows so that a lane owns
// the scores of ONE query and P never leaves registers, exp2-domain online softmax, rel-pos bias from a q.E^T table with far-field constants —
// restructured around what bound that kernel (tools/ax_stamps.sh: vector-unit ISSUE, ~280 vector / LDS instructions per 32-key tile against 24 MFMAs):
//   * ONE workgroup per CU, 256 queries, EIGHT waves (two per SIMD) of 32 queries each (one 32-query MFMA column block per wave; W8_NQB = 1 below):
//     a K / V tile is staged once per 256 queries instead of once per 128 (half the staging instructions per query, half the L2 -> LDS bytes: the 12
//     query-tile workgroups of a (clip, head) become 6). (The 4-wave x 64-query form — one wave per SIMD, every K / V fragment feeding two query blocks —
//     was built and measured SLOWER, 2.43 vs 2.00 ms: 501 registers, the compiler parks q pieces / scores in the accumulator file;
//     profiles/r04_attention_w8_steps.txt step 4. It survives as W8_NQB = 2.)
//   * 64-key tiles: one running-maximum update, one cross-half exchange and ONE rescale of the 32 output accumulators per 64 keys instead of per 32;
//   * K / V tiles arrive by LDS-DMA (buffer_load_dwordx4 ... lds: no staging registers, no ds_write; a row beyond T is fetched as row T - 1 —
//     finite, and its scores get the -inf key bias), double-buffered, one barrier per tile. The DMA writes 1 KB runs linearly (8 rows x 128 B), so the bank-conflict
//     freedom of the fragment reads comes from the SOURCE side: lane l of a DMA instruction fetches 16-byte chunk (l & 7) ^ swz(row) of its row and
//     the fragment reads apply the same XOR (cdna_hip_programming.md rule 21: linear destination + swizzled source + swizzled read). K rows use
//     swz = (row >> 1) & 7 (ds_read_b128: the 16 lanes of a group hit 16 distinct 16-byte slots of a 256-byte bank window), V rows
//     swz = ((row >> 1) & 1) << 2 (ds_read_b64_tr_b16: the 4 rows of a transposing read fall on 4 disjoint 64-byte quarters);
//   * the softmax's vector work per score: the scale, the bias and the subtraction of the running maximum are ONE fma (the maximum is taken over the
//     raw accumulators first: the scale is positive and the far-field bias is constant per (query, tile)); the 2^10 of the probability pieces is folded
//     into that fma's addend (p' = exp2(x + 10), exact power of two; the row sum is kept in the same scale and cancels in the final 1 / l); the
//     cross-half exchanges use v_permlane32_swap instead of ds_bpermute;
//   * the key bias (0 / finfo.min for a padded key / -inf beyond T) of the whole clip is tabulated in LDS once per workgroup together with a per-tile
//     "any key masked" bit mask, so the loop carries no mask loads, no ballot and no LDS stores.
//   * A TWO-STAGE SOFTWARE PIPELINE INSIDE EVERY WAVE (two waves per SIMD). What the builds of this file measured (profiles/r04_attention_w8_steps.txt):
//     with a tile as three dependent phases — S (24 MFMAs), softmax + split (vector unit), P.V (24 MFMAs) — the kernel took MFMA time + vector time of
//     BOTH waves of a SIMD (6 700 cycles per 64-key tile for 2 x 1 536 of MFMA issue; matrix pipe busy 0.50), whether the two waves ran in lockstep or
//     one phase apart: what a wave's MFMAs leave free on its SIMD is only usable by vector instructions that sit BETWEEN those MFMAs in the SAME wave's
//     stream (MI355X_MICROARCH.md: an MFMA holds the vector issue port for 8 of its 32 cycles; <= 5 issues hide per gap). So each interval i runs two
//     phases whose MFMAs and vector work belong to DIFFERENT tiles and are independent:
//         phase A:  S(i + 1) = K(i + 1) . Q^T   (24 MFMAs)   beside   max / exp2 / row sum of tile i  (-> alpha(i))
//         phase B:  O += V(i)^T . P(i)^T        (24 MFMAs)   beside   P(i) -> hi / lo pieces, one MFMA k-step ahead
//     (T15 of cdna_hip_programming.md), cut into slices of one MFMA + <= 5 vector instructions whose order is pinned with sched_barrier(0): left to
//     itself the compiler emitted the MFMAs and the vector work as separate blocks. Two waves of one SIMD arbitrate by age, not fairly (stamps: 1 645 +
//     1 445 cycles for the older wave's two phases against 3 015 + 1 146 for its partner, then 1 600 at the barrier: 5 700-5 800 per interval) — the
//     interval is the SUM of both waves' MFMA and vector cycles, so what moved the kernel after that were instruction COUNTS: the running-max rescale
//     deferred until the maximum has grown by 2^4 (step 8) and the distance table Q.E^T built with fp16 MFMAs (step 9): 1.88 ms per launch.
//     K is fetched two tiles ahead, V one: two buffers each. Per element the operations and their order are those of the unpipelined form.
// Near-diagonal tiles (rel-pos buckets -64 .. +8 around the wave's queries) and tiles with masked keys take the general path (per-score bias gather).
// Also serves HuBERT (12 heads, no rel-pos bias).
#include "at_common.h"
#include "w2vbert_kernels.h"
#include "split_scheme.h"

#include <cstdio>
#include <type_traits>

namespace at {

typedef __attribute__((address_space(3))) void w8_lds_void;

// -DW8_DEBUG_STAMPS (tools/w8_stamps.sh; never in the product build): waves 0 and 4 of one workgroup sum the cycle counter over the parts of their intervals
#ifdef W8_DEBUG_STAMPS
__device__ unsigned long long w8_stamps[2][12];
#define W8_T(i) const unsigned long long w8_t##i = __builtin_readcyclecounter()
#define W8_ACC(k, a_, b_) w8_d[k] += w8_t##b_ - w8_t##a_
#else
#define W8_T(i) do {} while (0)
#define W8_ACC(k, a_, b_) do {} while (0)
#endif

constexpr int W8_QB = 256, W8_KB = 64;
constexpr int W8_PLANE_B = W8_KB * 64 * 2;      // bytes of one piece of one K or V tile: 64 keys x 64 d x fp16
constexpr int W8_TILE_B = 2 * W8_PLANE_B;       // hi | lo of one K or V tile
constexpr int W8_KBUFS = 2, W8_VBUFS = 2;       // K(i + 1) read / K(i + 2) in flight; V(i) read / V(i + 1) in flight
constexpr int W8_KV_B = (W8_KBUFS + W8_VBUFS) * W8_TILE_B;
constexpr int W8_QE_LD = 73;                    // the 73 buckets: odd stride, conflict-free row reads
constexpr int W8_LDS_MAX = 163840;              // the key-bias table of a clip must fit the LDS left beside the tiles and the rel-pos table
constexpr float W8_SCALE2 = 0.125f * 1.4426950408889634f;
constexpr float W8_P_LOG2 = 10.0f;              // probabilities are split as p * 2^10
// Deferred rescale (cdna_hip_programming.md T13): when no query of the wave raises its running maximum by more than this (log2 domain), the maximum is
// kept and O is not rescaled — the probabilities of the tile are then at most 2^(10 + 4) = 16384 (fp16 pieces hold 65504; the split keeps its 22 bits at any
// magnitude, the row sum is fp32), and the 32-register multiply of O — pure vector work, which does not overlap with MFMAs on this chip — is skipped for all
// but the first tiles of a query block. Exact algebra: the deferred factor cancels in O / l; only the rounding of exp2's argument moves.
constexpr float W8_DEFER = 4.0f;
// EXPERIMENT (round 6, -DW8_P_PIECES=1; the product build keeps 2): the probabilities as ONE fp16 piece in P.V — two products per multiply-add (v lo . p, v hi . p)
// instead of three, and one conversion per pair of probabilities instead of three instructions. p * 2^10 <= 1024 has no range problem; its precision drops from
// 22 to 11 bits (relative 2^-12 per probability). Measured and judged in profiles/EXPERIMENTS.md.
#ifndef W8_P_PIECES
#define W8_P_PIECES 2
#endif
constexpr int W8_PP = W8_P_PIECES;

static size_t w8_lds_bytes(int T, bool relpos) {
    const int nkt = (T + W8_KB - 1) / W8_KB;
    return (size_t)W8_KV_B + (relpos ? (size_t)W8_QB * W8_QE_LD * 4 : 0) + (size_t)nkt * W8_KB * 4 + 16;
}

// max / sum of a value with its partner lane (lane ^ 32) on the vector unit (v_permlane32_swap: vdst[32:63] <-> src0[0:31]; fed the same register
// twice, lane i < 32 ends up with {x[i], x[i + 32]} and lane i >= 32 with {x[i - 32], x[i]})
__device__ __forceinline__ float w8_pair_max(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float w8_pair_sum(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// two probabilities (times 2^10) -> packed fp16 hi pair and lo pair: hi = rne(p), lo = rne(p - hi) with the exact difference formed inside the fma
// (v_fma_mixlo / mixhi_f16: (-hi as f16) * 1.0 + p, rounded once to fp16 — the same values as cvt, subtract, cvt, in 3 instructions instead of 5)
__device__ __forceinline__ void w8_split_pair(float p0, float p1, unsigned& hi, unsigned& lo) {
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(p0), "v"(p1));
    asm("v_fma_mixlo_f16 %0, -%1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "v"(p0));
    asm("v_fma_mixhi_f16 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(p1));
}

// W8_NQB = 32-query MFMA column blocks per wave: 1 -> 8 waves (two per SIMD: the product), 2 -> 4 waves (one per SIMD, K / V fragments shared by both
// blocks; measured slower, profiles/r04_attention_w8_steps.txt). A file-scope constant, not a template parameter: with the block count as a template
// argument hipcc (ROCm 7.2) silently dropped the kernel's HOST-side stub (the device code was complete, the library failed to load: undefined symbol).
constexpr int W8_NQB = 1;
template <bool RELPOS>
__global__ __launch_bounds__(512 / W8_NQB, W8_NQB == 1 ? 2 : 1) void relpos_attention_w8_kernel(const float* __restrict__ qkv, const float* __restrict__ amask,
                                                                     const _Float16* __restrict__ dist_pieces, float dist_inv_scale, float* __restrict__ ctx, int T, int hid,
                                                                     int* __restrict__ status, _Float16* __restrict__ ctx_pieces, long long rows_pad,
                                                                     int nheads, int nclips, const _Float16* __restrict__ kv_pieces) {
    constexpr int NQB = W8_NQB;
    typedef SchemeF16x2 SC;
    typedef _Float16 PT;
    typedef f16x8 V8;
    typedef f16x4 V4;
    constexpr float XS = XB_F16_ACT_SCALE;
    constexpr float S_SCALE2 = W8_SCALE2 / (XS * XS);
    RangeMax over;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int nkt = (T + W8_KB - 1) / W8_KB;
    unsigned char* KV = smem_raw;                                             // K tiles [2][hi | lo][64 rows][128 B], then V tiles [2][hi | lo][64][128 B]
    float* QE = reinterpret_cast<float*>(KV + W8_KV_B);                       // [256 queries][73]: log2(e)/8 * q . E[bucket] (absent without rel-pos)
    float* kbt = QE + (RELPOS ? W8_QB * W8_QE_LD : 0);                        // [nkt * 64] additive key bias
    int* tmask = reinterpret_cast<int*>(kbt + nkt * W8_KB);                   // [2] bit kt: tile kt holds a key whose bias is not 0
    const int tid = threadIdx.x, lane = tid & 63;
    constexpr int NW = 8 / NQB, QW = 32 * NQB, NT = 64 * NW;                  // waves, queries per wave, threads
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);                // queries l0 + QW wave .. (block qb: + 32 qb)
    const int l32 = lane & 31, hh = lane >> 5;
    // 1-D grid, XCD-aware (as attention_bf16x3.hip): XCD x takes a contiguous range of (clip, head) pairs and runs the query tiles of one pair back to
    // back, so its K / V rows are fetched into ONE L2
    const int nqt = (T + W8_QB - 1) / W8_QB;
    const int nblk = gridDim.x, per_xcd = (nblk + 7) >> 3;
    const int lid = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (lid >= nqt * nheads * nclips) return;
    const int qt = lid % nqt, pair = lid / nqt;
    const int h = pair % nheads, b = pair / nheads;
    const int l0 = qt * W8_QB;
    const long long rowbase = (long long)b * T;
    const int LD = 3 * hid;
    const float* qp = qkv + h * 64;

    // ---- LDS-DMA of the first tiles: they fly during the table set-up below ----------------------------------------------------------------------
    // ONE buffer descriptor over the four piece planes (k hi, k lo, v hi, v lo: [4][rows_pad][hid] fp16, < 4 GB), built from kernel arguments only so
    // that it stays in scalar registers (four per-clip descriptors ended up in vector registers and every DMA instruction inside a waterfall loop).
    // It cannot clip rows beyond this clip's T to zero, and those rows may be anything (the next clip; never-written padding rows of the workspace: a NaN
    // there would survive its exactly-zero probability), so a lane whose row is beyond T fetches row T - 1 instead: finite, and masked by the -inf key bias.
    const unsigned plane_b = (unsigned)(rows_pad * (long long)hid * 2);          // bytes of one piece plane
    const __amdgpu_buffer_rsrc_t kvrs = __builtin_amdgcn_make_buffer_rsrc((void*)kv_pieces, 0, (int)(4u * plane_b), 0x00020000);
    // wave w moves rows RW w .. RW w + RW - 1 (RW = 8 NQB) of each plane of a tile, 8 rows = 1 KB per instruction; lane l: row RW w + 8 j + (l >> 3),
    // LDS chunk l & 7; the source chunk is swizzled with the DESTINATION row
    constexpr int RW = 8 * NQB;
    const int d_row = wave * RW + (lane >> 3), d_cs = lane & 7;
    const unsigned clip_b = (unsigned)(rowbase * hid * 2 + h * 128);
    const int row_b = hid * 2;
    auto issue_k = [&](int kt, int buf) {
#pragma unroll
        for (int j = 0; j < NQB; ++j) {
            const int row = min(kt * W8_KB + d_row + 8 * j, T - 1);
            const unsigned vo = clip_b + (unsigned)(row * row_b) + ((d_cs ^ (((d_row + 8 * j) >> 1) & 7)) << 4);
#pragma unroll
            for (int p = 0; p < 2; ++p)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(kvrs, (w8_lds_void*)(KV + buf * W8_TILE_B + wave * (RW * 128) + p * W8_PLANE_B + j * 1024), 16, vo, p * plane_b, 0, 0);
        }
    };
    auto issue_v = [&](int kt, int buf) {
#pragma unroll
        for (int j = 0; j < NQB; ++j) {
            const int row = min(kt * W8_KB + d_row + 8 * j, T - 1);
            const unsigned vo = clip_b + (unsigned)(row * row_b) + ((d_cs ^ ((((d_row + 8 * j) >> 1) & 1) << 2)) << 4);
#pragma unroll
            for (int p = 0; p < 2; ++p)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(kvrs, (w8_lds_void*)(KV + (W8_KBUFS + buf) * W8_TILE_B + wave * (RW * 128) + p * W8_PLANE_B + j * 1024), 16, vo,
                                                         (2u + p) * plane_b, 0, 0);
        }
    };
    issue_k(0, 0);
    issue_v(0, 0);
    if (nkt > 1) issue_k(1, 1);

    // ---- key bias of the whole clip + per-tile "any key masked" bits ---------------------------------------------------------------------------
    const float FMIN = -3.4028234663852886e38f;
    if (tid < 2) tmask[tid] = 0;
    __syncthreads();
    for (int k = tid; k < nkt * W8_KB; k += NT) {
        float v = -INFINITY;
        if (k < T) v = amask[rowbase + k] != 0.f ? 0.f : FMIN;
        kbt[k] = v;
        if (v != 0.f) atomicOr(&tmask[k >> 11], 1 << ((k >> 6) & 31));
    }
    // ---- this lane's queries (column l32 of the wave's two blocks): two fp16 pieces of q[lq][ds * 16 + 8 hh .. + 7] * XS ---------------------------
    const int lq0 = l0 + wave * QW + l32;            // block qb: lq0 + 32 qb
    V8 qpc[NQB][2][4];
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
        const int lq = lq0 + 32 * qb;
        const int lc = lq < T ? lq : T - 1;
#pragma unroll
        for (int ds = 0; ds < 4; ++ds) {
            const f4 a = *reinterpret_cast<const f4*>(qp + (rowbase + lc) * LD + ds * 16 + 8 * hh);
            const f4 c = *reinterpret_cast<const f4*>(qp + (rowbase + lc) * LD + ds * 16 + 8 * hh + 4);
            V4 pa[2], pc[2];
            over |= split4<SC>(a, XS, pa);
            over |= split4<SC>(c, XS, pc);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) { qpc[qb][i][ds][k] = pa[i][k]; qpc[qb][i][ds][4 + k] = pc[i][k]; }
        }
    }
    // ---- rel-pos table QE[query][bucket] = log2(e)/8 * q . E[bucket] on the SAME fp16 MFMAs as the scores (round 4): E^T . Q^T is S^T with the 73 distance
    // embeddings in the place of the keys — three 32-bucket row tiles x 4 k-steps x 3 products = 36 MFMAs per query block against 160 fp32 MFMAs of 32
    // cycles each before (10 k cycles per SIMD and workgroup: 7 % of the kernel). The embeddings arrive pre-split (launch_dist_split: [piece][96][64] fp16
    // times a power of two, rows >= 73 zero) and are read as fragments straight from L2; q pieces are the ones the scores use.
    if constexpr (RELPOS) {
        const float qe_scale = W8_SCALE2 * dist_inv_scale / XS;
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) {
            float* row = QE + (wave * QW + qb * 32 + l32) * W8_QE_LD;
#pragma unroll
            for (int rt = 0; rt < 3; ++rt) {
                f16v acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                for (int ds = 0; ds < 4; ++ds) {
                    V8 ef[2];
#pragma unroll
                    for (int p = 0; p < 2; ++p) ef[p] = *reinterpret_cast<const V8*>(dist_pieces + (p * 96 + rt * 32 + l32) * 64 + ds * 16 + 8 * hh);
#pragma unroll
                    for (int t = 0; t < SC::NPROD; ++t) acc = SC::mfma(ef[SC::prod_a(t)], qpc[qb][SC::prod_w(t)][ds], acc);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {   // lane holds buckets 32 rt + 8 (r / 4) + 4 hh + r % 4 of query l32
                    const int bucket = rt * 32 + 8 * (r >> 2) + 4 * hh + (r & 3);
                    if (bucket < W8_QE_LD) row[bucket] = qe_scale * acc[r];
                }
            }
        }
    }
    f16v oacc[NQB][2];
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[qb][dt][r] = 0.f;
    float mrun[NQB], lrun[NQB], alpha[NQB];   // alpha: the rescale of O that the statistics of the current tile ask for
    bool rescale[NQB];                        // wave-uniform: whether they ask for one at all (deferred rescale, W8_DEFER)
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) { mrun[qb] = -INFINITY; lrun[qb] = 0.f; alpha[qb] = 0.f; }
    const int wl_min = l0 + wave * QW, wl_max = wl_min + QW - 1;
    // fragment offsets of this lane inside a tile (bytes)
    int koff[4];   // K: row l32 (+ 32 for the second half), chunk (2 ds + hh) ^ swzK(row)
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) koff[ds] = l32 * 128 + (((2 * ds + hh) ^ ((l32 >> 1) & 7)) << 4);
    int voff[2];   // V (transposing read): lane 4 q + p of a 16-lane group addresses row 4 hh + q, d columns dt * 32 + (lane & 16) + 4 p .. + 3
    {
        const int li = lane & 15, vq = li >> 2, vp = li & 3;
        const int c0 = ((lane & 16) >> 3) + (vp >> 1), sv = (vq >> 1) & 1;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) voff[dt] = (4 * hh + vq) * 128 + ((((dt ^ sv) << 2) + c0) << 4) + (vp & 1) * 8;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // K(0), V(0), K(1)
    __syncthreads();                                   // ... of every wave; QE, kbt, tmask complete
    const int tm0 = __builtin_amdgcn_readfirstlane(tmask[0]), tm1 = __builtin_amdgcn_readfirstlane(tmask[1]);
    typedef const __attribute__((address_space(3))) float lds_cf;
    typedef const __attribute__((address_space(3))) f4 lds_cf4;
    lds_cf* qe0 = (lds_cf*)(QE + (wave * QW + l32) * W8_QE_LD);              // block qb: + 32 qb rows (typed LDS pointers: through the lambdas generic ones became flat loads)
    lds_cf* kbt3 = (lds_cf*)kbt;
    float c_left[NQB], c_right[NQB];
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) { c_left[qb] = 0.f; c_right[qb] = 0.f; }
    if constexpr (RELPOS) {
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) { c_left[qb] = qe0[qb * 32 * W8_QE_LD]; c_right[qb] = qe0[qb * 32 * W8_QE_LD + 72]; }
    }

    // ---- the pieces of a tile ----------------------------------------------------------------------------------------------------------------------
    // S^T = K . Q^T of the tile in K buffer kbuf: lane holds s[qb][kh][r] = q_(lq0 + 32 qb) . k_(r0 + 32 kh + 8 (r / 4) + 4 hh + r % 4)
    auto s_tile = [&](f16v (&s)[NQB][2], int kbuf) {
        const unsigned char* Kb = KV + kbuf * W8_TILE_B;
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[qb][kh][r] = 0.f;
#pragma unroll
            for (int ds = 0; ds < 4; ++ds) {
                V8 kf[2];
#pragma unroll
                for (int p = 0; p < 2; ++p) kf[p] = *reinterpret_cast<const V8*>(Kb + p * W8_PLANE_B + kh * 32 * 128 + koff[ds]);
#pragma unroll
                for (int t = 0; t < SC::NPROD; ++t)
#pragma unroll
                    for (int qb = 0; qb < NQB; ++qb) s[qb][kh] = SC::mfma(kf[SC::prod_a(t)], qpc[qb][SC::prod_w(t)][ds], s[qb][kh]);
            }
        }
    };
    // tile kt for this wave's 64 queries: far field on one side (constant bias per query) and no masked key -> the fast softmax
    auto tile_fast = [&](int kt, bool& far_left) -> bool {
        const int r0 = kt * W8_KB;
        far_left = (r0 + W8_KB - 1) - wl_min <= -64;
        const bool far_right = r0 - wl_max >= 8;
        const bool masked = (((kt < 32 ? tm0 : tm1) >> (kt & 31)) & 1) != 0;
        return (!RELPOS || far_left || far_right) && !masked;
    };
    // the general form for one query block: per-score bias = (far-field constant | rel-pos bucket of key - query, gathered from the table) + the key's
    // mask bias; scores -> p' = p * 2^10 in place; running maximum / row sum updated; alpha = the rescale of O
    auto bias_pass = [&](f16v (&s)[2], int r0, int lq, lds_cf* qe, float c_far, auto near_) -> float {
        constexpr bool NEAR = decltype(near_)::value;
        float mx = -INFINITY;
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int key0 = kh * 32 + 8 * g + 4 * hh;                                  // keys key0 .. key0 + 3 = registers 4 g .. 4 g + 3
                const f4 kb4 = *(lds_cf4*)(kbt3 + r0 + key0);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float bias = c_far;
                    if constexpr (NEAR) {
                        int dd = (r0 + key0 + e) - lq;
                        dd = dd < -64 ? -64 : (dd > 8 ? 8 : dd);
                        bias = qe[dd + 64];
                    }
                    const float sc = fmaf(S_SCALE2, s[kh][4 * g + e], bias + kb4[e]);
                    s[kh][4 * g + e] = sc;
                    mx = fmaxf(mx, sc);
                }
            }
        return mx;
    };
    auto softmax_general = [&](f16v (&s)[2], int kt, int qb) {
        const int r0 = kt * W8_KB;
        const int bmin = wl_min + 32 * qb, bmax = bmin + 31;                                 // this block's queries
        const bool far_left = (r0 + W8_KB - 1) - bmin <= -64;
        const bool far_right = r0 - bmax >= 8;
        float mx;
        if (!RELPOS || far_left || far_right) mx = bias_pass(s, r0, 0, qe0, far_left ? c_left[qb] : c_right[qb], std::false_type{});
        else mx = bias_pass(s, r0, lq0 + 32 * qb, qe0 + qb * 32 * W8_QE_LD, 0.f, std::true_type{});
        mx = w8_pair_max(mx);
        const bool resc = __builtin_amdgcn_ballot_w64(mx > mrun[qb] + W8_DEFER) != 0ull;   // wave-uniform (mrun = -inf on the first tile: always)
        rescale[qb] = resc;
        const float mnew = resc ? fmaxf(mrun[qb], mx) : mrun[qb];
        const float sub = mnew - W8_P_LOG2;
        float rs = 0.f;
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(s[kh][r] - sub);
                s[kh][r] = p;
                rs += p;
            }
        rs = w8_pair_sum(rs);
        alpha[qb] = resc ? __builtin_amdgcn_exp2f(mrun[qb] - mnew) : 1.0f;
        lrun[qb] = lrun[qb] * alpha[qb] + rs;
        mrun[qb] = mnew;
    };
    // the fast form of the statistics, cut into 22 slices per query block (0-3 raw maximum of 8 scores each; 4 the new maximum, exponent offset and alpha;
    // 5-20 two probabilities each; 21 the row sum) so that phase A can put one slice behind each MFMA
    float f_mraw[NQB], f_mnew[NQB], f_bias2[NQB], f_rs[NQB], f_cfar[NQB];
    auto stat_slice = [&](f16v (&sa)[2], int qb, int k) {
        if (k < 4) {
            if (k == 0) { f_mraw[qb] = -INFINITY; f_rs[qb] = 0.f; }
#pragma unroll
            for (int e = 0; e < 8; ++e) f_mraw[qb] = fmaxf(f_mraw[qb], sa[k >> 1][8 * (k & 1) + e]);
        } else if (k == 4) {
            const float mx = w8_pair_max(fmaf(S_SCALE2, f_mraw[qb], f_cfar[qb]));
            const bool resc = __builtin_amdgcn_ballot_w64(mx > mrun[qb] + W8_DEFER) != 0ull;   // wave-uniform (mrun = -inf on the first tile: always)
            rescale[qb] = resc;
            f_mnew[qb] = resc ? fmaxf(mrun[qb], mx) : mrun[qb];
            f_bias2[qb] = (f_cfar[qb] - f_mnew[qb]) + W8_P_LOG2;
            alpha[qb] = resc ? __builtin_amdgcn_exp2f(mrun[qb] - f_mnew[qb]) : 1.0f;   // exp2(0) = 1 exactly for a lane whose maximum did not move, exp2(-inf) = 0 on the first tile
        } else if (k < 21) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int r = 2 * (k - 5) + e;
                const float p = __builtin_amdgcn_exp2f(fmaf(S_SCALE2, sa[r >> 4][r & 15], f_bias2[qb]));
                sa[r >> 4][r & 15] = p;
                f_rs[qb] += p;
            }
        } else if (k == 21) {
            const float rs = w8_pair_sum(f_rs[qb]);
            lrun[qb] = lrun[qb] * alpha[qb] + rs;
            mrun[qb] = f_mnew[qb];
        }
    };
    // ---- phase A, fast form: the 48 MFMAs of S(i + 1) (both query blocks) with the statistics of tile i behind them, slice by slice (block 0's 22
    // slices, then block 1's); K fragments are read two MFMA groups ahead
    auto s_tile_softmax = [&](f16v (&sb)[NQB][2], int kbuf, f16v (&sa)[NQB][2]) {
        const unsigned char* Kb = KV + kbuf * W8_TILE_B;
        V8 kf[3][2];   // fragments of MFMA group g = (kh = g >> 2, ds = g & 3) in kf[g % 3]
        auto load_k = [&](int g) {
#pragma unroll
            for (int p = 0; p < 2; ++p) kf[g % 3][p] = *reinterpret_cast<const V8*>(Kb + p * W8_PLANE_B + (g >> 2) * 32 * 128 + koff[g & 3]);
        };
        load_k(0);
        load_k(1);
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                for (int r = 0; r < 16; ++r) sb[qb][kh][r] = 0.f;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if (g + 2 < 8) load_k(g + 2);
#pragma unroll
            for (int t = 0; t < SC::NPROD; ++t)
#pragma unroll
                for (int qb = 0; qb < NQB; ++qb) {
                    sb[qb][g >> 2] = SC::mfma(kf[g % 3][SC::prod_a(t)], qpc[qb][SC::prod_w(t)][g & 3], sb[qb][g >> 2]);
                    const int sl = (3 * g + t) * NQB + qb;                  // 0 .. 24 NQB - 1
                    if (sl < 22) stat_slice(sa[0], 0, sl);
                    else if (NQB > 1 && sl < 44) stat_slice(sa[NQB - 1], NQB - 1, sl - 22);
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
    };
    // ---- phase B: the 48 MFMAs of O^T += V(i)^T . P(i)^T with the split of the probabilities into hi / lo fp16 pieces running ONE k-step ahead of the
    // MFMAs that consume them (B operand of k-step ks = registers 8 (ks & 1) .. + 7 of half ks >> 1 = keys 16 ks + {0..3, 8..11} + 4 hh;
    // p * 2^10 <= 1024: always fits). One pair of probabilities behind each of the first eight MFMAs of a k-step.
    auto pv_split = [&](const f16v (&s)[NQB][2], int vbuf) {
        const unsigned char* Vb = KV + (W8_KBUFS + vbuf) * W8_TILE_B;
        typedef short s4_ __attribute__((__vector_size__(4 * sizeof(short))));
        typedef unsigned u4_ __attribute__((ext_vector_type(4)));
        u4_ pp[2][NQB][2];   // [k-step parity][query block][piece]: 8 fp16 = 4 registers
        auto split_pair = [&](int ks, int qb, int j) {
            unsigned hi, lo;
            if constexpr (W8_PP == 2) {
                w8_split_pair(s[qb][ks >> 1][8 * (ks & 1) + j], s[qb][ks >> 1][8 * (ks & 1) + j + 1], hi, lo);
                pp[ks & 1][qb][1][j >> 1] = lo;
            } else {
                asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(s[qb][ks >> 1][8 * (ks & 1) + j]), "v"(s[qb][ks >> 1][8 * (ks & 1) + j + 1]));
            }
            pp[ks & 1][qb][0][j >> 1] = hi;
        };
        V8 vf[2][2];   // [(ks, dt) parity][piece]
        auto load_v = [&](int u) {   // u = 2 ks + dt
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const unsigned char* vb = Vb + p * W8_PLANE_B + (u >> 1) * 16 * 128 + voff[u & 1];
                const s4_ lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4_*)(vb));
                const s4_ hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4_*)(vb + 8 * 128));
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    vf[u & 1][p][k] = __builtin_bit_cast(PT, (short)lo[k]);
                    vf[u & 1][p][4 + k] = __builtin_bit_cast(PT, (short)hi[k]);
                }
            }
        };
        load_v(0);
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
            for (int j = 0; j < 8; j += 2) split_pair(0, qb, j);
        asm volatile("s_nop 1");     // (pieces written by inline asm -> MFMA operand: the compiler pads nothing behind an asm statement)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int ks = u >> 1, dt = u & 1;
            if (u + 1 < 8) load_v(u + 1);
            constexpr int T0 = W8_PP == 2 ? 0 : 1;                          // one probability piece: the products (v lo . p), (v hi . p) only
#pragma unroll
            for (int t = T0; t < SC::NPROD; ++t)
#pragma unroll
                for (int qb = 0; qb < NQB; ++qb) {
                    oacc[qb][dt] = SC::mfma(vf[u & 1][SC::prod_a(t)], __builtin_bit_cast(V8, pp[ks & 1][qb][SC::prod_w(t)]), oacc[qb][dt]);
                    const int sl = (dt * (SC::NPROD - T0) + (t - T0)) * NQB + qb;   // 0 .. 6 NQB - 1 (4 NQB - 1) within the k-step
                    if (ks + 1 < 4 && sl < 4 * NQB) split_pair(ks + 1, sl >> 2, 2 * (sl & 3));
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
    };
#ifdef W8_DEBUG_STAMPS
    unsigned long long w8_d[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long w8_begin = __builtin_readcyclecounter();
#endif
    // interval i: `sa` holds the raw scores of tile i, `sb` receives those of tile i + 1
    auto interval = [&](int i, f16v (&sa)[NQB][2], f16v (&sb)[NQB][2]) {
        W8_T(0);
        // Landed behind this wait + barrier: K(i + 1) and V(i) (issued one interval ago). Free behind it: K buffer i & 1 (tile i, last read by S(i) in
        // the interval that just ended) and V buffer (i + 1) & 1 (tile i - 1, last read by P.V(i - 1) there) -> refill them with K(i + 2) and V(i + 1)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        W8_T(1);
        __builtin_amdgcn_s_barrier();
        W8_T(2);
        const bool more = i + 1 < nkt;
        if (i + 2 < nkt) issue_k(i + 2, i & 1);
        if (more) issue_v(i + 1, (i + 1) & 1);
        bool far_left;
        const bool fast = tile_fast(i, far_left);
        W8_T(3);
        // phase A: the MFMAs of S(i + 1) beside the softmax statistics of tile i
        if (more && fast) {
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb) f_cfar[qb] = far_left ? c_left[qb] : c_right[qb];
            s_tile_softmax(sb, (i + 1) & 1, sa);
        } else {
            if (more) s_tile(sb, (i + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);   // (not interleaved: the general form's bias gathers need the registers)
            if (fast) {
#pragma unroll
                for (int qb = 0; qb < NQB; ++qb) {
                    f_cfar[qb] = far_left ? c_left[qb] : c_right[qb];
#pragma unroll
                    for (int k = 0; k < 22; ++k) stat_slice(sa[qb], qb, k);
                }
            } else {
#pragma unroll
                for (int qb = 0; qb < NQB; ++qb) softmax_general(sa[qb], i, qb);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        W8_T(4);
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb)
            if (rescale[qb]) {      // (rare after the first tiles: W8_DEFER)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[qb][dt][r] *= alpha[qb];
            }
        __builtin_amdgcn_sched_barrier(0);
        W8_T(5);
        // phase B: the MFMAs of P.V(i) beside the split of tile i's probabilities
        pv_split(sa, i & 1);
        W8_T(6);
#ifdef W8_DEBUG_STAMPS
        W8_ACC(0, 0, 1); W8_ACC(1, 1, 2); W8_ACC(2, 2, 3); W8_ACC(fast && more ? 3 : 4, 3, 4); W8_ACC(5, 4, 5); W8_ACC(6, 5, 6);
        w8_d[7] += (fast && more) ? 1 : 0;
#endif
    };
    f16v sA[NQB][2], sB[NQB][2];
    s_tile(sA, 0);
    for (int i = 0; i < nkt; i += 2) {
        interval(i, sA, sB);
        if (i + 1 < nkt) interval(i + 1, sB, sA);
    }
#ifdef W8_DEBUG_STAMPS
    if (blockIdx.x == (gridDim.x / 2) + 8 && (threadIdx.x == 0 || threadIdx.x == NT / 2)) {
        const int g = threadIdx.x ? 1 : 0;
        for (int k = 0; k < 8; ++k) w8_stamps[g][k] = w8_d[k];
        w8_stamps[g][8] = __builtin_readcyclecounter() - w8_begin;
        w8_stamps[g][9] = (unsigned long long)nkt;
    }
#endif
    // lane holds O[lq][dv = 32 dt + 8 (r / 4) + 4 hh + r % 4] * (XS * l'), l' = l * 2^10 (the probability scale cancels)
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
        const int lq = lq0 + 32 * qb;
        if (lq < T) {
            const float inv = (1.0f / lrun[qb]) * (1.0f / XS);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f4 v = {oacc[qb][dt][4 * g] * inv, oacc[qb][dt][4 * g + 1] * inv, oacc[qb][dt][4 * g + 2] * inv, oacc[qb][dt][4 * g + 3] * inv};
                    if (ctx_pieces)   // straight to the output projection's GEMM operand (split_scheme.h): an 8-byte store per piece
                        over |= store_pieces4<SC>(ctx_pieces, rows_pad * hid, rows_pad, rowbase + lq, h * 64 + dt * 32 + 8 * g + 4 * hh, v, XS);
                    else
                        *reinterpret_cast<f4*>(ctx + (rowbase + lq) * hid + h * 64 + dt * 32 + 8 * g + 4 * hh) = v;
                }
        }
    }
    range_publish(status, status ? status + 1 : nullptr, over);
}

bool relpos_attention_w8_eligible(int T, int heads, long long rows_pad, long long B, bool relpos) {
    return T >= 1 && T <= 4096 /* 64 tile bits */ && rows_pad * heads * 64 * 8 < (1ll << 32) /* one descriptor over the four planes */ && w8_lds_bytes(T, relpos) <= (size_t)W8_LDS_MAX && rows_pad >= B * T && (long long)T * heads * 64 * 2 < (1ll << 31);
}

int launch_relpos_attention_w8(const float* qkv, const float* amask, const __bf16* dist_pieces, float dist_scale, float* ctx, int B, int T, hipStream_t stream,
                               int heads, int* status, __bf16* ctx_pieces, long long rows_pad, const __bf16* kv_pieces) {
    AT_REQUIRE(kv_pieces != nullptr && relpos_attention_w8_eligible(T, heads, rows_pad, B, dist_pieces != nullptr),
               "relpos_attention_w8: needs pre-split k / v and a clip whose key-bias table fits LDS (T <= 1728 with rel-pos)");
    const long long nblk = (long long)((T + W8_QB - 1) / W8_QB) * heads * B;
    dim3 grid((unsigned)((nblk + 7) / 8 * 8));
    const size_t lds = w8_lds_bytes(T, dist_pieces != nullptr);
    // (the attribute is the maximum over every T this process will use: set it to the kernel's ceiling once per device)
    const size_t lds_max = W8_LDS_MAX;
    int rc;
    const int hid = heads * 64;
    _Float16* cp = reinterpret_cast<_Float16*>(ctx_pieces);
    const _Float16* kp = reinterpret_cast<const _Float16*>(kv_pieces);
    const _Float16* dp = reinterpret_cast<const _Float16*>(dist_pieces);
    const float dinv = dist_pieces ? 1.0f / dist_scale : 1.0f;
    if (dist_pieces) {
        static LdsAttrFlags lds_attr;
        rc = set_max_dynamic_lds(lds_attr, relpos_attention_w8_kernel<true>, lds_max);
        if (!rc) hipLaunchKernelGGL((relpos_attention_w8_kernel<true>), grid, dim3(512 / W8_NQB), lds, stream, qkv, amask, dp, dinv, ctx, T, hid, status, cp, rows_pad, heads, B, kp);
    } else {
        static LdsAttrFlags lds_attr;
        rc = set_max_dynamic_lds(lds_attr, relpos_attention_w8_kernel<false>, lds_max);
        if (!rc) hipLaunchKernelGGL((relpos_attention_w8_kernel<false>), grid, dim3(512 / W8_NQB), lds, stream, qkv, amask, dp, dinv, ctx, T, hid, status, cp, rows_pad, heads, B, kp);
    }
    if (rc) return rc;
    AT_CHECK_HIP(hipGetLastError());
#ifdef W8_DEBUG_STAMPS
    {
        static int printed = 0;
        if (printed < 3) {
            ++printed;
            (void)hipStreamSynchronize(stream);
            unsigned long long hb[2][12];
            (void)hipMemcpyFromSymbol(hb, HIP_SYMBOL(w8_stamps), sizeof(hb));
            for (int g = 0; g < 2; ++g) {
                const double n = (double)(hb[g][9] ? hb[g][9] : 1), nf = (double)(hb[g][7] ? hb[g][7] : 1), ng = n - hb[g][7] > 0 ? n - hb[g][7] : 1;
                std::fprintf(stderr, "w8 stamps B %d T %d heads %d wave %d (cycles per 64-key interval, 64 queries): dma wait %.0f  barrier %.0f  issue+classify %.0f  phase A fast %.0f (x%llu)  phase A general/last %.0f (x%.0f)  rescale %.0f  phase B %.0f | loop total %.0f per interval, %llu intervals\n",
                             B, T, heads, g * 4, hb[g][0] / n, hb[g][1] / n, hb[g][2] / n, hb[g][3] / nf, hb[g][7], hb[g][4] / ng, ng, hb[g][5] / n, hb[g][6] / n, hb[g][8] / n, hb[g][9]);
            }
        }
    }
#endif
    return 0;
}

// distance embeddings fp32 [80][64] (73 rows used) * scale -> fp16 pieces [2][96][64], rows >= 73 zero (the A operand of the rel-pos table's MFMAs)
__global__ __launch_bounds__(256) void dist_split_kernel(const float* __restrict__ e, _Float16* __restrict__ out, float scale) {
    const int q = blockIdx.x * 256 + threadIdx.x;      // one quad of one of 96 rows
    if (q >= 96 * 16) return;
    const int row = q >> 4, col = (q & 15) * 4;
    f4 v = {0.f, 0.f, 0.f, 0.f};
    if (row < 73) v = *reinterpret_cast<const f4*>(e + row * 64 + col);
    SchemeNoCheck<SchemeF16x2>::V4 p[2];
    split4<SchemeNoCheck<SchemeF16x2>>(v, scale, p);
#pragma unroll
    for (int i = 0; i < 2; ++i) *reinterpret_cast<f16x4*>(out + (i * 96 + row) * 64 + col) = p[i];
}
int launch_dist_split(const float* dist_emb, __bf16* out, float scale, hipStream_t stream) {
    hipLaunchKernelGGL(dist_split_kernel, dim3(6), dim3(256), 0, stream, dist_emb, reinterpret_cast<_Float16*>(out), scale);
    AT_CHECK_HIP(hipGetLastError());
    return 0;
}

// fp32 qkv rows [rows][3 hid] -> the row-major k / v pieces [which][piece][rows_pad][hid] * XB_F16_ACT_SCALE the fused projection's epilogue writes
// (XB_EPI_QKV, split_epilogue.h): the operator-level entry point of this kernel (at_op_relpos_attention_kvp) starts from fp32 rows like its twin
__global__ __launch_bounds__(256) void kv_rowmajor_split_kernel(const float* __restrict__ qkv, _Float16* __restrict__ out, long long rows, long long rows_pad,
                                                                int hid, int* __restrict__ status) {
    typedef SchemeF16x2 SC;
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;        // one quad of one row of k or v
    const int qpr = hid / 4;
    RangeMax over;
    if (q < rows * 2 * qpr) {
        const long long row = q / (2 * qpr);
        const int rem = (int)(q - row * 2 * qpr), which = rem / qpr, col = (rem - which * qpr) * 4;
        const f4 v = *reinterpret_cast<const f4*>(qkv + row * 3 * hid + (1 + which) * hid + col);
        SC::V4 p[2];
        over |= split4<SC>(v, XB_F16_ACT_SCALE, p);
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<SC::V4*>(out + ((long long)(which * 2 + i) * rows_pad + row) * hid + col) = p[i];
    }
    range_publish(status, status ? status + 1 : nullptr, over);
}

int launch_kv_rowmajor_split(const float* qkv, __bf16* out, long long rows, long long rows_pad, int hid, int* status, hipStream_t stream) {
    AT_REQUIRE(hid % 4 == 0 && rows_pad >= rows, "kv_rowmajor_split: bad shape");
    const long long quads = rows * 2 * (hid / 4);
    hipLaunchKernelGGL(kv_rowmajor_split_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, stream, qkv, reinterpret_cast<_Float16*>(out), rows, rows_pad, hid,
                       status);
    AT_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace at
