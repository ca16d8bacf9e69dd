// SEANet residual block at 128 channels, two-piece fp16 scheme, ROLE-SPLIT waves (round 2) — the block of seanet_res128x3.hip
//   out = ELU( [W1 | Wsc] . [ELU(W3 * ELU(x) + b3) | x] + (b1 + bsc) )          (k3 128->64, k1 64->128, k1 shortcut)
// with the same arithmetic per element (split_scheme.h, SchemeF16x2: operands as two fp16 pieces, three MFMA products, power-of-two scales,
// range verdict into the status word), restructured so that a SIMD always has two waves with different work:
//   * 8 waves; waves 0-3 ("C") keep W3 rows 16w .. 16w+15 (K = 384) in 96 registers and run conv3, waves 4-7 ("T") keep [W1 | Wsc] rows
//     32w .. 32w+31 (K = 192) in 96 registers and run the tail. In seanet_res128x3.hip one wave per SIMD holds both (192 registers) and does
//     staging, conv3, the h epilogue, the tail and the output epilogue one after the other: 23 % matrix-pipe busy, 40 % vector-unit active,
//     the rest waiting (PMC, profiles/r02_final_acoustic_pmc_derived.csv).
//   * software pipeline over the workgroup's tiles, ONE barrier per tile: in iteration k the C waves run conv3 of tile k (x tile k -> h
//     buffer k & 1) while the T waves run the tail of tile k-1 (h buffer (k-1) & 1 and the raw x of tile k-1 -> out); every wave then stages
//     its share of tile k+1 (loads issued at the top of the iteration: ELU, two splits, LDS stores). Three x buffers make that hazard-free:
//     tile k+1 is written to buffer (k+1) % 3 while buffers k % 3 and (k-1) % 3 are read.
//   * 32-row tiles: 3 x [Xe hi | Xe lo | Xr hi | Xr lo][34 rows][128 + 16] fp16 + 2 x [2][32][64] (swizzled) = 131 KB of LDS, one workgroup per CU.
// Results are bit-identical to seanet_res128x3_kernel<SchemeF16x2> (same products in the same order per output element); that kernel stays
// as the bf16-scheme instantiation and as this one's reference (tests/test_acoustic_gpu.py).
#include "gemm_core.h"
#include "encodec_kernels.h"
#include "split_scheme.h"

namespace at {

namespace {
constexpr int RS_TT = 32;                       // time rows per tile
constexpr int RS_XROWS = RS_TT + 2;             // row i <-> time t0 - 2 + i
constexpr int RS_LDX = 144;                     // x row stride (fp16 elements): + 32 B, conflict-free fragment reads (seanet_res128x3.hip)
constexpr int RS_LDH = 64;                      // h rows DENSE (128 B), 16-byte chunk index XORed with row & 7 (rs_hoff, round 6): fragment reads conflict-free, the
                                                // epilogue's 8-byte stores 2-way instead of 4-way (rows 160 B apart put 16 rows of one column on 4 bank pairs)
constexpr int RS_XP = RS_XROWS * RS_LDX;        // one piece plane of an x tile
constexpr int RS_XBUF = 4 * RS_XP;              // Xe hi, Xe lo, Xr hi, Xr lo
constexpr int RS_HP = RS_TT * RS_LDH;           // one piece plane of an h tile
constexpr int RS_HBUF = 2 * RS_HP;
constexpr int RS_CHUNKS = RS_XROWS * 32;        // float4 chunks of an input tile
constexpr int RS_PRE = (RS_CHUNKS + 511) / 512; // per thread: 3 (the last one partly)
constexpr size_t RS_LDS_BYTES = (size_t)(3 * RS_XBUF + 2 * RS_HBUF) * 2;
__device__ __forceinline__ int rs_hoff(int row, int chunk) { return row * RS_LDH + ((chunk ^ (row & 7)) << 3); }

// tile row of tail fragment column pos = 16 m + r16 when the output goes to the stride-5 consumer's phase planes (see tail_row below)
__device__ constexpr unsigned char kRsTailRows[32] = {2, 7, 12, 17, 0, 5, 10, 15, 20, 25, 30, 3, 22, 27, 8, 13,
                                                      4, 9, 14, 19, 1, 6, 11, 16, 21, 26, 31, 28, 24, 29, 18, 23};
constexpr bool rs_tail_rows_ok() {
    unsigned seen = 0;
    for (int m = 0; m < 2; ++m) {
        unsigned a8 = 0, b8 = 0;   // residues mod 8 of the two 8-row sets of a ds_read_b128 lane group
        for (int r = 0; r < 16; ++r) {
            const int row = kRsTailRows[16 * m + r];
            seen |= 1u << row;
            if (r >= 4 && r < 12) b8 |= 1u << (row & 7); else a8 |= 1u << (row & 7);
        }
        if (a8 != 0xffu || b8 != 0xffu) return false;
    }
    return seen == 0xffffffffu;
}
static_assert(rs_tail_rows_ok(), "tail row order: a permutation of 0..31 whose ds_read_b128 lane groups are distinct mod 8");
}  // namespace

__global__ __launch_bounds__(512, 1) void seanet_res128rs_kernel(Res64Args a) {
    typedef SchemeF16x2 SC;
    typedef SC::T PT;
    typedef SC::V8 V8;
    typedef SC::V4 V4;
    constexpr int NP = 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char rs_lds_raw[];
    PT* Xb = reinterpret_cast<PT*>(rs_lds_raw);          // [3][4 planes][34][144]
    PT* Hb = Xb + 3 * RS_XBUF;                           // [2][2 pieces][32][80]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int role_t = __builtin_amdgcn_readfirstlane(wave >> 2);   // 0: conv3 wave, 1: tail wave
    const int w = wave & 3;
    const int r16 = lane & 15, q = lane >> 4;
    const int L = a.L;
    const int tiles_per_clip = (L + RS_TT - 1) / RS_TT;
    const int total_tiles = a.B * tiles_per_clip;        // < 2^30: checked by the launcher
    const float sa = a.act_scale;
    const float rs3 = 1.0f / (a.act_scale * a.w3_scale), rst = 1.0f / (a.act_scale * a.wt_scale);
    RangeMax over;

    // ---- this wave's weights -> two fp16 pieces in 96 registers (MFMA A operand: row r16, k = 32 ks + 8 q .. + 7) ---------------------------
    // C wave: wreg[p][ks] = W3 row 16 w + r16, K step ks (12). T wave: wreg[p][6 n + ks] = [W1 | Wsc] row 32 w + 16 n + r16, K step ks (6).
    V8 wreg[NP][12];
    {
        const float scale = role_t ? a.wt_scale : a.w3_scale;
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const float* src = role_t ? a.wt + (w * 32 + (i / 6) * 16 + r16) * 192 + (i % 6) * 32 + q * 8
                                      : a.w3 + (w * 16 + r16) * 384 + i * 32 + q * 8;
            const f4 lo = *reinterpret_cast<const f4*>(src), hi = *reinterpret_cast<const f4*>(src + 4);
            V4 plo[NP], phi[NP];
            split4<SchemeNoCheck<SC>>(lo, scale, plo);
            split4<SchemeNoCheck<SC>>(hi, scale, phi);
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int k = 0; k < 4; ++k) { wreg[p][i][k] = plo[p][k]; wreg[p][i][4 + k] = phi[p][k]; }
        }
    }
    const f4 b3 = *reinterpret_cast<const f4*>(a.b3 + w * 16 + q * 4);
    f4 bt[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) bt[n] = *reinterpret_cast<const f4*>(a.bt + w * 32 + n * 16 + q * 4);

    // ---- input staging: chunk c = tid + 512 j -> (row = c / 32, float4 = c % 32) ---------------------------------------------------------------
    f4 pre[RS_PRE];
    auto prefetch = [&](int tile) {
        const int b = tile / tiles_per_clip;
        const int t0 = (tile - b * tiles_per_clip) * RS_TT;
        const float* xb = a.x + (long long)b * L * 128;
#pragma unroll
        for (int j = 0; j < RS_PRE; ++j) {
            int c = tid + 512 * j;
            c = c < RS_CHUNKS ? c : RS_CHUNKS - 1;   // the surplus threads of the last round re-load the last chunk and do not store it
            int tau = t0 - 2 + (c >> 5);
            tau = tau < 0 ? -tau : tau;            // causal reflect padding at the clip start
            tau = tau > L - 1 ? L - 1 : tau;       // rows past the end are never stored
            pre[j] = *reinterpret_cast<const f4*>(xb + (unsigned)(tau * 128 + (c & 31) * 4));
        }
    };
    auto stage = [&](int buf) {
        PT* X = Xb + buf * RS_XBUF;
#pragma unroll
        for (int j = 0; j < RS_PRE; ++j) {
            const int c = tid + 512 * j;
            if (c < RS_CHUNKS) {
                const int row = c >> 5, c4 = c & 31;          // float4 c4 = half (c4 & 1) of the 8-channel chunk c4 >> 1
                const f4 v = pre[j];
                const f4 e = {elu1(v.x), elu1(v.y), elu1(v.z), elu1(v.w)};
                V4 rp[NP], ep[NP];
                over |= split4<SC>(v, sa, rp);
                split4<SchemeNoCheck<SC>>(e, sa, ep);      // |ELU(x)| <= max(|x|, 1): covered by the check of x
                const int off = row * RS_LDX + ((c4 >> 1) << 3) + ((c4 & 1) << 2);
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    *reinterpret_cast<V4*>(X + p * RS_XP + off) = ep[p];
                    *reinterpret_cast<V4*>(X + (2 + p) * RS_XP + off) = rp[p];
                }
            }
        }
    };
    // Which tile row a fragment column of the TAIL stands for. The 1x1 tail may take its rows in any order; with the split output the 32 rows are grouped by
    // row % 5 (the consumer's phase planes: rows 5 i + c of one class are consecutive indices of one plane) so that the 32-byte piece stores of neighbouring
    // lanes join into longer runs. Round 6: the grouping is BANK-AWARE. A ds_read_b128 is served in 16-lane groups {r16 in 0-3 | 12-15 at one q, r16 in 4-11 at
    // the next q}; with row strides of 18 (x) and 10 (h) 16-byte slots a group is conflict-free exactly when its two sets of 8 rows are distinct mod 8 each.
    // Round 5's order (classes back to back: 7, 7, 6, 6, 6 rows) put rows 10 / 26 / 2 — all 2 mod 8 — into one set: 3-way conflicts on every tail read, 0.34
    // of the kernel's LDS cycles (PMC lds_conflict_share; tools/lds_bank_sim.py reproduces it). What a store instruction costs is the number of distinct runs its
    // 16 rows cover, whatever lanes hold them (the 4 lanes of one row are 16 apart anyway), so each row tile keeps THREE runs like round 5's order: row tile 0 =
    // class 2 whole (lanes 0-3, 12-13) + class 0 whole (lanes 4-10) + class 3's first three rows (lanes 11, 14, 15); row tile 1 = class 4 whole + class 1 whole +
    // class 3's last three — every 8-row set distinct mod 8 (checked at compile time). (A first bank-aware order with five runs per row tile was conflict-free
    // and 4 % SLOWER: profiles/EXPERIMENTS.md.)
    const bool phase_order = a.S != nullptr;
    auto tail_row = [&](int pos) { return phase_order ? (int)kRsTailRows[pos] : pos; };

    // per-lane constants of the loop (hoisted by hand: inside the role branch the table lookup was a dependent global load at the head of EVERY tail)
    int trow[2], thoff[2][2], hst[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        trow[m] = tail_row(16 * m + r16);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) thoff[m][ks] = rs_hoff(trow[m], ks * 4 + q);          // tail: h fragment of K step ks
        hst[m] = rs_hoff(16 * m + r16, 2 * w + (q >> 1)) + ((q & 1) << 2);                    // conv3 epilogue: channels 16 w + 4 q .. + 3 of row 16 m + r16
    }
    const int first = blockIdx.x, step = gridDim.x;
    const int K = first < total_tiles ? (total_tiles - first + step - 1) / step : 0;
    if (K == 0) return;
    prefetch(first);
    stage(0);
    __syncthreads();
    int xb_cur = 0, xb_prev = 2;   // x buffer of tile k, of tile k - 1; tile k + 1 goes to the third
    for (int k = 0; k <= K; ++k) {
        const int tile = first + k * step;
        const int xb_next = 3 - xb_cur - xb_prev;
        if (k + 1 < K) prefetch(tile + step);   // flies during this iteration's MFMAs
        __builtin_amdgcn_sched_barrier(0);
        if (role_t == 0) {
            if (k < K) {
                // ---- h[:, 16w..16w+15] = ELU(conv3(ELU(x)) + b3): output row j uses x rows j, j+1, j+2; K step ks = (tap, 32 channels) --------
                const PT* Xe = Xb + xb_cur * RS_XBUF;
                PT* Hs = Hb + (k & 1) * RS_HBUF;
                f4 acc[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
                auto xread = [&](int ks, V8 (&xf)[NP][2]) {
                    const int tap = ks >> 2, chunk = (ks & 3) * 4 + q;
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        const PT* src = Xe + (16 * m + r16 + tap) * RS_LDX + (chunk << 3);
#pragma unroll
                        for (int p = 0; p < NP; ++p) xf[p][m] = *reinterpret_cast<const V8*>(src + p * RS_XP);
                    }
                };
                V8 xa[NP][2], xq[NP][2];
                xread(0, xa);
#pragma unroll
                for (int ks = 0; ks < 12; ks += 2) {
                    xread(ks + 1, xq);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = 0; t < SC::NPROD; ++t)
#pragma unroll
                        for (int m = 0; m < 2; ++m) acc[m] = SC::mfma16(wreg[SC::prod_w(t)][ks], xa[SC::prod_a(t)][m], acc[m]);
                    if (ks + 2 < 12) xread(ks + 2, xa);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = 0; t < SC::NPROD; ++t)
#pragma unroll
                        for (int m = 0; m < 2; ++m) acc[m] = SC::mfma16(wreg[SC::prod_w(t)][ks + 1], xq[SC::prod_a(t)][m], acc[m]);
                }
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const f4 v = acc[m] * rs3 + b3;
                    const f4 e = {elu1(v.x), elu1(v.y), elu1(v.z), elu1(v.w)};
                    V4 hp[NP];
                    over |= split4<SC>(e, sa, hp);
                    const int off = hst[m];
#pragma unroll
                    for (int p = 0; p < NP; ++p) *reinterpret_cast<V4*>(Hs + p * RS_HP + off) = hp[p];
                }
            }
        } else if (k >= 1) {
            // ---- out[:, 32w..32w+31] = ELU([h | x] . [W1 | Wsc]^T + (b1 + bsc)) of tile k - 1: output row j uses h row j and x row j + 2 ----------
            const int ptile = tile - step;
            const int b = ptile / tiles_per_clip;
            const int t0 = (ptile - b * tiles_per_clip) * RS_TT;
            const PT* Xr = Xb + xb_prev * RS_XBUF + 2 * RS_XP;
            const PT* Hs = Hb + ((k - 1) & 1) * RS_HBUF;
            f4 acc[2][2];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[m][n] = f4{0.f, 0.f, 0.f, 0.f};
            auto tread = [&](int ks, V8 (&xf)[NP][2]) {
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const PT* src = ks < 2 ? Hs + thoff[m][ks] : Xr + (trow[m] + 2) * RS_LDX + (((ks - 2) * 4 + q) << 3);
                    const int ps = ks < 2 ? RS_HP : RS_XP;
#pragma unroll
                    for (int p = 0; p < NP; ++p) xf[p][m] = *reinterpret_cast<const V8*>(src + p * ps);
                }
            };
            V8 xa[NP][2], xq[NP][2];
            tread(0, xa);
#pragma unroll
            for (int ks = 0; ks < 6; ks += 2) {
                tread(ks + 1, xq);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < SC::NPROD; ++t)
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int n = 0; n < 2; ++n)
                            acc[m][n] = SC::mfma16(wreg[SC::prod_w(t)][6 * n + ks], xa[SC::prod_a(t)][m], acc[m][n]);
                if (ks + 2 < 6) tread(ks + 2, xa);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < SC::NPROD; ++t)
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int n = 0; n < 2; ++n)
                            acc[m][n] = SC::mfma16(wreg[SC::prod_w(t)][6 * n + ks + 1], xq[SC::prod_a(t)][m], acc[m][n]);
            }
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int t = t0 + trow[m];
                if (t < L) {
                    float* dst = a.out + ((long long)b * L + t) * 128 + w * 32 + q * 4;
                    const int plane = t % 5, idx = t / 5 + 1;
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        const f4 v = acc[m][n] * rst + bt[n];
                        const f4 o = {elu1(v.x), elu1(v.y), elu1(v.z), elu1(v.w)};
                        if (a.S) {   // the consumer is the stride-5 split GEMM: K-blocked, phase-major pieces (bf16x3 or f16x2: a.S_scheme)
                            const long long ps = (long long)a.B * 8 * 5 * a.Lp * 16;
                            const long long off = ((((long long)b * 8 + w * 2 + n) * 5 + plane) * a.Lp + idx) * 16 + q * 4;
                            if (a.S_scheme == XB_SCHEME_F16X2) {
                                V4 pp[2];
                                over |= split4<SC>(o, a.S_scale, pp);
                                PT* d = reinterpret_cast<PT*>(a.S) + off;
                                *reinterpret_cast<V4*>(d) = pp[0];
                                *reinterpret_cast<V4*>(d + ps) = pp[1];
                            } else {
                                SchemeBf16x3::V4 pp[3];
                                split4<SchemeBf16x3>(o, 1.0f, pp);
                                __bf16* d = a.S + off;
#pragma unroll
                                for (int p = 0; p < 3; ++p) *reinterpret_cast<SchemeBf16x3::V4*>(d + p * ps) = pp[p];
                            }
                        } else {
                            *reinterpret_cast<f4*>(dst + n * 16) = o;
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (k + 1 < K) stage(xb_next);
        __syncthreads();
        xb_prev = xb_cur; xb_cur = xb_next;
    }
    range_publish(a.status, a.status ? a.status + 1 : nullptr, over);
}

int launch_seanet_res128rs(const Res64Args& a, hipStream_t stream) {
    AT_REQUIRE(a.L >= 4 && a.B >= 1, "fused resblock needs at least 4 rows");
    AT_REQUIRE(a.scheme == XB_SCHEME_F16X2 && a.act_scale > 0.f && a.w3_scale > 0.f && a.wt_scale > 0.f, "res128rs: the fp16 scheme and its scales");
    const long long tiles = (long long)a.B * ((a.L + RS_TT - 1) / RS_TT);
    AT_REQUIRE(tiles < (1LL << 30) && (long long)a.L * 128 < (1LL << 30), "tile / offset arithmetic is 32-bit");
    const int grid = (int)(tiles < 256 ? tiles : 256);   // one workgroup per CU
    { static LdsAttrFlags lds_attr; if (int rc = set_max_dynamic_lds(lds_attr, seanet_res128rs_kernel, RS_LDS_BYTES)) return rc; }
    hipLaunchKernelGGL(seanet_res128rs_kernel, dim3(grid), dim3(512), RS_LDS_BYTES, stream, a);
    AT_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace at
