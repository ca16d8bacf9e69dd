// SEANet residual block at 128 channels on the bf16 matrix cores — the block of seanet_res128.hip
//   out = ELU( [W1 | Wsc] . [ELU(W3 * ELU(x) + b3) | x] + (b1 + bsc) )          (k3 128->64, k1 64->128, k1 shortcut)
// with every operand as an exact 3-way bf16 split and six v_mfma_f32_16x16x32_bf16 (16 cycles) per 32-wide K step instead of
// eight 32-cycle fp32 MFMAs (arithmetic and accuracy: gemm_bf16x3.hip). Same decomposition as the fp32 kernel: weights stationary
// in REGISTERS, split over the output channels — wave w keeps W3 rows 16w..16w+15 (K = 384) and [W1 | Wsc] rows 32w..32w+31
// (K = 192) as three bf16 pieces = 288 registers; activations stream through LDS in 64-row tiles:
//   Xe = split(ELU(x)), Xr = split(x): [3][72 rows][128 ch + 16 pad] bf16 each;  H = split(ELU(conv3 + b3)): [3][64][64 + 16 pad].
// The splits cost vector instructions (17 per input value: ELU 5 + two splits 6 each) which a single wave per SIMD cannot hide
// behind another wave; the MFMA time saved is larger.
// Rounds differently from the fp32 chain of the GEMM / seanet_res128 path: compared by tolerance and identical tokens
// (tests/test_acoustic_gpu.py::test_x3_kernels_match_fp32). (EnCodec architecture: SURVEY.md Appendix A.1.)
#include "gemm_core.h"
#include "encodec_kernels.h"
#include "split_scheme.h"

namespace at {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int RX_TT = 64;                    // time rows per tile
constexpr int RX_XROWS = 66;                 // row i <-> time t0 - 2 + i
constexpr int RX_PRE = (RX_XROWS * 32 + 255) / 256;   // float4 chunks per thread: 9
constexpr int RX_LDX = 144, RX_LDH = 80;     // row strides (bf16): + 32 B. The 16-byte pad of round 1 assumed a ds_read_b128 is served in groups of 16 CONSECUTIVE lanes; the hardware groups {0-3,12-15,20-27}, ... (MI355X_MICROARCH.md, LDS) made one bank quad collide in every group (PMC: 43-47 % of the LDS cycles were conflict replays). A row stride of 32 B mod 64 B x odd (stride / 16 B = 2 mod 4) is conflict-free under that grouping.
constexpr int RX_XP = RX_PRE * 8 * RX_LDX;   // bf16 elements of one piece of an x tile: 72 rows (the last 6 absorb the tail of the last chunk)
constexpr int RX_HP = RX_TT * RX_LDH;        // one piece of the h tile

// linear in (row, chunk): every fragment address is one per-lane base plus an immediate offset
__device__ __forceinline__ int rx_xoff(int row, int chunk) { return row * RX_LDX + (chunk << 3); }
__device__ __forceinline__ int rx_hoff(int row, int chunk) { return row * RX_LDH + (chunk << 3); }
__device__ __forceinline__ void rx_split(float v, __bf16& p1, __bf16& p2, __bf16& p3) {
    p1 = (__bf16)v;
    const float r1 = v - (float)p1;
    p2 = (__bf16)r1;
    p3 = (__bf16)(r1 - (float)p2);
}

// SC = operand scheme of the block's own contractions (see seanet_res64x3.hip): three bf16 pieces / six products, or (round 2, default) two
// fp16 pieces / three products with power-of-two activation / weight scales and a range status bit.
template <class SC>
__global__ __launch_bounds__(256, 1) void seanet_res128x3_kernel(Res64Args a) {
    typedef typename SC::T PT;
    typedef typename SC::V8 V8;
    typedef typename SC::V4 V4;
    constexpr int NP = SC::NP;
    constexpr int MT = NP == 2 ? 4 : 2;   // conv3 row tiles per pass: all four with two pieces (12 MFMAs = 192 cycles cover a fragment read; one wave per SIMD)
    extern __shared__ __attribute__((aligned(16))) unsigned char rx_lds_raw[];
    PT* Xe = reinterpret_cast<PT*>(rx_lds_raw);   // split(ELU(x))
    PT* Xr = Xe + NP * RX_XP;            // split(x)
    PT* Hs = Xr + NP * RX_XP;            // split(ELU(conv3 + b3))
    const float sa = SC::RANGE_CHECK ? a.act_scale : 1.0f;
    const float sw3 = SC::RANGE_CHECK ? a.w3_scale : 1.0f, swt = SC::RANGE_CHECK ? a.wt_scale : 1.0f;
    const float rs3 = SC::RANGE_CHECK ? 1.0f / (a.act_scale * a.w3_scale) : 1.0f, rst = SC::RANGE_CHECK ? 1.0f / (a.act_scale * a.wt_scale) : 1.0f;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, q = lane >> 4;
    const int L = a.L;
    const int tiles_per_clip = (L + RX_TT - 1) / RX_TT;
    const int total_tiles = a.B * tiles_per_clip;   // < 2^30: checked by the launcher
    RangeMax over;                               // fp16 range check of the f16x2 piece output

    // ---- weights -> 3 bf16 pieces in registers, once per workgroup (MFMA A operand: row r16, k = 32 ks + 8 q .. + 7) -------------
    V8 w3p[NP][12], wtp[NP][2][6];
    auto wsplit = [&](const float* src, float scale, V8 (&dst)[NP]) {
        const f4 lo = *reinterpret_cast<const f4*>(src), hi = *reinterpret_cast<const f4*>(src + 4);
        V4 plo[NP], phi[NP];
        split4<SchemeNoCheck<SC>>(lo, scale, plo);
        split4<SchemeNoCheck<SC>>(hi, scale, phi);
#pragma unroll
        for (int i = 0; i < NP; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) { dst[i][k] = plo[i][k]; dst[i][4 + k] = phi[i][k]; }
    };
#pragma unroll
    for (int ks = 0; ks < 12; ++ks) {
        V8 t[NP];
        wsplit(a.w3 + (wave * 16 + r16) * 384 + ks * 32 + q * 8, sw3, t);
#pragma unroll
        for (int i = 0; i < NP; ++i) w3p[i][ks] = t[i];
    }
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) {
            V8 t[NP];
            wsplit(a.wt + (wave * 32 + n * 16 + r16) * 192 + ks * 32 + q * 8, swt, t);
#pragma unroll
            for (int i = 0; i < NP; ++i) wtp[i][n][ks] = t[i];
        }
    const f4 b3 = *reinterpret_cast<const f4*>(a.b3 + wave * 16 + q * 4);

    // input staging: chunk c = tid + 256 j -> (row = c / 32, float4 = c % 32)
    f4 pre[RX_PRE];
    auto prefetch = [&](int tile) {
        const int b = tile / tiles_per_clip;
        const int t0 = (tile - b * tiles_per_clip) * RX_TT;
        const float* xb = a.x + (long long)b * L * 128;
#pragma unroll
        for (int j = 0; j < RX_PRE; ++j) {
            const int c = tid + 256 * j;
            int tau = t0 - 2 + (c >> 5);           // rows 66..71 of the last chunk are loaded (clamped) and never read
            tau = tau < 0 ? -tau : tau;            // causal reflect padding at the clip start
            tau = tau > L - 1 ? L - 1 : tau;       // rows past the end are never stored
            pre[j] = *reinterpret_cast<const f4*>(xb + (unsigned)(tau * 128 + (c & 31) * 4));
        }
    };
    if ((int)blockIdx.x < total_tiles) prefetch(blockIdx.x);
    // Which tile row a fragment column of the TAIL stands for. The 1x1 tail may take its rows in any order; with the split output the
    // 64 rows are sorted by (row % 5, row / 5): the 16 lanes of a row tile then hold consecutive indices of (mostly) one phase plane
    // and their 32-byte piece stores join into runs of up to 416 bytes instead of 96.
    const bool phase_order = a.S != nullptr;
    auto tail_row = [&](int pos) {
        if (!phase_order) return pos;
        const int pl = pos < 52 ? pos / 13 : 4;
        return 5 * (pos - 13 * pl) + pl;
    };
    for (int tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
        const int b = tile / tiles_per_clip;
        const int t0 = (tile - b * tiles_per_clip) * RX_TT;
        __syncthreads();   // previous tile's readers are done
#pragma unroll
        for (int j = 0; j < RX_PRE; ++j) {
            const int c = tid + 256 * j;
            const int row = c >> 5, c4 = c & 31;          // float4 c4 = half (c4 & 1) of the 8-channel chunk c4 >> 1
            const f4 v = pre[j];
            const f4 e = {elu1(v.x), elu1(v.y), elu1(v.z), elu1(v.w)};
            V4 rp[NP], ep[NP];
            over |= split4<SC>(v, sa, rp);
            split4<SchemeNoCheck<SC>>(e, sa, ep);      // |ELU(x)| <= max(|x|, 1): covered by the check of x
            const int off = rx_xoff(row, c4 >> 1) + ((c4 & 1) << 2);
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                *reinterpret_cast<V4*>(Xr + i * RX_XP + off) = rp[i];
                *reinterpret_cast<V4*>(Xe + i * RX_XP + off) = ep[i];
            }
        }
        __syncthreads();
        // ---- h[:, 16w..16w+15] = ELU(conv3(ELU(x)) + b3): output row j uses x rows j, j+1, j+2; K step ks = (tap, 32 channels) --------
#pragma unroll 1   // (unrolled, the compiler's schedule needs > 512 registers and spills weights)
        for (int mp = 0; mp < 4; mp += MT) {   // MT 16-row tiles at a time: 4 MT NP fragment registers per buffer
            f4 acc[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m] = f4{0.f, 0.f, 0.f, 0.f};
            auto xread = [&](int ks, V8 (&xf)[NP][MT]) {
                const int tap = ks >> 2, chunk = (ks & 3) * 4 + q;
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const PT* src = Xe + rx_xoff(16 * (mp + m) + r16 + tap, chunk);
#pragma unroll
                    for (int p = 0; p < NP; ++p) xf[p][m] = *reinterpret_cast<const V8*>(src + p * RX_XP);
                }
            };
            V8 xa[NP][MT], xb[NP][MT];
            xread(0, xa);
#pragma unroll
            for (int ks = 0; ks < 12; ks += 2) {
                xread(ks + 1, xb);
                __builtin_amdgcn_sched_barrier(0);   // keep the reads one step ahead of their MFMAs (one wave per SIMD)
#pragma unroll
                for (int t = 0; t < SC::NPROD; ++t)
#pragma unroll
                    for (int m = 0; m < MT; ++m) acc[m] = SC::mfma16(w3p[SC::prod_w(t)][ks], xa[SC::prod_a(t)][m], acc[m]);
                if (ks + 2 < 12) xread(ks + 2, xa);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < SC::NPROD; ++t)
#pragma unroll
                    for (int m = 0; m < MT; ++m) acc[m] = SC::mfma16(w3p[SC::prod_w(t)][ks + 1], xb[SC::prod_a(t)][m], acc[m]);
            }
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const f4 v = acc[m] * rs3 + b3;
                const f4 e = {elu1(v.x), elu1(v.y), elu1(v.z), elu1(v.w)};
                V4 hp[NP];
                over |= split4<SC>(e, sa, hp);
                const int off = rx_hoff(16 * (mp + m) + r16, 2 * wave + (q >> 1)) + ((q & 1) << 2);   // channels 16 w + 4 q .. + 3
#pragma unroll
                for (int i = 0; i < NP; ++i) *reinterpret_cast<V4*>(Hs + i * RX_HP + off) = hp[i];
            }
        }
        __syncthreads();
        if (tile + (int)gridDim.x < total_tiles) prefetch(tile + gridDim.x);   // flies during the tail MFMAs and the next tile's staging wait
        __builtin_amdgcn_sched_barrier(0);
        // ---- out[:, 32w..32w+31] = ELU([h | x] . [W1 | Wsc]^T + (b1 + bsc)): output row j uses h row j and x row j + 2 ----------------
#pragma unroll
        for (int mp = 0; mp < 4; mp += 2) {
            f4 acc[2][2];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[m][n] = f4{0.f, 0.f, 0.f, 0.f};
            auto tread = [&](int ks, V8 (&xf)[NP][2]) {
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int row = tail_row(16 * (mp + m) + r16);
                    const PT* src = ks < 2 ? Hs + rx_hoff(row, ks * 4 + q) : Xr + rx_xoff(row + 2, (ks - 2) * 4 + q);
                    const int ps = ks < 2 ? RX_HP : RX_XP;
#pragma unroll
                    for (int p = 0; p < NP; ++p) xf[p][m] = *reinterpret_cast<const V8*>(src + p * ps);
                }
            };
            V8 xa[NP][2], xb[NP][2];
            tread(0, xa);
#pragma unroll
            for (int ks = 0; ks < 6; ks += 2) {
                tread(ks + 1, xb);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < SC::NPROD; ++t)
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int n = 0; n < 2; ++n)
                            acc[m][n] = SC::mfma16(wtp[SC::prod_w(t)][n][ks], xa[SC::prod_a(t)][m], acc[m][n]);
                if (ks + 2 < 6) tread(ks + 2, xa);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < SC::NPROD; ++t)
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int n = 0; n < 2; ++n)
                            acc[m][n] = SC::mfma16(wtp[SC::prod_w(t)][n][ks + 1], xb[SC::prod_a(t)][m], acc[m][n]);
            }
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int t = t0 + tail_row((mp + m) * 16 + r16);
                if (t < L) {
                    float* dst = a.out + ((long long)b * L + t) * 128 + wave * 32 + q * 4;
                    const int plane = t % 5, idx = t / 5 + 1;
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        const f4 v = acc[m][n] * rst + *reinterpret_cast<const f4*>(a.bt + wave * 32 + n * 16 + q * 4);
                        f4 o;
                        o.x = elu1(v.x); o.y = elu1(v.y); o.z = elu1(v.z); o.w = elu1(v.w);
                        if (a.S) {   // the consumer is the stride-5 split GEMM: K-blocked, phase-major pieces (bf16x3 or f16x2: a.S_scheme)
                            const long long ps = (long long)a.B * 8 * 5 * a.Lp * 16;
                            const long long off = ((((long long)b * 8 + wave * 2 + n) * 5 + plane) * a.Lp + idx) * 16 + q * 4;
                            if (a.S_scheme == XB_SCHEME_F16X2) {
                                SchemeF16x2::V4 pp[2];
                                over |= split4<SchemeF16x2>(o, a.S_scale, pp);
                                _Float16* d = reinterpret_cast<_Float16*>(a.S) + off;
                                *reinterpret_cast<SchemeF16x2::V4*>(d) = pp[0];
                                *reinterpret_cast<SchemeF16x2::V4*>(d + ps) = pp[1];
                            } else {
                                bf16x4 s1, s2, s3;
#pragma unroll
                                for (int k = 0; k < 4; ++k) {
                                    __bf16 x1, x2, x3;
                                    rx_split(o[k], x1, x2, x3);
                                    s1[k] = x1; s2[k] = x2; s3[k] = x3;
                                }
                                __bf16* d = a.S + off;
                                *reinterpret_cast<bf16x4*>(d) = s1;
                                *reinterpret_cast<bf16x4*>(d + ps) = s2;
                                *reinterpret_cast<bf16x4*>(d + 2 * ps) = s3;
                            }
                        } else {
                            *reinterpret_cast<f4*>(dst + n * 16) = o;
                        }
                    }
                }
            }
        }
    }
    range_publish(a.status, a.status ? a.status + 1 : nullptr, over);
}

// index 0 of plane i = padded row i = time i - 5 = reflect of time 5 - i (plane (5 - i) % 5, index (5 - i) / 5 + 1)
__global__ void reflect_front5_kernel(__bf16* S, int B, int cblocks, int Lp, int npieces) {
    const int gid = blockIdx.x * 256 + threadIdx.x;          // (piece, clip, cblock, plane, 4-channel group)
    const int total = npieces * B * cblocks * 5 * 4;
    if (gid >= total) return;
    const int c4 = gid & 3, i = (gid >> 2) % 5, rest = (gid >> 2) / 5;   // rest = (piece * B + clip) * cblocks + cblock
    const int t = 5 - i;
    __bf16* base = S + (long long)rest * 5 * Lp * 16;
    *reinterpret_cast<bf16x4*>(base + ((long long)i * Lp) * 16 + c4 * 4) =
        *reinterpret_cast<const bf16x4*>(base + ((long long)(t % 5) * Lp + t / 5 + 1) * 16 + c4 * 4);
}

int launch_reflect_front5(__bf16* S, int B, int cblocks, int Lp, hipStream_t stream, int npieces) {
    const int total = npieces * B * cblocks * 5 * 4;
    hipLaunchKernelGGL(reflect_front5_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, S, B, cblocks, Lp, npieces);
    AT_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_seanet_res128x3(const Res64Args& a, hipStream_t stream) {
    AT_REQUIRE(a.L >= 4 && a.B >= 1, "fused resblock needs at least 4 rows");
    const long long tiles = (long long)a.B * ((a.L + RX_TT - 1) / RX_TT);
    AT_REQUIRE(tiles < (1LL << 30) && (long long)a.L * 128 < (1LL << 30), "tile / offset arithmetic is 32-bit");
    const int grid = (int)(tiles < 256 ? tiles : 256);
    if (a.scheme == XB_SCHEME_F16X2) {
        AT_REQUIRE(a.act_scale > 0.f && a.w3_scale > 0.f && a.wt_scale > 0.f, "res128x3: the fp16 scheme needs its scales");
        const size_t lds = (size_t)(4 * RX_XP + 2 * RX_HP) * 2;
        { static LdsAttrFlags lds_attr_0; if (int rc = set_max_dynamic_lds(lds_attr_0, seanet_res128x3_kernel<SchemeF16x2>, lds)) return rc; }
        hipLaunchKernelGGL(seanet_res128x3_kernel<SchemeF16x2>, dim3(grid), dim3(256), lds, stream, a);
    } else {
        const size_t lds = (size_t)(6 * RX_XP + 3 * RX_HP) * 2;
        { static LdsAttrFlags lds_attr_1; if (int rc = set_max_dynamic_lds(lds_attr_1, seanet_res128x3_kernel<SchemeBf16x3>, lds)) return rc; }
        hipLaunchKernelGGL(seanet_res128x3_kernel<SchemeBf16x3>, dim3(grid), dim3(256), lds, stream, a);
    }
    AT_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace at
