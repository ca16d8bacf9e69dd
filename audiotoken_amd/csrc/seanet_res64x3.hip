// SEANet residual block at 64 channels on the bf16 matrix cores — the block of seanet_res64.hip
//   out = ELU( [W1 | Wsc] . [ELU(W3 * ELU(x) + b3) | x] + (b1 + bsc) )          (k3 64->32, k1 32->64, k1 shortcut)
// with every operand as an exact 3-way bf16 split and six v_mfma_f32_16x16x32_bf16 per 32-wide K step (arithmetic and accuracy:
// gemm_bf16x3.hip; structure: seanet_res128x3.hip). Work split as in the fp32 kernel: conv3 wave = (channel tile w & 1, row half
// w >> 1), tail wave = channel tile w; the weights a wave needs are 108 registers of bf16 pieces, so TWO workgroups fit a CU
// (72 KB of LDS each) and one stages / splits (vector work) while the other multiplies — a bf16 MFMA leaves half of its cycles
// to vector issue, unlike the fp32 MFMA.
//   Xe = split(ELU(x)), Xr = split(x): [3][66 rows][64 ch + 8 pad] bf16 each;  H = split(ELU(conv3 + b3)): [3][64][32 + 8 pad]
// (the 16-byte pads put the 16 consecutive rows of a fragment read on 16 distinct bank groups with plain linear addresses).
// Rounds differently from the fp32 chain: compared by tolerance and identical tokens (tests/test_acoustic_gpu.py).
#include "gemm_core.h"
#include "encodec_kernels.h"
#include "split_scheme.h"

namespace at {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int RY_TT = 64;                    // time rows per tile
constexpr int RY_XROWS = 66;                 // row i <-> time t0 - 2 + i
constexpr int RY_CHUNKS = RY_XROWS * 16;     // float4 chunks of the input tile
constexpr int RY_PRE = (RY_CHUNKS + 255) / 256;
constexpr int RY_LDX = 80, RY_LDH = 40;      // row strides (bf16): x rows + 32 B (conflict-free fragment reads, see seanet_res128x3.hip); the h rows keep + 16 B (two workgroups per CU: 78.7 KB each)
constexpr int RY_XP = RY_XROWS * RY_LDX;     // bf16 elements of one piece of an x tile
constexpr int RY_HP = RY_TT * RY_LDH;

// SC = operand scheme (split_scheme.h): three bf16 pieces / six products, or (round 2, default) two fp16 pieces / three products with the
// activations scaled by a.act_scale and the two weight matrices by a.w3_scale / a.wt_scale (powers of two; the accumulators are rescaled
// before the bias); an activation beyond the fp16 range raises XB_STATUS_F16_OVERFLOW in *a.status.
template <class SC>
__global__ __launch_bounds__(256, 2) void seanet_res64x3_kernel(Res64Args a) {
    typedef typename SC::T PT;
    typedef typename SC::V8 V8;
    typedef typename SC::V4 V4;
    constexpr int NP = SC::NP;
    extern __shared__ __attribute__((aligned(16))) unsigned char ry_lds_raw[];
    PT* Xe = reinterpret_cast<PT*>(ry_lds_raw);   // split(ELU(x))
    PT* Xr = Xe + NP * RY_XP;            // split(x)
    PT* Hs = Xr + NP * RY_XP;            // split(ELU(conv3 + b3))
    const float sa = SC::RANGE_CHECK ? a.act_scale : 1.0f;
    const float rs3 = SC::RANGE_CHECK ? 1.0f / (a.act_scale * a.w3_scale) : 1.0f, rst = SC::RANGE_CHECK ? 1.0f / (a.act_scale * a.wt_scale) : 1.0f;
    RangeMax over;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, q = lane >> 4;
    const int L = a.L;
    const int tiles_per_clip = (L + RY_TT - 1) / RY_TT;
    const int total_tiles = a.B * tiles_per_clip;   // < 2^30: checked by the launcher
    const int cn = wave & 1, mh = wave >> 1;        // conv3: channel tile, row half

    // ---- weights -> 3 bf16 pieces in registers, once per workgroup (MFMA A operand: row r16, k = 32 ks + 8 q .. + 7) -------------
    V8 w3p[NP][6], wtp[NP][3];
    auto wsplit = [&](const float* src, float scale, V8 (&dst)[NP][6], int ks) {   // dst[piece][ks] (the [NP][3] array is passed through a cast below)
        const f4 lo = *reinterpret_cast<const f4*>(src), hi = *reinterpret_cast<const f4*>(src + 4);
        V4 plo[NP], phi[NP];
        split4<SchemeNoCheck<SC>>(lo, scale, plo);
        split4<SchemeNoCheck<SC>>(hi, scale, phi);
#pragma unroll
        for (int i = 0; i < NP; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) { dst[i][ks][k] = plo[i][k]; dst[i][ks][4 + k] = phi[i][k]; }
    };
    const float sw3 = SC::RANGE_CHECK ? a.w3_scale : 1.0f, swt = SC::RANGE_CHECK ? a.wt_scale : 1.0f;
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) wsplit(a.w3 + (cn * 16 + r16) * 192 + ks * 32 + q * 8, sw3, w3p, ks);
    {
        V8 wt6[NP][6];
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) wsplit(a.wt + (wave * 16 + r16) * 96 + ks * 32 + q * 8, swt, wt6, ks);
#pragma unroll
        for (int i = 0; i < NP; ++i)
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) wtp[i][ks] = wt6[i][ks];
    }
    const f4 b3 = *reinterpret_cast<const f4*>(a.b3 + cn * 16 + q * 4);
    const f4 bt = *reinterpret_cast<const f4*>(a.bt + wave * 16 + q * 4);

    // input staging: chunk c = tid + 256 j -> (row = c / 16, float4 = c % 16)
    f4 pre[RY_PRE];
    auto prefetch = [&](int tile) {
        const int b = tile / tiles_per_clip;
        const int t0 = (tile - b * tiles_per_clip) * RY_TT;
        const float* xb = a.x + (long long)b * L * 64;
#pragma unroll
        for (int j = 0; j < RY_PRE; ++j) {
            int c = tid + 256 * j;
            c = c < RY_CHUNKS ? c : RY_CHUNKS - 1;
            int tau = t0 - 2 + (c >> 4);
            tau = tau < 0 ? -tau : tau;            // causal reflect padding at the clip start
            tau = tau > L - 1 ? L - 1 : tau;       // rows past the end are never stored
            pre[j] = *reinterpret_cast<const f4*>(xb + (unsigned)(tau * 64 + (c & 15) * 4));
        }
    };
    if ((int)blockIdx.x < total_tiles) prefetch(blockIdx.x);

    for (int tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
        const int b = tile / tiles_per_clip;
        const int t0 = (tile - b * tiles_per_clip) * RY_TT;
        __syncthreads();   // previous tile's readers are done
#pragma unroll
        for (int j = 0; j < RY_PRE; ++j) {
            const int c = tid + 256 * j;
            if (c < RY_CHUNKS) {
                const int row = c >> 4, c4 = c & 15;
                const f4 v = pre[j];
                const f4 e = {elu1(v.x), elu1(v.y), elu1(v.z), elu1(v.w)};
                V4 rp[NP], ep[NP];
                over |= split4<SC>(v, sa, rp);
                split4<SchemeNoCheck<SC>>(e, sa, ep);      // |ELU(x)| <= max(|x|, 1): covered by the check of x
                const int off = row * RY_LDX + c4 * 4;
#pragma unroll
                for (int i = 0; i < NP; ++i) {
                    *reinterpret_cast<V4*>(Xr + i * RY_XP + off) = rp[i];
                    *reinterpret_cast<V4*>(Xe + i * RY_XP + off) = ep[i];
                }
            }
        }
        __syncthreads();
        // ---- h[32 mh .. + 31, 16 cn .. + 15] = ELU(conv3(ELU(x)) + b3): output row j uses x rows j, j+1, j+2; ks = (tap, 32 channels) ----
        {
            f4 acc[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
            auto xread = [&](int ks, V8 (&xf)[NP][2]) {
                const int tap = ks >> 1, chunk = (ks & 1) * 4 + q;
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const PT* src = Xe + (32 * mh + 16 * m + r16 + tap) * RY_LDX + chunk * 8;
#pragma unroll
                    for (int p = 0; p < NP; ++p) xf[p][m] = *reinterpret_cast<const V8*>(src + p * RY_XP);
                }
            };
            V8 xa[NP][2], xb[NP][2];
            xread(0, xa);
#pragma unroll
            for (int ks = 0; ks < 6; ks += 2) {
                xread(ks + 1, xb);
#pragma unroll
                for (int t = 0; t < SC::NPROD; ++t)
#pragma unroll
                    for (int m = 0; m < 2; ++m) acc[m] = SC::mfma16(w3p[SC::prod_w(t)][ks], xa[SC::prod_a(t)][m], acc[m]);
                if (ks + 2 < 6) xread(ks + 2, xa);
#pragma unroll
                for (int t = 0; t < SC::NPROD; ++t)
#pragma unroll
                    for (int m = 0; m < 2; ++m) acc[m] = SC::mfma16(w3p[SC::prod_w(t)][ks + 1], xb[SC::prod_a(t)][m], acc[m]);
            }
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const f4 v = acc[m] * rs3 + b3;
                const f4 e = {elu1(v.x), elu1(v.y), elu1(v.z), elu1(v.w)};
                V4 hp[NP];
                over |= split4<SC>(e, sa, hp);
                const int off = (32 * mh + 16 * m + r16) * RY_LDH + cn * 16 + q * 4;
#pragma unroll
                for (int i = 0; i < NP; ++i) *reinterpret_cast<V4*>(Hs + i * RY_HP + off) = hp[i];
            }
        }
        __syncthreads();
        if (tile + (int)gridDim.x < total_tiles) prefetch(tile + gridDim.x);   // flies during the tail MFMAs and the barrier wait
        // ---- out[:, 16w..16w+15] = ELU([h | x] . [W1 | Wsc]^T + (b1 + bsc)): output row j uses h row j and x row j + 2 ----------------
#pragma unroll 1
        for (int mp = 0; mp < 4; mp += 2) {
            f4 acc[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
            V8 xf[3][NP][2];   // [k step][piece][row tile]
#pragma unroll
            for (int ks = 0; ks < 3; ++ks)
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int row = 16 * (mp + m) + r16;
                    const PT* src = ks == 0 ? Hs + row * RY_LDH + q * 8 : Xr + (row + 2) * RY_LDX + ((ks - 1) * 4 + q) * 8;
                    const int ps = ks == 0 ? RY_HP : RY_XP;
#pragma unroll
                    for (int p = 0; p < NP; ++p) xf[ks][p][m] = *reinterpret_cast<const V8*>(src + p * ps);
                }
#pragma unroll
            for (int ks = 0; ks < 3; ++ks)
#pragma unroll
                for (int t = 0; t < SC::NPROD; ++t)
#pragma unroll
                    for (int m = 0; m < 2; ++m) acc[m] = SC::mfma16(wtp[SC::prod_w(t)][ks], xf[ks][SC::prod_a(t)][m], acc[m]);
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int t = t0 + (mp + m) * 16 + r16;
                if (t < L) {
                    const f4 v = acc[m] * rst + bt;
                    f4 o;
                    o.x = elu1(v.x); o.y = elu1(v.y); o.z = elu1(v.z); o.w = elu1(v.w);
                    *reinterpret_cast<f4*>(a.out + ((long long)b * L + t) * 64 + wave * 16 + q * 4) = o;
                }
            }
        }
    }
    if constexpr (SC::RANGE_CHECK)
        range_publish(a.status, a.status ? a.status + 1 : nullptr, over);
}

int launch_seanet_res64x3(const Res64Args& a, hipStream_t stream) {
    AT_REQUIRE(a.L >= 4 && a.B >= 1, "fused resblock needs at least 4 rows");
    const long long tiles = (long long)a.B * ((a.L + RY_TT - 1) / RY_TT);
    AT_REQUIRE(tiles < (1LL << 30) && (long long)a.L * 64 < (1LL << 30), "tile / offset arithmetic is 32-bit");
    const int grid = (int)(tiles < 512 ? tiles : 512);   // two workgroups per CU
    if (a.scheme == XB_SCHEME_F16X2) {
        AT_REQUIRE(a.act_scale > 0.f && a.w3_scale > 0.f && a.wt_scale > 0.f, "res64x3: the fp16 scheme needs its scales");
        const size_t lds = (size_t)(4 * RY_XP + 2 * RY_HP) * 2;
        { static LdsAttrFlags lds_attr_0; if (int rc = set_max_dynamic_lds(lds_attr_0, seanet_res64x3_kernel<SchemeF16x2>, lds)) return rc; }
        hipLaunchKernelGGL(seanet_res64x3_kernel<SchemeF16x2>, dim3(grid), dim3(256), lds, stream, a);
    } else {
        const size_t lds = (size_t)(6 * RY_XP + 3 * RY_HP) * 2;
        { static LdsAttrFlags lds_attr_1; if (int rc = set_max_dynamic_lds(lds_attr_1, seanet_res64x3_kernel<SchemeBf16x3>, lds)) return rc; }
        hipLaunchKernelGGL(seanet_res64x3_kernel<SchemeBf16x3>, dim3(grid), dim3(256), lds, stream, a);
    }
    AT_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace at
