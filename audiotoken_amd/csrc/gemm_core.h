// Core of the fp32 windowed GEMM (see gemm_f32.hip for the design notes): computes the BM x BN accumulator
// tile of one workgroup and leaves it in registers so that callers can attach their own epilogue.
#pragma once
#include "at_common.h"

namespace at {

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int BK = 32;   // default K tile; GemmTile's BKT parameter may select 16 (see gemm_f32.hip)

// Accumulator ownership after gemm_tile(): lane (r16 = lane&15, q = lane>>4) of wave (wm, wn) holds
//   acc[i][j] = out[m = m0 + wm*TM*16 + i*16 + r16][n = n0 + wn*TN*16 + j*16 + q*4 .. +3]
template <int BM, int BN, int WM, int WN, int PRO = PRO_NONE, int BKT = 32>
struct GemmTile {
    static_assert(BKT == 32 || BKT == 16, "K tile of 32 or 16");
    static constexpr int BK = BKT;                        // shadows at::BK inside the struct
    static constexpr int CPR = BKT / 4;                   // 16-byte chunks per LDS row
    static constexpr int RPP = 256 / CPR;                 // rows staged per pass of the 256 threads
    static constexpr int KG = BKT / 16;                   // 16-wide k groups per tile
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static constexpr int TM = BM / WM / 16;
    static constexpr int TN = BN / WN / 16;
    static constexpr int XCH = BM / RPP;                  // float4 chunks per thread, activation tile
    static constexpr int WCH = (BN >= RPP) ? BN / RPP : 1;  // float4 chunks per thread, weight tile
    static_assert(BM % RPP == 0, "BM must be a multiple of the rows per staging pass");
    // 16-B chunk swizzle: conflict-free ds_read_b128 (16 rows, same chunk) and ds_write_b128 (all chunks of a row)
    __device__ static __forceinline__ int swz(int row) { return BKT == 32 ? ((row >> 1) & 7) : ((row >> 1) & 3); }
    static constexpr size_t LDS_BYTES = (size_t)(BM + BN) * BK * 2 * sizeof(float);

    // Two bodies, chosen per workgroup by a scalar branch:
    //  FAST  — interior tile (every window row inside the clip, full M/N/K tiles, single A source): each thread keeps
    //          XCH + WCH precomputed pointers and a tile fetch is "pointer + kt*32" — ~20 VALU instead of ~340 per
    //          K step (PMC: 2.66 VALU per MFMA and 72 % MFMA-pipe busy before this split).
    //  !FAST — boundary tiles: reflect / zero padding, N/K tails, dual-source K; clamped unconditional loads + masks.
    __device__ static __forceinline__ void run(const GemmArgs& a, float* smem, int m0, int n0, int b, f4 (&acc)[TM][TN]) {
        const int Tlast_ = a.Tin - 1;
        // an M tail is fine for the FAST body: rows >= M are clamped to row M-1 when the pointers are set up and their
        // results are dropped by the epilogue
        const int m_hi = m0 + BM - 1 < a.M - 1 ? m0 + BM - 1 : a.M - 1;
        const bool fast = BN >= RPP && a.X2 == nullptr && (a.K % BK) == 0 && n0 + BN <= a.N &&
                          (a.ktaps == 1 || a.ldx == a.Cin) && m0 * a.stride - a.pad_left >= 0 &&
                          m_hi * a.stride - a.pad_left + a.ktaps - 1 <= Tlast_;
        // TAP body: boundary tiles of windowed convs (reflect / zero padding at the clip ends, N tails, rows that are not
        // contiguous windows such as grouped convs): a K tile lies inside one tap, so the row pointers are rebuilt only
        // when the tap changes (every Cin / BK tiles) instead of per K step.
        const bool tap_ok = BN >= RPP && (a.K % BK) == 0 && (a.X2 != nullptr ? (a.K1 % BK) == 0 : (a.ktaps == 1 || (a.Cin % BK) == 0));
        if (fast) run_impl<1>(a, smem, m0, n0, b, acc);
        else if (tap_ok) run_impl<2>(a, smem, m0, n0, b, acc);
        else run_impl<0>(a, smem, m0, n0, b, acc);
    }

    template <int MODE>   // 1 FAST, 2 TAP, 0 general
    __device__ static __forceinline__ void run_impl(const GemmArgs& a, float* smem, int m0, int n0, int b, f4 (&acc)[TM][TN]) {
        constexpr bool FAST = MODE == 1;
        constexpr bool TAP = MODE == 2;
        float* Xs = smem;                // [2][BM*32]
        float* Ws = smem + 2 * BM * BK;  // [2][BN*32]
        const int tid = threadIdx.x;
        const int lane = tid & 63;
        const int wave = tid >> 6;
        const int wm = wave / WN, wn = wave % WN;
        const int r16 = lane & 15, q = lane >> 4;
        const float* Xb = a.X + (long long)b * a.x_bstride;
        const int lrow = tid / CPR;  // 0..RPP-1
        const int kc = tid % CPR;    // 16-byte chunk within the K slice

        f4 xreg[XCH], wreg[WCH];
        f4 ureg[PRO == PRO_POWER ? XCH : 1];
        unsigned xok = 0u, wok = 0u;
        bool xsecond = false;
        const int nk = (a.K + BK - 1) / BK;

        // Every load below is UNCONDITIONAL (addresses clamped into range, result masked by a select afterwards):
        // a load under a per-element runtime branch makes hipcc serialise the whole tile fetch behind
        // s_waitcnt vmcnt(0) (CDNA guide §5 "three .s-level traps" (c)) — measured 70 -> ~110 TFLOP/s on this kernel.
        const int Mlast = a.M - 1, Nlast = a.N - 1, Tlast = a.Tin - 1;
        const float* xp[XCH];
        const float* wp[WCH];
        if (FAST) {
#pragma unroll
            for (int j = 0; j < XCH; ++j) {
                const int m = m0 + lrow + j * RPP;
                xp[j] = Xb + (long long)((m < a.M ? m : Mlast) * a.stride - a.pad_left) * a.ldx + kc * 4;
            }
#pragma unroll
            for (int j = 0; j < WCH; ++j) wp[j] = a.W + (long long)(n0 + lrow + j * RPP) * a.K + kc * 4;
        }
        int tap_cur = 0, tap_off = 0;   // TAP body: current tap and the offset of the K tile inside it
        if (TAP) {
#pragma unroll
            for (int j = 0; j < WCH; ++j) {
                const int n = n0 + lrow + j * RPP;
                wp[j] = a.W + (long long)(n < a.N ? n : Nlast) * a.K + kc * 4;   // columns >= N are dropped by the epilogue
            }
        }
        auto tap_rows = [&](int tap) {   // row pointers (and zero-padding mask) of this thread's rows for one tap
            xok = 0u;
#pragma unroll
            for (int j = 0; j < XCH; ++j) {
                const int m = m0 + lrow + j * RPP;
                int r = (m < a.M ? m : Mlast) * a.stride + tap - a.pad_left;
                const bool lo = r < 0, hi = r > Tlast;
                const bool ok = a.pad_mode != 0 || !(lo || hi);
                r = lo ? -r : (hi ? 2 * Tlast - r : r);
                r = r < 0 ? 0 : (r > Tlast ? Tlast : r);
                xp[j] = Xb + (long long)r * a.ldx + kc * 4;
                xok |= (ok ? 1u : 0u) << j;
            }
        };
        auto second_rows = [&]() {   // dual-source A (ktaps == 1): the K tiles from K1 on come from X2
            const float* X2b = a.X2 + (long long)b * a.x2_bstride;
#pragma unroll
            for (int j = 0; j < XCH; ++j) {
                const int m = m0 + lrow + j * RPP;
                xp[j] = X2b + (long long)(m < a.M ? m : Mlast) * a.ld2 + kc * 4;
            }
            xok = ~0u;
            xsecond = true;
        };
        auto load_tile = [&](int kt) {
            if (FAST) {
#pragma unroll
                for (int j = 0; j < XCH; ++j) {
                    xreg[j] = *reinterpret_cast<const f4*>(xp[j] + kt * BK);
                    if (PRO == PRO_POWER) ureg[j] = *reinterpret_cast<const f4*>(xp[j] + kt * BK + a.aux_off);
                }
#pragma unroll
                for (int j = 0; j < WCH; ++j) wreg[j] = *reinterpret_cast<const f4*>(wp[j] + kt * BK);
                return;
            }
            if (TAP) {   // tiles are requested in k order: advance the tap when the tile leaves it
                if (kt == 0) { tap_cur = 0; tap_off = 0; xsecond = false; tap_rows(0); }
                else {
                    tap_off += BK;
                    if (a.ktaps > 1 && tap_off >= a.Cin) { tap_off = 0; ++tap_cur; tap_rows(tap_cur); }
                    else if (a.X2 != nullptr && tap_cur == 0 && tap_off >= a.K1) { tap_off = 0; tap_cur = 1; second_rows(); }
                }
#pragma unroll
                for (int j = 0; j < XCH; ++j) {
                    xreg[j] = *reinterpret_cast<const f4*>(xp[j] + tap_off);
                    if (PRO == PRO_POWER) ureg[j] = *reinterpret_cast<const f4*>(xp[j] + tap_off + a.aux_off);
                }
#pragma unroll
                for (int j = 0; j < WCH; ++j) wreg[j] = *reinterpret_cast<const f4*>(wp[j] + kt * BK);
                return;
            }
            const int kk0 = kt * BK + kc * 4;
            const bool kvalid = kk0 < a.K;
            xok = 0u;
            wok = 0u;
            const int kk = kvalid ? kk0 : 0;
            const int tap = a.ktaps == 1 ? 0 : kk / a.Cin;
            // optional second A source for k >= K1 (ktaps == 1 only): A = [ pro(X) | X2 ] — lets a SEANet residual
            // block fuse "conv1x1(ELU(h)) + shortcut1x1(x)" into one GEMM over the concatenated K
            const bool second = a.X2 != nullptr && kk >= a.K1;
            xsecond = second;
            const int ci = second ? kk - a.K1 : kk - tap * a.Cin;
            const float* Xsrc = second ? a.X2 + (long long)b * a.x2_bstride : Xb;
            const int ldsrc = second ? a.ld2 : a.ldx;
#pragma unroll
            for (int j = 0; j < XCH; ++j) {
                const int m = m0 + lrow + j * RPP;
                const int mc = m < a.M ? m : Mlast;
                int r = mc * a.stride + tap - a.pad_left;
                bool ok = kvalid && (m < a.M);
                const bool lo = r < 0, hi = r > Tlast;
                ok = ok && (a.pad_mode != 0 || !(lo || hi));
                r = lo ? -r : (hi ? 2 * Tlast - r : r);
                r = r < 0 ? 0 : (r > Tlast ? Tlast : r);
                const float* src = Xsrc + (long long)r * ldsrc + ci;
                f4 v = *reinterpret_cast<const f4*>(src);
                if (PRO == PRO_POWER) ureg[j] = *reinterpret_cast<const f4*>(src + a.aux_off);
                xreg[j] = v;                 // raw: the mask is applied in store_tile so that nothing consumes the
                xok |= (ok ? 1u : 0u) << j;  // load before the MFMAs of the current tile have been issued
            }
#pragma unroll
            for (int j = 0; j < WCH; ++j) {
                const int n = n0 + lrow + j * RPP;
                const int nc = n < a.N ? n : Nlast;
                wreg[j] = *reinterpret_cast<const f4*>(a.W + (long long)nc * a.K + kk);
                wok |= ((kvalid && n < a.N) ? 1u : 0u) << j;
            }
        };
        auto store_tile = [&](int buf) {
#pragma unroll
            for (int j = 0; j < XCH; ++j) {
                const int row = lrow + j * RPP;
                f4 v = xreg[j];
                if (PRO == PRO_POWER) v = v * v + ureg[j] * ureg[j];
                if (!FAST && !((xok >> j) & 1u)) v = f4{0.f, 0.f, 0.f, 0.f};
                if (PRO == PRO_ELU && (FAST || !xsecond)) { v.x = elu1(v.x); v.y = elu1(v.y); v.z = elu1(v.z); v.w = elu1(v.w); }  // ELU(0) = 0
                *reinterpret_cast<f4*>(Xs + buf * BM * BK + row * BK + ((kc ^ swz(row)) << 2)) = v;
            }
#pragma unroll
            for (int j = 0; j < WCH; ++j) {
                const int row = lrow + j * RPP;
                if (BN >= RPP || row < BN)
                    *reinterpret_cast<f4*>(Ws + buf * BN * BK + row * BK + ((kc ^ swz(row)) << 2)) =
                        (FAST || TAP || ((wok >> j) & 1u)) ? wreg[j] : f4{0.f, 0.f, 0.f, 0.f};
            }
        };

#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
        if (nk == 0) return;

        const int sw = swz(r16);
        load_tile(0);
        store_tile(0);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int buf = kt & 1;
            const float* xs = Xs + buf * BM * BK + (wm * TM * 16 + r16) * BK;
            const float* ws = Ws + buf * BN * BK + (wn * TN * 16 + r16) * BK;
            // all 16-wide K groups' fragments are requested up front (KG register sets): the second group's LDS reads
            // land while the first group's 64 MFMAs issue, instead of being exposed between the groups
            f4 xa[KG][TM], wb[KG][TN];
#pragma unroll
            for (int g = 0; g < KG; ++g) {
                const int ch = ((4 * g + q) ^ sw) << 2;
#pragma unroll
                for (int i = 0; i < TM; ++i) xa[g][i] = *reinterpret_cast<const f4*>(xs + i * 16 * BK + ch);
#pragma unroll
                for (int j = 0; j < TN; ++j) wb[g][j] = *reinterpret_cast<const f4*>(ws + j * 16 * BK + ch);
            }
            // LDS fragment reads are requested first (they gate the first MFMA); the next tile's global loads issue while
            // those reads are in flight
            if (kt + 1 < nk) load_tile(kt + 1);
            // the MFMAs of a tile run in two halves (k order unchanged) with the next tile's ds_writes in between: they
            // issue in the shadow of the MFMA pipe instead of forming a separate non-MFMA phase in front of the barrier
#pragma unroll
            for (int half = 0; half < 2; ++half) {
#pragma unroll
                for (int s = 0; s < 2 * KG; ++s) {
                    const int step = half * 2 * KG + s;          // 0 .. 4*KG-1 in k order
                    const int g = step >> 2, e = step & 3;
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[g][j][e], xa[g][i][e], acc[i][j], 0, 0, 0);
                }
                if (half == 0 && kt + 1 < nk) store_tile(buf ^ 1);
            }
            __syncthreads();
        }
    }
};

}  // namespace at
