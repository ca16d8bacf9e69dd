// The two-piece fp16 split GEMM (gemm_bf16x3.h, XB_SCHEME_F16X2) for launches that fill the chip: C = epi(A . W^T) with both operands as
// hi + lo fp16 pieces in the K-blocked layout [piece][K/16][rows][16], three products per multiply-add on v_mfma_f32_16x16x32_f16.
//
// Structure ("two groups", measured against the alternatives in tools/f16x2_gemm.hip: 431-468 fp32-equivalent TFLOP/s on the conformer
// shapes = 1.3-1.4 PFLOP/s of issued MFMA on random data, against 295-347 for the register-staged 2-barrier kernel in gemm_bf16x3.hip):
//   * tile 256 x 256, 8 waves (4 x 2, each 64 x 128 = 4 x 8 MFMA tiles of 16 x 16, 128 accumulator registers), one workgroup per CU:
//     waves w and w + 4 share a SIMD;
//   * operands by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write): one ring slot = one 16-wide k-block of all four
//     pieces (32 KB, the global chunks are contiguous 8 KB runs); a K step is 32 = two slots; ring of two pairs (128 KB);
//   * every K step has two segments per wave — L: read the step's fragments (24 ds_read_b128, conflict-free in the linear [rows][16]
//     image: a lane's fragment is k-block lane >> 5, half (lane >> 4) & 1 of row lane & 15) and issue the DMA of the next step;
//     C: 96 MFMAs under s_setprio 1 — each closed by a workgroup barrier, and waves 4-7 run ONE BARRIER BEHIND waves 0-3: while one
//     group multiplies, its SIMD partners load. The matrix pipe alternates between the two waves of a SIMD instead of being fought
//     over and then left idle (MI355X_MICROARCH.md, two waves per SIMD);
//   * the 16x16x32 MFMA shape: same cycles per FLOP as 32x32x16, but the chip holds a higher clock on it under load (+12-20 % here);
//   * hazards: the slot pair of step s + 1 is refilled only after the barrier that closed the trailing group's reads of step s - 1 (every
//     read retires behind s_waitcnt lgkmcnt(0) before its barrier), and the issuing waves wait for their DMA (vmcnt(0)) before the barrier
//     that opens the leading group's reads of step s + 1 (see the loop). A first version let both groups issue in their own segment L and
//     wait after their own C: the trailing group's share then landed one barrier too late and the leaders read stale rows — found by
//     tests/test_ops_gpu.py::test_split_gemm_vs_float64, not by the prototype's single check shape.
// Windowed (conv1d) mode, epilogues and the fp16 range check are those of gemm_bf16x3.hip (split_epilogue.h).
#include "at_common.h"
#include "gemm_bf16x3.h"
#include "split_scheme.h"
#include "split_epilogue.h"
#include <type_traits>
#include <cstdlib>
#include <cstdio>

namespace at {

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

// Two tile shapes with IDENTICAL per-element arithmetic (the same MFMA instruction, k order and product order), so a result does not
// depend on the batch size that picked the shape: TI x TJ = 4 x 8 tiles of 16 x 16 per wave = 256 x 256 per workgroup (128 KB of LDS, one
// workgroup per CU) for launches that fill the chip, 2 x 4 = 128 x 128 (64 KB, two per CU) for small ones (single clips).
template <int TI, int TJ>
struct TgCfg {
    static constexpr int BM = 4 * TI * 16, BN = 2 * TJ * 16;
    static_assert(BM == BN, "square tiles: one chunk size for both operands");
    static constexpr int PIECE = BM * 16;              // 16-bit elements of one (piece, k-block) chunk of BM rows
    static constexpr int SLOT = 4 * PIECE;             // A hi, A lo, W hi, W lo of one k-block
    static constexpr size_t LDS_BYTES = (size_t)4 * SLOT * 2;
    static constexpr int DMA_PER_CHUNK = BM / 128;     // 1-KB pieces (32 rows) of a chunk per leading wave
};

// -DTG_DEBUG_STAMPS (tools/tg_stamps.sh; never in the product build): wave 0 (leading group) and wave 4 (trailing group) of workgroup 0 sum the
// cycle counter over the segments of their K steps; the launcher prints the averages per K step for the first launches of each shape
#ifdef TG_DEBUG_STAMPS
__device__ unsigned long long tg_stamps[2][8];
#define TG_T(i) const unsigned long long tg_t##i = __builtin_readcyclecounter()
#define TG_ACC(k, a_, b_) tg_d[k] += tg_t##b_ - tg_t##a_
#else
#define TG_T(i) do {} while (0)
#define TG_ACC(k, a_, b_) do {} while (0)
#endif

// WINDOWED = false: a plain linear layer (one tap, stride 1): the k-block -> row offset map is a multiplication; true: conv1d windows
// (per k-block two integer divisions on the scalar unit — kept off the linear layers' instruction stream)
// ga > 0: XCD-aware tile order. Workgroups b and b + 8 share an XCD (round-robin dispatch: speed only, never correctness): XCD x gets a
// contiguous range of m-tiles and walks it in groups of `ga` m-tiles x all n-tiles, m fastest — the ~32 tiles an XCD runs at a time then
// share `ga` activation tiles and 32 / ga weight tiles in ITS L2, and an activation tile is fetched into one L2 instead of all eight
// (n-fastest order: 8 x the activation bytes leave the Infinity Cache; the K = 4096 GEMM moved 12.6 GB per launch that way).
template <bool WINDOWED, int TI, int TJ>
__global__ __launch_bounds__(512, (TI * TJ <= 8) ? 4 : 2) void gemm_f16x2_tg_kernel(Bf16x3Args a, int ga) {
    using Cfg = TgCfg<TI, TJ>;
    constexpr int BM = Cfg::BM, TG_PIECE = Cfg::PIECE, TG_SLOT = Cfg::SLOT;
#ifdef TG_DEBUG_STAMPS
    const unsigned long long tg_entry = __builtin_readcyclecounter();
#endif
    typedef SchemeF16x2 SC;
    typedef _Float16 PT;
    typedef f16x8 V8;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    PT* lds = reinterpret_cast<PT*>(lds_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = __builtin_amdgcn_readfirstlane(wave >> 2);     // 0: leading group, 1: trailing group
    const int wm = wave & 3, wn = wave >> 2;                       // SIMD partners own the two column halves of the same 64 rows
    const int ntn = a.N / BM, ntm = a.Mpad / BM;
    int mt, nt;
    if (ga > 0) {
        const int mtn = ntm * a.batch;
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int lo = (int)((long long)mtn * xcd / 8), hi = (int)((long long)mtn * (xcd + 1) / 8);
        const int g = slot / (ga * ntn), base = lo + g * ga;
        const int gn = min(ga, hi - base);
        if (gn <= 0) return;
        const int r = slot - g * ga * ntn;
        if (r >= gn * ntn) return;
        nt = r / gn; mt = base + (r - nt * gn);
    } else {
        nt = blockIdx.x % ntn;   // n fastest: the activation tile is fetched once per row of blocks
        mt = blockIdx.x / ntn;
    }
    const int n0 = nt * BM;
    const int clip = mt / ntm, m0 = (mt - clip * ntm) * BM;
    const int cblocks = a.cblocks > 0 ? a.cblocks : a.K / 16;
    const int Lp = a.Lp > 0 ? a.Lp : a.Mpad;
    const long long a_clip = (long long)cblocks * a.stride * Lp * 16;   // elements of one clip of one piece
    const long long psA = a_clip * a.batch, psW = (long long)a.N * a.K;
    const int nk2 = a.K / 32;
    // DMA: a chunk (one piece of one operand, 256 rows x 32 B) = 8 pieces of 1 KB. Only the LEADING group issues DMA (its waves move rows
    // 64 w .. 64 w + 63 of all four chunks, 16 instructions per K step, in segment L while the trailing group multiplies): measured 2-7 %
    // ahead of sharing the issue between the groups, whose trailing half had to sit in front of its MFMAs (tools/f16x2_gemm.hip, variant S)
    const int srow = (wave & 3) * (BM / 4) + (lane >> 1), shalf = lane & 1;
    const PT* gA = reinterpret_cast<const PT*>(a.A) + clip * a_clip + ((long long)m0 + srow) * 16 + shalf * 8;
    const PT* gW = reinterpret_cast<const PT*>(a.W) + ((long long)n0 + srow) * 16 + shalf * 8;
    // window order of the k-blocks (gemm_bf16x3.h, xb_window_block), walked incrementally: issue() is called for kp = 0, 1, 2, ... in order, so
    // the (plane image, row offset) pair of the next k-block is one compare-and-carry away — no integer division in the loop
    const int w_taps = nk2 * 2 / cblocks, w_q = w_taps / a.stride, w_r = w_taps - w_q * a.stride;
    int w_t = 0, w_off = 0, w_p = 0;                 // plane image cbk * stride + p, row offset, plane
    // issue_A / issue_W: the activation / weight chunks of K step kp (k-blocks 2 kp, 2 kp + 1) -> ring slots 2 pair, 2 pair + 1
    auto issue_A = [&](int kp, int pair) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int kt = 2 * kp + h;
            long long ka;
            if constexpr (WINDOWED) {
                ka = ((long long)w_t * Lp + w_off) * 16;
                if (++w_off == (w_p < w_r ? w_q + 1 : w_q)) {          // next plane; planes beyond the taps (ktaps < stride) hold none
                    w_off = 0; ++w_t; ++w_p;
                    if (w_p == a.stride || (w_q == 0 && w_p == w_r)) { w_t += a.stride - w_p; w_p = 0; }
                }
            } else {
                ka = (long long)kt * Lp * 16;
            }
            PT* s = lds + (2 * pair + h) * TG_SLOT + (wave & 3) * (BM / 4) * 16;  // wave-uniform; the hardware adds lane * 16 B
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int j = 0; j < Cfg::DMA_PER_CHUNK; ++j)              // rows + 32 j: 512 elements further in both images
                    __builtin_amdgcn_global_load_lds((glb_void*)(gA + p * psA + ka + j * 512), (lds_void*)(s + p * TG_PIECE + j * 512), 16, 0, 0);
        }
    };
    auto issue_W = [&](int kp, int pair) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const long long kw = (long long)(2 * kp + h) * a.N * 16;
            PT* s = lds + (2 * pair + h) * TG_SLOT + (wave & 3) * (BM / 4) * 16;
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int j = 0; j < Cfg::DMA_PER_CHUNK; ++j)
                    __builtin_amdgcn_global_load_lds((glb_void*)(gW + p * psW + kw + j * 512), (lds_void*)(s + (2 + p) * TG_PIECE + j * 512), 16, 0, 0);
        }
    };
    const int fr = lane & 15, fq = lane >> 4;
    const int foff = (fq >> 1) * TG_SLOT + fr * 16 + (fq & 1) * 8;      // k-block, row, half
    f4 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
    // (Round 2 tried sharing the DMA issue between the groups — leaders: activation chunks one step ahead; trailers: weight chunks two steps ahead,
    // issued at the end of their read segment — which shortens the leaders' longest segment (in-kernel stamps: 1 800 -> 900 cycles, FFN GEMMs
    // +2.3 %), and a persistent-tile form on top of it (+1 %). With ONE workgroup per CU every test passed; with two co-resident workgroups (the
    // 128 x 128 shape) tiles of the chained acoustic GEMMs came out wrong sporadically — only when the trailing waves issue LDS-DMA, never with the
    // leaders issuing the same chunks. Not understood, so not shipped: tests/test_acoustic_gpu.py::test_repeated_encodes_are_identical caught it.
    // Also measured, same box A/B on semantic_m: issuing the DMA before the fragment reads of the segment (+-0), 2-4 of the weight instructions
    // between the first MFMAs of C instead of in L (-0.2 ... -0.7 %), staggering the CUs' tile phases by up to 15 us (+-0).)
#ifdef TG_DEBUG_STAMPS
    const unsigned long long tg_p0 = __builtin_readcyclecounter();
#endif
    if (grp == 0) { issue_A(0, 0); issue_W(0, 0); }
#ifdef TG_DEBUG_STAMPS
    const unsigned long long tg_p1 = __builtin_readcyclecounter();
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef TG_DEBUG_STAMPS
    const unsigned long long tg_p2 = __builtin_readcyclecounter();
#endif
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();      // the trailing group starts one barrier late
#ifdef TG_DEBUG_STAMPS
    const unsigned long long tg_p3 = __builtin_readcyclecounter();
#endif
#ifdef TG_DEBUG_STAMPS
    unsigned long long tg_d[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    for (int kp = 0; kp < nk2; ++kp) {
        TG_T(0);
        // ---- L -------------------------------------------------------------------------------------------------------------
        const PT* s = lds + (kp & 1) * 2 * TG_SLOT + foff;
        V8 xa[2][TI], wb[2][TJ];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
#pragma unroll
            for (int i = 0; i < TI; ++i) xa[p][i] = *reinterpret_cast<const V8*>(s + p * TG_PIECE + (wm * TI * 16 + i * 16) * 16);
#pragma unroll
            for (int j = 0; j < TJ; ++j) wb[p][j] = *reinterpret_cast<const V8*>(s + (2 + p) * TG_PIECE + (wn * TJ * 16 + j * 16) * 16);
        }
        // DMA of the NEXT step: its slot pair is free once the barrier that closed the trailing group's segment L of step kp - 1 has
        // passed (= the barrier in front of this segment, for the leaders), and it must have landed before the barrier that opens the
        // leading group's segment L of step kp + 1 (= the one that closes their C(kp): they wait there).
        if (grp == 0 && kp + 1 < nk2) { issue_A(kp + 1, (kp + 1) & 1); issue_W(kp + 1, (kp + 1) & 1); }
        __builtin_amdgcn_sched_barrier(0);
        TG_T(1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        TG_T(2);
        __builtin_amdgcn_sched_barrier(0);
        TG_T(3);
        __builtin_amdgcn_s_barrier();
        TG_T(4);
        // ---- C: hi.lo, lo.hi, hi.hi (smallest first) -------------------------------------------------------------------------
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[SC::prod_w(t)][j], xa[SC::prod_a(t)][i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        TG_T(5);
        if (grp == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the leaders' share of step kp + 1 has landed
        __builtin_amdgcn_sched_barrier(0);
        TG_T(6);
        __builtin_amdgcn_s_barrier();
        TG_T(7);
        TG_ACC(0, 0, 1); TG_ACC(1, 1, 2); TG_ACC(2, 2, 3); TG_ACC(3, 3, 4); TG_ACC(4, 4, 5); TG_ACC(5, 5, 6); TG_ACC(6, 6, 7);
    }
#ifdef TG_DEBUG_STAMPS
    const unsigned long long tg_loop_end = __builtin_readcyclecounter();
#endif
    if (grp == 0) __builtin_amdgcn_s_barrier();      // pairs the trailing group's last barrier
    // lane holds, per 16 x 16 tile (i, j): output row m = .. + lane & 15 and the 4 consecutive columns n = .. + 4 (lane >> 4) ..
    XbEpilogue<SC> ep(a, clip);
    // bias quads once per column tile, residual quads one row tile ahead of the stores (split_epilogue.h: inside apply() they would each
    // wait for the previous store)
    auto epilogue = [&](auto mode) {
        constexpr int E = decltype(mode)::value;
        constexpr bool PLAIN = E == XB_EPI_LINEAR || E == XB_EPI_GELU;
        f4 bj[TJ];
#pragma unroll
        for (int j = 0; j < TJ; ++j) bj[j] = ep.load_bias(n0 + wn * TJ * 16 + j * 16 + 4 * fq);
        const int mbase = m0 + wm * TI * 16 + fr;
        f4 rcur[TJ], rnext[TJ];
        auto load_row = [&](int i, f4 (&r)[TJ]) {
            const int m = mbase + i * 16;
#pragma unroll
            for (int j = 0; j < TJ; ++j)
                r[j] = (PLAIN && m < a.M) ? ep.load_residual(m, n0 + wn * TJ * 16 + j * 16 + 4 * fq) : f4{0.f, 0.f, 0.f, 0.f};
        };
        if constexpr (PLAIN) load_row(0, rcur);
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            const int m = mbase + i * 16;
            if constexpr (PLAIN) {
                if (i + 1 < TI) load_row(i + 1, rnext);
            }
            if (m < a.M) {
#pragma unroll
                for (int j = 0; j < TJ; ++j) ep.template apply_with<E>(m, n0 + wn * TJ * 16 + j * 16 + 4 * fq, acc[i][j], bj[j], rcur[j]);
            }
            if constexpr (PLAIN) {
#pragma unroll
                for (int j = 0; j < TJ; ++j) rcur[j] = rnext[j];
            }
        }
    };
    switch (a.epi) {
        case XB_EPI_SWISH_SPLIT: epilogue(std::integral_constant<int, XB_EPI_SWISH_SPLIT>{}); break;
        case XB_EPI_GELU_SPLIT: epilogue(std::integral_constant<int, XB_EPI_GELU_SPLIT>{}); break;
        case XB_EPI_ELU_SPLIT: epilogue(std::integral_constant<int, XB_EPI_ELU_SPLIT>{}); break;
        case XB_EPI_GLU: epilogue(std::integral_constant<int, XB_EPI_GLU>{}); break;
        case XB_EPI_GELU: epilogue(std::integral_constant<int, XB_EPI_GELU>{}); break;
        case XB_EPI_QKV: epilogue(std::integral_constant<int, XB_EPI_QKV>{}); break;
        case XB_EPI_RAW_ELU_SPLIT2: epilogue(std::integral_constant<int, XB_EPI_RAW_ELU_SPLIT2>{}); break;
        default: epilogue(std::integral_constant<int, XB_EPI_LINEAR>{}); break;
    }
    ep.finish();
#ifdef TG_DEBUG_STAMPS
    if (blockIdx.x == (gridDim.x * 3) / 4 && (tid == 0 || tid == 256)) {
        for (int i = 0; i < 7; ++i) tg_stamps[grp][i] = tg_d[i];
        tg_stamps[grp][7] = ((__builtin_readcyclecounter() - tg_loop_end) << 32) | ((tg_loop_end - tg_entry) & 0xffffffffull);   // epilogue | entry..loop end
        tg_stamps[grp][2] = ((tg_p0 - tg_entry) << 48) | ((tg_p1 - tg_p0) << 32) | ((tg_p2 - tg_p1) << 16) | ((tg_p3 - tg_p2) & 0xffff);   // prologue: setup | issue | DMA wait | barriers
    }
#endif
}

// eligibility: the fp16 scheme, whole 128-column tiles, K steps of 32, row padding of 256
bool gemm_f16x2_tg_eligible(const Bf16x3Args& a) {
    if (a.scheme != XB_SCHEME_F16X2) return false;
    return a.N % 128 == 0 && a.K % 32 == 0 && a.Mpad % 256 == 0;
}

template <bool WINDOWED, int TI, int TJ>
static int launch_tg(const Bf16x3Args& a, int ga, dim3 grid, hipStream_t stream) {
    using Cfg = TgCfg<TI, TJ>;
    { static LdsAttrFlags lds_attr; if (int rc = set_max_dynamic_lds(lds_attr, gemm_f16x2_tg_kernel<WINDOWED, TI, TJ>, Cfg::LDS_BYTES)) return rc; }
    hipLaunchKernelGGL((gemm_f16x2_tg_kernel<WINDOWED, TI, TJ>), grid, dim3(512), Cfg::LDS_BYTES, stream, a, ga);
    AT_CHECK_HIP(hipGetLastError());
#ifdef TG_DEBUG_STAMPS
    {
        static int printed = 0;
        if (printed < 12) {
            ++printed;
            (void)hipStreamSynchronize(stream);
            unsigned long long hbuf[2][8];
            (void)hipMemcpyFromSymbol(hbuf, HIP_SYMBOL(tg_stamps), sizeof(hbuf));
            const double nk = a.K / 32;
            for (int g = 0; g < 2; ++g) {
                std::fprintf(stderr, "tg stamps M %d N %d K %d tile %d %s: issue %.0f  lds-wait %.0f  (unused) %.0f  barrierA %.0f  mfma %.0f  dma-wait %.0f  barrierB %.0f (cycles per K step)\n",
                             a.M, a.N, a.K, TI * 64, g ? "trailers" : "leaders ", hbuf[g][0] / nk, hbuf[g][1] / nk, hbuf[g][2] / nk, hbuf[g][3] / nk,
                             hbuf[g][4] / nk, hbuf[g][5] / nk, hbuf[g][6] / nk);
                std::fprintf(stderr, "    whole tile: entry -> end of K loop %llu cycles (%d K steps), epilogue %llu; prologue: set-up %llu  issue %llu  DMA wait %llu  barriers %llu\n",
                             hbuf[g][7] & 0xffffffffull, (int)nk, hbuf[g][7] >> 32, hbuf[g][2] >> 48, (hbuf[g][2] >> 32) & 0xffff, (hbuf[g][2] >> 16) & 0xffff, hbuf[g][2] & 0xffff);
            }
        }
    }
#endif
    return 0;
}

int launch_gemm_f16x2_tg(const Bf16x3Args& a, hipStream_t stream) {
    // 256 x 256 tiles when they fill the chip, else 128 x 128 (4 x as many workgroups, two per CU): same arithmetic per element
    static const long long min_tiles = std::getenv("AUDIOTOKEN_F16X2_TG_MIN_TILES") ? std::atoll(std::getenv("AUDIOTOKEN_F16X2_TG_MIN_TILES")) : 256;
    const bool big = a.N % 256 == 0 && (long long)a.batch * (a.Mpad / 256) * (a.N / 256) >= min_tiles;
    const int bm = big ? 256 : 128;
    const int ntn = a.N / bm, mtn = a.batch * (a.Mpad / bm);
    static const int xcdmap = std::getenv("AUDIOTOKEN_XB_XCDMAP") ? std::atoi(std::getenv("AUDIOTOKEN_XB_XCDMAP")) : 1;
    int ga = 0;
    dim3 grid((unsigned)((long long)mtn * ntn));
    if (xcdmap && big && mtn >= 64) {
        ga = ntn >= 8 ? 4 : (32 / ntn > 16 ? 16 : 32 / ntn);
        const int per_xcd = ((mtn + 7) / 8 + ga - 1) / ga * ga * ntn;   // upper bound of one XCD's slots incl. the padding of its last group
        grid = dim3((unsigned)(8 * per_xcd));
    }
    const bool windowed = a.stride != 1 || (a.cblocks > 0 && a.cblocks != a.K / 16);
    if (big) return windowed ? launch_tg<true, 4, 8>(a, ga, grid, stream) : launch_tg<false, 4, 8>(a, ga, grid, stream);
    return windowed ? launch_tg<true, 2, 4>(a, ga, grid, stream) : launch_tg<false, 2, 4>(a, ga, grid, stream);
}

}  // namespace at
