// The two-piece fp16 split GEMM (gemm_bf16x3.h, XB_SCHEME_F16X2): C = epi(A . W^T) with both operands as hi + lo fp16 pieces in the K-blocked
// layout [piece][K/16][rows][16], three products per multiply-add on v_mfma_f32_16x16x32_f16.
//
// Structure ("two groups", measured against the alternatives in tools/f16x2_gemm.hip):
//   * tile 256 x 256, 8 waves (4 x 2, each 64 x 128 = 4 x 8 MFMA tiles of 16 x 16, 128 accumulator registers), one workgroup per CU:
//     waves w and w + 4 share a SIMD;
//   * operands by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write; the global chunks are contiguous 8 KB runs). A K step
//     is 32 = two 16-wide k-blocks of both pieces of both operands. LDS (160 KB): an ACTIVATION ring of two K steps (2 x 32 KB) and a WEIGHT
//     ring of THREE (3 x 32 KB);
//   * every K step has two segments per wave — L: read the step's fragments (24 ds_read_b128, conflict-free in the linear [rows][16]
//     image: a lane's fragment is k-block lane >> 5, half (lane >> 4) & 1 of row lane & 15) and issue DMA; C: 96 MFMAs under s_setprio 1 —
//     each closed by a workgroup barrier, and waves 4-7 run ONE BARRIER BEHIND waves 0-3: while one group multiplies, its SIMD partners
//     load. The matrix pipe alternates between the two waves of a SIMD instead of being fought over and then left idle;
//   * the DMA issue is SHARED between the groups (round 3): the leaders move the activation chunks of step kp + 1 in their L(kp) and wait for
//     them after their C(kp); the trailers move the weight chunks of step kp + 2 at the end of their L(kp) and wait for them one period
//     later at the same place — both before the barrier that opens the leaders' L of that step. The leaders' L (the longest segment of the
//     loop: 24 reads + 16 DMA instructions, 1 800 cycles against 1 600 of MFMAs) carries 8 DMA instructions, the trailers' L (which had
//     ~1 000 cycles of slack) the other 8;
//   * the 16x16x32 MFMA shape: same cycles per FLOP as 32x32x16, but the chip holds a higher clock on it under load (+12-20 % here);
//   * PERSISTENT tiles (round 3): a workgroup walks its XCD's tile list; the first chunks of the next tile are issued BEFORE the epilogue of
//     the current one (the whole ring is free behind the loop's last barrier), so a tile no longer starts with set-up + issue + DMA wait.
//
// Hazards of the ring (cdna_hip_programming.md, "Read a staged buffer one phase AFTER the wait that retires it"; MI355X_MICROARCH.md item 7:
// nothing orders a ds_read against an LDS-DMA write except the issuing wave's vmcnt plus a barrier):
//   RAW  a chunk is waited for (vmcnt(0) of the issuing waves) BEFORE a barrier that every reader passes before its reads;
//   WAR  a ring slot is refilled only after a barrier that closed the LAST segment in which ANY wave read it. s_waitcnt lgkmcnt(0) retires
//        the issuing wave's own reads only. Round 2's shared-issue variant broke exactly this rule: the trailers refilled the weight pair of
//        step kp in their own L(kp) after their own lgkmcnt(0), while the other three trailing waves could still be reading it (they share
//        the same 128 weight rows). With one workgroup per CU the waves' skew never exceeded the DMA's flight time; with two co-resident
//        workgroups (the 128 x 128 shape) a sibling held back by the other workgroup's s_setprio(1) MFMA segment read rows of step kp + 2:
//        the "sporadically wrong tiles" of round 2. Reduced reproducer: tools/lds_dma_war.hip (profiles/r03_lds_dma_war.jsonl: protocol 1
//        fails 1 launch in 60 at two workgroups per CU, every launch once a trailing wave is delayed, never a leading one; protocols 0 and
//        2 never). The fix is the third weight pair: the pair refilled in the trailers' L(kp) is the one of step kp - 1, whose last reads
//        (the trailers' L(kp - 1)) are two barriers back.
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
// depend on the batch size that picked the shape: TI x TJ = 4 x 8 tiles of 16 x 16 per wave = 256 x 256 per workgroup (160 KB of LDS, one
// workgroup per CU) for launches that fill the chip, 2 x 4 = 128 x 128 (80 KB, two per CU) for small ones (single clips).
template <int TI, int TJ>
struct TgCfg {
    static constexpr int BM = 4 * TI * 16, BN = 2 * TJ * 16;
    static_assert(BM == BN, "square tiles: one chunk size for both operands");
    static constexpr int PIECE = BM * 16;              // 16-bit elements of one (piece, k-block) chunk of BM rows
    static constexpr int KB = 2 * PIECE;               // one k-block of one operand: hi, lo
    static constexpr int PAIR = 2 * KB;                // one K step of one operand
    static constexpr int A_PAIRS = 2, W_PAIRS = 3;
    static constexpr size_t LDS_BYTES = (size_t)(A_PAIRS + W_PAIRS) * PAIR * 2;
    static constexpr int DMA_PER_CHUNK = BM / 128;     // 1-KB pieces (32 rows) of a chunk per issuing wave
    static constexpr int WG_PER_CU = (TI * TJ <= 8) ? 2 : 1;
};

// -DTG_DEBUG_STAMPS (tools/tg_stamps.sh; never in the product build): wave 0 (leading group) and wave 4 (trailing group) of one workgroup sum the
// cycle counter over the segments of the K steps of their FIRST tile; the launcher prints the averages per K step for the first launches of each shape
#ifdef TG_DEBUG_STAMPS
__device__ unsigned long long tg_stamps[2][16];
#define TG_T(i) const unsigned long long tg_t##i = __builtin_readcyclecounter()
#define TG_ACC(k, a_, b_) tg_d[k] += tg_t##b_ - tg_t##a_
#else
#define TG_T(i) do {} while (0)
#define TG_ACC(k, a_, b_) do {} while (0)
#endif

// WINDOWED = false: a plain linear layer (one tap, stride 1): the k-block -> row offset map is a multiplication; true: conv1d windows
// ga > 0: XCD-aware tile order. Workgroups b and b + 8 share an XCD (round-robin dispatch: speed only, never correctness): XCD x owns a
// contiguous range of m-tiles and walks it in groups of `ga` m-tiles x all n-tiles, m fastest — the ~32 tiles an XCD runs at a time then
// share `ga` activation tiles and 32 / ga weight tiles in ITS L2, and an activation tile is fetched into one L2 instead of all eight
// (n-fastest order: 8 x the activation bytes leave the Infinity Cache; the K = 4096 GEMM moved 12.6 GB per launch that way). Workgroup b
// takes the entries b >> 3, (b >> 3) + gridDim.x / 8, ... of its XCD's list. ga == 0: tiles blockIdx.x, + gridDim.x, ... n fastest.
// the divisors of the tile walk as FastDivU (at_common.h): ga * ntn, ntn, ntm, ga
struct TgSched { FastDivU gantn, ntn, ntm, ga; };

template <bool WINDOWED, int TI, int TJ>
__global__ __launch_bounds__(512, (TI * TJ <= 8) ? 4 : 2) void gemm_f16x2_tg_kernel(Bf16x3Args a, int ga, TgSched sd) {
    using Cfg = TgCfg<TI, TJ>;
    constexpr int BM = Cfg::BM, TG_PIECE = Cfg::PIECE, TG_KB = Cfg::KB, TG_PAIR = Cfg::PAIR;
#ifdef TG_DEBUG_STAMPS
    const unsigned long long tg_entry = __builtin_readcyclecounter();
#endif
    typedef SchemeF16x2 SC;
    typedef _Float16 PT;
    typedef f16x8 V8;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    PT* ldsA = reinterpret_cast<PT*>(lds_raw);
    PT* ldsW = ldsA + Cfg::A_PAIRS * TG_PAIR;
    // The lane index is re-derived (v_mbcnt, volatile: not hoisted) where it is needed instead of living across the K loop and the epilogue: at 256
    // registers per wave the persistent loop has none to spare (the build fails on any scratch use: Makefile `check`)
    auto lane_id = []() { int l; asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l)); return l; };
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int grp = wave >> 2;                                     // 0: leading group, 1: trailing group
    const int wm = wave & 3, wn = wave >> 2;                       // SIMD partners own the two column halves of the same 64 rows
    const int ntn = a.N / BM, ntm = a.Mpad / BM;
    const int mtn = ntm * a.batch;
    const int xcd = blockIdx.x & 7;
    const int x_lo = (int)((long long)mtn * xcd / 8), x_hi = (int)((long long)mtn * (xcd + 1) / 8);
    auto tile_at = [&](int it, int& mt, int& nt) -> bool {
        if (ga > 0) {
            const int q = (int)(blockIdx.x >> 3) + it * (int)(gridDim.x >> 3);
            if (q >= (x_hi - x_lo) * ntn) return false;
            const int g = (int)sd.gantn.div((unsigned)q), r = q - g * ga * ntn, base = x_lo + g * ga;
            const int gn = min(ga, x_hi - base);
            nt = gn == ga ? (int)sd.ga.div((unsigned)r) : r / gn;   // (the generic division only in an XCD's last, partial group)
            mt = base + (r - nt * gn);
        } else {
            const int t = (int)blockIdx.x + it * (int)gridDim.x;
            if (t >= mtn * ntn) return false;
            mt = (int)sd.ntn.div((unsigned)t);   // n fastest: the activation tile is fetched once per row of blocks
            nt = t - mt * ntn;
        }
        return true;
    };
    const int cblocks = a.cblocks > 0 ? a.cblocks : a.K / 16;
    const int Lp = a.Lp > 0 ? a.Lp : a.Mpad;
    const long long a_clip = (long long)cblocks * a.stride * Lp * 16;   // elements of one clip of one piece
    const long long psA = a_clip * a.batch, psW = (long long)a.N * a.K;
    const int nk2 = a.K / 32;
    // DMA: a chunk (one piece of one k-block of one operand, BM rows x 32 B) = BM / 32 pieces of 1 KB; issuing wave w & 3 moves rows
    // (BM / 4) (w & 3) .. of it. The source pointers are those of the tile whose chunks are being issued (the NEXT tile during an epilogue).
    // (wave-uniform base in scalar registers + ONE per-lane element offset shared by both operands: the loads take the saddr + voffset form)
    unsigned lane_el;
    { const int l = lane_id(); lane_el = (unsigned)(l >> 1) * 16u + (unsigned)(l & 1) * 8u; }
    const PT* uA = nullptr;
    const PT* uW = nullptr;
    // window order of the k-blocks (gemm_bf16x3.h, xb_window_block), walked incrementally: issue_A() is called for kp = 0, 1, 2, ... in order, so
    // the (plane image, row offset) pair of the next k-block is one compare-and-carry away — no integer division in the loop
    const int w_taps = nk2 * 2 / cblocks, w_q = w_taps / a.stride, w_r = w_taps - w_q * a.stride;
    int w_t = 0, w_off = 0, w_p = 0;                 // plane image cbk * stride + p, row offset, plane
    auto begin_stream = [&](int mt, int nt) {
        const int clip = (int)sd.ntm.div((unsigned)mt), m0 = (mt - clip * ntm) * BM;
        uA = reinterpret_cast<const PT*>(a.A) + clip * a_clip + ((long long)m0 + (wave & 3) * (BM / 4)) * 16;
        uW = reinterpret_cast<const PT*>(a.W) + ((long long)nt * BM + (wave & 3) * (BM / 4)) * 16;
        w_t = 0; w_off = 0; w_p = 0;
    };
    // issue_A / issue_W: the activation / weight chunks of K step kp (k-blocks 2 kp, 2 kp + 1) -> ring pair `pair` of that operand
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
            PT* s = ldsA + pair * TG_PAIR + h * TG_KB + (wave & 3) * (BM / 4) * 16;  // wave-uniform; the hardware adds lane * 16 B
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int j = 0; j < Cfg::DMA_PER_CHUNK; ++j)              // rows + 32 j: 512 elements further in both images
                    __builtin_amdgcn_global_load_lds((glb_void*)(uA + p * psA + ka + j * 512 + lane_el), (lds_void*)(s + p * TG_PIECE + j * 512), 16, 0, 0);
        }
    };
    auto issue_W = [&](int kp, int pair) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const long long kw = (long long)(2 * kp + h) * a.N * 16;
            PT* s = ldsW + pair * TG_PAIR + h * TG_KB + (wave & 3) * (BM / 4) * 16;
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int j = 0; j < Cfg::DMA_PER_CHUNK; ++j)
                    __builtin_amdgcn_global_load_lds((glb_void*)(uW + p * psW + kw + j * 512 + lane_el), (lds_void*)(s + p * TG_PIECE + j * 512), 16, 0, 0);
        }
    };
    // the first chunks of a tile: A(0), W(0) by the leaders, W(1) by the trailers (into pairs 0, 0, 1: every tile restarts the rings — the
    // whole LDS is free behind the last barrier of the previous tile's loop)
    auto issue_first = [&]() {
        if (grp == 0) { issue_A(0, 0); issue_W(0, 0); }
        else if (nk2 > 1) issue_W(1, 1);
    };
    RangeMax over;
    int mt, nt;
    if (!tile_at(0, mt, nt)) return;
    begin_stream(mt, nt);
#ifdef TG_DEBUG_STAMPS
    const unsigned long long tg_p0 = __builtin_readcyclecounter();
    unsigned long long tg_d[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tg_loop_begin = 0, tg_loop_end = 0, tg_epi_end = 0, tg_next_begin = 0;
    unsigned long long tg_all_loop = 0, tg_all_epi = 0, tg_all_gap = 0, tg_tiles = 0, tg_mark = tg_p0;   // sums over ALL tiles of this workgroup
#endif
    issue_first();
    for (int it = 0;; ++it) {
        const int e_clip = (int)sd.ntm.div((unsigned)mt), e_m0 = (mt - e_clip * ntm) * BM, e_n0 = nt * BM;
        f4 acc[TI][TJ];
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
        int foff;                                          // fragment offset of this lane: k-block, row, half
        { const int l = lane_id(); foff = (l >> 5) * TG_KB + (l & 15) * 16 + ((l >> 4) & 1) * 8; lane_el = (unsigned)(l >> 1) * 16u + (unsigned)(l & 1) * 8u; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the tile's first chunks (and the previous tile's stores)
        __builtin_amdgcn_s_barrier();
        if (grp == 1) __builtin_amdgcn_s_barrier();        // the trailing group starts one barrier late
#ifdef TG_DEBUG_STAMPS
        if (it == 0) tg_loop_begin = __builtin_readcyclecounter();
        if (it == 1) tg_next_begin = __builtin_readcyclecounter();
        { const unsigned long long t = __builtin_readcyclecounter(); tg_all_gap += t - tg_mark; tg_mark = t; }
#endif
        int wpair = 0;                                     // weight ring pair of step kp = kp % 3
        for (int kp = 0; kp < nk2; ++kp) {
            TG_T(0);
            // ---- L -------------------------------------------------------------------------------------------------------------
            const PT* sa = ldsA + (kp & 1) * TG_PAIR + foff;
            const PT* sw = ldsW + wpair * TG_PAIR + foff;
            V8 xa[2][TI], wb[2][TJ];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
#pragma unroll
                for (int i = 0; i < TI; ++i) xa[p][i] = *reinterpret_cast<const V8*>(sa + p * TG_PIECE + (wm * TI * 16 + i * 16) * 16);
#pragma unroll
                for (int j = 0; j < TJ; ++j) wb[p][j] = *reinterpret_cast<const V8*>(sw + p * TG_PIECE + (wn * TJ * 16 + j * 16) * 16);
            }
            const int wnext = wpair == 2 ? 0 : wpair + 1;
            // Leaders: A(kp + 1) into activation pair (kp + 1) & 1 — last read in the trailers' L(kp - 1), closed by the barrier in front of this
            // segment; awaited after C(kp), before the barrier that opens the leaders' L(kp + 1).
            if (grp == 0 && kp + 1 < nk2) issue_A(kp + 1, (kp + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            TG_T(1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            TG_T(2);
            // Trailers: W(kp + 1) (issued one period ago) has landed before the barrier below, which opens the leaders' L(kp + 1); then W(kp + 2)
            // into weight pair (kp + 2) % 3 = the pair of step kp - 1, last read in the trailers' L(kp - 1) two barriers back (NOT the pair this
            // segment reads: see WAR above).
            if (grp == 1) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (kp + 2 < nk2) issue_W(kp + 2, wnext == 2 ? 0 : wnext + 1);
            }
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
            if (grp == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the leaders' A(kp + 1) has landed
            __builtin_amdgcn_sched_barrier(0);
            TG_T(6);
            __builtin_amdgcn_s_barrier();
            TG_T(7);
#ifdef TG_DEBUG_STAMPS
            if (it == 0) { TG_ACC(0, 0, 1); TG_ACC(1, 1, 2); TG_ACC(2, 2, 3); TG_ACC(3, 3, 4); TG_ACC(4, 4, 5); TG_ACC(5, 5, 6); TG_ACC(6, 6, 7); }
#endif
            wpair = wnext;
        }
        if (grp == 0) __builtin_amdgcn_s_barrier();      // pairs the trailing group's last barrier: every read of the rings has retired
#ifdef TG_DEBUG_STAMPS
        if (it == 0) tg_loop_end = __builtin_readcyclecounter();
        { const unsigned long long t = __builtin_readcyclecounter(); tg_all_loop += t - tg_mark; tg_mark = t; }
#endif
        // the next tile's first chunks fly during this tile's epilogue
        int mt2 = 0, nt2 = 0;
        const bool more = tile_at(it + 1, mt2, nt2);
        if (more) { begin_stream(mt2, nt2); issue_first(); }
        // lane holds, per 16 x 16 tile (i, j): output row m = .. + lane & 15 and the 4 consecutive columns n = .. + 4 (lane >> 4) ..
        XbEpilogue<SC, true> ep(a, e_clip);
        const int el = lane_id(), fr = el & 15, fq = el >> 4;
        // bias quads once per column tile; residual quads one row tile ahead of the stores (split_epilogue.h: inside apply() each load would sit
        // behind the previous quad's store)
        // FULL: every row of the tile is below M (all tiles but a clip's last m-tile): no per-row predicate. With the predicate every store sat
        // behind a conditional branch, the compiler's wait-count pass gave up counting across them and put s_waitcnt vmcnt(0) in front of EVERY
        // store — each one waited for the acknowledgement of the one before it (~1 k cycles x 32 stores per lane = the 31-50 k-cycle residual
        // epilogues of the stamps); straight-line code keeps the stores in flight behind counted waits.
        auto epilogue = [&](auto mode, auto ph1, auto full_, auto res_) {
            constexpr int E = decltype(mode)::value;
            constexpr bool PH1 = decltype(ph1)::value;
            constexpr bool FULL = decltype(full_)::value;
            constexpr bool RES = decltype(res_)::value;    // a residual operand exists (uniform; a per-load null check is a branch per load)
            constexpr bool PLAIN = E == XB_EPI_LINEAR || E == XB_EPI_GELU;
            // ROW LAYOUT (256 x 256 tiles; the modes whose outputs are row-major: fp32 C, the q rows and k / v row-major pieces of the fused
            // projection, GLU): in the MFMA layout a lane owns 4 consecutive columns of ONE row per 16 x 16 tile, so a wave-instruction touches 16
            // rows x 64 B (32 B for the 8-byte forms) — half cache lines, 16 DRAM pages per instruction. In-kernel stamps (tools/tg_stamps.sh, all
            // tiles of a workgroup) gave 43-50 k cycles for the residual epilogue of a tile against 14-23 k for the modes that write contiguous
            // K-blocked pieces. Each wave therefore transposes one 16-row tile at a time through its own 8 KB of LDS (the activation pair 1 and weight
            // pair 2 are free while the next tile's first chunks land in pairs 0 / 0, 1): 8 ds_write_b128 in the MFMA layout, 8 ds_read_b128 with
            // lane l -> row 2 r + (l >> 5), columns 4 (l & 31) .. + 3, i.e. two rows x 512 contiguous bytes per instruction for the residual load and
            // the store. 16-byte chunks are XOR-swizzled with row & 7: conflict-free for both forms (writes: 8 contiguous lanes = 8 rows of one chunk;
            // reads: the lane groups of MI355X_MICROARCH.md stay distinct mod 16 under an XOR < 8). Wave-local: LDS operations of one wave execute
            // in order, no barrier. The arithmetic per element is unchanged (same operands, same order): bit-identical to the MFMA-layout path.
            constexpr bool ROWLAYOUT = TI * TJ > 8 && (PLAIN || E == XB_EPI_QKV || E == XB_EPI_GLU);
            if constexpr (ROWLAYOUT) {
                static_assert(TJ == 8, "row layout: 128-column wave tiles");
                float* wl = reinterpret_cast<float*>((wave < 4 ? ldsA + TG_PAIR : ldsW + 2 * TG_PAIR) + (wave & 3) * 4096);   // 8 KB per wave
                const int rrow = el >> 5, rc = el & 31;                                   // row-layout coordinates of this lane
                const int ncol = e_n0 + wn * TJ * 16 + rc * 4;
                const f4 bias4 = ep.load_bias(ncol);
                const int mrow0 = e_m0 + wm * TI * 16 + rrow;                              // + i * 16 + 2 r
                // Residual quads. The memory counter retires loads AND stores in issue order (MI355X_MICROARCH.md: "a load's data waits for every
                // older one of them"), so a residual load issued behind a row tile's stores is only usable once those stores have been acknowledged:
                // with the loads one (half) row tile ahead every row tile paid a store round trip + a load round trip (stamps: 39-50 k cycles per tile).
                // Here row tile 0 is requested first and row tile i + 1 right after the accumulators of row tile i have gone to LDS — whose 32
                // registers it takes over — BEFORE that row tile's stores: no load ever waits behind the stores of the row tile in front of it.
                if constexpr (PLAIN) {
                    // C / R addressed as (tile origin: scalar registers) + (32-bit lane offset inside the tile, < 256 rows x ldc): no 64-bit vector
                    // address arithmetic, one address register per access
                    const char* rb = ep.Rb ? reinterpret_cast<const char*>(ep.Rb + (long long)e_m0 * a.ldr + e_n0) : nullptr;
                    char* cb = reinterpret_cast<char*>(ep.Cb + (long long)e_m0 * a.ldc + e_n0);
                    const unsigned rowl = (unsigned)(wm * TI * 16 + rrow), coll = (unsigned)(wn * TJ * 16 + rc * 4);
                    const unsigned offC = (rowl * (unsigned)a.ldc + coll) * 4u, offR = (rowl * (unsigned)a.ldr + coll) * 4u;
                    const unsigned stepC = (unsigned)a.ldc * 8u, stepR = (unsigned)a.ldr * 8u;      // two rows further
                    f4 res[TI][8];
                    auto load_res = [&](int i) {
#pragma unroll
                        for (int k = 0; k < 8; ++k)
                            res[i][k] = (RES && (FULL || mrow0 + i * 16 + 2 * k < a.M)) ? *reinterpret_cast<const f4*>(rb + (offR + (unsigned)(i * 8 + k) * stepR)) : f4{0.f, 0.f, 0.f, 0.f};
                    };
                    load_res(0);
#pragma unroll
                    for (int i = 0; i < TI; ++i) {
#pragma unroll
                        for (int j = 0; j < TJ; ++j)
                            *reinterpret_cast<f4*>(wl + fr * 128 + (((j * 4 + fq) ^ (fr & 7)) << 2)) = acc[i][j];
                        if (i + 1 < TI) load_res(i + 1);
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            const int row = 2 * k + rrow;
                            const f4 q = *reinterpret_cast<const f4*>(wl + row * 128 + ((rc ^ (row & 7)) << 2));
                            if (FULL || mrow0 + i * 16 + 2 * k < a.M)
                                *reinterpret_cast<f4*>(cb + (offC + (unsigned)(i * 8 + k) * stepC)) = ep.template plain_value<E>(q, bias4, res[i][k]);
                        }
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < TI; ++i) {
#pragma unroll
                        for (int j = 0; j < TJ; ++j)
                            *reinterpret_cast<f4*>(wl + fr * 128 + (((j * 4 + fq) ^ (fr & 7)) << 2)) = acc[i][j];
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            const int row = 2 * k + rrow, m = mrow0 + i * 16 + 2 * k;
                            const f4 q = *reinterpret_cast<const f4*>(wl + row * 128 + ((rc ^ (row & 7)) << 2));
                            if (FULL || m < a.M) ep.template apply_with<E, false>(m, ncol, q, bias4, f4{0.f, 0.f, 0.f, 0.f});
                        }
                    }
                }
                return;
            }
            constexpr int NCH = 1, H = TJ / NCH;   // residual chunks per row tile (register path: the 128 x 128 shape and the K-blocked piece outputs)
            f4 bj[TJ];
#pragma unroll
            for (int j = 0; j < TJ; ++j) bj[j] = ep.load_bias(e_n0 + wn * TJ * 16 + j * 16 + 4 * fq);
            const int mbase = e_m0 + wm * TI * 16 + fr;
            f4 rcur[H], rnext[H];
            auto load_half = [&](int idx, f4 (&r)[H]) {
                const int m = mbase + (idx / NCH) * 16, j0 = (idx % NCH) * H;
#pragma unroll
                for (int j = 0; j < H; ++j)
                    r[j] = (PLAIN && RES && (FULL || m < a.M)) ? ep.load_residual(m, e_n0 + wn * TJ * 16 + (j0 + j) * 16 + 4 * fq) : f4{0.f, 0.f, 0.f, 0.f};
            };
            if constexpr (PLAIN) load_half(0, rcur);
#pragma unroll
            for (int idx = 0; idx < NCH * TI; ++idx) {
                const int m = mbase + (idx / NCH) * 16, j0 = (idx % NCH) * H;
                if constexpr (PLAIN) {
                    if (idx + 1 < NCH * TI) load_half(idx + 1, rnext);
                }
                if (FULL || m < a.M) {
#pragma unroll
                    for (int j = 0; j < H; ++j)
                        ep.template apply_with<E, PH1>(m, e_n0 + wn * TJ * 16 + (j0 + j) * 16 + 4 * fq, acc[idx / NCH][j0 + j], bj[j0 + j], rcur[j]);
                }
                if constexpr (PLAIN) {
#pragma unroll
                    for (int j = 0; j < H; ++j) rcur[j] = rnext[j];
                }
            }
        };
        using T_ = std::true_type;
        using F_ = std::false_type;
        const bool ph1 = a.Sphases == 1;   // the plain linear layers' piece outputs: no phase planes, no integer division per quad
        const bool full = e_m0 + BM <= a.M;
        const bool has_res = a.R != nullptr;
        auto run = [&](auto mode, auto p1) {
            constexpr int E_ = decltype(mode)::value;
            if constexpr (E_ == XB_EPI_LINEAR || E_ == XB_EPI_GELU) {
                if (has_res) { if (full) epilogue(mode, p1, T_{}, T_{}); else epilogue(mode, p1, F_{}, T_{}); }
                else { if (full) epilogue(mode, p1, T_{}, F_{}); else epilogue(mode, p1, F_{}, F_{}); }
            } else {
                if (full) epilogue(mode, p1, T_{}, F_{}); else epilogue(mode, p1, F_{}, F_{});
            }
        };
        switch (a.epi) {
            case XB_EPI_SWISH_SPLIT: if (ph1) run(std::integral_constant<int, XB_EPI_SWISH_SPLIT>{}, T_{}); else run(std::integral_constant<int, XB_EPI_SWISH_SPLIT>{}, F_{}); break;
            case XB_EPI_GELU_SPLIT: if (ph1) run(std::integral_constant<int, XB_EPI_GELU_SPLIT>{}, T_{}); else run(std::integral_constant<int, XB_EPI_GELU_SPLIT>{}, F_{}); break;
            case XB_EPI_ELU_SPLIT: run(std::integral_constant<int, XB_EPI_ELU_SPLIT>{}, F_{}); break;
            case XB_EPI_GLU: run(std::integral_constant<int, XB_EPI_GLU>{}, F_{}); break;
            case XB_EPI_GELU: run(std::integral_constant<int, XB_EPI_GELU>{}, F_{}); break;
            case XB_EPI_QKV: run(std::integral_constant<int, XB_EPI_QKV>{}, F_{}); break;
            case XB_EPI_RAW_ELU_SPLIT2: run(std::integral_constant<int, XB_EPI_RAW_ELU_SPLIT2>{}, F_{}); break;
            default: run(std::integral_constant<int, XB_EPI_LINEAR>{}, F_{}); break;
        }
        over |= ep.over;
#ifdef TG_DEBUG_STAMPS
        if (it == 0) tg_epi_end = __builtin_readcyclecounter();
        { const unsigned long long t = __builtin_readcyclecounter(); tg_all_epi += t - tg_mark; tg_mark = t; ++tg_tiles; }
#endif
        if (!more) break;
        mt = mt2; nt = nt2;
    }
    range_publish(a.status, a.status ? a.status + 1 : nullptr, over);
#ifdef TG_DEBUG_STAMPS
    if (blockIdx.x == (gridDim.x * 3) / 4 && (threadIdx.x == 0 || threadIdx.x == 256)) {
        for (int i = 0; i < 7; ++i) tg_stamps[grp][i] = tg_d[i];
        // epilogue | entry..loop end ; first tile's start-up | gap between the first tile's epilogue end and the second tile's first K step
        tg_stamps[grp][7] = ((tg_epi_end - tg_loop_end) << 32) | ((tg_loop_end - tg_entry) & 0xffffffffull);
        tg_stamps[grp][8] = ((tg_loop_begin - tg_p0) << 32) | ((tg_next_begin > tg_epi_end ? tg_next_begin - tg_epi_end : 0) & 0xffffffffull);
        tg_stamps[grp][9] = tg_all_loop; tg_stamps[grp][10] = tg_all_epi; tg_stamps[grp][11] = tg_all_gap; tg_stamps[grp][12] = tg_tiles;
        tg_stamps[grp][13] = __builtin_readcyclecounter() - tg_entry;
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
    Bf16x3Args b = a;
    b.fdS = FastDivU((unsigned)(a.Sphases > 0 ? a.Sphases : 1));
    b.fdS2 = FastDivU((unsigned)(a.S2phases > 0 ? a.S2phases : 1));
    const int ntn = a.N / Cfg::BM, ntm = a.Mpad / Cfg::BM;
    TgSched sd;
    sd.gantn = FastDivU((unsigned)((ga > 0 ? ga : 1) * ntn)); sd.ntn = FastDivU((unsigned)ntn); sd.ntm = FastDivU((unsigned)ntm); sd.ga = FastDivU((unsigned)(ga > 0 ? ga : 1));
    hipLaunchKernelGGL((gemm_f16x2_tg_kernel<WINDOWED, TI, TJ>), grid, dim3(512), Cfg::LDS_BYTES, stream, b, ga, sd);
    AT_CHECK_HIP(hipGetLastError());
#ifdef TG_DEBUG_STAMPS
    {
        static int printed = 0;
        if (printed < 12) {
            ++printed;
            (void)hipStreamSynchronize(stream);
            unsigned long long hbuf[2][16];
            (void)hipMemcpyFromSymbol(hbuf, HIP_SYMBOL(tg_stamps), sizeof(hbuf));
            const double nk = a.K / 32;
            for (int g = 0; g < 2; ++g) {
                std::fprintf(stderr, "tg stamps M %d N %d K %d epi %d tile %d grid %u %s: issue %.0f  lds-wait %.0f  dma-issue(trailers) %.0f  barrierA %.0f  mfma %.0f  dma-wait %.0f  barrierB %.0f (cycles per K step)\n",
                             a.M, a.N, a.K, a.epi, TI * 64, grid.x, g ? "trailers" : "leaders ", hbuf[g][0] / nk, hbuf[g][1] / nk, hbuf[g][2] / nk, hbuf[g][3] / nk,
                             hbuf[g][4] / nk, hbuf[g][5] / nk, hbuf[g][6] / nk);
                std::fprintf(stderr, "    first tile: entry -> end of K loop %llu cycles (%d K steps), epilogue %llu; start-up (first issue -> first K step) %llu; epilogue end -> next tile's first K step %llu\n",
                             hbuf[g][7] & 0xffffffffull, (int)nk, hbuf[g][7] >> 32, hbuf[g][8] >> 32, hbuf[g][8] & 0xffffffffull);
                const double nt_ = (double)(hbuf[g][12] ? hbuf[g][12] : 1);
                std::fprintf(stderr, "    all %llu tiles of this workgroup, per tile: K loop %.0f  epilogue %.0f  start (wait for first chunks + barriers) %.0f; workgroup lifetime %llu cycles\n",
                             hbuf[g][12], hbuf[g][9] / nt_, hbuf[g][10] / nt_, hbuf[g][11] / nt_, hbuf[g][13]);
            }
        }
    }
#endif
    return 0;
}

int launch_gemm_f16x2_tg(const Bf16x3Args& a, hipStream_t stream) {
    // 256 x 256 tiles when they fill the chip, else 128 x 128 (4 x as many workgroups, two per CU): same arithmetic per element
    const long long min_tiles = 256;
    const bool big = a.N % 256 == 0 && (long long)a.batch * (a.Mpad / 256) * (a.N / 256) >= min_tiles;
    const int bm = big ? 256 : 128;
    const int ntn = a.N / bm, mtn = a.batch * (a.Mpad / bm);
    const long long tiles = (long long)mtn * ntn;
    const int resident = device_cus() * (big ? 1 : 2);          // persistent workgroups the chip holds at once
    int ga = 0;
    dim3 grid((unsigned)(tiles < resident ? tiles : resident));
    if (big && mtn >= 64) {
        ga = ntn >= 8 ? 4 : (32 / ntn > 16 ? 16 : 32 / ntn);
        const long long per_xcd_tiles = (long long)((mtn + 7) / 8) * ntn;        // the longest XCD list
        const long long per_xcd = per_xcd_tiles < resident / 8 ? per_xcd_tiles : resident / 8;
        grid = dim3((unsigned)(8 * per_xcd));
    }
    const bool windowed = a.stride != 1 || (a.cblocks > 0 && a.cblocks != a.K / 16);
    if (big) return windowed ? launch_tg<true, 4, 8>(a, ga, grid, stream) : launch_tg<false, 4, 8>(a, ga, grid, stream);
    return windowed ? launch_tg<true, 2, 4>(a, ga, grid, stream) : launch_tg<false, 2, 4>(a, ga, grid, stream);
}

}  // namespace at
