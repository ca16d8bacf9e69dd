// SEANet stage 1 in ONE kernel: the 64-channel residual block (seanet_res64x3.hip) and the strided conv that consumes it
// (64 -> 128 channels, k = 8, stride 4: seanet_down64x3.hip), so that the block output — with 7.86 GB per 256 x 10 s batch the largest tensor of the
// path — never reaches HBM: 256 B in per row, 128 B out (instead of 256 in + 256 out + 256 in + 128 out for the pair).
//
// Both kernels keep their weights in registers (72 per lane for the block, 128 for the conv), which does not fit one wave. So the workgroup is
// ROLE-SPLIT, 12 waves = 3 per SIMD at 168 registers each:
//   waves 8..11 (block role):  conv3 -> h -> tail -> ELU -> split -> R pieces in LDS, one 64-row tile per iteration
//   waves 0..7  (conv role):   16 output channels each, the 16 outputs of the PREVIOUS tile from its R pieces (16 K steps x 3 products), and the
//                              staging of the NEXT tile's input, half a tile at a time: split(ELU(x)) / split(x) into the other X buffer while the
//                              other half's loads fly (one float4 per thread in flight)
// One iteration = two workgroup barriers (h ready / R ready + next X staged); the conv waves cut their 16 K steps in two at the first one. The
// block role's chain conv3 -> h -> tail is the critical path of an iteration (one workgroup per CU: nothing else hides its latency), so everything
// that is not on it — half of the vector work (the input ELUs and splits) — runs in the eight conv waves, whose matrix work is shorter.
// A workgroup walks a CONTIGUOUS run of tiles, so the conv's causal context (the last 4 block rows of the previous tile) is carried in LDS: the tail
// writes rows 60..63 a second time into the next R buffer (three buffers: the conv role still reads the previous one). At a clip start the context is
// the reflect padding (rows 1..4 mirrored), written by the same lanes; a run that starts inside a clip first runs the block role alone over the
// tile before it (one extra tile per run).
// Arithmetic: per output element the same products in the same order as seanet_res64x3_kernel<SchemeF16x2> followed by
// seanet_down64x3_kernel<SchemeF16x2> — results are bit-identical to that pair (tests/test_acoustic_gpu.py::test_fused_stage1_is_bit_identical).
// fp16 scheme only: with three bf16 pieces the conv's weights alone are 192 registers (the bf16x3 fallback keeps the two kernels).
#include "gemm_core.h"
#include "encodec_kernels.h"
#include "split_scheme.h"

namespace at {

namespace {

constexpr int RD_TT = 64;                    // block rows per tile = 16 conv outputs
constexpr int RD_XROWS = 66;                 // x rows staged: row i <-> time t0 - 2 + i
constexpr int RD_LDX = 80;                   // as seanet_res64x3.hip
constexpr int RD_LDH = 32;                   // h rows are DENSE (64 B) and XOR-swizzled by row: rd_hoff below (round 6)
constexpr int RD_XP = RD_XROWS * RD_LDX, RD_HP = RD_TT * RD_LDH;
constexpr int RD_PL = 17, RD_LDR = 80;       // R: local row i (time t0 - 4 + i, 68 rows) in plane i & 3 at index i >> 2 (seanet_down64x3.hip's layout)
constexpr int RD_PPAD = 8;                   // + 16 B per plane (round 6): see rd_off
constexpr int RD_PS = RD_PL * RD_LDR + RD_PPAD;
constexpr int RD_RP = 4 * RD_PS;             // elements of one piece of one R buffer
constexpr int RD_NBUF = 3;
constexpr int RD_XBUF = 2 * 2 * RD_XP;         // elements of one X buffer: [Xe | Xr][2 pieces]
constexpr int RD_LDS_BYTES = (2 * RD_XBUF + 2 * RD_HP + RD_NBUF * 2 * RD_RP) * 2;   // 158 336 of 163 840
constexpr int RD_THREADS = 768;

// LDS bank arithmetic (round 6; tools/lds_bank_sim.py reproduces the PMC conflict share of round 5, 0.20, from these address functions):
// * R stores: the 16 lanes of a ds_write_b64 group hold 16 consecutive block rows of one column = 4 planes x 4 indices. With planes 17 x 160 B apart both the plane
//   and the index step were = 8 dwords mod 32 banks: 4 addresses per bank, 16 LDS cycles per store instead of 4. 16 B more per plane make the plane step = 12 mod 32:
//   (2 i + 3 p) mod 8 takes every value twice — 2-way, the floor for 8-byte stores of 16 different 16-byte-aligned rows. The conv role's fragment reads stay inside
//   one plane (consecutive indices, 10 slots apart): conflict-free as before.
// * h: [64 rows][32 channels] fp16. Rows 80 B apart (5 slots, odd) made the tail's ds_read_b128 groups — 8 rows at chunk q, 8 rows at chunk q + 1 — collide 2-way.
//   Now dense 64-byte rows with the 16-byte chunk index XORed by (2 if row & 4) ^ (3 if row & 8): reads conflict-free, the epilogue's stores 2-way (as before).
__device__ __forceinline__ int rd_off(int row) { return (row & 3) * RD_PS + (row >> 2) * RD_LDR; }
// (the swizzle term depends on row bits 2, 3 only: for a fragment row 16 k + r16 it is a per-lane constant — rd_hswz(r16) — and the kernel adds it as one)
__device__ __forceinline__ int rd_hswz(int row) { return ((row >> 1) & 2) ^ (((row >> 3) & 1) * 3); }
__device__ __forceinline__ int rd_hoff(int row, int chunk) { return row * RD_LDH + ((chunk ^ rd_hswz(row)) << 3); }

// Input staging of a tile, half `half` (0 / 1): thread tid of the conv role owns the float4 chunk c = 512 half + tid of the tile's 64 NEW rows
// (local row 2 + c / 16 <-> time t0 + c / 16, float4 c % 16). Local rows 0, 1 (times t0 - 2, t0 - 1) are the previous tile's rows 64, 65: copied
// inside LDS by the block role, or — first tile of a clip — the reflect padding, which the owners of rows 3 and 4 write as a second copy.
__device__ __forceinline__ f4 rd_load(const float* x, int tile, int tiles_per_clip, int L, int half, int tid) {
    const int b = tile / tiles_per_clip;
    const int t0 = (tile - b * tiles_per_clip) * RD_TT;
    int tau = t0 + half * 32 + (tid >> 4);
    tau = tau > L - 1 ? L - 1 : tau;       // rows past the end only feed outputs that are never stored
    return *reinterpret_cast<const f4*>(x + (long long)b * L * 64 + (unsigned)(tau * 64 + (tid & 15) * 4));
}
// ... split into the X buffer at Xe ([split(ELU(x)) | split(x)][2 pieces]) at local row `row`; returns the range maximum of the chunk
__device__ __forceinline__ float rd_stage_row(SchemeF16x2::T* Xe, int row, int c4, const f4& v, float sa, bool mirror) {
    typedef SchemeF16x2 SC;
    SC::T* Xr = Xe + SC::NP * RD_XP;
    const int off = row * RD_LDX + c4 * 4, o2 = (4 - row) * RD_LDX + c4 * 4;   // o2: rows 3, 4 (times 1, 2) of a clip's first tile -> rows 1, 0
    float over;
    {   // the raw copy first, then the ELU copy: one set of pieces live at a time (the conv role has ~40 registers beside its weights)
        SC::V4 rp[SC::NP];
        over = split4<SC>(v, sa, rp);
#pragma unroll
        for (int i = 0; i < SC::NP; ++i) {
            *reinterpret_cast<SC::V4*>(Xr + i * RD_XP + off) = rp[i];
            if (mirror) *reinterpret_cast<SC::V4*>(Xr + i * RD_XP + o2) = rp[i];
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    {
        const f4 e = {elu1(v.x), elu1(v.y), elu1(v.z), elu1(v.w)};
        SC::V4 ep[SC::NP];
        split4<SchemeNoCheck<SC>>(e, sa, ep);      // |ELU(x)| <= max(|x|, 1): covered by the check of x
#pragma unroll
        for (int i = 0; i < SC::NP; ++i) {
            *reinterpret_cast<SC::V4*>(Xe + i * RD_XP + off) = ep[i];
            if (mirror) *reinterpret_cast<SC::V4*>(Xe + i * RD_XP + o2) = ep[i];
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    return over;
}

}  // namespace

__global__ __launch_bounds__(RD_THREADS, 1) void seanet_res64down_kernel(ResDown64Args a) {
    typedef SchemeF16x2 SC;
    typedef SC::T PT;
    typedef SC::V8 V8;
    typedef SC::V4 V4;
    constexpr int NP = SC::NP;
    extern __shared__ __attribute__((aligned(16))) unsigned char rd_lds_raw[];
    PT* Xs = reinterpret_cast<PT*>(rd_lds_raw);   // [2 buffers][split(ELU(x)) | split(x)][NP]
    PT* Hs = Xs + 2 * RD_XBUF;                    // split(ELU(conv3 + b3))
    PT* Rs = Hs + NP * RD_HP;                     // [3 buffers][NP][4 planes][17][80]: split(ELU(block output))
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, q = lane >> 4;
    const int L = a.L, Lo = L / 4;
    const int tiles_per_clip = (L + RD_TT - 1) / RD_TT;
    const long long total_tiles = (long long)a.B * tiles_per_clip;   // < 2^30: checked by the launcher
    // this workgroup's run of tiles [beg, end); the block role starts one tile earlier when the run begins inside a clip
    const int beg = (int)(total_tiles * blockIdx.x / gridDim.x), end = (int)(total_tiles * (blockIdx.x + 1) / gridDim.x);
    if (end <= beg) return;
    const int first = beg - (beg % tiles_per_clip != 0 ? 1 : 0);
    const int n_iter = end - first + 1;
    const float sa = a.act_scale;
    // iteration `it`: block role on tile first + it (X buffer it & 1 -> R buffer it % 3), conv role on tile first + it - 1 (R buffer (it - 1) % 3)
    // and staging tile first + it + 1 into X buffer (it + 1) & 1

    if (wave >= 8) {
        // =========================================== block role (seanet_res64x3.hip) ===========================================
        const int rw = wave - 8;
        __builtin_amdgcn_s_setprio(2);   // this role's chain (conv3 -> h -> tail) is the critical path of every iteration
        const int cn = rw & 1, mh = rw >> 1;        // conv3: channel tile, row half
        const float rs3 = 1.0f / (a.act_scale * a.w3_scale), rst = 1.0f / (a.act_scale * a.wt_scale);
        RangeMax over_h, over_out;
        V8 w3p[NP][6], wtp[NP][3];
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
        for (int ks = 0; ks < 6; ++ks) {
            V8 t[NP];
            wsplit(a.w3 + (cn * 16 + r16) * 192 + ks * 32 + q * 8, a.w3_scale, t);
#pragma unroll
            for (int i = 0; i < NP; ++i) w3p[i][ks] = t[i];
        }
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            V8 t[NP];
            wsplit(a.wt + (rw * 16 + r16) * 96 + ks * 32 + q * 8, a.wt_scale, t);
#pragma unroll
            for (int i = 0; i < NP; ++i) wtp[i][ks] = t[i];
        }
        const f4 b3 = *reinterpret_cast<const f4*>(a.b3 + cn * 16 + q * 4);
        const f4 bt = *reinterpret_cast<const f4*>(a.bt + rw * 16 + q * 4);
        // h offsets inside a row: rows are 16 k + r16, so the swizzle is this lane's constant
        const int h_st = r16 * RD_LDH + (((cn * 2 + (q >> 1)) ^ rd_hswz(r16)) << 3) + (q & 1) * 4;   // epilogue store: channels 16 cn + 4 q .. + 3
        const int h_ld = r16 * RD_LDH + ((q ^ rd_hswz(r16)) << 3);                                   // tail fragment: chunk q
        RangeMax over_x;
        if ((first % tiles_per_clip) != 0 && tid < 512 + 32) {
            // a run that starts inside a clip: local rows 0, 1 of its first tile from memory (32 lanes, prologue only)
            const int l = tid - 512;
            const int b = first / tiles_per_clip;
            const int tau = (first - b * tiles_per_clip) * RD_TT - 2 + (l >> 4);
            const f4 v = *reinterpret_cast<const f4*>(a.x + (long long)b * L * 64 + (unsigned)(tau * 64 + (l & 15) * 4));
            over_x |= rd_stage_row(Xs, l >> 4, l & 15, v, sa, false);
        }
        __syncthreads();   // X buffer 0 staged (prologue)
        int buf = 0;
        for (int it = 0; it < n_iter; ++it) {
            const int tile = first + it;
            const bool live = tile < end;
            const int t0 = (tile % tiles_per_clip) * RD_TT;
            const PT* Xe = Xs + (it & 1) * RD_XBUF;
            const PT* Xr = Xe + NP * RD_XP;
            if (wave == 8 && tile + 1 < end && (tile + 1) % tiles_per_clip != 0) {
                // local rows 0, 1 of the next tile = rows 64, 65 of this one: 64 lanes x 16 B = [Xe | Xr][2 pieces][2 rows][8 chunks]
                const int o = (lane >> 4) * RD_XP + ((lane >> 3) & 1) * RD_LDX + (lane & 7) * 8;
                *reinterpret_cast<V8*>(Xs + ((it + 1) & 1) * RD_XBUF + o) = *reinterpret_cast<const V8*>(Xs + (it & 1) * RD_XBUF + o + RD_TT * RD_LDX);
            }
            if (live) {
                // h[32 mh .. + 31, 16 cn .. + 15] = ELU(conv3(ELU(x)) + b3): output row j uses x rows j, j+1, j+2; ks = (tap, 32 channels)
                f4 acc[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
                auto xread = [&](int ks, V8 (&xf)[NP][2]) {
                    const int tap = ks >> 1, chunk = (ks & 1) * 4 + q;
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        const PT* src = Xe + (32 * mh + 16 * m + r16 + tap) * RD_LDX + chunk * 8;
#pragma unroll
                        for (int p = 0; p < NP; ++p) xf[p][m] = *reinterpret_cast<const V8*>(src + p * RD_XP);
                    }
                };
                // this chain is the critical path of the iteration: the next K step's fragments are in flight during the MFMAs of the current one
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
                    over_h |= split4<SC>(e, sa, hp);
                    const int off = (32 * mh + 16 * m) * RD_LDH + h_st;
#pragma unroll
                    for (int i = 0; i < NP; ++i) *reinterpret_cast<V4*>(Hs + i * RD_HP + off) = hp[i];
                }
            }
            __syncthreads();   // B: h ready
            if (live) {
                // out[:, 16 rw .. + 15] = ELU([h | x] . [W1 | Wsc]^T + (b1 + bsc)): output row j uses h row j and x row j + 2; written as the conv's
                // operand: pieces of local row j + 4 of R[buf]; rows 60..63 also as rows 0..3 of the next buffer, rows 1..4 of a clip's first tile
                // also mirrored into rows 3..0 (reflect padding)
                PT* Rc = Rs + buf * (NP * RD_RP);
                PT* Rn = Rs + (buf == RD_NBUF - 1 ? 0 : buf + 1) * (NP * RD_RP);
#pragma unroll 1
                for (int mp = 0; mp < 4; mp += 2) {
                    f4 acc[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
                    V8 xf[3][NP][2];   // all three K steps' fragments up front
#pragma unroll
                    for (int ks = 0; ks < 3; ++ks)
#pragma unroll
                        for (int m = 0; m < 2; ++m) {
                            const int row = 16 * (mp + m) + r16;
                            const PT* src = ks == 0 ? Hs + 16 * (mp + m) * RD_LDH + h_ld : Xr + (row + 2) * RD_LDX + ((ks - 1) * 4 + q) * 8;
                            const int ps = ks == 0 ? RD_HP : RD_XP;
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
                        const int j = (mp + m) * 16 + r16;
                        const f4 v = acc[m] * rst + bt;
                        const f4 o = {elu1(v.x), elu1(v.y), elu1(v.z), elu1(v.w)};
                        V4 op[NP];
                        over_out |= split4<SC>(o, sa, op);
                        const int col = rw * 16 + q * 4;
                        const int off = rd_off(j + 4) + col;
#pragma unroll
                        for (int i = 0; i < NP; ++i) *reinterpret_cast<V4*>(Rc + i * RD_RP + off) = op[i];
                        if (j >= RD_TT - 4) {
                            const int o2 = rd_off(j - (RD_TT - 4)) + col;
#pragma unroll
                            for (int i = 0; i < NP; ++i) *reinterpret_cast<V4*>(Rn + i * RD_RP + o2) = op[i];
                        }
                        if (t0 == 0 && j >= 1 && j <= 4) {
                            const int o2 = rd_off(4 - j) + col;
#pragma unroll
                            for (int i = 0; i < NP; ++i) *reinterpret_cast<V4*>(Rc + i * RD_RP + o2) = op[i];
                        }
                    }
                }
            }
            __syncthreads();   // C: R[buf] complete, the next X buffer staged; Hs free
            buf = buf == RD_NBUF - 1 ? 0 : buf + 1;
        }
        over_h |= over_x;
        range_publish(a.status_res, a.status_res ? a.status_res + 1 : nullptr, over_h);
        range_publish(a.status_down, a.status_down ? a.status_down + 1 : nullptr, over_out);
    } else {
        // ============================== conv role (seanet_down64x3.hip) + input staging of the block ==============================
        const float rs = 1.0f / (a.act_scale * a.wd_scale);
        RangeMax over_x;
        V8 wr[NP][16];
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const float* src = a.wd + (wave * 16 + r16) * 512 + ks * 32 + q * 8;
            const f4 lo = *reinterpret_cast<const f4*>(src), hi = *reinterpret_cast<const f4*>(src + 4);
            V4 plo[NP], phi[NP];
            split4<SchemeNoCheck<SC>>(lo, a.wd_scale, plo);
            split4<SchemeNoCheck<SC>>(hi, a.wd_scale, phi);
#pragma unroll
            for (int i = 0; i < NP; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) { wr[i][ks][k] = plo[i][k]; wr[i][ks][4 + k] = phi[i][k]; }
        }
        // input staging (rd_load / rd_stage_row): ONE chunk in flight per thread — half a tile is split while the other half's loads fly.
        // The thread index is re-derived where it is needed (v_mbcnt, volatile: not hoisted) instead of living across the loop as half a dozen
        // lane-constant offsets: this role has ~40 registers beside its weights and the build fails on any scratch use (Makefile `check`)
        auto tid_now = [&]() { int l; asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l)); return wave * 64 + l; };
        f4 pre;
        auto stage_half = [&](int tile, int half, PT* Xe) {
            const int t = tid_now();
            const int row = 2 + half * 32 + (t >> 4);
            over_x |= rd_stage_row(Xe, row, t & 15, pre, sa, half == 0 && (tile % tiles_per_clip) == 0 && (row == 3 || row == 4));
        };
        pre = rd_load(a.x, first, tiles_per_clip, L, 0, tid_now());
        stage_half(first, 0, Xs);
        pre = rd_load(a.x, first, tiles_per_clip, L, 1, tid_now());
        stage_half(first, 1, Xs);
        if (first + 1 < end) pre = rd_load(a.x, first + 1, tiles_per_clip, L, 0, tid_now());
        __syncthreads();   // X buffer 0 staged
        int buf = RD_NBUF - 1;   // the R buffer of the previous iteration's tile
        for (int it = 0; it < n_iter; ++it) {
            const int tile = first + it - 1;
            const bool live = it >= 1 && tile >= beg;
            const PT* X = Rs + buf * (NP * RD_RP);
            f4 acc = f4{0.f, 0.f, 0.f, 0.f};
            auto ksteps = [&](auto k0c, auto k1c) {
                constexpr int K0 = decltype(k0c)::value, K1 = decltype(k1c)::value;
                if (live) {
                    int fbase;
                    { const int l = tid_now() & 63; fbase = (l & 15) * RD_LDR + (l >> 4) * 8; }
#pragma unroll
                    for (int ks = K0; ks < K1; ++ks) {
                        const int tap = ks >> 1, ch = (ks & 1) * 4;
                        const PT* src = X + fbase + rd_off(tap) + ch * 8;   // = rd_off(4 r16 + tap) + (ch + q) * 8
                        V8 xa[NP];
#pragma unroll
                        for (int p = 0; p < NP; ++p) xa[p] = *reinterpret_cast<const V8*>(src + p * RD_RP);
#pragma unroll
                        for (int t = 0; t < SC::NPROD; ++t) acc = SC::mfma16(wr[SC::prod_w(t)][ks], xa[SC::prod_a(t)], acc);
                    }
                }
            };
            // the block role's next tile, half by half: split what was loaded half an iteration ago into the other X buffer, load the next half
            const int nt = first + it + 1;
            PT* Xn = Xs + ((it + 1) & 1) * RD_XBUF;
            if (nt < end) {
                stage_half(nt, 0, Xn);
                pre = rd_load(a.x, nt, tiles_per_clip, L, 1, tid_now());
            }
            ksteps(std::integral_constant<int, 0>{}, std::integral_constant<int, 8>{});
            __syncthreads();   // B
            if (nt < end) {
                stage_half(nt, 1, Xn);
                if (nt + 1 < end) pre = rd_load(a.x, nt + 1, tiles_per_clip, L, 0, tid_now());
            }
            ksteps(std::integral_constant<int, 8>{}, std::integral_constant<int, 16>{});
            if (live) {
                const int b = tile / tiles_per_clip;
                const int l = tid_now() & 63;
                const int u = (tile - b * tiles_per_clip) * (RD_TT / 4) + (l & 15);
                if (u < Lo) *reinterpret_cast<f4*>(a.out + ((long long)b * Lo + u) * 128 + wave * 16 + (l >> 4) * 4) = acc * rs + *reinterpret_cast<const f4*>(a.bd + wave * 16 + (l >> 4) * 4);
            }
            __syncthreads();   // C
            buf = buf == RD_NBUF - 1 ? 0 : buf + 1;
        }
        range_publish(a.status_res, a.status_res ? a.status_res + 1 : nullptr, over_x);
    }
}

int launch_seanet_res64down(const ResDown64Args& a, hipStream_t stream) {
    AT_REQUIRE(a.L >= 8 && a.L % 4 == 0 && a.B >= 1, "res64down: L % 4 == 0, L >= 8");
    AT_REQUIRE(a.act_scale > 0.f && a.w3_scale > 0.f && a.wt_scale > 0.f && a.wd_scale > 0.f, "res64down: the fp16 scheme needs its scales");
    const long long tiles = (long long)a.B * ((a.L + RD_TT - 1) / RD_TT);
    AT_REQUIRE(tiles < (1LL << 30) && (long long)a.L * 64 < (1LL << 30), "tile / offset arithmetic is 32-bit");
    const int cus = device_cus();
    const int grid = (int)(tiles < cus ? tiles : cus);
    { static LdsAttrFlags lds_attr; if (int rc = set_max_dynamic_lds(lds_attr, seanet_res64down_kernel, RD_LDS_BYTES)) return rc; }
    hipLaunchKernelGGL(seanet_res64down_kernel, dim3(grid), dim3(RD_THREADS), RD_LDS_BYTES, stream, a);
    AT_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace at
