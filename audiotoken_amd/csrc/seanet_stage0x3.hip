// SEANet encoder stage 0 on the bf16 matrix cores — the fused kernel of seanet_stage0.hip
//   waveform -> conv0 (1->32, k7) -> residual block (ELU, k3 32->16, ELU, k1 16->32, + k1 shortcut) -> ELU -> strided conv (32->64, k4 s2)
// with the three 32/16-channel contractions as exact 3-way bf16 splits on v_mfma_f32_16x16x32_bf16 (six per 32-wide K step;
// arithmetic and accuracy: gemm_bf16x3.hip). conv0 (K = 7 taps) stays on the fp32 MFMA exactly as in the fp32 kernel.
// Tile = 60 input samples (30 outputs; SX_ADV below): 64 rows of the block = 4 MFMA row tiles, one per wave (conv3 + tail per wave on its own rows,
// no barrier in between); the strided conv is split over the OUTPUT channels instead (wave = 16 channels, all 32 output slots), so a
// wave's weights are 132 registers of bf16 pieces and 65 KB of LDS per workgroup put TWO workgroups on a CU — one splits / applies
// ELUs (vector work) while the other multiplies; a bf16 MFMA leaves half of its cycles to vector issue.
// LDS (bf16 pieces, rows padded by 16 B so that fragment reads of 16 consecutive rows are conflict-free with linear addresses):
//   X0e = split(ELU(x0)): [pieces][66 rows][32 + 16]   row i <-> time t0 - 4 + i (64 rows from conv0 + 2 zero rows)
//   Hs  = split(ELU(conv3 + b3)):          [3][64][16 + 8]        row j <-> time t0 - 2 + j
//   Rs  = split(ELU(block output)):        [3][2 planes][34][32 + 8]: row j in plane j & 1 at j >> 1, so that the stride-2 rows of
//         the strided conv's fragment reads are consecutive.
// The tail's K = 48 = [h 16 | x0 32] runs as two K steps of 32 with a zero-weight quarter.
// Rounds differently from the fp32 chain: compared by tolerance and identical tokens (tests/test_acoustic_gpu.py).
// Causal reflect padding as in seanet_stage0.hip (conv0 evaluated at |t|, two mirrored rows of the block output); N % 2 == 0.
#include "gemm_core.h"
#include "encodec_kernels.h"
#include "gemm_bf16x3.h"
#include "split_scheme.h"
#include <cstdlib>

namespace at {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// Tile = 60 input samples (30 outputs): the block needs x0 rows 0 .. ADV + 3 = 64 rows = exactly FOUR MFMA row tiles of conv0, 8 (row tile, channel
// tile) units = two per wave. (Rounds 1-2: 62 samples / 31 outputs needed 66 rows = five row tiles, 10 units: waves 0 and 1 ran three units while
// waves 2 and 3 waited at the barrier after two, and 14 of the 80 rows were never used — a third of the conv0 phase for one more output per tile.)
// Rows 64, 65 of the x0 image (read by conv3 rows 62, 63, whose results only feed the two masked output slots) are zeroed once per workgroup.
constexpr int SX_ADV = 60;                 // input samples per tile
constexpr int SX_UO = 30;                  // outputs per tile
constexpr int SX_XUNITS = 8;               // conv0 units: 4 row tiles x 2 channel tiles
constexpr int SX_XROWS = 66;               // x0 rows: 64 written by conv0 + 2 zero rows
constexpr int SX_ROWS = 64;                // h / r rows
constexpr int SX_RIDX = 34;                // r rows per parity plane (the masked 32nd output reads rows 62..65)
constexpr int SX_LDX = 48, SX_LDH = 24, SX_LDR = 48;   // x0 / r rows + 32 B (conflict-free fragment reads, see seanet_res128x3.hip); h rows keep + 16 B (two workgroups per CU)
constexpr int SX_XP = SX_XROWS * SX_LDX, SX_HP = SX_ROWS * SX_LDH, SX_RP = 2 * SX_RIDX * SX_LDR;   // elements per piece
constexpr int SX_WAV = 88;                 // waveform segment: Wv[s] = wav[|t0 - 10 + s|]

// 4 consecutive channels * scale -> the NP pieces at element offset `off` (piece stride ps); returns the scheme's range verdict
template <class SC>
__device__ __forceinline__ float sx_store4(typename SC::T* base, int off, int ps, const f4& v, float scale) {
    typename SC::V4 p[SC::NP];
    const float over = split4<SC>(v, scale, p);
#pragma unroll
    for (int i = 0; i < SC::NP; ++i) *reinterpret_cast<typename SC::V4*>(base + i * ps + off) = p[i];
    return over;
}
// Write / read swizzle of the x0 and r images: 16-byte chunk c of a row lives at c ^ bit. The 8-byte piece stores of the 16 lanes of a store group go to
// 16 rows with one column; with rows 96 B apart, rows 4 apart share their 32 banks (ds_write_b64 banks repeat every 128 B: MI355X_MICROARCH.md, LDS)
// — a 4-way conflict on every store of conv0 and of the block's tail (PMC, round 2: 41 % of this kernel's LDS cycles were conflict replays). Flipping
// the chunk for every other group of 4 rows halves that (the other half needs an 8-byte-granular swizzle, which would split the 16-byte fragment
// reads); the fragment reads stay conflict-free: their lane groups {0-3,12-15,20-27} see, per column, rows whose flipped and unflipped halves
// cover disjoint bank octets.
__device__ __forceinline__ int sx_swz(int col, int bit) { return col ^ ((bit & 1) << 3); }
__device__ __forceinline__ int sx_xbit(int row) { return row >> 2; }                 // x0 image: row i
__device__ __forceinline__ int sx_rbit(int row) { return row ^ (row >> 3); }         // r image: plane (row & 1) ^ bit 2 of the in-plane index (row >> 1)
__device__ __forceinline__ f4 sx_elu4(const f4& v) { return f4{elu1(v.x), elu1(v.y), elu1(v.z), elu1(v.w)}; }

// SC = operand scheme of the three split contractions (conv3, tail, strided conv; conv0 stays on the fp32 MFMA): three bf16 pieces / six products,
// or (round 2, default) two fp16 pieces / three products with power-of-two activation / weight scales and a range status bit (seanet_res64x3.hip).
template <class SC>
__global__ __launch_bounds__(256, SC::NP == 2 ? 3 : 2) void seanet_stage0x3_kernel(Stage0Args a) {
    typedef typename SC::T PT;
    typedef typename SC::V8 V8;
    typedef typename SC::V4 V4;
    constexpr int NP = SC::NP;
    extern __shared__ __attribute__((aligned(16))) unsigned char sx_lds_raw[];
    PT* X0e = reinterpret_cast<PT*>(sx_lds_raw);   // split(ELU(x0)); the raw x0 is never stored (the shortcut is folded into conv0: Stage0Args::wsc0)
    PT* Hs = X0e + NP * SX_XP;
    PT* Rs = Hs + NP * SX_HP;
    float* Wv = reinterpret_cast<float*>(Rs + NP * SX_RP);
    const float sa = SC::RANGE_CHECK ? a.act_scale : 1.0f;
    const float sw3 = SC::RANGE_CHECK ? a.w3_scale : 1.0f, swt = SC::RANGE_CHECK ? a.wt_scale : 1.0f, swd = SC::RANGE_CHECK ? a.wd_scale : 1.0f;
    const float rs3 = 1.0f / (sa * sw3), rst = 1.0f / (sa * swt), rsd = 1.0f / (sa * swd);
    RangeMax over;
    float* B0s = Wv + 2 * SX_WAV;                    // conv0 bias [32]  (Wv: two segments, alternating tiles)
    float* Bs = B0s + 32;                            // b3 [16] | bt [32] | bd [64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, q = lane >> 4;
    const int N = a.N, L1 = N / 2;
    const int tiles_per_clip = (N + SX_ADV - 1) / SX_ADV;
    const int total_tiles = a.B * tiles_per_clip;   // < 2^30: checked by the launcher

    // ---- weights -> registers, once per workgroup ------------------------------------------------------------------
    for (int e = tid; e < NP * 2 * SX_LDX; e += 256) {   // x0 rows 64, 65 of every piece: zeros (see SX_ADV)
        const int p = e / (2 * SX_LDX), r = e - p * (2 * SX_LDX);
        X0e[p * SX_XP + 64 * SX_LDX + r] = (PT)0.f;
    }
    if (tid < 32) B0s[tid] = a.b0[tid];
    if (tid < 16) Bs[tid] = a.b3[tid];
    if (tid < 32) Bs[16 + tid] = a.bsc0[tid];   // Wsc . b0 + (b1 + bsc)
    if (tid < 64) Bs[48 + tid] = a.bd[tid];
    float w0f[2][2];   // conv0 as a K = 8 fp32 MFMA (7 taps + a zero column): A fragment w0f[nt][s] = W0[nt*16 + r16][4s + q]
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) w0f[nt][ks] = (4 * ks + q) < 7 ? a.w0[(nt * 16 + r16) * 7 + 4 * ks + q] : 0.f;
    float wscf[2][2];  // the shortcut of the block as the 7-tap conv Wsc . W0 of the waveform, same fragment form
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) wscf[nt][ks] = (4 * ks + q) < 7 ? a.wsc0[(nt * 16 + r16) * 7 + 4 * ks + q] : 0.f;
    // bf16 pieces of the A operands (row r16, k = 32 ks + 8 q .. + 7); `kmax` zero-fills the tail's K padding
    auto wsplit = [&](const float* row, int k0, int kmax, float scale, V8 (&dst)[NP]) {
        f4 lo, hi;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            lo[k] = (k0 + k) < kmax ? row[k0 + k] : 0.f;
            hi[k] = (k0 + 4 + k) < kmax ? row[k0 + 4 + k] : 0.f;
        }
        V4 plo[NP], phi[NP];
        split4<SchemeNoCheck<SC>>(lo, scale, plo);
        split4<SchemeNoCheck<SC>>(hi, scale, phi);
#pragma unroll
        for (int i = 0; i < NP; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) { dst[i][k] = plo[i][k]; dst[i][4 + k] = phi[i][k]; }
    };
    V8 w3p[NP][3], wtp[NP][2], wdp[NP][4];   // wtp: W1 only (K = 16, zero-padded to one K step of 32)
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
        V8 t[NP];
        wsplit(a.w3 + r16 * 96, ks * 32 + q * 8, 96, sw3, t);
#pragma unroll
        for (int i = 0; i < NP; ++i) w3p[i][ks] = t[i];
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        V8 t[NP];
        wsplit(a.wt + (nt * 16 + r16) * 48, q * 8, 16, swt, t);
#pragma unroll
        for (int i = 0; i < NP; ++i) wtp[i][nt] = t[i];
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        V8 t[NP];
        wsplit(a.wd + (wave * 16 + r16) * 128, ks * 32 + q * 8, 128, swd, t);
#pragma unroll
        for (int i = 0; i < NP; ++i) wdp[i][ks] = t[i];
    }

    auto fetch_wav = [&](int tile) -> float {
        if (tile >= total_tiles || tid >= SX_WAV) return 0.f;
        const int b = tile / tiles_per_clip;
        const int t0 = (tile - b * tiles_per_clip) * SX_ADV;
        int w = t0 - 10 + tid;
        w = w < 0 ? -w : w;
        w = w > N - 1 ? N - 1 : w;
        return a.wav[(long long)b * N + w];
    };
    // ---- B: conv0 of a tile at time |t0 - 4 + i| on the fp32 MFMA (as seanet_stage0.hip) -> split ELU copy. 8 units of
    //      (row tile, channel tile) over the 4 waves. Only the first tile of a clip needs the reflect index map ------------------------
    // waveform-segment indices of the two K = 4 tap groups of x0 row i (time t0 - 4 + i): only the first tile of a clip needs the reflect maps
    auto tap_index = [&](int t0, int i, int& a0, int& a1) {
        if (t0 == 0) {
            int tau = i - 4;
            tau = tau < 0 ? -tau : tau;
            a0 = tau + q - 6;
            a0 = (a0 < 0 ? -a0 : a0) + 10;
            a1 = tau + q - 2;
            a1 = (a1 < 0 ? -a1 : a1) + 10;
        } else {
            a0 = i + q;
            a1 = i + 4 + q;
        }
        a0 = a0 < SX_WAV - 1 ? a0 : SX_WAV - 1;   // rows >= 66 and the zero tap stay inside the (finite) segment
        a1 = a1 < SX_WAV - 1 ? a1 : SX_WAV - 1;
    };
    auto conv0_tile = [&](int tile, const float* Wseg) {
        if (tile >= total_tiles) return;
        const int t0 = (tile % tiles_per_clip) * SX_ADV;
        for (int unit = wave; unit < SX_XUNITS; unit += 4) {
            const int mt = unit >> 1, nt = unit & 1;
            const int i = mt * 16 + r16;
            int a0, a1;
            tap_index(t0, i, a0, a1);
            f4 acc = {0.f, 0.f, 0.f, 0.f};
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(nt ? w0f[1][0] : w0f[0][0], Wseg[a0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(nt ? w0f[1][1] : w0f[0][1], Wseg[a1], acc, 0, 0, 0);
            const f4 o = acc + *reinterpret_cast<const f4*>(B0s + nt * 16 + q * 4);
            const int off = i * SX_LDX + sx_swz(nt * 16 + q * 4, sx_xbit(i));
            over |= sx_store4<SC>(X0e, off, SX_XP, sx_elu4(o), sa);
        }
    };
    // Software pipeline over the workgroup's tiles (two barriers per tile): while tile t is in its strided-conv phase the same
    // waves run conv0 of tile t+1 into the (by then free) x0 buffers and park the waveform segment of tile t+2 in the other Wv buffer.
    const int stride = gridDim.x;
    {
        const float w0v = fetch_wav(blockIdx.x), w1v = fetch_wav(blockIdx.x + stride);
        __syncthreads();   // biases
        if (tid < SX_WAV) { Wv[tid] = w0v; Wv[SX_WAV + tid] = w1v; }
        __syncthreads();
        conv0_tile(blockIdx.x, Wv);
    }
    float wnext = fetch_wav(blockIdx.x + 2 * stride);
    __syncthreads();

    int par = 0;   // Wv[par] holds the segment of the current tile
    for (int tile = blockIdx.x; tile < total_tiles; tile += stride, par ^= 1) {
        const int b = tile / tiles_per_clip;
        const int t0 = (tile - b * tiles_per_clip) * SX_ADV;
        // ---- C + D on the wave's own 16 rows: h = ELU(conv3(ELU(x0)) + b3), row j uses x0 rows j..j+2 (K step = tap);
        //      r = ELU([h | x0] . [W1 | Wsc]^T + (b1 + bsc)), row j uses h row j and raw x0 row j + 2 -----------------------------
        {
            const int row = wave * 16 + r16;
            f4 acc = {0.f, 0.f, 0.f, 0.f};
            V8 xf[3][NP];
#pragma unroll
            for (int ks = 0; ks < 3; ++ks)
#pragma unroll
                for (int p = 0; p < NP; ++p) xf[ks][p] = *reinterpret_cast<const V8*>(X0e + p * SX_XP + (row + ks) * SX_LDX + sx_swz(q * 8, sx_xbit(row + ks)));
#pragma unroll
            for (int ks = 0; ks < 3; ++ks)
#pragma unroll
                for (int t = 0; t < SC::NPROD; ++t) acc = SC::mfma16(w3p[SC::prod_w(t)][ks], xf[ks][SC::prod_a(t)], acc);
            over |= sx_store4<SC>(Hs, row * SX_LDH + q * 4, SX_HP, sx_elu4(acc * rs3 + *reinterpret_cast<const f4*>(Bs + q * 4)), sa);
            // tail: W1 . h as ONE split K step (k 0..15 = h: lanes q < 2; k 16..31 zero weights, those lanes re-read finite h data) plus the
            // shortcut Wsc . x0[row + 2] as the folded 7-tap conv of the waveform on the fp32 MFMA (x0 row i <-> time t0 - 4 + i)
            const PT* s0 = Hs + row * SX_LDH + (q & 1) * 8;
            V8 tf[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) tf[p] = *reinterpret_cast<const V8*>(s0 + p * SX_HP);
            int a0, a1;
            tap_index(t0, row + 2, a0, a1);
            const float* Wseg = Wv + par * SX_WAV;
            f4 acc2[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}}, sc[2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                sc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wscf[nt][0], Wseg[a0], f4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                sc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wscf[nt][1], Wseg[a1], sc[nt], 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < SC::NPROD; ++t)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc2[nt] = SC::mfma16(wtp[SC::prod_w(t)][nt], tf[SC::prod_a(t)], acc2[nt]);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
                over |= sx_store4<SC>(Rs, ((row & 1) * SX_RIDX + (row >> 1)) * SX_LDR + sx_swz(nt * 16 + q * 4, sx_rbit(row)), SX_RP,
                                      sx_elu4((acc2[nt] * rst + sc[nt]) + *reinterpret_cast<const f4*>(Bs + 16 + nt * 16 + q * 4)), sa);
        }
        __syncthreads();
        if (t0 == 0) {   // reflect padding of the strided conv's input at the clip start: r[-1] = r[1], r[-2] = r[2]
            if (tid < 16 * NP) {
                const int p = tid >> 4, j = (tid >> 3) & 1, c = (tid & 7) * 4;   // j = 0 <-> t = -2 (copy of row 4); j = 1 <-> t = -1 (row 3)
                const int src = 4 - j;
                *reinterpret_cast<V4*>(Rs + p * SX_RP + ((j & 1) * SX_RIDX + (j >> 1)) * SX_LDR + sx_swz(c, sx_rbit(j))) =
                    *reinterpret_cast<const V4*>(Rs + p * SX_RP + ((src & 1) * SX_RIDX + (src >> 1)) * SX_LDR + sx_swz(c, sx_rbit(src)));
            }
            __syncthreads();
        }
        // ---- E: x1[u][16 wave .. + 15] = down0(ELU(r)): output u uses r rows 2u .. 2u+3 (K step = tap) ---------------------------
        {
            f4 acc[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
            V8 rf[4][NP][2];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int om = 0; om < 2; ++om)
#pragma unroll
                    for (int p = 0; p < NP; ++p)
                        rf[ks][p][om] = *reinterpret_cast<const V8*>(Rs + p * SX_RP + ((ks & 1) * SX_RIDX + om * 16 + r16 + (ks >> 1)) * SX_LDR +
                                                                     sx_swz(q * 8, sx_rbit(2 * (om * 16 + r16 + (ks >> 1)) + (ks & 1))));
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int t = 0; t < SC::NPROD; ++t)
#pragma unroll
                    for (int om = 0; om < 2; ++om)
                        acc[om] = SC::mfma16(wdp[SC::prod_w(t)][ks], rf[ks][SC::prod_a(t)][om], acc[om]);
            const f4 bd = *reinterpret_cast<const f4*>(Bs + 48 + wave * 16 + q * 4);
#pragma unroll
            for (int om = 0; om < 2; ++om) {
                const int u = om * 16 + r16, tout = t0 / 2 + u;
                if (u < SX_UO && tout < L1) *reinterpret_cast<f4*>(a.x1 + ((long long)b * L1 + tout) * 64 + wave * 16 + q * 4) = acc[om] * rsd + bd;
            }
        }
        // ---- B of the next tile, A of the one after it (the x0 buffers were last read before the barrier above; Wv[par] before the
        //      previous one) ------------------------------------------------------------------------------------------------------
        conv0_tile(tile + stride, Wv + (par ^ 1) * SX_WAV);
        if (tid < SX_WAV) Wv[par * SX_WAV + tid] = wnext;
        wnext = fetch_wav(tile + 3 * stride);
        __syncthreads();   // x0 of the next tile complete; Rs and Hs free
    }
    if constexpr (SC::RANGE_CHECK)
        range_publish(a.status, a.status ? a.status + 1 : nullptr, over);
}

template <class SC>
static int launch_stage0_scheme(const Stage0Args& a, hipStream_t stream, int grid) {
    constexpr int lds = (SC::NP * SX_XP + SC::NP * SX_HP + SC::NP * SX_RP) * 2 + (2 * SX_WAV + 32 + 112) * 4;
    { static LdsAttrFlags lds_attr; if (int rc = set_max_dynamic_lds(lds_attr, seanet_stage0x3_kernel<SC>, lds)) return rc; }
    hipLaunchKernelGGL(seanet_stage0x3_kernel<SC>, dim3(grid), dim3(256), lds, stream, a);
    AT_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_seanet_stage0x3(const Stage0Args& a, hipStream_t stream) {
    AT_REQUIRE(a.N % 2 == 0 && a.N >= 16 && a.B >= 1, "fused stage 0 needs an even sample count");
    AT_REQUIRE(a.wsc0 != nullptr && a.bsc0 != nullptr, "stage 0 needs the folded shortcut weights (Stage0Args::wsc0)");
    const long long tiles = (long long)a.B * ((a.N + SX_ADV - 1) / SX_ADV);
    AT_REQUIRE(tiles < (1LL << 30), "tile arithmetic is 32-bit");
    const int per_cu = a.scheme == XB_SCHEME_F16X2 ? 3 : 2;   // resident workgroups per CU (fp16 scheme: 51 KB LDS, 160 registers; measured 6.5 -> 6.0 ms vs two)
    const int grid = (int)(tiles < 256 * per_cu ? tiles : 256 * per_cu);
    if (a.scheme == XB_SCHEME_F16X2) {
        AT_REQUIRE(a.act_scale > 0.f && a.w3_scale > 0.f && a.wt_scale > 0.f && a.wd_scale > 0.f, "two-piece fp16 stage 0 needs its scales");
        return launch_stage0_scheme<SchemeF16x2>(a, stream, grid);
    }
    return launch_stage0_scheme<SchemeBf16x3>(a, stream, grid);
}

}  // namespace at
