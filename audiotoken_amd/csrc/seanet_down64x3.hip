// SEANet stage-1 strided conv (64 -> 128 channels, k = 8, stride 4, causal) on the bf16 matrix cores: the same register-stationary
// scheme as seanet_down64.hip, but with both operands as exact 3-way bf16 splits (see gemm_bf16x3.hip for the arithmetic and its
// measured accuracy) and six v_mfma_f32_16x16x32_bf16 (16 cycles each) per 32-wide K step instead of eight 32-cycle fp32 MFMAs:
// 6/16 of the matrix time. Weights: wave w keeps output channels 32w..32w+31 x K = 512 as three bf16 pieces in 384 registers
// (one wave per SIMD owns 512) for the lifetime of the persistent workgroup — unlike the split GEMM (bound by the L2 -> LDS
// operand stream) this kernel streams only the activations. A tile of 32 output rows = 132 input rows is split while it is staged
// into LDS ([3 pieces][132 rows][64 ch] bf16, 16-byte chunks XOR-swizzled by (row >> 2) & 7 so that the 16 rows 4 apart a fragment
// read touches fall on different banks), double-buffered: tile t+1 is split into the other buffer in the shadow of tile t's MFMAs
// (a bf16 MFMA leaves half of its cycles to VALU issue) and tile t+2's loads are in flight — one barrier per tile.
// The arithmetic differs from the fp32 k-ordered chain of the GEMM path in rounding only (both are within 2e-6 of float64): this
// kernel is compared with the unfused path by tolerance, not bit-identity (tests/test_acoustic_gpu.py).
#include "gemm_core.h"
#include "encodec_kernels.h"
#include "split_scheme.h"
#include <cstdlib>

namespace at {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int DX_TU = 32;                  // output rows per tile
constexpr int DX_ROWS = 4 * DX_TU + 4;     // input rows per tile: row i <-> time 4*u0 - 4 + i
constexpr int DX_CHUNKS = DX_ROWS * 16;    // float4 chunks of the input tile
constexpr int DX_PL = 40, DX_LD = 80;      // input row i lives in plane i & 3 at index i >> 2 (40 per plane: 33 used + the tail of the last chunk), 64 + 16 channels per row (+ 32 B: conflict-free fragment reads, see seanet_res128x3.hip)
constexpr int DX_PIECE = 4 * DX_PL * DX_LD; // bf16 elements of one piece of the tile
// physical LDS row: rows 32 apart (fragment lanes r16 and r16 + 8) swap odd/even so that they fall in different halves of the banks
// scheduling pattern for one K step: after each of the 12 MFMAs up to 4 vector instructions of the side work (the rest follows)
template <int NMFMA>
__device__ __forceinline__ void dx_interleave() {
#pragma unroll
    for (int i = 0; i < NMFMA; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
    }
}
// plane-major rows: the 16 rows 4 apart that a fragment read touches are consecutive (stride 144 B: 16 distinct bank groups) and every
// fragment address is a per-lane base plus a compile-time offset (the XOR swizzle this replaces cost ~40 address registers)
__device__ __forceinline__ int dx_off(int row) { return ((row & 3) * DX_PL + (row >> 2)) * DX_LD; }

// 8 waves, one 16-channel tile each: 16 K steps x NP weight fragments per lane (192 registers with three bf16 pieces, 128 with two fp16
// pieces), two waves per SIMD. SC = operand scheme (split_scheme.h): three bf16 pieces / six products, or (round 2, default) two fp16 pieces /
// three products with power-of-two activation / weight scales (a.act_scale, a.w_scale) and a range status bit. (The 4-wave x 384-register
// shape of round 1 lost its A/B — the compiler parked weights in AGPRs, ~22 cycles per MFMA — and is gone.)
template <class SC>
__global__ __launch_bounds__(512, 1) void seanet_down64x3_kernel(Down64Args a) {
    typedef typename SC::T PT;
    typedef typename SC::V8 V8;
    typedef typename SC::V4 V4;
    constexpr int NP = SC::NP;
    constexpr int NTHR = 512;
    constexpr int DX_PRE = (DX_CHUNKS + NTHR - 1) / NTHR;   // 16-byte chunks of the input tile per thread: 5
    constexpr int SJ = 6;                                   // K steps between two chunks of side work
    constexpr int DX_AHEAD = 12;                            // K steps between a chunk's global load and its split into LDS (chunks in flight: 2)
    extern __shared__ __attribute__((aligned(16))) unsigned char Xp_raw[];   // [2 buffers][NP pieces][4 planes][40][80]
    PT* Xp = reinterpret_cast<PT*>(Xp_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, q = lane >> 4;
    const int L = a.L, Lo = L / 4;
    const int tiles_per_clip = (Lo + DX_TU - 1) / DX_TU;
    const int total_tiles = a.B * tiles_per_clip;   // < 2^31: checked by the launcher
    const float sa = SC::RANGE_CHECK ? a.act_scale : 1.0f, sw = SC::RANGE_CHECK ? a.w_scale : 1.0f;
    const float rs = SC::RANGE_CHECK ? 1.0f / (a.act_scale * a.w_scale) : 1.0f;
    RangeMax over;

    // ---- weights -> pieces in registers, once: wr[p][ks] = split(W[16w + r16][32 ks + 8 q .. +7] * sw) ------------------------------
    V8 wr[NP][16];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        const float* src = a.w + (wave * 16 + r16) * 512 + ks * 32 + q * 8;
        const f4 lo = *reinterpret_cast<const f4*>(src), hi = *reinterpret_cast<const f4*>(src + 4);
        V4 plo[NP], phi[NP];
        split4<SchemeNoCheck<SC>>(lo, sw, plo);
        split4<SchemeNoCheck<SC>>(hi, sw, phi);
#pragma unroll
        for (int i = 0; i < NP; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) { wr[i][ks][k] = plo[i][k]; wr[i][ks][4 + k] = phi[i][k]; }
    }
    f4 pre[DX_PRE];
    // chunk j of a tile: 16-byte piece c = tid + 512 j of its 132 x 64 input rows
    auto load_chunk = [&](int j, const float* xb, int u0) {
        int c = tid + NTHR * j;
        c = c < DX_CHUNKS ? c : DX_CHUNKS - 1;
        int tau = 4 * u0 - 4 + (c >> 4);
        tau = tau < 0 ? -tau : tau;            // causal reflect padding at the clip start
        tau = tau > L - 1 ? L - 1 : tau;       // rows past the end only feed outputs that are never stored
        pre[j] = *reinterpret_cast<const f4*>(xb + (unsigned)(tau * 64 + (c & 15) * 4));
    };
    // split chunk j into the pieces of LDS buffer X
    auto stage_chunk = [&](int j, PT* X) {
        const int c = tid + NTHR * j;
        {   // chunks past the tile (the tail of the last j) land in the pad rows: no branch, so the work can sit between MFMAs
            const int row = c >> 4, c4 = c & 15;          // 4 channels 4 c4 .. 4 c4 + 3 = half of the 8-channel chunk c4 >> 1
            V4 pp[NP];
            over |= split4<SC>(pre[j], sa, pp);
            PT* d = X + dx_off(row) + c4 * 4;
#pragma unroll
            for (int i = 0; i < NP; ++i) *reinterpret_cast<V4*>(d + i * DX_PIECE) = pp[i];
        }
    };
    auto clip_base = [&](int tile, int& u0) {
        const int b = tile / tiles_per_clip;
        u0 = (tile - b * tiles_per_clip) * DX_TU;
        return a.x + (long long)b * L * 64;
    };
    const int stride = gridDim.x;
    if ((int)blockIdx.x < total_tiles) {
        int u0;
        const float* xb = clip_base(blockIdx.x, u0);
#pragma unroll
        for (int j = 0; j < DX_PRE; ++j) load_chunk(j, xb, u0);
#pragma unroll
        for (int j = 0; j < DX_PRE; ++j) stage_chunk(j, Xp);
        const int t1 = blockIdx.x + stride;
        xb = clip_base(t1 < total_tiles ? t1 : total_tiles - 1, u0);
#pragma unroll
        for (int j = 0; j < DX_PRE; ++j)
            if (SJ * j + 2 - DX_AHEAD < 0) load_chunk(j, xb, u0);   // what the previous iteration would have issued
    }
    __syncthreads();

    int buf = 0;
    for (int tile = blockIdx.x; tile < total_tiles; tile += stride, buf ^= 1) {
        const int b = tile / tiles_per_clip;
        const int u0 = (tile - b * tiles_per_clip) * DX_TU;
        const PT* X = Xp + buf * (NP * DX_PIECE);
        PT* Xn = Xp + (buf ^ 1) * (NP * DX_PIECE);
        // branch-free side work: past the end the last tile is loaded / staged again into buffers nobody reads
        int u1, u2;
        const int t1 = tile + stride, t2 = tile + 2 * stride;
        const float* xb1 = clip_base(t1 < total_tiles ? t1 : total_tiles - 1, u1);
        const float* xb2 = clip_base(t2 < total_tiles ? t2 : total_tiles - 1, u2);
        // One 16-row m-tile at a time (4 accumulator registers live). Spread over the 32 K steps: chunk j of the NEXT tile (loaded during
        // the previous iteration) is split into the other buffer and the same registers are refilled with chunk j of the tile after it —
        // the vector work sits between the MFMAs (a 16x16x32 MFMA holds vector issue for 8 of its 16 cycles) instead of in a phase of its own.
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            f4 acc = f4{0.f, 0.f, 0.f, 0.f};
            auto xread = [&](int ks, V8 (&xf)[NP]) {
                const int tap = ks >> 1, ch = (ks & 1) * 4 + q;    // 8-channel chunk of this lane's K slice
                const int row = 4 * (16 * m + r16) + tap;
                const PT* src = X + dx_off(row) + ch * 8;
#pragma unroll
                for (int p = 0; p < NP; ++p) xf[p] = *reinterpret_cast<const V8*>(src + p * DX_PIECE);
            };
            auto side_work = [&](int step) {   // step = 16 m + ks
                if (step % SJ == 2 && step / SJ < DX_PRE) stage_chunk(step / SJ, Xn);
                // chunk j is loaded DX_AHEAD steps before it is split: in this iteration for the later chunks, else in the previous one
#pragma unroll
                for (int j = 0; j < DX_PRE; ++j) {
                    if (SJ * j + 2 - DX_AHEAD >= 0 && step == SJ * j + 2 - DX_AHEAD) load_chunk(j, xb1, u1);
                    if (SJ * j + 2 - DX_AHEAD < 0 && step == SJ * j + 2 - DX_AHEAD + 32) load_chunk(j, xb2, u2);
                }
            };
            V8 xa[NP];
            // two waves per SIMD: the partner wave covers the LDS latency, one fragment set is enough
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                xread(ks, xa);
#pragma unroll
                for (int t = 0; t < SC::NPROD; ++t) acc = SC::mfma16(wr[SC::prod_w(t)][ks], xa[SC::prod_a(t)], acc);
                side_work(16 * m + ks);
                dx_interleave<SC::NPROD>();
                __builtin_amdgcn_sched_barrier(0);
            }
            const int u = u0 + m * 16 + r16;
            if (u < Lo) {
                float* dst = a.out + ((long long)b * Lo + u) * 128 + wave * 16 + q * 4;
                *reinterpret_cast<f4*>(dst) = acc * rs + *reinterpret_cast<const f4*>(a.b + wave * 16 + q * 4);
            }
        }
        __syncthreads();   // buffer buf ^ 1 is complete, buffer buf is free
    }
    if constexpr (SC::RANGE_CHECK)
        range_publish(a.status, a.status ? a.status + 1 : nullptr, over);
}

int launch_seanet_down64x3(const Down64Args& a, hipStream_t stream) {
    AT_REQUIRE(a.L >= 8 && a.L % 4 == 0 && a.B >= 1, "down64x3: L % 4 == 0, L >= 8");
    const long long tiles = (long long)a.B * ((a.L / 4 + DX_TU - 1) / DX_TU);
    AT_REQUIRE(tiles < (1LL << 30) && (long long)a.L * 64 < (1LL << 30), "tile / offset arithmetic is 32-bit");
    const int grid = (int)(tiles < 256 ? tiles : 256);
    if (a.scheme == XB_SCHEME_F16X2) {
        AT_REQUIRE(a.act_scale > 0.f && a.w_scale > 0.f, "down64x3: the fp16 scheme needs its scales");
        const size_t lds = (size_t)2 * 2 * DX_PIECE * 2;
        { static LdsAttrFlags lds_attr_0; if (int rc = set_max_dynamic_lds(lds_attr_0, seanet_down64x3_kernel<SchemeF16x2>, lds)) return rc; }
        hipLaunchKernelGGL(seanet_down64x3_kernel<SchemeF16x2>, dim3(grid), dim3(512), lds, stream, a);
    } else {
        const size_t lds = (size_t)2 * 3 * DX_PIECE * 2;
        { static LdsAttrFlags lds_attr_1; if (int rc = set_max_dynamic_lds(lds_attr_1, seanet_down64x3_kernel<SchemeBf16x3>, lds)) return rc; }
        hipLaunchKernelGGL(seanet_down64x3_kernel<SchemeBf16x3>, dim3(grid), dim3(512), lds, stream, a);
    }
    AT_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace at
