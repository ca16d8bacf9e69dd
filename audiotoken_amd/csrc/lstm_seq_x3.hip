// Whole-sequence LSTM layer with the recurrent product on the bf16 matrix cores: the persistent kernel of lstm_seq.hip (same
// decomposition idea, same hand-off protocol, same gate-interleaved weight / xg / h layouts) with W_hh and h_{t-1} as exact
// 3-way bf16 splits and six v_mfma_f32_16x16x32_bf16 per 32-wide K step (arithmetic and accuracy: gemm_bf16x3.hip).
// What changes in the decomposition: a workgroup owns 32 hidden units = 128 gate rows (wave w: two 16-row tiles = 8 units),
// its weights are 384 registers of bf16 pieces per lane (one wave per SIMD owns 512), so a group of 16 clips has 16 workgroups
// instead of 32 and ONE workgroup sits on a CU: 192 MFMAs of 16 cycles per step and wave instead of 2 x 128 of 32 cycles for the
// two co-resident fp32 workgroups, and half as many producers to wait for. h_{t-1} is split while it is staged into LDS
// ([pieces][16 clips][512 + 16 pad]: the 32-byte pad makes the fragment reads conflict-free).
// The result differs from the fp32 chain in rounding only (both ~1e-6 of float64 per step); it is compared with the fp32
// persistent kernel and the per-step path by tolerance and by identical tokens (tests/test_acoustic_gpu.py).
//
// Round 2: the kernel is templated on the operand scheme (split_scheme.h). Default is the two-piece fp16 scheme: h lies in (-1, 1), so
// h * 2^14 always fits fp16 (no range check needed) and W_hh is scaled per layer into [2^14, 2^15); three products per K step instead of
// six (48 MFMAs per wave and step instead of 96), 128 weight registers per lane instead of 192, two LDS piece planes instead of three.
// The 4-wave x 384-register shape of round 1 lost its A/B (8.4 vs 7.6 ms) and is gone.
#include "gemm_core.h"
#include "encodec_kernels.h"
#include "split_scheme.h"
#include <cstdlib>
#include <cstdio>
#include <vector>
#include <cmath>

namespace at {

// -DLX_DEBUG_STAMPS (tools/lstm_stamps.sh; never in the product build): workgroup 0 / thread 0 records the cycle counter at five points of every
// step and the launcher prints the average phase lengths of the launch to stderr. Round-2 figures (256 clips, T = 750, layer 1, cycles at ~2.1 GHz):
// wait for the 16 flags 2354 | h load + split + LDS + barrier 1565 | 48 MFMAs x 2 waves per SIMD 1736 | gates + store drain + barrier + flag
// 1945 | tail 152 = 3.7 us per step; with the line-sized publish of h and y (FAST below) 2049 | 1584 | 1712 | 1643 | 220 (one box, same
// session: 6.24 ms -> 5.06 ms for the two layers of 256 x 750 steps). A flag-less variant (producers publish packed piece pairs into a ring of sentinel-armed slots, consumers
// poll the data) shortened the publish to 822 cycles but lengthened the wait to 4009-4673 (the last of 128 scattered 16-byte write-throughs
// per workgroup lands later than one flag word behind a drain; polling the tile from every wave also slowed the stores): 8.1-11 ms vs 6.1 ms.
#ifdef LX_DEBUG_STAMPS
__device__ unsigned long long lx_stamps[8192 * 6];
#define LX_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0 && t < 8192) lx_stamps[t * 6 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define LX_STAMP(i) do {} while (0)
#endif

typedef unsigned int u4 __attribute__((ext_vector_type(4)));
constexpr float LX_H_SCALE = 16384.0f;   // fp16 scheme: h (|h| < 1) is split as h * 2^14

constexpr int LX_H = 512;
constexpr int LX_CLIPS = 16;       // clips per group = one MFMA row tile
constexpr int LX_SLICES = 16;      // workgroups per group: 32 hidden units each
constexpr int LX_LDH = LX_H + 16;  // LDS row stride (16-bit elements): + 32 B, conflict-free fragment reads under the real ds_read_b128 lane grouping (see seanet_res128x3.hip)
constexpr int LX_HP = LX_CLIPS * LX_LDH;       // elements of one piece
constexpr int LX_STATUS = 63;      // as lstm_seq.hip
constexpr int LX_FLAGS = 128;      // flags[16 groups][32 words] (16 used)
constexpr int LX_FLAG_STRIDE = 32;
constexpr int LX_MAX_GROUPS = 16;

__device__ __forceinline__ f4 lx_mfma(bf16x8 w, bf16x8 x, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x, c, 0, 0, 0); }
__device__ __forceinline__ f4 lx_mfma(f16x8 w, f16x8 x, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(w, x, c, 0, 0, 0); }

// 8 waves, one 16-row tile each (two waves per SIMD); SC = operand scheme; w_scale / h_scale / acc_scale: the fp16 scheme's powers of two
template <class SC>
__global__ __launch_bounds__(512, 1) void lstm_seq_x3_kernel(LstmSeqArgs a, float w_scale, float h_scale, float acc_scale) {
    typedef typename SC::T PT;
    typedef typename SC::V8 V8;
    typedef typename SC::V4 V4;
    constexpr int NP = SC::NP, NJ = 1;
    constexpr int NTHR = 512 / NJ, NST = 2048 / NTHR;   // threads; 16-byte staging chunks per thread
    extern __shared__ __attribute__((aligned(16))) unsigned char Hp_raw[];   // [NP][16 clips][528]
    PT* Hp = reinterpret_cast<PT*>(Hp_raw);
    __shared__ int abort_s;
    __shared__ __attribute__((aligned(16))) float Hx[LX_CLIPS][32];   // this workgroup's h_t slice, gathered for line-sized stores
    __shared__ __attribute__((aligned(16))) float Hy[LX_CLIPS][32];   // and its y_t = act(h_t + skip) slice (layer 2)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, q = lane >> 4;
    const int group = blockIdx.x % a.n_groups;
    const int slice = blockIdx.x / a.n_groups;     // 0..15: hidden units [32*slice, 32*slice + 32)
    const int b0 = group * LX_CLIPS;
    const int T = a.T;

    // row tile j of wave w = tile nt = 2w + j of the 8 tiles of this slice; in the weight layout of lstm_seq.hip (64-row blocks of 16 units,
    // rows = unit * 4 + gate) that is block 2*slice + (nt >> 2), rows 16*(nt & 3) ..; lane (r16, q) then owns unit .. + q, gates = acc[0..3]
    int unit[NJ];
    V8 wp[NP][NJ][16];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int nt = NJ * wave + j;
        const int blk = 2 * slice + (nt >> 2), sub = nt & 3;
        unit[j] = blk * 16 + sub * 4 + q;
        const float* wrow = a.w_hh + ((long long)blk * 64 + sub * 16 + r16) * LX_H;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const f4 lo = *reinterpret_cast<const f4*>(wrow + ks * 32 + q * 8), hi = *reinterpret_cast<const f4*>(wrow + ks * 32 + q * 8 + 4);
            V4 plo[NP], phi[NP];
            split4<SchemeNoCheck<SC>>(lo, w_scale, plo);
            split4<SchemeNoCheck<SC>>(hi, w_scale, phi);
#pragma unroll
            for (int i = 0; i < NP; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) { wp[i][j][ks][k] = plo[i][k]; wp[i][j][ks][4 + k] = phi[i][k]; }
        }
    }
    const int clip = b0 + r16;
    const bool clip_ok = clip < a.B;
    const long long own_row = (long long)(clip_ok ? clip : a.B - 1) * T;
    f4 bhh[NJ];
    float cst[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) cst[j] = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) bhh[j] = *reinterpret_cast<const f4*>(a.b_hh + unit[j] * 4);
    // h_{t-1} staging: thread -> 8 x (clip row, 16-B chunk) of the [16][512] tile: e = tid + 256*j -> row = e >> 7, chunk = tid & 127
    const __amdgpu_buffer_rsrc_t hrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.h_out, 0, (int)a.h_bytes, 0x00020000);
    int g_off[NST];
#pragma unroll
    for (int j = 0; j < NST; ++j) {
        const int row = (tid >> 7) + (NTHR / 128) * j, ch = tid & 127;
        const int cb = b0 + row < a.B ? b0 + row : a.B - 1;
        g_off[j] = (int)((((long long)cb * T) * LX_H + ch * 4) * 4);   // < 2^31: checked by the launcher
    }
    const int l_off0 = (tid >> 7) * LX_LDH + (tid & 127) * 4;            // + (NTHR / 128) j rows
    unsigned* flags = a.sync + LX_FLAGS + group * LX_FLAG_STRIDE;
    const PT* hb = Hp + r16 * LX_LDH + q * 8;                            // fragment base: + 32 ks elements, + piece

    // input-side gates and the skip input of a step: independent of the recurrence
    auto fetch = [&](int tt, f4 (&xg)[NJ], float (&sk)[NJ]) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            xg[j] = *reinterpret_cast<const f4*>(a.xg + (own_row + tt) * (4 * LX_H) + unit[j] * 4);
            sk[j] = 0.f;
            if (a.y_out) sk[j] = a.skip[(own_row + tt) * LX_H + unit[j]];
        }
    };
    // FAST (the fp16 scheme; the bf16 scheme has no registers to spare for it): the h_t slice is published by one wave as whole lines (below).
    // Tried on top of it and measured slower (stamps, tools/lstm_stamps.sh): every wave polling only the two producers whose slices it stages and
    // loading them without the workgroup barrier (staging 1580 -> 1030 cycles but the wait 2050 -> 3020: eight polling waves per workgroup slow
    // the flag's way to the pollers, as in the flag-less variant; 5.45 ms); fetching xg / skip one step ahead (the HBM loads then delay the h tile
    // loads they are issued behind: 5.6-6.4 ms) and polling the flags through the scalar memory path (5.6 ms).
    constexpr bool FAST = SC::NP == 2, YG = FAST;
    for (int t = 0; t < T; ++t) {
        LX_STAMP(0);
        f4 xg[NJ];
        float skipv[NJ];
        fetch(t, xg, skipv);   // issued before the wait
        __builtin_amdgcn_sched_barrier(0);
        f4 acc[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[j] = f4{0.f, 0.f, 0.f, 0.f};
        if (t > 0) {
            // ---- wait until all 16 slices of this group have published h_{t-1} (protocol: lstm_seq.hip) -------------------------
            if (wave == 0) {
                const unsigned target = (unsigned)t;
                unsigned spins = 0;
                int give_up = 0;
                for (;;) {
                    const unsigned f = __hip_atomic_load(flags + (lane & (LX_SLICES - 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (__builtin_amdgcn_ballot_w64(f < target) == 0ull) break;
                    ++spins;
                    if (spins > a.spin_limit) { give_up = 1; break; }
                    if ((spins & 1023u) == 0u &&
                        __hip_atomic_load(a.sync + LX_STATUS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { give_up = 1; break; }
                }
                if (lane == 0) {
                    abort_s = give_up;
                    if (give_up) __hip_atomic_store(a.sync + LX_STATUS, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            __syncthreads();       // also: every wave has finished reading the previous step's pieces
            if (abort_s) return;   // uniform: a member of the group is not making progress
            LX_STAMP(1);
            LX_STAMP(2);
            // ---- h_{t-1} [16][512] -> sc1 loads to registers -> split -> LDS pieces -----------------------------------------------
            u4 stage[NST];
            const int toff = (t - 1) * (LX_H * 4);
#pragma unroll
            for (int j = 0; j < NST; ++j) stage[j] = __builtin_amdgcn_raw_buffer_load_b128(hrsrc, g_off[j] + toff, 0, 16);   // aux 16 = sc1
#pragma unroll
            for (int j = 0; j < NST; ++j) {
                const f4 hv = {__uint_as_float(stage[j][0]), __uint_as_float(stage[j][1]), __uint_as_float(stage[j][2]), __uint_as_float(stage[j][3])};
                V4 pp[NP];
                split4<SchemeNoCheck<SC>>(hv, h_scale, pp);
                PT* d = Hp + l_off0 + (NTHR / 128) * j * LX_LDH;
#pragma unroll
                for (int i = 0; i < NP; ++i) *reinterpret_cast<V4*>(d + i * LX_HP) = pp[i];
            }
            __syncthreads();
            LX_STAMP(3);
            // ---- gates += h_{t-1} . W_slice^T: 16 K steps x 6 products x 2 row tiles = 192 MFMAs per wave; fragments one K step ahead ----
            V8 xa[NP], xb[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) xa[p] = *reinterpret_cast<const V8*>(hb + p * LX_HP);
#pragma unroll
            for (int ks = 0; ks < 16; ks += 2) {
#pragma unroll
                for (int p = 0; p < NP; ++p) xb[p] = *reinterpret_cast<const V8*>(hb + p * LX_HP + (ks + 1) * 32);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int tt = 0; tt < SC::NPROD; ++tt)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[j] = lx_mfma(wp[SC::prod_w(tt)][j][ks], xa[SC::prod_a(tt)], acc[j]);
                if (ks + 2 < 16) {
#pragma unroll
                    for (int p = 0; p < NP; ++p) xa[p] = *reinterpret_cast<const V8*>(hb + p * LX_HP + (ks + 2) * 32);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int tt = 0; tt < SC::NPROD; ++tt)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[j] = lx_mfma(wp[SC::prod_w(tt)][j][ks + 1], xb[SC::prod_a(tt)], acc[j]);
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[j] *= acc_scale;   // exact: a power of two (1 for the bf16 scheme)
        }
        LX_STAMP(4);
        // ---- cell update (torch CPU LSTMCell order: gates = (hW + b_hh) + igates; c = f*c + i*g unfused) -----------------------------
        float hn[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const f4 g = (acc[j] + bhh[j]) + xg[j];
            const float ig = lstm_sigmoid(g.x), fg = lstm_sigmoid(g.y), cg = lstm_tanh(g.z), og = lstm_sigmoid(g.w);
            const float c_new = __fadd_rn(__fmul_rn(fg, cst[j]), __fmul_rn(ig, cg));
            hn[j] = og * lstm_tanh(c_new);
            cst[j] = c_new;
            if (FAST) {
                Hx[r16][(NJ * wave + j) * 4 + q] = hn[j];
                if (a.y_out && YG) {
                    const float yv = hn[j] + skipv[j];
                    Hy[r16][(NJ * wave + j) * 4 + q] = a.y_elu ? elu1(yv) : yv;
                }
            } else if (clip_ok)
                __hip_atomic_store(reinterpret_cast<unsigned*>(a.h_out) + (own_row + t) * LX_H + unit[j], __float_as_uint(hn[j]),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // write-through (sc1): no release fence needed
        }
        // ---- publish: the slice [16 clips][32 units] is gathered in LDS and ONE wave stores it as whole 128-byte lines (two 16-byte
        // write-through stores per lane instead of 8 waves x 16 partial lines), drains and signals: no second barrier ----------------------
        if (!FAST) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains
        __syncthreads();
        if (!FAST) {
            if (tid == 0) __hip_atomic_store(flags + slice, (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (wave == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int idx = lane + 64 * i, row = idx >> 3, c4 = idx & 7;
                const u4 v = *reinterpret_cast<const u4*>(&Hx[row][c4 * 4]);
                if (b0 + row < a.B)
                    __builtin_amdgcn_raw_buffer_store_b128(v, hrsrc, (((b0 + row) * T + t) * LX_H + slice * 32 + c4 * 4) * 4, 0, 16);   // aux 16 = sc1: write-through; < 2^31: launcher
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(flags + slice, (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        LX_STAMP(5);
        if (YG) {
            if (a.y_out && wave == 1) {   // the y slice as whole lines too, by a wave that is not the publisher (plain stores: later kernels read them)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int idx = lane + 64 * i, row = idx >> 3, c4 = idx & 7;
                    if (b0 + row < a.B)
                        *reinterpret_cast<u4*>(a.y_out + ((long long)(b0 + row) * T + t) * LX_H + slice * 32 + c4 * 4) = *reinterpret_cast<const u4*>(&Hy[row][c4 * 4]);
                }
            }
        } else if (a.y_out && clip_ok) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const float yv = hn[j] + skipv[j];
                a.y_out[(own_row + t) * LX_H + unit[j]] = a.y_elu ? elu1(yv) : yv;
            }
        }
    }
}

// clips one launch can take: all workgroups must be resident (one per CU), 16 per group of 16 clips
int lstm_seq_x3_max_clips() {
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (cached[dev] == 0) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
        int groups = cus / LX_SLICES;
        groups = groups > LX_MAX_GROUPS ? LX_MAX_GROUPS : groups;
        cached[dev] = (groups < 1 ? 1 : groups) * LX_CLIPS;
    }
    return cached[dev];
}

template <class SC>
static int launch_lx(const LstmSeqArgs& a, float w_scale, float h_scale, hipStream_t stream) {
    const size_t lds = (size_t)SC::NP * LX_HP * 2;
    { static LdsAttrFlags lds_attr; if (int rc = set_max_dynamic_lds(lds_attr, lstm_seq_x3_kernel<SC>, lds)) return rc; }
    hipLaunchKernelGGL(lstm_seq_x3_kernel<SC>, dim3(a.n_groups * LX_SLICES), dim3(512), lds, stream, a, w_scale, h_scale, 1.0f / (w_scale * h_scale));
    AT_CHECK_HIP(hipGetLastError());
    return 0;
}

#ifdef LX_DEBUG_STAMPS
static void lx_print_stamps(int T, hipStream_t stream) {
    (void)hipStreamSynchronize(stream);
    static std::vector<unsigned long long> hbuf(8192 * 6);
    (void)hipMemcpyFromSymbol(hbuf.data(), HIP_SYMBOL(lx_stamps), hbuf.size() * 8);
    double ph[6] = {0, 0, 0, 0, 0, 0};
    const int n = (T < 8192 ? T : 8192) - 2;
    for (int t = 1; t <= n; ++t) {
        const unsigned long long* s0 = &hbuf[t * 6];
        for (int i = 0; i < 5; ++i) ph[i] += (double)(s0[i + 1] >= s0[i] ? s0[i + 1] - s0[i] : 0);
        ph[5] += (double)(hbuf[(t + 1) * 6] - s0[5]);
    }
    std::fprintf(stderr, "lstm stamps (cycles, avg over %d steps): wait %.0f  (unused) %.0f  load+split+lds %.0f  mfma %.0f  gates+publish %.0f  tail %.0f\n",
                 n, ph[0] / n, ph[1] / n, ph[2] / n, ph[3] / n, ph[4] / n, ph[5] / n);
}
#endif

int launch_lstm_seq_x3(const LstmSeqArgs& a_in, hipStream_t stream) {
    LstmSeqArgs a = a_in;
    AT_REQUIRE(a.B >= 1 && a.B <= lstm_seq_x3_max_clips() && a.T >= 1, "lstm_seq_x3: too many clips for one launch on this device");
    a.n_groups = (a.B + LX_CLIPS - 1) / LX_CLIPS;
    a.h_bytes = (long long)a.B * a.T * LX_H * 4;
    AT_REQUIRE(a.h_bytes < (1ll << 31), "lstm_seq_x3: h buffer exceeds the 2 GB buffer-descriptor range");
    AT_CHECK_HIP(hipMemsetAsync(a.sync + LX_FLAGS, 0, LX_MAX_GROUPS * LX_FLAG_STRIDE * sizeof(unsigned), stream));   // flags, every launch
    const int rc = a.w_scale_f16 > 0.f ? launch_lx<SchemeF16x2>(a, a.w_scale_f16, LX_H_SCALE, stream) : launch_lx<SchemeBf16x3>(a, 1.0f, 1.0f, stream);
#ifdef LX_DEBUG_STAMPS
    if (rc == 0) lx_print_stamps(a.T, stream);
#endif
    return rc;
}

}  // namespace at
