// HuBERT's grouped positional convolution (HF HubertPositionalConvEmbedding: Conv1d(768 -> 768, k 128, padding 64, groups 16), last output dropped,
// GELU; reference audiotoken/encoder.py:87-108 reaches it through HubertModel) on the two-piece fp16 scheme — the last large fp32-MFMA remnant of
// semantic_s (round 3: 16 windowed fp32 GEMMs of N = 48, 21 ms per 128 x 30 s step).
//   pos[b][t][48 g + co] = x[b][t][48 g + co] + gelu( bias[48 g + co] + sum_{j < 128, ci < 48} W[g][co][j][ci] * x[b][t + j - 64][48 g + ci] )
// N = 48 per group does not fit the split GEMM's 128-column tiles, and a windowed GEMM would re-fetch every input row once per tap from L2. Here the
// INPUT TILE IS RESIDENT IN LDS: a workgroup owns 256 output rows of one (clip, group); the 383 input rows x 48 channels it needs are split once into
// hi / lo fp16 pieces in LDS (row stride 112 B: odd multiple of 16 B), and the 128 taps are SHIFTED fragment reads of that one image — the K index of the
// contraction is (tap, channel) = 6144, walked in 192 steps of 32 = four 8-channel groups, each group a 16-byte read at row m + tap. The group's
// weights (1.2 MB as pieces, L2-resident: the grid runs one group at a time) stream through a 4-slot LDS ring by LDS-DMA, 6 KB per K step, pre-arranged
// at finalize as [group][K step][piece][k-block][48 rows][16] so that a step is six linear 1-KB DMA instructions.
// 8 waves, each 32 rows x 48 channels = 2 x 3 MFMA tiles (v_mfma_f32_16x16x32_f16, weights as the A operand so a lane owns 4 consecutive output
// channels of one row), 18 MFMAs per wave and step beside 10 ds_read_b128. ~450 MFLOP per tile at the MFMA rate = 110 k cycles.
#include "at_common.h"
#include "gemm_bf16x3.h"
#include "split_scheme.h"

namespace at {

typedef __attribute__((address_space(3))) void pc_lds_void;
typedef const __attribute__((address_space(1))) void pc_glb_void;

constexpr int PC_G = 16, PC_C = 48, PC_K = 128, PC_PAD = 64;      // groups, channels per group, taps, left padding
constexpr int PC_ROWS = 256;                                      // output rows per workgroup
constexpr int PC_IN_ROWS = PC_ROWS + PC_K;                        // input rows staged (383 needed)
constexpr int PC_XLD = 112;                                       // bytes per staged row per piece (96 of data)
constexpr int PC_X_BYTES = 2 * PC_IN_ROWS * PC_XLD;               // both pieces
constexpr int PC_STEP_BYTES = 2 * 2 * PC_C * 16 * 2;              // one K step of weights: [piece][k-block][48][16] fp16 = 6144 B
constexpr int PC_NSTEPS = PC_K * PC_C / 32;                       // 192
constexpr int PC_RING = 4;
constexpr size_t PC_LDS = (size_t)PC_X_BYTES + (size_t)PC_RING * PC_STEP_BYTES;

// folded fp32 weights [16 groups][48 out][128 taps][48 in] * scale -> [16][192 steps][2 pieces][2 k-blocks][48][16] fp16
__global__ __launch_bounds__(256) void posconv_weight_split_kernel(const float* __restrict__ w, _Float16* __restrict__ out, float scale) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;            // one (group, out row, k quad)
    const long long total = (long long)PC_G * PC_C * (PC_K * PC_C / 4);
    if (i >= total) return;
    const int kq = (int)(i % (PC_K * PC_C / 4)), row = (int)((i / (PC_K * PC_C / 4)) % PC_C), g = (int)(i / ((long long)PC_C * (PC_K * PC_C / 4)));
    const int k = kq * 4;
    const f4 v = *reinterpret_cast<const f4*>(w + ((long long)g * PC_C + row) * (PC_K * PC_C) + k);
    SchemeNoCheck<SchemeF16x2>::V4 p[2];
    split4<SchemeNoCheck<SchemeF16x2>>(v, scale, p);
    const int step = k >> 5, kb = (k >> 4) & 1, kk = k & 15;
#pragma unroll
    for (int pc = 0; pc < 2; ++pc)
        *reinterpret_cast<f16x4*>(out + ((((long long)g * PC_NSTEPS + step) * 2 + pc) * 2 + kb) * (PC_C * 16) + row * 16 + kk) = p[pc];
}

int launch_posconv_weight_split(const float* w, __bf16* out, float scale, hipStream_t stream) {
    const long long total = (long long)PC_G * PC_C * (PC_K * PC_C / 4);
    hipLaunchKernelGGL(posconv_weight_split_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, w, reinterpret_cast<_Float16*>(out), scale);
    AT_CHECK_HIP(hipGetLastError());
    return 0;
}
size_t posconv_weight_pieces_bytes() { return (size_t)PC_G * PC_NSTEPS * PC_STEP_BYTES; }

__global__ __launch_bounds__(512, 2) void hubert_posconv_kernel(const float* __restrict__ x, const _Float16* __restrict__ wp, const float* __restrict__ bias,
                                                                float* __restrict__ pos, int B, int T, float act_scale, float acc_scale, int* __restrict__ status) {
    typedef SchemeF16x2 SC;
    typedef f16x8 V8;
    extern __shared__ __attribute__((aligned(16))) unsigned char pc_lds[];
    unsigned char* X = pc_lds;                        // [piece][PC_IN_ROWS][112 B]
    unsigned char* Wr = pc_lds + PC_X_BYTES;          // [PC_RING][piece][k-block][48][16] fp16
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntt = (T + PC_ROWS - 1) / PC_ROWS;
    // group slowest: the ~1 500 workgroups of a group stream the same 1.2 MB of weight pieces through L2
    const int tt = (int)(blockIdx.x % ntt), b = (int)((blockIdx.x / ntt) % B), g = (int)(blockIdx.x / ((unsigned)ntt * B));
    const int t0 = tt * PC_ROWS;
    const float* xg = x + (long long)b * T * (PC_G * PC_C) + g * PC_C;
    const _Float16* wg = wp + (long long)g * PC_NSTEPS * (PC_STEP_BYTES / 2);
    RangeMax over;
    // weight ring: step s -> slot s % 4; waves 0-5 move one 1-KB piece each (6 KB per step)
    auto issue_w = [&](int s) {
        if (wave < 6)
            __builtin_amdgcn_global_load_lds((pc_glb_void*)(wg + (long long)s * (PC_STEP_BYTES / 2) + wave * 512 + lane * 8),
                                             (pc_lds_void*)(Wr + (s % PC_RING) * PC_STEP_BYTES + wave * 1024), 16, 0, 0);
    };
    issue_w(0); issue_w(1); issue_w(2);
    // ---- stage the input tile: rows t0 - 64 .. t0 + 319 of this clip (zeros outside [0, T)), split into hi / lo pieces ------------------------------
    for (int q = tid; q < PC_IN_ROWS * (PC_C / 4); q += 512) {
        const int r = q / (PC_C / 4), c = (q - r * (PC_C / 4)) * 4;
        const int t = t0 - PC_PAD + r;
        f4 v = {0.f, 0.f, 0.f, 0.f};
        if (t >= 0 && t < T) v = *reinterpret_cast<const f4*>(xg + (long long)t * (PC_G * PC_C) + c);
        SC::V4 p[2];
        over |= split4<SC>(v, act_scale, p);
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) *reinterpret_cast<f16x4*>(X + pc * (PC_IN_ROWS * PC_XLD) + r * PC_XLD + c * 2) = p[pc];
    }
    // ---- main loop -------------------------------------------------------------------------------------------------------------------------------------
    const int m16 = lane & 15, q4 = lane >> 4;
    f4 acc[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
    // B-operand (input) fragment of K step s, row tile i: lane (m16, q4) reads the 8 channels of 8-group 4 s + q4 = (tap, c8) at row 32 wave + 16 i + m16 + tap.
    // The groups repeat with period 3 steps = 2 taps: offsets for s % 3 = 0, 1, 2 are lane constants, the tap advance is (s / 3) * 2 rows.
    int xoff[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int grp = 4 * r + q4;
        xoff[r] = (wave * 32 + m16 + grp / 6) * PC_XLD + (grp % 6) * 16;
    }
    // A-operand (weight) fragment: lane (n = m16 (+ 16 j), q4): k-block q4 >> 1, half q4 & 1
    const int woff = ((q4 >> 1) * PC_C + m16) * 32 + (q4 & 1) * 16;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int s = 0; s < PC_NSTEPS; ++s) {
        // slot (s + 3) % 4 was read in step s - 1: every wave passed the barrier at the end of that step
        if (s + 3 < PC_NSTEPS) issue_w(s + 3);
        const unsigned char* Ws = Wr + (s % PC_RING) * PC_STEP_BYTES;
        const int xrow = (s / 3) * 2 * PC_XLD + xoff[s % 3];
        V8 wf[2][3], xf[2][2];
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) {
#pragma unroll
            for (int j = 0; j < 3; ++j) wf[pc][j] = *reinterpret_cast<const V8*>(Ws + pc * (2 * PC_C * 32) + j * 16 * 32 + woff);
#pragma unroll
            for (int i = 0; i < 2; ++i) xf[pc][i] = *reinterpret_cast<const V8*>(X + pc * (PC_IN_ROWS * PC_XLD) + i * 16 * PC_XLD + xrow);
        }
#pragma unroll
        for (int t = 0; t < SC::NPROD; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) acc[i][j] = SC::mfma16(wf[SC::prod_w(t)][j], xf[SC::prod_a(t)][i], acc[i][j]);
        // the next step's weights (issued three steps ago by waves 0-5) must have landed, and this step's slot must be free for step s + 4: one barrier
        if (wave < 6) {
            if (s + 3 < PC_NSTEPS) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
    }
    // ---- epilogue: lane holds rows m = t0 + 32 wave + 16 i + m16, channels 16 j + 4 q4 .. + 3 ---------------------------------------------------------
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int t = t0 + wave * 32 + i * 16 + m16;
        if (t >= T) continue;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int c = g * PC_C + j * 16 + q4 * 4;
            const long long o = ((long long)b * T + t) * (PC_G * PC_C) + c;
            const f4 bq = *reinterpret_cast<const f4*>(bias + c);
            const f4 r = *reinterpret_cast<const f4*>(x + o);
            f4 v = acc[i][j] * acc_scale + bq;
            v = f4{gelu_erf(v.x), gelu_erf(v.y), gelu_erf(v.z), gelu_erf(v.w)} + r;
            *reinterpret_cast<f4*>(pos + o) = v;
        }
    }
    range_publish(status, status ? status + 1 : nullptr, over);
}

int launch_hubert_posconv(const float* x, const __bf16* w_pieces, const float* bias, float* pos, int B, int T, float w_scale, int* status, hipStream_t stream) {
    AT_REQUIRE(x && w_pieces && bias && pos && B >= 1 && T >= 1, "hubert_posconv: bad arguments");
    { static LdsAttrFlags lds_attr; if (int rc = set_max_dynamic_lds(lds_attr, hubert_posconv_kernel, PC_LDS)) return rc; }
    const long long blocks = (long long)((T + PC_ROWS - 1) / PC_ROWS) * B * PC_G;
    AT_REQUIRE(blocks < (1ll << 31), "hubert_posconv: grid too large");
    hipLaunchKernelGGL(hubert_posconv_kernel, dim3((unsigned)blocks), dim3(512), PC_LDS, stream, x, reinterpret_cast<const _Float16*>(w_pieces), bias, pos, B, T,
                       XB_F16_ACT_SCALE, 1.0f / (XB_F16_ACT_SCALE * w_scale), status);
    AT_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace at
