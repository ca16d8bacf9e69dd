// The two LSTM layers of the SEANet encoder / decoder as ONE persistent launch, layer 2 one step behind layer 1 (round 4).
//
// lstm_seq_x3.hip runs a layer as 750 dependent steps of ~3.4 us — hand-off latency, not arithmetic — so two layers cost 2 x 750 steps
// and the input projection of layer 2 (a [B T] x 2048 x 512 GEMM) sits between them. Here three ROLES of workgroups run side by side, each
// role a copy of that kernel's decomposition (a group of 16 clips = 16 workgroups of 32 hidden units, weights as fp16 piece pairs in
// registers, h staged into LDS as pieces, line-sized write-through publishes, one flag word per workgroup):
//   role A  layer 1:               h1_t = cell(xg1_t + W_hh1 h1_{t-1})                  waits for the 16 A flags of step t - 1
//   role X  layer-2 input gates:   xg2_t = W_ih2 h1_t + b_ih2  (no recurrence)          waits for the 16 A flags of step t
//   role B  layer 2:               h2_t = cell(xg2_t + W_hh2 h2_{t-1}), y_t = h2 + skip waits for the 16 B flags of step t - 1 and ITS OWN X slice of step t
// The chains overlap: total ~ (T + 2) steps instead of 2 T, and the layer-2 projection GEMM and its split pass disappear. A group needs
// 48 co-resident workgroups (one per CU: 256 registers x 8 waves), so one launch takes at most 5 groups = 80 clips on 256 CUs: this is the
// SMALL-BATCH form (the decoder's 64-clip configuration, short encode batches); larger batches stay on the layer-by-layer kernels, where the
// chip is already full of layer-1 workgroups.
//
// Arithmetic is that of the layer-by-layer path, operation for operation: role X splits h1 with the activation scale of the projection
// GEMM (2^4) and W_ih2 with its finalize-time scale, accumulates the three products of every 32-wide K step in the GEMM's order
// (split_scheme.h prod order, K ascending) and adds the bias after the power-of-two rescale; roles A / B are lstm_seq_x3_kernel<SchemeF16x2>.
// tests/test_acoustic_gpu.py::test_pipelined_lstm_equals_layerwise compares the two paths bit for bit.
#include "gemm_core.h"
#include "encodec_kernels.h"
#include "split_scheme.h"

namespace at {

typedef unsigned int u4 __attribute__((ext_vector_type(4)));

constexpr int LP_H = 512;
constexpr int LP_CLIPS = 16;
constexpr int LP_SLICES = 16;
constexpr int LP_LDH = LP_H + 16;              // LDS row stride of a piece plane (16-bit elements), as lstm_seq_x3.hip
constexpr int LP_HP = LP_CLIPS * LP_LDH;
constexpr int LP_STATUS = 63;
constexpr int LP_FLAGS = 128;                  // flags[role][group][32 words] behind the status words (same region as lstm_seq_x3's)
constexpr int LP_FLAG_STRIDE = 32;
constexpr int LP_MAX_GROUPS = 5;
constexpr int LP_XLD = 132;                    // row stride (floats) of role X's gathered [16 clips][128 gate values]
constexpr float LP_H_SCALE = 16384.0f;         // recurrent products: h (|h| < 1) split as h * 2^14 (lstm_seq_x3.hip)

__global__ __launch_bounds__(512, 1) void lstm_pipe_kernel(LstmPipeArgs a) {
    typedef SchemeF16x2 SC;
    typedef typename SC::T PT;
    typedef typename SC::V8 V8;
    typedef typename SC::V4 V4;
    constexpr int NP = SC::NP, NST = 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char Hp_raw[];   // [2 pieces][16 clips][528]
    PT* Hp = reinterpret_cast<PT*>(Hp_raw);
    __shared__ int abort_s;
    __shared__ __attribute__((aligned(16))) float Hx[LP_CLIPS][32];
    __shared__ __attribute__((aligned(16))) float Hy[LP_CLIPS][32];
    __shared__ __attribute__((aligned(16))) float Xg[LP_CLIPS][LP_XLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, q = lane >> 4;
    const int per_role = a.n_groups * LP_SLICES;
    const int role = blockIdx.x / per_role;                 // 0 = A (layer 1), 1 = X (layer-2 input gates), 2 = B (layer 2)
    const int rem = blockIdx.x - role * per_role;
    const int group = rem % a.n_groups, slice = rem / a.n_groups;
    const int b0 = group * LP_CLIPS;
    const int T = a.T;
    const bool is_x = role == 1, is_b = role == 2;

    const float* w = role == 0 ? a.w_hh1 : is_x ? a.w_ih2 : a.w_hh2;
    const float* bias = role == 0 ? a.b_hh1 : is_x ? a.b_ih2 : a.b_hh2;
    const float w_scale = role == 0 ? a.ws_hh1 : is_x ? a.ws_ih2 : a.ws_hh2;
    const float h_scale = is_x ? a.act_scale : LP_H_SCALE;
    const float acc_scale = 1.0f / (w_scale * h_scale);

    // wave w owns the 16-row tile nt = w of this slice's 8 tiles (rows = unit * 4 + gate in 64-row blocks of 16 units, as lstm_seq.hip)
    const int blk = 2 * slice + (wave >> 2), sub = wave & 3;
    const int unit = blk * 16 + sub * 4 + q;
    V8 wp[NP][16];
    {
        const float* wrow = w + ((long long)blk * 64 + sub * 16 + r16) * LP_H;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const f4 lo = *reinterpret_cast<const f4*>(wrow + ks * 32 + q * 8), hi = *reinterpret_cast<const f4*>(wrow + ks * 32 + q * 8 + 4);
            V4 plo[NP], phi[NP];
            split4<SchemeNoCheck<SC>>(lo, w_scale, plo);
            split4<SchemeNoCheck<SC>>(hi, w_scale, phi);
#pragma unroll
            for (int i = 0; i < NP; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) { wp[i][ks][k] = plo[i][k]; wp[i][ks][4 + k] = phi[i][k]; }
        }
    }
    const int clip = b0 + r16;
    const bool clip_ok = clip < a.B;
    const long long own_row = (long long)(clip_ok ? clip : a.B - 1) * T;
    const f4 bv = *reinterpret_cast<const f4*>(bias + unit * 4);
    float cst = 0.f;
    // staging source: A and X read layer 1's h, B layer 2's; A / B publish into their own layer's buffer
    float* h_own = is_b ? a.h2 : a.h1;
    const __amdgpu_buffer_rsrc_t src = __builtin_amdgcn_make_buffer_rsrc((void*)(is_b ? a.h2 : a.h1), 0, (int)a.h_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t dst = __builtin_amdgcn_make_buffer_rsrc((void*)h_own, 0, (int)a.h_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t xgr = __builtin_amdgcn_make_buffer_rsrc((void*)a.xg2, 0, (int)a.xg_bytes, 0x00020000);
    int g_off[NST];
#pragma unroll
    for (int j = 0; j < NST; ++j) {
        const int row = (tid >> 7) + 4 * j, ch = tid & 127;
        const int cb = b0 + row < a.B ? b0 + row : a.B - 1;
        g_off[j] = (int)((((long long)cb * T) * LP_H + ch * 4) * 4);   // < 2^31: launcher
    }
    const int l_off0 = (tid >> 7) * LP_LDH + (tid & 127) * 4;
    unsigned* flags_base = a.sync + LP_FLAGS;
    unsigned* own_flags = flags_base + (role * LP_MAX_GROUPS + group) * LP_FLAG_STRIDE;
    // whose step counters this role waits for: X follows A's, A and B their own
    const unsigned* wait_flags = flags_base + ((is_b ? 2 : 0) * LP_MAX_GROUPS + group) * LP_FLAG_STRIDE;
    const unsigned* x_flag = flags_base + (1 * LP_MAX_GROUPS + group) * LP_FLAG_STRIDE + slice;
    const PT* hb = Hp + r16 * LP_LDH + q * 8;
    const int xg_lane_off = (int)(((own_row) * (4 * LP_H) + unit * 4) * 4);   // + t * 8192 bytes; < 2^31: launcher

    for (int t = 0; t < T; ++t) {
        f4 xg = {0.f, 0.f, 0.f, 0.f};
        float skipv = 0.f;
        if (role == 0) xg = *reinterpret_cast<const f4*>(a.xg1 + (own_row + t) * (4 * LP_H) + unit * 4);   // independent of the recurrence: before the wait
        if (is_b) skipv = a.skip[(own_row + t) * LP_H + unit];
        __builtin_amdgcn_sched_barrier(0);
        f4 acc = {0.f, 0.f, 0.f, 0.f};
        const bool stage_h = is_x || t > 0;
        if (role != 0 || t > 0) {
            if (wave == 0) {
                // lanes 0..15: the 16 step counters of the chain this role follows (X needs step t of A, i.e. counter >= t + 1);
                // role B, lane 16: its own slice of xg2_t (counter >= t + 1)
                const unsigned* fp = wait_flags + (lane & (LP_SLICES - 1));
                unsigned target = is_x ? (unsigned)(t + 1) : (unsigned)t;
                if (is_b && lane == 16) { fp = x_flag; target = (unsigned)(t + 1); }
                unsigned spins = 0;
                int give_up = 0;
                for (;;) {
                    const unsigned f = __hip_atomic_load(fp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (__builtin_amdgcn_ballot_w64(f < target) == 0ull) break;
                    ++spins;
                    if (spins > a.spin_limit) { give_up = 1; break; }
                    if ((spins & 1023u) == 0u &&
                        __hip_atomic_load(a.sync + LP_STATUS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { give_up = 1; break; }
                }
                if (lane == 0) {
                    abort_s = give_up;
                    if (give_up) __hip_atomic_store(a.sync + LP_STATUS, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            __syncthreads();       // also: every wave has finished reading the previous step's pieces
            if (abort_s) return;
        }
        if (is_b) {   // role X's write-through stores, read past this CU's caches (sc1) like the h tiles
            const u4 v = __builtin_amdgcn_raw_buffer_load_b128(xgr, xg_lane_off + t * (4 * LP_H * 4), 0, 16);
            xg = f4{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
        }
        if (stage_h) {
            u4 stage[NST];
            const int toff = (is_x ? t : t - 1) * (LP_H * 4);
#pragma unroll
            for (int j = 0; j < NST; ++j) stage[j] = __builtin_amdgcn_raw_buffer_load_b128(src, g_off[j] + toff, 0, 16);
#pragma unroll
            for (int j = 0; j < NST; ++j) {
                const f4 hv = {__uint_as_float(stage[j][0]), __uint_as_float(stage[j][1]), __uint_as_float(stage[j][2]), __uint_as_float(stage[j][3])};
                V4 pp[NP];
                split4<SchemeNoCheck<SC>>(hv, h_scale, pp);
                PT* d = Hp + l_off0 + 4 * j * LP_LDH;
#pragma unroll
                for (int i = 0; i < NP; ++i) *reinterpret_cast<V4*>(d + i * LP_HP) = pp[i];
            }
            __syncthreads();
            V8 xa[NP], xb[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) xa[p] = *reinterpret_cast<const V8*>(hb + p * LP_HP);
#pragma unroll
            for (int ks = 0; ks < 16; ks += 2) {
#pragma unroll
                for (int p = 0; p < NP; ++p) xb[p] = *reinterpret_cast<const V8*>(hb + p * LP_HP + (ks + 1) * 32);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int tt = 0; tt < SC::NPROD; ++tt)
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wp[SC::prod_w(tt)][ks], xa[SC::prod_a(tt)], acc, 0, 0, 0);
                if (ks + 2 < 16) {
#pragma unroll
                    for (int p = 0; p < NP; ++p) xa[p] = *reinterpret_cast<const V8*>(hb + p * LP_HP + (ks + 2) * 32);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int tt = 0; tt < SC::NPROD; ++tt)
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wp[SC::prod_w(tt)][ks + 1], xb[SC::prod_a(tt)], acc, 0, 0, 0);
            }
            acc *= acc_scale;   // exact: a power of two
        }
        if (is_x) {
            // ---- xg2_t slice [16 clips][32 units x 4 gates] = product + b_ih2: gathered in LDS, stored as whole 512-byte runs, write-through ----
            *reinterpret_cast<f4*>(&Xg[r16][wave * 16 + q * 4]) = acc + bv;
            __syncthreads();
            {
                const int row = tid >> 5, c4 = tid & 31;
                const u4 v = *reinterpret_cast<const u4*>(&Xg[row][c4 * 4]);
                if (b0 + row < a.B)
                    __builtin_amdgcn_raw_buffer_store_b128(v, xgr, (int)((((long long)(b0 + row) * T + t) * (4 * LP_H) + slice * 128 + c4 * 4) * 4), 0, 16);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_store(own_flags + slice, (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            continue;
        }
        // ---- cell update (torch CPU LSTMCell order, as lstm_seq_x3.hip) ----
        {
            const f4 g = (acc + bv) + xg;
            const float ig = lstm_sigmoid(g.x), fg = lstm_sigmoid(g.y), cg = lstm_tanh(g.z), og = lstm_sigmoid(g.w);
            const float c_new = __fadd_rn(__fmul_rn(fg, cst), __fmul_rn(ig, cg));
            const float hn = og * lstm_tanh(c_new);
            cst = c_new;
            Hx[r16][wave * 4 + q] = hn;
            if (is_b) {
                const float yv = hn + skipv;
                Hy[r16][wave * 4 + q] = a.y_elu ? elu1(yv) : yv;
            }
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int idx = lane + 64 * i, row = idx >> 3, c4 = idx & 7;
                const u4 v = *reinterpret_cast<const u4*>(&Hx[row][c4 * 4]);
                if (b0 + row < a.B)
                    __builtin_amdgcn_raw_buffer_store_b128(v, dst, (((b0 + row) * T + t) * LP_H + slice * 32 + c4 * 4) * 4, 0, 16);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(own_flags + slice, (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (is_b && wave == 1) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int idx = lane + 64 * i, row = idx >> 3, c4 = idx & 7;
                if (b0 + row < a.B)
                    *reinterpret_cast<u4*>(a.y_out + ((long long)(b0 + row) * T + t) * LP_H + slice * 32 + c4 * 4) = *reinterpret_cast<const u4*>(&Hy[row][c4 * 4]);
            }
        }
    }
}

// clips one pipelined launch can take: 48 co-resident workgroups (one per CU) per group of 16 clips
int lstm_pipe_max_clips() {
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (cached[dev] == 0) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
        int groups = cus / (3 * LP_SLICES);
        groups = groups > LP_MAX_GROUPS ? LP_MAX_GROUPS : groups;
        cached[dev] = groups < 1 ? -1 : groups * LP_CLIPS;
    }
    return cached[dev] < 0 ? 0 : cached[dev];
}

bool lstm_pipe_eligible(int B, int T) {
    return B >= 1 && T >= 1 && B <= lstm_pipe_max_clips() && (long long)B * T * 4 * LP_H * 4 < (1ll << 31);
}

int launch_lstm_pipe(const LstmPipeArgs& a_in, hipStream_t stream) {
    LstmPipeArgs a = a_in;
    AT_REQUIRE(lstm_pipe_eligible(a.B, a.T), "lstm_pipe: batch does not fit one pipelined launch");
    AT_REQUIRE(a.ws_hh1 > 0.f && a.ws_ih2 > 0.f && a.ws_hh2 > 0.f && a.act_scale > 0.f, "lstm_pipe: operand scales missing");
    a.n_groups = (a.B + LP_CLIPS - 1) / LP_CLIPS;
    a.h_bytes = (long long)a.B * a.T * LP_H * 4;
    a.xg_bytes = a.h_bytes * 4;
    AT_CHECK_HIP(hipMemsetAsync(a.sync + LP_FLAGS, 0, 3 * LP_MAX_GROUPS * LP_FLAG_STRIDE * sizeof(unsigned), stream));
    const size_t lds = (size_t)2 * LP_HP * 2;
    { static LdsAttrFlags lds_attr; if (int rc = set_max_dynamic_lds(lds_attr, lstm_pipe_kernel, lds)) return rc; }
    hipLaunchKernelGGL(lstm_pipe_kernel, dim3(3 * a.n_groups * LP_SLICES), dim3(512), lds, stream, a);
    AT_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace at
