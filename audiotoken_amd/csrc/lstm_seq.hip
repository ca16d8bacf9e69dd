// Whole-sequence LSTM layer in ONE persistent launch (EnCodec's 2 x LSTM(512), reference call site
// audiotoken/encoder.py:48 -> encodec SLSTM; HF restatement modeling_encodec.py:236-249).
//
// Why: the recurrence is 750 dependent steps of a [B,512]x[512,2048] product. As one launch per step it is bound
// by launch/dependency latency (measured 21-36 us per step for 3.4 us of MFMA work at B = 256).
//
// Decomposition (MI355X: 8 XCDs x 32 CUs, 160 KB LDS per CU):
//   * clips are cut into groups of 32; a group never talks to another group;
//   * inside a group, 32 workgroups each own 16 hidden units = 64 interleaved gate rows of W_hh
//     (64 x 512 fp32 = 128 KB) which stay RESIDENT IN LDS for all T steps — W_hh is read from HBM exactly once;
//   * per step a workgroup multiplies its rows with the group's h_{t-1} [32 x 512] on the fp32 MFMA
//     (256 MFMAs per wave = the fp32 peak for this work), applies the cell update to the (clip, unit) pairs its
//     lanes own — the cell state c never leaves registers — and publishes its 32 x 16 slice of h_t;
//   * the 32 workgroups of a group meet once per step at a monotonic counter.
// Group id = blockIdx % n_groups: with the observed round-robin dispatch the 32 members share one XCD and the
// exchanged h (64 KB per step) stays in that XCD's L2. That placement is a speed assumption only.
//
// Hand-off protocol (placement independent; CDNA guide G16 recipe R1 / MI355X_MICROARCH "Valid forms" row 1):
//   producer: h stores are write-through (sc1, agent-scope relaxed atomic stores) -> every wave s_waitcnt vmcnt(0)
//             -> workgroup barrier -> ONE lane: agent-scope atomic add on the group counter;
//   consumer: ONE wave polls the counter with sc1 loads (bounded spin) -> workgroup barrier -> every load of the
//             handed-off h is an sc1 buffer load to registers (bypasses the CU's L1, never a plain load).
// Counters are zeroed by a memset node in front of every launch. A spin that exceeds its bound sets status[0] = 1
// and the workgroup carries on (wrong output, reported by at_encodec_status; never a hang).
#include "gemm_core.h"
#include "encodec_kernels.h"

namespace at {

constexpr int LS_H = 512;          // hidden size
constexpr int LS_CLIPS = 32;       // clips per group
constexpr int LS_SLICES = 32;      // workgroups per group
constexpr int LS_ROWS = 64;        // gate rows per workgroup = 16 hidden units x 4 gates
constexpr int LS_KP = 128;         // k extent of one h phase in LDS
constexpr int LS_W_FLOATS = LS_ROWS * LS_H;        // 32768 floats = 128 KB
constexpr int LS_H_FLOATS = LS_CLIPS * LS_KP;      // 4096 floats  = 16 KB
constexpr unsigned LS_SPIN_LIMIT = 1u << 18;   // ~0.1-0.3 s of polling; normal waits are microseconds
constexpr int LS_STATUS = 63;                  // word of a.sync that reports a timed-out wait (sticky for the encode call)

typedef unsigned int u4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 1) void lstm_seq_kernel(LstmSeqArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ int abort_s;
    float* Ws = smem;                  // [64 rows][512], 16-B chunk ^= row & 15
    float* Hs = smem + LS_W_FLOATS;    // [32 clips][128], 16-B chunk ^= clip & 15
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, q = lane >> 4;
    const int group = blockIdx.x % a.n_groups;
    const int slice = blockIdx.x / a.n_groups;     // 0..31: hidden units [16*slice, 16*slice + 16)
    const int b0 = group * LS_CLIPS;
    const int T = a.T;

    // ---- W_hh slice -> LDS, once -------------------------------------------------------------------------
    {
        const float* wsrc = a.w_hh + (long long)slice * LS_ROWS * LS_H;
        for (int e = tid; e < LS_ROWS * (LS_H / 4); e += 256) {
            const int row = e >> 7, ch = e & 127;
            const f4 v = *reinterpret_cast<const f4*>(wsrc + row * LS_H + ch * 4);
            *reinterpret_cast<f4*>(Ws + row * LS_H + ((ch ^ (row & 15)) << 2)) = v;
        }
    }
    // lane ownership: wave -> clip tile mi and gate tiles {2*nh, 2*nh+1}; acc[jn][reg]: clip = b0 + mi*16 + r16,
    // unit = slice*16 + (2*nh + jn)*4 + q, gate = reg (i, f, g, o)
    const int mi = wave >> 1, nh = wave & 1;
    const int clip = b0 + mi * 16 + r16;
    const bool clip_ok = clip < a.B;
    const int clip_c = clip_ok ? clip : a.B - 1;
    int unit[2], col[2];
    f4 bhh[2];
    float cst[2] = {0.f, 0.f};
#pragma unroll
    for (int jn = 0; jn < 2; ++jn) {
        unit[jn] = slice * 16 + (2 * nh + jn) * 4 + q;
        col[jn] = unit[jn] * 4;
        bhh[jn] = *reinterpret_cast<const f4*>(a.b_hh + col[jn]);
    }
    // h_{t-1} staging: thread -> 4 x (clip row, 16-B chunk) of the [32][128] phase tile
    const __amdgpu_buffer_rsrc_t hrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.h_out, 0, (int)a.h_bytes, 0x00020000);
    int st_row[4], st_ch[4];
    bool st_ok[4];
    long long st_base[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int e = tid + 256 * j;
        st_row[j] = e >> 5;
        st_ch[j] = e & 31;
        const int cb = b0 + st_row[j];
        st_ok[j] = cb < a.B;
        st_base[j] = ((long long)(st_ok[j] ? cb : a.B - 1) * T) * LS_H + st_ch[j] * 4;
    }
    unsigned* counter = a.sync + group;
    __syncthreads();

    for (int t = 0; t < T; ++t) {
        // input-side gates for this step: independent of the recurrence, issued before the wait
        f4 xg[2];
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
            xg[jn] = *reinterpret_cast<const f4*>(a.xg + ((long long)clip_c * T + t) * (4 * LS_H) + col[jn]);
        f4 acc[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
        if (t > 0) {
            // ---- wait until all 32 slices of this group have published h_{t-1} ---------------------------
            if (wave == 0) {
                const unsigned target = (unsigned)LS_SLICES * (unsigned)t;
                unsigned spins = 0;
                int give_up = 0;
                while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                    ++spins;
                    if (spins > LS_SPIN_LIMIT) { give_up = 1; break; }
                    if ((spins & 1023u) == 0u &&
                        __hip_atomic_load(a.sync + LS_STATUS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { give_up = 1; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (lane == 0) {
                    abort_s = give_up;
                    if (give_up) __hip_atomic_store(a.sync + LS_STATUS, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            __syncthreads();
            if (abort_s) return;   // uniform: a member of the group is not making progress (e.g. not resident)
            // ---- gates += h_{t-1} . W_slice^T, 4 phases of 128 k ---------------------------------------------
            u4 stage[4];
            auto load_phase = [&](int ph) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const long long off = (st_base[j] + (long long)(t - 1) * LS_H + ph * LS_KP) * 4;
                    stage[j] = __builtin_amdgcn_raw_buffer_load_b128(hrsrc, (int)off, 0, 16);   // aux 16 = sc1
                }
            };
            load_phase(0);
            for (int ph = 0; ph < 4; ++ph) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    u4 v = stage[j];
                    if (!st_ok[j]) v = u4{0u, 0u, 0u, 0u};
                    *reinterpret_cast<u4*>(Hs + st_row[j] * LS_KP + ((st_ch[j] ^ (st_row[j] & 15)) << 2)) = v;
                }
                __syncthreads();
                if (ph + 1 < 4) load_phase(ph + 1);   // next phase's loads fly during this phase's MFMAs
                const float* hrow = Hs + (mi * 16 + r16) * LS_KP;
#pragma unroll
                for (int kg = 0; kg < 8; ++kg) {
                    const f4 hb = *reinterpret_cast<const f4*>(hrow + ((((kg << 2) + q) ^ r16) << 2));
                    f4 wa[2];
#pragma unroll
                    for (int jn = 0; jn < 2; ++jn) {
                        const int row = (2 * nh + jn) * 16 + r16;
                        wa[jn] = *reinterpret_cast<const f4*>(Ws + row * LS_H + (((((ph * 8 + kg) << 2) + q) ^ r16) << 2));
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int jn = 0; jn < 2; ++jn)
                            acc[jn] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[jn][e], hb[e], acc[jn], 0, 0, 0);
                }
                __syncthreads();   // Hs is rewritten by the next phase / next step
            }
        }
        // ---- cell update (torch CPU LSTMCell order: gates = (hW + b_hh) + igates; c = f*c + i*g unfused) ---------
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) {
            const f4 g = (acc[jn] + bhh[jn]) + xg[jn];
            const float ig = sigmoidf_(g.x), fg = sigmoidf_(g.y), cg = tanhf(g.z), og = sigmoidf_(g.w);
            const float c_new = __fadd_rn(__fmul_rn(fg, cst[jn]), __fmul_rn(ig, cg));
            const float h_new = og * tanhf(c_new);
            cst[jn] = c_new;
            if (clip_ok) {
                const long long oi = ((long long)clip * T + t) * LS_H + unit[jn];
                __hip_atomic_store(reinterpret_cast<unsigned*>(a.h_out) + oi, __float_as_uint(h_new), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);   // write-through (sc1): no release fence needed
                if (a.y_out) { const float yv = h_new + a.skip[oi]; a.y_out[oi] = a.y_elu ? elu1(yv) : yv; }
            }
        }
        // ---- publish: every storing wave drains, workgroup barrier, one lane signals ------------------------------
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

int lstm_seq_max_clips() { return 8 * LS_CLIPS; }

int launch_lstm_seq(const LstmSeqArgs& a_in, hipStream_t stream) {
    LstmSeqArgs a = a_in;
    AT_REQUIRE(a.B >= 1 && a.B <= lstm_seq_max_clips() && a.T >= 1, "lstm_seq: 1..256 clips per launch");
    a.n_groups = (a.B + LS_CLIPS - 1) / LS_CLIPS;
    a.h_bytes = (long long)a.B * a.T * LS_H * 4;
    AT_REQUIRE(a.h_bytes < (1ll << 31), "lstm_seq: h buffer exceeds the 2 GB buffer-descriptor range");
    const size_t lds = (size_t)(LS_W_FLOATS + LS_H_FLOATS) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        AT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_seq_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    AT_CHECK_HIP(hipMemsetAsync(a.sync, 0, 32 * sizeof(unsigned), stream));   // group counters, every launch
    hipLaunchKernelGGL(lstm_seq_kernel, dim3(a.n_groups * LS_SLICES), dim3(256), lds, stream, a);
    AT_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace at
