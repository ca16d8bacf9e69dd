// Whole-sequence LSTM layer in ONE persistent launch (EnCodec's 2 x LSTM(512), reference call site
// audiotoken/encoder.py:48 -> encodec SLSTM; HF restatement modeling_encodec.py:236-249).
//
// Why: the recurrence is 750 dependent steps of a [B,512]x[512,2048] product. As one launch per step it is bound
// by launch/dependency latency (measured 21-36 us per step for 3.4 us of MFMA work at B = 256).
//
// Decomposition (MI355X: 8 XCDs x 32 CUs, 512 registers per lane at one wave per SIMD, 160 KB LDS per CU):
//   * clips are cut into groups of 16 (one MFMA row tile); a group never talks to another group;
//   * inside a group, 32 workgroups each own 16 hidden units = 64 interleaved gate rows of W_hh. Those rows stay
//     RESIDENT IN REGISTERS, already in MFMA A-fragment order (wave w: gate rows 16w..16w+15 x 512 k = 128 registers
//     per lane) — W_hh is read from HBM exactly once and costs no LDS bandwidth per step;
//   * per step the group's h_{t-1} [16 x 512] (32 KB) is staged once into LDS and every wave streams its B
//     fragments from there (one ds_read_b128 per 4 MFMAs); 128 MFMAs per wave and step;
//   * ~200 registers and 32 KB of LDS put TWO workgroups — of two different groups — on a CU (16 groups x 32 = 512
//     workgroups at B = 256): while one waits for its group's hand-off (three dependent memory round trips, ~3 us)
//     the other one owns the MFMA pipe. With 32-clip groups and one workgroup per CU the pipe idled for that time
//     (7.9 us per step against 3.4 us of MFMA work);
//   * the cell update runs on the (clip, unit) pairs a lane's accumulators own — the cell state c never leaves
//     registers — and the workgroup publishes its 32 x 16 slice of h_t.
// Group id = blockIdx % n_groups: with the observed round-robin dispatch the 32 members share one XCD and the
// exchanged h (32 KB per step) stays close to it. That placement is a speed assumption only (rotating the second half of
// the grid against the first so that co-resident workgroups are guaranteed to differ in group measured 10 % slower).
//
// Hand-off protocol (placement independent; CDNA guide G16 recipe R1 / MI355X_MICROARCH "Valid forms" row 1, with a
// per-producer flag word instead of a shared counter — a write-through store is acknowledged ~3x sooner than a
// memory-side atomic RMW: measured ~580 vs ~1900 cycles):
//   producer: h stores are write-through (sc1, agent-scope relaxed atomic stores) -> every wave s_waitcnt vmcnt(0)
//             -> workgroup barrier -> ONE lane: agent-scope store of (t + 1) to flags[group][slice];
//   consumer: ONE wave polls the group's 32 flags (one 128-byte line, lane i <-> slice i, sc1 loads, bounded spin)
//             -> workgroup barrier -> every load of the handed-off h is an sc1 buffer load to registers
//             (bypasses the CU's L1, never a plain load).
// Flags are zeroed by a memset node in front of every launch. A spin that exceeds its bound sets the status word
// and the workgroup stops (wrong output, reported by at_encodec_status; never a hang).
#include "gemm_core.h"
#include "encodec_kernels.h"

namespace at {

constexpr int LS_H = 512;          // hidden size
constexpr int LS_CLIPS = 16;       // clips per group = one MFMA row tile
constexpr int LS_SLICES = 32;      // workgroups per group
constexpr int LS_ROWS = 64;        // gate rows per workgroup = 16 hidden units x 4 gates
constexpr int LS_KG = LS_H / 16;   // 32 k-groups of 16
constexpr int LS_H_FLOATS = LS_CLIPS * LS_H;   // 8192 floats = 32 KB
constexpr int LS_STATUS = 63;                  // word of a.sync that reports a timed-out wait (sticky for the encode call)
constexpr int LS_FLAGS = 128;                  // word offset of flags[16 groups][32 slices] in a.sync (1024 words)
constexpr int LS_MAX_GROUPS = 16;

typedef unsigned int u4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 2) void lstm_seq_kernel(LstmSeqArgs a) {
    extern __shared__ __attribute__((aligned(16))) float Hs[];   // [16 clips][512], 16-B chunk ^= clip & 15
    __shared__ int abort_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, q = lane >> 4;
    const int group = blockIdx.x % a.n_groups;
    const int slice = blockIdx.x / a.n_groups;     // 0..31: hidden units [16*slice, 16*slice + 16)
    const int b0 = group * LS_CLIPS;
    const int T = a.T;

    // lane ownership: wave -> gate tile (4 hidden units x 4 gates); acc[reg]: clip = b0 + r16, unit = slice*16 + wave*4 + q,
    // gate = reg (i, f, g, o)
    // ---- this wave's W_hh rows -> registers, once: wr[kg] = W[wave*16 + r16][kg*16 + q*4 .. +3] -------------------------
    f4 wr[LS_KG];
#pragma unroll
    for (int kg = 0; kg < LS_KG; ++kg)
        wr[kg] = *reinterpret_cast<const f4*>(a.w_hh + ((long long)slice * LS_ROWS + wave * 16 + r16) * LS_H + kg * 16 + q * 4);

    const int clip = b0 + r16;
    const bool clip_ok = clip < a.B;
    const long long own_row = (long long)(clip_ok ? clip : a.B - 1) * T;
    const int unit = slice * 16 + wave * 4 + q;
    const int col = unit * 4;
    const f4 bhh = *reinterpret_cast<const f4*>(a.b_hh + col);
    float cst = 0.f;
    // h_{t-1} staging: thread -> 8 x (clip row, 16-B chunk) of the [16][512] tile: e = tid + 256*j -> row = e >> 7,
    // chunk = tid & 127. Rows beyond B are clamped (their gates are computed on a copy of the last clip, never stored).
    const __amdgpu_buffer_rsrc_t hrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.h_out, 0, (int)a.h_bytes, 0x00020000);
    int g_off[8], l_off[8];   // byte offset of (row, chunk) at t = 0 in h_out; float offset in Hs
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int row = (tid >> 7) + 2 * j, ch = tid & 127;
        const int cb = b0 + row < a.B ? b0 + row : a.B - 1;
        g_off[j] = (int)((((long long)cb * T) * LS_H + ch * 4) * 4);   // < 2^31: checked by the launcher
        l_off[j] = row * LS_H + ((ch ^ (row & 15)) << 2);
    }
    unsigned* flags = a.sync + LS_FLAGS + group * LS_SLICES;
    // fragment address of k-group kg = 4m + r: chunk (kg*4 + q) ^ r16 = ((4m + (r ^ c>>2)) << 2) | (c & 3), c = q ^ r16
    const float* hp[4];
    {
        const int c = q ^ r16;
#pragma unroll
        for (int r = 0; r < 4; ++r) hp[r] = Hs + r16 * LS_H + ((((r ^ (c >> 2)) << 2) | (c & 3)) << 2);
    }

    for (int t = 0; t < T; ++t) {
        // input-side gates and the skip input of this step: independent of the recurrence, issued before the wait
        const f4 xg = *reinterpret_cast<const f4*>(a.xg + (own_row + t) * (4 * LS_H) + col);
        float skipv = 0.f;
        if (a.y_out) skipv = a.skip[(own_row + t) * LS_H + unit];
        __builtin_amdgcn_sched_barrier(0);   // keep these loads in front of the wait: their latency hides behind it
        f4 acc = {0.f, 0.f, 0.f, 0.f};
        if (t > 0) {
            // ---- wait until all 32 slices of this group have published h_{t-1} ---------------------------
            if (wave == 0) {
                const unsigned target = (unsigned)t;
                unsigned spins = 0;
                int give_up = 0;
                for (;;) {
                    const unsigned f = __hip_atomic_load(flags + (lane & (LS_SLICES - 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (__builtin_amdgcn_ballot_w64(f < target) == 0ull) break;
                    ++spins;
                    if (spins > a.spin_limit) { give_up = 1; break; }
                    if ((spins & 1023u) == 0u &&
                        __hip_atomic_load(a.sync + LS_STATUS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { give_up = 1; break; }
                }
                if (lane == 0) {
                    abort_s = give_up;
                    if (give_up) __hip_atomic_store(a.sync + LS_STATUS, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            __syncthreads();       // also: every wave has finished reading the previous step's Hs
            if (abort_s) return;   // uniform: a member of the group is not making progress (e.g. not resident)
            // ---- h_{t-1} [32][512] -> LDS: sc1 loads straight to registers, then ds_write ---------------------------
            u4 stage[8];
            const int toff = (t - 1) * (LS_H * 4);
#pragma unroll
            for (int j = 0; j < 8; ++j) stage[j] = __builtin_amdgcn_raw_buffer_load_b128(hrsrc, g_off[j] + toff, 0, 16);   // aux 16 = sc1
#pragma unroll
            for (int j = 0; j < 8; ++j) *reinterpret_cast<u4*>(Hs + l_off[j]) = stage[j];
            __syncthreads();
            // ---- gates += h_{t-1} . W_slice^T : 32 k-groups x 4 k-steps = 128 MFMAs per wave (one accumulator chain; the
            //      co-resident workgroup's wave fills the dependent-issue gaps) -----------------------------------------------
            // B fragments are fetched one k-group ahead so the LDS latency hides behind the 8 MFMAs in flight
            f4 hb = *reinterpret_cast<const f4*>(hp[0]);
#pragma unroll
            for (int kg = 0; kg < LS_KG; ++kg) {
                f4 hbn = hb;
                if (kg + 1 < LS_KG) hbn = *reinterpret_cast<const f4*>(hp[(kg + 1) & 3] + ((kg + 1) >> 2) * 64);
                __builtin_amdgcn_sched_barrier(0);   // hipcc otherwise re-serialises read -> wait -> MFMAs on one register set
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[kg][e], hb[e], acc, 0, 0, 0);
                hb = hbn;
            }
        }
        // ---- cell update (torch CPU LSTMCell order: gates = (hW + b_hh) + igates; c = f*c + i*g unfused) ---------
        float hn;
        {
            const f4 g = (acc + bhh) + xg;
            const float ig = lstm_sigmoid(g.x), fg = lstm_sigmoid(g.y), cg = lstm_tanh(g.z), og = lstm_sigmoid(g.w);
            const float c_new = __fadd_rn(__fmul_rn(fg, cst), __fmul_rn(ig, cg));
            hn = og * lstm_tanh(c_new);
            cst = c_new;
            if (clip_ok)
                __hip_atomic_store(reinterpret_cast<unsigned*>(a.h_out) + (own_row + t) * LS_H + unit, __float_as_uint(hn),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // write-through (sc1): no release fence needed
        }
        // ---- publish: every storing wave drains, workgroup barrier, one lane signals ------------------------------
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(flags + slice, (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // ---- y = h + skip: not part of the recurrence, so it goes out after the hand-off ------------------------------
        if (a.y_out && clip_ok) {
            const float yv = hn + skipv;
            a.y_out[(own_row + t) * LS_H + unit] = a.y_elu ? elu1(yv) : yv;
        }
    }
}

// All workgroups of a launch must be resident at once (they wait for each other): the number of groups per launch follows
// from what the CURRENT device can hold — CUs x resident workgroups per CU (occupancy query) / 32 slices — so a
// partitioned or CU-masked GPU gets smaller launches instead of a hand-off timeout. 256 CUs x 2 -> 16 groups = 256 clips.
int lstm_seq_max_clips() {
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (cached[dev] == 0) {
        int cus = 0, per_cu = 0;
        const size_t lds = (size_t)LS_H_FLOATS * sizeof(float);
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_seq_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(lstm_seq_kernel), 256, lds) != hipSuccess) per_cu = 1;
        int groups = (cus * (per_cu < 1 ? 1 : per_cu)) / LS_SLICES;
        groups = groups > LS_MAX_GROUPS ? LS_MAX_GROUPS : groups;
        cached[dev] = (groups < 1 ? 1 : groups) * LS_CLIPS;   // < 32 resident workgroups cannot run even one group: the status word reports it
    }
    return cached[dev];
}

int launch_lstm_seq(const LstmSeqArgs& a_in, hipStream_t stream) {
    LstmSeqArgs a = a_in;
    AT_REQUIRE(a.B >= 1 && a.B <= lstm_seq_max_clips() && a.T >= 1, "lstm_seq: too many clips for one launch on this device");
    a.n_groups = (a.B + LS_CLIPS - 1) / LS_CLIPS;
    a.h_bytes = (long long)a.B * a.T * LS_H * 4;
    AT_REQUIRE(a.h_bytes < (1ll << 31), "lstm_seq: h buffer exceeds the 2 GB buffer-descriptor range");
    const size_t lds = (size_t)LS_H_FLOATS * sizeof(float);
    { static LdsAttrFlags lds_attr_0; if (int rc = set_max_dynamic_lds(lds_attr_0, lstm_seq_kernel, lds)) return rc; }
    AT_CHECK_HIP(hipMemsetAsync(a.sync + LS_FLAGS, 0, LS_MAX_GROUPS * LS_SLICES * sizeof(unsigned), stream));   // flags, every launch
    hipLaunchKernelGGL(lstm_seq_kernel, dim3(a.n_groups * LS_SLICES), dim3(256), lds, stream, a);
    AT_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace at
