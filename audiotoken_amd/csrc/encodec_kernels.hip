// Kernels specific to the acoustic tokenizer (EnCodec 24 kHz): first conv (Cin = 1), fused LSTM step,
// residual-VQ search, RVQ decode (embed-sum), last decoder conv (Cout = 1).
// Everything GEMM-shaped goes through gemm_core.h; see encodec.cpp for the orchestration.
#include "gemm_core.h"
#include "encodec_kernels.h"

namespace at {

// ------------------------------------------------------------------------------------------------------
// conv0: [B][N] waveform -> [B][N][32], k = 7, causal (left reflect pad 6). HBM-bound: 4 B in, 128 B out
// per sample. One thread = one sample x 4 channels -> float4 store; 8 consecutive threads write one 128-B row.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv0_kernel(const float* __restrict__ wav, const float* __restrict__ w /*[32][7]*/,
                                                    const float* __restrict__ bias, float* __restrict__ out, int N, long long total) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total) return;
    const int cg = (int)(gid & 7);
    const long long bt = gid >> 3;
    const int t = (int)(bt % N);
    const float* x = wav + (bt - t);
    float xv[7];
#pragma unroll
    for (int tap = 0; tap < 7; ++tap) {
        int r = t + tap - 6;
        if (r < 0) r = -r;
        xv[tap] = x[r];
    }
    f4 o;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float* wc = w + (cg * 4 + c) * 7;
        float acc = 0.f;
#pragma unroll
        for (int tap = 0; tap < 7; ++tap) acc = fmaf(wc[tap], xv[tap], acc);
        o[c] = acc + bias[cg * 4 + c];
    }
    *reinterpret_cast<f4*>(out + bt * 32 + cg * 4) = o;
}

int launch_conv0(const float* wav, const float* w, const float* bias, float* out, int B, int N, hipStream_t stream) {
    const long long total = (long long)B * N * 8;
    const long long blocks = (total + 255) / 256;
    hipLaunchKernelGGL(conv0_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, wav, w, bias, out, N, total);
    AT_CHECK_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------------
// LSTM step: gates[B][2048] = h_{t-1} W_hh^T + b_hh + xg[:, t, :]; pointwise cell update fused as the epilogue.
// Gate columns are interleaved at load time (column 4*j + g, g in i,f,g,o) so that the 4 accumulator
// registers of a lane are the 4 gates of hidden unit j of one clip.
// Follows torch's CPU LSTMCell arithmetic: gates = linear_hh(h) + igates; c = f*c + i*g (two rounded
// products, one add — no fused multiply-add); h = o * tanh(c).
// ------------------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void lstm_step_kernel(GemmArgs a, LstmStepArgs s) {
    using Tile = GemmTile<BM, BN, WM, WN>;
    constexpr int TM = Tile::TM, TN = Tile::TN;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    f4 acc[TM][TN];
    Tile::run(a, smem, m0, n0, 0, acc);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int r16 = lane & 15, q = lane >> 4;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int b = m0 + wm * TM * 16 + i * 16 + r16;
        if (b >= a.M) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * TN * 16 + j * 16 + q * 4;
            if (n >= a.N) continue;
            const int unit = n >> 2;
            f4 hg = acc[i][j] + *reinterpret_cast<const f4*>(s.b_hh + n);
            f4 g = hg + *reinterpret_cast<const f4*>(s.xg + ((long long)b * s.T + s.t) * (4 * s.H) + n);
            const float ig = lstm_sigmoid(g.x), fg = lstm_sigmoid(g.y), cg = lstm_tanh(g.z), og = lstm_sigmoid(g.w);
            const long long ci = (long long)b * s.H + unit;
            const float c_prev = s.first ? 0.f : s.c[ci];
            const float c_new = __fadd_rn(__fmul_rn(fg, c_prev), __fmul_rn(ig, cg));
            const float h_new = og * lstm_tanh(c_new);
            s.c[ci] = c_new;
            const long long oi = ((long long)b * s.T + s.t) * s.H + unit;
            s.h_out[oi] = h_new;
            if (s.y_out) { const float yv = h_new + s.skip[oi]; s.y_out[oi] = s.y_elu ? elu1(yv) : yv; }
        }
    }
}

int launch_lstm_step(const GemmArgs& a, const LstmStepArgs& s, hipStream_t stream) {
    if (int rc = check_gemm_args(a)) return rc;
    GemmArgs g = a;
    if (s.first) g.K = 0;  // h_{-1} = 0: no recurrent product at t = 0
    constexpr int BM = 64, BN = 64;
    using Tile = GemmTile<BM, BN, 2, 2>;
    dim3 grid((g.M + BM - 1) / BM, (g.N + BN - 1) / BN, 1);
    hipLaunchKernelGGL((lstm_step_kernel<BM, BN, 2, 2>), grid, dim3(256), Tile::LDS_BYTES, stream, g, s);
    AT_CHECK_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------------
// Residual VQ search, all n_q stages in one launch.
//   per stage: dist[n] = -((|r|^2 - 2 r.E_n) + |E_n|^2), idx = first argmax, r -= E[idx]
// (encodec EuclideanCodebook.quantize; HF restatement modeling_encodec.py:364-369, 424-438).
// One workgroup = 128 frames, one wave = 32 frames. A wave keeps its frames' residual in registers in MFMA
// B-operand fragment order for the whole kernel (64 VGPRs): no activation ever returns to memory between
// stages. Codebook tiles (64 codes x 128 dims, 32 KB) stream L2 -> registers -> LDS, double-buffered, shared
// by the 4 waves. Dot products on the f32 MFMA (exact fp32), distance formula in the reference's operation
// order, running (max, first index) per lane, then a 2-step cross-quad reduction with index tie-break.
// ------------------------------------------------------------------------------------------------------
constexpr int RVQ_D = 128;
constexpr int RVQ_CODES = 1024;
constexpr int RVQ_CT = 64;                 // codes per LDS tile
constexpr int RVQ_ROWS = 128;              // frames per workgroup
constexpr int RVQ_TILE_FLOATS = RVQ_CT * RVQ_D;

__global__ __launch_bounds__(256) void rvq_encode_kernel(const float* __restrict__ x, long long rows, int T,
                                                         const float* __restrict__ codebooks, const float* __restrict__ e2,
                                                         int n_q, int16_t* __restrict__ codes) {
    extern __shared__ __attribute__((aligned(16))) float smem[];  // [2][64][128], chunk ^= row&15
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, q = lane >> 4;
    const long long row_base = (long long)blockIdx.x * RVQ_ROWS + wave * 32;

    // residual fragments: xr[i][kg] = x[row_i][kg*16 + q*4 .. +3]
    f4 xr[2][8];
    long long rowi[2];
    bool valid[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        long long r = row_base + i * 16 + r16;
        valid[i] = r < rows;
        rowi[i] = valid[i] ? r : rows - 1;
#pragma unroll
        for (int kg = 0; kg < 8; ++kg) xr[i][kg] = *reinterpret_cast<const f4*>(x + rowi[i] * RVQ_D + kg * 16 + q * 4);
    }

    const int st_chunk = tid & 31;  // staging: thread -> (row = tid>>5 + 8j, chunk)
    const int st_row = tid >> 5;
    f4 stage_reg[8];

    for (int stage = 0; stage < n_q; ++stage) {
        const float* E = codebooks + (long long)stage * RVQ_CODES * RVQ_D;
        const float* e2s = e2 + stage * RVQ_CODES;
        // |r|^2 per frame
        float s2[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float p = 0.f;
#pragma unroll
            for (int kg = 0; kg < 8; ++kg) {
                p = fmaf(xr[i][kg].x, xr[i][kg].x, p); p = fmaf(xr[i][kg].y, xr[i][kg].y, p);
                p = fmaf(xr[i][kg].z, xr[i][kg].z, p); p = fmaf(xr[i][kg].w, xr[i][kg].w, p);
            }
            p += __shfl_xor(p, 16);
            p += __shfl_xor(p, 32);
            s2[i] = p;
        }
        float best[2] = {-INFINITY, -INFINITY};
        int bidx[2] = {0, 0};

        auto load_codes = [&](int tile) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                stage_reg[j] = *reinterpret_cast<const f4*>(E + ((long long)tile * RVQ_CT + st_row + 8 * j) * RVQ_D + st_chunk * 4);
        };
        auto store_codes = [&](int buf) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int row = st_row + 8 * j;
                *reinterpret_cast<f4*>(smem + buf * RVQ_TILE_FLOATS + row * RVQ_D + ((st_chunk ^ (row & 15)) << 2)) = stage_reg[j];
            }
        };
        constexpr int NT = RVQ_CODES / RVQ_CT;
        __syncthreads();  // previous stage's readers are done with both buffers
        load_codes(0);
        store_codes(0);
        __syncthreads();
        for (int tile = 0; tile < NT; ++tile) {
            const int buf = tile & 1;
            if (tile + 1 < NT) load_codes(tile + 1);
            f4 acc[2][4];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
            const float* cs = smem + buf * RVQ_TILE_FLOATS + r16 * RVQ_D;
#pragma unroll
            for (int kg = 0; kg < 8; ++kg) {
                const int ch = (((kg << 2) + q) ^ r16) << 2;
                f4 wb[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) wb[j] = *reinterpret_cast<const f4*>(cs + j * 16 * RVQ_D + ch);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[j][e], xr[i][kg][e], acc[i][j], 0, 0, 0);
            }
            // lane holds dot[frame r16 of m-tile i][code tile*64 + j*16 + q*4 + reg]
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = tile * RVQ_CT + j * 16 + q * 4;
                const f4 e2v = *reinterpret_cast<const f4*>(e2s + n);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const float two_dot = 2.0f * acc[i][j][reg];
                        const float d = -__fadd_rn(__fsub_rn(s2[i], two_dot), e2v[reg]);
                        if (d > best[i]) { best[i] = d; bidx[i] = n + reg; }
                    }
                }
            }
            if (tile + 1 < NT) store_codes(buf ^ 1);
            __syncthreads();
        }
        // combine the four lane-quads that share a frame (first maximal index wins)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int off = 16; off <= 32; off <<= 1) {
                const float ob = __shfl_xor(best[i], off);
                const int oi = __shfl_xor(bidx[i], off);
                if (ob > best[i] || (ob == best[i] && oi < bidx[i])) { best[i] = ob; bidx[i] = oi; }
            }
            if (valid[i] && q == 0) {
                const long long r = rowi[i];
                const long long bb = r / T;
                const int tt = (int)(r - bb * T);
                codes[(bb * n_q + stage) * T + tt] = (int16_t)bidx[i];
            }
            // residual -= E[idx]
            const float* ev = E + (long long)bidx[i] * RVQ_D + q * 4;
#pragma unroll
            for (int kg = 0; kg < 8; ++kg) xr[i][kg] -= *reinterpret_cast<const f4*>(ev + kg * 16);
        }
    }
}

int launch_rvq_encode(const float* x, long long rows, int T, const float* codebooks, const float* e2, int n_q,
                      int16_t* codes, hipStream_t stream) {
    if (rows <= 0) return 0;
    const long long blocks = (rows + RVQ_ROWS - 1) / RVQ_ROWS;
    const size_t lds = 2 * RVQ_TILE_FLOATS * sizeof(float);
    hipLaunchKernelGGL(rvq_encode_kernel, dim3((unsigned)blocks), dim3(256), lds, stream, x, rows, T, codebooks, e2, n_q, codes);
    AT_CHECK_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------------
// RVQ decode: z[b][t][:] = sum_k E_k[codes[b][k][t]]  (encodec ResidualVectorQuantizer.decode;
// HF modeling_encodec.py:440-447 — summed in stage order starting from 0.0).
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rvq_decode_kernel(const int64_t* __restrict__ codes, int B, int K, int T,
                                                         const float* __restrict__ codebooks, float* __restrict__ z) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;  // one thread = one frame x 4 dims
    const long long total = (long long)B * T * 32;
    if (gid >= total) return;
    const int d4 = (int)(gid & 31);
    const long long bt = gid >> 5;
    const long long b = bt / T;
    const int t = (int)(bt - b * T);
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < K; ++k) {
        long long idx = codes[(b * K + k) * T + t];
        idx = idx < 0 ? 0 : (idx >= RVQ_CODES ? RVQ_CODES - 1 : idx);
        acc += *reinterpret_cast<const f4*>(codebooks + ((long long)k * RVQ_CODES + idx) * RVQ_D + d4 * 4);
    }
    *reinterpret_cast<f4*>(z + bt * RVQ_D + d4 * 4) = acc;
}

int launch_rvq_decode(const int64_t* codes, int B, int K, int T, const float* codebooks, float* z, hipStream_t stream) {
    const long long total = (long long)B * T * 32;
    const long long blocks = (total + 255) / 256;
    hipLaunchKernelGGL(rvq_decode_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, codes, B, K, T, codebooks, z);
    AT_CHECK_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------------
// Last decoder conv: ELU -> conv k7 32 -> 1 (causal reflect). [B][L][32] -> [B][L]. HBM-bound, 128 B in / 4 B out.
// 8 lanes per output sample (4 channels each: one coalesced 128-B row per tap), 3-step shuffle reduction.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_last_kernel(const float* __restrict__ x, const float* __restrict__ w /*[7][32]*/,
                                                        const float* __restrict__ bias, float* __restrict__ out, int L, long long total) {
    __shared__ f4 wsm[7 * 8];
    if (threadIdx.x < 56) wsm[threadIdx.x] = reinterpret_cast<const f4*>(w)[threadIdx.x];
    __syncthreads();
    long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const int cg = (int)(gid & 7);
    long long bt = gid >> 3;
    const bool live = bt < total;
    if (!live) bt = total - 1;
    const int t = (int)(bt % L);
    const float* xb = x + (bt - t) * 32;
    float acc = 0.f;
#pragma unroll
    for (int tap = 0; tap < 7; ++tap) {
        int r = t + tap - 6;
        if (r < 0) r = -r;
        const f4 v = *reinterpret_cast<const f4*>(xb + (long long)r * 32 + cg * 4);
        const f4 ww = wsm[tap * 8 + cg];
        acc = fmaf(elu1(v.x), ww.x, acc); acc = fmaf(elu1(v.y), ww.y, acc);
        acc = fmaf(elu1(v.z), ww.z, acc); acc = fmaf(elu1(v.w), ww.w, acc);
    }
    acc += __shfl_xor(acc, 1);
    acc += __shfl_xor(acc, 2);
    acc += __shfl_xor(acc, 4);
    if (live && cg == 0) out[bt] = acc + bias[0];
}

int launch_conv_last(const float* x, const float* w, const float* bias, float* out, int B, int L, hipStream_t stream) {
    const long long total = (long long)B * L;
    const long long blocks = (total * 8 + 255) / 256;
    hipLaunchKernelGGL(conv_last_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, w, bias, out, L, total);
    AT_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace at
