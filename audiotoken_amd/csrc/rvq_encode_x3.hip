// RVQ search (EnCodec ResidualVectorQuantizer.encode; HF modeling_encodec.py:424-438) with the frame x code dot products on the bf16
// matrix cores: the kernel of encodec_kernels.hip (all n_q stages in one launch, residual in registers, 64-code tiles streamed
// through LDS double-buffered, reference-order distance -( (|r|^2 - 2 r.e) + |e|^2 ), first maximal index) with both operands as
// exact 3-way bf16 splits and six v_mfma_f32_16x16x32_bf16 per 32-wide K step (arithmetic and accuracy: gemm_bf16x3.hip).
// The codebooks are split once at load time ([3][n_cb * 1024][128] bf16); the residual stays fp32 in registers (the update
// r -= E[idx] uses the fp32 codebook, exactly as before) and is re-split at the start of every stage.
#include "gemm_core.h"
#include "encodec_kernels.h"

namespace at {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

constexpr int QX_D = 128, QX_CODES = 1024, QX_CT = 64, QX_ROWS = 128;
constexpr int QX_LD = QX_D + 16;                // LDS row stride (bf16): + 32 B, conflict-free fragment reads under the real ds_read_b128 lane grouping (see seanet_res128x3.hip)
constexpr int QX_PIECE = QX_CT * QX_LD;         // elements of one piece of a code tile
constexpr int QX_TILE = 3 * QX_PIECE;

__global__ void split_plain_kernel(const float* __restrict__ x, long long n, __bf16* __restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    const __bf16 p1 = (__bf16)v;
    const float r1 = v - (float)p1;
    const __bf16 p2 = (__bf16)r1;
    out[i] = p1; out[n + i] = p2; out[2 * n + i] = (__bf16)(r1 - (float)p2);
}

int launch_split_plain(const float* x, long long n, __bf16* out, hipStream_t stream) {
    hipLaunchKernelGGL(split_plain_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x, n, out);
    AT_CHECK_HIP(hipGetLastError());
    return 0;
}

__global__ __launch_bounds__(256, 1) void rvq_encode_x3_kernel(const float* __restrict__ x, long long rows, int T,
                                                                const float* __restrict__ codebooks, const __bf16* __restrict__ cb_s,
                                                                long long cb_piece, const float* __restrict__ e2, int n_q,
                                                                int16_t* __restrict__ codes) {
    extern __shared__ __attribute__((aligned(16))) __bf16 qx_lds[];   // [2 buffers][3 pieces][64 codes][144]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, q = lane >> 4;
    const long long row_base = (long long)blockIdx.x * QX_ROWS + wave * 32;

    // residual, fp32, in B-fragment grouping: xr[i][ks][h] = x[row_i][32 ks + 8 q + 4 h .. + 3]
    f4 xr[2][4][2];
    long long rowi[2];
    bool valid[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const long long r = row_base + i * 16 + r16;
        valid[i] = r < rows;
        rowi[i] = valid[i] ? r : rows - 1;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) xr[i][ks][hh] = *reinterpret_cast<const f4*>(x + rowi[i] * QX_D + ks * 32 + q * 8 + hh * 4);
    }
    // staging of a code tile: 3 pieces x 64 rows x 256 B = 3072 chunks of 16 B, 12 per thread: chunk c = tid + 256 j -> row c >> 4, 16 B c & 15
    u4 stage_reg[12];
    constexpr int PW[6] = {2, 0, 1, 1, 0, 0}, PX[6] = {0, 2, 1, 0, 1, 0};   // smallest products first

    for (int stage = 0; stage < n_q; ++stage) {
        const float* E = codebooks + (long long)stage * QX_CODES * QX_D;
        const __bf16* Es = cb_s + (long long)stage * QX_CODES * QX_D;
        const float* e2s = e2 + stage * QX_CODES;
        // |r|^2 per frame and the three bf16 pieces of the residual
        float s2[2];
        bf16x8 xp[3][2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float p = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float v = xr[i][ks][hh][k];
                        p = fmaf(v, v, p);
                        const __bf16 x1 = (__bf16)v;
                        const float r1 = v - (float)x1;
                        const __bf16 x2 = (__bf16)r1;
                        xp[0][i][ks][hh * 4 + k] = x1; xp[1][i][ks][hh * 4 + k] = x2; xp[2][i][ks][hh * 4 + k] = (__bf16)(r1 - (float)x2);
                    }
            p += __shfl_xor(p, 16);
            p += __shfl_xor(p, 32);
            s2[i] = p;
        }
        float best[2] = {-INFINITY, -INFINITY};
        int bidx[2] = {0, 0};

        auto load_codes = [&](int tile) {
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = tid + 256 * j;
                    stage_reg[p * 4 + j] = *reinterpret_cast<const u4*>(Es + p * cb_piece + ((long long)tile * QX_CT + (c >> 4)) * QX_D + (c & 15) * 8);
                }
        };
        auto store_codes = [&](int buf) {
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = tid + 256 * j;
                    *reinterpret_cast<u4*>(qx_lds + buf * QX_TILE + p * QX_PIECE + (c >> 4) * QX_LD + (c & 15) * 8) = stage_reg[p * 4 + j];
                }
        };
        constexpr int NT = QX_CODES / QX_CT;
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
            const __bf16* cs = qx_lds + buf * QX_TILE + r16 * QX_LD + q * 8;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                bf16x8 wb[3][4];
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int j = 0; j < 4; ++j) wb[p][j] = *reinterpret_cast<const bf16x8*>(cs + p * QX_PIECE + j * 16 * QX_LD + ks * 32);
#pragma unroll
                for (int t = 0; t < 6; ++t)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[PW[t]][j], xp[PX[t]][i][ks], acc[i][j], 0, 0, 0);
            }
            // lane holds dot[frame r16 of m-tile i][code tile*64 + j*16 + q*4 + reg]
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = tile * QX_CT + j * 16 + q * 4;
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
            // residual -= E[idx] (fp32 codebook)
            const float* ev = E + (long long)bidx[i] * QX_D + q * 8;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) xr[i][ks][hh] -= *reinterpret_cast<const f4*>(ev + ks * 32 + hh * 4);
        }
    }
}

int launch_rvq_encode_x3(const float* x, long long rows, int T, const float* codebooks, const __bf16* cb_s, long long cb_piece,
                         const float* e2, int n_q, int16_t* codes, hipStream_t stream) {
    if (rows <= 0) return 0;
    const long long blocks = (rows + QX_ROWS - 1) / QX_ROWS;
    const size_t lds = (size_t)2 * QX_TILE * sizeof(__bf16);
    { static LdsAttrFlags lds_attr_0; if (int rc = set_max_dynamic_lds(lds_attr_0, rvq_encode_x3_kernel, lds)) return rc; }
    hipLaunchKernelGGL(rvq_encode_x3_kernel, dim3((unsigned)blocks), dim3(256), lds, stream, x, rows, T, codebooks, cb_s, cb_piece, e2, n_q, codes);
    AT_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace at
