// RVQ search (EnCodec ResidualVectorQuantizer.encode; HF modeling_encodec.py:424-438) with the frame x code dot products on the bf16
// matrix cores: the kernel of encodec_kernels.hip (all n_q stages in one launch, residual in registers, 64-code tiles streamed
// through LDS double-buffered, reference-order distance -( (|r|^2 - 2 r.e) + |e|^2 ), first maximal index) with both operands as
// exact 3-way bf16 splits and six v_mfma_f32_16x16x32_bf16 per 32-wide K step (arithmetic and accuracy: gemm_bf16x3.hip).
// The codebooks are split once at load time ([3][n_cb * 1024][128] bf16); the residual stays fp32 in registers (the update
// r -= E[idx] uses the fp32 codebook, exactly as before) and is re-split at the start of every stage.
// Round 2: the kernel is a template on the operand scheme (split_scheme.h). The default is two fp16 pieces / three products:
// codebooks split at load time as [2][n_cb * 1024][128] fp16 of E * cb_scale (one power of two for all codebooks, so the order of
// the distances is untouched), the residual split per stage as r * act_scale with a range check into the status word, and the
// accumulator multiplied by the exact power of two 1 / (act_scale * cb_scale) before it enters the reference-order distance.
#include "gemm_core.h"
#include "encodec_kernels.h"
#include "gemm_bf16x3.h"
#include "split_scheme.h"

namespace at {

typedef unsigned int u4 __attribute__((ext_vector_type(4)));

constexpr int QX_D = 128, QX_CODES = 1024, QX_CT = 64, QX_ROWS = 128;
constexpr int QX_LD = QX_D + 16;                // LDS row stride (bf16): + 32 B, conflict-free fragment reads under the real ds_read_b128 lane grouping (see seanet_res128x3.hip)
constexpr int QX_PIECE = QX_CT * QX_LD;         // elements of one piece of a code tile

template <class SC>
__global__ void split_plain_kernel(const float* __restrict__ x, long long n, float scale, typename SC::T* __restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    typename SC::T p[SC::NP];
    split_n<SC>(x[i] * scale, p);
#pragma unroll
    for (int k = 0; k < SC::NP; ++k) out[k * n + i] = p[k];
}

int launch_split_plain(const float* x, long long n, __bf16* out, hipStream_t stream, int scheme, float scale) {
    const dim3 grid((unsigned)((n + 255) / 256));
    if (scheme == XB_SCHEME_F16X2)
        hipLaunchKernelGGL(split_plain_kernel<SchemeF16x2>, grid, dim3(256), 0, stream, x, n, scale, reinterpret_cast<_Float16*>(out));
    else
        hipLaunchKernelGGL(split_plain_kernel<SchemeBf16x3>, grid, dim3(256), 0, stream, x, n, 1.0f, out);
    AT_CHECK_HIP(hipGetLastError());
    return 0;
}

template <class SC>
__global__ __launch_bounds__(256, 1) void rvq_encode_x3_kernel(const float* __restrict__ x, long long rows, int T,
                                                                const float* __restrict__ codebooks, const typename SC::T* __restrict__ cb_s,
                                                                long long cb_piece, const float* __restrict__ e2, int n_q,
                                                                int16_t* __restrict__ codes, float act_scale, float cb_scale, int* status) {
    typedef typename SC::T PT;
    typedef typename SC::V8 V8;
    typedef typename SC::V4 V4;
    constexpr int NP = SC::NP;
    constexpr int QX_TILE = NP * QX_PIECE;
    extern __shared__ __attribute__((aligned(16))) unsigned char qx_lds_raw[];   // [2 buffers][NP pieces][64 codes][144]
    PT* qx_lds = reinterpret_cast<PT*>(qx_lds_raw);
    const float sa = SC::RANGE_CHECK ? act_scale : 1.0f;
    const float rs = SC::RANGE_CHECK ? 1.0f / (act_scale * cb_scale) : 1.0f;
    RangeMax over;
    bool nonfinite = false;
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
    u4 stage_reg[NP * 4];

    for (int stage = 0; stage < n_q; ++stage) {
        const float* E = codebooks + (long long)stage * QX_CODES * QX_D;
        const PT* Es = cb_s + (long long)stage * QX_CODES * QX_D;
        const float* e2s = e2 + stage * QX_CODES;
        // |r|^2 per frame and the pieces of the residual
        float s2[2];
        V8 xp[NP][2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float p = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) p = fmaf(xr[i][ks][hh][k], xr[i][ks][hh][k], p);
                    V4 pc[NP];
                    over |= split4<SC>(xr[i][ks][hh], sa, pc);
#pragma unroll
                    for (int pi = 0; pi < NP; ++pi)
#pragma unroll
                        for (int k = 0; k < 4; ++k) xp[pi][i][ks][hh * 4 + k] = pc[pi][k];
                }
            p += __shfl_xor(p, 16);
            p += __shfl_xor(p, 32);
            s2[i] = p;
            nonfinite |= valid[i] && !(p <= 3.0e38f);   // |residual|^2: a NaN / infinity anywhere in the encoder ends up here
        }
        float best[2] = {-INFINITY, -INFINITY};
        int bidx[2] = {0, 0};

        auto load_codes = [&](int tile) {
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = tid + 256 * j;
                    stage_reg[p * 4 + j] = *reinterpret_cast<const u4*>(Es + p * cb_piece + ((long long)tile * QX_CT + (c >> 4)) * QX_D + (c & 15) * 8);
                }
        };
        auto store_codes = [&](int buf) {
#pragma unroll
            for (int p = 0; p < NP; ++p)
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
            const PT* cs = qx_lds + buf * QX_TILE + r16 * QX_LD + q * 8;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                V8 wb[NP][4];
#pragma unroll
                for (int p = 0; p < NP; ++p)
#pragma unroll
                    for (int j = 0; j < 4; ++j) wb[p][j] = *reinterpret_cast<const V8*>(cs + p * QX_PIECE + j * 16 * QX_LD + ks * 32);
#pragma unroll
                for (int t = 0; t < SC::NPROD; ++t)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[i][j] = SC::mfma16(wb[SC::prod_w(t)][j], xp[SC::prod_a(t)][i][ks], acc[i][j]);
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
                        const float two_dot = 2.0f * (acc[i][j][reg] * rs);
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
    if constexpr (SC::RANGE_CHECK)
        range_publish(status, status ? status + 1 : nullptr, over);
    if (status && nonfinite) atomicOr(status, XB_STATUS_NONFINITE);
}

template <class SC>
static int launch_rvq_scheme(const float* x, long long rows, int T, const float* codebooks, const void* cb_s, long long cb_piece,
                             const float* e2, int n_q, int16_t* codes, float act_scale, float cb_scale, int* status, hipStream_t stream) {
    const long long blocks = (rows + QX_ROWS - 1) / QX_ROWS;
    const size_t lds = (size_t)2 * SC::NP * QX_PIECE * 2;
    { static LdsAttrFlags lds_attr; if (int rc = set_max_dynamic_lds(lds_attr, rvq_encode_x3_kernel<SC>, lds)) return rc; }
    hipLaunchKernelGGL(rvq_encode_x3_kernel<SC>, dim3((unsigned)blocks), dim3(256), lds, stream, x, rows, T, codebooks,
                       static_cast<const typename SC::T*>(cb_s), cb_piece, e2, n_q, codes, act_scale, cb_scale, status);
    AT_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_rvq_encode_x3(const float* x, long long rows, int T, const float* codebooks, const __bf16* cb_s, long long cb_piece,
                         const float* e2, int n_q, int16_t* codes, hipStream_t stream, int scheme, float act_scale, float cb_scale, int* status) {
    if (rows <= 0) return 0;
    if (scheme == XB_SCHEME_F16X2) {
        AT_REQUIRE(act_scale > 0.f && cb_scale > 0.f, "two-piece fp16 RVQ needs its scales");
        return launch_rvq_scheme<SchemeF16x2>(x, rows, T, codebooks, cb_s, cb_piece, e2, n_q, codes, act_scale, cb_scale, status, stream);
    }
    return launch_rvq_scheme<SchemeBf16x3>(x, rows, T, codebooks, cb_s, cb_piece, e2, n_q, codes, 1.f, 1.f, status, stream);   // (status: the non-finite flag only)
}

}  // namespace at
