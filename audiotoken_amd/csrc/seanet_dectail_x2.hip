// SEANet decoder tail on the two-piece fp16 scheme (round 4): seanet_dectail.hip with its three contractions — the transposed conv 64 -> 32
// (k 4, stride 2) as a two-tap GEMM, the block's k3 conv 32 -> 16 and its tail [h | u] 48 -> 32 — moved from the fp32 matrix cores
// (v_mfma_f32_16x16x4_f32: 224 MFMAs of 32 cycles per wave and 120-sample tile) to operand splits on the fp16 ones (v_mfma_f32_16x16x32_f16:
// 90 MFMAs of 16 cycles): every activation is split into hi + lo fp16 pieces (x * 2^4) ONCE, where it is produced — the staged x rows, the
// transposed conv's output u (raw, for the shortcut) and ELU(u), the ELU'd conv3 output h — and lives in LDS as two piece planes; the weights are
// split per launch into register fragments (three products per 32-wide K step, smallest first: split_scheme.h). The last conv (32 -> 1, k 7) stays
// on the VALU in fp32, on ELU(r) kept as fp32 rows, exactly as in seanet_dectail.hip.
// Same tiling as the fp32 kernel (one workgroup = 120 output samples of one clip, 128 rows of u / h / r, 65 rows of x; reflect padding at the clip
// start by mirroring LDS rows; two workgroups per CU). Results differ from the fp32 kernel in rounding only (the scheme's products are exact in fp32
// up to the dropped lo x lo term, ~2^-22): compared by tolerance, tests/test_acoustic_gpu.py::test_fused_decoder_kernels_equal_unfused and the decoder
// goldens. Activations beyond |x| > 4094 raise the range status word (site "dec_res"): AcousticDecoder.verified() repeats on the fp32 kernel.
// (EnCodec architecture: SURVEY.md Appendix A.1; reference call site audiotoken/decoder.py:66-76.)
#include "gemm_core.h"
#include "encodec_kernels.h"
#include "split_scheme.h"

namespace at {

constexpr int DX_TO = 120;                  // output samples per tile
constexpr int DX_ROWS = 128;                // u / h / r rows per tile: row j <-> time t0 - 8 + j
constexpr int DX_XROWS = 65;                // x rows per tile: row i <-> time t0/2 - 5 + i
constexpr int DX_LDX = 80;                  // halves per x row (64 + 16): 160 B — b128 fragment reads spread evenly over the banks
constexpr int DX_LDU = 40;                  // halves per u row (32 + 8): 80 B
constexpr int DX_LDH = 24;                  // halves per h row (16 + 8): 48 B
constexpr int DX_LDR = 36;                  // floats per ELU(r) row
constexpr int DX_XP = DX_XROWS * DX_LDX;    // one piece plane of x (halves)
constexpr int DX_UP = DX_ROWS * DX_LDU;
constexpr int DX_UEP = (DX_ROWS + 2) * DX_LDU;   // two spare rows in front (rows -2, -1 of the k3 window)
constexpr int DX_HP = DX_ROWS * DX_LDH;
constexpr int DX_XR_BYTES = (2 * DX_XP * 2 > DX_ROWS * DX_LDR * 4) ? 2 * DX_XP * 2 : DX_ROWS * DX_LDR * 4;   // x pieces, later ELU(r) as fp32
constexpr int DX_LDS_BYTES = DX_XR_BYTES + 2 * DX_UP * 2 + 2 * DX_UEP * 2 + 2 * DX_HP * 2 + (7 * 32 + 64 + 16 + 32 + 4) * 4;
constexpr int DX_CHUNKS = DX_XROWS * 16;
constexpr int DX_PRE = (DX_CHUNKS + 255) / 256;

__global__ __launch_bounds__(256, 2) void seanet_dectail_x2_kernel(DecTailArgs a) {
    typedef SchemeF16x2 SC;
    typedef _Float16 PT;
    typedef f16x8 V8;
    typedef f16x4 V4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    PT* Xp = reinterpret_cast<PT*>(smem_raw);                           // [2][65][80]; re-used for ELU(r) (fp32) once the transposed conv is done
    float* Re = reinterpret_cast<float*>(smem_raw);                     // [128][36]
    PT* Up = reinterpret_cast<PT*>(smem_raw + DX_XR_BYTES);             // [2][128][40] raw u (shortcut input)
    PT* Uep = Up + 2 * DX_UP;                                           // [2][130][40] ELU(u); row j lives at index j + 2
    PT* Hp = Uep + 2 * DX_UEP;                                          // [2][128][24] ELU(conv3 + b3)
    float* Wl = reinterpret_cast<float*>(Hp + 2 * DX_HP);               // last conv [7][32]
    float* Bu = Wl + 7 * 32;                                            // biases: bu [64] | b3 [16] | bt [32] | bl [1]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, q = lane >> 4;
    const int L = a.L, Lout = 2 * L;
    const int tiles_per_clip = (Lout + DX_TO - 1) / DX_TO;
    const long long total_tiles = (long long)a.B * tiles_per_clip;
    const float as = a.act_scale;
    const float su = 1.0f / (as * a.wu_scale), s3 = 1.0f / (as * a.w3_scale), st = 1.0f / (as * a.wt_scale);   // exact: powers of two

    // ---- weights -> fp16 piece fragments in registers, once per workgroup (A operand: row r16 of the wave's tile, k = 32 ks + 8 q .. + 7) ----
    auto frag = [](const float* src, float scale, V8 (&out)[2]) {   // 8 consecutive fp32 weights -> hi / lo pieces
        const f4 lo4 = *reinterpret_cast<const f4*>(src), hi4 = *reinterpret_cast<const f4*>(src + 4);
        V4 pl[2], ph[2];
        split4<SchemeNoCheck<SC>>(lo4, scale, pl);
        split4<SchemeNoCheck<SC>>(hi4, scale, ph);
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int k = 0; k < 4; ++k) { out[p][k] = pl[p][k]; out[p][4 + k] = ph[p][k]; }
    };
    const V8 zero8 = {};
    V8 wu[4][2], w3[3][2], wt[2][2][2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) frag(a.wu + (wave * 16 + r16) * 128 + ks * 32 + q * 8, a.wu_scale, wu[ks]);
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) frag(a.w3 + r16 * 96 + ks * 32 + q * 8, a.w3_scale, w3[ks]);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        // K step 0 = the 16 h channels (lanes q >= 2 hold zeros on both sides), K step 1 = the 32 raw u channels
        const float* row = a.wt + (nt * 16 + r16) * 48;
        if (q < 2) frag(row + q * 8, a.wt_scale, wt[nt][0]);
        else { wt[nt][0][0] = zero8; wt[nt][0][1] = zero8; }
        frag(row + 16 + q * 8, a.wt_scale, wt[nt][1]);
    }
    if (tid < 224) Wl[tid] = a.wl[tid];
    if (tid < 64) Bu[tid] = a.bu[tid];
    if (tid < 16) Bu[64 + tid] = a.b3[tid];
    if (tid < 32) Bu[80 + tid] = a.bt[tid];
    if (tid == 0) Bu[112] = a.bl[0];
    // the two spare rows in front of ELU(u) are only ever READ (by halo rows whose results are never used): keep them finite, or their garbage
    // would reach the range census through the h rows computed from them
    if (tid < 40) {
        const int p = tid / 20, c = (tid % 20) * 4;
        *reinterpret_cast<V4*>(Uep + p * DX_UEP + c) = V4{};
    }

    f4 pre[DX_PRE];
    auto prefetch = [&](long long tile) {
        const long long b = tile / tiles_per_clip;
        const int t0 = (int)(tile - b * tiles_per_clip) * DX_TO;
        const float* xb = a.x + b * (long long)L * 64;
#pragma unroll
        for (int j = 0; j < DX_PRE; ++j) {
            int c = tid + 256 * j;
            c = c < DX_CHUNKS ? c : DX_CHUNKS - 1;
            const int tx = t0 / 2 - 5 + (c >> 4);
            const int txc = tx < 0 ? 0 : (tx > L - 1 ? L - 1 : tx);   // rows past the end only feed outputs that are never stored
            f4 v = *reinterpret_cast<const f4*>(xb + (long long)txc * 64 + (c & 15) * 4);
            if (tx < 0) v = f4{0.f, 0.f, 0.f, 0.f};                   // the transposed conv's zero left pad
            pre[j] = v;
        }
    };
    if ((long long)blockIdx.x < total_tiles) prefetch(blockIdx.x);
    RangeMax over;

    for (long long tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
        const long long b = tile / tiles_per_clip;
        const int t0 = (int)(tile - b * tiles_per_clip) * DX_TO;
        __syncthreads();   // previous tile's readers are done with every buffer
#pragma unroll
        for (int j = 0; j < DX_PRE; ++j) {
            const int c = tid + 256 * j;
            if (c < DX_CHUNKS) {
                V4 p[2];
                over |= split4<SC>(pre[j], as, p);
                PT* d = Xp + (c >> 4) * DX_LDX + (c & 15) * 4;
                *reinterpret_cast<V4*>(d) = p[0];
                *reinterpret_cast<V4*>(d + DX_XP) = p[1];
            }
        }
        __syncthreads();
        if (tile + gridDim.x < total_tiles) prefetch(tile + gridDim.x);   // flies during the MFMAs below
        // ---- transposed conv: wave w owns columns 16w..16w+15 of the [64 t][2 x 32] output = phase w >> 1, channels 16 (w & 1) ..;
        //      GEMM row m <-> t = t0/2 - 4 + m uses x rows m (tap 0: x[t-1]) and m + 1 (tap 1: x[t]); K step ks = tap ks >> 1, channels 32 (ks & 1) .. ----
        {
            f4 acc[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[m] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int tap = ks >> 1, c0 = (ks & 1) * 32;
                V8 xb[2][4];
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int m = 0; m < 4; ++m) xb[p][m] = *reinterpret_cast<const V8*>(Xp + p * DX_XP + (m * 16 + r16 + tap) * DX_LDX + c0 + q * 8);
#pragma unroll
                for (int t = 0; t < SC::NPROD; ++t)
#pragma unroll
                    for (int m = 0; m < 4; ++m) acc[m] = SC::mfma16(wu[ks][SC::prod_w(t)], xb[SC::prod_a(t)][m], acc[m]);
            }
            const f4 bu = *reinterpret_cast<const f4*>(Bu + wave * 16 + q * 4);
            const int ph = wave >> 1, co = (wave & 1) * 16 + q * 4;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int j = 2 * (m * 16 + r16) + ph;
                const f4 v = acc[m] * su + bu;
                V4 p[2];
                over |= split4<SC>(v, as, p);
                *reinterpret_cast<V4*>(Up + j * DX_LDU + co) = p[0];
                *reinterpret_cast<V4*>(Up + DX_UP + j * DX_LDU + co) = p[1];
                const f4 e = {elu1(v.x), elu1(v.y), elu1(v.z), elu1(v.w)};
                split4<SchemeNoCheck<SC>>(e, as, p);   // |ELU(v)| <= max(|v|, 1): covered by the check of v
                *reinterpret_cast<V4*>(Uep + (j + 2) * DX_LDU + co) = p[0];
                *reinterpret_cast<V4*>(Uep + DX_UEP + (j + 2) * DX_LDU + co) = p[1];
            }
        }
        __syncthreads();   // Up / Uep complete; Xp is dead from here on (Re takes its place)
        if (t0 == 0) {     // the k3 conv's reflect pad: u[-1] = u[1], u[-2] = u[2]  (row j <-> time j - 8), both pieces
            if (tid < 32) {
                const int p = tid >> 4, k = 1 + ((tid >> 3) & 1), c = (tid & 7) * 4;
                *reinterpret_cast<V4*>(Uep + p * DX_UEP + (8 - k + 2) * DX_LDU + c) = *reinterpret_cast<const V4*>(Uep + p * DX_UEP + (8 + k + 2) * DX_LDU + c);
            }
            __syncthreads();
        }
        // ---- block: two row tiles per wave; the h rows a wave writes are the ones it reads back (no barrier in between) ------
        {
            int row[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) row[i] = (2 * wave + i) * 16 + r16;
            f4 acc[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) {   // K step = tap ks, all 32 channels
                V8 xb[2][2];
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int i = 0; i < 2; ++i) xb[p][i] = *reinterpret_cast<const V8*>(Uep + p * DX_UEP + (row[i] + ks) * DX_LDU + q * 8);   // (row + tap - 2) + 2
#pragma unroll
                for (int t = 0; t < SC::NPROD; ++t)
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc[i] = SC::mfma16(w3[ks][SC::prod_w(t)], xb[SC::prod_a(t)][i], acc[i]);
            }
            const f4 b3 = *reinterpret_cast<const f4*>(Bu + 64 + q * 4);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const f4 v = acc[i] * s3 + b3;
                const f4 o = {elu1(v.x), elu1(v.y), elu1(v.z), elu1(v.w)};
                V4 p[2];
                over |= split4<SC>(o, as, p);
                *reinterpret_cast<V4*>(Hp + row[i] * DX_LDH + q * 4) = p[0];
                *reinterpret_cast<V4*>(Hp + DX_HP + row[i] * DX_LDH + q * 4) = p[1];
            }
            f4 acc2[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc2[i][nt] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                V8 xb[2][2];
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        if (ks == 0) xb[p][i] = q < 2 ? *reinterpret_cast<const V8*>(Hp + p * DX_HP + row[i] * DX_LDH + q * 8) : zero8;
                        else xb[p][i] = *reinterpret_cast<const V8*>(Up + p * DX_UP + row[i] * DX_LDU + q * 8);
                    }
#pragma unroll
                for (int t = 0; t < SC::NPROD; ++t)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) acc2[i][nt] = SC::mfma16(wt[nt][ks][SC::prod_w(t)], xb[SC::prod_a(t)][i], acc2[i][nt]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const f4 v = acc2[i][nt] * st + *reinterpret_cast<const f4*>(Bu + 80 + nt * 16 + q * 4);
                    const f4 o = {elu1(v.x), elu1(v.y), elu1(v.z), elu1(v.w)};
                    *reinterpret_cast<f4*>(Re + row[i] * DX_LDR + nt * 16 + q * 4) = o;
                }
        }
        __syncthreads();
        if (t0 == 0) {     // the k7 conv's reflect pad on ELU(r): r[-k] = r[k], k = 1..6
            if (tid < 48) {
                const int k = 1 + tid / 8, c = (tid & 7) * 4;
                *reinterpret_cast<f4*>(Re + (8 - k) * DX_LDR + c) = *reinterpret_cast<const f4*>(Re + (8 + k) * DX_LDR + c);
            }
            __syncthreads();
        }
        // ---- last conv (32 -> 1, k7) on the VALU, lane layout and reduction order of conv_last_kernel ----
        {
            const int cg = tid & 7;
            const float bl = Bu[112];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int o = it * 32 + (tid >> 3);           // output sample t0 + o  <->  row j = 8 + o
                const int oc = o < DX_TO ? o : DX_TO - 1;
                float acc = 0.f;
#pragma unroll
                for (int tap = 0; tap < 7; ++tap) {
                    const f4 v = *reinterpret_cast<const f4*>(Re + (8 + oc - 6 + tap) * DX_LDR + cg * 4);
                    const f4 ww = *reinterpret_cast<const f4*>(Wl + tap * 32 + cg * 4);
                    acc = fmaf(v.x, ww.x, acc); acc = fmaf(v.y, ww.y, acc);
                    acc = fmaf(v.z, ww.z, acc); acc = fmaf(v.w, ww.w, acc);
                }
                acc += __shfl_xor(acc, 1);
                acc += __shfl_xor(acc, 2);
                acc += __shfl_xor(acc, 4);
                const int tout = t0 + o;
                if (cg == 0 && o < DX_TO && tout < Lout) a.out[b * (long long)Lout + tout] = acc + bl;
            }
        }
    }
    range_publish(a.status, a.status ? a.status + 1 : nullptr, over);
}

int launch_seanet_dectail_x2(const DecTailArgs& a, hipStream_t stream) {
    AT_REQUIRE(a.L >= 8 && a.B >= 1, "fused decoder tail needs at least 8 input rows");
    AT_REQUIRE(a.act_scale > 0.f && a.wu_scale > 0.f && a.w3_scale > 0.f && a.wt_scale > 0.f, "seanet_dectail_x2: operand scales missing");
    const size_t lds = (size_t)DX_LDS_BYTES;
    { static LdsAttrFlags lds_attr_0; if (int rc = set_max_dynamic_lds(lds_attr_0, seanet_dectail_x2_kernel, lds)) return rc; }
    const long long tiles = (long long)a.B * ((2 * a.L + DX_TO - 1) / DX_TO);
    const int grid = (int)(tiles < 512 ? tiles : 512);   // two resident workgroups per CU
    hipLaunchKernelGGL(seanet_dectail_x2_kernel, dim3(grid), dim3(256), lds, stream, a);
    AT_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace at
