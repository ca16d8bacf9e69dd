// Acoustic tokenizer (EnCodec 24 kHz) — host side: weight intake / repacking and the launch sequence.
// C ABI in include/audiotoken_hip.h. Replaces reference AcousticEncoder / AcousticDecoder
// (audiotoken/encoder.py:29-57, audiotoken/decoder.py:50-76), whose arithmetic is the PyPI `encodec` model;
// architecture per SURVEY.md Appendix A.1.
//
// Data layout: every activation is time-major / channels-last [clip][t][c] so that conv windows are contiguous
// (see at_common.h). Weights are repacked once at finalize():
//   Conv1d  [Cout][Cin][k]      -> [Cout][k*Cin]            (tap-major rows, matches the window order)
//   ConvTr  [Cin][Cout][k=2s]   -> [s*Cout][2*Cin]          (phase p row block: [W[:, :, p+s] | W[:, :, p]])
//   LSTM    [4H][H] gate blocks -> rows 4*j + g             (gates of one unit adjacent; see lstm_step_kernel)
#include <map>
#include <string>
#include <vector>
#include <cstring>
#include <cmath>
#include <cstdlib>

#include "../../include/audiotoken_hip.h"
#include "at_common.h"
#include "encodec_kernels.h"
#include "gemm_bf16x3.h"

namespace at {

static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }
const char* last_error_cstr() { return g_last_error.c_str(); }

hipEvent_t Profiler::get_event() {
    if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
void Profiler::begin(const char* group, int launches, hipStream_t s) {
    if (!enabled) return;
    int g = -1;
    for (size_t i = 0; i < names.size(); ++i) if (names[i] == group) { g = (int)i; break; }
    if (g < 0) { names.push_back(group); g = (int)names.size() - 1; }
    Span sp{g, launches, get_event(), get_event()};
    (void)hipEventRecord(sp.a, s);
    spans.push_back(sp);
    open_ = (int)spans.size() - 1;
}
void Profiler::end(hipStream_t s) {
    if (!enabled || open_ < 0) return;
    (void)hipEventRecord(spans[open_].b, s);
    open_ = -1;
}
void Profiler::reset() {
    for (auto& sp : spans) { pool.push_back(sp.a); pool.push_back(sp.b); }
    spans.clear();
    names.clear();
    open_ = -1;
}
int Profiler::read(std::vector<float>& ms, std::vector<int>& launches) {
    ms.assign(names.size(), 0.f);
    launches.assign(names.size(), 0);
    for (auto& sp : spans) {
        if (hipEventSynchronize(sp.b) != hipSuccess) return -2;
        float t = 0.f;
        if (hipEventElapsedTime(&t, sp.a, sp.b) != hipSuccess) return -2;
        ms[sp.group] += t;
        launches[sp.group] += sp.launches;
    }
    return 0;
}
Profiler::~Profiler() {
    reset();
    for (auto e : pool) (void)hipEventDestroy(e);
}

struct HostTensor {
    std::vector<int64_t> shape;
    std::vector<float> data;
};

struct ConvW {
    const float* w = nullptr;
    const float* b = nullptr;
    int cin = 0, cout = 0, k = 0, stride = 1;
};

constexpr int kRatiosEnc[4] = {2, 4, 5, 8};
constexpr int kRatiosDec[4] = {8, 5, 4, 2};
constexpr int kH = 512;
constexpr int kDim = 128;
constexpr int kCodes = 1024;
constexpr bool kBf16x3AcousticDefault = true;
constexpr int kSubBatchDefault = 256;  // clips per pass through the 24 kHz..75 Hz conv stack (bounds the workspace)
inline int sub_batch() {
    static const int v = [] {
        const char* e = std::getenv("AUDIOTOKEN_SUBBATCH");
        const int n = e ? std::atoi(e) : 0;
        return n > 0 ? n : kSubBatchDefault;
    }();
    return v;
}

}  // namespace at

using namespace at;

struct at_encodec {
    int device = 0;
    bool finalized = false;
    bool has_decoder = false;
    std::map<std::string, HostTensor> staged;
    float* blob = nullptr;
    size_t blob_floats = 0;
    // encoder
    ConvW conv0, res[4][3], down[4], fin;
    const float *wih[2] = {}, *whh[2] = {}, *bih[2] = {}, *bhh[2] = {};
    // quantiser
    const float* codebooks = nullptr;  // [n_cb][1024][128]
    const float* e2 = nullptr;         // [n_cb][1024]
    int n_codebooks = 0;
    // decoder
    ConvW dconv0, dup[4], dres[4][3], dlast;
    const float *dwih[2] = {}, *dwhh[2] = {}, *dbih[2] = {}, *dbhh[2] = {};
    Profiler prof;
    bool fused_stage0 = true;       // conv0 + resblock + strided conv in one kernel (seanet_stage0.hip)
    bool fused_res64 = true;        // 64-channel residual block in one kernel (seanet_res64.hip)
    bool fused_res128 = true;       // 128-channel residual block in one kernel (seanet_res128.hip)
    bool fused_down64 = true;       // stage-1 strided conv with register-stationary weights (seanet_down64.hip)
    bool fused_stage1 = true;       // 64-channel block + stage-1 strided conv in one role-split kernel (seanet_res64down.hip; fp16 scheme, needs res64_x3 / down64_x3 / res_f16x2)
    bool down64_x3 = true;          // ... on the bf16 matrix cores with 3-way split operands (seanet_down64x3.hip); follows bf16x3
    bool down128_x3 = true;         // stage-2 strided conv as a windowed split-bf16 GEMM fed by seanet_res128x3's split epilogue; follows bf16x3
    const __bf16* down2_s = nullptr;
    bool rvq_x3 = true;             // RVQ search with the dot products on the bf16 matrix cores (rvq_encode_x3.hip); follows bf16x3
    const __bf16* cb_s = nullptr;   // codebooks as 3 bf16 pieces [3][n_cb * 1024][128]
    const piece_t* fin_f = nullptr; // final conv weight [128][7 * 512] * fin_fs as two fp16 pieces, K-blocks in window order (option "fin_f16x2")
    float fin_fs = 1.f;
    bool fin_f16x2 = true;
    const float* sc0_w = nullptr;   // stage 0: shortcut folded into conv0, [32][7] weights then [32] bias (Stage0Args::wsc0 / bsc0)
    const piece_t* dup_f[3] = {nullptr, nullptr, nullptr};   // decoder transposed convs [r * Cout][2 * Cin] * dup_fs as two fp16 pieces, window order (option "up_f16x2")
    float dup_fs[3] = {};
    bool up_f16x2 = true;
    bool res128_rs = true;          // 128-channel block (fp16 scheme): the role-split kernel (seanet_res128rs.hip) instead of seanet_res128x3.hip; same bits
    const __bf16* cb_f = nullptr;   // codebooks * cb_fs as 2 fp16 pieces [2][n_cb * 1024][128] (option "rvq_f16x2")
    float cb_fs = 1.f;
    bool rvq_f16x2 = true;
    bool lstm_x3 = true;            // persistent LSTM with the recurrent product on the bf16 matrix cores (lstm_seq_x3.hip); follows bf16x3
    bool res256_x3 = true;          // 256-channel block as two split-bf16 GEMMs chained between the stage-2 and stage-3 strided convs; follows bf16x3
    const __bf16 *res3c_s = nullptr, *res3t_s = nullptr;
    bool down256_x3 = true;         // stage-3 strided conv as a windowed split-bf16 GEMM behind a split pass; follows bf16x3
    const __bf16* down3_s = nullptr;
    bool stage0_x3 = true;          // fused stage 0 on the bf16 matrix cores (seanet_stage0x3.hip); follows bf16x3
    bool res64_x3 = true;           // 64-channel residual block on the bf16 matrix cores (seanet_res64x3.hip); follows bf16x3
    bool res128_x3 = true;          // 128-channel residual block on the bf16 matrix cores (seanet_res128x3.hip); follows bf16x3
    bool fused_dectail = true;      // decoder: last transposed conv + block + final conv in one kernel (seanet_dectail.hip)
    bool tail_f16x2 = true;         // ... with its contractions on the two-piece fp16 scheme (seanet_dectail_x2.hip)
    bool dec_chain = true;          // decoder stage 0 (256 channels): the block as two split GEMMs whose output is the next transposed conv's operand (seanet_dec256.hip)
    const piece_t* dchain_f[2] = {nullptr, nullptr};   // its k3 conv [128][3 * 256] (window order) and tail [256][128 + 256] as two fp16 pieces (scales: dres_fs[0])
    float dtail_up_fs = 0.f;        // power-of-two scale of the last transposed conv's weights for it
    bool bf16x3 = false;            // plain linear layers (LSTM input projections) on the split-bf16 GEMM ($AUDIOTOKEN_BF16X3_ACOUSTIC)
    const __bf16* wih_s[2] = {nullptr, nullptr};
    const __bf16* dwih_s[2] = {nullptr, nullptr};
    // the LSTM input projections also as two-piece fp16 operands (gemm_bf16x3.h, XB_SCHEME_F16X2): weights + their power-of-two scales
    const piece_t* wih_f[2] = {nullptr, nullptr};
    const piece_t* dwih_f[2] = {nullptr, nullptr};
    float wih_fs[2] = {1.f, 1.f}, dwih_fs[2] = {1.f, 1.f};
    // the stage 2-3 GEMM chain's weights as two fp16 pieces (+ scales): down2, res3 conv3, res3 tail, down3 (option "chain_f16x2")
    const piece_t* chain_f[4] = {nullptr, nullptr, nullptr, nullptr};
    float chain_fs[4] = {1.f, 1.f, 1.f, 1.f};
    bool chain_f16x2 = true;
    // fused residual blocks on the fp16 scheme (option "res_f16x2"): power-of-two scales of [conv3, tail] per stage, encoder / decoder
    float res_fs[4][2] = {}, dres_fs[4][2] = {};
    float down_fs[4] = {};    // strided convs (stage-1 fused kernel on the fp16 scheme)
    bool res_f16x2 = true;
    float whh_fs[2] = {0.f, 0.f}, dwhh_fs[2] = {0.f, 0.f};   // W_hh scales of the fp16-scheme LSTM recurrence (option "lstm_f16x2")
    bool lstm_f16x2 = true;
    bool lstm_pipe = true;          // batches of <= 80 clips: both LSTM layers in one pipelined launch (lstm_pipe.hip), same arithmetic
    bool ih_f16x2 = true;   // option "ih_f16x2": LSTM input projections on the fp16 scheme (three MFMA products instead of six)
    std::vector<void*> extra_allocs;
    int* range_tab = nullptr;   // device, {flag, census} per AcSite, zeroed at the start of every encode / decode (at_encodec_range_report reads it)
    int sub_batch = at::sub_batch();   // clips per pass through the conv stack: bounds the workspace (option "subbatch")
    bool persistent_lstm = false;   // whole-sequence persistent LSTM (needs one resident workgroup per CU for 256 CUs)
    unsigned lstm_spin_limit = 1u << 18;   // option "lstm_spin_limit": flag polls before a persistent-LSTM workgroup gives up
};

namespace {

const HostTensor* find(const at_encodec* h, const std::string& name) {
    auto it = h->staged.find(name);
    return it == h->staged.end() ? nullptr : &it->second;
}

struct Packer {
    std::vector<float> host;
    size_t add(const std::vector<float>& v) {
        size_t off = host.size();
        host.insert(host.end(), v.begin(), v.end());
        while (host.size() % 4) host.push_back(0.f);  // keep every tensor 16-byte aligned
        return off;
    }
};

// Conv1d weight [cout][cin][k] -> [cout][k][cin]
bool pack_conv(const at_encodec* h, const std::string& prefix, int cin, int cout, int k, Packer& p, size_t& w_off,
               size_t& b_off) {
    const HostTensor* w = find(h, prefix + ".weight");
    const HostTensor* b = find(h, prefix + ".bias");
    if (!w || !b) { set_error("missing tensor " + prefix + ".{weight,bias}"); return false; }
    if (w->shape != std::vector<int64_t>{cout, cin, k} || b->shape != std::vector<int64_t>{cout}) {
        set_error("bad shape for " + prefix);
        return false;
    }
    std::vector<float> out((size_t)cout * k * cin);
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci)
            for (int t = 0; t < k; ++t) out[((size_t)co * k + t) * cin + ci] = w->data[((size_t)co * cin + ci) * k + t];
    w_off = p.add(out);
    b_off = p.add(b->data);
    return true;
}

// ConvTranspose1d weight [cin][cout][k = 2s] -> rows (p*cout + co), cols [x[t-1] block | x[t] block]
bool pack_convtr(const at_encodec* h, const std::string& prefix, int cin, int cout, int s, Packer& p, size_t& w_off,
                 size_t& b_off) {
    const HostTensor* w = find(h, prefix + ".weight");
    const HostTensor* b = find(h, prefix + ".bias");
    if (!w || !b) { set_error("missing tensor " + prefix + ".{weight,bias}"); return false; }
    const int k = 2 * s;
    if (w->shape != std::vector<int64_t>{cin, cout, k} || b->shape != std::vector<int64_t>{cout}) {
        set_error("bad shape for " + prefix);
        return false;
    }
    std::vector<float> out((size_t)s * cout * 2 * cin);
    for (int ph = 0; ph < s; ++ph)
        for (int co = 0; co < cout; ++co) {
            float* row = &out[((size_t)ph * cout + co) * 2 * cin];
            for (int ci = 0; ci < cin; ++ci) {
                row[ci] = w->data[((size_t)ci * cout + co) * k + ph + s];  // x[t-1] contributes tap p+s
                row[cin + ci] = w->data[((size_t)ci * cout + co) * k + ph];  // x[t] contributes tap p
            }
        }
    std::vector<float> bias((size_t)s * cout);
    for (int ph = 0; ph < s; ++ph)
        for (int co = 0; co < cout; ++co) bias[(size_t)ph * cout + co] = b->data[co];
    w_off = p.add(out);
    b_off = p.add(bias);
    return true;
}

// residual block tail: [W1 (C x C/2) | Wsc (C x C)] rows concatenated along K, biases summed
bool pack_res_tail(const at_encodec* h, const std::string& p1, const std::string& psc, int C, Packer& p, size_t& w_off, size_t& b_off) {
    const HostTensor* w1 = find(h, p1 + ".weight");
    const HostTensor* b1 = find(h, p1 + ".bias");
    const HostTensor* ws = find(h, psc + ".weight");
    const HostTensor* bs = find(h, psc + ".bias");
    if (!w1 || !b1 || !ws || !bs) { set_error("missing tensor " + p1 + " / " + psc); return false; }
    if (w1->shape != std::vector<int64_t>{C, C / 2, 1} || ws->shape != std::vector<int64_t>{C, C, 1} ||
        b1->shape != std::vector<int64_t>{C} || bs->shape != std::vector<int64_t>{C}) {
        set_error("bad shape for " + p1 + " / " + psc);
        return false;
    }
    const int K = C / 2 + C;
    std::vector<float> w((size_t)C * K), b(C);
    for (int co = 0; co < C; ++co) {
        for (int ci = 0; ci < C / 2; ++ci) w[(size_t)co * K + ci] = w1->data[(size_t)co * (C / 2) + ci];
        for (int ci = 0; ci < C; ++ci) w[(size_t)co * K + C / 2 + ci] = ws->data[(size_t)co * C + ci];
        // reference order: shortcut(x) + block(x) -> (Wsc.x + bsc) + (W1.h + b1); the GEMM adds ONE bias to the
        // full dot product, so the two biases are pre-added (a 1-ulp reassociation, inside the 1e-3 budget)
        b[co] = bs->data[co] + b1->data[co];
    }
    w_off = p.add(w);
    b_off = p.add(b);
    return true;
}

bool pack_lstm(const at_encodec* h, const std::string& prefix, Packer& p, size_t off[2][4]) {
    for (int l = 0; l < 2; ++l) {
        const char* names[4] = {"weight_ih", "weight_hh", "bias_ih", "bias_hh"};
        for (int which = 0; which < 4; ++which) {
            const std::string key = prefix + ".lstm." + names[which] + "_l" + std::to_string(l);
            const HostTensor* t = find(h, key);
            if (!t) { set_error("missing tensor " + key); return false; }
            const bool is_w = which < 2;
            if ((is_w && t->shape != std::vector<int64_t>{4 * kH, kH}) || (!is_w && t->shape != std::vector<int64_t>{4 * kH})) {
                set_error("bad shape for " + key);
                return false;
            }
            const size_t cols = is_w ? kH : 1;
            std::vector<float> out(t->data.size());
            for (int g = 0; g < 4; ++g)
                for (int j = 0; j < kH; ++j)
                    std::memcpy(&out[((size_t)j * 4 + g) * cols], &t->data[((size_t)g * kH + j) * cols], cols * sizeof(float));
            off[l][which] = p.add(out);
        }
    }
    return true;
}

void set_conv(ConvW& c, const float* blob, size_t w_off, size_t b_off, int cin, int cout, int k, int stride) {
    c.w = blob + w_off; c.b = blob + b_off; c.cin = cin; c.cout = cout; c.k = k; c.stride = stride;
}

int out_len(int L, int stride) { return (L + stride - 1) / stride; }

// One causal conv as a windowed GEMM over `batch` clips.
int conv_gemm(const ConvW& c, const float* X, long long x_bstride, int Tin, float* C, long long c_bstride, int M, int batch,
              int pro, const float* R, long long r_bstride, hipStream_t stream, int pad_mode = 1, int epi = EPI_NONE) {
    GemmArgs a;
    a.X = X; a.x_bstride = x_bstride; a.Tin = Tin; a.Cin = c.cin; a.ldx = c.cin;
    a.ktaps = c.k; a.stride = c.stride; a.pad_left = c.k - c.stride; a.pad_mode = pad_mode;
    a.W = c.w; a.bias = c.b;
    a.C = C; a.c_bstride = c_bstride; a.ldc = c.cout;
    a.R = R; a.r_bstride = r_bstride; a.ldr = c.cout;
    a.M = M; a.N = c.cout; a.K = c.k * c.cin; a.batch = batch;
    a.pro = pro; a.epi = epi; a.alpha = 1.0f;
    return launch_gemm(a, stream);
}

// SEANet residual block: out = shortcut(x) + conv1(ELU(conv3(ELU(x)))) as TWO windowed GEMMs:
//   h   = ELU(conv3(ELU(x)))                             K = 3C,  N = C/2   (the inner ELU once per element, in the epilogue)
//   out = [h | x] . [W1 | Wsc]^T + (b1 + bsc)            K = C/2 + C, N = C   (dual-source A, weights concatenated
// at finalize) — one pass less over the block output than "shortcut, then accumulate".
int resblock(const ConvW (&r)[3], const float* x, float* hbuf, float* out, int L, int batch, hipStream_t stream, int epi = EPI_NONE) {
    const int C = r[2].cout;
    const long long xs = (long long)L * C, hs = (long long)L * (C / 2);
    if (int rc = conv_gemm(r[0], x, xs, L, hbuf, hs, L, batch, PRO_ELU, nullptr, 0, stream, 1, EPI_ELU)) return rc;
    GemmArgs a;
    a.X = hbuf; a.x_bstride = hs; a.Tin = L; a.Cin = C / 2; a.ldx = C / 2;
    a.X2 = x; a.x2_bstride = xs; a.ld2 = C; a.K1 = C / 2;
    a.W = r[1].w; a.bias = r[1].b;     // r[1] holds the concatenated [C][C/2 + C] weight and the summed bias
    a.C = out; a.c_bstride = xs; a.ldc = C;
    a.M = L; a.N = C; a.K = C / 2 + C; a.batch = batch; a.pro = PRO_NONE; a.epi = epi;
    return launch_gemm(a, stream);
}

// 2-layer LSTM + skip over [B][T][512]; xg/c/h0 are scratch. y = lstm(x) + x.
// Range table of a handle (device, zeroed per call): one {flag word, census word} pair per SITE = per place where activations are split into
// fp16 pieces. A split writer ORs XB_STATUS_F16_OVERFLOW into its site's flag word and raises the census word to the largest |x * scale| it saw
// (split_scheme.h, range_publish); at_encodec_range_report() returns the census, i.e. the measured headroom to 65504 per site.
enum AcSite { AS_STAGE0 = 0, AS_RES1, AS_DOWN1, AS_RES2, AS_DOWN2, AS_RES3_CONV, AS_RES3_TAIL, AS_LSTM_IH, AS_FINAL, AS_RVQ,
              AS_DEC_LSTM_IH, AS_DEC_UP, AS_DEC_RES, AC_NSITES };
static const char* const kAcSiteNames[AC_NSITES] = {"stage0", "res1", "down1", "res2", "down2", "res3_conv", "res3_tail", "lstm_ih", "final_conv_in", "rvq",
                                                   "dec_lstm_ih", "dec_up", "dec_res"};
// status word of the *_checked entry points: bit 0 = an LSTM hand-off wait gave up (sync[63]), bit 1 = fp16 range overflow at any site, bit 2 = a NaN /
// infinity reached the RVQ search (XB_STATUS_NONFINITE)
__global__ void status_combine_kernel(const unsigned* sync, const int* range_tab, int nsites, unsigned* out) {
    unsigned v = sync[63] ? 1u : 0u;
    for (int k = 0; k < nsites; ++k) v |= (unsigned)range_tab[2 * k] & (unsigned)(XB_STATUS_F16_OVERFLOW | XB_STATUS_NONFINITE);
    out[0] = v;
}
int launch_status_combine(const unsigned* sync, const int* range_tab, unsigned* out, hipStream_t stream) {
    hipLaunchKernelGGL(status_combine_kernel, dim3(1), dim3(1), 0, stream, sync, range_tab, (int)AC_NSITES, out);
    AT_CHECK_HIP(hipGetLastError());
    return 0;
}

int lstm_skip(const float* const wih[2], const float* const whh[2], const float* const bih[2], const float* const bhh[2],
              const float* x, float* xg, float* h0, float* h1, float* c, float* y, int B, int T, hipStream_t stream,
              Profiler& prof, unsigned* sync, bool persistent, int y_elu, const __bf16* const* wih_s = nullptr, __bf16* xs = nullptr,
              bool rec_x3 = false, unsigned spin_limit = 1u << 18, const piece_t* const* wih_f = nullptr, const float* wih_fs = nullptr,
              int* range_status = nullptr, const float* whh_fs = nullptr, float* xg2 = nullptr) {
    // small batches: both layers in one pipelined launch after layer 1's projection (lstm_pipe.hip) — same arithmetic, ~half the dependent steps
    const bool pipe = xg2 && persistent && rec_x3 && whh_fs && wih_f && wih_f[0] && wih_f[1] && wih_fs && xs && lstm_pipe_eligible(B, T);
    for (int layer = 0; layer < 2; ++layer) {
        const float* in = layer == 0 ? x : h0;
        float* hout = layer == 0 ? h0 : h1;
        GemmArgs g;
        g.X = in; g.x_bstride = 0; g.Tin = B * T; g.Cin = kH; g.ldx = kH;
        g.W = wih[layer]; g.bias = bih[layer];
        g.C = xg; g.ldc = 4 * kH; g.M = B * T; g.N = 4 * kH; g.K = kH; g.batch = 1;
        prof.begin("lstm_ih", 1, stream);
        if (wih_f && wih_f[layer] && xs) {   // two fp16 pieces per operand, three MFMA products (gemm_f16x2_tg.hip for full batches)
            const long long M = (long long)B * T, Mpad = (M + 255) / 256 * 256;
            if (int rc = launch_split_blocked(in, kH, M, Mpad, kH, xs, stream, XB_SCHEME_F16X2, XB_F16_ACT_SCALE, range_status)) return rc;
            Bf16x3Args a;
            a.A = xs; a.W = wih_f[layer]; a.bias = bih[layer]; a.M = (int)M; a.N = 4 * kH; a.K = kH; a.Mpad = (int)Mpad;
            a.epi = XB_EPI_LINEAR; a.C = xg; a.ldc = 4 * kH; a.alpha = 1.0f;
            a.scheme = XB_SCHEME_F16X2; a.acc_scale = 1.0f / (XB_F16_ACT_SCALE * wih_fs[layer]); a.split_scale = XB_F16_ACT_SCALE; a.status = range_status;
            if (int rc = launch_gemm_bf16x3(a, stream)) return rc;
        } else if (wih_s && wih_s[layer] && xs) {   // split-bf16 GEMM (gemm_bf16x3.hip): x -> 3 bf16 pieces, then 6 bf16 MFMAs per step
            const long long M = (long long)B * T, Mpad = (M + 255) / 256 * 256;
            if (int rc = launch_split_blocked(in, kH, M, Mpad, kH, xs, stream)) return rc;
            Bf16x3Args a;
            a.A = xs; a.W = wih_s[layer]; a.bias = bih[layer]; a.M = (int)M; a.N = 4 * kH; a.K = kH; a.Mpad = (int)Mpad;
            a.epi = XB_EPI_LINEAR; a.C = xg; a.ldc = 4 * kH; a.alpha = 1.0f;
            if (int rc = launch_gemm_bf16x3(a, stream)) return rc;
        } else if (int rc = launch_gemm(g, stream)) {
            return rc;
        }
        prof.end(stream);
        if (pipe) {
            LstmPipeArgs q;
            q.xg1 = xg; q.w_hh1 = whh[0]; q.b_hh1 = bhh[0]; q.w_ih2 = wih[1]; q.b_ih2 = bih[1]; q.w_hh2 = whh[1]; q.b_hh2 = bhh[1];
            q.h1 = h0; q.xg2 = xg2; q.h2 = h1; q.y_out = y; q.skip = x; q.sync = sync; q.B = B; q.T = T; q.y_elu = y_elu; q.spin_limit = spin_limit;
            q.ws_hh1 = whh_fs[0]; q.ws_ih2 = wih_fs[1]; q.ws_hh2 = whh_fs[1]; q.act_scale = XB_F16_ACT_SCALE;
            prof.begin("lstm_rec", 1, stream);
            if (int rc = launch_lstm_pipe(q, stream)) return rc;
            prof.end(stream);
            return 0;
        }
        if (persistent) {
            // whole sequence in one persistent launch per 256-clip block (lstm_seq.hip)
            const int maxc = rec_x3 ? lstm_seq_x3_max_clips() : lstm_seq_max_clips();
            prof.begin("lstm_rec", (B + maxc - 1) / maxc, stream);
            for (int c0 = 0; c0 < B; c0 += maxc) {
                LstmSeqArgs q;
                const long long ro = (long long)c0 * T;
                q.xg = xg + ro * 4 * kH; q.w_hh = whh[layer]; q.b_hh = bhh[layer]; q.h_out = hout + ro * kH;
                q.y_out = layer == 1 ? y + ro * kH : nullptr; q.skip = x + ro * kH; q.sync = sync;
                q.B = (B - c0) < maxc ? (B - c0) : maxc; q.T = T; q.n_groups = 0; q.h_bytes = 0; q.y_elu = y_elu; q.spin_limit = spin_limit;
                q.w_scale_f16 = (rec_x3 && whh_fs) ? whh_fs[layer] : 0.f;
                if (int rc = rec_x3 ? launch_lstm_seq_x3(q, stream) : launch_lstm_seq(q, stream)) return rc;
            }
            prof.end(stream);
            continue;
        }
        prof.begin("lstm_rec", T, stream);
        for (int t = 0; t < T; ++t) {
            GemmArgs s;
            s.X = hout + (long long)(t > 0 ? t - 1 : 0) * kH; s.x_bstride = 0; s.Tin = B; s.Cin = kH; s.ldx = T * kH;
            s.W = whh[layer]; s.M = B; s.N = 4 * kH; s.K = kH; s.batch = 1; s.ldc = 4 * kH;
            LstmStepArgs ls;
            ls.xg = xg; ls.b_hh = bhh[layer]; ls.c = c; ls.h_out = hout;
            ls.y_out = layer == 1 ? y : nullptr; ls.skip = x;
            ls.T = T; ls.t = t; ls.H = kH; ls.first = t == 0; ls.y_elu = y_elu;
            if (int rc = launch_lstm_step(s, ls, stream)) return rc;
        }
        prof.end(stream);
    }
    return 0;
}

constexpr int kPipeMaxClips = 80;   // lstm_pipe.hip: 5 groups of 16 clips x 48 workgroups on 256 CUs (the launcher checks the device)

struct EncPlan {
    int L[5];        // lengths: L[0] = N, L[s+1] = ceil(L[s]/ratio)
    int G;           // sub-batch
    size_t off_x[4], off_h[4], off_r[4];  // per-stage sub-batch buffers (floats)
    size_t off_x4, off_xg, off_xg2, off_h0, off_h1, off_c, off_y, off_emb, off_sync, off_xs;
    int Mpf = 0, Lpf = 0;   // final conv as a windowed GEMM: padded output rows / operand rows per clip
    int Mp3, Lp3; size_t off_s3;   // stage-3 strided conv the same way, its input split by a separate pass or by the block's tail GEMM
    int Mpc, Lpc; size_t off_ac3, off_at3;   // 256-channel block as two split-bf16 GEMMs: pieces of ELU(x) (2 front rows) and of [h | x]
    int Mp2, Lp2;    // stage-2 strided conv as a windowed split-bf16 GEMM: padded output rows, rows per phase plane of its input pieces
    size_t total_floats;
};

EncPlan make_plan(int B, int N, int sub) {
    EncPlan p;
    p.L[0] = N;
    for (int s = 0; s < 4; ++s) p.L[s + 1] = out_len(p.L[s], kRatiosEnc[s]);
    p.G = B < sub ? B : sub;
    size_t cur = 0;
    auto take = [&](size_t n) { size_t o = cur; cur += (n + 63) / 64 * 64; return o; };
    for (int s = 0; s < 4; ++s) {
        const size_t C = 32u << s;
        p.off_x[s] = take((size_t)p.G * p.L[s] * C);
        p.off_h[s] = take((size_t)p.G * p.L[s] * (C / 2));
        size_t rn = (size_t)p.G * p.L[s] * C;
        if (s == 2) {   // r[2] doubles as the K-blocked phase-major bf16 pieces of ELU(block output) (3 pieces x 2 B = 1.5 floats per element)
            p.Mp2 = (p.L[3] + 255) / 256 * 256;
            const int reach = p.Mp2 + (10 - 1) / 5, have = (p.L[2] + 5 + 4) / 5;
            p.Lp2 = ((have > reach ? have : reach) + 63) / 64 * 64;
            const size_t pn = (size_t)p.G * 5 * p.Lp2 * C * 3 / 2 + 64;
            rn = pn > rn ? pn : rn;
        }
        p.off_r[s] = take(rn);
    }
    {
        p.Mp3 = (p.L[4] + 255) / 256 * 256;
        const int reach = p.Mp3 + (16 - 1) / 8, have = (p.L[3] + 8 + 7) / 8;
        p.Lp3 = ((have > reach ? have : reach) + 63) / 64 * 64;
        p.off_s3 = take((size_t)p.G * 8 * p.Lp3 * 256 * 3 / 2 + 64);
        p.Mpc = (p.L[3] + 255) / 256 * 256;
        p.Lpc = (p.Mpc + 2 + 63) / 64 * 64;
        p.off_ac3 = take((size_t)p.G * p.Lpc * 256 * 3 / 2 + 64);
        p.off_at3 = take((size_t)p.G * p.Mpc * 384 * 3 / 2 + 64);
    }
    const size_t T = p.L[4];
    p.off_x4 = take((size_t)B * T * kH);
    p.off_xg = take((size_t)B * T * 4 * kH);
    p.off_xg2 = take(B <= kPipeMaxClips ? (size_t)B * T * 4 * kH : 0);   // layer-2 input gates of the pipelined LSTM launch (small batches)
    p.off_h0 = take((size_t)B * T * kH);
    p.off_h1 = take((size_t)B * T * kH);
    p.off_c = take((size_t)B * kH);
    p.off_y = take((size_t)B * T * kH);
    p.off_emb = take((size_t)B * T * kDim);
    p.off_sync = take(1024);
    // split copy of an LSTM layer's input (three bf16 pieces at most); the same region then holds the final conv's operand: the LSTM
    // output as two fp16 pieces in windowed layout [2][B][32][Lpf][16] (6 reflected front rows, output rows padded to 256 per clip)
    p.Mpf = ((int)T + 255) / 256 * 256;
    p.Lpf = p.Mpf + 8;
    const size_t xs_lstm = (((size_t)B * T + 255) / 256 * 256) * kH * 3 / 2, xs_fin = (size_t)B * p.Lpf * kH + 64;
    p.off_xs = take(xs_lstm > xs_fin ? xs_lstm : xs_fin);
    p.total_floats = cur;
    return p;
}

struct DecPlan {
    int L[5];  // L[0] = T, L[s+1] = L[s]*ratio
    int G;
    size_t off_z, off_x0, off_xg, off_xg2, off_h0, off_h1, off_c, off_y, off_sync, off_xs;
    size_t off_u[4], off_h[4], off_r[4];
    size_t off_ap;     // operand pieces of a transposed conv run as a windowed split GEMM: [2][G][Cin/16][Lpu][16] fp16 (one float per element)
    int dMpc = 0, dLpc = 0; size_t off_dac3 = 0, off_dat3 = 0;   // stage-0 block as split GEMMs: padded rows, k3 operand [2][G][16][dLpc][16], tail operand [2][G][24][dMpc][16]
    int Mpu[3], Lpu[3];   // per stage: padded output rows / operand rows per clip
    size_t total_floats;
};

DecPlan make_dec_plan(int B, int T, int sub) {
    DecPlan p;
    p.L[0] = T;
    for (int s = 0; s < 4; ++s) p.L[s + 1] = p.L[s] * kRatiosDec[s];
    p.G = B < sub ? B : sub;
    size_t cur = 0;
    auto take = [&](size_t n) { size_t o = cur; cur += (n + 63) / 64 * 64; return o; };
    p.off_z = take((size_t)B * T * kDim);
    p.off_x0 = take((size_t)B * T * kH);
    p.off_xg = take((size_t)B * T * 4 * kH);
    p.off_xg2 = take(B <= kPipeMaxClips ? (size_t)B * T * 4 * kH : 0);
    p.off_h0 = take((size_t)B * T * kH);
    p.off_h1 = take((size_t)B * T * kH);
    p.off_c = take((size_t)B * kH);
    p.off_y = take((size_t)B * T * kH);
    p.off_sync = take(1024);
    p.off_xs = take((((size_t)B * T + 255) / 256 * 256) * kH * 3 / 2);   // split-bf16 copy of an LSTM layer's input
    int C = kH;
    for (int s = 0; s < 4; ++s) {
        C /= 2;
        p.off_u[s] = take((size_t)p.G * p.L[s + 1] * C);
        p.off_h[s] = take((size_t)p.G * p.L[s + 1] * (C / 2));
        p.off_r[s] = take((size_t)p.G * p.L[s + 1] * C);
    }
    {
        size_t ap = 0;
        int Cin = kH;
        for (int s = 0; s < 3; ++s) {
            p.Mpu[s] = (p.L[s] + 255) / 256 * 256;
            p.Lpu[s] = p.Mpu[s] + 8;
            const size_t n = (size_t)p.G * Cin * p.Lpu[s];
            ap = n > ap ? n : ap;
            Cin /= 2;
        }
        p.off_ap = take(ap + 64);
    }
    p.dMpc = (p.L[1] + 255) / 256 * 256;
    p.dLpc = (p.dMpc + 2 + 63) / 64 * 64;
    p.off_dac3 = take((size_t)p.G * p.dLpc * 256 + 64);
    p.off_dat3 = take((size_t)p.G * p.dMpc * 384 + 64);
    p.total_floats = cur;
    return p;
}

}  // namespace

extern "C" {

int at_version(void) { return 1; }
const char* at_last_error(void) { return at::last_error_cstr(); }

at_encodec_t* at_encodec_create(int device_id) {
    int n = 0;
    if (!host_only_test() && (hipGetDeviceCount(&n) != hipSuccess || device_id < 0 || device_id >= n)) {
        set_error("at_encodec_create: no such HIP device " + std::to_string(device_id));
        return nullptr;
    }
    at_encodec* h = new at_encodec();
    h->device = device_id;
    return h;
}

int at_encodec_set_tensor(at_encodec_t* h, const char* name, const float* host_data, const int64_t* shape, int ndim) {
    AT_REQUIRE(h && name && host_data && shape && ndim >= 1 && ndim <= 4, "bad arguments");
    AT_REQUIRE(!h->finalized, "model already finalized");
    HostTensor t;
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) { t.shape.push_back(shape[i]); n *= (size_t)shape[i]; }
    t.data.assign(host_data, host_data + n);
    h->staged[name] = std::move(t);
    return 0;
}

int at_encodec_finalize(at_encodec_t* h, int with_decoder) {
    AT_REQUIRE(h && !h->finalized, "bad handle");
    DeviceGuard guard(h->device);
    AT_REQUIRE(guard.ok, "cannot select the handle's device");
    Packer p;
    struct Off { size_t w, b; };
    Off o_conv0, o_res[4][3], o_down[4], o_fin;
    size_t o_lstm[2][4];
    // conv0 keeps [32][7] (Cin = 1): tap-major == torch layout
    if (!pack_conv(h, "encoder.model.0.conv.conv", 1, 32, 7, p, o_conv0.w, o_conv0.b)) return -1;
    int C = 32, idx = 1;
    for (int s = 0; s < 4; ++s) {
        const std::string base = "encoder.model." + std::to_string(idx);
        if (!pack_conv(h, base + ".block.1.conv.conv", C, C / 2, 3, p, o_res[s][0].w, o_res[s][0].b)) return -1;
        if (!pack_res_tail(h, base + ".block.3.conv.conv", base + ".shortcut.conv.conv", C, p, o_res[s][1].w, o_res[s][1].b)) return -1;
        o_res[s][2] = o_res[s][1];
        if (!pack_conv(h, "encoder.model." + std::to_string(idx + 2) + ".conv.conv", C, 2 * C, 2 * kRatiosEnc[s], p,
                       o_down[s].w, o_down[s].b))
            return -1;
        C *= 2;
        idx += 3;
    }
    if (!pack_lstm(h, "encoder.model.13", p, o_lstm)) return -1;
    if (!pack_conv(h, "encoder.model.15.conv.conv", kH, kDim, 7, p, o_fin.w, o_fin.b)) return -1;

    // codebooks: consecutive layers 0..n-1
    int ncb = 0;
    while (find(h, "quantizer.vq.layers." + std::to_string(ncb) + "._codebook.embed")) ++ncb;
    AT_REQUIRE(ncb >= 1, "no codebooks (quantizer.vq.layers.0._codebook.embed) supplied");
    std::vector<float> cbs((size_t)ncb * kCodes * kDim), e2s((size_t)ncb * kCodes);
    for (int q = 0; q < ncb; ++q) {
        const std::string key = "quantizer.vq.layers." + std::to_string(q) + "._codebook.embed";
        const HostTensor* t = find(h, key);
        AT_REQUIRE(t->shape == (std::vector<int64_t>{kCodes, kDim}), "bad codebook shape");
        std::memcpy(&cbs[(size_t)q * kCodes * kDim], t->data.data(), (size_t)kCodes * kDim * sizeof(float));
        const HostTensor* e = find(h, key.substr(0, key.size() - 5) + "e2");
        if (e) {
            AT_REQUIRE(e->shape == (std::vector<int64_t>{kCodes}), "bad e2 shape");
            std::memcpy(&e2s[(size_t)q * kCodes], e->data.data(), kCodes * sizeof(float));
        } else {
            for (int n = 0; n < kCodes; ++n) {
                float acc = 0.f;
                for (int k = 0; k < kDim; ++k) { const float v = t->data[(size_t)n * kDim + k]; acc += v * v; }
                e2s[(size_t)q * kCodes + n] = acc;
            }
        }
    }
    const size_t o_cb = p.add(cbs), o_e2 = p.add(e2s);

    // decoder
    Off d_conv0 = {}, d_up[4] = {}, d_res[4][3] = {}, d_last = {};
    size_t d_lstm[2][4] = {};
    if (with_decoder) {
        if (!pack_conv(h, "decoder.model.0.conv.conv", kDim, kH, 7, p, d_conv0.w, d_conv0.b)) return -1;
        if (!pack_lstm(h, "decoder.model.1", p, d_lstm)) return -1;
        int Cd = kH, di = 3;
        for (int s = 0; s < 4; ++s) {
            if (!pack_convtr(h, "decoder.model." + std::to_string(di) + ".convtr.convtr", Cd, Cd / 2, kRatiosDec[s], p,
                             d_up[s].w, d_up[s].b))
                return -1;
            Cd /= 2;
            const std::string base = "decoder.model." + std::to_string(di + 1);
            if (!pack_conv(h, base + ".block.1.conv.conv", Cd, Cd / 2, 3, p, d_res[s][0].w, d_res[s][0].b)) return -1;
            if (!pack_res_tail(h, base + ".block.3.conv.conv", base + ".shortcut.conv.conv", Cd, p, d_res[s][1].w, d_res[s][1].b)) return -1;
            d_res[s][2] = d_res[s][1];
            di += 3;
        }
        if (!pack_conv(h, "decoder.model.15.conv.conv", 32, 1, 7, p, d_last.w, d_last.b)) return -1;
    }

    h->blob_floats = p.host.size();
    AT_CHECK_HIP(hipMalloc((void**)&h->blob, h->blob_floats * sizeof(float)));
    AT_CHECK_HIP(hipMemcpy(h->blob, p.host.data(), h->blob_floats * sizeof(float), hipMemcpyHostToDevice));
    const float* bl = h->blob;
    set_conv(h->conv0, bl, o_conv0.w, o_conv0.b, 1, 32, 7, 1);
    C = 32;
    for (int s = 0; s < 4; ++s) {
        set_conv(h->res[s][0], bl, o_res[s][0].w, o_res[s][0].b, C, C / 2, 3, 1);
        set_conv(h->res[s][1], bl, o_res[s][1].w, o_res[s][1].b, C / 2, C, 1, 1);
        set_conv(h->res[s][2], bl, o_res[s][2].w, o_res[s][2].b, C, C, 1, 1);
        set_conv(h->down[s], bl, o_down[s].w, o_down[s].b, C, 2 * C, 2 * kRatiosEnc[s], kRatiosEnc[s]);
        C *= 2;
    }
    set_conv(h->fin, bl, o_fin.w, o_fin.b, kH, kDim, 7, 1);
    for (int l = 0; l < 2; ++l) {
        h->wih[l] = bl + o_lstm[l][0]; h->whh[l] = bl + o_lstm[l][1]; h->bih[l] = bl + o_lstm[l][2]; h->bhh[l] = bl + o_lstm[l][3];
    }
    h->codebooks = bl + o_cb;
    h->e2 = bl + o_e2;
    h->n_codebooks = ncb;
    if (with_decoder) {
        set_conv(h->dconv0, bl, d_conv0.w, d_conv0.b, kDim, kH, 7, 1);
        int Cd = kH;
        for (int s = 0; s < 4; ++s) {
            // transposed conv as a k=2, stride-1, zero-left-pad GEMM with N = s*Cout
            set_conv(h->dup[s], bl, d_up[s].w, d_up[s].b, Cd, kRatiosDec[s] * (Cd / 2), 2, 1);
            Cd /= 2;
            set_conv(h->dres[s][0], bl, d_res[s][0].w, d_res[s][0].b, Cd, Cd / 2, 3, 1);
            set_conv(h->dres[s][1], bl, d_res[s][1].w, d_res[s][1].b, Cd / 2, Cd, 1, 1);
            set_conv(h->dres[s][2], bl, d_res[s][2].w, d_res[s][2].b, Cd, Cd, 1, 1);
        }
        set_conv(h->dlast, bl, d_last.w, d_last.b, 32, 1, 7, 1);
        for (int l = 0; l < 2; ++l) {
            h->dwih[l] = bl + d_lstm[l][0]; h->dwhh[l] = bl + d_lstm[l][1]; h->dbih[l] = bl + d_lstm[l][2]; h->dbhh[l] = bl + d_lstm[l][3];
        }
    }
    h->has_decoder = with_decoder != 0;
    {
        hipDeviceProp_t prop;
        AT_CHECK_HIP(hipGetDeviceProperties(&prop, h->device));
        const char* env = std::getenv("AUDIOTOKEN_LSTM_STEPWISE");
        h->persistent_lstm = prop.multiProcessorCount >= 256 && !(env && env[0] == '1');
    }
    h->staged.clear();
    {
        const char* e = std::getenv("AUDIOTOKEN_BF16X3_ACOUSTIC");
        h->bf16x3 = e ? std::atoi(e) != 0 : kBf16x3AcousticDefault;
        // which fused SEANet kernels use the split-bf16 variants: bit 0 stage-1 strided conv, bit 1 128-channel block, bit 2 64-channel block, bit 3 stage 0, bit 4 / 5 stage-2 / stage-3 strided conv (GEMM), bit 6 256-channel block (GEMMs), bit 7 LSTM recurrence, bit 8 RVQ search
        const char* m = std::getenv("AUDIOTOKEN_X3_KERNELS");
        const int mask = m ? std::atoi(m) : 511;
        h->down64_x3 = (mask & 1) != 0; h->res128_x3 = (mask & 2) != 0; h->res64_x3 = (mask & 4) != 0; h->stage0_x3 = (mask & 8) != 0;
        h->down128_x3 = (mask & 16) != 0;
        h->down256_x3 = (mask & 32) != 0;
        h->res256_x3 = (mask & 64) != 0;
        h->lstm_x3 = (mask & 128) != 0;
        h->rvq_x3 = (mask & 256) != 0;
    }
    if (h->bf16x3) {
        for (int dec = 0; dec < (with_decoder ? 2 : 1); ++dec)
            for (int l = 0; l < 2; ++l) {
                __bf16* d = nullptr;
                AT_CHECK_HIP(hipMalloc((void**)&d, (size_t)3 * 4 * kH * kH * sizeof(__bf16)));
                h->extra_allocs.push_back(d);
                if (int rc = launch_split_blocked(dec ? h->dwih[l] : h->wih[l], kH, 4 * kH, 4 * kH, kH, d, nullptr)) return rc;
                (dec ? h->dwih_s : h->wih_s)[l] = d;
                // the same weights as two fp16 pieces, scaled by a power of two into [2^14, 2^15)
                const size_t off = dec ? d_lstm[l][0] : o_lstm[l][0];
                float mx = 0.f;
                for (size_t i = 0; i < (size_t)4 * kH * kH; ++i) mx = std::fmax(mx, std::fabs(p.host[off + i]));
                const float sc = xb_weight_scale(mx);
                piece_t* f = nullptr;
                AT_CHECK_HIP(hipMalloc((void**)&f, (size_t)2 * 4 * kH * kH * sizeof(piece_t)));
                h->extra_allocs.push_back(f);
                if (int rc = launch_split_blocked(dec ? h->dwih[l] : h->wih[l], kH, 4 * kH, 4 * kH, kH, f, nullptr, XB_SCHEME_F16X2, sc, nullptr)) return rc;
                (dec ? h->dwih_f : h->wih_f)[l] = f;
                (dec ? h->dwih_fs : h->wih_fs)[l] = sc;
                // W_hh scale of the fp16-scheme recurrence (the kernel splits W_hh itself, once per launch)
                const size_t offh = dec ? d_lstm[l][1] : o_lstm[l][1];
                float mxh = 0.f;
                for (size_t i = 0; i < (size_t)4 * kH * kH; ++i) mxh = std::fmax(mxh, std::fabs(p.host[offh + i]));
                (dec ? h->dwhh_fs : h->whh_fs)[l] = xb_weight_scale(mxh);
            }
        {   // codebooks as plain (row-major) bf16 pieces for the RVQ search
            __bf16* d = nullptr;
            const long long n = (long long)h->n_codebooks * kCodes * kDim;
            AT_CHECK_HIP(hipMalloc((void**)&d, (size_t)3 * n * sizeof(__bf16)));
            h->extra_allocs.push_back(d);
            if (int rc = launch_split_plain(h->codebooks, n, d, nullptr)) return rc;
            h->cb_s = d;
            // and as two fp16 pieces of E * 2^k (one power of two for all codebooks: the order of the distances is untouched)
            float mx = 0.f;
            const size_t cb_off = o_cb;
            for (long long i = 0; i < n; ++i) mx = std::fmax(mx, std::fabs(p.host[cb_off + i]));
            h->cb_fs = xb_weight_scale(mx);
            __bf16* f = nullptr;
            AT_CHECK_HIP(hipMalloc((void**)&f, (size_t)2 * n * sizeof(__bf16)));
            h->extra_allocs.push_back(f);
            if (int rc = launch_split_plain(h->codebooks, n, f, nullptr, XB_SCHEME_F16X2, h->cb_fs)) return rc;
            h->cb_f = f;
        }
        {   // stage-2 strided conv weights [256][10 * 128] as K-blocked bf16 pieces
            __bf16* d = nullptr;
            AT_CHECK_HIP(hipMalloc((void**)&d, (size_t)3 * 256 * 1280 * sizeof(__bf16)));
            h->extra_allocs.push_back(d);
            if (int rc = launch_split_blocked(h->down[2].w, 1280, 256, 256, 1280, d, nullptr, XB_SCHEME_BF16X3, 1.0f, nullptr, 8, 5)) return rc;
            h->down2_s = d;
        }
        {   // stage-3 strided conv weights [512][16 * 256]
            __bf16* d = nullptr;
            AT_CHECK_HIP(hipMalloc((void**)&d, (size_t)3 * 512 * 4096 * sizeof(__bf16)));
            h->extra_allocs.push_back(d);
            if (int rc = launch_split_blocked(h->down[3].w, 4096, 512, 512, 4096, d, nullptr, XB_SCHEME_BF16X3, 1.0f, nullptr, 16, 8)) return rc;
            h->down3_s = d;
        }
        {   // 256-channel block: conv3 [128][3 * 256] and tail [256][128 + 256]
            __bf16 *d0 = nullptr, *d1 = nullptr;
            AT_CHECK_HIP(hipMalloc((void**)&d0, (size_t)3 * 128 * 768 * sizeof(__bf16)));
            AT_CHECK_HIP(hipMalloc((void**)&d1, (size_t)3 * 256 * 384 * sizeof(__bf16)));
            h->extra_allocs.push_back(d0); h->extra_allocs.push_back(d1);
            if (int rc = launch_split_blocked(h->res[3][0].w, 768, 128, 128, 768, d0, nullptr, XB_SCHEME_BF16X3, 1.0f, nullptr, 16, 1)) return rc;
            if (int rc = launch_split_blocked(h->res[3][1].w, 384, 256, 256, 384, d1, nullptr)) return rc;
            h->res3c_s = d0; h->res3t_s = d1;
        }
        {   // stage 0: Wsc . conv0 as one 7-tap conv of the waveform (float64 products, rounded once) for seanet_stage0x3.hip
            std::vector<float> f(32 * 7 + 32);
            const float* wt = p.host.data() + o_res[0][1].w;   // [32][16 + 32] = [W1 | Wsc]
            const float* bt = p.host.data() + o_res[0][1].b;   // b1 + bsc
            const float* w0 = p.host.data() + o_conv0.w;       // [32][7]
            const float* b0 = p.host.data() + o_conv0.b;
            for (int c = 0; c < 32; ++c) {
                for (int j = 0; j < 7; ++j) {
                    double acc = 0.0;
                    for (int k = 0; k < 32; ++k) acc += (double)wt[c * 48 + 16 + k] * (double)w0[k * 7 + j];
                    f[c * 7 + j] = (float)acc;
                }
                double accb = (double)bt[c];
                for (int k = 0; k < 32; ++k) accb += (double)wt[c * 48 + 16 + k] * (double)b0[k];
                f[32 * 7 + c] = (float)accb;
            }
            float* d = nullptr;
            AT_CHECK_HIP(hipMalloc((void**)&d, f.size() * sizeof(float)));
            h->extra_allocs.push_back(d);
            AT_CHECK_HIP(hipMemcpy(d, f.data(), f.size() * sizeof(float), hipMemcpyHostToDevice));
            h->sc0_w = d;
        }
        {   // power-of-two weight scales of the fused residual blocks' fp16 scheme (the kernels split their weights themselves, once per launch)
            auto wmax = [&](size_t off, size_t n) { float mx = 0.f; for (size_t i = 0; i < n; ++i) mx = std::fmax(mx, std::fabs(p.host[off + i])); return mx; };
            int Cc = 32;
            for (int s2 = 0; s2 < 4; ++s2) {
                h->res_fs[s2][0] = xb_weight_scale(wmax(o_res[s2][0].w, (size_t)(Cc / 2) * 3 * Cc));
                h->res_fs[s2][1] = xb_weight_scale(wmax(o_res[s2][1].w, (size_t)Cc * (Cc / 2 + Cc)));
                Cc *= 2;
            }
            Cc = 32;
            for (int s2 = 0; s2 < 4; ++s2) { h->down_fs[s2] = xb_weight_scale(wmax(o_down[s2].w, (size_t)2 * Cc * 2 * kRatiosEnc[s2] * Cc)); Cc *= 2; }
            if (with_decoder) {
                int Cd2 = kH / 2;
                for (int s2 = 0; s2 < 4; ++s2) {
                    h->dres_fs[s2][0] = xb_weight_scale(wmax(d_res[s2][0].w, (size_t)(Cd2 / 2) * 3 * Cd2));
                    h->dres_fs[s2][1] = xb_weight_scale(wmax(d_res[s2][1].w, (size_t)Cd2 * (Cd2 / 2 + Cd2)));
                    if (s2 == 0) {   // the 256-channel block's two weight matrices as fp16 pieces for the split GEMMs (as the encoder's chain_f[1], chain_f[2])
                        const float* src[2] = {h->dres[0][0].w, h->dres[0][1].w};
                        const int ns[2] = {128, 256}, ks[2] = {768, 384}, wcb[2] = {16, 0};
                        for (int j = 0; j < 2; ++j) {
                            piece_t* f = nullptr;
                            AT_CHECK_HIP(hipMalloc((void**)&f, (size_t)2 * ns[j] * ks[j] * sizeof(piece_t)));
                            h->extra_allocs.push_back(f);
                            if (int rc = launch_split_blocked(src[j], ks[j], ns[j], ns[j], ks[j], f, nullptr, XB_SCHEME_F16X2, h->dres_fs[0][j], nullptr, wcb[j], 1)) return rc;
                            h->dchain_f[j] = f;
                        }
                    }
                    if (s2 == 3) h->dtail_up_fs = xb_weight_scale(wmax(d_up[3].w, (size_t)64 * 128));   // the fused tail kernel's transposed conv [2 * 32][2 * 64]
                    if (s2 < 3) {   // transposed conv of this stage as a two-tap windowed split GEMM: [r * Cout][2 * Cin], Cin = 2 * Cd2
                        const int Cin_u = 2 * Cd2, Nu = kRatiosDec[s2] * Cd2, Ku = 2 * Cin_u;
                        if (Nu % 64 == 0 && Ku % 64 == 0) {
                            h->dup_fs[s2] = xb_weight_scale(wmax(d_up[s2].w, (size_t)Nu * Ku));
                            piece_t* f = nullptr;
                            AT_CHECK_HIP(hipMalloc((void**)&f, (size_t)2 * Nu * Ku * sizeof(piece_t)));
                            h->extra_allocs.push_back(f);
                            if (int rc = launch_split_blocked(h->dup[s2].w, Ku, Nu, Nu, Ku, f, nullptr, XB_SCHEME_F16X2, h->dup_fs[s2], nullptr, Cin_u / 16, 1)) return rc;
                            h->dup_f[s2] = f;
                        }
                    }
                    Cd2 /= 2;
                }
            }
        }
        {   // the same four weights as two fp16 pieces, each scaled by a power of two into [2^14, 2^15) (gemm_bf16x3.h, XB_SCHEME_F16X2)
            const float* src[4] = {h->down[2].w, h->res[3][0].w, h->res[3][1].w, h->down[3].w};
            const size_t off[4] = {o_down[2].w, o_res[3][0].w, o_res[3][1].w, o_down[3].w};
            const int ns[4] = {256, 128, 256, 512}, ks[4] = {1280, 768, 384, 4096};
            const int wcb[4] = {8, 16, 0, 16}, wst[4] = {5, 1, 1, 8};   // window description of the three convs (the tail is a plain linear layer)
            for (int j = 0; j < 4; ++j) {
                float mx = 0.f;
                for (size_t i = 0; i < (size_t)ns[j] * ks[j]; ++i) mx = std::fmax(mx, std::fabs(p.host[off[j] + i]));
                const float sc = xb_weight_scale(mx);
                piece_t* f = nullptr;
                AT_CHECK_HIP(hipMalloc((void**)&f, (size_t)2 * ns[j] * ks[j] * sizeof(piece_t)));
                h->extra_allocs.push_back(f);
                if (int rc = launch_split_blocked(src[j], ks[j], ns[j], ns[j], ks[j], f, nullptr, XB_SCHEME_F16X2, sc, nullptr, wcb[j], wst[j])) return rc;
                h->chain_f[j] = f;
                h->chain_fs[j] = sc;
            }
            {   // final conv [128][7 * 512]
                const size_t n = (size_t)kDim * 7 * kH;
                float mx = 0.f;
                for (size_t i = 0; i < n; ++i) mx = std::fmax(mx, std::fabs(p.host[o_fin.w + i]));
                h->fin_fs = xb_weight_scale(mx);
                piece_t* f = nullptr;
                AT_CHECK_HIP(hipMalloc((void**)&f, 2 * n * sizeof(piece_t)));
                h->extra_allocs.push_back(f);
                if (int rc = launch_split_blocked(h->fin.w, 7 * kH, kDim, kDim, 7 * kH, f, nullptr, XB_SCHEME_F16X2, h->fin_fs, nullptr, kH / 16, 1)) return rc;
                h->fin_f = f;
            }
        }
        AT_CHECK_HIP(hipDeviceSynchronize());
    }
    if (!host_only_test()) {
        AT_CHECK_HIP(hipMalloc((void**)&h->range_tab, 64 * sizeof(int)));
        h->extra_allocs.push_back(h->range_tab);
        AT_CHECK_HIP(hipMemset(h->range_tab, 0, 64 * sizeof(int)));
    }
    h->finalized = true;
    return 0;
}

void at_encodec_destroy(at_encodec_t* h) {
    if (!h) return;
    DeviceGuard guard(h->device);   // restores the caller's current device (destroy runs from garbage collection in Python)
    if (h->blob) (void)hipFree(h->blob);
    for (void* p : h->extra_allocs) (void)hipFree(p);
    delete h;
}

int at_encodec_num_codebooks(const at_encodec_t* h) { return h ? h->n_codebooks : 0; }

size_t at_encodec_workspace_bytes(const at_encodec_t* h, int B, int N) {
    if (B <= 0 || N <= 0) return 0;
    return make_plan(B, N, h ? h->sub_batch : sub_batch()).total_floats * sizeof(float);
}

static int encodec_encode_impl(at_encodec_t* h, const float* wav, const float* mask, int B, int N, int n_q, int16_t* codes, int* T_out,
                               float* emb_out, void* workspace, size_t workspace_bytes, at_stream_t stream_, unsigned* status_out) {
    (void)mask;  // the reference's AcousticEncoder.forward ignores attention_mask (audiotoken/encoder.py:44-52)
    AT_REQUIRE(h && h->finalized, "model not finalized");
    DeviceGuard guard(h->device);
    AT_REQUIRE(guard.ok, "cannot select the handle's device");
    AT_REQUIRE(wav && codes && workspace, "null pointer");
    AT_REQUIRE(B >= 1 && N >= 10, "need B >= 1 and N >= 10 samples");
    AT_REQUIRE(n_q >= 1 && n_q <= h->n_codebooks, "n_q out of range for the loaded codebooks");
    hipStream_t stream = (hipStream_t)stream_;
    const EncPlan p = make_plan(B, N, h->sub_batch);
    AT_REQUIRE(workspace_bytes >= p.total_floats * sizeof(float), "workspace too small");
    AT_REQUIRE(p.L[3] > 8, "clip too short for the strided convs");
    float* ws = (float*)workspace;
    const int T = p.L[4];
    if (T_out) *T_out = T;

    float* x4 = ws + p.off_x4;
    AT_CHECK_HIP(hipMemsetAsync(ws + p.off_sync, 0, 1024 * sizeof(unsigned), stream));   // LSTM flags + the LSTM status word
    AT_CHECK_HIP(hipMemsetAsync(h->range_tab, 0, 64 * sizeof(int), stream));
    for (int b0 = 0; b0 < B; b0 += p.G) {
        const int g = (B - b0) < p.G ? (B - b0) : p.G;
        static const char* kRes[4] = {"res0", "res1", "res2", "res3"};
        static const char* kDown[4] = {"down0", "down1", "down2", "down3"};
        Profiler& prof = h->prof;
        const bool fused0 = h->fused_stage0 && (N % 2 == 0);
        auto rs = [&](int site) { return h->range_tab + 2 * site; };
        if (fused0) {
            // conv0 + resblock(32) + ELU + strided conv in one kernel: 4 B in, 128 B out per sample (seanet_stage0.hip)
            Stage0Args sa;
            sa.wav = wav + (long long)b0 * N; sa.x1 = ws + p.off_x[1];
            sa.w0 = h->conv0.w; sa.b0 = h->conv0.b; sa.w3 = h->res[0][0].w; sa.b3 = h->res[0][0].b;
            sa.wt = h->res[0][1].w; sa.bt = h->res[0][1].b; sa.wd = h->down[0].w; sa.bd = h->down[0].b;
            sa.B = g; sa.N = N;
            sa.wsc0 = h->sc0_w; sa.bsc0 = h->sc0_w ? h->sc0_w + 32 * 7 : nullptr;
            if (h->res_f16x2) {
                sa.scheme = XB_SCHEME_F16X2; sa.act_scale = XB_F16_ACT_SCALE; sa.status = rs(AS_STAGE0);
                sa.w3_scale = h->res_fs[0][0]; sa.wt_scale = h->res_fs[0][1]; sa.wd_scale = h->down_fs[0];
            }
            prof.begin("stage0_fused", 1, stream);
            if (int rc = (h->stage0_x3 && h->bf16x3) ? launch_seanet_stage0x3(sa, stream) : launch_seanet_stage0(sa, stream)) return rc;
            prof.end(stream);
        } else {
            prof.begin("conv0", 1, stream);
            if (int rc = launch_conv0(wav + (long long)b0 * N, h->conv0.w, h->conv0.b, ws + p.off_x[0], g, N, stream)) return rc;
            prof.end(stream);
        }
        bool chain3 = false;   // stage-2 strided conv -> 256-channel block -> stage-3 strided conv as chained split GEMMs (no fp32 in between)
        // operand scheme of that chain: two fp16 pieces / three products (default) or three bf16 pieces / six products
        const bool cf = h->chain_f16x2 && h->chain_f[0] != nullptr;
        const int cnp = cf ? 2 : 3;
        auto chain_cfg = [&](Bf16x3Args& a, int j, const __bf16* w_bf16) {   // j: 0 = stage-2 strided conv, 1 = conv3 of the block, 2 = its tail, 3 = stage-3 conv
            if (cf) {
                a.W = h->chain_f[j]; a.scheme = XB_SCHEME_F16X2; a.acc_scale = 1.0f / (XB_F16_ACT_SCALE * h->chain_fs[j]);
                a.split_scale = XB_F16_ACT_SCALE; a.status = rs(j == 0 ? AS_DOWN2 : j == 1 ? AS_RES3_CONV : AS_RES3_TAIL);
            } else {
                a.W = w_bf16;
            }
        };
        for (int s = fused0 ? 1 : 0; s < 4; ++s) {
            const int C = 32 << s, L = p.L[s], Lo = p.L[s + 1];
            float* x = ws + p.off_x[s];
            float* r = ws + p.off_r[s];
            bool down2_gemm = false;
            float* out = s < 3 ? ws + p.off_x[s + 1] : x4 + (long long)b0 * T * kH;
            // stage 1 in one kernel (seanet_res64down.hip): the block output (the largest tensor of the path) stays in LDS
            const bool stage1 = s == 1 && h->fused_stage1 && h->fused_res64 && h->fused_down64 && h->bf16x3 && h->res64_x3 && h->down64_x3 && h->res_f16x2 &&
                                L % 4 == 0 && L >= 8;
            if (stage1) {
                ResDown64Args fa;
                fa.x = x; fa.out = out; fa.w3 = h->res[1][0].w; fa.b3 = h->res[1][0].b; fa.wt = h->res[1][1].w; fa.bt = h->res[1][1].b;
                fa.wd = h->down[1].w; fa.bd = h->down[1].b; fa.B = g; fa.L = L;
                fa.act_scale = XB_F16_ACT_SCALE; fa.w3_scale = h->res_fs[1][0]; fa.wt_scale = h->res_fs[1][1]; fa.wd_scale = h->down_fs[1];
                fa.status_res = rs(AS_RES1); fa.status_down = rs(AS_DOWN1);
                prof.begin("res1_down1", 1, stream);
                if (int rc = launch_seanet_res64down(fa, stream)) return rc;
                prof.end(stream);
                continue;
            }
            if (s == 1 && h->fused_res64) {
                // 64-channel block fused into one kernel: 256 B in + 256 B out per row (seanet_res64.hip)
                Res64Args ra;
                ra.x = x; ra.out = r; ra.w3 = h->res[1][0].w; ra.b3 = h->res[1][0].b; ra.wt = h->res[1][1].w; ra.bt = h->res[1][1].b;
                ra.B = g; ra.L = L;
                if (h->res_f16x2) { ra.scheme = XB_SCHEME_F16X2; ra.act_scale = XB_F16_ACT_SCALE; ra.w3_scale = h->res_fs[1][0]; ra.wt_scale = h->res_fs[1][1]; ra.status = rs(AS_RES1); }
                prof.begin("res1", 1, stream);
                if (int rc = (h->res64_x3 && h->bf16x3) ? launch_seanet_res64x3(ra, stream) : launch_seanet_res64(ra, stream)) return rc;
                prof.end(stream);
            } else if (s == 2 && h->fused_res128) {
                // 128-channel block fused: weights stationary in registers, h never leaves the CU (seanet_res128.hip)
                Res64Args ra;
                ra.x = x; ra.out = r; ra.w3 = h->res[2][0].w; ra.b3 = h->res[2][0].b; ra.wt = h->res[2][1].w; ra.bt = h->res[2][1].b;
                ra.B = g; ra.L = L;
                // with the strided conv as a split-bf16 GEMM the block writes that GEMM's operand pieces instead of fp32 rows
                down2_gemm = h->down128_x3 && h->res128_x3 && h->bf16x3 && h->down2_s && L % 5 == 0 && L >= 10;
                if (h->res_f16x2) { ra.scheme = XB_SCHEME_F16X2; ra.act_scale = XB_F16_ACT_SCALE; ra.w3_scale = h->res_fs[2][0]; ra.wt_scale = h->res_fs[2][1]; ra.status = rs(AS_RES2); }
                if (down2_gemm) {
                    ra.S = reinterpret_cast<__bf16*>(r); ra.Lp = p.Lp2;
                    if (cf) { ra.S_scheme = XB_SCHEME_F16X2; ra.S_scale = XB_F16_ACT_SCALE; ra.status = rs(AS_RES2); }
                }
                prof.begin("res2", down2_gemm ? 2 : 1, stream);
                const bool x3 = h->res128_x3 && h->bf16x3;
                if (int rc = (x3 && h->res128_rs && ra.scheme == XB_SCHEME_F16X2) ? launch_seanet_res128rs(ra, stream)
                             : x3 ? launch_seanet_res128x3(ra, stream) : launch_seanet_res128(ra, stream)) return rc;
                if (down2_gemm)
                    if (int rc = launch_reflect_front5(ra.S, g, 8, p.Lp2, stream, cnp)) return rc;
                prof.end(stream);
            } else if (s == 3 && chain3) {
                // conv3 (k3, 256 -> 128) on the ELU pieces the stage-2 GEMM wrote; its ELU_SPLIT epilogue fills K-blocks 0..7 of the tail's
                // operand (blocks 8..23 = the raw x pieces, also from the stage-2 GEMM); the tail's epilogue writes the stage-3 conv's operand
                __bf16* ac3 = reinterpret_cast<__bf16*>(ws + p.off_ac3);
                __bf16* at3 = reinterpret_cast<__bf16*>(ws + p.off_at3);
                __bf16* s3 = reinterpret_cast<__bf16*>(ws + p.off_s3);
                prof.begin(kRes[s], 3, stream);
                Bf16x3Args ca;
                ca.A = ac3; chain_cfg(ca, 1, h->res3c_s); ca.bias = h->res[3][0].b; ca.M = L; ca.Mpad = p.Mpc; ca.N = 128; ca.K = 768;
                ca.batch = g; ca.stride = 1; ca.cblocks = 16; ca.Lp = p.Lpc;
                ca.epi = XB_EPI_ELU_SPLIT; ca.S = at3; ca.Spad = p.Mpc; ca.Sphases = 1; ca.Sfront = 0; ca.Sblocks = 24; ca.Sblock0 = 0;
                if (int rc = launch_gemm_bf16x3(ca, stream)) return rc;
                Bf16x3Args ta;
                ta.A = at3; chain_cfg(ta, 2, h->res3t_s); ta.bias = h->res[3][1].b; ta.M = L; ta.Mpad = p.Mpc; ta.N = 256; ta.K = 384;
                ta.batch = g; ta.stride = 1; ta.cblocks = 24; ta.Lp = p.Mpc;
                ta.epi = XB_EPI_ELU_SPLIT; ta.S = s3; ta.Spad = p.Lp3; ta.Sphases = 8; ta.Sfront = 1;
                if (int rc = launch_gemm_bf16x3(ta, stream)) return rc;
                if (int rc = launch_reflect_front(s3, g, 16, 8, p.Lp3, 8, stream, cnp)) return rc;
                prof.end(stream);
            } else {
                prof.begin(kRes[s], 2, stream);
                // the block output is only ever consumed through ELU (by the strided conv): apply it once here
                if (int rc = resblock(h->res[s], x, ws + p.off_h[s], r, L, g, stream, EPI_ELU)) return rc;
                prof.end(stream);
            }
            prof.begin(kDown[s], 1, stream);
            if (s == 1 && h->fused_down64 && L % 4 == 0) {
                Down64Args da;
                da.x = r; da.out = out; da.w = h->down[1].w; da.b = h->down[1].b; da.B = g; da.L = L;
                if (h->res_f16x2) { da.scheme = XB_SCHEME_F16X2; da.act_scale = XB_F16_ACT_SCALE; da.w_scale = h->down_fs[1]; da.status = rs(AS_DOWN1); }
                if (int rc = (h->down64_x3 && h->bf16x3) ? launch_seanet_down64x3(da, stream) : launch_seanet_down64(da, stream)) return rc;
            } else if (s == 2 && down2_gemm) {
                Bf16x3Args ga;
                ga.A = reinterpret_cast<const __bf16*>(r); chain_cfg(ga, 0, h->down2_s); ga.bias = h->down[2].b;
                ga.M = Lo; ga.Mpad = p.Mp2; ga.N = 256; ga.K = 1280;
                ga.batch = g; ga.stride = 5; ga.cblocks = 8; ga.Lp = p.Lp2;
                chain3 = h->res256_x3 && h->down256_x3 && h->res3c_s && h->down3_s && Lo % 8 == 0 && Lo >= 16;
                if (chain3) {   // the next block reads pieces: raw x -> K-blocks 8..23 of its tail operand, ELU(x) (2 causal front rows) -> its conv3 operand
                    ga.epi = XB_EPI_RAW_ELU_SPLIT2;
                    ga.S = reinterpret_cast<__bf16*>(ws + p.off_at3); ga.Spad = p.Mpc; ga.Sphases = 1; ga.Sfront = 0; ga.Sblocks = 24; ga.Sblock0 = 8;
                    ga.S2 = reinterpret_cast<__bf16*>(ws + p.off_ac3); ga.S2pad = p.Lpc; ga.S2phases = 1; ga.S2front = 2;
                } else {
                    ga.epi = XB_EPI_LINEAR; ga.C = out; ga.ldc = 256;
                }
                if (int rc = launch_gemm_bf16x3(ga, stream)) return rc;
                if (chain3)
                    if (int rc = launch_reflect_front(ga.S2, g, 16, 1, p.Lpc, 2, stream, cnp)) return rc;
            } else if (s == 3 && h->down256_x3 && h->bf16x3 && h->down3_s && L % 8 == 0 && L >= 16) {
                __bf16* s3 = reinterpret_cast<__bf16*>(ws + p.off_s3);
                if (!chain3)
                    if (int rc = launch_split_phase_major(r, g, L, 256, 8, p.Lp3, s3, stream)) return rc;
                Bf16x3Args ga;
                ga.A = s3; ga.bias = h->down[3].b;
                if (chain3) chain_cfg(ga, 3, h->down3_s); else ga.W = h->down3_s;   // the stand-alone split pass writes bf16 pieces
                ga.M = Lo; ga.Mpad = p.Mp3; ga.N = 512; ga.K = 4096;
                ga.batch = g; ga.stride = 8; ga.cblocks = 16; ga.Lp = p.Lp3;
                ga.epi = XB_EPI_LINEAR; ga.C = out; ga.ldc = 512;
                if (int rc = launch_gemm_bf16x3(ga, stream)) return rc;
            } else if (int rc = conv_gemm(h->down[s], r, (long long)L * C, L, out, (long long)Lo * 2 * C, Lo, g, PRO_NONE, nullptr, 0, stream)) {
                return rc;
            }
            prof.end(stream);
        }
    }
    Profiler& prof = h->prof;
    float* y = ws + p.off_y;
    unsigned* sync = reinterpret_cast<unsigned*>(ws + p.off_sync);   // zeroed at the start of the call (the conv stack's range status lives in it)
    if (int rc = lstm_skip(h->wih, h->whh, h->bih, h->bhh, x4, ws + p.off_xg, ws + p.off_h0, ws + p.off_h1, ws + p.off_c, y, B, T, stream, prof,
                           sync, h->persistent_lstm, 1, h->bf16x3 ? h->wih_s : nullptr, reinterpret_cast<__bf16*>(ws + p.off_xs), h->bf16x3 && h->lstm_x3, h->lstm_spin_limit,
                           (h->bf16x3 && h->ih_f16x2) ? h->wih_f : nullptr, h->wih_fs, h->range_tab + 2 * AS_LSTM_IH, h->lstm_f16x2 ? h->whh_fs : nullptr,
                           (h->lstm_pipe && B <= kPipeMaxClips) ? ws + p.off_xg2 : nullptr))
        return rc;
    float* emb = emb_out ? emb_out : ws + p.off_emb;
    prof.begin("final_conv", 1, stream);
    if (h->bf16x3 && h->fin_f16x2 && h->fin_f && T > 6) {
        // y = ELU(lstm + skip) -> two fp16 pieces in windowed layout (6 reflected front rows), then the k = 7 conv as a windowed split GEMM
        __bf16* yp = reinterpret_cast<__bf16*>(ws + p.off_xs);
        int* range_status = h->range_tab + 2 * AS_FINAL;
        if (int rc = launch_split_windowed(y, B, T, kH, 1, 6, p.Lpf, yp, stream, XB_SCHEME_F16X2, XB_F16_ACT_SCALE, range_status)) return rc;
        Bf16x3Args fa;
        fa.A = yp; fa.W = h->fin_f; fa.bias = h->fin.b;
        fa.M = T; fa.Mpad = p.Mpf; fa.N = kDim; fa.K = 7 * kH;
        fa.batch = B; fa.stride = 1; fa.cblocks = kH / 16; fa.Lp = p.Lpf;
        fa.scheme = XB_SCHEME_F16X2; fa.acc_scale = 1.0f / (XB_F16_ACT_SCALE * h->fin_fs); fa.split_scale = XB_F16_ACT_SCALE; fa.status = range_status;
        fa.epi = XB_EPI_LINEAR; fa.C = emb; fa.ldc = kDim;
        if (int rc = launch_gemm_bf16x3(fa, stream)) return rc;
    } else if (int rc = conv_gemm(h->fin, y, (long long)T * kH, T, emb, (long long)T * kDim, T, B, PRO_NONE, nullptr, 0, stream)) {  // y holds ELU(lstm + skip)
        return rc;
    }
    prof.end(stream);
    prof.begin("rvq", 1, stream);
    const bool rf = h->rvq_f16x2 && h->cb_f;
    int rc = (h->rvq_x3 && h->bf16x3 && h->cb_s)
                 ? launch_rvq_encode_x3(emb, (long long)B * T, T, h->codebooks, rf ? h->cb_f : h->cb_s, (long long)h->n_codebooks * kCodes * kDim, h->e2, n_q,
                                        codes, stream, rf ? XB_SCHEME_F16X2 : XB_SCHEME_BF16X3, XB_F16_ACT_SCALE, h->cb_fs, h->range_tab + 2 * AS_RVQ)
                 : launch_rvq_encode(emb, (long long)B * T, T, h->codebooks, h->e2, n_q, codes, stream);
    prof.end(stream);
    if (rc) return rc;
    if (status_out) return launch_status_combine(sync, h->range_tab, status_out, stream);   // LSTM hand-off + every range verdict of the call, RVQ included
    return 0;
}

int at_encodec_encode(at_encodec_t* h, const float* wav, const float* mask, int B, int N, int n_q, int16_t* codes, int* T_out,
                      float* emb_out, void* workspace, size_t workspace_bytes, at_stream_t stream) {
    return encodec_encode_impl(h, wav, mask, B, N, n_q, codes, T_out, emb_out, workspace, workspace_bytes, stream, nullptr);
}

int at_encodec_encode_checked(at_encodec_t* h, const float* wav, const float* mask, int B, int N, int n_q, int16_t* codes, int* T_out,
                              float* emb_out, void* workspace, size_t workspace_bytes, at_stream_t stream, uint32_t* status_dev) {
    return encodec_encode_impl(h, wav, mask, B, N, n_q, codes, T_out, emb_out, workspace, workspace_bytes, stream, status_dev);
}

namespace {
struct BoolOption { const char* name; bool at_encodec::*member; };
const BoolOption kBoolOptions[] = {
    {"persistent_lstm", &at_encodec::persistent_lstm},
    {"fused_stage0", &at_encodec::fused_stage0},
    {"fused_res64", &at_encodec::fused_res64},
    {"fused_res128", &at_encodec::fused_res128},
    {"fused_down64", &at_encodec::fused_down64},
    {"fused_stage1", &at_encodec::fused_stage1},
    {"down64_x3", &at_encodec::down64_x3},
    {"rvq_x3", &at_encodec::rvq_x3},
    {"lstm_x3", &at_encodec::lstm_x3},
    {"res256_x3", &at_encodec::res256_x3},
    {"down256_x3", &at_encodec::down256_x3},
    {"down128_x3", &at_encodec::down128_x3},
    {"stage0_x3", &at_encodec::stage0_x3},
    {"res64_x3", &at_encodec::res64_x3},
    {"res128_x3", &at_encodec::res128_x3},
    {"fused_dectail", &at_encodec::fused_dectail},
    {"tail_f16x2", &at_encodec::tail_f16x2},
    {"dec_chain", &at_encodec::dec_chain},
    {"ih_f16x2", &at_encodec::ih_f16x2},
    {"res_f16x2", &at_encodec::res_f16x2},
    {"rvq_f16x2", &at_encodec::rvq_f16x2},
    {"fin_f16x2", &at_encodec::fin_f16x2},
    {"res128_rs", &at_encodec::res128_rs},
    {"up_f16x2", &at_encodec::up_f16x2},
    {"lstm_f16x2", &at_encodec::lstm_f16x2},
    {"lstm_pipe", &at_encodec::lstm_pipe},
    {"chain_f16x2", &at_encodec::chain_f16x2},
};
}  // namespace

int at_encodec_set_option(at_encodec_t* h, const char* name, int value) {
    AT_REQUIRE(h && name, "null pointer");
    const std::string n(name);
    for (const BoolOption& o : kBoolOptions)
        if (n == o.name) { h->*(o.member) = value != 0; return 0; }
    if (n == "lstm_spin_limit") { AT_REQUIRE(value >= 0, "lstm_spin_limit must be >= 0"); h->lstm_spin_limit = (unsigned)value; return 0; }
    if (n == "subbatch") { AT_REQUIRE(value >= 1, "subbatch must be >= 1"); h->sub_batch = value; return 0; }
    set_error(std::string("unknown option ") + name);
    return -1;
}

int at_encodec_get_option(const at_encodec_t* h, const char* name) {
    if (!h || !name) return -1;
    const std::string n(name);
    for (const BoolOption& o : kBoolOptions)
        if (n == o.name) return h->*(o.member) ? 1 : 0;
    if (n == "lstm_spin_limit") return (int)h->lstm_spin_limit;
    if (n == "subbatch") return h->sub_batch;
    return -1;
}

// The measured fp16 headroom of the LAST encode / decode call of this handle: for every site (at_encodec_range_sites) the largest |x * scale| a
// split writer saw, 0 for sites that did not run on the fp16 scheme; the scheme overflows at 65504. Synchronises the device.
int at_encodec_range_report(at_encodec_t* h, float* max_scaled, int cap) {
    AT_REQUIRE(h && h->finalized && h->range_tab && max_scaled && cap >= (int)AC_NSITES, "at_encodec_range_report: bad arguments");
    DeviceGuard guard(h->device);
    AT_REQUIRE(guard.ok, "cannot select the handle's device");
    int host[2 * AC_NSITES];
    AT_CHECK_HIP(hipDeviceSynchronize());
    AT_CHECK_HIP(hipMemcpy(host, h->range_tab, sizeof(host), hipMemcpyDeviceToHost));
    for (int k = 0; k < (int)AC_NSITES; ++k) { float f; std::memcpy(&f, &host[2 * k + 1], sizeof(f)); max_scaled[k] = f; }
    return (int)AC_NSITES;
}
int at_encodec_range_sites(char* names, size_t cap) {
    std::string s;
    for (int k = 0; k < (int)AC_NSITES; ++k) { s += kAcSiteNames[k]; s += "\n"; }
    if (!names || cap < s.size() + 1) return -(int)(s.size() + 1);
    std::memcpy(names, s.c_str(), s.size() + 1);
    return (int)AC_NSITES;
}

int at_encodec_profile(at_encodec_t* h, int enable) {
    AT_REQUIRE(h != nullptr, "null handle");
    h->prof.reset();
    h->prof.enabled = enable != 0;
    return 0;
}

int at_encodec_profile_read(at_encodec_t* h, char* names, size_t names_cap, float* total_ms, int* launches, int max_groups) {
    AT_REQUIRE(h && names && total_ms && launches, "null pointer");
    std::vector<float> ms;
    std::vector<int> ln;
    if (h->prof.read(ms, ln) != 0) { set_error("profile read: event query failed"); return -2; }
    std::string joined;
    int n = 0;
    for (size_t i = 0; i < h->prof.names.size() && n < max_groups; ++i, ++n) {
        joined += h->prof.names[i];
        joined += '\n';
        total_ms[n] = ms[i];
        launches[n] = ln[i];
    }
    AT_REQUIRE(joined.size() + 1 <= names_cap, "names buffer too small");
    std::memcpy(names, joined.c_str(), joined.size() + 1);
    return n;
}

size_t at_encodec_decode_workspace_bytes(const at_encodec_t* h, int B, int T) {
    if (B <= 0 || T <= 0) return 0;
    return make_dec_plan(B, T, h ? h->sub_batch : sub_batch()).total_floats * sizeof(float);
}

int at_encodec_decode(at_encodec_t* h, const int64_t* codes, int B, int K, int T, float* wav, void* workspace,
                      size_t workspace_bytes, at_stream_t stream_) {
    return at_encodec_decode_checked(h, codes, B, K, T, wav, workspace, workspace_bytes, stream_, nullptr);
}

int at_encodec_decode_checked(at_encodec_t* h, const int64_t* codes, int B, int K, int T, float* wav, void* workspace,
                              size_t workspace_bytes, at_stream_t stream_, uint32_t* status_dev) {
    AT_REQUIRE(h && h->finalized && h->has_decoder, "model not finalized with a decoder");
    DeviceGuard guard(h->device);
    AT_REQUIRE(guard.ok, "cannot select the handle's device");
    AT_REQUIRE(codes && wav && workspace, "null pointer");
    AT_REQUIRE(B >= 1 && T >= 7 && K >= 1 && K <= h->n_codebooks, "bad B/T/K");
    hipStream_t stream = (hipStream_t)stream_;
    const DecPlan p = make_dec_plan(B, T, h->sub_batch);
    AT_REQUIRE(workspace_bytes >= p.total_floats * sizeof(float), "workspace too small");
    float* ws = (float*)workspace;
    float* z = ws + p.off_z;
    Profiler& prof = h->prof;   // same HIP-event taps as the encoder (at_encodec_profile / at_encodec_profile_read)
    prof.begin("dec_rvq_conv0", 2, stream);
    if (int rc = launch_rvq_decode(codes, B, K, T, h->codebooks, z, stream)) return rc;
    float* x0 = ws + p.off_x0;
    if (int rc = conv_gemm(h->dconv0, z, (long long)T * kDim, T, x0, (long long)T * kH, T, B, PRO_NONE, nullptr, 0, stream)) return rc;
    prof.end(stream);
    float* y = ws + p.off_y;
    unsigned* sync = reinterpret_cast<unsigned*>(ws + p.off_sync);
    AT_CHECK_HIP(hipMemsetAsync(sync, 0, 1024 * sizeof(unsigned), stream));
    AT_CHECK_HIP(hipMemsetAsync(h->range_tab, 0, 64 * sizeof(int), stream));
    // every activation that is only consumed through ELU is stored already ELU'd (once per element, in the producer's
    // epilogue) so the transposed convs run the plain-linear GEMM path: y (LSTM + skip) and the block outputs of stages 0-2
    if (int rc = lstm_skip(h->dwih, h->dwhh, h->dbih, h->dbhh, x0, ws + p.off_xg, ws + p.off_h0, ws + p.off_h1, ws + p.off_c, y, B, T, stream, prof,
                           sync, h->persistent_lstm, 1, h->bf16x3 ? h->dwih_s : nullptr, reinterpret_cast<__bf16*>(ws + p.off_xs), h->bf16x3 && h->lstm_x3, h->lstm_spin_limit,
                           (h->bf16x3 && h->ih_f16x2) ? h->dwih_f : nullptr, h->dwih_fs, h->range_tab + 2 * AS_DEC_LSTM_IH, h->lstm_f16x2 ? h->dwhh_fs : nullptr,
                           (h->lstm_pipe && B <= kPipeMaxClips) ? ws + p.off_xg2 : nullptr))
        return rc;
    const int Lout = p.L[4];
    static const char* kUp[4] = {"dec_up0", "dec_up1", "dec_up2", "dec_up3"};
    static const char* kDRes[4] = {"dec_res0", "dec_res1", "dec_res2", "dec_res3"};
    for (int b0 = 0; b0 < B; b0 += p.G) {
        const int g = (B - b0) < p.G ? (B - b0) : p.G;
        const float* in = y + (long long)b0 * T * kH;
        int Cin = kH;
        bool tail_done = false;
        bool ap_ready = false;   // the next stage's transposed-conv operand already lies in `ap` as pieces
        for (int s = 0; s < 4; ++s) {
            const int Li = p.L[s], Lo = p.L[s + 1], Co = Cin / 2;
            if (s == 3 && h->fused_dectail && Li >= 8) {
                DecTailArgs da;
                da.x = in; da.out = wav + (long long)b0 * Lout;
                da.wu = h->dup[3].w; da.bu = h->dup[3].b; da.w3 = h->dres[3][0].w; da.b3 = h->dres[3][0].b;
                da.wt = h->dres[3][1].w; da.bt = h->dres[3][1].b; da.wl = h->dlast.w; da.bl = h->dlast.b;
                da.B = g; da.L = Li;
                prof.begin("dec_tail", 1, stream);
                if (h->tail_f16x2 && h->bf16x3 && h->dtail_up_fs > 0.f && h->dres_fs[3][0] > 0.f) {
                    da.act_scale = XB_F16_ACT_SCALE; da.wu_scale = h->dtail_up_fs; da.w3_scale = h->dres_fs[3][0]; da.wt_scale = h->dres_fs[3][1];
                    da.status = h->range_tab + 2 * AS_DEC_RES;
                    if (int rc = launch_seanet_dectail_x2(da, stream)) return rc;
                } else if (int rc = launch_seanet_dectail(da, stream)) {
                    return rc;
                }
                prof.end(stream);
                tail_done = true;
                break;
            }
            float* u = ws + p.off_u[s];
            prof.begin(kUp[s], 2, stream);
            // ConvTranspose1d(k = 2r, stride r) of the (already ELU'd) input, trimmed right by r, as one GEMM with N = r*Cout:
            // out[t][p*Cout + co] = x[t-1].W[:, co, p+r] + x[t].W[:, co, p]; [Li][r*Cout] is [Lo][Cout] in memory.
            if (s < 3 && h->bf16x3 && h->up_f16x2 && h->dup_f[s] && Li > 1) {
                // as a two-tap windowed split GEMM on the fp16 scheme: the (already ELU'd) input -> pieces with ONE ZERO front row (x[-1] = 0)
                __bf16* ap = reinterpret_cast<__bf16*>(ws + p.off_ap);
                int* range_status = h->range_tab + 2 * AS_DEC_UP;
                if (!ap_ready)   // (after the stage-0 chain the block's tail GEMM has already written these pieces)
                    if (int rc = launch_split_windowed(in, g, Li, Cin, 1, 1, p.Lpu[s], ap, stream, XB_SCHEME_F16X2, XB_F16_ACT_SCALE, range_status, 0)) return rc;
                ap_ready = false;
                Bf16x3Args ua;
                ua.A = ap; ua.W = h->dup_f[s]; ua.bias = h->dup[s].b;
                ua.M = Li; ua.Mpad = p.Mpu[s]; ua.N = kRatiosDec[s] * Co; ua.K = 2 * Cin;
                ua.batch = g; ua.stride = 1; ua.cblocks = Cin / 16; ua.Lp = p.Lpu[s];
                ua.scheme = XB_SCHEME_F16X2; ua.acc_scale = 1.0f / (XB_F16_ACT_SCALE * h->dup_fs[s]); ua.split_scale = XB_F16_ACT_SCALE; ua.status = range_status;
                ua.epi = XB_EPI_LINEAR; ua.C = u; ua.ldc = ua.N;
                if (int rc = launch_gemm_bf16x3(ua, stream)) return rc;
            } else if (int rc = conv_gemm(h->dup[s], in, (long long)Li * Cin, Li, u, (long long)Lo * Co, Li, g, PRO_NONE, nullptr, 0, stream, 0)) {
                return rc;
            }
            prof.end(stream);
            float* r = ws + p.off_r[s];
            prof.begin(kDRes[s], 1, stream);
            // stage 0 (256 channels) as the encoder's stage-3 block: one pass u -> ELU(u) pieces (+ reflect rows) and raw u pieces, the k3 conv and the tail as
            // split GEMMs; the tail's ELU -> pieces epilogue writes the NEXT transposed conv's operand (one zero front row): no fp32 block output, no split pass
            const bool chain0 = s == 0 && Co == 256 && h->dec_chain && h->bf16x3 && h->res_f16x2 && h->up_f16x2 && h->dchain_f[0] && h->dchain_f[1] && h->dup_f[1] && Lo >= 3;
            if (chain0) {
                __bf16* ac3 = reinterpret_cast<__bf16*>(ws + p.off_dac3);
                __bf16* at3 = reinterpret_cast<__bf16*>(ws + p.off_dat3);
                __bf16* apn = reinterpret_cast<__bf16*>(ws + p.off_ap);
                int* rs = h->range_tab + 2 * AS_DEC_RES;
                if (int rc = launch_zero_piece_rows(at3, (long long)2 * g * 24, p.dMpc, Lo, p.dMpc, stream)) return rc;
                if (int rc = launch_dec_res256_split(u, g, Lo, ac3, p.dLpc, at3, p.dMpc, XB_F16_ACT_SCALE, rs, stream)) return rc;
                Bf16x3Args ca;
                ca.A = ac3; ca.W = h->dchain_f[0]; ca.bias = h->dres[0][0].b; ca.M = Lo; ca.Mpad = p.dMpc; ca.N = 128; ca.K = 768;
                ca.batch = g; ca.stride = 1; ca.cblocks = 16; ca.Lp = p.dLpc;
                ca.scheme = XB_SCHEME_F16X2; ca.acc_scale = 1.0f / (XB_F16_ACT_SCALE * h->dres_fs[0][0]); ca.split_scale = XB_F16_ACT_SCALE; ca.status = rs;
                ca.epi = XB_EPI_ELU_SPLIT; ca.S = at3; ca.Spad = p.dMpc; ca.Sphases = 1; ca.Sfront = 0; ca.Sblocks = 24; ca.Sblock0 = 0;
                if (int rc = launch_gemm_bf16x3(ca, stream)) return rc;
                // the next stage's operand: [2][g][16][Lpu][16], row t at index t + 1; row 0 and the rows past the data zero
                if (int rc = launch_zero_piece_rows(apn, (long long)2 * g * 16, p.Lpu[1], 0, 1, stream)) return rc;
                if (int rc = launch_zero_piece_rows(apn, (long long)2 * g * 16, p.Lpu[1], Lo + 1, p.Lpu[1], stream)) return rc;
                Bf16x3Args ta;
                ta.A = at3; ta.W = h->dchain_f[1]; ta.bias = h->dres[0][1].b; ta.M = Lo; ta.Mpad = p.dMpc; ta.N = 256; ta.K = 384;
                ta.batch = g; ta.stride = 1; ta.cblocks = 24; ta.Lp = p.dMpc;
                ta.scheme = XB_SCHEME_F16X2; ta.acc_scale = 1.0f / (XB_F16_ACT_SCALE * h->dres_fs[0][1]); ta.split_scale = XB_F16_ACT_SCALE; ta.status = rs;
                ta.epi = XB_EPI_ELU_SPLIT; ta.S = apn; ta.Spad = p.Lpu[1]; ta.Sphases = 1; ta.Sfront = 1;
                if (int rc = launch_gemm_bf16x3(ta, stream)) return rc;
                ap_ready = true;
            } else if ((Co == 64 && h->fused_res64) || (Co == 128 && h->fused_res128)) {
                Res64Args ra;
                ra.x = u; ra.out = r; ra.w3 = h->dres[s][0].w; ra.b3 = h->dres[s][0].b; ra.wt = h->dres[s][1].w; ra.bt = h->dres[s][1].b;
                ra.B = g; ra.L = Lo;
                if (h->res_f16x2 && h->dres_fs[s][0] > 0.f) {   // the blocks' own contractions on the two-piece fp16 scheme, as in the encoder
                    ra.scheme = XB_SCHEME_F16X2; ra.act_scale = XB_F16_ACT_SCALE; ra.w3_scale = h->dres_fs[s][0]; ra.wt_scale = h->dres_fs[s][1];
                    ra.status = h->range_tab + 2 * AS_DEC_RES;
                }
                const bool x3_128 = h->res128_x3 && h->bf16x3;
                if (int rc = Co == 64 ? ((h->res64_x3 && h->bf16x3) ? launch_seanet_res64x3(ra, stream) : launch_seanet_res64(ra, stream))
                                      : (x3_128 && h->res128_rs && ra.scheme == XB_SCHEME_F16X2) ? launch_seanet_res128rs(ra, stream)
                                      : x3_128 ? launch_seanet_res128x3(ra, stream) : launch_seanet_res128(ra, stream)) return rc;
            } else {
                // the last block's output goes to conv_last, which applies the ELU itself
                if (int rc = resblock(h->dres[s], u, ws + p.off_h[s], r, Lo, g, stream, s < 3 ? EPI_ELU : EPI_NONE)) return rc;
            }
            prof.end(stream);
            in = r;
            Cin = Co;
        }
        if (!tail_done) {
            prof.begin("dec_tail", 1, stream);
            if (int rc = launch_conv_last(in, h->dlast.w, h->dlast.b, wav + (long long)b0 * Lout, g, Lout, stream)) return rc;
            prof.end(stream);
        }
    }
    if (status_dev) return launch_status_combine(sync, h->range_tab, status_dev, stream);   // LSTM hand-off + every range verdict of the call
    return 0;
}

int at_op_gemm(const at_gemm_desc* d, at_stream_t stream) {
    AT_REQUIRE(d != nullptr, "null descriptor");
    GemmArgs a;
    a.X = d->X; a.x_bstride = d->x_bstride; a.Tin = d->Tin; a.Cin = d->Cin; a.ldx = d->ldx;
    a.ktaps = d->ktaps; a.stride = d->stride; a.pad_left = d->pad_left; a.pad_mode = d->pad_mode;
    a.W = d->W; a.bias = d->bias; a.C = d->C; a.c_bstride = d->c_bstride; a.ldc = d->ldc;
    a.R = d->R; a.r_bstride = d->r_bstride; a.ldr = d->ldr;
    a.M = d->M; a.N = d->N; a.K = d->K; a.batch = d->batch; a.pro = d->pro; a.epi = d->epi; a.alpha = d->alpha;
    a.aux_off = d->aux_off; a.row_mask = d->row_mask;
    return launch_gemm(a, (hipStream_t)stream);
}

int at_op_rvq_encode(const float* x, int64_t rows, int T, const float* codebooks, const float* e2, int n_q, int16_t* codes,
                     at_stream_t stream) {
    AT_REQUIRE(x && codebooks && e2 && codes && T >= 1 && n_q >= 1, "bad arguments");
    return launch_rvq_encode(x, rows, T, codebooks, e2, n_q, codes, (hipStream_t)stream);
}

// The RVQ search with split dot products (rvq_encode_x3.hip) — what the product runs: scheme 1 = two fp16 pieces / three products (default, option
// "rvq_f16x2"), 0 = three bf16 pieces / six products (its range fallback). The codebooks are split into `workspace` first, as finalize() does.
int at_op_rvq_encode_split(const float* x, int64_t rows, int T, const float* codebooks, const float* e2, int n_q, int16_t* codes, int scheme,
                           float cb_max_abs, void* workspace, size_t workspace_bytes, int32_t* status_dev, at_stream_t stream_) {
    using namespace at;
    AT_REQUIRE(x && codebooks && e2 && codes && workspace && T >= 1 && n_q >= 1, "at_op_rvq_encode_split: bad arguments");
    AT_REQUIRE(scheme == XB_SCHEME_BF16X3 || scheme == XB_SCHEME_F16X2, "at_op_rvq_encode_split: scheme 0 (bf16x3) or 1 (f16x2)");
    hipStream_t stream = (hipStream_t)stream_;
    const long long n = (long long)n_q * kCodes * kDim;
    const int np = xb_pieces(scheme);
    AT_REQUIRE(workspace_bytes >= (size_t)np * n * sizeof(piece_t) + 2 * sizeof(int), "at_op_rvq_encode_split: workspace too small (pieces * n_q * 1024 * 128 * 2 + 8 bytes)");
    __bf16* pieces = reinterpret_cast<__bf16*>(workspace);
    int* pair = reinterpret_cast<int*>(reinterpret_cast<char*>(workspace) + (size_t)np * n * sizeof(piece_t));   // {flag, census} of this call
    const float cs = scheme == XB_SCHEME_F16X2 ? xb_weight_scale(cb_max_abs) : 1.0f;
    AT_CHECK_HIP(hipMemsetAsync(pair, 0, 2 * sizeof(int), stream));
    if (status_dev) AT_CHECK_HIP(hipMemsetAsync(status_dev, 0, sizeof(int32_t), stream));
    if (int rc = launch_split_plain(codebooks, n, pieces, stream, scheme, cs)) return rc;
    if (int rc = launch_rvq_encode_x3(x, rows, T, codebooks, pieces, n, e2, n_q, codes, stream, scheme, scheme == XB_SCHEME_F16X2 ? XB_F16_ACT_SCALE : 1.0f, cs, pair)) return rc;
    return launch_range_combine(pair, 1, reinterpret_cast<int*>(status_dev), stream);
}

}  // extern "C"
