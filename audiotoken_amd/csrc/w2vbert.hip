// semantic_m tokenizer (log-mel front-end -> Wav2Vec2-BERT conformer -> LayerNorm -> VQ) — host side.
// C ABI in include/audiotoken_hip.h. Replaces reference Wav2VecBertEncoder (audiotoken/encoder.py:111-186):
// processor (audiotoken/processors.py), HF Wav2Vec2BertModel with the reference's rel-pos SDPA attention
// (audiotoken/modeling_wav2vec2_bert.py:20-80), non-affine LayerNorm and VectorQuantize lookup.
// Arithmetic per SURVEY.md Appendix A.2 / A.3.
//
// The conformer is fp32 on the f32 matrix cores: token ids must equal the reference's fp32 CPU result and the
// bf16 autocast the reference uses on GPU is not reproducible. The front-end's frame
// arithmetic and DFT run in f64 (see w2vbert_kernels.hip: frame_prep_kernel / dft_f64_kernel for why).
//
// Weight repacking at finalize():
//   DFT            generated here in double: [520][400], rows 0..256 cos, 260..516 -sin (rest zero)
//   mel            [257][80] (reference layout) -> [80][260]
//   q,k,v Linear   -> one [3072][1024] matrix (one GEMM, one pass over the LayerNorm output)
//   pointwise_conv1 [2048][1024][1] -> rows interleaved (a_c, b_c) so GLU is the GEMM epilogue
//   depthwise_conv [1024][1][31] -> [31][1024]
//   distance_embedding [73][64] -> [80][64] zero padded (MFMA row tiles)
#include <map>
#include <string>
#include <vector>
#include <cstring>
#include <cmath>

#include "../../include/audiotoken_hip.h"
#include "at_common.h"
#include "w2vbert_kernels.h"
#include <cstdlib>
#include "gemm_bf16x3.h"
#include "packed_model.h"

namespace at {
const char* last_error_cstr();
}
using namespace at;

namespace {
constexpr int kHid = 1024, kFfn = 4096, kFeat = 160, kMel = 80, kFrame = 400, kHop = 160;
constexpr int kSpecLd = 520, kImOff = 260, kCodes = 2048, kBuckets = 73;

struct HostTensor {
    std::vector<int64_t> shape;
    std::vector<float> data;
};

struct LayerW {
    const float *ln_ffn1_g, *ln_ffn1_b, *w1a, *b1a, *w1b, *b1b;
    const float *ln_att_g, *ln_att_b, *wqkv, *bqkv, *dist, *wo, *bo;
    const float *ln_conv_g, *ln_conv_b, *pw1, *dw, *ln_dw_g, *ln_dw_b, *pw2;
    const float *ln_ffn2_g, *ln_ffn2_b, *w2a, *b2a, *w2b, *b2b;
    const float *ln_fin_g, *ln_fin_b;
    // the eight linear layers as 16-bit operand pieces (gemm_bf16x3.hip), per scheme [XB_SCHEME_*][W_*]; split lazily per scheme.
    // wscale: the power of two the fp16 scheme multiplied that weight by (max |w s| in [2^14, 2^15))
    const piece_t* ws[2][8] = {};
    float wscale[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
    const piece_t* dist_s = nullptr;   // the distance embeddings as fp16 pieces [2][96][64] (attention_f16x2_w8.hip's rel-pos table MFMAs; f16x2 scheme only)
    float dist_scale = 1.f;
    // f16x2: the power of two every activation is multiplied by before it is split, per split site (WSite). XB_F16_ACT_SCALE (16) everywhere, except that
    // the LayerNorm-fed sites get the PROVABLE scale of xb_ln_site_scale() (gemm_bf16x3.h) when the LayerNorm's gains are large enough for 16 to overflow (finalize)
    float site_scale[10] = {XB_F16_ACT_SCALE, XB_F16_ACT_SCALE, XB_F16_ACT_SCALE, XB_F16_ACT_SCALE, XB_F16_ACT_SCALE,
                            XB_F16_ACT_SCALE, XB_F16_ACT_SCALE, XB_F16_ACT_SCALE, XB_F16_ACT_SCALE, XB_F16_ACT_SCALE};
};
enum { W_1A = 0, W_1B, W_2A, W_2B, W_QKV, W_O, W_PW1, W_PW2 };
// arithmetic of the linear layers: the fp32 MFMA, or operand splits on the 16-bit matrix cores (gemm_bf16x3.h)
enum { ARITH_F32 = 0, ARITH_BF16X3 = 1, ARITH_F16X2 = 2 };
}  // namespace

struct at_w2vbert {
    int device = 0;
    bool finalized = false;
    std::map<std::string, HostTensor> staged;
    DeviceArena arena;          // every device allocation of finalize(), in order (packed_model.h: export / import of the finalized model)
    PackedHeader imp{};         // import_packed: the exporter's record (layer count, flags) while finalize is replayed
    std::vector<int> split_seq; // the schemes whose weight pieces exist, in the order they were split (= their order in the arena)
    int* range_tab = nullptr;   // device, {flag, census} per (layer, WSite), zeroed at the start of every encode (at_w2vbert_range_report / _layer_status read it)
    std::vector<int> layer_arith;   // per conformer layer: -1 = the handle's arithmetic, else ARITH_BF16X3 / ARITH_F16X2 for that layer only (option "layer_arith:<i>")
    const float *window = nullptr, *melw = nullptr;
    double* dft64 = nullptr;  // [520][400] DFT matrix in double (see dft_f64_kernel)
    const float *fp_ln_g = nullptr, *fp_ln_b = nullptr, *fp_w = nullptr, *fp_b = nullptr;
    std::vector<LayerW> layers;
    const float *codebook = nullptr, *e2 = nullptr;
    const piece_t* cb_s[2] = {};   // the code book as operand pieces, per scheme (the VQ score GEMM on the split kernel; option "vq_split")
    float cb_scale = 1.f;
    bool vq_split = true;
    bool vq_refine = true;      // option "vq_refine" (round 5): near-tie codes re-evaluated exactly (vq_argmax_kernel); 0 = the expanded fp32 form alone, as rounds 1-4
    int arith = ARITH_F16X2;   // linear layers: ARITH_* ($AUDIOTOKEN_SEMANTIC_ARITH = f32 | bf16x3 | f16x2; option "arith")
    bool split_done[2] = {false, false};
    std::map<const float*, float> wmax;   // max |w| of every uploaded tensor (the fp16 scheme's weight scales)
    bool dwconv_stream = true;  // option "dwconv_stream": depthwise conv + LayerNorm + swish on the streaming kernel (dwconv_stream.hip)
    int attn_w8 = -1;           // option "attn_w8": 1 / 0 = the 8-wave 64-key LDS-DMA attention (attention_f16x2_w8.hip) / its round-3 twin; -1 = $AUDIOTOKEN_ATTN_W8, default 1
    Profiler prof;
};

namespace {

const HostTensor* find(const at_w2vbert* h, const std::string& name) {
    auto it = h->staged.find(name);
    return it == h->staged.end() ? nullptr : &it->second;
}

// upload one packed tensor into its own device allocation (the model is ~1.8 GB: no second full host copy)
const float* upload(at_w2vbert* h, const std::vector<float>& v) {
    const size_t n = (v.size() + 3) / 4 * 4;
    float* d = static_cast<float*>(h->arena.alloc(n * sizeof(float)));
    if (!d) return nullptr;
    if (hipMemcpy(d, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    float mx = 0.f;
    for (float x : v) mx = std::fmax(mx, std::fabs(x));
    h->arena.blocks.back().wmax = mx;
    h->wmax[d] = mx;
    return d;
}
// import_packed: the tensor's bytes are already in the blob — take the next slice and the recorded max |w|
const float* reserve(at_w2vbert* h, size_t n_floats) {
    float* d = static_cast<float*>(h->arena.alloc((n_floats + 3) / 4 * 4 * sizeof(float)));
    if (d) h->wmax[d] = h->arena.blocks.back().wmax;
    return d;
}

const float* take(at_w2vbert* h, const std::string& name, std::vector<int64_t> shape, bool& ok) {
    if (h->arena.importing) {
        size_t n = 1;
        for (int64_t d : shape) n *= (size_t)d;
        const float* d = reserve(h, n);
        if (!d) ok = false;
        return d;
    }
    const HostTensor* t = find(h, name);
    if (!t) { set_error("missing tensor " + name); ok = false; return nullptr; }
    if (t->shape != shape) { set_error("bad shape for " + name); ok = false; return nullptr; }
    const float* d = upload(h, t->data);
    if (!d) { set_error("device allocation/copy failed for " + name); ok = false; }
    return d;
}

int frames_of(int N) { return N >= kFrame ? 1 + (N - kFrame) / kHop : 0; }
int tokens_of(int N, int mult) {
    int t = frames_of(N) / 2;
    if (mult > 0 && t % mult) t += mult - t % mult;
    return t;
}

struct Plan {
    int F, Tp;
    size_t off_frames, off_fmask, off_spec, off_logmel, off_stats, off_feats, off_amask, off_x, off_t1, off_big, off_tok, off_t1s, off_bigs, off_kvs;
    size_t Mpad;
    size_t total_floats;
};

Plan make_plan(int B, int N, int mult) {
    Plan p;
    p.F = frames_of(N);
    p.Tp = tokens_of(N, mult);
    size_t cur = 0;
    auto takef = [&](size_t n) { size_t o = cur; cur += (n + 63) / 64 * 64; return o; };
    const size_t M = (size_t)B * p.Tp, BF = (size_t)B * p.F;
    p.off_frames = takef(BF * kFrame * 2);  // float64 frames
    p.off_fmask = takef(BF);
    p.off_spec = takef(BF * kSpecLd);
    p.off_logmel = takef(BF * kMel);
    p.off_stats = takef((size_t)B * 2 * kMel);
    p.off_feats = takef(M * kFeat);
    p.off_amask = takef(M);
    p.off_x = takef(M * kHid);
    p.off_t1 = takef(M * kHid);
    p.off_big = takef(M * kFfn);
    p.off_tok = takef(M);
    p.Mpad = (M + 255) / 256 * 256;                       // split-bf16 operands: 3 pieces x 2 bytes = 1.5 floats per element
    p.off_t1s = takef(p.Mpad * kHid * 3 / 2);
    p.off_bigs = takef(p.Mpad * kFfn * 3 / 2);
    p.off_kvs = takef(p.Mpad * kHid * 2 * 2 / 2);        // k and v as two fp16 pieces each (XB_EPI_QKV): 2 x 2 x 2 bytes per element
    p.total_floats = cur;
    return p;
}

int linear(const float* X, int K, const float* W, const float* bias, float* C, int N, long long M, int epi, float alpha,
           const float* R, const float* row_mask, int ldc, hipStream_t stream) {
    GemmArgs a;
    a.X = X; a.x_bstride = 0; a.Tin = (int)M; a.Cin = K; a.ldx = K;
    a.W = W; a.bias = bias; a.C = C; a.ldc = ldc; a.R = R; a.ldr = ldc;
    a.M = (int)M; a.N = N; a.K = K; a.batch = 1; a.epi = epi; a.alpha = alpha; a.row_mask = row_mask;
    return launch_gemm(a, stream);
}

// Split the eight linear layers of every conformer layer into the 16-bit pieces of `scheme` (once per scheme)
int split_weights(at_w2vbert* h, int scheme) {
    if (h->split_done[scheme]) return 0;
    const int np = xb_pieces(scheme);
    for (LayerW& L : h->layers) {
        const float* src[8] = {L.w1a, L.w1b, L.w2a, L.w2b, L.wqkv, L.wo, L.pw1, L.pw2};
        const int ns[8] = {kFfn, kHid, kFfn, kHid, 3 * kHid, kHid, 2 * kHid, kHid}, ks[8] = {kHid, kFfn, kHid, kFfn, kHid, kHid, kHid, kHid};
        for (int j = 0; j < 8; ++j) {
            const int n = ns[j], k = ks[j];
            piece_t* d = static_cast<piece_t*>(h->arena.alloc((size_t)np * n * k * sizeof(piece_t)));
            if (!d) return -1;
            float sc = 1.0f;
            if (scheme == XB_SCHEME_F16X2) {
                auto it = h->wmax.find(src[j]);
                AT_REQUIRE(it != h->wmax.end(), "weight maximum not recorded");
                sc = xb_weight_scale(it->second);
                L.wscale[j] = sc;
            }
            if (!h->arena.importing)   // import_packed: the pieces are in the blob
                if (int rc = launch_split_blocked(src[j], k, n, n, k, d, nullptr, scheme, sc, nullptr)) return rc;
            L.ws[scheme][j] = d;
        }
        if (scheme == XB_SCHEME_F16X2) {   // distance embeddings -> pieces for the attention kernel's rel-pos table
            piece_t* d = static_cast<piece_t*>(h->arena.alloc((size_t)2 * 96 * 64 * sizeof(piece_t)));
            if (!d) return -1;
            auto it = h->wmax.find(L.dist);
            AT_REQUIRE(it != h->wmax.end(), "distance embedding maximum not recorded");
            L.dist_scale = xb_weight_scale(it->second);
            if (!h->arena.importing)
                if (int rc = launch_dist_split(L.dist, d, L.dist_scale, nullptr)) return rc;
            L.dist_s = d;
        }
    }
    if (h->codebook) {   // the VQ score GEMM dots = LN(x) . E^T [M x 1024] x [1024 x 2048]
        piece_t* d = static_cast<piece_t*>(h->arena.alloc((size_t)np * kCodes * kHid * sizeof(piece_t)));
        if (!d) return -1;
        float sc = 1.0f;
        if (scheme == XB_SCHEME_F16X2) {
            auto it = h->wmax.find(h->codebook);
            AT_REQUIRE(it != h->wmax.end(), "code book maximum not recorded");
            sc = xb_weight_scale(it->second);
            h->cb_scale = sc;
        }
        if (!h->arena.importing)
            if (int rc = launch_split_blocked(h->codebook, kHid, kCodes, kCodes, kHid, d, nullptr, scheme, sc, nullptr)) return rc;
        h->cb_s[scheme] = d;
    }
    AT_CHECK_HIP(hipDeviceSynchronize());
    h->split_done[scheme] = true;
    h->split_seq.push_back(scheme);
    return 0;
}

// One split-operand GEMM of the conformer: C / S = epi(A . W^T) with A given as pieces (gemm_bf16x3.hip)
// Sites of the handle's range table (gemm_bf16x3.h, launch_range_combine): where activations become fp16 pieces. The same site in every layer.
enum WSite { WS_LN_FFN1 = 0, WS_FFN1_HIDDEN, WS_LN_ATTN, WS_QKV_KV, WS_ATTENTION, WS_LN_CONV, WS_DWCONV, WS_LN_FFN2, WS_FFN2_HIDDEN, WS_OTHER, W_NSITES };
static const char* const kWSiteNames[W_NSITES] = {"ln_ffn1", "ffn1_hidden", "ln_attn", "qkv_kv", "attention", "ln_conv", "dwconv_out", "ln_ffn2", "ffn2_hidden", "other"};
static_assert((int)W_NSITES == 10, "LayerW::site_scale has one entry per WSite");
constexpr int kRangeLayers = 64;                              // rows of the range table: one per conformer layer
constexpr int kRangeInts = kRangeLayers * 2 * (int)W_NSITES;
struct SplitCtx {
    int scheme; int* tab; const LayerW* L;
    int* site(int k) const { return tab ? tab + 2 * k : nullptr; }
    // the activation scale of split site k of this layer (1 on the bf16x3 scheme: full fp32 exponent range)
    float act_scale(int k = WS_OTHER) const { return scheme == XB_SCHEME_F16X2 ? (L ? L->site_scale[k] : XB_F16_ACT_SCALE) : 1.0f; }
};
// which split site produced the A operand of linear layer w
int a_site_of(int w) {
    switch (w) {
        case W_1A: return WS_LN_FFN1;  case W_1B: return WS_FFN1_HIDDEN;  case W_2A: return WS_LN_FFN2;  case W_2B: return WS_FFN2_HIDDEN;
        case W_QKV: return WS_LN_ATTN; case W_O: return WS_ATTENTION;     case W_PW1: return WS_LN_CONV; default: return WS_DWCONV;
    }
}
int gemm_split(const SplitCtx& c, const piece_t* A, const LayerW& L, int w, const float* bias, int N, int K, long long M, long long Mpad, int epi,
               float alpha, float* C, const float* R, int ldc, piece_t* S, hipStream_t stream) {
    Bf16x3Args a;
    a.A = A; a.W = L.ws[c.scheme][w]; a.bias = bias; a.M = (int)M; a.N = N; a.K = K; a.Mpad = (int)Mpad;
    a.epi = epi; a.C = C; a.ldc = ldc; a.R = R; a.ldr = ldc; a.alpha = alpha; a.S = S; a.Spad = (int)Mpad;
    const int out_site = w == W_1A ? WS_FFN1_HIDDEN : w == W_2A ? WS_FFN2_HIDDEN : WS_OTHER;
    a.scheme = c.scheme; a.status = c.site(out_site);
    if (c.scheme == XB_SCHEME_F16X2) { a.acc_scale = 1.0f / (c.act_scale(a_site_of(w)) * L.wscale[w]); a.split_scale = c.act_scale(out_site); }
    return launch_gemm_bf16x3(a, stream);
}

}  // namespace

extern "C" {

at_w2vbert_t* at_w2vbert_create(int device_id) {
    int n = 0;
    if (!host_only_test() && (hipGetDeviceCount(&n) != hipSuccess || device_id < 0 || device_id >= n)) {
        set_error("at_w2vbert_create: no such HIP device " + std::to_string(device_id));
        return nullptr;
    }
    at_w2vbert* h = new at_w2vbert();
    h->device = device_id;
    return h;
}

int at_w2vbert_set_tensor(at_w2vbert_t* h, const char* name, const float* host_data, const int64_t* shape, int ndim) {
    AT_REQUIRE(h && name && host_data && shape && ndim >= 1 && ndim <= 4, "bad arguments");
    AT_REQUIRE(!h->finalized, "model already finalized");
    HostTensor t;
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) { t.shape.push_back(shape[i]); n *= (size_t)shape[i]; }
    t.data.assign(host_data, host_data + n);
    h->staged[name] = std::move(t);
    return 0;
}

// finalize(): staged host tensors -> device. With the arena in import mode (at_w2vbert_import_packed) the same code REPLAYS the allocation order over
// the packed blob: no host tensor is read, nothing is uploaded or split — only the pointers and scales are rebuilt.
static int finalize_impl(at_w2vbert* h) {
    const bool imp = h->arena.importing;
    bool ok = true;
    // ---- front-end tables --------------------------------------------------------------------
    h->window = take(h, "frontend.window", {kFrame}, ok);
    if (!ok) return -1;
    if (imp) {
        h->melw = reserve(h, (size_t)kMel * kImOff);
        h->dft64 = static_cast<double*>(h->arena.alloc((size_t)kSpecLd * kFrame * sizeof(double)));
        AT_REQUIRE(h->melw && h->dft64, "import_packed: front-end tables");
    } else {
        const HostTensor* mf = find(h, "frontend.mel_filters");
        AT_REQUIRE(mf && mf->shape == (std::vector<int64_t>{257, kMel}), "frontend.mel_filters [257,80] missing");
        std::vector<float> m((size_t)kMel * kImOff, 0.f);
        for (int k = 0; k < 257; ++k)
            for (int j = 0; j < kMel; ++j) m[(size_t)j * kImOff + k] = mf->data[(size_t)k * kMel + j];
        h->melw = upload(h, m);
        std::vector<double> d((size_t)kSpecLd * kFrame, 0.0);
        const double two_pi = 6.283185307179586476925286766559;
        for (int k = 0; k < 257; ++k)
            for (int t = 0; t < kFrame; ++t) {
                const int ph = (int)(((long long)k * t) % 512);  // exact argument reduction
                const double ang = two_pi * ph / 512.0;
                d[(size_t)k * kFrame + t] = std::cos(ang);
                d[(size_t)(kImOff + k) * kFrame + t] = -std::sin(ang);
            }
        h->dft64 = static_cast<double*>(h->arena.alloc(d.size() * sizeof(double)));
        AT_REQUIRE(h->melw != nullptr && h->dft64 != nullptr, "device allocation failed (front-end tables)");
        AT_CHECK_HIP(hipMemcpy(h->dft64, d.data(), d.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    // ---- feature projection ------------------------------------------------------------------
    h->fp_ln_g = take(h, "feature_projection.layer_norm.weight", {kFeat}, ok);
    h->fp_ln_b = take(h, "feature_projection.layer_norm.bias", {kFeat}, ok);
    h->fp_w = take(h, "feature_projection.projection.weight", {kHid, kFeat}, ok);
    h->fp_b = take(h, "feature_projection.projection.bias", {kHid}, ok);
    if (!ok) return -1;
    // ---- conformer layers ----------------------------------------------------------------------
    int nl = imp ? h->imp.n_layers : 0;
    while (!imp && find(h, "encoder.layers." + std::to_string(nl) + ".ffn1_layer_norm.weight")) ++nl;
    for (int i = 0; i < nl; ++i) {
        const std::string p = "encoder.layers." + std::to_string(i);
        LayerW L{};
        L.ln_ffn1_g = take(h, p + ".ffn1_layer_norm.weight", {kHid}, ok);
        L.ln_ffn1_b = take(h, p + ".ffn1_layer_norm.bias", {kHid}, ok);
        L.w1a = take(h, p + ".ffn1.intermediate_dense.weight", {kFfn, kHid}, ok);
        L.b1a = take(h, p + ".ffn1.intermediate_dense.bias", {kFfn}, ok);
        L.w1b = take(h, p + ".ffn1.output_dense.weight", {kHid, kFfn}, ok);
        L.b1b = take(h, p + ".ffn1.output_dense.bias", {kHid}, ok);
        L.ln_att_g = take(h, p + ".self_attn_layer_norm.weight", {kHid}, ok);
        L.ln_att_b = take(h, p + ".self_attn_layer_norm.bias", {kHid}, ok);
        if (!ok) return -1;
        if (imp) {
            L.wqkv = reserve(h, (size_t)3 * kHid * kHid);
            L.bqkv = reserve(h, (size_t)3 * kHid);
            L.dist = reserve(h, (size_t)80 * 64);
            AT_REQUIRE(L.wqkv && L.bqkv && L.dist, "import_packed: attention tensors");
        } else {
            std::vector<float> w((size_t)3 * kHid * kHid), b((size_t)3 * kHid);
            const char* nm[3] = {"linear_q", "linear_k", "linear_v"};
            for (int j = 0; j < 3; ++j) {
                const HostTensor* wt = find(h, p + ".self_attn." + nm[j] + ".weight");
                const HostTensor* bt = find(h, p + ".self_attn." + nm[j] + ".bias");
                AT_REQUIRE(wt && bt && wt->shape == (std::vector<int64_t>{kHid, kHid}) && bt->shape == (std::vector<int64_t>{kHid}),
                           "attention projection tensors missing or mis-shaped");
                std::memcpy(&w[(size_t)j * kHid * kHid], wt->data.data(), (size_t)kHid * kHid * sizeof(float));
                std::memcpy(&b[(size_t)j * kHid], bt->data.data(), kHid * sizeof(float));
            }
            L.wqkv = upload(h, w);
            L.bqkv = upload(h, b);
            const HostTensor* de = find(h, p + ".self_attn.distance_embedding.weight");
            AT_REQUIRE(de && de->shape == (std::vector<int64_t>{kBuckets, 64}), "distance_embedding [73,64] missing");
            std::vector<float> e((size_t)80 * 64, 0.f);
            std::memcpy(e.data(), de->data.data(), (size_t)kBuckets * 64 * sizeof(float));
            L.dist = upload(h, e);
            AT_REQUIRE(L.wqkv && L.bqkv && L.dist, "device allocation failed");
        }
        L.wo = take(h, p + ".self_attn.linear_out.weight", {kHid, kHid}, ok);
        L.bo = take(h, p + ".self_attn.linear_out.bias", {kHid}, ok);
        L.ln_conv_g = take(h, p + ".conv_module.layer_norm.weight", {kHid}, ok);
        L.ln_conv_b = take(h, p + ".conv_module.layer_norm.bias", {kHid}, ok);
        if (!ok) return -1;
        if (imp) {
            L.pw1 = reserve(h, (size_t)2 * kHid * kHid);
            L.dw = reserve(h, (size_t)31 * kHid);
            AT_REQUIRE(L.pw1 && L.dw, "import_packed: conv-module tensors");
        } else {
            const HostTensor* pw = find(h, p + ".conv_module.pointwise_conv1.weight");
            AT_REQUIRE(pw && pw->shape == (std::vector<int64_t>{2 * kHid, kHid, 1}), "pointwise_conv1 [2048,1024,1] missing");
            std::vector<float> w((size_t)2 * kHid * kHid);
            for (int c = 0; c < kHid; ++c) {
                std::memcpy(&w[(size_t)(2 * c) * kHid], &pw->data[(size_t)c * kHid], kHid * sizeof(float));
                std::memcpy(&w[(size_t)(2 * c + 1) * kHid], &pw->data[(size_t)(kHid + c) * kHid], kHid * sizeof(float));
            }
            L.pw1 = upload(h, w);
            const HostTensor* dw = find(h, p + ".conv_module.depthwise_conv.weight");
            AT_REQUIRE(dw && dw->shape == (std::vector<int64_t>{kHid, 1, 31}), "depthwise_conv [1024,1,31] missing");
            std::vector<float> d((size_t)31 * kHid);
            for (int c = 0; c < kHid; ++c)
                for (int j = 0; j < 31; ++j) d[(size_t)j * kHid + c] = dw->data[(size_t)c * 31 + j];
            L.dw = upload(h, d);
            AT_REQUIRE(L.pw1 && L.dw, "device allocation failed");
        }
        L.ln_dw_g = take(h, p + ".conv_module.depthwise_layer_norm.weight", {kHid}, ok);
        L.ln_dw_b = take(h, p + ".conv_module.depthwise_layer_norm.bias", {kHid}, ok);
        L.pw2 = take(h, p + ".conv_module.pointwise_conv2.weight", {kHid, kHid, 1}, ok);
        L.ln_ffn2_g = take(h, p + ".ffn2_layer_norm.weight", {kHid}, ok);
        L.ln_ffn2_b = take(h, p + ".ffn2_layer_norm.bias", {kHid}, ok);
        L.w2a = take(h, p + ".ffn2.intermediate_dense.weight", {kFfn, kHid}, ok);
        L.b2a = take(h, p + ".ffn2.intermediate_dense.bias", {kFfn}, ok);
        L.w2b = take(h, p + ".ffn2.output_dense.weight", {kHid, kFfn}, ok);
        L.b2b = take(h, p + ".ffn2.output_dense.bias", {kHid}, ok);
        L.ln_fin_g = take(h, p + ".final_layer_norm.weight", {kHid}, ok);
        L.ln_fin_b = take(h, p + ".final_layer_norm.bias", {kHid}, ok);
        if (!ok) return -1;
        {
            auto mx = [&](const float* d) { auto it = h->wmax.find(d); return it == h->wmax.end() ? 0.f : it->second; };
            L.site_scale[WS_LN_FFN1] = xb_ln_site_scale(mx(L.ln_ffn1_g), mx(L.ln_ffn1_b), kHid);
            L.site_scale[WS_LN_ATTN] = xb_ln_site_scale(mx(L.ln_att_g), mx(L.ln_att_b), kHid);
            L.site_scale[WS_LN_CONV] = xb_ln_site_scale(mx(L.ln_conv_g), mx(L.ln_conv_b), kHid);
            L.site_scale[WS_DWCONV] = xb_ln_site_scale(mx(L.ln_dw_g), mx(L.ln_dw_b), kHid);
            L.site_scale[WS_LN_FFN2] = xb_ln_site_scale(mx(L.ln_ffn2_g), mx(L.ln_ffn2_b), kHid);
        }
        h->layers.push_back(L);
        // free the staged host copies of this layer early
        for (auto it = h->staged.begin(); it != h->staged.end();)
            it = it->first.compare(0, p.size() + 1, p + ".") == 0 ? h->staged.erase(it) : std::next(it);
    }
    // ---- VQ codebook (state-dict key _codebook.embed [1, 2048, 1024], reference audiotoken/utils.py:331-339) ---
    if (imp) {
        if (h->imp.flags & 1) {
            h->codebook = reserve(h, (size_t)kCodes * kHid);
            h->e2 = reserve(h, kCodes);
            AT_REQUIRE(h->codebook && h->e2, "import_packed: code book");
        }
    } else if (const HostTensor* cb = find(h, "vq._codebook.embed")) {
        AT_REQUIRE((cb->shape == std::vector<int64_t>{1, kCodes, kHid}) || (cb->shape == std::vector<int64_t>{kCodes, kHid}),
                   "vq._codebook.embed must be [1,2048,1024]");
        h->codebook = upload(h, cb->data);
        std::vector<float> e2(kCodes);
        if (const HostTensor* e = find(h, "vq._codebook.e2")) {
            AT_REQUIRE(e->data.size() == (size_t)kCodes, "bad e2 shape");
            e2 = e->data;
        } else {
            for (int n2 = 0; n2 < kCodes; ++n2) {
                float acc = 0.f;
                for (int k = 0; k < kHid; ++k) { const float v = cb->data[(size_t)n2 * kHid + k]; acc += v * v; }
                e2[n2] = acc;
            }
        }
        h->e2 = upload(h, e2);
        AT_REQUIRE(h->codebook && h->e2, "device allocation failed (codebook)");
    }
    h->staged.clear();
    if (imp) {
        h->arith = h->imp.arith;
    } else {
        h->arith = ARITH_F16X2;
        if (const char* e = std::getenv("AUDIOTOKEN_SEMANTIC_ARITH")) {
            const std::string v(e);
            AT_REQUIRE(v == "f32" || v == "bf16x3" || v == "f16x2", "AUDIOTOKEN_SEMANTIC_ARITH must be f32, bf16x3 or f16x2");
            h->arith = v == "f32" ? ARITH_F32 : v == "bf16x3" ? ARITH_BF16X3 : ARITH_F16X2;
        }
    }
    if (imp) {
        // the exporter's splits in ITS order (flags bits 1-2 = count, bits 3.. = one bit per split: 1 = bf16x3). Normally one: the default scheme at
        // finalize; two when the per-batch range fallback had run there (the other scheme is split lazily, and the handle's current arithmetic may be either)
        const int n = (h->imp.flags >> 1) & 3;
        for (int i = 0; i < n; ++i)
            if (int rc = split_weights(h, ((h->imp.flags >> (3 + i)) & 1) ? XB_SCHEME_BF16X3 : XB_SCHEME_F16X2)) return rc;
    } else if (h->arith != ARITH_F32) {
        if (int rc = split_weights(h, h->arith == ARITH_F16X2 ? XB_SCHEME_F16X2 : XB_SCHEME_BF16X3)) return rc;
    }
    if (!host_only_test() && !h->range_tab) {   // run-time state, not part of the packed model
        // the table and, behind it, its per-encode initial image: row 0 = {flag 0, census 0} per site; every further row {flag 0, LINK to row 0's census word of
        // that site} (split_scheme.h, range_publish): one flag word per (row, site), one census word per site
        AT_CHECK_HIP(hipMalloc((void**)&h->range_tab, 2 * kRangeInts * sizeof(int)));
        std::vector<int> init(kRangeInts, 0);
        for (int r = 1; r < kRangeLayers; ++r)
            for (int k = 0; k < (int)W_NSITES; ++k) init[(r * (int)W_NSITES + k) * 2 + 1] = -(r * (int)W_NSITES * 2);
        AT_CHECK_HIP(hipMemcpy(h->range_tab + kRangeInts, init.data(), kRangeInts * sizeof(int), hipMemcpyHostToDevice));
        AT_CHECK_HIP(hipMemcpy(h->range_tab, init.data(), kRangeInts * sizeof(int), hipMemcpyHostToDevice));
    }
    h->finalized = true;
    return 0;
}

int at_w2vbert_finalize(at_w2vbert_t* h) {
    AT_REQUIRE(h && !h->finalized, "bad handle");
    DeviceGuard guard(h->device);
    AT_REQUIRE(guard.ok, "cannot select the handle's device");
    return finalize_impl(h);
}

// ---- the finalized model as one device blob (packed_model.h) ------------------------------------------------------------------------------
static int packed_flags(const at_w2vbert* h) {
    int f = (h->codebook ? 1 : 0) | ((int)h->split_seq.size() << 1);
    for (size_t i = 0; i < h->split_seq.size(); ++i) f |= (h->split_seq[i] == XB_SCHEME_BF16X3 ? 1 : 0) << (3 + i);
    return f;
}
int64_t at_w2vbert_packed_bytes(at_w2vbert_t* h) {
    if (!h || !h->finalized) { set_error("at_w2vbert_packed_bytes: model not finalized"); return -1; }
    return (int64_t)h->arena.packed_bytes();
}
int64_t at_w2vbert_packed_meta(at_w2vbert_t* h, void* host_dst, int64_t cap) {
    if (!h || !h->finalized) { set_error("at_w2vbert_packed_meta: model not finalized"); return -1; }
    return packed_write_meta(h->arena, PACKED_MODEL_W2VBERT, (int)h->layers.size(), packed_flags(h), h->arith, host_dst, cap);
}
int at_w2vbert_export_packed(at_w2vbert_t* h, void* device_dst, int64_t bytes, void* stream) {
    AT_REQUIRE(h && h->finalized, "at_w2vbert_export_packed: model not finalized");
    DeviceGuard guard(h->device);
    AT_REQUIRE(guard.ok, "cannot select the handle's device");
    return packed_export(h->arena, device_dst, bytes, (hipStream_t)stream);
}
int at_w2vbert_import_packed(at_w2vbert_t* h, const void* host_meta, int64_t meta_bytes, const void* device_src, int64_t bytes, void* stream) {
    AT_REQUIRE(h && !h->finalized && h->staged.empty(), "at_w2vbert_import_packed needs a fresh handle (no set_tensor, no finalize)");
    DeviceGuard guard(h->device);
    AT_REQUIRE(guard.ok, "cannot select the handle's device");
    if (int rc = packed_begin_import(h->arena, PACKED_MODEL_W2VBERT, host_meta, meta_bytes, device_src, bytes, (hipStream_t)stream, &h->imp)) return rc;
    int rc = finalize_impl(h);
    if (!rc) rc = packed_end_import(h->arena);
    if (rc) {
        // a failed import leaves an EMPTY handle that can only be destroyed (or imported into again): not a half-built model that reports `finalized`
        h->finalized = false;
        h->arena.importing = false;
        h->layers.clear();
        h->split_seq.clear();
        h->split_done[0] = h->split_done[1] = false;
        h->wmax.clear();
        h->codebook = h->e2 = nullptr;
        h->cb_s[0] = h->cb_s[1] = nullptr;
        h->arena.free_all();
        if (h->range_tab) { (void)hipFree(h->range_tab); h->range_tab = nullptr; }
    }
    return rc;
}

void at_w2vbert_destroy(at_w2vbert_t* h) {
    if (!h) return;
    DeviceGuard guard(h->device);   // restores the caller's current device (destroy runs from garbage collection in Python)
    h->arena.free_all();
    if (h->range_tab) (void)hipFree(h->range_tab);
    delete h;
}

int at_w2vbert_num_layers(const at_w2vbert_t* h) { return h ? (int)h->layers.size() : 0; }
int at_w2vbert_num_tokens(int N, int pad_to_multiple_of) { return tokens_of(N, pad_to_multiple_of); }

size_t at_w2vbert_workspace_bytes(const at_w2vbert_t* h, int B, int N, int pad_to_multiple_of) {
    (void)h;
    if (B <= 0 || N < kFrame) return 0;
    return make_plan(B, N, pad_to_multiple_of).total_floats * sizeof(float);
}

int at_w2vbert_profile(at_w2vbert_t* h, int enable) {
    AT_REQUIRE(h != nullptr, "null handle");
    h->prof.reset();
    h->prof.enabled = enable != 0;
    return 0;
}

int at_w2vbert_profile_read(at_w2vbert_t* h, char* names, size_t names_cap, float* total_ms, int* launches, int max_groups) {
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

int at_w2vbert_set_option(at_w2vbert_t* h, const char* name, int value) {
    AT_REQUIRE(h && h->finalized && name, "bad handle");
    const std::string n(name);
    if (n == "arith") {
        AT_REQUIRE(value == ARITH_F32 || value == ARITH_BF16X3 || value == ARITH_F16X2, "arith: 0 = f32 MFMA, 1 = bf16x3, 2 = f16x2");
        DeviceGuard guard(h->device);
        AT_REQUIRE(guard.ok, "cannot select the handle's device");
        if (value != ARITH_F32)
            if (int rc = split_weights(h, value == ARITH_F16X2 ? XB_SCHEME_F16X2 : XB_SCHEME_BF16X3)) return rc;
        h->arith = value;
        return 0;
    }
    if (n.rfind("layer_arith:", 0) == 0) {   // "layer_arith:<i>": -1 = follow "arith", 1 = bf16x3, 2 = f16x2 for conformer layer i only
        const int li = std::atoi(n.c_str() + 12);
        AT_REQUIRE(li >= 0 && li < (int)h->layers.size(), "layer_arith: no such layer");
        AT_REQUIRE(value == -1 || value == ARITH_BF16X3 || value == ARITH_F16X2, "layer_arith:<i>: -1 = the handle's arithmetic, 1 = bf16x3, 2 = f16x2");
        if (value > 0) {
            DeviceGuard guard(h->device);
            AT_REQUIRE(guard.ok, "cannot select the handle's device");
            if (int rc = split_weights(h, value == ARITH_F16X2 ? XB_SCHEME_F16X2 : XB_SCHEME_BF16X3)) return rc;
        }
        if (h->layer_arith.size() < h->layers.size()) h->layer_arith.resize(h->layers.size(), -1);
        h->layer_arith[li] = value;
        return 0;
    }
    if (n == "dwconv_stream") { h->dwconv_stream = value != 0; return 0; }
    if (n == "vq_split") { h->vq_split = value != 0; return 0; }
    if (n == "vq_refine") { h->vq_refine = value != 0; return 0; }
    if (n == "attn_w8") { h->attn_w8 = value < 0 ? -1 : (value != 0); return 0; }
    set_error("at_w2vbert_set_option: unknown option " + n);
    return -1;
}

int at_w2vbert_get_option(const at_w2vbert_t* h, const char* name) {
    if (!h || !name) return -1;
    if (std::string(name) == "arith") return h->arith;
    if (std::string(name).rfind("layer_arith:", 0) == 0) {
        const int li = std::atoi(name + 12);
        return (li >= 0 && li < (int)h->layer_arith.size()) ? h->layer_arith[li] : -1;
    }
    if (std::string(name) == "dwconv_stream") return h->dwconv_stream ? 1 : 0;
    if (std::string(name) == "vq_split") return h->vq_split ? 1 : 0;
    if (std::string(name) == "vq_refine") return h->vq_refine ? 1 : 0;
    if (std::string(name) == "attn_w8") return h->attn_w8;
    return -1;
}

int at_w2vbert_encode(at_w2vbert_t* h, const float* wav, const float* mask, int B, int N, int pad_to_multiple_of, int n_layers,
                      int16_t* tokens, int* T_out, float* features_out, float* attn_mask_out, float* hidden_out, void* workspace,
                      size_t workspace_bytes, at_stream_t stream_) {
    return at_w2vbert_encode_checked(h, wav, mask, B, N, pad_to_multiple_of, n_layers, tokens, T_out, features_out, attn_mask_out, hidden_out,
                                     workspace, workspace_bytes, stream_, nullptr);
}

int at_w2vbert_encode_checked(at_w2vbert_t* h, const float* wav, const float* mask, int B, int N, int pad_to_multiple_of, int n_layers,
                              int16_t* tokens, int* T_out, float* features_out, float* attn_mask_out, float* hidden_out, void* workspace,
                              size_t workspace_bytes, at_stream_t stream_, int32_t* status_dev) {
    AT_REQUIRE(h && h->finalized, "model not finalized");
    DeviceGuard guard(h->device);
    AT_REQUIRE(guard.ok, "cannot select the handle's device");
    AT_REQUIRE(wav && workspace, "null pointer");
    AT_REQUIRE(B >= 1 && N >= kFrame + kHop, "need B >= 1 and at least two frames (N >= 560 samples)");
    AT_REQUIRE(n_layers >= 0 && n_layers <= (int)h->layers.size(), "n_layers exceeds the loaded layers");
    AT_REQUIRE(tokens == nullptr || h->codebook != nullptr, "tokens requested but no VQ codebook loaded");
    hipStream_t stream = (hipStream_t)stream_;
    const Plan p = make_plan(B, N, pad_to_multiple_of);
    AT_REQUIRE(workspace_bytes >= p.total_floats * sizeof(float), "workspace too small");
    AT_REQUIRE(p.Tp >= 1, "clip too short");
    float* ws = (float*)workspace;
    const int F = p.F, T = p.Tp;
    const long long M = (long long)B * T, BF = (long long)B * F;
    if (T_out) *T_out = T;
    Profiler& prof = h->prof;
    if (status_dev) AT_CHECK_HIP(hipMemsetAsync(status_dev, 0, sizeof(int32_t), stream));
    const bool split = h->arith != ARITH_F32;
    AT_REQUIRE(n_layers <= kRangeLayers, "more conformer layers than range-table rows");
    AT_CHECK_HIP(hipMemcpyAsync(h->range_tab, h->range_tab + kRangeInts, kRangeInts * sizeof(int), hipMemcpyDeviceToDevice, stream));   // flags 0, census 0 / links
    // arithmetic per layer: the handle's, unless that layer is pinned to another split scheme (option "layer_arith:<i>": what the product's range
    // fallback sets for a layer whose activations do not fit fp16 — the other layers stay on f16x2). Each layer has its own row of the range table.
    auto arith_of = [&](int li) { return (split && li < (int)h->layer_arith.size() && h->layer_arith[li] > 0) ? h->layer_arith[li] : h->arith; };
    auto ctx_of = [&](int li) { return SplitCtx{arith_of(li) == ARITH_F16X2 ? XB_SCHEME_F16X2 : XB_SCHEME_BF16X3, h->range_tab + li * 2 * (int)W_NSITES, &h->layers[li]}; };
    const SplitCtx sc_model{h->arith == ARITH_F16X2 ? XB_SCHEME_F16X2 : XB_SCHEME_BF16X3, nullptr, nullptr};   // the VQ score GEMM (no range site, non-affine LayerNorm: scale 16)

    // ---- log-mel front-end (reference processors.py) -------------------------------------------
    double* frames = reinterpret_cast<double*>(ws + p.off_frames);
    float* fmask = ws + p.off_fmask;
    float* spec = ws + p.off_spec;
    float* logmel = ws + p.off_logmel;
    float* feats = features_out ? features_out : ws + p.off_feats;
    float* amask = attn_mask_out ? attn_mask_out : ws + p.off_amask;
    prof.begin("frontend", 5, stream);
    if (int rc = launch_frame_prep(wav, mask, h->window, frames, fmask, B, N, F, stream, reinterpret_cast<int*>(status_dev))) return rc;
    if (int rc = launch_dft_f64(frames, h->dft64, spec, BF, kSpecLd, stream)) return rc;
    {   // |X|^2 folded into the mel projection's prologue, log(max(., floor)) into its epilogue
        GemmArgs a;
        a.X = spec; a.Tin = (int)BF; a.Cin = kImOff; a.ldx = kSpecLd; a.W = h->melw; a.C = logmel; a.ldc = kMel;
        a.M = (int)BF; a.N = kMel; a.K = kImOff; a.batch = 1; a.pro = PRO_POWER; a.aux_off = kImOff; a.epi = EPI_LOGFLOOR;
        if (int rc = launch_gemm(a, stream)) return rc;
    }
    if (int rc = launch_fbank_normalize(logmel, fmask, ws + p.off_stats, feats, amask, B, F, T, stream)) return rc;
    prof.end(stream);

    // ---- feature projection; padded rows zeroed (HF encoder entry) ------------------------------
    float* x = ws + p.off_x;
    float* t1 = ws + p.off_t1;
    float* big = ws + p.off_big;
    piece_t* t1s = reinterpret_cast<piece_t*>(ws + p.off_t1s);
    piece_t* bigs = reinterpret_cast<piece_t*>(ws + p.off_bigs);
    piece_t* kvs = reinterpret_cast<piece_t*>(ws + p.off_kvs);
    const bool attn_kvp = true;   // k / v as pieces from the projection's epilogue whenever the arithmetic is f16x2
    prof.begin("feature_projection", 2, stream);
    if (int rc = launch_layernorm(feats, h->fp_ln_g, h->fp_ln_b, nullptr, t1, M, kFeat, stream)) return rc;
    if (int rc = linear(t1, kFeat, h->fp_w, h->fp_b, x, kHid, M, EPI_NONE, 1.f, nullptr, amask, kHid, stream)) return rc;
    prof.end(stream);

    const long long Mpad = (long long)p.Mpad;
    for (int li = 0; li < n_layers; ++li) {
        const LayerW& L = h->layers[li];
        const SplitCtx sc = ctx_of(li);
        const int attn_arith = arith_of(li);   // attention follows the layer's arithmetic (0: the fp32-MFMA kernel)
        if (split) {
            // Split arithmetic: every GEMM operand is produced directly as K-blocked pieces — LayerNorm (launch_layernorm_split), the first
            // FFN GEMM's swish epilogue, the attention kernel's context and the depthwise-conv kernel's output — so no fp32 activation is
            // written only to be re-read by a split pass.
            if (li == 0) {   // layers > 0: the previous layer's final LayerNorm wrote these pieces in the same pass (launch_layernorm2_split below)
                prof.begin("layernorm", 1, stream);
                if (int rc = launch_layernorm_split(x, L.ln_ffn1_g, L.ln_ffn1_b, nullptr, nullptr, t1s, M, Mpad, kHid, sc.scheme, sc.act_scale(WS_LN_FFN1), sc.site(WS_LN_FFN1), stream)) return rc;
                prof.end(stream);
            }
            prof.begin("ffn", 2, stream);
            if (int rc = gemm_split(sc, t1s, L, W_1A, L.b1a, kFfn, kHid, M, Mpad, XB_EPI_SWISH_SPLIT, 1.f, nullptr, nullptr, kFfn, bigs, stream)) return rc;
            if (int rc = gemm_split(sc, bigs, L, W_1B, L.b1b, kHid, kFfn, M, Mpad, XB_EPI_LINEAR, 0.5f, x, x, kHid, nullptr, stream)) return rc;
            prof.end(stream);

            prof.begin("layernorm", 1, stream);
            if (int rc = launch_layernorm_split(x, L.ln_att_g, L.ln_att_b, nullptr, nullptr, t1s, M, Mpad, kHid, sc.scheme, sc.act_scale(WS_LN_ATTN), sc.site(WS_LN_ATTN), stream)) return rc;
            prof.end(stream);
            prof.begin("attn_proj", 1, stream);
            // f16x2: the projection's epilogue writes k and v directly as fp16 pieces (q stays fp32 for the rel-pos table); the attention kernel
            // then stages K / V tiles without splitting them
            const bool kvp = attn_arith == ARITH_F16X2 && sc.scheme == XB_SCHEME_F16X2 && attn_kvp;
            if (kvp) {
                Bf16x3Args qa;
                qa.A = t1s; qa.W = L.ws[sc.scheme][W_QKV]; qa.bias = L.bqkv; qa.M = (int)M; qa.N = 3 * kHid; qa.K = kHid; qa.Mpad = (int)Mpad;
                qa.epi = XB_EPI_QKV; qa.C = big; qa.ldc = 3 * kHid; qa.S = kvs; qa.Spad = (int)Mpad; qa.qkv_hid = kHid;
                qa.scheme = sc.scheme; qa.status = sc.site(WS_QKV_KV); qa.acc_scale = 1.0f / (sc.act_scale(WS_LN_ATTN) * L.wscale[W_QKV]); qa.split_scale = XB_F16_ACT_SCALE;   // k / v pieces: the attention kernel's fixed 16
                if (int rc = launch_gemm_bf16x3(qa, stream)) return rc;
            } else if (int rc = gemm_split(sc, t1s, L, W_QKV, L.bqkv, 3 * kHid, kHid, M, Mpad, XB_EPI_LINEAR, 1.f, big, nullptr, 3 * kHid, nullptr, stream)) {
                return rc;
            }
            prof.end(stream);
            prof.begin("attention", 1, stream);
            if (attn_arith > 0) {
                if (int rc = launch_relpos_attention(big, amask, L.dist, nullptr, B, T, stream, 16, attn_arith, sc.site(WS_ATTENTION), t1s, Mpad, kvp ? kvs : nullptr, h->attn_w8, L.dist_s, L.dist_scale)) return rc;
            } else {
                if (int rc = launch_relpos_attention(big, amask, L.dist, t1, B, T, stream, 16, 0, nullptr)) return rc;
                if (int rc = launch_split_blocked(t1, kHid, M, Mpad, kHid, t1s, stream, sc.scheme, sc.act_scale(WS_ATTENTION), sc.site(WS_ATTENTION))) return rc;
            }
            prof.end(stream);
            prof.begin("attn_proj", 1, stream);
            if (int rc = gemm_split(sc, t1s, L, W_O, L.bo, kHid, kHid, M, Mpad, XB_EPI_LINEAR, 1.f, x, x, kHid, nullptr, stream)) return rc;
            prof.end(stream);

            prof.begin("layernorm", 1, stream);
            if (int rc = launch_layernorm_split(x, L.ln_conv_g, L.ln_conv_b, amask, nullptr, t1s, M, Mpad, kHid, sc.scheme, sc.act_scale(WS_LN_CONV), sc.site(WS_LN_CONV), stream)) return rc;
            prof.end(stream);
            prof.begin("conv_module", 3, stream);
            if (int rc = gemm_split(sc, t1s, L, W_PW1, nullptr, 2 * kHid, kHid, M, Mpad, XB_EPI_GLU, 1.f, big, nullptr, kHid, nullptr, stream)) return rc;
            if (int rc = (h->dwconv_stream ? launch_dwconv_stream : launch_dwconv_ln_swish)(big, L.dw, L.ln_dw_g, L.ln_dw_b, nullptr, B, T, stream, t1s, Mpad, sc.scheme, sc.act_scale(WS_DWCONV), sc.site(WS_DWCONV))) return rc;
            if (int rc = gemm_split(sc, t1s, L, W_PW2, nullptr, kHid, kHid, M, Mpad, XB_EPI_LINEAR, 1.f, x, x, kHid, nullptr, stream)) return rc;
            prof.end(stream);

            prof.begin("layernorm", 1, stream);
            if (int rc = launch_layernorm_split(x, L.ln_ffn2_g, L.ln_ffn2_b, nullptr, nullptr, t1s, M, Mpad, kHid, sc.scheme, sc.act_scale(WS_LN_FFN2), sc.site(WS_LN_FFN2), stream)) return rc;
            prof.end(stream);
            prof.begin("ffn", 2, stream);
            if (int rc = gemm_split(sc, t1s, L, W_2A, L.b2a, kFfn, kHid, M, Mpad, XB_EPI_SWISH_SPLIT, 1.f, nullptr, nullptr, kFfn, bigs, stream)) return rc;
            if (int rc = gemm_split(sc, bigs, L, W_2B, L.b2b, kHid, kFfn, M, Mpad, XB_EPI_LINEAR, 0.5f, x, x, kHid, nullptr, stream)) return rc;
            prof.end(stream);
            prof.begin("layernorm", 1, stream);
            if (li + 1 < n_layers) {
                // final_layer_norm of this layer and ffn1_layer_norm of the next in ONE pass over the residual stream: x = LN(x) as fp32 rows and
                // t1s = split(LN'(x)) (bit-identical to the two launches; saves one read of x per layer)
                const LayerW& Ln = h->layers[li + 1];
                const SplitCtx scn = ctx_of(li + 1);   // the pieces are the NEXT layer's operand: its scheme, its range row
                if (int rc = launch_layernorm2_split(x, L.ln_fin_g, L.ln_fin_b, x, Ln.ln_ffn1_g, Ln.ln_ffn1_b, t1s, M, Mpad, kHid, scn.scheme, scn.act_scale(WS_LN_FFN1), scn.site(WS_LN_FFN1), stream)) return rc;
            } else if (int rc = launch_layernorm(x, L.ln_fin_g, L.ln_fin_b, nullptr, x, M, kHid, stream)) {
                return rc;
            }
            prof.end(stream);
            continue;
        }
        // ---- fp32 MFMA path --------------------------------------------------------------------------------------------------
        prof.begin("ffn", 3, stream);
        if (int rc = launch_layernorm(x, L.ln_ffn1_g, L.ln_ffn1_b, nullptr, t1, M, kHid, stream)) return rc;
        if (int rc = linear(t1, kHid, L.w1a, L.b1a, big, kFfn, M, EPI_SWISH, 1.f, nullptr, nullptr, kFfn, stream)) return rc;
        if (int rc = linear(big, kFfn, L.w1b, L.b1b, x, kHid, M, EPI_NONE, 0.5f, x, nullptr, kHid, stream)) return rc;
        prof.end(stream);
        prof.begin("attn_proj", 3, stream);
        if (int rc = launch_layernorm(x, L.ln_att_g, L.ln_att_b, nullptr, t1, M, kHid, stream)) return rc;
        if (int rc = linear(t1, kHid, L.wqkv, L.bqkv, big, 3 * kHid, M, EPI_NONE, 1.f, nullptr, nullptr, 3 * kHid, stream)) return rc;
        prof.end(stream);
        prof.begin("attention", 1, stream);
        if (int rc = launch_relpos_attention(big, amask, L.dist, t1, B, T, stream, 16, h->arith, nullptr)) return rc;
        prof.end(stream);
        prof.begin("attn_proj", 0, stream);
        if (int rc = linear(t1, kHid, L.wo, L.bo, x, kHid, M, EPI_NONE, 1.f, x, nullptr, kHid, stream)) return rc;
        prof.end(stream);
        prof.begin("conv_module", 4, stream);
        if (int rc = launch_layernorm(x, L.ln_conv_g, L.ln_conv_b, amask, t1, M, kHid, stream)) return rc;
        if (int rc = linear(t1, kHid, L.pw1, nullptr, big, 2 * kHid, M, EPI_GLU, 1.f, nullptr, nullptr, kHid, stream)) return rc;
        if (int rc = (h->dwconv_stream ? launch_dwconv_stream : launch_dwconv_ln_swish)(big, L.dw, L.ln_dw_g, L.ln_dw_b, t1, B, T, stream, nullptr, 0, 0, 1.0f, nullptr)) return rc;
        if (int rc = linear(t1, kHid, L.pw2, nullptr, x, kHid, M, EPI_NONE, 1.f, x, nullptr, kHid, stream)) return rc;
        prof.end(stream);
        prof.begin("ffn", 4, stream);
        if (int rc = launch_layernorm(x, L.ln_ffn2_g, L.ln_ffn2_b, nullptr, t1, M, kHid, stream)) return rc;
        if (int rc = linear(t1, kHid, L.w2a, L.b2a, big, kFfn, M, EPI_SWISH, 1.f, nullptr, nullptr, kFfn, stream)) return rc;
        if (int rc = linear(big, kFfn, L.w2b, L.b2b, x, kHid, M, EPI_NONE, 0.5f, x, nullptr, kHid, stream)) return rc;
        if (int rc = launch_layernorm(x, L.ln_fin_g, L.ln_fin_b, nullptr, x, M, kHid, stream)) return rc;
        prof.end(stream);
    }
    if (status_dev)   // every site's range verdict of this call -> the caller's status word
        if (int rc = launch_range_combine(h->range_tab, n_layers * (int)W_NSITES, reinterpret_cast<int*>(status_dev), stream)) return rc;
    if (hidden_out) AT_CHECK_HIP(hipMemcpyAsync(hidden_out, x, (size_t)M * kHid * sizeof(float), hipMemcpyDeviceToDevice, stream));

    if (tokens) {
        // non-affine LayerNorm (reference encoder.py:138-143,176) then nearest code (encoder.py:180-181)
        prof.begin("vq", 3, stream);
        if (split && h->vq_split && h->cb_s[sc_model.scheme]) {
            // the score GEMM on the split kernel: the LayerNorm writes its fp32 rows (|x|^2 of the distance) and the operand pieces in one pass. Non-affine
            // LayerNorm output is bounded by sqrt(1024) = 32: x 16 cannot leave the fp16 range, so this site has no range word.
            if (int rc = launch_layernorm_split(x, nullptr, nullptr, nullptr, t1, t1s, M, Mpad, kHid, sc_model.scheme, sc_model.act_scale(), nullptr, stream)) return rc;
            Bf16x3Args va;
            va.A = t1s; va.W = h->cb_s[sc_model.scheme]; va.bias = nullptr; va.M = (int)M; va.N = kCodes; va.K = kHid; va.Mpad = (int)Mpad;
            va.epi = XB_EPI_LINEAR; va.C = big; va.ldc = kCodes; va.R = nullptr; va.ldr = kCodes; va.alpha = 1.f;
            va.scheme = sc_model.scheme; va.status = nullptr;
            if (sc_model.scheme == XB_SCHEME_F16X2) { va.acc_scale = 1.0f / (XB_F16_ACT_SCALE * h->cb_scale); va.split_scale = XB_F16_ACT_SCALE; }
            if (int rc = launch_gemm_bf16x3(va, stream)) return rc;
        } else {
            if (int rc = launch_layernorm(x, nullptr, nullptr, nullptr, t1, M, kHid, stream)) return rc;
            if (int rc = linear(t1, kHid, h->codebook, nullptr, big, kCodes, M, EPI_NONE, 1.f, nullptr, nullptr, kCodes, stream)) return rc;
        }
        if (int rc = launch_vq_argmax(t1, big, h->e2, tokens, M, kHid, kCodes, stream, reinterpret_cast<int*>(status_dev), 0, h->vq_refine ? h->codebook : nullptr)) return rc;
        prof.end(stream);
    }
    return 0;
}

// The measured fp16 headroom of the LAST encode of this handle: per site (at_w2vbert_range_sites) the largest |x * scale| a split writer saw over all
// layers (0: the site did not run on the fp16 scheme); the scheme overflows at 65504. Synchronises the device.
int at_w2vbert_range_report(at_w2vbert_t* h, float* max_scaled, int cap) {
    AT_REQUIRE(h && h->finalized && h->range_tab && max_scaled && cap >= (int)W_NSITES, "at_w2vbert_range_report: bad arguments");
    DeviceGuard guard(h->device);
    AT_REQUIRE(guard.ok, "cannot select the handle's device");
    std::vector<int> host(kRangeInts);
    AT_CHECK_HIP(hipDeviceSynchronize());
    AT_CHECK_HIP(hipMemcpy(host.data(), h->range_tab, kRangeInts * sizeof(int), hipMemcpyDeviceToHost));
    for (int k = 0; k < (int)W_NSITES; ++k) {
        float f;   // row 0 holds the one census word of the site; the other rows' second words are links to it (split_scheme.h)
        std::memcpy(&f, &host[k * 2 + 1], sizeof(f));
        max_scaled[k] = f;
    }
    return (int)W_NSITES;
}
// Per conformer layer, the OR of its sites' status flags in the LAST encode (bit 1 = an activation of that layer left the fp16 range; an overflow turns
// into infinities that later layers flag too: the FIRST flagged layer is the cause). Returns the number of layers written. Synchronises the device.
int at_w2vbert_layer_status(at_w2vbert_t* h, int32_t* flags, int cap) {
    AT_REQUIRE(h && h->finalized && h->range_tab && flags && cap >= 1, "at_w2vbert_layer_status: bad arguments");
    DeviceGuard guard(h->device);
    AT_REQUIRE(guard.ok, "cannot select the handle's device");
    std::vector<int> host(kRangeInts);
    AT_CHECK_HIP(hipDeviceSynchronize());
    AT_CHECK_HIP(hipMemcpy(host.data(), h->range_tab, kRangeInts * sizeof(int), hipMemcpyDeviceToHost));
    const int n = std::min<int>({cap, (int)h->layers.size(), kRangeLayers});
    for (int l = 0; l < n; ++l) {
        int v = 0;
        for (int k = 0; k < (int)W_NSITES; ++k) v |= host[(l * (int)W_NSITES + k) * 2];
        flags[l] = v;
    }
    return n;
}
// The activation scale of every (layer, site): scales[l * n_sites + k], n_sites = at_w2vbert_range_sites. 16 everywhere unless a LayerNorm's gains force
// the provable scale of a LayerNorm-fed site below that (xb_ln_site_scale). Returns the number of floats written. Host-only.
int at_w2vbert_site_scales(const at_w2vbert_t* h, float* scales, int cap) {
    AT_REQUIRE(h && h->finalized && scales, "at_w2vbert_site_scales: bad arguments");
    const int n = (int)h->layers.size() * (int)W_NSITES;
    AT_REQUIRE(cap >= n, "at_w2vbert_site_scales: buffer too small");
    for (size_t l = 0; l < h->layers.size(); ++l)
        for (int k = 0; k < (int)W_NSITES; ++k) scales[l * W_NSITES + k] = h->layers[l].site_scale[k];
    return n;
}
int at_w2vbert_range_sites(char* names, size_t cap) {
    std::string s;
    for (int k = 0; k < (int)W_NSITES; ++k) { s += kWSiteNames[k]; s += "\n"; }
    if (!names || cap < s.size() + 1) return -(int)(s.size() + 1);
    std::memcpy(names, s.c_str(), s.size() + 1);
    return (int)W_NSITES;
}

/* ---- operator-level entry points for the parity tests ---------------------------------------------------- */
int at_op_layernorm(const float* x, const float* gamma, const float* beta, const float* row_mask, float* y, int64_t rows, int D,
                    at_stream_t stream) {
    AT_REQUIRE(x && y, "null pointer");
    return launch_layernorm(x, gamma, beta, row_mask, y, rows, D, (hipStream_t)stream);
}

int at_op_relpos_attention(const float* qkv, const float* attn_mask, const float* dist_emb80, float* ctx, int B, int T,
                           at_stream_t stream) {
    AT_REQUIRE(qkv && attn_mask && dist_emb80 && ctx && B >= 1 && T >= 1, "bad arguments");
    return launch_relpos_attention(qkv, attn_mask, dist_emb80, ctx, B, T, (hipStream_t)stream);
}

int at_op_relpos_attention_kvp(const float* qkv, const float* attn_mask, const float* dist_emb80, float dist_max_abs, float* ctx, int B, int T, int heads, int w8,
                               void* kv_workspace, size_t kv_workspace_bytes, int32_t* status_dev, at_stream_t stream) {
    AT_REQUIRE(qkv && attn_mask && ctx && kv_workspace && B >= 1 && T >= 1 && heads >= 1 && heads <= 64, "bad arguments");
    const long long rows = (long long)B * T, rows_pad = (rows + 255) / 256 * 256;
    const int hid = heads * 64;
    const size_t kv_bytes = (size_t)4 * rows_pad * hid * 2, dist_bytes = (size_t)2 * 96 * 64 * 2;
    AT_REQUIRE(kv_workspace_bytes >= kv_bytes + dist_bytes, "kv workspace too small: 4 * ceil256(B * T) * heads * 64 * 2 + 24576 bytes");
    if (int rc = launch_kv_rowmajor_split(qkv, static_cast<__bf16*>(kv_workspace), rows, rows_pad, hid, status_dev, (hipStream_t)stream)) return rc;
    __bf16* dist_s = nullptr;
    float dist_scale = 1.0f;
    if (dist_emb80) {   // what finalize() does once per layer: the distance embeddings as fp16 pieces times a power of two
        dist_s = reinterpret_cast<__bf16*>(static_cast<char*>(kv_workspace) + kv_bytes);
        dist_scale = xb_weight_scale(dist_max_abs);
        if (int rc = launch_dist_split(dist_emb80, dist_s, dist_scale, (hipStream_t)stream)) return rc;
    }
    return launch_relpos_attention(qkv, attn_mask, dist_emb80, ctx, B, T, (hipStream_t)stream, heads, 2, status_dev, nullptr, rows_pad,
                                   static_cast<const __bf16*>(kv_workspace), w8, dist_s, dist_scale);
}

int at_op_dwconv_ln_swish(const float* g, const float* w31x1024, const float* gamma, const float* beta, float* out, int B, int T,
                          at_stream_t stream) {
    AT_REQUIRE(g && w31x1024 && gamma && beta && out && B >= 1 && T >= 1, "bad arguments");
    return launch_dwconv_ln_swish(g, w31x1024, gamma, beta, out, B, T, (hipStream_t)stream);
}

int at_op_dwconv_stream(const float* g, const float* w31x1024, const float* gamma, const float* beta, float* out, int B, int T,
                        at_stream_t stream) {
    AT_REQUIRE(g && w31x1024 && gamma && beta && out && B >= 1 && T >= 1, "bad arguments");
    return launch_dwconv_stream(g, w31x1024, gamma, beta, out, B, T, (hipStream_t)stream);
}

int at_op_vq_argmax(const float* x, const float* dots, const float* e2, int16_t* out, int64_t rows, int D, int C, at_stream_t stream) {
    AT_REQUIRE(x && dots && e2 && out && D % 4 == 0 && C % 4 == 0, "bad arguments");
    return launch_vq_argmax(x, dots, e2, out, rows, D, C, (hipStream_t)stream);
}

int at_op_vq_argmax_refined(const float* x, const float* dots, const float* e2, const float* codebook, int16_t* out, int64_t rows, int D, int C, at_stream_t stream) {
    AT_REQUIRE(x && dots && e2 && codebook && out && D % 4 == 0 && C % 4 == 0, "bad arguments");
    return launch_vq_argmax(x, dots, e2, out, rows, D, C, (hipStream_t)stream, nullptr, 0, codebook);
}

}  // extern "C"
