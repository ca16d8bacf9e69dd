// semantic_s tokenizer (mHuBERT-base -> LayerNorm -> k-means assignment) — host side. C ABI in
// include/audiotoken_hip.h. Replaces reference HubertEncoder (audiotoken/encoder.py:60-108): HF HubertModel
// hidden_states[output_layer = 11], non-affine LayerNorm(768), torch.cdist to 1000 centres + argmin.
// Arithmetic per SURVEY.md Appendix A.4 (HF modeling_hubert.py). fp32 on the f32 matrix cores throughout.
//
// Weight repacking at finalize():
//   feature-extractor convs [512][Cin][k] -> [512][k*Cin] (tap-major, channels-last windows); conv 0 keeps [512][10]
//   positional conv (weight-norm folded by the caller) [768][48][128], groups 16 -> 16 x [48][128*48]
//   q,k,v Linear -> one [2304][768]
#include <map>
#include <string>
#include <vector>
#include <cstring>

#include "../../include/audiotoken_hip.h"
#include "at_common.h"
#include "hubert_kernels.h"
#include "gemm_bf16x3.h"
#include "packed_model.h"
#include <cstdlib>
#include "w2vbert_kernels.h"

using namespace at;

namespace {
constexpr int kCd = 512, kHid = 768, kFfn = 3072, kHeads = 12, kPosK = 128, kGroups = 16, kGc = 48, kCenters = 1000, kCentersPad = 1024;
constexpr int kKs[7] = {10, 3, 3, 3, 3, 2, 2}, kSt[7] = {5, 2, 2, 2, 2, 2, 2};

struct HostTensor {
    std::vector<int64_t> shape;
    std::vector<float> data;
};
struct LayerW {
    const float *wqkv, *bqkv, *wo, *bo, *ln1_g, *ln1_b, *w1, *b1, *w2, *b2, *ln2_g, *ln2_b;
    // the four linear layers as 16-bit operand pieces per scheme [XB_SCHEME_*][HW_*] (gemm_bf16x3.h); wscale: the fp16 scheme's weight scales
    const piece_t* ws[2][4] = {};
    float wscale[4] = {1.f, 1.f, 1.f, 1.f};
    // f16x2: the activation scales of the two LayerNorm-fed split sites of this layer — the input of the q/k/v projection (written by the previous
    // layer's final_layer_norm, or encoder.layer_norm for layer 0) and the input of the first FFN GEMM (this layer's layer_norm): 16 unless the
    // LayerNorm's gains force the provable scale below that (xb_ln_site_scale, gemm_bf16x3.h)
    float xs_qkv = XB_F16_ACT_SCALE, xs_ffn = XB_F16_ACT_SCALE;
};
enum { HW_QKV = 0, HW_O, HW_1, HW_2 };
enum { ARITH_F32 = 0, ARITH_BF16X3 = 1, ARITH_F16X2 = 2 };
}  // namespace

struct at_hubert {
    int device = 0;
    bool finalized = false;
    std::map<std::string, HostTensor> staged;
    DeviceArena arena;          // every device allocation of finalize(), in order (packed_model.h: export / import of the finalized model)
    PackedHeader imp{};         // import_packed: the exporter's record (layer count, flags) while finalize is replayed
    std::vector<int> split_seq; // the schemes whose weight pieces exist, in the order they were split (= their order in the arena)
    int* range_tab = nullptr;   // device, {flag, census} per (row, HSite): row 0 = conv feature encoder + positional conv, row 1 + l = transformer layer l; zeroed per encode
    std::vector<int> layer_arith;   // per transformer layer: -1 = the handle's arithmetic, else ARITH_BF16X3 / ARITH_F16X2 for that layer only (option "layer_arith:<i>")
    const float* conv_w[7] = {};
    const float *gn_g = nullptr, *gn_b = nullptr, *fp_ln_g = nullptr, *fp_ln_b = nullptr, *fp_w = nullptr, *fp_b = nullptr;
    const float *pos_w = nullptr, *pos_b = nullptr, *enc_ln_g = nullptr, *enc_ln_b = nullptr;
    std::vector<LayerW> layers;
    const float *centers = nullptr, *c2 = nullptr;
    const piece_t* conv_ws[2][7] = {};   // conv weights of layers 1..6 as operand pieces, per scheme
    const piece_t* pos_ws = nullptr;     // positional-conv weights as fp16 pieces in the per-K-step layout of hubert_posconv.hip (f16x2 scheme only)
    float pos_wscale = 1.f;
    const piece_t* cen_s[2] = {};        // the k-means centres as operand pieces per scheme, rows padded 1000 -> 1024 (zero rows): the score GEMM on the split kernel
    float cen_scale = 1.f;
    bool kmeans_split = true;            // option "kmeans_split": that GEMM on the split kernel instead of the fp32 MFMA (as at_w2vbert's "vq_split")
    bool vq_refine = true;               // option "vq_refine" (round 5): near-tie centres re-evaluated exactly (vq_argmax_kernel)
    bool ln_split = true;                // option "ln_split" (round 5): the post-LN LayerNorms write the fp32 residual stream AND the next GEMM's operand pieces in one pass (launch_layernorm_split, D = 768) instead of LayerNorm + a separate split pass; bit-identical
    bool posconv_split = true;           // option "posconv_split": the LDS-resident grouped conv kernel (hubert_posconv.hip) instead of 16 fp32 windowed GEMMs
    float conv_wscale[7] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
    int arith = ARITH_F16X2;   // linear layers + conv chain: ARITH_* ($AUDIOTOKEN_SEMANTIC_ARITH = f32 | bf16x3 | f16x2; option "arith")
    bool split_done[2] = {false, false};
    int attn_w8 = -1;          // option "attn_w8" (as at_w2vbert): the round-4 attention kernel (attention_f16x2_w8.hip) or its round-3 twin
    std::map<const float*, float> wmax;   // max |w| of every uploaded tensor
    Profiler prof;
};

namespace {

const HostTensor* find(const at_hubert* h, const std::string& name) {
    auto it = h->staged.find(name);
    return it == h->staged.end() ? nullptr : &it->second;
}
const float* upload(at_hubert* h, const std::vector<float>& v) {
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
const float* reserve(at_hubert* h, size_t n_floats) {
    float* d = static_cast<float*>(h->arena.alloc((n_floats + 3) / 4 * 4 * sizeof(float)));
    if (d) h->wmax[d] = h->arena.blocks.back().wmax;
    return d;
}
const float* take(at_hubert* h, const std::string& name, std::vector<int64_t> shape, bool& ok) {
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

struct Plan {
    int L[8];   // L[0] = N, L[i+1] = frames after conv i
    size_t off_a, off_b, off_part, off_ss, off_fmask, off_x, off_t1, off_big, off_pos, off_xs, off_bigs, off_kvs;
    size_t Mpad;
    int Lp[8];              // rows per phase plane of the split-bf16 input of conv i (i = 1..6)
    int Mp[8];              // padded output rows per clip of conv i
    size_t off_sa, off_sb;  // split-bf16 ping / pong buffers of the conv chain
    size_t total_floats;
};
Plan make_plan(int B, int N) {
    Plan p;
    p.L[0] = N;
    for (int i = 0; i < 7; ++i) p.L[i + 1] = p.L[i] >= kKs[i] ? (p.L[i] - kKs[i]) / kSt[i] + 1 : 0;
    size_t cur = 0;
    auto takef = [&](size_t n) { size_t o = cur; cur += (n + 63) / 64 * 64; return o; };
    const size_t T = p.L[7], M = (size_t)B * T;
    p.off_a = takef((size_t)B * p.L[1] * kCd);     // ping: conv 0, 2, 4, 6 outputs
    p.off_b = takef((size_t)B * p.L[2] * kCd);     // pong: conv 1, 3, 5 outputs
    p.off_part = takef((size_t)B * hub_ws_nchunk(p.L[1] > 0 ? p.L[1] : 1) * 65 * 2);   // doubles
    p.off_ss = takef((size_t)B * kCd * 2);
    p.off_fmask = takef(M);
    p.off_x = takef(M * kHid);
    p.off_t1 = takef(M * kHid);
    p.off_pos = takef(M * kHid);
    p.off_big = takef(M * kFfn);
    p.Mpad = (M + 255) / 256 * 256;                 // split-bf16 operands: 3 pieces x 2 bytes = 1.5 floats per element
    {
        size_t need[2] = {0, 0};
        for (int i = 1; i < 7; ++i) {
            p.Mp[i] = (p.L[i + 1] + 255) / 256 * 256;
            // phase-major time axis: kSt planes of Lp rows each; the last tile of tap j reads rows up to Mp + (k - 1) / stride
            const int reach = p.Mp[i] + (kKs[i] - 1) / kSt[i], have = (p.L[i] + kSt[i] - 1) / kSt[i];
            p.Lp[i] = ((have > reach ? have : reach) + 63) / 64 * 64;
            const size_t fl = (size_t)B * kSt[i] * p.Lp[i] * kCd * 3 / 2 + 64;   // 3 pieces x 2 bytes per element, in floats
            if (fl > need[i & 1]) need[i & 1] = fl;
        }
        p.off_sa = takef(need[1]);   // inputs of conv 1, 3, 5
        p.off_sb = takef(need[0]);   // inputs of conv 2, 4, 6
    }
    p.off_xs = takef(p.Mpad * kHid * 3 / 2);
    p.off_bigs = takef(p.Mpad * kFfn * 3 / 2);
    p.off_kvs = takef(p.Mpad * kHid * 2);           // k and v as two fp16 pieces each (XB_EPI_QKV)
    p.total_floats = cur;
    return p;
}

// Sites of the handle's range table (gemm_bf16x3.h, launch_range_combine): where activations become fp16 pieces
enum HSite { HS_CONV0 = 0, HS_FE_CONV, HS_X_IN, HS_QKV_KV, HS_ATTENTION, HS_FFN_HIDDEN, HS_OTHER, H_NSITES };
static const char* const kHSiteNames[H_NSITES] = {"conv0_out", "feature_convs", "layer_input", "qkv_kv", "attention", "ffn_hidden", "other"};
constexpr int kRangeRows = 33;                               // rows of the range table: the front end + up to 32 transformer layers
constexpr int kRangeInts = kRangeRows * 2 * (int)H_NSITES;
struct SplitCtx {
    int scheme; int* tab;
    int* site(int k) const { return tab ? tab + 2 * k : nullptr; }
    float act_scale() const { return scheme == XB_SCHEME_F16X2 ? XB_F16_ACT_SCALE : 1.0f; }
};

// Split the conv chain's and the transformer's weights into the 16-bit pieces of `scheme` (once per scheme)
int split_weights(at_hubert* h, int scheme) {
    if (h->split_done[scheme]) return 0;
    const int np = xb_pieces(scheme);
    auto one = [&](const float* src, int n, int k, const piece_t** dst, float* scale_out, int win_cblocks = 0, int win_stride = 1) -> int {
        piece_t* d = static_cast<piece_t*>(h->arena.alloc((size_t)np * n * k * sizeof(piece_t)));
        if (!d) return -1;
        float sc = 1.0f;
        if (scheme == XB_SCHEME_F16X2) {
            auto it = h->wmax.find(src);
            AT_REQUIRE(it != h->wmax.end(), "weight maximum not recorded");
            sc = xb_weight_scale(it->second);
            *scale_out = sc;
        }
        if (!h->arena.importing)   // import_packed: the pieces are in the blob
            if (int rc = launch_split_blocked(src, k, n, n, k, d, nullptr, scheme, sc, nullptr, win_cblocks, win_stride)) return rc;
        *dst = d;
        return 0;
    };
    for (int i = 1; i < 7; ++i)
        if (int rc = one(h->conv_w[i], kCd, kKs[i] * kCd, &h->conv_ws[scheme][i], &h->conv_wscale[i], kCd / 16, kSt[i])) return rc;   // window order
    if (scheme == XB_SCHEME_F16X2) {   // the positional conv's weights in hubert_posconv.hip's layout ([group][K step][piece][k-block][48][16])
        piece_t* d = static_cast<piece_t*>(h->arena.alloc(posconv_weight_pieces_bytes()));
        if (!d) return -1;
        auto it = h->wmax.find(h->pos_w);
        AT_REQUIRE(it != h->wmax.end(), "weight maximum not recorded");
        h->pos_wscale = xb_weight_scale(it->second);
        if (!h->arena.importing)
            if (int rc = launch_posconv_weight_split(h->pos_w, d, h->pos_wscale, nullptr)) return rc;
        h->pos_ws = d;
    }
    for (LayerW& L : h->layers) {
        const float* src[4] = {L.wqkv, L.wo, L.w1, L.w2};
        const int ns[4] = {3 * kHid, kHid, kFfn, kHid}, ks[4] = {kHid, kHid, kHid, kFfn};
        for (int j = 0; j < 4; ++j)
            if (int rc = one(src[j], ns[j], ks[j], &L.ws[scheme][j], &L.wscale[j])) return rc;
    }
    if (h->centers) {   // k-means centres [1000][768] -> pieces of 1024 rows (the last 24 zero: their scores are never read)
        piece_t* d = static_cast<piece_t*>(h->arena.alloc((size_t)np * kCentersPad * kHid * sizeof(piece_t)));
        if (!d) return -1;
        float sc = 1.0f;
        if (scheme == XB_SCHEME_F16X2) {
            auto it = h->wmax.find(h->centers);
            AT_REQUIRE(it != h->wmax.end(), "weight maximum not recorded");
            sc = xb_weight_scale(it->second);
            h->cen_scale = sc;
        }
        if (!h->arena.importing)
            if (int rc = launch_split_blocked(h->centers, kHid, kCenters, kCentersPad, kHid, d, nullptr, scheme, sc, nullptr)) return rc;
        h->cen_s[scheme] = d;
    }
    AT_CHECK_HIP(hipDeviceSynchronize());
    h->split_done[scheme] = true;
    h->split_seq.push_back(scheme);
    return 0;
}

// C = epi(X . W^T) through the split GEMM: X fp32 row-major [M][K] is split into xs first (unless it already is: X == nullptr)
// x_scale: the f16x2 scale of the A operand (the pieces in xs, or what X is split with here); 16 except at the LayerNorm-fed sites
int linear_split(const SplitCtx& c, const float* X, int K, const piece_t* xs, piece_t* xs_w, const LayerW& L, int w, const float* bias, float* C, int N,
                 long long M, long long Mpad, int epi, const float* R, int ldc, piece_t* S, hipStream_t stream, float x_scale = XB_F16_ACT_SCALE) {
    if (c.scheme != XB_SCHEME_F16X2) x_scale = 1.0f;
    if (X) {
        if (int rc = launch_split_blocked(X, K, M, Mpad, K, xs_w, stream, c.scheme, x_scale, c.site(HS_X_IN))) return rc;
        xs = xs_w;
    }
    Bf16x3Args a;
    a.A = xs; a.W = L.ws[c.scheme][w]; a.bias = bias; a.M = (int)M; a.N = N; a.K = K; a.Mpad = (int)Mpad;
    a.epi = epi; a.C = C; a.ldc = ldc; a.R = R; a.ldr = ldc; a.alpha = 1.0f; a.S = S; a.Spad = (int)Mpad;
    a.scheme = c.scheme; a.status = c.site(w == HW_1 ? HS_FFN_HIDDEN : HS_OTHER);
    if (c.scheme == XB_SCHEME_F16X2) { a.acc_scale = 1.0f / (x_scale * L.wscale[w]); a.split_scale = XB_F16_ACT_SCALE; }
    return launch_gemm_bf16x3(a, stream);
}

int linear(const float* X, int K, const float* W, const float* bias, float* C, int N, long long M, int epi, const float* R,
           const float* row_mask, int ldc, hipStream_t stream) {
    GemmArgs a;
    a.X = X; a.Tin = (int)M; a.Cin = K; a.ldx = K;
    a.W = W; a.bias = bias; a.C = C; a.ldc = ldc; a.R = R; a.ldr = ldc;
    a.M = (int)M; a.N = N; a.K = K; a.batch = 1; a.epi = epi; a.row_mask = row_mask;
    return launch_gemm(a, stream);
}

}  // namespace

extern "C" {

at_hubert_t* at_hubert_create(int device_id) {
    int n = 0;
    if (!host_only_test() && (hipGetDeviceCount(&n) != hipSuccess || device_id < 0 || device_id >= n)) {
        set_error("at_hubert_create: no such HIP device " + std::to_string(device_id));
        return nullptr;
    }
    at_hubert* h = new at_hubert();
    h->device = device_id;
    return h;
}

int at_hubert_set_tensor(at_hubert_t* h, const char* name, const float* host_data, const int64_t* shape, int ndim) {
    AT_REQUIRE(h && name && host_data && shape && ndim >= 1 && ndim <= 4, "bad arguments");
    AT_REQUIRE(!h->finalized, "model already finalized");
    HostTensor t;
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) { t.shape.push_back(shape[i]); n *= (size_t)shape[i]; }
    t.data.assign(host_data, host_data + n);
    h->staged[name] = std::move(t);
    return 0;
}

// finalize(): staged host tensors -> device. With the arena in import mode (at_hubert_import_packed) the same code REPLAYS the allocation order over the
// packed blob: no host tensor is read, nothing is uploaded or split — only the pointers and scales are rebuilt.
static int finalize_impl(at_hubert* h) {
    const bool imp = h->arena.importing;
    bool ok = true;
    for (int i = 0; i < 7; ++i) {
        if (imp) {
            h->conv_w[i] = reserve(h, (size_t)kCd * kKs[i] * (i == 0 ? 1 : kCd));
            AT_REQUIRE(h->conv_w[i] != nullptr, "import_packed: feature-extractor weights");
            continue;
        }
        const std::string key = "feature_extractor.conv_layers." + std::to_string(i) + ".conv.weight";
        const HostTensor* t = find(h, key);
        const int cin = i == 0 ? 1 : kCd, k = kKs[i];
        AT_REQUIRE(t && t->shape == (std::vector<int64_t>{kCd, cin, k}), "missing / mis-shaped " + key);
        std::vector<float> w((size_t)kCd * k * cin);
        for (int co = 0; co < kCd; ++co)
            for (int ci = 0; ci < cin; ++ci)
                for (int tp = 0; tp < k; ++tp) w[((size_t)co * k + tp) * cin + ci] = t->data[((size_t)co * cin + ci) * k + tp];
        h->conv_w[i] = upload(h, w);
        AT_REQUIRE(h->conv_w[i] != nullptr, "device allocation failed");
    }
    h->gn_g = take(h, "feature_extractor.conv_layers.0.layer_norm.weight", {kCd}, ok);
    h->gn_b = take(h, "feature_extractor.conv_layers.0.layer_norm.bias", {kCd}, ok);
    h->fp_ln_g = take(h, "feature_projection.layer_norm.weight", {kCd}, ok);
    h->fp_ln_b = take(h, "feature_projection.layer_norm.bias", {kCd}, ok);
    h->fp_w = take(h, "feature_projection.projection.weight", {kHid, kCd}, ok);
    h->fp_b = take(h, "feature_projection.projection.bias", {kHid}, ok);
    h->pos_b = take(h, "encoder.pos_conv_embed.conv.bias", {kHid}, ok);
    h->enc_ln_g = take(h, "encoder.layer_norm.weight", {kHid}, ok);
    h->enc_ln_b = take(h, "encoder.layer_norm.bias", {kHid}, ok);
    if (!ok) return -1;
    if (imp) {
        h->pos_w = reserve(h, (size_t)kHid * kPosK * kGc);
        AT_REQUIRE(h->pos_w != nullptr, "import_packed: positional conv");
    } else {   // grouped positional conv: folded weight [768][48][128] -> per group [48 out][128 taps][48 in]
        const HostTensor* t = find(h, "encoder.pos_conv_embed.conv.weight");
        AT_REQUIRE(t && t->shape == (std::vector<int64_t>{kHid, kGc, kPosK}), "encoder.pos_conv_embed.conv.weight [768,48,128] (weight-norm folded) missing");
        std::vector<float> w((size_t)kHid * kPosK * kGc);
        for (int co = 0; co < kHid; ++co)
            for (int ci = 0; ci < kGc; ++ci)
                for (int tp = 0; tp < kPosK; ++tp)
                    w[((size_t)co * kPosK + tp) * kGc + ci] = t->data[((size_t)co * kGc + ci) * kPosK + tp];
        h->pos_w = upload(h, w);
        AT_REQUIRE(h->pos_w != nullptr, "device allocation failed");
    }
    int nl = imp ? h->imp.n_layers : 0;
    while (!imp && find(h, "encoder.layers." + std::to_string(nl) + ".layer_norm.weight")) ++nl;
    for (int i = 0; i < nl; ++i) {
        const std::string p = "encoder.layers." + std::to_string(i);
        LayerW L{};
        if (imp) {
            L.wqkv = reserve(h, (size_t)3 * kHid * kHid);
            L.bqkv = reserve(h, (size_t)3 * kHid);
        } else {
            std::vector<float> w((size_t)3 * kHid * kHid), b((size_t)3 * kHid);
            const char* nm[3] = {"q_proj", "k_proj", "v_proj"};
            for (int j = 0; j < 3; ++j) {
                const HostTensor* wt = find(h, p + ".attention." + nm[j] + ".weight");
                const HostTensor* bt = find(h, p + ".attention." + nm[j] + ".bias");
                AT_REQUIRE(wt && bt && wt->shape == (std::vector<int64_t>{kHid, kHid}) && bt->shape == (std::vector<int64_t>{kHid}),
                           "attention projection tensors missing or mis-shaped");
                std::memcpy(&w[(size_t)j * kHid * kHid], wt->data.data(), (size_t)kHid * kHid * sizeof(float));
                std::memcpy(&b[(size_t)j * kHid], bt->data.data(), kHid * sizeof(float));
            }
            L.wqkv = upload(h, w);
            L.bqkv = upload(h, b);
        }
        AT_REQUIRE(L.wqkv && L.bqkv, "device allocation failed");
        L.wo = take(h, p + ".attention.out_proj.weight", {kHid, kHid}, ok);
        L.bo = take(h, p + ".attention.out_proj.bias", {kHid}, ok);
        L.ln1_g = take(h, p + ".layer_norm.weight", {kHid}, ok);
        L.ln1_b = take(h, p + ".layer_norm.bias", {kHid}, ok);
        L.w1 = take(h, p + ".feed_forward.intermediate_dense.weight", {kFfn, kHid}, ok);
        L.b1 = take(h, p + ".feed_forward.intermediate_dense.bias", {kFfn}, ok);
        L.w2 = take(h, p + ".feed_forward.output_dense.weight", {kHid, kFfn}, ok);
        L.b2 = take(h, p + ".feed_forward.output_dense.bias", {kHid}, ok);
        L.ln2_g = take(h, p + ".final_layer_norm.weight", {kHid}, ok);
        L.ln2_b = take(h, p + ".final_layer_norm.bias", {kHid}, ok);
        if (!ok) return -1;
        {
            auto mx = [&](const float* d) { auto it = h->wmax.find(d); return it == h->wmax.end() ? 0.f : it->second; };
            const float* pg = h->layers.empty() ? h->enc_ln_g : h->layers.back().ln2_g;
            const float* pb = h->layers.empty() ? h->enc_ln_b : h->layers.back().ln2_b;
            L.xs_qkv = xb_ln_site_scale(mx(pg), mx(pb), kHid);
            L.xs_ffn = xb_ln_site_scale(mx(L.ln1_g), mx(L.ln1_b), kHid);
        }
        h->layers.push_back(L);
    }
    if (imp) {
        if (h->imp.flags & 1) {
            h->centers = reserve(h, (size_t)kCenters * kHid);
            h->c2 = reserve(h, kCenters);
            AT_REQUIRE(h->centers && h->c2, "import_packed: k-means centres");
        }
    } else if (const HostTensor* c = find(h, "kmeans.cluster_centers_")) {
        AT_REQUIRE(c->shape.size() == 2 && c->shape[1] == kHid && c->shape[0] % 4 == 0, "kmeans.cluster_centers_ must be [C,768], C % 4 == 0");
        h->centers = upload(h, c->data);
        const int C = (int)c->shape[0];
        AT_REQUIRE(C == kCenters, "this build is sized for 1000 centres");
        std::vector<float> c2(C);
        if (const HostTensor* e = find(h, "kmeans.c2")) {
            AT_REQUIRE(e->data.size() == (size_t)C, "bad kmeans.c2 shape");
            c2 = e->data;
        } else {
            for (int n2 = 0; n2 < C; ++n2) {
                float acc = 0.f;
                for (int k = 0; k < kHid; ++k) { const float v = c->data[(size_t)n2 * kHid + k]; acc += v * v; }
                c2[n2] = acc;
            }
        }
        h->c2 = upload(h, c2);
        AT_REQUIRE(h->centers && h->c2, "device allocation failed");
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
        for (int r = 1; r < kRangeRows; ++r)
            for (int k = 0; k < (int)H_NSITES; ++k) init[(r * (int)H_NSITES + k) * 2 + 1] = -(r * (int)H_NSITES * 2);
        AT_CHECK_HIP(hipMemcpy(h->range_tab + kRangeInts, init.data(), kRangeInts * sizeof(int), hipMemcpyHostToDevice));
        AT_CHECK_HIP(hipMemcpy(h->range_tab, init.data(), kRangeInts * sizeof(int), hipMemcpyHostToDevice));
    }
    h->finalized = true;
    return 0;
}

int at_hubert_finalize(at_hubert_t* h) {
    AT_REQUIRE(h && !h->finalized, "bad handle");
    DeviceGuard guard(h->device);
    AT_REQUIRE(guard.ok, "cannot select the handle's device");
    return finalize_impl(h);
}

// ---- the finalized model as one device blob (packed_model.h) ------------------------------------------------------------------------------
static int packed_flags(const at_hubert* h) {
    int f = (h->centers ? 1 : 0) | ((int)h->split_seq.size() << 1);
    for (size_t i = 0; i < h->split_seq.size(); ++i) f |= (h->split_seq[i] == XB_SCHEME_BF16X3 ? 1 : 0) << (3 + i);
    return f;
}
int64_t at_hubert_packed_bytes(at_hubert_t* h) {
    if (!h || !h->finalized) { set_error("at_hubert_packed_bytes: model not finalized"); return -1; }
    return (int64_t)h->arena.packed_bytes();
}
int64_t at_hubert_packed_meta(at_hubert_t* h, void* host_dst, int64_t cap) {
    if (!h || !h->finalized) { set_error("at_hubert_packed_meta: model not finalized"); return -1; }
    return packed_write_meta(h->arena, PACKED_MODEL_HUBERT, (int)h->layers.size(), packed_flags(h), h->arith, host_dst, cap);
}
int at_hubert_export_packed(at_hubert_t* h, void* device_dst, int64_t bytes, void* stream) {
    AT_REQUIRE(h && h->finalized, "at_hubert_export_packed: model not finalized");
    DeviceGuard guard(h->device);
    AT_REQUIRE(guard.ok, "cannot select the handle's device");
    return packed_export(h->arena, device_dst, bytes, (hipStream_t)stream);
}
int at_hubert_import_packed(at_hubert_t* h, const void* host_meta, int64_t meta_bytes, const void* device_src, int64_t bytes, void* stream) {
    AT_REQUIRE(h && !h->finalized && h->staged.empty(), "at_hubert_import_packed needs a fresh handle (no set_tensor, no finalize)");
    DeviceGuard guard(h->device);
    AT_REQUIRE(guard.ok, "cannot select the handle's device");
    if (int rc = packed_begin_import(h->arena, PACKED_MODEL_HUBERT, host_meta, meta_bytes, device_src, bytes, (hipStream_t)stream, &h->imp)) return rc;
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
        h->centers = h->c2 = nullptr;
        h->pos_ws = nullptr;
        for (int s = 0; s < 2; ++s)
            for (int j = 0; j < 7; ++j) h->conv_ws[s][j] = nullptr;
        h->arena.free_all();
        if (h->range_tab) { (void)hipFree(h->range_tab); h->range_tab = nullptr; }
    }
    return rc;
}

void at_hubert_destroy(at_hubert_t* h) {
    if (!h) return;
    DeviceGuard guard(h->device);   // restores the caller's current device
    h->arena.free_all();
    if (h->range_tab) (void)hipFree(h->range_tab);
    delete h;
}

int at_hubert_num_layers(const at_hubert_t* h) { return h ? (int)h->layers.size() : 0; }

int at_hubert_num_tokens(int N) {
    int L = N;
    for (int i = 0; i < 7; ++i) L = L >= kKs[i] ? (L - kKs[i]) / kSt[i] + 1 : 0;
    return L;
}

size_t at_hubert_workspace_bytes(const at_hubert_t* h, int B, int N) {
    (void)h;
    if (B <= 0 || at_hubert_num_tokens(N) <= 0) return 0;
    return make_plan(B, N).total_floats * sizeof(float);
}

int at_hubert_set_option(at_hubert_t* h, const char* name, int value) {
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
    if (n.rfind("layer_arith:", 0) == 0) {   // "layer_arith:<l>": -1 = follow "arith", 1 = bf16x3, 2 = f16x2 for transformer layer l only
        const int li = std::atoi(n.c_str() + 12);
        AT_REQUIRE(li >= 0 && li < (int)h->layers.size(), "layer_arith: no such layer");
        AT_REQUIRE(value == -1 || value == ARITH_BF16X3 || value == ARITH_F16X2, "layer_arith:<l>: -1 = the handle's arithmetic, 1 = bf16x3, 2 = f16x2");
        if (value > 0) {
            DeviceGuard guard(h->device);
            AT_REQUIRE(guard.ok, "cannot select the handle's device");
            if (int rc = split_weights(h, value == ARITH_F16X2 ? XB_SCHEME_F16X2 : XB_SCHEME_BF16X3)) return rc;
        }
        if (h->layer_arith.size() < h->layers.size()) h->layer_arith.resize(h->layers.size(), -1);
        h->layer_arith[li] = value;
        return 0;
    }
    if (n == "attn_w8") { h->attn_w8 = value < 0 ? -1 : (value != 0); return 0; }
    if (n == "posconv_split") { h->posconv_split = value != 0; return 0; }
    if (n == "ln_split") { h->ln_split = value != 0; return 0; }
    if (n == "vq_refine") { h->vq_refine = value != 0; return 0; }
    if (n == "kmeans_split") { h->kmeans_split = value != 0; return 0; }
    set_error("at_hubert_set_option: unknown option " + n);
    return -1;
}

int at_hubert_get_option(const at_hubert_t* h, const char* name) {
    if (!h || !name) return -1;
    if (std::string(name) == "arith") return h->arith;
    if (std::string(name).rfind("layer_arith:", 0) == 0) {
        const int li = std::atoi(name + 12);
        return (li >= 0 && li < (int)h->layer_arith.size()) ? h->layer_arith[li] : -1;
    }
    if (std::string(name) == "attn_w8") return h->attn_w8;
    if (std::string(name) == "posconv_split") return h->posconv_split ? 1 : 0;
    if (std::string(name) == "ln_split") return h->ln_split ? 1 : 0;
    if (std::string(name) == "vq_refine") return h->vq_refine ? 1 : 0;
    if (std::string(name) == "kmeans_split") return h->kmeans_split ? 1 : 0;
    return -1;
}

int at_hubert_encode(at_hubert_t* h, const float* wav, const float* mask, int B, int N, int n_layers, int16_t* tokens, int* T_out,
                     float* hidden_out, void* workspace, size_t workspace_bytes, at_stream_t stream_) {
    return at_hubert_encode_checked(h, wav, mask, B, N, n_layers, tokens, T_out, hidden_out, workspace, workspace_bytes, stream_, nullptr);
}

int at_hubert_encode_checked(at_hubert_t* h, const float* wav, const float* mask, int B, int N, int n_layers, int16_t* tokens, int* T_out,
                             float* hidden_out, void* workspace, size_t workspace_bytes, at_stream_t stream_, int32_t* status_dev) {
    AT_REQUIRE(h && h->finalized, "model not finalized");
    DeviceGuard guard(h->device);
    AT_REQUIRE(guard.ok, "cannot select the handle's device");
    AT_REQUIRE(wav && workspace, "null pointer");
    AT_REQUIRE(n_layers >= 0 && n_layers <= (int)h->layers.size(), "n_layers exceeds the loaded layers");
    AT_REQUIRE(tokens == nullptr || h->centers != nullptr, "tokens requested but no k-means centres loaded");
    const Plan p = make_plan(B, N);
    AT_REQUIRE(B >= 1 && p.L[7] >= 1, "clip too short (needs at least 400 samples)");
    AT_REQUIRE(workspace_bytes >= p.total_floats * sizeof(float), "workspace too small");
    hipStream_t stream = (hipStream_t)stream_;
    float* ws = (float*)workspace;
    const int T = p.L[7];
    const long long M = (long long)B * T;
    if (T_out) *T_out = T;
    Profiler& prof = h->prof;
    if (status_dev) AT_CHECK_HIP(hipMemsetAsync(status_dev, 0, sizeof(int32_t), (hipStream_t)stream_));
    const bool split = h->arith != ARITH_F32;
    AT_REQUIRE(n_layers + 1 <= kRangeRows, "more transformer layers than range-table rows");
    AT_CHECK_HIP(hipMemcpyAsync(h->range_tab, h->range_tab + kRangeInts, kRangeInts * sizeof(int), hipMemcpyDeviceToDevice, (hipStream_t)stream_));   // flags 0, census 0 / links
    const SplitCtx sc{h->arith == ARITH_F16X2 ? XB_SCHEME_F16X2 : XB_SCHEME_BF16X3, h->range_tab};   // the front end (row 0) and the k-means GEMM
    // transformer layer l: its own row of the range table and, when the range fallback has pinned it (option "layer_arith:<l>"), its own arithmetic
    auto arith_of = [&](int li) { return (split && li < (int)h->layer_arith.size() && h->layer_arith[li] > 0) ? h->layer_arith[li] : h->arith; };
    auto ctx_of = [&](int li) { return SplitCtx{arith_of(li) == ARITH_F16X2 ? XB_SCHEME_F16X2 : XB_SCHEME_BF16X3, h->range_tab + (1 + li) * 2 * (int)H_NSITES}; };

    // ---- conv feature encoder (7 valid strided convs, GroupNorm after the first, GELU) ----------------------
    float* bufs[2] = {ws + p.off_a, ws + p.off_b};
    prof.begin("feature_extractor", 9, stream);
    // conv0 + GroupNorm + GELU: statistics from float64 waveform moments, one pass over the output (hubert_kernels.hip)
    if (split) {
        // the six 512 -> 512 convs as windowed split-bf16 GEMMs (gemm_bf16x3.hip): conv0 writes the K-blocked bf16 pieces of its
        // output, every conv's GELU epilogue writes the next conv's input the same way, the last one writes fp32 features
        piece_t* sb[2] = {reinterpret_cast<piece_t*>(ws + p.off_sb), reinterpret_cast<piece_t*>(ws + p.off_sa)};   // [i & 1]
        if (int rc = launch_hub_conv0_gn_gelu(wav, h->conv_w[0], h->gn_g, h->gn_b, ws + p.off_part, ws + p.off_ss, nullptr, B, N, p.L[1], stream,
                                              sb[1], p.Lp[1], sc.scheme, sc.act_scale(), sc.site(HS_CONV0)))
            return rc;
        for (int i = 1; i < 7; ++i) {
            Bf16x3Args a;
            a.A = sb[i & 1]; a.W = h->conv_ws[sc.scheme][i]; a.M = p.L[i + 1]; a.Mpad = p.Mp[i]; a.N = kCd; a.K = kKs[i] * kCd;
            a.batch = B; a.stride = kSt[i]; a.cblocks = kCd / 16; a.Lp = p.Lp[i];
            a.scheme = sc.scheme; a.status = sc.site(HS_FE_CONV);
            if (sc.scheme == XB_SCHEME_F16X2) { a.acc_scale = 1.0f / (XB_F16_ACT_SCALE * h->conv_wscale[i]); a.split_scale = XB_F16_ACT_SCALE; }
            if (i < 6) { a.epi = XB_EPI_GELU_SPLIT; a.S = sb[(i + 1) & 1]; a.Spad = p.Lp[i + 1]; a.Sphases = kSt[i + 1]; }
            else { a.epi = XB_EPI_GELU; a.C = bufs[6 & 1]; a.ldc = kCd; }
            if (int rc = launch_gemm_bf16x3(a, stream)) return rc;
        }
    } else {
    if (int rc = launch_hub_conv0_gn_gelu(wav, h->conv_w[0], h->gn_g, h->gn_b, ws + p.off_part, ws + p.off_ss, bufs[0], B, N, p.L[1], stream))
        return rc;
    for (int i = 1; i < 7; ++i) {
        GemmArgs a;
        a.X = bufs[(i - 1) & 1]; a.x_bstride = (long long)p.L[i] * kCd; a.Tin = p.L[i]; a.Cin = kCd; a.ldx = kCd;
        a.ktaps = kKs[i]; a.stride = kSt[i]; a.pad_left = 0; a.pad_mode = 0;
        a.W = h->conv_w[i];
        a.C = bufs[i & 1]; a.c_bstride = (long long)p.L[i + 1] * kCd; a.ldc = kCd;
        a.M = p.L[i + 1]; a.N = kCd; a.K = kKs[i] * kCd; a.batch = B; a.epi = EPI_GELU;
        if (int rc = launch_gemm(a, stream)) return rc;
    }
    }
    prof.end(stream);
    const float* feats = bufs[6 & 1];   // [B][T][512]

    // ---- feature projection, zero padded frames, positional conv, LayerNorm (HF encoder entry) --------------
    float* fmask = ws + p.off_fmask;
    float* x = ws + p.off_x;
    float* t1 = ws + p.off_t1;
    float* pos = ws + p.off_pos;
    piece_t* xs = reinterpret_cast<piece_t*>(ws + p.off_xs);
    piece_t* bigs = reinterpret_cast<piece_t*>(ws + p.off_bigs);
    piece_t* kvs = reinterpret_cast<piece_t*>(ws + p.off_kvs);
    const bool attn_kvp = true;   // k / v as pieces from the projection's epilogue whenever the arithmetic is f16x2
    const int attn_arith = h->arith;   // attention follows the linear layers' arithmetic (0: the fp32-MFMA kernel)
    const long long Mpad = (long long)p.Mpad;
    float* big = ws + p.off_big;
    prof.begin("projection_posconv", 20, stream);
    if (int rc = launch_hub_frame_mask(mask, fmask, B, N, T, stream)) return rc;
    if (int rc = launch_layernorm(feats, h->fp_ln_g, h->fp_ln_b, nullptr, t1, M, kCd, stream)) return rc;
    if (int rc = linear(t1, kCd, h->fp_w, h->fp_b, x, kHid, M, EPI_NONE, nullptr, fmask, kHid, stream)) return rc;
    if (split && sc.scheme == XB_SCHEME_F16X2 && h->pos_ws && h->posconv_split) {
        // all 16 groups in one launch on the split scheme, the input tile resident in LDS (hubert_posconv.hip)
        if (int rc = launch_hubert_posconv(x, h->pos_ws, h->pos_b, pos, B, T, h->pos_wscale, sc.site(HS_X_IN), stream)) return rc;
    } else
    for (int g = 0; g < kGroups; ++g) {   // pos[b][t][g*48 + co] = x + gelu(conv_g(x) + bias)
        GemmArgs a;
        a.X = x + g * kGc; a.x_bstride = (long long)T * kHid; a.Tin = T; a.Cin = kGc; a.ldx = kHid;
        a.ktaps = kPosK; a.stride = 1; a.pad_left = kPosK / 2; a.pad_mode = 0;
        a.W = h->pos_w + (size_t)g * kGc * kPosK * kGc; a.bias = h->pos_b + g * kGc;
        a.C = pos + g * kGc; a.c_bstride = (long long)T * kHid; a.ldc = kHid;
        a.R = x + g * kGc; a.r_bstride = (long long)T * kHid; a.ldr = kHid;
        a.M = T; a.N = kGc; a.K = kPosK * kGc; a.batch = B; a.epi = EPI_GELU;
        if (int rc = launch_gemm(a, stream)) return rc;
    }
    // xs_ready: the LayerNorm that wrote x also wrote the pieces of x the next split GEMM reads (option "ln_split"; the scheme / scale / range row are the
    // CONSUMING layer's)
    bool xs_ready = false;
    auto ln_to = [&](const float* src, const float* g, const float* b, int consumer_layer, bool for_qkv) -> int {
        if (split && h->ln_split && consumer_layer < n_layers) {
            const SplitCtx c = ctx_of(consumer_layer);
            const LayerW& Lc = h->layers[consumer_layer];
            const float sc_x = c.scheme == XB_SCHEME_F16X2 ? (for_qkv ? Lc.xs_qkv : Lc.xs_ffn) : 1.0f;
            xs_ready = true;
            return launch_layernorm_split(src, g, b, nullptr, x, xs, M, Mpad, kHid, c.scheme, sc_x, c.site(HS_X_IN), stream);
        }
        xs_ready = false;
        return launch_layernorm(src, g, b, nullptr, x, M, kHid, stream);
    };
    if (int rc = ln_to(pos, h->enc_ln_g, h->enc_ln_b, 0, true)) return rc;
    prof.end(stream);

    for (int li = 0; li < n_layers; ++li) {
        const LayerW& L = h->layers[li];
        const SplitCtx scl = ctx_of(li);              // this layer's scheme and range row
        const int attn_arith_l = split ? arith_of(li) : attn_arith;
        prof.begin("attn_proj", 3, stream);
        // f16x2: the projection's epilogue writes k / v as fp16 pieces, the attention kernel stages them unsplit and writes its context as the
        // output projection's operand pieces (as in w2vbert.hip)
        const bool kvp = split && attn_arith_l == ARITH_F16X2 && scl.scheme == XB_SCHEME_F16X2 && attn_kvp;
        if (kvp) {
            if (!xs_ready)
                if (int rc = launch_split_blocked(x, kHid, M, Mpad, kHid, xs, stream, scl.scheme, L.xs_qkv, scl.site(HS_X_IN))) return rc;   // (kvp: the scheme is f16x2)
            Bf16x3Args qa;
            qa.A = xs; qa.W = L.ws[scl.scheme][HW_QKV]; qa.bias = L.bqkv; qa.M = (int)M; qa.N = 3 * kHid; qa.K = kHid; qa.Mpad = (int)Mpad;
            qa.epi = XB_EPI_QKV; qa.C = big; qa.ldc = 3 * kHid; qa.S = kvs; qa.Spad = (int)Mpad; qa.qkv_hid = kHid;
            qa.scheme = scl.scheme; qa.status = scl.site(HS_QKV_KV); qa.acc_scale = 1.0f / (L.xs_qkv * L.wscale[HW_QKV]); qa.split_scale = XB_F16_ACT_SCALE;
            if (int rc = launch_gemm_bf16x3(qa, stream)) return rc;
        } else if (split) {
            if (int rc = linear_split(scl, xs_ready ? nullptr : x, kHid, xs, xs, L, HW_QKV, L.bqkv, big, 3 * kHid, M, Mpad, XB_EPI_LINEAR, nullptr, 3 * kHid, nullptr, stream, L.xs_qkv)) return rc;
        } else if (int rc = linear(x, kHid, L.wqkv, L.bqkv, big, 3 * kHid, M, EPI_NONE, nullptr, nullptr, 3 * kHid, stream)) {
            return rc;
        }
        prof.end(stream);
        prof.begin("attention", 1, stream);
        const bool ctx_as_pieces = split && attn_arith_l > 0;
        if (int rc = launch_relpos_attention(big, fmask, nullptr, ctx_as_pieces ? nullptr : t1, B, T, stream, kHeads, attn_arith_l, scl.site(HS_ATTENTION),
                                             ctx_as_pieces ? xs : nullptr, Mpad, kvp ? kvs : nullptr, h->attn_w8)) return rc;
        prof.end(stream);
        prof.begin("attn_proj", 0, stream);
        if (split) {
            if (int rc = linear_split(scl, ctx_as_pieces ? nullptr : t1, kHid, xs, xs, L, HW_O, L.bo, x, kHid, M, Mpad, XB_EPI_LINEAR, x, kHid, nullptr, stream)) return rc;
        } else if (int rc = linear(t1, kHid, L.wo, L.bo, x, kHid, M, EPI_NONE, x, nullptr, kHid, stream)) {
            return rc;
        }
        if (int rc = ln_to(x, L.ln1_g, L.ln1_b, li, false)) return rc;
        prof.end(stream);
        prof.begin("ffn", 3, stream);
        if (split) {   // hidden activation written split by the first GEMM's epilogue
            if (int rc = linear_split(scl, xs_ready ? nullptr : x, kHid, xs, xs, L, HW_1, L.b1, nullptr, kFfn, M, Mpad, XB_EPI_GELU_SPLIT, nullptr, kFfn, bigs, stream, L.xs_ffn)) return rc;
            if (int rc = linear_split(scl, nullptr, kFfn, bigs, nullptr, L, HW_2, L.b2, x, kHid, M, Mpad, XB_EPI_LINEAR, x, kHid, nullptr, stream)) return rc;
        } else {
            if (int rc = linear(x, kHid, L.w1, L.b1, big, kFfn, M, EPI_GELU, nullptr, nullptr, kFfn, stream)) return rc;
            if (int rc = linear(big, kFfn, L.w2, L.b2, x, kHid, M, EPI_NONE, x, nullptr, kHid, stream)) return rc;
        }
        if (int rc = ln_to(x, L.ln2_g, L.ln2_b, li + 1, true)) return rc;     // (the last layer: plain LayerNorm, nothing consumes pieces)
        prof.end(stream);
    }
    if (status_dev)   // every site's range verdict of this call -> the caller's status word
        if (int rc = launch_range_combine(h->range_tab, (1 + n_layers) * (int)H_NSITES, reinterpret_cast<int*>(status_dev), stream)) return rc;
    if (hidden_out) AT_CHECK_HIP(hipMemcpyAsync(hidden_out, x, (size_t)M * kHid * sizeof(float), hipMemcpyDeviceToDevice, stream));
    if (tokens) {
        prof.begin("kmeans", 3, stream);
        if (int rc = launch_layernorm(x, nullptr, nullptr, nullptr, t1, M, kHid, stream)) return rc;
        if (split && h->kmeans_split && h->cen_s[sc.scheme]) {
            // the score GEMM on the split kernel against the centres padded to 1024 rows; the non-affine LayerNorm output is bounded by sqrt(768) = 27.7, so
            // x 16 cannot leave the fp16 range: no range site
            if (int rc = launch_split_blocked(t1, kHid, M, Mpad, kHid, xs, stream, sc.scheme, sc.act_scale(), nullptr)) return rc;
            Bf16x3Args va;
            va.A = xs; va.W = h->cen_s[sc.scheme]; va.bias = nullptr; va.M = (int)M; va.N = kCentersPad; va.K = kHid; va.Mpad = (int)Mpad;
            va.epi = XB_EPI_LINEAR; va.C = big; va.ldc = kCentersPad; va.R = nullptr; va.ldr = kCentersPad; va.alpha = 1.f;
            va.scheme = sc.scheme; va.status = nullptr;
            if (sc.scheme == XB_SCHEME_F16X2) { va.acc_scale = 1.0f / (XB_F16_ACT_SCALE * h->cen_scale); va.split_scale = XB_F16_ACT_SCALE; }
            if (int rc = launch_gemm_bf16x3(va, stream)) return rc;
            if (int rc = launch_vq_argmax(t1, big, h->c2, tokens, M, kHid, kCenters, stream, reinterpret_cast<int*>(status_dev), kCentersPad, h->vq_refine ? h->centers : nullptr)) return rc;
        } else {
            if (int rc = linear(t1, kHid, h->centers, nullptr, big, kCenters, M, EPI_NONE, nullptr, nullptr, kCenters, stream)) return rc;
            if (int rc = launch_vq_argmax(t1, big, h->c2, tokens, M, kHid, kCenters, stream, reinterpret_cast<int*>(status_dev), 0, h->vq_refine ? h->centers : nullptr)) return rc;
        }
        prof.end(stream);
    }
    return 0;
}

// The measured fp16 headroom of the LAST encode of this handle (see at_w2vbert_range_report)
int at_hubert_range_report(at_hubert_t* h, float* max_scaled, int cap) {
    AT_REQUIRE(h && h->finalized && h->range_tab && max_scaled && cap >= (int)H_NSITES, "at_hubert_range_report: bad arguments");
    DeviceGuard guard(h->device);
    AT_REQUIRE(guard.ok, "cannot select the handle's device");
    std::vector<int> host(kRangeInts);
    AT_CHECK_HIP(hipDeviceSynchronize());
    AT_CHECK_HIP(hipMemcpy(host.data(), h->range_tab, kRangeInts * sizeof(int), hipMemcpyDeviceToHost));
    for (int k = 0; k < (int)H_NSITES; ++k) {
        float f;   // row 0 holds the one census word of the site; the other rows' second words are links to it (split_scheme.h)
        std::memcpy(&f, &host[k * 2 + 1], sizeof(f));
        max_scaled[k] = f;
    }
    return (int)H_NSITES;
}
// Status flags of the LAST encode per part of the model: flags[0] = conv feature encoder + positional conv, flags[1 + l] = transformer layer l (the OR of its
// split sites; bit 1 = an activation left the fp16 range; the FIRST flagged entry is the cause, later ones inherit its infinities). Returns the number of
// entries written (1 + layers, <= cap). Synchronises the device. (As at_w2vbert_layer_status.)
int at_hubert_layer_status(at_hubert_t* h, int32_t* flags, int cap) {
    AT_REQUIRE(h && h->finalized && h->range_tab && flags && cap >= 1, "at_hubert_layer_status: bad arguments");
    DeviceGuard guard(h->device);
    AT_REQUIRE(guard.ok, "cannot select the handle's device");
    std::vector<int> host(kRangeInts);
    AT_CHECK_HIP(hipDeviceSynchronize());
    AT_CHECK_HIP(hipMemcpy(host.data(), h->range_tab, kRangeInts * sizeof(int), hipMemcpyDeviceToHost));
    const int n = std::min<int>({cap, 1 + (int)h->layers.size(), kRangeRows});
    for (int r = 0; r < n; ++r) {
        int v = 0;
        for (int k = 0; k < (int)H_NSITES; ++k) v |= host[(r * (int)H_NSITES + k) * 2];
        flags[r] = v;
    }
    return n;
}
// The activation scales of the two LayerNorm-fed split sites of every transformer layer: scales[2 l] = the q/k/v projection's input, scales[2 l + 1] = the
// first FFN GEMM's input (16 unless a LayerNorm's gains force the provable scale below that). Returns the number of floats written. Host-only.
int at_hubert_site_scales(const at_hubert_t* h, float* scales, int cap) {
    AT_REQUIRE(h && h->finalized && scales, "at_hubert_site_scales: bad arguments");
    const int n = 2 * (int)h->layers.size();
    AT_REQUIRE(cap >= n, "at_hubert_site_scales: buffer too small");
    for (size_t l = 0; l < h->layers.size(); ++l) { scales[2 * l] = h->layers[l].xs_qkv; scales[2 * l + 1] = h->layers[l].xs_ffn; }
    return n;
}
int at_hubert_range_sites(char* names, size_t cap) {
    std::string s;
    for (int k = 0; k < (int)H_NSITES; ++k) { s += kHSiteNames[k]; s += "\n"; }
    if (!names || cap < s.size() + 1) return -(int)(s.size() + 1);
    std::memcpy(names, s.c_str(), s.size() + 1);
    return (int)H_NSITES;
}

int at_hubert_profile(at_hubert_t* h, int enable) {
    AT_REQUIRE(h != nullptr, "null handle");
    h->prof.reset();
    h->prof.enabled = enable != 0;
    return 0;
}

int at_hubert_profile_read(at_hubert_t* h, char* names, size_t names_cap, float* total_ms, int* launches, int max_groups) {
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

}  // extern "C"
