// at_required_tensors: the named host tensors (after weight-norm folding) each model's finalize() needs, as a machine-readable list.
// A loader for a new checkpoint format can check its output against this list before touching the device; tests/test_checkpoints_*.py
// check it against HF save_pretrained directories and (on the GPU) against what finalize() actually accepts.
#include <string>
#include <vector>
#include <cstring>

#include "../../include/audiotoken_hip.h"
#include "at_common.h"

namespace {
struct Entry { std::string name; std::vector<int> shape; };

void conv(std::vector<Entry>& v, const std::string& base, int cout, int cin, int k) {
    v.push_back({base + ".weight", {cout, cin, k}});
    v.push_back({base + ".bias", {cout}});
}
void lstm(std::vector<Entry>& v, const std::string& base) {
    for (int l = 0; l < 2; ++l) {
        const std::string s = std::to_string(l);
        v.push_back({base + ".lstm.weight_ih_l" + s, {2048, 512}});
        v.push_back({base + ".lstm.weight_hh_l" + s, {2048, 512}});
        v.push_back({base + ".lstm.bias_ih_l" + s, {2048}});
        v.push_back({base + ".lstm.bias_hh_l" + s, {2048}});
    }
}

std::vector<Entry> encodec(int n_codebooks, bool decoder) {
    std::vector<Entry> v;
    const int ratios_enc[4] = {2, 4, 5, 8}, ratios_dec[4] = {8, 5, 4, 2};
    conv(v, "encoder.model.0.conv.conv", 32, 1, 7);
    int C = 32, idx = 1;
    for (int s = 0; s < 4; ++s) {
        const std::string base = "encoder.model." + std::to_string(idx);
        conv(v, base + ".block.1.conv.conv", C / 2, C, 3);
        conv(v, base + ".block.3.conv.conv", C, C / 2, 1);
        conv(v, base + ".shortcut.conv.conv", C, C, 1);
        conv(v, "encoder.model." + std::to_string(idx + 2) + ".conv.conv", 2 * C, C, 2 * ratios_enc[s]);
        C *= 2;
        idx += 3;
    }
    lstm(v, "encoder.model.13");
    conv(v, "encoder.model.15.conv.conv", 128, 512, 7);
    for (int q = 0; q < n_codebooks; ++q) v.push_back({"quantizer.vq.layers." + std::to_string(q) + "._codebook.embed", {1024, 128}});
    if (decoder) {
        conv(v, "decoder.model.0.conv.conv", 512, 128, 7);
        lstm(v, "decoder.model.1");
        int Cd = 512, di = 3;
        for (int s = 0; s < 4; ++s) {
            v.push_back({"decoder.model." + std::to_string(di) + ".convtr.convtr.weight", {Cd, Cd / 2, 2 * ratios_dec[s]}});   // ConvTranspose1d: [in][out][k]
            v.push_back({"decoder.model." + std::to_string(di) + ".convtr.convtr.bias", {Cd / 2}});
            Cd /= 2;
            const std::string base = "decoder.model." + std::to_string(di + 1);
            conv(v, base + ".block.1.conv.conv", Cd / 2, Cd, 3);
            conv(v, base + ".block.3.conv.conv", Cd, Cd / 2, 1);
            conv(v, base + ".shortcut.conv.conv", Cd, Cd, 1);
            di += 3;
        }
        conv(v, "decoder.model.15.conv.conv", 1, 32, 7);
    }
    return v;
}

std::vector<Entry> w2vbert(int n_layers, bool vq) {
    std::vector<Entry> v;
    const int H = 1024, F = 4096;
    v.push_back({"frontend.window", {400}});
    v.push_back({"frontend.mel_filters", {257, 80}});
    v.push_back({"feature_projection.layer_norm.weight", {160}});
    v.push_back({"feature_projection.layer_norm.bias", {160}});
    v.push_back({"feature_projection.projection.weight", {H, 160}});
    v.push_back({"feature_projection.projection.bias", {H}});
    auto ln = [&](const std::string& b) { v.push_back({b + ".weight", {H}}); v.push_back({b + ".bias", {H}}); };
    auto lin = [&](const std::string& b, int o, int i) { v.push_back({b + ".weight", {o, i}}); v.push_back({b + ".bias", {o}}); };
    for (int i = 0; i < n_layers; ++i) {
        const std::string p = "encoder.layers." + std::to_string(i);
        ln(p + ".ffn1_layer_norm");
        lin(p + ".ffn1.intermediate_dense", F, H);
        lin(p + ".ffn1.output_dense", H, F);
        ln(p + ".self_attn_layer_norm");
        lin(p + ".self_attn.linear_q", H, H);
        lin(p + ".self_attn.linear_k", H, H);
        lin(p + ".self_attn.linear_v", H, H);
        v.push_back({p + ".self_attn.distance_embedding.weight", {73, 64}});
        lin(p + ".self_attn.linear_out", H, H);
        ln(p + ".conv_module.layer_norm");
        v.push_back({p + ".conv_module.pointwise_conv1.weight", {2 * H, H, 1}});
        v.push_back({p + ".conv_module.depthwise_conv.weight", {H, 1, 31}});
        ln(p + ".conv_module.depthwise_layer_norm");
        v.push_back({p + ".conv_module.pointwise_conv2.weight", {H, H, 1}});
        ln(p + ".ffn2_layer_norm");
        lin(p + ".ffn2.intermediate_dense", F, H);
        lin(p + ".ffn2.output_dense", H, F);
        ln(p + ".final_layer_norm");
    }
    if (vq) v.push_back({"vq._codebook.embed", {1, 2048, H}});
    return v;
}

std::vector<Entry> hubert(int n_layers, bool kmeans) {
    std::vector<Entry> v;
    const int Cd = 512, H = 768, F = 3072;
    const int ks[7] = {10, 3, 3, 3, 3, 2, 2};
    for (int i = 0; i < 7; ++i) v.push_back({"feature_extractor.conv_layers." + std::to_string(i) + ".conv.weight", {Cd, i == 0 ? 1 : Cd, ks[i]}});
    v.push_back({"feature_extractor.conv_layers.0.layer_norm.weight", {Cd}});
    v.push_back({"feature_extractor.conv_layers.0.layer_norm.bias", {Cd}});
    v.push_back({"feature_projection.layer_norm.weight", {Cd}});
    v.push_back({"feature_projection.layer_norm.bias", {Cd}});
    v.push_back({"feature_projection.projection.weight", {H, Cd}});
    v.push_back({"feature_projection.projection.bias", {H}});
    v.push_back({"encoder.pos_conv_embed.conv.weight", {H, 48, 128}});
    v.push_back({"encoder.pos_conv_embed.conv.bias", {H}});
    v.push_back({"encoder.layer_norm.weight", {H}});
    v.push_back({"encoder.layer_norm.bias", {H}});
    auto lin = [&](const std::string& b, int o, int i) { v.push_back({b + ".weight", {o, i}}); v.push_back({b + ".bias", {o}}); };
    for (int i = 0; i < n_layers; ++i) {
        const std::string p = "encoder.layers." + std::to_string(i);
        lin(p + ".attention.q_proj", H, H);
        lin(p + ".attention.k_proj", H, H);
        lin(p + ".attention.v_proj", H, H);
        lin(p + ".attention.out_proj", H, H);
        v.push_back({p + ".layer_norm.weight", {H}});
        v.push_back({p + ".layer_norm.bias", {H}});
        lin(p + ".feed_forward.intermediate_dense", F, H);
        lin(p + ".feed_forward.output_dense", H, F);
        v.push_back({p + ".final_layer_norm.weight", {H}});
        v.push_back({p + ".final_layer_norm.bias", {H}});
    }
    if (kmeans) v.push_back({"kmeans.cluster_centers_", {1000, H}});
    return v;
}
}  // namespace

extern "C" int at_required_tensors(const char* model, int n, int with_extras, char* buf, size_t cap) {
    using namespace at;
    AT_REQUIRE(model && n >= 0, "at_required_tensors: bad arguments");
    const std::string m(model);
    std::vector<Entry> v;
    if (m == "encodec") v = encodec(n, with_extras != 0);
    else if (m == "w2vbert") v = w2vbert(n, with_extras != 0);
    else if (m == "hubert") v = hubert(n, with_extras != 0);
    else { set_error("at_required_tensors: unknown model " + m); return -1; }
    std::string out;
    for (const Entry& e : v) {
        out += e.name;
        for (int d : e.shape) { out += ' '; out += std::to_string(d); }
        out += '\n';
    }
    if (buf == nullptr || out.size() + 1 > cap) return -(int)(out.size() + 1);   // negative = bytes needed
    std::memcpy(buf, out.c_str(), out.size() + 1);
    return (int)v.size();
}
