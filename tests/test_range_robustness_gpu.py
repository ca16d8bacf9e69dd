"""GPU: the two-piece fp16 arithmetic on ranges it was not tuned on. The activation scale of the f16x2 kernels is a fixed
power of two (16: overflow beyond |x| > 4094), so (1) the per-site CENSUS (at_*_range_report: the largest |x * scale| every split site saw) measures
how close the synthetic-weight activations come to 65504; (2) LayerNorm gains / conv weights scaled x8 and x64 must still give the oracle's tokens —
x8 without leaving f16x2 (zero fallback batches); (3) when a batch does overflow, the fallback is per batch (test_acoustic_gpu / test_semantic_gpu)."""
import math

import numpy as np
import pytest
import torch

from audiotoken_amd import weights as W
from tests import parity as P

pytestmark = pytest.mark.gpu


def _headroom(rep):
    live = {k: v for k, v in rep.items() if v > 0}
    worst = max(live.values())
    return live, math.log2(65504.0 / worst)


def _scaled_w2vbert(gain, which):
    """which = "operands": the gains / biases of the LayerNorms in front of the FFNs and of the conv module, and the value projection, times `gain`
    — every split site's operands grow (LayerNorm outputs, FFN hidden, dwconv output, v, the attention context) while the softmax logits do not;
    "all": every LayerNorm of every layer, the attention one included — q and k grow too, the logits with gain^2: the network itself becomes
    ill-conditioned in fp32 (the oracle differs from its own float64 evaluation by 1e-2 in the quantised vectors at x8)."""
    w = dict(W.synth_w2vbert_weights(n_layers=3, seed=9, with_vq=True))
    g = np.float32(gain)
    for k in list(w):
        if not k.startswith("encoder.layers."):
            continue
        if "layer_norm" in k and (which == "all" or "self_attn_layer_norm" not in k) and "final_layer_norm" not in k:
            w[k] = (w[k] * g).astype(np.float32)
        if which == "operands" and ".self_attn.linear_v." in k:
            w[k] = (w[k] * g).astype(np.float32)
    return w


@pytest.mark.parametrize("which,gain", [("operands", 1.0), ("operands", 8.0), ("operands", 64.0), ("all", 8.0)])
def test_semantic_m_scaled_operands(cuda_device, which, gain):
    """A 3-layer conformer whose GEMM operands are `gain` times larger than the synthetic weights make them (see _scaled_w2vbert). Tokens must equal
    the oracle's on the same weights — at x1 and x8 on f16x2 itself (status 0, no fallback batch); the census shows the headroom shrinking by
    log2(gain) bits. For "all" the bar is the measured one of tests/parity.py (a flip needs an oracle margin below twice the measured vector
    difference, itself bounded by the oracle's own float32-vs-float64 difference)."""
    from audiotoken_amd.configs import Wav2VecBertConfig
    from audiotoken_amd.encoder import Wav2VecBertEncoder
    from oracle import w2vbert_ref as R
    w = _scaled_w2vbert(gain, which)
    enc = Wav2VecBertEncoder(Wav2VecBertConfig(output_layer=3), device="cuda:0", quantize=True, weights=w)
    wav = torch.from_numpy(W.synth_waveform(2, 48000, 16000, seed=41))
    mask = torch.ones_like(wav)
    x, m = wav.cuda(), mask.cuda()
    toks, taps = enc(x, m, return_taps=True)
    status = enc.last_status()
    live, bits = _headroom(enc.range_report())
    print(f"[range] semantic_m {which} x{gain:g}: status {status}, census {dict((k, round(v, 1)) for k, v in live.items())}, headroom {bits:.1f} bits")
    assert set(live) >= {"ln_ffn1", "ffn1_hidden", "ln_attn", "qkv_kv", "attention", "ln_conv", "dwconv_out", "ln_ffn2", "ffn2_hidden"}
    if gain <= 8.0:
        assert status == 0 and bits > 0, "f16x2 must hold at x8"
    if status != 0:
        toks = enc.verified(toks, x, m)
        assert enc.fallback_batches == 1 and enc.get_option("arith") == 2
        _, taps = enc(x, m, return_taps=True)     # (taps of the f16x2 run are invalid after an overflow; not compared then)
    wt = {k: torch.from_numpy(v) for k, v in w.items()}
    ref, margins = R.semantic_m_encode(wt, wav, mask, 2, 3, return_margins=True)
    feats, am = R.processor(wav, mask, 2)
    valid = am.bool().unsqueeze(1)
    if which == "all":
        x_ref = R.layer_norm(R.encoder_hidden_state(wt, feats, am, 3), wt, None, 1024)
        # the oracle evaluated exactly: float64 front-end AND float64 network. tools/ln_gain_probe.py: the reference's fp32 front-end leaves its
        # features 5e-4 from exact; after the feature projection that is 1.8e-3 at scale 4.4 — the HIP path, whose front-end is
        # float64, sits that far from the fp32 oracle for EVERY arithmetic (f32, bf16x3, f16x2 alike) — and this network multiplies it by ~300
        wt64 = {k: v.double() for k, v in wt.items()}
        feats64, _ = R.processor(wav.double(), mask.double(), 2)
        x_ref_exact = torch.nn.functional.layer_norm(R.encoder_hidden_state(wt64, feats64, am.double(), 3), (1024,)).float()
        x_got = torch.nn.functional.layer_norm(taps["hidden"].cpu(), (1024,))
        P.assert_tokens_equal_or_explained_by_delta(toks, ref, margins, x_got, x_ref, P.VQ_TIE, f"[range] semantic_m {which} x{gain:g}", valid, x_ref_exact)
    else:
        P.assert_tokens_equal_or_explained(toks, ref, margins, P.VQ_TIE, f"[range] semantic_m {which} x{gain:g}", valid)


@pytest.mark.parametrize("gain", [1.0, 8.0, 64.0])
def test_acoustic_conv_weight_scale(cuda_device, gain):
    """The stage-1 strided conv's weight (weight_g of its weight-norm pair) times `gain`: every later activation of the SEANet encoder, the LSTM
    input and the RVQ residual grow with it. Tokens equal the oracle's; x8 stays on f16x2."""
    from audiotoken_amd.configs import AcousticEncoderConfig
    from audiotoken_amd.encoder import AcousticEncoder
    from oracle import encodec_ref as R
    w = dict(W.synth_encodec_weights(seed=7, with_decoder=False))
    key = "encoder.model.6.conv.conv.weight_g"     # ELU, conv 64 -> 128 k8 s4 (layer 5-6)
    assert key in w
    w[key] = (w[key] * np.float32(gain)).astype(np.float32)
    enc = AcousticEncoder(AcousticEncoderConfig(bandwidth=6), device="cuda:0", weights=w)
    wav = torch.from_numpy(W.synth_waveform(3, 48000, 24000, seed=42))
    x = wav.cuda()
    codes = enc(x, None)
    status = enc.last_status()
    live, bits = _headroom(enc.range_report())
    print(f"[range] acoustic down1 x{gain:g}: status {status}, census {dict((k, round(v, 1)) for k, v in live.items())}, headroom {bits:.1f} bits")
    assert set(live) >= {"stage0", "res1", "down1", "res2", "down2", "res3_conv", "res3_tail", "lstm_ih", "final_conv_in", "rvq"}
    if gain <= 8.0:
        assert status == 0 and bits > 0, "f16x2 must hold at x8"
    codes = enc.verified(codes, x, None)
    assert enc.fallback_batches == (0 if status == 0 else 1)
    ref, margins = R.acoustic_encode(w, wav, 8, return_margins=True)
    P.assert_rvq_equal_or_explained(codes, ref, margins, P.RVQ_TIE, f"[range] acoustic down1 x{gain:g}")


def test_semantic_s_census(cuda_device):
    from audiotoken_amd.configs import HubertEncoderConfig
    from audiotoken_amd.hubert import HubertEncoder, hubert_processor
    w = W.synth_hubert_weights(3, 0, True)
    enc = HubertEncoder(HubertEncoderConfig(output_layer=3), device="cuda:0", quantize=True, weights=w)
    wav = hubert_processor(torch.from_numpy(W.synth_waveform(1, 32000, 16000, seed=43)))
    enc(wav.cuda(), torch.ones_like(wav).cuda())
    assert enc.last_status() == 0
    live, bits = _headroom(enc.range_report())
    print(f"[range] semantic_s: census {dict((k, round(v, 1)) for k, v in live.items())}, headroom {bits:.1f} bits")
    assert set(live) >= {"conv0_out", "feature_convs", "layer_input", "qkv_kv", "attention", "ffn_hidden"} and bits > 0


@pytest.mark.parametrize("tokenizer", ["acoustic", "semantic_m", "semantic_s"])
@pytest.mark.parametrize("bad", [float("nan"), float("inf")], ids=["nan", "inf"])
def test_nonfinite_input_is_reported(cuda_device, tokenizer, bad):
    """A NaN / infinity in the caller's waveform: the range bookkeeping's running maximum drops NaNs (fmaxf), so without a check such a batch would
    finish with status 0. The quantisers (RVQ, VQ, k-means) raise status bit 2 when a vector they quantise is not finite; `verified` logs it, counts it
    and returns the ids as they are (the reference emits arbitrary ids there without a diagnostic). A clean batch on the same handle reads 0 again."""
    if tokenizer == "acoustic":
        from audiotoken_amd.configs import AcousticEncoderConfig
        from audiotoken_amd.encoder import AcousticEncoder
        enc = AcousticEncoder(AcousticEncoderConfig(bandwidth=6), device="cuda:0", weights=W.synth_encodec_weights(seed=0, with_decoder=False))
        wav = torch.from_numpy(W.synth_waveform(2, 24000, 24000, seed=5))
    elif tokenizer == "semantic_m":
        from audiotoken_amd.configs import Wav2VecBertConfig
        from audiotoken_amd.encoder import Wav2VecBertEncoder
        enc = Wav2VecBertEncoder(Wav2VecBertConfig(output_layer=2), device="cuda:0", quantize=True, weights=W.synth_w2vbert_weights(n_layers=2, seed=9, with_vq=True))
        wav = torch.from_numpy(W.synth_waveform(2, 32000, 16000, seed=5))
    else:
        from audiotoken_amd.configs import HubertEncoderConfig
        from audiotoken_amd.hubert import HubertEncoder
        enc = HubertEncoder(HubertEncoderConfig(output_layer=2), device="cuda:0", weights=W.synth_hubert_weights(2, 0, True))
        wav = torch.from_numpy(W.synth_waveform(2, 32000, 16000, seed=5))
    mask = torch.ones_like(wav)
    clean = enc(wav.cuda(), mask.cuda())
    assert enc.last_status() == 0
    dirty = wav.clone()
    dirty[1, 12345] = bad
    toks = enc(dirty.cuda(), mask.cuda())
    status = enc.last_status()
    print(f"[range] {tokenizer}: a {bad} sample in clip 1 -> status {status}")
    assert status & 4, f"{tokenizer}: a non-finite input finished with status {status}"
    out = enc.verified(toks, dirty.cuda(), mask.cuda())
    assert enc.nonfinite_batches == 1 and out.shape == clean.shape
    again = enc(wav.cuda(), mask.cuda())
    assert enc.last_status() == 0 and torch.equal(again, clean)


OUTLIER_CHANNELS = (5, 261, 700, 1019)


def test_semantic_m_per_channel_outliers(cuda_device):
    """Real transformer checkpoints carry a few channels whose LayerNorm gains are 100-1000 x the median; uniform gains do not
    model that. Here FOUR channels of every FFN-input LayerNorm (ffn1 / ffn2, gain and bias) of every layer are x 256: the split sites downstream see a
    256-fold spread inside one operand row. Tokens must equal the oracle's on the same weights, on the f16x2 arithmetic itself (status 0, no fallback
    batch) — the fixed activation scale (x 16, overflow at |x| > 4094) must hold without a per-site rescue — and the census shows what is left."""
    from audiotoken_amd.configs import Wav2VecBertConfig
    from audiotoken_amd.encoder import Wav2VecBertEncoder
    from oracle import w2vbert_ref as R
    w = dict(W.synth_w2vbert_weights(n_layers=3, seed=9, with_vq=True))
    n = 0
    for k in list(w):
        if k.startswith("encoder.layers.") and ("ffn1_layer_norm" in k or "ffn2_layer_norm" in k):
            v = w[k].copy()
            v[list(OUTLIER_CHANNELS)] *= np.float32(256.0)
            w[k] = v
            n += 1
    assert n == 3 * 2 * 2
    enc = Wav2VecBertEncoder(Wav2VecBertConfig(output_layer=3), device="cuda:0", quantize=True, weights=w)
    wav = torch.from_numpy(W.synth_waveform(2, 48000, 16000, seed=41))
    mask = torch.ones_like(wav)
    toks = enc(wav.cuda(), mask.cuda())
    status = enc.last_status()
    live, bits = _headroom(enc.range_report())
    print(f"[range] semantic_m, 4 outlier channels x256 in every FFN LayerNorm: status {status}, census {dict((k, round(v, 1)) for k, v in live.items())}, headroom {bits:.1f} bits")
    assert status == 0 and bits > 0, "four x256 channels must not leave the fp16 range"
    toks = enc.verified(toks, wav.cuda(), mask.cuda())
    assert enc.fallback_batches == 0
    wt = {k: torch.from_numpy(v) for k, v in w.items()}
    ref, margins = R.semantic_m_encode(wt, wav, mask, 2, 3, return_margins=True)
    P.assert_tokens_equal_or_explained(toks, ref, margins, P.VQ_TIE, "[range] semantic_m per-channel outliers x256")


def test_acoustic_per_channel_outliers(cuda_device):
    """The same for the SEANet encoder: four output channels of the stage-1 strided conv (encoder.model.6: weight_g and bias) x 256."""
    from audiotoken_amd.configs import AcousticEncoderConfig
    from audiotoken_amd.encoder import AcousticEncoder
    from oracle import encodec_ref as R
    w = dict(W.synth_encodec_weights(seed=7, with_decoder=False))
    for key in ("encoder.model.6.conv.conv.weight_g", "encoder.model.6.conv.conv.bias"):
        v = w[key].copy()
        v[[3, 40, 77, 120]] *= np.float32(256.0)
        w[key] = v
    enc = AcousticEncoder(AcousticEncoderConfig(bandwidth=6), device="cuda:0", weights=w)
    wav = torch.from_numpy(W.synth_waveform(3, 48000, 24000, seed=42))
    codes = enc(wav.cuda(), None)
    status = enc.last_status()
    live, bits = _headroom(enc.range_report())
    print(f"[range] acoustic, 4 outlier channels x256 in encoder.model.6: status {status}, census {dict((k, round(v, 1)) for k, v in live.items())}, headroom {bits:.1f} bits")
    codes = enc.verified(codes, wav.cuda(), None)
    print(f"    fallback batches {enc.fallback_batches}")
    ref, margins = R.acoustic_encode(w, wav, 8, return_margins=True)
    P.assert_rvq_equal_or_explained(codes, ref, margins, P.RVQ_TIE, "[range] acoustic per-channel outliers x256")
    assert status == 0 and enc.fallback_batches == 0, "four x256 channels must not leave the fp16 range"


def test_semantic_m_bad_layer_is_pinned_not_the_batch(cuda_device):
    """Per-layer range fallback (round 4). ONE conformer layer whose activations leave the fp16 range (layer 1: the bias of one FFN hidden unit at 6 000,
    swish(.) x 16 > 65504) — what a real checkpoint with an activation outlier does on every batch. The status word reports it, the per-layer flags name
    layer 1 first (layer 2 only inherits its infinities), verified() moves layer 1 alone to bf16x3 for good and repeats the batch: tokens equal the
    oracle's, the NEXT batch runs clean without a repeat (round 3 repeated every batch on bf16x3 for all layers). Mixed arithmetic per layer is also checked
    on healthy weights (the fused final-LayerNorm / next-layer-LayerNorm pass writes the NEXT layer's scheme)."""
    from audiotoken_amd.configs import Wav2VecBertConfig
    from audiotoken_amd.encoder import Wav2VecBertEncoder
    from oracle import w2vbert_ref as R
    wav = torch.from_numpy(W.synth_waveform(2, 48000, 16000, seed=43))
    mask = torch.ones_like(wav)
    x, m = wav.cuda(), mask.cuda()
    valid = R.processor(wav, mask, 2)[1].bool().unsqueeze(1)

    healthy = W.synth_w2vbert_weights(n_layers=3, seed=9, with_vq=True)
    enc = Wav2VecBertEncoder(Wav2VecBertConfig(output_layer=3), device="cuda:0", quantize=True, weights=healthy)
    ref, margins = R.semantic_m_encode({k: torch.from_numpy(v) for k, v in healthy.items()}, wav, mask, 2, 3, return_margins=True)
    for pinned in ((0,), (1,), (2,), (0, 2)):
        for l in pinned:
            enc.set_option(f"layer_arith:{l}", 1)
        toks = enc(x, m)
        assert enc.last_status() == 0 and [enc.get_option(f"layer_arith:{l}") for l in range(3)] == [1 if l in pinned else -1 for l in range(3)]
        P.assert_tokens_equal_or_explained(toks, ref, margins, P.VQ_TIE, f"[range] semantic_m layers {pinned} on bf16x3", valid)
        for l in pinned:
            enc.set_option(f"layer_arith:{l}", -1)

    w = dict(healthy)
    b = w["encoder.layers.1.ffn1.intermediate_dense.bias"].copy()
    b[0] = 6000.0
    w["encoder.layers.1.ffn1.intermediate_dense.bias"] = b
    enc = Wav2VecBertEncoder(Wav2VecBertConfig(output_layer=3), device="cuda:0", quantize=True, weights=w)
    ref, margins = R.semantic_m_encode({k: torch.from_numpy(v) for k, v in w.items()}, wav, mask, 2, 3, return_margins=True)
    toks = enc(x, m)
    assert enc.last_status() & 2, "the poisoned layer must overflow the fp16 range"
    flags = enc.layer_status()
    assert len(flags) == 3 and flags[0] == 0 and flags[1] & 2, flags
    toks = enc.verified(toks, x, m)
    # round 5 (ADVICE): the FIRST overflowing batch is repeated with layer 1 on bf16x3 and the layer goes back to f16x2 (the outlier might have come with the input) ...
    assert enc.pinned_layers == [] and enc.fallback_batches == 1 and enc.get_option("arith") == 2 and enc.get_option("layer_arith:1") == -1
    P.assert_tokens_equal_or_explained(toks, ref, margins, P.VQ_TIE, "[range] semantic_m layer 1 on bf16x3 for one batch (repeat)", valid)
    toks1 = enc(x, m)                           # ... the second one pins it (a property of the checkpoint)
    assert enc.last_status() & 2
    toks1 = enc.verified(toks1, x, m)
    assert enc.pinned_layers == [1] and enc.fallback_batches == 2 and enc.get_option("layer_arith:1") == 1 and torch.equal(toks1, toks)
    toks2 = enc(x, m)                           # the next batch: no overflow, no repeat
    assert enc.last_status() == 0 and torch.equal(enc.verified(toks2, x, m), toks) and enc.fallback_batches == 2
    enc.unpin_layers()
    enc(x, m)
    assert enc.last_status() & 2 and enc.pinned_layers == []


def test_semantic_s_bad_layer_is_pinned_not_the_batch(cuda_device):
    """The per-layer range fallback for the HuBERT tokenizer (as test_semantic_m_bad_layer_is_pinned_not_the_batch): transformer layer 1's FFN hidden
    overflows the fp16 range on every batch (one bias at 6 000: GELU(.) x 16 > 65504). layer_status()[0] (the conv front end) stays clean, entry 2
    (= layer 1) is the first flagged one; verified() pins that layer to bf16x3 and repeats the batch; the next batch runs clean; mixed arithmetic per layer
    equals the oracle on healthy weights."""
    from audiotoken_amd.configs import HubertEncoderConfig
    from audiotoken_amd.hubert import HubertEncoder, hubert_processor
    from oracle import hubert_ref as R
    wav = hubert_processor(torch.from_numpy(W.synth_waveform(2, 48000, 16000, seed=47)))
    mask = torch.ones_like(wav)
    x, m = wav.cuda(), mask.cuda()
    healthy = W.synth_hubert_weights(3, 19, True)
    enc = HubertEncoder(HubertEncoderConfig(output_layer=3), device="cuda:0", quantize=True, weights=healthy)
    ref, margins = R.semantic_s_encode(healthy, wav, mask, 3, return_margins=True)
    for pinned in ((0,), (2,), (0, 1)):
        for l in pinned:
            enc.set_option(f"layer_arith:{l}", 1)
        toks = enc(x, m)
        assert enc.last_status() == 0
        P.assert_tokens_equal_or_explained(toks, ref, margins, P.VQ_TIE, f"[range] semantic_s layers {pinned} on bf16x3")
        for l in pinned:
            enc.set_option(f"layer_arith:{l}", -1)

    w = dict(healthy)
    key = "encoder.layers.1.feed_forward.intermediate_dense.bias"
    b = np.array(w[key], dtype=np.float32).copy()
    b[0] = 6000.0
    w[key] = b
    enc = HubertEncoder(HubertEncoderConfig(output_layer=3), device="cuda:0", quantize=True, weights=w)
    ref, margins = R.semantic_s_encode(w, wav, mask, 3, return_margins=True)
    toks = enc(x, m)
    assert enc.last_status() & 2, "the poisoned layer must overflow the fp16 range"
    flags = enc.layer_status()
    assert len(flags) == 4 and flags[0] == 0 and flags[1] == 0 and flags[2] & 2, flags
    toks = enc.verified(toks, x, m)
    assert enc.pinned_layers == [] and enc.fallback_batches == 1 and enc.get_option("arith") == 2 and enc.get_option("layer_arith:1") == -1
    P.assert_tokens_equal_or_explained(toks, ref, margins, P.VQ_TIE, "[range] semantic_s layer 1 on bf16x3 for one batch (repeat)")
    toks1 = enc(x, m)
    assert enc.last_status() & 2
    toks1 = enc.verified(toks1, x, m)
    assert enc.pinned_layers == [1] and enc.fallback_batches == 2 and enc.get_option("layer_arith:1") == 1 and torch.equal(toks1, toks)
    toks2 = enc(x, m)
    assert enc.last_status() == 0 and torch.equal(enc.verified(toks2, x, m), toks) and enc.fallback_batches == 2
    enc.unpin_layers()
    enc(x, m)
    assert enc.last_status() & 2 and enc.pinned_layers == []


# ---- round 5: provable activation scales of the LayerNorm-fed split sites --------------------------------------------------------------------------
def test_ln_fed_sites_have_provable_scales(cuda_device):
    """|LN(x)_k| <= sqrt(D) |gamma_k| + |beta_k| for any input, so finalize can pick, per LayerNorm-fed split site, a power-of-two activation scale that cannot
    overflow fp16 whatever the data (include/audiotoken_hip.h, at_w2vbert_site_scales). (1) Ordinary gains: every site keeps 16 — nothing computed in
    rounds 2-4 moves. (2) One ffn1 LayerNorm with gains x 400 (sqrt(1024) x 400 x 1.2 x 16 = 245 000 > 65 504: the fixed 16 COULD overflow, and does on a row
    dominated by one channel): that site's scale drops to the provable one, the other sites keep 16, the encode reports status 0 with NO fallback batch and
    the tokens equal the oracle's on the same weights — also for the adversarial input that reaches the bound."""
    from audiotoken_amd.configs import Wav2VecBertConfig
    from audiotoken_amd.encoder import Wav2VecBertEncoder
    from oracle import w2vbert_ref as R
    w = dict(W.synth_w2vbert_weights(n_layers=3, seed=11, with_vq=True))
    enc = Wav2VecBertEncoder(Wav2VecBertConfig(output_layer=3), device="cuda:0", quantize=True, weights=w)
    assert all(s == 16.0 for v in enc.site_scales().values() for s in v)
    key = "encoder.layers.1.ffn1_layer_norm.weight"
    g = w[key].copy()
    g[[5, 300, 777]] *= np.float32(400.0)
    w[key] = g
    enc = Wav2VecBertEncoder(Wav2VecBertConfig(output_layer=3), device="cuda:0", quantize=True, weights=w)
    scales = enc.site_scales()
    bound = 32.0 * float(np.abs(g).max()) + float(np.abs(w[key.replace("weight", "bias")]).max())
    want = 16.0
    while want * bound > 65000.0:
        want /= 2
    print(f"[range] provable scale of ln_ffn1, layer 1: bound {bound:.0f} -> scale {scales['ln_ffn1'][1]} (fixed 16 would reach {16 * bound:.0f})")
    assert want < 16.0 and scales["ln_ffn1"] == [16.0, want, 16.0]
    assert all(s == 16.0 for k, v in scales.items() if k != "ln_ffn1" for s in v)
    wav = torch.from_numpy(W.synth_waveform(2, 48000, 16000, seed=44))
    mask = torch.ones_like(wav)
    toks = enc(wav.cuda(), mask.cuda())
    assert enc.last_status() == 0
    assert enc.verified(toks, wav.cuda(), mask.cuda()) is toks and enc.fallback_batches == 0 and enc.pinned_layers == []
    live, bits = _headroom(enc.range_report())
    print(f"[range] gains x 400 on three channels: census {dict((k, round(v, 1)) for k, v in live.items())}, headroom {bits:.1f} bits")
    wt = {k: torch.from_numpy(v) for k, v in w.items()}
    ref, margins = R.semantic_m_encode(wt, wav, mask, 2, 3, return_margins=True)
    P.assert_tokens_equal_or_explained(toks, ref, margins, P.VQ_TIE, "[range] semantic_m, one LayerNorm with gains x 400")


def test_hubert_ln_fed_sites_have_provable_scales(cuda_device):
    from audiotoken_amd.configs import HubertEncoderConfig
    from audiotoken_amd.hubert import HubertEncoder, hubert_processor
    from oracle import hubert_ref as R
    w = dict(W.synth_hubert_weights(3, 5, True))
    enc = HubertEncoder(HubertEncoderConfig(output_layer=3), device="cuda:0", quantize=True, weights=w)
    assert all(s == (16.0, 16.0) for s in enc.site_scales())
    key = "encoder.layers.0.final_layer_norm.weight"          # writes the residual stream layer 1's q/k/v projection reads
    g = w[key].copy()
    g[[9, 400]] *= np.float32(500.0)
    w[key] = g
    enc = HubertEncoder(HubertEncoderConfig(output_layer=3), device="cuda:0", quantize=True, weights=w)
    scales = enc.site_scales()
    print(f"[range] HuBERT provable scales: {scales}")
    assert scales[0] == (16.0, 16.0) and scales[1][0] < 16.0 and scales[1][1] == 16.0 and scales[2] == (16.0, 16.0)
    wav = hubert_processor(torch.from_numpy(W.synth_waveform(1, 48000, 16000, seed=45)))
    mask = torch.ones_like(wav)
    toks = enc(wav.cuda(), mask.cuda())
    assert enc.last_status() == 0
    ref, margins = R.semantic_s_encode(w, wav, mask, 3, return_margins=True)
    P.assert_tokens_equal_or_explained(toks, ref, margins, P.VQ_TIE, "[range] semantic_s, one LayerNorm with gains x 500")
