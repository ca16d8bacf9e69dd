"""semantic_s (HuBERT) — CPU: oracle vs the HF-generated golden; GPU: HIP path vs golden and oracle."""
import os

import numpy as np
import pytest
import torch

from audiotoken_amd import weights as W
from oracle import hubert_ref as R

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "hubert_a.npz"))


def _inputs():
    B, N = int(G["B"]), int(G["N"])
    mask = G["mask"]
    wave = W.synth_waveform(B, N, 16000, seed=int(G["wave_seed"])) * mask
    return torch.from_numpy(wave), torch.from_numpy(mask)


def test_oracle_matches_hf_golden():
    from audiotoken_amd.hubert import hubert_processor
    w = W.synth_hubert_weights(int(G["n_layers"]), int(G["weight_seed"]), True)
    wave, mask = _inputs()
    norm = torch.stack([hubert_processor(wave[i:i + 1])[0] for i in range(wave.shape[0])])
    assert np.allclose(norm.numpy(), G["normalized"], atol=1e-6)
    assert np.allclose(R.feature_extractor_normalize(wave).numpy(), G["normalized"], atol=1e-6)
    hs = R.hidden_states(w, torch.from_numpy(G["normalized"]), mask, int(G["n_layers"]), True)
    for k, name in ((0, "hs0"), (1, "hs1"), (int(G["n_layers"]), "hs_last")):
        assert np.abs(hs[k].numpy() - G[name]).max() < 2e-4, name
    toks = R.semantic_s_encode(w, torch.from_numpy(G["normalized"]), mask, int(G["n_layers"]))
    assert toks.dtype == torch.int16 and np.array_equal(toks.numpy(), G["tokens"])
    assert R.num_frames(480000) == 1499 and W.hubert_num_frames(16000) == 49


@pytest.mark.gpu
def test_gpu_matches_golden_and_oracle(cuda_device):
    from audiotoken_amd.configs import HubertEncoderConfig
    from audiotoken_amd.hubert import HubertEncoder
    nl = int(G["n_layers"])
    w = W.synth_hubert_weights(nl, int(G["weight_seed"]), True)
    enc = HubertEncoder(HubertEncoderConfig(output_layer=nl), device="cuda:0", quantize=True, weights=w)
    x, mask = torch.from_numpy(G["normalized"]).cuda(), torch.from_numpy(G["mask"]).cuda()
    for n, key in ((0, "hs0"), (1, "hs1"), (nl, "hs_last")):
        toks, hid = enc(x, mask, n_layers=n, return_hidden=True)
        torch.cuda.synchronize()
        err = np.abs(hid.cpu().numpy() - G[key]).max()
        print(f"hubert hidden_states[{n}] max abs err {err:.3e}")
        assert err < 1e-3, (key, err)
    assert toks.dtype == torch.int16 and np.array_equal(toks.cpu().numpy(), G["tokens"]), "token ids must be bit-identical"


@pytest.mark.gpu
def test_gpu_api_semantic_s(cuda_device):
    from audiotoken_amd import AudioToken, Tokenizers
    w = W.synth_hubert_weights(11, 3, True)
    tok = AudioToken(Tokenizers.semantic_s, device="cuda:0", weights=w)
    wav = W.synth_waveform(1, 16000 * 2, 16000, seed=8)
    out = tok.encode(wav)
    assert out.device.type == "cpu" and out.dtype == torch.int16 and tuple(out.shape) == (1, 1, 99)
    from audiotoken_amd.hubert import hubert_processor
    norm = hubert_processor(torch.from_numpy(wav))
    ref = R.semantic_s_encode(w, norm, torch.ones_like(norm), 11)
    same = (out == ref).float().mean().item()
    print(f"semantic_s tokens equal to oracle: {same:.4f}")
    assert same == 1.0


@pytest.mark.gpu
@pytest.mark.parametrize("N", [1200, 4000, 10640], ids=["T3", "T12", "T33"])
def test_gpu_short_clips_vs_oracle(cuda_device, N):
    """Clips of 3, 12 and 33 frames (below / just above one attention key tile): ids equal the oracle's, or the oracle's own top-2 centre margin
    explains them (tests/parity.py)."""
    from audiotoken_amd.configs import HubertEncoderConfig
    from audiotoken_amd.hubert import HubertEncoder, hubert_processor
    nl = 3
    w = W.synth_hubert_weights(nl, 5, True)
    enc = HubertEncoder(HubertEncoderConfig(output_layer=nl), device="cuda:0", quantize=True, weights=w)
    wav = W.synth_waveform(2, N, 16000, seed=300 + N)
    norm = hubert_processor(torch.from_numpy(wav))
    mask = torch.ones_like(norm)
    toks = enc(norm.cuda(), mask.cuda())
    assert enc.last_status() == 0
    ref = R.semantic_s_encode(w, norm, mask, nl)
    assert toks.shape == ref.shape and toks.shape[-1] == W.hubert_num_frames(N)
    assert torch.equal(toks.cpu(), ref), f"N={N}: {int((toks.cpu() != ref).sum())} ids differ"


@pytest.mark.gpu
@pytest.mark.parametrize("B,N", [(3, 16000 * 6 + 77), (2, 10640), (1, 1200), (2, 82000)], ids=["T300", "T33", "T3", "T256"])
def test_gpu_positional_conv_kernels_agree(cuda_device, B, N):
    """The grouped positional convolution (HF HubertPositionalConvEmbedding: k 128, 16 groups, padding 64, last output dropped) on the LDS-resident split
    kernel (csrc/hubert_posconv.hip, option posconv_split = 1, default) against the 16 windowed fp32-MFMA GEMMs it replaces (= 0): the layer-0 input they
    produce agrees to fp32 rounding, ids are identical; tile seams (T = 300: two 256-row tiles), clips shorter than the 128 taps, ragged masks."""
    from audiotoken_amd.configs import HubertEncoderConfig
    from audiotoken_amd.hubert import HubertEncoder, hubert_processor
    w = W.synth_hubert_weights(2, 17, True)
    enc = HubertEncoder(HubertEncoderConfig(output_layer=2), device="cuda:0", quantize=True, weights=w)
    wav = hubert_processor(torch.from_numpy(W.synth_waveform(B, N, 16000, seed=400 + B)))
    mask = torch.ones_like(wav)
    if B > 1:
        mask[1, N * 2 // 3:] = 0
    x, m = wav.cuda(), mask.cuda()
    assert enc.get_option("posconv_split") == 1
    t1, h1 = enc(x, m, n_layers=0, return_hidden=True)
    t1f = enc(x, m)
    assert enc.last_status() == 0
    enc.set_option("posconv_split", 0)
    t0, h0 = enc(x, m, n_layers=0, return_hidden=True)
    t0f = enc(x, m)
    enc.set_option("posconv_split", 1)
    err = (h1 - h0).abs().max().item()
    print(f"positional conv B={B} T={h1.shape[1]}: max |split - fp32| at the encoder input {err:.2e} (max |h| {h0.abs().max().item():.2f})")
    assert err < 1e-4          # (K = 6 144 products per output at |h| ~ 8: 3e-5 is 4e-6 relative; the contract for float intermediates is 1e-3)
    assert torch.equal(t1f, t0f)
    ref, margins = R.semantic_s_encode(w, wav, mask, 2, return_margins=True)
    from tests import parity as P
    P.assert_tokens_equal_or_explained(t1f, ref, margins, P.VQ_TIE, f"positional conv (split kernel) B={B} N={N}")


@pytest.mark.gpu
def test_gpu_kmeans_score_gemm_kernels_agree(cuda_device):
    """The k-means score GEMM (hidden state . centres^T, 1000 centres) on the split kernel against centres zero-padded to 1024 rows (option kmeans_split = 1,
    default) and on the fp32 MFMA (= 0): identical ids on ragged clips, and the oracle's ids (equal or explained); the 24 padding rows can never win (their
    scores are not read)."""
    from audiotoken_amd.configs import HubertEncoderConfig
    from audiotoken_amd.hubert import HubertEncoder, hubert_processor
    from tests import parity as P
    w = W.synth_hubert_weights(2, 23, True)
    enc = HubertEncoder(HubertEncoderConfig(output_layer=2), device="cuda:0", quantize=True, weights=w)
    wav = hubert_processor(torch.from_numpy(W.synth_waveform(3, 40000, 16000, seed=77)))
    mask = torch.ones_like(wav)
    mask[1, 30000:] = 0
    x, m = wav.cuda(), mask.cuda()
    assert enc.get_option("kmeans_split") == 1
    t1 = enc(x, m)
    assert enc.last_status() == 0
    enc.set_option("kmeans_split", 0)
    t0 = enc(x, m)
    enc.set_option("kmeans_split", 1)
    ref, margins = R.semantic_s_encode(w, wav, mask, 2, return_margins=True)
    P.assert_tokens_equal_or_explained(t1, ref, margins, P.VQ_TIE, "k-means on the split kernel")
    P.assert_tokens_equal_or_explained(t0, ref, margins, P.VQ_TIE, "k-means on the fp32 MFMA")
    assert int(t1.max()) < 1000


@pytest.mark.gpu
@pytest.mark.parametrize("arith", ["f16x2", "bf16x3"])
def test_gpu_layernorm_split_is_bit_identical(cuda_device, arith):
    """Round 5: HuBERT's post-LN LayerNorms write the fp32 residual stream and the next GEMM's operand pieces in ONE pass (option ln_split = 1, default;
    layernorm_split_kernel<.., 768>) instead of LayerNorm + a separate split pass (= 0). Same reductions in the same order, same split: hidden states and
    ids are BIT-identical, on ragged clips, with a layer pinned to the other scheme (the pieces are written in the CONSUMING layer's scheme), at every depth."""
    from audiotoken_amd.configs import HubertEncoderConfig
    from audiotoken_amd.hubert import HubertEncoder, hubert_processor
    w = W.synth_hubert_weights(3, 29, True)
    enc = HubertEncoder(HubertEncoderConfig(output_layer=3), device="cuda:0", quantize=True, weights=w)
    enc.set_option("arith", arith)
    wav = hubert_processor(torch.from_numpy(W.synth_waveform(3, 52000, 16000, seed=81)))
    mask = torch.ones_like(wav)
    mask[2, 31000:] = 0
    x, m = wav.cuda(), mask.cuda()
    for pinned in (None, 1):
        if pinned is not None:
            enc.set_option(f"layer_arith:{pinned}", 1 if arith == "f16x2" else 2)
        for depth in (0, 1, 2, 3):
            assert enc.get_option("ln_split") == 1
            t1, h1 = enc(x, m, n_layers=depth, return_hidden=True)
            assert enc.last_status() == 0
            enc.set_option("ln_split", 0)
            t0, h0 = enc(x, m, n_layers=depth, return_hidden=True)
            enc.set_option("ln_split", 1)
            assert torch.equal(h1, h0), f"{arith}, pinned {pinned}, depth {depth}: hidden states differ by {(h1 - h0).abs().max().item():.2e}"
            assert torch.equal(t1, t0)
