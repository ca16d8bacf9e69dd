"""CPU: the semantic_m oracle restatement against golden vectors produced by the reference's own files
(processors.py, modeling_wav2vec2_bert.py) and HF Wav2Vec2BertModel (tests/golden/make_golden.py)."""
import glob
import os

import numpy as np
import pytest
import torch

from audiotoken_amd import prng
from audiotoken_amd import weights as W
from oracle import w2vbert_ref as R

G = os.path.join(os.path.dirname(__file__), "golden")
FBANK = sorted(glob.glob(os.path.join(G, "fbank_*.npz")))


@pytest.mark.parametrize("path", FBANK, ids=[os.path.basename(p) for p in FBANK])
def test_processor_matches_reference(path):
    g = np.load(path)
    f, m = R.processor(torch.from_numpy(g["wave"]), torch.from_numpy(g["mask"]), int(g["pad_to_multiple_of"]))
    assert tuple(f.shape) == g["input_features"].shape
    assert np.array_equal(m.numpy(), g["attention_mask"])
    # same op order as the reference -> bit-identical on the same torch build; allow fp noise across builds
    assert np.allclose(f.numpy(), g["input_features"], atol=2e-5, rtol=0)


def test_token_count_rule():
    # T' = pad2(floor((1 + floor((N-400)/160)) / 2)); 30 s -> 1500 with 1499 valid
    for n, t in ((480000, 1500), (160000, 500), (3200, 10)):
        nf = R.num_frames(n)
        tp = nf // 2
        tp += tp % 2
        assert tp == t


def test_attention_matches_reference():
    g = np.load(os.path.join(G, "attention_a.npz"))
    w = W.synth_w2vbert_weights(n_layers=1, seed=int(g["weight_seed"]), with_vq=False)
    B, T = int(g["B"]), int(g["T"])
    x = torch.from_numpy(prng.irwin_hall("attn.x", (B, T, 1024), 1.0, int(g["x_seed"])))
    mask = torch.from_numpy(g["mask"])
    add = ((1.0 - mask[:, None, None, :]) * torch.finfo(torch.float32).min).expand(B, 1, T, T)
    out = R.relpos_attention(w, "encoder.layers.0.self_attn", x, add)
    assert np.allclose(out.numpy(), g["out"], atol=2e-5, rtol=1e-5)


def test_conformer_matches_hf():
    g = np.load(os.path.join(G, "conformer_a.npz"))
    n_layers = int(g["n_layers"])
    w = W.synth_w2vbert_weights(n_layers=n_layers, seed=int(g["weight_seed"]), with_vq=True)
    B, N = int(g["B"]), int(g["N"])
    wave = W.synth_waveform(B, N, 16000, seed=int(g["wave_seed"]))
    mask = g["mask"]
    wave = wave * mask
    feats, am = R.processor(torch.from_numpy(wave), torch.from_numpy(mask), 2)
    assert np.array_equal(am.numpy(), g["attention_mask"])
    hs = R.encoder_hidden_state(w, feats, am, n_layers, return_all=True)
    for k, name in ((0, "hs0"), (1, "hs1"), (n_layers, "hs_last")):
        err = np.abs(hs[k].numpy() - g[name]).max()
        assert err < 5e-4, (name, err)
    e = R.layer_norm(hs[n_layers], w, None, 1024)
    embed = torch.from_numpy(w["vq._codebook.embed"][0])
    idx, margin = R.vq_assign(e, embed, return_margin=True)
    # vector_quantize_pytorch's sqrt(clamp(x2+y2-2xy)) argmax == torch.cdist argmin except at genuine near-ties
    diff = idx.numpy() != g["tokens_cdist"].astype(np.int64)
    assert not (diff & (margin.numpy() > 1e-4)).any()
    toks = R.semantic_m_encode(w, torch.from_numpy(wave), torch.from_numpy(mask), 2, n_layers)
    assert toks.dtype == torch.int16 and tuple(toks.shape) == (B, 1, feats.shape[1])
    assert np.array_equal(toks[:, 0].numpy().astype(np.int64), idx.numpy())
