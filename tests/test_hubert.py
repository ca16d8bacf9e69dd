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
