"""CPU: the EnCodec oracle restatement against the committed golden vectors (made by HF EncodecModel with
the same synthetic weights — tests/golden/make_golden.py)."""
import glob
import os

import numpy as np
import pytest
import torch

from audiotoken_amd import weights as W
from oracle import encodec_ref as R

CASES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "encodec_*.npz")))


@pytest.fixture(scope="module")
def enc_weights():
    return W.synth_encodec_weights(seed=0)


def _edges(x, n=16):
    T = x.shape[-1]
    return x if T <= 2 * n else torch.cat([x[..., :n], x[..., -n:]], dim=-1)


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p) for p in CASES])
def test_oracle_matches_golden(path, enc_weights):
    g = np.load(path)
    B, N, n_q = int(g["B"]), int(g["N"]), int(g["n_q"])
    wav = torch.from_numpy(W.synth_waveform(B, N, 24000, seed=int(g["wave_seed"])))
    emb, stages = R.seanet_encode(enc_weights, wav, return_stages=True)
    # stage taps: oracle stage list index -> HF layer index
    hf_idx = [0, 1, 3, 4, 6, 7, 9, 10, 12, 13]
    for st, li in zip(stages[:-1], hf_idx):
        ref = torch.from_numpy(g[f"stage{li}"])
        assert torch.allclose(_edges(st), ref, atol=2e-5, rtol=1e-5), f"stage {li}"
    assert torch.allclose(emb, torch.from_numpy(g["emb"]), atol=5e-5, rtol=1e-5)
    toks = R.acoustic_encode(enc_weights, wav, n_q)
    assert toks.dtype == torch.int16 and tuple(toks.shape) == (B, n_q, -(-N // 320))
    assert np.array_equal(toks.numpy(), g["tokens"]), "token ids differ from the golden"
    dec = R.acoustic_decode(enc_weights, toks)
    assert tuple(dec.shape) == (1, B * 320 * toks.shape[-1])
    assert np.allclose(dec.numpy(), g["decoded"], atol=1e-4, rtol=1e-4)


def test_bandwidth_to_nq():
    assert [R.bandwidth_to_nq(b) for b in (1.5, 3, 6, 12, 24)] == [2, 4, 8, 16, 32]


def test_reflect_short_input_rule():
    # modeling_encodec.py:139-155: inputs not longer than the pad are zero-extended before reflecting
    x = torch.arange(1.0, 4.0).view(1, 1, 3)
    y = R.pad1d_reflect(x, 6, 0)
    assert y.shape[-1] == 9
    assert torch.equal(y[0, 0, -3:], x[0, 0])
