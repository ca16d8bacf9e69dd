"""GPU parity of the acoustic tokenizer (AcousticEncoder through the C ABI) against the CPU oracle and the
committed HF-generated golden vectors."""
import glob
import os

import numpy as np
import pytest
import torch

from audiotoken_amd import weights as W
from oracle import encodec_ref as R

pytestmark = pytest.mark.gpu
CASES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "encodec_*.npz")))


@pytest.fixture(scope="module")
def enc_weights():
    return W.synth_encodec_weights(seed=0)


@pytest.fixture(scope="module")
def encoders(cuda_device, enc_weights):
    from audiotoken_amd.configs import AcousticEncoderConfig
    from audiotoken_amd.encoder import AcousticEncoder
    return {nq: AcousticEncoder(AcousticEncoderConfig(bandwidth=bw), device="cuda:0", weights=enc_weights)
            for nq, bw in ((2, 1.5), (4, 3), (8, 6), (16, 12))}


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p) for p in CASES])
def test_encode_matches_golden(path, encoders):
    g = np.load(path)
    B, N, n_q = int(g["B"]), int(g["N"]), int(g["n_q"])
    wav = torch.from_numpy(W.synth_waveform(B, N, 24000, seed=int(g["wave_seed"])))
    enc = encoders[n_q]
    codes, emb = enc(wav.cuda(), torch.ones_like(wav).cuda(), return_embeddings=True)
    torch.cuda.synchronize()
    assert codes.dtype == torch.int16 and tuple(codes.shape) == (B, n_q, -(-N // 320))
    emb_ref = torch.from_numpy(g["emb"]).permute(0, 2, 1)  # golden is [B,128,T]
    err = (emb.cpu() - emb_ref).abs().max().item()
    print(f"{os.path.basename(path)}: emb max abs err {err:.3e} (|emb| max {emb_ref.abs().max().item():.2f})")
    assert err < 1e-3, "float intermediates must stay within 1e-3 of the fp32 reference"
    assert np.array_equal(codes.cpu().numpy(), g["tokens"]), "token ids must be bit-identical"


def test_encode_matches_oracle_ragged_batch(encoders, enc_weights):
    # N not a multiple of 320 (right 'extra' reflect padding) and more clips than one 64-frame RVQ tile
    B, N, n_q = 5, 12345, 8
    wav = torch.from_numpy(W.synth_waveform(B, N, 24000, seed=99))
    codes, emb = encoders[n_q](wav.cuda(), None, return_embeddings=True)
    emb_ref = R.seanet_encode(enc_weights, wav).permute(0, 2, 1)
    err = (emb.cpu() - emb_ref).abs().max().item()
    assert err < 1e-3, err
    ref = R.rvq_encode(enc_weights, emb_ref.permute(0, 2, 1), n_q).transpose(0, 1).to(torch.int16)
    assert torch.equal(codes.cpu(), ref)


def test_mask_is_ignored(encoders):
    # reference AcousticEncoder.forward never reads attention_mask (audiotoken/encoder.py:44-52)
    wav = torch.from_numpy(W.synth_waveform(2, 6400, 24000, seed=5)).cuda()
    a = encoders[8](wav, torch.ones_like(wav))
    b = encoders[8](wav, torch.zeros_like(wav))
    assert torch.equal(a, b)
