#!/usr/bin/env python3
"""Round 6, P.V one-piece experiment: hidden-state error at FULL depth (19 conformer layers) against the fp32 oracle and its float64 evaluation, per weight family —
run once with the product library and once with AUDIOTOKEN_HIP_LIB=tools/_lib_p1.so (attention_f16x2_w8.hip built with -DW8_P_PIECES=1).
    python tools/p1_depth.py [n_clips]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from audiotoken_amd import synthetic as S
from audiotoken_amd import weights as W
from audiotoken_amd.configs import Wav2VecBertConfig
from audiotoken_amd.encoder import Wav2VecBertEncoder
from oracle import w2vbert_ref as R

n_clips = int(sys.argv[1]) if len(sys.argv) > 1 else 2
torch.set_num_threads(16)
tag = "P1 (one-piece probabilities)" if "p1" in os.environ.get("AUDIOTOKEN_HIP_LIB", "") else "product (two pieces)"
for fam in ("uniform", "trained_like"):
    w = W.synth_w2vbert_weights(n_layers=19, seed=0, with_vq=True, family=fam)
    wav = torch.from_numpy(S.speech_like_waveform(n_clips, 160000, 16000, seed=32000))
    mask = torch.ones_like(wav)
    enc = Wav2VecBertEncoder(Wav2VecBertConfig(), device="cuda:0", quantize=True, weights=w)
    toks, taps = enc(wav.cuda(), mask.cuda(), return_taps=True)
    assert enc.last_status() == 0
    ln = lambda h: torch.nn.functional.layer_norm(h, (1024,))
    hid = ln(taps["hidden"].cpu())
    w32 = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in w.items()}
    f32_, am = R.processor(wav, mask, 2)
    h32 = ln(R.encoder_hidden_state(w32, f32_, am, 19))
    w64 = {k: v.double() for k, v in w32.items()}
    f64_, am64 = R.processor(wav.double(), mask.double(), 2)
    h64 = ln(R.encoder_hidden_state(w64, f64_, am64, 19)).float()
    ref, margins = R.semantic_m_encode(w32, wav, mask, 2, 19, return_margins=True)
    d = toks.cpu().reshape(-1).long() != ref.reshape(-1).long()
    print(f"[p1-depth] {tag}, {fam} weights, {n_clips} x 10 s, 19 layers: max |LN(h) - LN(h oracle32)| {float((hid - h32).abs().max()):.3e} (99.9 % quantile "
          f"{float((hid - h32).abs().flatten().kthvalue(int(0.999 * hid.numel())).values):.3e}); vs oracle64 {float((hid - h64).abs().max()):.3e}; oracle32 vs oracle64 "
          f"{float((h32 - h64).abs().max()):.3e}; ids differ {int(d.sum())} of {d.numel()}, {int((d & (margins.reshape(-1) >= 1e-3)).sum())} at an oracle margin >= 1e-3", flush=True)
    del enc
