#!/usr/bin/env python3
"""Round 5: where do ids that differ from the oracle on the trained_like weight family come from? For a few clips of the semantic_s bench batch: the HIP
path in each arithmetic (f16x2 / bf16x3 / f32) against the oracle in float32 AND float64, layer by layer — max |LN-normalised hidden difference| per
depth — and the flip counts. If oracle32 vs oracle64 is as far apart as HIP vs oracle64 the network itself amplifies fp32 rounding (no fp32
implementation, the reference's own on another BLAS included, reproduces those ids); if only f16x2 is far, it is the two-piece arithmetic.
    python tools/family_probe.py [semantic_s] [family] [n_clips] [raw]
`raw` undoes the q / k column compensation of the trained_like HuBERT family (weights.synth_hubert_weights): the chaotic variant whose log is
profiles/r05_family_probe_semantic_s_raw_massive.txt."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F

from audiotoken_amd import synthetic as S
from audiotoken_amd import weights as W

which = sys.argv[1] if len(sys.argv) > 1 else "semantic_s"
family = sys.argv[2] if len(sys.argv) > 2 else "trained_like"
n_clips = int(sys.argv[3]) if len(sys.argv) > 3 else 4
torch.set_num_threads(min(32, len(os.sched_getaffinity(0))))
dev = torch.device("cuda:0")

if which == "semantic_s":
    from audiotoken_amd.configs import HubertEncoderConfig
    from audiotoken_amd.hubert import HubertEncoder
    from oracle import hubert_ref as R
    NL, D = 11, 768
    w = W.synth_hubert_weights(NL, 0, True, family=family)
    if len(sys.argv) > 4 and sys.argv[4] == "raw" and family == "trained_like":
        mc = W.massive_channels("hubert", 768, 0)
        for i in range(1, NL):
            g = w[f"encoder.layers.{i - 1}.final_layer_norm.weight"][mc]
            for nm in ("q_proj", "k_proj"):
                w[f"encoder.layers.{i}.attention.{nm}.weight"][:, mc] *= g[None, :]
    full = S.semantic_s_batch(128, 480000, dev)
    wav = full[list(range(0, 128, 8))[:n_clips]].contiguous()
    del full
    enc = HubertEncoder(HubertEncoderConfig(), device="cuda:0", quantize=True, weights=w)
    mask = torch.ones_like(wav)
    w32 = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in w.items()}
    w64 = {k: v.double() for k, v in w32.items()}
    states32 = R.hidden_states(w32, wav.cpu(), mask.cpu(), NL, return_all=True)
    states64 = R.hidden_states(w64, wav.cpu().double(), mask.cpu().double(), NL, return_all=True)
    centers = w32["kmeans.cluster_centers_"]

    def hip_hidden(k):
        return enc(wav, mask, n_layers=k, return_hidden=True)[1].cpu()

    def assign(e):
        return R.kmeans_assign(e, centers.to(e.dtype), return_margin=True)
else:
    from audiotoken_amd.configs import Wav2VecBertConfig
    from audiotoken_amd.encoder import Wav2VecBertEncoder
    from oracle import w2vbert_ref as R
    NL, D = 19, 1024
    secs = 10
    w = W.synth_w2vbert_weights(NL, 0, True, family=family)
    wav = torch.from_numpy(S.speech_like_waveform(n_clips, secs * 16000, 16000, seed=32000)).to(dev)
    enc = Wav2VecBertEncoder(Wav2VecBertConfig(), device="cuda:0", quantize=True, weights=w)
    mask = torch.ones_like(wav)
    w32 = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in w.items()}
    w64 = {k: v.double() for k, v in w32.items()}
    f32_, am = R.processor(wav.cpu(), mask.cpu(), 2)
    f64_, _ = R.processor(wav.cpu().double(), mask.cpu().double(), 2)
    states32 = R.encoder_hidden_state(w32, f32_, am, NL, return_all=True)
    states64 = R.encoder_hidden_state(w64, f64_, am.double(), NL, return_all=True)
    # the fp32 oracle fed the EXACT features: separates the front end's 5e-4 (reference fp32 framing / DFT) from the network's own rounding
    states32x = R.encoder_hidden_state(w32, f64_.float(), am, NL, return_all=True)
    centers = w32["vq._codebook.embed"].reshape(-1, D)

    def hip_hidden(k):
        return enc(wav, mask, n_layers=k, return_taps=True)[1]["hidden"].cpu()

    def assign(e):
        idx, m = R.vq_assign(e.float(), centers, return_margin=True)
        return idx, m


def ln(x):
    return F.layer_norm(x, (D,))


print(f"{which}, {family} weights, {n_clips} clips; per depth: max over positions of max_k |LN(h)_k difference|  (and the 99.9 % quantile)")
extra = which == "semantic_m"
print(f"{'depth':>5} | {'oracle32 vs 64':>22} | " + (f"{'o32(exact feats) vs 64':>22} | " if extra else "") + " | ".join(f"{a + ' vs 64':>22}" for a in ("f16x2", "bf16x3", "f32")))
for k in range(NL + 1):
    ref64 = ln(states64[k])
    row = []
    d = (ln(states32[k]).double() - ref64).abs().amax(-1)
    row.append(f"{float(d.max()):10.2e} {float(d.flatten().quantile(0.999)):10.2e}")
    if extra:
        d = (ln(states32x[k]).double() - ref64).abs().amax(-1)
        row.append(f"{float(d.max()):10.2e} {float(d.flatten().quantile(0.999)):10.2e}")
    for a in ("f16x2", "bf16x3", "f32"):
        enc.set_option("arith", a)
        d = (ln(hip_hidden(k)).double() - ref64).abs().amax(-1)
        row.append(f"{float(d.max()):10.2e} {float(d.flatten().quantile(0.999)):10.2e}")
    print(f"{k:>5} | " + " | ".join(row), flush=True)

idx64, m64 = assign(ln(states64[NL]))
idx32, m32 = assign(ln(states32[NL]))
print(f"ids: oracle32 vs oracle64 differ at {int((idx32 != idx64).sum())} of {idx64.numel()} positions "
      f"({int(((idx32 != idx64) & (m32 >= 1e-3)).sum())} of them with an oracle32 margin >= 1e-3)")
for a in ("f16x2", "bf16x3", "f32"):
    enc.set_option("arith", a)
    toks = enc(wav, mask)[:, 0].cpu().long()
    print(f"ids: HIP {a:7s} vs oracle32: {int((toks != idx32).sum())} differ ({int(((toks != idx32) & (m32 >= 1e-3)).sum())} unexplained); "
          f"vs oracle64: {int((toks != idx64).sum())} differ ({int(((toks != idx64) & (m64 >= 1e-3)).sum())} at an oracle64 margin >= 1e-3); status {enc.last_status()}")
