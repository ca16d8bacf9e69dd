"""semantic_m counterpart of flip_probe.py: for the tokens where the HIP path and the oracle differ in parity_sweep's configuration,
print the oracle's top-2 VQ distance margin and the hidden-state error at that token."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from audiotoken_amd import weights as W
from audiotoken_amd.configs import Wav2VecBertConfig
from audiotoken_amd.encoder import Wav2VecBertEncoder
from oracle import w2vbert_ref as RW

for s in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    w = W.synth_w2vbert_weights(n_layers=4, seed=200 + s, with_vq=True)
    wt = {k: torch.from_numpy(v) for k, v in w.items()}
    enc = Wav2VecBertEncoder(Wav2VecBertConfig(output_layer=4), device="cuda:0", quantize=True, weights=w)
    wav = torch.from_numpy(W.synth_waveform(2, 64000, 16000, seed=600 + s))
    mask = torch.ones_like(wav); mask[1, 40000 + 1000 * s:] = 0; wav = wav * mask
    toks, taps = enc(wav.cuda(), mask.cuda(), 2, n_layers=4, return_taps=True)
    feats, am = RW.processor(wav, mask, 2)
    h = RW.encoder_hidden_state(wt, feats, am, 4)
    e = RW.layer_norm(h, wt, None, RW.HIDDEN)
    embed = wt["vq._codebook.embed"].reshape(-1, 1024)
    herr = (taps["hidden"].cpu() - h).abs()
    for b in range(2):
        d = -torch.cdist(e[b], embed)
        top2 = d.topk(2, dim=-1)
        ref, got = top2.indices[:, 0], toks.cpu()[b, 0].long()
        for t in ((got != ref) & am[b].bool()).nonzero().flatten().tolist():
            print(f"seed {s} clip {b} token {t}: oracle {int(ref[t])} gpu {int(got[t])}, oracle top-2 margin {float(top2.values[t, 0] - top2.values[t, 1]):.3e} "
                  f"at distance {float(-top2.values[t, 0]):.2f}, hidden err at token {float(herr[b, t].max()):.2e} (max over all {float(herr.max()):.2e})")
    del enc
print("done")
