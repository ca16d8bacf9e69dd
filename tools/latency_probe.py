"""Single-clip latency of AudioToken.encode() (the reference's interactive path, core.py:120-196) for the three tokenizers."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from audiotoken_amd import AudioToken, Tokenizers, weights as W

for name, sr, secs, kw in (("acoustic", 24000, 10, dict(num_codebooks=8, weights=W.synth_encodec_weights(seed=0, with_decoder=False))),
                           ("semantic_s", 16000, 30, dict(weights=W.synth_hubert_weights(11, 0, True))),
                           ("semantic_m", 16000, 30, dict(weights=W.synth_w2vbert_weights(n_layers=19, seed=0, with_vq=True)))):
    tok = AudioToken(Tokenizers(name), device="cuda:0", **kw)
    wav = W.synth_waveform(1, sr * secs, sr, seed=3)
    tok.encode(wav); torch.cuda.synchronize()
    t = []
    for _ in range(5):
        t0 = time.perf_counter(); out = tok.encode(wav); t.append((time.perf_counter() - t0) * 1e3)
    print(f"{name}: {secs} s clip -> tokens {tuple(out.shape)} in {min(t):.1f} ms (median {sorted(t)[2]:.1f}) = {secs / (min(t) * 1e-3):.0f}x real time", flush=True)
    del tok
    torch.cuda.empty_cache()
