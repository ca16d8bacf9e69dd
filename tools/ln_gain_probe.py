"""Where does the HIP path leave the oracle when EVERY LayerNorm gain is x8 (softmax logits x64)? Hidden state after 0..3 layers: HIP (f16x2, bf16x3, f32)
vs the fp32 oracle vs the oracle in float64."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from audiotoken_amd import weights as W
from audiotoken_amd.configs import Wav2VecBertConfig
from audiotoken_amd.encoder import Wav2VecBertEncoder
from oracle import w2vbert_ref as R

gain = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
w = dict(W.synth_w2vbert_weights(n_layers=3, seed=9, with_vq=True))
for k in list(w):
    if k.startswith("encoder.layers.") and "layer_norm" in k and "final_layer_norm" not in k:
        w[k] = (w[k] * np.float32(gain)).astype(np.float32)
wav = torch.from_numpy(W.synth_waveform(2, 48000, 16000, seed=41)); mask = torch.ones_like(wav)
wt = {k: torch.from_numpy(v) for k, v in w.items()}
wt64 = {k: v.double() for k, v in wt.items()}
feats, am = R.processor(wav, mask, 2)
h32 = R.encoder_hidden_state(wt, feats, am, 3, return_all=True)
h64 = R.encoder_hidden_state(wt64, feats.double(), am.double(), 3, return_all=True)
enc = Wav2VecBertEncoder(Wav2VecBertConfig(output_layer=3), device="cuda:0", quantize=False, weights=w)
for arith in ("f16x2", "bf16x3", "f32"):
    enc.set_option("arith", arith)
    for nl in range(4):
        h = enc(wav.cuda(), mask.cuda(), n_layers=nl).cpu()
        st = enc.last_status()
        print(f"gain {gain:g} arith {arith} after {nl} layers: HIP vs oracle32 {(h - h32[nl]).abs().max():.3e}  HIP vs f64 {(h.double() - h64[nl]).abs().max():.3e}  "
              f"oracle32 vs f64 {(h32[nl].double() - h64[nl]).abs().max():.3e}  scale {h64[nl].abs().max():.1f} status {st}")
