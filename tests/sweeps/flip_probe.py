"""Explain id mismatches vs the oracle: for the frames where ids differ, print the oracle's top-2 distance margin at the first
differing codebook level (a margin at fp32 rounding level = a legitimate near-tie) and the embedding error there."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from audiotoken_amd import weights as W
from audiotoken_amd.configs import AcousticEncoderConfig
from audiotoken_amd.encoder import AcousticEncoder
from oracle import encodec_ref as RE

s = int(sys.argv[1]) if len(sys.argv) > 1 else 2
w = W.synth_encodec_weights(seed=100 + s, with_decoder=False)
enc = AcousticEncoder(AcousticEncoderConfig(bandwidth=6), device="cuda:0", weights=w)
wav = torch.from_numpy(W.synth_waveform(3, 72000 + 320 * s, 24000, seed=500 + s))
got, emb = enc(wav.cuda(), None, return_embeddings=True)
got, emb = got.cpu(), emb.cpu()
emb_ref = RE.seanet_encode(w, wav)                      # [B,128,T]
ref, margins = RE.rvq_encode(w, emb_ref, 8, return_margins=True)
ref, margins = ref.transpose(0, 1).to(torch.int16), margins.transpose(0, 1)      # [B, n_q, T]
print("emb max abs err", (emb - emb_ref.permute(0, 2, 1)).abs().max().item())
bad = (got != ref).nonzero()
frames = sorted({(int(b), int(t)) for b, _, t in bad})
for b, t in frames:
    lv = int((got[b, :, t] != ref[b, :, t]).nonzero()[0])
    print(f"clip {b} frame {t}: first differing level {lv}, oracle ids {ref[b, :, t].tolist()} gpu {got[b, :, t].tolist()}, "
          f"oracle top-2 margin at that level {float(margins[b, lv, t]):.3e}")
