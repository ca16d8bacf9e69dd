"""Debug aid: compare an encoder option on/off and report where the embeddings differ."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from audiotoken_amd import weights as W
from audiotoken_amd.configs import AcousticEncoderConfig
from audiotoken_amd.encoder import AcousticEncoder
opt = sys.argv[1] if len(sys.argv) > 1 else "res128_x3"
w = W.synth_encodec_weights(seed=0)
enc = AcousticEncoder(AcousticEncoderConfig(bandwidth=6), device="cuda:0", weights=w)
for B, N in ((1, 320 * 8), (1, 320 * 40), (2, 24000), (3, 6400)):
    wav = torch.from_numpy(W.synth_waveform(B, N, 24000, seed=3)).cuda()
    enc.set_option(opt, 1); c1, e1 = enc(wav, None, return_embeddings=True); e1 = e1.clone()
    enc.set_option(opt, 0); c0, e0 = enc(wav, None, return_embeddings=True)
    bad = torch.isnan(e1).any(dim=1) if e1.dim() == 3 else torch.isnan(e1)
    d = (e0 - e1).abs()
    print(B, N, "shape", tuple(e1.shape), "nan frames", int(torch.isnan(e1).sum()), "max diff", float(torch.nan_to_num(d, nan=-1).max()),
          "first nan idx", torch.nonzero(torch.isnan(e1))[:3].tolist())
