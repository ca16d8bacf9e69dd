import sys, torch, numpy as np
sys.path.insert(0, ".")
from audiotoken_amd import weights as W
from audiotoken_amd.configs import AcousticEncoderConfig
from audiotoken_amd.encoder import AcousticEncoder
enc = AcousticEncoder(AcousticEncoderConfig(bandwidth=6), device="cuda:0", weights=W.synth_encodec_weights(seed=0, with_decoder=False))
wav = torch.from_numpy(W.synth_waveform(40, 24000, 24000, seed=3)).cuda()
ref = None
for it in range(6):
    c, e = enc(wav, None, return_embeddings=True)
    torch.cuda.synchronize()
    if ref is None: ref = e.clone(); continue
    d = (e - ref).abs().amax(dim=2)   # [B, T]
    bad = (d > 1e-4).nonzero()
    print(it, "bad frames", len(bad), "clips", sorted(set(bad[:,0].tolist()))[:20], "t range", (bad[:,1].min().item(), bad[:,1].max().item()) if len(bad) else None)
