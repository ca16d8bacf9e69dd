import os, sys, time
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/bench.py") else os.getcwd())
import torch
from audiotoken_amd import weights as W
from audiotoken_amd.configs import AcousticDecoderConfig, num_codebooks_to_bandwidth
from audiotoken_amd.decoder import AcousticDecoder
dec = AcousticDecoder(config=AcousticDecoderConfig(bandwidth=num_codebooks_to_bandwidth(8)), device="cuda:0", weights=W.synth_encodec_weights(seed=0))
B, T = (int(sys.argv[1]) if len(sys.argv) > 1 else 64), 750
codes = torch.randint(0, 1024, (B, 8, T), dtype=torch.long, device="cuda")
out = dec(codes); torch.cuda.synchronize()
print(out.shape)
t=time.time()
for _ in range(3): out = dec(codes)
torch.cuda.synchronize()
ms=(time.time()-t)/3*1e3
print(f"decode {B}x10s: {ms:.1f} ms -> {B*10/(ms*1e-3):.0f} audio-s/s")
