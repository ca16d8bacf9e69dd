"""Soak test of the persistent-kernel hand-offs: many acoustic encodes of the same batch must give identical tokens and a
clean status word every time (a lost or stale hand-off in lstm_seq_kernel would show up as a differing checksum)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audiotoken_amd import weights as W
from audiotoken_amd.configs import AcousticEncoderConfig, num_codebooks_to_bandwidth
from audiotoken_amd.encoder import AcousticEncoder

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
enc = AcousticEncoder(AcousticEncoderConfig(bandwidth=num_codebooks_to_bandwidth(8)), device="cuda:0",
                      weights=W.synth_encodec_weights(seed=0, with_decoder=False))
sums = set()
t0 = time.time()
for B, N in ((256, 240000), (77, 24000 * 3 + 640)):
    wav = torch.from_numpy(W.synth_waveform(min(B, 16), N, 24000, seed=5)).cuda().repeat((B + 15) // 16, 1)[:B].contiguous()
    wav = wav * torch.linspace(0.5, 1.0, B, device="cuda").unsqueeze(1)
    ref = None
    for i in range(iters):
        codes = enc(wav, None)
        torch.cuda.synchronize()
        assert enc.last_status() == 0, f"hand-off timeout at iteration {i}"
        if ref is None:
            ref = codes.clone()
        else:
            assert torch.equal(ref, codes), f"tokens changed at iteration {i} (B={B})"
    print(f"B={B} N={N}: {iters} identical encodes, checksum {int(ref.long().sum())}", flush=True)
print(f"soak ok in {time.time() - t0:.1f} s")
