"""Determinism soak of the benchmark's own batches: every repeat of an encode must reproduce the first one's tokens bit for bit (the role-split / persistent
kernels hand data between waves and workgroups through LDS and memory flags: a dropped or reused hand-off shows up as a moved id), and the status words stay 0.
    python tools/soak.py [acoustic_repeats] [semantic_m_repeats] [semantic_s_repeats] [small-batch encode / decode repeats]
"""
import sys
import time

import torch

sys.path.insert(0, ".")
from audiotoken_amd import synthetic as S
from audiotoken_amd import weights as W
from audiotoken_amd.configs import AcousticEncoderConfig, HubertEncoderConfig, Wav2VecBertConfig
from audiotoken_amd.encoder import AcousticEncoder, Wav2VecBertEncoder
from audiotoken_amd.hubert import HubertEncoder

n_ac = int(sys.argv[1]) if len(sys.argv) > 1 else 200
n_sm = int(sys.argv[2]) if len(sys.argv) > 2 else 40
n_ss = int(sys.argv[3]) if len(sys.argv) > 3 else 40
dev = torch.device("cuda", 0)


def soak(name, enc, call, n, pinned=None):
    ref = call().clone()
    torch.cuda.synchronize()
    cs = S.token_checksum(ref) if ref.dtype == torch.int16 else float(ref.double().abs().sum().item())
    bad = 0
    t0 = time.perf_counter()
    for i in range(n):
        out = call()
        if not torch.equal(out, ref):
            bad += 1
            print(f"{name}: repeat {i} differs in {int((out != ref).sum())} ids")
        if enc.last_status() != 0:
            bad += 1
            print(f"{name}: repeat {i} status {enc.last_status()}")
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{name}: {n} repeats, {bad} bad, checksum {cs}" + (f" (pinned: {cs == pinned})" if pinned is not None else "") + f", {dt / n * 1e3:.1f} ms per repeat incl. compare")
    return bad


bad = 0
enc = AcousticEncoder(AcousticEncoderConfig(bandwidth=6), device="cuda:0", weights=W.synth_encodec_weights(seed=0, with_decoder=False))
wav = S.acoustic_batch(256, 240000, dev, 0)
mask = torch.ones_like(wav)
bad += soak("acoustic 256 x 10 s", enc, lambda: enc(wav, mask), n_ac, S.PINNED_CHECKSUMS[("uniform", "acoustic")])
del enc, wav, mask
enc = Wav2VecBertEncoder(Wav2VecBertConfig(output_layer=19), device="cuda:0", quantize=True, weights=W.synth_w2vbert_weights(n_layers=19, seed=0, with_vq=True))
wav = S.semantic_m_batch(64, 480000, dev, 0)
mask = torch.ones_like(wav)
bad += soak("semantic_m 64 x 30 s", enc, lambda: enc(wav, mask), n_sm, S.PINNED_CHECKSUMS[("uniform", "semantic_m")])
del enc, wav, mask
enc = HubertEncoder(HubertEncoderConfig(output_layer=11), device="cuda:0", quantize=True, weights=W.synth_hubert_weights(n_layers=11, seed=0, with_kmeans=True))
wav = S.semantic_s_batch(128, 480000, dev, 0)
mask = torch.ones_like(wav)
bad += soak("semantic_s 128 x 30 s", enc, lambda: enc(wav, mask), n_ss, S.PINNED_CHECKSUMS[("uniform", "semantic_s")])
del enc, wav, mask
torch.cuda.empty_cache()
# round 4: the pipelined two-layer LSTM launch (three chained hand-off roles; <= 80 clips) in the encoder and the decoder's 64-clip configuration
# (pipelined LSTM, stage-0 split-GEMM chain, fp16-scheme tail kernel)
from audiotoken_amd.configs import AcousticDecoderConfig
from audiotoken_amd.decoder import AcousticDecoder
n_dec = int(sys.argv[4]) if len(sys.argv) > 4 else 200
w = W.synth_encodec_weights(seed=0)
enc = AcousticEncoder(AcousticEncoderConfig(bandwidth=6), device="cuda:0", weights=w)
wav = S.acoustic_batch(64, 240000, dev, 0)
assert enc.get_option("lstm_pipe") == 1
bad += soak("acoustic 64 x 10 s (pipelined LSTM)", enc, lambda: enc(wav, None), n_dec)
codes = enc(wav, None).long()
dec = AcousticDecoder(AcousticDecoderConfig(bandwidth=6), device="cuda:0", weights=w)
bad += soak("acoustic decode 64 x 10 s", dec, lambda: dec(codes), n_dec)
print("soak", "FAILED" if bad else "ok")
sys.exit(1 if bad else 0)
