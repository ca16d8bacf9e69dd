"""Randomised parity sweep on the GPU box: several weight seeds x waveform seeds, HIP path vs the CPU oracle.
Prints the fraction of identical token ids per tokenizer (expected 1.0; a near-tie flip would show up as < 1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from audiotoken_amd import weights as W

nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
t0 = time.time()

from audiotoken_amd.configs import AcousticEncoderConfig, Wav2VecBertConfig, HubertEncoderConfig
from audiotoken_amd.encoder import AcousticEncoder, Wav2VecBertEncoder
from audiotoken_amd.hubert import HubertEncoder, hubert_processor
from oracle import encodec_ref as RE, w2vbert_ref as RW, hubert_ref as RH

tot = {"acoustic": [0, 0], "semantic_m": [0, 0], "semantic_s": [0, 0]}
dec_err = 0.0
for s in range(nseeds):
    # acoustic: 3 clips x 3 s, 8 codebooks
    w = W.synth_encodec_weights(seed=100 + s, with_decoder=False)
    enc = AcousticEncoder(AcousticEncoderConfig(bandwidth=6), device="cuda:0", weights=w)
    wav = torch.from_numpy(W.synth_waveform(3, 72000 + 320 * s, 24000, seed=500 + s))
    got = enc(wav.cuda(), None).cpu()
    ref = RE.acoustic_encode(w, wav, 8)
    tot["acoustic"][0] += int((got == ref).sum()); tot["acoustic"][1] += ref.numel()
    del enc
    # decoder: the oracle's codes -> waveform, max abs error
    from audiotoken_amd.configs import AcousticDecoderConfig
    from audiotoken_amd.decoder import AcousticDecoder
    wd = W.synth_encodec_weights(seed=100 + s)
    dec = AcousticDecoder(config=AcousticDecoderConfig(bandwidth=6), device="cuda:0", weights=wd)
    codes = torch.randint(0, 1024, (2, 8, 40 + s), dtype=torch.long)
    got_w = dec(codes.cuda()).cpu()
    ref_w = RE.acoustic_decode(wd, codes)
    dec_err = max(dec_err, float((got_w.reshape(-1) - ref_w.reshape(-1)).abs().max()))
    del dec
    # semantic_m: 4 conformer layers, 2 clips x 4 s, one ragged
    w = W.synth_w2vbert_weights(n_layers=4, seed=200 + s, with_vq=True)
    enc = Wav2VecBertEncoder(Wav2VecBertConfig(output_layer=4), device="cuda:0", quantize=True, weights=w)
    wav = torch.from_numpy(W.synth_waveform(2, 64000, 16000, seed=600 + s))
    mask = torch.ones_like(wav); mask[1, 40000 + 1000 * s:] = 0; wav = wav * mask
    got = enc(wav.cuda(), mask.cuda()).cpu()
    wt = {k: torch.from_numpy(v) for k, v in w.items()}
    ref = RW.semantic_m_encode(wt, wav, mask, 2, 4)
    _, am = RW.processor(wav, mask, 2)
    valid = am.bool().unsqueeze(1)
    tot["semantic_m"][0] += int((got == ref)[valid].sum()); tot["semantic_m"][1] += int(valid.sum())
    del enc
    # semantic_s: 3 transformer layers, 2 clips x 3 s
    w = W.synth_hubert_weights(3, 300 + s, True)
    enc = HubertEncoder(HubertEncoderConfig(output_layer=3), device="cuda:0", quantize=True, weights=w)
    wav = torch.from_numpy(W.synth_waveform(2, 48000, 16000, seed=700 + s))
    norm = torch.stack([hubert_processor(wav[i:i + 1])[0] for i in range(2)])
    got = enc(norm.cuda(), torch.ones_like(norm).cuda()).cpu()
    ref = RH.semantic_s_encode(w, norm, torch.ones_like(norm), 3)
    tot["semantic_s"][0] += int((got == ref).sum()); tot["semantic_s"][1] += ref.numel()
    del enc
    print(f"seed {s}: " + ", ".join(f"{k} {v[0]}/{v[1]}" for k, v in tot.items()), flush=True)
print({k: v[0] / max(v[1], 1) for k, v in tot.items()}, f"decoder max abs err {dec_err:.2e}", f"{time.time() - t0:.0f} s")
