#!/usr/bin/env python3
"""One-off evidence beyond the driver-run suite (which oracle-checks 18 of 64 / 16 of 128 semantic clips per weight family within its time limit): EVERY clip of the
semantic bench batches against the CPU oracle, on the bar of tests/parity.py (equal, or an oracle top-2 margin < 1e-3).
    python tools/full_batch_oracle.py [semantic_m|semantic_s] [uniform|trained_like]      -> profiles/r05_full_batch_oracle.txt (appended by the caller)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from audiotoken_amd import synthetic as S
from audiotoken_amd import weights as W
from tests import parity as P

which = sys.argv[1] if len(sys.argv) > 1 else "semantic_m"
family = sys.argv[2] if len(sys.argv) > 2 else "uniform"
torch.set_num_threads(min(32, len(os.sched_getaffinity(0))))
dev = torch.device("cuda:0")
t0 = time.time()
if which == "semantic_m":
    from audiotoken_amd.configs import Wav2VecBertConfig
    from audiotoken_amd.encoder import Wav2VecBertEncoder
    from oracle import w2vbert_ref as R
    w = W.synth_w2vbert_weights(n_layers=19, seed=0, with_vq=True, family=family)
    enc = Wav2VecBertEncoder(Wav2VecBertConfig(), device="cuda:0", quantize=True, weights=w)
    wav = S.semantic_m_batch(64, 480000, dev)
    wt = {k: torch.from_numpy(v) for k, v in w.items()}
    oracle = lambda x, m: R.semantic_m_encode(wt, x, m, 2, 19, return_margins=True)
    step = 4
else:
    from audiotoken_amd.configs import HubertEncoderConfig
    from audiotoken_amd.hubert import HubertEncoder
    from oracle import hubert_ref as R
    w = W.synth_hubert_weights(11, 0, True, family=family)
    enc = HubertEncoder(HubertEncoderConfig(), device="cuda:0", quantize=True, weights=w)
    wav = S.semantic_s_batch(128, 480000, dev)
    oracle = lambda x, m: R.semantic_s_encode(w, x, m, 11, return_margins=True)
    step = 8
mask = torch.ones_like(wav)
toks = enc.verified(enc(wav, mask), wav, mask)
assert enc.last_status() == 0
n_ids = n_diff = n_bad = n_tie = 0
for c0 in range(0, wav.shape[0], step):
    ref, m = oracle(wav[c0:c0 + step].cpu(), mask[c0:c0 + step].cpu())
    n, bad, _ = P.explain_token_mismatches(toks[c0:c0 + step], ref, m, P.VQ_TIE)
    n_ids += ref.numel(); n_diff += n; n_bad += bad; n_tie += int((m < P.VQ_TIE).sum())
print(f"[full-batch] {which}, {family} weights: ALL {wav.shape[0]} clips x 30 s of the bench batch: {n_diff} of {n_ids} ids differ from the oracle, {n_bad} unexplained "
      f"({n_tie} positions have an oracle margin < 1e-3); token_checksum {S.token_checksum(toks)} (pinned {S.PINNED_CHECKSUMS[(family, which)]}); "
      f"fallback_batches {enc.fallback_batches}; {time.time() - t0:.0f} s")
assert n_bad == 0
