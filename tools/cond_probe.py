#!/usr/bin/env python3
"""Round 5: WHY does the HIP path flip more ids than the fp32 oracle on the trained_like conformer with a data-fitted code book, although its hidden states are
closer to float64 in the max norm? Per position at depth 19: the L2 norm of the LayerNorm-normalised difference to the float64 oracle, split into the massive
channels and the rest, for the fp32 oracle (its own fp32 front end), the fp32 oracle fed exact features, and the HIP arithmetics."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
from audiotoken_amd import weights as W
from tests.test_fullsize_gpu import fitted_semantic_m
from oracle import w2vbert_ref as R
torch.set_num_threads(min(32, len(os.sched_getaffinity(0))))
enc, w, wav = fitted_semantic_m("trained_like", n_test=2)
mask = torch.ones_like(wav)
w32 = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in w.items()}
w64 = {k: v.double() for k, v in w32.items()}
f32_, am = R.processor(wav, mask, 2)
f64_, _ = R.processor(wav.double(), mask.double(), 2)
ln = lambda x: F.layer_norm(x, (1024,))
h64 = ln(R.encoder_hidden_state(w64, f64_, am.double(), 19))
h32 = ln(R.encoder_hidden_state(w32, f32_, am, 19)).double()
h32x = ln(R.encoder_hidden_state(w32, f64_.float(), am, 19)).double()
mc = W.massive_channels("w2vbert", 1024, 0)
rest = np.setdiff1d(np.arange(1024), mc)
cb = w32["vq._codebook.embed"].reshape(-1, 1024)
idx64, m64 = R.vq_assign(h64.float(), cb, return_margin=True)
print("massive channels", mc, "| typical |LN(h)| there", float(h64[..., mc].abs().mean()), "| elsewhere", float(h64[..., rest].abs().mean()))
def report(name, h):
    d = h - h64
    l2, l2m, l2r = d.norm(dim=-1), d[..., mc].norm(dim=-1), d[..., rest].norm(dim=-1)
    idx, _ = R.vq_assign(h.float(), cb, return_margin=True)
    fl = idx != idx64
    print(f"{name:28s} L2 diff to float64: median {float(l2.median()):.2e} max {float(l2.max()):.2e} | massive-channel part median {float(l2m.median()):.2e} | "
          f"rest median {float(l2r.median()):.2e} | max |elem| {float(d.abs().max()):.2e} | ids (same CPU quantiser) differ {int(fl.sum())} of {fl.numel()}, "
          f"{int((fl & (m64 >= 1e-3)).sum())} at a margin >= 1e-3")
report("fp32 oracle (own front end)", h32)
report("fp32 oracle, exact features", h32x)
for a in ("f16x2", "bf16x3", "f32"):
    enc.set_option("arith", a)
    toks, taps = enc(wav.cuda(), mask.cuda(), return_taps=True)
    report("HIP " + a, ln(taps["hidden"].cpu()).double())
    fl = toks[:, 0].cpu().long() != idx64
    print(f"{'':28s} HIP's own quantiser on its hidden states: differ from float64 ids {int(fl.sum())}, {int((fl & (m64 >= 1e-3)).sum())} at a margin >= 1e-3")
