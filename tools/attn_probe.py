"""Attention accuracy probe (GPU box): at_op_relpos_attention (default arithmetic: f16x2) and the torch-CPU fp32 oracle against a float64 evaluation,
with q and k scaled so that the softmax logits grow by `g^2`. Separates "the HIP attention is less accurate than fp32 at large logits" from "the
network is ill-conditioned there for every fp32 implementation"."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from audiotoken_amd import _cabi, prng, weights as W
from oracle import w2vbert_ref as R

lib = _cabi.load()
w = W.synth_w2vbert_weights(n_layers=1, seed=11, with_vq=False)
p = "encoder.layers.0.self_attn"
B, T = 1, 300
x = torch.from_numpy(prng.irwin_hall("attn.long", (B, T, 1024), 1.0, 2))
mask = torch.ones(B, T)
for g in (1.0, 2.0, 4.0, 8.0):
    wg = dict(w)
    for n in "qk":
        wg[f"{p}.linear_{n}.weight"] = w[f"{p}.linear_{n}.weight"] * g
        wg[f"{p}.linear_{n}.bias"] = w[f"{p}.linear_{n}.bias"] * g
    add = ((1.0 - mask[:, None, None, :]) * torch.finfo(torch.float32).min).expand(B, 1, T, T)
    ref32 = R.relpos_attention(wg, p, x, add)
    w64 = {k: torch.from_numpy(v).double() for k, v in wg.items()}
    ref64 = R.relpos_attention(w64, p, x.double(), add.double())
    wq = torch.cat([torch.from_numpy(wg[f"{p}.linear_{n}.weight"]) for n in "qkv"])
    bq = torch.cat([torch.from_numpy(wg[f"{p}.linear_{n}.bias"]) for n in "qkv"])
    qkv = F.linear(x, wq, bq).reshape(B * T, 3072).cuda().contiguous()
    de = torch.zeros(80, 64); de[:73] = torch.from_numpy(w[f"{p}.distance_embedding.weight"])
    ctx = torch.full((B * T, 1024), float("nan"), device="cuda")
    md, ded = mask.reshape(-1).cuda(), de.cuda()
    _cabi.check(lib.at_op_relpos_attention(qkv.data_ptr(), md.data_ptr(), ded.data_ptr(), ctx.data_ptr(), B, T, _cabi.current_stream_handle(torch.device("cuda:0"))), "attn")
    torch.cuda.synchronize()
    out = F.linear(ctx.cpu().reshape(B, T, 1024), torch.from_numpy(w[f"{p}.linear_out.weight"]), torch.from_numpy(w[f"{p}.linear_out.bias"]))
    q = F.linear(x, torch.from_numpy(wg[f"{p}.linear_q.weight"]), torch.from_numpy(wg[f"{p}.linear_q.bias"]))
    print(f"g={g:g}: |q|max {q.abs().max():.1f}  HIP vs f64 {(out.double() - ref64).abs().max():.3e}   oracle fp32 vs f64 {(ref32.double() - ref64).abs().max():.3e}   (output scale {ref64.abs().max():.2f})")
