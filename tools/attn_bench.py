"""A/B of the two f16x2 attention kernels on the semantic_m bench shape (64 clips x 1500 frames x 16 heads) through at_op_relpos_attention_kvp:
interleaved rounds in one process (cdna_hip_programming.md rule 24), random data, max difference of the two outputs.
  python tools/attn_bench.py [B T heads]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audiotoken_amd import _cabi

lib = _cabi.load()
dev = torch.device("cuda:0")
B, T, heads = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (64, 1500, 16)
hid = heads * 64
torch.manual_seed(0)
qkv = torch.randn(B * T, 3 * hid, device=dev)
mask = torch.ones(B * T, device=dev)
de = (torch.randn(80, 64, device=dev) * 0.5) if heads == 16 else None
rows_pad = (B * T + 255) // 256 * 256
ws = torch.zeros(4 * rows_pad * hid + 2 * 96 * 64, dtype=torch.float16, device=dev)
st = _cabi.current_stream_handle(dev)
from ctypes import c_void_p
outs = {}
def run(w8, ctx):
    _cabi.check(lib.at_op_relpos_attention_kvp(qkv.data_ptr(), mask.data_ptr(), _cabi.ptr(de), float(de.abs().max()) if de is not None else 0.0, ctx.data_ptr(), B, T, heads, w8, ws.data_ptr(),
                                               ws.numel() * 2, None, st), "attn")
for w8 in (1, 0):
    outs[w8] = torch.empty(B * T, hid, device=dev)
    run(w8, outs[w8])
torch.cuda.synchronize()
print("max |w8 - r3| =", (outs[1] - outs[0]).abs().max().item(), " max |out| =", outs[0].abs().max().item(), flush=True)
# the split pass is part of the op entry (0.2 ms): time it alone through T = 1 rows?  -> simply report both with the same overhead
times = {1: [], 0: []}
for rnd in range(5):
    for w8 in (1, 0):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            run(w8, outs[w8])
        e1.record()
        torch.cuda.synchronize()
        times[w8].append(e0.elapsed_time(e1) / 5)
fl = 4.0 * B * heads * T * T * 64
for w8 in (1, 0):
    t = sorted(times[w8])
    print(f"w8={w8}: median {t[len(t)//2]:.3f} ms  min {t[0]:.3f} ms (incl. the k / v split pass)  -> {fl / (t[len(t)//2] * 1e-3) / 1e12:.0f} TFLOP/s algorithmic", flush=True)
