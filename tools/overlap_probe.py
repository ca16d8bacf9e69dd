"""Probe: does a GEMM on a second stream make progress inside the persistent LSTM kernel's MFMA gaps?
Runs the acoustic encode alone, a GEMM loop alone, and both concurrently; prints wall times."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audiotoken_amd import _cabi, weights as W
from audiotoken_amd.configs import AcousticEncoderConfig, num_codebooks_to_bandwidth
from audiotoken_amd.encoder import AcousticEncoder

lib = _cabi.load()
dev = torch.device("cuda:0")
enc = AcousticEncoder(AcousticEncoderConfig(bandwidth=num_codebooks_to_bandwidth(8)), device="cuda:0",
                      weights=W.synth_encodec_weights(seed=0, with_decoder=False))
B, N = 256, 240000
wav = torch.from_numpy(W.synth_waveform(16, N, 24000, seed=5)).cuda().repeat(16, 1).contiguous()
enc(wav, None); torch.cuda.synchronize()

M, Nn, K = 192000, 2048, 512
X = torch.randn(M, K, device=dev); Wt = torch.randn(Nn, K, device=dev) * 0.03; out = torch.empty(M, Nn, device=dev)
d = _cabi.GemmDesc()
d.X, d.x_bstride, d.Tin, d.Cin, d.ldx = X.data_ptr(), 0, M, K, K
d.ktaps, d.stride, d.pad_left, d.pad_mode = 1, 1, 0, 0
d.W, d.bias = Wt.data_ptr(), 0
d.C, d.c_bstride, d.ldc = out.data_ptr(), 0, Nn
d.R, d.r_bstride, d.ldr = 0, 0, Nn
d.M, d.N, d.K, d.batch, d.pro, d.epi, d.alpha = M, Nn, K, 1, 0, 0, 1.0
side = torch.cuda.Stream()
NG = int(sys.argv[1]) if len(sys.argv) > 1 else 8

def gemms():
    with torch.cuda.stream(side):
        h = _cabi.current_stream_handle(dev)
        for _ in range(NG):
            lib.at_op_gemm(C.byref(d), h)

def timed(fn):
    torch.cuda.synchronize(); t = time.time(); fn(); torch.cuda.synchronize(); return (time.time() - t) * 1e3

gemms(); torch.cuda.synchronize()
enc.enable_profile(True)
ta = min(timed(lambda: enc(wav, None)) for _ in range(3))
tg = min(timed(gemms) for _ in range(3))
def both():
    gemms(); enc(wav, None)
tb = min(timed(both) for _ in range(3))
enc.enable_profile(True); both(); torch.cuda.synchronize()
prof = enc.read_profile()
print(f"encode alone {ta:.2f} ms, {NG} GEMMs alone {tg:.2f} ms, together {tb:.2f} ms (sum {ta + tg:.2f})")
print({k: round(v[0], 2) for k, v in prof.items()})
