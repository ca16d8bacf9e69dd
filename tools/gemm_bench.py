"""Micro-benchmark of the windowed fp32 GEMM through the C ABI on the shapes of the two tokenizers."""
import ctypes as C, os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audiotoken_amd import _cabi

lib = _cabi.load()
dev = torch.device("cuda:0")
shapes = [  # (name, M, N, K, epi)
    ("ffn1 96000x4096x1024 swish", 96000, 4096, 1024, 1),
    ("ffn2 96000x1024x4096 res", 96000, 1024, 4096, 0),
    ("qkv 96000x3072x1024", 96000, 3072, 1024, 0),
    ("out 96000x1024x1024", 96000, 1024, 1024, 0),
    ("glu 96000x2048x1024", 96000, 2048, 1024, 5),
    ("lstm_ih 192000x2048x512", 192000, 2048, 512, 0),
]
if len(sys.argv) > 1:
    shapes = [s for s in shapes if sys.argv[1] in s[0]]
res = {}
for name, M, N, K, epi in shapes:
    X = torch.randn(M, K, device=dev)
    Wt = torch.randn(N, K, device=dev) * 0.03
    b = torch.randn(N, device=dev)
    ldc = N // 2 if epi == 5 else N
    out = torch.empty(M, ldc, device=dev)
    d = _cabi.GemmDesc()
    d.X, d.x_bstride, d.Tin, d.Cin, d.ldx = X.data_ptr(), 0, M, K, K
    d.ktaps, d.stride, d.pad_left, d.pad_mode = 1, 1, 0, 0
    d.W, d.bias = Wt.data_ptr(), (0 if epi == 5 else b.data_ptr())
    d.C, d.c_bstride, d.ldc = out.data_ptr(), 0, ldc
    d.R, d.r_bstride, d.ldr = 0, 0, ldc
    d.M, d.N, d.K, d.batch, d.pro, d.epi, d.alpha = M, N, K, 1, 0, epi, 1.0
    st = _cabi.current_stream_handle(dev)
    for _ in range(2):
        _cabi.check(lib.at_op_gemm(C.byref(d), st), "gemm")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 5
    e0.record()
    for _ in range(n):
        lib.at_op_gemm(C.byref(d), st)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12
    res[name] = (round(ms, 3), round(tf, 1))
    print(f"{name:34s} {ms:8.3f} ms  {tf:6.1f} TFLOP/s", flush=True)
    del X, Wt, out
