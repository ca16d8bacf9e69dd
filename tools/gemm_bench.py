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
convs = [  # (name, batch, Tin, Cin, k, stride, N, elu)
    ("down0 k4s2 32->64", 32, 240000, 32, 4, 2, 64, 1),
    ("down1 k8s4 64->128", 32, 120000, 64, 8, 4, 128, 1),
    ("down2 k10s5 128->256", 32, 30000, 128, 10, 5, 256, 1),
    ("down3 k16s8 256->512", 32, 6000, 256, 16, 8, 512, 1),
    ("res1 k3 64->32", 32, 120000, 64, 3, 1, 32, 1),
    ("res2 k3 128->64", 32, 30000, 128, 3, 1, 64, 1),
    ("res3 k3 256->128", 32, 6000, 256, 3, 1, 128, 1),
    ("final k7 512->128", 256, 750, 512, 7, 1, 128, 1),
]
if len(sys.argv) > 1 and sys.argv[1] == "full":   # the acoustic bench shapes at the full 256-clip batch, producer-side ELU (PRO_NONE)
    convs = [("down2 k10s5 128->256 B256", 256, 30000, 128, 10, 5, 256, 0), ("down3 k16s8 256->512 B256", 256, 6000, 256, 16, 8, 512, 0),
             ("res3 k3 256->128 B256 elu", 256, 6000, 256, 3, 1, 128, 1), ("final k7 512->128 B256", 256, 750, 512, 7, 1, 128, 0)]
    shapes = [("flat 192000x512x4096", 192000, 512, 4096, 0), ("flat 1536000x256x1280", 1536000, 256, 1280, 0)]
    sys.argv[1] = ""
if len(sys.argv) > 1:
    convs = [c for c in convs if sys.argv[1] in c[0]]
for name, Bc, Tin, Cin, k, st, N, elu in convs:
    X = torch.randn(Bc, Tin, Cin, device=dev)
    K = k * Cin
    Wt = torch.randn(N, K, device=dev) * 0.03
    b = torch.randn(N, device=dev)
    M = -(-Tin // st)
    out = torch.empty(Bc, M, N, device=dev)
    d = _cabi.GemmDesc()
    d.X, d.x_bstride, d.Tin, d.Cin, d.ldx = X.data_ptr(), Tin * Cin, Tin, Cin, Cin
    d.ktaps, d.stride, d.pad_left, d.pad_mode = k, st, k - st, 1
    d.W, d.bias = Wt.data_ptr(), b.data_ptr()
    d.C, d.c_bstride, d.ldc = out.data_ptr(), M * N, N
    d.R, d.r_bstride, d.ldr = 0, 0, N
    d.M, d.N, d.K, d.batch, d.pro, d.epi, d.alpha = M, N, K, Bc, elu, 0, 1.0
    st_ = _cabi.current_stream_handle(dev)
    for _ in range(2):
        _cabi.check(lib.at_op_gemm(C.byref(d), st_), "gemm")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 5
    e0.record()
    for _ in range(n):
        lib.at_op_gemm(C.byref(d), st_)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    tf = 2.0 * Bc * M * N * K / (ms * 1e-3) / 1e12
    gb = 4.0 * Bc * (Tin * Cin + M * N) / (ms * 1e-3) / 1e9
    print(f"{name:34s} {ms:8.3f} ms  {tf:6.1f} TFLOP/s  {gb:7.0f} GB/s", flush=True)
    del X, Wt, out
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
