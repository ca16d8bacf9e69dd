"""GPU parity of the operator-level kernels (through the C ABI) against the CPU oracle / torch-CPU fp32."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from audiotoken_amd import _cabi, prng
from oracle import encodec_ref as R
from tests import parity as P

pytestmark = pytest.mark.gpu


def _rand(name, shape, lo=-1.0, hi=1.0):
    return torch.from_numpy(prng.uniform(name, shape, lo, hi, seed=7))


def run_gemm(dev, X, Wt, bias, *, Tin, Cin, ktaps=1, stride=1, pad_left=0, pad_mode=0, M, N, batch=1, pro=0, epi=0,
             alpha=1.0, R_=None, ldx=None):
    """X: [batch, Tin, Cin] (or strided), Wt: [N, K]."""
    lib = _cabi.load()
    Xd, Wd = X.to(dev).contiguous(), Wt.to(dev).contiguous()
    bd = bias.to(dev).contiguous() if bias is not None else None
    out = torch.full((batch, M, N), float("nan"), device=dev)
    Rd = R_.to(dev).contiguous() if R_ is not None else None
    d = _cabi.GemmDesc()
    d.X, d.x_bstride, d.Tin, d.Cin, d.ldx = Xd.data_ptr(), Xd.stride(0) if batch > 1 else 0, Tin, Cin, ldx or Cin
    d.ktaps, d.stride, d.pad_left, d.pad_mode = ktaps, stride, pad_left, pad_mode
    d.W, d.bias = Wd.data_ptr(), _cabi.ptr(bd)
    d.C, d.c_bstride, d.ldc = out.data_ptr(), M * N, N
    d.R, d.r_bstride, d.ldr = _cabi.ptr(Rd), M * N, N
    d.M, d.N, d.K, d.batch, d.pro, d.epi, d.alpha = M, N, ktaps * Cin, batch, pro, epi, alpha
    _cabi.check(lib.at_op_gemm(C.byref(d), _cabi.current_stream_handle(dev)), "at_op_gemm")
    torch.cuda.synchronize()
    return out.cpu()


def _close(got, ref, tol=2e-5):
    scale = ref.abs().max().item() + 1e-6
    err = (got - ref).abs().max().item()
    assert not torch.isnan(got).any(), "NaN in output (unwritten elements?)"
    assert err <= tol * scale * 8, f"max err {err:.3e} vs scale {scale:.3e}"


@pytest.mark.parametrize("M,N,K,epi,alpha,res", [
    (300, 1024, 160, 0, 1.0, False), (257, 128, 96, 1, 0.5, True), (64, 2048, 512, 0, 1.0, False),
    (1000, 16, 96, 0, 1.0, False), (129, 32, 48, 2, 1.0, True), (513, 64, 64, 3, 1.0, False), (700, 640, 1024, 0, 1.0, False),
])
def test_linear(cuda_device, M, N, K, epi, alpha, res):
    X = _rand("lin.x", (1, M, K))
    Wt = _rand("lin.w", (N, K), -0.1, 0.1)
    b = _rand("lin.b", (N,))
    Rr = _rand("lin.r", (1, M, N)) if res else None
    got = run_gemm(cuda_device, X, Wt, b, Tin=M, Cin=K, M=M, N=N, epi=epi, alpha=alpha, R_=Rr)
    ref = F.linear(X.double(), Wt.double(), b.double())
    if epi == 1:
        ref = F.silu(ref)
    elif epi == 2:
        ref = F.elu(ref)
    elif epi == 3:
        ref = F.gelu(ref)
    ref = ref * alpha
    if res:
        ref = ref + Rr.double()
    _close(got, ref.float())


@pytest.mark.parametrize("B,T,Cin,Cout,k,s,elu", [
    (3, 1000, 32, 16, 3, 1, True), (2, 1001, 32, 64, 4, 2, True), (2, 203, 256, 512, 16, 8, True),
    (1, 77, 512, 128, 7, 1, True), (2, 333, 64, 128, 8, 4, False), (2, 205, 128, 256, 10, 5, True), (2, 500, 16, 32, 1, 1, True),
])
def test_causal_conv(cuda_device, B, T, Cin, Cout, k, s, elu):
    x = _rand("cv.x", (B, T, Cin), -2, 2)            # channels-last
    w = _rand("cv.w", (Cout, Cin, k), -0.2, 0.2)     # torch layout
    b = _rand("cv.b", (Cout,))
    wp = w.permute(0, 2, 1).reshape(Cout, k * Cin)   # [Cout][tap][Cin]
    M = -(-T // s)
    got = run_gemm(cuda_device, x, wp, b, Tin=T, Cin=Cin, ktaps=k, stride=s, pad_left=k - s, pad_mode=1, M=M, N=Cout,
                   batch=B, pro=1 if elu else 0)
    xin = x.permute(0, 2, 1)
    ref = R.conv1d_causal(F.elu(xin) if elu else xin, w, b, s).permute(0, 2, 1)
    assert ref.shape == got.shape
    _close(got, ref)


@pytest.mark.parametrize("B,T,Cin,Cout,s", [(2, 50, 64, 32, 2), (1, 33, 512, 256, 8), (2, 40, 256, 128, 5)])
def test_transposed_conv_as_phase_gemm(cuda_device, B, T, Cin, Cout, s):
    x = _rand("ct.x", (B, T, Cin), -2, 2)
    w = _rand("ct.w", (Cin, Cout, 2 * s), -0.2, 0.2)  # torch ConvTranspose1d layout
    b = _rand("ct.b", (Cout,))
    wp = torch.empty(s, Cout, 2 * Cin)
    for p in range(s):
        wp[p, :, :Cin] = w[:, :, p + s].t()
        wp[p, :, Cin:] = w[:, :, p].t()
    got = run_gemm(cuda_device, x, wp.reshape(s * Cout, 2 * Cin), b.repeat(s), Tin=T, Cin=Cin, ktaps=2, stride=1, pad_left=1,
                   pad_mode=0, M=T, N=s * Cout, batch=B, pro=1)
    got = got.reshape(B, T * s, Cout)
    ref = R.convtr1d_causal(F.elu(x.permute(0, 2, 1)), w, b, s).permute(0, 2, 1)
    assert ref.shape == got.shape
    _close(got, ref)


@pytest.mark.parametrize("kernel", ["f16x2", "bf16x3", "fp32"])
@pytest.mark.parametrize("rows,T,n_q", [(1000, 125, 8), (77, 11, 16), (4096, 512, 2)])
def test_rvq_encode(cuda_device, rows, T, n_q, kernel):
    """The residual-VQ search against the oracle (encodec ResidualVectorQuantizer.encode) on every kernel that can run it:
    "f16x2" = rvq_encode_x3_kernel<SchemeF16x2> — WHAT SHIPS (option rvq_f16x2, default) — "bf16x3" = its range fallback, "fp32" = the fp32-MFMA
    kernel behind at_op_rvq_encode (round 2 tested only that one at op level)."""
    lib = _cabi.load()
    w = {f"quantizer.vq.layers.{q}._codebook.embed": prng.irwin_hall(f"cb{q}", (1024, 128), 1.2 * 0.75 ** q, 3) for q in range(n_q)}
    x = torch.from_numpy(prng.irwin_hall("rvq.x", (rows, 128), 2.0, 3))
    B = rows // T
    emb = x.reshape(B, T, 128).permute(0, 2, 1)
    ref, margins = R.rvq_encode(w, emb, n_q, return_margins=True)       # [n_q, B, T]
    cb = torch.stack([torch.from_numpy(w[f"quantizer.vq.layers.{q}._codebook.embed"]) for q in range(n_q)])
    e2 = torch.stack([c.t().pow(2).sum(0) for c in cb])
    dev = cuda_device
    xd, cbd, e2d = x.to(dev), cb.to(dev).contiguous(), e2.to(dev).contiguous()
    out = torch.full((B, n_q, T), -1, dtype=torch.int16, device=dev)
    if kernel == "fp32":
        _cabi.check(lib.at_op_rvq_encode(xd.data_ptr(), rows, T, cbd.data_ptr(), e2d.data_ptr(), n_q, out.data_ptr(),
                                         _cabi.current_stream_handle(dev)), "at_op_rvq_encode")
    else:
        scheme = 1 if kernel == "f16x2" else 0
        nbytes = (2 if scheme else 3) * n_q * 1024 * 128 * 2 + 8
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        status = torch.zeros(1, dtype=torch.int32, device=dev)
        _cabi.check(lib.at_op_rvq_encode_split(xd.data_ptr(), rows, T, cbd.data_ptr(), e2d.data_ptr(), n_q, out.data_ptr(), scheme, float(cb.abs().max()),
                                               ws.data_ptr(), nbytes, status.data_ptr(), _cabi.current_stream_handle(dev)), "at_op_rvq_encode_split")
        torch.cuda.synchronize()
        assert int(status.item()) == 0
    torch.cuda.synchronize()
    got = out.cpu().permute(1, 0, 2)                                     # [n_q, B, T] like the oracle
    n_ids, n_frames, bad = P.explain_rvq_mismatches(got.permute(1, 0, 2), ref.permute(1, 0, 2), margins.permute(1, 0, 2), P.RVQ_TIE)
    print(f"rvq {kernel} rows={rows} n_q={n_q}: {n_ids} differing ids in {n_frames} frames, {bad} frames not explained by a near-tie; min margin {margins.min().item():.2e}")
    assert bad == 0
    if kernel == "fp32":
        assert n_ids == 0, "the fp32 kernel follows the oracle's operation order: ids must be equal"


@pytest.mark.parametrize("kernel,scheme", [(2, 0), (2, 1), (1, 1)])
@pytest.mark.parametrize("M,N,K", [(200, 256, 1024), (303, 512, 1024), (1500, 1024, 4096), (3000, 4096, 1024)])
def test_split_gemm_vs_float64(cuda_device, M, N, K, kernel, scheme):
    """The split-operand GEMM (bf16x3: kernel 2 / scheme 0; f16x2: register-staged kernel 2 and two-group kernel 1) against numpy float64:
    error not above the k-ordered fp32 FMA chain's scale (a few 1e-7 of the row scale), on ragged M (row padding, several M tiles)."""
    lib = _cabi.load()
    x = prng.irwin_hall(f"sg.x{M}", (M, K), 1.0, 3)
    w = prng.irwin_hall(f"sg.w{N}", (N, K), 0.05, 3)
    b = prng.irwin_hall("sg.b", (N,), 0.5, 3)
    ref = x.astype(np.float64) @ w.astype(np.float64).T + b.astype(np.float64)
    dev = cuda_device
    xd, wd, bd = torch.from_numpy(x).to(dev), torch.from_numpy(w).to(dev), torch.from_numpy(b).to(dev)
    out = torch.full((M, N), float("nan"), dtype=torch.float32, device=dev)
    np_ = 3 if scheme == 0 else 2
    nbytes = ((M + 255) // 256 * 256 + N) * K * np_ * 2
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    _cabi.check(lib.at_op_gemm_split(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), out.data_ptr(), M, N, K, scheme, float(np.abs(w).max()), kernel,
                                     ws.data_ptr(), nbytes, status.data_ptr(), _cabi.current_stream_handle(dev)), "at_op_gemm_split")
    torch.cuda.synchronize()
    assert int(status.item()) == 0
    got = out.cpu().numpy().astype(np.float64)
    err = np.abs(got - ref)
    chain = np.abs((torch.from_numpy(x) @ torch.from_numpy(w).t() + torch.from_numpy(b)).numpy().astype(np.float64) - ref).max()
    print(f"split gemm M={M} N={N} K={K} scheme={scheme} kernel={kernel}: max err {err.max():.2e} (torch fp32 on CPU: {chain:.2e}), ref rms {np.sqrt((ref ** 2).mean()):.2f}")
    assert np.isfinite(got).all()
    assert err.max() <= 1.5e-5 * np.sqrt(K / 1024.0) * max(1.0, np.sqrt((ref ** 2).mean()))


def test_split_gemm_xcd_order_and_soak(cuda_device):
    """A launch large enough for the 256 x 256 shape WITH the XCD-aware tile order (>= 64 m-tiles), against float64 — every tile must be
    produced exactly once — and 25 repeats that must be bit-identical: the two-group kernel's LDS-DMA ring is a hand-written
    producer / consumer protocol, and a landing race shows up as run-to-run differences long before it shows up as a wrong token."""
    lib = _cabi.load()
    M, N, K = 20000, 1024, 1024
    x = prng.irwin_hall("soak.x", (M, K), 1.0, 5)
    w = prng.irwin_hall("soak.w", (N, K), 0.05, 5)
    ref = x.astype(np.float64) @ w.astype(np.float64).T
    dev = cuda_device
    xd, wd = torch.from_numpy(x).to(dev), torch.from_numpy(w).to(dev)
    nbytes = ((M + 255) // 256 * 256 + N) * K * 2 * 2
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    first = None
    for it in range(25):
        out = torch.full((M, N), float("nan"), dtype=torch.float32, device=dev)
        _cabi.check(lib.at_op_gemm_split(xd.data_ptr(), wd.data_ptr(), 0, out.data_ptr(), M, N, K, 1, float(np.abs(w).max()), 0,
                                         ws.data_ptr(), nbytes, status.data_ptr(), _cabi.current_stream_handle(dev)), "at_op_gemm_split")
        torch.cuda.synchronize()
        assert int(status.item()) == 0
        if first is None:
            first = out
            err = np.abs(out.cpu().numpy().astype(np.float64) - ref).max()
            print(f"split gemm (product dispatch) M={M} N={N} K={K}: max err vs float64 {err:.2e}")
            assert err <= 1.5e-5 * max(1.0, np.sqrt((ref ** 2).mean()))
        else:
            assert torch.equal(out, first), f"run {it} differs from run 0"


def _conv_split(dev, x, wp, b, k, s, reps=1):
    """x [B, L, Cin] channels-last, wp [Cout, k * Cin] tap-major -> list of `reps` outputs [B, L / s, Cout] of at_op_conv_split."""
    lib = _cabi.load()
    B, L, Cin = x.shape
    Cout = wp.shape[0]
    M = L // s
    Lp = (M + 255) // 256 * 256 + (k - 1) // s + 1
    nbytes = 2 * (B * Cin * s * Lp + Cout * k * Cin) * 2
    xd, wd, bd = x.to(dev).contiguous(), wp.to(dev).contiguous(), b.to(dev).contiguous()
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    outs = []
    for _ in range(reps):
        out = torch.full((B, M, Cout), float("nan"), dtype=torch.float32, device=dev)
        _cabi.check(lib.at_op_conv_split(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), out.data_ptr(), B, L, Cin, Cout, k, s, float(wp.abs().max()),
                                         ws.data_ptr(), nbytes, status.data_ptr(), _cabi.current_stream_handle(dev)), "at_op_conv_split")
        torch.cuda.synchronize()
        assert int(status.item()) == 0
        outs.append(out)
    return outs


@pytest.mark.parametrize("B,L,Cin,Cout,k,s,tile", [
    (3, 1000, 128, 256, 10, 5, 128), (2, 2048, 256, 512, 16, 8, 128), (2, 777, 512, 128, 7, 1, 128), (5, 640, 256, 128, 3, 1, 128),
    (40, 6000, 128, 256, 10, 5, 128), (64, 12000, 128, 256, 10, 5, 256),
])
def test_conv_split_windowed_vs_oracle_and_soak(cuda_device, B, L, Cin, Cout, k, s, tile):
    """The WINDOWED instantiations of the two-group GEMM (gemm_f16x2_tg_kernel<true, 2, 4> at two workgroups per CU, <true, 4, 8> with the XCD-aware
    persistent tile walk) as a causal conv1d against the CPU oracle's conv (encodec SConv1d), then 25 bit-identical repeats: a landing or refill
    race of the LDS-DMA ring (gemm_f16x2_tg.hip, RAW / WAR) shows up as run-to-run differences long before it shows up as a wrong token."""
    x = torch.from_numpy(prng.uniform(f"cs.x{B}.{L}", (B, L, Cin), -2.0, 2.0, seed=11))
    w = torch.from_numpy(prng.uniform(f"cs.w{Cout}.{k}", (Cout, Cin, k), -0.2, 0.2, seed=11))
    b = torch.from_numpy(prng.uniform("cs.b", (Cout,), -1.0, 1.0, seed=11))
    wp = w.permute(0, 2, 1).reshape(Cout, k * Cin).contiguous()
    reps = 25 if B >= 5 else 3
    outs = _conv_split(cuda_device, x, wp, b, k, s, reps)
    nchk = min(B, 3)
    ref = R.conv1d_causal(x[:nchk].permute(0, 2, 1).double(), w.double(), b.double(), s).permute(0, 2, 1).float()
    got = outs[0][:nchk].cpu()
    assert ref.shape == got.shape
    err = (got - ref).abs().max().item()
    print(f"conv split B={B} L={L} {Cin}->{Cout} k={k} s={s} ({tile}-row tiles): max err {err:.2e} at scale {ref.abs().max().item():.1f}; {reps} repeats")
    assert not torch.isnan(outs[0]).any(), "NaN in the output (unwritten tiles?)"
    assert err <= 2e-5 * max(1.0, ref.abs().max().item())
    for i, o in enumerate(outs[1:], 1):
        assert torch.equal(o, outs[0]), f"run {i} differs from run 0"


def test_split_gemm_small_tile_soak(cuda_device):
    """gemm_f16x2_tg_kernel<false, 2, 4>: the 128 x 128 shape with TWO workgroups per CU (628 tiles on 512 persistent workgroups), the configuration
    in which round 2's shared-DMA variant produced sporadically wrong tiles (root cause and reproducer: tools/lds_dma_war.hip). Against float64, then
    25 bit-identical repeats."""
    lib = _cabi.load()
    M, N, K = 20000, 512, 1024
    x = prng.irwin_hall("soak2.x", (M, K), 1.0, 5)
    w = prng.irwin_hall("soak2.w", (N, K), 0.05, 5)
    ref = x.astype(np.float64) @ w.astype(np.float64).T
    dev = cuda_device
    xd, wd = torch.from_numpy(x).to(dev), torch.from_numpy(w).to(dev)
    nbytes = ((M + 255) // 256 * 256 + N) * K * 2 * 2
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    first = None
    for it in range(25):
        out = torch.full((M, N), float("nan"), dtype=torch.float32, device=dev)
        _cabi.check(lib.at_op_gemm_split(xd.data_ptr(), wd.data_ptr(), 0, out.data_ptr(), M, N, K, 1, float(np.abs(w).max()), 1,
                                         ws.data_ptr(), nbytes, status.data_ptr(), _cabi.current_stream_handle(dev)), "at_op_gemm_split")
        torch.cuda.synchronize()
        assert int(status.item()) == 0
        if first is None:
            first = out
            err = np.abs(out.cpu().numpy().astype(np.float64) - ref).max()
            print(f"split gemm (two-group kernel, 128-row tiles) M={M} N={N} K={K}: max err vs float64 {err:.2e}")
            assert err <= 1.5e-5 * max(1.0, np.sqrt((ref ** 2).mean()))
        else:
            assert torch.equal(out, first), f"run {it} differs from run 0"


# ---- round 5: nearest code with the near-ties re-evaluated exactly ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("D,Cn", [(1024, 2048), (768, 1000)])
def test_vq_argmax_refined_equals_exact_arithmetic(cuda_device, D, Cn):
    """The quantiser of the two semantic tokenizers (reference: vector_quantize_pytorch / torch.cdist + argmin, audiotoken/encoder.py:100-101,180-181) on the
    adversarial case a trained quantiser presents: centres that sit IN the data (rows = other centres + 1 % noise, so the nearest centre is ~0.3 away while
    |x|^2 + |e|^2 ~ 2 D) with three massive channels carrying most of the norm. The expanded fp32 form — even with exact fp64 dot products rounded to fp32, let alone
    a GEMM's accumulation error — mis-ranks near-ties there; the refined kernel must return exactly the argmin of the float64 distances (first index at exact
    ties: duplicated centres), and the plain kernel must agree with it wherever the exact margin is comfortably large."""
    lib = _cabi.load()
    g = torch.Generator().manual_seed(11)
    cb = torch.randn(Cn, D, generator=g) * 0.05
    cb[:, [5, 77, 300]] += torch.randn(Cn, 3, generator=g) * 12.0                      # massive channels: norm^2 ~ 430 + 2.5
    cb[Cn - 8:] = cb[:8]                                                               # exact duplicates: ties go to the LOWER index
    rows = 4096
    x = cb[torch.randint(0, Cn, (rows,), generator=g)] + 0.01 * torch.randn(rows, D, generator=g)
    x64, c64 = x.double(), cb.double()
    d2 = (x64 * x64).sum(-1, keepdim=True) + (c64 * c64).sum(-1)[None] - 2.0 * x64 @ c64.t()
    # first index among the exact minima (a dgemm gives duplicated columns values 1e-13 apart: "equal" = within 1e-9 of the minimum)
    want = ((d2 <= d2.min(-1, keepdim=True).values + 1e-9).float().argmax(-1))
    top2 = (-d2.clamp_min(0).sqrt()).topk(2, dim=-1).values
    margin = top2[:, 0] - top2[:, 1]
    dots = (x64 @ c64.t()).float()                                                     # the best a score GEMM could deliver: exact dots rounded to fp32
    # ... and what it does deliver: an accumulation error of ~3e-4 at |x . e| ~ 400
    noisy = dots + (torch.rand(dots.shape, generator=g) - 0.5) * 6e-4
    e2 = (cb * cb).sum(-1)
    dev = cuda_device
    xd, cd, e2d = x.to(dev), cb.to(dev), e2.to(dev)
    for name, dd in (("exact dots", dots), ("dots with a GEMM-sized error", noisy)):
        out_r = torch.empty(rows, dtype=torch.int16, device=dev)
        out_p = torch.empty(rows, dtype=torch.int16, device=dev)
        dd_d = dd.to(dev).contiguous()
        _cabi.check(lib.at_op_vq_argmax_refined(xd.data_ptr(), dd_d.data_ptr(), e2d.data_ptr(), cd.data_ptr(), out_r.data_ptr(), rows, D, Cn, _cabi.current_stream_handle(dev)), "refined")
        _cabi.check(lib.at_op_vq_argmax(xd.data_ptr(), dd_d.data_ptr(), e2d.data_ptr(), out_p.data_ptr(), rows, D, Cn, _cabi.current_stream_handle(dev)), "plain")
        torch.cuda.synchronize()
        bad_r = int((out_r.cpu().long() != want).sum())
        plain_off = out_p.cpu().long() != want
        print(f"vq_argmax D={D} C={Cn}, {name}: refined differs from the float64 argmin at {bad_r} of {rows} rows; the expanded fp32 form alone at {int(plain_off.sum())} "
              f"({int((plain_off & (margin >= 1e-3)).sum())} of them at an exact margin >= 1e-3); rows with a margin < 1e-2: {int((margin < 1e-2).sum())}")
        assert bad_r == 0
        assert not bool((plain_off & (margin > 0.05)).any())
