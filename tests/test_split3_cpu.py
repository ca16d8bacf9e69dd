"""Arithmetic of the split-bf16 kernels, checked on the host with torch's bfloat16 (round-to-nearest-even, the
rounding `v_cvt_pk_bf16_f32` applies): the 3-way split is EXACT, and the six leading cross products reproduce a product to 2^-24."""
import torch


def split3(v: torch.Tensor):
    p1 = v.to(torch.bfloat16)
    r1 = v - p1.float()
    p2 = r1.to(torch.bfloat16)
    p3 = (r1 - p2.float()).to(torch.bfloat16)
    return p1, p2, p3


def _samples(n, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(n, generator=g) * torch.exp(torch.randn(n, generator=g) * 3.0)


def test_three_bf16_pieces_reconstruct_fp32_exactly():
    v = _samples(2_000_000, 0)
    p1, p2, p3 = split3(v)
    assert torch.equal(p1.double() + p2.double() + p3.double(), v.double())
    # each piece is at most half an ulp of the previous one
    assert float((p2.float().abs() / v.abs()).max()) <= 2.0 ** -8
    assert float((p3.float().abs() / v.abs()).max()) <= 2.0 ** -16


def test_six_cross_products_match_the_product_to_fp32_precision():
    a, w = _samples(1_000_000, 1), _samples(1_000_000, 2)
    a1, a2, a3 = (p.double() for p in split3(a))
    w1, w2, w3 = (p.double() for p in split3(w))
    six = a1 * w3 + a3 * w1 + a2 * w2 + a1 * w2 + a2 * w1 + a1 * w1     # the order the kernels accumulate in: smallest first
    exact = a.double() * w.double()
    rel = ((six - exact).abs() / exact.abs()).max()
    assert float(rel) <= 2.0 ** -24, float(rel)     # the three dropped terms a2*w3, a3*w2, a3*w3
