"""Counter-based PRNG for bit-reproducible synthetic weights and waveforms.

SURVEY.md §7 step 1: synthetic weights must be bit-identical in the build
container and on the GPU box, independent of torch's generator.  Every value is
a pure integer function (SplitMix64 finaliser) of (stream name, element index),
turned into a float32 by an exact 24-bit scaling, so no libm call is involved.
"""
from __future__ import annotations

import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def stream_seed(name: str, seed: int = 0) -> np.uint64:
    """FNV-1a 64-bit hash of the stream name, xor-folded with an integer seed."""
    h = 0xCBF29CE484222325
    for ch in name.encode("utf-8"):
        h ^= ch
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    h ^= (seed * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    return np.uint64(h)


def _mix(z: np.ndarray) -> np.ndarray:
    z = (z ^ (z >> np.uint64(30))) * _M1
    z = (z ^ (z >> np.uint64(27))) * _M2
    return z ^ (z >> np.uint64(31))


def uniform01(name: str, n: int, seed: int = 0) -> np.ndarray:
    """n float32 values in [0, 1), exact multiples of 2^-24."""
    with np.errstate(over="ignore"):
        idx = np.arange(1, n + 1, dtype=np.uint64)
        z = _mix(stream_seed(name, seed) + idx * _GOLDEN)
    return ((z >> np.uint64(40)).astype(np.float32)) * np.float32(2.0 ** -24)


def uniform(name: str, shape, lo: float, hi: float, seed: int = 0) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    u = uniform01(name, n, seed)
    out = np.float32(lo) + (np.float32(hi) - np.float32(lo)) * u
    return out.astype(np.float32).reshape(shape)


def irwin_hall(name: str, shape, std: float, seed: int = 0) -> np.ndarray:
    """Approximately normal(0, std): sum of 4 uniforms, centred and scaled (exact float32 ops)."""
    n = int(np.prod(shape)) if len(shape) else 1
    acc = np.zeros(n, dtype=np.float32)
    for j in range(4):
        acc = acc + uniform01(f"{name}#ih{j}", n, seed)
    # var of sum of 4 U(0,1) = 4/12
    out = (acc - np.float32(2.0)) * np.float32(std * (3.0 ** 0.5))
    return out.astype(np.float32).reshape(shape)


def exp_exact(x: np.ndarray) -> np.ndarray:
    """exp(x) in float64 from IEEE +, * and ldexp only (no libm: the same bits on every host): x = k ln2 + r with |r| <= ln2 / 2, exp(r) by its
    Taylor series to degree 13 (truncation error < 1e-17 relative), Horner form with separately rounded operations."""
    x = np.asarray(x, dtype=np.float64)
    k = np.rint(x * 1.4426950408889634)
    r = (x - k * 0.6931471803691238) - k * 1.9082149292705877e-10     # two-constant Cody-Waite reduction
    p = np.full_like(r, 1.0 / 6227020800.0)
    for c in (479001600.0, 39916800.0, 3628800.0, 362880.0, 40320.0, 5040.0, 720.0, 120.0, 24.0, 6.0, 2.0, 1.0, 1.0):
        p = p * r + 1.0 / c
    return np.ldexp(p, k.astype(np.int64))


def _bits(name: str, n: int, seed: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        idx = np.arange(1, n + 1, dtype=np.uint64)
        return _mix(stream_seed(name, seed) + idx * _GOLDEN)


def heavy_tailed(name: str, shape, std: float, seed: int = 0) -> np.ndarray:
    """Zero mean, standard deviation `std`, power-law tails of index 4 (the tail of Student-t with nu = 4): an approximately normal draw (the sum of
    the three 21-bit fields of one 64-bit word, centred: Irwin-Hall of order 3) times a Pareto scale u^(-1/4) — two IEEE square roots, no libm.
    E[u^(-1/2)] = 2, hence the 1/sqrt(2). The scale is capped at 2^4.5, so the largest entry of a big matrix is ~48 standard deviations out (trained
    transformer weights: 20-50); excess kurtosis ~20."""
    n = int(np.prod(shape)) if len(shape) else 1
    b = _bits(name, n, seed)
    m21 = np.uint64((1 << 21) - 1)
    s3 = ((b & m21) + ((b >> np.uint64(21)) & m21) + ((b >> np.uint64(42)) & m21)).astype(np.float64)
    z = (s3 - 1.5 * ((1 << 21) - 1)) * (2.0 / (1 << 21))                  # variance 3/12 * 4 = 1
    u = np.maximum((_bits(f"{name}#tail", n, seed) >> np.uint64(11)).astype(np.float64) * 2.0 ** -53, 2.0 ** -18)   # scale <= 22.6: entries <= ~48 std
    out = z / np.sqrt(np.sqrt(u)) * (std / np.sqrt(2.0))
    return out.astype(np.float32).reshape(shape)


def log_normal(name: str, shape, sigma: float, seed: int = 0, median: float = 1.0) -> np.ndarray:
    """median * exp(sigma n), n approximately normal(0, 1) (|n| <= 3.46): LayerNorm gains of trained models."""
    n = int(np.prod(shape)) if len(shape) else 1
    z = irwin_hall(name, (n,), 1.0, seed).astype(np.float64)
    return (median * exp_exact(sigma * z)).astype(np.float32).reshape(shape)
