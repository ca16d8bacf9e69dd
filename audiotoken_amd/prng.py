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
