"""TEST INFRASTRUCTURE ONLY (imported by tests/; never by audiotoken_amd/): an independent CPU oracle for the resampler of the input side (N3).

The reference resamples every streamed chunk with ``torchaudio.transforms.Resample(orig, new)`` at its defaults (reference audiotoken/utils.py:82-98 and
``convert_audio`` utils.py:26-44): ``resampling_method="sinc_interp_hann"``, ``lowpass_filter_width=6``, ``rolloff=0.99``. torchaudio 2.x is not installed
offline, so its PUBLISHED algorithm (``torchaudio/functional/functional.py``: ``_get_sinc_resample_kernel`` + ``_apply_sinc_resample_kernel``) is restated
here — deliberately NOT as the polyphase table that ``audiotoken_amd/audio_io.resample_table`` builds, and sharing no code with it: every output sample is
evaluated on its own from the continuous-time form of the same filter, with exact integer time arithmetic.

With o = orig / gcd, n = new / gcd, base = min(o, n) * rolloff, L = lowpass_filter_width: output sample j lies at input time tau_j = j * o / n, and

    y[j] = (base / o) * sum_m x[m] * sinc(pi * t) * cos^2(pi * t / (2 L)),      t = base * (m - tau_j) / o = base * (m n - j o) / (n o),  |t| < L

(x[m] = 0 outside the chunk: torchaudio zero-pads ``width`` samples on the left and ``width + o`` on the right; taps with |t| >= L are exactly zero
because torchaudio clamps t to [-L, L] where the Hann window vanishes). torchaudio builds its kernel in float64 and casts it to float32 before the
convolution (``dtype=None`` path); the same rounding of each tap is applied here, the accumulation runs in float64. Output length: ceil(new * len / orig).
Parity status: pinned to the published formula, NOT to a torchaudio run ("parity unpinned" against the package itself — it does not exist offline).
"""
from __future__ import annotations

import math

import numpy as np


def resampled_length(length: int, orig_freq: int, new_freq: int) -> int:
    g = math.gcd(int(orig_freq), int(new_freq))
    return -((-(int(new_freq) // g) * int(length)) // (int(orig_freq) // g))


def sinc_interp_hann(x: np.ndarray, orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99, block: int = 4096) -> np.ndarray:
    """x float [L] -> float64 [ceil(new * L / orig)], each output sample evaluated directly (see the module docstring)."""
    x = np.asarray(x, dtype=np.float64).reshape(-1)
    if int(orig_freq) == int(new_freq):
        return x.copy()
    g = math.gcd(int(orig_freq), int(new_freq))
    o, n = int(orig_freq) // g, int(new_freq) // g
    base = min(o, n) * rolloff
    L = float(lowpass_filter_width)
    half = int(math.ceil(L * o / base)) + 1                       # input samples either side of tau_j that can carry a non-zero tap
    n_out = resampled_length(len(x), orig_freq, new_freq)
    out = np.zeros(n_out, dtype=np.float64)
    d = np.arange(-half, half + 1, dtype=np.int64)[None, :]
    for j0 in range(0, n_out, block):
        j = np.arange(j0, min(n_out, j0 + block), dtype=np.int64)[:, None]
        m = (j * o) // n + d                                      # candidate input indices around tau_j
        num = m * n - j * o                                       # exact: (m - tau_j) * n
        t = base * num.astype(np.float64) / float(n * o)
        inside = np.abs(t) < L
        tc = np.clip(t, -L, L)
        window = np.cos(tc * math.pi / L / 2.0) ** 2
        pt = tc * math.pi
        with np.errstate(invalid="ignore", divide="ignore"):
            sinc = np.where(pt == 0.0, 1.0, np.sin(pt) / pt)
        tap = (sinc * window * (base / o)).astype(np.float32).astype(np.float64)   # torchaudio casts its float64 kernel to float32
        tap = np.where(inside, tap, 0.0)
        ok = (m >= 0) & (m < len(x))
        xm = np.where(ok, x[np.clip(m, 0, len(x) - 1)], 0.0)
        out[j0:j0 + len(j)] = (xm * tap).sum(axis=1)
    return out
