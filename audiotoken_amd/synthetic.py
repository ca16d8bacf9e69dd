"""The synthetic batches of the headline benchmark (seeded waveforms, clip i depends only on seed 1234 + i), shared by
``bench.py`` and ``tests/test_fullsize_gpu.py`` so that the batch the driver-run parity tests oracle-check IS the batch the bench times.
Since round 5 every clip of a batch is distinct and speech-like (``speech_like_waveform``; rounds 1-4: 8-16 stationary 4-sine clips repeated
and rescaled — still available as ``kind="sines"``). Rank r of a multi-GPU run owns clips [r B, (r + 1) B) of the global batch."""
from __future__ import annotations

import numpy as np
import torch

from . import weights as W

# token_checksum (sum of all ids as int64) of each batch at rank 0 with the synthetic seed-0 weights, as produced by the library's DEFAULT
# arithmetic; pinned so that a change of arithmetic cannot move ids unnoticed (round 1 -> 2 moved the acoustic one 775616966 -> 775618559).
# tests/test_fullsize_gpu.py asserts them; bench.py reports `checksum_pinned`.
# Keyed by (weight family, tokenizer); the batches are the kind="speech" ones at the BASELINE sizes (256 x 10 s, 64 x 30 s, 128 x 30 s), rank 0.
# (Rounds 2-4 pinned 775618559 / 49212128 / 51528128 for the repeated 4-sine batches; re-pinned ONCE in round 5 when the clips became distinct.)
PINNED_CHECKSUMS = {
    ("uniform", "acoustic"): 779361555, ("uniform", "semantic_m"): 85477913, ("uniform", "semantic_s"): 130741104,
    ("trained_like", "acoustic"): 742693889, ("trained_like", "semantic_m"): 82653249, ("trained_like", "semantic_s"): 134409645,
}


# ---- speech-like clips (round 5) -----------------------------------------------------------------------------------------------------------------
# What the stationary 4-sine clips of rounds 1-4 (weights.synth_waveform) lack and real corpora have: onsets and decays, pauses at the noise floor,
# gliding pitch under a formant envelope, syllable-rate amplitude modulation, a clipped stretch, 40 dB of level spread between clips. Clip i is a pure
# function of (seed + i) — every clip of a batch is distinct — and of IEEE arithmetic plus numpy's float64 sin / cos (as
# synth_waveform). Gains go through prng.exp_exact.
def _speech_clip(n: int, sr: int, s: int) -> np.ndarray:
    from . import prng
    U = prng.uniform01("speech.events", 4096, s).astype(np.float64)
    ui = iter(U)
    nxt = lambda lo, hi: lo + (hi - lo) * next(ui)
    x = np.zeros(n, dtype=np.float64)
    noise = prng.irwin_hall("speech.noise", (n,), 1.0, s).astype(np.float64)
    level_db = -40.0 * nxt(0.0, 1.0)                                   # 40 dB of level spread between clips
    formants = (nxt(300, 900), nxt(900, 2500), nxt(2500, 3500))
    fm, fm_ph = nxt(3.0, 8.0), nxt(0.0, 2 * np.pi)                     # syllable-rate amplitude modulation
    f0 = nxt(80.0, 300.0)
    pos, n_voiced, clipped_at = int(nxt(0.0, 0.3) * sr), 0, int(nxt(1.0, 4.0))
    while pos < n:
        kind = next(ui)
        if kind < 0.25:                                                # pause: 100-800 ms of noise floor
            pos += int(nxt(0.1, 0.8) * sr)
            continue
        dur = int((nxt(0.04, 0.15) if kind < 0.4 else nxt(0.15, 0.6)) * sr)
        end = min(n, pos + dur)
        L = end - pos
        if L < 16:
            break
        t = np.arange(L, dtype=np.float64) / sr
        att, rel = max(1, int(0.005 * sr)), max(1, int(0.03 * sr))     # 5 ms onset, 30 ms decay
        env = np.minimum(1.0, np.arange(L) / att) * np.minimum(1.0, (L - np.arange(L)) / rel)
        env = env * (0.6 + 0.4 * np.sin(2 * np.pi * fm * (pos / sr + t) + fm_ph))
        if kind < 0.4:                                                 # unvoiced burst: differenced (high-passed) noise
            seg = np.diff(noise[pos:end], prepend=0.0) * 0.25
        else:                                                          # voiced: harmonic stack, f0 glide, formant envelope
            f0 = min(320.0, max(70.0, f0 * nxt(0.8, 1.25)))
            f1 = f0 * nxt(0.7, 1.4)
            ph = 2 * np.pi * (f0 * t + 0.5 * (f1 - f0) / max(L / sr, 1e-3) * t * t) + nxt(0.0, 2 * np.pi)
            H = int(min(24, 0.45 * sr / max(f0, f1)))
            c2, s_prev, s_cur = 2.0 * np.cos(ph), np.zeros(L), np.sin(ph)
            seg = np.zeros(L)
            fmid = 0.5 * (f0 + f1)
            for h in range(1, H + 1):                                  # sin((h + 1) ph) = 2 cos(ph) sin(h ph) - sin((h - 1) ph)
                fh = h * fmid
                a = sum(1.0 / (1.0 + ((fh - F) / (80.0 + 0.06 * F)) ** 2) for F in formants) + 0.15 / h
                seg += a * s_cur
                s_prev, s_cur = s_cur, c2 * s_cur - s_prev
            seg *= 0.35
            n_voiced += 1
            if n_voiced == clipped_at:                                 # one over-driven, hard-clipped stretch per clip
                seg = np.clip(4.0 * seg, -1.0, 1.0)
        x[pos:end] += env * seg
        pos = end
    gain = float(prng.exp_exact(np.array([level_db * 0.11512925464970229]))[0])     # 10^(dB / 20)
    x = np.clip(x, -1.0, 1.0) * gain + 1e-3 * noise                    # noise floor at -60 dBFS under everything
    return np.clip(x, -1.0, 1.0).astype(np.float32)


def speech_like_waveform(n_clips: int, n_samples: int, sample_rate: int, seed: int = 1234, first_clip: int = 0) -> np.ndarray:
    """``float32 [n_clips, n_samples]`` in [-1, 1]; clip i depends only on ``seed + first_clip + i`` (generated on a thread pool)."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    try:
        nt = len(os.sched_getaffinity(0))
    except AttributeError:  # pragma: no cover
        nt = os.cpu_count() or 1
    out = np.empty((n_clips, n_samples), dtype=np.float32)

    def one(i):
        out[i] = _speech_clip(n_samples, sample_rate, seed + first_clip + i)

    with ThreadPoolExecutor(max_workers=max(1, min(16, nt))) as ex:
        list(ex.map(one, range(n_clips)))
    return out


def _clips(B: int, N: int, sr: int, rank: int, kind: str) -> np.ndarray:
    if kind == "speech":
        return speech_like_waveform(B, N, sr, seed=1234, first_clip=rank * B)
    if kind == "sines":
        return W.synth_waveform(B, N, sr, seed=1234, first_clip=rank * B)
    raise ValueError(f"clip kind {kind!r}: 'speech' or 'sines'")


def acoustic_batch(B: int, N: int, dev, rank: int = 0, kind: str = "speech") -> torch.Tensor:
    """BASELINE configs[1] at B = 256, N = 240 000: ``float32 [B, N]`` @24 kHz on `dev`, B distinct clips."""
    return torch.from_numpy(_clips(B, N, 24000, rank, kind)).to(dev)


def semantic_m_batch(B: int, N: int, dev, rank: int = 0, kind: str = "speech") -> torch.Tensor:
    """BASELINE configs[3] per-GPU share at B = 64, N = 480 000: ``float32 [B, N]`` @16 kHz on `dev`, B distinct clips (mask: all ones)."""
    return torch.from_numpy(_clips(B, N, 16000, rank, kind)).to(dev)


def semantic_s_batch(B: int, N: int, dev, rank: int = 0, kind: str = "speech") -> torch.Tensor:
    """BASELINE configs[2] at B = 128, N = 480 000: B distinct clips, each normalised by ``hubert_processor`` (the reference's host-side
    transform, encoder.py:20-26)."""
    from .hubert import hubert_processor
    host = _clips(B, N, 16000, rank, kind)
    for i in range(B):
        host[i] = hubert_processor(torch.from_numpy(host[i:i + 1]))[0].numpy()
    return torch.from_numpy(host).to(dev)


def token_checksum(tokens: torch.Tensor) -> int:
    return int(tokens.to(torch.int64).sum().item())
