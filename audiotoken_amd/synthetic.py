"""The synthetic batches of the headline benchmark (seeded waveforms, clip i depends only on seed 1234 + i), shared by
``bench.py`` and ``tests/test_fullsize_gpu.py`` so that the batch the driver-run parity tests oracle-check IS the batch the bench times:
a few host-generated clips repeated along the batch and scaled by ``linspace(0.5, 1.0, B)`` on the device (distinct rows without
regenerating 256 x 10 s on the host). Rank r of a multi-GPU run owns clips [r B, (r + 1) B) of the global batch."""
from __future__ import annotations

import numpy as np
import torch

from . import weights as W

# token_checksum (sum of all ids as int64) of each batch at rank 0 with the synthetic seed-0 weights, as produced by the library's DEFAULT
# arithmetic; pinned so that a change of arithmetic cannot move ids unnoticed (round 1 -> 2 moved the acoustic one 775616966 -> 775618559).
# tests/test_fullsize_gpu.py asserts them; bench.py reports `checksum_pinned`.
PINNED_CHECKSUMS = {"acoustic": 775618559, "semantic_m": 49212128, "semantic_s": 51528128}


def _repeat_scaled(base: torch.Tensor, B: int) -> torch.Tensor:
    gen_B = base.shape[0]
    wav = base.repeat((B + gen_B - 1) // gen_B, 1)[:B].contiguous()
    if B > gen_B:
        wav = (wav * torch.linspace(0.5, 1.0, B, device=base.device).unsqueeze(1)).contiguous()
    return wav


def acoustic_batch(B: int, N: int, dev, rank: int = 0) -> torch.Tensor:
    """BASELINE configs[1] at B = 256, N = 240 000: ``float32 [B, N]`` @24 kHz on `dev`."""
    base = torch.from_numpy(W.synth_waveform(min(B, 16), N, 24000, seed=1234, first_clip=rank * B)).to(dev)
    return _repeat_scaled(base, B)


def semantic_m_batch(B: int, N: int, dev, rank: int = 0) -> torch.Tensor:
    """BASELINE configs[3] per-GPU share at B = 64, N = 480 000: ``float32 [B, N]`` @16 kHz on `dev` (mask: all ones)."""
    base = torch.from_numpy(W.synth_waveform(min(B, 8), N, 16000, seed=1234, first_clip=rank * B)).to(dev)
    return _repeat_scaled(base, B)


def semantic_s_batch(B: int, N: int, dev, rank: int = 0) -> torch.Tensor:
    """BASELINE configs[2] at B = 128, N = 480 000: clips normalised per clip by ``hubert_processor`` (the reference's host-side transform,
    encoder.py:20-26), then repeated — NOT rescaled (the transform fixes the scale)."""
    from .hubert import hubert_processor
    gen_B = min(B, 8)
    host = W.synth_waveform(gen_B, N, 16000, seed=1234, first_clip=rank * B)
    host = np.stack([hubert_processor(torch.from_numpy(host[i:i + 1]))[0].numpy() for i in range(gen_B)])
    return torch.from_numpy(host).to(dev).repeat((B + gen_B - 1) // gen_B, 1)[:B].contiguous()


def token_checksum(tokens: torch.Tensor) -> int:
    return int(tokens.to(torch.int64).sum().item())
