"""CPU oracle for the batch-harness bookkeeping (numpy, integer arithmetic). TEST INFRASTRUCTURE ONLY.

Restates reference ``AudioBatchDataset._iter_chunk`` (audiotoken/datasets.py:75-105), ``AudioConfig.length_tokens``
(audiotoken/configs.py:213-218) and ``save_audio_tokens`` (audiotoken/utils.py:199-225). Pinned by
tests/golden/harness_a.npz, produced by driving the reference's own datasets.py / utils.py through stubs
(tests/golden/make_golden.py)."""
from __future__ import annotations

import math
import os
from typing import List, Tuple

import numpy as np


def segments(length: int, sample_rate: int, chunk_size: int, model_token_rate: int) -> List[Tuple[int, int, int, int]]:
    """-> [(start_idx, end_idx, n_valid, length_tokens)] for one waveform of `length` samples."""
    seg = chunk_size * sample_rate
    length_tokens = math.ceil((length / sample_rate) * model_token_rate)
    out = []
    for i in range(0, length, seg):
        n = min(seg, length - i)
        if n < 3200:            # datasets.py:95-97
            continue
        out.append((i, min(i + seg, length), n, length_tokens))
    return out


def segment_arrays(wave: np.ndarray, sample_rate: int, chunk_size: int, pad_token: float = 0.0):
    """wave [L] -> list of (segment[seg], mask[seg])."""
    seg = chunk_size * sample_rate
    out = []
    for i in range(0, wave.shape[0], seg):
        s = wave[i:i + seg]
        if s.shape[0] < 3200:
            continue
        m = np.ones(s.shape[0], dtype=np.float32)
        if s.shape[0] < seg:
            pad = seg - s.shape[0]
            s = np.concatenate([s, np.full(pad, pad_token, dtype=s.dtype)])
            m = np.concatenate([m, np.zeros(pad, dtype=np.float32)])
        out.append((s, m))
    return out


def save_tokens(tokens: np.ndarray, file_name: str, length_tokens: int, root_dir: str) -> str:
    """utils.py:199-225 (stem = basename up to the first dot; trim; append when the file exists)."""
    stem = file_name.split('/')[-1].split('.')[0]
    path = os.path.join(root_dir, f"{stem}.npy")
    t = tokens[:, :length_tokens]
    if os.path.exists(path):
        t = np.hstack([np.load(path), t])
    np.save(path, t)
    return path
