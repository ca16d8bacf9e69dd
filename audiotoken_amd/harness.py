"""Batch harness semantics of ``encode_batch_files``: fixed-length segmentation + mask, short-tail skip, zero
padding, per-row trim to ``length_tokens`` and the append-on-save ``.npy`` format.

Mirrors reference ``audiotoken/datasets.py:18-20,75-105`` (``collate_fn``, ``AudioBatchDataset._iter_chunk``) and
``audiotoken/utils.py:199-225,342-353,367-396`` (``save_audio_tokens``, ``sanitize_path``, ``save_rel_audio_tokens``),
including the quirks listed in SURVEY.md Appendix B (5, 6, 12, 13). CPU-side bookkeeping; never on the device.
"""
from __future__ import annotations

import os
from copy import deepcopy
from pathlib import Path
from typing import Callable, Iterator, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from .configs import AudioConfig
from .logger import get_logger

logger = get_logger(__name__)

MIN_SEGMENT_SAMPLES = 3200  # datasets.py:95-97 — "0.2 s" at 16 kHz, applied at every sample rate


def collate_fn(batch):
    """datasets.py:18-20."""
    segments, attention_masks, file_names = zip(*batch)
    return torch.stack(segments), torch.stack(attention_masks), file_names


def iter_chunk(waveform: torch.Tensor, file_name: str, *, sample_rate: int, chunk_size: int, model_token_rate: int,
               pad_token: Optional[int] = 0, transform: Optional[Callable] = None
               ) -> Iterator[Tuple[torch.Tensor, torch.Tensor, AudioConfig]]:
    """``AudioBatchDataset._iter_chunk`` (datasets.py:75-105). ``waveform`` is ``[1, L]``.

    Yields ``(segment[segment_length], mask[segment_length], AudioConfig)``; every segment cut from this call
    carries the length of the WHOLE waveform handed in."""
    segment_length = chunk_size * sample_rate
    stride = int(segment_length)
    length = waveform.shape[-1]
    if transform:
        waveform = transform(waveform)
    audio_config = AudioConfig(file_name=file_name, length_seconds=length / sample_rate, length_samples=length,
                               model_token_rate=model_token_rate)
    for i in range(0, length, stride):
        segment = waveform[0, i:i + segment_length]
        attention_mask = torch.ones(segment.shape[0])
        audio_config.start_idx = i
        audio_config.end_idx = min(i + segment_length, length)
        if segment.shape[-1] < MIN_SEGMENT_SAMPLES:
            logger.warning(f'File segment {i // sample_rate} of {file_name} is too short. Skipping')
            continue
        if segment.shape[0] < segment_length:
            pad = segment_length - segment.shape[0]
            attention_mask = F.pad(attention_mask, (0, pad), value=0)
            segment = F.pad(segment, (0, pad), value=pad_token)
        yield segment, attention_mask, deepcopy(audio_config)


def sanitize_path(path) -> str:
    """utils.py:342-353: expand ~, absolutise, resolve, mkdir -p."""
    path = Path(path).expanduser()
    if not path.is_absolute():
        path = path.absolute()
    path = path.resolve()
    if not path.exists():
        path.mkdir(parents=True, exist_ok=True)
    return str(path)


def _append_npy(save_path: str, tokens: np.ndarray) -> None:
    if os.path.exists(save_path):   # reference appends: re-running on the same files duplicates tokens (README:89-90)
        prev = np.load(save_path)
        np.save(save_path, np.hstack([prev, tokens]))
    else:
        np.save(save_path, tokens)


def save_audio_tokens(tokens: torch.Tensor, audio_pointer: AudioConfig, root_dir: str) -> None:
    """utils.py:199-225: ``<stem>.npy`` with stem = basename up to the FIRST dot; trim to length_tokens; append if the
    file exists; errors are logged and swallowed."""
    try:
        filename = audio_pointer.file_name.split('/')[-1].split('.')[0]
        save_path = os.path.join(root_dir, f'{filename}.npy')
        arr = tokens.cpu().numpy()
        arr = arr[:, :audio_pointer.length_tokens]
        _append_npy(save_path, arr)
    except Exception as e:  # noqa: BLE001 — reference behaviour
        logger.error(f'Error saving tokens for {audio_pointer.file_name} with error {e}')


def save_rel_audio_tokens(tokens: torch.Tensor, audio_pointer: AudioConfig, root_dir: str, rel_dir: str) -> None:
    """utils.py:367-396: as above but keeps the directory tree relative to ``rel_dir`` and strips only the last
    extension."""
    try:
        arr = tokens.cpu().numpy()
        arr = arr[:, :audio_pointer.length_tokens]
        rel_path = os.path.dirname(os.path.relpath(audio_pointer.file_name, start=rel_dir))
        if rel_path.startswith("..") or os.path.isabs(rel_path):
            # DELIBERATE deviation (round 5): a name that is not under rel_dir — a member of a tar / zip found by the directory scan carries the member's own
            # relative name — makes the reference's os.path.join(root_dir, "../../..") write OUTSIDE outdir, relative to the current directory (utils.py:374-376).
            # Such token files go to outdir itself here.
            logger.warning(f'{audio_pointer.file_name} is not under {rel_dir}: its tokens are saved in {root_dir} (the reference would write outside it)')
            rel_path = ""
        output_path = os.path.join(root_dir, rel_path)
        os.makedirs(output_path, exist_ok=True)
        filename = os.path.splitext(os.path.basename(audio_pointer.file_name))[0]
        _append_npy(os.path.join(output_path, f'{filename}.npy'), arr)
    except Exception as e:  # noqa: BLE001
        logger.error(f'Error saving tokens for {audio_pointer.file_name} with error {e}')


def batched(items: Iterator, batch_size: int) -> Iterator[List]:
    buf: List = []
    for it in items:
        buf.append(it)
        if len(buf) == batch_size:
            yield buf
            buf = []
    if buf:
        yield buf
