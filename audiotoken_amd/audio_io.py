"""Audio input side ("next"): WAV reading, mono mix and windowed-sinc resampling.

The reference decodes with torchaudio/ffmpeg (``audiotoken/utils.py:26-101``) — absent here and CPU-side, outside the
measured hot path. This module covers PCM/float WAV through the standard library + scipy and restates
torchaudio's default ``Resample`` (sinc_interp_hann, lowpass_filter_width=6, rolloff=0.99) from its published
algorithm; torchaudio is not installed, so the resampler is UNPINNED against it. Other containers raise.
"""
from __future__ import annotations

import math
import os
import struct
from typing import Iterator, Tuple

import numpy as np
import torch

from .configs import AUDIO_EXTS


class AudioDecodeError(Exception):
    """A file this build cannot turn into a mono waveform: a codec it does not ship (flac / mp3 / ogg need ffmpeg), a damaged or truncated
    header, more than one channel. ``encode_batch_files`` skips such a file, records it in ``AudioToken.skipped_files`` and reports it at
    the end of the run; anything else (a bug in resampling or chunking) propagates like in the reference (audiotoken/datasets.py __iter__)."""


def _load_wav(path_or_file) -> Tuple[torch.Tensor, int]:
    from scipy.io import wavfile
    sr, data = wavfile.read(path_or_file if hasattr(path_or_file, "read") else str(path_or_file))
    if data.dtype == np.int16:
        x = data.astype(np.float32) / 32768.0
    elif data.dtype == np.int32:
        x = data.astype(np.float32) / 2147483648.0
    elif data.dtype == np.uint8:
        x = (data.astype(np.float32) - 128.0) / 128.0
    else:
        x = data.astype(np.float32)
    if x.ndim == 1:
        x = x[:, None]
    return torch.from_numpy(np.ascontiguousarray(x.T)), int(sr)


def load(path, file_stream=None) -> Tuple[torch.Tensor, int]:
    """``path`` names the audio (its extension selects the decoder); ``file_stream`` optionally supplies the bytes
    (members of tar / zip archives)."""
    ext = os.path.splitext(str(path))[1].lower()
    if ext == ".wav":
        return _load_wav(file_stream if file_stream is not None else path)
    if ext in AUDIO_EXTS:
        raise NotImplementedError(f"decoding {ext} needs ffmpeg/torchaudio, which this build does not ship; convert to WAV")
    raise ValueError(f"unsupported audio file {path}")


def resample(wave: torch.Tensor, orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99) -> torch.Tensor:
    """Windowed-sinc polyphase resampling with torchaudio ``Resample`` defaults. wave [C, L] float32."""
    if orig_freq == new_freq:
        return wave
    g = math.gcd(int(orig_freq), int(new_freq))
    o, n = int(orig_freq) // g, int(new_freq) // g
    base = min(o, n) * rolloff
    width = math.ceil(lowpass_filter_width * o / base)
    idx = torch.arange(-width, width + o, dtype=torch.float64)[None, None] / o
    t = torch.arange(0, -n, -1, dtype=torch.float64)[:, None, None] / n + idx
    t = (t * base).clamp(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    scale = base / o
    kernels = torch.where(t == 0, torch.tensor(1.0, dtype=torch.float64), t.sin() / t) * window * scale
    kernels = kernels.to(torch.float32)                      # [n, 1, 2*width + o]
    C, L = wave.shape
    x = torch.nn.functional.pad(wave.reshape(C, 1, L), (width, width + o))
    y = torch.nn.functional.conv1d(x, kernels, stride=o)     # [C, n, frames]
    y = y.transpose(1, 2).reshape(C, -1)
    target = math.ceil(n * L / o)
    return y[..., :target]


def convert_audio(audio: torch.Tensor, sample_rate: int, target_sample_rate: int) -> torch.Tensor:
    """Reference ``convert_audio`` (audiotoken/utils.py:26-44): stereo -> mono mean, >2 channels raises."""
    num_channels = audio.shape[0]
    if num_channels == 2:
        audio = audio.mean(-2, keepdim=True)
    elif num_channels != 1:
        raise RuntimeError("Only mono or stereo audio is supported")
    if sample_rate != target_sample_rate:
        audio = resample(audio, sample_rate, target_sample_rate)
    return audio


def read_audio(x, model_sample_rate: int) -> torch.Tensor:
    """Reference ``read_audio`` (audiotoken/utils.py:47-68): ``[1, num_samples]`` float32 at the model's rate."""
    audio, sr = load(x)
    assert audio.dim() == 2, f"Audio needs to be 2D array, provided {audio.dim()}D for {x}"
    return convert_audio(audio, sr, model_sample_rate)


def process_audio_chunks(file_name, target_sample_rate: int, chunk_size: int, file_stream=None) -> Iterator[Tuple[torch.Tensor, str]]:
    """Reference ``process_audio_chunks`` (audiotoken/utils.py:71-101): ``chunk_size``-second chunks at the SOURCE
    rate, each resampled on its own (so chunk seams follow the reference), yielded as ``([1, n], file_name)``."""
    try:
        audio, sr = load(file_name, file_stream)
    except (NotImplementedError, ValueError, EOFError, OSError, struct.error) as e:   # codec not shipped / unsupported extension / damaged or truncated header / unreadable
        raise AudioDecodeError(f"{file_name}: {type(e).__name__}: {e}") from e
    if audio.shape[0] != 1:
        raise AudioDecodeError(f"Audio needs to be mono, provided {audio.shape[0]} channels for {file_name}")
    step = int(chunk_size * sr)
    for i in range(0, audio.shape[-1], step):
        chunk = audio[:, i:i + step]
        if sr != target_sample_rate:
            chunk = resample(chunk, sr, target_sample_rate)
        yield chunk, str(file_name)


def iterate_tar(x, model_sample_rate: int, chunk_size: int = 30) -> Iterator[Tuple[torch.Tensor, str]]:
    """Reference ``iterate_tar`` (audiotoken/utils.py:139-169): member by member, each through process_audio_chunks."""
    import io
    import tarfile
    with tarfile.open(x, "r") as tar:
        for member in tar.getmembers():
            if not member.isfile():
                continue
            fh = tar.extractfile(member)
            if fh is None:
                continue
            yield from process_audio_chunks(member.name, model_sample_rate, chunk_size, io.BytesIO(fh.read()))


def iterate_zip(x, model_sample_rate: int, chunk_size: int = 30) -> Iterator[Tuple[torch.Tensor, str]]:
    """Reference ``iterate_zip`` (audiotoken/utils.py:104-136)."""
    import io
    import zipfile
    with zipfile.ZipFile(x, "r") as zf:
        for info in zf.infolist():
            if info.is_dir():
                continue
            yield from process_audio_chunks(info.filename, model_sample_rate, chunk_size, io.BytesIO(zf.read(info.filename)))
