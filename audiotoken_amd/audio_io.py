"""Audio input side: container decoding, mono check and windowed-sinc resampling.

The reference decodes with torchaudio / ffmpeg (``audiotoken/utils.py:26-101``) — absent here. This module covers
  * WAV (8 / 16 / 24 / 32-bit PCM, 32-bit float) through scipy,
  * FLAC through the library's own host decoder (``csrc/flac_decode.hip``, RFC 9639; MD5 of the decoded samples verified here),
both also as tar / zip members, and restates torchaudio's default ``Resample`` (sinc_interp_hann, lowpass_filter_width=6, rolloff=0.99) from its
published algorithm. torchaudio is not installed, so the resampler is UNPINNED against it (pinned against a float64 evaluation of the same formula:
tests/test_io_loaders_cpu.py). '.mp3' / '.ogg' / '.opus' (lossy codecs: a decoder each) raise ``AudioDecodeError``; ``encode_batch_files`` records and
skips such files.

Two consumers: the HOST path (``process_audio_chunks``: float32 tensors, per-chunk resampling on the CPU, exactly the reference's data flow) and the
DEVICE path (``feeder.py``: ``decode_raw`` hands over the file's samples in their storage format; conversion, resampling and segmentation run in one HIP
kernel with the table ``resample_table`` builds).
"""
from __future__ import annotations

import ctypes as C
import functools
import hashlib
import math
import os
import struct
from dataclasses import dataclass
from typing import Iterator, Optional, Tuple

import numpy as np

from .configs import AUDIO_EXTS


class AudioDecodeError(Exception):
    """A file this build cannot turn into a mono waveform: a codec it does not ship (mp3 / ogg / opus need a lossy decoder), a damaged or truncated
    file, more than one channel. ``encode_batch_files`` skips such a file, records it in ``AudioToken.skipped_files`` and reports it at
    the end of the run; anything else (a bug in resampling or chunking) propagates like in the reference (audiotoken/datasets.py __iter__)."""


@dataclass
class RawAudio:
    """A decoded file in its storage format: ``pcm`` [channels, samples] (int16 / int32 / uint8 / float32), float value = (pcm - offset) * scale."""
    pcm: np.ndarray
    sample_rate: int
    scale: float
    offset: float = 0.0
    pinned: object = None     # (device feeder) the page-locked tensor `pcm` is a view of, when the file was read straight into one

    def to_float(self):
        import torch
        x = self.pcm.astype(np.float32)
        if self.offset:
            x = x - np.float32(self.offset)
        if self.pcm.dtype != np.float32:
            x = x * np.float32(self.scale)
        return torch.from_numpy(np.ascontiguousarray(x))


def wav_probe(path):
    """Header of a plain RIFF / WAVE file whose ``data`` chunk can go to the device exactly as it lies in the file (the device feeder reads it straight into
    pinned memory: feeder.py): ``(numpy dtype, sample_rate, data_offset, data_bytes, scale, offset)`` — or None for everything else (more than one channel,
    24-bit samples, WAVE_FORMAT_EXTENSIBLE with another sub-format, RF64 / RIFX, a data chunk that runs past the end of the file ...), which the general
    reader (`decode_raw`, scipy) handles or rejects with its own messages."""
    try:
        size = os.path.getsize(path)
        with open(path, "rb") as f:
            head = f.read(12)
            if len(head) < 12 or head[:4] != b"RIFF" or head[8:12] != b"WAVE":
                return None
            fmt = None
            pos = 12
            while pos + 8 <= size:
                f.seek(pos)
                hdr = f.read(8)
                if len(hdr) < 8:
                    return None
                cid, csize = hdr[:4], struct.unpack("<I", hdr[4:])[0]
                if cid == b"fmt ":
                    body = f.read(min(csize, 40))
                    if len(body) < 16:
                        return None
                    tag, ch, sr, _, align, bits = struct.unpack("<HHIIHH", body[:16])
                    if tag == 0xFFFE and len(body) >= 26:          # WAVE_FORMAT_EXTENSIBLE: the real tag opens the sub-format GUID
                        tag = struct.unpack("<H", body[24:26])[0]
                    fmt = (tag, ch, sr, align, bits)
                elif cid == b"data":
                    if fmt is None:
                        return None
                    tag, ch, sr, align, bits = fmt
                    kinds = {(1, 16): (np.dtype("<i2"), 1.0 / 32768.0, 0.0), (1, 32): (np.dtype("<i4"), 1.0 / 2147483648.0, 0.0),
                             (1, 8): (np.dtype(np.uint8), 1.0 / 128.0, 128.0), (3, 32): (np.dtype("<f4"), 1.0, 0.0)}
                    if ch != 1 or (tag, bits) not in kinds or align != bits // 8 or sr <= 0:
                        return None
                    dtype, scale, offset = kinds[(tag, bits)]
                    if csize == 0 or pos + 8 + csize > size or csize % dtype.itemsize:
                        return None
                    return dtype, int(sr), pos + 8, int(csize), scale, offset
                pos += 8 + csize + (csize & 1)
    except (OSError, struct.error):
        return None
    return None


def _wav_raw(path_or_file) -> RawAudio:
    from scipy.io import wavfile
    sr, data = wavfile.read(path_or_file if hasattr(path_or_file, "read") else str(path_or_file))
    if data.ndim == 1:
        data = data[:, None]
    pcm = np.ascontiguousarray(data.T)
    if pcm.dtype == np.int16:
        return RawAudio(pcm, int(sr), 1.0 / 32768.0)
    if pcm.dtype == np.int32:                       # 24-bit samples arrive left-justified in int32 (scipy), 32-bit as they are
        return RawAudio(pcm, int(sr), 1.0 / 2147483648.0)
    if pcm.dtype == np.uint8:
        return RawAudio(pcm, int(sr), 1.0 / 128.0, 128.0)
    return RawAudio(pcm.astype(np.float32, copy=False), int(sr), 1.0)


def _flac_raw(data: bytes, name) -> RawAudio:
    from . import _cabi
    lib = _cabi.load()
    sr, ch, bits, total = C.c_int(), C.c_int(), C.c_int(), C.c_int64()
    md5 = (C.c_uint8 * 16)()
    buf = np.frombuffer(data, dtype=np.uint8)
    if lib.at_flac_info(buf.ctypes.data, len(data), C.byref(sr), C.byref(ch), C.byref(bits), C.byref(total), md5) != 0:
        raise AudioDecodeError(f"{name}: {_cabi.last_error()}")
    if total.value <= 0:
        raise AudioDecodeError(f"{name}: FLAC stream without a sample count in STREAMINFO (streamed encodes are not supported)")
    # STREAMINFO is untrusted input (36-bit sample count, 8 channels). A CONSTANT sub-frame codes a whole block of up to 65 535 samples in ~10 bytes, so the sample count a file of len(data) bytes can hold is bounded by 65 535 samples per 10 bytes, not by the
    # header's word; anything beyond that, and an allocation the host refuses, is a damaged file to skip — not a MemoryError that aborts the whole run
    if total.value > (len(data) // 10 + 1) * 65535 or not (1 <= ch.value <= 8):
        raise AudioDecodeError(f"{name}: STREAMINFO claims {total.value} samples x {ch.value} channels in a {len(data)}-byte file")
    try:
        out = np.empty((ch.value, total.value), dtype=np.int32)
    except (MemoryError, OverflowError, ValueError) as e:
        raise AudioDecodeError(f"{name}: cannot hold {total.value} samples x {ch.value} channels ({type(e).__name__})") from e
    n = lib.at_flac_decode(buf.ctypes.data, len(data), out.ctypes.data, total.value)
    if n != total.value:
        raise AudioDecodeError(f"{name}: {_cabi.last_error()}")
    want = bytes(md5)
    if any(want):                                   # an all-zero MD5 means "not computed" (RFC 9639 section 8.2)
        nb = (bits.value + 7) // 8
        inter = np.ascontiguousarray(out.T)
        if nb == 2:
            raw = inter.astype("<i2").tobytes()
        elif nb == 4:
            raw = inter.astype("<i4").tobytes()
        elif nb == 1:
            raw = inter.astype("i1").tobytes()
        else:                                        # 3 bytes per sample, little-endian
            raw = inter.astype("<i4").view(np.uint8).reshape(-1, 4)[:, :3].tobytes()
        if hashlib.md5(raw).digest() != want:
            raise AudioDecodeError(f"{name}: decoded samples do not match the MD5 in STREAMINFO")
    if bits.value <= 16:
        return RawAudio(out.astype(np.int16), sr.value, 1.0 / float(1 << (bits.value - 1)))
    return RawAudio(out, sr.value, 1.0 / float(1 << (bits.value - 1)))


def decode_raw(path, file_stream=None) -> RawAudio:
    """``path`` names the audio (its extension selects the decoder); ``file_stream`` optionally supplies the bytes (members of tar / zip archives).
    Raises AudioDecodeError for anything that cannot be decoded."""
    ext = os.path.splitext(str(path))[1].lower()
    try:
        if ext == ".wav":
            return _wav_raw(file_stream if file_stream is not None else path)
        if ext == ".flac":
            if file_stream is not None:
                data = file_stream.read()
            else:
                with open(path, "rb") as f:
                    data = f.read()
            return _flac_raw(data, path)
    except AudioDecodeError:
        raise
    except (ValueError, EOFError, OSError, struct.error) as e:   # damaged or truncated header / unreadable
        raise AudioDecodeError(f"{path}: {type(e).__name__}: {e}") from e
    if ext in AUDIO_EXTS:
        raise AudioDecodeError(f"{path}: decoding {ext} needs a lossy-codec decoder (ffmpeg in the reference), which this build does not ship; convert to WAV or FLAC")
    raise AudioDecodeError(f"{path}: unsupported audio file extension")


def load(path, file_stream=None):
    """torchaudio.load's role: ``(float32 [channels, samples] in [-1, 1), sample_rate)``."""
    raw = decode_raw(path, file_stream)
    return raw.to_float(), raw.sample_rate


@functools.lru_cache(maxsize=32)
def resample_table(orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    """torchaudio ``_get_sinc_resample_kernel`` (sinc_interp_hann) restated: ``(kernels float32 [n, 2 * width + o], tap ranges int32 [n, 2], o, n, width)``
    with o = orig / gcd, n = new / gcd. ``ranges[p] = [lo, hi)`` bounds the non-zero taps of phase p (the device kernel skips the exact zeros)."""
    import torch
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
    kernels = kernels.to(torch.float32)                      # [n, 1, 2 * width + o]
    k2 = kernels[:, 0].numpy()
    ranges = np.zeros((n, 2), dtype=np.int32)
    for p in range(n):
        nz = np.flatnonzero(k2[p])
        ranges[p] = (int(nz[0]), int(nz[-1]) + 1) if len(nz) else (0, 0)
    return kernels, ranges, o, n, width


def resampled_length(length: int, orig_freq: int, new_freq: int) -> int:
    """Output samples of ``Resample(orig, new)`` on ``length`` input samples: ceil(new * length / orig) (torchaudio ``_apply_sinc_resample_kernel``)."""
    if orig_freq == new_freq:
        return int(length)
    g = math.gcd(int(orig_freq), int(new_freq))
    return int(math.ceil((int(new_freq) // g) * int(length) / (int(orig_freq) // g)))


def resample(wave, orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    """Windowed-sinc polyphase resampling with torchaudio ``Resample`` defaults. wave [C, L] float32 (host)."""
    import torch
    if orig_freq == new_freq:
        return wave
    kernels, _, o, n, width = resample_table(int(orig_freq), int(new_freq), lowpass_filter_width, rolloff)
    Cn, L = wave.shape
    x = torch.nn.functional.pad(wave.reshape(Cn, 1, L), (width, width + o))
    y = torch.nn.functional.conv1d(x, kernels, stride=o)     # [C, n, frames]
    y = y.transpose(1, 2).reshape(Cn, -1)
    return y[..., :resampled_length(L, orig_freq, new_freq)]


def convert_audio(audio, sample_rate: int, target_sample_rate: int):
    """Reference ``convert_audio`` (audiotoken/utils.py:26-44): stereo -> mono mean, >2 channels raises."""
    num_channels = audio.shape[0]
    if num_channels == 2:
        audio = audio.mean(-2, keepdim=True)
    elif num_channels != 1:
        raise RuntimeError("Only mono or stereo audio is supported")
    if sample_rate != target_sample_rate:
        audio = resample(audio, sample_rate, target_sample_rate)
    return audio


def read_audio(x, model_sample_rate: int):
    """Reference ``read_audio`` (audiotoken/utils.py:47-68): ``[1, num_samples]`` float32 at the model's rate."""
    audio, sr = load(x)
    assert audio.dim() == 2, f"Audio needs to be 2D array, provided {audio.dim()}D for {x}"
    return convert_audio(audio, sr, model_sample_rate)


def process_audio_chunks(file_name, target_sample_rate: int, chunk_size: int, file_stream=None) -> Iterator[Tuple["object", str]]:
    """Reference ``process_audio_chunks`` (audiotoken/utils.py:71-101): ``chunk_size``-second chunks at the SOURCE
    rate, each resampled on its own (so chunk seams follow the reference), yielded as ``([1, n], file_name)``."""
    audio, sr = load(file_name, file_stream)
    if audio.shape[0] != 1:
        raise AudioDecodeError(f"Audio needs to be mono, provided {audio.shape[0]} channels for {file_name}")
    step = int(chunk_size * sr)
    for i in range(0, audio.shape[-1], step):
        chunk = audio[:, i:i + step]
        if sr != target_sample_rate:
            chunk = resample(chunk, sr, target_sample_rate)
        yield chunk, str(file_name)


def archive_members(x) -> Iterator[Tuple[str, bytes]]:
    """``(member name, bytes)`` of every regular file of a tar / zip archive, in archive order (reference iterate_tar / iterate_zip, utils.py:104-169)."""
    import tarfile
    import zipfile
    if str(x).endswith((".zip", ".ZIP")):
        with zipfile.ZipFile(x, "r") as zf:
            for info in zf.infolist():
                if not info.is_dir():
                    yield info.filename, zf.read(info.filename)
        return
    with tarfile.open(x, "r") as tar:
        for member in tar.getmembers():
            if not member.isfile():
                continue
            fh = tar.extractfile(member)
            if fh is not None:
                yield member.name, fh.read()


def _iterate_archive(x, model_sample_rate: int, chunk_size: int, on_skip=None):
    import io
    for name, data in archive_members(x):
        try:
            chunks = list(process_audio_chunks(name, model_sample_rate, chunk_size, io.BytesIO(data)))
        except AudioDecodeError as e:
            # one undecodable member (a README, a stereo file, an mp3) must not abort a run whose earlier files have already been appended to:
            # with a callback it is recorded and skipped like an undecodable plain file; without one the error propagates (as in the reference)
            if on_skip is None:
                raise
            on_skip(f"{x}:{name}", str(e))
            continue
        yield from chunks


def iterate_tar(x, model_sample_rate: int, chunk_size: int = 30, on_skip=None):
    """Reference ``iterate_tar`` (audiotoken/utils.py:139-169): member by member, each through process_audio_chunks."""
    return _iterate_archive(x, model_sample_rate, chunk_size, on_skip)


def iterate_zip(x, model_sample_rate: int, chunk_size: int = 30, on_skip=None):
    """Reference ``iterate_zip`` (audiotoken/utils.py:104-136)."""
    return _iterate_archive(x, model_sample_rate, chunk_size, on_skip)
