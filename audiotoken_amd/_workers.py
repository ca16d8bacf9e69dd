"""Top-level functions for the decode-ahead WORKER PROCESSES of ``encode_batch_files`` (the reference's DataLoader workers: audiotoken/core.py:259-267,
datasets.py:107-139). Spawned children import this module only — numpy, the ctypes binding and (when a chunk must be resampled) torch-CPU — never the
encoders, and never touch the GPU: the parent has HIP initialised, so workers are SPAWNED, not forked."""
from __future__ import annotations


def decode_chunks(file_path: str, sample_rate: int, chunk_size: int):
    """One plain audio file -> ("ok", [float32 ndarray [1, n] per streamed chunk]) or ("skip", reason) for an AudioDecodeError. Any other exception
    propagates to the parent through the future, like a DataLoader worker's."""
    from .audio_io import AudioDecodeError, process_audio_chunks
    try:
        return "ok", [c.numpy() for c, _ in process_audio_chunks(file_path, sample_rate, chunk_size)]
    except AudioDecodeError as e:
        return "skip", str(e)
