"""Device-side feeder of ``encode_batch_files``: files -> the ``[B, segment_length]`` float32 batch + mask ON THE DEVICE.

The reference feeds its encoder from DataLoader worker processes that decode (ffmpeg), resample every ``chunk_size``-second chunk
(``torchaudio.transforms.Resample``), cut / pad fixed-length segments and collate float32 batches on the host
(reference audiotoken/utils.py:71-101, datasets.py:75-139, core.py:244-267). One MI355X consumes 15 k audio-seconds per second; the host side of that
data flow — ~10^11 multiply-adds per second for the 44.1 -> 16 kHz Resample alone, three float32 passes over every sample — is what would set the
throughput. So the split here is:

  host    what needs no samples: container decoding to the file's STORAGE format (int16 PCM as it is: 2 bytes per sample; FLAC through the library's C++
          decoder; both release the GIL, a thread pool runs them ``num_workers`` files ahead, in order), headers, lengths, and the index arithmetic of
          reference ``process_audio_chunks`` + ``AudioBatchDataset._iter_chunk`` (chunk boundaries at the source rate, ``ceil(n * L / o)`` resampled
          lengths, segment starts, the < 3200-sample skip, ``AudioConfig`` per segment) — a few integers per segment;
  device  everything that touches samples, in ONE kernel per batch (csrc/audio_device.hip: ``at_segments_from_pcm``): int -> float conversion, the
          per-chunk windowed-sinc resampling (same kernel table as the host twin ``audio_io.resample``), segmentation, zero padding, mask.

The batches are the ones the host path (``AudioToken._chunk_stream`` + ``collate_fn``) produces, row for row: identical at the model's sample rate
(integer * power of two is exact), within fp32 summation order (<= 1e-6) when resampled — tests/test_feeder_gpu.py. Since round 5 also for
Tokenizers.semantic_s: its per-chunk transform (``hubert_processor``: zero mean / unit variance over every streamed chunk, reference encoder.py:20-26 applied at
datasets.py:78-79) is ``transform="zmuv"`` here — two more small launches take the chunk's float64 moments in a fixed order, the segment kernel normalises
(``at_segments_from_pcm_zmuv``); rows equal the host transform's within float32 summation error.
"""
from __future__ import annotations

import ctypes as C
import io
import threading
import time
from collections import deque
from copy import deepcopy
from typing import Callable, Iterator, List, Optional

import numpy as np
import torch

from . import _cabi
from .audio_io import AudioDecodeError, RawAudio, archive_members, decode_raw, resample_table, resampled_length, wav_probe
from .configs import AUDIO_EXTS, TAR_EXTS, ZIP_EXTS, AudioConfig
from .harness import MIN_SEGMENT_SAMPLES
from .logger import get_logger
from .prefetch import background, ordered_map

logger = get_logger(__name__)

_FMT = {np.dtype(np.int16): _cabi.PCM_S16, np.dtype(np.int32): _cabi.PCM_S32, np.dtype(np.float32): _cabi.PCM_F32, np.dtype(np.uint8): _cabi.PCM_U8}
_STAGE_BYTES = 32 << 20
_POOL_GRAIN = 1 << 20            # pinned read buffers come in multiples of 1 MiB
_POOL_KEEP_BYTES = 512 << 20     # idle pinned buffers kept for reuse; beyond that they are released
_POOL_MAX_FILE = 512 << 20       # larger files take the staged path (two 32 MiB pinned buffers) instead of one pinned buffer of their size


class _PinnedPool:
    """Page-locked read buffers for the decode workers: a plain PCM WAV file is read from the page cache STRAIGHT into one of these (one copy, in the
    worker thread, GIL released) and goes to the device from there by DMA — the main thread no longer copies every sample into a staging buffer
    (that copy, ~9 GB/s on one core, was the largest host cost per batch: 369 MB for 256 files of 30 s). Buffers return when their copy has finished."""

    def __init__(self):
        self._lock = threading.Lock()
        self._free = {}          # capacity -> [tensors]
        self._kept = 0
        self.allocations = 0     # buffers page-locked so far (a miss of the free lists)

    def take(self, nbytes: int) -> torch.Tensor:
        cap = max(_POOL_GRAIN, (nbytes + _POOL_GRAIN - 1) // _POOL_GRAIN * _POOL_GRAIN)
        with self._lock:
            lst = self._free.get(cap)
            if lst:
                self._kept -= cap
                return lst.pop()
            self.allocations += 1
        return torch.empty(cap, dtype=torch.uint8).pin_memory()

    def give(self, t: torch.Tensor) -> None:
        cap = t.numel()
        with self._lock:
            if self._kept + cap <= _POOL_KEEP_BYTES:
                self._free.setdefault(cap, []).append(t)
                self._kept += cap


_POOL = _PinnedPool()            # one per process: page-locking is slow, the buffers are reused across encode_batch_files calls


class DeviceFeeder:
    """``for segments, masks, file_pointers, ready in DeviceFeeder(...).batches(files, batch_size)``: ``segments`` / ``masks`` are device tensors
    produced on the feeder's own stream; ``ready`` is the event the consumer's stream must wait for. ``timings`` accumulates host seconds per stage
    (decode wait, upload, descriptors + launch) for bench.py's files leg."""

    def __init__(self, device, model_sample_rate: int, chunk_size: int, model_token_rate: int, pad_token: Optional[int] = 0, num_workers: int = 0,
                 on_skip: Optional[Callable[[str, str], None]] = None, transform: Optional[str] = None):
        assert transform in (None, "zmuv"), "the device feeder knows one per-chunk transform: 'zmuv' (hubert_processor)"
        self.transform = transform
        self.device = torch.device(device)
        assert self.device.type == "cuda", "the device feeder needs a HIP device"
        self.lib = _cabi.load()
        self.sr = int(model_sample_rate)
        self.chunk_size = int(chunk_size)
        self.token_rate = int(model_token_rate)
        self.pad_value = float(pad_token or 0)
        self.num_workers = int(num_workers)
        self.on_skip = on_skip or (lambda name, why: None)
        self.seg_len = self.chunk_size * self.sr
        self.stream = torch.cuda.Stream(device=self.device)
        self._tables = {}
        self._stage = [torch.empty(_STAGE_BYTES, dtype=torch.uint8).pin_memory() for _ in range(2)]
        self._stage_free = [None, None]
        self._slot = 0
        self._pool = _POOL
        self._inflight = deque()  # (event, pinned buffer) of direct uploads still in flight, oldest first
        self.timings = {"decode_wait_s": 0.0, "upload_s": 0.0, "launch_s": 0.0, "bytes_uploaded": 0, "files": 0, "segments": 0, "pinned_allocs": 0}
        self._allocs0 = _POOL.allocations

    # ---- host: decoding (worker threads; scipy's WAV reader and the C++ FLAC decoder release the GIL) ------------------------------------------
    def _decode(self, file_path: str):
        """One unit of host work -> an iterable of (name, RawAudio). Undecodable inputs are reported through ``on_skip`` and yield nothing."""
        if file_path.endswith(AUDIO_EXTS):
            try:
                direct = self._read_pinned(file_path)
                return [(file_path, direct if direct is not None else decode_raw(file_path))]
            except AudioDecodeError as e:
                self.on_skip(file_path, str(e))
                return []
        if file_path.endswith(TAR_EXTS) or file_path.endswith(ZIP_EXTS):
            def members():
                for name, data in archive_members(file_path):
                    try:
                        yield name, decode_raw(name, io.BytesIO(data))
                    except AudioDecodeError as e:
                        self.on_skip(f"{file_path}:{name}", str(e))
            return background(members) if self.num_workers > 0 else members()
        self.on_skip(file_path, "unsupported extension")
        return []

    def _read_pinned(self, file_path: str) -> Optional[RawAudio]:
        """A mono PCM / float32 WAV file read from the file straight into a pinned buffer (worker thread); None = take the general reader."""
        if not file_path.lower().endswith(".wav"):
            return None
        hdr = wav_probe(file_path)
        if hdr is None or hdr[3] > _POOL_MAX_FILE:
            return None
        dtype, sr, off, nbytes, scale, offset = hdr
        buf = self._pool.take(nbytes)
        try:
            view = memoryview(buf.numpy())[:nbytes]
            with open(file_path, "rb", buffering=0) as f:
                f.seek(off)
                got = 0
                while got < nbytes:
                    n = f.readinto(view[got:])
                    if not n:
                        raise AudioDecodeError(f"{file_path}: file shrank while it was read")
                    got += n
        except OSError as e:             # unreadable / vanished: the same report the general reader gives
            self._pool.give(buf)
            raise AudioDecodeError(f"{file_path}: {type(e).__name__}: {e}") from e
        except BaseException:
            self._pool.give(buf)
            raise
        pcm = buf.numpy()[:nbytes].view(dtype.newbyteorder("=")).reshape(1, -1)
        return RawAudio(pcm, sr, scale, offset, pinned=buf)

    # ---- device: uploads on the feeder's stream — straight from a worker's pinned read buffer, or through two pinned staging buffers -------------------
    def _reap(self, wait: bool = False) -> None:
        """Return the pinned buffers whose copy has finished (the copies complete in issue order on the feeder's one stream: only the oldest is asked)."""
        q = self._inflight
        while q and (wait or q[0][0].query()):
            ev, buf = q.popleft()
            if wait:
                ev.synchronize()
            self._pool.give(buf)

    def _upload_raw(self, raw: RawAudio) -> torch.Tensor:
        if raw.pinned is None:
            return self._upload(raw.pcm[0])
        nbytes = raw.pcm.nbytes
        with torch.cuda.stream(self.stream):
            dev = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            dev.copy_(raw.pinned[:nbytes], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self._inflight.append((ev, raw.pinned))
        self.timings["bytes_uploaded"] += int(nbytes)
        self._reap()
        return dev

    def _upload(self, arr: np.ndarray) -> torch.Tensor:
        with torch.cuda.stream(self.stream):          # (never held across a yield: the consumer's encode must stay on ITS stream)
            return self._upload_on_stream(arr)

    def _upload_on_stream(self, arr: np.ndarray) -> torch.Tensor:
        flat = np.ascontiguousarray(arr).reshape(-1)
        raw = flat.view(np.uint8)
        dev = torch.empty(raw.size, dtype=torch.uint8, device=self.device)
        pos = 0
        while pos < raw.size:
            n = min(_STAGE_BYTES, raw.size - pos)
            s = self._slot
            if self._stage_free[s] is not None:
                self._stage_free[s].synchronize()                 # the copy that last used this staging buffer has finished
            self._stage[s][:n].numpy()[:] = raw[pos:pos + n]
            dev[pos:pos + n].copy_(self._stage[s][:n], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
            self._stage_free[s] = ev
            self._slot ^= 1
            pos += n
        self.timings["bytes_uploaded"] += int(raw.size)
        return dev

    def _table(self, orig: int):
        key = int(orig)
        if key not in self._tables:
            kernels, ranges, o, n, width = resample_table(key, self.sr)
            blob = np.concatenate([kernels[:, 0].numpy().reshape(-1).view(np.uint8), ranges.reshape(-1).view(np.uint8)])
            self._tables[key] = (self._upload(blob), o, n, width)
        return self._tables[key]

    # ---- the index arithmetic of process_audio_chunks + _iter_chunk, no samples touched --------------------------------------------------------------
    def _segments_of(self, name: str, raw: RawAudio, pcm_dev: torch.Tensor):
        """Yields (SegmentDesc fields, AudioConfig) for every segment of one decoded file, in the order the host path yields them."""
        L = raw.pcm.shape[-1]
        fmt = _FMT[raw.pcm.dtype]
        native = raw.sample_rate == self.sr
        table_ptr, o, n, width = (0, 1, 1, 0)
        if not native:
            t, o, n, width = self._table(raw.sample_rate)
            table_ptr = t.data_ptr()
        step = int(self.chunk_size * raw.sample_rate)
        for c0 in range(0, L, step):
            clen = min(step, L - c0)
            Lr = clen if native else resampled_length(clen, raw.sample_rate, self.sr)
            cfg = AudioConfig(file_name=str(name), length_seconds=Lr / self.sr, length_samples=Lr, model_token_rate=self.token_rate)
            for i in range(0, Lr, self.seg_len):
                valid = min(self.seg_len, Lr - i)
                cfg.start_idx = i
                cfg.end_idx = min(i + self.seg_len, Lr)
                if valid < MIN_SEGMENT_SAMPLES:
                    logger.warning(f'File segment {i // self.sr} of {name} is too short. Skipping')
                    continue
                yield (pcm_dev.data_ptr(), table_ptr, c0, clen, i, valid, fmt, float(raw.scale), o, n, width, Lr), deepcopy(cfg), pcm_dev

    def _launch(self, rows, pointers, keep):
        t0 = time.perf_counter()
        B = len(rows)
        descs = (_cabi.SegmentDesc * B)(*[_cabi.SegmentDesc(*r) for r in rows])
        blob = np.frombuffer(descs, dtype=np.uint8)
        with torch.cuda.stream(self.stream):
            d_dev = self._upload_on_stream(blob)
            segs = torch.empty((B, self.seg_len), dtype=torch.float32, device=self.device)
            masks = torch.empty((B, self.seg_len), dtype=torch.float32, device=self.device)
            if self.transform == "zmuv":
                max_len = max(r[11] for r in rows)
                nbytes = self.lib.at_segments_zmuv_workspace_bytes(B, max_len)
                ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
                _cabi.check(self.lib.at_segments_from_pcm_zmuv(d_dev.data_ptr(), B, self.seg_len, max_len, self.pad_value, 1e-7, segs.data_ptr(), masks.data_ptr(),
                                                               ws.data_ptr(), nbytes, C.c_void_p(self.stream.cuda_stream)), "at_segments_from_pcm_zmuv")
                del ws
            else:
                _cabi.check(self.lib.at_segments_from_pcm(d_dev.data_ptr(), B, self.seg_len, self.pad_value, segs.data_ptr(), masks.data_ptr(),
                                                          C.c_void_p(self.stream.cuda_stream)), "at_segments_from_pcm")
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self.timings["launch_s"] += time.perf_counter() - t0
        self.timings["segments"] += B
        del keep, d_dev        # (allocated on the feeder's stream: the allocator reuses them only behind the kernel that reads them)
        return segs, masks, tuple(pointers), ev

    def batches(self, files: List[str], batch_size: int) -> Iterator:
        rows, pointers, keep = [], [], []
        it = iter(ordered_map(self._decode, [str(f) for f in files], self.num_workers))
        while True:
            t0 = time.perf_counter()
            source = next(it, None)
            self.timings["decode_wait_s"] += time.perf_counter() - t0
            if source is None:
                break
            try:
                for name, raw in source:
                    if raw.pcm.shape[0] != 1:
                        self.on_skip(str(name), f"Audio needs to be mono, provided {raw.pcm.shape[0]} channels for {name}")
                        continue
                    t0 = time.perf_counter()
                    pcm_dev = self._upload_raw(raw)
                    self.timings["upload_s"] += time.perf_counter() - t0
                    self.timings["files"] += 1
                    for row, cfg, ref in self._segments_of(name, raw, pcm_dev):
                        rows.append(row)
                        pointers.append(cfg)
                        keep.append(ref)
                        if len(rows) == batch_size:
                            yield self._launch(rows, pointers, keep)
                            rows, pointers, keep = [], [], []
            finally:
                close = getattr(source, "close", None)
                if close is not None:
                    close()
        if rows:
            yield self._launch(rows, pointers, keep)
        self._reap(wait=True)
        self.timings["pinned_allocs"] = _POOL.allocations - self._allocs0
