"""``AudioToken`` — the reference's public façade (audiotoken/core.py:27-359) over the MI355X HIP library.

Same constructor, attributes, method signatures, return shapes/dtypes and exception types; the differences are
forced by the environment and stated here:
* ``device`` must be a HIP device (``"cuda[:i]"`` under PyTorch-ROCm). The reference's ``device="cpu"`` default
  would need a CPU path, which this package deliberately does not have — constructing with "cpu" raises
  ``ValueError`` at ``load_encoder`` time.
* checkpoints cannot be downloaded (no network): pass ``weights=`` / ``quantizer=`` kwargs or set the
  ``AUDIOTOKEN_*`` environment variables (configs.py); with nothing given, synthetic weights are used and a
  warning is logged.
* ``compile`` is accepted and ignored (there is no tracing compiler in the path; kernels are precompiled HIP).
* ``encode_batch_files`` shards files over ranks when ``torch.distributed`` is initialised (one process per GPU).
"""
from __future__ import annotations

import os
import time
from pathlib import Path
from typing import Callable, List, Optional, Union

import numpy as np
import torch

from .configs import (AUDIO_EXTS, TAR_EXTS, ZIP_EXTS, AcousticDecoderConfig, AcousticEncoderConfig, EncoderConfig, HubertEncoderConfig, Tokenizers,
                      Wav2VecBertConfig, num_codebooks_to_bandwidth)
from .harness import batched, collate_fn, iter_chunk, sanitize_path, save_audio_tokens, save_rel_audio_tokens
from .logger import get_logger

logger = get_logger(__name__, log_file=None, level="WARNING")


def _in_flight(start, items, depth):
    """``start(item)`` for up to ``depth`` items ahead of the consumer, results in order (``start`` returns immediately: it submits work elsewhere)."""
    from collections import deque
    pending = deque()
    for it in items:
        pending.append(start(it))
        if len(pending) >= max(1, depth):
            yield pending.popleft()
    while pending:
        yield pending.popleft()


class AudioToken:
    def __init__(self, tokenizer: Tokenizers, device: str = "cpu", compile: bool = False, **kwargs):
        """Reference ``AudioToken.__init__`` (core.py:28-71). Supported kwargs: ``num_codebooks`` in {2,4,8,16}
        (default 16 — the reference's actual default, core.py:67), ``weights``, ``quantizer``."""
        self.tokenizer_name = Tokenizers(tokenizer)  # ValueError on unknown names, like the reference's StrEnum
        self.encoder: Optional[torch.nn.Module] = None
        self.decoder: Optional[torch.nn.Module] = None
        self.model_config: EncoderConfig
        self.transform_func: Optional[Callable] = None
        self.compile = compile
        self.kwargs = kwargs
        self.device = device
        self.skipped_files: List[tuple] = []   # (path, reason) of the inputs the last encode_batch_files could not decode
        self.rank_probe: Optional[dict] = None    # checksums of the multi-rank start-up probe (load_encoder under torch.distributed), else None
        self.num_codebooks = kwargs.get("num_codebooks", 16)
        assert self.num_codebooks in [2, 4, 8, 16], "num_codebooks must be one of [2, 4, 8, 16]"
        self.load_config()

    def load_config(self):
        """core.py:73-90."""
        if self.tokenizer_name == Tokenizers.acoustic:
            self.model_config = AcousticEncoderConfig(bandwidth=num_codebooks_to_bandwidth(self.num_codebooks))
        elif self.tokenizer_name == Tokenizers.semantic_s:
            self.model_config = HubertEncoderConfig()
        elif self.tokenizer_name == Tokenizers.semantic_m:
            self.model_config = Wav2VecBertConfig()
        else:
            raise ValueError(f"Tokenizer {self.tokenizer_name} not supported")
        if self.kwargs.get("weights") is not None and not isinstance(self.kwargs["weights"], dict):
            self.model_config.weights = self.kwargs["weights"]
        if self.kwargs.get("quantizer") is not None and hasattr(self.model_config, "quantizer_path"):
            self.model_config.quantizer_path = self.kwargs["quantizer"]
        self.model_sample_rate = self.model_config.model_sample_rate

    def _weights_kw(self):
        w = self.kwargs.get("weights")
        return w if isinstance(w, dict) else None

    def load_encoder(self):
        """core.py:92-118 (lazy construction on first use). The reference is single-device; here, when ``torch.distributed`` is initialised with more than one
        rank, the model is built ONCE: rank 0 reads the checkpoint (the ``weights=`` path / dict, the ``AUDIOTOKEN_*`` variables, or the synthetic default),
        the other ranks receive it over RCCL — EnCodec as one flat tensor, the semantic tokenizers as rank 0's FINALIZED model (one packed device blob:
        no N-fold checkpoint read, fold, upload or split; distributed.encoder_on_all_ranks) — and every rank then encodes the same 2-clip probe: a rank whose
        tokens differ from rank 0's raises on ALL ranks before the first batch (``self.rank_probe`` keeps the checksums). ``broadcast_weights=False`` keeps
        per-rank loading (every rank must then be given the same checkpoint; the probe still runs unless ``rank_probe=False``). Collective: every rank of the
        group must construct its ``AudioToken`` and reach its first ``encode*`` call."""
        if self.encoder is not None:
            return
        from . import distributed as D
        dist = D.active()
        dev = torch.device(self.device)
        shared = dist is not None and self.kwargs.get("broadcast_weights", True)
        wkw = self._weights_kw()
        if self.tokenizer_name == Tokenizers.acoustic:
            from .encoder import AcousticEncoder
            if shared:
                from .encoder import encodec_weights_from
                wkw = D.weights_on_all_ranks(lambda: encodec_weights_from(wkw if wkw is not None else self.model_config.weights, with_decoder=False),
                                             dev, dist, "acoustic encoder")
            self.encoder = AcousticEncoder(device=self.device, config=self.model_config, weights=wkw)
        elif self.tokenizer_name == Tokenizers.semantic_s:
            from .hubert import HubertEncoder, hubert_processor
            local = lambda: HubertEncoder(config=self.model_config, device=self.device, weights=wkw)
            self.encoder = (D.encoder_on_all_ranks(local, lambda p: HubertEncoder(config=self.model_config, device=self.device, packed=p), dev, dist, "semantic_s encoder")
                            if shared else local())
            self.transform_func = hubert_processor
        elif self.tokenizer_name == Tokenizers.semantic_m:
            from .encoder import Wav2VecBertEncoder
            local = lambda: Wav2VecBertEncoder(config=self.model_config, device=self.device, quantize=True, weights=wkw)
            self.encoder = (D.encoder_on_all_ranks(local, lambda p: Wav2VecBertEncoder(config=self.model_config, device=self.device, quantize=True, packed=p),
                                                   dev, dist, "semantic_m encoder") if shared else local())
        else:
            raise ValueError(f"Tokenizer {self.tokenizer_name} not supported")
        self.encoder.eval()
        self.rank_probe = None
        if dist is not None and self.kwargs.get("rank_probe", True):
            x = D.probe_batch(self.model_config.model_sample_rate, self.device, self.transform_func)
            self.rank_probe = D.ranks_agree_on_probe(lambda w: self.encoder(w, torch.ones_like(w)), x, dev, dist, str(self.tokenizer_name))

    def encode(self, audio: Union[torch.Tensor, np.ndarray, os.PathLike, bytes, Path], chunk_size: Optional[int] = None) -> torch.Tensor:
        """core.py:120-185. ``(1, num_samples)`` array/tensor or a path -> tokens ``(1, K, T)`` on the CPU
        (``(K, sum T)`` when a path is encoded with ``chunk_size`` — the reference drops the batch dim there)."""
        self.load_encoder()
        if isinstance(audio, np.ndarray):
            assert audio.ndim == 2, "Audio must be 2D array"
            assert audio.shape[0] == 1, "Audio must mono"
            return self._encode_single(torch.from_numpy(audio))
        elif isinstance(audio, torch.Tensor):
            assert audio.ndim == 2, "Audio must be 2D array"
            assert audio.shape[0] == 1, "Audio must mono"
            return self._encode_single(audio)
        elif isinstance(audio, os.PathLike) or isinstance(audio, Path):
            from .audio_io import process_audio_chunks, read_audio
            if chunk_size is None:
                logger.warning("Chunking not provided. Encoding the complete audio file at once. May run out of memory for larger audio files.")
                return self._encode_single(read_audio(audio, self.model_config.model_sample_rate))
            processed = [self._encode_single(chunk)[0]
                         for chunk, _ in process_audio_chunks(audio, self.model_config.model_sample_rate, chunk_size)]
            return torch.cat(processed, dim=-1)
        elif isinstance(audio, bytes):
            raise NotImplementedError("Encoding bytes not supported yet")
        else:
            raise ValueError(f"Unsupported input type {type(audio)}. Should be one of: {np.ndarray, os.PathLike, bytes, Path}")

    def _encode_single(self, audio: torch.Tensor) -> torch.Tensor:
        """core.py:187-196."""
        if self.transform_func:
            audio = self.transform_func(audio)
        input_batch = audio.to(self.device)
        attention_mask = torch.ones_like(input_batch, device=self.device)
        toks = self.encoder(input_batch, attention_mask)
        out = toks.cpu()                      # the reference's synchronisation point
        if hasattr(self.encoder, "verified"):  # device status of that call (LSTM hand-off / fp16 range): repeat on the safe path if set
            checked = self.encoder.verified(toks, input_batch, attention_mask)
            if checked is not toks:
                out = checked.cpu()
        return out

    def _chunk_stream(self, files, chunk_size: int, num_workers: int = 0, worker_processes: bool = False):
        """File -> streamed ``chunk_size``-second chunks -> segments (reference datasets.py:107-139). Decoding and resampling run ``num_workers`` files
        ahead of the consumer — in SPAWNED worker processes for plain audio files when ``worker_processes`` (the reference's DataLoader workers,
        core.py:259-267; the parent has the GPU initialised, so never forked), else on a thread pool; archives are streamed member by member by a
        background thread either way (members are not random-access). The segment order equals the sequential one."""
        from .audio_io import AudioDecodeError, iterate_tar, iterate_zip, process_audio_chunks
        from .prefetch import background, ordered_map
        sr = self.model_config.model_sample_rate
        pool = None
        if worker_processes and num_workers > 0:
            import multiprocessing as mp
            from concurrent.futures import ProcessPoolExecutor
            pool = ProcessPoolExecutor(max_workers=num_workers, mp_context=mp.get_context("spawn"))

        def skipped(name, why):   # an undecodable file / archive member: recorded and skipped
            logger.error(f"Skipping {name}: {why}")
            self.skipped_files.append((name, why))

        def load(file_path: str):
            """One unit of host work: plain audio files are decoded completely; archives return a streaming source. A file that
            cannot be decoded (AudioDecodeError: a codec this build does not ship, more than one channel, a damaged header) is skipped, recorded
            in ``self.skipped_files`` and reported at the end of the run — it must not abort a run whose earlier files have already been
            appended to. Any other exception propagates, as in the reference (datasets.py __iter__)."""
            if file_path.endswith(AUDIO_EXTS):
                if pool is not None:
                    from ._workers import decode_chunks
                    return pool.submit(decode_chunks, file_path, sr, chunk_size)
                try:
                    return list(process_audio_chunks(file_path, sr, chunk_size))
                except AudioDecodeError as e:
                    skipped(file_path, str(e))
                    return []
            if file_path.endswith(TAR_EXTS):
                return background(lambda: iterate_tar(file_path, sr, chunk_size, skipped)) if num_workers > 0 else iterate_tar(file_path, sr, chunk_size, skipped)
            if file_path.endswith(ZIP_EXTS):
                return background(lambda: iterate_zip(file_path, sr, chunk_size, skipped)) if num_workers > 0 else iterate_zip(file_path, sr, chunk_size, skipped)
            logger.error(f"File {file_path} not supported for processing. Only {AUDIO_EXTS + TAR_EXTS + ZIP_EXTS} supported")
            self.skipped_files.append((file_path, "unsupported extension"))
            return []

        def resolve(file_path, source):
            if pool is not None and hasattr(source, "result"):      # a worker process's answer: numpy chunks, or the reason the file was skipped
                kind, payload = source.result()
                if kind == "skip":
                    skipped(file_path, payload)
                    return []
                return [(torch.from_numpy(c), file_path) for c in payload]
            return source

        names = [str(f) for f in files]
        try:
            # with processes `load` only SUBMITS (the thread pool of ordered_map is not needed: in-line submission keeps num_workers futures in flight)
            sources = ordered_map(lambda f: (f, load(f)), names, num_workers if pool is None else 0) if pool is None else _in_flight(lambda f: (f, load(f)), names, num_workers)
            for file_path, source in sources:
                source = resolve(file_path, source)
                try:
                    for waveform, file_name in source:
                        yield from iter_chunk(waveform, file_name, sample_rate=self.model_config.model_sample_rate, chunk_size=chunk_size,
                                              model_token_rate=self.model_config.model_token_rate, pad_token=self.model_config.pad_token,
                                              transform=self.transform_func)
                finally:   # an exception in the consumer (or an abandoned run) must not leave an archive's producer thread and its handle behind
                    close = getattr(source, "close", None)
                    if close is not None:
                        close()
        finally:
            if pool is not None:
                pool.shutdown(wait=False, cancel_futures=True)

    def encode_batch_files(self, batch_size: int, outdir: os.PathLike, chunk_size: int = 30, num_workers: int = 12,
                           audio_files: Optional[List[os.PathLike]] = None, audio_dir: Optional[Union[os.PathLike, Path]] = None,
                           **dataloader_kwargs) -> None:
        """core.py:198-289. Files -> ``chunk_size``-second segments -> batches -> encoder -> per-row trimmed
        ``<stem>.npy`` (append semantics as in the reference). ``num_workers`` files are decoded ahead of the device, in order (the reference's
        DataLoader workers; 0 = inline). On a HIP device, for the tokenizers without a host-side transform (acoustic, semantic_m), the samples never become
        float32 on the host: the DEVICE FEEDER (feeder.py) uploads the decoded PCM and converts / resamples / segments / pads in one kernel per batch
        (``device_feeder=False`` keeps the reference's host data flow). Host path: a thread pool by default — the heavy work (file read, FLAC decode, the
        resampling conv1d) releases the GIL — or spawned worker PROCESSES with ``worker_processes=True`` (the reference's arrangement; archives are always
        streamed by a thread). Under ``torch.distributed`` every rank takes whole files, balanced by size (distributed.shard_by_size): all chunks of a file stay
        on one rank, preserving the append order (``shard_across_ranks=False``: this rank takes every file it was given — for callers that have already
        split the work, e.g. one directory per rank)."""
        self.load_encoder()
        self.skipped_files = []
        assert audio_files or audio_dir, "Either audio_files or audio_dir must be provided"
        assert not (audio_files and audio_dir), "Provide either audio_files or audio_dir, not both"
        outdir = sanitize_path(outdir)
        if audio_files is not None:
            files = [str(f) for f in audio_files]
        else:
            # every file under audio_dir with one of the extensions — the set the reference's `glob.iglob(f"{audio_dir}/**/*{ext}", recursive=True)` per
            # extension finds (datasets.py:47-50; glob does not descend into or match dot-names) — in ONE walk instead of fourteen, sorted (the sharding
            # below needs every rank to see the same order)
            exts = AUDIO_EXTS + TAR_EXTS + ZIP_EXTS
            files = []
            seen = set()    # glob follows symlinked sub-directories (datasets laid out as symlink farms); so does this walk, once per real directory
            try:            # the root counts as seen: a link cycle back to it must not list its own files a second time
                st = os.stat(str(audio_dir))
                seen.add((st.st_dev, st.st_ino))
            except OSError:
                pass
            for d, dirs, names in os.walk(str(audio_dir), followlinks=True):
                keep = []
                for x in dirs:
                    if x.startswith("."):
                        continue
                    try:
                        st = os.stat(os.path.join(d, x))
                    except OSError:
                        continue
                    if (st.st_dev, st.st_ino) not in seen:
                        seen.add((st.st_dev, st.st_ino))
                        keep.append(x)
                dirs[:] = keep
                files.extend(os.path.join(d, n) for n in names if n.endswith(exts) and not n.startswith("."))
            files.sort()
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 and dataloader_kwargs.get("shard_across_ranks", True):
            # duration-aware: whole files by greedy LPT on their sizes (distributed.shard_by_size). Rank 0 stats the list ONCE and broadcasts the sizes (N_files
            # stats instead of N_files x world on a shared filesystem; and every rank provably shards the same numbers)
            import hashlib
            from .distributed import collective_device, shard_by_size
            digest = hashlib.sha256("\0".join(files).encode("utf-8", "surrogateescape")).hexdigest()
            sizes = [([os.path.getsize(f) if os.path.exists(f) else 0 for f in files], digest)] if dist.get_rank() == 0 else [None]
            # the pickled list travels on THIS rank's device under RCCL (not torch's current device: a caller that never called set_device would put every rank on cuda:0)
            dist.broadcast_object_list(sizes, src=0, device=collective_device(torch.device(self.device), dist))
            sizes, digest0 = sizes[0]
            # every rank learns whether ALL ranks hold rank 0's list: a rank that differs must stop the others too, not let them encode a shard of a list it does not share
            from .distributed import gather_scalars
            same = len(sizes) == len(files) and digest0 == digest
            votes = gather_scalars([1.0 if same else 0.0], torch.device(self.device), dist)
            bad = [r for r, v in enumerate(votes) if v[0] != 1.0]
            assert not bad, f"ranks {bad} see a different file list than rank 0: encode_batch_files needs the same audio_files / audio_dir on every rank"
            files = [files[i] for i in shard_by_size(sizes, dist.get_rank(), dist.get_world_size())]
        start_time = time.time()
        on_gpu = torch.device(self.device).type == "cuda"
        copy_stream = torch.cuda.Stream(device=self.device) if on_gpu else None

        def upload(batch):
            """Collate a batch and start its host->device copy (pinned staging, side stream) so it overlaps the encode of
            the batch before it; returns (ids, masks, file_pointers, ready_event)."""
            input_ids, attention_masks, file_pointers = collate_fn(batch)
            if not on_gpu:
                return input_ids.to(self.device), attention_masks.to(self.device), file_pointers, None
            with torch.cuda.stream(copy_stream):
                ids = input_ids.pin_memory().to(self.device, non_blocking=True)
                masks = attention_masks.pin_memory().to(self.device, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(copy_stream)
            return ids, masks, file_pointers, ev

        # Device feeder (feeder.py): decoding stays on the host, sample conversion / per-chunk resampling / segmentation / padding run in one HIP kernel per
        # batch — for semantic_s including its per-chunk zero-mean / unit-variance transform (feeder.py, transform="zmuv"); a custom transform_func and
        # `device_feeder=False` keep the host data flow of the reference.
        from .hubert import hubert_processor as _zmuv
        dev_transform = "zmuv" if self.transform_func is _zmuv else None     # semantic_s: the per-chunk normalisation runs in the feeder's kernels
        use_feeder = on_gpu and (self.transform_func is None or dev_transform) and dataloader_kwargs.get("device_feeder", True)
        self.feeder_timings = None
        if use_feeder:
            from .feeder import DeviceFeeder

            def skipped(name, why):
                logger.error(f"Skipping {name}: {why}")
                self.skipped_files.append((name, why))
            feeder = DeviceFeeder(self.device, self.model_config.model_sample_rate, chunk_size, self.model_config.model_token_rate,
                                  self.model_config.pad_token, num_workers, skipped, transform=dev_transform)
            self.feeder_timings = feeder.timings
            staged_iter = feeder.batches(files, batch_size)
            stage_next = lambda: next(staged_iter, None)
        else:
            batches = batched(self._chunk_stream(files, chunk_size, num_workers, bool(dataloader_kwargs.get("worker_processes", False))), batch_size)

            def stage_next():
                b = next(batches, None)
                return upload(b) if b is not None else None
        # host seconds per stage of the loop (bench.py's files leg): `stage` = producing the next batch (decode wait + upload + descriptors, or the host
        # chunk stream + collate + upload), `encode_call` = enqueueing the encode, `device_wait` = blocked on the device (the status read of verified / the
        # first .cpu()), `save` = the per-row trim + append to the .npy files (of the batch BEFORE, while the device encodes the current one)
        rt = self.run_timings = {"stage_s": 0.0, "encode_call_s": 0.0, "device_wait_s": 0.0, "save_s": 0.0, "batches": 0, "rows": 0}
        def save(tokens, pointers):
            for tokens_batch, file_pointer in zip(tokens, pointers):
                if audio_files is not None:
                    save_audio_tokens(tokens_batch, file_pointer, str(outdir))
                else:
                    save_rel_audio_tokens(tokens_batch, file_pointer, str(outdir), str(audio_dir))

        t0 = time.perf_counter()
        staged = stage_next()
        rt["stage_s"] += time.perf_counter() - t0
        pending = None    # (tokens on the host, file pointers) of the batch before: written while the device encodes the next one, in batch order
        try:
            while staged is not None:
                input_ids, attention_masks, file_pointers, ev = staged
                t0 = time.perf_counter()
                if ev is not None:
                    torch.cuda.current_stream(self.device).wait_event(ev)
                    input_ids.record_stream(torch.cuda.current_stream(self.device))
                    attention_masks.record_stream(torch.cuda.current_stream(self.device))
                encoded_audio = self.encoder(input_ids, attention_masks)      # asynchronous on the device
                t1 = time.perf_counter()
                if pending is not None:
                    p, pending = pending, None      # ownership first: an interrupt inside the save must not make the `finally` below append the rows again
                    save(*p)
                t2 = time.perf_counter()
                staged = stage_next()                                         # the next batch is decoded / uploaded / cut while this one encodes
                t3 = time.perf_counter()
                if hasattr(self.encoder, "verified"):   # the copy below synchronises anyway: check the call's device status first
                    encoded_audio = self.encoder.verified(encoded_audio, input_ids, attention_masks)
                pending = (encoded_audio.cpu(), file_pointers)   # ONE device-to-host copy per batch (a per-row .cpu() inside the save would synchronise B times)
                t4 = time.perf_counter()
                rt["encode_call_s"] += t1 - t0; rt["save_s"] += t2 - t1; rt["stage_s"] += t3 - t2; rt["device_wait_s"] += t4 - t3
                rt["batches"] += 1; rt["rows"] += len(file_pointers)
        finally:
            # also when the encode / staging of batch k raised: the verified tokens of batch k - 1 are on the host and belong in their files (earlier
            # batches are already there) — the save is deferred by one batch, it must not be lost by it
            if pending is not None:
                t0 = time.perf_counter()
                p, pending = pending, None
                save(*p)
                rt["save_s"] += time.perf_counter() - t0
            try:
                self._end_of_run()
            except Exception as e:   # bookkeeping must not mask the exception that ended the run
                logger.error(f"encode_batch_files: end-of-run bookkeeping failed: {type(e).__name__}: {e}")
        rt["total_s"] = time.time() - start_time
        logger.debug(f"Encoding batch files took: {time.time() - start_time:.2f}s")
        if self.skipped_files:
            logger.error(f"encode_batch_files: {len(self.skipped_files)} input(s) were skipped and have NO token file (AudioToken.skipped_files): "
                         + "; ".join(f"{p} ({why})" for p, why in self.skipped_files[:8]) + (" ..." if len(self.skipped_files) > 8 else ""))

    def _end_of_run(self):
        """End of an encode_batch_files run: layers the range fallback moved to bf16x3 because of THIS run's inputs go back to f16x2 (a loud or clipped file
        must not slow down, or change the rounding of, every later run of the process); what happened is kept in ``run_summary``."""
        enc = self.encoder
        self.run_summary = {"fallback_batches": getattr(enc, "fallback_batches", 0), "pinned_layers": sorted(set(getattr(enc, "pinned_layers", []) or [])),
                            "nonfinite_batches": getattr(enc, "nonfinite_batches", 0), "skipped_files": len(self.skipped_files)}
        if self.run_summary["pinned_layers"]:
            logger.error(f"encode_batch_files: layers {self.run_summary['pinned_layers']} ran on bf16x3 for part of this run (fp16 range overflow); restored to f16x2")
        if hasattr(enc, "unpin_layers"):
            enc.unpin_layers()

    def load_decoder(self, **kwargs):
        """core.py:291-315 — only the acoustic decoder exists here (the semantic decoders are out of scope)."""
        if self.decoder is None:
            if self.tokenizer_name == Tokenizers.acoustic:
                from .decoder import AcousticDecoder
                cfg = AcousticDecoderConfig(bandwidth=num_codebooks_to_bandwidth(self.num_codebooks))
                if self.kwargs.get("weights") is not None and not isinstance(self.kwargs["weights"], dict):
                    cfg.weights = self.kwargs["weights"]
                wkw = self._weights_kw()
                from . import distributed as D
                dist = D.active()
                if dist is not None and self.kwargs.get("broadcast_weights", True):   # as load_encoder: rank 0 reads the checkpoint, one RCCL broadcast
                    from .encoder import encodec_weights_from
                    wkw = D.weights_on_all_ranks(lambda: encodec_weights_from(wkw if wkw is not None else cfg.weights, with_decoder=True),
                                                 torch.device(self.device), dist, "acoustic decoder")
                self.decoder = AcousticDecoder(config=cfg, device=self.device, weights=wkw, **kwargs)
            elif self.tokenizer_name in (Tokenizers.semantic_s, Tokenizers.semantic_m):
                raise NotImplementedError("semantic decoders (autoregressive GPT + bark fine model) are out of scope of the MI355X hot path")
            else:
                raise ValueError(f"Tokenizer {self.tokenizer_name} not supported")
            self.decoder.eval()

    def decode(self, tokens: Union[torch.Tensor, np.ndarray, os.PathLike, Path], **kwargs) -> torch.Tensor:
        """core.py:317-353: tokens ``(B, K, T)`` -> audio ``(1, B*320*T)`` on the CPU."""
        self.load_decoder(**kwargs)
        if isinstance(tokens, np.ndarray):
            return self._decode_single(torch.from_numpy(tokens))
        elif isinstance(tokens, torch.Tensor):
            return self._decode_single(tokens)
        elif isinstance(tokens, os.PathLike) or isinstance(tokens, Path):
            tokens_mem = torch.load(tokens, map_location="cpu")
            return self._decode_single(tokens_mem)
        else:
            raise ValueError(f"Unsupported input type {type(tokens)}. Should be one of: {np.ndarray, os.PathLike, Path}")

    def _decode_single(self, tokens: torch.Tensor) -> torch.Tensor:
        """core.py:355-359."""
        input_batch = tokens.to(dtype=torch.long)
        toks = self.decoder(input_batch)
        out = toks.cpu()
        if hasattr(self.decoder, "verified"):
            checked = self.decoder.verified(toks, input_batch)
            if checked is not toks:
                out = checked.cpu()
        return out
