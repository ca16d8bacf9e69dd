"""Clip-level data parallelism: one process per GPU, weights broadcast once over RCCL/xGMI, no steady-state
collective. The reference is single-device (audiotoken/core.py:66); this module is what lets
``encode_batch_files``-style work shard across the 8 GPUs of a node."""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import torch


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """Items (clips / files) owned by ``rank``: contiguous blocks, sizes differing by at most one.
    All chunks of one file stay on one rank so the per-file append order of the reference
    (audiotoken/utils.py:214-217) is preserved."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return list(range(start, start + base + (1 if rank < rem else 0)))


def shard_by_size(sizes: Sequence[int], rank: int, world: int) -> List[int]:
    """Items owned by ``rank`` when the items have very different costs (audio files: bytes as the proxy for seconds): greedy longest-processing-time
    assignment — items in decreasing size, each to the rank with the least work so far (ties: the lowest rank; equal sizes: the lower index first) —
    which every rank computes identically from the same list. Whole items only, so all chunks of a file stay on one rank (the reference's per-file append
    order, audiotoken/utils.py:214-217); a rank's items keep their original relative order. The reference balances dynamically (its DataLoader workers
    pull files from a queue, audiotoken/datasets.py:107-139); contiguous blocks by COUNT left ranks idle behind a few long files. Worst case of LPT:
    4/3 - 1/(3 world) of the optimum; with many files of bounded size the loads differ by at most the smallest file."""
    order = sorted(range(len(sizes)), key=lambda i: (-int(sizes[i]), i))
    load = [0] * world
    owner = [0] * len(sizes)
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        owner[i] = r
        load[r] += max(int(sizes[i]), 1)
    return [i for i in range(len(sizes)) if owner[i] == rank]


def collective_device(device: torch.device, dist) -> torch.device:
    """Where a collective's tensors live: the rank's device under RCCL ("nccl"); host memory under gloo (the CPU tests, and bench.py's
    ``--shared-device`` rehearsal of the N > 1 path on a one-GPU box, where every rank drives cuda:0 and RCCL cannot be used)."""
    return torch.device("cpu") if dist.get_backend() == "gloo" else device


def broadcast_weights(weights: Optional[Dict[str, np.ndarray]], device: torch.device, dist=None, src: int = 0):
    """Rank ``src`` holds the weight dict; every rank returns an identical dict.

    One metadata broadcast (names/shapes) + ONE flat float32 tensor broadcast (backend "nccl" = RCCL on ROCm,
    ring/tree over xGMI; "gloo" on CPU in the tests)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return weights
    rank = dist.get_rank()
    meta = [[(k, tuple(v.shape)) for k, v in weights.items()]] if rank == src else [None]
    cdev = collective_device(device, dist)
    dist.broadcast_object_list(meta, src=src, device=cdev)   # the pickled layout travels on THIS rank's device under RCCL, not torch's current one
    layout = meta[0]
    total = int(sum(int(np.prod(s)) for _, s in layout))
    if rank == src:
        flat = torch.from_numpy(np.concatenate([np.asarray(weights[k], dtype=np.float32).reshape(-1) for k, _ in layout]))
        flat = flat.to(cdev)
    else:
        flat = torch.empty(total, dtype=torch.float32, device=cdev)
    dist.broadcast(flat, src=src)
    host = flat.cpu().numpy()
    out: Dict[str, np.ndarray] = {}
    off = 0
    for k, shp in layout:
        n = int(np.prod(shp))
        out[k] = host[off:off + n].reshape(shp).copy()
        off += n
    return out


def broadcast_packed(packed, device: torch.device, dist=None, src: int = 0):
    """Rank ``src`` holds ``(meta: bytes, blob: uint8 tensor)`` of a finalized model (``encoder.export_packed()``); every rank returns the pair, the blob
    on ``device``. Two collectives: the sizes, then meta and blob as uint8 tensors — the blob goes device to device (backend "nccl" = RCCL over xGMI),
    never through the host, and the receivers rebuild the model over it with ``Encoder(packed=...)`` instead of re-reading, re-folding, re-uploading
    and re-splitting the checkpoint on every rank."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return packed
    rank = dist.get_rank()
    cdev = collective_device(device, dist)   # gloo (tests / one-GPU rehearsal): the blob is staged through the host; RCCL: device to device
    sizes = torch.tensor([len(packed[0]), packed[1].numel()] if rank == src else [0, 0], dtype=torch.int64, device=cdev)
    dist.broadcast(sizes, src=src)
    n_meta, n_blob = int(sizes[0].item()), int(sizes[1].item())
    if rank == src:
        meta_t = torch.frombuffer(bytearray(packed[0]), dtype=torch.uint8).to(cdev)
        blob = packed[1].to(cdev)
    else:
        meta_t = torch.empty(n_meta, dtype=torch.uint8, device=cdev)
        blob = torch.empty(n_blob, dtype=torch.uint8, device=cdev)
    dist.broadcast(meta_t, src=src)
    dist.broadcast(blob, src=src)
    return bytes(meta_t.cpu().numpy().tobytes()), blob.to(device)


def gather_scalars(values: Sequence[float], device: torch.device, dist=None) -> List[List[float]]:
    """All-gather a few per-rank scalars (benchmark reporting only)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [list(values)]
    t = torch.tensor(list(values), dtype=torch.float64, device=collective_device(device, dist))
    outs = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(outs, t)
    return [o.tolist() for o in outs]


def ranks_agree_on_probe(encode, probe: torch.Tensor, device: torch.device, dist=None, what: str = "encoder") -> dict:
    """Start-up self-check of a multi-rank run: every rank encodes the SAME small probe batch with the model it ended up with (rank 0 built it from the
    checkpoint, the others received it by broadcast_weights / broadcast_packed + import_packed) and the token checksums — sum of ids and a position-weighted
    sum, so a permutation cannot cancel — are all-gathered. Any rank that differs from rank 0 raises on EVERY rank before a single timed step or token file:
    a wrong import on one rank must not produce a fast, wrong scaling curve (the reference is single-device, audiotoken/core.py:66: nothing to mirror)."""
    toks = encode(probe).to(torch.int64).reshape(-1)
    w = torch.arange(1, toks.numel() + 1, device=toks.device, dtype=torch.int64) % 8191
    mine = [float(toks.sum().item()), float((toks * w).sum().item() % (1 << 50))]
    rows = gather_scalars(mine, device, dist)
    bad = [r for r, row in enumerate(rows) if row != rows[0]]
    if bad:
        raise RuntimeError(f"{what}: ranks {bad} encode the start-up probe differently from rank 0 ({[rows[r] for r in bad]} vs {rows[0]}): "
                           "a rank's model differs (packed-model import / weight broadcast) — refusing to run")
    return {"ranks": len(rows), "token_checksum": int(rows[0][0]), "weighted_checksum": int(rows[0][1])}


# ---- model distribution for the product (AudioToken.load_encoder / load_decoder) ----------------------------------------------------------------------
def active(dist=None):
    """``torch.distributed`` when it is initialised with more than one rank, else None."""
    if dist is None:
        import torch.distributed as dist
    return dist if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 else None


def _rank0_first(step, device: torch.device, dist, what: str, src: int = 0):
    """Run ``step()`` on rank ``src`` only and tell every rank whether it worked BEFORE any tensor collective: a checkpoint that rank 0 cannot read must raise
    on all ranks, not leave world - 1 of them waiting in a broadcast."""
    res, err = None, None
    if dist.get_rank() == src:
        try:
            res = step()
        except Exception as e:   # re-raised below on every rank
            err = f"{type(e).__name__}: {e}"
    box = [err]
    dist.broadcast_object_list(box, src=src, device=collective_device(device, dist))
    if box[0] is not None:
        raise RuntimeError(f"{what}: rank {src} could not build the model ({box[0]})")
    return res


def weights_on_all_ranks(load: Callable[[], Dict[str, np.ndarray]], device: torch.device, dist, what: str = "weights", src: int = 0):
    """EnCodec (34 MB): rank ``src`` reads the checkpoint (``load()``), every rank returns the same name -> array dict (one flat RCCL broadcast)."""
    w = _rank0_first(load, device, dist, what, src)
    return broadcast_weights(w, device, dist, src)


def encoder_on_all_ranks(build_local: Callable[[], "torch.nn.Module"], build_from_packed: Callable[[tuple], "torch.nn.Module"], device: torch.device, dist,
                         what: str = "encoder", src: int = 0):
    """The semantic tokenizers (1.8-4.5 GB of weights): rank ``src`` reads, folds, uploads, splits and finalizes the checkpoint ONCE (``build_local()``), exports
    the finalized model as one packed device blob, and the other ranks rebuild their handle over the broadcast blob (``build_from_packed((meta, blob))``) —
    SURVEY.md §8(e); the reference is single-device (audiotoken/core.py:92-118 loads per process)."""
    def build_and_export():
        e = build_local()
        return e, e.export_packed()
    enc, packed = _rank0_first(build_and_export, device, dist, what, src) or (None, None)
    packed = broadcast_packed(packed, device, dist, src)
    if dist.get_rank() != src:
        enc = build_from_packed(packed)
    del packed
    return enc


def probe_batch(sample_rate: int, device, transform: Optional[Callable] = None) -> torch.Tensor:
    """The start-up probe: two distinct speech-like 2 s clips (synthetic.py, fixed seed — the same bytes on every rank), through the tokenizer's per-clip
    host transform when it has one (semantic_s)."""
    from . import synthetic as S
    x = torch.from_numpy(S.speech_like_waveform(2, 2 * sample_rate, sample_rate, seed=987654))
    if transform is not None:
        x = torch.stack([transform(x[i:i + 1])[0] for i in range(2)])
    return x.to(device)
