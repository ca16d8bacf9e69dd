#!/usr/bin/env python3
"""Headline benchmark: audio-seconds tokenized per wall-second on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload acoustic|semantic_m] [--no-cpu-baseline]

One "step" = one pass of the hot path (the reference's ``self.encoder(input_batch, attention_mask)`` call,
audiotoken/core.py:276) over one synthetic batch that is already resident in HBM. At N=1 the workload is
BASELINE.json configs[1]: Tokenizers.acoustic, 256 clips x 10 s @ 24 kHz, 8 codebooks. With N>1 (launched by
torch.distributed.run, one rank per GPU) every rank encodes its own 256-clip shard — clips are independent, so
there is no data-path collective ("weak" scaling); RCCL is used only for the start barrier, the weight
broadcast check and the max-over-ranks time.

Prints ONE JSON line on rank 0 (contract in the task statement) including
  "roofline":     dominant kernel group's achieved rate vs the gfx950 peak, timed with HIP events on the launch stream
  "cpu_baseline": the CPU oracle (a port of the reference's CPU path) timed on this host on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
F32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense f32-input MFMA peak


def acoustic_flops_per_clip(N: int, n_q: int):
    """Algorithmic FLOPs of one clip, per kernel group, 2 FLOP per MAC."""
    L = [N]
    for r in (2, 4, 5, 8):
        L.append(-(-L[-1] // r))
    g = {"conv0": 2.0 * N * 7 * 32}
    C = 32
    for s, r in enumerate((2, 4, 5, 8)):
        g[f"res{s}"] = 2.0 * L[s] * (3 * C * (C // 2) + (C // 2) * C + C * C)
        g[f"down{s}"] = 2.0 * L[s + 1] * (2 * r * C) * (2 * C)
        C *= 2
    T = L[4]
    g["lstm_ih"] = 2.0 * T * 512 * 2048 * 2
    g["lstm_rec"] = 2.0 * T * 512 * 2048 * 2
    g["final_conv"] = 2.0 * T * 7 * 512 * 128
    g["rvq"] = 2.0 * T * n_q * 1024 * 128
    return g, T


def acoustic_bytes_per_clip(N: int, n_q: int):
    """Algorithmic (compulsory) HBM bytes per kernel group as launched today: each group reads its input
    activation once and writes its output once (fp32, channels-last)."""
    L = [N]
    for r in (2, 4, 5, 8):
        L.append(-(-L[-1] // r))
    g = {"conv0": 4.0 * N * (1 + 32)}
    C = 32
    for s in range(4):
        g[f"res{s}"] = 4.0 * L[s] * (C + C)
        g[f"down{s}"] = 4.0 * (L[s] * C + L[s + 1] * 2 * C)
        C *= 2
    T = L[4]
    g["lstm_ih"] = 4.0 * T * (512 + 2048) * 2
    g["lstm_rec"] = 4.0 * T * (2048 + 512 * 2) * 2
    g["final_conv"] = 4.0 * T * (512 + 128)
    g["rvq"] = 4.0 * T * 128 + 2.0 * T * n_q
    return g


def host_threads() -> int:
    """Threads for the CPU baseline: the cores this process may actually run on, capped at 16 — the box's
    logical core count (os.cpu_count()) can be far above its CPU quota and the oracle's 750-step LSTM loop
    collapses under oversubscription."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:  # pragma: no cover
        n = os.cpu_count() or 1
    return max(1, min(16, n))


def cpu_baseline_acoustic(n_q: int, budget_s: float = 15.0):
    """Time the CPU oracle (oracle/encodec_ref.py — a torch-CPU fp32 port of the reference's CPU encode path,
    reference audiotoken/encoder.py:44-57 with device='cpu') on a bounded sample of the same workload:
    10 s @24 kHz clips, batch 2, repeated until ~budget_s of CPU work (a 1 s probe sizes the sample)."""
    from audiotoken_amd import weights as W
    from oracle import encodec_ref as R

    torch.set_num_threads(host_threads())
    w = W.synth_encodec_weights(seed=0, with_decoder=False)
    wt = {k: torch.from_numpy(v) for k, v in w.items()}
    with torch.no_grad():
        probe = torch.from_numpy(W.synth_waveform(1, 24000, 24000, seed=1))
        R.acoustic_encode(wt, probe, n_q)  # warm-up (thread pool, weight-norm folds)
        t0 = time.perf_counter()
        R.acoustic_encode(wt, probe, n_q)
        per_audio_s = time.perf_counter() - t0
        # bounded sample: batch of 10 s clips that should take <= budget_s
        clips = int(max(1, min(8, budget_s / max(per_audio_s * 10.0, 1e-3))))
        wav = torch.from_numpy(W.synth_waveform(clips, 240000, 24000, seed=1234))
        t0 = time.perf_counter()
        R.acoustic_encode(wt, wav, n_q)
        t_total = time.perf_counter() - t0
    return {"value": round(clips * 10.0 / t_total, 3), "unit": "audio-s/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{clips} clip(s) x 10 s @24 kHz in one batch, n_q={n_q}, oracle/encodec_ref.py (torch-CPU fp32), "
                      f"{t_total:.1f} s of CPU work"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="acoustic", choices=["acoustic"])
    ap.add_argument("--batch", type=int, default=256, help="clips per GPU per step (BASELINE config: 256)")
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--num-codebooks", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist_mod.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        dist = dist_mod
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch N>1 with torch.distributed.run)"

    from audiotoken_amd import weights as W
    from audiotoken_amd import _cabi
    from audiotoken_amd.configs import AcousticEncoderConfig, num_codebooks_to_bandwidth
    from audiotoken_amd.encoder import AcousticEncoder
    from audiotoken_amd.distributed import broadcast_weights

    n_q = args.num_codebooks
    B, N = args.batch, int(round(args.seconds * 24000))
    # weights: rank 0 generates, RCCL broadcast over xGMI to the other ranks
    weights = W.synth_encodec_weights(seed=0, with_decoder=False) if rank == 0 else None
    weights = broadcast_weights(weights, dev, dist)
    enc = AcousticEncoder(AcousticEncoderConfig(bandwidth=num_codebooks_to_bandwidth(n_q)), device=str(dev), weights=weights)
    # synthetic clips: rank r owns clips [r*B, (r+1)*B) of the global batch
    gen_B = min(B, 16)
    base = torch.from_numpy(W.synth_waveform(gen_B, N, 24000, seed=1234, first_clip=rank * B)).to(dev)
    wav = base.repeat((B + gen_B - 1) // gen_B, 1)[:B].contiguous()
    if B > gen_B:  # make repeated clips distinct without regenerating on the host
        wav = (wav * torch.linspace(0.5, 1.0, B, device=dev).unsqueeze(1)).contiguous()
    mask = torch.ones_like(wav)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        codes = enc(wav, mask)
    barrier()
    enc.enable_profile(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        codes = enc(wav, mask)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    if dist is not None:
        dist.barrier()
    elapsed = t1 - t0
    prof = enc.read_profile()  # {group: (total ms over the timed region, launches)}
    enc.enable_profile(False)
    if dist is not None:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    checksum = int(codes.to(torch.int64).sum().item())

    if rank == 0:
        audio_s = world * B * args.seconds * args.steps
        value = audio_s / elapsed
        flops, T = acoustic_flops_per_clip(N, n_q)
        nbytes = acoustic_bytes_per_clip(N, n_q)
        breakdown = {}
        for k, (ms, launches) in prof.items():
            per_step_ms = ms / args.steps
            breakdown[k] = {"ms_per_step": round(per_step_ms, 3), "launches_per_step": launches // args.steps,
                            "tflops": round(flops[k] * B / (per_step_ms * 1e-3) / 1e12, 2) if per_step_ms > 0 else None,
                            "gbs": round(nbytes[k] * B / (per_step_ms * 1e-3) / 1e9, 1) if per_step_ms > 0 else None}
        dom = max(breakdown, key=lambda k: breakdown[k]["ms_per_step"])
        d = breakdown[dom]
        # which roofline binds the dominant group: compare time at peak for its flops vs its bytes
        t_mfma = flops[dom] * B / (F32_MFMA_PEAK_TFLOPS * 1e12)
        t_hbm = nbytes[dom] * B / (HBM_PEAK_GBS * 1e9)
        launches = max(1, d["launches_per_step"])
        if t_hbm >= t_mfma:
            roof = {"bound": "hbm", "achieved": d["gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(d["gbs"] / HBM_PEAK_GBS, 4)}
        else:
            roof = {"bound": "mfma", "achieved": d["tflops"], "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(d["tflops"] / F32_MFMA_PEAK_TFLOPS, 4)}
        roof.update({"kernel": dom, "launches_per_step": launches, "avg_launch_ms": round(d["ms_per_step"] / launches, 4),
                     "traffic": None})
        out = {
            "metric": "audio-sec tokenized / wall-sec", "value": round(value, 2), "unit": "audio-s/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"Tokenizers.acoustic encode, {B} clips x {args.seconds:g} s @24 kHz per GPU, num_codebooks={n_q}",
                       "clips_per_gpu": B, "samples_per_clip": N, "frames_per_clip": T, "weights": "synthetic seed 0",
                       "parallelism": f"clip-sharded x{world}, no data-path collective"},
            "roofline": roof, "breakdown": breakdown, "token_checksum": checksum,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline_acoustic(n_q)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
