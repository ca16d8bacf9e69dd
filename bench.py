#!/usr/bin/env python3
"""Headline benchmark: audio-seconds tokenized per wall-second on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload both|acoustic|semantic_m] [--no-cpu-baseline]

One "step" = one pass of the hot path (the reference's ``self.encoder(input_batch, attention_mask)`` call,
audiotoken/core.py:276) over one synthetic batch that is already resident in HBM. At N=1 the workload is
BASELINE.json configs[1]: Tokenizers.acoustic, 256 clips x 10 s @ 24 kHz, 8 codebooks. With N>1 (launched by
torch.distributed.run, one rank per GPU) every rank encodes its own 256-clip shard — clips are independent, so
there is no data-path collective ("weak" scaling); RCCL is used only for the start barrier, the weight
broadcast check and the max-over-ranks time.

The metric names two tokenizers ("acoustic + semantic_m"). The top-level fields of the JSON line are the acoustic
workload (configs[1]); with --workload both (default) the same line carries a "semantic_m" object (BASELINE configs[3]
per-GPU share: 64 clips x 30 s @16 kHz, 19 conformer layers, VQ 2048) with its own value / roofline / cpu_baseline and
a "combined" figure = audio-seconds of both / (t_acoustic + t_semantic_m).

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
BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak (no sparsity)
# Linear layers of the two semantic tokenizers run as exact 3-way bf16 splits (6 bf16 MFMAs per fp32-equivalent step) unless disabled
BF16X3 = os.environ.get("AUDIOTOKEN_BF16X3", "1") != "0"
BF16X3_GROUPS = ("ffn", "attn_proj", "feature_extractor")   # feature_extractor: HuBERT only (its six 512->512 convs)
# acoustic kernel groups that execute as exact 3-way bf16 splits (library defaults; same switches as csrc/encodec.hip)
BF16X3_ACOUSTIC = os.environ.get("AUDIOTOKEN_BF16X3_ACOUSTIC", "1") != "0"
_X3_MASK = int(os.environ.get("AUDIOTOKEN_X3_KERNELS", "511"))
ACOUSTIC_X3_GROUPS = tuple(g for bit, g in enumerate(("down1", "res2", "res1", "stage0_fused", "down2", "down3", "res3", "lstm_rec", "rvq"))
                           if BF16X3_ACOUSTIC and (_X3_MASK >> bit) & 1) + (("lstm_ih",) if BF16X3_ACOUSTIC else ())


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
    g["stage0_fused"] = g["conv0"] + g["res0"] + g["down0"]
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
    g = {"conv0": 4.0 * N * (1 + 32), "stage0_fused": 4.0 * (N + L[1] * 64)}
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
        clips = int(max(1, min(48, budget_s / max(per_audio_s * 10.0, 1e-3))))
        wav = torch.from_numpy(W.synth_waveform(clips, 240000, 24000, seed=1234))
        t0 = time.perf_counter()
        R.acoustic_encode(wt, wav, n_q)
        t_total = time.perf_counter() - t0
    return {"value": round(clips * 10.0 / t_total, 3), "unit": "audio-s/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{clips} clip(s) x 10 s @24 kHz in one batch, n_q={n_q}, oracle/encodec_ref.py (torch-CPU fp32), "
                      f"{t_total:.1f} s of CPU work"}


def semantic_flops_per_clip(T: int, n_layers: int, F: int):
    """Algorithmic FLOPs per clip by kernel group (bucketed rel-pos)."""
    H, Fd = 1024, 4096
    g = {
        "frontend": 2.0 * F * (400 * 514 + 257 * 80),
        "feature_projection": 2.0 * T * 160 * H,
        "ffn": n_layers * 2 * (2.0 * T * H * Fd * 2),
        "attn_proj": n_layers * (2.0 * T * H * H * 4),
        "attention": n_layers * (4.0 * T * T * H + 2.0 * T * 80 * H),
        "conv_module": n_layers * (2.0 * T * H * 2 * H + 2.0 * T * H * H + 2.0 * T * 31 * H),
        "vq": 2.0 * T * H * 2048,
    }
    return g


def cpu_baseline_semantic(n_layers: int, budget_s: float = 20.0):
    """CPU oracle (oracle/w2vbert_ref.py, torch-CPU fp32 port of reference Wav2VecBertEncoder.forward on device='cpu')
    on a bounded sample: one clip whose length is sized by a 1 s probe to ~budget_s of CPU work (max 30 s audio).
    Uses the first min(n_layers, 4) synthetic layers' weights cyclically to bound host memory/time of weight
    generation — the arithmetic per layer is identical."""
    from audiotoken_amd import weights as W
    from oracle import w2vbert_ref as R

    torch.set_num_threads(host_threads())
    nl_w = min(n_layers, 2)
    w = W.synth_w2vbert_weights(n_layers=nl_w, seed=0, with_vq=True)
    wt = {k: torch.from_numpy(v) for k, v in w.items()}
    for i in range(nl_w, n_layers):   # alias layer i -> layer i % nl_w (no copies)
        for k in list(wt):
            if k.startswith(f"encoder.layers.{i % nl_w}."):
                wt[k.replace(f"encoder.layers.{i % nl_w}.", f"encoder.layers.{i}.", 1)] = wt[k]
    with torch.no_grad():
        probe = torch.from_numpy(W.synth_waveform(1, 16000, 16000, seed=1))
        R.semantic_m_encode(wt, probe, torch.ones_like(probe), 2, n_layers)
        t0 = time.perf_counter()
        R.semantic_m_encode(wt, probe, torch.ones_like(probe), 2, n_layers)
        per_s = time.perf_counter() - t0
        secs = float(max(1.0, min(30.0, budget_s / max(per_s, 1e-3))))
        n = int(secs * 16000)
        clips = int(max(1, min(4, budget_s / max(per_s * secs, 1e-3)))) if secs >= 30.0 else 1
        wav = torch.from_numpy(W.synth_waveform(clips, n, 16000, seed=1234))
        t0 = time.perf_counter()
        R.semantic_m_encode(wt, wav, torch.ones_like(wav), 2, n_layers)
        t_total = time.perf_counter() - t0
    return {"value": round(clips * n / 16000.0 / t_total, 3), "unit": "audio-s/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{clips} clip(s) x {n / 16000.0:.1f} s @16 kHz, {n_layers} conformer layers, oracle/w2vbert_ref.py (torch-CPU fp32), "
                      f"{t_total:.1f} s of CPU work"}


def hubert_flops_per_clip(N: int, n_layers: int):
    L = [N]
    for k, st in zip((10, 3, 3, 3, 3, 2, 2), (5, 2, 2, 2, 2, 2, 2)):
        L.append((L[-1] - k) // st + 1)
    T = L[7]
    fe = 2.0 * L[1] * 10 * 512 + sum(2.0 * L[i + 1] * k * 512 * 512 for i, k in zip(range(1, 7), (3, 3, 3, 3, 2, 2)))
    g = {
        "feature_extractor": fe,
        "projection_posconv": 2.0 * T * 512 * 768 + 2.0 * T * 768 * 128 * 48,
        "attn_proj": n_layers * 2.0 * T * 768 * 768 * 4,
        "attention": n_layers * 4.0 * T * T * 768,
        "ffn": n_layers * 2.0 * T * 768 * 3072 * 2,
        "kmeans": 2.0 * T * 768 * 1000,
    }
    return g, T


def run_hubert(args, rank, world, dev, dist):
    from audiotoken_amd import weights as W
    from audiotoken_amd.configs import HubertEncoderConfig
    from audiotoken_amd.hubert import HubertEncoder, hubert_processor
    from audiotoken_amd.distributed import broadcast_weights

    nl, B, secs = 11, args.hub_batch, args.sem_seconds
    N = int(round(secs * 16000))
    weights = W.synth_hubert_weights(n_layers=nl, seed=0, with_kmeans=True) if rank == 0 else None
    weights = broadcast_weights(weights, dev, dist)
    enc = HubertEncoder(HubertEncoderConfig(output_layer=nl), device=str(dev), quantize=True, weights=weights)
    del weights
    gen_B = min(B, 8)
    host = W.synth_waveform(gen_B, N, 16000, seed=1234, first_clip=rank * B)
    host = np.stack([hubert_processor(torch.from_numpy(host[i:i + 1]))[0].numpy() for i in range(gen_B)])
    wav = torch.from_numpy(host).to(dev).repeat((B + gen_B - 1) // gen_B, 1)[:B].contiguous()
    mask = torch.ones_like(wav)
    enc(wav, mask)
    for _ in range(max(0, args.warmup - 1)):
        enc(wav, mask)
    enc.enable_profile(True)
    elapsed, toks = timed_steps(lambda: enc(wav, mask), args.steps, 0, dist)
    prof = enc.read_profile()
    enc.enable_profile(False)
    elapsed = max_over_ranks(elapsed, dev, dist)
    flops, T = hubert_flops_per_clip(N, nl)
    breakdown = {}
    for k, (ms, launches) in prof.items():
        per = ms / args.steps
        breakdown[k] = {"ms_per_step": round(per, 3), "launches_per_step": launches // args.steps,
                        "tflops": round(flops[k] * B / (per * 1e-3) / 1e12, 2) if per > 0 else None, "gbs": None}
    res = {
        "value": round(world * B * secs * args.steps / elapsed, 2), "unit": "audio-s/s", "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "dtype": "f32 (linear layers: exact bf16x3 splits, fp32 accumulate)" if BF16X3 else "f32",
        "config": {"workload": f"Tokenizers.semantic_s encode, {B} clips x {secs:g} s @16 kHz per GPU, mHuBERT-base 11 layers, k-means 1000",
                   "clips_per_gpu": B, "samples_per_clip": N, "tokens_per_clip": T, "weights": "synthetic seed 0"},
        "roofline": roofline_of(breakdown, flops, None, B, BF16X3_GROUPS if BF16X3 else ()), "breakdown": breakdown,
        "token_checksum": int(toks.to(torch.int64).sum().item()),
        "total_tflops": round(sum(flops.values()) * B * args.steps / elapsed / 1e12, 2),
    }
    del enc
    torch.cuda.empty_cache()
    return res


def timed_steps(enc_call, steps, warmup, dist):
    for _ in range(warmup):
        out = enc_call()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = enc_call()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
    return elapsed, out


def max_over_ranks(x: float, dev, dist) -> float:
    if dist is None:
        return x
    t = torch.tensor([x], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def measured_traffic(group: str):
    """HBM bytes per launch of a kernel group from the committed rocprofv3 PMC passes (profiles/r01_traffic.json); None if
    that group was not profiled. PMC collection needs rocprofv3, so it cannot run inside the timed benchmark."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_traffic.json")) as f:
            return json.load(f)["kernels"][group]["traffic_bytes_per_launch"]
    except Exception:
        return None


def roofline_of(breakdown, flops, nbytes, B, split_groups=()):
    dom = max(breakdown, key=lambda k: breakdown[k]["ms_per_step"])
    d = breakdown[dom]
    if dom in split_groups:
        # executed arithmetic: six bf16 MFMAs per fp32-equivalent multiply-add, priced against the dense bf16 MFMA peak
        ach = round(6.0 * d["tflops"], 2)
        return {"bound": "mfma", "achieved": ach, "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / BF16_MFMA_PEAK_TFLOPS, 4),
                "kernel": dom, "launches_per_step": max(1, d["launches_per_step"]),
                "avg_launch_ms": round(d["ms_per_step"] / max(1, d["launches_per_step"]), 4), "traffic": measured_traffic(dom),
                "note": f"bf16 MFMA executing exact 3-way operand splits: {d['tflops']} fp32-equivalent TFLOP/s x 6 products"}
    t_mfma = flops[dom] * B / (F32_MFMA_PEAK_TFLOPS * 1e12)
    t_hbm = (nbytes[dom] * B / (HBM_PEAK_GBS * 1e9)) if nbytes is not None else 0.0
    launches = max(1, d["launches_per_step"])
    if t_hbm >= t_mfma:
        roof = {"bound": "hbm", "achieved": d["gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(d["gbs"] / HBM_PEAK_GBS, 4)}
    else:
        roof = {"bound": "mfma", "achieved": d["tflops"], "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(d["tflops"] / F32_MFMA_PEAK_TFLOPS, 4)}
    roof.update({"kernel": dom, "launches_per_step": launches, "avg_launch_ms": round(d["ms_per_step"] / launches, 4),
                 "traffic": measured_traffic(dom)})
    return roof


def run_acoustic(args, rank, world, dev, dist):
    from audiotoken_amd import weights as W
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
    enc(wav, mask)  # allocate workspace outside the timed region
    enc.enable_profile(False)
    # warm-up untimed, then the timed region with the HIP-event taps on
    for _ in range(args.warmup):
        enc(wav, mask)
    enc.enable_profile(True)
    elapsed, codes = timed_steps(lambda: enc(wav, mask), args.steps, 0, dist)
    prof = enc.read_profile()
    enc.enable_profile(False)
    elapsed = max_over_ranks(elapsed, dev, dist)
    checksum = int(codes.to(torch.int64).sum().item())
    flops, T = acoustic_flops_per_clip(N, n_q)
    nbytes = acoustic_bytes_per_clip(N, n_q)
    breakdown = {}
    for k, (ms, launches) in prof.items():
        per = ms / args.steps
        breakdown[k] = {"ms_per_step": round(per, 3), "launches_per_step": launches // args.steps,
                        "tflops": round(flops[k] * B / (per * 1e-3) / 1e12, 2) if per > 0 else None,
                        "gbs": round(nbytes[k] * B / (per * 1e-3) / 1e9, 1) if per > 0 else None}
    res = {
        "value": round(world * B * args.seconds * args.steps / elapsed, 2), "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "elapsed": elapsed, "audio_s_per_step": world * B * args.seconds,
        "config": {"workload": f"Tokenizers.acoustic encode, {B} clips x {args.seconds:g} s @24 kHz per GPU, num_codebooks={n_q}",
                   "clips_per_gpu": B, "samples_per_clip": N, "frames_per_clip": T, "weights": "synthetic seed 0",
                   "parallelism": f"clip-sharded x{world}, no data-path collective"},
        "roofline": roofline_of(breakdown, flops, nbytes, B, ACOUSTIC_X3_GROUPS), "breakdown": breakdown, "token_checksum": checksum,
    }
    # PCIe-inclusive rate (host waveforms in pinned memory -> H2D -> encode -> D2H tokens), reported beside `value`, never as it
    try:
        host = wav.cpu().pin_memory()
        out_host = torch.empty(codes.shape, dtype=codes.dtype).pin_memory()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2):
            d = host.to(dev, non_blocking=True)
            out_host.copy_(enc(d, None), non_blocking=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 2
        res["pcie_inclusive"] = {"value": round(world * B * args.seconds / dt, 2), "unit": "audio-s/s", "ms_per_step": round(dt * 1e3, 3),
                                 "note": "pinned host waveforms -> H2D -> encode -> D2H tokens, serialized on one stream"}
        del host, out_host, d
    except Exception as e:  # pragma: no cover - informational only
        res["pcie_inclusive"] = {"error": f"{type(e).__name__}: {e}"}
    del enc
    torch.cuda.empty_cache()
    return res


def run_decode(args, rank, world, dev, dist):
    """BASELINE configs[4] (C5): 64 clips x 10 s of acoustic tokens -> waveform through the HIP decoder (A11). Extra line only."""
    from audiotoken_amd import weights as W
    from audiotoken_amd.configs import AcousticDecoderConfig, num_codebooks_to_bandwidth
    from audiotoken_amd.decoder import AcousticDecoder
    from audiotoken_amd.distributed import broadcast_weights

    weights = W.synth_encodec_weights(seed=0) if rank == 0 else None
    weights = broadcast_weights(weights, dev, dist)
    dec = AcousticDecoder(config=AcousticDecoderConfig(bandwidth=num_codebooks_to_bandwidth(args.num_codebooks)), device=str(dev), weights=weights)
    B, T = 64, int(round(args.seconds * 75))
    g = torch.Generator().manual_seed(1234 + rank)
    codes = torch.randint(0, 1024, (B, args.num_codebooks, T), generator=g, dtype=torch.long).to(dev)
    dec(codes)
    elapsed, out = timed_steps(lambda: dec(codes), args.steps, args.warmup, dist)
    elapsed = max_over_ranks(elapsed, dev, dist)
    res = {"value": round(world * B * args.seconds * args.steps / elapsed, 2), "unit": "audio-s/s", "ms_per_step": round(elapsed / args.steps * 1e3, 3),
           "config": {"workload": f"Tokenizers.acoustic decode, {B} clips x {args.seconds:g} s, num_codebooks={args.num_codebooks}"},
           "checksum": float(out.double().abs().sum().item())}
    del dec
    torch.cuda.empty_cache()
    return res


def run_semantic(args, rank, world, dev, dist):
    from audiotoken_amd import weights as W
    from audiotoken_amd.configs import Wav2VecBertConfig
    from audiotoken_amd.encoder import Wav2VecBertEncoder
    from audiotoken_amd.distributed import broadcast_weights

    nl = args.sem_layers
    B, secs = args.sem_batch, args.sem_seconds
    N = int(round(secs * 16000))
    weights = W.synth_w2vbert_weights(n_layers=nl, seed=0, with_vq=True) if rank == 0 else None
    weights = broadcast_weights(weights, dev, dist)
    enc = Wav2VecBertEncoder(Wav2VecBertConfig(output_layer=nl), device=str(dev), quantize=True, weights=weights)
    del weights
    gen_B = min(B, 8)
    base = torch.from_numpy(W.synth_waveform(gen_B, N, 16000, seed=1234, first_clip=rank * B)).to(dev)
    wav = base.repeat((B + gen_B - 1) // gen_B, 1)[:B].contiguous()
    if B > gen_B:
        wav = (wav * torch.linspace(0.5, 1.0, B, device=dev).unsqueeze(1)).contiguous()
    mask = torch.ones_like(wav)
    enc(wav, mask)
    for _ in range(max(0, args.warmup - 1)):
        enc(wav, mask)
    enc.enable_profile(True)
    elapsed, toks = timed_steps(lambda: enc(wav, mask), args.steps, 0, dist)
    prof = enc.read_profile()
    enc.enable_profile(False)
    elapsed = max_over_ranks(elapsed, dev, dist)
    T = toks.shape[-1]
    F = 1 + (N - 400) // 160
    flops = semantic_flops_per_clip(T, nl, F)
    breakdown = {}
    for k, (ms, launches) in prof.items():
        per = ms / args.steps
        breakdown[k] = {"ms_per_step": round(per, 3), "launches_per_step": launches // args.steps,
                        "tflops": round(flops[k] * B / (per * 1e-3) / 1e12, 2) if per > 0 else None, "gbs": None}
    res = {
        "value": round(world * B * secs * args.steps / elapsed, 2), "unit": "audio-s/s", "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "elapsed": elapsed, "audio_s_per_step": world * B * secs,
        "dtype": "f32 (linear layers: exact bf16x3 splits, fp32 accumulate)" if BF16X3 else "f32",
        "config": {"workload": f"Tokenizers.semantic_m encode, {B} clips x {secs:g} s @16 kHz per GPU, {nl} conformer layers, VQ 2048x1024",
                   "clips_per_gpu": B, "samples_per_clip": N, "tokens_per_clip": T, "weights": "synthetic seed 0",
                   "parallelism": f"clip-sharded x{world}, no data-path collective"},
        "roofline": roofline_of(breakdown, flops, None, B, BF16X3_GROUPS if BF16X3 else ()), "breakdown": breakdown,
        "token_checksum": int(toks.to(torch.int64).sum().item()),
        "total_tflops": round(sum(flops.values()) * B * args.steps / elapsed / 1e12, 2),
    }
    del enc
    torch.cuda.empty_cache()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="all", choices=["all", "both", "acoustic", "semantic_m", "semantic_s"])
    ap.add_argument("--hub-batch", type=int, default=128, help="semantic_s clips per GPU per step (BASELINE configs[2]: 128)")
    ap.add_argument("--batch", type=int, default=256, help="acoustic clips per GPU per step (BASELINE configs[1]: 256)")
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--num-codebooks", type=int, default=8)
    ap.add_argument("--sem-batch", type=int, default=64, help="semantic_m clips per GPU per step (BASELINE configs[3]: 512/8)")
    ap.add_argument("--sem-seconds", type=float, default=30.0)
    ap.add_argument("--sem-layers", type=int, default=19)
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

    ac = sem = None
    sem_err = None
    hub = hub_err = None
    if args.workload in ("all", "both", "acoustic"):
        ac = run_acoustic(args, rank, world, dev, dist)
    dec = None
    if args.workload == "all":
        try:
            dec = run_decode(args, rank, world, dev, dist)
        except Exception as e:
            dec = {"error": f"{type(e).__name__}: {e}"}
    if args.workload in ("all", "semantic_s"):
        try:
            hub = run_hubert(args, rank, world, dev, dist)
        except Exception as e:
            if args.workload == "semantic_s":
                raise
            hub_err = f"{type(e).__name__}: {e}"
    if args.workload in ("all", "both", "semantic_m"):
        try:
            sem = run_semantic(args, rank, world, dev, dist)
        except Exception as e:  # keep the acoustic line even if the second workload cannot run on this box
            if args.workload == "semantic_m":
                raise
            sem_err = f"{type(e).__name__}: {e}"

    if rank == 0:
        primary = ac if ac is not None else (sem if sem is not None else hub)
        out = {
            "metric": "audio-sec tokenized / wall-sec", "value": primary["value"], "unit": "audio-s/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": primary["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (convs, LSTM, RVQ dot products: exact bf16x3 splits, fp32 accumulate)" if ACOUSTIC_X3_GROUPS else "f32", "data": "synthetic",
            "config": primary["config"], "roofline": primary["roofline"], "breakdown": primary["breakdown"],
            "token_checksum": primary["token_checksum"],
        }
        if "pcie_inclusive" in primary:
            out["pcie_inclusive"] = primary["pcie_inclusive"]
        if not args.no_cpu_baseline and world == 1:
            if ac is not None:
                out["cpu_baseline"] = cpu_baseline_acoustic(args.num_codebooks)
            elif sem is not None:
                out["cpu_baseline"] = cpu_baseline_semantic(args.sem_layers)
        if ac is not None and sem is not None:
            s = {k: v for k, v in sem.items() if k not in ("elapsed", "audio_s_per_step")}
            if not args.no_cpu_baseline and world == 1:
                s["cpu_baseline"] = cpu_baseline_semantic(args.sem_layers)
            out["semantic_m"] = s
            tot_audio = (ac["audio_s_per_step"] + sem["audio_s_per_step"]) * args.steps
            out["combined"] = {"value": round(tot_audio / (ac["elapsed"] + sem["elapsed"]), 2), "unit": "audio-s/s",
                               "definition": "audio-seconds of both workloads / (t_acoustic + t_semantic_m)"}
        elif sem_err:
            out["semantic_m"] = {"error": sem_err}
        if dec is not None:
            out["acoustic_decode"] = dec
        if hub is not None and primary is not hub:
            out["semantic_s"] = hub
        elif hub_err:
            out["semantic_s"] = {"error": hub_err}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
