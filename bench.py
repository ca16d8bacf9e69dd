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

Multi-GPU: `python bench.py --gpus N` with no WORLD_SIZE in the environment starts N FRESH child ranks itself
(`python -m torch.distributed.run --nproc-per-node N ... bench.py ...`, before this process has touched the GPU) and relays rank 0's
JSON line; under an external `torch.distributed.run` (WORLD_SIZE set) it is simply one of the ranks.

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
# kernel groups behind each option of the two-piece fp16 scheme (three MFMA products per multiply-add instead of six); the live
# option values are read back from the handle (at_encodec_get_option)
ACOUSTIC_F16X2_OPTIONS = {"chain_f16x2": ("down2", "res3", "down3"), "ih_f16x2": ("lstm_ih",), "lstm_f16x2": ("lstm_rec",),
                          "res_f16x2": ("stage0_fused", "res1", "down1", "res2"), "rvq_f16x2": ("rvq",), "fin_f16x2": ("final_conv",)}


def acoustic_f16x2_groups(enc):
    return tuple(g for opt, groups in ACOUSTIC_F16X2_OPTIONS.items() if enc.get_option(opt) == 1 for g in groups if g in ACOUSTIC_X3_GROUPS or g == "final_conv")


def free_port() -> int:
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def launch_children(n: int, argv, script: str = None, extra_env=None, timeout: float = None):
    """Start `n` fresh ranks of this script (one process per GPU) with torch.distributed.run on 127.0.0.1 and relay their output.
    Called BEFORE the parent has initialised the GPU; the parent never re-execs, it waits for the children and returns
    (exit code, last JSON line printed by rank 0 or None)."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), script or os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    env.update(extra_env or {})
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True, timeout=timeout)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
    return proc.returncode, line


def init_ranks(backend: str, dev):
    """(rank, world, dist-or-None) from the torch.distributed.run environment; backend "nccl" is RCCL on ROCm."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world == 1:
        return rank, world, None
    import torch.distributed as dist_mod
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend == "nccl":
        dist_mod.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist_mod.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, dist_mod


def rank_report(rank: int, per_rank_ms: float, dev, dist):
    """All-gather (rank id, ms per step) so the JSON line shows that N ranks really ran: {"rccl_ranks": [...], "per_rank_ms": [...]}."""
    from audiotoken_amd.distributed import gather_scalars
    rows = gather_scalars([float(rank), float(per_rank_ms)], dev, dist)
    return {"rccl_ranks": [int(r[0]) for r in rows], "per_rank_ms": [round(r[1], 3) for r in rows]}


def run_selftest(args, rank, world, dev, dist):
    """Launcher / collective plumbing only (no encoder, runs on CPU with gloo): weight broadcast, barrier-bracketed timed region,
    max over ranks, rank all-gather. Used by tests/test_bench_launcher_cpu.py; never part of a reported number."""
    from audiotoken_amd.distributed import broadcast_weights, shard_indices
    w = {"a": np.arange(12, dtype=np.float32).reshape(3, 4), "b": np.ones(5, dtype=np.float32)} if rank == 0 else None
    t0 = time.perf_counter()
    w = broadcast_weights(w, dev, dist)
    bcast_ms = (time.perf_counter() - t0) * 1e3
    assert float(w["a"].sum()) == 66.0 and w["b"].shape == (5,)
    mine = shard_indices(args.batch, rank, world)
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    acc = 0
    for _ in range(args.steps):
        acc += sum(mine)
    elapsed = max(time.perf_counter() - t0, 1e-6)
    if dist is not None:
        dist.barrier()
    ms = elapsed / args.steps * 1e3
    elapsed = max_over_ranks(elapsed, dev, dist)
    res = {"value": round(world * len(mine) * args.steps / elapsed, 2), "ms_per_step": round(elapsed / args.steps * 1e3, 6),
           "config": {"workload": "selftest (launcher and collectives only)", "items_per_rank": len(mine)},
           "roofline": None, "breakdown": {}, "token_checksum": int(acc), "broadcast_ms": round(bcast_ms, 3), "rank_ms": ms}
    return res


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
    enc.enable_profile(False)
    elapsed, toks, per_step = timed_steps(lambda: enc(wav, mask), args.steps, max(0, args.warmup - 1), dist)
    elapsed = max_over_ranks(elapsed, dev, dist)
    prof = tapped_breakdown(enc, lambda: enc(wav, mask), min(args.steps, 2))
    flops, T = hubert_flops_per_clip(N, nl)
    arith = enc.get_option("arith")
    products = {0: 1, 1: 6, 2: 3}[arith]
    assert enc.last_status() == 0, "semantic_s status word non-zero (fp16 range overflow): the timed run is invalid"
    breakdown = {}
    for k, (per, launches) in prof.items():
        breakdown[k] = {"ms_per_step": round(per, 3), "launches_per_step": launches,
                        "tflops": round(flops[k] * B / (per * 1e-3) / 1e12, 2) if per > 0 else None, "gbs": None}
    res = {
        "value": round(world * B * secs * args.steps / elapsed, 2), "unit": "audio-s/s", "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "dtype": {0: "f32", 1: "f32 (linear layers and convs: three bf16 pieces per operand, six MFMA products, fp32 accumulate)",
                  2: "f32 (linear layers and convs: two fp16 pieces per operand, three MFMA products, fp32 accumulate)"}[arith],
        "config": {"workload": f"Tokenizers.semantic_s encode, {B} clips x {secs:g} s @16 kHz per GPU, mHuBERT-base 11 layers, k-means 1000",
                   "clips_per_gpu": B, "samples_per_clip": N, "tokens_per_clip": T, "weights": "synthetic seed 0"},
        "roofline": roofline_of(breakdown, flops, None, B, BF16X3_GROUPS if arith else (), "semantic_s", products), "breakdown": breakdown,
        "token_checksum": int(toks.to(torch.int64).sum().item()),
        "total_tflops": round(sum(flops.values()) * B * args.steps / elapsed / 1e12, 2),
    }
    del enc
    torch.cuda.empty_cache()
    return res


def timed_steps(enc_call, steps, warmup, dist):
    """W untimed warm-up steps, then EXACTLY `steps` steps bracketed by barrier + synchronize on both sides (the contract's timed
    region). Also returns the per-step device times from HIP events on the launch stream (for the median)."""
    for _ in range(warmup):
        out = enc_call()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(steps):
        out = enc_call()
        evs[i + 1].record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
    per_step = [evs[i].elapsed_time(evs[i + 1]) for i in range(steps)]
    return elapsed, out, per_step


def tapped_breakdown(enc, enc_call, steps: int):
    """A second, short loop with the library's HIP-event taps on (they are OFF in the timed region): ms per kernel group."""
    enc.enable_profile(True)
    for _ in range(steps):
        enc_call()
    torch.cuda.synchronize()
    prof = enc.read_profile()
    enc.enable_profile(False)
    return {k: (ms / steps, launches // steps) for k, (ms, launches) in prof.items()}


def pipelined_pcie(enc_call, host_in: torch.Tensor, dev, iters: int):
    """SURVEY.md §8(d) wall definition: pinned host waveforms -> H2D -> encode -> D2H tokens, pipelined the way
    AudioToken.encode_batch_files runs it (core.py): the copy of batch i+1 flies on a copy stream during the encode of batch i,
    tokens come back on the compute stream. Returns the median and mean time per batch over `iters` batches in steady state."""
    copy_stream = torch.cuda.Stream(device=dev)
    main = torch.cuda.current_stream(dev)
    bufs = [torch.empty_like(host_in, device=dev) for _ in range(2)]
    ready = [torch.cuda.Event() for _ in range(2)]
    free = [torch.cuda.Event() for _ in range(2)]
    out_host = None
    done = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]

    def upload(i):
        with torch.cuda.stream(copy_stream):
            copy_stream.wait_event(free[i % 2])
            bufs[i % 2].copy_(host_in, non_blocking=True)
            ready[i % 2].record(copy_stream)

    for f in free:
        f.record(main)
    upload(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    done[0].record(main)
    for i in range(iters):
        main.wait_event(ready[i % 2])
        if i + 1 < iters:
            upload(i + 1)
        toks = enc_call(bufs[i % 2])
        free[i % 2].record(main)
        if out_host is None:
            out_host = torch.empty(toks.shape, dtype=toks.dtype).pin_memory()
        out_host.copy_(toks, non_blocking=True)
        done[i + 1].record(main)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    per = sorted(done[i].elapsed_time(done[i + 1]) for i in range(iters))
    return {"median_ms": per[len(per) // 2], "mean_ms": wall / iters * 1e3, "iters": iters}


def max_over_ranks(x: float, dev, dist) -> float:
    if dist is None:
        return x
    t = torch.tensor([x], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def measured_traffic(group: str, workload: str = "acoustic"):
    """HBM bytes per launch of a kernel group from the committed rocprofv3 PMC passes (the newest profiles/r02_*_traffic.json, falling back to
    round 1's); None if that group was not profiled. PMC collection needs rocprofv3, so it cannot run inside the timed benchmark."""
    for name in ("r02_final_traffic.json", "r02_v2_traffic.json", "r01_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                doc = json.load(f)
            k = doc.get(workload, doc).get("kernels", doc.get("kernels", {}))
            if group in k:
                return k[group]["traffic_bytes_per_launch"]
        except Exception:
            continue
    return None


def roofline_of(breakdown, flops, nbytes, B, split_groups=(), workload: str = "acoustic", products: int = 6):
    """Roofline of the dominant kernel group. `achieved` is ALGORITHMIC work (SURVEY.md §8(d) FLOPs or bytes per clip x clips per
    launch) / the group's measured time. For groups that run as exact operand splits on the bf16/fp16 matrix cores the same
    object also carries the EXECUTED rate (`products` MFMA products per multiply-add): that one is pipe utilisation, not work."""
    dom = max(breakdown, key=lambda k: breakdown[k]["ms_per_step"])
    d = breakdown[dom]
    launches = max(1, d["launches_per_step"])
    common = {"kernel": dom, "launches_per_step": launches, "avg_launch_ms": round(d["ms_per_step"] / launches, 4),
              "traffic": measured_traffic(dom, workload)}
    if dom in split_groups:
        alg = d["tflops"]
        ex = round(products * alg, 2)
        roof = {"bound": "mfma", "achieved": alg, "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(alg / BF16_MFMA_PEAK_TFLOPS, 4),
                "achieved_algorithmic": alg, "achieved_executed": ex, "products_per_mac": products,
                "frac_executed": round(ex / BF16_MFMA_PEAK_TFLOPS, 4), "frac_algorithmic_vs_f32_mfma_peak": round(alg / F32_MFMA_PEAK_TFLOPS, 4),
                "note": f"fp32-grade contraction as exact operand splits on the 16-bit matrix cores: {products} MFMA products per multiply-add; "
                        "`achieved`/`frac` count the algorithmic FLOPs against the dense bf16 peak, `*_executed` the issued MFMA FLOPs"}
        roof.update(common)
        return roof
    t_mfma = flops[dom] * B / (F32_MFMA_PEAK_TFLOPS * 1e12)
    t_hbm = (nbytes[dom] * B / (HBM_PEAK_GBS * 1e9)) if nbytes is not None else 0.0
    if t_hbm >= t_mfma:
        roof = {"bound": "hbm", "achieved": d["gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(d["gbs"] / HBM_PEAK_GBS, 4)}
    else:
        roof = {"bound": "mfma", "achieved": d["tflops"], "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(d["tflops"] / F32_MFMA_PEAK_TFLOPS, 4), "achieved_algorithmic": d["tflops"], "achieved_executed": d["tflops"]}
    roof.update(common)
    return roof


def median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2]


def run_acoustic(args, rank, world, dev, dist):
    from audiotoken_amd import weights as W
    from audiotoken_amd.configs import AcousticEncoderConfig, num_codebooks_to_bandwidth
    from audiotoken_amd.encoder import AcousticEncoder
    from audiotoken_amd.distributed import broadcast_weights

    n_q = args.num_codebooks
    B, N = args.batch, int(round(args.seconds * 24000))
    # weights: rank 0 generates, RCCL broadcast over xGMI to the other ranks
    weights = W.synth_encodec_weights(seed=0, with_decoder=False) if rank == 0 else None
    t0 = time.perf_counter()
    weights = broadcast_weights(weights, dev, dist)
    bcast_ms = (time.perf_counter() - t0) * 1e3
    enc = AcousticEncoder(AcousticEncoderConfig(bandwidth=num_codebooks_to_bandwidth(n_q)), device=str(dev), weights=weights)
    # synthetic clips: rank r owns clips [r*B, (r+1)*B) of the global batch
    gen_B = min(B, 16)
    base = torch.from_numpy(W.synth_waveform(gen_B, N, 24000, seed=1234, first_clip=rank * B)).to(dev)
    wav = base.repeat((B + gen_B - 1) // gen_B, 1)[:B].contiguous()
    if B > gen_B:  # make repeated clips distinct without regenerating on the host
        wav = (wav * torch.linspace(0.5, 1.0, B, device=dev).unsqueeze(1)).contiguous()
    mask = torch.ones_like(wav)
    enc(wav, mask)  # allocate workspace outside the timed region
    enc.enable_profile(False)   # no event taps inside the timed region
    elapsed, codes, per_step = timed_steps(lambda: enc(wav, mask), args.steps, args.warmup, dist)
    status = enc.last_status()
    assert status == 0, f"persistent LSTM hand-off status {status}: the timed run is invalid"
    rank_ms = elapsed / args.steps * 1e3
    elapsed = max_over_ranks(elapsed, dev, dist)
    checksum = int(codes.to(torch.int64).sum().item())
    prof = tapped_breakdown(enc, lambda: enc(wav, mask), min(args.steps, 3))
    flops, T = acoustic_flops_per_clip(N, n_q)
    nbytes = acoustic_bytes_per_clip(N, n_q)
    breakdown = {}
    for k, (per, launches) in prof.items():
        breakdown[k] = {"ms_per_step": round(per, 3), "launches_per_step": launches,
                        "tflops": round(flops[k] * B / (per * 1e-3) / 1e12, 2) if per > 0 else None,
                        "gbs": round(nbytes[k] * B / (per * 1e-3) / 1e9, 1) if per > 0 else None}
    f16_groups = acoustic_f16x2_groups(enc)
    res = {
        "value": round(world * B * args.seconds * args.steps / elapsed, 2), "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "median_ms_per_step": round(median(per_step), 3),
        "elapsed": elapsed, "audio_s_per_step": world * B * args.seconds, "rank_ms": rank_ms, "broadcast_ms": round(bcast_ms, 1),
        "config": {"workload": f"Tokenizers.acoustic encode, {B} clips x {args.seconds:g} s @24 kHz per GPU, num_codebooks={n_q}",
                   "clips_per_gpu": B, "samples_per_clip": N, "frames_per_clip": T, "weights": "synthetic seed 0",
                   "parallelism": f"clip-sharded x{world}, no data-path collective"},
        "roofline": roofline_of(breakdown, flops, nbytes, B, ACOUSTIC_X3_GROUPS, "acoustic",
                                3 if max(breakdown, key=lambda k: breakdown[k]["ms_per_step"]) in f16_groups else 6), "breakdown": breakdown,
        "mfma_products_per_mac": {**{g: (3 if g in f16_groups else 6) for g in ACOUSTIC_X3_GROUPS}, "final_conv": 3 if "final_conv" in f16_groups else 1},
        "breakdown_note": "HIP-event taps of a second short loop (taps are off in the timed region)", "token_checksum": checksum,
        "lstm_handoff_status": status,
    }
    # SURVEY.md §8(d) wall (first H2D enqueue -> last token D2H), pipelined as encode_batch_files runs it. Reported beside `value`
    # (which, by the bench contract, is the rate with inputs resident in HBM), never as it.
    try:
        host = wav.cpu().pin_memory()
        pp = pipelined_pcie(lambda d: enc(d, None), host, dev, max(10, args.steps))
        res["pcie_inclusive"] = {"value": round(world * B * args.seconds / (pp["median_ms"] * 1e-3), 2), "unit": "audio-s/s",
                                 "median_ms_per_step": round(pp["median_ms"], 3), "mean_ms_per_step": round(pp["mean_ms"], 3), "iters": pp["iters"],
                                 "note": "pinned host waveforms -> H2D on a copy stream during the previous encode -> encode -> D2H tokens; median over batches"}
        del host
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
    elapsed, out, _ = timed_steps(lambda: dec(codes), args.steps, args.warmup, dist)
    if hasattr(dec, "last_status"):
        assert dec.last_status() == 0, "persistent LSTM hand-off status non-zero: the timed decode is invalid"
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
    enc.enable_profile(False)
    elapsed, toks, per_step = timed_steps(lambda: enc(wav, mask), args.steps, max(0, args.warmup - 1), dist)
    rank_ms = elapsed / args.steps * 1e3
    elapsed = max_over_ranks(elapsed, dev, dist)
    prof = tapped_breakdown(enc, lambda: enc(wav, mask), min(args.steps, 2))
    T = toks.shape[-1]
    F = 1 + (N - 400) // 160
    flops = semantic_flops_per_clip(T, nl, F)
    arith = enc.get_option("arith")                    # 0 f32 MFMA, 1 bf16x3 (six products), 2 f16x2 (three products)
    products = {0: 1, 1: 6, 2: 3}[arith]
    assert enc.last_status() == 0, "semantic_m status word non-zero (fp16 range overflow): the timed run is invalid"
    breakdown = {}
    ln_bytes = {"layernorm": 8.0 * T * 1024 * (4 * nl) + 8.0 * T * 1024 * nl}   # LayerNorm -> pieces: 4 B in + 4 B out per element; final LN 4 + 4
    for k, (per, launches) in prof.items():
        breakdown[k] = {"ms_per_step": round(per, 3), "launches_per_step": launches,
                        "tflops": round(flops[k] * B / (per * 1e-3) / 1e12, 2) if per > 0 and k in flops else None,
                        "gbs": round(ln_bytes[k] * B / (per * 1e-3) / 1e9, 1) if per > 0 and k in ln_bytes else None}
    flops = dict(flops, layernorm=0.0)
    res = {
        "value": round(world * B * secs * args.steps / elapsed, 2), "unit": "audio-s/s", "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "median_ms_per_step": round(median(per_step), 3), "elapsed": elapsed, "audio_s_per_step": world * B * secs, "rank_ms": rank_ms,
        "dtype": {0: "f32", 1: "f32 (linear layers: three bf16 pieces per operand, six MFMA products, fp32 accumulate)",
                  2: "f32 (linear layers: two fp16 pieces per operand, three MFMA products, fp32 accumulate)"}[arith],
        "config": {"workload": f"Tokenizers.semantic_m encode, {B} clips x {secs:g} s @16 kHz per GPU, {nl} conformer layers, VQ 2048x1024",
                   "clips_per_gpu": B, "samples_per_clip": N, "tokens_per_clip": T, "weights": "synthetic seed 0",
                   "parallelism": f"clip-sharded x{world}, no data-path collective",
                   "note": "BASELINE configs[3] is 512 clips over 8 GPUs = 64 per GPU; at N=1 one step is one such 64-clip micro-batch"},
        "roofline": roofline_of(breakdown, flops, None, B, ("ffn", "attn_proj", "conv_module") if arith else (), "semantic_m", products), "breakdown": breakdown,
        "breakdown_note": "HIP-event taps of a second short loop (taps are off in the timed region)",
        "token_checksum": int(toks.to(torch.int64).sum().item()),
        "total_tflops": round(sum(flops.values()) * B * args.steps / elapsed / 1e12, 2),
    }
    del enc
    torch.cuda.empty_cache()
    return res


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="all", choices=["all", "both", "acoustic", "semantic_m", "semantic_s", "selftest"])
    ap.add_argument("--hub-batch", type=int, default=128, help="semantic_s clips per GPU per step (BASELINE configs[2]: 128)")
    ap.add_argument("--batch", type=int, default=256, help="acoustic clips per GPU per step (BASELINE configs[1]: 256)")
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--num-codebooks", type=int, default=8)
    ap.add_argument("--sem-batch", type=int, default=64, help="semantic_m clips per GPU per step (BASELINE configs[3]: 512/8)")
    ap.add_argument("--sem-seconds", type=float, default=30.0)
    ap.add_argument("--sem-layers", type=int, default=19)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo + --workload selftest run on CPU (tests)")
    return ap.parse_args(argv)


def main(argv=None):
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # self-launch: N fresh child processes, one per GPU; this parent has not touched the GPU and only relays the result
        if args.backend == "nccl":
            n_dev = torch.cuda.device_count()   # counting devices does not initialise the GPU
            if n_dev < args.gpus:
                print(f"bench.py: --gpus {args.gpus} but this node exposes {n_dev} device(s)", file=sys.stderr)
                return 2
        rc, line = launch_children(args.gpus, sys.argv[1:] if argv is None else list(argv))
        if line is not None:
            print(line, flush=True)
        return rc if rc != 0 or line is not None else 1

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.workload == "selftest" and args.backend == "gloo":
        dev = torch.device("cpu")
    else:
        assert torch.cuda.is_available(), "bench.py needs a HIP device"
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    rank, world, dist = init_ranks(args.backend, dev)
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    ac = sem = None
    sem_err = None
    hub = hub_err = None
    if args.workload == "selftest":
        ac = run_selftest(args, rank, world, dev, dist)
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

    primary = ac if ac is not None else (sem if sem is not None else hub)
    ranks = rank_report(rank, primary.get("rank_ms", primary["ms_per_step"]), dev, dist)   # collective: every rank calls it
    if rank == 0:
        out = {
            "metric": "audio-sec tokenized / wall-sec", "value": primary["value"], "unit": "audio-s/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": primary["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("f32 (contractions as operand splits on the 16-bit matrix cores with fp32 accumulate; per kernel group two fp16 pieces / three "
                      "products or three bf16 pieces / six products: see mfma_products_per_mac; final conv on the fp32 MFMA)")
                     if ACOUSTIC_X3_GROUPS else "f32", "data": "synthetic",
            "config": primary["config"], "roofline": primary["roofline"], "breakdown": primary["breakdown"],
            "token_checksum": primary["token_checksum"],
            "rccl_ranks": ranks["rccl_ranks"], "per_rank_ms": ranks["per_rank_ms"],
            "backend": args.backend if world > 1 else None,
        }
        for k in ("median_ms_per_step", "pcie_inclusive", "broadcast_ms", "breakdown_note", "lstm_handoff_status", "mfma_products_per_mac"):
            if k in primary:
                out[k] = primary[k]
        want_cpu = not args.no_cpu_baseline and world == 1 and args.workload != "selftest"
        if not want_cpu:
            out["cpu_baseline"] = None
            out["cpu_baseline_note"] = ("skipped: --no-cpu-baseline" if args.no_cpu_baseline else
                                        "skipped at N>1: the CPU oracle is timed on rank 0 at N=1 only")
        else:
            if ac is not None:
                out["cpu_baseline"] = cpu_baseline_acoustic(args.num_codebooks)
            elif sem is not None:
                out["cpu_baseline"] = cpu_baseline_semantic(args.sem_layers)
        if ac is not None and sem is not None:
            s = {k: v for k, v in sem.items() if k not in ("elapsed", "audio_s_per_step", "rank_ms")}
            if want_cpu:
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
    return 0


if __name__ == "__main__":
    sys.exit(main())
