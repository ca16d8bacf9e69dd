#!/usr/bin/env python3
"""8-rank readiness without an 8-GPU box: N ranks on ONE device (gloo through host memory) run `AudioToken.encode_batch_files` on ONE
shared directory of mixed-length files. Checks, on hardware: the LPT sharding (whole files by size, every rank the same answer from rank 0's stat), disjoint and
complete outputs, token files byte-identical to a single-process run, the device feeder on N processes x `num_workers` decode threads at once. Reports the wall
seconds and the host CPU seconds (time.process_time) per rank and max / min over ranks. NOT a throughput measurement: N processes share one GPU.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29533 tools/n8_shared_dir.py <workdir> [tokenizer] [n_files]"""
import filecmp
import os
import shutil
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
from scipy.io import wavfile

from audiotoken_amd import AudioToken, Tokenizers
from audiotoken_amd import synthetic as S
from audiotoken_amd import weights as W
from audiotoken_amd.distributed import gather_scalars

work = sys.argv[1]
which = sys.argv[2] if len(sys.argv) > 2 else "acoustic"
n_files = int(sys.argv[3]) if len(sys.argv) > 3 else 96
workers = int(sys.argv[4]) if len(sys.argv) > 4 else 8
rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
dist.init_process_group("gloo", rank=rank, world_size=world)
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
sr = 24000 if which == "acoustic" else 16000
src_dir, out_n, out_1 = os.path.join(work, "in"), os.path.join(work, f"out_world{world}"), os.path.join(work, "out_world1")
if rank == 0:
    shutil.rmtree(work, ignore_errors=True)
    os.makedirs(src_dir)
    rng = np.random.default_rng(5)
    for i in range(n_files):
        secs = float(rng.choice([2.0, 3.5, 7.0, 11.0, 19.0, 31.0, 47.0], p=[0.2, 0.2, 0.2, 0.15, 0.1, 0.1, 0.05]))      # skewed: a few long files
        rate = sr if i % 5 else (44100 if which != "acoustic" else 48000)                                              # every fifth file needs resampling
        x = S.speech_like_waveform(1, int(secs * rate) + 17 * i, rate, seed=7000 + i)[0]
        sub = os.path.join(src_dir, f"spk{i % 7}")
        os.makedirs(sub, exist_ok=True)
        wavfile.write(os.path.join(sub, f"utt{i:03d}.wav"), rate, np.round(x * 30000).astype(np.int16))
dist.barrier()
# the PRODUCT path (round 6): AudioToken.load_encoder distributes the model — acoustic: a checkpoint FILE that only rank 0 reads (the other ranks get a path that
# does not exist), one flat broadcast; semantic_s: only rank 0 holds weights, the others rebuild from its finalized packed model — and runs the start-up probe
if which == "acoustic":
    ckpt = os.path.join(work, "encodec_24khz.th")
    if rank == 0:
        torch.save({k: torch.from_numpy(v) for k, v in W.synth_encodec_weights(seed=0, with_decoder=False).items()}, ckpt)
    dist.barrier()
    tok = AudioToken(Tokenizers.acoustic, device="cuda:0", weights=ckpt if rank == 0 else ckpt + ".only-rank-0-reads-it", num_codebooks=8)
else:
    tok = AudioToken(getattr(Tokenizers, which), device="cuda:0", weights=W.synth_hubert_weights(11, 0, True) if rank == 0 else None)
tok.load_encoder()
pr = tok.rank_probe
assert pr is not None and pr["ranks"] == world
dist.barrier()
t0, c0 = time.perf_counter(), time.process_time()
tok.encode_batch_files(batch_size=16, outdir=out_n, chunk_size=10, audio_dir=src_dir, num_workers=workers)
torch.cuda.synchronize()
wall, cpu = time.perf_counter() - t0, time.process_time() - c0
rows = gather_scalars([wall, cpu, float(tok.run_timings["rows"]), float(tok.run_timings["batches"]), float(len(tok.skipped_files))], dev, dist)
dist.barrier()
if rank == 0:
    t0 = time.perf_counter()
    tok.encode_batch_files(batch_size=16, outdir=out_1, chunk_size=10, audio_dir=src_dir, num_workers=workers, shard_across_ranks=False)
    torch.cuda.synchronize()
    t1 = time.perf_counter() - t0
    rel = lambda root: sorted(os.path.relpath(os.path.join(d, f), root) for d, _, fs in os.walk(root) for f in fs)
    a, b = rel(out_n), rel(out_1)
    same = a == b and all(filecmp.cmp(os.path.join(out_n, f), os.path.join(out_1, f), shallow=False) for f in a)
    walls, cpus = [r[0] for r in rows], [r[1] for r in rows]
    print(f"[n{world}-shared-dir] {which}: {n_files} files (2-47 s, every fifth resampled) in ONE directory tree, {world} ranks on one device, {workers} decode threads per rank")
    print(f"[n{world}-shared-dir] start-up probe: all {pr['ranks']} ranks encode it alike (checksum {pr['token_checksum']})")
    print(f"[n{world}-shared-dir] token files: {len(a)} written by the {world} ranks, {len(b)} by the single-process run, byte-identical: {same}")
    print(f"[n{world}-shared-dir] rows per rank {[int(r[2]) for r in rows]} (max/min {max(r[2] for r in rows) / max(1.0, min(r[2] for r in rows)):.2f}), batches {[int(r[3]) for r in rows]}, skipped {[int(r[4]) for r in rows]}")
    print(f"[n{world}-shared-dir] wall s per rank {[round(x, 2) for x in walls]} (max/min {max(walls) / min(walls):.2f}); host CPU s per rank {[round(x, 2) for x in cpus]} (sum {sum(cpus):.1f}); single process {t1:.2f} s")
    assert same and len(a) == n_files, "world-N token files differ from the single-process run"
dist.barrier()
dist.destroy_process_group()
