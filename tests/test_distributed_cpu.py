"""CPU, world_size 2, gloo: the N>1 path — weight broadcast and clip/file sharding (no data-path collective)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from audiotoken_amd.distributed import broadcast_packed, broadcast_weights, gather_scalars, shard_indices


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from audiotoken_amd import weights as W
        w = None
        if rank == 0:
            w = {k: v for k, v in W.synth_encodec_weights(seed=0, with_decoder=False, n_codebooks=2).items()}
        got = broadcast_weights(w, torch.device("cpu"), dist)
        ref = W.synth_encodec_weights(seed=0, with_decoder=False, n_codebooks=2)
        assert list(got) == list(ref)
        assert all(np.array_equal(got[k], ref[k]) and got[k].shape == ref[k].shape for k in ref)
        mine = shard_indices(11, rank, world)
        allv = gather_scalars([float(len(mine)), float(sum(mine))], torch.device("cpu"), dist)
        assert sum(v[0] for v in allv) == 11 and sum(v[1] for v in allv) == sum(range(11))
        # the finalized model as (host record, one blob): every rank ends with rank 0's bytes (blob on the device given; CPU tensors here)
        meta0, blob0 = bytes(range(200)) * 3, torch.arange(100003, dtype=torch.int64).to(torch.uint8)
        meta, blob = broadcast_packed((meta0, blob0) if rank == 0 else None, torch.device("cpu"), dist)
        assert meta == meta0 and blob.dtype == torch.uint8 and torch.equal(blob, blob0)
        np.save(os.path.join(tmp, f"ok{rank}.npy"), np.array(mine))
    finally:
        dist.destroy_process_group()


def test_broadcast_and_sharding_world2(tmp_path):
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = np.load(tmp_path / "ok0.npy"), np.load(tmp_path / "ok1.npy")
    assert sorted(list(a) + list(b)) == list(range(11)) and len(a) - len(b) in (0, 1)


@pytest.mark.parametrize("n,world", [(0, 4), (3, 8), (512, 8), (10, 3)])
def test_shard_indices_partition(n, world):
    parts = [shard_indices(n, r, world) for r in range(world)]
    flat = [i for p in parts for i in p]
    assert flat == list(range(n))                      # contiguous blocks in rank order
    assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


# ---- encode_batch_files under torch.distributed (world 2, gloo) -------------------------------------------------------------------
class _HashEncoder(torch.nn.Module):
    """Stand-in encoder (no device): the token at frame t is a function of that frame's samples and of NOTHING else (not the batch row, not
    the rank), so two runs produce identical files exactly when every segment reaches the encoder with the same content and in the same order
    per file."""

    def forward(self, x, m):
        B, N = x.shape
        T = -(-N // 320)
        pad = torch.nn.functional.pad(x * m, (0, T * 320 - N)).reshape(B, T, 320)
        a = (pad.abs().sum(-1) * 1000.0).round().to(torch.int64) % 1024
        b = (pad[..., ::7].sum(-1).abs() * 1000.0).round().to(torch.int64) % 1024
        return torch.stack([a, b], 1).to(torch.int16)


def _write_inputs(d, sr=24000):
    """7 WAV files of 1.2 .. 4.6 s (multi-chunk at chunk_size 1, ragged tails) + a tar of two members."""
    import io
    import tarfile
    import wave as wavmod
    from audiotoken_amd import weights as W
    names = []
    for i in range(7):
        x = W.synth_waveform(1, int(sr * (1.2 + 0.55 * i)) + 37 * i, sr, seed=50 + i)[0]
        p = os.path.join(d, f"clip{i}.v{i}.wav")        # dots in the stem: save_audio_tokens keeps only the first part (reference quirk B.6)
        with wavmod.open(p, "wb") as f:
            f.setnchannels(1); f.setsampwidth(2); f.setframerate(sr)
            f.writeframes((np.clip(x, -1, 1) * 32767).astype(np.int16).tobytes())
        names.append(p)
    tp = os.path.join(d, "bundle.tar")
    with tarfile.open(tp, "w") as tar:
        for j in range(2):
            x = W.synth_waveform(1, sr * 2 + 999 * j, sr, seed=70 + j)[0]
            buf = io.BytesIO()
            with wavmod.open(buf, "wb") as f:
                f.setnchannels(1); f.setsampwidth(2); f.setframerate(sr)
                f.writeframes((np.clip(x, -1, 1) * 32767).astype(np.int16).tobytes())
            info = tarfile.TarInfo(f"member{j}.wav"); info.size = buf.tell(); buf.seek(0)
            tar.addfile(info, buf)
    names.append(tp)
    return names


def _encode_files(names, outdir, workers):
    from audiotoken_amd import AudioToken, Tokenizers
    tok = AudioToken(Tokenizers.acoustic, device="cpu", num_codebooks=2)
    tok.encoder = _HashEncoder()
    tok.load_encoder = lambda: None
    tok.encode_batch_files(batch_size=3, outdir=outdir, chunk_size=1, num_workers=workers, audio_files=names)
    return tok


def _batch_worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        names = sorted(os.path.join(tmp, "in", n) for n in os.listdir(os.path.join(tmp, "in")))
        _encode_files(names, os.path.join(tmp, f"out_rank{rank}"), workers=2)
    finally:
        dist.destroy_process_group()


def test_encode_batch_files_world2_equals_world1(tmp_path):
    """Reference core.py:198-289 under the clip/file sharding of SURVEY.md §8(e): with torch.distributed initialised (world 2) every rank takes whole
    FILES (greedy LPT on their sizes, distributed.shard_by_size) — all chunks of a file on one rank, so the reference's per-file append order
    (utils.py:214-217) survives — the two ranks' file sets are disjoint and complete, and every token file is byte-identical to the single-process run."""
    os.makedirs(tmp_path / "in")
    names = sorted(_write_inputs(str(tmp_path / "in")))
    _encode_files(names, str(tmp_path / "out_world1"), workers=0)
    ref = {n: np.load(tmp_path / "out_world1" / n) for n in sorted(os.listdir(tmp_path / "out_world1"))}
    assert len(ref) == 9 and all(v.dtype == np.int16 and v.shape[0] == 2 for v in ref.values())       # 7 files + 2 tar members
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_batch_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got0 = {n: np.load(tmp_path / "out_rank0" / n) for n in os.listdir(tmp_path / "out_rank0")}
    got1 = {n: np.load(tmp_path / "out_rank1" / n) for n in os.listdir(tmp_path / "out_rank1")}
    assert not (set(got0) & set(got1)), "a file was encoded by both ranks"
    assert set(got0) | set(got1) == set(ref), "the ranks' file sets do not cover the input"
    for n, v in {**got0, **got1}.items():
        assert v.shape == ref[n].shape and np.array_equal(v, ref[n]), f"{n}: tokens differ from the single-process run"
    # the split is the one shard_by_size gives for the inputs' sizes, and it balances the BYTES (the proxy for audio seconds), not the file count
    from audiotoken_amd.distributed import shard_by_size
    sizes = [os.path.getsize(n) for n in names]
    own = [[os.path.basename(names[i]) for i in shard_by_size(sizes, r, 2)] for r in range(2)]
    stem = lambda f: {"bundle.tar": ["member0.npy", "member1.npy"]}.get(f, [f.split(".")[0] + ".npy"])
    assert set(got0) == {s for f in own[0] for s in stem(f)} and set(got1) == {s for f in own[1] for s in stem(f)}
    work = [sum(sizes[i] for i in shard_by_size(sizes, r, 2)) for r in range(2)]
    assert max(work) / min(work) <= 1.15, work


def _skew_worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        names = sorted(os.path.join(tmp, "in", n) for n in os.listdir(os.path.join(tmp, "in")))
        _encode_files(names, os.path.join(tmp, f"w{world}_rank{rank}"), workers=0)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_skewed_directory_is_balanced_by_duration(tmp_path, world):
    """A directory whose first file (in sorted order) is 10 x as long as the other 29: contiguous blocks by COUNT would give rank 0 the long file plus its
    share of the short ones; the duration-aware split keeps max / min work per rank within 1.15, every file is encoded exactly once and every token file is
    byte-identical to the single-process run (VERDICT round 3, weak #14 / next #6a)."""
    import wave as wavmod
    from audiotoken_amd import weights as W
    from audiotoken_amd.distributed import shard_by_size, shard_indices
    sr = 24000
    os.makedirs(tmp_path / "in")
    for i in range(30):
        secs = 10.0 if i == 0 else 1.0 + 0.01 * i
        x = W.synth_waveform(1, int(sr * secs), sr, seed=900 + i)[0]
        with wavmod.open(str(tmp_path / "in" / f"a{i:02d}.wav"), "wb") as f:
            f.setnchannels(1); f.setsampwidth(2); f.setframerate(sr)
            f.writeframes((np.clip(x, -1, 1) * 32767).astype(np.int16).tobytes())
    names = sorted(str(tmp_path / "in" / n) for n in os.listdir(tmp_path / "in"))
    sizes = [os.path.getsize(n) for n in names]
    work = [sum(sizes[i] for i in shard_by_size(sizes, r, world)) for r in range(world)]
    by_count = [sum(sizes[i] for i in shard_indices(len(sizes), r, world)) for r in range(world)]
    print(f"world {world}: bytes per rank by duration {work} (max/min {max(work) / min(work):.3f}); by count {by_count} (max/min {max(by_count) / min(by_count):.2f})")
    assert max(work) / min(work) <= 1.15 < max(by_count) / min(by_count)
    assert sorted(i for r in range(world) for i in shard_by_size(sizes, r, world)) == list(range(30))
    _encode_files(names, str(tmp_path / "w1"), workers=0)
    ref = {n: np.load(tmp_path / "w1" / n) for n in os.listdir(tmp_path / "w1")}
    port = 33500 + (os.getpid() % 2000) + world
    mp.spawn(_skew_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    seen = {}
    for r in range(world):
        for n in os.listdir(tmp_path / f"w{world}_rank{r}"):
            assert n not in seen, f"{n} was encoded by two ranks"
            seen[n] = np.load(tmp_path / f"w{world}_rank{r}" / n)
    assert set(seen) == set(ref)
    assert all(np.array_equal(seen[n], ref[n]) for n in ref)


# ---- round 5: start-up self-check — every rank must encode a probe like rank 0 ---------------------------------------------------------------------------
def _probe_worker(rank, world, port, corrupt_rank, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from audiotoken_amd.distributed import ranks_agree_on_probe
        enc = _HashEncoder()
        probe = torch.linspace(-1, 1, 2 * 6400).reshape(2, 6400)
        bias = 0.001 if rank == corrupt_rank else 0.0        # a "model" that differs on one rank (what a wrong import_packed would be)
        try:
            res = ranks_agree_on_probe(lambda x: enc(x + bias, torch.ones_like(x)), probe, torch.device("cpu"), dist, "probe")
            q.put((rank, "ok", res["ranks"]))
        except RuntimeError as e:
            q.put((rank, "raised", str(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("corrupt_rank", [-1, 2])
def test_ranks_agree_on_probe_world4(corrupt_rank):
    """bench.py's start-up self-check (audiotoken_amd/distributed.ranks_agree_on_probe): with identical models every rank returns the shared checksums; with ONE
    rank's model perturbed EVERY rank raises (nobody enters a timed region or writes a token file), naming the rank."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000) + (7 if corrupt_rank >= 0 else 0)
    mp.spawn(_probe_worker, args=(4, port, corrupt_rank, q), nprocs=4, join=True)
    got = sorted(q.get(timeout=30) for _ in range(4))
    if corrupt_rank < 0:
        assert all(s == "ok" and n == 4 for _, s, n in got), got
    else:
        assert all(s == "raised" and f"ranks [{corrupt_rank}]" in msg for _, s, msg in got), got


# ---- round 6: the PRODUCT distributes the model — AudioToken.load_encoder under torch.distributed ----------------------------------------------------
class _PackedEncoder(torch.nn.Module):
    """Stand-in for Wav2VecBertEncoder / HubertEncoder (no device, no HIP library): same constructor keywords, `export_packed()` / `packed=`. `weights` is a
    path to a .npy "checkpoint"; the tokens depend on every weight. `corrupt_import` makes the packed import wrong on the ranks listed in the environment."""
    loads = 0   # checkpoint reads in this process

    def __init__(self, config=None, device="cpu", quantize=True, weights=None, packed=None):
        super().__init__()
        if packed is not None:
            meta, blob = packed
            assert meta == b"stand-in v1" and blob.dtype == torch.uint8
            self.w = blob.clone().view(torch.float32)
            if str(dist.get_rank()) in os.environ.get("AT_TEST_CORRUPT_IMPORT", "").split(","):
                self.w[3] += 0.5
        else:
            type(self).loads += 1
            self.w = torch.from_numpy(np.load(weights if weights is not None else config.weights))   # FileNotFoundError on a rank without the file

    def export_packed(self):
        return b"stand-in v1", self.w.clone().view(torch.uint8)

    def forward(self, x, m):
        B, N = x.shape
        T = N // 320
        f = (x * m)[:, :T * 320].reshape(B, T, 320)
        k = self.w[:320]
        return ((f * k).sum(-1).abs() * 4096.0).round().to(torch.int64).remainder(2048).to(torch.int16).unsqueeze(1)


def _product_worker(rank, world, port, tmp, case, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if case == "corrupt":
        os.environ["AT_TEST_CORRUPT_IMPORT"] = "1"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import audiotoken_amd.encoder as E
        from audiotoken_amd import AudioToken, Tokenizers
        E.Wav2VecBertEncoder = _PackedEncoder
        ckpt = os.path.join(tmp, "ckpt.npy")
        # only rank 0 can read the checkpoint: the other ranks are given a path that does not exist (case "missing": rank 0 cannot either)
        path = ckpt if (rank == 0 and case != "missing") or case == "per_rank" else os.path.join(tmp, "not-here.npy")
        kw = {"broadcast_weights": False} if case == "per_rank" else {}
        tok = AudioToken(Tokenizers.semantic_m, device="cpu", weights=path, **kw)
        try:
            tok.load_encoder()
            q.put((rank, "ok", _PackedEncoder.loads, float(tok.encoder.w.double().sum()), tok.rank_probe))
        except RuntimeError as e:
            q.put((rank, "raised", _PackedEncoder.loads, str(e), None))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", ["shared", "per_rank", "corrupt", "missing"])
def test_audiotoken_load_encoder_distributes_the_model(tmp_path, case):
    """AudioToken.load_encoder with world 2 (gloo, stand-in encoder): rank 0 reads the checkpoint ONCE and exports its finalized model, rank 1 rebuilds from the
    broadcast blob without touching the checkpoint, both run the start-up probe and hold its checksums; `broadcast_weights=False` keeps per-rank loading;
    a rank whose import went wrong makes EVERY rank raise before the first batch; a checkpoint rank 0 cannot read raises on every rank (nobody waits)."""
    rng = np.random.default_rng(5)
    np.save(tmp_path / "ckpt.npy", rng.standard_normal(4096).astype(np.float32))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + (os.getpid() % 2000) + 11 * ["shared", "per_rank", "corrupt", "missing"].index(case)
    mp.spawn(_product_worker, args=(2, port, str(tmp_path), case, q), nprocs=2, join=True)
    got = sorted(q.get(timeout=30) for _ in range(2))
    want = float(np.load(tmp_path / "ckpt.npy").astype(np.float64).sum())
    if case == "shared":
        assert [g[1] for g in got] == ["ok", "ok"], got
        assert [g[2] for g in got] == [1, 0]                                   # checkpoint reads: rank 0 once, rank 1 never
        assert all(abs(g[3] - want) < 1e-6 for g in got)
        assert got[0][4] == got[1][4] and got[0][4]["ranks"] == 2
    elif case == "per_rank":
        assert [g[1] for g in got] == ["ok", "ok"] and [g[2] for g in got] == [1, 1] and got[0][4]["ranks"] == 2, got
    elif case == "corrupt":
        assert all(g[1] == "raised" and "ranks [1]" in g[3] for g in got), got
    else:
        assert all(g[1] == "raised" and "rank 0 could not build the model" in g[3] and "FileNotFoundError" in g[3] for g in got), got


class _DictEncoder(torch.nn.Module):
    """Stand-in for AcousticEncoder: keeps the weight dict it was constructed with."""

    def __init__(self, config=None, device="cpu", weights=None):
        super().__init__()
        assert isinstance(weights, dict), type(weights)
        self.weights = weights
        self.k = torch.from_numpy(np.concatenate([np.asarray(v, dtype=np.float32).reshape(-1) for v in weights.values()])[:320].copy())

    def forward(self, x, m):
        B, N = x.shape
        T = N // 320
        return ((x[:, :T * 320].reshape(B, T, 320) * self.k).sum(-1).abs() * 4096.0).round().to(torch.int64).remainder(1024).to(torch.int16).unsqueeze(1)


def _acoustic_worker(rank, world, port, tmp, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import audiotoken_amd.encoder as E
        from audiotoken_amd import AudioToken, Tokenizers
        from audiotoken_amd import weights as W
        E.AcousticEncoder = _DictEncoder
        ckpt = os.path.join(tmp, "encodec.th")
        if rank == 0:
            torch.save({k: torch.from_numpy(v) for k, v in W.synth_encodec_weights(seed=3, with_decoder=False, n_codebooks=2).items()}, ckpt)
        dist.barrier()
        tok = AudioToken(Tokenizers.acoustic, device="cpu", num_codebooks=2, weights=ckpt if rank == 0 else os.path.join(tmp, "absent.th"))
        tok.load_encoder()
        w = tok.encoder.weights
        q.put((rank, len(w), float(sum(np.asarray(v, dtype=np.float64).sum() for v in w.values())), tok.rank_probe["token_checksum"]))
    finally:
        dist.destroy_process_group()


def test_audiotoken_acoustic_weights_come_from_rank0(tmp_path):
    """The EnCodec checkpoint is read on rank 0 only (rank 1's path does not exist) and reaches rank 1 as one flat broadcast; both ranks pass the probe."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    mp.spawn(_acoustic_worker, args=(2, 36900 + (os.getpid() % 2000), str(tmp_path), q), nprocs=2, join=True)
    a, b = sorted(q.get(timeout=30) for _ in range(2))
    assert a[1:] == b[1:] and a[1] > 50, (a, b)


def _mismatch_worker(rank, world, port, tmp, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        names = sorted(os.path.join(tmp, "in", n) for n in os.listdir(os.path.join(tmp, "in")))
        if rank == 1:
            names = names[:-1] + [names[-1] + ".other"]          # the same COUNT, one different name: the length check alone would not see it
        try:
            _encode_files(names, os.path.join(tmp, f"out_rank{rank}"), workers=0)
            q.put((rank, "ok", ""))
        except AssertionError as e:
            q.put((rank, "raised", str(e)))
    finally:
        dist.destroy_process_group()


def test_ranks_with_different_file_lists_all_stop(tmp_path):
    """the size broadcast used to compare list LENGTHS only. Now a digest of the list travels with the sizes and every rank learns the verdict of all
    ranks: a rank holding a different list (same length) makes EVERY rank raise before a single token file is written."""
    os.makedirs(tmp_path / "in")
    _write_inputs(str(tmp_path / "in"))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    mp.spawn(_mismatch_worker, args=(2, 37900 + (os.getpid() % 2000), str(tmp_path), q), nprocs=2, join=True)
    got = sorted(q.get(timeout=30) for _ in range(2))
    assert all(s == "raised" and "ranks [1] see a different file list" in msg for _, s, msg in got), got
    assert not os.path.exists(tmp_path / "out_rank0") or not os.listdir(tmp_path / "out_rank0")
