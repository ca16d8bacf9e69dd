"""Host-side prefetch (audiotoken_amd/prefetch.py): order, error propagation, and that encode_batch_files writes the same
token files with decode-ahead workers as it does inline (the reference's DataLoader-worker semantics, datasets.py:107-139)."""
import os
import threading
import time
import wave

import numpy as np
import pytest
import torch

from audiotoken_amd import weights as W
from audiotoken_amd.prefetch import background, ordered_map


def test_ordered_map_keeps_order_and_runs_ahead():
    seen = []
    lock = threading.Lock()

    def fn(i):
        time.sleep(0.02 * (5 - i % 5))     # later items finish first
        with lock:
            seen.append(i)
        return i * i

    out = list(ordered_map(fn, range(12), 4))
    assert out == [i * i for i in range(12)]
    assert seen != sorted(seen)              # it really ran out of order
    assert list(ordered_map(fn, range(3), 0)) == [0, 1, 4]


def test_ordered_map_raises_at_the_failing_position():
    def fn(i):
        if i == 3:
            raise ValueError("boom")
        return i

    got = []
    with pytest.raises(ValueError, match="boom"):
        for v in ordered_map(fn, range(8), 3):
            got.append(v)
    assert got == [0, 1, 2]


def test_background_generator_order_and_errors():
    assert list(background(lambda: iter(range(50)), depth=3)) == list(range(50))

    def bad():
        yield 1
        raise RuntimeError("stream broke")

    it = background(bad)
    assert next(it) == 1
    with pytest.raises(RuntimeError, match="stream broke"):
        next(it)


def _write_wav(path, x, sr):
    with wave.open(str(path), "wb") as f:
        f.setnchannels(1); f.setsampwidth(2); f.setframerate(sr)
        f.writeframes((np.clip(x, -1, 1) * 32767).astype("<i2").tobytes())


def test_batch_files_same_output_with_workers(tmp_path):
    from audiotoken_amd import AudioToken, Tokenizers
    sr = 24000
    names = []
    for i in range(7):
        p = tmp_path / f"c{i}.wav"
        _write_wav(p, W.synth_waveform(1, sr * (1 + i % 3) + 700 * i, sr if i % 2 == 0 else 16000, seed=40 + i)[0], sr if i % 2 == 0 else 16000)
        names.append(p)

    class Fake(torch.nn.Module):                # deterministic stand-in encoder: tokens depend on the batch contents
        def forward(self, x, m):
            s = (x.abs().sum(1) * 1000).to(torch.int64) % 997
            t = x.shape[1] // 320
            return (s.view(-1, 1, 1) + torch.arange(2 * t).view(1, 2, t)).to(torch.int16)

    outs = []
    for workers in (0, 4):
        tok = AudioToken(Tokenizers.acoustic, device="cpu", num_codebooks=2)
        tok.encoder = Fake()
        tok.load_encoder = lambda: None
        out = tmp_path / f"o{workers}"
        tok.encode_batch_files(batch_size=3, outdir=out, chunk_size=1, num_workers=workers, audio_files=names)
        outs.append({n: np.load(out / n) for n in sorted(os.listdir(out))})
    assert outs[0].keys() == outs[1].keys() and len(outs[0]) == 7
    for k in outs[0]:
        assert np.array_equal(outs[0][k], outs[1][k]), k


def test_background_starts_eagerly_and_stops_when_abandoned():
    """The producer thread starts with the object (decode-ahead) and ends when the consumer goes away with the queue full."""
    import threading
    import time
    from audiotoken_amd.prefetch import background
    started = threading.Event()
    produced = []

    def gen():
        started.set()
        for i in range(1000):
            produced.append(i)
            yield i

    it = background(gen, depth=2)
    assert started.wait(2.0), "the producer must start before the first next()"
    assert next(it) == 0 and next(it) == 1
    t = it._thread
    it.close()                       # consumer stops early: queue full, producer blocked in put()
    t.join(2.0)
    assert not t.is_alive(), "an abandoned producer must not stay blocked"
    assert len(produced) < 1000


def test_background_forwards_exceptions_and_end():
    from audiotoken_amd.prefetch import background

    def gen():
        yield 1
        raise RuntimeError("boom")

    it = background(gen)
    assert next(it) == 1
    import pytest
    with pytest.raises(RuntimeError, match="boom"):
        next(it)
    assert list(background(lambda: iter([1, 2, 3]))) == [1, 2, 3]


def test_background_stops_when_garbage_collected_or_consumer_fails():
    """The producer thread must not keep the iterator alive (it references the queue, the event and the generator factory only): dropping the
    object without close() — what an exception in the consumer does — ends the thread and runs the generator's finally block (the tar / zip handle)."""
    import gc
    import threading
    import time
    from audiotoken_amd.prefetch import background
    closed = threading.Event()

    def endless():
        try:
            i = 0
            while True:
                yield i
                i += 1
        finally:
            closed.set()

    b = background(endless, depth=2)
    assert next(b) == 0
    th = b._thread
    del b
    gc.collect()
    th.join(3.0)
    assert not th.is_alive(), "the producer thread outlived its abandoned iterator"
    assert closed.wait(1.0), "the generator's finally block did not run"

    # explicit close(), and join()
    closed.clear()
    b = background(endless, depth=1)
    assert [next(b), next(b)] == [0, 1]
    b.close()
    assert b.join(3.0) and closed.wait(1.0)
    with pytest.raises(StopIteration):
        next(b)


def test_worker_processes_give_the_sequential_result(tmp_path):
    """encode_batch_files(worker_processes=True): plain files are decoded / resampled in SPAWNED worker processes (the reference's DataLoader workers,
    core.py:259-267), archives by a background thread — same token files, same skip list as the inline run; an undecodable file inside the process pool
    is reported, not fatal."""
    import tarfile
    import numpy as np
    import torch
    from scipy.io import wavfile
    from audiotoken_amd import AudioToken, Tokenizers
    from audiotoken_amd import weights as W
    G = os.path.join(os.path.dirname(__file__), "golden")
    for i, (sr, secs) in enumerate([(24000, 2.3), (44100, 1.7), (16000, 3.1), (24000, 0.6)]):
        wavfile.write(str(tmp_path / f"w{i}.wav"), sr, np.round(W.synth_waveform(1, int(sr * secs), sr, seed=40 + i)[0] * 20000).astype(np.int16))
    (tmp_path / "bad.wav").write_bytes(b"RIFF\x00\x00")
    with tarfile.open(tmp_path / "t.tar", "w") as tar:
        tar.add(os.path.join(G, "flac_a.flac"), arcname="m/flac_a.flac")
    files = [tmp_path / "w0.wav", tmp_path / "bad.wav", tmp_path / "w1.wav", tmp_path / "t.tar", tmp_path / "w2.wav", os.path.join(G, "flac_c.flac"), tmp_path / "w3.wav"]

    class Fake(torch.nn.Module):
        def forward(self, x, m):
            return ((x * m).reshape(x.shape[0], 75, -1).abs().sum(-1) * 997).to(torch.int64).remainder(1024).to(torch.int16)[:, None, :].repeat(1, 2, 1)

    outs = {}
    for name, kw in (("inline", dict(num_workers=0)), ("procs", dict(num_workers=3, worker_processes=True)), ("threads", dict(num_workers=3))):
        tok = AudioToken(Tokenizers.acoustic, device="cpu", num_codebooks=2)
        tok.encoder = Fake()
        tok.load_encoder = lambda: None
        tok.encode_batch_files(batch_size=4, outdir=tmp_path / name, chunk_size=1, audio_files=files, **kw)
        outs[name] = ({n: np.load(tmp_path / name / n) for n in sorted(os.listdir(tmp_path / name))}, [os.path.basename(p) for p, _ in tok.skipped_files])
    ref, ref_skipped = outs["inline"]
    assert sorted(ref) == ["flac_a.npy", "flac_c.npy", "w0.npy", "w1.npy", "w2.npy", "w3.npy"] and ref_skipped == ["bad.wav"]
    for name in ("procs", "threads"):
        got, skipped = outs[name]
        assert skipped == ref_skipped and sorted(got) == sorted(ref)
        assert all(np.array_equal(got[n], ref[n]) for n in ref), name
