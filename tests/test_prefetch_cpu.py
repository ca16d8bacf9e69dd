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
