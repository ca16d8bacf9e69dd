"""CPU: batch-harness semantics (product host code and the numpy oracle) against golden I/O captured from the
reference's datasets.py / utils.py (tests/golden/harness_a.npz)."""
import os

import numpy as np
import pytest
import torch

from audiotoken_amd import harness as H
from audiotoken_amd import weights as W
from audiotoken_amd.configs import AudioConfig
from oracle import harness_ref as R

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "harness_a.npz"))


@pytest.mark.parametrize("tag", ["a", "b", "c", "d", "e"])
def test_iter_chunk_matches_reference(tag):
    length, sr, chunk, rate = (int(v) for v in G[f"cfg_{tag}"])
    wave = torch.from_numpy(W.synth_waveform(1, length, sr, seed=3))
    rows, sums = [], []
    for seg, mask, cfg in H.iter_chunk(wave, f"x/y/clip_{tag}.v2.wav", sample_rate=sr, chunk_size=chunk, model_token_rate=rate):
        assert seg.shape == mask.shape == (chunk * sr,)
        assert float(seg[int(mask.sum()):].abs().sum()) == 0.0   # zero padding (pad_token = 0)
        rows.append([cfg.start_idx, cfg.end_idx, int(mask.sum().item()), cfg.length_tokens, seg.shape[0]])
        sums.append(float(seg.double().sum().item()))
    assert np.array_equal(np.array(rows, dtype=np.int64).reshape(-1, 5), G[f"seg_{tag}"])
    assert np.allclose(sums, G[f"sum_{tag}"], rtol=0, atol=1e-9)
    # numpy oracle agrees with the reference too
    o = R.segments(length, sr, chunk, rate)
    assert [list(r[:4]) for r in rows] == [[a, b, n, lt] for a, b, n, lt in o]
    arrs = R.segment_arrays(wave[0].numpy(), sr, chunk)
    assert len(arrs) == len(rows)


def test_save_audio_tokens_trim_and_append(tmp_path):
    ptr = AudioConfig(file_name="x/y/clip.v2.wav", length_seconds=1.0, model_token_rate=50)
    t1 = torch.arange(2 * 60, dtype=torch.int16).reshape(2, 60)
    H.save_audio_tokens(t1, ptr, str(tmp_path))
    assert sorted(os.listdir(tmp_path)) == list(G["save_files"])      # stem cut at the FIRST dot
    assert np.array_equal(np.load(tmp_path / "clip.npy"), G["save_first"])
    H.save_audio_tokens(t1 + 1000, ptr, str(tmp_path))
    got = np.load(tmp_path / "clip.npy")
    assert got.dtype == np.int16 and np.array_equal(got, G["save_second"])
    # oracle
    d2 = tmp_path / "o"
    d2.mkdir()
    R.save_tokens(t1.numpy(), ptr.file_name, ptr.length_tokens, str(d2))
    R.save_tokens((t1 + 1000).numpy(), ptr.file_name, ptr.length_tokens, str(d2))
    assert np.array_equal(np.load(d2 / "clip.npy"), G["save_second"])


def test_save_rel_audio_tokens_keeps_tree(tmp_path):
    src = tmp_path / "in" / "a" / "b"
    src.mkdir(parents=True)
    ptr = AudioConfig(file_name=str(src / "clip.v2.wav"), length_seconds=0.5, model_token_rate=50)
    H.save_rel_audio_tokens(torch.zeros(1, 40, dtype=torch.int16), ptr, str(tmp_path / "out"), str(tmp_path / "in"))
    out = np.load(tmp_path / "out" / "a" / "b" / "clip.v2.npy")       # only the LAST extension is stripped here
    assert out.shape == (1, 25)


def test_save_errors_are_swallowed(tmp_path):
    ptr = AudioConfig(file_name="f.wav")                               # length_tokens raises ValueError inside
    H.save_audio_tokens(torch.zeros(1, 4, dtype=torch.int16), ptr, str(tmp_path))
    assert os.listdir(tmp_path) == []


def test_sanitize_path(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    p = H.sanitize_path("rel/dir")
    assert os.path.isabs(p) and os.path.isdir(p)
    assert H.sanitize_path("~").startswith("/")


def test_length_tokens():
    assert AudioConfig("f", length_seconds=69100 / 16000, model_token_rate=50).length_tokens == 216
    with pytest.raises(ValueError):
        AudioConfig("f").length_tokens
