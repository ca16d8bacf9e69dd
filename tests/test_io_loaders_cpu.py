"""CPU: input side (WAV / tar / zip streaming, chunk seams) and the real-checkpoint loaders (file formats the
reference loads: EnCodec .th state dict, HF safetensors + VQ .pkl, HF HuBERT + joblib k-means)."""
import io
import os
import tarfile
import zipfile

import numpy as np
import pytest
import torch
from scipy.io import wavfile

from audiotoken_amd import audio_io as A
from audiotoken_amd import weights as W


def _write_wav(path, x, sr):
    wavfile.write(str(path), sr, (np.clip(x, -1, 1) * 32767).astype(np.int16))


def test_read_audio_and_chunks(tmp_path):
    sr = 16000
    x = W.synth_waveform(1, sr * 3 + 100, sr, seed=2)[0]
    _write_wav(tmp_path / "a.wav", x, sr)
    y = A.read_audio(tmp_path / "a.wav", sr)
    assert tuple(y.shape) == (1, sr * 3 + 100) and y.dtype == torch.float32
    assert np.abs(y[0].numpy() - x).max() < 1e-4
    chunks = list(A.process_audio_chunks(tmp_path / "a.wav", sr, 1))
    assert [c.shape[1] for c, _ in chunks] == [sr, sr, sr, 100]
    assert all(name.endswith("a.wav") for _, name in chunks)
    up = A.read_audio(tmp_path / "a.wav", 24000)          # resampled on load
    assert up.shape[1] == int(np.ceil((sr * 3 + 100) * 1.5))
    stereo = np.stack([x, -x], 1)
    wavfile.write(str(tmp_path / "s.wav"), sr, stereo.astype(np.float32))
    assert float(A.read_audio(tmp_path / "s.wav", sr).abs().max()) < 1e-6       # mono mix of (x, -x)
    with pytest.raises(A.AudioDecodeError):                # lossy codecs: no decoder in this build
        A.load(tmp_path / "a.mp3")


def test_tar_and_zip_members(tmp_path):
    sr = 24000
    xs = {f"d/{i}.wav": W.synth_waveform(1, sr + i * 10, sr, seed=i)[0] for i in range(2)}
    bufs = {}
    for name, x in xs.items():
        b = io.BytesIO()
        wavfile.write(b, sr, x)
        bufs[name] = b.getvalue()
    with tarfile.open(tmp_path / "t.tar", "w") as tf:
        for name, data in bufs.items():
            ti = tarfile.TarInfo(name)
            ti.size = len(data)
            tf.addfile(ti, io.BytesIO(data))
    with zipfile.ZipFile(tmp_path / "z.zip", "w") as zf:
        for name, data in bufs.items():
            zf.writestr(name, data)
    for it in (A.iterate_tar(tmp_path / "t.tar", sr, 30), A.iterate_zip(tmp_path / "z.zip", sr, 30)):
        got = list(it)
        assert [n for _, n in got] == list(xs)
        for (c, n) in got:
            assert np.array_equal(c[0].numpy(), xs[n])


def test_encodec_checkpoint_roundtrip(tmp_path):
    from audiotoken_amd.encoder import fold_encodec_weights, load_encodec_checkpoint
    w = W.synth_encodec_weights(seed=1, with_decoder=True, n_codebooks=4)
    sd = {k: torch.from_numpy(v) for k, v in w.items()}
    sd["quantizer.vq.layers.0._codebook.inited"] = torch.tensor([1.0])      # buffers present in the real .th file
    sd["quantizer.vq.layers.0._codebook.cluster_size"] = torch.zeros(1024)
    sd["quantizer.vq.layers.0._codebook.embed_avg"] = torch.zeros(1024, 128)
    torch.save(sd, tmp_path / "encodec_24khz.th")
    got = load_encodec_checkpoint(str(tmp_path / "encodec_24khz.th"))
    f_ref, f_got = fold_encodec_weights(w), fold_encodec_weights(got)
    assert list(f_ref) == list(f_got)
    assert all(np.array_equal(f_ref[k], f_got[k]) for k in f_ref)
    assert "encoder.model.0.conv.conv.weight" in f_got and "quantizer.vq.layers.3._codebook.e2" in f_got
    assert not any(k.endswith(("inited", "cluster_size", "embed_avg", "weight_g", "weight_v")) for k in f_got)


def test_w2vbert_checkpoint_roundtrip(tmp_path):
    from safetensors.numpy import save_file
    from audiotoken_amd.encoder import load_w2vbert_checkpoint
    w = W.synth_w2vbert_weights(n_layers=1, seed=2, with_vq=True)
    save_file({k: v for k, v in w.items() if not k.startswith("vq.")}, str(tmp_path / "model.safetensors"))
    torch.save({"_codebook.embed": torch.from_numpy(w["vq._codebook.embed"]), "_codebook.cluster_size": torch.zeros(1, 2048)},
               tmp_path / "vq.pkl")
    got = load_w2vbert_checkpoint(str(tmp_path), str(tmp_path / "vq.pkl"))
    for k, v in w.items():
        assert np.array_equal(got[k], v), k


def test_hubert_checkpoint_roundtrip(tmp_path):
    import joblib
    from safetensors.numpy import save_file
    from sklearn.cluster import KMeans
    from audiotoken_amd.hubert import fold_hubert_weights, load_hubert_checkpoint
    w = W.synth_hubert_weights(n_layers=1, seed=3, with_kmeans=True)
    save_file({("hubert." + k): v for k, v in w.items() if not k.startswith("kmeans.")}, str(tmp_path / "model.safetensors"))
    km = KMeans(n_clusters=1000)
    km.cluster_centers_ = w["kmeans.cluster_centers_"].astype(np.float64)      # sklearn stores float64
    joblib.dump(km, tmp_path / "km.bin")
    got = load_hubert_checkpoint(str(tmp_path), str(tmp_path / "km.bin"))
    f_ref, f_got = fold_hubert_weights(w, 1), fold_hubert_weights(got, 1)
    assert sorted(f_ref) == sorted(f_got)
    assert all(np.array_equal(f_ref[k], f_got[k]) for k in f_ref)
    assert tuple(f_got["encoder.pos_conv_embed.conv.weight"].shape) == (768, 48, 128)


def test_batch_files_sharding_logic(tmp_path, monkeypatch):
    """encode_batch_files: segmentation / batching / per-file save order, with a stand-in encoder (no device)."""
    from audiotoken_amd import AudioToken, Tokenizers
    sr = 24000
    names = []
    for i in range(3):
        _write_wav(tmp_path / f"f{i}.wav", W.synth_waveform(1, sr * (i + 1) + 500 * i, sr, seed=i)[0], sr)
        names.append(tmp_path / f"f{i}.wav")
    tok = AudioToken(Tokenizers.acoustic, device="cpu", num_codebooks=2)
    calls = []

    class Fake(torch.nn.Module):
        def forward(self, x, m):
            calls.append((tuple(x.shape), m.sum(1).tolist()))
            return torch.arange(x.shape[0] * 2 * 75, dtype=torch.int16).reshape(x.shape[0], 2, 75)

    tok.encoder = Fake()
    monkeypatch.setattr(tok, "load_encoder", lambda: None)
    tok.encode_batch_files(batch_size=4, outdir=tmp_path / "o", chunk_size=1, audio_files=names)
    segs = [s for shp, ms in calls for s in ms]
    assert all(shp[1] == sr for shp, _ in calls) and max(shp[0] for shp, _ in calls) == 4
    assert segs == [sr, sr, sr, sr, sr, sr]            # f1's 500-sample and f2's 1000-sample tails are < 3200: skipped
    out = {n: np.load(tmp_path / "o" / n) for n in sorted(os.listdir(tmp_path / "o"))}
    assert {k: v.shape for k, v in out.items()} == {"f0.npy": (2, 75), "f1.npy": (2, 150), "f2.npy": (2, 225)}


def test_undecodable_files_are_skipped_visibly_and_bugs_propagate(tmp_path, monkeypatch):
    """A file this build cannot decode (a damaged FLAC; an mp3: no lossy decoder here; a stereo WAV; a damaged header) is skipped, but the caller can see it:
    ``AudioToken.skipped_files`` lists it and the run logs a summary. Any other exception (a bug in resampling / chunking) propagates, like
    the reference's dataset iterator (audiotoken/datasets.py __iter__) lets it."""
    from audiotoken_amd import AudioToken, Tokenizers
    from audiotoken_amd import audio_io as A
    sr = 24000
    _write_wav(tmp_path / "good.wav", W.synth_waveform(1, sr, sr, seed=1)[0], sr)
    (tmp_path / "music.flac").write_bytes(b"fLaC\x00\x00\x00\x22")
    (tmp_path / "broken.wav").write_bytes(b"RIFF\x00\x00")
    from scipy.io import wavfile
    wavfile.write(str(tmp_path / "stereo.wav"), sr, np.zeros((sr, 2), np.float32))
    tok = AudioToken(Tokenizers.acoustic, device="cpu", num_codebooks=2)

    class Fake(torch.nn.Module):
        def forward(self, x, m):
            return torch.zeros((x.shape[0], 2, 75), dtype=torch.int16)

    tok.encoder = Fake()
    monkeypatch.setattr(tok, "load_encoder", lambda: None)
    (tmp_path / "song.mp3").write_bytes(b"ID3\x03\x00\x00")
    files = [tmp_path / n for n in ("music.flac", "good.wav", "broken.wav", "stereo.wav", "notes.txt", "song.mp3")]
    tok.encode_batch_files(batch_size=2, outdir=tmp_path / "o", chunk_size=1, audio_files=files, num_workers=2)
    assert sorted(os.listdir(tmp_path / "o")) == ["good.npy"]
    skipped = {os.path.basename(p): why for p, why in tok.skipped_files}
    assert set(skipped) == {"music.flac", "broken.wav", "stereo.wav", "notes.txt", "song.mp3"}
    assert "flac" in skipped["music.flac"] and "mono" in skipped["stereo.wav"] and skipped["notes.txt"] == "unsupported extension"
    assert "lossy-codec decoder" in skipped["song.mp3"]
    # a second run starts from an empty list
    tok.encode_batch_files(batch_size=2, outdir=tmp_path / "o2", chunk_size=1, audio_files=[tmp_path / "good.wav"], num_workers=0)
    assert tok.skipped_files == []
    # a bug is not a decode error: it must surface
    _write_wav(tmp_path / "sr16.wav", W.synth_waveform(1, 16000, 16000, seed=2)[0], 16000)
    monkeypatch.setattr(A, "resample", lambda *a, **k: (_ for _ in ()).throw(ZeroDivisionError("bug in resample")))
    with pytest.raises(ZeroDivisionError):
        tok.encode_batch_files(batch_size=2, outdir=tmp_path / "o3", chunk_size=1, audio_files=[tmp_path / "sr16.wav"], num_workers=0)


def test_wav_probe_matches_the_general_reader(tmp_path):
    """`audio_io.wav_probe` (the header walk behind the device feeder's read-into-pinned-memory path) against scipy's reader: same dtype, rate and
    samples for the formats it accepts; None — i.e. the general reader decides — for everything else (stereo, 24-bit, truncated, odd chunks first)."""
    rng = np.random.default_rng(3)
    cases = {"s16": (rng.integers(-32768, 32767, 5001).astype(np.int16), 22050), "s32": (rng.integers(-2**31, 2**31 - 1, 777).astype(np.int32), 48000),
             "u8": (rng.integers(0, 255, 1234).astype(np.uint8), 8000), "f32": (rng.standard_normal(999).astype(np.float32), 16000)}
    for name, (data, sr) in cases.items():
        p = str(tmp_path / f"{name}.wav")
        wavfile.write(p, sr, data)
        hdr = A.wav_probe(p)
        assert hdr is not None, name
        dtype, rate, off, nbytes, scale, offset = hdr
        raw = A.decode_raw(p)
        assert rate == sr == raw.sample_rate and nbytes == data.nbytes and scale == raw.scale and offset == raw.offset
        with open(p, "rb") as f:
            f.seek(off)
            got = np.frombuffer(f.read(nbytes), dtype=dtype)
        assert got.dtype == raw.pcm.dtype and np.array_equal(got, raw.pcm[0]) and np.array_equal(got, data)
    # an extra chunk (LIST, odd length -> pad byte) in front of `data` is walked over
    p = str(tmp_path / "list.wav")
    wavfile.write(p, 16000, cases["s16"][0])
    b = open(p, "rb").read()
    i = b.index(b"data")
    extra = b"LIST" + (5).to_bytes(4, "little") + b"abcde" + b"\0"
    b2 = b[:i] + extra + b[i:]
    b2 = b2[:4] + (len(b2) - 8).to_bytes(4, "little") + b2[8:]
    open(p, "wb").write(b2)
    hdr = A.wav_probe(p)
    assert hdr is not None and hdr[2] == i + len(extra) + 8 and np.array_equal(A.decode_raw(p).pcm[0], cases["s16"][0])
    # not for the fast path
    stereo = str(tmp_path / "stereo.wav")
    wavfile.write(stereo, 16000, np.stack([cases["s16"][0], cases["s16"][0]], 1))
    assert A.wav_probe(stereo) is None
    trunc = str(tmp_path / "trunc.wav")
    open(trunc, "wb").write(b[:-100])
    assert A.wav_probe(trunc) is None
    s24 = str(tmp_path / "s24.wav")     # 24-bit PCM: 3-byte samples need unpacking
    hdr24 = b[:i]
    fi = hdr24.index(b"fmt ") + 8
    fmt = bytearray(hdr24[fi:fi + 16])
    fmt[12:14] = (3).to_bytes(2, "little"); fmt[14:16] = (24).to_bytes(2, "little"); fmt[8:12] = (16000 * 3).to_bytes(4, "little")
    body = bytes(range(240))
    raw24 = hdr24[:fi] + bytes(fmt) + hdr24[fi + 16:] + b"data" + len(body).to_bytes(4, "little") + body
    raw24 = raw24[:4] + (len(raw24) - 8).to_bytes(4, "little") + raw24[8:]
    open(s24, "wb").write(raw24)
    assert A.wav_probe(s24) is None and A.decode_raw(s24).pcm.shape == (1, 80)
    assert A.wav_probe(str(tmp_path / "missing.wav")) is None
    junk = str(tmp_path / "junk.wav")
    open(junk, "wb").write(b"not a wave file at all")
    assert A.wav_probe(junk) is None
