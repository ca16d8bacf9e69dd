"""GPU: the device feeder of encode_batch_files (audiotoken_amd/feeder.py + csrc/audio_device.hip) against the host data flow it
replaces (AudioToken._chunk_stream + collate_fn = reference utils.py:71-101 + datasets.py:75-139): the same rows in the same order — bit-identical at the
model's sample rate, within fp32 summation order (1e-6) when a chunk is resampled — the same AudioConfig per row, and the same token files end to end."""
import os
import tarfile

import numpy as np
import pytest
import torch

from audiotoken_amd import audio_io as A
from audiotoken_amd import weights as W
from audiotoken_amd.harness import collate_fn

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _host_rows(tok, files, chunk):
    rows = list(tok._chunk_stream([str(f) for f in files], chunk, 0))
    segs, masks, cfgs = collate_fn(rows)
    return segs, masks, cfgs


def _feeder_rows(tok, files, chunk, batch_size, workers=0, transform=None):
    from audiotoken_amd.feeder import DeviceFeeder
    skipped = []
    f = DeviceFeeder("cuda:0", tok.model_config.model_sample_rate, chunk, tok.model_config.model_token_rate, tok.model_config.pad_token, workers,
                     lambda n, why: skipped.append((n, why)), transform=transform)
    segs, masks, cfgs = [], [], []
    for s, m, ptrs, ev in f.batches([str(x) for x in files], batch_size):
        torch.cuda.current_stream().wait_event(ev)
        segs.append(s.cpu()); masks.append(m.cpu()); cfgs.extend(ptrs)
    return torch.cat(segs), torch.cat(masks), cfgs, skipped, f.timings


def _tok(name="acoustic"):
    from audiotoken_amd import AudioToken, Tokenizers
    t = AudioToken(getattr(Tokenizers, name), device="cuda:0", num_codebooks=2)
    t.skipped_files = []
    return t


def _write(path, x, sr, kind):
    from scipy.io import wavfile
    if kind == "s16":
        wavfile.write(str(path), sr, np.round(x * 20000).astype(np.int16))
    elif kind == "f32":
        wavfile.write(str(path), sr, x.astype(np.float32))
    elif kind == "u8":
        wavfile.write(str(path), sr, (np.round(x * 100) + 128).astype(np.uint8))
    elif kind == "s32":
        wavfile.write(str(path), sr, np.round(x * (1 << 30)).astype(np.int32))


def _same_configs(a, b):
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert (os.path.basename(x.file_name), x.length_samples, x.start_idx, x.end_idx, x.length_tokens) == \
               (os.path.basename(y.file_name), y.length_samples, y.start_idx, y.end_idx, y.length_tokens)
        assert abs(x.length_seconds - y.length_seconds) < 1e-12


@pytest.mark.parametrize("kind", ["s16", "f32", "u8", "s32"])
def test_native_rate_rows_are_bit_identical(cuda_device, tmp_path, kind):
    """Files at the model's rate: 2.7 chunks (a padded last segment), exactly one chunk, and a tail below the 3200-sample rule (dropped by both)."""
    tok = _tok()
    sr = 24000
    lens = [int(sr * 2.7), sr, sr + 1000]
    files = []
    for i, n in enumerate(lens):
        p = tmp_path / f"f{i}.wav"
        _write(p, W.synth_waveform(1, n, sr, seed=300 + i)[0], sr, kind)
        files.append(p)
    hs, hm, hc = _host_rows(tok, files, 1)
    ds, dm, dc, skipped, _ = _feeder_rows(tok, files, 1, batch_size=3)
    assert skipped == [] and hs.shape == ds.shape == (5, sr)      # 3 + 1 + 1 rows; the 1000-sample tail of the third file is dropped by both
    assert torch.equal(hs, ds) and torch.equal(hm, dm)
    _same_configs(hc, dc)


@pytest.mark.parametrize("src,dst_tok", [(44100, "semantic_m"), (48000, "acoustic"), (8000, "semantic_m"), (22050, "acoustic")])
def test_resampled_rows_match_the_host_resampler(cuda_device, tmp_path, src, dst_tok):
    """Per-chunk resampling on the device (44.1 -> 16 k, 48 -> 24 k, 8 -> 16 k up-sampling, 22.05 -> 24 k): the ceil(n * L / o) length rule of every chunk,
    the chunk seams (each chunk is filtered on its own: zeros beyond its ends, like the reference), the padded last segment and its mask."""
    tok = _tok(dst_tok)
    dst = tok.model_config.model_sample_rate
    n = int(src * 2.37) + 11
    p = tmp_path / "x.wav"
    _write(p, W.synth_waveform(1, n, src, seed=77)[0], src, "s16")
    hs, hm, hc = _host_rows(tok, [p], 1)
    ds, dm, dc, skipped, _ = _feeder_rows(tok, [p], 1, batch_size=2)
    assert skipped == [] and hs.shape == ds.shape
    assert torch.equal(hm, dm), "masks (i.e. the resampled length of every chunk) differ"
    _same_configs(hc, dc)
    err = (hs - ds).abs().max().item()
    print(f"{src} -> {dst}: {hs.shape[0]} rows, max |device - host| {err:.2e} (max |x| {hs.abs().max().item():.3f})")
    assert err <= 1e-6
    # and against a float64 evaluation of the same polyphase sum (the summation-order-free reference of both)
    kernels, _, o, nn, width = A.resample_table(src, dst)
    raw = A.decode_raw(p)
    x = raw.to_float()[0, :src].double()                      # first chunk
    xp = torch.nn.functional.pad(x, (width, width + o))
    Lr = A.resampled_length(src, src, dst)
    y64 = torch.nn.functional.conv1d(xp[None, None], kernels.double(), stride=o)[0].t().reshape(-1)[:Lr]
    e_dev = (ds[0, :Lr].double() - y64[:dst]).abs().max().item()
    e_host = (hs[0, :Lr].double() - y64[:dst]).abs().max().item()
    print(f"    vs float64: device {e_dev:.2e}, host conv1d {e_host:.2e}")
    assert e_dev <= 5e-7
    # and against the INDEPENDENT oracle (oracle/resample_ref.py: torchaudio's published sinc_interp_hann evaluated per output sample in float64, no
    # polyphase table, no code shared with audio_io.resample_table) on the whole first chunk, chunk ends included, and on the (shorter) last chunk
    from oracle import resample_ref as RR
    full = raw.to_float()[0].numpy()
    first = RR.sinc_interp_hann(full[:src], src, dst)
    assert len(first) == dst
    e_first = float(np.abs(ds[0, :dst].double().numpy() - first).max())
    tail = RR.sinc_interp_hann(full[2 * src:], src, dst)                     # third streamed chunk: 0.37 s + 11 samples
    assert len(tail) == A.resampled_length(n - 2 * src, src, dst)
    e_tail = float(np.abs(ds[2, :len(tail)].double().numpy() - tail).max())
    print(f"    vs the independent oracle: first chunk {e_first:.2e}, last (short) chunk {e_tail:.2e}")
    assert e_first <= 5e-7 and e_tail <= 5e-7
    assert float(ds[2, len(tail):].abs().max()) == 0.0 and float(dm[2, :len(tail)].min()) == 1.0 and float(dm[2, len(tail):].max()) == 0.0


def test_flac_tar_and_skips_through_the_feeder(cuda_device, tmp_path):
    """FLAC (16-bit -> int16 PCM, 24-bit -> int32 PCM on the device), a tar with a FLAC member and a README, a stereo file and an mp3: the feeder yields the
    host path's rows and reports the same skips."""
    tok = _tok("semantic_m")
    with tarfile.open(tmp_path / "x.tar", "w") as tar:
        tar.add(os.path.join(G, "flac_a.flac"), arcname="d/flac_a.flac")
        tar.add(__file__, arcname="d/README.txt")
    (tmp_path / "song.mp3").write_bytes(b"ID3\x03")
    files = [os.path.join(G, "flac_a.flac"), os.path.join(G, "flac_c.flac"), tmp_path / "x.tar", os.path.join(G, "flac_b.flac"), tmp_path / "song.mp3"]
    tok.skipped_files = []
    hs, hm, hc = _host_rows(tok, files, 1)
    host_skipped = sorted(os.path.basename(n) for n, _ in tok.skipped_files)
    ds, dm, dc, skipped, timings = _feeder_rows(tok, files, 1, batch_size=4, workers=2)
    assert sorted(os.path.basename(n) for n, _ in skipped) == host_skipped == ["README.txt", "flac_b.flac", "song.mp3"]
    assert hs.shape == ds.shape and torch.equal(hm, dm)
    _same_configs(hc, dc)
    nat = [i for i, c in enumerate(hc) if c.file_name.endswith("flac_a.flac")]            # 16 kHz: the model's rate -> exact
    assert torch.equal(hs[nat], ds[nat])
    assert (hs - ds).abs().max().item() <= 1e-6
    assert timings["files"] == 3 and timings["segments"] == hs.shape[0]


def test_token_files_are_the_same_with_and_without_the_feeder(cuda_device, tmp_path):
    """encode_batch_files end to end on the acoustic tokenizer: a 24 kHz WAV, a 44.1 kHz WAV, a 16 kHz FLAC — the token files written through the device
    feeder equal the ones written through the host data flow (device_feeder=False), and, for the 44.1 kHz file, the tokens of the host-resampled
    waveform encoded directly."""
    from scipy.io import wavfile
    from audiotoken_amd import AudioToken, Tokenizers
    w = W.synth_encodec_weights(seed=0, with_decoder=False)
    _write(tmp_path / "n.wav", W.synth_waveform(1, 24000 * 3 + 500, 24000, seed=5)[0], 24000, "s16")
    _write(tmp_path / "r.wav", W.synth_waveform(1, int(44100 * 2.2), 44100, seed=6)[0], 44100, "s16")
    files = [tmp_path / "n.wav", tmp_path / "r.wav", os.path.join(G, "flac_a.flac")]
    tok = AudioToken(Tokenizers.acoustic, device="cuda:0", num_codebooks=8, weights=w)
    tok.encode_batch_files(batch_size=3, outdir=tmp_path / "dev", chunk_size=1, audio_files=files, num_workers=2)
    assert tok.feeder_timings is not None and tok.feeder_timings["segments"] == 3 + 3 + 2     # (the 500-sample tail of n.wav is below the 3200-sample rule)
    tok.encode_batch_files(batch_size=3, outdir=tmp_path / "host", chunk_size=1, audio_files=files, num_workers=0, device_feeder=False)
    assert tok.feeder_timings is None
    for name in ("n.npy", "r.npy", "flac_a.npy"):
        a, b = np.load(tmp_path / "dev" / name), np.load(tmp_path / "host" / name)
        assert a.shape == b.shape and a.dtype == np.int16
        same = float((a == b).mean())
        print(f"{name}: {a.shape} tokens, device feeder == host path at {same:.4f}")
        assert np.array_equal(a, b)
    # the 44.1 kHz file: host-resampled chunks, encoded directly
    chunks = [c for c, _ in A.process_audio_chunks(tmp_path / "r.wav", 24000, 1)]
    direct = []
    for c in chunks:
        seg = torch.zeros(1, 24000)
        seg[0, :c.shape[1]] = c[0]
        t = tok.encoder(seg.cuda(), torch.ones_like(seg).cuda()).cpu().numpy()[0]
        direct.append(t[:, :int(np.ceil(c.shape[1] / 24000 * 75))])
    assert np.array_equal(np.hstack(direct), np.load(tmp_path / "dev" / "r.npy"))


# ---- round 5: Tokenizers.semantic_s through the device feeder (its per-chunk transform on the device) ---------------------------------------------------
@pytest.mark.parametrize("src", [16000, 44100, 8000])
def test_semantic_s_rows_match_the_host_transform(cuda_device, tmp_path, src):
    """The reference normalises every streamed chunk with hubert_processor BEFORE cutting and padding it (audiotoken/encoder.py:20-26 applied at
    datasets.py:78-79). The feeder's transform="zmuv" does it in its kernels (float64 chunk moments in a fixed order, then (x - mean) / sqrt(var + 1e-7)): rows
    equal the host flow's (numpy float32 mean / var on the host-resampled chunk) to <= 1e-6 relative to unit-variance samples, masks and AudioConfigs are
    identical, padding stays 0, and two runs are bit-identical. Files: 2.37 chunks (a padded last row whose moments span only its own samples), speech-like
    and 4-sine content, a quiet file (level -46 dB: the normalisation multiplies by ~200)."""
    from audiotoken_amd import synthetic as S
    from audiotoken_amd.hubert import hubert_processor
    tok = _tok("semantic_s")
    tok.transform_func = hubert_processor
    dst = tok.model_config.model_sample_rate
    assert dst == 16000
    n = int(src * 2.37) + 11
    files = []
    for i, x in enumerate((S.speech_like_waveform(1, n, src, seed=900 + src)[0], W.synth_waveform(1, n, src, seed=78)[0] * 0.005, W.synth_waveform(1, src, src, seed=79)[0])):
        p = tmp_path / f"x{i}.wav"
        _write(p, x / max(1e-9, np.abs(x).max()) * (0.9 if i != 1 else 0.005), src, "s16")
        files.append(p)
    hs, hm, hc = _host_rows(tok, files, 1)
    ds, dm, dc, skipped, _ = _feeder_rows(tok, files, 1, batch_size=4, transform="zmuv")
    assert skipped == [] and hs.shape == ds.shape == (7, dst)
    assert torch.equal(hm, dm)
    _same_configs(hc, dc)
    err = (hs - ds).abs().max().item()
    print(f"semantic_s rows from {src} Hz: max |device - host hubert_processor| {err:.2e} (max |row| {hs.abs().max().item():.2f}); "
          f"row means {ds.sum(1).div(dm.sum(1)).abs().max().item():.1e}")
    assert err <= 1e-6 * max(1.0, hs.abs().max().item())
    assert float(ds[dm == 0].abs().max()) == 0.0                                # padding is pad_token, not (0 - mean) / std
    valid = dm[2] == 1                                                            # the short last chunk of file 0: its OWN moments
    assert abs(float(ds[2][valid].mean())) < 1e-5 and abs(float(ds[2][valid].var(unbiased=False)) - 1.0) < 1e-4
    ds2 = _feeder_rows(tok, files, 1, batch_size=4, transform="zmuv")[0]
    assert torch.equal(ds, ds2)


def test_semantic_s_token_files_through_the_feeder(cuda_device, tmp_path):
    """encode_batch_files of Tokenizers.semantic_s end to end: the device feeder is now used (round 5) and writes the token files the host data flow writes."""
    from audiotoken_amd import AudioToken, Tokenizers
    from audiotoken_amd import synthetic as S
    w = W.synth_hubert_weights(11, 0, True)
    _write(tmp_path / "n.wav", S.speech_like_waveform(1, 16000 * 3 + 500, 16000, seed=15)[0] * 4, 16000, "s16")
    _write(tmp_path / "r.wav", W.synth_waveform(1, int(44100 * 2.2), 44100, seed=16)[0], 44100, "s16")
    files = [tmp_path / "n.wav", tmp_path / "r.wav", os.path.join(G, "flac_a.flac")]
    tok = AudioToken(Tokenizers.semantic_s, device="cuda:0", weights=w)
    tok.encode_batch_files(batch_size=3, outdir=tmp_path / "dev", chunk_size=1, audio_files=files, num_workers=2)
    assert tok.feeder_timings is not None and tok.feeder_timings["segments"] == 3 + 3 + 2
    tok.encode_batch_files(batch_size=3, outdir=tmp_path / "host", chunk_size=1, audio_files=files, num_workers=0, device_feeder=False)
    assert tok.feeder_timings is None
    for name in ("n.npy", "r.npy", "flac_a.npy"):
        a, b = np.load(tmp_path / "dev" / name), np.load(tmp_path / "host" / name)
        assert a.shape == b.shape and a.dtype == np.int16 and a.shape[0] == 1
        print(f"{name}: {a.shape} tokens, device feeder == host path at {float((a == b).mean()):.4f}")
        assert np.array_equal(a, b)
