"""CPU: the FLAC input path (VERDICT round 3, next #2c). The decoder is host C++ inside the product library
(audiotoken_amd/csrc/flac_decode.hip: no device call), so these tests need the built library but no GPU. Fixtures and the independent Python ENCODER that
wrote them: tests/golden/make_flac.py. Pinning: self-pinned — RFC 9639 restated twice, in opposite directions and two languages, plus the STREAMINFO MD5; no
libFLAC-produced file exists offline."""
import io
import os
import tarfile

import numpy as np
import pytest
import torch

from audiotoken_amd import audio_io as A

G = os.path.join(os.path.dirname(__file__), "golden")
PCM = np.load(os.path.join(G, "flac_pcm.npz"))
META = {"a": (16000, 1, 16), "b": (44100, 2, 16), "c": (48000, 1, 24)}


@pytest.mark.parametrize("k", "abc")
def test_fixture_decodes_to_its_pcm(k):
    """a: mono 16-bit, fixed / LPC / verbatim subframes, 4- and 5-bit Rice parameters, a short last block with an explicit 16-bit block size;
    b: stereo, all four channel assignments (independent, left-side, side-right, mid-side); c: 24-bit, wasted bits, constant subframes, escaped
    partitions, the sample size taken from STREAMINFO. Every frame's CRC-8 / CRC-16 and the stream's MD5 are checked by the decoder / loader."""
    raw = A.decode_raw(os.path.join(G, f"flac_{k}.flac"))
    sr, ch, bits = META[k]
    assert raw.sample_rate == sr and raw.pcm.shape == PCM[k].shape and raw.pcm.shape[0] == ch
    assert raw.pcm.dtype == (np.int16 if bits <= 16 else np.int32)
    assert np.array_equal(raw.pcm.astype(np.int64), PCM[k].astype(np.int64))
    x = raw.to_float()
    assert x.dtype == torch.float32 and torch.equal(x, torch.from_numpy(PCM[k].astype(np.float32) / np.float32(1 << (bits - 1))))


def test_wav_and_flac_of_the_same_pcm_are_the_same_audio(tmp_path):
    """torchaudio normalises both containers to float32 = integer / 2^(bits - 1): the same PCM must give the same tensor, the same chunks and — through
    encode_batch_files with a stand-in encoder — byte-identical token files."""
    from scipy.io import wavfile
    from audiotoken_amd import AudioToken, Tokenizers
    sr = 16000
    wavfile.write(str(tmp_path / "a.wav"), sr, PCM["a"][0].astype(np.int16))
    xw, srw = A.load(tmp_path / "a.wav")
    xf, srf = A.load(os.path.join(G, "flac_a.flac"))
    assert srw == srf == sr and torch.equal(xw, xf)
    cw = list(A.process_audio_chunks(tmp_path / "a.wav", 24000, 1))
    cf = list(A.process_audio_chunks(os.path.join(G, "flac_a.flac"), 24000, 1))
    assert len(cw) == len(cf) == 2 and all(torch.equal(a[0], b[0]) for a, b in zip(cw, cf))

    class Fake(torch.nn.Module):
        def forward(self, x, m):   # tokens that depend on every sample of the batch row
            return (x.reshape(x.shape[0], 75, -1).sum(-1) * 1000).to(torch.int16)[:, None, :].repeat(1, 2, 1)

    out = {}
    for name, path in (("wav", tmp_path / "a.wav"), ("flac", os.path.join(G, "flac_a.flac"))):
        tok = AudioToken(Tokenizers.acoustic, device="cpu", num_codebooks=2)
        tok.encoder = Fake()
        tok.load_encoder = lambda: None
        tok.encode_batch_files(batch_size=2, outdir=tmp_path / name, chunk_size=1, audio_files=[path], num_workers=0)
        assert tok.skipped_files == []
        (only,) = os.listdir(tmp_path / name)
        out[name] = np.load(tmp_path / name / only)
    assert out["wav"].shape == (2, 113) and np.array_equal(out["wav"], out["flac"])


def test_stereo_flac_is_rejected_like_a_stereo_wav():
    x, sr = A.load(os.path.join(G, "flac_b.flac"))
    assert x.shape == (2, 17640) and sr == 44100
    with pytest.raises(A.AudioDecodeError, match="mono"):
        list(A.process_audio_chunks(os.path.join(G, "flac_b.flac"), 16000, 1))
    assert A.read_audio(os.path.join(G, "flac_b.flac"), 44100).shape == (1, 17640)     # read_audio mixes stereo down (reference utils.py:26-44)


def test_damaged_flac_raises_decode_error(tmp_path):
    blob = bytearray(open(os.path.join(G, "flac_a.flac"), "rb").read())
    for name, mutate in (("crc", lambda b: b.__setitem__(len(b) // 2, b[len(b) // 2] ^ 0x10)), ("cut", lambda b: b.__delitem__(slice(len(b) - 4000, None))),
                         ("md5", lambda b: b.__setitem__(30, b[30] ^ 0xff))):
        bad = bytearray(blob)
        mutate(bad)
        p = tmp_path / f"{name}.flac"
        p.write_bytes(bytes(bad))
        with pytest.raises(A.AudioDecodeError):
            A.decode_raw(p)
    with pytest.raises(A.AudioDecodeError):
        A.decode_raw(tmp_path / "none.flac", io.BytesIO(b"fLaC\x00\x00\x00\x22"))


def test_flac_member_of_a_tar(tmp_path):
    with tarfile.open(tmp_path / "x.tar", "w") as tar:
        tar.add(os.path.join(G, "flac_a.flac"), arcname="d/flac_a.flac")
        info = tarfile.TarInfo("d/README.txt")
        info.size = 5
        tar.addfile(info, io.BytesIO(b"hello"))
    skipped = []
    chunks = list(A.iterate_tar(tmp_path / "x.tar", 16000, 1, lambda n, why: skipped.append(n)))
    assert [c.shape[1] for c, _ in chunks] == [16000, 8000] and all(n == "d/flac_a.flac" for _, n in chunks)
    assert len(skipped) == 1 and skipped[0].endswith("README.txt")
    with pytest.raises(A.AudioDecodeError):       # without a callback the error propagates, as in the reference
        list(A.iterate_tar(tmp_path / "x.tar", 16000, 1))


def test_encoder_of_the_fixtures_is_deterministic():
    """tests/golden/make_flac.py must reproduce the committed fixture bytes (the fixtures are data: seeded synthetic PCM through a committed encoder)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_flac", os.path.join(G, "make_flac.py"))
    mf = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mf)
    kinds = ["fixed", "lpc", "lpc", "verbatim", "fixed", "lpc"]
    blob, _ = mf.encode(PCM["a"][:, :8192 + 777], 16000, 16, 4096, lambda i: {"kind": kinds[i % len(kinds)], "method": i % 2})
    raw = A.decode_raw("x.flac", io.BytesIO(blob))
    assert np.array_equal(raw.pcm, PCM["a"][:, :8192 + 777])
    ref = open(os.path.join(G, "flac_a.flac"), "rb").read()
    assert blob[42:42 + 2000] == ref[42:42 + 2000]          # same first frames as the committed file (the header differs: length, MD5)


# ---- externally produced files (round 5) -------------------------------------------------------------------------------------------------------------
# The three example files of RFC 9639 Appendix D, byte for byte (tests/golden/rfc9639_d{1,2,3}.flac; 57 / 227 / 73 bytes). D.2 carries the vendor string
# "reference libFLAC 1.3.3 20190804": these are libFLAC's output, not this repository's encoder — the only FLAC bytes available offline that the author of
# csrc/flac_decode.hip did not write. The decoder checks every frame's CRC-8 and CRC-16 and the STREAMINFO MD5 of the decoded PCM, so a wrong byte in a
# fixture or a wrong sample out of the decoder cannot pass. Between them: verbatim subframes with wasted bits (D.1), PADDING / SEEKTABLE / VORBIS_COMMENT
# blocks, side-channel stereo, fixed predictors, Rice partitions, a short last block (D.2), an 8-bit 32 kHz LPC subframe of order 3 (D.3).
RFC_EXAMPLES = {
    "rfc9639_d1.flac": (44100, 16, [[25588], [10416]]),
    "rfc9639_d2.flac": (44100, 16, [[10372, 18041, 14942, 17876, 15627, 17899, 16242, 18077, 16824, 18263, 17295, -14418, -15201, -14508, -15195, -14818, -15486, -15349, -16054],
                                    [6070, 10545, 8743, 10449, 9143, 10463, 9502, 10569, 9840, 10680, 10113, -8428, -8895, -8476, -8896, -8653, -9072, -8958, -9410]]),
    "rfc9639_d3.flac": (32000, 8, [[0, 79, 111, 78, 8, -61, -90, -68, -13, 42, 67, 53, 13, -27, -46, -38, -12, 14, 24, 19, 6, -4, -5, 0]]),
}


@pytest.mark.parametrize("name", sorted(RFC_EXAMPLES))
def test_rfc9639_appendix_examples_decode(name):
    sr, bits, pcm = RFC_EXAMPLES[name]
    path = os.path.join(G, name)
    raw = A.decode_raw(path)
    assert raw.sample_rate == sr
    got = (raw.to_float().double() * float(1 << (bits - 1))).round().long()
    assert got.tolist() == pcm
    if name == "rfc9639_d2.flac":
        assert b"reference libFLAC 1.3.3 20190804" in open(path, "rb").read()
    # a flipped payload bit must be caught by the frame CRC (i.e. the fixtures above really were verified, not just parsed)
    data = bytearray(open(path, "rb").read())
    data[-4] ^= 0x10
    with pytest.raises(A.AudioDecodeError):
        A.decode_raw(name, io.BytesIO(bytes(data)))
