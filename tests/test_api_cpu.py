"""CPU: the reference's API surface (names, constants, argument checks, exception types) — no compute."""
import numpy as np
import pytest
import torch

import audiotoken_amd
from audiotoken_amd import AUDIO_EXTS, TAR_EXTS, ZIP_EXTS, AudioToken, Tokenizers
from audiotoken_amd.configs import (AcousticEncoderConfig, Wav2VecBertConfig, HubertEncoderConfig, bandwidth_to_num_codebooks,
                                    num_codebooks_to_bandwidth)
from audiotoken_amd.encoder import encodec_bandwidth_to_nq


def test_exports_and_enum_values():
    assert set(audiotoken_amd.__all__) >= {"AudioToken", "Tokenizers", "AUDIO_EXTS", "TAR_EXTS", "ZIP_EXTS", "read_audio"}
    assert [t.value for t in Tokenizers] == ["acoustic", "semantic_s", "semantic_m"]
    assert Tokenizers("acoustic") is Tokenizers.acoustic and str(Tokenizers.semantic_m) == "semantic_m"
    assert AUDIO_EXTS == ('.mp3', '.flac', '.wav', '.ogg', '.opus') and '.tar.gz' in TAR_EXTS and ZIP_EXTS == ('.zip', '.ZIP')
    assert callable(audiotoken_amd.read_audio)


def test_ctor_contract():
    t = AudioToken(Tokenizers.acoustic)
    assert t.num_codebooks == 16 and t.device == "cpu" and t.encoder is None and t.decoder is None   # core.py:58-67
    assert t.model_sample_rate == 24000 and t.model_config.bandwidth == 12 and t.model_config.model_token_rate == 75
    s = AudioToken("semantic_m", device="cuda:0")
    assert s.model_sample_rate == 16000 and s.model_config.output_layer == 19 and s.model_config.model_token_rate == 50
    assert AudioToken("semantic_s").model_config.output_layer == 11
    with pytest.raises(AssertionError):
        AudioToken(Tokenizers.acoustic, num_codebooks=3)
    with pytest.raises(ValueError):
        AudioToken("whisper")


def test_bandwidth_maps():
    for nq in (2, 4, 8, 16):
        bw = num_codebooks_to_bandwidth(nq)
        assert bandwidth_to_num_codebooks(bw) == nq == encodec_bandwidth_to_nq(bw)


def test_encode_argument_checks_raise_reference_exceptions(monkeypatch):
    t = AudioToken(Tokenizers.acoustic, num_codebooks=8)
    monkeypatch.setattr(t, "load_encoder", lambda: None)   # argument checks happen before any device work
    with pytest.raises(AssertionError):
        t.encode(np.zeros((2, 100), dtype=np.float32))
    with pytest.raises(AssertionError):
        t.encode(torch.zeros(100))
    with pytest.raises(NotImplementedError):
        t.encode(b"abc")
    with pytest.raises(ValueError):
        t.encode(123)
    with pytest.raises(AssertionError):
        t.encode_batch_files(batch_size=2, outdir="/tmp/x")
    monkeypatch.setattr(t, "load_decoder", lambda **k: None)
    with pytest.raises(ValueError):
        t.decode(123)


def test_cpu_device_is_rejected_loudly():
    t = AudioToken(Tokenizers.acoustic, num_codebooks=8)   # reference default device="cpu"
    with pytest.raises(ValueError, match="MI355X"):
        t.load_encoder()


def test_resample_shapes_and_dc_gain():
    from audiotoken_amd.audio_io import convert_audio, resample
    x = torch.ones(1, 44100)
    y = resample(x, 44100, 16000)
    assert y.shape == (1, 16000)
    assert abs(float(y[0, 2000:14000].mean()) - 1.0) < 1e-3
    assert convert_audio(torch.randn(2, 1000), 16000, 16000).shape == (1, 1000)
    with pytest.raises(RuntimeError):
        convert_audio(torch.randn(3, 1000), 16000, 16000)
