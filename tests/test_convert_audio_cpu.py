"""Mirror of the reference's only arithmetic unit test, ``test/convert_audio.py`` (TestAudioConversion): the same
mono/stereo x source-rate x target-rate matrix and the 3-channel error case, for ``audiotoken_amd.audio_io.convert_audio``
(reference ``audiotoken/utils.py:26-44``). The reference compares against ``encodec.utils.convert_audio`` (torchaudio's
``Resample``); neither package exists offline, so the resampler is pinned through properties torchaudio's kernel has:
output length ``ceil(n * new / orig)``, linearity (stereo down-mix commutes with resampling), identity at equal rates and
reproduction of a band-limited tone."""
import math

import numpy as np
import pytest
import torch

from audiotoken_amd.audio_io import convert_audio

SRC = [44100, 48000, 24000, 16000]
DST = [16000, 22050, 24000, 48000]


def _tone(sr, n, f=440.0, ch=1, seed=0):
    t = torch.arange(n, dtype=torch.float64) / sr
    g = torch.Generator().manual_seed(seed)
    ph = torch.rand(ch, 1, generator=g, dtype=torch.float64) * 2 * math.pi
    return (0.5 * torch.sin(2 * math.pi * f * t.unsqueeze(0) + ph)).float(), ph


@pytest.mark.parametrize("sr", SRC)
@pytest.mark.parametrize("tr", DST)
def test_mono_matrix(sr, tr):
    n = sr // 4
    x, ph = _tone(sr, n)
    y = convert_audio(x, sr, tr)
    assert y.dtype == torch.float32 and y.dim() == 2 and y.shape[0] == 1
    if sr == tr:
        assert torch.equal(y, x)
        return
    assert y.shape[1] == math.ceil(n * tr / sr)
    # a 440 Hz tone is far below both Nyquist rates: away from the edges the resampled signal is the same tone
    t = torch.arange(y.shape[1], dtype=torch.float64) / tr
    ref = 0.5 * torch.sin(2 * math.pi * 440.0 * t + ph[0])
    m = slice(64, y.shape[1] - 64)
    assert (y[0, m].double() - ref[m]).abs().max().item() < 2e-3


@pytest.mark.parametrize("sr", SRC)
@pytest.mark.parametrize("tr", DST)
def test_stereo_matrix(sr, tr):
    n = sr // 8
    x, _ = _tone(sr, n, ch=2, seed=sr + tr)
    y = convert_audio(x, sr, tr)
    assert y.shape[0] == 1
    # down-mix first (reference utils.py:33-35), then resample: equals the mean of the separately converted channels
    y0 = convert_audio(x[0:1], sr, tr)
    y1 = convert_audio(x[1:2], sr, tr)
    assert torch.allclose(y, 0.5 * (y0 + y1), atol=1e-6)


def test_three_channels_raise():
    with pytest.raises(RuntimeError):
        convert_audio(torch.zeros(3, 1000), 16000, 24000)


def test_noise_energy_is_preserved_when_upsampling():
    g = torch.Generator().manual_seed(7)
    x = torch.randn(1, 16000, generator=g) * 0.1
    y = convert_audio(x, 16000, 48000)
    ex, ey = float((x ** 2).mean()), float((y ** 2).mean())
    # band-limited interpolation keeps the power up to the transition band of the width-6 Hann-windowed sinc (~8 % of a
    # white spectrum) and never adds any
    assert 0.85 < ey / ex <= 1.0
