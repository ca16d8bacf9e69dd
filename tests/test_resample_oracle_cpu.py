"""CPU: the host resampler (audiotoken_amd/audio_io.resample — the polyphase conv1d twin of the device kernel) against oracle/resample_ref.py, an
independent per-output-sample float64 evaluation of torchaudio's published ``sinc_interp_hann`` filter (no table, no shared code): reference
audiotoken/utils.py:82-98 resamples every chunk with ``torchaudio.transforms.Resample`` defaults. Whole chunks are compared, i.e. INCLUDING the chunk
ends where the zero padding of ``_apply_sinc_resample_kernel`` decides the values. torchaudio itself is absent offline: pinned to the formula, not to a run."""
import numpy as np
import pytest
import torch

from audiotoken_amd import audio_io as A
from audiotoken_amd import synthetic as S
from oracle import resample_ref as R

PAIRS = [(44100, 16000), (48000, 24000), (8000, 16000), (22050, 24000), (16000, 24000), (32000, 16000)]


@pytest.mark.parametrize("src,dst", PAIRS)
def test_host_resampler_matches_the_independent_oracle(src, dst):
    n = int(src * 0.73) + 13
    rng = np.random.default_rng(src + dst)
    noise = (rng.standard_normal(n) * 0.3).clip(-1, 1).astype(np.float32)
    speech = S.speech_like_waveform(1, n, src, seed=src)[0]
    speech = speech / max(1e-6, np.abs(speech).max()) * 0.9
    for name, x in (("noise", noise), ("speech-like", speech)):
        got = A.resample(torch.from_numpy(x)[None], src, dst)[0].double().numpy()
        ref = R.sinc_interp_hann(x, src, dst)
        assert got.shape == ref.shape == (R.resampled_length(n, src, dst),) and A.resampled_length(n, src, dst) == len(ref)
        err = np.abs(got - ref)
        edge = max(err[:64].max(), err[-64:].max())
        print(f"{src} -> {dst} ({name}): max |host - oracle| {err.max():.2e} (chunk ends {edge:.2e}), max |y| {np.abs(ref).max():.3f}")
        assert err.max() <= 1e-6


def test_table_taps_equal_the_direct_formula():
    """Every tap of audio_io.resample_table (what the DEVICE kernel multiplies with) equals the oracle's direct evaluation of the filter at that tap's time —
    checked through impulses: resampling a unit impulse at position m reads column m of the filter."""
    for src, dst in ((44100, 16000), (8000, 16000)):
        n = 400
        for pos in (0, 1, 57, 199, n - 1):
            x = np.zeros(n, dtype=np.float32)
            x[pos] = 1.0
            got = A.resample(torch.from_numpy(x)[None], src, dst)[0].double().numpy()
            ref = R.sinc_interp_hann(x, src, dst)
            assert np.abs(got - ref).max() <= 1e-9, (src, dst, pos)      # single products: only the shared fp32 rounding of the taps is involved


def test_oracle_properties():
    # a band-limited tone comes back as the same tone (away from the ends), the length rule is torchaudio's ceil
    sr, tr, n = 44100, 16000, 8000
    t = np.arange(n) / sr
    x = 0.5 * np.sin(2 * np.pi * 440.0 * t + 0.3)
    y = R.sinc_interp_hann(x, sr, tr)
    assert len(y) == int(np.ceil(n * tr / sr))
    tt = np.arange(len(y)) / tr
    assert np.abs(y - 0.5 * np.sin(2 * np.pi * 440.0 * tt + 0.3))[64:-64].max() < 2e-3
    assert np.array_equal(R.sinc_interp_hann(x, 16000, 16000), x)
