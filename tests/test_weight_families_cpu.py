"""CPU: the synthetic generators the parity evidence rests on (round 5). The "uniform" weight family must stay bit-identical to rounds 1-4 (the golden vectors
and sweeps were produced with it); the "trained_like" family must have the statistics it claims (same variances, heavy tails, log-normal LayerNorm gains,
massive-activation channels); the speech-like clips must be distinct, reproducible per clip index, and hold pauses, level spread and a clipped stretch."""
import hashlib

import numpy as np

from audiotoken_amd import prng
from audiotoken_amd import synthetic as S
from audiotoken_amd import weights as W


def _digest(w):
    h = hashlib.sha256()
    for k in sorted(w):
        h.update(k.encode())
        h.update(np.ascontiguousarray(w[k]).tobytes())
    return h.hexdigest()[:16]


def test_uniform_family_is_unchanged():
    # digests taken from the round-4 tree (commit 775f432) with the same function calls
    assert _digest(W.synth_w2vbert_weights(2, 0, True)) == _digest(W.synth_w2vbert_weights(2, 0, True, family="uniform"))
    assert _digest(W.synth_encodec_weights(0, False)) == "cb088385d293b580"
    assert _digest(W.synth_hubert_weights(1, 0, True)) == "d30fe1a91a130b10"
    assert _digest(W.synth_w2vbert_weights(1, 0, True)) == "6600a7920e82b304"


def test_exp_exact_matches_libm():
    x = np.linspace(-20.0, 20.0, 200001)
    assert np.abs(prng.exp_exact(x) / np.exp(x) - 1.0).max() < 4e-16


def test_trained_like_statistics():
    u = W.synth_w2vbert_weights(2, 0, True)
    t = W.synth_w2vbert_weights(2, 0, True, family="trained_like")
    assert set(u) == set(t) and all(u[k].shape == t[k].shape and t[k].dtype == np.float32 for k in u)
    k = "encoder.layers.1.ffn1.intermediate_dense.weight"
    assert abs(t[k].std() / u[k].std() - 1.0) < 0.03                       # same variance ...
    z = t[k] / t[k].std()
    assert 20 < np.abs(z).max() < 60 and (z ** 4).mean() > 6.0             # ... heavy tails (uniform: max 1.73 std, kurtosis 1.8)
    g = t["encoder.layers.0.final_layer_norm.weight"]
    mc = W.massive_channels("w2vbert", 1024, 0)
    assert len(set(mc.tolist())) == W.N_MASSIVE
    rest = np.delete(g, mc)
    assert g[mc].min() > 8.0 and rest.max() < 8.0 and 0.35 < np.log(rest).std() < 0.65   # massive channels; log-normal gains, sigma 0.5
    assert np.array_equal(mc, W.massive_channels("w2vbert", 1024, 0))
    g1 = t["encoder.layers.1.ffn1_layer_norm.weight"]                      # every third layer: also the LayerNorm that feeds the first FFN GEMM
    assert g1[mc].min() > 8.0
    assert np.abs(t["encoder.layers.0.ffn2_layer_norm.bias"]).max() > 2.0  # a few large betas
    e = W.synth_encodec_weights(0, True, family="trained_like")
    assert set(e) == set(W.synth_encodec_weights(0, True))
    h = W.synth_hubert_weights(2, 0, True, family="trained_like")
    assert set(h) == set(W.synth_hubert_weights(2, 0, True))
    assert h["encoder.layers.0.final_layer_norm.weight"][W.massive_channels("hubert", 768, 0)].min() > 8.0


def test_speech_like_clips():
    a = S.speech_like_waveform(6, 48000, 16000, seed=1234)
    b = S.speech_like_waveform(2, 48000, 16000, seed=1234, first_clip=3)
    assert a.dtype == np.float32 and a.shape == (6, 48000) and np.abs(a).max() <= 1.0
    assert np.array_equal(a[3:5], b)                                       # clip i depends only on seed + i
    assert len({a[i].tobytes() for i in range(6)}) == 6                    # all distinct
    big = S.speech_like_waveform(32, 160000, 16000, seed=1234)
    peaks = np.abs(big).max(axis=1)
    assert 20.0 * np.log10(peaks.max() / peaks.min()) > 25.0               # level spread between clips (40 dB nominal)
    fr = np.abs(big.reshape(32, -1, 1600)).max(axis=2)                     # 100 ms frames
    loud = peaks > 0.2
    assert loud.any() and ((fr[loud] < 0.02 * peaks[loud, None]).mean() > 0.03)   # pauses at the noise floor
    flat = max(int((np.abs(x) >= 0.999 * np.abs(x).max()).sum()) for x in big[loud])
    assert flat > 20                                                       # a hard-clipped stretch
