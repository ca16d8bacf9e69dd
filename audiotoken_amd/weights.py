"""Weight dictionaries for the tokenizer models: synthetic generation and layout constants.

A model's weights are a flat ``dict[str, np.ndarray(float32)]`` whose keys follow the
checkpoint formats the reference loads, so a real checkpoint can be dropped in later
(SURVEY.md §8(f) N2):

* EnCodec 24 kHz — state-dict keys of ``encodec.EncodecModel.encodec_model_24khz()``
  (reference call site ``audiotoken/encoder.py:38``): ``encoder.model.{i}.conv.conv.{weight_g,
  weight_v,bias}``, ``encoder.model.{i}.block.{1,3}.conv.conv.*``, ``encoder.model.{i}.shortcut.conv.conv.*``,
  ``encoder.model.13.lstm.{weight_ih,weight_hh,bias_ih,bias_hh}_l{0,1}``,
  ``decoder.model.{i}.convtr.convtr.*``, ``quantizer.vq.layers.{k}._codebook.embed``.
* Wav2Vec2-BERT — HF ``Wav2Vec2BertModel`` state-dict keys (reference call site
  ``audiotoken/encoder.py:132``) plus ``vq._codebook.embed`` ``[1, 2048, 1024]``
  (``audiotoken/encoder.py:147-161``, ``audiotoken/utils.py:331-339``).

No pretrained weights exist offline, so benchmarks and parity tests use the synthetic
generators below (counter-based PRNG, bit-reproducible everywhere — see ``prng.py``).
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np

from . import prng

# ----------------------------------------------------------------------------------------------
# EnCodec 24 kHz architecture constants
# ----------------------------------------------------------------------------------------------
ENCODEC_RATIOS_ENC = (2, 4, 5, 8)      # encoder downsampling strides, in order
ENCODEC_RATIOS_DEC = (8, 5, 4, 2)      # decoder upsampling strides, in order
ENCODEC_FILTERS = 32
ENCODEC_DIM = 128
ENCODEC_CODEBOOK = 1024
ENCODEC_MAX_NQ = 32
ENCODEC_HOP = 320
ENCODEC_LSTM = 512


def encodec_conv_specs() -> List[Tuple[str, int, int, int, int]]:
    """(key prefix, c_in, c_out, kernel, stride) of every Conv1d of the SEANet *encoder*, in forward order."""
    specs = [("encoder.model.0.conv.conv", 1, ENCODEC_FILTERS, 7, 1)]
    c = ENCODEC_FILTERS
    idx = 1
    for r in ENCODEC_RATIOS_ENC:
        specs.append((f"encoder.model.{idx}.block.1.conv.conv", c, c // 2, 3, 1))
        specs.append((f"encoder.model.{idx}.block.3.conv.conv", c // 2, c, 1, 1))
        specs.append((f"encoder.model.{idx}.shortcut.conv.conv", c, c, 1, 1))
        specs.append((f"encoder.model.{idx + 2}.conv.conv", c, 2 * c, 2 * r, r))
        c *= 2
        idx += 3
    # idx == 13 -> LSTM, 14 ELU, 15 final conv
    specs.append(("encoder.model.15.conv.conv", c, ENCODEC_DIM, 7, 1))
    return specs


def encodec_decoder_specs() -> List[Tuple[str, str, int, int, int, int]]:
    """(kind, key prefix, c_in, c_out, kernel, stride) for the SEANet *decoder*; kind in {conv, convtr}."""
    c = ENCODEC_FILTERS * 16
    specs = [("conv", "decoder.model.0.conv.conv", ENCODEC_DIM, c, 7, 1)]
    idx = 3  # 1 = LSTM, 2 = ELU
    for r in ENCODEC_RATIOS_DEC:
        specs.append(("convtr", f"decoder.model.{idx}.convtr.convtr", c, c // 2, 2 * r, r))
        c //= 2
        specs.append(("conv", f"decoder.model.{idx + 1}.block.1.conv.conv", c, c // 2, 3, 1))
        specs.append(("conv", f"decoder.model.{idx + 1}.block.3.conv.conv", c // 2, c, 1, 1))
        specs.append(("conv", f"decoder.model.{idx + 1}.shortcut.conv.conv", c, c, 1, 1))
        idx += 3
    # idx == 15 -> ELU at 14, final conv at 15
    specs.append(("conv", "decoder.model.15.conv.conv", c, 1, 7, 1))
    return specs


def _wn_pair(w: Dict[str, np.ndarray], prefix: str, shape, norm_dim0: int, seed: int):
    """weight_g / weight_v / bias for one weight-normalised conv. ``shape`` is the torch weight shape."""
    w[f"{prefix}.weight_v"] = prng.uniform(f"{prefix}.weight_v", shape, -1.0, 1.0, seed)
    w[f"{prefix}.weight_g"] = prng.uniform(f"{prefix}.weight_g", (norm_dim0, 1, 1), 0.9, 1.5, seed)


def synth_encodec_weights(seed: int = 0, with_decoder: bool = True, n_codebooks: int = ENCODEC_MAX_NQ) -> Dict[str, np.ndarray]:
    """Synthetic EnCodec-24kHz weights (encoder, RVQ codebooks, optionally decoder)."""
    w: Dict[str, np.ndarray] = {}
    for prefix, cin, cout, k, _s in encodec_conv_specs():
        _wn_pair(w, prefix, (cout, cin, k), cout, seed)
        w[f"{prefix}.bias"] = prng.uniform(f"{prefix}.bias", (cout,), -0.1, 0.1, seed)
    a = 1.0 / np.sqrt(ENCODEC_LSTM)
    for which in ("encoder.model.13", "decoder.model.1"):
        if which.startswith("decoder") and not with_decoder:
            continue
        for layer in range(2):
            for nm, shape in (("weight_ih", (4 * ENCODEC_LSTM, ENCODEC_LSTM)), ("weight_hh", (4 * ENCODEC_LSTM, ENCODEC_LSTM)),
                              ("bias_ih", (4 * ENCODEC_LSTM,)), ("bias_hh", (4 * ENCODEC_LSTM,))):
                key = f"{which}.lstm.{nm}_l{layer}"
                w[key] = prng.uniform(key, shape, -a, a, seed)
    if with_decoder:
        for kind, prefix, cin, cout, k, _s in encodec_decoder_specs():
            if kind == "conv":
                _wn_pair(w, prefix, (cout, cin, k), cout, seed)
            else:  # ConvTranspose1d weight is [in, out, k]; weight-norm dim 0 = input channel
                _wn_pair(w, prefix, (cin, cout, k), cin, seed)
            w[f"{prefix}.bias"] = prng.uniform(f"{prefix}.bias", (cout,), -0.1, 0.1, seed)
    for q in range(n_codebooks):
        key = f"quantizer.vq.layers.{q}._codebook.embed"
        scale = 1.2 * (0.75 ** q)
        w[key] = prng.irwin_hall(key, (ENCODEC_CODEBOOK, ENCODEC_DIM), scale, seed)
    return w


def fold_weight_norm(g: np.ndarray, v: np.ndarray) -> np.ndarray:
    """``W = g * v / ||v||`` with the norm over all dims but 0 (torch ``weight_norm(dim=0)``).

    Uses ``torch._weight_norm`` on the host — the very function torch's weight-norm parametrisation
    evaluates on every forward of the reference model — so the folded tensor handed to the device is
    bit-identical to the weight the reference convolves with on CPU.
    """
    import torch

    out = torch._weight_norm(torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)),
                             torch.from_numpy(np.ascontiguousarray(g, dtype=np.float32)), 0)
    return out.numpy()


# ----------------------------------------------------------------------------------------------
# synthetic waveforms (SURVEY.md §8(d) "synthetic inputs")
# ----------------------------------------------------------------------------------------------
def synth_waveform(n_clips: int, n_samples: int, sample_rate: int, seed: int = 1234, first_clip: int = 0) -> np.ndarray:
    """``float32 [n_clips, n_samples]`` in [-1, 1]: 0.3*sum of 4 sines (80..4000 Hz) + 0.05*noise, clipped.

    Clip ``i`` depends only on ``seed + first_clip + i`` so shards of a batch are reproducible per rank.
    """
    out = np.empty((n_clips, n_samples), dtype=np.float32)
    t = np.arange(n_samples, dtype=np.float64) / float(sample_rate)
    for i in range(n_clips):
        s = seed + first_clip + i
        f = 80.0 + 3920.0 * prng.uniform01("wave.freq", 4, s).astype(np.float64)
        ph = 2.0 * np.pi * prng.uniform01("wave.phase", 4, s).astype(np.float64)
        x = np.zeros(n_samples, dtype=np.float64)
        for j in range(4):
            x += np.sin(2.0 * np.pi * f[j] * t + ph[j])
        noise = prng.irwin_hall("wave.noise", (n_samples,), 1.0, s).astype(np.float64)
        out[i] = np.clip(0.3 * x + 0.05 * noise, -1.0, 1.0).astype(np.float32)
    return out
