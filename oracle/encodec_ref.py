"""CPU oracle for the acoustic tokenizer (EnCodec 24 kHz SEANet encoder/decoder + residual VQ).

TEST INFRASTRUCTURE — never imported by the product path (see oracle/__init__.py).

The reference calls the PyPI package ``encodec`` (not vendored, unpinned in
``requirements.txt:5``; only the 0.1.x line exists) at ``audiotoken/encoder.py:38-52`` and
``audiotoken/decoder.py:60-72``. Its published algorithm is restated here in plain torch-CPU fp32
functional ops, following SURVEY.md Appendix A.1 and the HF ``transformers`` 5.15.0 restatement
(``transformers/models/encodec/modeling_encodec.py``: conv padding rules ``:126-176``, transposed conv
trim ``:206-233``, LSTM+skip ``:236-249``, res-block ``:252-284``, encoder/decoder stacks ``:287-347``,
codebook search ``:364-369``, RVQ ``:424-447``). Pinned by tests/golden/encodec_*.npz, which were produced
by HF ``EncodecModel`` itself with the same synthetic weights (tests/golden/make_golden.py).

Weights: dict with the original ``encodec`` checkpoint key names (audiotoken_amd/weights.py).
Layout here is torch's channel-first ``[B, C, T]``.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

RATIOS_ENC = (2, 4, 5, 8)
RATIOS_DEC = (8, 5, 4, 2)


def _t(w: Dict[str, np.ndarray], key: str) -> torch.Tensor:
    v = w[key]
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(v))


def folded(w, prefix: str) -> torch.Tensor:
    """weight-norm fold: W = g * v / ||v|| over all dims but 0 (modeling_encodec.py:99-106 applies
    torch weight_norm with the default dim=0; for ConvTranspose1d dim 0 is the *input* channel)."""
    return torch._weight_norm(_t(w, prefix + ".weight_v"), _t(w, prefix + ".weight_g"), 0)


def pad1d_reflect(x: torch.Tensor, left: int, right: int) -> torch.Tensor:
    """modeling_encodec.py:139-155: reflect pad, zero-extending short inputs first."""
    length = x.shape[-1]
    max_pad = max(left, right)
    extra = 0
    if length <= max_pad:
        extra = max_pad - length + 1
        x = F.pad(x, (0, extra))
    y = F.pad(x, (left, right), mode="reflect")
    end = y.shape[-1] - extra
    return y[..., :end]


def conv1d_causal(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, stride: int) -> torch.Tensor:
    """Causal EnCodec conv (modeling_encodec.py:126-176): left pad k-s, right pad `extra`, reflect."""
    k = weight.shape[-1]
    padding_total = k - stride
    length = x.shape[-1]
    n_frames = math.ceil((length - k + padding_total) / stride + 1) - 1
    ideal = n_frames * stride + k - padding_total
    extra = ideal - length
    x = pad1d_reflect(x, padding_total, extra)
    return F.conv1d(x, weight, bias, stride=stride)


def convtr1d_causal(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, stride: int) -> torch.Tensor:
    """Causal transposed conv, trim right k-s (modeling_encodec.py:206-233, trim_right_ratio=1)."""
    k = weight.shape[-1]
    y = F.conv_transpose1d(x, weight, bias, stride=stride)
    pad_right = k - stride
    return y[..., : y.shape[-1] - pad_right]


def resblock(w, prefix: str, x: torch.Tensor) -> torch.Tensor:
    """modeling_encodec.py:252-284 with compress=2, kernel (3,1), dilation 1, conv shortcut."""
    h = F.elu(x)
    h = conv1d_causal(h, folded(w, f"{prefix}.block.1.conv.conv"), _t(w, f"{prefix}.block.1.conv.conv.bias"), 1)
    h = F.elu(h)
    h = conv1d_causal(h, folded(w, f"{prefix}.block.3.conv.conv"), _t(w, f"{prefix}.block.3.conv.conv.bias"), 1)
    sc = conv1d_causal(x, folded(w, f"{prefix}.shortcut.conv.conv"), _t(w, f"{prefix}.shortcut.conv.conv.bias"), 1)
    return sc + h


def lstm_skip(w, prefix: str, x: torch.Tensor) -> torch.Tensor:
    """2-layer LSTM over time plus skip (modeling_encodec.py:236-249). x: [B, C, T].

    Explicit recurrence, torch gate order i, f, g, o; h0 = c0 = 0.
    """
    B, C, T = x.shape
    seq = x.permute(2, 0, 1)  # [T, B, C]
    inp = seq
    for layer in range(2):
        w_ih = _t(w, f"{prefix}.lstm.weight_ih_l{layer}")
        w_hh = _t(w, f"{prefix}.lstm.weight_hh_l{layer}")
        b_ih = _t(w, f"{prefix}.lstm.bias_ih_l{layer}")
        b_hh = _t(w, f"{prefix}.lstm.bias_hh_l{layer}")
        H = w_hh.shape[1]
        h = torch.zeros(B, H, dtype=x.dtype)     # (dtype of the input: the conditioning study evaluates this restatement in float64 as well)
        c = torch.zeros(B, H, dtype=x.dtype)
        outs = []
        xg = F.linear(inp, w_ih, b_ih)  # [T, B, 4H]
        for t in range(T):
            gates = xg[t] + F.linear(h, w_hh, b_hh)
            i, f, g, o = gates.chunk(4, dim=1)
            c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
            h = torch.sigmoid(o) * torch.tanh(c)
            outs.append(h)
        inp = torch.stack(outs, 0)
    return (inp + seq).permute(1, 2, 0)


def seanet_encode(w, wav: torch.Tensor, return_stages: bool = False):
    """[B, N] float32 -> emb [B, 128, ceil(N/320)] (modeling_encodec.py:287-314)."""
    stages: List[torch.Tensor] = []
    x = wav.unsqueeze(1)
    x = conv1d_causal(x, folded(w, "encoder.model.0.conv.conv"), _t(w, "encoder.model.0.conv.conv.bias"), 1)
    stages.append(x)
    idx = 1
    for r in RATIOS_ENC:
        x = resblock(w, f"encoder.model.{idx}", x)
        stages.append(x)
        x = F.elu(x)
        p = f"encoder.model.{idx + 2}.conv.conv"
        x = conv1d_causal(x, folded(w, p), _t(w, p + ".bias"), r)
        stages.append(x)
        idx += 3
    x = lstm_skip(w, "encoder.model.13", x)
    stages.append(x)
    x = F.elu(x)
    x = conv1d_causal(x, folded(w, "encoder.model.15.conv.conv"), _t(w, "encoder.model.15.conv.conv.bias"), 1)
    stages.append(x)
    return (x, stages) if return_stages else x


def bandwidth_to_nq(bandwidth: float, frame_rate: int = 75, codebook_size: int = 1024) -> int:
    """modeling_encodec.py:416-422: n_q = max(1, floor(bw*1000 / (log2(1024)*75)))."""
    return int(max(1, math.floor(bandwidth * 1000 / (math.log2(codebook_size) * frame_rate))))


def codebook_search(x: torch.Tensor, embed: torch.Tensor) -> torch.Tensor:
    """modeling_encodec.py:364-369: dist = -(|x|^2 - 2 x@E^T + |e|^2); first maximal index."""
    e = embed.t()
    s = x.pow(2).sum(1, keepdim=True)
    dist = -(s - 2 * x @ e + e.pow(2).sum(0, keepdim=True))
    return dist.max(dim=-1).indices


def rvq_encode(w, emb: torch.Tensor, n_q: int, return_margins: bool = False):
    """emb [B, 128, T] -> int64 codes [n_q, B, T] (modeling_encodec.py:424-438)."""
    B, D, T = emb.shape
    residual = emb.permute(0, 2, 1).reshape(-1, D)
    out = []
    margins = []
    for q in range(n_q):
        embed = _t(w, f"quantizer.vq.layers.{q}._codebook.embed")
        e = embed.t()
        s = residual.pow(2).sum(1, keepdim=True)
        dist = -(s - 2 * residual @ e + e.pow(2).sum(0, keepdim=True))
        idx = dist.max(dim=-1).indices
        if return_margins:
            top2 = dist.topk(2, dim=-1).values
            margins.append((top2[:, 0] - top2[:, 1]).reshape(B, T))
        residual = residual - F.embedding(idx, embed)
        out.append(idx.reshape(B, T))
    codes = torch.stack(out)
    return (codes, torch.stack(margins)) if return_margins else codes


def acoustic_encode(w, wav: torch.Tensor, n_q: int, return_margins: bool = False):
    """Reference AcousticEncoder.forward (audiotoken/encoder.py:44-57): int16 [B, n_q, T]. With return_margins also the
    oracle's own top-2 distance margin of every choice, [B, n_q, T] (what a differing id is explained by, or not)."""
    emb = seanet_encode(w, wav)
    if return_margins:
        codes, margins = rvq_encode(w, emb, n_q, return_margins=True)
        return codes.transpose(0, 1).to(torch.int16), margins.transpose(0, 1)
    codes = rvq_encode(w, emb, n_q)
    return codes.transpose(0, 1).to(torch.int16)


def rvq_decode(w, codes: torch.Tensor) -> torch.Tensor:
    """codes int64 [K, B, T] -> [B, 128, T] sum of codebook rows (modeling_encodec.py:440-447)."""
    out = None
    for q in range(codes.shape[0]):
        embed = _t(w, f"quantizer.vq.layers.{q}._codebook.embed")
        e = F.embedding(codes[q], embed).permute(0, 2, 1)
        out = e if out is None else out + e
    return out


def seanet_decode(w, z: torch.Tensor) -> torch.Tensor:
    """[B, 128, T] -> [B, 1, 320*T] (modeling_encodec.py:317-347)."""
    x = conv1d_causal(z, folded(w, "decoder.model.0.conv.conv"), _t(w, "decoder.model.0.conv.conv.bias"), 1)
    x = lstm_skip(w, "decoder.model.1", x)
    idx = 3
    for r in RATIOS_DEC:
        x = F.elu(x)
        p = f"decoder.model.{idx}.convtr.convtr"
        x = convtr1d_causal(x, folded(w, p), _t(w, p + ".bias"), r)
        x = resblock(w, f"decoder.model.{idx + 1}", x)
        idx += 3
    x = F.elu(x)
    x = conv1d_causal(x, folded(w, "decoder.model.15.conv.conv"), _t(w, "decoder.model.15.conv.conv.bias"), 1)
    return x


def acoustic_decode(w, tokens: torch.Tensor) -> torch.Tensor:
    """Reference AcousticDecoder.forward (audiotoken/decoder.py:66-76): tokens [B, K, T] -> float32 [1, B*320*T]."""
    codes = tokens.to(torch.long).transpose(0, 1)
    z = rvq_decode(w, codes)
    out = seanet_decode(w, z)
    return out.reshape(-1).to(torch.float32).unsqueeze(0)
