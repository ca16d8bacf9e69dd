"""CPU oracle for the semantic_s tokenizer: mHuBERT-base (HuBERT-base architecture) hidden state 11 -> non-affine
LayerNorm -> nearest of 1000 k-means centres.

TEST INFRASTRUCTURE — never imported by the product path (see oracle/__init__.py).

Restates reference ``HubertEncoder.__call__`` (audiotoken/encoder.py:87-108) and ``hubert_processor``
(encoder.py:20-26 -> HF ``Wav2Vec2FeatureExtractor``: per-clip zero-mean / unit-variance, eps 1e-7). The model is the
dependency ``transformers`` ``HubertModel`` (unpinned, requirements.txt:4), restated from HF 5.15.0
``models/hubert/modeling_hubert.py``: conv feature encoder ``:106-213`` (GroupNorm after conv 0 only, GELU), feature
projection ``:216-231``, positional conv (k128, groups 16, weight-norm dim 2, last frame dropped) ``:45-103``, post-LN
encoder layer ``:371-404``, encoder entry ``:417-440``, frame-level mask ``:664-693``. Pinned by
tests/golden/hubert_a.npz (HF ``HubertModel(HubertConfig())`` with the same synthetic weights). k-means centres come from a
joblib pickle of sklearn ``KMeans`` in the reference (only ``cluster_centers_`` is used) — no file offline: UNPINNED.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch
import torch.nn.functional as F

KERNELS = (10, 3, 3, 3, 3, 2, 2)
STRIDES = (5, 2, 2, 2, 2, 2, 2)
HEADS, HEAD_DIM, HIDDEN = 12, 64, 768


def _t(w, key):
    v = w[key]
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(v))


def feature_extractor_normalize(wave: torch.Tensor) -> torch.Tensor:
    """Wav2Vec2FeatureExtractor.zero_mean_unit_var_norm without attention mask: (x - mean) / sqrt(var + 1e-7), per clip,
    computed in numpy float32 by HF."""
    x = wave.numpy()
    out = np.stack([(r - r.mean()) / np.sqrt(r.var() + 1e-7) for r in x]).astype(np.float32)
    return torch.from_numpy(out)


def num_frames(n: int) -> int:
    for k, s in zip(KERNELS, STRIDES):
        n = (n - k) // s + 1
    return n


def conv_features(w, wave: torch.Tensor) -> torch.Tensor:
    """[B, N] -> [B, T, 512] (modeling_hubert.py:178-213)."""
    h = wave[:, None]
    for i, s in enumerate(STRIDES):
        h = F.conv1d(h, _t(w, f"feature_extractor.conv_layers.{i}.conv.weight"), stride=s)
        if i == 0:
            h = F.group_norm(h, 512, _t(w, "feature_extractor.conv_layers.0.layer_norm.weight"),
                             _t(w, "feature_extractor.conv_layers.0.layer_norm.bias"), 1e-5)
        h = F.gelu(h)
    return h.transpose(1, 2)


def frame_mask(mask: torch.Tensor, T: int) -> torch.Tensor:
    """modeling_hubert.py:679-693: frames [0, out_len(sum(mask))) are valid. -> bool [B, T]."""
    lens = mask.sum(-1)
    for k, s in zip(KERNELS, STRIDES):
        lens = torch.div(lens - k, s, rounding_mode="floor") + 1
    lens = lens.to(torch.long)
    return torch.arange(T)[None, :] < lens[:, None]


def pos_conv(w, h: torch.Tensor) -> torch.Tensor:
    wt = torch._weight_norm(_t(w, "encoder.pos_conv_embed.conv.weight_v"), _t(w, "encoder.pos_conv_embed.conv.weight_g"), 2)
    y = F.conv1d(h.transpose(1, 2), wt, _t(w, "encoder.pos_conv_embed.conv.bias"), padding=64, groups=16)[:, :, :-1]
    return F.gelu(y).transpose(1, 2)


def attention(w, p: str, x: torch.Tensor, add_mask: Optional[torch.Tensor]) -> torch.Tensor:
    B, T, _ = x.shape
    def proj(n):
        return F.linear(x, _t(w, f"{p}.{n}.weight"), _t(w, f"{p}.{n}.bias")).view(B, T, HEADS, HEAD_DIM).transpose(1, 2)
    q, k, v = proj("q_proj"), proj("k_proj"), proj("v_proj")
    a = torch.matmul(q, k.transpose(2, 3)) * (HEAD_DIM ** -0.5)
    if add_mask is not None:
        a = a + add_mask
    a = F.softmax(a, dim=-1)
    o = torch.matmul(a, v).transpose(1, 2).reshape(B, T, HIDDEN)
    return F.linear(o, _t(w, f"{p}.out_proj.weight"), _t(w, f"{p}.out_proj.bias"))


def hidden_states(w, wave: torch.Tensor, mask: Optional[torch.Tensor], n_layers: int = 11, return_all: bool = False):
    feats = conv_features(w, wave)
    T = feats.shape[1]
    h = F.layer_norm(feats, (512,), _t(w, "feature_projection.layer_norm.weight"), _t(w, "feature_projection.layer_norm.bias"), 1e-5)
    h = F.linear(h, _t(w, "feature_projection.projection.weight"), _t(w, "feature_projection.projection.bias"))
    add_mask = None
    if mask is not None:
        fm = frame_mask(mask, T)
        h = h * fm.unsqueeze(-1)
        add_mask = torch.zeros(fm.shape[0], 1, 1, T).masked_fill(~fm[:, None, None, :], torch.finfo(torch.float32).min)
    h = h + pos_conv(w, h)
    h = F.layer_norm(h, (HIDDEN,), _t(w, "encoder.layer_norm.weight"), _t(w, "encoder.layer_norm.bias"), 1e-5)
    states = [h]
    for i in range(n_layers):
        p = f"encoder.layers.{i}"
        h = h + attention(w, p + ".attention", h, add_mask)
        h = F.layer_norm(h, (HIDDEN,), _t(w, p + ".layer_norm.weight"), _t(w, p + ".layer_norm.bias"), 1e-5)
        f = F.linear(h, _t(w, p + ".feed_forward.intermediate_dense.weight"), _t(w, p + ".feed_forward.intermediate_dense.bias"))
        f = F.linear(F.gelu(f), _t(w, p + ".feed_forward.output_dense.weight"), _t(w, p + ".feed_forward.output_dense.bias"))
        h = F.layer_norm(h + f, (HIDDEN,), _t(w, p + ".final_layer_norm.weight"), _t(w, p + ".final_layer_norm.bias"), 1e-5)
        states.append(h)
    return states if return_all else h


def kmeans_assign(e: torch.Tensor, centers: torch.Tensor, return_margin: bool = False):
    """torch.cdist + argmin(keepdim) (reference encoder.py:100-101)."""
    d = torch.cdist(e, centers)
    idx = d.argmin(dim=-1)
    if return_margin:
        top2 = (-d).topk(2, dim=-1).values
        return idx, top2[..., 0] - top2[..., 1]
    return idx


def semantic_s_encode(w, wave: torch.Tensor, mask: torch.Tensor, n_layers: int = 11, return_margins: bool = False):
    """Reference HubertEncoder.__call__ (audiotoken/encoder.py:87-108): int16 [B, 1, T]. `wave` is already normalised
    by hubert_processor (the reference applies it as transform_func before batching)."""
    h = hidden_states(w, wave, mask, n_layers)
    e = F.layer_norm(h, (HIDDEN,))
    if return_margins:   # also the oracle's own top-2 distance margin per token, [B, 1, T]
        idx, margin = kmeans_assign(e, _t(w, "kmeans.cluster_centers_"), return_margin=True)
        return idx.unsqueeze(-1).transpose(1, 2).to(torch.int16), margin.unsqueeze(-1).transpose(1, 2)
    idx = kmeans_assign(e, _t(w, "kmeans.cluster_centers_"))
    return idx.unsqueeze(-1).transpose(1, 2).to(torch.int16)
