"""CPU oracle for the semantic_m tokenizer: log-mel front-end -> Wav2Vec2-BERT conformer (first 19 layers)
-> non-affine LayerNorm -> L2-nearest code of a 2048 x 1024 codebook.

TEST INFRASTRUCTURE — never imported by the product path (see oracle/__init__.py).

Restates, in plain torch-CPU fp32:
* ``Wav2VecBertProcessor`` — reference ``audiotoken/processors.py:8-266`` with mel helpers
  ``audiotoken/utils.py:286-328`` (reference-authored; pinned by tests/golden/fbank_*.npz which were produced by
  importing the reference's own files, see tests/golden/make_golden.py).
* the rel-pos SDPA attention the reference patches into HF — ``audiotoken/modeling_wav2vec2_bert.py:20-80``
  (pinned by tests/golden/attention_*.npz, same provenance).
* the rest of the conformer layer / encoder — dependency ``transformers`` (unpinned in ``requirements.txt:4``;
  restated from the published model code, HF 5.15.0 ``models/wav2vec2_bert/modeling_wav2vec2_bert.py``:
  feature projection ``:119-131``, feed-forward ``:134-154``, conv module ``:157-226``, layer ``:398-461``,
  encoder entry (zero padded rows, additive mask) ``:491-531``); pinned by tests/golden/conformer_*.npz
  (HF ``Wav2Vec2BertModel`` + the reference patch, synthetic weights).
* ``vector_quantize_pytorch.VectorQuantize`` eval forward (dependency, unpinned ``requirements.txt:10``, not
  installed): identity projections, fp32, ``idx = argmax(-cdist(x, embed))`` with the package's own
  ``cdist = sqrt(clamp(x2 + y2 - 2xy, 0))``; call site ``audiotoken/encoder.py:147-161,180``. PARITY UNPINNED for the
  package itself (absent); the formula is cross-checked against ``torch.cdist`` + ``argmin`` in the tests.
Only hidden state 19 is consumed (``audiotoken/encoder.py:172-175``, ``configs.py:128``) so layers 19-20 of the 21
are never evaluated here.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

MEL_FLOOR = 1.192092955078125e-07
FRAME, HOP, NFFT, NMEL = 400, 160, 512, 80
LEFT_MAX, RIGHT_MAX = 64, 8
HEADS, HEAD_DIM, HIDDEN = 16, 64, 1024


def _t(w, key: str) -> torch.Tensor:
    v = w[key]
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(v))


# --------------------------------------------------------------------------------------------------
# front-end (processors.py)
# --------------------------------------------------------------------------------------------------
def hertz_to_mel(f: torch.Tensor) -> torch.Tensor:
    """utils.py:286-296 (Kaldi mel)."""
    return 1127.0 * torch.log(1.0 + (f / 700.0))


def mel_filter_bank() -> torch.Tensor:
    """processors.py:8-26 + utils.py:313-328: triangles built IN MEL SPACE (the Hz conversion on :19 is
    overwritten on :21) over 256 bins of width 16000/512 from 20 Hz to 8000 Hz; one zero row appended (:77).
    Returns [257, 80]."""
    mel_min = hertz_to_mel(torch.tensor(20.0))
    mel_max = hertz_to_mel(torch.tensor(8000.0))
    filter_freqs = torch.linspace(mel_min, mel_max, NMEL + 2)
    fft_bin_width = 16000 / (256 * 2)
    fft_freqs = hertz_to_mel(fft_bin_width * torch.arange(256))
    filter_diff = torch.diff(filter_freqs)
    slopes = filter_freqs.unsqueeze(0) - fft_freqs.unsqueeze(1)
    down = -slopes[:, :-2] / filter_diff[:-1]
    up = slopes[:, 2:] / filter_diff[1:]
    fb = torch.maximum(torch.zeros(1), torch.minimum(down, up))
    return F.pad(fb, (0, 0, 0, 1))


def povey_window() -> torch.Tensor:
    """processors.py:75: hann(400, periodic=False) ** 0.85."""
    return torch.pow(torch.hann_window(FRAME, periodic=False), 0.85)


def num_frames(n_samples: int) -> int:
    """processors.py:158."""
    return int(1 + math.floor((n_samples - FRAME) / HOP))


def log_mel(wave: torch.Tensor) -> torch.Tensor:
    """processors.py:137-190, frame loop vectorised (same per-frame arithmetic, same operation order).
    wave [B, N] -> [B, F, 80]."""
    x = wave * (2 ** 15)
    nf = num_frames(x.shape[1])
    frames = x.unfold(1, FRAME, HOP)[:, :nf].clone()           # [B, F, 400]
    frames = frames - frames.mean(dim=2, keepdim=True)          # :168-169
    prev = frames[..., :-1].clone()
    frames[..., 1:] = frames[..., 1:] - 0.97 * prev             # :171-172 (pre-update neighbours)
    frames[..., 0] = frames[..., 0] * (1 - 0.97)                # :173
    # (.to(dtype) is the identity for the reference's float32; a float64 `wave` evaluates the same formulas with an exact front-end — used by
    # tests/test_edge_inputs_gpu.py to measure how much of the reference's output on an ill-conditioned input is its own rounding noise)
    frames = frames * povey_window().to(frames.dtype)           # :175
    buf = F.pad(frames, (0, NFFT - FRAME))
    spec = torch.fft.rfft(buf)                                  # :177
    power = spec.abs().pow(2.0)                                 # :181
    mel = torch.matmul(power, mel_filter_bank().to(power.dtype))  # :184
    mel = torch.maximum(mel, torch.tensor(MEL_FLOOR, dtype=mel.dtype))
    return torch.log(mel)


def frame_mask(mask: torch.Tensor, nf: int) -> torch.Tensor:
    """processors.py:80-115: a frame is valid iff all of its 400 samples are valid. [B, N] -> [B, F]."""
    m = F.avg_pool1d(mask.unsqueeze(1), kernel_size=FRAME, stride=HOP, padding=0).squeeze(1)[:, :nf]
    return torch.where(m == 1, m, 0)


def processor(wave: torch.Tensor, mask: torch.Tensor, pad_to_multiple_of: int = 2) -> Tuple[torch.Tensor, torch.Tensor]:
    """Wav2VecBertProcessor.forward (processors.py:209-266): returns (input_features [B,T',160], attention_mask [B,T'])."""
    feats = log_mel(wave)
    nf = feats.shape[1]
    fm = frame_mask(mask, nf).unsqueeze(-1).expand(-1, -1, NMEL)
    masked = feats * fm                                          # :128
    cnt = fm.sum(dim=1, keepdim=True).clamp(min=1)
    mean = masked.sum(dim=1, keepdim=True) / cnt
    var = (((masked - mean) ** 2) * fm).sum(dim=1, keepdim=True) / cnt   # population variance (:133)
    feats = (feats - mean) / torch.sqrt(var + 1e-7)             # :242
    rem = nf % 2
    if rem:
        feats, fm = feats[:, : nf - rem], fm[:, : nf - rem]
    B = feats.shape[0]
    feats = feats.reshape(B, (nf - rem) // 2, 2 * NMEL)
    fm = fm.reshape(B, (nf - rem) // 2, 2 * NMEL)
    n = feats.shape[1]
    P = 0
    if pad_to_multiple_of > 0 and n % pad_to_multiple_of:
        P = pad_to_multiple_of - n % pad_to_multiple_of
    out = torch.where(fm == 0, 1.0, feats)                      # :200 (padding_value = 1)
    out = F.pad(out, (0, 0, 0, P), value=1.0)
    am = F.pad(fm[:, :, 0], (0, P), value=0)
    return out, torch.where(am == 1, am, 0)


# --------------------------------------------------------------------------------------------------
# conformer (HF Wav2Vec2BertModel + the reference's attention patch)
# --------------------------------------------------------------------------------------------------
def layer_norm(x, w, prefix: Optional[str], dim: int):
    if prefix is None:
        return F.layer_norm(x, (dim,), None, None, 1e-5)
    return F.layer_norm(x, (dim,), _t(w, prefix + ".weight"), _t(w, prefix + ".bias"), 1e-5)


def feed_forward(w, prefix: str, x):
    h = F.linear(x, _t(w, prefix + ".intermediate_dense.weight"), _t(w, prefix + ".intermediate_dense.bias"))
    h = F.silu(h)
    return F.linear(h, _t(w, prefix + ".output_dense.weight"), _t(w, prefix + ".output_dense.bias"))


def relpos_attention(w, prefix: str, x: torch.Tensor, add_mask: Optional[torch.Tensor]) -> torch.Tensor:
    """audiotoken/modeling_wav2vec2_bert.py:20-80. x [B,T,1024]; add_mask [B,1,T,T] additive (finfo.min at padded keys)."""
    B, T, _ = x.shape
    q = F.linear(x, _t(w, prefix + ".linear_q.weight"), _t(w, prefix + ".linear_q.bias")).view(B, T, HEADS, HEAD_DIM).transpose(1, 2)
    k = F.linear(x, _t(w, prefix + ".linear_k.weight"), _t(w, prefix + ".linear_k.bias")).view(B, T, HEADS, HEAD_DIM).transpose(1, 2)
    v = F.linear(x, _t(w, prefix + ".linear_v.weight"), _t(w, prefix + ".linear_v.bias")).view(B, T, HEADS, HEAD_DIM).transpose(1, 2)
    pos_l = torch.arange(T).view(-1, 1)
    pos_r = torch.arange(T).view(1, -1)
    dist = torch.clamp(pos_r - pos_l, -LEFT_MAX, RIGHT_MAX)
    pe = F.embedding(dist + LEFT_MAX, _t(w, prefix + ".distance_embedding.weight"))      # [T,T,64]
    bias = torch.einsum("bhld,lrd->bhlr", q, pe) / math.sqrt(HEAD_DIM)                    # :57-58
    if add_mask is not None:
        bias = bias + add_mask                                                           # :61-64
    ctx = F.scaled_dot_product_attention(q, k, v, attn_mask=bias, scale=1 / math.sqrt(HEAD_DIM))
    ctx = ctx.transpose(1, 2).reshape(B, T, HEADS * HEAD_DIM)
    return F.linear(ctx, _t(w, prefix + ".linear_out.weight"), _t(w, prefix + ".linear_out.bias"))


def conv_module(w, prefix: str, x: torch.Tensor, mask: Optional[torch.Tensor]) -> torch.Tensor:
    """HF modeling_wav2vec2_bert.py:157-226. x [B,T,1024], mask [B,T] (1 = valid)."""
    h = layer_norm(x, w, prefix + ".layer_norm", HIDDEN)
    if mask is not None:
        h = h.masked_fill(~mask.bool().unsqueeze(-1), 0.0)
    h = h.transpose(1, 2)
    h = F.conv1d(h, _t(w, prefix + ".pointwise_conv1.weight"))
    h = F.glu(h, dim=1)
    h = F.pad(h, (30, 0))
    h = F.conv1d(h, _t(w, prefix + ".depthwise_conv.weight"), groups=HIDDEN)
    h = layer_norm(h.transpose(1, 2), w, prefix + ".depthwise_layer_norm", HIDDEN).transpose(1, 2)
    h = F.silu(h)
    h = F.conv1d(h, _t(w, prefix + ".pointwise_conv2.weight"))
    return h.transpose(1, 2)


def conformer_layer(w, i: int, x, add_mask, mask):
    """HF modeling_wav2vec2_bert.py:423-461."""
    p = f"encoder.layers.{i}"
    x = x + 0.5 * feed_forward(w, p + ".ffn1", layer_norm(x, w, p + ".ffn1_layer_norm", HIDDEN))
    # HF writes `hidden*0.5 + residual`; addition is commutative in IEEE so the value is identical
    x = x + relpos_attention(w, p + ".self_attn", layer_norm(x, w, p + ".self_attn_layer_norm", HIDDEN), add_mask)
    x = x + conv_module(w, p + ".conv_module", x, mask)
    x = x + 0.5 * feed_forward(w, p + ".ffn2", layer_norm(x, w, p + ".ffn2_layer_norm", HIDDEN))
    return layer_norm(x, w, p + ".final_layer_norm", HIDDEN)


def encoder_hidden_state(w, feats: torch.Tensor, mask: Optional[torch.Tensor], n_layers: int = 19, return_all: bool = False):
    """hidden_states[n_layers] of Wav2Vec2BertModel(input_features, attention_mask, output_hidden_states=True)."""
    h = layer_norm(feats, w, "feature_projection.layer_norm", feats.shape[-1])
    h = F.linear(h, _t(w, "feature_projection.projection.weight"), _t(w, "feature_projection.projection.bias"))
    add_mask = None
    if mask is not None:
        h = h.masked_fill(~mask.bool().unsqueeze(-1), 0.0)
        am = (1.0 - mask[:, None, None, :].to(h.dtype)) * torch.finfo(h.dtype).min
        add_mask = am.expand(am.shape[0], 1, am.shape[-1], am.shape[-1])
    states = [h]
    for i in range(n_layers):
        h = conformer_layer(w, i, h, add_mask, mask)
        states.append(h)
    return states if return_all else h


def vq_cdist(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """vector_quantize_pytorch's own cdist: sqrt(clamp(x2 + y2 - 2 x.y, min=0)). x [n,d], y [c,d]."""
    x2 = (x ** 2).sum(-1)
    y2 = (y ** 2).sum(-1)
    xy = (x @ y.t()) * -2
    return (x2.unsqueeze(1) + y2.unsqueeze(0) + xy).clamp(min=0).sqrt()


def vq_assign(x: torch.Tensor, embed: torch.Tensor, return_margin: bool = False):
    """VectorQuantize eval forward -> indices: argmax(-cdist) (first maximal index). x [..., 1024], embed [2048, 1024]."""
    shape = x.shape[:-1]
    d = -vq_cdist(x.reshape(-1, x.shape[-1]).float(), embed.float())
    idx = d.argmax(dim=-1).reshape(shape)
    if return_margin:
        top2 = d.topk(2, dim=-1).values
        return idx, (top2[:, 0] - top2[:, 1]).reshape(shape)
    return idx


def semantic_m_encode(w, wave: torch.Tensor, mask: torch.Tensor, pad_to_multiple_of: int = 2, n_layers: int = 19, return_margins: bool = False):
    """Reference Wav2VecBertEncoder.forward with quantize=True (audiotoken/encoder.py:163-184): int16 [B, 1, T'].
    With return_margins also the oracle's own top-2 distance margin per token, [B, 1, T']."""
    feats, am = processor(wave, mask, pad_to_multiple_of)
    h = encoder_hidden_state(w, feats, am, n_layers)
    e = layer_norm(h, w, None, HIDDEN)
    embed = _t(w, "vq._codebook.embed")
    if return_margins:
        clusters, margin = vq_assign(e, embed.reshape(-1, embed.shape[-1]), return_margin=True)
        return clusters.unsqueeze(-1).transpose(1, 2).to(torch.int16), margin.unsqueeze(-1).transpose(1, 2)
    clusters = vq_assign(e, embed.reshape(-1, embed.shape[-1]))
    return clusters.unsqueeze(-1).transpose(1, 2).to(torch.int16)
