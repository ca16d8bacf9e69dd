"""semantic_s: ``HubertEncoder`` and ``hubert_processor`` with the reference's protocol
(audiotoken/encoder.py:20-26, 60-108), backed by libaudiotoken_hip.so."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Union

import numpy as np
import torch

from . import _cabi
from . import weights as W
from .configs import HubertEncoderConfig
from .encoder import _device_index
from .logger import get_logger

logger = get_logger(__name__)


def hubert_processor(audio: torch.Tensor, processor=None) -> torch.Tensor:
    """Reference ``hubert_processor`` (encoder.py:20-26) = HF ``Wav2Vec2FeatureExtractor(do_normalize=True)`` on one
    clip: zero mean / unit variance over the whole array, ``(x - mean) / sqrt(var + 1e-7)`` in float32. Host-side
    transform applied before batching, exactly where the reference applies it (core.py:188-189, datasets.py:78-79)."""
    x = np.asarray(audio, dtype=np.float32)
    y = (x - x.mean()) / np.sqrt(x.var() + 1e-7)
    return torch.from_numpy(y.astype(np.float32))


def fold_hubert_weights(w: Dict[str, np.ndarray], n_layers: int) -> Dict[str, np.ndarray]:
    out: Dict[str, np.ndarray] = {}
    for k, v in w.items():
        if k.startswith("encoder.layers.") and int(k.split(".")[2]) >= n_layers:
            continue   # layers past the consumed hidden state are dead compute
        if k.endswith("pos_conv_embed.conv.weight_g") or k.endswith("parametrizations.weight.original0"):
            base = k.rsplit(".", 1)[0] if k.endswith("weight_g") else k[: -len(".parametrizations.weight.original0")]
            vkey = base + ".weight_v" if k.endswith("weight_g") else base + ".parametrizations.weight.original1"
            folded = torch._weight_norm(torch.from_numpy(np.ascontiguousarray(w[vkey], dtype=np.float32)),
                                        torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)), 2)
            out["encoder.pos_conv_embed.conv.weight"] = folded.numpy()
        elif k.endswith("weight_v") or k.endswith("original1") or k == "masked_spec_embed":
            continue
        elif k == "kmeans.cluster_centers_":
            c = np.ascontiguousarray(v, dtype=np.float32)
            out[k] = c
            out["kmeans.c2"] = (torch.from_numpy(c) ** 2).sum(-1).numpy()
        else:
            out[k] = np.ascontiguousarray(v, dtype=np.float32)
    return out


HUBERT_ARCH = {"hidden_size": 768, "intermediate_size": 3072, "num_attention_heads": 12, "conv_dim": [512] * 7, "conv_kernel": [10, 3, 3, 3, 3, 2, 2],
               "conv_stride": [5, 2, 2, 2, 2, 2, 2], "conv_bias": False, "feat_extract_norm": "group", "do_stable_layer_norm": False,
               "num_conv_pos_embeddings": 128, "num_conv_pos_embedding_groups": 16, "hidden_act": "gelu"}


def load_hubert_checkpoint(model_dir: str, quantizer_path: Optional[str]) -> Dict[str, np.ndarray]:
    """The reference's ``HubertModel.from_pretrained`` directory (config.json + model.safetensors / pytorch_model.bin, possibly sharded,
    keys possibly under ``hubert.``) + the joblib k-means pickle whose ``cluster_centers_`` it reads (reference encoder.py:72, 84-85)."""
    sd = W.read_hf_state_dict(model_dir, strip_prefixes=("hubert.",))
    W.check_hf_config(model_dir, HUBERT_ARCH, "semantic_s checkpoint")
    if quantizer_path:
        import joblib
        sd["kmeans.cluster_centers_"] = np.asarray(joblib.load(quantizer_path).cluster_centers_, dtype=np.float32)
    return sd


class HubertEncoder(torch.nn.Module):
    """Drop-in for reference ``HubertEncoder`` (audiotoken/encoder.py:60-108)."""

    def __init__(self, config: HubertEncoderConfig = None, device: str = "cuda:0", quantize: bool = True,
                 weights: Optional[Union[str, Dict[str, np.ndarray]]] = None, packed=None):
        """``packed`` = ``(meta, blob)`` from another rank's ``export_packed()``: the finalized model is rebuilt over that device blob (``weights`` is ignored)."""
        super().__init__()
        config = config or HubertEncoderConfig()
        self.config = config
        self.quantize = quantize
        self.output_layer = config.output_layer
        self.lib = _cabi.load()
        self.device_index = _device_index(device)
        self.device = torch.device("cuda", self.device_index)
        if packed is not None:
            self.handle = self.lib.at_hubert_create(self.device_index)
            if not self.handle:
                raise _cabi.HipLibraryError(f"at_hubert_create failed: {_cabi.last_error()}")
            _cabi.import_packed(self.lib, "hubert", self.handle, packed[0], packed[1].to(self.device))
            self._finish_init()
            return
        if weights is None:
            weights = config.weights
        if weights is None:
            logger.warning("No HuBERT checkpoint given (weights=/AUDIOTOKEN_HUBERT_WEIGHTS): synthetic weights, seed 0")
            weights = W.synth_hubert_weights(n_layers=self.output_layer, seed=0, with_kmeans=True)
        elif isinstance(weights, (str, bytes)):
            weights = load_hubert_checkpoint(weights, config.quantizer_path)
        self.handle = self.lib.at_hubert_create(self.device_index)
        if not self.handle:
            raise _cabi.HipLibraryError(f"at_hubert_create failed: {_cabi.last_error()}")
        for name, arr in fold_hubert_weights(weights, self.output_layer).items():
            _cabi.set_tensor(self.lib, self.lib.at_hubert_set_tensor, self.handle, name, arr)
        _cabi.check(self.lib.at_hubert_finalize(self.handle), "at_hubert_finalize")
        self._finish_init()

    def export_packed(self):
        """(meta bytes, uint8 device blob): this finalized model for ``HubertEncoder(packed=...)`` on the other ranks of a node."""
        return _cabi.export_packed(self.lib, "hubert", self.handle, self.device)

    def _finish_init(self) -> None:
        if self.lib.at_hubert_num_layers(self.handle) < self.output_layer:
            raise ValueError(f"checkpoint has too few transformer layers for output_layer={self.output_layer}")
        self._ws: Optional[torch.Tensor] = None
        self._status = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.fallback_batches = 0
        self.nonfinite_batches = 0
        self.pinned_layers = []      # transformer layers `verified` moved to bf16x3 for good (their activations do not fit the fp16 range)
        self.layer_overflows = {}    # {layer: batches on which it overflowed}: a layer is pinned from the PIN_AFTER-th such batch on (Wav2VecBertEncoder.PIN_AFTER)

    def __del__(self):
        h = self.__dict__.pop("handle", None)
        if h:
            self.lib.at_hubert_destroy(h)

    ARITH = {"f32": 0, "bf16x3": 1, "f16x2": 2}

    def set_option(self, name: str, value) -> None:
        """"arith": "f32" | "bf16x3" | "f16x2" (or 0/1/2) — arithmetic of the linear layers and the 512->512 convs."""
        if isinstance(value, str):
            value = self.ARITH[value]
        _cabi.check(self.lib.at_hubert_set_option(self.handle, name.encode(), int(value)), f"at_hubert_set_option({name})")

    def get_option(self, name: str) -> int:
        return int(self.lib.at_hubert_get_option(self.handle, name.encode()))

    def last_status(self) -> int:
        """0 = ok; bit 1 (2) = an activation overflowed the fp16 range of the f16x2 arithmetic (synchronises the device)."""
        return int(self._status.item())

    def range_report(self) -> Dict[str, float]:
        """{site: largest |x * scale| its split writers saw in the LAST call}; the f16x2 arithmetic overflows at 65504."""
        return _cabi.range_report(self.lib, "hubert", self.handle)

    def site_scales(self):
        """[(scale of the q/k/v projection's input, scale of the first FFN GEMM's input) per transformer layer]: 16, or the provable scale of a LayerNorm
        whose gains are too large for 16 (include/audiotoken_hip.h, at_hubert_site_scales)."""
        buf = (C.c_float * 128)()
        n = self.lib.at_hubert_site_scales(self.handle, buf, 128)
        if n < 0:
            raise _cabi.HipLibraryError(f"at_hubert_site_scales failed: {_cabi.last_error()}")
        return [(float(buf[2 * l]), float(buf[2 * l + 1])) for l in range(n // 2)]

    def layer_status(self):
        """Status flags of the LAST call per part: [0] = conv feature encoder + positional conv, [1 + l] = transformer layer l (bit 1 = fp16 overflow)."""
        import ctypes as C
        buf = (C.c_int32 * 64)()
        n = self.lib.at_hubert_layer_status(self.handle, buf, 64)
        if n < 0:
            raise _cabi.HipLibraryError(f"at_hubert_layer_status failed: {_cabi.last_error()}")
        return [int(buf[i]) for i in range(n)]

    PIN_AFTER = 2   # as Wav2VecBertEncoder.PIN_AFTER

    def unpin_layers(self) -> None:
        for layer in set(self.pinned_layers):
            self.set_option(f"layer_arith:{layer}", -1)
        self.pinned_layers = []
        self.layer_overflows = {}

    def verified(self, tokens: torch.Tensor, input_batch: torch.Tensor, attention_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Product-path guard (see Wav2VecBertEncoder.verified): on an fp16 range overflow in a TRANSFORMER layer that layer moves to bf16x3 and the batch is
        repeated (the layer returns to f16x2 afterwards unless it was its PIN_AFTER-th overflowing batch); an overflow in the conv feature encoder / positional conv (a property of the input's level) repeats THIS batch with
        arith=bf16x3 for the whole model, then switches back."""
        status = self.last_status()
        if status == 0:
            return tokens
        if status & 4 and not status & 2:   # (with bit 1 set the infinity descends from the flagged fp16 overflow: the repeat below cures it)
            # a NaN / infinity reached the quantiser (a non-finite sample in the waveform, as a rule): no kernel choice changes that. The reference emits
            # arbitrary ids for such input without a diagnostic; here it is at least logged and counted. The ids are returned as they are.
            self.nonfinite_batches += 1
            logger.error(f"semantic_s encode: a NaN or an infinity reached the quantiser (status {status}); check the input waveform. "
                         f"The token ids of this batch are meaningless (non-finite batch #{self.nonfinite_batches})")
            if status & ~4 == 0:
                return tokens
        self.fallback_batches += 1
        transient = []    # layers moved for THIS batch only (their first overflow): restored below
        try:
            for _ in range(3):
                flags = self.layer_status()
                bad = [i for i, f in enumerate(flags) if f & 2]
                if not bad or bad[0] == 0:      # nothing per layer to act on, or the front end itself: the whole-batch repeat below
                    break
                layer = bad[0] - 1
                self.layer_overflows[layer] = self.layer_overflows.get(layer, 0) + 1
                pin = self.layer_overflows[layer] >= self.PIN_AFTER
                (self.pinned_layers if pin else transient).append(layer)
                logger.error(f"semantic_s encode reported status {status}: an activation of transformer layer {layer} exceeded the fp16 range of the f16x2 arithmetic "
                             f"(batch #{self.layer_overflows[layer]} on which it did). The tokens of this batch were discarded; layer {layer} runs on bf16x3 (option "
                             f"layer_arith:{layer} = 1) " + ("from now on" if pin else "for this batch") + f", this batch is re-encoded (fallback batch #{self.fallback_batches})")
                self.set_option(f"layer_arith:{layer}", 1)
                tokens = self.forward(input_batch, attention_mask)
                status = self.last_status()
                if not status & 2:
                    if status & 4:
                        self.nonfinite_batches += 1
                        logger.error(f"a NaN or an infinity reached the quantiser with layer {layer} on bf16x3 too (non-finite batch #{self.nonfinite_batches}): check the input waveform")
                    return tokens
        finally:
            for layer in transient:
                self.set_option(f"layer_arith:{layer}", -1)
        logger.error(f"semantic_s encode reported status {status} (an activation exceeded the fp16 range of the f16x2 arithmetic): "
                     f"the tokens of this batch were discarded; re-encoding THIS batch with arith=bf16x3 (fallback batch #{self.fallback_batches})")
        saved = self.get_option("arith")
        self.set_option("arith", "bf16x3")
        try:
            tokens = self.forward(input_batch, attention_mask)
            if self.last_status() & 4:          # still non-finite on the safe kernels: it came with the input, not from the fp16 range
                self.nonfinite_batches += 1
                logger.error(f"a NaN or an infinity reached the quantiser on the fallback kernels too (non-finite batch #{self.nonfinite_batches}): check the input waveform")
            if self.last_status() & ~4 != 0:   # (bit 2, non-finite input, is not something a repeat can clear)
                raise _cabi.HipLibraryError("semantic_s encode failed twice (status non-zero with bf16x3 arithmetic)")
        finally:
            self.set_option("arith", saved)
        return tokens

    @torch.no_grad()
    def forward(self, input_batch: torch.Tensor, attention_mask: Optional[torch.Tensor] = None, n_layers: Optional[int] = None,
                return_hidden: bool = False):
        """``float32 [B, N]`` (normalised) + mask -> ``int16 [B, 1, T]`` on the device (hidden state if quantize=False)."""
        assert input_batch.dim() == 2, "input_batch must be [B, N]"
        x = input_batch.to(device=self.device, dtype=torch.float32).contiguous()
        m = None if attention_mask is None else attention_mask.to(device=self.device, dtype=torch.float32).contiguous()
        B, N = x.shape
        T = self.lib.at_hubert_num_tokens(N)
        nl = self.output_layer if n_layers is None else n_layers
        tokens = torch.empty((B, 1, T), dtype=torch.int16, device=self.device) if self.quantize else None
        hidden = torch.empty((B, T, 768), dtype=torch.float32, device=self.device) if (return_hidden or not self.quantize) else None
        nbytes = self.lib.at_hubert_workspace_bytes(self.handle, B, N)
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = None
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        t_out = C.c_int(0)
        with torch.cuda.device(self.device):
            rc = self.lib.at_hubert_encode_checked(self.handle, x.data_ptr(), _cabi.ptr(m), B, N, nl, _cabi.ptr(tokens), C.byref(t_out),
                                                   _cabi.ptr(hidden), self._ws.data_ptr(), nbytes, _cabi.current_stream_handle(self.device),
                                                   self._status.data_ptr())
        _cabi.check(rc, "at_hubert_encode_checked")
        assert t_out.value == T
        if return_hidden:
            return tokens, hidden
        return tokens if self.quantize else hidden

    __call__ = torch.nn.Module.__call__   # the reference defines __call__ directly (encoder.py:87); same call protocol

    def enable_profile(self, on: bool) -> None:
        _cabi.check(self.lib.at_hubert_profile(self.handle, 1 if on else 0), "at_hubert_profile")

    def read_profile(self) -> Dict[str, tuple]:
        names = C.create_string_buffer(4096)
        ms = (C.c_float * 64)()
        ln = (C.c_int * 64)()
        n = self.lib.at_hubert_profile_read(self.handle, names, 4096, ms, ln, 64)
        if n < 0:
            raise _cabi.HipLibraryError(f"at_hubert_profile_read failed: {_cabi.last_error()}")
        keys = names.value.decode().split("\n")[:n]
        return {k: (float(ms[i]), int(ln[i])) for i, k in enumerate(keys)}
