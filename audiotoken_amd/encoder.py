"""Encoder callables with the reference's protocol: ``encoder(input_batch, attention_mask) -> int16 [B, K, T]``
(reference audiotoken/encoder.py:29-186), backed by libaudiotoken_hip.so through the C ABI.

PyTorch is used for device tensors, the caching allocator (workspace) and the current stream only.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Dict, Optional, Union

import numpy as np
import torch

from . import _cabi
from . import weights as W
from .configs import AcousticEncoderConfig
from .logger import get_logger

logger = get_logger(__name__)


def _device_index(device: Union[str, torch.device]) -> int:
    dev = torch.device(device)
    if dev.type != "cuda":
        raise ValueError(
            f"audiotoken_amd runs on MI355X only (device 'cuda[:i]' under PyTorch-ROCm); got {device!r}. "
            "There is no CPU path in this package.")
    return dev.index if dev.index is not None else torch.cuda.current_device()


def encodec_bandwidth_to_nq(bandwidth: float) -> int:
    """``n_q = max(1, floor(bw*1000 / (log2(1024)*75)))`` — encodec ResidualVectorQuantizer
    (call site reference audiotoken/encoder.py:50-52)."""
    return int(max(1, math.floor(bandwidth * 1000 / (10 * 75))))


def fold_encodec_weights(w: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """encodec checkpoint dict (weight_g / weight_v pairs) -> folded ``.weight`` tensors for the C ABI."""
    out: Dict[str, np.ndarray] = {}
    for k, v in w.items():
        if k.endswith(".weight_g"):
            base = k[: -len(".weight_g")]
            out[base + ".weight"] = W.fold_weight_norm(v, w[base + ".weight_v"])
        elif k.endswith(".weight_v"):
            continue
        elif k.endswith("._codebook.embed"):
            out[k] = np.ascontiguousarray(v, dtype=np.float32)
            e = torch.from_numpy(out[k])
            # |e|^2 exactly as the reference forms it: embed.t().pow(2).sum(0, keepdim=True)
            out[k[: -len("embed")] + "e2"] = e.t().pow(2).sum(0).numpy()
        elif k.startswith(("encoder.", "decoder.", "quantizer.")) and not k.endswith(("inited", "cluster_size", "embed_avg")):
            out[k] = np.ascontiguousarray(v, dtype=np.float32)
    return out


def load_encodec_checkpoint(path: str) -> Dict[str, np.ndarray]:
    """Read an ``encodec_24khz-*.th`` state dict (torch.save format) into numpy arrays."""
    sd = torch.load(path, map_location="cpu", weights_only=True)
    return {k: v.float().numpy() for k, v in sd.items() if torch.is_tensor(v)}


def encodec_weights_from(weights, with_decoder: bool) -> Dict[str, np.ndarray]:
    """``weights`` as the constructors accept it — None (synthetic, seed 0, logged), a checkpoint path, or a name -> array dict — as the un-folded dict."""
    if weights is None:
        logger.warning("No EnCodec checkpoint given (weights=/AUDIOTOKEN_ENCODEC_WEIGHTS): using synthetic weights, seed 0")
        return W.synth_encodec_weights(seed=0, with_decoder=with_decoder)
    if isinstance(weights, (str, bytes, os.PathLike)):
        return load_encodec_checkpoint(weights)
    return weights


class _EncodecHandle:
    """Owns one ``at_encodec_t`` (device weights live inside the library)."""

    def __init__(self, device: Union[str, torch.device], weights: Optional[Union[str, Dict[str, np.ndarray]]],
                 with_decoder: bool):
        self.lib = _cabi.load()
        self.device_index = _device_index(device)
        self.device = torch.device("cuda", self.device_index)
        folded = fold_encodec_weights(encodec_weights_from(weights, with_decoder))
        self.handle = self.lib.at_encodec_create(self.device_index)
        if not self.handle:
            raise _cabi.HipLibraryError(f"at_encodec_create failed: {_cabi.last_error()}")
        for name, arr in folded.items():
            if not with_decoder and name.startswith("decoder."):
                continue
            _cabi.set_tensor(self.lib, self.lib.at_encodec_set_tensor, self.handle, name, arr)
        _cabi.check(self.lib.at_encodec_finalize(self.handle, 1 if with_decoder else 0), "at_encodec_finalize")
        self.n_codebooks = self.lib.at_encodec_num_codebooks(self.handle)

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h:
            self.lib.at_encodec_destroy(h)

    # benchmark taps: HIP events recorded by the library on the launch stream, {group: (total ms, launches)}
    def enable_profile(self, on: bool) -> None:
        _cabi.check(self.lib.at_encodec_profile(self.handle, 1 if on else 0), "at_encodec_profile")

    def read_profile(self) -> Dict[str, tuple]:
        names = C.create_string_buffer(4096)
        ms = (C.c_float * 64)()
        ln = (C.c_int * 64)()
        n = self.lib.at_encodec_profile_read(self.handle, names, 4096, ms, ln, 64)
        if n < 0:
            raise _cabi.HipLibraryError(f"at_encodec_profile_read failed: {_cabi.last_error()}")
        keys = names.value.decode().split("\n")[:n]
        return {k: (float(ms[i]), int(ln[i])) for i, k in enumerate(keys)}


class AcousticEncoder(torch.nn.Module):
    """Drop-in for reference ``AcousticEncoder`` (audiotoken/encoder.py:29-57)."""

    def __init__(self, config: AcousticEncoderConfig = None, device: str = "cuda:0",
                 weights: Optional[Union[str, Dict[str, np.ndarray]]] = None):
        super().__init__()
        config = config or AcousticEncoderConfig()
        self.config = config
        self._h = _EncodecHandle(device, weights if weights is not None else config.weights, with_decoder=False)
        self.device = self._h.device
        self.n_q = encodec_bandwidth_to_nq(config.bandwidth)
        if self.n_q > self._h.n_codebooks:
            raise ValueError(f"bandwidth {config.bandwidth} needs {self.n_q} codebooks, checkpoint has {self._h.n_codebooks}")
        self._ws: Optional[torch.Tensor] = None
        self._status = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.fallback_batches = 0    # batches `verified` repeated on the bf16x3 kernels (fp16 range overflow)
        self.nonfinite_batches = 0   # batches whose activations held a NaN / infinity at the quantiser (status bit 2)

    def _workspace(self, nbytes: int) -> torch.Tensor:
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = None
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        return self._ws

    def set_option(self, name: str, value: int) -> None:
        _cabi.check(self._h.lib.at_encodec_set_option(self._h.handle, name.encode(), int(value)), f"at_encodec_set_option({name})")

    def get_option(self, name: str) -> int:
        return int(self._h.lib.at_encodec_get_option(self._h.handle, name.encode()))

    def _sized_workspace(self, B: int, N: int):
        """Workspace for a [B, N] encode. The conv stack runs in sub-batches of `subbatch` clips (default 256, which needs
        ~0.22 GB per clip-10-s); when the allocation does not fit, the sub-batch is halved (option "subbatch") and the
        request repeated — results do not depend on it (clips are independent), only the workspace size does."""
        lib = self._h.lib
        sub = getattr(self, "_subbatch", None)
        while True:
            nbytes = lib.at_encodec_workspace_bytes(self._h.handle, B, N)
            try:
                return nbytes, self._workspace(nbytes)
            except torch.OutOfMemoryError:
                sub = max(1, (sub or min(B, 256)) // 2)
                if sub < 1 or getattr(self, "_subbatch", None) == 1:
                    raise
                logger.warning(f"workspace of {nbytes / 2**30:.1f} GiB does not fit: conv-stack sub-batch -> {sub} clips")
                self.set_option("subbatch", sub)
                self._subbatch = sub
                torch.cuda.empty_cache()

    def last_status(self) -> int:
        """0 = ok; bit 0 = a bounded wait inside the persistent LSTM kernel gave up; bit 1 = fp16 range overflow in an f16x2 kernel (stage 2-3
        convs, LSTM input projection) (synchronises the device)."""
        return int(self._status.item())

    def range_report(self) -> Dict[str, float]:
        """{site: largest |x * scale| its split writers saw in the LAST call}: the measured headroom of the two-piece fp16 arithmetic, which
        overflows at 65504 (0.0: the site did not run on that scheme). Synchronises the device."""
        return _cabi.range_report(self._h.lib, "encodec", self._h.handle)

    RANGE_OPTIONS = ("ih_f16x2", "chain_f16x2", "res_f16x2", "rvq_f16x2", "fin_f16x2")

    def verified(self, codes: torch.Tensor, input_batch: torch.Tensor, attention_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Product-path guard, called where the caller synchronises anyway (tokens leaving the device). The status word of the call that
        produced `codes` decides:
        * bit 0, the persistent LSTM's hand-off timed out (another process on the GPU, a partitioned device: not all 256 workgroups
          resident): a property of the MACHINE — log it, switch to the per-step LSTM launches for the rest of the handle's life and repeat;
        * bit 1, an activation exceeded the fp16 range of the f16x2 kernels: a property of THIS BATCH — repeat it on the bf16x3 kernels (fp32
          exponent range), count it in ``fallback_batches`` and switch back: the next batch runs on f16x2 again (round 2 switched the
          handle for good, so one outlier batch halved the throughput of the rest of a run)."""
        status = self.last_status()
        if status == 0:
            return codes
        if status & 4 and not status & 2:   # (with bit 1 set the infinity descends from the flagged fp16 overflow: the repeat below cures it)
            # a NaN / infinity reached the quantiser (a non-finite sample in the waveform, as a rule): no kernel choice changes that. The reference emits
            # arbitrary ids for such input without a diagnostic; here it is at least logged and counted. The ids are returned as they are.
            self.nonfinite_batches += 1
            logger.error(f"acoustic encode: a NaN or an infinity reached the quantiser (status {status}); check the input waveform. "
                         f"The token ids of this batch are meaningless (non-finite batch #{self.nonfinite_batches})")
            if status & ~4 == 0:
                return codes
        if status & 1:
            if self.get_option("lstm_pipe") == 1 and input_batch.shape[0] <= 80:
                # the pipelined two-layer launch (lstm_pipe.hip) needs 48 co-resident workgroups per 16 clips, the layer-by-layer one 16: try that first
                logger.error(f"persistent LSTM hand-off timed out (status {status}): the tokens of this batch were discarded; "
                             "re-encoding with the layer-by-layer persistent LSTM (option lstm_pipe=0) from now on")
                self.set_option("lstm_pipe", 0)
            else:
                logger.error(f"persistent LSTM hand-off timed out (status {status}): the tokens of this batch were discarded; "
                             "re-encoding with per-step LSTM launches (option persistent_lstm=0) from now on")
                self.set_option("persistent_lstm", 0)
        saved = {}
        if status & 2:
            self.fallback_batches += 1
            logger.error(f"an activation exceeded the fp16 range of the f16x2 kernels (SEANet convs, LSTM input projection, final conv, RVQ search; status {status}): "
                         f"the tokens of this batch were discarded; re-encoding THIS batch with the bf16x3 kernels (fallback batch #{self.fallback_batches})")
            for opt in self.RANGE_OPTIONS:
                saved[opt] = self.get_option(opt)
                self.set_option(opt, 0)
        try:
            codes = self.forward(input_batch, attention_mask)
            if self.last_status() & 1 and self.get_option("persistent_lstm") == 1:   # the layer-by-layer persistent launch timed out as well
                logger.error("persistent LSTM hand-off timed out again: re-encoding with per-step LSTM launches (option persistent_lstm=0) from now on")
                self.set_option("persistent_lstm", 0)
                codes = self.forward(input_batch, attention_mask)
            if self.last_status() & 4:          # still non-finite on the safe kernels: it came with the input, not from the fp16 range
                self.nonfinite_batches += 1
                logger.error(f"a NaN or an infinity reached the quantiser on the fallback kernels too (non-finite batch #{self.nonfinite_batches}): check the input waveform")
            if self.last_status() & ~4 != 0:   # (bit 2, non-finite input, is not something a repeat can clear)
                raise _cabi.HipLibraryError("acoustic encode failed twice (status non-zero on the fallback kernels)")
        finally:
            for opt, v in saved.items():
                self.set_option(opt, v)
        return codes

    @torch.no_grad()
    def forward(self, input_batch: torch.Tensor, attention_mask: Optional[torch.Tensor] = None,
                return_embeddings: bool = False):
        """``float32 [B, N]`` on the device (+ ignored mask) -> ``int16 [B, n_q, ceil(N/320)]`` on the device."""
        assert input_batch.dim() == 2, "input_batch must be [B, N]"
        x = input_batch.to(device=self.device, dtype=torch.float32).contiguous()
        B, N = x.shape
        T = -(-N // W.ENCODEC_HOP)
        lib = self._h.lib
        codes = torch.empty((B, self.n_q, T), dtype=torch.int16, device=self.device)
        emb = torch.empty((B, T, W.ENCODEC_DIM), dtype=torch.float32, device=self.device) if return_embeddings else None
        nbytes, ws = self._sized_workspace(B, N)
        t_out = C.c_int(0)
        with torch.cuda.device(self.device):
            stream = _cabi.current_stream_handle(self.device)
            rc = lib.at_encodec_encode_checked(self._h.handle, x.data_ptr(), 0, B, N, self.n_q, codes.data_ptr(), C.byref(t_out),
                                               _cabi.ptr(emb), ws.data_ptr(), nbytes, stream, self._status.data_ptr())
        _cabi.check(rc, "at_encodec_encode_checked")
        assert t_out.value == T
        logger.info(f'Codes shape: {codes.shape}')
        if return_embeddings:
            return codes, emb
        return codes

    # ---- benchmark taps (HIP events recorded by the library on the launch stream) -----------------
    def enable_profile(self, on: bool) -> None:
        self._h.enable_profile(on)

    def read_profile(self) -> Dict[str, tuple]:
        return self._h.read_profile()


# ======================================================================================================
# semantic_m: Wav2Vec2-BERT + VQ
# ======================================================================================================
def frontend_tables() -> Dict[str, np.ndarray]:
    """Povey window and mel filter bank, built with the reference's formulas so the device uses the very same
    fp32 tables (reference audiotoken/processors.py:8-26,66-78; audiotoken/utils.py:286-328): triangles built in
    mel space over 256 bins of width 16000/512 from 20 Hz to 8 kHz (Kaldi mel), plus one zero row."""
    def hz2mel(f):
        return 1127.0 * torch.log(1.0 + (f / 700.0))

    filt = torch.linspace(hz2mel(torch.tensor(20.0)), hz2mel(torch.tensor(8000.0)), 80 + 2)
    fft_freqs = hz2mel((16000 / 512) * torch.arange(256))
    diff = torch.diff(filt)
    slopes = filt.unsqueeze(0) - fft_freqs.unsqueeze(1)
    fb = torch.maximum(torch.zeros(1), torch.minimum(-slopes[:, :-2] / diff[:-1], slopes[:, 2:] / diff[1:]))
    fb = torch.nn.functional.pad(fb, (0, 0, 0, 1))
    window = torch.pow(torch.hann_window(400, periodic=False), 0.85)
    return {"frontend.mel_filters": fb.numpy(), "frontend.window": window.numpy()}


W2VBERT_ARCH = {"hidden_size": 1024, "intermediate_size": 4096, "num_attention_heads": 16, "feature_projection_input_dim": 160,
                "position_embeddings_type": "relative_key", "left_max_position_embeddings": 64, "right_max_position_embeddings": 8,
                "conv_depthwise_kernel_size": 31, "hidden_act": "swish"}


def load_w2vbert_checkpoint(model_dir: str, quantizer_path: Optional[str]) -> Dict[str, np.ndarray]:
    """The reference's ``Wav2Vec2BertModel.from_pretrained(config.model_id)`` directory (``w2vbert2_l21/``: config.json +
    model.safetensors, possibly sharded, 21 conformer layers of which 19 are used) + the VQ ``.pkl`` state dict it ``torch.load``s
    (reference audiotoken/encoder.py:132,156-161; audiotoken/configs.py:114-134; audiotoken/utils.py:331-339) -> numpy dict."""
    W.check_hf_config(model_dir, W2VBERT_ARCH, "semantic_m checkpoint")
    w = W.read_hf_state_dict(model_dir, strip_prefixes=("wav2vec2_bert.",))
    if quantizer_path:
        sd = torch.load(quantizer_path, map_location="cpu", weights_only=True)
        if "_codebook.embed" not in sd:
            raise ValueError(f"{quantizer_path}: no '_codebook.embed' (expected a vector_quantize_pytorch VectorQuantize state dict)")
        w["vq._codebook.embed"] = sd["_codebook.embed"].float().numpy()
    return w


class Wav2VecBertEncoder(torch.nn.Module):
    """Drop-in for reference ``Wav2VecBertEncoder`` with ``quantize=True`` (audiotoken/encoder.py:111-186)."""

    def __init__(self, config=None, device: str = "cuda:0", quantize: bool = True,
                 weights: Optional[Union[str, Dict[str, np.ndarray]]] = None, packed=None):
        """``packed`` = ``(meta, blob)`` from another rank's ``export_packed()`` (``distributed.broadcast_packed``): the finalized model is rebuilt
        over that device blob — no checkpoint is read, nothing is uploaded or split here (``weights`` is ignored)."""
        super().__init__()
        from .configs import Wav2VecBertConfig
        config = config or Wav2VecBertConfig()
        self.config = config
        self.quantize = quantize
        self.output_layer = config.output_layer
        self.lib = _cabi.load()
        self.device_index = _device_index(device)
        self.device = torch.device("cuda", self.device_index)
        if packed is not None:
            self.handle = self.lib.at_w2vbert_create(self.device_index)
            if not self.handle:
                raise _cabi.HipLibraryError(f"at_w2vbert_create failed: {_cabi.last_error()}")
            _cabi.import_packed(self.lib, "w2vbert", self.handle, packed[0], packed[1].to(self.device))
            self._finish_init()
            return
        if weights is None:
            weights = config.weights
        if weights is None:
            logger.warning("No Wav2Vec2-BERT checkpoint given (weights=/AUDIOTOKEN_W2VBERT_WEIGHTS): synthetic weights, seed 0")
            weights = W.synth_w2vbert_weights(n_layers=self.output_layer, seed=0, with_vq=True)
        elif isinstance(weights, (str, bytes)):
            weights = load_w2vbert_checkpoint(weights, config.quantizer_path)
        self.handle = self.lib.at_w2vbert_create(self.device_index)
        if not self.handle:
            raise _cabi.HipLibraryError(f"at_w2vbert_create failed: {_cabi.last_error()}")
        tensors = dict(frontend_tables())
        for k, v in weights.items():
            if k.startswith("encoder.layers."):
                if int(k.split(".")[2]) >= self.output_layer:   # layers past the consumed hidden state are dead compute
                    continue
            elif not k.startswith(("feature_projection.", "vq.")):
                continue
            tensors[k] = v
        if "vq._codebook.embed" in tensors:
            e = torch.from_numpy(np.ascontiguousarray(tensors["vq._codebook.embed"], dtype=np.float32)).reshape(-1, 1024)
            tensors["vq._codebook.e2"] = (e ** 2).sum(-1).numpy()   # y2 of vector_quantize_pytorch's cdist
        for name, arr in tensors.items():
            _cabi.set_tensor(self.lib, self.lib.at_w2vbert_set_tensor, self.handle, name, arr)
        _cabi.check(self.lib.at_w2vbert_finalize(self.handle), "at_w2vbert_finalize")
        self._finish_init()

    def export_packed(self):
        """(meta bytes, uint8 device blob): this finalized model for ``Wav2VecBertEncoder(packed=...)`` on the other ranks of a node."""
        return _cabi.export_packed(self.lib, "w2vbert", self.handle, self.device)

    def _finish_init(self) -> None:
        self.n_layers = self.lib.at_w2vbert_num_layers(self.handle)
        if self.n_layers < self.output_layer:
            raise ValueError(f"checkpoint has {self.n_layers} conformer layers, output_layer={self.output_layer} needs that many")
        self._ws: Optional[torch.Tensor] = None
        self._status = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.fallback_batches = 0    # batches `verified` repeated (fp16 range overflow)
        self.pinned_layers = []      # conformer layers `verified` moved to bf16x3 for good (their activations do not fit the fp16 range)
        self.layer_overflows = {}    # {layer: batches on which it overflowed}: a layer is pinned from the PIN_AFTER-th such batch on
        self.nonfinite_batches = 0

    def __del__(self):
        h = self.__dict__.pop("handle", None)
        if h:
            self.lib.at_w2vbert_destroy(h)

    ARITH = {"f32": 0, "bf16x3": 1, "f16x2": 2}
    # Range fallback policy: a layer's FIRST overflowing batch is repeated with that layer on bf16x3 and the layer goes back to f16x2 (the outlier may have
    # come with the input: one loud or clipped file must not slow down — or change the rounding of — the rest of a run); from its PIN_AFTER-th overflowing batch
    # on the layer stays on bf16x3 (an activation outlier that is a property of the checkpoint would repeat every batch otherwise). AudioToken unpins at the
    # end of encode_batch_files and records what happened in `run_summary`.
    PIN_AFTER = 2

    def set_option(self, name: str, value) -> None:
        """"arith": "f32" | "bf16x3" | "f16x2" (or 0/1/2) — arithmetic of the linear layers (include/audiotoken_hip.h)."""
        if isinstance(value, str):
            value = self.ARITH[value]
        _cabi.check(self.lib.at_w2vbert_set_option(self.handle, name.encode(), int(value)), f"at_w2vbert_set_option({name})")

    def get_option(self, name: str) -> int:
        return int(self.lib.at_w2vbert_get_option(self.handle, name.encode()))

    def last_status(self) -> int:
        """0 = ok; bit 1 (2) = an activation overflowed the fp16 range of the f16x2 arithmetic (synchronises the device)."""
        return int(self._status.item())

    def range_report(self) -> Dict[str, float]:
        """{site: largest |x * scale| its split writers saw in the LAST call, over all layers}; the f16x2 arithmetic overflows at 65504."""
        return _cabi.range_report(self.lib, "w2vbert", self.handle)

    def verified(self, tokens: torch.Tensor, input_batch: torch.Tensor, mask: Optional[torch.Tensor] = None, **kw) -> torch.Tensor:
        """Product-path guard, called where the caller synchronises anyway: if the call that produced `tokens` reported an fp16 range
        overflow, log it, find the conformer layer that caused it, move THAT layer to the bf16x3 arithmetic (full fp32 exponent range) and repeat this
        batch (``fallback_batches``); every other layer stays on f16x2. The layer returns to f16x2 after the batch unless it is its PIN_AFTER-th overflowing
        batch (then it stays: ``pinned_layers``)."""
        status = self.last_status()
        if status == 0:
            return tokens
        if status & 4 and not status & 2:   # (with bit 1 set the infinity descends from the flagged fp16 overflow: the repeat below cures it)
            # a NaN / infinity reached the quantiser (a non-finite sample in the waveform, as a rule): no kernel choice changes that. The reference emits
            # arbitrary ids for such input without a diagnostic; here it is at least logged and counted. The ids are returned as they are.
            self.nonfinite_batches += 1
            logger.error(f"semantic_m encode: a NaN or an infinity reached the quantiser (status {status}); check the input waveform. "
                         f"The token ids of this batch are meaningless (non-finite batch #{self.nonfinite_batches})")
            if status & ~4 == 0:
                return tokens
        self.fallback_batches += 1
        # Which layer? Every layer has its own row of range flags; an overflow turns into infinities that all later layers flag too, so the FIRST flagged
        # layer is the cause. That layer alone is moved to bf16x3 (full fp32 exponent range); it stays there once it has overflowed on PIN_AFTER batches
        # (an activation outlier of the checkpoint: the next batch would overflow at the same place). The other layers keep f16x2, so a model with one such
        # layer pays ~1 / n_layers of the bf16x3 price instead of a repeat of every batch. Up to three layers are found this way per batch; beyond that the
        # whole batch is repeated on bf16x3 as in round 3.
        transient = []    # layers moved for THIS batch only (their first overflow): restored below
        try:
            for _ in range(3):
                bad = [l for l, f in enumerate(self.layer_status()) if f & 2]
                if not bad:
                    break
                layer = bad[0]
                self.layer_overflows[layer] = self.layer_overflows.get(layer, 0) + 1
                pin = self.layer_overflows[layer] >= self.PIN_AFTER
                (self.pinned_layers if pin else transient).append(layer)
                logger.error(f"semantic_m encode reported status {status}: an activation of conformer layer {layer} exceeded the fp16 range of the f16x2 "
                             f"arithmetic (batch #{self.layer_overflows[layer]} on which it did). The tokens of this batch were discarded; layer {layer} runs on "
                             f"bf16x3 (option layer_arith:{layer} = 1) " + ("from now on" if pin else "for this batch") +
                             f", this batch is re-encoded (fallback batch #{self.fallback_batches})")
                self.set_option(f"layer_arith:{layer}", 1)
                tokens = self.forward(input_batch, mask, **kw)
                status = self.last_status()
                if not status & 2:
                    if status & 4:
                        self.nonfinite_batches += 1
                        logger.error(f"a NaN or an infinity reached the quantiser with layer {layer} on bf16x3 too (non-finite batch #{self.nonfinite_batches}): check the input waveform")
                    return tokens
        finally:
            for layer in transient:
                self.set_option(f"layer_arith:{layer}", -1)
        logger.error(f"semantic_m encode still reports status {status}: re-encoding THIS batch with arith=bf16x3 for every layer")
        saved = self.get_option("arith")
        self.set_option("arith", "bf16x3")
        try:
            tokens = self.forward(input_batch, mask, **kw)
            if self.last_status() & 4:          # still non-finite on the safe kernels: it came with the input, not from the fp16 range
                self.nonfinite_batches += 1
                logger.error(f"a NaN or an infinity reached the quantiser on the fallback kernels too (non-finite batch #{self.nonfinite_batches}): check the input waveform")
            if self.last_status() & ~4 != 0:   # (bit 2, non-finite input, is not something a repeat can clear)
                raise _cabi.HipLibraryError("semantic_m encode failed twice (status non-zero with bf16x3 arithmetic)")
        finally:
            self.set_option("arith", saved)
        return tokens

    def site_scales(self) -> Dict[str, list]:
        """{site: [scale per conformer layer]} — the power of two each split site multiplies its activations by (16, or the provable scale of a
        LayerNorm-fed site whose gains are too large for 16: include/audiotoken_hip.h, at_w2vbert_site_scales)."""
        names = C.create_string_buffer(2048)
        ns = self.lib.at_w2vbert_range_sites(names, 2048)
        buf = (C.c_float * (64 * 16))()
        n = self.lib.at_w2vbert_site_scales(self.handle, buf, 64 * 16)
        if ns <= 0 or n < 0:
            raise _cabi.HipLibraryError(f"at_w2vbert_site_scales failed: {_cabi.last_error()}")
        keys = names.value.decode().split("\n")[:ns]
        return {k: [float(buf[l * ns + i]) for l in range(n // ns)] for i, k in enumerate(keys)}

    def layer_status(self):
        """Per conformer layer, the OR of its split sites' status flags in the LAST call (bit 1 = fp16 range overflow in that layer). Synchronises."""
        buf = (C.c_int32 * 64)()
        n = self.lib.at_w2vbert_layer_status(self.handle, buf, 64)
        if n < 0:
            raise _cabi.HipLibraryError(f"at_w2vbert_layer_status failed: {_cabi.last_error()}")
        return [int(buf[i]) for i in range(n)]

    def unpin_layers(self) -> None:
        """Return every layer the range fallback moved to bf16x3 to the handle's arithmetic."""
        for layer in set(self.pinned_layers):
            self.set_option(f"layer_arith:{layer}", -1)
        self.pinned_layers = []
        self.layer_overflows = {}

    def _workspace(self, nbytes: int) -> torch.Tensor:
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = None
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        return self._ws

    @torch.no_grad()
    def forward(self, input_batch: torch.Tensor, mask: Optional[torch.Tensor] = None, pad_to_multiple_of: int = 2,
                n_layers: Optional[int] = None, return_taps: bool = False):
        """``float32 [B, N]`` @16 kHz + ``float32 [B, N]`` mask -> ``int16 [B, 1, T']`` on the device.
        With ``quantize=False`` returns the selected hidden state ``[B, T', 1024]`` like the reference."""
        assert input_batch.dim() == 2, "Input tensor must have shape [batch, time]"
        x = input_batch.to(device=self.device, dtype=torch.float32).contiguous()
        m = None if mask is None else mask.to(device=self.device, dtype=torch.float32).contiguous()
        B, N = x.shape
        nl = self.output_layer if n_layers is None else n_layers
        T = self.lib.at_w2vbert_num_tokens(N, pad_to_multiple_of)
        want_tokens = self.quantize
        tokens = torch.empty((B, 1, T), dtype=torch.int16, device=self.device) if want_tokens else None
        feats = amask = hidden = None
        if return_taps or not want_tokens:
            hidden = torch.empty((B, T, 1024), dtype=torch.float32, device=self.device)
        if return_taps:
            feats = torch.empty((B, T, 160), dtype=torch.float32, device=self.device)
            amask = torch.empty((B, T), dtype=torch.float32, device=self.device)
        nbytes = self.lib.at_w2vbert_workspace_bytes(self.handle, B, N, pad_to_multiple_of)
        ws = self._workspace(nbytes)
        t_out = C.c_int(0)
        with torch.cuda.device(self.device):
            rc = self.lib.at_w2vbert_encode_checked(self.handle, x.data_ptr(), _cabi.ptr(m), B, N, pad_to_multiple_of, nl,
                                                    _cabi.ptr(tokens), C.byref(t_out), _cabi.ptr(feats), _cabi.ptr(amask),
                                                    _cabi.ptr(hidden), ws.data_ptr(), nbytes, _cabi.current_stream_handle(self.device),
                                                    self._status.data_ptr())
        _cabi.check(rc, "at_w2vbert_encode_checked")
        assert t_out.value == T
        if return_taps:
            return tokens, {"input_features": feats, "attention_mask": amask, "hidden": hidden}
        return tokens if want_tokens else hidden

    def enable_profile(self, on: bool) -> None:
        _cabi.check(self.lib.at_w2vbert_profile(self.handle, 1 if on else 0), "at_w2vbert_profile")

    def read_profile(self) -> Dict[str, tuple]:
        names = C.create_string_buffer(4096)
        ms = (C.c_float * 64)()
        ln = (C.c_int * 64)()
        n = self.lib.at_w2vbert_profile_read(self.handle, names, 4096, ms, ln, 64)
        if n < 0:
            raise _cabi.HipLibraryError(f"at_w2vbert_profile_read failed: {_cabi.last_error()}")
        keys = names.value.decode().split("\n")[:n]
        return {k: (float(ms[i]), int(ln[i])) for i, k in enumerate(keys)}
