"""Decoder callable with the reference's protocol: ``decoder(tokens int64 [B, K, T]) -> float32 [1, B*320*T]``
(reference audiotoken/decoder.py:50-76), backed by libaudiotoken_hip.so. Only the acoustic decoder is in scope:
the semantic decoders (nanoGPT sampling + bark, decoder.py:79-245) are stochastic and need private weights."""
from __future__ import annotations

from typing import Dict, Optional, Union

import numpy as np
import torch

from . import _cabi
from . import weights as W
from .configs import AcousticDecoderConfig
from .encoder import _EncodecHandle
from .logger import get_logger

logger = get_logger(__name__)


class AcousticDecoder(torch.nn.Module):
    """Drop-in for reference ``AcousticDecoder`` (audiotoken/decoder.py:50-76)."""

    def __init__(self, config: AcousticDecoderConfig = None, device: str = "cuda:0",
                 weights: Optional[Union[str, Dict[str, np.ndarray]]] = None):
        super().__init__()
        config = config or AcousticDecoderConfig()
        self.config = config
        self._h = _EncodecHandle(device, weights if weights is not None else config.weights, with_decoder=True)
        self.device = self._h.device
        self._ws: Optional[torch.Tensor] = None
        self._status = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.fallback_batches = 0    # batches `verified` repeated without the f16x2 kernels (fp16 range overflow)

    def last_status(self) -> int:
        """0 = ok, 1 = a bounded wait inside the persistent LSTM kernel gave up (synchronises the device)."""
        return int(self._status.item())

    def verified(self, wav: torch.Tensor, tokens: torch.Tensor) -> torch.Tensor:
        """As AcousticEncoder.verified: on an LSTM hand-off time-out repeat the decode with per-step LSTM launches."""
        status = self.last_status()
        if status == 0:
            return wav
        if status & 1:
            if self.get_option("lstm_pipe") == 1 and tokens.shape[0] <= 80:
                logger.error(f"persistent LSTM hand-off timed out in the decoder (status {status}): waveform discarded; "
                             "decoding again with the layer-by-layer persistent LSTM (option lstm_pipe=0) from now on")
                self.set_option("lstm_pipe", 0)
            else:
                logger.error(f"persistent LSTM hand-off timed out in the decoder (status {status}): waveform discarded; "
                             "decoding again with per-step LSTM launches (option persistent_lstm=0) from now on")
                self.set_option("persistent_lstm", 0)
        saved = {}
        if status & 2:
            self.fallback_batches += 1
            logger.error(f"fp16 range overflow in the decoder's f16x2 kernels (LSTM input projection, residual blocks, transposed convs; status {status}): "
                         "waveform discarded; decoding THIS batch again without them (options ih_f16x2=0, res_f16x2=0, up_f16x2=0, tail_f16x2=0)")
            for opt in ("ih_f16x2", "res_f16x2", "up_f16x2", "tail_f16x2"):
                saved[opt] = self.get_option(opt)
                self.set_option(opt, 0)
        try:
            wav = self.forward(tokens)
            if self.last_status() & 1 and self.get_option("persistent_lstm") == 1:   # the layer-by-layer persistent launch timed out as well
                logger.error("persistent LSTM hand-off timed out again in the decoder: decoding with per-step LSTM launches (option persistent_lstm=0) from now on")
                self.set_option("persistent_lstm", 0)
                wav = self.forward(tokens)
            if self.last_status() != 0:
                raise _cabi.HipLibraryError("acoustic decode failed twice (status non-zero on the fallback kernels)")
        finally:
            for opt, v in saved.items():
                self.set_option(opt, v)
        return wav

    def set_option(self, name: str, value: int) -> None:
        """Kernel-selection switches of the library (results are bit-identical either way; used by the parity tests)."""
        _cabi.check(self._h.lib.at_encodec_set_option(self._h.handle, name.encode(), int(value)), f"at_encodec_set_option({name})")

    def get_option(self, name: str) -> int:
        return int(self._h.lib.at_encodec_get_option(self._h.handle, name.encode()))

    def range_report(self):
        """{site: largest |x * scale| its split writers saw in the LAST decode} (see AcousticEncoder.range_report)."""
        return _cabi.range_report(self._h.lib, "encodec", self._h.handle)

    def enable_profile(self, on: bool) -> None:
        """HIP-event taps per kernel group (bench.py only; off by default)."""
        self._h.enable_profile(on)

    def read_profile(self) -> Dict[str, tuple]:
        return self._h.read_profile()

    @torch.no_grad()
    def forward(self, input_batch: torch.Tensor) -> torch.Tensor:
        assert input_batch.dim() == 3, "tokens must be [B, K, T]"
        codes = input_batch.to(device=self.device, dtype=torch.long).contiguous()
        B, K, T = codes.shape
        lib = self._h.lib
        out = torch.empty((1, B * W.ENCODEC_HOP * T), dtype=torch.float32, device=self.device)
        nbytes = lib.at_encodec_decode_workspace_bytes(self._h.handle, B, T)
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = None
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            rc = lib.at_encodec_decode_checked(self._h.handle, codes.data_ptr(), B, K, T, out.data_ptr(), self._ws.data_ptr(), nbytes,
                                               _cabi.current_stream_handle(self.device), self._status.data_ptr())
        _cabi.check(rc, "at_encodec_decode_checked")
        return out
