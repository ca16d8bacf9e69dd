"""Constants and config dataclasses of the tokenizers (mirrors reference audiotoken/configs.py).

The reference downloads checkpoints from the HF hub while its config classes are being defined
(audiotoken/configs.py:55-58, 65-70, 114-134); there is no network here, so paths are plain optional
fields filled from constructor kwargs / environment variables instead. Everything else (names, rates,
layer indices, extension lists, ``AudioConfig.length_tokens``) is kept identical.
"""
from __future__ import annotations

import enum
import os
from dataclasses import dataclass
from math import ceil
from typing import Optional

AUDIO_EXTS = ('.mp3', '.flac', '.wav', '.ogg', '.opus')
TAR_EXTS = ('.tar', '.tar.gz', '.tgz', '.tar.bz2', '.tbz', '.tar.xz', '.txz')
ZIP_EXTS = ('.zip', '.ZIP')


class Tokenizers(str, enum.Enum):
    """Reference audiotoken/configs.py:20-23 (a StrEnum whose values equal the member names)."""
    acoustic = "acoustic"
    semantic_s = "semantic_s"
    semantic_m = "semantic_m"

    def __str__(self) -> str:  # StrEnum semantics on Python 3.10
        return str(self.value)


@dataclass
class EncoderConfig:
    model_id: str
    model_sample_rate: int
    model_token_rate: int
    pad_token: Optional[int]


@dataclass
class AcousticEncoderConfig(EncoderConfig):
    """Reference audiotoken/configs.py:33-39."""
    model_id: str = 'encodec'
    model_sample_rate: int = 24_000
    bandwidth: float = 12
    model_token_rate: int = 75
    pad_token: Optional[int] = 0
    weights: Optional[str] = os.environ.get("AUDIOTOKEN_ENCODEC_WEIGHTS")


@dataclass
class AcousticDecoderConfig(AcousticEncoderConfig):
    """Reference audiotoken/configs.py:41-47."""
    bandwidth: float = 6


@dataclass
class HubertEncoderConfig(EncoderConfig):
    """Reference audiotoken/configs.py:49-59."""
    model_id: str = 'voidful/mhubert-base'
    model_sample_rate: int = 16_000
    output_layer: int = 11
    model_token_rate: int = 50
    quantizer_path: Optional[str] = os.environ.get("AUDIOTOKEN_HUBERT_KMEANS")
    pad_token: Optional[int] = 0
    weights: Optional[str] = os.environ.get("AUDIOTOKEN_HUBERT_WEIGHTS")


@dataclass
class Wav2VecBertConfig(EncoderConfig):
    """Reference audiotoken/configs.py:112-135 (21-layer trimmed w2v-bert-2.0, hidden state 19, VQ 2048x1024)."""
    model_id: str = 'cmeraki/audiotoken:w2vbert2_l21'
    model_sample_rate: int = 16_000
    model_token_rate: int = 50
    output_layer: int = 19
    quantizer_path: Optional[str] = os.environ.get("AUDIOTOKEN_W2VBERT_VQ")
    pad_token: Optional[int] = 0
    weights: Optional[str] = os.environ.get("AUDIOTOKEN_W2VBERT_WEIGHTS")


@dataclass
class AudioConfig:
    """Per-chunk metadata (reference audiotoken/configs.py:190-218)."""
    file_name: str
    start_idx: Optional[int] = None
    end_idx: Optional[int] = None
    length_seconds: Optional[float] = None
    length_samples: Optional[int] = None
    model_token_rate: Optional[int] = None

    @property
    def length_tokens(self) -> int:
        if self.model_token_rate is None or self.length_seconds is None:
            raise ValueError("Model token rate or length of the audio file is not provided")
        return ceil(self.length_seconds * self.model_token_rate)


def bandwidth_to_num_codebooks(bandwidth: float) -> int:
    """Reference audiotoken/utils.py:418-429."""
    return {1.5: 2, 3: 4, 6: 8, 12: 16, 24: 32}[bandwidth]


def num_codebooks_to_bandwidth(num_codebooks: int) -> float:
    """Reference audiotoken/utils.py:432-443."""
    return {2: 1.5, 4: 3, 8: 6, 16: 12}[num_codebooks]
