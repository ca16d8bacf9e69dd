"""audiotoken_amd — MI355X-native hot path of cmeraki/audiotoken behind the reference's API surface.

``from audiotoken_amd import AudioToken, Tokenizers, AUDIO_EXTS, TAR_EXTS, ZIP_EXTS, read_audio``
(reference audiotoken/__init__.py:1-3).
"""
from .configs import AUDIO_EXTS, TAR_EXTS, ZIP_EXTS, Tokenizers  # noqa: F401

__all__ = ["AudioToken", "Tokenizers", "AUDIO_EXTS", "TAR_EXTS", "ZIP_EXTS", "read_audio"]


def __getattr__(name):
    # lazy: importing the package must not require torch/HIP until the API objects are used
    if name == "AudioToken":
        from .core import AudioToken
        return AudioToken
    if name == "read_audio":
        from .audio_io import read_audio
        return read_audio
    raise AttributeError(name)
