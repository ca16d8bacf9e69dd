"""CPU: the C-ABI library loads and exports every symbol include/audiotoken_hip.h declares (no compute)."""
import os
import re

from audiotoken_amd import _cabi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "audiotoken_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(at_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    syms = declared_symbols()
    assert syms, "no declarations parsed"
    assert sorted(_cabi.SIGNATURES) == syms


def test_library_exports_every_declared_symbol():
    lib = _cabi.load()
    for s in declared_symbols():
        assert hasattr(lib, s), s
    assert lib.at_version() >= 1


def test_create_without_gpu_fails_loudly_not_silently():
    import torch
    if torch.cuda.is_available():
        return
    lib = _cabi.load()
    h = lib.at_encodec_create(0)
    assert not h
    assert "device" in _cabi.last_error()
