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


def test_missing_library_fails_loudly():
    """No CPU or eager fallback: without the built .so, importing the binding's consumers raises HipLibraryError."""
    import subprocess
    import sys
    code = ("import os; os.environ['AUDIOTOKEN_HIP_LIB'] = '/nonexistent/libaudiotoken_hip.so'\n"
            "from audiotoken_amd import _cabi\n"
            "try:\n"
            "    _cabi.load()\n"
            "except _cabi.HipLibraryError as e:\n"
            "    print('RAISED', 'no CPU fallback' in str(e))\n")
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert "RAISED True" in out.stdout, out.stdout + out.stderr
