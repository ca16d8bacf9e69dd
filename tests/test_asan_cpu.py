"""AddressSanitizer run of the library's HOST side (`make -C audiotoken_amd/csrc asan`): tensor staging, the finalize() packers up to their first
device call, argument / descriptor validation and the error plumbing, on this GPU-less machine. Any ASan report fails the test."""
import os
import subprocess
import sys
import glob

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASAN_LIB = os.path.join(ROOT, "audiotoken_amd", "lib", "libaudiotoken_hip_asan.so")

DRIVER = r'''
import ctypes as C, numpy as np, sys
from audiotoken_amd import _cabi, weights as W
from audiotoken_amd.encoder import fold_encodec_weights, frontend_tables
from audiotoken_amd.hubert import fold_hubert_weights
lib = _cabi.load()
assert lib.at_version() >= 1
# required-tensor lists: too-small buffers must report the size, never write
for model, n in (("encodec", 8), ("w2vbert", 2), ("hubert", 2)):
    need = -lib.at_required_tensors(model.encode(), n, 1, None, 0)
    small = C.create_string_buffer(8)
    assert lib.at_required_tensors(model.encode(), n, 1, small, 8) == -need
    assert len(_cabi.required_tensors(model, n, True)) > 10
assert lib.at_required_tensors(b"nope", 1, 1, None, 0) == -1 and b"unknown model" in lib.at_last_error()
# argument validation without a handle
assert lib.at_encodec_set_tensor(None, b"x", None, None, 1) != 0
assert lib.at_w2vbert_num_tokens(480000, 2) == 1500 and lib.at_hubert_num_tokens(480000) == 1499
assert lib.at_op_gemm(None, None) != 0
d = _cabi.GemmDesc()
assert lib.at_op_gemm(C.byref(d), None) != 0          # an all-zero descriptor is refused before any launch
# staging + the packers of finalize(): handles without a device ($AUDIOTOKEN_HOST_ONLY_TEST); the first device call fails cleanly
def stage(create, set_tensor, tensors):
    h = create(0)
    assert h
    for name, arr in tensors.items():
        _cabi.set_tensor(lib, set_tensor, h, name, arr)
    return h
h = stage(lib.at_encodec_create, lib.at_encodec_set_tensor, fold_encodec_weights(W.synth_encodec_weights(seed=0, with_decoder=True, n_codebooks=4)))
bad = np.zeros((3, 3), dtype=np.float32)
_cabi.set_tensor(lib, lib.at_encodec_set_tensor, h, "encoder.model.15.conv.conv.weight", bad)      # wrong shape: refused by the packer
assert lib.at_encodec_finalize(h, 1) != 0 and len(lib.at_last_error()) > 0
lib.at_encodec_destroy(h)
h = stage(lib.at_encodec_create, lib.at_encodec_set_tensor, fold_encodec_weights(W.synth_encodec_weights(seed=0, with_decoder=True, n_codebooks=4)))
rc = lib.at_encodec_finalize(h, 1)       # packs every conv / LSTM / codebook on the host, then hipMalloc fails (no device here)
assert rc != 0
lib.at_encodec_destroy(h)
w = dict(frontend_tables(), **W.synth_w2vbert_weights(n_layers=1, seed=0, with_vq=True))
h = stage(lib.at_w2vbert_create, lib.at_w2vbert_set_tensor, w)
assert lib.at_w2vbert_finalize(h) != 0
lib.at_w2vbert_destroy(h)
h = stage(lib.at_hubert_create, lib.at_hubert_set_tensor, fold_hubert_weights(W.synth_hubert_weights(1, 0, True), 1))
assert lib.at_hubert_finalize(h) != 0
lib.at_hubert_destroy(h)
print("ASAN_DRIVER_OK")
'''


@pytest.mark.skipif(not os.path.exists(ASAN_LIB), reason="sanitizer build not present (make -C audiotoken_amd/csrc asan)")
def test_host_side_under_address_sanitizer():
    import shutil
    if shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc"):   # bring an existing sanitizer build up to date (incremental)
        mk = subprocess.run(["make", "-j8", "-C", os.path.join(ROOT, "audiotoken_amd", "csrc"), "asan"], capture_output=True, text=True, timeout=1500)
        assert mk.returncode == 0, mk.stderr[-3000:]
    rt = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    assert rt, "ASan runtime of the ROCm clang not found"
    env = dict(os.environ, LD_PRELOAD=rt[-1], AUDIOTOKEN_HIP_LIB=ASAN_LIB, AUDIOTOKEN_HOST_ONLY_TEST="1",
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=97:verify_asan_link_order=0", PYTHONPATH=ROOT)
    p = subprocess.run([sys.executable, "-c", DRIVER], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert "ERROR: AddressSanitizer" not in p.stderr, p.stderr[-4000:]
    assert p.returncode == 0 and "ASAN_DRIVER_OK" in p.stdout, (p.returncode, p.stdout[-2000:], p.stderr[-4000:])
