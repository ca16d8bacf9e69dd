"""ctypes binding of libaudiotoken_hip.so (C ABI: include/audiotoken_hip.h).

The HIP library IS the product path: if it cannot be loaded this module raises — there is no CPU or
PyTorch fallback anywhere in ``audiotoken_amd``.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

_LIB_PATH = Path(__file__).resolve().parent / "lib" / "libaudiotoken_hip.so"
_lib = None


class HipLibraryError(RuntimeError):
    pass


class SegmentDesc(C.Structure):
    """Mirror of ``at_segment_desc`` (include/audiotoken_hip.h): one output row of at_segments_from_pcm."""
    _fields_ = [("pcm", C.c_void_p), ("table", C.c_void_p), ("chunk_off", C.c_int64), ("chunk_len", C.c_int32), ("out_start", C.c_int32),
                ("valid_len", C.c_int32), ("fmt", C.c_int32), ("scale", C.c_float), ("o", C.c_int32), ("n", C.c_int32), ("width", C.c_int32),
                ("chunk_out_len", C.c_int32)]


PCM_S16, PCM_S32, PCM_F32, PCM_U8 = 0, 1, 2, 3


class GemmDesc(C.Structure):
    """Mirror of ``at_gemm_desc`` (include/audiotoken_hip.h)."""
    _fields_ = [
        ("X", C.c_void_p), ("x_bstride", C.c_int64), ("Tin", C.c_int32), ("Cin", C.c_int32), ("ldx", C.c_int32),
        ("ktaps", C.c_int32), ("stride", C.c_int32), ("pad_left", C.c_int32), ("pad_mode", C.c_int32),
        ("W", C.c_void_p), ("bias", C.c_void_p),
        ("C", C.c_void_p), ("c_bstride", C.c_int64), ("ldc", C.c_int32),
        ("R", C.c_void_p), ("r_bstride", C.c_int64), ("ldr", C.c_int32),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("batch", C.c_int32),
        ("pro", C.c_int32), ("epi", C.c_int32), ("alpha", C.c_float),
        ("aux_off", C.c_int32), ("row_mask", C.c_void_p),
    ]


# name -> (restype, argtypes); every symbol include/audiotoken_hip.h declares
SIGNATURES = {
    "at_version": (C.c_int, []),
    "at_last_error": (C.c_char_p, []),
    "at_encodec_create": (C.c_void_p, [C.c_int]),
    "at_encodec_set_tensor": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int]),
    "at_encodec_finalize": (C.c_int, [C.c_void_p, C.c_int]),
    "at_encodec_destroy": (None, [C.c_void_p]),
    "at_encodec_num_codebooks": (C.c_int, [C.c_void_p]),
    "at_encodec_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int]),
    "at_encodec_encode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                    C.POINTER(C.c_int), C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "at_encodec_encode_checked": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                            C.POINTER(C.c_int), C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "at_encodec_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "at_encodec_profile": (C.c_int, [C.c_void_p, C.c_int]),
    "at_encodec_profile_read": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_float), C.POINTER(C.c_int), C.c_int]),
    "at_encodec_decode_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int]),
    "at_encodec_decode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                    C.c_size_t, C.c_void_p]),
    "at_encodec_decode_checked": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                            C.c_size_t, C.c_void_p, C.c_void_p]),
    "at_w2vbert_create": (C.c_void_p, [C.c_int]),
    "at_w2vbert_set_tensor": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int]),
    "at_w2vbert_finalize": (C.c_int, [C.c_void_p]),
    "at_w2vbert_packed_bytes": (C.c_int64, [C.c_void_p]),
    "at_w2vbert_packed_meta": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_int64]),
    "at_w2vbert_export_packed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "at_w2vbert_import_packed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]),
    "at_w2vbert_destroy": (None, [C.c_void_p]),
    "at_w2vbert_num_layers": (C.c_int, [C.c_void_p]),
    "at_w2vbert_num_tokens": (C.c_int, [C.c_int, C.c_int]),
    "at_w2vbert_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "at_w2vbert_encode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                    C.POINTER(C.c_int), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "at_w2vbert_encode_checked": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                            C.POINTER(C.c_int), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "at_w2vbert_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "at_w2vbert_get_option": (C.c_int, [C.c_void_p, C.c_char_p]),
    "at_encodec_get_option": (C.c_int, [C.c_void_p, C.c_char_p]),
    "at_w2vbert_profile": (C.c_int, [C.c_void_p, C.c_int]),
    "at_w2vbert_profile_read": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_float), C.POINTER(C.c_int), C.c_int]),
    "at_hubert_create": (C.c_void_p, [C.c_int]),
    "at_hubert_set_tensor": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int]),
    "at_hubert_finalize": (C.c_int, [C.c_void_p]),
    "at_hubert_packed_bytes": (C.c_int64, [C.c_void_p]),
    "at_hubert_packed_meta": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_int64]),
    "at_hubert_export_packed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "at_hubert_import_packed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]),
    "at_hubert_destroy": (None, [C.c_void_p]),
    "at_hubert_num_layers": (C.c_int, [C.c_void_p]),
    "at_hubert_num_tokens": (C.c_int, [C.c_int]),
    "at_hubert_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int]),
    "at_hubert_encode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int),
                                   C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "at_hubert_encode_checked": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int),
                                           C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "at_hubert_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "at_hubert_get_option": (C.c_int, [C.c_void_p, C.c_char_p]),
    "at_hubert_profile": (C.c_int, [C.c_void_p, C.c_int]),
    "at_hubert_profile_read": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_float), C.POINTER(C.c_int), C.c_int]),
    "at_encodec_range_sites": (C.c_int, [C.c_char_p, C.c_size_t]),
    "at_encodec_range_report": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_int]),
    "at_w2vbert_range_sites": (C.c_int, [C.c_char_p, C.c_size_t]),
    "at_w2vbert_range_report": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_int]),
    "at_w2vbert_site_scales": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_int]),
    "at_hubert_site_scales": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_int]),
    "at_w2vbert_layer_status": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.c_int]),
    "at_hubert_layer_status": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.c_int]),
    "at_hubert_range_sites": (C.c_int, [C.c_char_p, C.c_size_t]),
    "at_hubert_range_report": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_int]),
    "at_segments_from_pcm": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "at_clock_stamp": (C.c_int, [C.c_void_p, C.c_void_p]),
    "at_segments_zmuv_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "at_segments_from_pcm_zmuv": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "at_flac_info": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int64), C.c_void_p]),
    "at_flac_decode": (C.c_int64, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int64]),
    "at_op_layernorm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "at_op_relpos_attention": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "at_op_relpos_attention_kvp": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t,
                                             C.c_void_p, C.c_void_p]),
    "at_op_dwconv_ln_swish": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "at_op_dwconv_stream": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "at_op_vq_argmax": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "at_op_vq_argmax_refined": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "at_op_gemm": (C.c_int, [C.POINTER(GemmDesc), C.c_void_p]),
    "at_required_tensors": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.c_size_t]),
    "at_op_gemm_split": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int,
                                   C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "at_op_conv_split": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                  C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "at_op_rvq_encode_split": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_float,
                                        C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "at_op_rvq_encode": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
}


def lib_path() -> Path:
    return Path(os.environ.get("AUDIOTOKEN_HIP_LIB", str(_LIB_PATH)))


def load():
    """Load the shared library once; raise HipLibraryError if it is missing or incomplete."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not path.exists():
        raise HipLibraryError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"(or `make -C audiotoken_amd/csrc`). audiotoken_amd has no CPU fallback.")
    try:
        lib = C.CDLL(str(path))
    except OSError as e:  # pragma: no cover - environment specific
        raise HipLibraryError(f"cannot load {path}: {e}") from e
    for name, (restype, argtypes) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HipLibraryError(f"{path} does not export {name}") from e
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def required_tensors(model: str, n: int, with_extras: bool = True):
    """{name: shape} of the host tensors `model`'s finalize() needs (include/audiotoken_hip.h: at_required_tensors)."""
    lib = load()
    need = -lib.at_required_tensors(model.encode(), n, 1 if with_extras else 0, None, 0)
    if need <= 1:   # -1 = error (a real list always needs more than one byte)
        raise HipLibraryError(f"at_required_tensors({model}) failed: {last_error()}")
    buf = C.create_string_buffer(need)
    cnt = lib.at_required_tensors(model.encode(), n, 1 if with_extras else 0, buf, need)
    out = {}
    for line in buf.value.decode().splitlines():
        parts = line.split()
        out[parts[0]] = tuple(int(x) for x in parts[1:])
    assert len(out) == cnt
    return out


def last_error() -> str:
    msg = load().at_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc: int, what: str):
    if rc != 0:
        raise HipLibraryError(f"{what} failed (code {rc}): {last_error()}")


def ptr(t) -> int:
    """Device/host address of a torch tensor (or None -> NULL)."""
    return 0 if t is None else t.data_ptr()


def current_stream_handle(device) -> int:
    import torch

    return torch.cuda.current_stream(device).cuda_stream


def set_tensor(lib, fn, handle, name: str, arr: np.ndarray):
    arr = np.ascontiguousarray(arr, dtype=np.float32)
    shape = (C.c_int64 * arr.ndim)(*arr.shape)
    check(fn(handle, name.encode(), arr.ctypes.data_as(C.c_void_p), shape, arr.ndim), f"set_tensor({name})")


def range_report(lib, model: str, handle) -> dict:
    """{site: largest |x * scale| the site's split writers saw in the handle's last call} (include/audiotoken_hip.h, at_*_range_report); the
    fp16 scheme overflows at 65504. Synchronises the device."""
    names = C.create_string_buffer(2048)
    n = getattr(lib, f"at_{model}_range_sites")(names, 2048)
    if n < 0:
        raise HipLibraryError(f"at_{model}_range_sites failed")
    vals = (C.c_float * 64)()
    check(getattr(lib, f"at_{model}_range_report")(handle, vals, 64) - n, f"at_{model}_range_report")
    keys = names.value.decode().split("\n")[:n]
    return {k: float(vals[i]) for i, k in enumerate(keys)}


def export_packed(lib, model: str, handle, device):
    """(meta: bytes, blob: uint8 device tensor) of a finalized handle (include/audiotoken_hip.h, at_*_export_packed): the model as ONE device blob
    plus a small host record, for ``import_packed`` on another rank."""
    import torch
    n = getattr(lib, f"at_{model}_packed_meta")(handle, None, 0)
    if n < 0:
        raise HipLibraryError(f"at_{model}_packed_meta failed: {last_error()}")
    buf = C.create_string_buffer(int(n))
    if getattr(lib, f"at_{model}_packed_meta")(handle, buf, n) != n:
        raise HipLibraryError(f"at_{model}_packed_meta failed: {last_error()}")
    nbytes = getattr(lib, f"at_{model}_packed_bytes")(handle)
    if nbytes < 0:
        raise HipLibraryError(f"at_{model}_packed_bytes failed: {last_error()}")
    blob = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
    with torch.cuda.device(device):
        check(getattr(lib, f"at_{model}_export_packed")(handle, blob.data_ptr(), int(nbytes), current_stream_handle(device)), f"at_{model}_export_packed")
        torch.cuda.current_stream(device).synchronize()
    return buf.raw, blob


def import_packed(lib, model: str, handle, meta: bytes, blob) -> None:
    """Rebuild a finalized model on a FRESH handle from ``export_packed``'s pair (the blob is copied into the handle's own allocation)."""
    import torch
    assert blob.dtype == torch.uint8 and blob.is_cuda and blob.is_contiguous()
    with torch.cuda.device(blob.device):
        check(getattr(lib, f"at_{model}_import_packed")(handle, meta, len(meta), blob.data_ptr(), blob.numel(), current_stream_handle(blob.device)),
              f"at_{model}_import_packed")

