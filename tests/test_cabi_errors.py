"""The C ABI's error contract (include/audiotoken_hip.h): every entry point returns an error code and leaves a message in
at_last_error(); nothing aborts, nothing falls back. CPU part: the pure helper functions and the no-GPU failures; GPU
part: argument validation of the encode / decode / operator calls."""
import ctypes as C

import numpy as np
import pytest
import torch

from audiotoken_amd import _cabi, weights as W


def test_token_count_helpers_match_the_reference_formulas():
    lib = _cabi.load()
    from oracle import hubert_ref, w2vbert_ref
    for n in (400, 16000, 47999, 480000):
        assert lib.at_hubert_num_tokens(n) == hubert_ref.num_frames(n)
    for n, pad in ((16000, 2), (480000, 2), (12345, 2), (16000, 1)):
        feats, _ = w2vbert_ref.processor(torch.zeros(1, n), torch.ones(1, n), pad)
        assert lib.at_w2vbert_num_tokens(n, pad) == feats.shape[1]


def test_null_handles_are_rejected_with_a_message():
    lib = _cabi.load()
    assert lib.at_encodec_finalize(None, 0) != 0 and _cabi.last_error()
    assert lib.at_w2vbert_finalize(None) != 0 and _cabi.last_error()
    assert lib.at_hubert_finalize(None) != 0 and _cabi.last_error()
    assert lib.at_op_gemm(None, None) != 0 and "null" in _cabi.last_error().lower()
    assert lib.at_encodec_workspace_bytes(None, 0, 0) == 0


@pytest.mark.gpu
def test_encode_argument_validation(cuda_device):
    from audiotoken_amd.configs import AcousticEncoderConfig
    from audiotoken_amd.encoder import AcousticEncoder
    enc = AcousticEncoder(AcousticEncoderConfig(bandwidth=6), device="cuda:0", weights=W.synth_encodec_weights(seed=0, with_decoder=False))
    lib, h = enc._h.lib, enc._h.handle
    B, N = 2, 3200
    wav = torch.zeros(B, N, device="cuda")
    codes = torch.zeros(B, 8, 10, dtype=torch.int16, device="cuda")
    nbytes = lib.at_encodec_workspace_bytes(h, B, N)
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    st = _cabi.current_stream_handle(torch.device("cuda:0"))
    ok = lib.at_encodec_encode(h, wav.data_ptr(), None, B, N, 8, codes.data_ptr(), None, None, ws.data_ptr(), nbytes, st)
    assert ok == 0
    # workspace too small, too many codebooks, null output: error code + message, the handle stays usable
    assert lib.at_encodec_encode(h, wav.data_ptr(), None, B, N, 8, codes.data_ptr(), None, None, ws.data_ptr(), nbytes // 2, st) != 0
    assert "workspace" in _cabi.last_error()
    assert lib.at_encodec_encode(h, wav.data_ptr(), None, B, N, 64, codes.data_ptr(), None, None, ws.data_ptr(), nbytes, st) != 0
    assert _cabi.last_error()
    assert lib.at_encodec_encode(h, wav.data_ptr(), None, B, N, 8, None, None, None, ws.data_ptr(), nbytes, st) != 0
    assert lib.at_encodec_set_option(h, b"no_such_option", 1) != 0
    # decode on a handle finalized without the decoder
    assert lib.at_encodec_decode(h, codes.data_ptr(), B, 8, 10, wav.data_ptr(), ws.data_ptr(), nbytes, st) != 0
    assert "decoder" in _cabi.last_error()
    assert lib.at_encodec_encode(h, wav.data_ptr(), None, B, N, 8, codes.data_ptr(), None, None, ws.data_ptr(), nbytes, st) == 0
    torch.cuda.synchronize()


@pytest.mark.gpu
def test_finalize_reports_missing_tensors(cuda_device):
    lib = _cabi.load()
    h = lib.at_encodec_create(0)
    assert h
    try:
        arr = np.zeros((32, 1, 7), dtype=np.float32)
        _cabi.set_tensor(lib, lib.at_encodec_set_tensor, h, "encoder.model.0.conv.conv.weight", arr)
        assert lib.at_encodec_finalize(h, 0) != 0
        assert _cabi.last_error()
    finally:
        lib.at_encodec_destroy(h)


@pytest.mark.gpu
def test_gemm_descriptor_validation(cuda_device):
    lib = _cabi.load()
    x = torch.zeros(64, 30, device="cuda")
    d = _cabi.GemmDesc()
    d.X, d.x_bstride, d.Tin, d.Cin, d.ldx = x.data_ptr(), 0, 64, 30, 30        # Cin not a multiple of 4
    d.ktaps, d.stride, d.pad_left, d.pad_mode = 1, 1, 0, 0
    d.W, d.bias, d.C, d.c_bstride, d.ldc = x.data_ptr(), 0, x.data_ptr(), 0, 32
    d.R, d.r_bstride, d.ldr = 0, 0, 32
    d.M, d.N, d.K, d.batch, d.pro, d.epi, d.alpha = 64, 32, 30, 1, 0, 0, 1.0
    assert lib.at_op_gemm(C.byref(d), _cabi.current_stream_handle(torch.device("cuda:0"))) != 0
    assert "multiple" in _cabi.last_error()
