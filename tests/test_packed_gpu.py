"""The finalized semantic models as ONE device blob (include/audiotoken_hip.h: at_*_export_packed / at_*_import_packed): a handle rebuilt from another
handle's export must behave exactly like the original — same tokens, same per-site range census, the bf16x3 range fallback still available — and a record
that does not match this build's allocation order must be refused, not half-imported. This is the N > 1 start-up path (bench.py
setup_semantic / run_hubert at --gpus > 1): one rank finalizes, the others import what one RCCL broadcast delivered."""
import struct

import numpy as np
import pytest
import torch

from audiotoken_amd import _cabi
from audiotoken_amd import weights as W

pytestmark = pytest.mark.gpu


def _w2vbert(n_layers=2, **kw):
    from audiotoken_amd.configs import Wav2VecBertConfig
    from audiotoken_amd.encoder import Wav2VecBertEncoder
    if "packed" not in kw:
        kw["weights"] = W.synth_w2vbert_weights(n_layers=n_layers, seed=3, with_vq=True)
    return Wav2VecBertEncoder(Wav2VecBertConfig(output_layer=n_layers), device="cuda:0", quantize=True, **kw)


def _hubert(n_layers=2, **kw):
    from audiotoken_amd.configs import HubertEncoderConfig
    from audiotoken_amd.hubert import HubertEncoder
    if "packed" not in kw:
        kw["weights"] = W.synth_hubert_weights(n_layers=n_layers, seed=3, with_kmeans=True)
    return HubertEncoder(HubertEncoderConfig(output_layer=n_layers), device="cuda:0", quantize=True, **kw)


@pytest.mark.parametrize("make", [_w2vbert, _hubert], ids=["semantic_m", "semantic_s"])
def test_import_equals_original(cuda_device, make):
    a = make()
    wav = torch.from_numpy(W.synth_waveform(3, 16000 * 2 + 777, 16000, seed=41)).cuda()
    mask = torch.ones_like(wav)
    mask[1, 20000:] = 0
    ta = a(wav, mask)
    assert a.last_status() == 0
    census_a = a.range_report()
    meta, blob = a.export_packed()
    assert blob.dtype == torch.uint8 and blob.is_cuda and blob.numel() % 256 == 0 and len(meta) > 40
    b = make(packed=(meta, blob))
    del blob                       # the importer owns a copy
    tb = b(wav, mask)
    assert b.last_status() == 0
    assert torch.equal(ta, tb)
    assert b.range_report() == census_a
    # the per-batch range fallback (arith = bf16x3: a lazy second split of the fp32 weights that travelled in the blob) works on the import too
    a.set_option("arith", "bf16x3"); b.set_option("arith", "bf16x3")
    assert torch.equal(a(wav, mask), b(wav, mask)) and b.last_status() == 0
    # ... and a handle exported AFTER that second split round-trips as well (flag bit 1 of the record: both schemes are in the blob)
    meta2, blob2 = a.export_packed()
    assert blob2.numel() > len(meta) and len(meta2) > len(meta)
    c = make(packed=(meta2, blob2))
    assert c.get_option("arith") == a.get_option("arith")
    assert torch.equal(c(wav, mask), a(wav, mask))
    c.set_option("arith", "f16x2")
    assert torch.equal(c(wav, mask), ta)


@pytest.mark.parametrize("make,model", [(_w2vbert, "w2vbert"), (_hubert, "hubert")], ids=["semantic_m", "semantic_s"])
def test_mismatching_records_are_refused(cuda_device, make, model):
    a = make()
    meta, blob = a.export_packed()
    lib = _cabi.load()

    def fresh():
        h = getattr(lib, f"at_{model}_create")(0)
        assert h
        return h

    def try_import(m, bl):
        h = fresh()
        try:
            with pytest.raises(_cabi.HipLibraryError):
                _cabi.import_packed(lib, model, h, m, bl)
        finally:
            getattr(lib, f"at_{model}_destroy")(h)

    # header: magic, version, model, n_blocks, blob_bytes (u64), n_layers, flags, arith, reserved = 40 bytes; then {u64 bytes, f32 wmax, u32} per block
    bad_magic = b"XXXX" + meta[4:]
    try_import(bad_magic, blob)
    other_model = meta[:8] + struct.pack("<I", 3 - struct.unpack_from("<I", meta, 8)[0]) + meta[12:]
    try_import(other_model, blob)
    try_import(meta[: len(meta) - 8], blob)                          # truncated record
    try_import(meta, blob[: blob.numel() - 256].contiguous())         # blob shorter than the record says
    one_layer_less = meta[:24] + struct.pack("<i", struct.unpack_from("<i", meta, 24)[0] - 1) + meta[28:]
    try_import(one_layer_less, blob)                                  # this build's finalize would make fewer blocks than the blob holds
    sz = struct.unpack_from("<Q", meta, 40 + 16 * 5)[0]
    wrong_size = meta[: 40 + 16 * 5] + struct.pack("<Q", sz + 256) + meta[40 + 16 * 5 + 8:]
    try_import(wrong_size, blob)                                      # sizes no longer add up
    # a handle that already holds tensors / is finalized cannot import
    with pytest.raises(_cabi.HipLibraryError):
        _cabi.import_packed(lib, model, a.handle, meta, blob)
    # the good record still imports after all that
    b = make(packed=(meta, blob))
    wav = torch.from_numpy(W.synth_waveform(1, 16000, 16000, seed=5)).cuda()
    assert torch.equal(a(wav, torch.ones_like(wav)), b(wav, torch.ones_like(wav)))
