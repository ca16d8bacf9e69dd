"""GPU parity of the semantic_m path (C ABI) against the CPU oracle and the reference-generated golden vectors."""
import glob
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from audiotoken_amd import _cabi, prng
from audiotoken_amd import weights as W
from oracle import w2vbert_ref as R

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
FBANK = sorted(glob.glob(os.path.join(G, "fbank_*.npz")))


def _stream(dev):
    return _cabi.current_stream_handle(dev)


@pytest.fixture(scope="module")
def enc3(cuda_device):
    from audiotoken_amd.configs import Wav2VecBertConfig
    from audiotoken_amd.encoder import Wav2VecBertEncoder
    w = W.synth_w2vbert_weights(n_layers=3, seed=5, with_vq=True)
    cfg = Wav2VecBertConfig(output_layer=3)
    return Wav2VecBertEncoder(cfg, device="cuda:0", quantize=True, weights=w), w


@pytest.mark.parametrize("path", FBANK, ids=[os.path.basename(p) for p in FBANK])
def test_frontend_matches_reference(path, enc3):
    enc, _ = enc3
    g = np.load(path)
    wave, mask = torch.from_numpy(g["wave"]).cuda(), torch.from_numpy(g["mask"]).cuda()
    _, taps = enc(wave, mask, int(g["pad_to_multiple_of"]), n_layers=0, return_taps=True)
    torch.cuda.synchronize()
    got_mask = taps["attention_mask"].cpu().numpy()
    assert got_mask.shape == g["attention_mask"].shape, (got_mask.shape, g["attention_mask"].shape)
    assert np.array_equal(got_mask, g["attention_mask"]), np.nonzero(got_mask != g["attention_mask"])
    d = np.abs(taps["input_features"].cpu().numpy() - g["input_features"])
    err = d.max()
    print(f"{os.path.basename(path)}: input_features max abs err {err:.3e} at {np.unravel_index(d.argmax(), d.shape)} mean {d.mean():.2e}")
    assert err < 1e-3


@pytest.mark.parametrize("rows,D,affine,masked", [(1000, 1024, True, False), (333, 160, True, False), (257, 1024, False, True)])
def test_layernorm(cuda_device, rows, D, affine, masked):
    lib = _cabi.load()
    x = torch.from_numpy(prng.irwin_hall("ln.x", (rows, D), 2.0, 1)) + 0.5
    g = torch.from_numpy(prng.uniform("ln.g", (D,), 0.5, 1.5, 1)) if affine else None
    b = torch.from_numpy(prng.uniform("ln.b", (D,), -0.5, 0.5, 1)) if affine else None
    m = (torch.from_numpy(prng.uniform01("ln.m", rows, 1)) > 0.3).float() if masked else None
    ref = F.layer_norm(x, (D,), g, b, 1e-5)
    if masked:
        ref = ref * m.unsqueeze(1)
    xd = x.cuda()
    y = torch.empty_like(xd)
    gd, bd, md = (t.cuda() if t is not None else None for t in (g, b, m))
    _cabi.check(lib.at_op_layernorm(xd.data_ptr(), _cabi.ptr(gd), _cabi.ptr(bd), _cabi.ptr(md), y.data_ptr(), rows, D,
                                    _stream(cuda_device)), "at_op_layernorm")
    torch.cuda.synchronize()
    assert (y.cpu() - ref).abs().max().item() < 2e-5


def test_attention_matches_reference(cuda_device):
    """at_op_relpos_attention runs the kernel the product runs by default — relpos_attention_x3_kernel<SchemeF16x2> (two fp16 pieces, three
    products; $AUDIOTOKEN_SEMANTIC_ARITH selects another) — against the golden produced by the reference's own modeling_wav2vec2_bert.py:20-80."""
    lib = _cabi.load()
    g = np.load(os.path.join(G, "attention_a.npz"))
    w = W.synth_w2vbert_weights(n_layers=1, seed=int(g["weight_seed"]), with_vq=False)
    B, T = int(g["B"]), int(g["T"])
    x = torch.from_numpy(prng.irwin_hall("attn.x", (B, T, 1024), 1.0, int(g["x_seed"])))
    p = "encoder.layers.0.self_attn"
    wq = torch.cat([torch.from_numpy(w[f"{p}.linear_{n}.weight"]) for n in "qkv"])
    bq = torch.cat([torch.from_numpy(w[f"{p}.linear_{n}.bias"]) for n in "qkv"])
    qkv = F.linear(x, wq, bq).reshape(B * T, 3072).cuda().contiguous()
    de = torch.zeros(80, 64)
    de[:73] = torch.from_numpy(w[f"{p}.distance_embedding.weight"])
    ctx = torch.full((B * T, 1024), float("nan"), device="cuda")
    mask = torch.from_numpy(g["mask"]).reshape(-1).cuda()
    ded = de.cuda()
    _cabi.check(lib.at_op_relpos_attention(qkv.data_ptr(), mask.data_ptr(), ded.data_ptr(), ctx.data_ptr(), B, T,
                                           _stream(cuda_device)), "at_op_relpos_attention")
    torch.cuda.synchronize()
    out = F.linear(ctx.cpu().reshape(B, T, 1024), torch.from_numpy(w[f"{p}.linear_out.weight"]), torch.from_numpy(w[f"{p}.linear_out.bias"]))
    err = (out - torch.from_numpy(g["out"])).abs().max().item()
    print(f"attention: max abs err after out-proj {err:.3e}")
    assert err < 1e-4


@pytest.mark.parametrize("B,T", [(2, 1500), (1, 130), (3, 64)])
def test_attention_long_vs_oracle(cuda_device, B, T):
    """T spanning many key tiles (far-field bias constants) and ragged masks, against the oracle attention (the op entry runs the product's default
    arithmetic: relpos_attention_x3_kernel<SchemeF16x2>; the fp32 / bf16x3 forms are compared in test_arith_options_agree)."""
    lib = _cabi.load()
    w = W.synth_w2vbert_weights(n_layers=1, seed=11, with_vq=False)
    p = "encoder.layers.0.self_attn"
    x = torch.from_numpy(prng.irwin_hall("attn.long", (B, T, 1024), 1.0, 2))
    mask = torch.ones(B, T)
    if B > 1:
        mask[1, T * 2 // 3:] = 0
    add = ((1.0 - mask[:, None, None, :]) * torch.finfo(torch.float32).min).expand(B, 1, T, T)
    ref = R.relpos_attention(w, p, x, add)
    wq = torch.cat([torch.from_numpy(w[f"{p}.linear_{n}.weight"]) for n in "qkv"])
    bq = torch.cat([torch.from_numpy(w[f"{p}.linear_{n}.bias"]) for n in "qkv"])
    qkv = F.linear(x, wq, bq).reshape(B * T, 3072).cuda().contiguous()
    de = torch.zeros(80, 64)
    de[:73] = torch.from_numpy(w[f"{p}.distance_embedding.weight"])
    ctx = torch.full((B * T, 1024), float("nan"), device="cuda")
    md, ded = mask.reshape(-1).cuda(), de.cuda()   # keep references: temporaries would be recycled by the allocator
    _cabi.check(lib.at_op_relpos_attention(qkv.data_ptr(), md.data_ptr(), ded.data_ptr(),
                                           ctx.data_ptr(), B, T, _stream(cuda_device)), "at_op_relpos_attention")
    torch.cuda.synchronize()
    out = F.linear(ctx.cpu().reshape(B, T, 1024), torch.from_numpy(w[f"{p}.linear_out.weight"]), torch.from_numpy(w[f"{p}.linear_out.bias"]))
    err = (out - ref).abs().max().item()
    print(f"attention B={B} T={T}: max abs err {err:.3e}")
    assert err < 1e-4


def _attention_kvp(lib, dev, qkv, mask, de, B, T, heads, w8):
    """at_op_relpos_attention_kvp: k / v as the row-major fp16 pieces the fused projection writes, then the product's attention kernel
    (w8 = 1: attention_f16x2_w8.hip, w8 = 0: its round-3 twin)."""
    hid = heads * 64
    rows_pad = (B * T + 255) // 256 * 256
    ws = torch.full((4 * rows_pad * hid + 2 * 96 * 64,), float("nan"), dtype=torch.float16, device="cuda")   # padding rows stay NaN: must never reach a result
    ctx = torch.full((B * T, hid), float("nan"), device="cuda")
    status = torch.zeros(2, dtype=torch.int32, device="cuda")
    dmax = float(de.abs().max().item()) if de is not None else 0.0
    _cabi.check(lib.at_op_relpos_attention_kvp(qkv.data_ptr(), mask.data_ptr(), _cabi.ptr(de), dmax, ctx.data_ptr(), B, T, heads, w8,
                                               ws.data_ptr(), ws.numel() * 2, status.data_ptr(), _stream(dev)), "at_op_relpos_attention_kvp")
    torch.cuda.synchronize()
    assert int(status[0].item()) == 0
    return ctx


@pytest.mark.parametrize("w8", [1, 0], ids=["w8", "r3"])
def test_attention_kvp_matches_reference(cuda_device, w8):
    """The kernels the product runs (pre-split k / v) against the golden produced by the reference's own modeling_wav2vec2_bert.py:20-80."""
    lib = _cabi.load()
    g = np.load(os.path.join(G, "attention_a.npz"))
    w = W.synth_w2vbert_weights(n_layers=1, seed=int(g["weight_seed"]), with_vq=False)
    B, T = int(g["B"]), int(g["T"])
    x = torch.from_numpy(prng.irwin_hall("attn.x", (B, T, 1024), 1.0, int(g["x_seed"])))
    p = "encoder.layers.0.self_attn"
    wq = torch.cat([torch.from_numpy(w[f"{p}.linear_{n}.weight"]) for n in "qkv"])
    bq = torch.cat([torch.from_numpy(w[f"{p}.linear_{n}.bias"]) for n in "qkv"])
    qkv = F.linear(x, wq, bq).reshape(B * T, 3072).cuda().contiguous()
    de = torch.zeros(80, 64)
    de[:73] = torch.from_numpy(w[f"{p}.distance_embedding.weight"])
    mask, ded = torch.from_numpy(g["mask"]).reshape(-1).cuda(), de.cuda()
    ctx = _attention_kvp(lib, cuda_device, qkv, mask, ded, B, T, 16, w8)
    out = F.linear(ctx.cpu().reshape(B, T, 1024), torch.from_numpy(w[f"{p}.linear_out.weight"]), torch.from_numpy(w[f"{p}.linear_out.bias"]))
    err = (out - torch.from_numpy(g["out"])).abs().max().item()
    print(f"attention kvp w8={w8}: max abs err after out-proj {err:.3e}")
    assert err < 1e-4


@pytest.mark.parametrize("w8", [1, 0], ids=["w8", "r3"])
@pytest.mark.parametrize("B,T", [(2, 1500), (1, 130), (3, 64), (2, 257), (1, 1), (2, 1499)])
def test_attention_kvp_long_vs_oracle(cuda_device, B, T, w8):
    """Many key tiles (far-field constants on both sides, near-diagonal gathers, a last tile cut by T), ragged masks (one clip padded from 2/3 on,
    i.e. tiles that are wholly and partly masked) — against the oracle attention."""
    lib = _cabi.load()
    w = W.synth_w2vbert_weights(n_layers=1, seed=11, with_vq=False)
    p = "encoder.layers.0.self_attn"
    x = torch.from_numpy(prng.irwin_hall("attn.long", (B, T, 1024), 1.0, 2))
    mask = torch.ones(B, T)
    if B > 1:
        mask[1, T * 2 // 3:] = 0
    add = ((1.0 - mask[:, None, None, :]) * torch.finfo(torch.float32).min).expand(B, 1, T, T)
    ref = R.relpos_attention(w, p, x, add)
    wq = torch.cat([torch.from_numpy(w[f"{p}.linear_{n}.weight"]) for n in "qkv"])
    bq = torch.cat([torch.from_numpy(w[f"{p}.linear_{n}.bias"]) for n in "qkv"])
    qkv = F.linear(x, wq, bq).reshape(B * T, 3072).cuda().contiguous()
    de = torch.zeros(80, 64)
    de[:73] = torch.from_numpy(w[f"{p}.distance_embedding.weight"])
    md, ded = mask.reshape(-1).cuda(), de.cuda()
    ctx = _attention_kvp(lib, cuda_device, qkv, md, ded, B, T, 16, w8)
    out = F.linear(ctx.cpu().reshape(B, T, 1024), torch.from_numpy(w[f"{p}.linear_out.weight"]), torch.from_numpy(w[f"{p}.linear_out.bias"]))
    err = (out - ref).abs().max().item()
    print(f"attention kvp w8={w8} B={B} T={T}: max abs err {err:.3e}")
    assert err < 1e-4


@pytest.mark.parametrize("w8", [1, 0], ids=["w8", "r3"])
def test_attention_kvp_no_relpos_12_heads(cuda_device, w8):
    """HuBERT's use of the kernel (HF HubertAttention: 12 heads x 64, no position bias, additive finfo.min key mask) against torch fp32; logits up to ~ +-20
    and one head with a spiked key so that the running maximum jumps inside a later tile (the rescale path)."""
    lib = _cabi.load()
    B, T, heads = 2, 700, 12
    hid = heads * 64
    qkv = torch.from_numpy(prng.irwin_hall("attn.h12", (B * T, 3 * hid), 1.0, 7)) * 1.5
    qkv[400, hid:hid + 64] *= 6.0          # key 400 of clip 0, head 0
    mask = torch.ones(B, T)
    mask[1, 500:] = 0
    q, k, v = (qkv[:, i * hid:(i + 1) * hid].reshape(B, T, heads, 64).transpose(1, 2) for i in range(3))
    logits = (q @ k.transpose(-1, -2)) * 0.125 + ((1.0 - mask[:, None, None, :]) * torch.finfo(torch.float32).min)
    ref = (torch.softmax(logits, dim=-1) @ v).transpose(1, 2).reshape(B * T, hid)
    ctx = _attention_kvp(lib, cuda_device, qkv.cuda().contiguous(), mask.reshape(-1).cuda(), None, B, T, heads, w8)
    err = (ctx.cpu() - ref).abs().max().item()
    print(f"attention kvp w8={w8} 12 heads no rel-pos: max abs err {err:.3e} (max |logit| {logits[logits > -1e30].abs().max().item():.1f})")
    assert err < 2e-5


def test_dwconv_ln_swish(cuda_device):
    lib = _cabi.load()
    B, T = 2, 77
    g = torch.from_numpy(prng.irwin_hall("dw.g", (B, T, 1024), 1.0, 4))
    w = torch.from_numpy(prng.uniform("dw.w", (1024, 1, 31), -0.3, 0.3, 4))
    gm = torch.from_numpy(prng.uniform("dw.gm", (1024,), 0.5, 1.5, 4))
    bt = torch.from_numpy(prng.uniform("dw.bt", (1024,), -0.5, 0.5, 4))
    h = F.conv1d(F.pad(g.transpose(1, 2), (30, 0)), w, groups=1024).transpose(1, 2)
    ref = F.silu(F.layer_norm(h, (1024,), gm, bt, 1e-5))
    out = torch.full((B * T, 1024), float("nan"), device="cuda")
    wt = w[:, 0, :].t().contiguous().cuda()
    gd, gmd, btd = g.cuda(), gm.cuda(), bt.cuda()
    _cabi.check(lib.at_op_dwconv_ln_swish(gd.data_ptr(), wt.data_ptr(), gmd.data_ptr(), btd.data_ptr(),
                                          out.data_ptr(), B, T, _stream(cuda_device)), "at_op_dwconv_ln_swish")
    torch.cuda.synchronize()
    assert (out.cpu().reshape(B, T, 1024) - ref).abs().max().item() < 5e-5


@pytest.mark.parametrize("B,T", [(2, 77), (1, 8), (3, 5), (1, 1500), (5, 333), (64, 40)])
def test_dwconv_stream_is_bit_identical(cuda_device, B, T):
    """dwconv_stream.hip (one channel per thread walking along time, LayerNorm moments reduced over the register-stationary kernel's tree through LDS)
    against dwconv_ln_swish_kernel: the outputs must be IDENTICAL — T below one 8-row iteration, not a multiple of it, several time segments per clip
    (B = 1: 256 segments), one segment per clip (B = 64), inputs with large and tiny rows."""
    lib = _cabi.load()
    g = torch.from_numpy(prng.irwin_hall(f"dws.g{B}.{T}", (B, T, 1024), 1.0, 4))
    g[0, : min(T, 3)] *= 1e3
    g[-1, -1] *= 1e-6
    w = torch.from_numpy(prng.uniform("dws.w", (31, 1024), -0.3, 0.3, 4))
    gm = torch.from_numpy(prng.uniform("dws.gm", (1024,), 0.5, 1.5, 4))
    bt = torch.from_numpy(prng.uniform("dws.bt", (1024,), -0.5, 0.5, 4))
    gd, wd, gmd, btd = g.cuda(), w.cuda(), gm.cuda(), bt.cuda()
    outs = []
    for fn in (lib.at_op_dwconv_ln_swish, lib.at_op_dwconv_stream):
        out = torch.full((B * T, 1024), float("nan"), device="cuda")
        _cabi.check(fn(gd.data_ptr(), wd.data_ptr(), gmd.data_ptr(), btd.data_ptr(), out.data_ptr(), B, T, _stream(cuda_device)), "dwconv op")
        torch.cuda.synchronize()
        outs.append(out)
    assert not torch.isnan(outs[1]).any()
    diff = (outs[0] != outs[1]).sum().item()
    assert diff == 0, f"{diff} of {outs[0].numel()} elements differ, max {(outs[0] - outs[1]).abs().max().item():.3e}"


def test_dwconv_stream_option_keeps_tokens(enc3):
    """The whole encoder with the streaming kernel (default) and with the register-stationary one: identical tokens and hidden states."""
    enc, w = enc3
    wav = torch.from_numpy(W.synth_waveform(3, 16000 * 3 + 123, 16000, seed=77)).cuda()
    mask = torch.ones_like(wav)
    mask[2, 30000:] = 0
    try:
        assert enc.get_option("dwconv_stream") == 1
        t1 = enc(wav, mask)
        enc.set_option("dwconv_stream", 0)
        t0 = enc(wav, mask)
        assert torch.equal(t1, t0)
    finally:
        enc.set_option("dwconv_stream", 1)


def test_vq_split_option_keeps_tokens(enc3):
    """The VQ score GEMM (non-affine LayerNorm output x code book) on the split kernel (default, option vq_split) and on the fp32 MFMA: the same tokens —
    both are fp32-grade dot products, the argmax only moves where two codes tie to ~1e-6 (none in this batch; the bench batch's pinned checksum and the
    oracle comparisons of tests/test_fullsize_gpu.py cover the size)."""
    enc, w = enc3
    wav = torch.from_numpy(W.synth_waveform(4, 16000 * 3 + 777, 16000, seed=78)).cuda()
    mask = torch.ones_like(wav)
    mask[1, 25000:] = 0
    try:
        assert enc.get_option("vq_split") == 1
        t1 = enc(wav, mask)
        enc.set_option("vq_split", 0)
        t0 = enc(wav, mask)
        valid = (t1 >= 0)
        assert torch.equal(t1, t0), f"{int((t1 != t0).sum())} of {t1.numel()} ids differ between the split and the fp32 score GEMM"
        assert valid.all()
    finally:
        enc.set_option("vq_split", 1)


def test_conformer_matches_hf_golden(enc3):
    enc, w = enc3
    g = np.load(os.path.join(G, "conformer_a.npz"))
    B, N = int(g["B"]), int(g["N"])
    mask = torch.from_numpy(g["mask"])
    wave = torch.from_numpy(W.synth_waveform(B, N, 16000, seed=int(g["wave_seed"]))) * mask
    for nl, key in ((0, "hs0"), (1, "hs1"), (3, "hs_last")):
        toks, taps = enc(wave.cuda(), mask.cuda(), 2, n_layers=nl, return_taps=True)
        torch.cuda.synchronize()
        assert np.array_equal(taps["attention_mask"].cpu().numpy(), g["attention_mask"])
        err = np.abs(taps["hidden"].cpu().numpy() - g[key]).max()
        print(f"hidden_states[{nl}] max abs err {err:.3e}")
        assert err < 1e-3, (key, err)
    # tokens: bit-identical to the oracle (vector_quantize_pytorch formula) on the same inputs
    ref = R.semantic_m_encode(w, wave, mask, 2, 3)
    assert toks.dtype == torch.int16 and tuple(toks.shape) == tuple(ref.shape)
    assert torch.equal(toks.cpu(), ref)


def test_tokens_vs_oracle_ragged(enc3):
    enc, w = enc3
    B, N = 3, 16000 * 2 + 250
    wave = W.synth_waveform(B, N, 16000, seed=31)
    mask = np.ones((B, N), dtype=np.float32)
    mask[1, 17000:] = 0
    mask[2, 4000:] = 0
    wave = wave * mask
    toks = enc(torch.from_numpy(wave).cuda(), torch.from_numpy(mask).cuda(), 2)
    ref = R.semantic_m_encode(w, torch.from_numpy(wave), torch.from_numpy(mask), 2, 3)
    feats, am = R.processor(torch.from_numpy(wave), torch.from_numpy(mask), 2)
    valid = am.bool().unsqueeze(1)
    same = (toks.cpu() == ref)
    # EVERY position is compared: the reference's save-time trim keeps ceil(sec * 50) tokens per streamed chunk, which includes
    # positions whose frames are padding (token mask 0) — their ids reach the .npy files, so they are part of what "the same tokens" means
    print(f"tokens equal: valid {same[valid].float().mean().item():.4f}, all {same.float().mean().item():.4f} ({int((~valid).sum())} padded positions compared too)")
    assert same.all(), f"{int((~same).sum())} token ids differ ({int((~same[valid]).sum())} of them at valid positions)"


@pytest.mark.parametrize("N", [4000, 10480, 10800, 880], ids=["T11", "T32", "T33", "T2"])
def test_tokens_vs_oracle_short_clips(enc3, N):
    """Clips shorter than one attention key tile (32 frames), exactly one, one plus a frame, and two frames: the double-buffered K / V prologue with a
    single tile, the depthwise-conv kernel below its 8-row iteration and with more time segments than rows, the LayerNorm / GEMM padding rows.
    Token ids at valid positions must equal the oracle's."""
    enc, w = enc3
    B = 3
    wave = W.synth_waveform(B, N, 16000, seed=131 + N)
    mask = np.ones((B, N), dtype=np.float32)
    mask[1, N // 2:] = 0
    wave = wave * mask
    toks = enc(torch.from_numpy(wave).cuda(), torch.from_numpy(mask).cuda(), 2)
    assert enc.last_status() == 0
    ref = R.semantic_m_encode(w, torch.from_numpy(wave), torch.from_numpy(mask), 2, 3)
    feats, am = R.processor(torch.from_numpy(wave), torch.from_numpy(mask), 2)
    valid = am.bool().unsqueeze(1)
    assert toks.shape == ref.shape
    same = (toks.cpu() == ref)
    print(f"N={N}: {same.numel()} ids compared, {int((~valid).sum())} of them at padded positions (persisted by the reference's trim: SURVEY.md App. B.7)")
    assert same.all(), f"N={N}: {int((~same).sum())} token ids differ ({int((~same[valid]).sum())} at valid positions)"


def test_encode_batch_files_semantic_m(tmp_path):
    """Files -> 2 s chunks -> zero padding + mask -> 19-layer HIP encoder -> trimmed .npy, against the oracle run on the
    same padded chunks (reference core.py:198-289 + datasets.py:75-105 semantics incl. the sample mask)."""
    from scipy.io import wavfile
    from audiotoken_amd import AudioToken, Tokenizers
    sr, chunk = 16000, 2
    base = W.synth_w2vbert_weights(n_layers=2, seed=9, with_vq=True)
    w = dict(base)
    for i in range(2, 19):   # 19 layers by aliasing two generated ones (no copies): same arithmetic per layer
        for k in list(base):
            if k.startswith(f"encoder.layers.{i % 2}."):
                w[k.replace(f"encoder.layers.{i % 2}.", f"encoder.layers.{i}.", 1)] = base[k]
    waves = {"a.wav": W.synth_waveform(1, sr * 3 + 4000, sr, seed=61)[0], "b.x.wav": W.synth_waveform(1, sr * 2, sr, seed=62)[0]}
    for name, x in waves.items():
        wavfile.write(str(tmp_path / name), sr, x)
    tok = AudioToken(Tokenizers.semantic_m, device="cuda:0", weights=w)
    out = tmp_path / "out"
    tok.encode_batch_files(batch_size=2, outdir=out, chunk_size=chunk, audio_files=[tmp_path / n for n in waves])
    assert sorted(os.listdir(out)) == ["a.npy", "b.npy"]
    wt = {k: torch.from_numpy(v) for k, v in w.items()}
    for name, x in waves.items():
        got = np.load(out / (name.split(".")[0] + ".npy"))
        pieces = []
        for i in range(0, len(x), sr * chunk):
            seg = x[i:i + sr * chunk]
            if len(seg) < 3200:
                continue
            padded = np.zeros(sr * chunk, dtype=np.float32)
            padded[:len(seg)] = seg
            m = np.zeros(sr * chunk, dtype=np.float32)
            m[:len(seg)] = 1
            ref, mg = R.semantic_m_encode(wt, torch.from_numpy(padded)[None], torch.from_numpy(m)[None], 2, 19, return_margins=True)
            _, am = R.processor(torch.from_numpy(padded)[None], torch.from_numpy(m)[None], 2)
            keep = int(np.ceil(len(seg) / sr * 50))
            pieces.append((ref[0].numpy()[:, :keep], mg[0].numpy()[:, :keep], am.numpy()[:, :keep]))
        ref_all = np.hstack([p[0] for p in pieces])
        margins = np.hstack([p[1] for p in pieces])
        valid = np.hstack([p[2] for p in pieces]) > 0
        assert got.dtype == np.int16 and got.shape == ref_all.shape, (got.shape, ref_all.shape)
        # the trim keeps ceil(sec*50) tokens, which can include positions whose frames are padding (token mask 0): the reference writes those ids to the
        # file too, so EVERY id of the file must equal the oracle's or sit on an oracle near-tie
        from tests import parity as P
        n = P.assert_tokens_equal_or_explained(torch.from_numpy(got)[None], torch.from_numpy(ref_all)[None], torch.from_numpy(margins)[None],
                                               P.VQ_TIE, f"{name}: all {got.shape[1]} tokens of the file ({int((~valid).sum())} of them at padded positions)")
        assert n <= 2


def test_fp16_range_overflow_is_reported_and_recovered():
    """Activations beyond the fp16 range of the f16x2 arithmetic (here: a LayerNorm gain of 1e4 in layer 0) must raise the status word,
    and Wav2VecBertEncoder.verified must re-encode with the bf16x3 arithmetic (full fp32 exponent range)."""
    from audiotoken_amd.configs import Wav2VecBertConfig
    from audiotoken_amd.encoder import Wav2VecBertEncoder
    w = dict(W.synth_w2vbert_weights(n_layers=2, seed=5, with_vq=True))
    w["encoder.layers.0.ffn1_layer_norm.weight"] = w["encoder.layers.0.ffn1_layer_norm.weight"] * 1.0e4
    cfg = Wav2VecBertConfig(output_layer=2)
    wav = torch.from_numpy(W.synth_waveform(2, 16000, 16000, seed=8)).cuda()
    mask = torch.ones_like(wav)
    safe = Wav2VecBertEncoder(cfg, device="cuda:0", quantize=True, weights=w)
    safe.set_option("arith", "bf16x3")
    ref = safe(wav, mask).clone()
    assert safe.last_status() == 0
    enc = Wav2VecBertEncoder(cfg, device="cuda:0", quantize=True, weights=w)
    assert enc.get_option("arith") == 2
    toks = enc(wav, mask)
    assert enc.last_status() & 2, "the overflow must be visible in the status word"
    toks = enc.verified(toks, wav, mask)
    assert enc.last_status() == 0 and enc.fallback_batches == 1
    assert enc.get_option("arith") == 2, "the range fallback must not outlive the batch"
    assert torch.equal(toks, ref)


def test_arith_options_agree(enc3):
    """f32 MFMA, bf16x3 and f16x2 arithmetic of the linear layers: same tokens, hidden states within 1e-3 of each other."""
    enc, w = enc3
    wave = torch.from_numpy(W.synth_waveform(2, 16000 * 2, 16000, seed=77)).cuda()
    mask = torch.ones_like(wave)
    outs = {}
    try:
        for name in ("f32", "bf16x3", "f16x2"):
            enc.set_option("arith", name)
            toks, taps = enc(wave, mask, 2, return_taps=True)
            assert enc.last_status() == 0
            outs[name] = (toks.clone(), taps["hidden"].clone())
    finally:
        enc.set_option("arith", "f16x2")
    for name in ("bf16x3", "f16x2"):
        assert (outs[name][1] - outs["f32"][1]).abs().max().item() < 1e-3, name
        assert torch.equal(outs[name][0], outs["f32"][0]), name
