"""GPU parity of the acoustic tokenizer (AcousticEncoder through the C ABI) against the CPU oracle and the
committed HF-generated golden vectors."""
import glob
import os

import numpy as np
import pytest
import torch

from audiotoken_amd import weights as W
from oracle import encodec_ref as R
from tests import parity as P

pytestmark = pytest.mark.gpu
CASES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "encodec_*.npz")))


@pytest.fixture(scope="module")
def enc_weights():
    return W.synth_encodec_weights(seed=0)


@pytest.fixture(scope="module")
def encoders(cuda_device, enc_weights):
    from audiotoken_amd.configs import AcousticEncoderConfig
    from audiotoken_amd.encoder import AcousticEncoder
    return {nq: AcousticEncoder(AcousticEncoderConfig(bandwidth=bw), device="cuda:0", weights=enc_weights)
            for nq, bw in ((2, 1.5), (4, 3), (8, 6), (16, 12))}


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p) for p in CASES])
def test_encode_matches_golden(path, encoders):
    g = np.load(path)
    B, N, n_q = int(g["B"]), int(g["N"]), int(g["n_q"])
    wav = torch.from_numpy(W.synth_waveform(B, N, 24000, seed=int(g["wave_seed"])))
    enc = encoders[n_q]
    codes, emb = enc(wav.cuda(), torch.ones_like(wav).cuda(), return_embeddings=True)
    torch.cuda.synchronize()
    assert codes.dtype == torch.int16 and tuple(codes.shape) == (B, n_q, -(-N // 320))
    emb_ref = torch.from_numpy(g["emb"]).permute(0, 2, 1)  # golden is [B,128,T]
    err = (emb.cpu() - emb_ref).abs().max().item()
    print(f"{os.path.basename(path)}: emb max abs err {err:.3e} (|emb| max {emb_ref.abs().max().item():.2f})")
    assert err < 1e-3, "float intermediates must stay within 1e-3 of the fp32 reference"
    assert np.array_equal(codes.cpu().numpy(), g["tokens"]), "token ids must be bit-identical"


def test_encode_matches_oracle_ragged_batch(encoders, enc_weights):
    # N not a multiple of 320 (right 'extra' reflect padding) and more clips than one 64-frame RVQ tile
    B, N, n_q = 5, 12345, 8
    wav = torch.from_numpy(W.synth_waveform(B, N, 24000, seed=99))
    codes, emb = encoders[n_q](wav.cuda(), None, return_embeddings=True)
    emb_ref = R.seanet_encode(enc_weights, wav).permute(0, 2, 1)
    err = (emb.cpu() - emb_ref).abs().max().item()
    assert err < 1e-3, err
    ref = R.rvq_encode(enc_weights, emb_ref.permute(0, 2, 1), n_q).transpose(0, 1).to(torch.int16)
    assert torch.equal(codes.cpu(), ref)


def test_mask_is_ignored(encoders):
    # reference AcousticEncoder.forward never reads attention_mask (audiotoken/encoder.py:44-52)
    wav = torch.from_numpy(W.synth_waveform(2, 6400, 24000, seed=5)).cuda()
    a = encoders[8](wav, torch.ones_like(wav))
    b = encoders[8](wav, torch.zeros_like(wav))
    assert torch.equal(a, b)


# ---- decoder and the AudioToken facade (A10) on the device -----------------------------
@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p) for p in CASES])
def test_decode_matches_golden(path, enc_weights):
    from audiotoken_amd.configs import AcousticDecoderConfig
    from audiotoken_amd.decoder import AcousticDecoder
    g = np.load(path)
    dec = AcousticDecoder(AcousticDecoderConfig(), device="cuda:0", weights=enc_weights)
    toks = torch.from_numpy(g["tokens"]).long()
    wav = dec(toks.cuda())
    torch.cuda.synchronize()
    ref = g["decoded"]
    assert wav.dtype == torch.float32 and tuple(wav.shape) == ref.shape
    err = np.abs(wav.cpu().numpy() - ref).max()
    rel = np.linalg.norm(wav.cpu().numpy() - ref) / np.linalg.norm(ref)
    print(f"{os.path.basename(path)}: decoded max abs err {err:.3e}, relative L2 {rel:.3e}")
    assert err < 1e-3 and rel < 1e-4


def test_audiotoken_roundtrip_api(enc_weights):
    from audiotoken_amd import AudioToken, Tokenizers
    tok = AudioToken(Tokenizers.acoustic, device="cuda:0", num_codebooks=8, weights=enc_weights)
    wav = W.synth_waveform(1, 24000, 24000, seed=77)
    codes = tok.encode(wav)                                        # numpy [1, N]
    assert codes.device.type == "cpu" and codes.dtype == torch.int16 and tuple(codes.shape) == (1, 8, 75)
    ref = R.acoustic_encode(enc_weights, torch.from_numpy(wav), 8)
    assert torch.equal(codes, ref)
    audio = tok.decode(codes)
    assert audio.device.type == "cpu" and tuple(audio.shape) == (1, 24000)
    ref_audio = R.acoustic_decode(enc_weights, ref)
    assert (audio - ref_audio).abs().max().item() < 1e-3
    # encode -> decode -> encode is stable in shape and the decoder is deterministic
    assert torch.equal(tok.decode(codes), audio)


def test_encode_batch_files_end_to_end(tmp_path, enc_weights):
    """Files -> chunk/pad/mask -> batches -> HIP encoder -> trimmed .npy (reference core.py:198-289 semantics)."""
    from scipy.io import wavfile
    from audiotoken_amd import AudioToken, Tokenizers
    sr = 24000
    waves = {"one.wav": W.synth_waveform(1, sr * 2 + 5000, sr, seed=1)[0], "two.take2.wav": W.synth_waveform(1, sr + 100, sr, seed=2)[0]}
    for name, x in waves.items():
        wavfile.write(str(tmp_path / name), sr, x)                  # float32 WAV
    tok = AudioToken(Tokenizers.acoustic, device="cuda:0", num_codebooks=4, weights=enc_weights)
    out = tmp_path / "out"
    tok.encode_batch_files(batch_size=3, outdir=out, chunk_size=1, audio_files=[tmp_path / n for n in waves])
    assert sorted(os.listdir(out)) == ["one.npy", "two.npy"]        # stem cut at the first dot
    for name, x in waves.items():
        got = np.load(out / (name.split(".")[0] + ".npy"))
        # reference semantics: every 1 s chunk is segmented, zero padded to 1 s, encoded independently, trimmed to
        # ceil(len_chunk_seconds * 75) tokens, and appended
        pieces = []
        for i in range(0, len(x), sr):
            chunk = x[i:i + sr]
            if len(chunk) < 3200:
                continue
            padded = np.zeros(sr, dtype=np.float32)
            padded[:len(chunk)] = chunk
            ref = R.acoustic_encode(enc_weights, torch.from_numpy(padded)[None], 4)[0].numpy()
            pieces.append(ref[:, :int(np.ceil(len(chunk) / sr * 75))])
        ref_all = np.hstack(pieces)
        assert got.dtype == np.int16 and got.shape == ref_all.shape
        assert np.array_equal(got, ref_all)


def test_persistent_lstm_equals_stepwise(encoders, enc_weights):
    """The whole-sequence persistent LSTM kernel and the one-launch-per-step path run the same MFMA order:
    embeddings and codes must be bit-identical, for full and ragged (B % 32 != 0) groups."""
    enc = encoders[8]
    enc.set_option("lstm_x3", 0)   # the split-bf16 recurrence rounds differently: test_x3_kernels_match_fp32
    for B, N in ((37, 6400), (3, 9600), (64, 3200)):
        wav = torch.from_numpy(W.synth_waveform(B, N, 24000, seed=B)).cuda()
        enc.set_option("persistent_lstm", 1)
        c1, e1 = enc(wav, None, return_embeddings=True)
        assert enc.last_status() == 0, "persistent LSTM: a bounded wait gave up"
        enc.set_option("persistent_lstm", 0)
        c0, e0 = enc(wav, None, return_embeddings=True)
        enc.set_option("persistent_lstm", 1)
        assert torch.equal(e0, e1), (B, (e0 - e1).abs().max().item())
        assert torch.equal(c0, c1)
    enc.set_option("lstm_x3", 1)


def test_pipelined_lstm_equals_layerwise(encoders, enc_weights):
    """lstm_pipe.hip (both LSTM layers in one launch, layer 2 one step behind layer 1, its input gates computed by a third role of
    workgroups instead of the projection GEMM) against the layer-by-layer launches: the same products in the same order, so embeddings,
    codes and decoded waveforms are bit-identical — full, ragged (B % 16 != 0) and maximal (80-clip) groups, and over repeats (the
    hand-off protocol of three chained roles is a race if it is wrong)."""
    from audiotoken_amd.configs import AcousticDecoderConfig
    from audiotoken_amd.decoder import AcousticDecoder
    enc = encoders[8]
    assert enc.get_option("lstm_pipe") == 1
    for B, N in ((5, 9600), (33, 6400), (80, 3200), (16, 24000)):
        wav = torch.from_numpy(W.synth_waveform(B, N, 24000, seed=100 + B)).cuda()
        c1, e1 = enc(wav, None, return_embeddings=True)
        assert enc.last_status() == 0
        enc.set_option("lstm_pipe", 0)
        c0, e0 = enc(wav, None, return_embeddings=True)
        enc.set_option("lstm_pipe", 1)
        assert torch.equal(e0, e1), (B, (e0 - e1).abs().max().item())
        assert torch.equal(c0, c1)
        for _ in range(5):
            c2, e2 = enc(wav, None, return_embeddings=True)
            assert torch.equal(e2, e1) and enc.last_status() == 0
    dec = AcousticDecoder(AcousticDecoderConfig(), device="cuda:0", weights=enc_weights)
    g = torch.Generator().manual_seed(9)
    for B, T in ((3, 25), (64, 75), (21, 40)):
        codes = torch.randint(0, 1024, (B, 8, T), generator=g, dtype=torch.long).cuda()
        w1 = dec(codes).clone()
        assert dec.last_status() == 0
        dec.set_option("lstm_pipe", 0)
        w0 = dec(codes).clone()
        dec.set_option("lstm_pipe", 1)
        assert torch.equal(w0, w1), (B, T, (w0 - w1).abs().max().item())
        for _ in range(5):
            assert torch.equal(dec(codes), w1) and dec.last_status() == 0


def test_fused_stage0_equals_unfused(encoders):
    """seanet_stage0_kernel (conv0 + resblock + strided conv fused) keeps the MFMA / tap order of the separate kernels:
    embeddings and codes must be bit-identical, including the reflect-padded clip start and a ragged last tile."""
    enc = encoders[8]
    enc.set_option("down64_x3", 0)   # the split-bf16 variants round differently: test_x3_kernels_match_fp32 below
    enc.set_option("res128_x3", 0)
    enc.set_option("res64_x3", 0)
    enc.set_option("stage0_x3", 0)
    enc.set_option("down128_x3", 0)
    enc.set_option("down256_x3", 0)
    enc.set_option("res256_x3", 0)
    for opt in ("fused_stage0", "fused_res64", "fused_res128", "fused_down64"):   # (res128_x3 / down64_x3 are off here)
        for B, N in ((3, 6400), (2, 24000 + 320 * 3), (5, 3200 + 640), (2, 9999)):
            wav = torch.from_numpy(W.synth_waveform(B, N, 24000, seed=B + 100)).cuda()
            enc.set_option(opt, 1)
            c1, e1 = enc(wav, None, return_embeddings=True)
            enc.set_option(opt, 0)
            c0, e0 = enc(wav, None, return_embeddings=True)
            enc.set_option(opt, 1)
            assert torch.equal(e0, e1), (opt, B, N, (e0 - e1).abs().max().item())
            assert torch.equal(c0, c1)
    enc.set_option("down64_x3", 1)
    enc.set_option("res128_x3", 1)
    enc.set_option("res64_x3", 1)
    enc.set_option("stage0_x3", 1)
    enc.set_option("down128_x3", 1)
    enc.set_option("down256_x3", 1)
    enc.set_option("res256_x3", 1)


_ORACLE_CACHE = {}


def _oracle_codes(B, N):
    """Oracle ids + top-2 margins of the seeded batch test_x3_kernels_match_fp32 uses (cached per shape: the oracle is CPU work)."""
    if (B, N) not in _ORACLE_CACHE:
        w = W.synth_encodec_weights(seed=0, with_decoder=True)
        wav = torch.from_numpy(W.synth_waveform(B, N, 24000, seed=B + 300))
        _ORACLE_CACHE[(B, N)] = R.acoustic_encode(w, wav, 8, return_margins=True)
    return _ORACLE_CACHE[(B, N)]


@pytest.mark.parametrize("opt", ["down64_x3", "res128_x3", "res64_x3", "stage0_x3", "down128_x3", "down256_x3", "res256_x3", "lstm_x3", "rvq_x3"])
def test_x3_kernels_match_fp32(encoders, opt):
    """seanet_down64x3_kernel / seanet_res128x3_kernel (exact 3-way bf16 splits, six bf16 MFMAs) against the fp32-MFMA kernels:
    a different rounding of the same sums, so embeddings agree to ~1e-6 of their scale rather than bit for bit, and the tokens
    are the same."""
    enc = encoders[8]
    for B, N in ((3, 6400), (2, 24000 + 320 * 3), (5, 3200 + 640), (2, 9999), (40, 24000), (1, 3200), (2, 3200 + 40 * 5)):
        wav = torch.from_numpy(W.synth_waveform(B, N, 24000, seed=B + 300)).cuda()
        enc.set_option(opt, 1)
        c1, e1 = enc(wav, None, return_embeddings=True)
        enc.set_option(opt, 0)
        c0, e0 = enc(wav, None, return_embeddings=True)
        enc.set_option(opt, 1)
        scale = e0.abs().max().item()
        assert (e0 - e1).abs().max().item() <= 2e-5 * scale, (opt, B, N, (e0 - e1).abs().max().item(), scale)
        # both variants: ids equal to the oracle's, or explained by an oracle near-tie at the first differing stage
        ref, margins = _oracle_codes(B, N)
        for name, c in (("x3", c1), ("fp32", c0)):
            P.assert_rvq_equal_or_explained(c, ref, margins, P.RVQ_TIE, f"{opt} {name} B={B} N={N}")


def test_repeated_encodes_are_identical(encoders):
    """The persistent LSTM's inter-workgroup hand-off (flags + write-through stores) must never drop or reuse a step: 25
    encodes of one ragged batch (two 32-clip groups, one partial) give identical tokens and a clean status word."""
    enc = encoders[8]
    wav = torch.from_numpy(W.synth_waveform(41, 24000 * 2 + 320 * 7, 24000, seed=77)).cuda()
    ref = None
    for it in range(25):
        codes = enc(wav, None)
        torch.cuda.synchronize()
        assert enc.last_status() == 0
        if ref is None:
            ref = codes.clone()
        else:
            assert torch.equal(ref, codes), it


def test_subbatch_option_does_not_change_tokens(encoders):
    """The conv stack's sub-batch (workspace bound, shrunk automatically on allocation failure) is invisible in the output."""
    enc = encoders[8]
    wav = torch.from_numpy(W.synth_waveform(11, 24000 + 320 * 5, 24000, seed=91)).cuda()
    ref = enc(wav, None).clone()
    try:
        for sub in (1, 4, 8):
            enc.set_option("subbatch", sub)
            assert torch.equal(enc(wav, None), ref), sub
    finally:
        enc.set_option("subbatch", 256)


def test_fused_decoder_kernels_equal_unfused(enc_weights):
    """seanet_dectail_kernel (transposed conv + block + final conv) and the fused residual blocks keep the accumulation order of
    the GEMM path: decoded waveforms are bit-identical, including the reflect-padded clip start and ragged last tiles."""
    from audiotoken_amd.configs import AcousticDecoderConfig
    from audiotoken_amd.decoder import AcousticDecoder
    dec = AcousticDecoder(config=AcousticDecoderConfig(bandwidth=6), device="cuda:0", weights=enc_weights)
    g = torch.Generator().manual_seed(5)
    for B, T in ((2, 7), (3, 25), (1, 40), (5, 13)):
        codes = torch.randint(0, 1024, (B, 8, T), generator=g, dtype=torch.long).cuda()
        x3 = dec(codes).clone()
        dec.set_option("res128_x3", 0)   # the split-bf16 blocks round differently: compared by tolerance
        dec.set_option("res64_x3", 0)
        dec.set_option("tail_f16x2", 0)  # so does the tail kernel on the fp16 scheme (seanet_dectail_x2.hip)
        ref = dec(codes).clone()
        assert (ref - x3).abs().max().item() <= 2e-5 * ref.abs().max().item(), (B, T, (ref - x3).abs().max().item())
        for opt in ("fused_dectail", "fused_res64", "fused_res128"):
            dec.set_option(opt, 0)
            got = dec(codes)
            dec.set_option(opt, 1)
            assert torch.equal(ref, got), (opt, B, T, (ref - got).abs().max().item())
        dec.set_option("res128_x3", 1)
        dec.set_option("res64_x3", 1)
        dec.set_option("tail_f16x2", 1)
        only_tail = dec(codes)           # the fp16-scheme tail alone against its fp32 twin
        dec.set_option("tail_f16x2", 0)
        fp32_tail = dec(codes)
        dec.set_option("tail_f16x2", 1)
        assert dec.last_status() == 0
        assert (only_tail - fp32_tail).abs().max().item() <= 2e-5 * fp32_tail.abs().max().item(), (B, T, (only_tail - fp32_tail).abs().max().item())
        dec.set_option("dec_chain", 0)   # stage 0's 256-channel block as fp32 GEMMs instead of the split-GEMM chain (seanet_dec256.hip)
        unchained = dec(codes)
        dec.set_option("dec_chain", 1)
        assert (only_tail - unchained).abs().max().item() <= 2e-5 * unchained.abs().max().item(), (B, T, (only_tail - unchained).abs().max().item())


F16X2_OPTIONS = ["chain_f16x2", "ih_f16x2", "res_f16x2", "rvq_f16x2", "fin_f16x2", "lstm_f16x2"]
X3_OPTIONS = ["down64_x3", "res128_x3", "res64_x3", "stage0_x3", "down128_x3", "down256_x3", "res256_x3", "lstm_x3", "rvq_x3"]


def test_all_fp32_and_all_x3_whole_path(encoders):
    """The two settings people actually use — every split-bf16 kernel off ($AUDIOTOKEN_X3_KERNELS=0) and all on (511, the default) — through
    the whole encoder against the oracle: ids equal or explained, embeddings within 1e-3."""
    enc = encoders[8]
    B, N = 6, 24000 + 320 * 11 + 7
    wav = torch.from_numpy(W.synth_waveform(B, N, 24000, seed=808))
    w = W.synth_encodec_weights(seed=0)
    ref, margins = R.acoustic_encode(w, wav, 8, return_margins=True)
    emb_ref = R.seanet_encode(w, wav).permute(0, 2, 1)
    try:
        for value, f16, name in ((0, 1, "all fp32-MFMA kernels"), (1, 1, "all split kernels, two fp16 pieces (default)"),
                                 (1, 0, "all split kernels, three bf16 pieces (the range-overflow fallback)")):
            for opt in X3_OPTIONS:
                enc.set_option(opt, value)
            for opt in F16X2_OPTIONS:
                enc.set_option(opt, f16)
            codes, emb = enc(wav.cuda(), None, return_embeddings=True)
            assert enc.last_status() == 0
            assert (emb.cpu() - emb_ref).abs().max().item() < 1e-3
            P.assert_rvq_equal_or_explained(codes, ref, margins, P.RVQ_TIE, name)
    finally:
        for opt in X3_OPTIONS + F16X2_OPTIONS:
            enc.set_option(opt, 1)


def test_role_split_res128_is_bit_identical(encoders):
    """seanet_res128rs.hip (conv3 waves / tail waves, 32-row tiles, one barrier per tile) performs the products of
    seanet_res128x3_kernel<SchemeF16x2> in the same order per output element: embeddings and codes must be IDENTICAL, at ragged lengths
    (stage-2 lengths that are not multiples of 32, and of 5: the fp32-row output instead of the piece output) and batch sizes."""
    enc = encoders[8]
    try:
        for B, N, seed in ((5, 24000 + 320 * 7, 1), (3, 24000 * 3 + 320 * 5 + 13, 2), (2, 320 * 9, 3), (17, 320 * 40, 4)):
            wav = torch.from_numpy(W.synth_waveform(B, N, 24000, seed=900 + seed)).cuda()
            enc.set_option("res128_rs", 1)
            c1, e1 = enc(wav, None, return_embeddings=True)
            assert enc.last_status() == 0
            enc.set_option("res128_rs", 0)
            c0, e0 = enc(wav, None, return_embeddings=True)
            assert enc.last_status() == 0
            assert torch.equal(e1, e0), f"role-split block differs at B={B} N={N}: max {(e1 - e0).abs().max().item():.3e}"
            assert torch.equal(c1, c0)
    finally:
        enc.set_option("res128_rs", 1)


def test_fused_stage1_is_bit_identical(encoders):
    """seanet_res64down.hip (block role + conv role in one workgroup, the block output kept in LDS, the conv's causal context carried from tile to
    tile) performs the products of seanet_res64x3_kernel + seanet_down64x3_kernel (fp16 scheme) in the same order per output element: embeddings and
    codes must be IDENTICAL — at stage-1 lengths that are not multiples of 64 (partial last tile), below one tile, with several clips per workgroup
    run and with runs that start inside a clip (B * tiles not a multiple of the grid), and the range census of both sites must still be reported."""
    enc = encoders[8]
    try:
        for B, N, seed in ((5, 24000 + 320 * 7, 1), (3, 24000 * 3 + 320 * 5 + 13, 2), (2, 320 * 9, 3), (17, 320 * 40, 4), (1, 240000, 5), (300, 320 * 9, 6)):
            wav = torch.from_numpy(W.synth_waveform(B, N, 24000, seed=950 + seed)).cuda()
            enc.set_option("fused_stage1", 1)
            c1, e1 = enc(wav, None, return_embeddings=True)
            assert enc.last_status() == 0
            live1 = enc.range_report()
            enc.set_option("fused_stage1", 0)
            c0, e0 = enc(wav, None, return_embeddings=True)
            assert enc.last_status() == 0
            live0 = enc.range_report()
            assert torch.equal(e1, e0), f"fused stage 1 differs at B={B} N={N}: max {(e1 - e0).abs().max().item():.3e}"
            assert torch.equal(c1, c0)
            if N % 8 == 0:   # stage-1 length N / 2 divisible by 4: the fused kernel (and seanet_down64x3) ran, else the GEMM path did in both passes
                assert live1["res1"] > 0 and live1["down1"] > 0
                # the census of the fused kernel also covers rows past a clip's end (clamped copies of its last row): never below the pair's
                assert live1["res1"] >= live0["res1"] * 0.999 and live1["down1"] >= live0["down1"] * 0.999
    finally:
        enc.set_option("fused_stage1", 1)


def test_fp16_range_overflow_is_reported_and_recovered(enc_weights):
    """A waveform far outside [-1, 1] (x 3e4) overflows the fp16 range of the two-piece kernels: the status word must say so (bit 1) and
    AcousticEncoder.verified must hand back the tokens of the three-bf16-piece kernels (full fp32 exponent range), equal to the oracle's or explained.
    The fallback is PER BATCH: the options are restored, the next (ordinary) batch runs on f16x2 again with a clean status, and the count is kept."""
    from audiotoken_amd.configs import AcousticEncoderConfig
    from audiotoken_amd.encoder import AcousticEncoder
    enc = AcousticEncoder(AcousticEncoderConfig(bandwidth=6), device="cuda:0", weights=enc_weights)
    wav = torch.from_numpy(W.synth_waveform(3, 24000 + 320 * 3, 24000, seed=23)) * 3e4
    w = W.synth_encodec_weights(seed=0)
    ref, margins = R.acoustic_encode(w, wav, 8, return_margins=True)
    x = wav.cuda()
    codes = enc(x, None)
    assert enc.last_status() & 2, "the range overflow was not reported"
    before = {o: enc.get_option(o) for o in F16X2_OPTIONS}
    codes = enc.verified(codes, x, None)
    assert enc.last_status() == 0 and enc.fallback_batches == 1
    assert {o: enc.get_option(o) for o in F16X2_OPTIONS} == before, "the range fallback must not outlive the batch"
    P.assert_rvq_equal_or_explained(codes, ref, margins, P.RVQ_TIE, "after the range-overflow fallback")
    quiet = (x / 3e4).contiguous()
    c2 = enc(quiet, None)
    assert enc.last_status() == 0 and enc.verified(c2, quiet, None) is c2 and enc.fallback_batches == 1


def test_both_fallbacks_together(enc_weights):
    """The option combination AcousticEncoder.verified can actually produce: the range fallback (every "*_f16x2" option of RANGE_OPTIONS off:
    three bf16 pieces) WHILE the machine fallback is active (persistent_lstm = 0: one launch per LSTM step). Round 2 tested all-on, all-off and
    bf16x3 triples but not this pair. Tokens against the oracle, and against the default path on an ordinary batch."""
    from audiotoken_amd.configs import AcousticEncoderConfig
    from audiotoken_amd.encoder import AcousticEncoder
    enc = AcousticEncoder(AcousticEncoderConfig(bandwidth=6), device="cuda:0", weights=enc_weights)
    wav = torch.from_numpy(W.synth_waveform(5, 24000 + 320 * 2, 24000, seed=29))
    x = wav.cuda()
    default = enc(x, None).clone()
    assert enc.last_status() == 0
    try:
        enc.set_option("persistent_lstm", 0)
        for o in AcousticEncoder.RANGE_OPTIONS:
            enc.set_option(o, 0)
        both = enc(x, None).clone()
        assert enc.last_status() == 0
    finally:
        enc.set_option("persistent_lstm", 1)
        for o in AcousticEncoder.RANGE_OPTIONS:
            enc.set_option(o, 1)
    w = W.synth_encodec_weights(seed=0)
    ref, margins = R.acoustic_encode(w, wav, 8, return_margins=True)
    P.assert_rvq_equal_or_explained(both, ref, margins, P.RVQ_TIE, "range fallback + per-step LSTM")
    P.assert_rvq_equal_or_explained(default, ref, margins, P.RVQ_TIE, "default path, same batch")
    assert torch.equal(enc(x, None), default), "the options were not restored"


def test_forced_lstm_timeout_is_reported_and_recovered(enc_weights):
    """A persistent-LSTM workgroup that gives up (spin limit 0: the first unready poll) must (1) terminate, (2) set the status word,
    and (3) never reach the caller as tokens: AcousticEncoder.verified re-encodes with per-step LSTM launches. Same for the decoder."""
    from audiotoken_amd.configs import AcousticDecoderConfig, AcousticEncoderConfig
    from audiotoken_amd.decoder import AcousticDecoder
    from audiotoken_amd.encoder import AcousticEncoder
    enc = AcousticEncoder(AcousticEncoderConfig(bandwidth=6), device="cuda:0", weights=enc_weights)
    wav = torch.from_numpy(W.synth_waveform(20, 24000 + 640, 24000, seed=17)).cuda()
    good = enc(wav, None).clone()
    assert enc.last_status() == 0
    enc.set_option("lstm_spin_limit", 0)
    bad = enc(wav, None)
    assert enc.last_status() == 1, "a give-up must be visible in the status word"
    fixed = enc.verified(bad, wav, None)          # logs, switches to per-step launches, re-encodes
    assert enc.last_status() == 0
    assert torch.equal(fixed, good)
    # the decoder runs the same kernel and has the same guard
    dec = AcousticDecoder(config=AcousticDecoderConfig(bandwidth=6), device="cuda:0", weights=enc_weights)
    codes = good[:, :, :40].long().contiguous()
    ref = dec(codes).clone()
    assert dec.last_status() == 0
    dec.set_option("lstm_spin_limit", 0)
    out = dec(codes)
    assert dec.last_status() == 1
    out = dec.verified(out, codes)
    assert dec.last_status() == 0
    assert (out - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()   # per-step fp32 LSTM vs the split-bf16 persistent one


def test_audiotoken_encode_recovers_from_timeout(enc_weights):
    """The same through the facade: AudioToken.encode must return valid tokens although the first launch times out."""
    from audiotoken_amd import AudioToken, Tokenizers
    tok = AudioToken(Tokenizers.acoustic, device="cuda:0", num_codebooks=8, weights=enc_weights)
    wav = W.synth_waveform(1, 24000, 24000, seed=3)
    ref = tok.encode(wav)
    tok.encoder.set_option("lstm_spin_limit", 0)
    # a single clip is one 16-clip group of 16 workgroups: the first step's hand-off polls are not all ready at once
    got = tok.encode(wav)
    assert torch.equal(got, ref)
