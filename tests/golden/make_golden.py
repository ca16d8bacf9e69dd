"""Generate the committed golden vectors (run in the BUILD container only: needs HF `transformers`
and, for the semantic fixtures, read access to /root/reference).

    python tests/golden/make_golden.py [encodec] [fbank] [attention] [conformer] [harness]

Outputs small .npz files next to this script. Inputs are regenerated from the in-repo counter-based PRNG
(audiotoken_amd/prng.py), so fixtures hold only seeds/shapes and the expected outputs.

* encodec_*.npz — produced by HF ``transformers`` 5.15.0 ``EncodecModel`` (the reference's dependency `encodec`
  is not installed; HF's model is the same published architecture) loaded with the
  synthetic weights of ``audiotoken_amd.weights.synth_encodec_weights(seed)``.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from audiotoken_amd import weights as W  # noqa: E402


def edges(x: torch.Tensor, n: int = 16) -> np.ndarray:
    """First and last n time steps of a [B, C, T] activation (padding effects live at the edges)."""
    T = x.shape[-1]
    if T <= 2 * n:
        return x.numpy().copy()
    return torch.cat([x[..., :n], x[..., -n:]], dim=-1).numpy().copy()


def make_encodec():
    from _hf_encodec import build_hf_encodec
    seed = 0
    w = W.synth_encodec_weights(seed=seed)
    model = build_hf_encodec(w)
    cases = [("a", 2, 8000, 8), ("b", 3, 7777, 4), ("c", 1, 24000, 16), ("d", 2, 3201, 2)]
    for tag, B, N, n_q in cases:
        wav = torch.from_numpy(W.synth_waveform(B, N, 24000, seed=1234))
        bw = {2: 1.5, 4: 3.0, 8: 6.0, 16: 12.0}[n_q]
        with torch.no_grad():
            h = wav.unsqueeze(1)
            stage_edges = {}
            for i, layer in enumerate(model.encoder.layers):
                h = layer(h)
                if i in (0, 1, 3, 4, 6, 7, 9, 10, 12, 13):
                    stage_edges[f"stage{i}"] = edges(h)
            emb = h
            codes = model.quantizer.encode(emb, bw)          # [n_q, B, T]
            toks = codes.transpose(0, 1).to(torch.int16)     # reference encoder.py:54
            dec = model.decoder(model.quantizer.decode(codes))  # [B, 1, 320*T]
            wav_out = dec.reshape(-1).to(torch.float32).unsqueeze(0)  # reference decoder.py:76
        np.savez_compressed(
            os.path.join(HERE, f"encodec_{tag}.npz"),
            weight_seed=seed, wave_seed=1234, B=B, N=N, n_q=n_q,
            emb=emb.numpy(), tokens=toks.numpy(), decoded=wav_out.numpy().astype(np.float32), **stage_edges)
        print("encodec", tag, emb.shape, toks.shape, wav_out.shape)


def make_fbank():
    """Reference-authored front-end: /root/reference/audiotoken/processors.py via _ref_loader stubs."""
    from _ref_loader import load_reference_modules
    mods = load_reference_modules()
    proc = mods["processors"].Wav2VecBertProcessor(feature_size=80, num_mel_bins=80, sampling_rate=16000, stride=2,
                                                   padding_value=1)
    cases = {
        # tag: (B, N, valid lengths (None = full), pad_to_multiple_of, noise level added on top of the bench waveform)
        "a": (3, 16000 * 2 + 123, [None, 20000, 9000], 2, 0.0),
        "b": (2, 16000, [None, 3200], 2, 0.0),
        "c": (2, 16000 * 3 + 77, [None, 30000], 10, 0.02),
    }
    for tag, (B, N, lens, mult, noise) in cases.items():
        wave = W.synth_waveform(B, N, 16000, seed=77)
        if noise:
            from audiotoken_amd import prng
            wave = np.clip(wave + noise * prng.irwin_hall("fbank.noise", (B, N), 1.0, 5), -1, 1).astype(np.float32)
        mask = np.ones((B, N), dtype=np.float32)
        for i, ln in enumerate(lens):
            if ln is not None:
                mask[i, ln:] = 0
                wave[i, ln:] = 0  # datasets.py:102: pad_token = 0
        with torch.no_grad():
            out = proc(torch.from_numpy(wave), torch.from_numpy(mask), mult)
        np.savez_compressed(os.path.join(HERE, f"fbank_{tag}.npz"), wave=wave, mask=mask, pad_to_multiple_of=mult,
                            input_features=out["input_features"].numpy(), attention_mask=out["attention_mask"].numpy())
        print("fbank", tag, out["input_features"].shape, out["attention_mask"].sum(1))


def _hf_w2vbert(n_layers, w):
    from transformers import Wav2Vec2BertConfig, Wav2Vec2BertModel
    from transformers.models.wav2vec2_bert.modeling_wav2vec2_bert import Wav2Vec2BertSelfAttention
    from _ref_loader import load_reference_modules
    mods = load_reference_modules()
    Wav2Vec2BertSelfAttention.forward = mods["modeling_wav2vec2_bert"].forward   # reference encoder.py:14-15
    model = Wav2Vec2BertModel(Wav2Vec2BertConfig(num_hidden_layers=n_layers)).eval()
    sd = model.state_dict()
    for k, v in w.items():
        if k.startswith("vq."):
            continue
        assert k in sd and tuple(sd[k].shape) == tuple(v.shape), k
        sd[k] = torch.from_numpy(v.copy())
    model.load_state_dict(sd)
    return model


def make_attention():
    """The reference's SDPA rel-pos attention (modeling_wav2vec2_bert.py:20-80) bound to HF's attention module."""
    from audiotoken_amd import prng
    w = W.synth_w2vbert_weights(n_layers=1, seed=3, with_vq=False)
    model = _hf_w2vbert(1, w)
    attn = model.encoder.layers[0].self_attn
    B, T = 2, 90
    x = torch.from_numpy(prng.irwin_hall("attn.x", (B, T, 1024), 1.0, 9))
    mask = torch.ones(B, T)
    mask[1, 61:] = 0
    add = (1.0 - mask[:, None, None, :]) * torch.finfo(torch.float32).min
    add = add.expand(B, 1, T, T)
    with torch.no_grad():
        out = attn(x, attention_mask=add.clone())[0]
    np.savez_compressed(os.path.join(HERE, "attention_a.npz"), weight_seed=3, x_seed=9, B=B, T=T, mask=mask.numpy(),
                        out=out.numpy())
    print("attention", out.shape)


def make_conformer():
    """HF Wav2Vec2BertModel (3 layers, synthetic weights) + the reference attention patch: hidden_states."""
    from _ref_loader import load_reference_modules
    mods = load_reference_modules()
    n_layers = 3
    w = W.synth_w2vbert_weights(n_layers=n_layers, seed=5, with_vq=True)
    model = _hf_w2vbert(n_layers, w)
    proc = mods["processors"].Wav2VecBertProcessor(feature_size=80, num_mel_bins=80, sampling_rate=16000, stride=2, padding_value=1)
    B, N = 2, 16000 + 400
    wave = W.synth_waveform(B, N, 16000, seed=21)
    mask = np.ones((B, N), dtype=np.float32)
    mask[1, 9000:] = 0
    wave[1, 9000:] = 0
    with torch.no_grad():
        po = proc(torch.from_numpy(wave), torch.from_numpy(mask), 2)
        hs = model(po["input_features"], attention_mask=po["attention_mask"], output_hidden_states=True).hidden_states
        e = torch.nn.functional.layer_norm(hs[n_layers], (1024,))          # reference encoder.py:138-143,176
        embed = torch.from_numpy(w["vq._codebook.embed"][0])
        idx = torch.cdist(e.reshape(-1, 1024), embed).argmin(-1).reshape(B, -1)   # L2-nearest code (cdist cross-check)
    np.savez_compressed(os.path.join(HERE, "conformer_a.npz"), weight_seed=5, wave_seed=21, n_layers=n_layers, B=B, N=N,
                        mask=mask, attention_mask=po["attention_mask"].numpy(),
                        hs0=hs[0].numpy(), hs1=hs[1].numpy(), hs_last=hs[n_layers].numpy(), tokens_cdist=idx.numpy().astype(np.int16))
    print("conformer", hs[n_layers].shape, idx.shape)


def make_harness():
    """Reference-authored harness: datasets.py::_iter_chunk and utils.py::save_audio_tokens via stubs."""
    import tempfile
    from _ref_loader import load_reference_modules
    mods = load_reference_modules()
    DS = mods["datasets"].AudioBatchDataset
    out = {}
    cases = [("a", 69100, 16000, 2, 50), ("b", 240000, 24000, 10, 75), ("c", 24000 * 3 + 1, 24000, 3, 75),
             ("d", 3199, 16000, 1, 50), ("e", 16000 * 5 + 3300, 16000, 5, 50)]
    for tag, length, sr, chunk, rate in cases:
        ds = object.__new__(DS)   # no feeder process
        ds.sample_rate, ds.model_token_rate, ds.transform, ds.pad_token = sr, rate, None, 0
        ds.chunk_size, ds.segment_length, ds.stride = chunk, chunk * sr, chunk * sr
        wave = torch.from_numpy(W.synth_waveform(1, length, sr, seed=3))
        rows = []
        sums = []
        for seg, mask, cfg in ds._iter_chunk(wave, f"x/y/clip_{tag}.v2.wav"):
            rows.append([cfg.start_idx, cfg.end_idx, int(mask.sum().item()), cfg.length_tokens, seg.shape[0]])
            sums.append(float(seg.double().sum().item()))
        out[f"seg_{tag}"] = np.array(rows, dtype=np.int64).reshape(-1, 5)
        out[f"sum_{tag}"] = np.array(sums)
        out[f"cfg_{tag}"] = np.array([length, sr, chunk, rate], dtype=np.int64)
    # save_audio_tokens: trim + append
    AC = mods["configs"].AudioConfig
    save = mods["utils"].save_audio_tokens
    with tempfile.TemporaryDirectory() as d:
        ptr = AC(file_name="x/y/clip.v2.wav", length_seconds=1.0, model_token_rate=50)
        t1 = torch.arange(2 * 60, dtype=torch.int16).reshape(2, 60)
        save(t1, ptr, d)
        first = np.load(os.path.join(d, "clip.npy"))
        save(t1 + 1000, ptr, d)
        second = np.load(os.path.join(d, "clip.npy"))
        out["save_first"], out["save_second"] = first, second
        out["save_files"] = np.array(sorted(os.listdir(d)))
    np.savez_compressed(os.path.join(HERE, "harness_a.npz"), **out)
    print("harness", {k: v.shape for k, v in out.items()})


def make_hubert():
    """HF HubertModel(HubertConfig()) (= HuBERT-base architecture) with synthetic weights, 3 layers."""
    from transformers import HubertConfig, HubertModel
    n_layers = 3
    w = W.synth_hubert_weights(n_layers=n_layers, seed=7, with_kmeans=True)
    model = HubertModel(HubertConfig(num_hidden_layers=n_layers)).eval()
    sd = model.state_dict()
    for k, v in w.items():
        if k.startswith("kmeans."):
            continue
        hk = k.replace("conv.weight_g", "conv.parametrizations.weight.original0").replace("conv.weight_v", "conv.parametrizations.weight.original1")
        assert hk in sd and tuple(sd[hk].shape) == tuple(v.shape), (k, hk)
        sd[hk] = torch.from_numpy(v.copy())
    model.load_state_dict(sd)
    B, N = 2, 16000
    wave = W.synth_waveform(B, N, 16000, seed=41)
    mask = np.ones((B, N), dtype=np.float32)
    mask[1, 9000:] = 0
    wave[1, 9000:] = 0
    from transformers import Wav2Vec2FeatureExtractor
    fe = Wav2Vec2FeatureExtractor(feature_size=1, sampling_rate=16000, padding_value=0.0, do_normalize=True, return_attention_mask=False)
    norm = np.stack([fe(wave[i], sampling_rate=16000, return_tensors="np").input_values[0] for i in range(B)])
    with torch.no_grad():
        out = model(torch.from_numpy(norm), attention_mask=torch.from_numpy(mask), output_hidden_states=True)
        hs = out.hidden_states
        e = torch.nn.functional.layer_norm(hs[n_layers], (768,))
        d = torch.cdist(e, torch.from_numpy(w["kmeans.cluster_centers_"]))
        toks = torch.argmin(d, dim=-1, keepdim=True).transpose(1, 2).to(torch.int16)
    np.savez_compressed(os.path.join(HERE, "hubert_a.npz"), weight_seed=7, wave_seed=41, n_layers=n_layers, B=B, N=N, mask=mask,
                        normalized=norm.astype(np.float32), hs0=hs[0].numpy(), hs1=hs[1].numpy(), hs_last=hs[n_layers].numpy(),
                        tokens=toks.numpy())
    print("hubert", hs[0].shape, toks.shape)


if __name__ == "__main__":
    which = sys.argv[1:] or ["encodec"]
    torch.manual_seed(0)
    for name in which:
        globals()[f"make_{name}"]()
