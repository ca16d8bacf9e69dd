"""GPU, SURVEY.md §8(f) N2 end to end: checkpoint FILES in the reference's on-disk layouts -> the loaders -> at_*_set_tensor / finalize -> HIP
encode, against the same weights passed as an in-memory dict and against the CPU oracle. The reference builds its encoders from files
(audiotoken/encoder.py:38,72,84-85,132,156-161; configs.py:112-134; utils.py:331-339); no pretrained weights exist offline, so the synthetic
weights are written in those layouts: encodec ``.th`` (torch-saved state dict with weight_g / weight_v pairs and codebook buffers), a SHARDED
``model.safetensors`` directory + ``config.json`` + the VectorQuantize ``.pkl``, a HuBERT ``save_pretrained`` directory + a joblib k-means."""
import json
import os

import numpy as np
import pytest
import torch

from audiotoken_amd import weights as W
from tests import parity as P

pytestmark = pytest.mark.gpu


def _write_sharded_safetensors(d, sd, n_shards, prefix=""):
    """{name: array} -> model-0000i-of-0000n.safetensors + model.safetensors.index.json, keys optionally under a task-head prefix."""
    from safetensors.torch import save_file
    os.makedirs(d, exist_ok=True)
    names = sorted(sd)
    weight_map = {}
    for i in range(n_shards):
        part = {prefix + k: torch.from_numpy(np.ascontiguousarray(sd[k])) for k in names[i::n_shards]}
        fn = f"model-{i + 1:05d}-of-{n_shards:05d}.safetensors"
        save_file(part, os.path.join(d, fn))
        weight_map.update({k: fn for k in part})
    with open(os.path.join(d, "model.safetensors.index.json"), "w") as fh:
        json.dump({"metadata": {}, "weight_map": weight_map}, fh)


def test_acoustic_from_th_file(cuda_device, tmp_path):
    from audiotoken_amd import AudioToken, Tokenizers
    from oracle import encodec_ref as R
    w = W.synth_encodec_weights(seed=3, with_decoder=True, n_codebooks=32)        # the 24 kHz checkpoint ships decoder + 32 codebooks
    sd = {k: torch.from_numpy(v) for k, v in w.items()}
    for q in range(32):                                                              # buffers of encodec's EuclideanCodebook
        sd[f"quantizer.vq.layers.{q}._codebook.inited"] = torch.tensor([1.0])
        sd[f"quantizer.vq.layers.{q}._codebook.cluster_size"] = torch.ones(1024)
        sd[f"quantizer.vq.layers.{q}._codebook.embed_avg"] = sd[f"quantizer.vq.layers.{q}._codebook.embed"].clone()
    path = tmp_path / "encodec_24khz-d7cc33bc.th"
    torch.save(sd, path)
    wav = W.synth_waveform(1, 48000, 24000, seed=31)
    from_file = AudioToken(Tokenizers.acoustic, device="cuda:0", num_codebooks=8, weights=str(path)).encode(wav)
    from_dict = AudioToken(Tokenizers.acoustic, device="cuda:0", num_codebooks=8, weights=w).encode(wav)
    assert from_file.dtype == torch.int16 and tuple(from_file.shape) == (1, 8, 150)
    assert torch.equal(from_file, from_dict), "file-loaded and in-memory weights give different tokens"
    ref, margins = R.acoustic_encode(w, torch.from_numpy(wav), 8, return_margins=True)
    P.assert_rvq_equal_or_explained(from_file, ref, margins, P.RVQ_TIE, "[N2] acoustic from encodec .th")


def test_semantic_m_from_sharded_safetensors_dir(cuda_device, tmp_path):
    from audiotoken_amd import AudioToken, Tokenizers
    from audiotoken_amd.encoder import W2VBERT_ARCH
    from oracle import w2vbert_ref as R
    nl = 3
    w = W.synth_w2vbert_weights(n_layers=nl, seed=5, with_vq=True)
    d = str(tmp_path / "w2vbert2_l21")
    model_sd = {k: v for k, v in w.items() if not k.startswith("vq.")}
    model_sd["masked_spec_embed"] = np.zeros(1024, np.float32)                      # present in the real checkpoint, unused in eval
    _write_sharded_safetensors(d, model_sd, 3)
    with open(os.path.join(d, "config.json"), "w") as fh:
        json.dump(dict(W2VBERT_ARCH, model_type="wav2vec2-bert", num_hidden_layers=nl), fh)
    vq = tmp_path / "vq.pkl"                                                        # VectorQuantize state dict (reference utils.py:331-339)
    torch.save({"_codebook.embed": torch.from_numpy(w["vq._codebook.embed"]), "_codebook.cluster_size": torch.zeros(1, 2048),
                "_codebook.embed_avg": torch.zeros(1, 2048, 1024), "_codebook.initted": torch.tensor([True])}, vq)
    wav = W.synth_waveform(1, 64000, 16000, seed=32)

    def tok(**kw):
        t = AudioToken(Tokenizers.semantic_m, device="cuda:0", **kw)
        t.model_config.output_layer = nl                                            # the fixture holds 3 conformer layers, not 21
        return t.encode(wav)
    from_file = tok(weights=d, quantizer=str(vq))
    from_dict = tok(weights=w)
    assert from_file.dtype == torch.int16 and tuple(from_file.shape) == (1, 1, 200)
    assert torch.equal(from_file, from_dict), "file-loaded and in-memory weights give different tokens"
    wt = {k: torch.from_numpy(v) for k, v in w.items()}
    x = torch.from_numpy(wav)
    ref, margins = R.semantic_m_encode(wt, x, torch.ones_like(x), 2, nl, return_margins=True)
    _, am = R.processor(x, torch.ones_like(x), 2)
    P.assert_tokens_equal_or_explained(from_file, ref, margins, P.VQ_TIE, "[N2] semantic_m from sharded safetensors + VQ .pkl", am.bool().unsqueeze(1))


def test_semantic_s_from_hubert_dir_and_joblib(cuda_device, tmp_path):
    import joblib
    from sklearn.cluster import KMeans
    from audiotoken_amd import AudioToken, Tokenizers
    from audiotoken_amd.hubert import HUBERT_ARCH, hubert_processor
    from oracle import hubert_ref as R
    nl = 3
    w = W.synth_hubert_weights(nl, 6, True)
    d = str(tmp_path / "mhubert-base")
    _write_sharded_safetensors(d, {k: v for k, v in w.items() if not k.startswith("kmeans.")}, 2, prefix="hubert.")
    with open(os.path.join(d, "config.json"), "w") as fh:
        json.dump(dict(HUBERT_ARCH, model_type="hubert", num_hidden_layers=nl), fh)
    km = KMeans(n_clusters=1000)
    km.cluster_centers_ = w["kmeans.cluster_centers_"].astype(np.float64)         # sklearn stores float64 (reference encoder.py:84-85)
    joblib.dump(km, tmp_path / "mhubert_base_vp_en_es_fr_it3_L11_km1000.bin")
    wav = W.synth_waveform(1, 48000, 16000, seed=33)

    def tok(**kw):
        t = AudioToken(Tokenizers.semantic_s, device="cuda:0", **kw)
        t.model_config.output_layer = nl
        return t.encode(wav)
    from_file = tok(weights=d, quantizer=str(tmp_path / "mhubert_base_vp_en_es_fr_it3_L11_km1000.bin"))
    from_dict = tok(weights=w)
    assert from_file.dtype == torch.int16 and tuple(from_file.shape) == (1, 1, 149)
    assert torch.equal(from_file, from_dict), "file-loaded and in-memory weights give different tokens"
    norm = hubert_processor(torch.from_numpy(wav))
    ref, margins = R.semantic_s_encode(w, norm, torch.ones_like(norm), nl, return_margins=True)
    P.assert_tokens_equal_or_explained(from_file, ref, margins, P.VQ_TIE, "[N2] semantic_s from HuBERT directory + joblib k-means")
