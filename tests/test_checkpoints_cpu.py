"""N2: real checkpoint LAYOUTS through the loaders — directories written by HF ``save_pretrained`` from randomly initialised models of the
reference's architectures (reference audiotoken/encoder.py:72,132; configs.py:114-134), a torch-saved encodec ``.th`` — checked against the
library's own list of required tensors (at_required_tensors). No device needed."""
import json
import os

import numpy as np
import pytest
import torch

from audiotoken_amd import _cabi
from audiotoken_amd import weights as W


@pytest.fixture(scope="module")
def w2vbert_dir(tmp_path_factory):
    """Wav2Vec2BertModel with the w2v-bert-2.0 dimensions and 3 layers, saved SHARDED (the 21-layer original is 2.3 GB)."""
    from transformers import Wav2Vec2BertConfig, Wav2Vec2BertModel
    torch.manual_seed(0)
    cfg = Wav2Vec2BertConfig(num_hidden_layers=3)
    model = Wav2Vec2BertModel(cfg).eval()
    d = tmp_path_factory.mktemp("w2vbert2_l3")
    model.save_pretrained(str(d), max_shard_size="150MB")
    return str(d), model


def test_w2vbert_save_pretrained_layout(w2vbert_dir, tmp_path):
    from audiotoken_amd.encoder import load_w2vbert_checkpoint
    d, model = w2vbert_dir
    assert os.path.exists(os.path.join(d, "model.safetensors.index.json")), "the fixture must exercise the sharded layout"
    torch.save({"_codebook.embed": torch.randn(1, 2048, 1024), "_codebook.cluster_size": torch.zeros(1, 2048), "_codebook.embed_avg": torch.zeros(1, 2048, 1024),
                "_codebook.initted": torch.tensor([True])}, tmp_path / "vq.pkl")
    w = load_w2vbert_checkpoint(d, str(tmp_path / "vq.pkl"))
    # the front-end tables come from frontend_tables(), everything else from the checkpoint
    from audiotoken_amd.encoder import frontend_tables
    have = dict(frontend_tables(), **w)
    assert W.missing_tensors("w2vbert", 3, True, have) == []
    assert W.missing_tensors("w2vbert", 2, True, have) == []          # output_layer < checkpoint depth: the extra layer is simply unused
    assert W.missing_tensors("w2vbert", 4, False, have) != []         # and a deeper request is reported, not silently short
    sd = model.state_dict()
    k = "encoder.layers.1.self_attn.distance_embedding.weight"
    assert np.array_equal(w[k], sd[k].numpy())
    k = "encoder.layers.2.conv_module.depthwise_conv.weight"
    assert tuple(w[k].shape) == (1024, 1, 31) and np.array_equal(w[k], sd[k].numpy())


def test_w2vbert_prefixed_and_wrong_architecture(w2vbert_dir, tmp_path):
    """Keys under a task-head prefix load the same; a config.json of another architecture is refused with a message."""
    from safetensors.torch import save_file
    from audiotoken_amd.encoder import load_w2vbert_checkpoint
    d, model = w2vbert_dir
    sd = {("wav2vec2_bert." + k): v.contiguous() for k, v in model.state_dict().items() if k.startswith(("feature_projection.", "encoder.layers.0."))}
    save_file(sd, str(tmp_path / "model.safetensors"))
    w = load_w2vbert_checkpoint(str(tmp_path), None)
    assert "encoder.layers.0.ffn1.intermediate_dense.weight" in w and not any(k.startswith("wav2vec2_bert.") for k in w)
    cfg = json.load(open(os.path.join(d, "config.json")))
    cfg["hidden_size"] = 768
    json.dump(cfg, open(tmp_path / "config.json", "w"))
    with pytest.raises(ValueError, match="hidden_size"):
        load_w2vbert_checkpoint(str(tmp_path), None)


def test_hubert_save_pretrained_layout(tmp_path):
    """HubertModel(HubertConfig()) = HuBERT-base, 12 layers; the tokenizer consumes hidden state 11 (reference encoder.py:94-98)."""
    import joblib
    from sklearn.cluster import KMeans
    from transformers import HubertConfig, HubertModel
    from audiotoken_amd.hubert import fold_hubert_weights, load_hubert_checkpoint
    torch.manual_seed(1)
    model = HubertModel(HubertConfig()).eval()
    model.save_pretrained(str(tmp_path / "hubert"))
    km = KMeans(n_clusters=1000)
    km.cluster_centers_ = np.random.default_rng(0).standard_normal((1000, 768))
    joblib.dump(km, tmp_path / "km.bin")
    sd = load_hubert_checkpoint(str(tmp_path / "hubert"), str(tmp_path / "km.bin"))
    folded = fold_hubert_weights(sd, 11)
    assert W.missing_tensors("hubert", 11, True, folded) == []
    assert not any(k.startswith("encoder.layers.11.") for k in folded), "layer 12 is dead compute and must be dropped"
    # the positional conv's weight-norm parametrisation is folded exactly as torch applies it
    ref = model.encoder.pos_conv_embed.conv.weight.detach().numpy()
    assert np.allclose(folded["encoder.pos_conv_embed.conv.weight"], ref, atol=1e-7)
    assert folded["kmeans.cluster_centers_"].dtype == np.float32


def test_encodec_th_layout(tmp_path):
    """The PyPI encodec checkpoint is a torch-saved state dict with weight_g / weight_v pairs and codebook buffers."""
    from audiotoken_amd.encoder import fold_encodec_weights, load_encodec_checkpoint
    w = W.synth_encodec_weights(seed=4, with_decoder=True, n_codebooks=32)      # the 24 kHz model ships 32 codebooks
    sd = {k: torch.from_numpy(v) for k, v in w.items()}
    for q in range(32):
        sd[f"quantizer.vq.layers.{q}._codebook.inited"] = torch.tensor([1.0])
        sd[f"quantizer.vq.layers.{q}._codebook.cluster_size"] = torch.ones(1024)
        sd[f"quantizer.vq.layers.{q}._codebook.embed_avg"] = sd[f"quantizer.vq.layers.{q}._codebook.embed"].clone()
    torch.save(sd, tmp_path / "encodec_24khz-d7cc33bc.th")
    folded = fold_encodec_weights(load_encodec_checkpoint(str(tmp_path / "encodec_24khz-d7cc33bc.th")))
    assert W.missing_tensors("encodec", 32, True, folded) == []
    assert W.missing_tensors("encodec", 8, False, folded) == []


def test_required_tensor_lists_are_well_formed():
    for model, n in (("encodec", 16), ("w2vbert", 19), ("hubert", 11)):
        r = _cabi.required_tensors(model, n, True)
        assert len(r) > 20 and all(len(s) >= 1 and all(d > 0 for d in s) for s in r.values())
    with pytest.raises(_cabi.HipLibraryError):
        _cabi.required_tensors("nope", 1, True)
