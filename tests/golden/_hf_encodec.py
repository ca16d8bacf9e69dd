"""Helper for make_golden.py: load an `encodec`-keyed weight dict into HF EncodecModel (build container only)."""
import re
import torch


def to_hf_key(k: str):
    """Original encodec key -> HF transformers 5.15 EncodecModel state-dict key."""
    m = re.match(r"quantizer\.vq\.layers\.(\d+)\._codebook\.embed", k)
    if m:
        return f"quantizer.layers.{m.group(1)}.codebook.embed"
    k2 = k.replace(".model.", ".layers.")
    k2 = k2.replace(".conv.conv.", ".conv.").replace(".convtr.convtr.", ".conv.")
    k2 = k2.replace(".conv.weight_g", ".conv.parametrizations.weight.original0")
    k2 = k2.replace(".conv.weight_v", ".conv.parametrizations.weight.original1")
    return k2


def build_hf_encodec(weights):
    from transformers import EncodecModel, EncodecConfig
    model = EncodecModel(EncodecConfig()).eval()
    sd = model.state_dict()
    used = set()
    for k, v in weights.items():
        hk = to_hf_key(k)
        assert hk in sd, (k, hk)
        assert tuple(sd[hk].shape) == tuple(v.shape), (k, hk, sd[hk].shape, v.shape)
        sd[hk] = torch.from_numpy(v.copy())
        used.add(hk)
    missing = [k for k in sd if k not in used and not any(s in k for s in ("inited", "cluster_size", "embed_avg"))]
    assert not missing, missing[:5]
    model.load_state_dict(sd)
    return model
