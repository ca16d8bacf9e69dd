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


if __name__ == "__main__":
    which = sys.argv[1:] or ["encodec"]
    torch.manual_seed(0)
    for name in which:
        globals()[f"make_{name}"]()
