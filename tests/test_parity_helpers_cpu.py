"""The equal-or-explained bookkeeping itself (tests/parity.py) on hand-made cases, and the oracles' margin outputs."""
import numpy as np
import pytest
import torch

from tests import parity as P


def test_token_mismatch_bookkeeping():
    ref = torch.tensor([[[1, 2, 3, 4]]])
    got = torch.tensor([[[1, 9, 3, 7]]])
    margins = torch.tensor([[[5.0, 1e-5, 5.0, 0.5]]])
    assert P.explain_token_mismatches(got, ref, margins, 1e-3) == (2, 1, 0.5)
    valid = torch.tensor([[[1, 1, 1, 0]]])
    assert P.explain_token_mismatches(got, ref, margins, 1e-3, valid)[:2] == (1, 0)
    with pytest.raises(AssertionError):
        P.assert_tokens_equal_or_explained(got, ref, margins, 1e-3, "case")
    assert P.assert_tokens_equal_or_explained(got, ref, margins, 1e-3, "case", valid) == 1


def test_rvq_first_stage_rule():
    ref = torch.zeros(1, 3, 4, dtype=torch.long)
    got = ref.clone()
    got[0, 1, 2] = 5; got[0, 2, 2] = 6          # frame 2: first difference at stage 1 (near-tie), stage 2 follows
    got[0, 2, 3] = 7                            # frame 3: first difference at stage 2 with a wide margin -> unexplained
    margins = torch.full((1, 3, 4), 4.0)
    margins[0, 1, 2] = 1e-6
    assert P.explain_rvq_mismatches(got, ref, margins, 1e-3) == (3, 2, 1)
    with pytest.raises(AssertionError):
        P.assert_rvq_equal_or_explained(got, ref, margins, 1e-3, "case")


def test_oracle_margins_are_consistent():
    from audiotoken_amd import weights as W
    from oracle import encodec_ref as R
    w = W.synth_encodec_weights(seed=3, with_decoder=False)
    wav = torch.from_numpy(W.synth_waveform(2, 3200, 24000, seed=5))
    codes, margins = R.acoustic_encode(w, wav, 4, return_margins=True)
    assert torch.equal(codes, R.acoustic_encode(w, wav, 4))
    assert tuple(margins.shape) == tuple(codes.shape) and bool((margins >= 0).all())
