"""The parity bar for token ids: equal to the oracle's, or EXPLAINED — the oracle's own top-2 distance margin at the (first) differing
choice is below a near-tie bound, i.e. the two candidates are closer than the fp32 rounding noise of the embedding that any two fp32
implementations (two BLAS builds of the reference included) differ by. Nothing else is accepted; the counts are printed."""
import torch

# near-tie bounds, in the units of the oracle's distance: EnCodec's -(|x|^2 - 2 x.e + |e|^2) at embedding scale ~10 (squared distances
# ~100-400), cdist (Euclidean) at ~40 for the two semantic tokenizers. 1e-3 is the contract's tolerance for float intermediates.
RVQ_TIE = 1e-3
VQ_TIE = 1e-3


def explain_token_mismatches(got: torch.Tensor, ref: torch.Tensor, margins: torch.Tensor, tie: float, valid: torch.Tensor = None):
    """got / ref / margins [B, 1, T] (one choice per position). Returns (n_differing, n_unexplained, largest margin among mismatches)."""
    got, ref = got.cpu().long(), ref.cpu().long()
    mism = got != ref
    if valid is not None:
        mism = mism & valid.cpu().bool()
    n = int(mism.sum())
    if n == 0:
        return 0, 0, 0.0
    m = margins.cpu()[mism]
    return n, int((m >= tie).sum()), float(m.max())


def explain_rvq_mismatches(got: torch.Tensor, ref: torch.Tensor, margins: torch.Tensor, tie: float):
    """got / ref / margins [B, n_q, T]: residual VQ — once a stage differs the later stages quantise a different residual, so a
    frame is explained by the margin at its FIRST differing stage. Returns (differing ids, differing frames, unexplained frames)."""
    got, ref = got.cpu().long(), ref.cpu().long()
    mism = got != ref
    frames = mism.any(dim=1)                                   # [B, T]
    if not bool(frames.any()):
        return 0, 0, 0
    first = mism.float().argmax(dim=1)                         # first differing stage per frame
    m0 = margins.cpu().gather(1, first.unsqueeze(1)).squeeze(1)
    unexplained = frames & (m0 >= tie)
    return int(mism.sum()), int(frames.sum()), int(unexplained.sum())


def assert_tokens_equal_or_explained(got, ref, margins, tie, what, valid=None):
    n, bad, worst = explain_token_mismatches(got, ref, margins, tie, valid)
    total = int(valid.sum()) if valid is not None else got.numel()
    print(f"{what}: {n} of {total} ids differ from the oracle, {bad} not explained by an oracle top-2 margin < {tie:g} (largest margin among them {worst:.2e})")
    assert bad == 0, f"{what}: {bad} token ids differ from the oracle without a near-tie"
    return n


def assert_rvq_equal_or_explained(got, ref, margins, tie, what):
    n_ids, n_frames, bad = explain_rvq_mismatches(got, ref, margins, tie)
    print(f"{what}: {n_ids} of {got.numel()} ids differ ({n_frames} frames), {bad} frames not explained by an oracle top-2 margin < {tie:g} at the first differing stage")
    assert bad == 0, f"{what}: {bad} frames differ from the oracle without a near-tie"
    return n_ids


# A second, measured explanation for inputs that amplify rounding noise (an impulse: LayerNorm of an almost constant vector): Euclidean
# distance is 1-Lipschitz in the quantised vector, so two implementations whose vectors differ by delta = ||x_got - x_ref||_2 at a position can
# only pick different codes there when the oracle's top-2 margin is <= 2 delta. The contract allows float intermediates to differ by 1e-3 per
# element; the bar uses the MEASURED delta of that position (normally ~1e-4), and separately bounds the per-element difference: <= FLOAT_TOL, or —
# where the reference's own output is ill-conditioned — <= NOISE_FACTOR x the amount by which the ORACLE differs from ITSELF when its front-end
# (framing, DFT, log-mel, per-bin normalisation) is evaluated in float64 instead of float32 (`x_ref_exact`): the reference's own rounding noise at
# that position, which no second implementation (another BLAS build of the reference included) reproduces.
FLOAT_TOL = 1e-3
NOISE_FACTOR = 4.0
# NOT the contract's bar: this second explanation exists for two STRESS suites only (inputs / weights on which the reference itself is ill-conditioned).
# Tests of BASELINE workloads, golden vectors and bench batches use assert_tokens_equal_or_explained (equal, or an oracle near-tie < 1e-3) and nothing else;
# the guard below makes that a property of the code, not of a reviewer's attention.
DELTA_HELPER_ALLOWED_IN = ("test_edge_inputs_gpu.py", "test_range_robustness_gpu.py", "test_parity_helpers_cpu.py")


def assert_tokens_equal_or_explained_by_delta(got, ref, margins, x_got, x_ref, tie, what, valid=None, x_ref_exact=None):
    """got / ref / margins [B, 1, T]; x_got / x_ref [B, T, D] = the vectors that were quantised (HIP path / oracle); x_ref_exact = the oracle's
    vectors with a float64 front-end (optional). A differing id must have an oracle margin < tie, or <= 2 ||x_got - x_ref||_2 at its position with
    max |x_got - x_ref| <= max(FLOAT_TOL, NOISE_FACTOR max |x_ref - x_ref_exact|) there. Prints the counts."""
    import inspect
    import os
    caller = os.path.basename(inspect.stack()[1].filename)
    assert caller in DELTA_HELPER_ALLOWED_IN, f"assert_tokens_equal_or_explained_by_delta is a stress-test bar only (called from {caller})"
    got, ref = got.cpu().long(), ref.cpu().long()
    mism = (got != ref)[:, 0]                                   # [B, T]
    if valid is not None:
        mism = mism & valid.cpu().bool().reshape(mism.shape)
    d = (x_got.cpu().float() - x_ref.cpu().float())
    delta, dmax = d.norm(dim=-1), d.abs().amax(dim=-1)          # [B, T]
    cap = torch.full_like(dmax, FLOAT_TOL)
    noise = torch.zeros_like(dmax)
    if x_ref_exact is not None:
        noise = (x_ref.cpu().float() - x_ref_exact.cpu().float()).abs().amax(dim=-1)
        cap = torch.maximum(cap, NOISE_FACTOR * noise)
    m = margins.cpu()[:, 0]
    by_tie = mism & (m < tie)
    by_delta = mism & ~by_tie & (m <= 2.0 * delta) & (dmax <= cap)
    bad = mism & ~by_tie & ~by_delta
    total = int(valid.sum()) if valid is not None else mism.numel()
    scope = valid.cpu().bool().reshape(mism.shape) if valid is not None else torch.ones_like(mism)
    print(f"{what}: {int(mism.sum())} of {total} ids differ from the oracle: {int(by_tie.sum())} at an oracle top-2 margin < {tie:g}, {int(by_delta.sum())} at a margin <= "
          f"2 x the measured vector difference (largest such margin {float(m[by_delta].max()) if bool(by_delta.any()) else 0.0:.2e}), {int(bad.sum())} unexplained; "
          f"max |x_got - x_ref| over the compared positions {float(dmax[scope].max()):.2e}, the oracle's own float32-vs-float64-front-end difference there "
          f"{float(noise[scope].max()):.2e}")
    assert int(bad.sum()) == 0, f"{what}: {int(bad.sum())} token ids differ from the oracle without an explanation"
    # away from the ill-conditioned positions the float contract holds as stated
    over = scope & (dmax > cap)
    assert not bool(over.any()), f"{what}: quantised vectors differ by {float(dmax[over].max()):.2e} > the tolerance at {int(over.sum())} positions"
    return int(mism.sum())
