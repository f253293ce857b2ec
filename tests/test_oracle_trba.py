"""Oracle TRBA restatement vs golden vectors generated from the reference's model files."""
import os

import numpy as np
import pytest
import torch

from manuscript_ocr_amd import synth
from oracle import trba_model as otm


def _net(seed):
    net = otm.TRBANet(194, 256)
    net.load_state_dict(synth.trba_state_dict(194, 256, seed=seed), strict=True)
    return net.eval()


@pytest.mark.parametrize("tag,B,h,w", [("b4_32x100", 4, 32, 100), ("b2_64x256", 2, 64, 256)])
def test_trba_golden(golden_dir, tag, B, h, w):
    g = np.load(os.path.join(golden_dir, "trba.npz"))
    seed = int(g["seed"])
    net = _net(seed)
    crops = synth.synth_crops(seed + 2, B, h, w)
    x = torch.from_numpy(((crops.astype(np.float32) - 127.5) * np.float32(1 / 127.5)).transpose(0, 3, 1, 2).copy())
    with torch.no_grad():
        f = net.cnn(x)
        enc = net.encode(x)
        gl, gi = net(x, max_len=25, mode="greedy")
        bl, bi = net(x, max_len=25, mode="beam", beam_size=8, alpha=0.9, temperature=1.7)
        bl5, bi5 = net(x, max_len=25, mode="beam", beam_size=5, alpha=0.0, temperature=1.0)
    # same torch ops in the same order as the reference -> bit-exact
    assert np.array_equal(f.numpy(), g[f"{tag}_cnn"])
    assert np.array_equal(enc.numpy(), g[f"{tag}_enc"])
    for got_l, got_i, key in ((gl, gi, "greedy"), (bl, bi, "beam"), (bl5, bi5, "beam5")):
        assert np.array_equal(got_i.numpy(), g[f"{tag}_{key}_ids"]), key
        assert np.array_equal(got_l.numpy(), g[f"{tag}_{key}_logits"]), key


def test_charset_and_decode_tokens():
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "manuscript_ocr_amd",
                        "recognizers", "_trba", "configs", "charset.txt")
    itos, stoi = otm.load_charset(path)
    assert len(itos) == 194 and itos[:4] == ["<PAD>", "<SOS>", "<EOS>", " "] and "<BLANK>" not in stoi
    assert otm.decode_tokens([4, 0, 5, 2, 6], itos, 0, 2, None) == "ab"
    assert otm.decode_tokens([], itos, 0, 2, None) == ""


def test_oracle_self_sensitivity():
    """How chaotic the all-random-weights decoder is, measured on the CPU oracle alone: a 1e-7 (rounding-level) perturbation
    of the input (1e-5 at the encoder output) moves its own logits by 2e-5..9e-5 of the largest |logit| on 90 % of the rows
    and by up to ~3e-2 on the worst row over the 25-26 chained steps; where its ids change, they change at a near-tie of its
    own decode (conftest.compare_decodes).  This is the yardstick for the tolerances of
    tests/test_gpu_trba.py::test_trba_random_weights_*."""
    from conftest import compare_decodes, oracle_decode_chunks
    N = 64
    net = _net(20260128)
    crops = synth.synth_crops(77, N, 32, 100)
    x = torch.from_numpy(((crops.astype(np.float32) - 127.5) * np.float32(1 / 127.5)).transpose(0, 3, 1, 2).copy())
    xp = x + 1e-7 * torch.randn(x.shape, generator=torch.Generator().manual_seed(1))
    for mode, steps in (("greedy", 26), ("beam", 25)):
        a, b = oracle_decode_chunks(net, x, mode), oracle_decode_chunks(net, xp, mode)
        ids = np.full((N, steps), -1, np.int64)
        lg = np.zeros((N, steps, 194), np.float32)
        trun = np.zeros(N, np.int32)
        for i, r in enumerate(b):
            T = len(r["ids"])
            ids[i, :T], lg[i, :T], trun[i] = r["ids"], r["logits"], T
        rep = compare_decodes(ids, trun, lg, a, mode, logit_rtol=5e-2)
        assert not rep["hard"], (mode, rep["hard"])
        assert len(rep["ties"]) <= 3, (mode, rep["ties"])
        assert 1e-5 < np.quantile(rep["row_logit_err_rel"], 0.9) < 3e-4, (mode, np.quantile(rep["row_logit_err_rel"], 0.9))
