"""Parity at BASELINE.json's full sizes (configs[1]-[3]): one 1536x2048 page through the EAST network, 256 crops
through TRBA, against the oracle on the same seeded inputs (the oracle needs a few seconds of host CPU for these)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CHARSET = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "manuscript_ocr_amd", "recognizers", "_trba",
                       "configs", "charset.txt")


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def test_east_full_resolution_maps(gpu):
    """configs[1] network input 1536x2048: score within 1e-4, geometry within 1e-3 of its max, and the set of
    above-threshold pixels identical except where the CPU score is within 1e-4 of the threshold."""
    from manuscript_ocr_amd import synth
    from manuscript_ocr_amd.detectors._east.net import EastNet
    from oracle import east_model as oem
    from oracle import imgproc
    H, W = 1536, 2048
    sd = synth.east_state_dict(seed=20260128)
    page = synth.synth_page(100, H, W)[0]
    ref = oem.EASTNet()
    ref.load_state_dict(sd)
    ref.eval()
    with torch.no_grad():
        r = ref(torch.from_numpy(imgproc.east_preprocess(page, W, H)))
    rs, rg = r["score"][0, 0].numpy(), r["geometry"][0].permute(1, 2, 0).numpy()
    score, geo = EastNet(sd, torch.float32).forward(torch.from_numpy(page[None]).cuda())
    s, g = score[0].cpu().numpy(), geo[0].cpu().numpy()
    assert np.abs(s - rs).max() < 1e-4
    assert np.abs(g - rg).max() < 1e-3 * np.abs(rg).max()
    thr = np.float32(0.6)
    diff = (s > thr) != (rs > thr)
    assert np.all(np.abs(rs[diff] - thr) < 1e-4)
    assert rs.std() > 0.02 and 100 < (rs > thr).sum() < rs.size - 100, "degenerate map"


def test_trba_batch256_beam_text(gpu):
    """configs[2]: 256 crops @32x100 in ONE reference chunk (batch_size=256): identical token ids, texts and run
    length; confidences within 1e-4."""
    from manuscript_ocr_amd import synth
    from manuscript_ocr_amd.recognizers import TRBA
    from conftest import compare_texts
    from oracle import trba_model as otm
    seed = 20260128
    sd = synth.trba_state_dict_confident(194, 256, seed=seed)
    rec = TRBA(state_dict=sd, config={"img_h": 32, "img_w": 100, "max_len": 25, "hidden_size": 256}, device="cuda")
    canv = synth.synth_crops(7, 256, 32, 100)
    got = rec.predict(list(canv), batch_size=256)
    net = otm.TRBANet(194, 256)
    net.load_state_dict(sd)
    net.eval()
    x = torch.from_numpy(((canv.astype(np.float32) - 127.5) * np.float32(1 / 127.5)).transpose(0, 3, 1, 2).copy())
    with torch.no_grad():
        lg, ids = net(x, max_len=25, mode="beam", beam_size=8, alpha=0.9, temperature=1.7)
    itos, _ = otm.load_charset(CHARSET)
    exp = otm.texts_and_confidences(lg, ids, itos, 0, 2, None)
    texts_e = [r["text"] for r in exp]
    same = compare_texts([r["text"] for r in got], exp, itos, max_ties=2)
    np.testing.assert_allclose([got[i]["confidence"] for i in same], [exp[i]["confidence"] for i in same], atol=1e-4)
    assert len(set(texts_e)) >= 20
