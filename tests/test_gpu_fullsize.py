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
    """configs[1] network input 1536x2048: score within 1e-4, geometry within 2.5e-6 of its max (conftest.assert_maps_close), and
    the set of above-threshold pixels identical except where the CPU score is within 1e-4 of the threshold."""
    from conftest import assert_maps_close
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
    assert_maps_close(s, g, rs, rg, "east 1536x2048")
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


def _boxes_of(page):
    return np.array([[c for pt in w.polygon for c in pt] + [w.detection_confidence] for w in page.blocks[0].words], dtype=np.float32).reshape(-1, 9)


def test_east_batch8_full_resolution(gpu):
    """configs[1] as stated: batch = 8 pages @ 2048x1536 in ONE launch sequence.  Pages 0 and 7 against the oracle's CPU network
    (score 1e-4 abs, geometry 2.5e-6 of its max); every page of the batch bit-identical to the same page run alone (batching
    changes the GEMM M only: same tiles, same summation order)."""
    from conftest import assert_maps_close
    from manuscript_ocr_amd import synth
    from manuscript_ocr_amd.detectors._east.net import EastNet
    from oracle import east_model as oem
    from oracle import imgproc
    H, W = 1536, 2048
    sd = synth.east_state_dict(seed=20260128)
    pages = np.stack([synth.synth_page(100 + i, H, W)[0] for i in range(8)])
    net = EastNet(sd, torch.float32)
    score, geo = net.forward(torch.from_numpy(pages).cuda())
    s_all, g_all = score.cpu().numpy(), geo.cpu().numpy()
    ref = oem.EASTNet()
    ref.load_state_dict(sd)
    ref.eval()
    for i in (0, 7):
        with torch.no_grad():
            r = ref(torch.from_numpy(imgproc.east_preprocess(pages[i], W, H)))
        rs, rg = r["score"][0, 0].numpy(), r["geometry"][0].permute(1, 2, 0).numpy()
        assert_maps_close(s_all[i], g_all[i], rs, rg, f"east batch of 8, page {i}")
    for i in range(8):
        s1, g1 = net.forward(torch.from_numpy(pages[i:i + 1]).cuda())
        assert np.array_equal(s1[0].cpu().numpy(), s_all[i]) and np.array_equal(g1[0].cpu().numpy(), g_all[i]), i
    assert len({s_all[i].tobytes() for i in range(8)}) == 8


def test_config4_share_3072x4096_pipeline(gpu):
    """BASELINE configs[4], one GPU's geometry: pages @ 4096x3072 with the native 3072x4096 network input (768x1024 maps,
    injected).  Two pages through Pipeline.predict_batch; page 0 (2066 words: above the 2048 boxes the box filters used to
    hold in LDS; 57 k candidates) against the oracle end to end — boxes bit-exact, reading order, the texts of its first 768 crops
    under the near-tie rule, confidences 1e-4 (the CPU recogniser on all 2066 was 45 s of the driver's 900 s GPU tier; every crop
    still has to carry a text); page 1 against the oracle's detector post-processing + reading order (bit-exact) with every word
    recognised.  Peak reserved device memory is asserted (Winograd workspaces come from one arena per stream)."""
    from conftest import compare_texts
    from manuscript_ocr_amd import Pipeline, synth
    from manuscript_ocr_amd.detectors import EAST
    from manuscript_ocr_amd.recognizers import TRBA
    from oracle import east_post as P
    from oracle import lanms as L
    from oracle import pipeline_glue as G
    from oracle import trba_model as otm
    from test_gpu_pipeline import _oracle_pipeline
    H, W = 3072, 4096
    cfg = {"img_h": 32, "img_w": 100, "max_len": 25, "hidden_size": 256}
    tsd = synth.trba_state_dict_confident(194, 256, seed=20260128)
    det = EAST(state_dict=synth.east_state_dict(), target_size=(W, H), device="cuda")
    rec = TRBA(state_dict=tsd, config=cfg, device="cuda")
    assert det._max_candidates() == 384 * 512
    pipe = Pipeline(detector=det, recognizer=rec)
    pages, maps = [], []
    for seed in (1000, 1001):
        pg, rects = synth.synth_page(seed, H, W)
        pages.append(pg)
        maps.append(synth.synth_maps(rects, (H, W), (H // 4, W // 4), seed))
    mo = (torch.from_numpy(np.stack([m[0] for m in maps])).cuda(), torch.from_numpy(np.stack([m[1] for m in maps])).cuda())
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    got = pipe.predict_batch(pages, _maps_override=mo)
    torch.cuda.synchronize()
    peak_gb = torch.cuda.max_memory_reserved() / 2 ** 30
    print(f"configs[4] share, 2 pages: peak reserved {peak_gb:.1f} GiB, words {[len(p.blocks[0].words) for p in got]}")
    assert peak_gb < 48, peak_gb
    ref_net = otm.TRBANet(194, 256)
    ref_net.load_state_dict(tsd)
    ref_net.eval()
    itos, _ = otm.load_charset(CHARSET)
    # page 0: the whole reference path on the CPU
    exp = _oracle_pipeline(pages[0], maps[0][0], maps[0][1], ref_net, itos, cfg, max_text=768)
    gw = got[0].blocks[0].words
    assert len(gw) == len(exp) and len(exp) > 2048
    n_text = ties = unchecked = 0
    for a, b in zip(gw, exp):
        assert [tuple(p) for p in a.polygon] == [tuple(p) for p in b["polygon"]]
        assert a.detection_confidence == b["det"]
        if b["text"] == "?unchecked":
            assert a.text is not None and a.recognition_confidence is not None
            unchecked += 1
        elif b["rec"] is None:
            assert a.text is None and a.recognition_confidence is None
        elif a.text != b["text"]:
            ties += len(compare_texts([a.text], [b], itos)) == 0
        else:
            assert abs(a.recognition_confidence - b["rec"]) < 1e-4
            n_text += 1
    assert n_text > 740 and ties <= 2 and n_text + ties + unchecked > 2000, (n_text, ties, unchecked)
    # page 1: detector post-processing and reading order against the oracle, bit for bit
    quads = P.east_postprocess(maps[1][0], maps[1][1], (H, W), (W, H), L.locality_aware_nms)
    polys = [q[:8].reshape(4, 2).tolist() for q in quads]
    order, kept, crops = G.order_and_crop(polys, pages[1], 5)
    gw1 = got[1].blocks[0].words
    assert len(gw1) == len(order) > 1500
    assert [[tuple(p) for p in w.polygon] for w in gw1] == [[tuple(p) for p in polys[wi]] for wi in order]
    assert [w.detection_confidence for w in gw1] == [float(quads[wi][8]) for wi in order]
    assert sum(w.text is not None for w in gw1) == len(kept)


def test_east_dense_page_above_old_capacities(gpu):
    """Detector post-processing at the configs[4] map size on pages that exceed the round-1 capacities: (a) > 65536 candidate
    cells (one third of the 768x1024 map above threshold), (b) > 2048 boxes after the NMS, through the device box filters
    (per-box arrays in the workspace instead of LDS), (c) the same page through the HOST fallback tail: all bit-identical
    to the oracle (decode -> LANMS -> expand -> scale -> contained -> anomalies -> axis-aligned)."""
    from manuscript_ocr_amd import synth
    from manuscript_ocr_amd.detectors import EAST
    from oracle import east_post as P
    from oracle import lanms as L
    H, W = 3072, 4096
    det = EAST(state_dict=synth.east_state_dict(), target_size=(W, H), device="cuda")
    # (a) long, tall words: 0.35 of the cells above threshold
    rects = np.array([(60 + 2000 * c, 40 + 126 * r, 60 + 2000 * c + 1900, 40 + 126 * r + 120) for r in range(23) for c in range(2)], dtype=np.float64)
    dense = synth.synth_maps(rects, (H, W), (H // 4, W // 4), 5)
    # (b) a tight layout: ~2460 words
    page_b, rects_b = synth.synth_page(1000, H, W, line_pitch=30, word_h=24)
    many = synth.synth_maps(rects_b, (H, W), (H // 4, W // 4), 6)
    page = np.zeros((H, W, 3), dtype=np.uint8)
    mo = (torch.from_numpy(np.stack([dense[0], many[0]])).cuda(), torch.from_numpy(np.stack([dense[1], many[1]])).cuda())
    res = det.predict_batch([page, page], _maps_override=mo)
    n_cand = [len(P.decode_quads_from_maps(m[0], m[1], 0.6, 4.0, 2)) for m in (dense, many)]
    assert n_cand[0] > 65536, n_cand
    exp = [P.east_postprocess(m[0], m[1], (H, W), (W, H), L.locality_aware_nms) for m in (dense, many)]
    assert len(exp[1]) > 2048, len(exp[1])
    for r, e in zip(res, exp):
        got = _boxes_of(r["page"])
        assert got.shape == e.shape and np.array_equal(got, e), (got.shape, e.shape)
    det.device_tail = False  # (c) host fallback tail (what a page above 16384 boxes takes)
    res_h = det.predict_batch([page, page], _maps_override=mo)
    for r, e in zip(res_h, exp):
        assert np.array_equal(_boxes_of(r["page"]), e)


def test_config3_16_pages_pipelined_schedule(gpu):
    """BASELINE configs[3] as stated and as bench.py times it: batches of 16 pages @ 2048x1536 (native 1536x2048 network input,
    injected maps) through Pipeline.submit_batch -> advance_batch -> collect_batch with bench.py's software-pipelined schedule
    (two stream sets; batch i+1's detector is on the device and batch i+1 is advanced before batch i is collected).
    Three batches A (seeds 200-215, the bench's pages), B (seeds 300-315), A again:
      * all 16 pages of A and of B: boxes, detection confidences and reading order bit-equal to the oracle
        (east_postprocess + pipeline_glue.order_and_crop = reference infer.py:319-363 + _pipeline.py:100-137), every word that
        the reference crops carries a text;
      * texts / confidences of one page of A and one page of B against the oracle's CPU recogniser (near-tie rule, 1e-4);
      * A's Pages are the same whether A runs alone or with B in flight around it, and the same the second time."""
    from conftest import compare_texts
    from manuscript_ocr_amd import Pipeline, synth
    from manuscript_ocr_amd.detectors import EAST
    from manuscript_ocr_amd.recognizers import TRBA
    from oracle import east_post as P
    from oracle import lanms as L
    from oracle import pipeline_glue as G
    from oracle import trba_model as otm
    from test_gpu_pipeline import _oracle_pipeline
    H, W, NP = 1536, 2048, 16
    cfg = {"img_h": 32, "img_w": 100, "max_len": 25, "hidden_size": 256}
    tsd = synth.trba_state_dict_confident(194, 256, seed=20260128)
    det = EAST(state_dict=synth.east_state_dict(seed=20260128), target_size=(W, H), device="cuda")
    rec = TRBA(state_dict=tsd, config=cfg, device="cuda")
    pipe = Pipeline(detector=det, recognizer=rec)
    pipe.stream_sets = 2

    def make(seed0):
        pages, maps = [], []
        for i in range(NP):
            pg, rects = synth.synth_page(seed0 + i, H, W)
            pages.append(pg)
            maps.append(synth.synth_maps(rects, (H, W), (H // 4, W // 4), seed0 + i))
        dev = torch.from_numpy(np.stack(pages)).cuda()
        mo = (torch.from_numpy(np.stack([m[0] for m in maps])).cuda(), torch.from_numpy(np.stack([m[1] for m in maps])).cuda())
        return pages, maps, dev, mo

    A, B = make(200), make(300)

    def submit(batch):
        return pipe.submit_batch(batch[0], pages_dev=batch[2], _maps_override=batch[3])

    def flat(pages):
        return [(tuple(map(tuple, w.polygon)), w.detection_confidence, w.text, w.recognition_confidence)
                for p in pages for w in p.blocks[0].words]

    alone = flat(pipe.collect_batch(submit(A)))          # A with nothing else in flight
    torch.cuda.synchronize()
    # bench.py run_steps: invariant at the top of the loop = batch i advanced, batch i+1 submitted
    seq = [A, B, A]
    adv, sub, nsub, outs = [], [], 0, []
    for i in range(len(seq)):
        while len(adv) < 2 and (sub or nsub < len(seq)):
            if not sub:
                sub.append(submit(seq[nsub]))
                nsub += 1
            adv.append(pipe.advance_batch(sub.pop(0)))
            if nsub < len(seq):
                sub.append(submit(seq[nsub]))
                nsub += 1
        outs.append(pipe.collect_batch(adv.pop(0)))
    torch.cuda.synchronize()
    assert flat(outs[0]) == alone, "batch A changed when batch B was in flight"
    assert flat(outs[2]) == alone, "batch A changed on its second pass"
    assert flat(outs[1]) != alone
    # every page of both batches against the oracle's detector post-processing + reading order, bit for bit
    n_words = 0
    for batch, got_pages in ((A, outs[0]), (B, outs[1])):
        for pi in range(NP):
            s, g = batch[1][pi]
            quads = P.east_postprocess(s, g, (H, W), (W, H), L.locality_aware_nms)
            polys = [q[:8].reshape(4, 2).tolist() for q in quads]
            order, kept, crops = G.order_and_crop(polys, batch[0][pi], 5)
            gw = got_pages[pi].blocks[0].words
            assert len(gw) == len(order) > 300, (pi, len(gw), len(order))
            assert [[tuple(p) for p in w.polygon] for w in gw] == [[tuple(p) for p in polys[wi]] for wi in order], pi
            assert [w.detection_confidence for w in gw] == [float(quads[wi][8]) for wi in order], pi
            assert [k for k, w in enumerate(gw) if w.text is not None] == kept, pi
            n_words += len(gw)
    assert n_words > 2 * NP * 400
    # recogniser output of one page per batch against the CPU path
    ref_net = otm.TRBANet(194, 256)
    ref_net.load_state_dict(tsd)
    ref_net.eval()
    itos, _ = otm.load_charset(CHARSET)
    for batch, got_pages, pi in ((A, outs[0], 3), (B, outs[1], 12)):
        exp = _oracle_pipeline(batch[0][pi], batch[1][pi][0], batch[1][pi][1], ref_net, itos, cfg, max_text=320)   # CPU recogniser: 320 of ~480 crops
        gw = got_pages[pi].blocks[0].words
        assert len(gw) == len(exp)
        n_text = ties = 0
        for a, b in zip(gw, exp):
            if b["text"] == "?unchecked":
                assert a.text is not None and a.recognition_confidence is not None
            elif b["rec"] is None:
                assert a.text is None and a.recognition_confidence is None
            elif a.text != b["text"]:
                ties += len(compare_texts([a.text], [b], itos)) == 0
            else:
                assert abs(a.recognition_confidence - b["rec"]) < 1e-4
                n_text += 1
        assert n_text > 300 and ties <= 1, (pi, n_text, ties)
