"""Oracle pieces pinned by fixtures the REFERENCE's own code produced (tests/golden/gen_golden.py: box_tail, pipeline_order,
trba_post), and the product's host twins of the same steps checked against the same fixtures.

  box_tail.npz         EAST._scale_boxes_to_original / _polygon_area_batch / _remove_area_anomalies / _convert_to_axis_aligned
                       (/root/reference/src/manuscript/detectors/_east/infer.py:134-182,216-233) on 9 box sets x 4 detector
                       configurations x 3 page sizes
  pipeline_order.json  Pipeline.predict's ordering / re-match / min_text_size / crop block (_pipeline.py:100-140,204-221) run
                       by the reference Pipeline with stand-in plugins (protocol of tests/test_pipeline_api_compatibility.py:15-93)
  trba_post.npz        TRBA.predict's chunk loop + log_softmax + decode_tokens + confidence
                       (recognizers/_trba/__init__.py:374-432) on the reference network, chunk sizes 32 / 2 / 3
The contained-box filter (infer.py:184-214, cv2.pointPolygonTest) stays "parity unpinned": cv2 is absent.
"""
import json
import os
import zlib

import numpy as np
import pytest
import torch

from manuscript_ocr_amd import synth
from manuscript_ocr_amd.detectors._east import post as host_post
from oracle import east_post as P
from oracle import pipeline_glue as G
from oracle import trba_model as otm


def _biteq(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and a.tobytes() == b.tobytes()


def _box_tail_cases(golden_dir):
    g = np.load(os.path.join(golden_dir, "box_tail.npz"))
    meta = json.loads(bytes(g["meta"]).decode())
    return g, meta


def test_box_tail_oracle_vs_reference(golden_dir):
    g, meta = _box_tail_cases(golden_dir)
    assert len(meta) == 108
    changed = 0
    for m in meta:
        boxes, tag = g[m["set"] + "_in"], m["tag"]
        scaled = P.scale_boxes_to_original(boxes.copy(), tuple(m["orig"]), m["target"])
        assert _biteq(scaled, g[tag + "_scaled"]), tag
        assert _biteq(P.polygon_area_batch(scaled[:, :8].reshape(-1, 4, 2)), g[tag + "_areas"]), tag
        kept = P.remove_area_anomalies(scaled, m["remove"], m["sigma"], m["min_count"])
        assert _biteq(kept, g[tag + "_kept"]), tag
        assert _biteq(P.convert_to_axis_aligned(kept), g[tag + "_aligned"]), tag
        changed += len(kept) != len(boxes)
    assert changed >= 30  # the fixtures do exercise the 5-sigma filter, its M > min_count gate and the sigma knob


def test_box_tail_product_host_vs_reference(golden_dir):
    """manuscript_ocr_amd/detectors/_east/post.py (the host fallback of msocr_east_box_tail) against the same fixtures."""
    g, meta = _box_tail_cases(golden_dir)
    for m in meta:
        boxes, tag, T = g[m["set"] + "_in"], m["tag"], m["target"]
        scaled = host_post.scale_boxes(boxes.copy(), tuple(m["orig"]), (T, T))
        assert _biteq(scaled, g[tag + "_scaled"]), tag
        assert _biteq(host_post.quad_areas(scaled[:, :8].reshape(-1, 4, 2)), g[tag + "_areas"]), tag
        kept = host_post.remove_area_anomalies(scaled, m["remove"], m["sigma"], m["min_count"])
        assert _biteq(kept, g[tag + "_kept"]), tag
        assert _biteq(host_post.to_axis_aligned(kept), g[tag + "_aligned"]), tag


def test_box_tail_native_host_twin_vs_reference(golden_dir):
    """msocr_east_box_tail_host runs the device kernel's __host__ __device__ code on the CPU: scale -> contained -> 5 sigma ->
    axis-aligned in one call.  On fixture sets without contained boxes its output must equal the reference's chain."""
    from manuscript_ocr_amd import _native as nat
    lib = nat.lib()
    g, meta = _box_tail_cases(golden_dir)
    ran = 0
    for m in meta:
        boxes, tag, T = g[m["set"] + "_in"], m["tag"], m["target"]
        scaled = g[tag + "_scaled"]
        if len(boxes) == 0 or len(P.remove_fully_contained_boxes(scaled)) != len(scaled):
            continue  # the contained-box filter would act: outside what the reference fixture pins
        out = np.zeros((len(boxes), 9), np.float32)
        cnt = np.zeros(1, np.int32)
        oh, ow = m["orig"]
        rc = lib.msocr_east_box_tail_host(np.ascontiguousarray(boxes).ctypes.data, len(boxes), 0.0, 0.0,
                                          ow / T, oh / T, 1, int(m["remove"]), m["sigma"], m["min_count"],
                                          out.ctypes.data, cnt.ctypes.data)
        assert rc == 0, tag
        assert _biteq(out[:cnt[0]], g[tag + "_aligned"]), tag
        ran += 1
    assert ran >= 30


def _order_cases(golden_dir):
    with open(os.path.join(golden_dir, "pipeline_order.json")) as f:
        return json.load(f)


def test_pipeline_order_oracle_vs_reference(golden_dir):
    cases = _order_cases(golden_dir)
    assert len(cases) == 8 and sum(len(c["polygons"]) for c in cases) > 350
    for ci, c in enumerate(cases):
        img = np.random.default_rng(c["image_seed"]).integers(0, 256, size=(c["h"], c["w"], 3), dtype=np.uint8)
        order, kept, crops = G.order_and_crop(c["polygons"], img, c["min_text_size"])
        assert order == c["order"], ci
        assert [[list(r.shape), zlib.crc32(np.ascontiguousarray(r).tobytes())] for r in crops] == c["crops"], ci
        word_text = {}  # duplicates put the SAME Word twice into the block: the later crop's text overwrites (_pipeline.py:149-162)
        for k, pos in enumerate(kept):
            word_text[order[pos]] = f"w{k}"
        assert [word_text.get(wi) for wi in order] == c["texts"], ci
        assert c["rec_calls"] == (1 if crops else 0)


def test_pipeline_order_product_vs_reference(golden_dir):
    """The product Pipeline (generic plugin branch, host reading order) with stand-in plugins on the same pages."""
    from manuscript_ocr_amd import Pipeline
    from manuscript_ocr_amd.detectors._types import Block, Page, Word

    class Det:
        def __init__(self, polys):
            self.words = [Word(polygon=p, detection_confidence=0.5) for p in polys]

        def predict(self, image, vis=False, profile=False):
            return {"page": Page(blocks=[Block(words=list(self.words))]), "vis_image": None, "score_map": None, "geo_map": None}

    class Rec:
        def __init__(self):
            self.calls = []

        def predict(self, images):
            self.calls.append([[list(im.shape), zlib.crc32(np.ascontiguousarray(im).tobytes())] for im in images])
            return [{"text": f"w{i}", "confidence": 1.0 / (1 + i)} for i in range(len(images))]

    for ci, c in enumerate(_order_cases(golden_dir)):
        img = np.random.default_rng(c["image_seed"]).integers(0, 256, size=(c["h"], c["w"], 3), dtype=np.uint8)
        det, rec = Det(c["polygons"]), Rec()
        page = Pipeline(detector=det, recognizer=rec, min_text_size=c["min_text_size"]).predict(img)
        ident = {id(w): i for i, w in enumerate(det.words)}
        words = page.blocks[0].words
        assert [ident[id(w)] for w in words] == c["order"], ci
        assert [w.text for w in words] == c["texts"], ci
        assert len(rec.calls) == c["rec_calls"], ci
        assert (rec.calls[0] if rec.calls else []) == c["crops"], ci


def test_native_reading_order_host_vs_reference(golden_dir):
    """msocr_reading_order_host (the C++ twin of the device reading-order kernel) reproduces the reference's word order."""
    from manuscript_ocr_amd import _native as nat
    lib = nat.lib()
    for ci, c in enumerate(_order_cases(golden_dir)):
        if not c["polygons"]:
            continue
        boxes = np.array([[int(v) for v in G.word_box(p)] for p in c["polygons"]], dtype=np.int32)
        order = np.empty(len(boxes), dtype=np.int32)
        assert lib.msocr_reading_order_host(boxes.ctypes.data, len(boxes), 0.6, float("inf"), order.ctypes.data) == 0
        assert order.tolist() == c["order"], ci


def _trba_net(seed):
    net = otm.TRBANet(194, 256)
    net.load_state_dict(synth.trba_state_dict(194, 256, seed=seed), strict=True)
    return net.eval()


def _charset():
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "manuscript_ocr_amd", "recognizers",
                        "_trba", "configs", "charset.txt")
    return otm.load_charset(path)[0]


def _oracle_predict(net, crops, itos, bs, mode, **kw):
    """TRBA.predict's chunk loop on the oracle network + oracle.trba_model.texts_and_confidences."""
    x = torch.from_numpy(((crops.astype(np.float32) - 127.5) * np.float32(1 / 127.5)).transpose(0, 3, 1, 2).copy())
    res = []
    with torch.no_grad():
        for i in range(0, len(x), bs):
            lg, ids = net(x[i:i + bs], max_len=25, mode=mode, **kw)
            res += otm.texts_and_confidences(lg, ids, itos, 0, 2, None)
    return res


@pytest.mark.parametrize("tag,B,h,w", [("b7_32x100", 7, 32, 100), ("b3_64x256", 3, 64, 256)])
def test_trba_post_oracle_vs_reference(golden_dir, tag, B, h, w):
    g = np.load(os.path.join(golden_dir, "trba_post.npz"))
    seed = int(g["seed"])
    net, itos = _trba_net(seed), _charset()
    crops = synth.synth_crops(seed + 5, B, h, w)
    moved = 0
    for mode, key, kw in (("greedy", "greedy", {}), ("beam", "beam8", dict(beam_size=8, alpha=0.9, temperature=1.7)),
                          ("beam", "beam5", dict(beam_size=5, alpha=0.0, temperature=1.0))):
        for bs in (32, 2, 3):
            r = _oracle_predict(net, crops, itos, bs, mode, **kw)
            k = f"{tag}_{key}_bs{bs}"
            assert [x["text"] for x in r] == g[k + "_text"].tolist(), k
            assert np.array_equal(np.array([x["confidence"] for x in r], dtype=np.float64), g[k + "_conf"]), k
        moved += not np.array_equal(g[f"{tag}_{key}_bs32_conf"], g[f"{tag}_{key}_bs2_conf"])
    assert moved >= 1  # the fixtures do show the chunk-composition dependence of the confidence (run length)


def test_trba_post_on_committed_logits(golden_dir):
    """The post-step ALONE: oracle.texts_and_confidences on the reference logits / ids stored in trba.npz equals what the
    reference's TRBA.predict returned for the same four crops."""
    g, gp = np.load(os.path.join(golden_dir, "trba.npz")), np.load(os.path.join(golden_dir, "trba_post.npz"))
    itos = _charset()
    for mode in ("greedy", "beam"):
        r = otm.texts_and_confidences(torch.from_numpy(g[f"b4_32x100_{mode}_logits"]), torch.from_numpy(g[f"b4_32x100_{mode}_ids"]),
                                      itos, 0, 2, None)
        assert [x["text"] for x in r] == gp[f"npz_b4_32x100_{mode}_text"].tolist()
        assert np.array_equal(np.array([x["confidence"] for x in r], dtype=np.float64), gp[f"npz_b4_32x100_{mode}_conf"])
