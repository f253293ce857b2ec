"""Generate golden vectors by running the REFERENCE's own source files.

Run once in the build container (needs /root/reference):
    python tests/golden/gen_golden.py
Outputs small fixtures next to this file.  Only DATA is stored (inputs, seeds,
expected outputs) — never reference source.  See _refload.py for how single
reference files are loaded behind inert stubs.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import _refload  # noqa: E402
from manuscript_ocr_amd import synth  # noqa: E402
from oracle import east_model as oem  # noqa: E402
from oracle import trba_model as otm  # noqa: E402


def perturb_bn(sd, seed):
    """Deterministic non-trivial BN statistics/affine (applied to reference AND oracle)."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, v in sd.items():
        v = v.clone()
        if k.endswith("running_var"):
            v = 0.5 + torch.rand(v.shape, generator=g)
        elif k.endswith("running_mean"):
            v = 0.2 * torch.randn(v.shape, generator=g)
        out[k] = v
    # BN affine: identify by the sibling running_mean key
    for k in list(out):
        if k.endswith("running_mean"):
            base = k[: -len("running_mean")]
            out[base + "weight"] = 0.7 + 0.6 * torch.rand(out[base + "weight"].shape, generator=g)
            out[base + "bias"] = 0.2 * torch.randn(out[base + "bias"].shape, generator=g)
    return out


def gen_lanms():
    ref = _refload.ref_lanms()
    out = {}
    rng = np.random.default_rng(11)
    cases = {}
    # (a) candidates decoded from injected maps (realistic: many near-duplicates per word)
    utils = _refload.ref_east_utils()
    for name, (H, W, seed) in {"page_small": (256, 384, 3), "page_mid": (512, 768, 4)}.items():
        rects = synth.synth_layout(seed, H, W)
        score, geo = synth.synth_maps(rects, (H, W), (H // 4, W // 4), seed)
        cand = utils.decode_quads_from_maps(score, geo, 0.6, 4.0, 2)
        cases[name] = cand
        out[f"{name}_score"], out[f"{name}_geo"] = score, geo
        out[f"{name}_decoded"] = cand
    # (b) random rotated quads, tie-free
    def rot_quads(n, span):
        c = rng.uniform(20, span, size=(n, 2))
        wh = rng.uniform(8, 60, size=(n, 2))
        th = rng.uniform(-0.6, 0.6, size=n)
        base = np.array([[-1, -1], [1, -1], [1, 1], [-1, 1]], dtype=np.float64) * 0.5
        q = np.empty((n, 9), dtype=np.float32)
        for i in range(n):
            R = np.array([[np.cos(th[i]), -np.sin(th[i])], [np.sin(th[i]), np.cos(th[i])]])
            pts = (base * wh[i]) @ R.T + c[i]
            k = rng.integers(0, 4)  # random vertex rotation; keep winding
            pts = np.roll(pts, k, axis=0)
            q[i, :8] = pts.reshape(-1)
            q[i, 8] = rng.uniform(0.6, 1.0)
        assert len(np.unique(q[:, 0])) == n and len(np.unique(q[:, 8])) == n
        return q
    cases["rot_1"] = rot_quads(1, 100)
    cases["rot_50"] = rot_quads(50, 200)
    cases["rot_600"] = rot_quads(600, 700)
    # (c) reversed winding (clockwise in image coords flipped) -> exercises empty intersections
    rv = rot_quads(40, 150)
    rv[:, :8] = rv[:, :8].reshape(-1, 4, 2)[:, ::-1].reshape(-1, 8)
    cases["rot_rev_40"] = rv
    for name, boxes in cases.items():
        for thr in (0.2,):
            res = ref.locality_aware_nms(boxes, thr)
            out[f"{name}_in"] = boxes.astype(np.float32)
            out[f"{name}_out"] = res.astype(np.float32)
            print("lanms", name, boxes.shape, "->", res.shape)
    np.savez_compressed(os.path.join(HERE, "lanms.npz"), **out)


def gen_east_post():
    utils = _refload.ref_east_utils()
    out = {}
    H, W, seed = 192, 256, 21
    rects = synth.synth_layout(seed, H * 4, W * 4)
    score, geo = synth.synth_maps(rects, (H * 4, W * 4), (H, W), seed)
    # sprinkle isolated above-threshold pixels incl. last row/col cells and sub-threshold cell centres
    rng = np.random.default_rng(5)
    ys, xs = rng.integers(0, H, 300), rng.integers(0, W, 300)
    score[ys, xs] = rng.uniform(0.55, 0.99, 300).astype(np.float32)
    out["score"], out["geo"] = score, geo
    for q in (1, 2, 4):
        out[f"decoded_q{q}"] = utils.decode_quads_from_maps(score, geo, 0.6, 4.0, q)
    out["decoded_thr09_q2"] = utils.decode_quads_from_maps(score, geo, 0.9, 4.0, 2)
    out["decoded_empty"] = utils.decode_quads_from_maps(np.zeros_like(score), geo, 0.6, 4.0, 2)
    lan = _refload.ref_lanms().locality_aware_nms(out["decoded_q2"], 0.2)
    out["lanms_q2"] = lan
    out["expanded"] = utils.expand_boxes(lan, 0.9, 0.9)
    out["expanded_w05_h0"] = utils.expand_boxes(lan, 0.5, 0.0)
    # degenerate quads for expand: zero-area, repeated vertices, reversed winding
    deg = np.array([
        [0, 0, 10, 0, 10, 0, 0, 0, 0.7],
        [5, 5, 5, 5, 5, 5, 5, 5, 0.8],
        [0, 0, 0, 10, 10, 10, 10, 0, 0.9],
    ], dtype=np.float32)
    out["deg_in"] = deg
    out["deg_expanded"] = utils.expand_boxes(deg, 0.9, 0.9)
    np.savez_compressed(os.path.join(HERE, "east_post.npz"), **out)
    print("east_post", {k: v.shape for k, v in out.items()})


def gen_east_decoder_head():
    rm = _refload.ref_east_model()
    seed = 1234
    torch.manual_seed(seed)
    ref_dec, ref_head = rm.FeatureMergingBranchResNet(), rm.OutputHead()
    torch.manual_seed(seed)
    my_dec, my_head = oem.FeatureMergingBranchResNet(), oem.OutputHead()
    for a, b in ((ref_dec, my_dec), (ref_head, my_head)):
        sa, sb = a.state_dict(), b.state_dict()
        assert list(sa) == list(sb)
        assert all(torch.equal(sa[k], sb[k]) for k in sa), "seeded construction differs from reference"
    sd = perturb_bn(ref_dec.state_dict(), seed + 1)
    ref_dec.load_state_dict(sd)
    ref_dec.eval(), ref_head.eval()
    g = torch.Generator().manual_seed(seed + 2)
    feats = {
        "res1": torch.randn(1, 256, 16, 24, generator=g),
        "res2": torch.randn(1, 512, 8, 12, generator=g),
        "res3": torch.randn(1, 1024, 4, 6, generator=g),
        "res4": torch.randn(1, 2048, 2, 3, generator=g),
    }
    with torch.no_grad():
        h1 = ref_dec(feats)
        score, geo = ref_head(h1)
    np.savez_compressed(os.path.join(HERE, "east_decoder_head.npz"), seed=seed,
                        h1=h1.numpy(), score=score.numpy(), geo=geo.numpy())
    print("east_decoder_head", h1.shape, float(h1.abs().mean()), float(score.mean()))


def gen_trba():
    rt = _refload.ref_trba_model()
    out = {}
    seed = 4321
    for tag, (B, h, w, max_len) in {"b4_32x100": (4, 32, 100, 25), "b2_64x256": (2, 64, 256, 25)}.items():
        torch.manual_seed(seed)
        ref = rt.TRBAModel(num_classes=194, blank_id=None)
        torch.manual_seed(seed)
        mine = otm.TRBANet(194, 256)
        sa, sb = ref.state_dict(), mine.state_dict()
        assert list(sa) == list(sb), [k for k in sa if k not in sb]
        assert all(torch.equal(sa[k], sb[k]) for k in sa), "seeded construction differs from reference"
        ref.load_state_dict(synth.trba_state_dict(194, 256, seed=seed), strict=True)
        ref.eval()
        crops = synth.synth_crops(seed + 2, B, h, w)
        x = torch.from_numpy(((crops.astype(np.float32) - 127.5) * np.float32(1 / 127.5)).transpose(0, 3, 1, 2).copy())
        with torch.no_grad():
            f = ref.cnn(x)
            enc = ref.encode(x)
            gl, gi = ref(x, is_train=False, batch_max_length=max_len, mode="greedy")
            bl, bi = ref(x, is_train=False, batch_max_length=max_len, mode="beam", beam_size=8, alpha=0.9, temperature=1.7)
            bl5, bi5 = ref(x, is_train=False, batch_max_length=max_len, mode="beam", beam_size=5, alpha=0.0, temperature=1.0)
        out[f"{tag}_cnn"] = f.numpy()
        out[f"{tag}_enc"] = enc.numpy()
        out[f"{tag}_greedy_logits"], out[f"{tag}_greedy_ids"] = gl.numpy(), gi.numpy()
        out[f"{tag}_beam_logits"], out[f"{tag}_beam_ids"] = bl.numpy(), bi.numpy()
        out[f"{tag}_beam5_logits"], out[f"{tag}_beam5_ids"] = bl5.numpy(), bi5.numpy()
        print("trba", tag, f.shape, enc.shape, gl.shape, bl.shape, "enc|mean|", float(enc.abs().mean()),
              "greedy ids", gi[0, :8].tolist(), "beam ids", bi[0, :8].tolist())
    out["seed"] = seed
    np.savez_compressed(os.path.join(HERE, "trba.npz"), **out)


def gen_pipeline_glue():
    utils = _refload.ref_east_utils()
    rng = np.random.default_rng(77)
    cases = []
    fixed = [
        [(10, 10, 50, 30), (60, 10, 100, 30), (10, 50, 50, 70)],
        [(10, 10, 55, 30), (50, 10, 100, 30)],
        [(0, 0, 100, 100), (10, 10, 90, 90), (20, 20, 80, 80)],
        [],
        [(5, 5, 6, 6)],
        [(10, 10, 60, 40), (10, 10, 60, 40), (70, 12, 120, 38)],
    ]
    for boxes in fixed:
        cases.append(boxes)
    for n in (8, 40, 150):
        rects = synth.synth_layout(int(rng.integers(1 << 30)), 600, 900)
        idx = rng.permutation(len(rects))[:n]
        b = []
        for r in rects[idx]:
            j = rng.integers(-6, 7, size=4)
            b.append(tuple(int(v) for v in (r + j)))
        cases.append(b)
    res = []
    for boxes in cases:
        boxes_np = [tuple(np.int32(v) for v in b) for b in boxes]  # the pipeline passes np.int32 tuples
        res.append({
            "boxes": [list(map(int, b)) for b in boxes],
            "resolved": [list(map(int, b)) for b in utils.resolve_intersections(boxes_np)],
            "sorted": [list(map(int, b)) for b in utils.sort_boxes_reading_order(boxes_np)],
            "sorted_res": [list(map(int, b)) for b in utils.sort_boxes_reading_order_with_resolutions(boxes_np)],
        })
    with open(os.path.join(HERE, "pipeline_glue.json"), "w") as f:
        json.dump(res, f)
    print("pipeline_glue", len(res), "cases")


def gen_box_tail():
    """EAST box tail of infer.py:134-182,216-233 (plain NumPy; the contained-box filter :184-214 needs
    cv2.pointPolygonTest and stays unpinned).  EAST.__new__ skips the constructor (weights download, torchvision)."""
    infer = _refload.ref_east_infer()
    utils = sys.modules["refman.detectors._east.utils"]
    lanms = sys.modules["refman.detectors._east.lanms"]
    rng = np.random.default_rng(2026)
    out = {}
    sets = {}
    # (a) a real page: decode -> LANMS -> expand, M > 30, plus one huge outlier box
    H, W, seed = 192, 256, 31
    rects = synth.synth_layout(seed, H * 4, W * 4)
    score, geo = synth.synth_maps(rects, (H * 4, W * 4), (H, W), seed)
    q = utils.expand_boxes(lanms.locality_aware_nms(utils.decode_quads_from_maps(score, geo, 0.6, 4.0, 2), 0.2), 0.9, 0.9)
    big = np.array([[3, 5, 1000.5, 7, 1001, 700.25, 2, 699, 0.77]], dtype=np.float32)
    sets["page_outlier"] = np.vstack([q[:20], big, q[20:]]).astype(np.float32)
    sets["page_plain"] = q.astype(np.float32)
    # (b) M <= 30 (filter inactive even with an outlier), M == 31 boundary, empty, single
    sets["few_outlier"] = np.vstack([q[:29], big]).astype(np.float32)
    sets["m31_outlier"] = np.vstack([q[:30], big]).astype(np.float32)
    sets["empty"] = np.zeros((0, 9), dtype=np.float32)
    sets["single"] = q[:1].astype(np.float32)
    # (c) identical areas (std == 0 -> untouched) and random rotated / self-intersecting quads with fractional coords
    sq = np.tile(np.array([[0, 0, 16, 0, 16, 8, 0, 8, 0.9]], dtype=np.float32), (40, 1))
    sq[:, 0:8:2] += (np.arange(40, dtype=np.float32) * 32)[:, None]
    sets["equal_areas"] = sq
    rq = rng.uniform(-50, 1500, size=(200, 9)).astype(np.float32)
    rq[:, 8] = rng.uniform(0.6, 1.0, 200).astype(np.float32)
    sets["random200"] = rq
    heavy = rng.uniform(0, 60, size=(64, 9)).astype(np.float32)      # several outliers, sigma 2
    heavy[::16, :8] *= 40
    sets["multi_outlier"] = heavy
    cfgs = {"default": (1280, True, 5.0, 30), "t1024_s2": (1024, True, 2.0, 30), "off": (1280, False, 5.0, 30),
            "min10_s3": (1536, True, 3.0, 10)}
    origs = [(1536, 2048), (4250, 5390), (720, 1280)]
    meta = []
    for name, boxes in sets.items():
        out[f"{name}_in"] = boxes
        for cname, (T, rm, sig, mn) in cfgs.items():
            det = infer.EAST.__new__(infer.EAST)
            det.target_size, det.remove_area_anomalies, det.anomaly_sigma_threshold, det.anomaly_min_box_count = T, rm, sig, mn
            for oi, orig in enumerate(origs):
                tag = f"{name}__{cname}__{oi}"
                scaled = det._scale_boxes_to_original(boxes.copy(), orig)
                areas = infer.EAST._polygon_area_batch(scaled[:, :8].reshape(-1, 4, 2))
                kept = det._remove_area_anomalies(scaled)
                aa = det._convert_to_axis_aligned(kept)
                out[f"{tag}_scaled"], out[f"{tag}_areas"], out[f"{tag}_kept"], out[f"{tag}_aligned"] = scaled, areas, kept, aa
                meta.append({"tag": tag, "set": name, "target": T, "remove": rm, "sigma": sig, "min_count": mn, "orig": list(orig),
                             "n_in": int(len(boxes)), "n_kept": int(len(kept))})
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "box_tail.npz"), **out)
    print("box_tail", len(meta), "cases;", {m["tag"]: (m["n_in"], m["n_kept"]) for m in meta if m["n_in"] != m["n_kept"]})


def _page_image(seed, h, w):
    return np.random.default_rng(seed).integers(0, 256, size=(h, w, 3), dtype=np.uint8)


def gen_pipeline_order():
    """The reference's Pipeline.predict (_pipeline.py:56-176) run with stand-in plugins that follow its plugin protocol
    (tests/test_pipeline_api_compatibility.py:15-93: a detector returning {"page": Page}, a recogniser returning one
    {"text","confidence"} per crop).  Records what the reference did with each page: the new word order, which words got a
    crop, the crops' shapes and CRC32s, and the text given to every word."""
    import zlib
    pl = _refload.ref_pipeline()
    T = sys.modules["refman.detectors._types"]
    rng = np.random.default_rng(99)

    def quad(x0, y0, x1, y1):
        return [[x0, y0], [x1, y0], [x1, y1], [x0, y1]]

    pages = []
    # 1 plain three words (the reference's own DummyDetector page)
    pages.append((60, 320, [quad(10.0, 10.0, 100.0, 50.0), quad(110.0, 10.0, 200.0, 50.0), quad(210.0, 10.0, 300.0, 50.0)], 5))
    # 2 duplicates + an overlapping pair + fractional coordinates (int32 truncation) + a rotated quad
    pages.append((200, 400, [quad(10.7, 10.2, 60.9, 40.5), quad(10.1, 10.9, 60.2, 40.1), quad(55.5, 12.5, 120.5, 38.5),
                             [[150.2, 20.8], [230.6, 10.3], [236.1, 44.9], [155.7, 55.4]], quad(10.0, 80.0, 90.0, 110.0),
                             quad(10.0, 80.0, 90.0, 110.0), quad(100.0, 84.0, 180.0, 114.0)], 5))
    # 3 sub-min_text_size (w or h < 5), exactly 5, out-of-page (negative / beyond the borders / fully outside)
    pages.append((120, 300, [quad(5.0, 5.0, 9.0, 40.0), quad(20.0, 5.0, 25.0, 10.0), quad(40.0, 5.0, 100.0, 9.9),
                             quad(-20.5, -10.5, 30.0, 30.0), quad(250.0, 80.0, 340.0, 150.0), quad(400.0, 10.0, 460.0, 40.0),
                             quad(120.0, 50.0, 200.0, 80.0), quad(-50.0, 60.0, -10.0, 90.0)], 5))
    # 4 nested boxes (resolve_intersections shrinks repeatedly), min_text_size 12
    pages.append((300, 300, [quad(0.0, 0.0, 200.0, 200.0), quad(10.0, 10.0, 190.0, 190.0), quad(20.0, 20.0, 180.0, 180.0),
                             quad(210.0, 5.0, 290.0, 40.0), quad(215.0, 30.0, 295.0, 70.0), quad(50.0, 220.0, 61.0, 231.0)], 12))
    # 5 empty page, 6 all words too small (recogniser must not be called)
    pages.append((50, 50, [], 5))
    pages.append((50, 50, [quad(1.0, 1.0, 4.0, 30.0), quad(10.0, 10.0, 40.0, 13.0)], 5))
    # 7-8 dense synthetic layouts with jitter (overlaps between neighbours), 150 and 400 words
    for n, (h, w) in ((150, (600, 900)), (400, (1200, 1600))):
        rects = synth.synth_layout(int(rng.integers(1 << 30)), h, w)
        idx = rng.permutation(len(rects))[:n]
        polys = []
        for r in rects[idx]:
            j = rng.uniform(-7, 7, size=4)
            x0, y0, x1, y1 = (r + j).tolist()
            polys.append(quad(round(x0, 2), round(y0, 2), round(x1, 2), round(y1, 2)))
        pages.append((h, w, polys, 5))

    class Det:
        def __init__(self, polys):
            self.words = [T.Word(polygon=p, detection_confidence=0.5) for p in polys]

        def predict(self, image, vis=False, profile=False):
            return {"page": T.Page(blocks=[T.Block(words=list(self.words))]), "vis_image": None, "score_map": None, "geo_map": None}

    class Rec:
        def __init__(self):
            self.calls = []

        def predict(self, images):
            self.calls.append([(list(im.shape), zlib.crc32(np.ascontiguousarray(im).tobytes())) for im in images])
            return [{"text": f"w{i}", "confidence": 1.0 / (1 + i)} for i in range(len(images))]

    res = []
    for pi, (h, w, polys, mts) in enumerate(pages):
        img = _page_image(1000 + pi, h, w)
        det, rec = Det(polys), Rec()
        page = pl.Pipeline(detector=det, recognizer=rec, min_text_size=mts).predict(img)
        ident = {id(wd): i for i, wd in enumerate(det.words)}
        words = page.blocks[0].words
        res.append({
            "h": h, "w": w, "image_seed": 1000 + pi, "min_text_size": mts, "polygons": polys,
            "order": [ident[id(wd)] for wd in words],
            "texts": [wd.text for wd in words],
            "rec_calls": len(rec.calls),
            "crops": [[s, int(c)] for s, c in (rec.calls[0] if rec.calls else [])],
        })
        print("pipeline_order page", pi, len(polys), "words ->", len(words), "ordered,", len(res[-1]["crops"]), "crops")
    with open(os.path.join(HERE, "pipeline_order.json"), "w") as f:
        json.dump(res, f)


def gen_trba_post():
    """The reference's TRBA.predict (recognizers/_trba/__init__.py:374-432) — chunking, log_softmax, decode_tokens and the
    confidence — run on pre-sized canvases.  TRBA.__new__ skips the constructor (weight download); `_preprocess_image` is
    replaced by the Normalize + CHW step alone (A.Normalize(mean=.5,std=.5,max_pixel_value=255) restated; the cv2 resize is
    a no-op for canvases that already have the network size), so what is pinned is everything after preprocessing."""
    tw = _refload.ref_trba_wrapper()
    seed = 4321
    charset = os.path.join(ROOT, "manuscript_ocr_amd", "recognizers", "_trba", "configs", "charset.txt")
    ref_charset = "/root/reference/src/manuscript/recognizers/_trba/configs/charset.txt"
    assert open(charset, encoding="utf-8").read() == open(ref_charset, encoding="utf-8").read()
    out = {}
    for tag, (B, h, w, max_len) in {"b7_32x100": (7, 32, 100, 25), "b3_64x256": (3, 64, 256, 25)}.items():
        torch.manual_seed(seed)
        model = tw.TRBAModel(num_classes=194, blank_id=None)
        model.load_state_dict(synth.trba_state_dict(194, 256, seed=seed), strict=True)
        model.eval()
        rec = tw.TRBA.__new__(tw.TRBA)
        rec.model, rec.max_length, rec.device = model, max_len, torch.device("cpu")
        rec.itos, rec.stoi = tw.load_charset(ref_charset)
        rec.pad_id, rec.sos_id, rec.eos_id = rec.stoi["<PAD>"], rec.stoi["<SOS>"], rec.stoi["<EOS>"]
        rec.blank_id = rec.stoi.get("<BLANK>", None)
        rec._preprocess_image = lambda im: torch.from_numpy(
            ((im.astype(np.float32) - 127.5) * np.float32(1 / 127.5)).transpose(2, 0, 1).copy()).unsqueeze(0)
        crops = synth.synth_crops(seed + 5, B, h, w)
        for mode, kw in (("greedy", {}), ("beam", {}), ("beam", {"beam_size": 5, "temperature": 1.0, "alpha": 0.0})):
            for bs in (32, 2, 3):
                r = rec.predict(list(crops), batch_size=bs, mode=mode, **kw)
                key = f"{tag}_{mode}{kw.get('beam_size', 8) if mode == 'beam' else ''}_bs{bs}"
                out[key + "_text"] = np.array([x["text"] for x in r])
                out[key + "_conf"] = np.array([x["confidence"] for x in r], dtype=np.float64)
                print("trba_post", key, [x["text"][:6] for x in r][:3], out[key + "_conf"][:3])
    # the post-step alone on the logits / ids that trba.npz already holds (one chunk of 4)
    g = np.load(os.path.join(HERE, "trba.npz"))
    crops4 = synth.synth_crops(seed + 2, 4, 32, 100)
    torch.manual_seed(seed)
    model = tw.TRBAModel(num_classes=194, blank_id=None)
    model.load_state_dict(synth.trba_state_dict(194, 256, seed=seed), strict=True)
    model.eval()
    rec.model, rec.max_length = model, 25
    for mode in ("greedy", "beam"):
        r = rec.predict(list(crops4), batch_size=32, mode=mode)
        out[f"npz_b4_32x100_{mode}_text"] = np.array([x["text"] for x in r])
        out[f"npz_b4_32x100_{mode}_conf"] = np.array([x["confidence"] for x in r], dtype=np.float64)
    out["seed"] = seed
    np.savez_compressed(os.path.join(HERE, "trba_post.npz"), **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["lanms", "east_post", "east_decoder_head", "trba", "pipeline_glue", "box_tail", "pipeline_order", "trba_post"]
    for w in which:
        globals()["gen_" + w]()
