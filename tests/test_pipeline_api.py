"""Plugin-boundary contract of Pipeline with fake detector / recogniser — the cases of the reference's
tests/test_pipeline_api_compatibility.py:99-321, restated against manuscript_ocr_amd (CPU only)."""
import numpy as np
import pytest
from PIL import Image

from manuscript_ocr_amd import Pipeline
from manuscript_ocr_amd.detectors._types import Block, Page, Word


def _words():
    return [
        Word(polygon=[[10.0, 10.0], [100.0, 10.0], [100.0, 50.0], [10.0, 50.0]], detection_confidence=0.95),
        Word(polygon=[[110.0, 10.0], [200.0, 10.0], [200.0, 50.0], [110.0, 50.0]], detection_confidence=0.92),
        Word(polygon=[[210.0, 10.0], [300.0, 10.0], [300.0, 50.0], [210.0, 50.0]], detection_confidence=0.88),
    ]


class DummyDetector:
    def __init__(self, return_type="dict", words=None):
        self.return_type, self.words, self.calls = return_type, words, []

    def predict(self, image, vis=False, profile=False):
        self.calls.append((vis, profile))
        page = Page(blocks=[Block(words=self.words if self.words is not None else _words())])
        if self.return_type == "dict":
            return {"page": page, "vis_image": None, "score_map": None, "geo_map": None}
        if self.return_type == "tuple":
            return (page, None)
        if self.return_type == "none":
            return {"page": None}
        return page


class DummyRecognizer:
    def __init__(self, kind="dict"):
        self.call_count, self.kind, self.seen = 0, kind, None

    def predict(self, images):
        self.call_count += 1
        self.seen = images
        if self.kind == "tuple":
            return [(f"word{i + 1}", 0.9 - i * 0.05) for i in range(len(images))]
        if self.kind == "str":
            return [f"word{i + 1}" for i in range(len(images))]
        return [{"text": f"word{i + 1}", "confidence": 0.9 - i * 0.05} for i in range(len(images))]


IMG = np.zeros((100, 400, 3), dtype=np.uint8)


@pytest.mark.parametrize("rt", ["dict", "tuple", "page"])
def test_detector_return_shapes(rt):
    rec = DummyRecognizer()
    res = Pipeline(detector=DummyDetector(rt), recognizer=rec).predict(IMG, recognize_text=True, vis=False)
    assert isinstance(res, Page) and len(res.blocks) == 1 and len(res.blocks[0].words) == 3
    assert [w.text for w in res.blocks[0].words] == ["word1", "word2", "word3"]
    assert res.blocks[0].words[0].recognition_confidence == 0.9
    assert rec.call_count == 1 and len(rec.seen) == 3 and rec.seen[0].shape == (40, 90, 3)


def test_detector_called_with_vis_false_and_profile_passthrough():
    det = DummyDetector()
    Pipeline(detector=det, recognizer=DummyRecognizer()).predict(IMG, vis=False, profile=False)
    assert det.calls == [(False, False)]


def test_none_page_raises():
    with pytest.raises(RuntimeError, match="Detector did not return a Page result."):
        Pipeline(detector=DummyDetector("none"), recognizer=DummyRecognizer()).predict(IMG)


@pytest.mark.parametrize("kind,conf", [("tuple", 0.9), ("str", None)])
def test_recognizer_result_shapes(kind, conf):
    res = Pipeline(detector=DummyDetector(), recognizer=DummyRecognizer(kind)).predict(IMG)
    assert res.blocks[0].words[0].text == "word1" and res.blocks[0].words[0].recognition_confidence == conf


def test_without_recognition():
    rec = DummyRecognizer()
    res = Pipeline(detector=DummyDetector(), recognizer=rec).predict(IMG, recognize_text=False)
    assert rec.call_count == 0 and res.blocks[0].words[0].text is None


def test_visualization_returns_pil():
    res, vis = Pipeline(detector=DummyDetector(), recognizer=DummyRecognizer()).predict(IMG, recognize_text=True, vis=True)
    assert isinstance(res, Page) and isinstance(vis, Image.Image)
    res, vis = Pipeline(detector=DummyDetector(), recognizer=DummyRecognizer()).predict(IMG, recognize_text=False, vis=True)
    assert isinstance(vis, Image.Image)


def test_get_text_and_min_text_size():
    p = Pipeline(detector=DummyDetector(), recognizer=DummyRecognizer())
    text = p.get_text(p.predict(IMG))
    assert text == "word1 word2 word3"
    small = [Word(polygon=[[10.0, 10.0], [12.0, 10.0], [12.0, 12.0], [10.0, 12.0]], detection_confidence=0.95)]
    rec = DummyRecognizer()
    Pipeline(detector=DummyDetector(words=small), recognizer=rec, min_text_size=5).predict(IMG)
    assert rec.call_count == 0


def test_reading_order_and_process_batch():
    words = list(reversed(_words())) + [Word(polygon=[[10.0, 60.0], [100.0, 60.0], [100.0, 95.0], [10.0, 95.0]], detection_confidence=0.5)]
    p = Pipeline(detector=DummyDetector(words=words), recognizer=DummyRecognizer())
    page = p.predict(IMG)
    xs = [w.polygon[0] for w in page.blocks[0].words]
    assert xs == [(10.0, 10.0), (110.0, 10.0), (210.0, 10.0), (10.0, 60.0)]
    pages = p.process_batch([IMG, IMG])
    assert len(pages) == 2 and all(isinstance(x, Page) for x in pages)
    assert p.min_text_size == 5 and p.detector is not None and p.recognizer is not None


def test_public_signatures_match_the_reference_kwargs():
    """Drop-in contract: constructor / predict keyword names, order and defaults of the reference's public API
    (detectors/_east/infer.py:28-43,235-241; recognizers/_trba/__init__.py:37-44,290-299; _pipeline.py:18-24,56-62), followed
    only by this package's keyword-only extras."""
    import inspect

    from manuscript_ocr_amd import Pipeline
    from manuscript_ocr_amd.detectors import EAST
    from manuscript_ocr_amd.recognizers import TRBA

    def params(fn):
        return [(n, p.default) for n, p in inspect.signature(fn).parameters.items()
                if n != "self" and p.kind in (p.POSITIONAL_OR_KEYWORD,)]

    assert params(EAST.__init__) == [
        ("weights_path", None), ("device", None), ("target_size", 1280), ("expand_ratio_w", 0.9), ("expand_ratio_h", 0.9),
        ("score_thresh", 0.6), ("iou_threshold", 0.2), ("score_geo_scale", 0.25), ("quantization", 2),
        ("axis_aligned_output", True), ("remove_area_anomalies", True), ("anomaly_sigma_threshold", 5.0),
        ("anomaly_min_box_count", 30)]
    assert params(EAST.predict) == [("img_or_path", inspect.Parameter.empty), ("vis", False), ("profile", False),
                                    ("return_maps", False), ("sort_reading_order", False)]
    assert params(TRBA.__init__) == [("model_path", None), ("charset_path", None), ("config_path", None), ("device", "auto")]
    assert params(TRBA.predict) == [("images", inspect.Parameter.empty), ("batch_size", 32), ("mode", "beam"), ("beam_size", 8),
                                    ("temperature", 1.7), ("alpha", 0.9)]
    assert params(Pipeline.__init__) == [("detector", None), ("recognizer", None), ("min_text_size", 5)]
    assert params(Pipeline.predict) == [("image", inspect.Parameter.empty), ("recognize_text", True), ("vis", False), ("profile", False)]
    assert [n for n, _ in params(Pipeline.get_text)] == ["page"]
