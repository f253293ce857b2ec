"""Pipeline — drop-in for the reference orchestration
(/root/reference/src/manuscript/_pipeline.py:17-221): detect -> reading-order sort -> crop ->
recognise -> annotate, behind the same duck-typed plugin protocol
(docs/PIPELINE_API.md: detector.predict(image, vis=False, profile=...) -> {"page"}|tuple|Page,
recognizer.predict(List[np.ndarray]) -> [{"text","confidence"}|(text, conf)|other]).

`process_batch` is broken upstream (it calls a non-existent `self.process`, _pipeline.py:187);
here it is the per-image `predict`.  `predict_batch` is the MI355X fast path: when detector and
recogniser are this package's EAST/TRBA it runs the detector once for all pages and the
recogniser once for all crops of all pages (results identical to per-page predict).
"""
import time
from typing import List, Optional, Union

import numpy as np
from PIL import Image

from .detectors import EAST, read_image, sort_boxes_reading_order_with_resolutions, visualize_page
from .recognizers import TRBA


def _word_aabb(word):
    poly = np.array(word.polygon, dtype=np.int32)  # float -> int32 truncation (reference :106)
    x_min, y_min = np.min(poly, axis=0)
    x_max, y_max = np.max(poly, axis=0)
    return (x_min, y_min, x_max, y_max), poly


class Pipeline:
    def __init__(self, detector: Optional[EAST] = None, recognizer: Optional[TRBA] = None, min_text_size: int = 5):
        self.detector = detector if detector is not None else EAST()
        self.recognizer = recognizer if recognizer is not None else TRBA()
        self.min_text_size = min_text_size

    # ------------------------------------------------------------------------------------- helpers
    @staticmethod
    def _page_of(det_out):
        if isinstance(det_out, dict):
            page = det_out.get("page")
        elif isinstance(det_out, tuple):
            page = det_out[0]
        else:
            page = det_out
        if page is None:
            raise RuntimeError("Detector did not return a Page result.")
        return page

    def _order_and_crop(self, page, image_array):
        """_pipeline.py:102-137: reorder every block in reading order, collect crops of words >= min_text_size."""
        words, crops = [], []
        for block in page.blocks:
            boxes = [_word_aabb(w)[0] for w in block.words]
            first = {}
            for k, bx in enumerate(boxes):  # first equal word wins, as the reference's tuple comparison (:113-121)
                first.setdefault(tuple(int(v) for v in bx), k)
            block.words = [block.words[first[tuple(int(v) for v in bx)]] for bx in sort_boxes_reading_order_with_resolutions(boxes)]
            for word in block.words:
                (x0, y0, x1, y1), poly = _word_aabb(word)
                if (x1 - x0) >= self.min_text_size and (y1 - y0) >= self.min_text_size:
                    region = self._extract_word_image(image_array, poly)
                    if region is not None and region.size > 0:
                        words.append(word)
                        crops.append(region)
        return words, crops

    @staticmethod
    def _assign(words, results):
        for word, result in zip(words, results):
            if isinstance(result, dict):
                text, confidence = result.get("text", ""), result.get("confidence", None)
            elif isinstance(result, tuple) and len(result) == 2:
                text, confidence = result
            else:
                text, confidence = (str(result) if result is not None else ""), None
            word.text = text
            word.recognition_confidence = confidence

    # ------------------------------------------------------------------------------------- API
    def predict(self, image: Union[str, np.ndarray, Image.Image], recognize_text: bool = True, vis: bool = False,
                profile: bool = False):
        start = time.time()
        t0 = time.time()
        page = self._page_of(self.detector.predict(image, vis=False, profile=profile))
        if profile:
            print(f"Detection: {time.time() - t0:.3f}s")
        if not recognize_text:
            if vis:
                arr = read_image(image)
                pil = image if isinstance(image, Image.Image) else Image.fromarray(arr)
                return page, visualize_page(pil, page, show_order=False)
            return page
        t0 = time.time()
        image_array = read_image(image)
        if profile:
            print(f"Load image for crops: {time.time() - t0:.3f}s")
        t0 = time.time()
        words, crops = self._order_and_crop(page, image_array)
        if profile:
            print(f"Extract {len(crops)} crops: {time.time() - t0:.3f}s")
        if crops:
            t0 = time.time()
            results = self.recognizer.predict(crops)
            if profile:
                print(f"Recognition: {time.time() - t0:.3f}s")
            self._assign(words, [results[i] for i in range(len(words))])
        if profile:
            print(f"Pipeline total: {time.time() - start:.3f}s")
        if vis:
            pil = image if isinstance(image, Image.Image) else Image.fromarray(image_array)
            return page, visualize_page(pil, page, show_order=True)
        return page

    def predict_batch(self, images: List[Union[str, np.ndarray]], recognize_text: bool = True, profile: bool = False,
                      pages_dev=None, _maps_override=None):
        """Equally sized pages -> list of Pages, same results as per-page `predict`.

        MI355X fast path (detector and recogniser are this package's EAST / TRBA): the pages are uploaded once,
        the detector runs once for the batch, word crops are cut, resized and padded ON THE DEVICE from the resident
        pages (no host crop, no per-crop upload) and the recogniser runs once over all crops of all pages — while the
        decode run lengths that enter the confidences still follow the reference's per-page / per-`batch_size` chunking.
        `pages_dev`: optional [N,H,W,3] u8 device tensor already holding `images` (benchmarks: inputs resident in HBM)."""
        native = isinstance(self.detector, EAST) and isinstance(self.recognizer, TRBA)
        if not native:
            return [self.predict(im, recognize_text=recognize_text, profile=profile) for im in images]
        import torch

        from . import ops
        tm = {}
        t0 = time.perf_counter()
        arrays = [read_image(im) for im in images]
        if pages_dev is None:
            pages_dev = torch.from_numpy(np.ascontiguousarray(np.stack(arrays))).to(self.detector.device)
        results = self.detector.predict_batch(arrays, profile=profile, _pages_dev=pages_dev, _maps_override=_maps_override)
        pages = [self._page_of(r) for r in results]
        tm["detect"] = time.perf_counter() - t0
        self.last_profile = tm
        if not recognize_text:
            return pages
        t0 = time.perf_counter()
        rec = self.recognizer
        all_words, boxes, page_ids, spans = [], [], [], []
        for pi, page in enumerate(pages):
            n0 = len(all_words)
            for block in page.blocks:
                if not block.words:
                    continue
                # AABBs of all words at once: np.array(polygon, int32) truncates toward zero (_pipeline.py:106)
                polys = np.array([w.polygon for w in block.words], dtype=np.float64).astype(np.int32)
                mins, maxs = polys.min(axis=1), polys.max(axis=1)
                aabbs = [(a[0], a[1], b[0], b[1]) for a, b in zip(mins, maxs)]
                first = {}
                for k, bx in enumerate(aabbs):  # "first equal word wins" of the reference's O(n^2) re-match (:113-121)
                    first.setdefault(tuple(int(v) for v in bx), k)
                order = [first[tuple(int(v) for v in bx)] for bx in sort_boxes_reading_order_with_resolutions(aabbs)]
                old_words = block.words
                block.words = [old_words[k] for k in order]
                for k in order:
                    x0, y0, x1, y1 = aabbs[k]
                    if (x1 - x0) >= self.min_text_size and (y1 - y0) >= self.min_text_size:
                        all_words.append(old_words[k])
                        boxes.append((x0, y0, x1, y1))
                        page_ids.append(pi)
            spans.append([n0, len(all_words) - n0])
        tm["order"] = time.perf_counter() - t0
        if not boxes:
            return pages
        t0 = time.perf_counter()
        H, W = arrays[0].shape[:2]
        desc, keep = ops.crop_descriptors(boxes, page_ids, (H, W), rec.img_h, rec.img_w)
        if not keep.all():  # empty clamped crops are skipped by the reference (_pipeline.py:135)
            all_words = [w for w, k in zip(all_words, keep) if k]
            kept_pages = np.asarray(page_ids)[keep]
            spans, n0 = [], 0
            for pi in range(len(pages)):
                c = int((kept_pages == pi).sum())
                spans.append([n0, c])
                n0 += c
        if len(desc):
            canv = ops.crop_resize_pad(pages_dev, desc, rec.img_h, rec.img_w)
            tm["crop"] = time.perf_counter() - t0
            t0 = time.perf_counter()
            out = rec.recognize_canvases(canv, spans=[tuple(s) for s in spans if s[1] > 0])
            tm["recognize"] = time.perf_counter() - t0
            t0 = time.perf_counter()
            self._assign(all_words, rec._results(*out))
            tm["assign"] = time.perf_counter() - t0
        if profile:
            print("Pipeline.predict_batch stages (s):", {k: round(v, 4) for k, v in tm.items()})
        return pages

    def process_batch(self, images: List[Union[str, np.ndarray, Image.Image]], recognize_text: bool = True, vis: bool = False,
                      profile: bool = False):
        results = []
        for img in images:
            res = self.predict(img, recognize_text=recognize_text, vis=vis, profile=profile)
            results.append(res[0] if vis else res)
        return results

    def get_text(self, page) -> str:
        lines = []
        for block in page.blocks:
            ordered = sorted(block.words, key=lambda w: min(p[0] for p in w.polygon))
            texts = [w.text for w in ordered if getattr(w, "text", None)]
            if texts:
                lines.append(" ".join(texts))
        return "\n".join(lines)

    def _extract_word_image(self, image: np.ndarray, polygon: np.ndarray) -> Optional[np.ndarray]:
        try:
            x_min, y_min = np.min(polygon, axis=0)
            x_max, y_max = np.max(polygon, axis=0)
            h, w = image.shape[:2]
            x1, y1 = max(0, int(x_min)), max(0, int(y_min))
            x2, y2 = min(w, int(x_max)), min(h, int(y_max))
            region = image[y1:y2, x1:x2]  # a view of the page, never mutated
            return region if region.size > 0 else None
        except Exception:
            return None
