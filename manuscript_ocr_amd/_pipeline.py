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
            new_order = []
            for bx in sort_boxes_reading_order_with_resolutions(boxes):
                for w, wb in zip(block.words, boxes):  # first equal word wins, as the reference's tuple comparison
                    if wb == bx:
                        new_order.append(w)
                        break
            block.words = new_order
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

    def predict_batch(self, images: List[Union[str, np.ndarray]], recognize_text: bool = True, profile: bool = False):
        """Equally sized pages -> list of Pages: one detector pass for the batch, one recogniser pass for all crops."""
        if not (hasattr(self.detector, "predict_batch") and isinstance(self.detector, EAST)):
            return [self.predict(im, recognize_text=recognize_text, profile=profile) for im in images]
        arrays = [read_image(im) for im in images]
        pages = [self._page_of(r) for r in self.detector.predict_batch(arrays, profile=profile)]
        if not recognize_text:
            return pages
        all_words, all_crops, spans = [], [], []
        for page, arr in zip(pages, arrays):
            words, crops = self._order_and_crop(page, arr)
            spans.append((len(all_words), len(words)))
            all_words += words
            all_crops += crops
        if all_crops:
            # per page the reference calls recognizer.predict(crops) separately: its batch_size chunks (and hence the
            # decode run length that enters the confidences) restart at every page -> keep that chunking.
            results = []
            for s, n in spans:
                if n:
                    results += self.recognizer.predict(all_crops[s:s + n])
            self._assign(all_words, results)
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
