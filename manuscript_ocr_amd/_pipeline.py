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
import contextlib
import gc
import time
from typing import List, Optional, Union

import numpy as np
from PIL import Image

from .detectors import EAST, read_image, sort_boxes_reading_order_with_resolutions, visualize_page
from .recognizers import TRBA


@contextlib.contextmanager
def _gc_paused():
    """The batch path creates ~10^5 small container objects per step (Words, polygons, result rows); with the cyclic GC
    enabled, generation-2 passes over them land in the middle of the enqueue path (measured ~10 ms per group).  Nothing
    here forms reference cycles, so reference counting alone frees everything; the collector's state is restored on exit."""
    was = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()


def _reading_order(aabbs_i32):
    """Indices of the words in reading order: sort_boxes_reading_order_with_resolutions + the reference's "first word with an
    equal box" re-match (_pipeline.py:113-121), evaluated by the host helper msocr_reading_order_host (same arithmetic)."""
    from . import _native as nat
    boxes = np.ascontiguousarray(aabbs_i32, dtype=np.int32)
    order = np.empty(len(boxes), dtype=np.int32)
    nat.check(nat.lib().msocr_reading_order_host(boxes.ctypes.data, len(boxes), 0.6, float("inf"), order.ctypes.data), "reading_order_host")
    return order.tolist()


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
        if (isinstance(self.detector, EAST) and isinstance(self.recognizer, TRBA) and recognize_text and not profile
                and getattr(self, "native_fast_path", True)):
            # both plugins are this package's: same result through the device path (crops cut, resized and padded on the
            # device from the uploaded page, one recogniser pass) — tests/test_gpu_pipeline.py pins batch == per-page
            page = self.predict_batch([image])[0]  # a JPEG path is decoded on the device (ingest.py), arrays are uploaded
            if vis:
                pil = image if isinstance(image, Image.Image) else Image.fromarray(read_image(image))
                return page, visualize_page(pil, page, show_order=True)
            return page
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

    def _order_boxes(self, page):
        """Reading-order reorder of every block (in place) + AABBs of the words that pass min_text_size.
        Same semantics as _pipeline.py:102-133 of the reference, vectorised."""
        words, boxes = [], []
        for block in page.blocks:
            if not block.words:
                continue
            # AABBs of all words at once: np.array(polygon, int32) truncates toward zero (_pipeline.py:106)
            polys = np.array([w.polygon for w in block.words], dtype=np.float64).astype(np.int32)
            mins, maxs = polys.min(axis=1), polys.max(axis=1)
            aabbs = [(a[0], a[1], b[0], b[1]) for a, b in zip(mins, maxs)]
            order = _reading_order(np.concatenate([mins, maxs], axis=1))
            old_words = block.words
            block.words = [old_words[k] for k in order]
            for k in order:
                x0, y0, x1, y1 = aabbs[k]
                if (x1 - x0) >= self.min_text_size and (y1 - y0) >= self.min_text_size:
                    words.append(old_words[k])
                    boxes.append((x0, y0, x1, y1))
        return words, boxes

    def predict_batch(self, images: List[Union[str, np.ndarray]], recognize_text: bool = True, profile: bool = False,
                      pages_dev=None, sub_batches: int = 0, _maps_override=None):
        """Pages -> list of Pages, same results as per-page `predict`.  Pages of different sizes are processed as one group per
        size (the word crops are cut from the ORIGINAL pages, which only stack when they are equally sized; the detector alone
        batches any mix: EAST.predict_batch), results returned in input order.

        MI355X fast path (detector and recogniser are this package's EAST / TRBA): the pages are uploaded once,
        word crops are cut, resized and padded ON THE DEVICE from the resident pages (no host crop, no per-crop upload)
        and the recogniser runs over all crops of a sub-batch at once — while the decode run lengths that enter the
        confidences still follow the reference's per-page / per-`batch_size` chunking.  The batch is processed as
        `sub_batches` groups on separate HIP streams in a software pipeline, so the host stages of one group (box
        filters, reading order, crop descriptors) overlap the device work of the others.
        `pages_dev`: optional [N,H,W,3] u8 device tensor already holding `images` (benchmarks: inputs resident in HBM).
        = collect_batch(submit_batch(...)); call the two halves yourself to overlap consecutive batches."""
        native = isinstance(self.detector, EAST) and isinstance(self.recognizer, TRBA)
        if not native:
            return [self.predict(im, recognize_text=recognize_text, profile=profile) for im in images]
        if pages_dev is None and len(images) > 1:
            shapes = [self._shape_of(im) for im in images]
            if len(set(shapes)) > 1:
                # ragged batch: every size group's detector work is enqueued before the first group is collected
                groups = {}
                for i, sh in enumerate(shapes):
                    groups.setdefault(sh, []).append(i)
                def maps_of(idx):  # injected maps (tests / benchmarks) follow their pages
                    if _maps_override is None:
                        return None
                    import torch
                    sel = torch.tensor(idx, device=_maps_override[0].device)
                    return (_maps_override[0].index_select(0, sel), _maps_override[1].index_select(0, sel))
                handles = []
                for idx in groups.values():
                    mo = maps_of(idx)  # kept alive in `handles` until the group is collected: the detector streams read it asynchronously
                    handles.append((idx, self.submit_batch([images[i] for i in idx], recognize_text, profile, None, sub_batches, mo), mo))
                out = [None] * len(images)
                for idx, h, _mo in handles:
                    for i, pg in zip(idx, self.collect_batch(h)):
                        out[i] = pg
                return out
        return self.collect_batch(self.submit_batch(images, recognize_text, profile, pages_dev, sub_batches, _maps_override))

    @staticmethod
    def _shape_of(im):
        """(height, width) of a page without decoding it twice: arrays know theirs, files are asked through PIL's header parse."""
        if isinstance(im, np.ndarray):
            return tuple(im.shape[:2])
        try:
            with Image.open(im) as f:
                return (f.height, f.width)
        except Exception:
            return tuple(read_image(im).shape[:2])

    def submit_batch(self, images, recognize_text: bool = True, profile: bool = False, pages_dev=None, sub_batches: int = 0,
                     _maps_override=None, _device_entropy=None):
        """Stage 1 of `predict_batch`: upload (if needed) and enqueue every group's detector work; returns a handle
        without synchronising.  Consecutive submits alternate between two sets of streams, so the detector work of
        batch i+1 can be enqueued before `collect_batch` of batch i and fills the device while batch i drains."""
        if not (isinstance(self.detector, EAST) and isinstance(self.recognizer, TRBA)):
            raise TypeError("submit_batch/collect_batch need this package's EAST and TRBA plugins")
        import torch

        det = self.detector
        # image ingest: a JPEG file is decoded ON THE DEVICE (host Huffman stage + HIP reconstruction, ingest.py) — the page's
        # pixels never exist on the host, `arrays` then only carries the shape; everything else goes through read_image
        arrays, decoded = [], []
        dec, ingest_pending = [None] * len(images), None
        # Which Huffman stage for files with restart intervals: ingest's policy (the device kernel unless an interval is so long that
        # its serial chain loses to a host core; `pipeline.device_entropy = True / False` or MSOCR_JPEG_DEVICE_ENTROPY force one).
        # From files, same box: device stage 79.5-80.1, host pool 78.2-79.2 pages/s against 83.0 resident.
        if _device_entropy is None:
            _device_entropy = getattr(self, "device_entropy", None)
        ing = torch.cuda.current_stream()
        with torch.cuda.stream(ing):
            if pages_dev is None and getattr(self, "device_ingest", True):
                from . import ingest
                # Huffman stage on the device for files with restart intervals, on a host thread pool otherwise; the device stage's
                # verdict on corrupt streams is read in advance_batch (ingest.check_pending), not here
                dec, ingest_pending = ingest.read_images_device(list(images), det.device, device_entropy=_device_entropy, defer_status=True)
            for im, t in zip(images, dec):
                if t is not None:
                    arrays.append(np.broadcast_to(np.uint8(0), tuple(t.shape)))
                else:
                    arrays.append(read_image(im))
                decoded.append(t)
            if len({a.shape for a in arrays}) != 1:
                raise ValueError("predict_batch needs equally sized pages")
            N = len(arrays)
            if pages_dev is None:
                if all(t is not None for t in decoded):
                    pages_dev = torch.stack(decoded)
                elif not any(t is not None for t in decoded):
                    pages_dev = torch.from_numpy(np.ascontiguousarray(np.stack(arrays))).to(det.device)
                else:
                    pages_dev = torch.stack([t if t is not None else torch.from_numpy(np.ascontiguousarray(a)).to(det.device)
                                             for t, a in zip(decoded, arrays)])
        # groups per batch.  Round 1 needed 8 groups of 2 pages to hide its host stages behind other groups' device work; with the
        # reading order on the device and Page assembly off the enqueue path, larger launch sequences win (bigger GEMM M, fewer
        # launches): 16 pages measured 44.0 / 45.4 / 45.7 pages/s at 8 / 4 / 2 groups with 4 hardware queues and 36.3 / 46.8 / 48.1
        # with 8; one group ties two at lower memory, two keeps a second detector sequence in flight
        nsub = sub_batches or (2 if N >= 8 else 1)
        nsub = max(1, min(nsub, N))
        bounds = [(N * k // nsub, N * (k + 1) // nsub) for k in range(nsub)]
        main = torch.cuda.current_stream()
        if getattr(self, "serialize_streams", False):  # profiling aid: same launches, no cross-stream kernel overlap
            streams = det_streams = [main] * nsub
        else:
            # Two stream sets alternate between consecutive batches.  Per group: a HIGH-priority stream for the detector and a
            # normal one for crops + recogniser.  Detection is the short head of a group's work and the host needs its boxes
            # before it can enqueue the long recogniser tail: with equal priorities the detector kernels of batch i+1 share the
            # chip fairly with the recogniser of batch i and finish together with it, so the next recogniser work is enqueued
            # only when the device has already drained (measured: 5 % idle); at high priority they overtake it.
            nsets = max(1, int(getattr(self, "stream_sets", 2)))  # batches that may be in flight at once
            # Every stream keeps its own allocator pool (stream-ordered reuse without waiting for the device), so reserved memory
            # grows with the number of streams in flight x the per-page activation footprint: measured 96 GB at 16 pages x
            # 1536 x 2048 with two stream sets = 1.9 KB per page pixel.  When two sets would not fit comfortably (configs[4]:
            # 16 pages x 3072 x 4096 -> 380 GB) consecutive batches share ONE set of streams: half the memory, still
            # stream-ordered, a little less overlap between batches — instead of an allocator that thrashes at the 288 GB limit.
            H_, W_ = arrays[0].shape[:2]
            if nsets > 1 and 1900.0 * N * H_ * W_ > 0.6 * torch.cuda.get_device_properties(pages_dev.device).total_memory:
                nsets = 1
            if not hasattr(self, "_stream_sets") or len(self._stream_sets) != nsets:
                self._stream_sets, self._det_stream_sets, self._set_idx = [[] for _ in range(nsets)], [[] for _ in range(nsets)], 0
            self._set_idx = (self._set_idx + 1) % nsets
            pool, dpool = self._stream_sets[self._set_idx], self._det_stream_sets[self._set_idx]
            hi_prio = -1 if getattr(self, "det_stream_priority", True) else 0
            while len(pool) < nsub:
                pool.append(torch.cuda.Stream())
                dpool.append(torch.cuda.Stream(priority=hi_prio))
            streams, det_streams = pool[:nsub], dpool[:nsub]
        from . import ops
        det_handles, ro_handles, det_events = [], [], []
        H, W = arrays[0].shape[:2]
        for (lo, hi), st in zip(bounds, det_streams):
            if st is not main:
                st.wait_stream(main)
            with torch.cuda.stream(st):
                mo = None if _maps_override is None else (_maps_override[0][lo:hi], _maps_override[1][lo:hi])
                dh = det.detect_start(pages_dev[lo:hi], mo)
                det_handles.append(dh)
                # reading order + crop descriptors of the group's pages on the device, right behind the box filters: the host
                # then needs only the crop COUNTS to enqueue the recogniser (Page / Word assembly moves to collect_batch)
                ro = None
                if recognize_text and dh[5] is not None and getattr(self, "device_order", True):
                    ro = ops.reading_order_crops(dh[5], dh[6], (H, W), self.min_text_size, self.recognizer.img_h, self.recognizer.img_w,
                                                 page_base=lo)
                ro_handles.append(ro)
                ev = torch.cuda.Event()
                ev.record(st)  # detector outputs of this group complete
                det_events.append(ev)
        return {"arrays": arrays, "pages_dev": pages_dev, "bounds": bounds, "streams": streams, "det_streams": det_streams, "main": main,
                "det_handles": det_handles, "ro_handles": ro_handles, "det_events": det_events, "recognize_text": recognize_text, "profile": profile,
                "ingest_pending": ingest_pending,
                "resubmit": (lambda: self.submit_batch(images, recognize_text, profile, None, sub_batches, _maps_override, _device_entropy=False))}

    def advance_batch(self, h):
        """Stage 2 of `predict_batch` for a handle from `submit_batch`: per group — wait for its boxes, run the host
        tail + reading order, then enqueue device crops + the recogniser (asynchronous).  Idempotent.  Calling it for
        batch i+1 BEFORE `collect_batch` of batch i keeps recogniser work queued on the device while the host
        annotates batch i (bench.py does)."""
        if h.get("groups") is not None:
            return h
        import torch

        from . import ingest, ops
        if h.get("ingest_pending") is not None:
            # the device Huffman stage's verdict on this batch's files (queued right behind the kernels, long done by now): a corrupt
            # stream gets what the host decoder's verdict gives it — the whole batch is read again through the host path (rare)
            bad = ingest.check_pending(h["ingest_pending"])
            h["ingest_pending"] = None
            if bad:
                h2 = h["resubmit"]()
                h.clear()
                h.update(h2)
        det, rec = self.detector, self.recognizer
        arrays, pages_dev, bounds, streams = h["arrays"], h["pages_dev"], h["bounds"], h["streams"]
        recognize_text, profile = h["recognize_text"], h["profile"]
        tm = {"detect_wait+tail": 0.0, "order": 0.0, "crop+enqueue": 0.0, "recognize_wait": 0.0, "assign": 0.0}
        N = len(arrays)
        H, W = arrays[0].shape[:2]
        pages, groups = [None] * N, []
        with _gc_paused():
            for (lo, hi), st, dst, dh, ro, dev_ev in zip(bounds, streams, h["det_streams"], h["det_handles"], h["ro_handles"], h["det_events"]):
                if ro is not None:
                    # device path: wait for the group's crop counts only (4 bytes per page), enqueue crops + recogniser
                    with torch.cuda.stream(dst):
                        t0 = time.perf_counter()
                        nc_h = ro[3].cpu().numpy()
                        tm["detect_wait+tail"] += time.perf_counter() - t0
                    if bool((nc_h >= 0).all()):
                        t0 = time.perf_counter()
                        grp = {"words": None, "spans": [], "handle": None, "ro": ro, "det": dh, "lohi": (lo, hi), "dst": dst, "det_event": dev_ev}
                        off = 0
                        for c in nc_h.tolist():
                            grp["spans"].append([off, c])
                            off += c
                        if off:
                            spans = [tuple(sp) for sp in grp["spans"] if sp[1] > 0]
                            use_graph = getattr(rec, "use_graphs", False)
                            prepared = None
                            if not use_graph:
                                with torch.cuda.stream(dst):  # the small blocking upload rides the high-priority stream
                                    prepared = rec.prepare_chunks(off, spans)
                            if st is not h["main"]:
                                st.wait_stream(h["main"])  # the page upload
                            st.wait_stream(dst)
                            with torch.cuda.stream(st):
                                ro[2].record_stream(st)
                                desc_dev = torch.cat([ro[2][pi, :c] for pi, c in enumerate(nc_h.tolist()) if c])
                                if use_graph:  # crop + encode + decode as one hipGraph replay
                                    grp["handle"] = rec.recognize_start_graph(pages_dev, desc_dev, spans, upload_stream=dst)
                                if grp["handle"] is None:
                                    if prepared is None:  # graph path declined (first call of a bucket, ...): plain launches
                                        with torch.cuda.stream(dst):
                                            prepared = rec.prepare_chunks(off, spans)
                                        st.wait_stream(dst)
                                    canv = ops.crop_resize_pad(pages_dev, None, rec.img_h, rec.img_w, desc_dev=desc_dev)
                                    grp["handle"] = rec.recognize_start(canv, spans=spans, prepared=prepared)
                        tm["crop+enqueue"] += time.perf_counter() - t0
                        groups.append(grp)
                        continue
                with torch.cuda.stream(dst):
                    t0 = time.perf_counter()
                    res = det.detect_finish(dh, arrays[lo:hi], profile=profile)
                    tm["detect_wait+tail"] += time.perf_counter() - t0
                if st is not h["main"]:
                    st.wait_stream(h["main"])  # the page upload
                with torch.cuda.stream(st):
                    grp = {"words": [], "spans": [], "handle": None}
                    if recognize_text:
                        t0 = time.perf_counter()
                        boxes, page_ids = [], []
                        for pi, r in enumerate(res):
                            page = self._page_of(r)
                            pages[lo + pi] = page
                            words, bxs = self._order_boxes(page)
                            grp["spans"].append([len(grp["words"]), len(words)])
                            grp["words"] += words
                            boxes += bxs
                            page_ids += [lo + pi] * len(bxs)
                        tm["order"] += time.perf_counter() - t0
                        t0 = time.perf_counter()
                        if boxes:
                            desc, keep = ops.crop_descriptors(boxes, page_ids, (H, W), rec.img_h, rec.img_w)
                            if not keep.all():  # empty clamped crops are skipped by the reference (_pipeline.py:135)
                                grp["words"] = [w for w, k in zip(grp["words"], keep) if k]
                                kept_pages = np.asarray(page_ids)[keep]
                                grp["spans"], n0 = [], 0
                                for pi in range(lo, hi):
                                    c = int((kept_pages == pi).sum())
                                    grp["spans"].append([n0, c])
                                    n0 += c
                            if len(desc):
                                spans = [tuple(s) for s in grp["spans"] if s[1] > 0]
                                up = dst if getattr(self, "upload_on_det_stream", True) else st
                                with torch.cuda.stream(up):  # the two small blocking uploads ride the high-priority stream
                                    desc_dev = torch.from_numpy(desc.astype("int32", copy=False)).to(det.device)
                                    prepared = rec.prepare_chunks(len(desc), spans)
                                if st is not up:
                                    st.wait_stream(up)
                                canv = ops.crop_resize_pad(pages_dev, desc, rec.img_h, rec.img_w, desc_dev=desc_dev)
                                grp["handle"] = rec.recognize_start(canv, spans=spans, prepared=prepared)
                        tm["crop+enqueue"] += time.perf_counter() - t0
                    else:
                        for pi, r in enumerate(res):
                            pages[lo + pi] = self._page_of(r)
                    groups.append(grp)
        h["pages"], h["groups"], h["tm"] = pages, groups, tm
        return h

    def collect_batch(self, h):
        """Stages 2-3 of `predict_batch` for a handle from `submit_batch` -> list of Pages (stage 2 = `advance_batch`,
        skipped when already done; stage 3 = per group: finish the recogniser (sync) and annotate the words)."""
        import torch

        self.advance_batch(h)
        rec = self.recognizer
        streams, main, profile = h["streams"], h["main"], h["profile"]
        pages, groups, tm = h["pages"], h["groups"], h["tm"]
        with _gc_paused():
            for grp, st in zip(groups, streams):
                if grp.get("ro") is not None:
                    # device-ordered group: Page / Word assembly happens here, off the path that feeds the device
                    lo, hi = grp["lohi"]
                    # read the boxes back on a copy stream of their own that only waits for the group's detector outputs: on the
                    # detector stream these copies would queue behind the detector work of the batch after next (it shares that
                    # stream), on the recogniser stream behind this batch's whole recogniser
                    if not hasattr(self, "_copy_stream"):
                        self._copy_stream = torch.cuda.Stream(priority=-1)
                    self._copy_stream.wait_event(grp["det_event"])
                    with torch.cuda.stream(self._copy_stream):
                        t0 = time.perf_counter()
                        res = self.detector.detect_finish(grp["det"], h["arrays"][lo:hi], profile=profile)
                        order_h, keep_h = grp["ro"][0].cpu().numpy(), grp["ro"][1].cpu().numpy()
                        tm["detect_wait+tail"] += time.perf_counter() - t0
                    t0 = time.perf_counter()
                    grp["words"] = []
                    for pi, r in enumerate(res):
                        page = self._page_of(r)
                        pages[lo + pi] = page
                        k0 = 0
                        for block in page.blocks:  # this package's EAST returns one block (infer.py:390)
                            nw = len(block.words)
                            if nw:
                                old = block.words
                                block.words = [old[k] for k in order_h[pi, k0:k0 + nw].tolist()]
                                grp["words"] += [block.words[pos] for pos in np.flatnonzero(keep_h[pi, k0:k0 + nw]).tolist()]
                            k0 += nw
                    tm["order"] += time.perf_counter() - t0
                if grp["handle"] is not None:
                    with torch.cuda.stream(st):
                        t0 = time.perf_counter()
                        ids, trun, conf = rec.recognize_finish(grp["handle"], spans=[tuple(s) for s in grp["spans"] if s[1] > 0])
                        tm["recognize_wait"] += time.perf_counter() - t0
                    t0 = time.perf_counter()
                    texts = rec.texts(ids, trun)
                    for word, text, c in zip(grp["words"], texts, conf.tolist()):
                        word.text = text
                        word.recognition_confidence = c
                    tm["assign"] += time.perf_counter() - t0
                if st is not main:
                    main.wait_stream(st)
            for dst in h["det_streams"]:
                if dst is not main:
                    main.wait_stream(dst)
        self.last_profile = tm
        if profile:
            print("Pipeline.predict_batch host stages (s):", {k: round(v, 4) for k, v in tm.items()})
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
