"""EAST detector plugin — drop-in for the reference class
(/root/reference/src/manuscript/detectors/_east/infer.py:27-402, inference half).

Same constructor knobs, same `predict` signature, return dict and exceptions.  The
network, quad decode and locality-aware NMS run on the MI355X through libmsocr.so
(manuscript_ocr_amd/csrc); there is no CPU execution path — constructing the detector
without a HIP device raises.

Extensions (keyword-only, all optional):
  precision   "fp32" (parity mode: f32 tensors and accumulation; the 1x1 / Winograd-domain GEMMs run on the bf16 matrix pipes with
              each operand split exactly into three bf16 terms) | "fp32-exact" (exact-f32 MFMA everywhere) |
              "bf16" (throughput mode, f32 accumulate)
  state_dict  in-memory weights in the reference key layout (offline: no download is possible)
  target_size may also be a (W, H) tuple: native non-square network input (multiples of 32)
  predict_batch(pages)  list/array of same-sized RGB pages -> list of result dicts, one launch
                        sequence for the whole batch (the reference is strictly one page per call)
"""
import time
from pathlib import Path
from typing import Any, Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from ... import ops
from .._types import Block, Page, Word
from . import post
from .net import EastNet
from .utils import read_image, sort_boxes_reading_order_with_resolutions, visualize_page

_DEFAULT_WEIGHT_LOCATIONS = (
    Path("weights") / "east_quad_23_05.pth",
    Path.home() / ".manuscript" / "east" / "east_quad_23_05.pth",
)


class EAST:
    def __init__(
        self,
        weights_path: Optional[Union[str, Path]] = None,
        device: Optional[str] = None,
        target_size: Union[int, Tuple[int, int]] = 1280,
        expand_ratio_w: float = 0.9,
        expand_ratio_h: float = 0.9,
        score_thresh: float = 0.6,
        iou_threshold: float = 0.2,
        score_geo_scale: float = 0.25,
        quantization: int = 2,
        axis_aligned_output: bool = True,
        remove_area_anomalies: bool = True,
        anomaly_sigma_threshold: float = 5.0,
        anomaly_min_box_count: int = 30,
        *,
        precision: str = "fp32",
        state_dict: Optional[Dict[str, torch.Tensor]] = None,
        max_candidates: Optional[int] = None,
        use_graphs: bool = False,
    ):
        self.device = device or ("cuda" if torch.cuda.is_available() else "cpu")
        if not str(self.device).startswith("cuda") or not torch.cuda.is_available():
            raise RuntimeError(
                f"manuscript_ocr_amd.EAST runs only on a HIP device (MI355X); device={self.device!r}, "
                f"torch.cuda.is_available()={torch.cuda.is_available()}. There is no CPU fallback."
            )
        if state_dict is None:
            if weights_path is None:
                # the reference downloads east_quad_23_05.pth with gdown (infer.py:96-107); no network here
                found = next((p for p in _DEFAULT_WEIGHT_LOCATIONS if p.exists()), None)
                if found is None:
                    raise FileNotFoundError(
                        "EAST weights not found: pass weights_path=... (or state_dict=...), or place east_quad_23_05.pth under "
                        "./weights/ or ~/.manuscript/east/ (automatic download is unavailable offline)."
                    )
                weights_path = found
            if not Path(weights_path).exists():
                raise FileNotFoundError(f"EAST weights not found: {weights_path}")
            state_dict = torch.load(str(weights_path), map_location="cpu", weights_only=True)
        self.precision = precision
        if precision not in ("fp32", "fp32-exact", "bf16"):
            raise ValueError(f"precision must be 'fp32', 'fp32-exact' or 'bf16', got {precision!r}")
        dtype = torch.bfloat16 if precision == "bf16" else torch.float32
        # "fp32": f32 tensors; 1x1 convolutions and Winograd-domain GEMMs on the bf16 matrix pipes with exactly split operands
        # (csrc/conv_split.hip); "fp32-exact": exact-f32 MFMA in every layer
        self.model = EastNet(state_dict, dtype=dtype, device=self.device, split=(False if precision == "fp32-exact" else None))

        self.target_size = target_size
        self.score_geo_scale = score_geo_scale
        self.expand_ratio_w = expand_ratio_w
        self.expand_ratio_h = expand_ratio_h
        self.score_thresh = score_thresh
        self.iou_threshold = iou_threshold
        self.quantization = quantization
        self.axis_aligned_output = axis_aligned_output
        self.remove_area_anomalies = remove_area_anomalies
        self.anomaly_sigma_threshold = anomaly_sigma_threshold
        self.anomaly_min_box_count = anomaly_min_box_count
        # capacity of the candidate / box buffers per page.  None = the exact upper bound of decode_quads_from_maps for the
        # network input: one candidate per q x q cell of the 1/4-resolution map (utils.py:349-356), so decode cannot overflow
        self.max_candidates = max_candidates
        self.use_graphs = bool(use_graphs)  # hipGraph replay of the static detect sequence in detect_start (BASELINE configs[3])
        self._graphs: Dict[Any, Dict[str, Any]] = {}
        if abs(1.0 / score_geo_scale - 4.0) > 1e-9:
            raise ValueError("the network emits maps at 1/4 resolution (east.py:126-127): score_geo_scale must be 0.25")

    # ------------------------------------------------------------------------------------- helpers
    def _target_wh(self):
        t = self.target_size
        return (int(t), int(t)) if np.isscalar(t) else (int(t[0]), int(t[1]))

    def _max_candidates(self) -> int:
        if self.max_candidates is not None:
            return int(self.max_candidates)
        tw, th = self._target_wh()
        q = max(1, int(self.quantization))
        return max(1, (th // 4 // q) * (tw // 4 // q))

    def detect_device(self, pages_dev, maps_override=None):
        """pages_dev [N,h,w,3] u8 on the device (any size) -> device tensors
        (score, geo, boxes [N,max_cand,9], nbox [N]).  Resize, network, decode and LANMS, all HIP.
        A LIST of [h_i,w_i,3] tensors is a ragged batch: the reference resizes every page to the network input first
        (infer.py:304), so pages of any mix of sizes are resized on the device one by one, stacked and sent through the network
        together; only the box tail, which scales back to each page's own size, runs per size group."""
        tw, th = self._target_wh()
        if isinstance(pages_dev, (list, tuple)):
            sizes = [(int(t.shape[0]), int(t.shape[1])) for t in pages_dev]
            pages_dev = torch.cat([t[None] if sizes[i] == (th, tw) else ops.resize_linear_u8(t[None].contiguous(), th, tw)
                                   for i, t in enumerate(pages_dev)])
        else:
            sizes = [(int(pages_dev.shape[1]), int(pages_dev.shape[2]))] * int(pages_dev.shape[0])
            if sizes and sizes[0] != (th, tw):
                pages_dev = ops.resize_linear_u8(pages_dev, th, tw)
        score, geo = self.model.forward(pages_dev)
        if maps_override is not None:  # benchmark / parity harness: injected maps
            score.copy_(maps_override[0], non_blocking=True)
            geo.copy_(maps_override[1], non_blocking=True)
        cand, counts = ops.east_decode(score, geo, self.score_thresh, 1.0 / self.score_geo_scale, self.quantization,
                                       self._max_candidates())
        boxes, nbox = ops.east_lanms(cand, counts, self.iou_threshold)
        fboxes = fn = None
        if getattr(self, "device_tail", True):
            # infer.py:340-356 on the device: expand, scale back to the page, contained boxes, area anomalies, axis-aligned
            def tail(b, n, hw):
                return ops.east_box_tail(b, n, self.expand_ratio_w, self.expand_ratio_h, hw[1] / tw, hw[0] / th,
                                         self.axis_aligned_output, self.remove_area_anomalies, self.anomaly_sigma_threshold,
                                         self.anomaly_min_box_count)
            if len(set(sizes)) <= 1:
                fboxes, fn = tail(boxes, nbox, sizes[0])
            else:  # one launch per page size, results back in page order
                fboxes, fn = torch.empty_like(boxes), torch.empty_like(nbox)
                for hw in sorted(set(sizes)):
                    idx = torch.tensor([i for i, s_ in enumerate(sizes) if s_ == hw], device=boxes.device)
                    fb, fc = tail(boxes.index_select(0, idx).contiguous(), nbox.index_select(0, idx).contiguous(), hw)
                    fboxes.index_copy_(0, idx, fb)
                    fn.index_copy_(0, idx, fc)
        return score, geo, boxes, nbox, counts, fboxes, fn

    def _host_tail(self, quads: np.ndarray, orig_hw) -> np.ndarray:
        """infer.py:340-356 on the (M,9) f32 NMS output."""
        q = post.expand_boxes(quads, self.expand_ratio_w, self.expand_ratio_h)
        q = post.scale_boxes(q, orig_hw, self._target_wh())
        q = post.remove_contained(q)
        q = post.remove_area_anomalies(q, self.remove_area_anomalies, self.anomaly_sigma_threshold, self.anomaly_min_box_count)
        return post.to_axis_aligned(q) if self.axis_aligned_output else q

    @staticmethod
    def _words(quads: np.ndarray) -> List[Word]:
        """infer.py:358-363.  The scores are checked against the DTO's [0, 1] range for the whole page at once; valid pages (every
        page a sigmoid produced) build their Words without running the per-object validator again — same field values, same types
        (tuples of Python floats), ~10x less host time per page."""
        q = np.asarray(quads, dtype=np.float32).reshape(-1, 9)
        if len(q) == 0:
            return []
        if not bool(np.all((q[:, 8] >= 0.0) & (q[:, 8] <= 1.0))):  # out of range / NaN: raise exactly as the validated constructor does
            return [Word(polygon=r[:8].reshape(4, 2).tolist(), detection_confidence=float(r[8])) for r in q]
        return [Word.model_construct(polygon=[(r[0], r[1]), (r[2], r[3]), (r[4], r[5]), (r[6], r[7])], detection_confidence=r[8])
                for r in q.tolist()]

    @staticmethod
    def _sort_words(words: List[Word]) -> List[Word]:
        def aabb(w):
            poly = np.array(w.polygon, dtype=np.int32)
            (x0, y0), (x1, y1) = np.min(poly, axis=0), np.max(poly, axis=0)
            return (x0, y0, x1, y1)

        boxes = [aabb(w) for w in words]
        out = []
        for bx in sort_boxes_reading_order_with_resolutions(boxes):
            for w, wb in zip(words, boxes):
                if wb == bx:
                    out.append(w)
                    break
        return out

    # ------------------------------------------------------------------------------------- API
    def detect_start(self, pages_dev: torch.Tensor, maps_override=None):
        """Enqueue resize + network + decode + LANMS for [N,h,w,3] u8 device pages on the CURRENT stream; no sync.

        With `use_graphs` the whole sequence (~110 launches, static shapes) is captured once per input shape into a
        hipGraph and replayed: the page bytes are copied into the instance's static input buffer, the outputs live in
        the graph's private pool until `detect_finish` has read them.  The first call of a shape runs eagerly (lazy
        one-time kernel attributes must not fall into a capture); an instance still in flight is never replayed — a
        second one is captured (two batches in flight = two instances per group)."""
        if not self.use_graphs or ops.PROFILE is not None or isinstance(pages_dev, (list, tuple)):
            return self.detect_device(pages_dev, maps_override) + (None,)
        key = (tuple(pages_dev.shape), None if maps_override is None else (maps_override[0].data_ptr(), maps_override[1].data_ptr()))
        pool = self._graphs.setdefault(key, {"warm": False, "inst": []})
        if not pool["warm"]:
            pool["warm"] = True
            return self.detect_device(pages_dev, maps_override) + (None,)
        inst = next((i for i in pool["inst"] if not i["busy"]), None)
        if inst is None:
            if len(pool["inst"]) >= 4:
                return self.detect_device(pages_dev, maps_override) + (None,)
            for _ in range(2 if not pool["inst"] else 1):  # the first capture of a shape makes TWO instances: consecutive
                inp = torch.empty_like(pages_dev)          # batches overlap (submit i+1 before collect i), so both are needed
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    out = self.detect_device(inp, maps_override)
                pool["inst"].append({"graph": graph, "inp": inp, "out": out, "busy": False})
            inst = pool["inst"][-1]
        inst["busy"] = True
        inst["inp"].copy_(pages_dev, non_blocking=True)
        inst["graph"].replay()
        return inst["out"] + (inst,)

    def detect_finish(self, handle, imgs, vis=False, profile=False, return_maps=False, sort_reading_order=False):
        """Wait for `handle` (from detect_start), then the host tail (infer.py:340-390) -> list of result dicts."""
        inst = handle[7]
        try:
            t0 = time.time()
            score, geo, boxes, nbox, counts, fboxes, fn, _ = handle
            nbox_h = nbox.cpu().numpy()
            counts_h = counts.cpu().numpy()
            if np.any(counts_h < 0):
                raise RuntimeError(f"more than max_candidates={self._max_candidates()} cells above threshold; raise max_candidates "
                                   "(None = the exact bound for the network input)")
            fn_h = fn.cpu().numpy() if fn is not None else None
            on_device = fn_h is not None and bool(np.all(fn_h >= 0))  # a page above the device tail capacity (16384 boxes) leaves the tail to the host
            if on_device:
                final_h = fboxes[:, : max(int(fn_h.max()), 1)].cpu().numpy()
            else:
                boxes_h = boxes[:, : max(int(nbox_h.max()), 1)].cpu().numpy()
            if profile:
                print(f"  Model inference + decode + NMS (device wait): {time.time() - t0:.3f}s")
                print(f"    Boxes after NMS: {[int(v) for v in nbox_h]}")
            t_dev = time.time() - t0
            results = []
            for n, img in enumerate(imgs):
                quads = final_h[n, : fn_h[n]] if on_device else self._host_tail(boxes_h[n, : nbox_h[n]], img.shape[:2])
                words = self._words(quads)
                if sort_reading_order and words:
                    words = self._sort_words(words)
                page = Page(blocks=[Block(words=words)])
                results.append({
                    "page": page,
                    "vis_image": visualize_page(img, page, show_order=False) if vis else None,
                    "score_map": score[n].cpu().numpy() if return_maps else None,
                    "geo_map": geo[n].permute(2, 0, 1).contiguous().cpu().numpy() if return_maps else None,
                })
            self.last_profile = {"device_wait": t_dev, "host_tail": time.time() - t0 - t_dev}
            return results
        finally:
            if inst is not None:
                inst["busy"] = False  # every output of the graph instance has been copied out (or the call failed)

    def predict_batch(self, images: Sequence[np.ndarray], vis=False, profile=False, return_maps=False,
                      sort_reading_order=False, _maps_override=None, _pages_dev=None) -> List[Dict[str, Any]]:
        imgs, decoded = [], []
        for im in images:
            t = None
            if _pages_dev is None and not vis and getattr(self, "device_ingest", True):
                from ... import ingest
                t = ingest.read_image_device(im, self.device)  # JPEG file: decoded on the device, the host keeps only the shape
            imgs.append(np.broadcast_to(np.uint8(0), tuple(t.shape)) if t is not None else read_image(im))
            decoded.append(t)
        if _pages_dev is not None:
            pages = _pages_dev
        else:
            pages = [t if t is not None else torch.from_numpy(np.ascontiguousarray(a)).to(self.device) for t, a in zip(decoded, imgs)]
            if len({im.shape for im in imgs}) == 1:
                pages = torch.stack(pages)  # equally sized pages: one tensor (hipGraph path, one resize launch); else a ragged list
        return self.detect_finish(self.detect_start(pages, _maps_override), imgs, vis, profile, return_maps, sort_reading_order)

    def predict(self, img_or_path: Union[str, Path, np.ndarray], vis: bool = False, profile: bool = False,
                return_maps: bool = False, sort_reading_order: bool = False) -> Dict[str, Any]:
        """Same contract as the reference EAST.predict (infer.py:235-402): keys
        {"page","vis_image","score_map","geo_map"}; FileNotFoundError / TypeError on bad input."""
        if not isinstance(img_or_path, (str, Path, np.ndarray)):
            raise TypeError(f"Unsupported type for image input: {type(img_or_path)}")  # read_image's contract (utils.py:494-495)
        return self.predict_batch([img_or_path], vis=vis, profile=profile, return_maps=return_maps,
                                  sort_reading_order=sort_reading_order)[0]
