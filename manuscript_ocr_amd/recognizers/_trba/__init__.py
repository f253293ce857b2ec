"""TRBA recogniser plugin — drop-in for the reference class
(/root/reference/src/manuscript/recognizers/_trba/__init__.py:23-434, inference half).

Same constructor (model_path / charset_path / config_path / device / weights_path alias), same
`predict` signature, outputs and exceptions.  SE-ResNet31, the BiLSTM encoder and the attention
decoder (greedy and beam) run on the MI355X through libmsocr.so; there is no CPU execution path.

Extensions (keyword-only): precision="fp32"|"bf16" for the CNN (recurrent/attention stages are
always exact f32), state_dict=... / config=... for in-memory weights (nothing can be downloaded
offline), device_batch=2048 rows per launch sequence.  Confidences reproduce the reference's
dependence on `batch_size` chunks (its decode loop stops per chunk, model.py:215,254).
"""
import json
import os
from pathlib import Path
from typing import Any, Dict, List, Optional, Tuple, Union

import numpy as np
import torch
from PIL import Image

from .net import TrbaNet
from .transforms import decode_tokens, load_charset, resize_and_pad


class _GraphLease:
    """Holds one captured recogniser-graph instance for a handle; released by recognize_finish (also when it raises) or when the
    handle is dropped without ever being finished.  An event recorded right behind the replay says when the instance's static
    buffers may be rewritten: a dropped handle waits for THAT event only — never a device-wide synchronize from a finalizer (the
    garbage collector may run it while another stream capture is in progress, and a synchronize would invalidate that capture)."""

    def __init__(self, inst):
        self.inst = inst
        self.done = torch.cuda.Event()
        self.done.record()  # on the replay's stream, behind the replay

    def release(self):
        if self.inst is not None:
            self.inst["busy"] = False
            self.inst = None

    def __del__(self):
        if self.inst is None:
            return
        try:
            if torch.cuda.is_current_stream_capturing():
                return  # leave the instance marked busy: it is simply never reused (a new one is captured on demand)
            if not self.done.query():
                self.done.synchronize()
        except Exception:
            return  # the wait failed: do not hand the instance out again
        self.inst["busy"] = False
        self.inst = None


class TRBA:
    _DEFAULT_PRESET_NAME = "exp_1_baseline"
    _DEFAULT_STORAGE_ROOT = Path.home() / ".manuscript" / "trba"
    _DEFAULT_WEIGHTS_FILENAME = "weights.pth"
    _DEFAULT_CONFIG_FILENAME = "config.json"

    def __init__(self, model_path: Optional[str] = None, charset_path: Optional[str] = None, config_path: Optional[str] = None,
                 device: str = "auto", **kwargs: Any):
        weights_path = kwargs.pop("weights_path", None)
        precision = kwargs.pop("precision", "fp32")
        state_dict = kwargs.pop("state_dict", None)
        config = kwargs.pop("config", None)
        self.device_batch = int(kwargs.pop("device_batch", 2048))
        self.use_graphs = bool(kwargs.pop("use_graphs", False))  # hipGraph replay of crop + encode + beam decode (recognize_start_graph)
        self.graph_cache_buckets = int(kwargs.pop("graph_cache_buckets", 8))  # captured (stream, pages, row bucket) keys kept
        self._graphs: Dict[Any, Dict[str, Any]] = {}
        if kwargs:
            raise TypeError(f"Unexpected keyword argument(s): {', '.join(kwargs.keys())}")
        if weights_path is not None and model_path is not None:
            if os.path.abspath(os.fspath(weights_path)) != os.path.abspath(os.fspath(model_path)):
                raise ValueError("Provide either model_path or weights_path, but not both with different values.")

        if state_dict is None:
            self.model_path, resolved_config = self._resolve_paths(weights_path if weights_path is not None else model_path, config_path)
        else:
            self.model_path, resolved_config = None, (os.fspath(config_path) if config_path is not None else None)
            if resolved_config is not None and not os.path.exists(resolved_config):
                raise FileNotFoundError(f"Config file not found: {resolved_config}")

        if charset_path is None:
            charset_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs", "charset.txt")
        self.charset_path = os.fspath(charset_path)
        self.config_path = resolved_config
        if not os.path.exists(self.charset_path):
            raise FileNotFoundError(f"Charset file not found: {self.charset_path}")
        if config is None:
            if self.config_path is not None:
                with open(self.config_path, "r", encoding="utf-8") as f:
                    config = json.load(f)
            else:
                config = {}
        self.max_length = config.get("max_len", 25)
        self.hidden_size = config.get("hidden_size", 256)
        self.img_h = config.get("img_h", 64)
        self.img_w = config.get("img_w", 256)

        if device == "auto":
            self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        else:
            self.device = torch.device(device)
        self.itos, self.stoi = load_charset(self.charset_path)
        self.pad_id = self.stoi["<PAD>"]
        self.sos_id = self.stoi["<SOS>"]
        self.eos_id = self.stoi["<EOS>"]
        self.blank_id = self.stoi.get("<BLANK>", None)

        if self.device.type != "cuda" or not torch.cuda.is_available():
            raise RuntimeError(
                f"manuscript_ocr_amd.TRBA runs only on a HIP device (MI355X); device={self.device}, "
                f"torch.cuda.is_available()={torch.cuda.is_available()}. There is no CPU fallback."
            )
        if state_dict is None:
            obj = torch.load(self.model_path, map_location="cpu", weights_only=True)
            state_dict = obj["model_state"] if isinstance(obj, dict) and "model_state" in obj else obj  # training/utils.py:54-59
        self.precision = precision
        if precision not in ("fp32", "fp32-exact", "bf16"):
            raise ValueError(f"precision must be 'fp32', 'fp32-exact' or 'bf16', got {precision!r}")
        self.model = TrbaNet(state_dict, len(self.itos), self.hidden_size, torch.bfloat16 if precision == "bf16" else torch.float32,
                             self.device, split=(False if precision == "fp32-exact" else None))

    # ------------------------------------------------------------------------------------- paths
    def _resolve_paths(self, model_path, config_path) -> Tuple[str, Optional[str]]:
        if model_path is None:
            target = self._DEFAULT_STORAGE_ROOT.expanduser() / self._DEFAULT_PRESET_NAME
            for cand in (Path("weights") / "trba" / self._DEFAULT_WEIGHTS_FILENAME, target / self._DEFAULT_WEIGHTS_FILENAME):
                if cand.exists():
                    model_path = os.fspath(cand)
                    break
            else:  # the reference downloads trba_exp_1_64.pth/.json with gdown (__init__.py:207-243); no network here
                raise FileNotFoundError(
                    "TRBA weights not found: pass model_path=... (or state_dict=...), or place weights.pth (+config.json) under "
                    "./weights/trba/ or ~/.manuscript/trba/exp_1_baseline/ (automatic download is unavailable offline)."
                )
        resolved = os.fspath(model_path)
        if not os.path.exists(resolved):
            raise FileNotFoundError(f"Model checkpoint not found: {resolved}")
        if config_path is not None:
            cfg = os.fspath(config_path)
            if not os.path.exists(cfg):
                raise FileNotFoundError(f"Config file not found: {cfg}")
        else:
            wf = Path(resolved)
            cfg = next((os.fspath(c) for c in (wf.with_suffix(".json"), wf.parent / "config.json") if c.exists()), None)
        return resolved, cfg

    # ------------------------------------------------------------------------------------- preprocessing
    def _load_rgb(self, image) -> np.ndarray:
        if isinstance(image, str):
            if not os.path.exists(image):
                raise FileNotFoundError(f"Image file not found: {image}")
            try:
                with Image.open(image) as im:
                    return np.array(im.convert("RGB"))
            except Exception:
                raise ValueError(f"Cannot read image: {image}")
        if isinstance(image, Image.Image):
            return np.array(image.convert("RGB"))
        if isinstance(image, np.ndarray):
            return image
        raise ValueError(f"Unsupported image type: {type(image)}")

    def _canvases(self, images) -> np.ndarray:
        """ResizeAndPadA on the host (u8), one [img_h, img_w, 3] canvas per crop."""
        return np.stack([resize_and_pad(self._load_rgb(im), self.img_h, self.img_w) for im in images])

    # ------------------------------------------------------------------------------------- device path
    def _device_batches(self, N, spans, batch_size):
        """Reference chunks (slices of `batch_size` rows inside each span = one model call of the reference) packed into
        launches of at most `device_batch` rows that never split a chunk.  -> (bounds [(lo, hi)], per-launch int32 meta
        = chunk ids of the rows followed by the chunk sizes)."""
        chunk_of = np.full(N, -1, dtype=np.int64)
        chunks = []
        for s0, cnt in (spans if spans is not None else [(0, N)]):
            for c0 in range(s0, s0 + cnt, batch_size):
                c1 = min(c0 + batch_size, s0 + cnt)
                chunk_of[c0:c1] = len(chunks)
                chunks.append((c0, c1))
        contiguous = len(chunks) > 0 and chunks[0][0] == 0 and chunks[-1][1] == N and all(a[1] == b[0] for a, b in zip(chunks, chunks[1:]))
        if not contiguous:  # rows outside every span, or spans out of order: no chunk information (all steps run)
            return [(s, min(s + self.device_batch, N)) for s in range(0, N, self.device_batch)], None
        bounds, metas, lo, first = [], [], 0, 0
        for k, (c0, c1) in enumerate(chunks + [(N, N)]):
            if k == len(chunks) or (c1 - lo > self.device_batch and c0 > lo):
                ids = chunk_of[lo:c0] - first
                sizes = np.array([b - a for a, b in chunks[first:k]], dtype=np.int64)
                bounds.append((lo, c0))
                metas.append(np.concatenate([ids, sizes]).astype(np.int32))
                lo, first = c0, k
        return bounds, metas

    def prepare_chunks(self, N, spans, batch_size=32):
        """Chunk metadata of `recognize_start` for N rows, uploaded on the CURRENT stream (one small blocking copy).  Callers
        that pipeline several streams do this on a high-priority stream and pass the result as `prepared=` so the copy does
        not queue behind the recogniser work of other groups."""
        bounds, metas = self._device_batches(N, spans, batch_size)
        meta_dev = torch.from_numpy(np.concatenate(metas)).to(self.device) if metas is not None else None
        return bounds, metas, meta_dev

    def recognize_start(self, canvases_dev: torch.Tensor, mode="beam", beam_size=8, temperature=1.7, alpha=0.9, spans=None,
                        batch_size=32, prepared=None):
        """Phase 1 — encode + decode of [N,img_h,img_w,3] u8 device canvases, enqueued on the CURRENT stream without any
        synchronisation.  Returns a handle for `recognize_finish`.  `spans`/`batch_size` (the same values `recognize_finish`
        will get) tell the beam kernel which rows share a reference chunk, so it can stop a chunk where the reference's
        loop does; without them every row runs all max_length steps (same results)."""
        if mode not in ("greedy", "beam"):
            raise ValueError(f"Unknown mode: {mode}")
        N = canvases_dev.shape[0]
        if mode == "beam" and prepared is not None:
            bounds, metas, meta_dev = prepared
            if meta_dev is not None:
                meta_dev.record_stream(torch.cuda.current_stream())
        elif mode == "beam" and spans is not None:
            bounds, metas, meta_dev = self.prepare_chunks(N, spans, batch_size)  # before any encoder work is queued here
        else:
            bounds, metas, meta_dev = [(s, min(s + self.device_batch, N)) for s in range(0, N, self.device_batch)], None, None
        parts, off = [], 0
        for k, (lo, hi) in enumerate(bounds):
            cv = canvases_dev[lo:hi]
            batch_H, proj_H = self.model.encode(cv)
            if mode == "greedy":
                parts.append(self.model.greedy(batch_H, proj_H, self.max_length, self.sos_id, self.eos_id, self.blank_id))
            else:
                chunks = None
                if meta_dev is not None:
                    nrows, nch = hi - lo, len(metas[k]) - (hi - lo)
                    chunks = (meta_dev[off:off + nrows], meta_dev[off + nrows:off + nrows + nch],
                              torch.zeros((2 * nch,), dtype=torch.int32, device=self.device))
                    off += nrows + nch
                parts.append(self.model.beam(batch_H, proj_H, self.max_length, beam_size, alpha, temperature, self.sos_id, self.eos_id,
                                             self.blank_id, chunks))
        return {"parts": parts, "N": N, "mode": mode, "beam": beam_size, "bounds": bounds}

    def recognize_start_graph(self, pages_dev: torch.Tensor, desc_dev: torch.Tensor, spans, batch_size=32, beam_size=8,
                              temperature=1.7, alpha=0.9, upload_stream=None):
        """`recognize_start` for crops that are still descriptors on the device (ops.reading_order_crops), as ONE hipGraph replay:
        crop + ResizeAndPadA, SE-ResNet31, BiLSTMs and the beam decode of up to `device_batch` rows are captured once per
        (page tensor, row bucket) and replayed (BASELINE configs[3]: "hipGraph-captured").  The row count is rounded up to a
        multiple of 32; the padding rows repeat crop 0 and form a reference chunk of their own, so they can neither change a real
        row's result nor delay a real chunk's early exit.  Falls back to the eager path (returns None) for empty or oversized
        batches, while kernels are being profiled, and for the first call of a bucket (lazy one-time kernel attributes must not
        fall into a capture).  Results are bit-identical to the eager path (tests/test_gpu_pipeline.py)."""
        from ... import ops
        M = int(desc_dev.shape[0])
        Mcap = (M + 31) // 32 * 32
        if M == 0 or Mcap > self.device_batch or ops.PROFILE is not None:
            return None
        nch_cap = Mcap // batch_size + len(spans) + 2  # chunks of the real rows (<= rows/batch_size + one per page) + the padding chunk
        # one pool per launch stream: the groups of a batch are in flight together, each replays its own instances
        key = (torch.cuda.current_stream().cuda_stream, pages_dev.data_ptr(), tuple(pages_dev.shape), Mcap, nch_cap, batch_size, beam_size,
               float(temperature), float(alpha))
        pool = self._graphs.pop(key, None) or {"warm": False, "inst": []}
        self._graphs[key] = pool  # most recently used last
        # bounded cache: the key holds the page tensor's address, so callers whose pages move would otherwise capture
        # without end, each instance pinning a private pool of up to device_batch crops of activations.  Least recently used
        # buckets whose instances are all idle are dropped.
        while len(self._graphs) > self.graph_cache_buckets:
            victim = next((k for k, v in self._graphs.items() if k != key and not any(i["busy"] for i in v["inst"])), None)
            if victim is None:
                break
            del self._graphs[victim]
        if not pool["warm"]:
            pool["warm"] = True
            return None
        # chunk metadata of the real rows (as prepare_chunks) + one chunk for the padding rows, into the instance's static buffer
        bounds, metas = self._device_batches(M, spans, batch_size)
        if metas is None or len(bounds) != 1:
            return None
        ids, sizes = metas[0][:M], metas[0][M:]
        if len(sizes) + 1 > nch_cap:
            return None
        meta = np.zeros(Mcap + nch_cap, dtype=np.int32)
        meta[:M] = ids
        meta[M:Mcap] = len(sizes)
        meta[Mcap:Mcap + len(sizes)] = sizes
        meta[Mcap + len(sizes)] = max(Mcap - M, 1)
        inst = next((i for i in pool["inst"] if not i["busy"]), None)
        if inst is None:
            if len(pool["inst"]) >= 4:
                return None
            for _ in range(2 if not pool["inst"] else 1):  # consecutive batches overlap: two instances per bucket
                dbuf = torch.zeros((Mcap, 8), dtype=torch.int32, device=self.device)
                mbuf = torch.zeros((Mcap + nch_cap,), dtype=torch.int32, device=self.device)
                dbuf[:M].copy_(desc_dev)
                dbuf[M:] = desc_dev[0]
                mbuf.copy_(torch.from_numpy(meta).to(self.device))
                torch.cuda.current_stream().synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    canv = ops.crop_resize_pad(pages_dev, None, self.img_h, self.img_w, desc_dev=dbuf)
                    state = torch.zeros((2 * nch_cap,), dtype=torch.int32, device=self.device)
                    batch_H, proj_H = self.model.encode(canv)
                    part = self.model.beam(batch_H, proj_H, self.max_length, beam_size, alpha, temperature, self.sos_id, self.eos_id,
                                           self.blank_id, (mbuf[:Mcap], mbuf[Mcap:], state))
                pool["inst"].append({"graph": graph, "desc": dbuf, "meta": mbuf, "part": part, "busy": False})
            inst = pool["inst"][-1]
        inst["busy"] = True
        inst["desc"][:M].copy_(desc_dev, non_blocking=True)
        if Mcap > M:
            inst["desc"][M:] = desc_dev[0]
        # the small blocking upload of the chunk metadata must not queue behind the recogniser work already on this stream
        cur = torch.cuda.current_stream()
        up = upload_stream if upload_stream is not None else cur
        with torch.cuda.stream(up):
            meta_dev = torch.from_numpy(meta).to(self.device)
        if up is not cur:
            cur.wait_stream(up)
            meta_dev.record_stream(cur)
        inst["meta"].copy_(meta_dev, non_blocking=True)
        inst["graph"].replay()
        return {"parts": [inst["part"]], "N": Mcap, "M_real": M, "mode": "beam", "beam": beam_size, "bounds": [(0, Mcap)],
                "graph_inst": _GraphLease(inst)}

    def recognize_finish(self, handle, batch_size=32, spans=None, return_logits=False):
        """Phases 2-3 — derive the reference's per-chunk run lengths, back-track (beam) and reduce confidences.

        `spans` = [(start, count), ...] groups of rows that the reference would have passed to ONE predict() call
        (one page each); inside a span rows are chunked by `batch_size` and every chunk stops at its own step
        (greedy: first step where every row emits EOS; beam: once every beam of every row is finished) — that run
        length enters the confidences.  Only ids and one float per row cross PCIe (msocr_seq_confidence)."""
        from ... import _native as nat
        from ... import ops
        try:
            return self._recognize_finish(handle, batch_size, spans, return_logits)
        finally:
            if handle.get("graph_inst") is not None:  # also on an exception: the instance must not stay leased for ever
                handle["graph_inst"].release()

    def _recognize_finish(self, handle, batch_size, spans, return_logits):
        from ... import _native as nat
        from ... import ops
        parts, N, mode, beam_size = handle["parts"], handle["N"], handle["mode"], handle["beam"]
        M_real = handle.get("M_real", N)  # a graph replay carries padding rows behind the real ones
        spans = spans if spans is not None else [(0, M_real)]
        steps = self.max_length + 1 if mode == "greedy" else self.max_length
        trun = np.ones(N, dtype=np.int32)
        if mode == "greedy":
            ids_h = np.concatenate([p[1].cpu().numpy() for p in parts])
            for s0, cnt in spans:
                for c0 in range(s0, s0 + cnt, batch_size):
                    c1 = min(c0 + batch_size, s0 + cnt)
                    hit = np.flatnonzero(np.all(ids_h[c0:c1] == self.eos_id, axis=0))
                    trun[c0:c1] = (hit[0] + 1) if len(hit) else steps
        else:
            fin_h = np.concatenate([p[1].cpu().numpy() for p in parts])
            for s0, cnt in spans:
                for c0 in range(s0, s0 + cnt, batch_size):
                    c1 = min(c0 + batch_size, s0 + cnt)
                    trun[c0:c1] = fin_h[c0:c1].max()
        self.last_run_length_sum = getattr(self, "last_run_length_sum", 0) + int(trun[:M_real].sum())
        self.last_rows = getattr(self, "last_rows", 0) + M_real
        trun_dev = torch.from_numpy(trun).to(self.device)
        ids_out, conf_out, logit_out = [], [], []
        for k, (s, hi) in enumerate(handle["bounds"]):
            B = hi - s
            tr = trun_dev[s:s + B]
            if mode == "greedy":
                lg, ids = parts[k]
            else:
                lg, ids = self.model.beam_finalize(parts[k][0], B, steps, beam_size, tr)
            conf = torch.empty((B,), dtype=torch.float32, device=self.device)
            nat.check(nat.lib().msocr_seq_confidence(lg.data_ptr(), ids.data_ptr(), tr.data_ptr(), B, self.model.V, steps, conf.data_ptr(),
                                                     ops._stream()), "seq_confidence")
            ids_out.append(ids), conf_out.append(conf)
            if return_logits:
                logit_out.append(lg.cpu().numpy())
            parts[k] = None
        ids_h = torch.cat(ids_out).cpu().numpy()[:M_real]
        conf_h = torch.cat(conf_out).cpu().numpy()[:M_real]
        if return_logits:
            return ids_h, trun[:M_real], conf_h, np.concatenate(logit_out)[:M_real]
        return ids_h, trun[:M_real], conf_h

    def recognize_canvases(self, canvases_dev: torch.Tensor, batch_size=32, mode="beam", beam_size=8, temperature=1.7, alpha=0.9,
                           spans=None, return_logits=False):
        """canvases [N,img_h,img_w,3] u8 on device -> (ids [N,steps] i32, t_run [N] i32, conf [N] f32[, logits]) on host."""
        return self.recognize_finish(self.recognize_start(canvases_dev, mode, beam_size, temperature, alpha,
                                                          spans if spans is not None else [(0, canvases_dev.shape[0])], batch_size),
                                     batch_size, spans, return_logits)

    def texts(self, ids, trun) -> List[str]:
        """decode_tokens (transforms.py:196-206) over the first t_run ids of every row, vectorised: the text ends at the
        first EOS; PAD (and BLANK) ids are skipped."""
        ids = np.asarray(ids)
        T = ids.shape[1] if ids.ndim == 2 else 0
        if not len(ids) or T == 0:
            return ["" for _ in range(len(ids))]
        pos = np.arange(T)[None, :]
        stop = (ids == self.eos_id) | (pos >= np.asarray(trun)[:, None])
        end = np.where(stop.any(axis=1), stop.argmax(axis=1), T)
        itos, skip = self.itos, {self.pad_id, self.blank_id}
        rows = ids.tolist()
        return ["".join([itos[t] for t in row[:e] if t not in skip]) for row, e in zip(rows, end.tolist())]

    def _results(self, ids, trun, conf) -> List[Dict[str, Any]]:
        """__init__.py:415-432: decode_tokens over the t_run generated ids; confidence computed on the device."""
        return [{"text": decode_tokens(ids[j, : int(trun[j])], self.itos, self.pad_id, self.eos_id, self.blank_id),
                 "confidence": float(conf[j])} for j in range(len(ids))]

    # ------------------------------------------------------------------------------------- API
    def predict(self, images, batch_size: int = 32, mode: str = "beam", beam_size: int = 8, temperature: float = 1.7,
                alpha: float = 0.9) -> List[Dict[str, Any]]:
        """Same contract as the reference TRBA.predict (__init__.py:290-434)."""
        images_list = images if isinstance(images, list) else [images]
        if mode not in ("greedy", "beam"):
            raise ValueError(f"Unknown mode: {mode}")
        if not images_list:
            return []
        canv = torch.from_numpy(self._canvases(images_list)).to(self.device, non_blocking=True)
        return self._results(*self.recognize_canvases(canv, batch_size, mode, beam_size, temperature, alpha))
