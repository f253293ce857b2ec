"""Image ingest on the device: JPEG file -> RGB u8 tensor in HBM (msocr_jpeg_* of libmsocr.so, csrc/jpeg.hip).

Replaces the file branch of the reference's read_image (detectors/_east/utils.py:477-497: cv2.imread + BGR->RGB, PIL
fallback — both libjpeg-turbo with default settings).  Files written with a restart interval (DRI) are decoded entirely on
the MI355X — the file's bytes are uploaded as they are and one thread per interval runs the Huffman stage
(`entropy_batch_device`); for the others the serial Huffman decode runs on the host.  Dequantisation, inverse DCT, chroma
upsampling and colour conversion always run on the device, so the decoded page (9.4 MB at 2048x1536, 69 MB for the
reference's 5390x4250 example page) is produced in HBM instead of crossing PCIe.
Formats outside the kernel's scope (progressive, CMYK, 12-bit, PNG, ...) return None: callers fall back to read_image.
"""
import ctypes
import os

import numpy as np

from . import _native as nat


def _parse(data: bytes):
    info = nat.JpegInfo()
    buf = (ctypes.c_uint8 * len(data)).from_buffer_copy(data)
    rc = nat.lib().msocr_jpeg_parse_host(ctypes.addressof(buf), len(data), ctypes.byref(info))
    if rc != 0 or not info.supported:
        return None, buf
    return info, buf


def jpeg_coefficients(data: bytes):
    """Host stage: (info, int16 coefficient array) of a supported JPEG, or None."""
    info, buf = _parse(data)
    if info is None:
        return None
    coef = np.empty(int(info.coef_total), dtype=np.int16)
    rc = nat.lib().msocr_jpeg_entropy_decode_host(ctypes.addressof(buf), len(data), ctypes.byref(info), coef.ctypes.data)
    if rc != 0:
        return None
    return info, coef


def decode_jpeg_host(data: bytes):
    """The whole decode on the CPU through the host twin of the device stage (tests; not a product path)."""
    r = jpeg_coefficients(data)
    if r is None:
        return None
    info, coef = r
    out = np.empty((info.height, info.width, 3), dtype=np.uint8)
    nat.check(nat.lib().msocr_jpeg_reconstruct_host(ctypes.byref(info), coef.ctypes.data, out.ctypes.data), "jpeg_reconstruct_host")
    return out


def decode_jpeg_device(data: bytes, device="cuda", device_entropy=True):
    """JPEG bytes -> [H, W, 3] u8 tensor on the device (current stream), or None when the stream is not supported.  A stream with
    a restart interval takes the device Huffman stage (`device_entropy=False`: the host decoder, as for all other streams)."""
    import torch

    from . import ops
    info, buf = _parse(data)
    if info is None:
        return None
    if device_entropy:
        batch = ScanBatch([(info, buf, len(data))])
        if batch.n_pages:
            coef, status = entropy_batch_device(batch, device)
            img = _reconstruct(info, coef, device, torch, ops)
            return img if int(status.cpu()[0]) == 0 else None
    coef = np.empty(int(info.coef_total), dtype=np.int16)
    if nat.lib().msocr_jpeg_entropy_decode_host(ctypes.addressof(buf), len(data), ctypes.byref(info), coef.ctypes.data) != 0:
        return None
    return _reconstruct(info, torch.from_numpy(coef).to(device, non_blocking=True), device, torch, ops)


def _prepare(ptr, n, info, bytes_base):
    """Marker walk of one parsed stream -> (descriptor bytes, interval bounds) or None (no restart interval / host decoder's case)."""
    lib = nat.lib()
    mcus = (int(info.blocks_w[0]) // int(info.hs[0])) * (int(info.blocks_h[0]) // int(info.vs[0]))
    d = np.zeros(int(lib.msocr_jpeg_scan_desc_bytes()), dtype=np.uint8)
    b = np.empty(2 * mcus, dtype=np.uint32)   # at most one interval per MCU
    niv = int(lib.msocr_jpeg_scan_prepare_host(ptr, n, ctypes.byref(info), bytes_base, d.ctypes.data, b.ctypes.data, mcus))
    return None if niv <= 0 else (d, b[: 2 * niv])


class ScanBatch:
    """Host side of the device entropy decode for a batch of parsed streams: descriptors, interval bounds, per-page bases and the
    files' bytes laid out as msocr_jpeg_entropy_decode_device takes them.  `pages[i]` = index into `descs` of stream i, or -1 (no
    restart interval, or a marker sequence the host decoder must judge).  `parsed[i]` = (info, ctypes buffer, length) or None;
    `prepared[i]` (optional) = what `_prepare` returned for stream i with `bytes_base[i]` as its base inside `bytes`."""

    def __init__(self, parsed, prepared=None, bytes_=None):
        self.pages, self.infos = [], []
        descs, bounds, chunks, base = [], [], [], []
        pos = coef_base = first = 0
        self.max_intervals = 0
        for i, pr in enumerate(parsed):
            if pr is None:
                self.pages.append(-1)
                continue
            info, buf, n = pr
            r = prepared[i] if prepared is not None else _prepare(ctypes.addressof(buf), n, info, pos)
            if r is None:
                self.pages.append(-1)
                continue
            self.pages.append(len(descs))
            self.infos.append((info, coef_base))
            descs.append(r[0])
            bounds.append(r[1])
            base.append((coef_base, first))
            if prepared is None:
                chunks.append(np.frombuffer(buf, dtype=np.uint8, count=n))
                pad = (-n) % 16
                if pad:
                    chunks.append(np.zeros(pad, dtype=np.uint8))
                pos += n + pad
            coef_base += int(info.coef_total)
            first += len(r[1]) // 2
            self.max_intervals = max(self.max_intervals, len(r[1]) // 2)
        self.n_pages = len(descs)
        self.coef_total = coef_base
        if self.n_pages:
            self.descs = np.stack(descs)
            self.bounds = np.concatenate(bounds)
            self.page_base = np.array(base, dtype=np.int64)
            self.bytes = bytes_ if prepared is not None else np.concatenate(chunks)


def entropy_batch_host_twin(batch: ScanBatch):
    """The kernel's per-interval decoder on the CPU (tests; not a product path) -> (int16 coefficients of the batch, status)."""
    coef = np.empty(batch.coef_total, dtype=np.int16)
    status = np.empty(batch.n_pages, dtype=np.int32)
    nat.check(nat.lib().msocr_jpeg_entropy_decode_intervals_host(batch.bytes.ctypes.data, batch.descs.ctypes.data, batch.n_pages,
                                                                  batch.bounds.ctypes.data, batch.page_base.ctypes.data, coef.ctypes.data,
                                                                  batch.coef_total, status.ctypes.data), "jpeg_entropy_decode_intervals_host")
    return coef, status


def entropy_batch_device(batch: ScanBatch, device="cuda", bytes_dev=None):
    """Uploads the batch's file bytes (unless `bytes_dev` already holds them), descriptors and interval bounds and runs the Huffman
    stage on the device (current stream) -> (int16 coefficient tensor of the batch, int32 status tensor [n_pages]); nothing is
    waited for."""
    import torch

    from . import ops
    up = lambda a: torch.from_numpy(a).to(device)
    if bytes_dev is None:
        bytes_dev = torch.from_numpy(batch.bytes).pin_memory().to(device, non_blocking=True)
    descs_dev, bounds_dev, base_dev = up(batch.descs), up(batch.bounds), up(batch.page_base)
    coef = torch.empty(batch.coef_total, dtype=torch.int16, device=device)
    status = torch.empty(batch.n_pages, dtype=torch.int32, device=device)
    nat.check(nat.lib().msocr_jpeg_entropy_decode_device(bytes_dev.data_ptr(), descs_dev.data_ptr(), batch.n_pages, batch.max_intervals,
                                                          bounds_dev.data_ptr(), base_dev.data_ptr(), coef.data_ptr(), batch.coef_total,
                                                          status.data_ptr(), ops._stream()), "jpeg_entropy_decode_device")
    return coef, status


def _reconstruct(info, coef_dev, device, torch, ops):
    ws = torch.empty((nat.lib().msocr_jpeg_workspace_bytes(ctypes.byref(info)),), dtype=torch.uint8, device=device)
    img = torch.empty((info.height, info.width, 3), dtype=torch.uint8, device=device)
    nat.check(nat.lib().msocr_jpeg_reconstruct(ctypes.byref(info), coef_dev.data_ptr(), ws.data_ptr(), img.data_ptr(), ops._stream()),
              "jpeg_reconstruct")
    return img


def _read_and_parse(path):
    """File -> (info, ctypes byte buffer, length), or None (not a file / not a supported JPEG)."""
    if not isinstance(path, (str, os.PathLike)) or not os.path.isfile(path):
        return None
    with open(path, "rb") as f:
        data = f.read()
    if data[:2] != b"\xff\xd8":
        return None
    info, buf = _parse(data)
    return None if info is None else (info, buf, len(data))


_POOL = None
_SLOTS = {}   # slot -> [pinned tensor, event of the last upload from it]; reused across batches (one reader thread at a time)


def _slot_buffer(slot, n, torch, dtype=None):
    ent = _SLOTS.get(slot)
    if ent is None or ent[0].numel() < n:
        ent = _SLOTS[slot] = [torch.empty(n + n // 4, dtype=dtype or torch.int16).pin_memory(), None]
    elif ent[1] is not None:
        ent[1].synchronize()   # the previous batch's upload from this buffer has left the host
    return ent


def _load(path, arr, off, n, want_device):
    """Worker: file -> its slice of the pinned batch buffer, header parse, marker walk.  -> (info, prepared or None) or None."""
    try:
        with open(path, "rb") as f:
            if f.readinto(memoryview(arr[off: off + n])) != n:
                return None
    except OSError:
        return None
    if n < 4 or arr[off] != 0xFF or arr[off + 1] != 0xD8:
        return None
    info = nat.JpegInfo()
    ptr = arr.ctypes.data + off
    if nat.lib().msocr_jpeg_parse_host(ptr, n, ctypes.byref(info)) != 0 or not info.supported:
        return None
    return info, (_prepare(ptr, n, info, off) if want_device else None)


def check_pending(pending):
    """Deferred verdict of the device Huffman stage (`read_images_device(..., defer_status=True)`): waits for the status words of
    that batch (a copy that was queued right behind the kernels; by the time a caller asks, long done) -> indices of the pages whose
    stream the kernel flagged as bad (they must be read again through the host path)."""
    if pending is None:
        return []
    st_host, ev, idx = pending
    ev.synchronize()
    return [i for i, s in zip(idx, st_host.tolist()) if s != 0]


def read_images_device(paths, device="cuda", device_entropy=None, defer_status=False):
    """A batch of files -> list of device RGB tensors (None where read_image must take over).
    A thread pool reads every file into its slice of ONE pinned batch buffer (reused across batches), parses its headers and walks
    its markers (the ctypes calls release the GIL).  Files with a restart interval: the buffer is uploaded as it is and the Huffman
    stage of the whole batch is ONE kernel launch (`entropy_batch_device`, one thread per interval) — unless the intervals are so
    long that the serial chain inside one of them would take longer than a host core needs for the file
    (MSOCR_JPEG_DEVICE_MAX_INTERVAL bytes, default 8192: 1.8 ms per KB of interval on the device against ~15 ms per 1.6 MB file on a
    host core; `device_entropy=True` / False or MSOCR_JPEG_DEVICE_ENTROPY=1 / 0 force one path).
    Files without: the entropy decode is one serial bit stream per FILE, but files are independent: the pool decodes one page per
    core into per-slot PINNED coefficient buffers that live across batches (fresh 9 MB arrays per page made the threads serialise
    on page faults), this thread uploads and launches the reconstruction page by page as the decodes finish.
    defer_status=True -> (list, pending): the device path's one host wait — the kernel's per-page verdict — is NOT taken here; the
    caller asks `check_pending(pending)` later (the pipeline does, when it waits for the detector anyway), so that submitting a
    batch never waits for the device."""
    global _POOL
    import torch
    from concurrent.futures import ThreadPoolExecutor

    from . import ops
    env = os.environ.get("MSOCR_JPEG_DEVICE_ENTROPY")
    if device_entropy is None and env is not None:
        device_entropy = env != "0"
    max_iv = int(os.environ.get("MSOCR_JPEG_DEVICE_MAX_INTERVAL", "8192"))
    if _POOL is None:
        # one process per GPU: the ranks of a node share its cores (LOCAL_WORLD_SIZE is set by torch.distributed.run)
        share = (os.cpu_count() or 2) // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
        _POOL = ThreadPoolExecutor(max_workers=max(1, min(32, share - 1)), thread_name_prefix="msocr-jpeg")
    sizes = [os.path.getsize(p) if isinstance(p, (str, os.PathLike)) and os.path.isfile(p) else -1 for p in paths]
    offs, total = [], 0
    for n in sizes:
        offs.append(total)
        total += (max(n, 0) + 15) // 16 * 16
    pending = None
    if total == 0:
        return ([None] * len(paths), None) if defer_status else [None] * len(paths)
    ent = _slot_buffer("bytes", total, torch, torch.uint8)
    arr = ent[0].numpy()
    want = device_entropy is not False
    jobs = [(_POOL.submit(_load, p, arr, o, n, want) if n >= 0 else None) for p, o, n in zip(paths, offs, sizes)]
    loaded = [j.result() if j is not None else None for j in jobs]
    on_dev = {}
    if want:
        prepared = [None if (r is None or r[1] is None) else r[1] for r in loaded]
        if device_entropy is None:   # the policy: no interval of the page longer than max_iv bytes
            prepared = [None if (r is None or int((r[1][1::2] - r[1][0::2]).max()) > max_iv) else r for r in prepared]
        parsed = [None if r is None else (r[0], None, n) for r, n in zip(loaded, sizes)]
        batch = ScanBatch(parsed, prepared, arr)
        if batch.n_pages:
            bytes_dev = ent[0][:total].to(device, non_blocking=True)
            ent[1] = torch.cuda.Event()
            ent[1].record()
            coef, status = entropy_batch_device(batch, device, bytes_dev)
            imgs = [_reconstruct(info, coef[base:], device, torch, ops) for info, base in batch.infos]
            idx = [i for i, k in enumerate(batch.pages) if k >= 0]
            if defer_status:
                st_host = torch.empty(batch.n_pages, dtype=torch.int32).pin_memory()
                st_host.copy_(status, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                pending = (st_host, ev, idx)
                on_dev = {i: imgs[batch.pages[i]] for i in idx}
            else:
                bad = status.cpu().numpy()   # the one wait of this path: a bad stream must go to the host reader, as the host decoder's verdict would
                on_dev = {i: (imgs[batch.pages[i]] if bad[batch.pages[i]] == 0 else None) for i in idx}
    lib = nat.lib()
    futs = []
    for i, r in enumerate(loaded):
        if r is None or i in on_dev:
            futs.append(None)
            continue
        slot = _slot_buffer(i, int(r[0].coef_total), torch)   # main thread: allocation / pinning is not done from the workers
        futs.append((_POOL.submit(lib.msocr_jpeg_entropy_decode_host, arr.ctypes.data + offs[i], sizes[i], ctypes.byref(r[0]), slot[0].data_ptr()), slot))
    out = []
    for i, (r, f) in enumerate(zip(loaded, futs)):
        if i in on_dev:
            out.append(on_dev[i])
            continue
        if f is None or f[0].result() != 0:
            out.append(None)
            continue
        info, slot = r[0], f[1]
        coef_dev = slot[0][: int(info.coef_total)].to(device, non_blocking=True)
        slot[1] = torch.cuda.Event()
        slot[1].record()
        out.append(_reconstruct(info, coef_dev, device, torch, ops))
    return (out, pending) if defer_status else out


def read_image_device(path, device="cuda"):
    """File -> device RGB tensor through the JPEG path, or None (not a file / not a supported JPEG: use read_image)."""
    return read_images_device([path], device)[0]
