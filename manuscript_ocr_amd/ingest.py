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


class ScanBatch:
    """Host side of the device entropy decode for a batch of parsed streams: descriptors, interval bounds and the files' bytes laid
    out as msocr_jpeg_entropy_decode_device takes them.  `pages[i]` = index into `descs` of stream i, or -1 (no restart interval,
    or a marker sequence the host decoder must judge)."""

    def __init__(self, parsed):
        lib = nat.lib()
        dsz = int(lib.msocr_jpeg_scan_desc_bytes())
        self.pages = []
        self.infos = []
        descs, bounds, chunks = [], [], []
        bytes_base = coef_base = first = 0
        self.max_intervals = 0
        for pr in parsed:
            if pr is None:
                self.pages.append(-1)
                continue
            info, buf, n = pr
            mcus = (int(info.blocks_w[0]) // int(info.hs[0])) * (int(info.blocks_h[0]) // int(info.vs[0]))
            d = np.zeros(dsz, dtype=np.uint8)
            b = np.empty(2 * mcus, dtype=np.uint32)   # at most one interval per MCU
            niv = int(lib.msocr_jpeg_scan_prepare_host(ctypes.addressof(buf), n, ctypes.byref(info), bytes_base, coef_base, first,
                                                       d.ctypes.data, b.ctypes.data, mcus))
            if niv <= 0:
                self.pages.append(-1)
                continue
            self.pages.append(len(descs))
            self.infos.append((info, coef_base))
            descs.append(d)
            bounds.append(b[: 2 * niv])
            chunks.append(np.frombuffer(buf, dtype=np.uint8, count=n))
            pad = (-n) % 16
            if pad:
                chunks.append(np.zeros(pad, dtype=np.uint8))
            bytes_base += n + pad
            coef_base += int(info.coef_total)
            first += niv
            self.max_intervals = max(self.max_intervals, niv)
        self.n_pages = len(descs)
        self.coef_total = coef_base
        if self.n_pages:
            self.descs = np.stack(descs)
            self.bounds = np.concatenate(bounds)
            self.bytes = np.concatenate(chunks)


def entropy_batch_host_twin(batch: ScanBatch):
    """The kernel's per-interval decoder on the CPU (tests; not a product path) -> (int16 coefficients of the batch, status)."""
    coef = np.empty(batch.coef_total, dtype=np.int16)
    status = np.empty(batch.n_pages, dtype=np.int32)
    nat.check(nat.lib().msocr_jpeg_entropy_decode_intervals_host(batch.bytes.ctypes.data, batch.descs.ctypes.data, batch.n_pages,
                                                                  batch.bounds.ctypes.data, coef.ctypes.data, batch.coef_total,
                                                                  status.ctypes.data), "jpeg_entropy_decode_intervals_host")
    return coef, status


def entropy_batch_device(batch: ScanBatch, device="cuda"):
    """Uploads the batch's file bytes, descriptors and interval bounds and runs the Huffman stage on the device (current stream)
    -> (int16 coefficient tensor of the batch, int32 status tensor [n_pages]); nothing is waited for."""
    import torch

    from . import ops
    up = lambda a: torch.from_numpy(a).pin_memory().to(device, non_blocking=True)
    bytes_dev, descs_dev, bounds_dev = up(batch.bytes), up(batch.descs), up(batch.bounds)
    coef = torch.empty(batch.coef_total, dtype=torch.int16, device=device)
    status = torch.empty(batch.n_pages, dtype=torch.int32, device=device)
    nat.check(nat.lib().msocr_jpeg_entropy_decode_device(bytes_dev.data_ptr(), descs_dev.data_ptr(), batch.n_pages, batch.max_intervals,
                                                          bounds_dev.data_ptr(), coef.data_ptr(), batch.coef_total, status.data_ptr(),
                                                          ops._stream()), "jpeg_entropy_decode_device")
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


def _entropy_into(parsed, coef_ptr):
    info, buf, n = parsed
    return nat.lib().msocr_jpeg_entropy_decode_host(ctypes.addressof(buf), n, ctypes.byref(info), coef_ptr)


_POOL = None
_SLOTS = {}   # slot -> [pinned int16 tensor, event of the last upload from it]


def _slot_buffer(slot, n, torch):
    ent = _SLOTS.get(slot)
    if ent is None or ent[0].numel() < n:
        ent = _SLOTS[slot] = [torch.empty(n, dtype=torch.int16).pin_memory(), None]
    elif ent[1] is not None:
        ent[1].synchronize()   # the previous batch's upload from this buffer has left the host
    return ent


def read_images_device(paths, device="cuda", device_entropy=None):
    """A batch of files -> list of device RGB tensors (None where read_image must take over).
    Files with a restart interval: the Huffman stage of the whole batch is ONE kernel launch on the device (`ScanBatch`,
    `entropy_batch_device`); the host reads the files and walks their markers, nothing else (`device_entropy=False` or
    MSOCR_JPEG_DEVICE_ENTROPY=0 sends them through the host decoder too).
    Files without: the entropy decode is one serial bit stream per FILE, but files are independent: the host stages of a batch run
    on a thread pool (the ctypes calls release the GIL), one page per core, into per-slot PINNED coefficient buffers that live across
    batches — fresh 9 MB arrays per page made the threads serialise on page faults (40 pages/s against 57 for the serial loop), and
    pinned memory lets the upload run asynchronously.  This thread uploads and launches the reconstruction page by page as the
    decodes finish."""
    global _POOL
    import torch
    from concurrent.futures import ThreadPoolExecutor

    from . import ops
    if device_entropy is None:
        device_entropy = os.environ.get("MSOCR_JPEG_DEVICE_ENTROPY", "1") != "0"
    if _POOL is None:
        _POOL = ThreadPoolExecutor(max_workers=max(1, min(32, (os.cpu_count() or 2) - 1)), thread_name_prefix="msocr-jpeg")
    parsed = list(_POOL.map(_read_and_parse, paths)) if len(paths) > 1 else [_read_and_parse(p) for p in paths]
    on_dev = {}
    if device_entropy and any(pr is not None for pr in parsed):
        batch = ScanBatch(parsed)
        if batch.n_pages:
            coef, status = entropy_batch_device(batch, device)
            imgs = [_reconstruct(info, coef[base:], device, torch, ops) for info, base in batch.infos]
            bad = status.cpu().numpy()   # the one wait of this path: a bad stream must go to the host reader, as the host decoder's verdict would
            on_dev = {i: (imgs[k] if bad[k] == 0 else None) for i, k in enumerate(batch.pages) if k >= 0}
    futs = []
    for i, pr in enumerate(parsed):
        if pr is None or i in on_dev:
            futs.append(None)
            continue
        ent = _slot_buffer(i, int(pr[0].coef_total), torch)   # main thread: allocation / pinning is not done from the workers
        futs.append((_POOL.submit(_entropy_into, pr, ent[0].data_ptr()), ent))
    out = []
    for i, (pr, f) in enumerate(zip(parsed, futs)):
        if i in on_dev:
            out.append(on_dev[i])
            continue
        if f is None or f[0].result() != 0:
            out.append(None)
            continue
        info, ent = pr[0], f[1]
        coef_dev = ent[0][: int(info.coef_total)].to(device, non_blocking=True)
        ent[1] = torch.cuda.Event()
        ent[1].record()
        out.append(_reconstruct(info, coef_dev, device, torch, ops))
    return out


def read_image_device(path, device="cuda"):
    """File -> device RGB tensor through the JPEG path, or None (not a file / not a supported JPEG: use read_image)."""
    return read_images_device([path], device)[0]
