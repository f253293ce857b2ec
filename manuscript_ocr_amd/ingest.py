"""Image ingest on the device: JPEG file -> RGB u8 tensor in HBM (msocr_jpeg_* of libmsocr.so, csrc/jpeg.hip).

Replaces the file branch of the reference's read_image (detectors/_east/utils.py:477-497: cv2.imread + BGR->RGB, PIL
fallback — both libjpeg-turbo with default settings).  The serial Huffman decode runs on the host; dequantisation, inverse
DCT, chroma upsampling and colour conversion run on the MI355X, so the decoded page (9.4 MB at 2048x1536, 69 MB for the
reference's 5390x4250 example page) is produced in HBM instead of crossing PCIe (only the 2-byte coefficients do).
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


def decode_jpeg_device(data: bytes, device="cuda"):
    """JPEG bytes -> [H, W, 3] u8 tensor on the device (current stream), or None when the stream is not supported."""
    import torch

    from . import ops
    r = jpeg_coefficients(data)
    if r is None:
        return None
    info, coef = r
    coef_dev = torch.from_numpy(coef).to(device, non_blocking=True)
    ws = torch.empty((nat.lib().msocr_jpeg_workspace_bytes(ctypes.byref(info)),), dtype=torch.uint8, device=device)
    out = torch.empty((info.height, info.width, 3), dtype=torch.uint8, device=device)
    nat.check(nat.lib().msocr_jpeg_reconstruct(ctypes.byref(info), coef_dev.data_ptr(), ws.data_ptr(), out.data_ptr(), ops._stream()),
              "jpeg_reconstruct")
    return out


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


def read_images_device(paths, device="cuda"):
    """A batch of files -> list of device RGB tensors (None where read_image must take over).  The entropy decode of a baseline
    JPEG without restart markers is one serial bit stream per FILE, but files are independent: the host stages of a batch run on a
    thread pool (the ctypes calls release the GIL), one page per core, into per-slot PINNED coefficient buffers that live across
    batches — fresh 9 MB arrays per page made the threads serialise on page faults (40 pages/s against 57 for the serial loop), and
    pinned memory lets the upload run asynchronously.  This thread uploads and launches the reconstruction page by page as the
    decodes finish."""
    global _POOL
    import torch
    if len(paths) <= 1:
        return [read_image_device(p, device) for p in paths]
    from concurrent.futures import ThreadPoolExecutor

    from . import ops
    if _POOL is None:
        _POOL = ThreadPoolExecutor(max_workers=max(1, min(32, (os.cpu_count() or 2) - 1)), thread_name_prefix="msocr-jpeg")
    parsed = list(_POOL.map(_read_and_parse, paths))
    futs = []
    for i, pr in enumerate(parsed):
        if pr is None:
            futs.append(None)
            continue
        ent = _slot_buffer(i, int(pr[0].coef_total), torch)   # main thread: allocation / pinning is not done from the workers
        futs.append((_POOL.submit(_entropy_into, pr, ent[0].data_ptr()), ent))
    out = []
    for pr, f in zip(parsed, futs):
        if f is None or f[0].result() != 0:
            out.append(None)
            continue
        info, ent = pr[0], f[1]
        coef_dev = ent[0][: int(info.coef_total)].to(device, non_blocking=True)
        ent[1] = torch.cuda.Event()
        ent[1].record()
        ws = torch.empty((nat.lib().msocr_jpeg_workspace_bytes(ctypes.byref(info)),), dtype=torch.uint8, device=device)
        img = torch.empty((info.height, info.width, 3), dtype=torch.uint8, device=device)
        nat.check(nat.lib().msocr_jpeg_reconstruct(ctypes.byref(info), coef_dev.data_ptr(), ws.data_ptr(), img.data_ptr(), ops._stream()),
                  "jpeg_reconstruct")
        out.append(img)
    return out


def read_image_device(path, device="cuda"):
    """File -> device RGB tensor through the JPEG path, or None (not a file / not a supported JPEG: use read_image)."""
    if not isinstance(path, (str, os.PathLike)) or not os.path.isfile(path):
        return None
    with open(path, "rb") as f:
        data = f.read()
    if data[:2] != b"\xff\xd8":
        return None
    return decode_jpeg_device(data, device)
