"""NumPy restatement of the image preprocessing on the path (TEST INFRASTRUCTURE).

Reference call sites:
  EAST preprocess   detectors/_east/infer.py:127-132,304-305  (cv2.resize bilinear ->
                    ToTensor -> Normalize(.5,.5))
  TRBA preprocess   recognizers/_trba/data/transforms.py:62-120,185-193
                    (ResizeAndPadA -> A.Normalize(.5,.5,max 255) -> CHW)
The arithmetic of cv2.resize lives in OpenCV (opencv-python, requirements.txt),
absent here and not vendored: restated from OpenCV's published imgproc/resize.cpp
(8-bit INTER_LINEAR = 11-bit fixed-point two-pass; INTER_AREA = float area
table, integer-scale fast path).  PARITY UNPINNED for the resize itself — the
numeric parity chain starts at the resized tensor (SURVEY.md §7 hard part 6).
"""
import math

import numpy as np

_COEF_BITS = 11
_ONE = 1 << _COEF_BITS


def _rint_short(v):
    return np.clip(np.rint(v), -32768, 32767).astype(np.int32)


def _linear_tab(dst, src):
    """Per-output index (s0, s1) and fixed-point (a0, a1) along one axis (x-axis rule)."""
    scale = 1.0 / (float(dst) / float(src))
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    return s, f


def resize_linear_u8(img, dw, dh):
    """cv2.resize(img, (dw, dh), interpolation=INTER_LINEAR) for uint8 HxWxC."""
    sh, sw = img.shape[:2]
    if (dw, dh) == (sw, sh):
        return img.copy()
    if sw == 2 * dw and sh == 2 * dh:  # resize.cpp: LINEAR with exact 2x decimation == AREA fast
        return resize_area_u8(img, dw, dh)
    src = img.astype(np.int32)
    # horizontal
    sx, fx = _linear_tab(dw, sw)
    lo = sx < 0
    fx = np.where(lo, np.float32(0), fx)
    sx = np.where(lo, 0, sx)
    hi = sx >= sw - 1
    fx = np.where(hi, np.float32(0), fx)
    sx = np.where(hi, sw - 1, sx)
    a0 = _rint_short((np.float32(1) - fx) * np.float32(_ONE))
    a1 = _rint_short(fx * np.float32(_ONE))
    sx1 = np.minimum(sx + 1, sw - 1)
    hbuf = src[:, sx] * a0[None, :, None] + src[:, sx1] * a1[None, :, None]  # (sh, dw, C) int
    # vertical (rows clamped, coefficients NOT reset)
    sy, fy = _linear_tab(dh, sh)
    b0 = _rint_short((np.float32(1) - fy) * np.float32(_ONE))
    b1 = _rint_short(fy * np.float32(_ONE))
    r0 = np.clip(sy, 0, sh - 1)
    r1 = np.clip(sy + 1, 0, sh - 1)
    S0 = hbuf[r0] >> 4
    S1 = hbuf[r1] >> 4
    out = (((b0[:, None, None] * S0) >> 16) + ((b1[:, None, None] * S1) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def _area_tab(ssize, dsize):
    scale = float(ssize) / float(dsize)
    tab = []  # (di, si, alpha)
    for dx in range(dsize):
        fsx1 = dx * scale
        fsx2 = fsx1 + scale
        cell = min(scale, ssize - fsx1)
        sx1 = math.ceil(fsx1)
        sx2 = math.floor(fsx2)
        sx2 = min(sx2, ssize - 1)
        sx1 = min(sx1, sx2)
        if sx1 - fsx1 > 1e-3:
            tab.append((dx, sx1 - 1, np.float32((sx1 - fsx1) / cell)))
        for sx in range(sx1, sx2):
            tab.append((dx, sx, np.float32(1.0 / cell)))
        if fsx2 - sx2 > 1e-3:
            tab.append((dx, sx2, np.float32(min(min(fsx2 - sx2, 1.0), cell) / cell)))
    return tab


def resize_area_u8(img, dw, dh):
    """cv2.resize(..., interpolation=INTER_AREA) for uint8 HxWxC, shrinking in both axes."""
    sh, sw = img.shape[:2]
    if (dw, dh) == (sw, sh):
        return img.copy()
    if dw > sw or dh > sh:
        # OpenCV: INTER_AREA with any up-scaling axis falls back to the linear kernel
        # with area-style coordinates; not reachable from ResizeAndPadA except by
        # rounding of the other axis. Restated as plain linear (unpinned).
        return resize_linear_u8(img, dw, dh)
    if sw % dw == 0 and sh % dh == 0:  # integer-scale fast path
        kx, ky = sw // dw, sh // dh
        s = img.astype(np.int32).reshape(dh, ky, dw, kx, -1).sum(axis=(1, 3))
        if kx == 2 and ky == 2:
            return ((s + 2) >> 2).astype(np.uint8)
        return np.clip(np.rint(s.astype(np.float32) * np.float32(1.0 / (kx * ky))), 0, 255).astype(np.uint8)
    xtab, ytab = _area_tab(sw, dw), _area_tab(sh, dh)
    C = img.shape[2]
    src = img.astype(np.float32)
    hbuf = np.zeros((sh, dw, C), dtype=np.float32)
    for di, si, a in xtab:  # sequential float accumulation in table order
        hbuf[:, di] += src[:, si] * a
    out = np.zeros((dh, dw, C), dtype=np.float32)
    for di, si, b in ytab:
        out[di] += hbuf[si] * b
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def east_preprocess(img_u8, target_w, target_h):
    """HxWx3 u8 RGB -> 1x3xTHxTW f32 in [-1,1] (infer.py:304-305)."""
    r = resize_linear_u8(img_u8, target_w, target_h)
    x = r.astype(np.float32) / np.float32(255.0)  # ToTensor
    x = (x - np.float32(0.5)) / np.float32(0.5)  # Normalize(.5,.5)
    return np.ascontiguousarray(x.transpose(2, 0, 1))[None]


def resize_and_pad(img, img_h, img_w):
    """ResizeAndPadA.apply with align_h='left', align_v='center' (transforms.py:85-120)."""
    if img.ndim == 2:
        img = np.repeat(img[:, :, None], 3, axis=2)
    elif img.shape[2] == 4:
        img = img[:, :, :3]
    h, w = img.shape[:2]
    scale = min(img_h / max(h, 1), img_w / max(w, 1))
    new_w = max(1, int(round(w * scale)))  # Python banker's rounding
    new_h = max(1, int(round(h * scale)))
    if new_h < h or new_w < w:
        r = resize_area_u8(img, new_w, new_h)
    else:
        r = resize_linear_u8(img, new_w, new_h)
    canvas = np.full((img_h, img_w, 3), 255, dtype=img.dtype)
    x0 = 0
    y0 = (img_h - new_h) // 2
    x0 = max(0, min(x0, img_w - new_w))
    y0 = max(0, min(y0, img_h - new_h))
    canvas[y0:y0 + new_h, x0:x0 + new_w] = r
    return canvas


def trba_preprocess(img, img_h, img_w):
    """u8 crop -> 3 x img_h x img_w f32.  A.Normalize(mean=.5,std=.5,max_pixel_value=255):
    (x - 127.5) * (1/127.5) in f32."""
    c = resize_and_pad(img, img_h, img_w).astype(np.float32)
    c = (c - np.float32(0.5 * 255.0)) * np.float32(1.0 / (0.5 * 255.0))
    return np.ascontiguousarray(c.transpose(2, 0, 1))
