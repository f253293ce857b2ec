"""Host-side crop preprocessing and token decoding of the recogniser.

  load_charset / decode_tokens   <- reference recognizers/_trba/data/transforms.py:39-59,196-206
  resize_and_pad                 <- ResizeAndPadA.apply, transforms.py:85-120 (align left / centre)
cv2 is not a dependency: INTER_AREA / INTER_LINEAR for uint8 are implemented here from
OpenCV's published algorithms (table-driven float area resize; 11-bit fixed-point bilinear).
Round 1 resizes crops on the host and normalises on the device (msocr_normalize_u8 mode 1);
a device crop+resize kernel is the next item.
"""
import math

import numpy as np


def load_charset(charset_path):
    itos = []
    with open(charset_path, "r", encoding="utf-8") as f:
        for line in f:
            tok = line.rstrip("\n")
            if tok == "":
                continue
            itos.append(tok)
    return itos, {s: i for i, s in enumerate(itos)}


def decode_tokens(ids, itos, pad_id, eos_id, blank_id=None):
    chars = []
    for t in ids:
        t = int(t)
        if t == eos_id:
            break
        if t == pad_id or (blank_id is not None and t == blank_id):
            continue
        chars.append(itos[t])
    return "".join(chars)


# ----------------------------------------------------------------------------------------- resize
def _sat_short(v):
    return np.clip(np.rint(v), -32768, 32767).astype(np.int32)


def _axis_linear(dst, src):
    scale = 1.0 / (float(dst) / float(src))
    f = ((np.arange(dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    return s, (f - s.astype(np.float32)).astype(np.float32)


def resize_linear(img, dw, dh):
    sh, sw = img.shape[:2]
    if (dw, dh) == (sw, sh):
        return img.copy()
    if sw == 2 * dw and sh == 2 * dh:
        return resize_area(img, dw, dh)
    src = img.astype(np.int32)
    sx, fx = _axis_linear(dw, sw)
    fx = np.where((sx < 0) | (sx >= sw - 1), np.float32(0), fx)
    sx = np.clip(sx, 0, sw - 1)
    a0, a1 = _sat_short((np.float32(1) - fx) * np.float32(2048)), _sat_short(fx * np.float32(2048))
    rows = src[:, sx] * a0[None, :, None] + src[:, np.minimum(sx + 1, sw - 1)] * a1[None, :, None]
    sy, fy = _axis_linear(dh, sh)
    b0, b1 = _sat_short((np.float32(1) - fy) * np.float32(2048)), _sat_short(fy * np.float32(2048))
    top, bot = rows[np.clip(sy, 0, sh - 1)] >> 4, rows[np.clip(sy + 1, 0, sh - 1)] >> 4
    out = (((b0[:, None, None] * top) >> 16) + ((b1[:, None, None] * bot) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def _area_table(ssize, dsize):
    scale = float(ssize) / float(dsize)
    di, si, al = [], [], []
    for d in range(dsize):
        f1 = d * scale
        f2 = f1 + scale
        cell = min(scale, ssize - f1)
        s1 = math.ceil(f1)
        s2 = min(math.floor(f2), ssize - 1)
        s1 = min(s1, s2)
        if s1 - f1 > 1e-3:
            di.append(d), si.append(s1 - 1), al.append((s1 - f1) / cell)
        for s in range(s1, s2):
            di.append(d), si.append(s), al.append(1.0 / cell)
        if f2 - s2 > 1e-3:
            di.append(d), si.append(s2), al.append(min(min(f2 - s2, 1.0), cell) / cell)
    return np.asarray(di), np.asarray(si), np.asarray(al, dtype=np.float32)


def resize_area(img, dw, dh):
    sh, sw = img.shape[:2]
    if (dw, dh) == (sw, sh):
        return img.copy()
    if dw > sw or dh > sh:
        return resize_linear(img, dw, dh)
    if sw % dw == 0 and sh % dh == 0:
        kx, ky = sw // dw, sh // dh
        s = img.astype(np.int32).reshape(dh, ky, dw, kx, -1).sum(axis=(1, 3))
        if kx == 2 and ky == 2:
            return ((s + 2) >> 2).astype(np.uint8)
        return np.clip(np.rint(s.astype(np.float32) * np.float32(1.0 / (kx * ky))), 0, 255).astype(np.uint8)
    src = img.astype(np.float32)
    xd, xs, xa = _area_table(sw, dw)
    yd, ys, ya = _area_table(sh, dh)
    hbuf = np.zeros((sh, dw, img.shape[2]), dtype=np.float32)
    for d, s, a in zip(xd, xs, xa):  # sequential f32 accumulation in table order, like OpenCV
        hbuf[:, d] += src[:, s] * a
    out = np.zeros((dh, dw, img.shape[2]), dtype=np.float32)
    for d, s, b in zip(yd, ys, ya):
        out[d] += hbuf[s] * b
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def resize_and_pad(img, img_h, img_w):
    """Aspect-preserving resize (AREA if any axis shrinks else LINEAR), pasted left / vertically centred on white."""
    if img.ndim == 2:
        img = np.repeat(img[:, :, None], 3, axis=2)
    elif img.shape[2] == 4:
        img = img[:, :, :3]
    h, w = img.shape[:2]
    scale = min(img_h / max(h, 1), img_w / max(w, 1))
    new_w, new_h = max(1, int(round(w * scale))), max(1, int(round(h * scale)))
    small = resize_area(img, new_w, new_h) if (new_h < h or new_w < w) else resize_linear(img, new_w, new_h)
    canvas = np.full((img_h, img_w, 3), 255, dtype=img.dtype)
    x0 = max(0, min(0, img_w - new_w))
    y0 = max(0, min((img_h - new_h) // 2, img_h - new_h))
    canvas[y0:y0 + new_h, x0:x0 + new_w] = small
    return canvas
