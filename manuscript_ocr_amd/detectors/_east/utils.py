"""Host helpers of the detector/pipeline boundary.

  read_image                                 <- reference detectors/_east/utils.py:477-497
  resolve_intersections                      <- :500-547
  sort_boxes_reading_order                   <- :550-607
  sort_boxes_reading_order_with_resolutions  <- :610-644
  draw_quads / visualize_page                <- :42-220 (same parameters and output contract; cv2's fillPoly /
                                                GaussianBlur / polylines restated with PIL + NumPy)
cv2 is not a dependency here: files are decoded with PIL (RGB), which is the reference's own
fallback branch.
"""
from pathlib import Path

import numpy as np
from PIL import Image, ImageDraw, ImageOps


def read_image(img_or_path):
    if isinstance(img_or_path, (str, Path)):
        try:
            with Image.open(str(img_or_path)) as pil_img:
                # cv2.imread (the reference's first branch, utils.py:480) applies the Exif orientation of the file;
                # parity unpinned (cv2 absent): restated with PIL's transpose of the same eight cases
                return np.array(ImageOps.exif_transpose(pil_img).convert("RGB"))
        except Exception as e:  # missing or undecodable file
            raise FileNotFoundError(f"Cannot read image with cv2 or PIL: {img_or_path}. Error: {e}")
    if isinstance(img_or_path, np.ndarray):
        return img_or_path
    raise TypeError(f"Unsupported type for image input: {type(img_or_path)}")


def _shrink(x0, y0, x1, y1):
    return int(x1 - (x1 - x0) * 0.1), int(y1 - (y1 - y0) * 0.1)


def resolve_intersections(boxes):
    """Shrink both members of every intersecting pair by 10 % (right/bottom edge, int truncation), at most
    50 sweeps in (i, j>i) order — the reference's O(n^2) Python double loop (utils.py:500-547), evaluated
    with one vectorised "next intersecting j" search per hit.  Exactly equivalent: boxes only ever shrink
    towards their fixed top-left corner, so a j that does not intersect box i now cannot intersect it later
    in the same sweep of i."""
    n = len(boxes)
    if n == 0:
        return []
    out = [tuple(b) for b in boxes]
    arr = np.array([[int(v) for v in b] for b in boxes], dtype=np.int64).reshape(n, 4)
    for _ in range(50):
        dirty = False
        for i in range(n - 1):
            j0 = i + 1
            while j0 < n:
                bi = arr[i]
                rest = arr[j0:]
                hit = ~((bi[2] <= rest[:, 0]) | (rest[:, 2] <= bi[0]) | (bi[3] <= rest[:, 1]) | (rest[:, 3] <= bi[1]))
                k = np.flatnonzero(hit)
                if len(k) == 0:
                    break
                j = j0 + int(k[0])
                for t in (i, j):
                    x0, y0, x1, y1 = out[t]
                    nx, ny = _shrink(x0, y0, x1, y1)
                    out[t] = (x0, y0, nx, ny)
                    arr[t, 2], arr[t, 3] = nx, ny
                dirty = True
                j0 = j + 1
        if not dirty:
            break
    return out


def sort_boxes_reading_order(boxes, y_tol_ratio=0.6, x_gap_ratio=np.inf):
    """Group into lines (|cy - mean line cy| <= y_tol*avg_h, first accepting line), sort lines by mean cy
    and words by x_min (utils.py:550-607).  Line means are kept as running sums: with integer box
    coordinates every centre is a multiple of 0.5, so the sums are exact and equal np.mean's."""
    if not boxes:
        return []
    avg_h = np.mean([b[3] - b[1] for b in boxes])
    tol, gap = avg_h * y_tol_ratio, avg_h * x_gap_ratio
    exact = all(float(v).is_integer() for b in boxes for v in b)
    lines, sums, maxx = [], [], []
    for b in sorted(boxes, key=lambda b: (b[1] + b[3]) / 2):
        cy = (b[1] + b[3]) / 2
        home = -1
        for li in range(len(lines)):
            line_cy = sums[li] / len(lines[li]) if exact else np.mean([(v[1] + v[3]) / 2 for v in lines[li]])
            if abs(cy - line_cy) <= tol and (b[0] - maxx[li]) <= gap:
                home = li
                break
        if home < 0:
            lines.append([b]), sums.append(float(cy)), maxx.append(b[2])
        else:
            lines[home].append(b)
            sums[home] += float(cy)
            maxx[home] = max(maxx[home], b[2])
    keyed = [(s / len(ln) if exact else np.mean([(v[1] + v[3]) / 2 for v in ln]), ln) for s, ln in zip(sums, lines)]
    keyed.sort(key=lambda t: t[0])  # stable, like list.sort on the reference's key
    return [b for _, ln in keyed for b in sorted(ln, key=lambda b: b[0])]


def sort_boxes_reading_order_with_resolutions(boxes, y_tol_ratio=0.6, x_gap_ratio=np.inf):
    shrunk = resolve_intersections(boxes)
    back = dict(zip(shrunk, boxes))  # identical shrunk boxes collapse; the later one wins (as the reference)
    return [back[b] for b in sort_boxes_reading_order(shrunk, y_tol_ratio, x_gap_ratio)]


def _gaussian_blur_f32(mask, k):
    """cv2.GaussianBlur(mask, (k, k), 0) restated: separable kernel exp(-(x - c)^2 / (2 sigma^2)) normalised to 1 with OpenCV's
    sigma = 0.3 * ((k - 1) * 0.5 - 1) + 0.8 for a non-positive sigma argument, BORDER_REFLECT_101 (parity unpinned: cv2 absent;
    OpenCV uses fixed tables for k <= 7, which the default k = 11 does not hit)."""
    sigma = 0.3 * ((k - 1) * 0.5 - 1) + 0.8
    x = np.arange(k, dtype=np.float64) - (k - 1) / 2
    g = np.exp(-(x * x) / (2 * sigma * sigma))
    g = (g / g.sum()).astype(np.float32)
    r = k // 2
    h, w = mask.shape
    if h <= r or w <= r:  # reflect_101 needs at least r + 1 samples; tiny images: clamp instead
        padded = np.pad(mask, r, mode="edge")
    else:
        padded = np.pad(mask, r, mode="reflect")
    tmp = np.zeros((h + 2 * r, w), dtype=np.float32)
    for i in range(k):
        tmp += g[i] * padded[:, i:i + w]
    out = np.zeros((h, w), dtype=np.float32)
    for i in range(k):
        out += g[i] * tmp[i:i + h, :]
    return out


def draw_quads(image, quads, color=(0, 0, 0), thickness=1, dark_alpha=0.5, blur_ksize=11):
    """The reference's overlay style (detectors/_east/utils.py:42-92): the page darkened by `dark_alpha` outside the quads, the
    inside kept at full brightness through a Gaussian-blurred polygon mask, quad outlines on top.  -> PIL.Image (RGB).
    cv2.fillPoly / GaussianBlur / polylines are restated with PIL + NumPy (cv2 is not a dependency here); vertex coordinates are
    truncated to int32 as the reference does."""
    img = np.asarray(image).copy()
    if quads is None or len(quads) == 0:
        return Image.fromarray(img)
    quads = np.asarray(quads)
    h, w = img.shape[:2]
    dark_bg = (img.astype(np.float32) * (1 - dark_alpha)).astype(np.uint8)
    mask_img = Image.new("L", (w, h), 0)
    md = ImageDraw.Draw(mask_img)
    polys = [[(int(x), int(y)) for x, y in np.asarray(q[:8]).reshape(4, 2).astype(np.int32)] for q in quads]
    for pts in polys:
        md.polygon(pts, fill=1, outline=1)
    k = blur_ksize if blur_ksize % 2 == 1 else blur_ksize + 1
    mask = np.clip(_gaussian_blur_f32(np.asarray(mask_img, dtype=np.float32), k), 0.0, 1.0)[:, :, None]
    out = img.astype(np.float32) * mask + dark_bg.astype(np.float32) * (1 - mask)
    out_img = Image.fromarray(np.clip(out, 0, 255).astype(np.uint8))
    d = ImageDraw.Draw(out_img)
    for pts in polys:
        d.line(pts + [pts[0]], fill=tuple(int(c) for c in color), width=int(thickness))
    return out_img


def visualize_page(image, page, *, show_order=False, color=(0, 0, 255), thickness=2, dark_alpha=0.3, blur_ksize=11,
                   line_color=(0, 255, 0), number_color=(255, 255, 255), number_bg=(0, 0, 0)):
    """Drop-in for the reference's visualize_page (detectors/_east/utils.py:95-220): same keyword-only parameters and defaults,
    same output contract — an RGB PIL image of the page's size with every word's quad drawn in the EAST overlay style
    (draw_quads) and, with show_order, green lines joining consecutive word centres plus a numbered 24 x 24 box (1-based) at each
    centre.  A page without words returns the input itself (PIL input) or Image.fromarray(input)."""
    img = np.array(image.convert("RGB")) if isinstance(image, Image.Image) else np.asarray(image).copy()
    quads, words = [], []
    for block in page.blocks:
        for w in block.words:
            quads.append(np.array(w.polygon).reshape(-1))
            words.append(w)
    if not quads:
        return Image.fromarray(img) if isinstance(image, np.ndarray) else image
    out = draw_quads(img, np.stack(quads, axis=0), color=color, thickness=thickness, dark_alpha=dark_alpha, blur_ksize=blur_ksize)
    if show_order:
        draw = ImageDraw.Draw(out)
        centers = []
        for w in words:
            xs, ys = [p[0] for p in w.polygon], [p[1] for p in w.polygon]
            centers.append((sum(xs) / len(xs), sum(ys) / len(ys)))
        for p, c in zip(centers, centers[1:]):
            draw.line([p, c], fill=line_color, width=3)
        for idx, (cx, cy) in enumerate(centers, start=1):
            draw.rectangle([cx - 12, cy - 12, cx + 12, cy + 12], fill=number_bg)
            draw.text((cx - 6, cy - 8), str(idx), fill=number_color)
    return out
