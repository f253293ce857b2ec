"""Host helpers of the detector/pipeline boundary.

  read_image                                 <- reference detectors/_east/utils.py:477-497
  resolve_intersections                      <- :500-547
  sort_boxes_reading_order                   <- :550-607
  sort_boxes_reading_order_with_resolutions  <- :610-644
  visualize_page                             <- :42-220 (minimal PIL renderer; the reference's
                                                cv2 drawing is out of the hot-path scope)
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


def visualize_page(image, page, show_order=False, color=(0, 200, 0), width=2):
    """Draw word polygons (and reading-order indices) on a copy of the image -> PIL.Image."""
    pil = image.copy() if isinstance(image, Image.Image) else Image.fromarray(np.asarray(image))
    pil = pil.convert("RGB")
    draw = ImageDraw.Draw(pil)
    k = 0
    for block in page.blocks:
        for w in block.words:
            pts = [(float(x), float(y)) for x, y in w.polygon]
            if len(pts) >= 2:
                draw.line(pts + [pts[0]], fill=color, width=width)
            if show_order and pts:
                draw.text(pts[0], str(k), fill=(220, 0, 0))
            k += 1
    return pil
