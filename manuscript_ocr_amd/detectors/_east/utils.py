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
from PIL import Image, ImageDraw


def read_image(img_or_path):
    if isinstance(img_or_path, (str, Path)):
        try:
            with Image.open(str(img_or_path)) as pil_img:
                return np.array(pil_img.convert("RGB"))
        except Exception as e:  # missing or undecodable file
            raise FileNotFoundError(f"Cannot read image with cv2 or PIL: {img_or_path}. Error: {e}")
    if isinstance(img_or_path, np.ndarray):
        return img_or_path
    raise TypeError(f"Unsupported type for image input: {type(img_or_path)}")


def _overlap(a, b):
    return not (a[2] <= b[0] or b[2] <= a[0] or a[3] <= b[1] or b[3] <= a[1])


def _shrink(b):
    x0, y0, x1, y1 = b
    return (x0, y0, int(x1 - (x1 - x0) * 0.1), int(y1 - (y1 - y0) * 0.1))


def resolve_intersections(boxes):
    """Shrink both members of every intersecting pair by 10 % (right/bottom edge, int truncation),
    at most 50 sweeps in (i, j>i) order."""
    out = list(boxes)
    n = len(out)
    for _ in range(50):
        dirty = False
        for i in range(n):
            for j in range(i + 1, n):
                if _overlap(out[i], out[j]):
                    out[i], out[j] = _shrink(out[i]), _shrink(out[j])
                    dirty = True
        if not dirty:
            break
    return out


def sort_boxes_reading_order(boxes, y_tol_ratio=0.6, x_gap_ratio=np.inf):
    if not boxes:
        return []
    avg_h = np.mean([b[3] - b[1] for b in boxes])
    lines = []
    for b in sorted(boxes, key=lambda b: (b[1] + b[3]) / 2):
        cy = (b[1] + b[3]) / 2
        home = None
        for ln in lines:  # first line that accepts the box
            if abs(cy - np.mean([(v[1] + v[3]) / 2 for v in ln])) <= avg_h * y_tol_ratio and \
                    (b[0] - max(v[2] for v in ln)) <= avg_h * x_gap_ratio:
                home = ln
                break
        if home is None:
            lines.append([b])
        else:
            home.append(b)
    lines.sort(key=lambda ln: np.mean([(b[1] + b[3]) / 2 for b in ln]))
    return [b for ln in lines for b in sorted(ln, key=lambda b: b[0])]


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
