"""Restatement of the Pipeline glue between detector and recogniser (TEST INFRASTRUCTURE).

Reference:
  resolve_intersections                      detectors/_east/utils.py:500-547
  sort_boxes_reading_order                   detectors/_east/utils.py:550-607
  sort_boxes_reading_order_with_resolutions  detectors/_east/utils.py:610-644
  Pipeline.predict sort+crop section         _pipeline.py:100-137
  Pipeline._extract_word_image               _pipeline.py:204-221
Pinned by tests/golden/pipeline_glue.json (generated from the reference file).
"""
import numpy as np


def _intersect(a, b):
    return not (a[2] <= b[0] or b[2] <= a[0] or a[3] <= b[1] or b[3] <= a[1])


def resolve_intersections(boxes):
    res = list(boxes)
    for _ in range(50):
        changed = False
        for i in range(len(res)):
            for j in range(i + 1, len(res)):
                if _intersect(res[i], res[j]):
                    x0, y0, x1, y1 = res[i]
                    u0, v0, u1, v1 = res[j]
                    res[i] = (x0, y0, int(x1 - (x1 - x0) * 0.1), int(y1 - (y1 - y0) * 0.1))
                    res[j] = (u0, v0, int(u1 - (u1 - u0) * 0.1), int(v1 - (v1 - v0) * 0.1))
                    changed = True
        if not changed:
            break
    return res


def sort_boxes_reading_order(boxes, y_tol_ratio=0.6, x_gap_ratio=np.inf):
    if not boxes:
        return []
    avg_h = np.mean([b[3] - b[1] for b in boxes])
    lines = []
    for b in sorted(boxes, key=lambda b: (b[1] + b[3]) / 2):
        cy = (b[1] + b[3]) / 2
        for ln in lines:
            line_cy = np.mean([(v[1] + v[3]) / 2 for v in ln])
            last_x1 = max(v[2] for v in ln)
            if abs(cy - line_cy) <= avg_h * y_tol_ratio and (b[0] - last_x1) <= avg_h * x_gap_ratio:
                ln.append(b)
                break
        else:
            lines.append([b])
    lines.sort(key=lambda ln: np.mean([(b[1] + b[3]) / 2 for b in ln]))
    for ln in lines:
        ln.sort(key=lambda b: b[0])
    return [b for ln in lines for b in ln]


def sort_boxes_reading_order_with_resolutions(boxes, y_tol_ratio=0.6, x_gap_ratio=np.inf):
    compressed = resolve_intersections(boxes)
    mapping = {c: o for c, o in zip(compressed, boxes)}  # identical shrunk boxes collapse, later wins
    return [mapping[b] for b in sort_boxes_reading_order(compressed, y_tol_ratio, x_gap_ratio)]


def word_box(polygon):
    poly = np.array(polygon, dtype=np.int32)  # float -> int32 truncation toward zero
    x0, y0 = np.min(poly, axis=0)
    x1, y1 = np.max(poly, axis=0)
    return (x0, y0, x1, y1)


def order_and_crop(polygons, image, min_text_size=5):
    """_pipeline.py:102-137 for one block.

    polygons: list of 4x2 float lists.  Returns (order, kept, crops): `order` =
    indices of polygons in the reordered block (first-equal-word matching, so
    duplicates can repeat/drop exactly as the reference does), `kept` = positions
    in `order` that produced a crop, `crops` = clamped AABB views of `image`."""
    boxes = [word_box(p) for p in polygons]
    sorted_boxes = sort_boxes_reading_order_with_resolutions(boxes)
    order = []
    for bx in sorted_boxes:
        for wi, wb in enumerate(boxes):
            if wb == bx:
                order.append(wi)
                break
    kept, crops = [], []
    H, W = image.shape[:2]
    for pos, wi in enumerate(order):
        x0, y0, x1, y1 = boxes[wi]
        if (x1 - x0) >= min_text_size and (y1 - y0) >= min_text_size:
            a, b = max(0, int(x0)), max(0, int(y0))
            c, d = min(W, int(x1)), min(H, int(y1))
            region = image[b:d, a:c]
            if region.size > 0:
                kept.append(pos)
                crops.append(region)
    return order, kept, crops
