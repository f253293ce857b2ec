"""NumPy restatement of the EAST post-processing (TEST INFRASTRUCTURE).

Reference (all under /root/reference/src/manuscript/detectors/_east/):
  decode_quads_from_maps            utils.py:328-381
  expand_boxes                      utils.py:384-422
  EAST._scale_boxes_to_original     infer.py:134-147
  EAST._convert_to_axis_aligned     infer.py:149-172
  EAST._polygon_area_batch          infer.py:174-182
  EAST._is_quad_inside              infer.py:184-192  (cv2.pointPolygonTest, restated)
  EAST._remove_fully_contained_boxes infer.py:194-214
  EAST._remove_area_anomalies       infer.py:216-233
Pinned by tests/golden/east_post.npz (decode, expand from the reference file);
the cv2.pointPolygonTest restatement is "parity unpinned" (cv2 absent).
"""
import numpy as np


def decode_quads_from_maps(score_map, geo_map, score_thresh, scale, quantization=1):
    """score_map (H,W) f32, geo_map (H,W,8) f32 -> (N,9) f32, rows in (y,x) order."""
    if score_map.ndim == 3 and score_map.shape[0] == 1:
        score_map = score_map.squeeze(0)
    ys, xs = np.where(score_map > score_thresh)  # strict, f32(thr) comparison
    if len(ys) == 0:
        return np.zeros((0, 9), dtype=np.float32)
    if quantization > 1:
        q = quantization
        ysq = (ys // q) * q + q // 2
        xsq = (xs // q) * q + q // 2
        uniq = np.unique(np.column_stack([ysq, xsq]), axis=0)  # lexicographic (y, x)
        ys, xs = uniq[:, 0], uniq[:, 1]
    offs = geo_map[ys, xs].astype(np.float32)  # (N,8)
    out = np.empty((len(ys), 9), dtype=np.float32)
    # utils.py:373-374: int64*float -> f64 ; f32*4.0 exact ; f64 add ; one rounding to f32
    sc = float(scale)
    xs64 = xs.astype(np.float64) * sc
    ys64 = ys.astype(np.float64) * sc
    off_s = (offs * np.float32(sc)).astype(np.float64)
    out[:, 0:8:2] = (xs64[:, None] + off_s[:, 0::2]).astype(np.float32)
    out[:, 1:8:2] = (ys64[:, None] + off_s[:, 1::2]).astype(np.float32)
    out[:, 8] = score_map[ys, xs]
    return out


def expand_boxes(quads, expand_w=0.0, expand_h=0.0):
    """All arithmetic in f32, as the reference's NumPy expression (utils.py:384-422)."""
    if len(quads) == 0 or (expand_w == 0 and expand_h == 0):
        return quads
    coords = quads[:, :8].reshape(-1, 4, 2)
    scores = quads[:, 8:9]
    x, y = coords[:, :, 0], coords[:, :, 1]
    area = np.sum(x * np.roll(y, -1, axis=1) - np.roll(x, -1, axis=1) * y, axis=1)
    sign = np.sign(area).reshape(-1, 1, 1)
    sign[sign == 0] = 1
    p_prev = np.roll(coords, 1, axis=1)
    p_curr = coords
    p_next = np.roll(coords, -1, axis=1)
    edge1 = p_curr - p_prev
    edge2 = p_next - p_curr
    len1 = np.linalg.norm(edge1, axis=2, keepdims=True)
    len2 = np.linalg.norm(edge2, axis=2, keepdims=True)
    n1 = sign * np.stack([edge1[..., 1], -edge1[..., 0]], axis=2) / (len1 + 1e-6)
    n2 = sign * np.stack([edge2[..., 1], -edge2[..., 0]], axis=2) / (len2 + 1e-6)
    n_avg = n1 + n2
    norm = np.linalg.norm(n_avg, axis=2, keepdims=True)
    n_avg = np.divide(n_avg, norm, out=np.zeros_like(n_avg), where=norm > 0)
    offset = np.minimum(len1, len2)
    scale_xy = np.array([1 + expand_w, 1 + expand_h], dtype=np.float32).reshape(1, 1, 2)
    delta = (scale_xy - 1.0) * offset
    new_coords = p_curr + delta * n_avg
    return np.hstack([new_coords.reshape(-1, 8), scores]).astype(np.float32)


def scale_boxes_to_original(boxes, orig_size, target_size):
    """target_size: int (square, reference) or (W, H)."""
    if len(boxes) == 0:
        return boxes
    orig_h, orig_w = orig_size
    tw, th = (target_size, target_size) if np.isscalar(target_size) else target_size
    scaled = boxes.copy()
    scaled[:, 0:8:2] *= orig_w / tw
    scaled[:, 1:8:2] *= orig_h / th
    return scaled


def convert_to_axis_aligned(quads):
    if len(quads) == 0:
        return quads
    aligned = quads.copy()
    c = aligned[:, :8].reshape(-1, 4, 2)
    x0, x1 = c[:, :, 0].min(axis=1), c[:, :, 0].max(axis=1)
    y0, y1 = c[:, :, 1].min(axis=1), c[:, :, 1].max(axis=1)
    aligned[:, :8] = np.stack([x0, y0, x1, y0, x1, y1, x0, y1], axis=1)
    return aligned


def polygon_area_batch(polys):
    if polys.size == 0:
        return np.zeros((0,), dtype=np.float32)
    x, y = polys[:, :, 0], polys[:, :, 1]
    return 0.5 * np.abs(np.sum(x * np.roll(y, -1, axis=1) - y * np.roll(x, -1, axis=1), axis=1))


def point_polygon_test_sign(contour, pt):
    """Restatement of cv2.pointPolygonTest(contour, pt, measureDist=False) for a
    float32 contour: +1 inside, 0 on an edge/vertex, -1 outside (OpenCV
    geometry.cpp, the non-integer branch: even-odd crossing count with an exact
    on-edge test `dist == 0`, evaluated in double)."""
    n = len(contour)
    f32 = np.float32
    x, y = f32(pt[0]), f32(pt[1])
    counter = 0
    v = (f32(contour[n - 1][0]), f32(contour[n - 1][1]))
    for i in range(n):
        v0 = v
        v = (f32(contour[i][0]), f32(contour[i][1]))
        if (v0[1] <= y and v[1] <= y) or (v0[1] > y and v[1] > y) or (v0[0] < x and v[0] < x):
            if y == v[1] and (x == v[0] or (y == v0[1] and ((v0[0] <= x <= v[0]) or (v[0] <= x <= v0[0])))):
                return 0
            continue
        # OpenCV: (double)(pt.y - v0.y)*(v.x - v0.x) - (double)(pt.x - v0.x)*(v.y - v0.y)
        # -> differences in float, products and subtraction in double
        dist = float(y - v0[1]) * float(v[0] - v0[0]) - float(x - v0[0]) * float(v[1] - v0[1])
        if dist == 0:
            return 0
        if v[1] < v0[1]:
            dist = -dist
        counter += dist > 0
    return -1 if counter % 2 == 0 else 1


def is_quad_inside(inner, outer):
    contour = outer.reshape(-1, 2).astype(np.float32)
    for p in inner.astype(np.float32):
        if point_polygon_test_sign(contour, p) < 0:
            return False
    return True


def remove_fully_contained_boxes(quads):
    if len(quads) <= 1:
        return quads
    coords = quads[:, :8].reshape(-1, 4, 2)
    areas = polygon_area_batch(coords)
    keep = np.ones(len(quads), dtype=bool)
    order = np.argsort(areas, kind="stable")
    for idx in order:
        if not keep[idx]:
            continue
        inner, inner_area = coords[idx], areas[idx]
        for jdx in range(len(quads)):
            if idx == jdx or not keep[jdx]:
                continue
            if areas[jdx] + 1e-6 < inner_area:
                continue
            if is_quad_inside(inner, coords[jdx]):
                keep[idx] = False
                break
    return quads[keep]


def remove_area_anomalies(quads, enabled=True, sigma=5.0, min_count=30):
    if not enabled or len(quads) == 0 or len(quads) <= min_count:
        return quads
    coords = quads[:, :8].reshape(-1, 4, 2)
    areas = polygon_area_batch(coords).astype(np.float32)
    mean, std = float(np.mean(areas)), float(np.std(areas))
    if std == 0.0:
        return quads
    keep = areas <= mean + sigma * std
    if not np.any(keep):
        return quads
    return quads[keep]


def east_postprocess(score_map, geo_hwc, orig_size, target_size, lanms_fn, score_thresh=0.6,
                     iou_threshold=0.2, quantization=2, scale=4.0, expand_w=0.9, expand_h=0.9,
                     axis_aligned=True, remove_anomalies=True, sigma=5.0, min_count=30):
    """infer.py:319-356 end to end on CPU maps -> (M,9) f32 output quads."""
    q = decode_quads_from_maps(score_map, geo_hwc, score_thresh, scale, quantization)
    q = lanms_fn(q, iou_threshold)
    q = expand_boxes(q, expand_w, expand_h)
    q = scale_boxes_to_original(q, orig_size, target_size)
    q = remove_fully_contained_boxes(q)
    q = remove_area_anomalies(q, remove_anomalies, sigma, min_count)
    return convert_to_axis_aligned(q) if axis_aligned else q
