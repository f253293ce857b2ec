"""Host-side tail of EAST.predict after the device NMS (small M, NumPy f32).

Mirrors /root/reference/src/manuscript/detectors/_east/utils.py:384-422 (expand_boxes) and
infer.py:134-233 (_scale_boxes_to_original, _convert_to_axis_aligned, _polygon_area_batch,
_is_quad_inside, _remove_fully_contained_boxes, _remove_area_anomalies).  The quad-in-quad
test is vectorised over all (inner, outer) pairs instead of per-point cv2.pointPolygonTest
calls; it evaluates OpenCV's even-odd crossing rule with the same float/double operation order.
Device versions of these filters are a "next" item.
"""
import numpy as np


def expand_boxes(quads, expand_w=0.0, expand_h=0.0):
    if len(quads) == 0 or (expand_w == 0 and expand_h == 0):
        return quads
    pts = quads[:, :8].reshape(-1, 4, 2)
    x, y = pts[:, :, 0], pts[:, :, 1]
    area = np.sum(x * np.roll(y, -1, axis=1) - np.roll(x, -1, axis=1) * y, axis=1)
    sign = np.sign(area).reshape(-1, 1, 1)
    sign[sign == 0] = 1
    prev_pt, next_pt = np.roll(pts, 1, axis=1), np.roll(pts, -1, axis=1)
    e_in, e_out = pts - prev_pt, next_pt - pts
    l_in = np.linalg.norm(e_in, axis=2, keepdims=True)
    l_out = np.linalg.norm(e_out, axis=2, keepdims=True)
    n_in = sign * np.stack([e_in[..., 1], -e_in[..., 0]], axis=2) / (l_in + 1e-6)
    n_out = sign * np.stack([e_out[..., 1], -e_out[..., 0]], axis=2) / (l_out + 1e-6)
    bis = n_in + n_out
    nrm = np.linalg.norm(bis, axis=2, keepdims=True)
    bis = np.divide(bis, nrm, out=np.zeros_like(bis), where=nrm > 0)
    reach = np.minimum(l_in, l_out)
    k = np.array([1 + expand_w, 1 + expand_h], dtype=np.float32).reshape(1, 1, 2)
    moved = pts + ((k - 1.0) * reach) * bis
    return np.hstack([moved.reshape(-1, 8), quads[:, 8:9]]).astype(np.float32)


def scale_boxes(boxes, orig_hw, target_wh):
    if len(boxes) == 0:
        return boxes
    out = boxes.copy()
    out[:, 0:8:2] *= orig_hw[1] / target_wh[0]
    out[:, 1:8:2] *= orig_hw[0] / target_wh[1]
    return out


def to_axis_aligned(quads):
    if len(quads) == 0:
        return quads
    out = quads.copy()
    c = out[:, :8].reshape(-1, 4, 2)
    x0, x1, y0, y1 = c[:, :, 0].min(1), c[:, :, 0].max(1), c[:, :, 1].min(1), c[:, :, 1].max(1)
    out[:, :8] = np.stack([x0, y0, x1, y0, x1, y1, x0, y1], axis=1)
    return out


def quad_areas(pts):
    if pts.size == 0:
        return np.zeros((0,), dtype=np.float32)
    x, y = pts[:, :, 0], pts[:, :, 1]
    return 0.5 * np.abs(np.sum(x * np.roll(y, -1, axis=1) - y * np.roll(x, -1, axis=1), axis=1))


def _points_not_outside(px, py, contour):
    """px, py: (P,) f32 points; contour (4,2) f32.  True where pointPolygonTest(...) >= 0."""
    n = len(contour)
    counter = np.zeros(px.shape, dtype=np.int64)
    on_edge = np.zeros(px.shape, dtype=bool)
    done = np.zeros(px.shape, dtype=bool)  # OpenCV returns at the first on-edge hit
    v = contour[n - 1]
    for i in range(n):
        v0, v = v, contour[i]
        skip = ((v0[1] <= py) & (v[1] <= py)) | ((v0[1] > py) & (v[1] > py)) | ((v0[0] < px) & (v[0] < px))
        hit = skip & (py == v[1]) & ((px == v[0]) | ((py == v0[1]) & (((v0[0] <= px) & (px <= v[0])) | ((v[0] <= px) & (px <= v0[0])))))
        dist = (py - v0[1]).astype(np.float64) * np.float64(v[0] - v0[0]) - (px - v0[0]).astype(np.float64) * np.float64(v[1] - v0[1])
        zero = ~skip & (dist == 0)
        new_edge = (hit | zero) & ~done
        on_edge |= new_edge
        done |= new_edge
        if v[1] < v0[1]:
            dist = -dist
        counter += (~skip & ~done & (dist > 0)).astype(np.int64)
    return on_edge | (counter % 2 == 1)


def remove_contained(quads):
    """Drop quads whose 4 vertices all lie inside/on a kept quad of area >= own - 1e-6 (ascending area)."""
    m = len(quads)
    if m <= 1:
        return quads
    pts = quads[:, :8].reshape(-1, 4, 2).astype(np.float32)
    areas = quad_areas(quads[:, :8].reshape(-1, 4, 2))
    # inside[i, j]: all vertices of i are not outside j  (M x M, M is a few hundred)
    inside = np.zeros((m, m), dtype=bool)
    px, py = pts[:, :, 0].reshape(-1), pts[:, :, 1].reshape(-1)
    # cheap bounding-box prefilter: a vertex outside j's bbox is outside j
    bx0, bx1 = pts[:, :, 0].min(1), pts[:, :, 0].max(1)
    by0, by1 = pts[:, :, 1].min(1), pts[:, :, 1].max(1)
    cand = (bx0[:, None] >= bx0[None, :]) & (bx1[:, None] <= bx1[None, :]) & (by0[:, None] >= by0[None, :]) & (by1[:, None] <= by1[None, :])
    for j in range(m):
        idx = np.nonzero(cand[:, j])[0]
        if len(idx) == 0:
            continue
        sel = (idx[:, None] * 4 + np.arange(4)[None, :]).reshape(-1)
        ok = _points_not_outside(px[sel], py[sel], pts[j]).reshape(-1, 4).all(axis=1)
        inside[idx[ok], j] = True
    keep = np.ones(m, dtype=bool)
    np.fill_diagonal(inside, False)
    big_enough = (areas[None, :] + 1e-6) >= areas[:, None]  # [i, j]: j may contain i
    inside &= big_enough
    for i in np.argsort(areas, kind="stable"):
        if np.any(inside[i] & keep):
            keep[i] = False
    return quads[keep]


def remove_area_anomalies(quads, enabled=True, sigma=5.0, min_count=30):
    if not enabled or len(quads) == 0 or len(quads) <= min_count:
        return quads
    areas = quad_areas(quads[:, :8].reshape(-1, 4, 2)).astype(np.float32)
    mean, std = float(np.mean(areas)), float(np.std(areas))
    if std == 0.0:
        return quads
    keep = areas <= mean + sigma * std
    return quads[keep] if np.any(keep) else quads
