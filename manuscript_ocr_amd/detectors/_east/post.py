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


def _pairs_not_outside(inner, outer):
    """inner, outer: (P,4,2) f32.  -> (P,4) bool: vertex k of inner[p] is inside or on the boundary of outer[p],
    i.e. cv2.pointPolygonTest(outer, pt, False) >= 0 (OpenCV's even-odd crossing rule, differences in f32,
    products in f64, early return 0 on an exact edge hit)."""
    px, py = inner[:, :, 0], inner[:, :, 1]  # (P,4)
    counter = np.zeros(px.shape, dtype=np.int64)
    on_edge = np.zeros(px.shape, dtype=bool)
    for i in range(4):
        v0 = outer[:, (i - 1) % 4][:, None, :]  # (P,1,2)
        v = outer[:, i][:, None, :]
        v0x, v0y, vx, vy = v0[..., 0], v0[..., 1], v[..., 0], v[..., 1]
        skip = ((v0y <= py) & (vy <= py)) | ((v0y > py) & (vy > py)) | ((v0x < px) & (vx < px))
        hit = skip & (py == vy) & ((px == vx) | ((py == v0y) & (((v0x <= px) & (px <= vx)) | ((vx <= px) & (px <= v0x)))))
        dist = (py - v0y).astype(np.float64) * (vx - v0x).astype(np.float64) - (px - v0x).astype(np.float64) * (vy - v0y).astype(np.float64)
        zero = ~skip & (dist == 0)
        live = ~on_edge  # OpenCV returns at the first on-edge hit
        on_edge |= (hit | zero) & live
        dist = np.where(vy < v0y, -dist, dist)
        counter += (~skip & live & ~zero & (dist > 0)).astype(np.int64)
    return on_edge | (counter % 2 == 1)


def remove_contained(quads):
    """Drop quads whose 4 vertices all lie inside/on a kept quad of area >= own - 1e-6 (ascending area)."""
    m = len(quads)
    if m <= 1:
        return quads
    pts = quads[:, :8].reshape(-1, 4, 2).astype(np.float32)
    areas = quad_areas(quads[:, :8].reshape(-1, 4, 2))
    # inside[i, j]: all vertices of i are not outside j.  Bounding-box prefilter first (a vertex outside j's
    # bbox is outside j), then the surviving (i, j) pairs in one vectorised even-odd evaluation.
    inside = np.zeros((m, m), dtype=bool)
    bx0, bx1 = pts[:, :, 0].min(1), pts[:, :, 0].max(1)
    by0, by1 = pts[:, :, 1].min(1), pts[:, :, 1].max(1)
    cand = (bx0[:, None] >= bx0[None, :]) & (bx1[:, None] <= bx1[None, :]) & (by0[:, None] >= by0[None, :]) & (by1[:, None] <= by1[None, :])
    np.fill_diagonal(cand, False)
    ii, jj = np.nonzero(cand)
    if len(ii):
        inside[ii, jj] = _pairs_not_outside(pts[ii], pts[jj]).all(axis=1)
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
