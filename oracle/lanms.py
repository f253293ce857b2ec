"""ctypes wrapper over oracle/lanms.c (TEST INFRASTRUCTURE; see lanms.c header)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(_HERE, "lanms.c")):
            build()
        L = ctypes.CDLL(so)
        d, i64 = ctypes.c_double, ctypes.c_int64
        P = ctypes.c_void_p
        L.orc_polygon_area.restype = d
        L.orc_polygon_area.argtypes = [P, ctypes.c_int]
        L.orc_compute_intersection.argtypes = [P, P, P, P, P]
        L.orc_clip_polygon.restype = ctypes.c_int
        L.orc_clip_polygon.argtypes = [P, ctypes.c_int, P, P, P]
        L.orc_polygon_intersection.restype = ctypes.c_int
        L.orc_polygon_intersection.argtypes = [P, ctypes.c_int, P, ctypes.c_int, P]
        L.orc_polygon_iou.restype = d
        L.orc_polygon_iou.argtypes = [P, ctypes.c_int, P, ctypes.c_int]
        L.orc_normalize_polygon.argtypes = [P, P, P]
        L.orc_standard_nms.restype = i64
        L.orc_standard_nms.argtypes = [P, P, i64, d, P]
        L.orc_locality_aware_nms.restype = i64
        L.orc_locality_aware_nms.argtypes = [P, i64, d, P, P]
        _LIB = L
    return _LIB


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def polygon_area(poly):
    p = _f64(poly)
    return lib().orc_polygon_area(p.ctypes.data, p.shape[0])


def compute_intersection(p1, p2, A, B):
    p1, p2, A, B = map(_f64, (p1, p2, A, B))
    out = np.empty(2)
    lib().orc_compute_intersection(p1.ctypes.data, p2.ctypes.data, A.ctypes.data, B.ctypes.data, out.ctypes.data)
    return out


def clip_polygon(subject, A, B):
    s, A, B = map(_f64, (subject, A, B))
    out = np.empty((20, 2))
    c = lib().orc_clip_polygon(s.ctypes.data, s.shape[0], A.ctypes.data, B.ctypes.data, out.ctypes.data)
    return out[:c], c


def polygon_intersection(poly1, poly2):
    a, b = _f64(poly1), _f64(poly2)
    out = np.empty((20, 2))
    c = lib().orc_polygon_intersection(a.ctypes.data, a.shape[0], b.ctypes.data, b.shape[0], out.ctypes.data)
    return out[:c].copy()


def polygon_iou(poly1, poly2):
    a, b = _f64(poly1), _f64(poly2)
    return lib().orc_polygon_iou(a.ctypes.data, a.shape[0], b.ctypes.data, b.shape[0])


def should_merge(poly1, poly2, thr):
    return polygon_iou(poly1, poly2) > thr


def normalize_polygon(ref, poly):
    r, p = _f64(ref), _f64(poly)
    out = np.empty((4, 2))
    lib().orc_normalize_polygon(r.ctypes.data, p.ctypes.data, out.ctypes.data)
    return out


def standard_nms(polys, scores, thr):
    p = _f64(polys).reshape(-1, 4, 2)
    s = _f64(scores)
    if p.size == 0:
        return p, s
    keep = np.empty(len(s), dtype=np.int64)
    k = lib().orc_standard_nms(p.ctypes.data, s.ctypes.data, len(s), float(thr), keep.ctypes.data)
    keep = keep[:k]
    return p[keep], s[keep]


def locality_aware_nms(boxes, iou_threshold, return_merged_count=False):
    if boxes is None or len(boxes) == 0:
        z = np.zeros((0, 9), dtype=np.float32)
        return (z, 0) if return_merged_count else z
    b = np.ascontiguousarray(boxes, dtype=np.float32)
    out = np.empty_like(b)
    nm = ctypes.c_int64(0)
    m = lib().orc_locality_aware_nms(b.ctypes.data, len(b), float(iou_threshold), out.ctypes.data, ctypes.byref(nm))
    res = out[:m].copy()
    return (res, nm.value) if return_merged_count else res
