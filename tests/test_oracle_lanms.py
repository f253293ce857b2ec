"""Oracle (oracle/lanms.c) vs the reference's known-answer tests and golden vectors.

The known-answer cases restate /root/reference/tests/detectors/east/test_lanms.py:18-188
(values only); lanms.npz / east_post.npz hold outputs of the reference's own lanms.py.
"""
import os

import numpy as np
import pytest

from oracle import lanms as L

SQ4 = np.array([[0, 0], [4, 0], [4, 4], [0, 4]], dtype=np.float64)
UNIT = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], dtype=np.float64)


def test_area_square_triangle_degenerate():
    assert L.polygon_area(UNIT) == pytest.approx(1.0, rel=1e-12)
    assert L.polygon_area(np.array([[0, 0], [2, 0], [0, 2]], dtype=np.float64)) == pytest.approx(2.0)
    assert L.polygon_area(np.array([[0, 0], [1, 0]], dtype=np.float64)) == pytest.approx(0.0)


def test_compute_intersection():
    r = L.compute_intersection([0, 0], [2, 2], [0, 2], [2, 0])
    np.testing.assert_allclose(r, [1, 1], rtol=1e-12)
    r = L.compute_intersection([0, 0], [1, 1], [2, 2], [3, 3])  # parallel -> p1
    np.testing.assert_allclose(r, [0, 0])


def test_clip_polygon_cases():
    c, n = L.clip_polygon(SQ4, [2, 5], [2, -1])
    assert n == 4
    np.testing.assert_allclose(c, [[2, 0], [4, 0], [4, 4], [2, 4]])
    c, n = L.clip_polygon(UNIT, [100, 0], [100, 1])
    assert n == 4
    np.testing.assert_allclose(c, UNIT)
    c, n = L.clip_polygon(UNIT + 1, [0, 0], [0, 1])
    assert n == 0 and c.shape == (0, 2)


def test_intersection_iou_merge():
    inter = L.polygon_intersection(SQ4, SQ4 + 2)
    np.testing.assert_allclose(inter, [[2, 2], [4, 2], [4, 4], [2, 4]])
    assert L.polygon_intersection(UNIT, UNIT + 2).shape == (0, 2)
    assert L.polygon_iou(SQ4, SQ4 + 2) == pytest.approx(4 / 28, rel=1e-12)
    assert L.polygon_iou(UNIT, UNIT) == pytest.approx(1.0)
    assert L.polygon_iou(UNIT, UNIT + 2) == pytest.approx(0.0)
    assert L.should_merge(SQ4, SQ4 + 2, 0.1) and not L.should_merge(SQ4, SQ4 + 2, 0.2)
    assert not L.should_merge(UNIT, UNIT, 1.0) and L.should_merge(UNIT, UNIT, 0.999)


def test_normalize_polygon_all_variants():
    np.testing.assert_allclose(L.normalize_polygon(SQ4, np.array([[4, 4], [0, 4], [0, 0], [4, 0]], dtype=np.float64)), SQ4)
    for start in range(4):
        for var in (np.vstack([UNIT[(i + start) % 4] for i in range(4)]),
                    np.vstack([UNIT[(start - i) % 4] for i in range(4)])):
            np.testing.assert_allclose(L.normalize_polygon(UNIT, var), UNIT)


def test_standard_and_locality_nms_counts():
    polys = [SQ4, SQ4 + 1, SQ4 + 10]
    kp, ks = L.standard_nms(polys, [0.9, 0.8, 0.7], 0.1)
    assert len(kp) == 2
    boxes = np.array([
        [0, 0, 4, 0, 4, 4, 0, 4, 0.9], [1, 1, 5, 1, 5, 5, 1, 5, 0.8],
        [10, 10, 14, 10, 14, 14, 10, 14, 0.7], [11, 11, 15, 11, 15, 15, 11, 15, 0.6]], dtype=np.float32)
    assert L.locality_aware_nms(boxes, 0.1).shape == (2, 9)
    assert L.locality_aware_nms(np.zeros((0, 9), np.float32), 0.5).shape == (0, 9)


@pytest.mark.parametrize("name", ["page_small", "page_mid", "rot_1", "rot_50", "rot_600", "rot_rev_40"])
def test_lanms_golden_bit_exact(golden_dir, name):
    g = np.load(os.path.join(golden_dir, "lanms.npz"))
    out = L.locality_aware_nms(g[f"{name}_in"], 0.2)
    exp = g[f"{name}_out"]
    assert out.shape == exp.shape
    assert np.array_equal(out.view(np.uint32), exp.view(np.uint32))


def test_lanms_golden_from_post(golden_dir):
    g = np.load(os.path.join(golden_dir, "east_post.npz"))
    out = L.locality_aware_nms(g["decoded_q2"], 0.2)
    assert np.array_equal(out.view(np.uint32), g["lanms_q2"].view(np.uint32))
