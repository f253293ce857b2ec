"""Oracle Pipeline-glue restatement vs goldens from the reference's utils.py."""
import json
import os

import numpy as np

from oracle import pipeline_glue as G


def test_glue_golden(golden_dir):
    cases = json.load(open(os.path.join(golden_dir, "pipeline_glue.json")))
    assert len(cases) >= 9
    for c in cases:
        boxes = [tuple(np.int32(v) for v in b) for b in c["boxes"]]
        assert [list(map(int, b)) for b in G.resolve_intersections(boxes)] == c["resolved"]
        assert [list(map(int, b)) for b in G.sort_boxes_reading_order(boxes)] == c["sorted"]
        assert [list(map(int, b)) for b in G.sort_boxes_reading_order_with_resolutions(boxes)] == c["sorted_res"]


def test_order_and_crop():
    img = np.arange(100 * 400 * 3, dtype=np.uint8).reshape(100, 400, 3)
    polys = [[[110.0, 10.0], [200.0, 10.0], [200.0, 50.0], [110.0, 50.0]],
             [[10.9, 10.2], [100.7, 10.0], [100.0, 50.0], [10.0, 50.9]],
             [[10.0, 60.0], [12.0, 60.0], [12.0, 62.0], [10.0, 62.0]]]
    order, kept, crops = G.order_and_crop(polys, img, 5)
    assert order == [1, 0, 2] and kept == [0, 1]
    assert crops[0].shape == (40, 90, 3) and crops[0].base is not None  # a view, truncated coords
    assert np.array_equal(crops[0], img[10:50, 10:100])
