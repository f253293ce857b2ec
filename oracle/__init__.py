"""oracle/ — CPU restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY.

Nothing in the shipped package (manuscript_ocr_amd/) imports this directory.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
it, and there only as the checker / the reported CPU baseline.

Pinning status (see DESIGN.md "Oracle"):
  * lanms / decode / expand / reading-order / Pipeline glue / TRBA model /
    EAST decoder+head: PINNED against outputs of the reference's own source
    files run in the build container (tests/golden/gen_golden.py, fixtures in
    tests/golden/*.npz) and against every known-answer case of the reference's
    tests/detectors/east/test_lanms.py.
  * EAST ResNet-50 backbone: the arithmetic lives in torchvision
    (requirements.txt pins >=0.12,<0.23), which is absent here and not vendored
    by the reference -> restated from the published ResNet v1.5 definition,
    "parity unpinned" for the backbone alone.
  * cv2.resize / cv2.pointPolygonTest / albumentations.Normalize: cv2 absent,
    restated from OpenCV's published fixed-point algorithms, "parity unpinned".
"""
