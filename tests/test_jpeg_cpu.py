"""Image ingest: the JPEG decoder of libmsocr.so against PIL (libjpeg-turbo defaults = what the reference's
read_image produces through cv2.imread / PIL, detectors/_east/utils.py:477-497) — bit for bit, on the CPU, through the host
entropy decoder and the HOST twin of the device reconstruction stage (the same __host__ __device__ code the kernels run)."""
import io

import numpy as np
import pytest
from PIL import Image

from manuscript_ocr_amd import ingest, synth


def _pil_decode(data):
    with Image.open(io.BytesIO(data)) as im:
        return np.array(im.convert("RGB"))


def _encode(arr, **kw):
    b = io.BytesIO()
    Image.fromarray(arr).save(b, format="JPEG", **kw)
    return b.getvalue()


def _test_images():
    rng = np.random.default_rng(5)
    page = synth.synth_page(3, 203, 317)[0]                      # odd sizes: partial MCUs on both axes
    noise = rng.integers(0, 256, size=(64, 80, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:97, 0:131]
    smooth = np.stack([(xx * 2) % 256, (yy * 3) % 256, (xx + yy) % 256], axis=2).astype(np.uint8)
    tiny = rng.integers(0, 256, size=(3, 5, 3), dtype=np.uint8)  # downsampled width <= 2 would need w <= 4: see `narrow`
    narrow = rng.integers(0, 256, size=(40, 3, 3), dtype=np.uint8)
    return {"page": page, "noise": noise, "smooth": smooth, "tiny": tiny, "narrow": narrow}


@pytest.mark.parametrize("subsampling", [0, 1, 2])
@pytest.mark.parametrize("quality", [30, 75, 95, 100])
def test_jpeg_decode_equals_pil(subsampling, quality):
    for name, arr in _test_images().items():
        data = _encode(arr, quality=quality, subsampling=subsampling)
        got = ingest.decode_jpeg_host(data)
        assert got is not None, (name, "unsupported")
        exp = _pil_decode(data)
        assert got.shape == exp.shape and np.array_equal(got, exp), (name, subsampling, quality,
                                                                    int(np.abs(got.astype(int) - exp.astype(int)).max()))


def test_jpeg_grayscale_restart_and_optimized_tables():
    arr = _test_images()["page"]
    gray = np.array(Image.fromarray(arr).convert("L"))
    b = io.BytesIO()
    Image.fromarray(gray).save(b, format="JPEG", quality=85)
    assert np.array_equal(ingest.decode_jpeg_host(b.getvalue()), _pil_decode(b.getvalue()))
    for kw in ({"optimize": True}, {"restart_marker_blocks": 3}, {"restart_marker_rows": 1}):
        try:
            data = _encode(arr, quality=80, subsampling=2, **kw)
        except TypeError:
            continue
        assert np.array_equal(ingest.decode_jpeg_host(data), _pil_decode(data)), kw


def test_jpeg_unsupported_streams_are_reported():
    arr = _test_images()["smooth"]
    assert ingest.decode_jpeg_host(_encode(arr, quality=80, progressive=True)) is None      # SOF2: host decoder
    assert ingest.decode_jpeg_host(b"not a jpeg") is None
    data = _encode(arr, quality=80)
    assert ingest.decode_jpeg_host(data[: len(data) // 2]) is not None or True               # truncated: must not crash
    cmyk = io.BytesIO()
    Image.fromarray(arr).convert("CMYK").save(cmyk, format="JPEG")
    assert ingest.decode_jpeg_host(cmyk.getvalue()) is None


def test_jpeg_restart_intervals_every_subsampling_and_awkward_sizes():
    """Restart markers with each chroma layout, widths / heights around the MCU sizes (8, 16) incl. 1-pixel images."""
    rng = np.random.default_rng(9)
    for (h, w) in ((1, 1), (7, 9), (8, 16), (15, 17), (16, 33), (31, 47), (64, 3), (2, 130)):
        arr = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        for sub in (0, 1, 2):
            for kw in ({}, {"restart_marker_blocks": 1}, {"restart_marker_rows": 1}):
                try:
                    data = _encode(arr, quality=60, subsampling=sub, **kw)
                except TypeError:
                    continue
                got = ingest.decode_jpeg_host(data)
                assert got is not None and np.array_equal(got, _pil_decode(data)), (h, w, sub, kw)
