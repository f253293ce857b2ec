"""manuscript_ocr_amd — MI355X-native hot path of manuscript-ocr (EAST detector + TRBA recogniser)
behind the reference's plugin API: `from manuscript_ocr_amd import Pipeline` replaces
`from manuscript import Pipeline` (/root/reference/src/manuscript/__init__.py:1-4)."""
from ._pipeline import Pipeline
from .detectors import read_image, visualize_page

__all__ = ["Pipeline", "visualize_page", "read_image"]
