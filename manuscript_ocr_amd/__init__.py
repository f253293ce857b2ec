"""manuscript_ocr_amd — MI355X-native hot path of manuscript-ocr (EAST detector + TRBA recogniser)
behind the reference's plugin API: `from manuscript_ocr_amd import Pipeline` replaces
`from manuscript import Pipeline` (/root/reference/src/manuscript/__init__.py:1-4)."""
import os as _os

# The engine keeps several independent launch sequences in flight (detector / recogniser streams of two batches).  The HIP runtime
# multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and a hardware queue runs its kernels in order, so with 4
# queues independent streams wait behind each other: measured 45.7 pages/s at 4 queues against 48.1 at 8 on the headline workload,
# 107 against 152 pages/s detector-only.  The runtime reads the variable when it initialises, i.e. at the first
# device call of the process: a value the user has set is respected.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from ._pipeline import Pipeline  # noqa: E402
from .detectors import read_image, visualize_page  # noqa: E402

__all__ = ["Pipeline", "visualize_page", "read_image"]
