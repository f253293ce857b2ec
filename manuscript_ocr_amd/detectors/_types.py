"""Result objects exchanged across the plugin boundary.

Field names, optionality and the [0, 1] range checks are those of the reference DTOs
(/root/reference/src/manuscript/detectors/_types.py:5-33) so that code written against
`manuscript.detectors._types` keeps working; everything else here is this package's own.
"""
from typing import List, Optional, Tuple

from pydantic import BaseModel, Field

_UNIT = dict(ge=0.0, le=1.0)


class Word(BaseModel):
    """One detected word: outline in page pixels, detector score, and (after recognition) its transcription."""

    polygon: List[Tuple[float, float]] = Field(..., description="(x, y) vertices of the word outline, page pixel coordinates")
    detection_confidence: float = Field(..., description="EAST score of the kept quad", **_UNIT)
    text: Optional[str] = Field(None, description="transcription written by Pipeline; None until recognised")
    recognition_confidence: Optional[float] = Field(None, description="mean per-step token probability from the recogniser", **_UNIT)


class Block(BaseModel):
    """A group of words; the detector emits exactly one block per page."""

    words: List[Word]


class Page(BaseModel):
    """All blocks of one page image."""

    blocks: List[Block]
