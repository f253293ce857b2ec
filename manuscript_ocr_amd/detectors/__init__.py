"""Detector plugins.  Public names match `manuscript.detectors` of the reference so imports port 1:1."""
from ._east import EAST
from ._east import utils as _utils

read_image = _utils.read_image
visualize_page = _utils.visualize_page
sort_boxes_reading_order = _utils.sort_boxes_reading_order
sort_boxes_reading_order_with_resolutions = _utils.sort_boxes_reading_order_with_resolutions

__all__ = ["EAST", "visualize_page", "read_image", "sort_boxes_reading_order", "sort_boxes_reading_order_with_resolutions"]
