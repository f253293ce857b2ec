from ._east import EAST
from ._east.utils import (
    read_image,
    sort_boxes_reading_order,
    sort_boxes_reading_order_with_resolutions,
    visualize_page,
)

__all__ = ["EAST", "visualize_page", "read_image", "sort_boxes_reading_order", "sort_boxes_reading_order_with_resolutions"]
