"""layers/box_utils.py -> the helpers eval.py's metric code imports (eval.py:5) plus crop / sanitize_coordinates."""
from yolact_amd.layers.box_utils import (center_size, crop, intersect, jaccard, mask_iou, point_form,   # noqa: F401
                                         sanitize_coordinates)
