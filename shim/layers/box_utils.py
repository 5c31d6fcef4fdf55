"""layers/box_utils.py -> the helpers eval.py's metric code imports (eval.py:5) plus crop / sanitize_coordinates."""
from yolact_amd.layers.box_utils import (center_size, crop, intersect, jaccard, mask_bits, mask_iou, mask_iou_bits,   # noqa: F401
                                         point_form, sanitize_coordinates)
