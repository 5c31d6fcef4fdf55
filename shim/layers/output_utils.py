"""layers/output_utils.py -> device postprocess (eval.py:8)."""
from yolact_amd.layers.output_utils import postprocess, undo_image_transformation   # noqa: F401
