"""The data-format side of the hot path: what `from data import ...` gives eval.py that the build
owns — `COCODetection` (pull_item), `COCOAnnotationTransform`, `get_label_map`, `MEANS`, `STD`.  `cfg` / `set_cfg` live in
yolact_amd.config."""
from ..coco import get_label_map                                            # noqa: F401
from ..utils.augmentations import MEANS, STD                                 # noqa: F401
from .coco import COCOAnnotationTransform, COCODetection, COCOIndex, ann_to_mask   # noqa: F401
from .jpeg import imread                                                     # noqa: F401
