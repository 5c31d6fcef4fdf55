"""layers/functions/detection.py -> the device Detect."""
from yolact_amd.layers.detection import Detect                          # noqa: F401
