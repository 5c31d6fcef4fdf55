"""layers/interpolate.py: utils/functions.py:7 imports InterpolateModule at module import time."""
from yolact_amd.modules import InterpolateModule                        # noqa: F401
