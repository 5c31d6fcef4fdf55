"""yolact_amd — MI355X (gfx950) native YOLACT inference hot path behind the reference's Python API.

    from yolact_amd import Yolact, Detect, postprocess, set_cfg
"""
from .config import cfg, set_cfg, active_cfg, CONFIGS  # noqa: F401


def __getattr__(name):   # lazy: importing the package must not require torch.cuda or the built library
    if name == 'Yolact':
        from .yolact import Yolact
        return Yolact
    if name == 'Detect':
        from .layers.detection import Detect
        return Detect
    if name == 'postprocess':
        from .layers.output_utils import postprocess
        return postprocess
    raise AttributeError(name)
