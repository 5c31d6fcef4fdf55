"""Drop-in for the reference's top-level `yolact` module (yolact.py): `from yolact import Yolact` (eval.py:2).

Put this directory AHEAD of the reference checkout on sys.path (PYTHONPATH=<repo>/shim:<repo>:<reference>): eval.py,
data/, utils/ stay the reference's own files; `yolact`, `layers.*` and `dcn_v2` resolve here and forward to the
MI355X engine.  Re-exports only — no logic lives in shim/."""
from yolact_amd.yolact import Yolact                                    # noqa: F401
from yolact_amd.modules import FPN, FastMaskIoUNet, PredictionModule    # noqa: F401  (names yolact.py defines)
