"""Drop-in for the reference's `layers` package as eval.py uses it (eval.py:5,8): layers.box_utils,
layers.output_utils, layers.functions.Detect, layers.interpolate.  (layers.modules = MultiBoxLoss is training-only.)"""
from .functions import *                                                # noqa: F401,F403
