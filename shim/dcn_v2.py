"""Drop-in for external/DCNv2's `dcn_v2` module: backbone.py:7-11 does `from dcn_v2 import DCN`.
The parameter container keeps the reference's names (weight, bias, conv_offset_mask.{weight,bias}; dcn_v2.py:97-116) so
YOLACT++ checkpoints load; the arithmetic is `ymi_dcn_v2_forward_f32` inside the engine's plan."""
from yolact_amd.modules import DCN                                      # noqa: F401
