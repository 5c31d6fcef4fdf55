"""Drop-in for external/DCNv2's `dcn_v2` module: backbone.py:7-11 does `from dcn_v2 import DCN`, external/DCNv2/test.py:11
`from dcn_v2 import dcn_v2_conv, DCNv2, DCN`.  Same names, argument lists and parameter layout (weight, bias,
conv_offset_mask.{weight,bias}; dcn_v2.py:55-128), so YOLACT++ checkpoints load; inside a Yolact plan the arithmetic is
`ymi_dcn_v2_forward_f32` ops of the engine, called on their own the modules run the same entry point (yolact_amd/dcn_v2.py).
The deformable RoI pooling half of the reference module (dcn_v2.py:131-303) is not used by YOLACT and is not provided."""
from yolact_amd.dcn_v2 import DCN, DCNv2, dcn_v2_conv                   # noqa: F401
