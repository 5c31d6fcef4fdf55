"""ctypes binding of libyolact_amd.so (the C ABI declared in include/yolact_amd.h).

The product path has NO fallback: if the HIP library is missing or a call fails, a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libyolact_amd.so')

ABI_VERSION = 8

ACT_NONE, ACT_RELU, ACT_LEAKY01, ACT_TANH, ACT_SIGMOID = 0, 1, 2, 3, 4
RES_NONE, RES_ADD, RES_BILINEAR = 0, 1, 2
TILE_AUTO, TILE_128x128, TILE_128x64, TILE_64x64, TILE_128x32, TILE_64x128 = 0, 1, 2, 3, 4, 5
TILE_32x32_K4, TILE_64x32_K2, TILE_32x64_K2 = 6, 7, 8
TILE_64x64_S3, TILE_64x64_S4, TILE_64x128_S3, TILE_128x64_S3, TILE_32x32_K4_S4, TILE_64x32_K2_S3, TILE_32x64_K2_S3 = range(9, 16)
TILE_NAMES = {1: '128x128', 2: '128x64', 3: '64x64', 4: '128x32', 5: '64x128', 6: '32x32k4', 7: '64x32k2', 8: '32x64k2',
              9: '64x64s3', 10: '64x64s4', 11: '64x128s3', 12: '128x64s3', 13: '32x32k4s4', 14: '64x32k2s3', 15: '32x64k2s3',
              16: '128x128w8', 17: '256x128w8', 18: '128x256w8', 19: '128x128s3', 20: '128x128w8s3', 21: '256x128w8s3',
              22: '128x128w8s4'}
TILE_128x128_W8, TILE_256x128_W8, TILE_128x256_W8 = 16, 17, 18
TILE_128x128_S3, TILE_128x128_W8_S3, TILE_256x128_W8_S3, TILE_128x128_W8_S4 = 19, 20, 21, 22
TILE_WG_128x256 = 23                # csrc/wgemm.hip (round 6): persistent producer / consumer grouped GEMM, Winograd path only (| TILE_H2, planes)
TILE_X3 = 32                        # tile | TILE_X3: bf16x3 split-precision variant of the same block tile (include/yolact_amd.h)
X3_BASE_TILES = (1, 2, 3, 5, 6, 7, 8, 9, 11, 12, 16, 17, 19, 20, 21, 22)
for _t in X3_BASE_TILES:
    TILE_NAMES[_t | TILE_X3] = TILE_NAMES[_t] + 'x3'
BASIC_TILES = (1, 2, 3, 4, 5)      # available for every loader (stem, DCN)
TILE_H2 = 64                        # tile | TILE_H2: fp16x2 split-precision variant (two fp16 pieces, 3 MFMAs per product)
H2_BASE_TILES = X3_BASE_TILES + (18, 13, 14, 15)   # + 128x256w8: one column block for the Cout = 256 Winograd GEMMs (V read once);
                                                  # + the deeper-pipelined K-split tiles (batch 1: ~23 % of the kernel time sat in 32x32k4)
for _t in H2_BASE_TILES:
    TILE_NAMES[_t | TILE_H2] = TILE_NAMES[_t] + 'h2'
TILE_NAMES[TILE_WG_128x256 | TILE_H2] = 'wg128x256h2'
TILE_DCNP = 128                     # TILE_H2 | TILE_DCNP | DCNP_*: the pipelined DCNv2 gather-GEMM (csrc/dcn.hip); the low bits are ITS tile enum
DCNP_64x128, DCNP_64x128_W8, DCNP_64x64, DCNP_128x128_W8, DCNP_128x64_W8, DCNP_32x128 = 1, 2, 3, 4, 5, 6
DCNP_96x128_W6, DCNP_128x128_W8_R1, DCNP_160x128_W10, DCNP_192x128_W12, DCNP_64x256_W8, DCNP_96x256_W12, DCNP_128x256_W16 = range(7, 14)
DCNP_TILES = {1: 'dcnp64x128', 2: 'dcnp64x128w8', 3: 'dcnp64x64', 4: 'dcnp128x128w8', 5: 'dcnp128x64w8', 6: 'dcnp32x128',
              7: 'dcnp96x128w6', 8: 'dcnp128x128w8r1', 9: 'dcnp160x128w10', 10: 'dcnp192x128w12',
              11: 'dcnp64x256w8', 12: 'dcnp96x256w12', 13: 'dcnp128x256w16',
              14: 'dcnp128x256w8t', 15: 'dcnp128x128w4t', 16: 'dcnp256x128w8t',
              17: 'dcnp128x32w4', 18: 'dcnp256x32w8', 19: 'dcnp64x32w2'}
DCNP_128x32_W4, DCNP_256x32_W8, DCNP_64x32_W2 = 17, 18, 19
DCNP_PLAIN_ONLY = (14, 15, 16, 17, 18, 19)     # 64x64 wave tiles, 32-column tiles: ordinary convolutions only
# the weight-stationary streaming kernel (csrc/wstat.hip): Cout <= 32 (.._x32) / <= 64 (.._x64), ordinary convolutions, no residual
WS_TILES = {20: 'ws128x32w4', 21: 'ws256x32w8', 22: 'ws256x32w4', 23: 'ws512x32w8',
            24: 'ws128x64w4', 25: 'ws256x64w8', 26: 'ws256x64w4', 27: 'ws512x64w8'}
DCNP_WS_128x32_W4, DCNP_WS_512x64_W8 = 20, 27
DCNP_PATCH_C64 = 28                 # csrc/patch.hip: 3x3 / s1 / p1, 64 -> 64, input patch in LDS, filters in registers
PATCH_TILES = {28: 'patch8x16c64'}
# csrc/pcconv.hip (round 6): producer / consumer blocks (4 consumer + 4 producer waves); ordinary convolutions
PC_TILES = {29: 'pc128x128'}
DCNP_PC_128x128 = 29
# csrc/patch2.hip (round 6): 3x3 / s1 / p1, input patch of a 256- / 192-pixel tile in LDS, filters streamed, 128 output channels per block
PATCH2_TILES = {30: 'patch2p256', 31: 'patch2p192'}
DCNP_PATCH2_256, DCNP_PATCH2_192 = 30, 31
for _t, _n in list(DCNP_TILES.items()) + list(WS_TILES.items()) + list(PATCH_TILES.items()) + list(PC_TILES.items()) + list(PATCH2_TILES.items()):
    TILE_NAMES[_t | TILE_H2 | TILE_DCNP] = _n
WINO_PLANES = 1024                  # tune-table flag on a Winograd GEMM tile id: V written as fp16x2 planes (ymi_wino_desc.v_planes)
KSPLIT_TILES = (6, 7, 8, 13, 14, 15, 6 | 32, 7 | 32, 8 | 32, 6 | 64, 7 | 64, 8 | 64, 13 | 64, 14 | 64, 15 | 64)   # different (still deterministic) fp32 summation order than the unsplit tiles


class ConvSeg(C.Structure):
    _fields_ = [('n0', C.c_int32), ('n1', C.c_int32), ('act', C.c_int32), ('row_stride', C.c_int32),
                ('batch_stride', C.c_int64), ('ptr', C.c_void_p)]


class ConvDesc(C.Structure):
    _fields_ = [('x', C.c_void_p), ('w', C.c_void_p), ('scale', C.c_void_p), ('bias', C.c_void_p),
                ('res', C.c_void_p),
                ('B', C.c_int32), ('H', C.c_int32), ('W', C.c_int32), ('Cin', C.c_int32), ('ldx', C.c_int32),
                ('Ho', C.c_int32), ('Wo', C.c_int32), ('Cout', C.c_int32),
                ('kh', C.c_int32), ('kw', C.c_int32), ('stride', C.c_int32), ('pad', C.c_int32),
                ('Kpad', C.c_int32), ('res_mode', C.c_int32), ('res_ld', C.c_int32),
                ('res_H', C.c_int32), ('res_W', C.c_int32), ('res_after_act', C.c_int32),
                ('nseg', C.c_int32), ('tile', C.c_int32), ('cin_alg', C.c_int32), ('split_k', C.c_int32),
                ('seg', ConvSeg * 3), ('w_x3', C.c_void_p), ('split_ws', C.c_void_p), ('cout_alg', C.c_int32),
                ('_pad2', C.c_int32),
                ('w_h2', C.c_void_p), ('scale_h2', C.c_void_p), ('winv_h2', C.c_void_p), ('x_amax', C.c_void_p),
                ('y_amax', C.c_void_p), ('x_amax_mul', C.c_float), ('_pad3', C.c_int32)]


class WinoDesc(C.Structure):
    _fields_ = [('x', C.c_void_p), ('u', C.c_void_p), ('scale', C.c_void_p), ('bias', C.c_void_p), ('y', C.c_void_p),
                ('V', C.c_void_p), ('M', C.c_void_p),
                ('B', C.c_int32), ('H', C.c_int32), ('W', C.c_int32), ('C', C.c_int32), ('Cout', C.c_int32),
                ('act', C.c_int32), ('tile', C.c_int32), ('nseg', C.c_int32), ('m', C.c_int32), ('_pad0', C.c_int32),
                ('seg', ConvSeg * 3), ('u_x3', C.c_void_p), ('cout_alg', C.c_int32), ('v_planes', C.c_int32),
                ('u_h2', C.c_void_p), ('uinv_h2', C.c_void_p), ('x_amax', C.c_void_p), ('y_amax', C.c_void_p),
                ('x_up', C.c_void_p), ('up_relu', C.c_int32), ('_pad4', C.c_int32),
                ('proj_w_h2', C.c_void_p), ('proj_scale_h2', C.c_void_p), ('proj_bias', C.c_void_p),
                ('proj_y', C.c_void_p), ('proj_y_amax', C.c_void_p),
                ('proj_cout', C.c_int32), ('proj_ldy', C.c_int32), ('proj_act', C.c_int32), ('_pad5', C.c_int32)]


class DcnDesc(C.Structure):
    _fields_ = [('conv', ConvDesc), ('offmask', C.c_void_p), ('ldo', C.c_int32), ('mask_is_prob', C.c_int32),
                ('om_layout', C.c_int32), ('_pad1', C.c_int32)]


class DetectDesc(C.Structure):
    _fields_ = [('conf', C.c_void_p), ('loc', C.c_void_p), ('coef', C.c_void_p), ('priors', C.c_void_p),
                ('B', C.c_int32), ('P', C.c_int32), ('C', C.c_int32), ('D', C.c_int32),
                ('conf_is_logits', C.c_int32), ('top_k', C.c_int32), ('max_det', C.c_int32),
                ('conf_thresh', C.c_float), ('nms_thresh', C.c_float), ('cross_class', C.c_int32),
                ('conf_ld', C.c_int32), ('_pad1', C.c_int32),
                ('scores_t', C.c_void_p), ('keep', C.c_void_p), ('num_keep', C.c_void_p),
                ('maxsc', C.c_void_p), ('argmax', C.c_void_p),
                ('cand_score', C.c_void_p), ('cand_prior', C.c_void_p),
                ('out_count', C.c_void_p), ('out_box', C.c_void_p), ('out_score', C.c_void_p),
                ('out_class', C.c_void_p), ('out_coef', C.c_void_p), ('out_prior', C.c_void_p), ('out_rec', C.c_void_p)]


class JpegInfo(C.Structure):
    _fields_ = [('width', C.c_int32), ('height', C.c_int32), ('ncomp', C.c_int32), ('progressive', C.c_int32),
                ('orientation', C.c_int32), ('color', C.c_int32), ('out_width', C.c_int32), ('out_height', C.c_int32),
                ('hs', C.c_int32 * 3), ('vs', C.c_int32 * 3), ('hf', C.c_int32 * 3), ('vf', C.c_int32 * 3),
                ('bw', C.c_int32 * 3), ('bh', C.c_int32 * 3), ('dw', C.c_int32 * 3), ('dh', C.c_int32 * 3),
                ('coef_count', C.c_int64), ('plane_bytes', C.c_int64)]


class ChainDesc(C.Structure):
    _fields_ = [('x', C.c_void_p), ('res', C.c_void_p), ('y', C.c_void_p), ('z', C.c_void_p),
                ('w_a_h2', C.c_void_p), ('w_b_h2', C.c_void_p),
                ('scale_a_h2', C.c_void_p), ('bias_a', C.c_void_p), ('scale_b_h2', C.c_void_p), ('bias_b', C.c_void_p),
                ('x_amax', C.c_void_p), ('y_amax', C.c_void_p), ('z_amax', C.c_void_p),
                ('M', C.c_int64), ('ldx', C.c_int32), ('res_ld', C.c_int32), ('ldy', C.c_int32), ('ldz', C.c_int32),
                ('k_a', C.c_int32), ('n_a', C.c_int32), ('n_b', C.c_int32), ('cout_pad_a', C.c_int32), ('cout_pad_b', C.c_int32),
                ('act_a', C.c_int32), ('act_b', C.c_int32), ('_pad0', C.c_int32),
                ('res_amax', C.c_void_p), ('gain_a', C.c_float), ('bias_max_a', C.c_float)]


class StemDesc(C.Structure):
    _fields_ = [('x', C.c_void_p), ('y', C.c_void_p), ('B', C.c_int32), ('H', C.c_int32), ('W', C.c_int32), ('cout_pad', C.c_int32),
                ('w_h2', C.c_void_p), ('scale_h2', C.c_void_p), ('bias', C.c_void_p), ('y_amax', C.c_void_p),
                ('kpad', C.c_int32), ('_pad0', C.c_int32)]


class PlanOp(C.Structure):
    _fields_ = [('kind', C.c_int32), ('stream', C.c_int32), ('section', C.c_int32), ('_pad', C.c_int32), ('desc', C.c_void_p),
                ('p', C.c_void_p * 3), ('i', C.c_int64 * 8), ('f', C.c_double * 2)]


(OP_NOP, OP_CONV, OP_WINO, OP_DCN, OP_CHAIN, OP_STEM, OP_INPUT, OP_BILINEAR, OP_MAXPOOL, OP_BILINEAR_ADD, OP_RECORD, OP_WAIT,
 OP_MEMSET) = range(13)
SEC_NONE, SEC_BACKBONE, SEC_FPN, SEC_PROTO, SEC_HEADS = range(5)


class MaskIouShape(C.Structure):
    _fields_ = [('A', C.c_int32), ('B', C.c_int32), ('n', C.c_int64)]


class RleShape(C.Structure):
    _fields_ = [('N', C.c_int32), ('h', C.c_int32), ('w', C.c_int32), ('cap', C.c_int32)]


# ymi_workspace_bytes selectors (include/yolact_amd.h YMI_WS_*)
(WS_WINO_V, WS_WINO_M, WS_SPLITK, WS_MASK_IOU, WS_JPEG_COEFS, WS_JPEG_PLANES, WS_DETECT_SCORES_T, WS_DETECT_PER_PRIOR,
 WS_DETECT_CAND, WS_DETECT_REC, WS_AMAX_SLOT, WS_RLE_COUNTS) = range(1, 13)

EFORMAT, EUNSUPPORTED = -4, -5

# every symbol include/yolact_amd.h declares: (name, restype, argtypes)
_P, _I, _F = C.c_void_p, C.c_int, C.c_float
SYMBOLS = [
    ('ymi_abi_version', C.c_int, []),
    ('ymi_strerror', C.c_char_p, [C.c_int]),
    ('ymi_conv2d_nhwc_f32', C.c_int, [C.POINTER(ConvDesc), _P]),
    ('ymi_conv3x3_winograd_f32', C.c_int, [C.POINTER(WinoDesc), _P]),
    ('ymi_conv_flops', C.c_double, [C.POINTER(ConvDesc)]),
    ('ymi_conv_pick_tile', C.c_int, [C.POINTER(ConvDesc)]),
    ('ymi_amax_f32', C.c_int, [_P, C.c_long, _P, _P]),
    ('ymi_nchw_to_nhwc4_f32', C.c_int, [_P, _P, _I, _I, _I, _I, _P]),
    ('ymi_nchw_to_nhwc4_amax_f32', C.c_int, [_P, _P, _I, _I, _I, _I, _P, _P]),
    ('ymi_nhwc_to_nchw_f32', C.c_int, [_P, _P, _I, _I, _I, _I, _P]),
    ('ymi_maxpool3x3s2_nhwc_f32', C.c_int, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    ('ymi_bilinear_nhwc_f32', C.c_int, [_P, _P, _I, _I, _I, _I, _I, _I, _F, _F, _I, _P]),
    ('ymi_bilinear_add_nhwc_f32', C.c_int, [_P, _P, _I, _I, _I, _I, _I, _I, _P, _P]),
    ('ymi_conv2d_direct_nhwc_f32', C.c_int, [_P, _P, _P, _P] + [_I] * 12 + [_P]),
    ('ymi_global_maxpool_nhwc_f32', C.c_int, [_P, _P, _I, _I, _I, _P]),
    ('ymi_detect_f32', C.c_int, [C.POINTER(DetectDesc), _P]),
    ('ymi_lincomb_crop_f32', C.c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    ('ymi_mask_upsample_f32', C.c_int, [_P, _P, _I, _I, _I, _I, _I, _F, _P]),
    ('ymi_lincomb_crop_batch_f32', C.c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    ('ymi_mask_upsample_batch_f32', C.c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _P]),
    ('ymi_boxes_to_pixels', C.c_int, [_P, _P, _I, _I, _I, _P]),
    ('ymi_dcn_v2_forward_f32', C.c_int, [C.POINTER(DcnDesc), _P]),
    ('ymi_composite_masks_u8', C.c_int, [_P, _P, _P, _I, _I, _I, _F, _P, _P]),
    ('ymi_mask_iou_f32', C.c_int, [_P, _P, _I, _I, C.c_long, _I, _P, _P, _P]),
    ('ymi_jaccard_f32', C.c_int, [_P, _P, _I, _I, _I, _P, _P]),
    ('ymi_mask_bits_f32', C.c_int, [_P, _I, C.c_long, _P, _P]),
    ('ymi_mask_upsample_bits', C.c_int, [_P, _I, _I, _I, _I, _I, _F, _P, _P]),
    ('ymi_mask_iou_bits', C.c_int, [_P, _P, _I, _I, C.c_long, _I, _P, _P]),
    ('ymi_stem_pool_f32', C.c_int, [_P, _P]),
    ('ymi_pointwise_chain_f32', C.c_int, [_P, _P]),
    ('ymi_mask_rle_f32', C.c_int, [_P, _I, _I, _I, _P, _P, _I, _P]),
    ('ymi_mask_rle_upsampled_f32', C.c_int, [_P, _I, _I, _I, _I, _I, _F, _P, _P, _I, _P]),
    ('ymi_rle_to_string', C.c_int, [_P, _P, _I, _I, _P, _P, _I, _P]),
    ('ymi_fast_base_transform_f32', C.c_int, [_P, _P, _I, _I, _I, _I, _I, C.POINTER(C.c_float), C.POINTER(C.c_float), _I, _I, _P]),
    ('ymi_jpeg_parse', C.c_int, [_P, C.c_size_t, C.POINTER(JpegInfo)]),
    ('ymi_jpeg_decode_coefs', C.c_int, [_P, C.c_size_t, _P, C.c_int64, _P, C.POINTER(JpegInfo)]),
    ('ymi_jpeg_reconstruct_bgr_u8', C.c_int, [C.POINTER(JpegInfo), _P, _P, _P, _P, _P]),
    ('ymi_coco_poly_fill_u8', C.c_int, [_P, _I, _I, _I, _P]),
    ('ymi_coco_rle_fill_u8', C.c_int, [_P, C.c_long, _I, _I, _P]),
    ('ymi_coco_rle_string_fill_u8', C.c_int, [C.c_char_p, C.c_long, _I, _I, _P]),
    ('ymi_event_create', C.c_int, [C.POINTER(C.c_void_p)]),
    ('ymi_event_destroy', C.c_int, [_P]),
    ('ymi_plan_run', C.c_int, [C.POINTER(PlanOp), _I, _I, _P, _P, C.POINTER(C.c_void_p), _I, _I, C.POINTER(C.c_int32)]),
    ('ymi_workspace_bytes', C.c_int64, [_I, _P]),
    ('ymi_calib_mfma_f16', C.c_int, [_P, _I, _I, C.POINTER(C.c_double), _P]),
    ('ymi_calib_hbm_copy', C.c_int, [_P, _P, C.c_long, C.POINTER(C.c_double), _P]),
    ('ymi_calib_l2_read', C.c_int, [_P, C.c_long, _I, _I, _P, C.POINTER(C.c_double), _P]),
    ('ymi_calib_latency', C.c_int, [_P, C.c_long, _I, _I, _P, _P]),
    ('ymi_debug_set_trace', C.c_int, [_P, C.c_long]),
    ('ymi_prof_enable', C.c_int, [_I]),
    ('ymi_prof_count', C.c_int, []),
    ('ymi_prof_read', C.c_int, [_I, C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_int32),
                                C.POINTER(C.c_int32)]),
    ('ymi_prof_reset', C.c_int, []),
]

_lib = None


def lib():
    """Load (once) and return the HIP library; loud failure if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                'yolact_amd: HIP library %s not found. Build it with `python -c "import __graft_entry__ as g; '
                'g.build()"` or `make -C yolact_amd/csrc`. There is no CPU fallback.' % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(l, name)   # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        if l.ymi_abi_version() != ABI_VERSION:
            raise RuntimeError('yolact_amd: ABI mismatch: library %d, binding %d' % (l.ymi_abi_version(), ABI_VERSION))
        _lib = l
    return _lib


def check(rc: int, what: str = ''):
    if rc != 0:
        msg = lib().ymi_strerror(rc).decode()
        raise RuntimeError('yolact_amd: %s failed: %s (code %d)' % (what or 'call', msg, rc))


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_cuda(t, name='tensor'):
    if not t.is_cuda:
        raise RuntimeError('yolact_amd: %s must live on the GPU (MI355X/HIP); there is no CPU path in the product '
                           '(the CPU oracle lives under oracle/ and is test-only)' % name)
