"""`postprocess` — layers/output_utils.py:15-122 on device (lincomb masks).

Signature, defaults, return types and the empty-result sentinel are the reference's.  The body is two kernels
(csrc/mask.hip): prototype x coefficient combination on the fp32 matrix cores with fused sigmoid + crop into a
small [N,ph,pw] buffer, then one bandwidth kernel that upsamples to (h, w), binarises and writes the [N,h,w]
float32 result once.  YOLACT++ mask re-scoring (FastMaskIoUNet) runs between the two like output_utils.py:79-88.
"""
from __future__ import annotations

import ctypes as C

import torch

from .. import _lib as L
from ..config import active_cfg, act_name, is_lincomb


def _empty(device):
    return [torch.empty(0, device=device)] * 4   # "4 copies of the same thing" like output_utils.py:40


def postprocess(det_output, w, h, batch_idx=0, interpolation_mode='bilinear', visualize_lincomb=False,
                crop_masks=True, score_threshold=0):
    cfg = active_cfg()
    dets = det_output[batch_idx]
    net = dets['net']
    dets = dets['detection']
    if dets is None:
        return [torch.Tensor()] * 4
    if score_threshold > 0:
        keep = dets['score'] > score_threshold
        for k in dets:
            if k != 'proto':
                dets[k] = dets[k][keep]
        if dets['score'].size(0) == 0:
            return [torch.Tensor()] * 4
    classes, boxes, scores, coef = dets['class'], dets['box'], dets['score'], dets['mask']
    if not (is_lincomb(cfg) and cfg.eval_mask_branch):
        raise NotImplementedError('only mask_type.lincomb is on the hot path')
    if interpolation_mode != 'bilinear':
        raise NotImplementedError("interpolation_mode %r (eval.py always uses 'bilinear')" % interpolation_mode)
    if visualize_lincomb:
        raise NotImplementedError('display_lincomb is a matplotlib debug helper, out of scope')
    if act_name(cfg.mask_proto_mask_activation) != 'sigmoid':
        raise NotImplementedError('mask activation other than sigmoid')
    proto = dets['proto']
    L.require_cuda(proto, "dets['proto']")
    dev = proto.device
    ph, pw, D = proto.shape
    N = int(coef.shape[0])
    lib = L.lib()
    with torch.cuda.device(dev):
        s = L.stream_ptr()
        proto_c, coef_c, boxes_c = proto.contiguous(), coef.contiguous().float(), boxes.contiguous().float()
        masks_lo = torch.empty(N, ph, pw, device=dev)
        L.check(lib.ymi_lincomb_crop_f32(proto_c.data_ptr(), coef_c.data_ptr(), boxes_c.data_ptr(), masks_lo.data_ptr(),
                                         ph, pw, D, N, 1 if crop_masks else 0, s), 'ymi_lincomb_crop_f32')
        if getattr(cfg, 'use_maskiou', False):
            maskiou_p = net.maskiou_forward(masks_lo)                       # [N, 80]
            maskiou_p = torch.gather(maskiou_p, 1, classes.unsqueeze(1)).squeeze(1)
            if cfg.rescore_mask:
                scores = scores * maskiou_p if cfg.rescore_bbox else [scores, scores * maskiou_p]
        masks = torch.empty(N, h, w, device=dev)
        L.check(lib.ymi_mask_upsample_f32(masks_lo.data_ptr(), masks.data_ptr(), N, ph, pw, h, w, C.c_float(0.5), s),
                'ymi_mask_upsample_f32')
        boxes_px = torch.empty(N, 4, dtype=torch.int64, device=dev)
        L.check(lib.ymi_boxes_to_pixels(boxes_c.data_ptr(), boxes_px.data_ptr(), N, w, h, s), 'ymi_boxes_to_pixels')
    return classes, scores, boxes_px, masks


def undo_image_transformation(img, w, h):
    raise NotImplementedError('display helper (output_utils.py:128-144, needs cv2) — out of scope for the hot path')
