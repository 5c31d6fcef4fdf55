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


def _lowres_masks(det_output, w, h, batch_idx, interpolation_mode, visualize_lincomb, crop_masks, score_threshold):
    """The part of `postprocess` up to the upsample (output_utils.py:35-88): score filter, checks, lincomb + sigmoid + crop
    at prototype resolution, YOLACT++ re-scoring, integer boxes.  None for an empty result."""
    cfg = active_cfg()
    dets = det_output[batch_idx]
    net = dets['net']
    dets = dets['detection']
    if dets is None:
        return None
    if score_threshold > 0:
        keep = dets['score'] > score_threshold
        for k in dets:
            if k != 'proto':
                dets[k] = dets[k][keep]
        if dets['score'].size(0) == 0:
            return None
    classes, boxes, scores, coef = dets['class'], dets['box'], dets['score'], dets['mask']
    if not is_lincomb(cfg):
        raise NotImplementedError('only mask_type.lincomb is on the hot path')
    if not cfg.eval_mask_branch:
        # output_utils.py:58,97-122 with cfg.eval_mask_branch False (eval.py --detect, eval.py:1067-1068): neither mask branch runs;
        # classes / scores / integer boxes are returned and the 4th value stays what dets['mask'] holds (the coefficient rows)
        L.require_cuda(boxes, "dets['box']")
        with torch.cuda.device(boxes.device):
            boxes_c = boxes.contiguous().float()
            boxes_px = torch.empty(boxes_c.shape[0], 4, dtype=torch.int64, device=boxes.device)
            L.check(L.lib().ymi_boxes_to_pixels(boxes_c.data_ptr(), boxes_px.data_ptr(), int(boxes_c.shape[0]), w, h, L.stream_ptr()),
                    'ymi_boxes_to_pixels')
        return classes, scores, boxes_px, None
    if interpolation_mode != 'bilinear':
        raise NotImplementedError("interpolation_mode %r (eval.py always uses 'bilinear')" % interpolation_mode)
    if visualize_lincomb:
        raise NotImplementedError('display_lincomb is a matplotlib debug helper, out of scope')
    if act_name(cfg.mask_proto_mask_activation) != 'sigmoid':
        raise NotImplementedError('mask activation other than sigmoid')
    proto = dets['proto']
    if proto is None:
        raise RuntimeError('postprocess: this detection was gathered from another rank (Yolact.forward_sharded) and carries no '
                           'prototypes; masks are assembled on the rank that computed the image')
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
        boxes_px = torch.empty(N, 4, dtype=torch.int64, device=dev)
        L.check(lib.ymi_boxes_to_pixels(boxes_c.data_ptr(), boxes_px.data_ptr(), N, w, h, s), 'ymi_boxes_to_pixels')
    return classes, scores, boxes_px, masks_lo


def postprocess(det_output, w, h, batch_idx=0, interpolation_mode='bilinear', visualize_lincomb=False,
                crop_masks=True, score_threshold=0):
    r = _lowres_masks(det_output, w, h, batch_idx, interpolation_mode, visualize_lincomb, crop_masks, score_threshold)
    if r is None:
        return [torch.Tensor()] * 4
    classes, scores, boxes_px, masks_lo = r
    if masks_lo is None:                       # cfg.eval_mask_branch False: boxes only (see _lowres_masks)
        return classes, scores, boxes_px, det_output[batch_idx]['detection']['mask']
    N, ph, pw = masks_lo.shape
    with torch.cuda.device(masks_lo.device):
        masks = torch.empty(N, h, w, device=masks_lo.device)
        L.check(L.lib().ymi_mask_upsample_f32(masks_lo.data_ptr(), masks.data_ptr(), N, ph, pw, h, w, C.c_float(0.5),
                                              L.stream_ptr()), 'ymi_mask_upsample_f32')
    return classes, scores, boxes_px, masks


def postprocess_rle(det_output, w, h, batch_idx=0, crop_masks=True, score_threshold=0, fused=False):
    """`postprocess` for the COCO result path (eval.py:403-429 under --output_coco_json): same classes / scores / boxes,
    but the masks come back as the COCO RLE records `Detections.add_mask` would have produced from them
    ({'size': [h, w], 'counts': str}); only the strings leave the device.  Empty result: ([], [], [], []).

    fused=True: ONE kernel upsamples, thresholds and run-length encodes the prototype-resolution masks
    (`ymi_mask_rle_upsampled_f32`), so the [N,h,w] float masks (121 MB per image at 550 x 550) are never allocated or
    written.  Byte-identical records either way.  Measured (100 masks, 138^2 -> 550^2, profiles/r02_rle_fused_probe.json): the
    fused kernel is the slower of the two on this chip (0.63 vs 0.54 ms: the 242 MB HBM round trip it saves costs less than
    evaluating the interpolation in both passes of a column walk), so it is the opt-in for callers that care about the
    footprint, not the default."""
    from ..coco import rle_encode, rle_encode_lowres
    r = _lowres_masks(det_output, w, h, batch_idx, 'bilinear', False, crop_masks, score_threshold)
    if r is None:
        return [], [], [], []
    classes, scores, boxes_px, masks_lo = r
    if masks_lo is None:
        raise RuntimeError('postprocess_rle: cfg.eval_mask_branch is False (eval.py --detect): there are no masks to encode')
    if fused:
        return classes, scores, boxes_px, rle_encode_lowres(masks_lo, h, w, 0.5)
    N, ph, pw = masks_lo.shape
    with torch.cuda.device(masks_lo.device):
        masks = torch.empty(N, h, w, device=masks_lo.device)
        L.check(L.lib().ymi_mask_upsample_f32(masks_lo.data_ptr(), masks.data_ptr(), N, ph, pw, h, w, C.c_float(0.5),
                                              L.stream_ptr()), 'ymi_mask_upsample_f32')
    return classes, scores, boxes_px, rle_encode(masks)


def postprocess_bits(det_output, w, h, batch_idx=0, crop_masks=True, score_threshold=0):
    """`postprocess` for the metric path (eval.py:403-440 prep_metrics): same classes / scores / integer boxes, but the masks come
    back as BITS — int64 [N, ceil(h*w/64)], bit i of word j = pixel 64 j + i of the flat [h, w] mask — written by one kernel
    that upsamples and thresholds the prototype-resolution masks (the float kernel's arithmetic: every bit equals the pixel
    `postprocess` would have written).  3.8 MB instead of 121 MB per image at 550 x 550; layers.box_utils.mask_iou_bits scores
    them against bit-packed ground truth with popcounts, bit-identical to mask_iou on the float masks.  Empty: ([], [], [], None)."""
    rec = det_output[batch_idx]
    dets = rec['detection']
    if dets is not None and dets.get('proto') is None and 'mask_bits' in dets:
        # a detection gathered from another rank by Yolact.forward_sharded(masks='bits'): its masks were assembled where its
        # prototypes live (postprocess_bits_batch) and arrived as bits for the size `mask_size`
        if tuple(rec.get('mask_size', ())) != (h, w):
            raise RuntimeError('postprocess_bits: the gathered masks were assembled for (h, w) = %s, not %s'
                               % (rec.get('mask_size'), (h, w)))
        if score_threshold > 0:
            keep = dets['score'] > score_threshold
            for k in dets:
                if k != 'proto':
                    dets[k] = dets[k][keep]
            if dets['score'].size(0) == 0:
                return [], [], [], None
        boxes = dets['box'].contiguous().float()
        if boxes.is_cuda:
            with torch.cuda.device(boxes.device):
                boxes_px = torch.empty(boxes.shape[0], 4, dtype=torch.int64, device=boxes.device)
                L.check(L.lib().ymi_boxes_to_pixels(boxes.data_ptr(), boxes_px.data_ptr(), int(boxes.shape[0]), w, h, L.stream_ptr()),
                        'ymi_boxes_to_pixels')
        else:
            raise RuntimeError('postprocess_bits: gathered detections must live on the GPU (there is no CPU path)')
        return dets['class'], dets['score'], boxes_px, dets['mask_bits']
    r = _lowres_masks(det_output, w, h, batch_idx, 'bilinear', False, crop_masks, score_threshold)
    if r is None:
        return [], [], [], None
    classes, scores, boxes_px, masks_lo = r
    if masks_lo is None:
        raise RuntimeError('postprocess_bits: cfg.eval_mask_branch is False (eval.py --detect): there are no masks to pack')
    N, ph, pw = masks_lo.shape
    W64 = (h * w + 63) // 64
    with torch.cuda.device(masks_lo.device):
        bits = torch.empty(N, W64, dtype=torch.int64, device=masks_lo.device)
        L.check(L.lib().ymi_mask_upsample_bits(masks_lo.data_ptr(), N, ph, pw, h, w, C.c_float(0.5), bits.data_ptr(),
                                               L.stream_ptr()), 'ymi_mask_upsample_bits')
    return classes, scores, boxes_px, bits


def postprocess_batch(dev_out, w, h, crop_masks=True, net=None):
    """postprocess() for a whole batch with NO per-image Python loop and no host synchronisation: the fixed-capacity
    device outputs of `Yolact.forward_device` (count [B], box [B,cap,4], score, cls, coef [B,cap,D], proto [B,ph,pw,D])
    go through ONE lincomb+sigmoid+crop launch and ONE upsample+threshold launch (output_utils.py:69-99 per image in the
    reference).  Returns fixed-capacity tensors: classes [B,cap] i64, scores [B,cap], boxes [B,cap,4] i64 pixels,
    masks [B,cap,h,w] f32 {0,1}, count [B] i32; rows >= count[b] are unspecified.  `masks[b, :count[b]]` equals what
    postprocess(preds, w, h, batch_idx=b) returns bit for bit (same kernels).

    YOLACT++ (cfg.use_maskiou; output_utils.py:79-88, yolact.py:363-375): FastMaskIoUNet runs ONCE over all B * cap cropped
    prototype-resolution masks (one chain of six launches for the batch instead of one per image; `net` = the model that owns
    maskiou_net, default dev_out['net']) and 'scores' follows the reference's structure: [box scores, box scores * maskiou]
    (two [B,cap] tensors) with cfg.rescore_mask and not cfg.rescore_bbox, the product alone with rescore_bbox (what eval.py's
    prep_display forces, eval.py:147-152).  Row (b, i < count[b]) equals the per-image postprocess() bit for bit.

    cfg.eval_mask_branch False (eval.py --detect): 'masks' is None, boxes / classes / scores as usual (output_utils.py:58,97-122)."""
    cfg = active_cfg()
    if not is_lincomb(cfg) or act_name(cfg.mask_proto_mask_activation) != 'sigmoid':
        raise NotImplementedError('only mask_type.lincomb with sigmoid masks is on the hot path')
    if not cfg.eval_mask_branch:
        box, count = dev_out['box'], dev_out['count']
        L.require_cuda(box, "dev_out['box']")
        B, cap = box.shape[0], box.shape[1]
        with torch.cuda.device(box.device):
            boxes_px = torch.empty(B, cap, 4, dtype=torch.int64, device=box.device)
            L.check(L.lib().ymi_boxes_to_pixels(box.data_ptr(), boxes_px.data_ptr(), B * cap, w, h, L.stream_ptr()), 'ymi_boxes_to_pixels')
        return {'classes': dev_out['cls'], 'scores': dev_out['score'], 'boxes': boxes_px, 'masks': None, 'count': count}
    proto, coef, box, count = dev_out['proto'], dev_out['coef'], dev_out['box'], dev_out['count']
    L.require_cuda(proto, "dev_out['proto']")
    B, ph, pw, D = proto.shape
    cap = coef.shape[1]
    dev = proto.device
    lib = L.lib()
    with torch.cuda.device(dev):
        s = L.stream_ptr()
        use_miou = bool(getattr(cfg, 'use_maskiou', False))
        # (YOLACT++: rows past count[b] also go through FastMaskIoUNet, so they must hold numbers, not uninitialised memory)
        masks_lo = (torch.zeros if use_miou else torch.empty)(B, cap, ph, pw, device=dev)
        L.check(lib.ymi_lincomb_crop_batch_f32(proto.data_ptr(), coef.data_ptr(), box.data_ptr(), count.data_ptr(),
                                               masks_lo.data_ptr(), B, cap, ph, pw, D, 1 if crop_masks else 0, s),
                'ymi_lincomb_crop_batch_f32')
        scores = dev_out['score']
        if use_miou:
            net = net if net is not None else dev_out.get('net')
            if net is None:
                raise RuntimeError('postprocess_batch: cfg.use_maskiou needs the model that owns maskiou_net (net=...)')
            miou = net.maskiou_forward(masks_lo.view(B * cap, ph, pw)).view(B, cap, -1)            # [B, cap, 80]
            miou = torch.gather(miou, 2, dev_out['cls'].clamp(0, miou.shape[2] - 1).unsqueeze(-1)).squeeze(-1)
            if cfg.rescore_mask:
                scores = scores * miou if cfg.rescore_bbox else [scores, scores * miou]
        masks = torch.empty(B, cap, h, w, device=dev)
        L.check(lib.ymi_mask_upsample_batch_f32(masks_lo.data_ptr(), count.data_ptr(), masks.data_ptr(), B, cap, ph, pw, h,
                                                w, C.c_float(0.5), s), 'ymi_mask_upsample_batch_f32')
        boxes_px = torch.empty(B, cap, 4, dtype=torch.int64, device=dev)
        L.check(lib.ymi_boxes_to_pixels(box.data_ptr(), boxes_px.data_ptr(), B * cap, w, h, s), 'ymi_boxes_to_pixels')
    return {'classes': dev_out['cls'], 'scores': scores, 'boxes': boxes_px, 'masks': masks, 'count': count}


def postprocess_bits_batch(dev_out, w, h, crop_masks=True):
    """postprocess_bits for a whole fixed-capacity batch (the data-parallel mask path: yolact_amd.parallel / Yolact.forward_sharded
    with masks='bits'): ONE lincomb + sigmoid + crop launch and ONE upsample + threshold-into-bits launch for all B * cap
    detections.  Returns {'classes', 'scores', 'boxes' [B,cap,4] i64, 'bits' int64 [B, cap, ceil(h*w/64)], 'count', 'size': (h, w)};
    bits[b, i < count[b]] equals what postprocess_bits(preds, w, h, batch_idx=b) returns, rows past count[b] are all-zero words.
    3.8 MB per image at 550 x 550 and cap 100 instead of the 121 MB of float masks — the form in which masks travel to the
    gather root ("masks are assembled where the prototypes live and only gathered if the caller needs them")."""
    cfg = active_cfg()
    if not (is_lincomb(cfg) and cfg.eval_mask_branch) or act_name(cfg.mask_proto_mask_activation) != 'sigmoid':
        raise NotImplementedError('only mask_type.lincomb with sigmoid masks (and cfg.eval_mask_branch) has masks to pack')
    proto, coef, box, count = dev_out['proto'], dev_out['coef'], dev_out['box'], dev_out['count']
    L.require_cuda(proto, "dev_out['proto']")
    B, ph, pw, D = proto.shape
    cap = coef.shape[1]
    dev = proto.device
    lib = L.lib()
    W64 = (h * w + 63) // 64
    with torch.cuda.device(dev):
        s = L.stream_ptr()
        masks_lo = torch.zeros(B, cap, ph, pw, device=dev)          # rows past count[b] stay zero -> all-zero bit rows
        L.check(lib.ymi_lincomb_crop_batch_f32(proto.data_ptr(), coef.data_ptr(), box.data_ptr(), count.data_ptr(),
                                               masks_lo.data_ptr(), B, cap, ph, pw, D, 1 if crop_masks else 0, s),
                'ymi_lincomb_crop_batch_f32')
        bits = torch.empty(B, cap, W64, dtype=torch.int64, device=dev)
        L.check(lib.ymi_mask_upsample_bits(masks_lo.data_ptr(), B * cap, ph, pw, h, w, C.c_float(0.5), bits.data_ptr(), s),
                'ymi_mask_upsample_bits')
        boxes_px = torch.empty(B, cap, 4, dtype=torch.int64, device=dev)
        L.check(lib.ymi_boxes_to_pixels(box.data_ptr(), boxes_px.data_ptr(), B * cap, w, h, s), 'ymi_boxes_to_pixels')
    return {'classes': dev_out['cls'], 'scores': dev_out['score'], 'boxes': boxes_px, 'bits': bits, 'count': count, 'size': (h, w)}


MEANS = (103.94, 116.78, 123.68)     # data/config.py:28-29, BGR order
STD = (57.38, 57.12, 58.40)


def undo_image_transformation(img, w, h):
    """output_utils.py:128-144 (display helper of eval.py's prep_display when the frame was not kept): transformed
    [3,H,W] tensor -> float RGB ndarray [h,w,3] in 0..1.  Host-side numpy like the reference; the final resize is
    cv2.resize when OpenCV is installed, otherwise the same half-pixel bilinear mapping through torch."""
    import numpy as np
    cfg = active_cfg()
    img_numpy = img.permute(1, 2, 0).cpu().numpy()
    img_numpy = img_numpy[:, :, (2, 1, 0)]                      # to BGR
    tr = cfg.backbone.transform
    name = tr if isinstance(tr, str) else ('resnet' if tr.normalize else 'vgg' if tr.subtract_means else 'other')
    if name == 'resnet':
        img_numpy = (img_numpy * np.array(STD) + np.array(MEANS)) / 255.0
    elif name == 'vgg':
        img_numpy = (img_numpy / 255.0 + np.array(MEANS) / 255.0).astype(np.float32)
    img_numpy = img_numpy[:, :, (2, 1, 0)]                      # to RGB
    img_numpy = np.clip(img_numpy, 0, 1)
    try:
        import cv2
        if hasattr(cv2, 'resize'):
            return cv2.resize(img_numpy, (w, h))
    except ImportError:
        pass
    t = torch.from_numpy(np.ascontiguousarray(img_numpy)).permute(2, 0, 1).unsqueeze(0)
    t = torch.nn.functional.interpolate(t, (h, w), mode='bilinear', align_corners=False)
    return t.squeeze(0).permute(1, 2, 0).contiguous().numpy()
