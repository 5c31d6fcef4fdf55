"""`FastBaseTransform` — utils/augmentations.py:616-658 on device (the step right before
`Yolact.forward` in eval.py's evalimage / evalvideo, eval.py:596-597,692-695).

Same module interface: `FastBaseTransform()(img)` with img `[n, h, w, c]` float BGR on the GPU returns `[n, 3, S, S]`
normalised RGB (NCHW).  One HIP kernel (csrc/preprocess.hip) instead of permute + interpolate + normalise + index.
`to_nhwc4(img)` returns the engine's native `[n, S, S, 4]` layout directly.
"""
from __future__ import annotations

import ctypes as C
from math import sqrt

import torch

from .. import _lib as L
from ..config import active_cfg

MEANS = (103.94, 116.78, 123.68)     # data/config.py:28-29 (BGR order)
STD = (57.38, 57.12, 58.40)


def calc_size_preserve_ar(img_w, img_h, max_size):
    """Resize.calc_size_preserve_ar, utils/augmentations.py:132-138."""
    ratio = sqrt(img_w / img_h)
    return int(max_size * ratio), int(max_size / ratio)


def _transform_mode(cfg):
    tr = cfg.backbone.transform
    if isinstance(tr, str):          # yolact_amd.config stores the transform by name
        tr = {'resnet': dict(normalize=True, subtract_means=False, to_float=False, channel_order='RGB'),
              'vgg': dict(normalize=False, subtract_means=True, to_float=False, channel_order='RGB'),
              'darknet': dict(normalize=False, subtract_means=False, to_float=True, channel_order='RGB')}[tr]
        get = tr.get
    else:
        get = lambda k: getattr(tr, k)   # noqa: E731  (the reference's Config object)
    if get('channel_order') != 'RGB':
        raise NotImplementedError          # like augmentations.py:652-653
    if get('normalize'):
        return 0
    if get('subtract_means'):
        return 1
    if get('to_float'):
        return 2
    return 3


class FastBaseTransform(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self._mean = (C.c_float * 3)(*MEANS)
        self._std = (C.c_float * 3)(*STD)

    def _run(self, img, nhwc4):
        L.require_cuda(img, 'image batch')
        cfg = active_cfg()
        if img.dim() != 4 or img.shape[3] != 3:
            raise ValueError('expected [n, h, w, 3] BGR, got %s' % (tuple(img.shape),))
        n, h, w, _ = img.shape
        if getattr(cfg, 'preserve_aspect_ratio', False):
            ow, oh = calc_size_preserve_ar(w, h, cfg.max_size)
        else:
            oh = ow = cfg.max_size
        img = img.detach().to(torch.float32).contiguous()
        out = torch.empty((n, oh, ow, 4) if nhwc4 else (n, 3, oh, ow), dtype=torch.float32, device=img.device)
        with torch.cuda.device(img.device):
            L.check(L.lib().ymi_fast_base_transform_f32(img.data_ptr(), out.data_ptr(), n, h, w, oh, ow, self._mean,
                                                        self._std, _transform_mode(cfg), 1 if nhwc4 else 0,
                                                        L.stream_ptr()), 'ymi_fast_base_transform_f32')
        return out

    def forward(self, img):
        return self._run(img, False)

    def to_nhwc4(self, img):
        return self._run(img, True)


class BaseTransform:
    """`BaseTransform` — utils/augmentations.py:601-612 (the `transform` eval.py hands COCODetection, eval.py:1097): the
    pipeline ConvertFromInts -> Resize(resize_gt=False) -> BackboneTransform(cfg.backbone.transform, mean, std, 'BGR').

    The image half is the FastBaseTransform kernel (bilinear resize with half-pixel centres, normalisation, BGR -> RGB): `img`
    is the device tensor data.jpeg.imread returned (uint8 / float BGR [h,w,3]; a numpy array is uploaded) and comes back as a
    float32 [S,S,3] device view (HWC like the reference's array, so `.permute(2, 0, 1)` gives the network input).
    cv2.resize and this kernel evaluate the same bilinear formula in fp32 but round the sample coordinate differently
    (cv2: double -> float; here: fp32 throughout, as F.interpolate), so values agree to ~1e-5 of the normalised range, not
    bit for bit.  The ground-truth half is the reference's host code: `Resize(resize_gt=False)` leaves masks / boxes at
    their size and drops boxes narrower than cfg.discard_box_width / height (utils/augmentations.py:170-178)."""

    def __init__(self, mean=MEANS, std=STD):
        self._fbt = FastBaseTransform()
        self._fbt._mean = (C.c_float * 3)(*mean)
        self._fbt._std = (C.c_float * 3)(*std)

    def __call__(self, img, masks=None, boxes=None, labels=None):
        import numpy as np
        cfg = active_cfg()
        if not torch.is_tensor(img):
            img = torch.from_numpy(np.ascontiguousarray(img))
        if not img.is_cuda:
            if not torch.cuda.is_available():
                L.require_cuda(img, 'image')
            img = img.cuda()
        out = self._fbt(img.unsqueeze(0))[0].permute(1, 2, 0)      # [S,S,3] view of the NCHW result
        if boxes is not None and labels is not None:
            w = boxes[:, 2] - boxes[:, 0]
            h = boxes[:, 3] - boxes[:, 1]
            keep = (w > getattr(cfg, 'discard_box_width', 4 / 550)) * (h > getattr(cfg, 'discard_box_height', 4 / 550))
            masks = masks[keep]
            boxes = boxes[keep]
            labels['labels'] = labels['labels'][keep]
            labels['num_crowds'] = (labels['labels'] < 0).sum()
        return out, masks, boxes, labels
