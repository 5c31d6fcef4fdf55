"""CPU restatement of `COCODetection.pull_item` + `BaseTransform` (data/coco.py:100-176, utils/augmentations.py:601-612,
:129-180, :566-596).  *** TEST INFRASTRUCTURE ONLY ***

  image     jpeg_oracle.imread_bgr           (cv2.imread, pinned to libjpeg-turbo)
  masks     coco_mask.ann_to_mask            (pycocotools annToMask, restated; see that file for what is pinned)
  target    COCOAnnotationTransform          (data/coco.py:19-51, restated verbatim in numpy)
  transform ConvertFromInts -> Resize(resize_gt=False) -> BackboneTransform('BGR' in): cv2.resize(float32, INTER_LINEAR)
            restated from OpenCV's published resize.cpp (coordinates: fx = float((dx + 0.5) * scale - 0.5) with scale a
            double; horizontal pass then vertical pass in float32); discard of narrow boxes; (x - mean) / std; BGR -> RGB.
            cv2 is not installed, so `cv2_resize_linear_f32` is PARITY UNPINNED against OpenCV itself; it is the same bilinear
            formula as F.interpolate(align_corners=False) up to the rounding of the sample coordinate, which the test bounds.
"""
from __future__ import annotations

import json
import os.path as osp
from collections import defaultdict

import numpy as np

from . import coco_mask, jpeg_oracle

MEANS = (103.94, 116.78, 123.68)
STD = (57.38, 57.12, 58.40)


def cv2_resize_linear_f32(img: np.ndarray, ow: int, oh: int) -> np.ndarray:
    """img float32 [h,w,c] -> [oh,ow,c], OpenCV INTER_LINEAR for CV_32F (resize.cpp: resizeGeneric_ / HResizeLinear /
    VResizeLinear with float coefficients)."""
    h, w = img.shape[:2]

    def coords(n_out, n_in):
        scale = n_in / n_out                                      # double
        f = ((np.arange(n_out, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = (f - s.astype(np.float32)).astype(np.float32)
        lo = s < 0
        s[lo], f[lo] = 0, 0
        hi = s >= n_in - 1
        s[hi], f[hi] = n_in - 1, 0
        return s, np.minimum(s + 1, n_in - 1), f

    sx, sx1, fx = coords(ow, w)
    sy, sy1, fy = coords(oh, h)
    src = img.astype(np.float32)
    a0, a1 = (np.float32(1) - fx)[None, :, None], fx[None, :, None]
    rows = src[:, sx] * a0 + src[:, sx1] * a1                     # horizontal pass, float32
    b0, b1 = (np.float32(1) - fy)[:, None, None], fy[:, None, None]
    return (rows[sy] * b0 + rows[sy1] * b1).astype(np.float32)


def base_transform(img_bgr_u8, masks, boxes, labels, max_size=550, mode='normalize', discard_w=4 / 550, discard_h=4 / 550):
    img = img_bgr_u8.astype(np.float32)                            # ConvertFromInts
    img = cv2_resize_linear_f32(img, max_size, max_size)           # Resize(resize_gt=False)
    if boxes is not None:
        w = boxes[:, 2] - boxes[:, 0]
        h = boxes[:, 3] - boxes[:, 1]
        keep = (w > discard_w) * (h > discard_h)
        masks = masks[keep]
        boxes = boxes[keep]
        labels['labels'] = labels['labels'][keep]
        labels['num_crowds'] = (labels['labels'] < 0).sum()
    mean, std = np.array(MEANS, dtype=np.float32), np.array(STD, dtype=np.float32)
    if mode == 'normalize':
        img = (img - mean) / std
    elif mode == 'subtract_means':
        img = img - mean
    elif mode == 'to_float':
        img = img / 255
    img = img[:, :, [2, 1, 0]]                                     # 'BGR' -> 'RGB'
    return img.astype(np.float32), masks, boxes, labels


def annotation_transform(target, width, height, label_map):
    scale = np.array([width, height, width, height])
    res = []
    for obj in target:
        if 'bbox' in obj:
            bbox = obj['bbox']
            label_idx = obj['category_id']
            if label_idx >= 0:
                label_idx = label_map[label_idx] - 1
            final_box = list(np.array([bbox[0], bbox[1], bbox[0] + bbox[2], bbox[1] + bbox[3]]) / scale)
            final_box.append(label_idx)
            res += [final_box]
    return res


def pull_item(root, info_file, index, label_map, has_gt=True, max_size=550, mode='normalize'):
    """-> (img float32 [3,S,S], target [n,5] | None, masks [n,h,w] uint8 | None, height, width, num_crowds); raises
    LookupError where the reference would resample a random other item (no ground truth left after the discard)."""
    ds = json.load(open(info_file))
    img_to_anns = defaultdict(list)
    for a in ds.get('annotations', []):
        img_to_anns[a['image_id']].append(a)
    imgs = {i['id']: i for i in ds['images']}
    ids = list(img_to_anns.keys())
    if len(ids) == 0 or not has_gt:
        ids = list(imgs.keys())
    img_id = ids[index]
    target = [dict(x) for x in img_to_anns[img_id]] if has_gt else []
    crowd = [x for x in target if ('iscrowd' in x and x['iscrowd'])]
    target = [x for x in target if not ('iscrowd' in x and x['iscrowd'])]
    num_crowds = len(crowd)
    for x in crowd:
        x['category_id'] = -1
    target += crowd
    file_name = imgs[img_id]['file_name']
    if file_name.startswith('COCO'):
        file_name = file_name.split('_')[-1]
    img = jpeg_oracle.imread_bgr(open(osp.join(root, file_name), 'rb').read())
    height, width, _ = img.shape
    masks = None
    if len(target) > 0:
        masks = np.stack([coco_mask.ann_to_mask(obj, height, width) for obj in target], axis=0)
        target = annotation_transform(target, width, height, label_map)
    if len(target) > 0:
        target = np.array(target)
        img, masks, boxes, labels = base_transform(img, masks, target[:, :4], {'num_crowds': num_crowds, 'labels': target[:, 4]},
                                                   max_size, mode)
        num_crowds = labels['num_crowds']
        target = np.hstack((boxes, np.expand_dims(labels['labels'], axis=1)))
        if target.shape[0] == 0:
            raise LookupError('no ground truth left: the reference resamples a random item here')
    else:
        img, _, _, _ = base_transform(img, None, None, None, max_size, mode)
        masks, target = None, None
    return img.transpose(2, 0, 1), target, masks, height, width, num_crowds
