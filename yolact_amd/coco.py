"""COCO result wire format: eval.py's `Detections` (eval.py:300-340) with the mask encoding on
the device.

The reference's `prep_metrics` copies every full-resolution float mask to the host (`masks.view(-1,h,w).cpu().numpy()`,
eval.py:422-423: 121 MB per image at 550 x 550) and `add_mask` run-length encodes each one with
`pycocotools.mask.encode(np.asfortranarray(mask.astype(np.uint8)))` (eval.py:320-324).  Here `rle_encode` keeps the
masks where `postprocess` left them, runs `ymi_mask_rle_f32` + `ymi_rle_to_string` (csrc/rle.hip) and brings back only
the ASCII strings (a few hundred bytes per mask).  `Detections` has the reference's methods, argument meaning and JSON
layout; `add_masks` is the batched entry the device path wants (one launch pair per image).
"""
from __future__ import annotations

import ctypes as C
import json

import numpy as np
import torch

from . import _lib as L
from .config import active_cfg

_MISSING_COCO_IDS = (12, 26, 29, 30, 45, 66, 68, 69, 71, 83)
# COCO's 80 category ids -> contiguous 1..80 (the table data/config.py:46-55 spells out)
COCO_LABEL_MAP = {cid: i + 1 for i, cid in enumerate(c for c in range(1, 91) if c not in _MISSING_COCO_IDS)}


def get_label_map(cfg=None):
    """data/config.py:... `get_label_map` (eval.py:283-288 consumer): identity when the dataset has no map."""
    cfg = cfg if cfg is not None else active_cfg()
    ds = getattr(cfg, 'dataset', None)
    lm = getattr(ds, 'label_map', None) if ds is not None else COCO_LABEL_MAP
    if lm is None:
        n = len(getattr(ds, 'class_names', ())) or (cfg.num_classes - 1)
        return {x + 1: x + 1 for x in range(n)}
    return lm


def _encode_strings(launch_counts, N, h, w, dev, cap):
    """Run `launch_counts(counts, nruns, cap, stream)` + the string kernel, growing `cap` until no mask is truncated."""
    lib = L.lib()
    with torch.cuda.device(dev):
        s = L.stream_ptr()
        while True:
            counts = torch.empty(N, cap, dtype=torch.int32, device=dev)
            nruns = torch.empty(N, dtype=torch.int32, device=dev)
            cap_chars = 7 * cap                       # a 32-bit (delta) count needs at most 7 characters of 5 bits
            text = torch.empty(N, cap_chars, dtype=torch.uint8, device=dev)
            nchars = torch.empty(N, dtype=torch.int32, device=dev)
            launch_counts(counts, nruns, cap, s)
            L.check(lib.ymi_rle_to_string(counts.data_ptr(), nruns.data_ptr(), N, cap, text.data_ptr(), nchars.data_ptr(),
                                          cap_chars, s), 'rle_to_string')
            both = torch.stack((nruns, nchars)).cpu()              # the one synchronising read
            need = int(both[0].max())
            if need <= cap:
                break
            cap = 1 << (need - 1).bit_length()                     # some mask was truncated: retry with room for it
        lens = both[1].tolist()
        width = max(lens)
        host = text[:, :width].cpu().numpy()
    return [{'size': [h, w], 'counts': host[i, :lens[i]].tobytes().decode('ascii')} for i in range(N)]


def rle_encode(masks: torch.Tensor, cap: int = 4096):
    """masks: [N,h,w] float32 CUDA tensor (postprocess' {0,1} masks).  Returns N dicts {'size': [h, w], 'counts': str},
    byte-identical to pycocotools.mask.encode(...)['counts'].decode('ascii').  `cap` = initial run capacity per mask
    (grown automatically when a mask has more runs)."""
    L.require_cuda(masks, 'masks')
    if masks.dim() != 3:
        raise ValueError('expected [N,h,w] masks, got %s' % (tuple(masks.shape),))
    N, h, w = (int(v) for v in masks.shape)
    if N == 0:
        return []
    masks = masks.detach().to(torch.float32).contiguous()
    lib = L.lib()

    def launch(counts, nruns, cap, s):
        L.check(lib.ymi_mask_rle_f32(masks.data_ptr(), N, h, w, counts.data_ptr(), nruns.data_ptr(), cap, s), 'mask_rle')
    return _encode_strings(launch, N, h, w, masks.device, cap)


def rle_encode_lowres(masks_lo: torch.Tensor, h: int, w: int, thresh: float = 0.5, cap: int = 4096):
    """masks_lo: [N,ph,pw] float32 CUDA tensor — the cropped sigmoid masks at prototype resolution (what `postprocess`
    holds before its upsample, output_utils.py:69-88).  Returns the records rle_encode would return for
    `F.interpolate(masks_lo, (h, w), 'bilinear', align_corners=False) > thresh` without materialising those masks: upsample,
    threshold and run-length encoding are one kernel (same arithmetic as the upsample kernel, byte-identical strings)."""
    L.require_cuda(masks_lo, 'masks_lo')
    if masks_lo.dim() != 3:
        raise ValueError('expected [N,ph,pw] masks, got %s' % (tuple(masks_lo.shape),))
    N, ph, pw = (int(v) for v in masks_lo.shape)
    if N == 0:
        return []
    masks_lo = masks_lo.detach().to(torch.float32).contiguous()
    lib = L.lib()

    def launch(counts, nruns, cap, s):
        L.check(lib.ymi_mask_rle_upsampled_f32(masks_lo.data_ptr(), N, ph, pw, int(h), int(w), C.c_float(thresh),
                                               counts.data_ptr(), nruns.data_ptr(), cap, s), 'mask_rle_upsampled')
    return _encode_strings(launch, N, int(h), int(w), masks_lo.device, cap)


def rle_counts(masks: torch.Tensor, cap: int = 4096):
    """The raw run lengths (list of int lists) — pycocotools' uncompressed RLE."""
    L.require_cuda(masks, 'masks')
    N, h, w = (int(v) for v in masks.shape)
    if N == 0:
        return []
    masks = masks.detach().to(torch.float32).contiguous()
    lib = L.lib()
    with torch.cuda.device(masks.device):
        while True:
            counts = torch.empty(N, cap, dtype=torch.int32, device=masks.device)
            nruns = torch.empty(N, dtype=torch.int32, device=masks.device)
            L.check(lib.ymi_mask_rle_f32(masks.data_ptr(), N, h, w, counts.data_ptr(), nruns.data_ptr(), cap, L.stream_ptr()),
                    'mask_rle')
            n = nruns.cpu()
            if int(n.max()) <= cap:
                break
            cap = 1 << (int(n.max()) - 1).bit_length()
        c = counts.cpu().numpy().view(np.uint32)
    return [c[i, :int(n[i])].tolist() for i in range(N)]


class Detections:
    """eval.py:300-340.  `bbox_path` / `mask_path` replace the reference's global `args.bbox_det_file` /
    `args.mask_det_file` (eval.py:91-94)."""

    def __init__(self, bbox_path='results/bbox_detections.json', mask_path='results/mask_detections.json', label_map=None):
        self.bbox_data = []
        self.mask_data = []
        self.bbox_path, self.mask_path = bbox_path, mask_path
        lm = label_map if label_map is not None else get_label_map()
        self._coco_cats = {v - 1: k for k, v in lm.items()}       # prep_coco_cats, eval.py:283-288

    def get_coco_cat(self, transformed_cat_id):
        return self._coco_cats[int(transformed_cat_id)]

    def add_bbox(self, image_id: int, category_id: int, bbox, score: float):
        """bbox = (x1, y1, x2, y2); stored as [x, y, w, h] rounded to one decimal (eval.py:306-318)."""
        bbox = [bbox[0], bbox[1], bbox[2] - bbox[0], bbox[3] - bbox[1]]
        bbox = [round(float(x) * 10) / 10 for x in bbox]
        self.bbox_data.append({'image_id': int(image_id), 'category_id': self.get_coco_cat(category_id), 'bbox': bbox,
                               'score': float(score)})

    def add_mask(self, image_id: int, category_id: int, segmentation, score: float):
        """segmentation: one full-size [h,w] mask (CUDA tensor), eval.py:320-330."""
        if not torch.is_tensor(segmentation):
            raise TypeError('yolact_amd encodes masks on the GPU: pass the CUDA tensor postprocess returned, not a numpy '
                            'copy (there is no CPU path)')
        self.add_masks(image_id, [category_id], segmentation[None], [score])

    def add_masks(self, image_id: int, category_ids, masks: torch.Tensor, scores):
        """All masks of an image at once: one RLE launch pair, same records as add_mask called per detection."""
        for cat, rle, sc in zip(category_ids, rle_encode(masks), scores):
            self.mask_data.append({'image_id': int(image_id), 'category_id': self.get_coco_cat(cat), 'segmentation': rle,
                                   'score': float(sc)})

    def add_image(self, image_id: int, classes, boxes, box_scores, masks: torch.Tensor, mask_scores=None):
        """The loop of prep_metrics under --output_coco_json (eval.py:420-429) for one image; `masks` stays on the GPU."""
        classes = [int(c) for c in (classes.tolist() if torch.is_tensor(classes) else classes)]
        boxes = boxes.detach().cpu().numpy() if torch.is_tensor(boxes) else np.asarray(boxes)
        to_list = lambda v: v.detach().cpu().tolist() if torch.is_tensor(v) else list(v)   # noqa: E731
        box_scores = to_list(box_scores)
        mask_scores = box_scores if mask_scores is None else to_list(mask_scores)
        keep = [i for i in range(len(classes)) if (boxes[i, 3] - boxes[i, 1]) * (boxes[i, 2] - boxes[i, 0]) > 0]
        for i in keep:
            self.add_bbox(image_id, classes[i], boxes[i, :], box_scores[i])
        if keep:
            idx = torch.as_tensor(keep, device=masks.device)
            self.add_masks(image_id, [classes[i] for i in keep], masks.index_select(0, idx), [mask_scores[i] for i in keep])

    def add_records(self, image_id: int, classes, boxes, box_scores, rles, mask_scores=None):
        """add_image for masks that are already COCO RLE records (layers.output_utils.postprocess_rle): the same filter
        (boxes of positive area, eval.py:426) and the same JSON records, without full-resolution masks."""
        classes = [int(c) for c in (classes.tolist() if torch.is_tensor(classes) else classes)]
        boxes = boxes.detach().cpu().numpy() if torch.is_tensor(boxes) else np.asarray(boxes)
        to_list = lambda v: v.detach().cpu().tolist() if torch.is_tensor(v) else list(v)   # noqa: E731
        box_scores = to_list(box_scores)
        mask_scores = box_scores if mask_scores is None else to_list(mask_scores)
        for i in range(len(classes)):
            if (boxes[i, 3] - boxes[i, 1]) * (boxes[i, 2] - boxes[i, 0]) > 0:
                self.add_bbox(image_id, classes[i], boxes[i, :], box_scores[i])
                self.mask_data.append({'image_id': int(image_id), 'category_id': self.get_coco_cat(classes[i]),
                                       'segmentation': rles[i], 'score': float(mask_scores[i])})

    def dump(self):
        for data, path in ((self.bbox_data, self.bbox_path), (self.mask_data, self.mask_path)):
            with open(path, 'w') as f:
                json.dump(data, f)
