"""`COCODetection` / `COCOAnnotationTransform` — data/coco.py:13-176 — and the slice of `pycocotools.coco.COCO` they use.

The reference's eval loop reads every validation item through `dataset.pull_item(i)` (eval.py:936):
    cv2.imread -> annToMask per object -> target transform -> BaseTransform -> torch tensor.
Here the image half runs on the GPU (data/jpeg.py: host entropy decode, device reconstruction; BaseTransform = the
FastBaseTransform kernel), the annotation half is host code like the reference's: the JSON index is plain Python, the
polygon / RLE rasteriser is native (csrc/coco_host.cpp, pycocotools' maskApi.c restated).  Same constructor arguments,
same return tuple, same crowd ordering, same "no ground truth -> resample" rule; the image tensor is on the GPU (the
reference returns a CPU tensor that eval.py then moves with `.cuda()`, eval.py:938-940).
"""
from __future__ import annotations

import ctypes as C
import json
import os.path as osp
import random
from collections import defaultdict

import numpy as np
import torch

from .. import _lib as L
from ..coco import get_label_map
from . import jpeg


class COCOIndex:
    """The part of pycocotools.coco.COCO that data/coco.py touches: `imgs`, `anns`, `imgToAnns`, `getAnnIds(imgIds=)`,
    `loadAnns`, `loadImgs`, `annToMask` (PythonAPI/pycocotools/coco.py createIndex / getAnnIds / annToRLE)."""

    def __init__(self, annotation_file=None):
        self.dataset, self.anns, self.imgs, self.cats = {}, {}, {}, {}
        self.imgToAnns = defaultdict(list)
        if annotation_file is not None:
            with open(annotation_file) as f:
                self.dataset = json.load(f)
            if not isinstance(self.dataset, dict):
                raise ValueError('annotation file format %s not supported' % type(self.dataset))
            self.createIndex()

    def createIndex(self):
        for ann in self.dataset.get('annotations', []):
            self.imgToAnns[ann['image_id']].append(ann)
            self.anns[ann['id']] = ann
        for img in self.dataset.get('images', []):
            self.imgs[img['id']] = img
        for cat in self.dataset.get('categories', []):
            self.cats[cat['id']] = cat

    def getAnnIds(self, imgIds=()):
        imgIds = imgIds if isinstance(imgIds, (list, tuple)) else [imgIds]
        if len(imgIds) == 0:
            return [a['id'] for a in self.dataset.get('annotations', [])]
        return [a['id'] for i in imgIds if i in self.imgToAnns for a in self.imgToAnns[i]]

    def loadAnns(self, ids=()):
        return [self.anns[i] for i in ids] if isinstance(ids, (list, tuple)) else [self.anns[ids]]

    def loadImgs(self, ids=()):
        return [self.imgs[i] for i in ids] if isinstance(ids, (list, tuple)) else [self.imgs[ids]]

    def annToMask(self, ann) -> np.ndarray:
        """uint8 [h,w] of {0,1} (pycocotools returns the same values in a Fortran-ordered array)."""
        t = self.imgs[ann['image_id']]
        return ann_to_mask(ann, t['height'], t['width'])


def ann_to_mask(ann, h: int, w: int) -> np.ndarray:
    """pycocotools annToRLE + decode: polygon list (union) | uncompressed RLE | compressed RLE string."""
    lib = L.lib()
    mask = np.zeros((h, w), dtype=np.uint8)
    seg = ann['segmentation']
    if isinstance(seg, list):
        for poly in seg:
            xy = np.ascontiguousarray(poly, dtype=np.float64)
            if xy.size < 2 or xy.size % 2:
                raise ValueError('polygon with %d coordinates' % xy.size)
            L.check(lib.ymi_coco_poly_fill_u8(xy.ctypes.data, xy.size // 2, h, w, mask.ctypes.data), 'ymi_coco_poly_fill_u8')
        return mask
    size = seg.get('size')
    if size is not None and (int(size[0]) != h or int(size[1]) != w):
        raise ValueError('RLE size %s does not match the image (%d, %d)' % (size, h, w))
    counts = seg['counts']
    if isinstance(counts, list):
        c = np.ascontiguousarray(counts, dtype=np.uint32)
        L.check(lib.ymi_coco_rle_fill_u8(c.ctypes.data, c.size, h, w, mask.ctypes.data), 'ymi_coco_rle_fill_u8')
    else:
        s = counts.encode('ascii') if isinstance(counts, str) else bytes(counts)
        L.check(lib.ymi_coco_rle_string_fill_u8(s, len(s), h, w, mask.ctypes.data), 'ymi_coco_rle_string_fill_u8')
    return mask


class COCOAnnotationTransform:
    """data/coco.py:19-51: COCO annotation dicts -> one row [xmin, ymin, xmax, ymax, label_idx] per annotation that has a
    'bbox', corners relative to the image size; label_idx = label_map[category_id] - 1, crowds (category_id < 0) keep -1."""

    def __init__(self):
        self.label_map = get_label_map()

    def __call__(self, target, width, height):
        rows = []
        size = np.array([width, height, width, height])
        for ann in target:
            box = ann.get('bbox')
            if box is None:
                print('No bbox found for object ', ann)
                continue
            x, y, bw, bh = box
            cat = ann['category_id']
            label = self.label_map[cat] - 1 if cat >= 0 else cat
            rows.append(list(np.array([x, y, x + bw, y + bh]) / size) + [label])
        return rows


def _split_crowds(anns):
    """Crowd annotations (iscrowd truthy) go to the END of the list and get category_id -1 (data/coco.py:117-128: both in
    training and in evaluation they are neutral regions, and eval.py slices them off by count)."""
    crowd = [a for a in anns if a.get('iscrowd')]
    solid = [a for a in anns if not a.get('iscrowd')]
    for a in crowd:
        a['category_id'] = -1
    return solid + crowd, len(crowd)


class COCODetection(torch.utils.data.Dataset):
    """data/coco.py:54-212 with the reference's constructor arguments and return values.  `transform` is called the way
    the reference calls it — (img, masks, boxes, {'num_crowds', 'labels'}) — and with
    yolact_amd.utils.augmentations.BaseTransform the image goes in and comes out as a device tensor."""

    def __init__(self, image_path, info_file, transform=None, target_transform=None, dataset_name='MS COCO', has_gt=True,
                 device=None):
        self.root = image_path
        self.coco = COCOIndex(info_file)
        annotated = list(self.coco.imgToAnns.keys())
        self.ids = annotated if (annotated and has_gt) else list(self.coco.imgs.keys())
        self.transform = transform
        self.target_transform = COCOAnnotationTransform()      # data/coco.py:86 ignores the argument as well
        self.name = dataset_name
        self.has_gt = has_gt
        self.device = device

    def __len__(self):
        return len(self.ids)

    def __getitem__(self, index):
        image, boxes, masks, _, _, num_crowds = self.pull_item(index)
        return image, (boxes, masks, num_crowds)

    def _path(self, img_id):
        name = self.coco.loadImgs(img_id)[0]['file_name']
        if name.startswith('COCO'):                  # COCO2014 "COCO_val2014_%012d.jpg" -> the "%012d.jpg" the download script writes
            name = name.rsplit('_', 1)[-1]
        path = osp.join(self.root, name)
        assert osp.exists(path), 'Image path does not exist: {}'.format(path)
        return path

    def pull_item(self, index):
        """-> (image [3,S,S] float32 on the GPU, target [n,5] float ndarray, masks [n,h,w] uint8 ndarray, height, width,
        num_crowds) — data/coco.py:100-176.  No annotations (has_gt=False): target and masks are None (what the
        reference's else-branch assigns, data/coco.py:163-167; its next line then dereferences the None and raises — an
        evident slip, not reproduced)."""
        img_id = self.ids[index]
        anns, num_crowds = [], 0
        if self.has_gt:
            mine = self.coco.loadAnns(self.coco.getAnnIds(imgIds=img_id))
            anns, num_crowds = _split_crowds([a for a in mine if a['image_id'] == img_id])

        img = jpeg.imread(self._path(img_id), self.device)                     # uint8 BGR [h,w,3], device
        height, width = int(img.shape[0]), int(img.shape[1])

        masks = target = None
        if anns:
            # data/coco.py:146-148: annToMask rasterises at the JSON's (height, width); the flat masks are then reshaped to
            # the DECODED size (they differ only for EXIF-rotated files: the reference carries on, and so does this)
            masks = np.vstack([self.coco.annToMask(a).reshape(-1) for a in anns]).reshape(-1, height, width)
            target = np.array(self.target_transform(anns, width, height))

        if self.transform is not None:
            if target is not None and len(target):
                meta = {'num_crowds': num_crowds, 'labels': target[:, 4]}
                img, masks, boxes, meta = self.transform(img, masks, target[:, :4], meta)
                num_crowds = meta['num_crowds']                               # (the transform may have dropped boxes)
                target = np.hstack((boxes, meta['labels'][:, None]))
                if len(target) == 0:
                    print('Warning: Augmentation output an example with no ground truth. Resampling...')
                    return self.pull_item(random.randint(0, len(self.ids) - 1))
            else:
                # the reference feeds a dummy box through the transform and then reports "no ground truth"
                dummy = {'num_crowds': 0, 'labels': np.array([0])}
                img = self.transform(img, np.zeros((1, height, width)), np.array([[0., 0., 1., 1.]]), dummy)[0]
                masks = target = None

        if img.dim() == 3 and img.shape[-1] == 3:     # HWC (the transform's convention) -> CHW
            img = img.permute(2, 0, 1)
        return img, target, masks, height, width, num_crowds

    def pull_image(self, index):
        """The decoded image (uint8 BGR [h,w,3], on the GPU) — data/coco.py:178-192 returns cv2.imread's array."""
        return jpeg.imread(self._path(self.ids[index]), self.device)

    def pull_anno(self, index):
        return self.coco.loadAnns(self.coco.getAnnIds(imgIds=self.ids[index]))

    def __repr__(self):
        return '%s(%d images, root=%r, has_gt=%s)' % (type(self).__name__, len(self), self.root, self.has_gt)
