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
    """data/coco.py:19-51: COCO annotation dicts -> [[xmin, ymin, xmax, ymax, label_idx], ...] in relative coordinates."""

    def __init__(self):
        self.label_map = get_label_map()

    def __call__(self, target, width, height):
        scale = np.array([width, height, width, height])
        res = []
        for obj in target:
            if 'bbox' in obj:
                bbox = obj['bbox']
                label_idx = obj['category_id']
                if label_idx >= 0:
                    label_idx = self.label_map[label_idx] - 1
                final_box = list(np.array([bbox[0], bbox[1], bbox[0] + bbox[2], bbox[1] + bbox[3]]) / scale)
                final_box.append(label_idx)
                res += [final_box]
            else:
                print('No bbox found for object ', obj)
        return res


class COCODetection(torch.utils.data.Dataset):
    """data/coco.py:54-212.  `transform` is called like the reference's (img, masks, boxes, labels-dict); with
    yolact_amd.utils.augmentations.BaseTransform the image goes in and comes out as a device tensor."""

    def __init__(self, image_path, info_file, transform=None, target_transform=None, dataset_name='MS COCO', has_gt=True,
                 device=None):
        self.root = image_path
        self.coco = COCOIndex(info_file)
        self.ids = list(self.coco.imgToAnns.keys())
        if len(self.ids) == 0 or not has_gt:
            self.ids = list(self.coco.imgs.keys())
        self.transform = transform
        self.target_transform = COCOAnnotationTransform()      # data/coco.py:86 ignores the argument as well
        self.name = dataset_name
        self.has_gt = has_gt
        self.device = device

    def __getitem__(self, index):
        im, gt, masks, h, w, num_crowds = self.pull_item(index)
        return im, (gt, masks, num_crowds)

    def __len__(self):
        return len(self.ids)

    def _path(self, img_id):
        file_name = self.coco.loadImgs(img_id)[0]['file_name']
        if file_name.startswith('COCO'):             # COCO2014 names -> the %012d.jpg the download script writes
            file_name = file_name.split('_')[-1]
        path = osp.join(self.root, file_name)
        assert osp.exists(path), 'Image path does not exist: {}'.format(path)
        return path

    def pull_item(self, index):
        """-> (image [3,S,S] float32 on the GPU, target [n,5] float ndarray, masks [n,h,w] uint8 ndarray, height, width,
        num_crowds) — data/coco.py:100-176."""
        img_id = self.ids[index]
        if self.has_gt:
            ann_ids = self.coco.getAnnIds(imgIds=img_id)
            target = [x for x in self.coco.loadAnns(ann_ids) if x['image_id'] == img_id]
        else:
            target = []
        crowd = [x for x in target if ('iscrowd' in x and x['iscrowd'])]
        target = [x for x in target if not ('iscrowd' in x and x['iscrowd'])]
        num_crowds = len(crowd)
        for x in crowd:
            x['category_id'] = -1
        target += crowd                              # crowd annotations at the end of the array

        img = jpeg.imread(self._path(img_id), self.device)
        height, width, _ = img.shape

        masks = None
        if len(target) > 0:
            masks = np.stack([ann_to_mask(obj, height, width) for obj in target], axis=0)
        if self.target_transform is not None and len(target) > 0:
            target = self.target_transform(target, width, height)

        if self.transform is not None:
            if len(target) > 0:
                target = np.array(target)
                img, masks, boxes, labels = self.transform(img, masks, target[:, :4],
                                                           {'num_crowds': num_crowds, 'labels': target[:, 4]})
                num_crowds = labels['num_crowds']
                labels = labels['labels']
                target = np.hstack((boxes, np.expand_dims(labels, axis=1)))
            else:
                img, _, _, _ = self.transform(img, np.zeros((1, height, width), dtype=np.float64),
                                              np.array([[0, 0, 1, 1]]), {'num_crowds': 0, 'labels': np.array([0])})
                masks = None
                target = None

        if target is not None and not isinstance(target, list) and target.shape[0] == 0:
            print('Warning: Augmentation output an example with no ground truth. Resampling...')
            return self.pull_item(random.randint(0, len(self.ids) - 1))

        if img.dim() == 3 and img.shape[2] == 3:     # HWC (the transform's convention) -> CHW
            img = img.permute(2, 0, 1)
        return img, target, masks, height, width, num_crowds

    def pull_image(self, index):
        """The decoded image (uint8 BGR [h,w,3], on the GPU) — data/coco.py:178-192 returns cv2.imread's array."""
        return jpeg.imread(self._path(self.ids[index]), self.device)

    def pull_anno(self, index):
        img_id = self.ids[index]
        return self.coco.loadAnns(self.coco.getAnnIds(imgIds=img_id))

    def __repr__(self):
        return 'Dataset %s\n    Number of datapoints: %d\n    Root Location: %s\n' % (self.__class__.__name__, len(self), self.root)
