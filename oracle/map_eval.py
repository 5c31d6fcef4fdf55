"""CPU restatement of the reference's mAP evaluator.  *** TEST INFRASTRUCTURE ONLY ***

eval.py:386-470 (`prep_metrics`, non-crowd part), :472-548 (`APDataObject`), :1005-1031 (`calc_map`);
layers/box_utils.py:54-80 (`jaccard`), :98-113 (`mask_iou`).  Used to state the BASELINE metric's "mask mAP parity"
on pseudo ground truth: the same GT is scored against the reference's detections (golden) and against
the HIP path's detections.  Pinned by tests/golden/map.npz, which holds the tables the reference's OWN prep_metrics /
calc_map produced in the build container (oracle/make_golden_map.py).
"""
from collections import OrderedDict

import numpy as np
import torch

IOU_THRESHOLDS = [x / 100 for x in range(50, 100, 5)]      # eval.py:31


def mask_iou(masks_a, masks_b):
    """box_utils.py:98-113: masks [a, n] / [b, n] float -> IoU [a, b]."""
    intersection = masks_a @ masks_b.t()
    area_a = masks_a.sum(dim=1).unsqueeze(1)
    area_b = masks_b.sum(dim=1).unsqueeze(0)
    return intersection / (area_a + area_b - intersection)


def jaccard(box_a, box_b):
    """box_utils.py:54-80 (non-batched, non-crowd): point-form boxes [A,4], [B,4] -> IoU [A,B]."""
    mx = torch.min(box_a[:, None, 2:], box_b[None, :, 2:])
    mn = torch.max(box_a[:, None, :2], box_b[None, :, :2])
    inter = torch.clamp(mx - mn, min=0).prod(2)
    area_a = ((box_a[:, 2] - box_a[:, 0]) * (box_a[:, 3] - box_a[:, 1])).unsqueeze(1)
    area_b = ((box_b[:, 2] - box_b[:, 0]) * (box_b[:, 3] - box_b[:, 1])).unsqueeze(0)
    return inter / (area_a + area_b - inter)


class APDataObject:
    """eval.py:472-548."""

    def __init__(self):
        self.data_points = []
        self.num_gt_positives = 0

    def push(self, score, is_true):
        self.data_points.append((score, is_true))

    def add_gt_positives(self, n):
        self.num_gt_positives += n

    def is_empty(self):
        return len(self.data_points) == 0 and self.num_gt_positives == 0

    def get_ap(self):
        if self.num_gt_positives == 0:
            return 0
        self.data_points.sort(key=lambda x: -x[0])
        precisions, recalls = [], []
        num_true = num_false = 0
        for datum in self.data_points:
            if datum[1]:
                num_true += 1
            else:
                num_false += 1
            precisions.append(num_true / (num_true + num_false))
            recalls.append(num_true / self.num_gt_positives)
        for i in range(len(precisions) - 1, 0, -1):
            if precisions[i] > precisions[i - 1]:
                precisions[i - 1] = precisions[i]
        y_range = [0] * 101
        x_range = np.array([x / 100 for x in range(101)])
        indices = np.searchsorted(np.array(recalls), x_range, side='left')
        for bar_idx, precision_idx in enumerate(indices):
            if precision_idx < len(precisions):
                y_range[bar_idx] = precisions[precision_idx]
        return sum(y_range) / len(y_range)


def new_ap_data(num_classes):
    """eval.py:884-887."""
    return {'box': [[APDataObject() for _ in range(num_classes)] for _ in IOU_THRESHOLDS],
            'mask': [[APDataObject() for _ in range(num_classes)] for _ in IOU_THRESHOLDS]}


def prep_metrics(ap_data, classes, scores, boxes, masks, gt, gt_masks, h, w):
    """eval.py:386-470 without the crowd / COCO-json branches.  classes [n] int, scores [n] (or 2-list: box scores,
    mask scores), boxes [n,4] absolute pixels, masks [n,h,w] {0,1}; gt [g,5] relative xyxy + class; gt_masks [g,h,w]."""
    gt_boxes = torch.tensor(np.asarray(gt)[:, :4], dtype=torch.float32).clone()
    gt_boxes[:, [0, 2]] *= w
    gt_boxes[:, [1, 3]] *= h
    gt_classes = list(np.asarray(gt)[:, 4].astype(int))
    gt_masks = torch.as_tensor(np.asarray(gt_masks), dtype=torch.float32).view(-1, h * w)
    if classes.shape[0] == 0:
        return
    classes = list(classes.cpu().numpy().astype(int))
    if isinstance(scores, (list, tuple)):
        box_scores = list(scores[0].cpu().numpy().astype(float))
        mask_scores = list(scores[1].cpu().numpy().astype(float))
    else:
        box_scores = mask_scores = list(scores.cpu().numpy().astype(float))
    masks = masks.reshape(-1, h * w).float().cpu()
    boxes = boxes.cpu()
    num_pred, num_gt = len(classes), len(gt_classes)
    mask_iou_cache = mask_iou(masks, gt_masks)
    bbox_iou_cache = jaccard(boxes.float(), gt_boxes.float())
    box_indices = sorted(range(num_pred), key=lambda i: -box_scores[i])
    mask_indices = sorted(box_indices, key=lambda i: -mask_scores[i])
    iou_types = [('box', lambda i, j: bbox_iou_cache[i, j].item(), lambda i: box_scores[i], box_indices),
                 ('mask', lambda i, j: mask_iou_cache[i, j].item(), lambda i: mask_scores[i], mask_indices)]
    for _class in set(classes + gt_classes):
        num_gt_for_class = sum(1 for x in gt_classes if x == _class)
        for iou_idx, iou_threshold in enumerate(IOU_THRESHOLDS):
            for iou_type, iou_func, score_func, indices in iou_types:
                gt_used = [False] * len(gt_classes)
                ap_obj = ap_data[iou_type][iou_idx][_class]
                ap_obj.add_gt_positives(num_gt_for_class)
                for i in indices:
                    if classes[i] != _class:
                        continue
                    max_iou_found, max_match_idx = iou_threshold, -1
                    for j in range(num_gt):
                        if gt_used[j] or gt_classes[j] != _class:
                            continue
                        iou = iou_func(i, j)
                        if iou > max_iou_found:
                            max_iou_found, max_match_idx = iou, j
                    if max_match_idx >= 0:
                        gt_used[max_match_idx] = True
                        ap_obj.push(score_func(i), True)
                    else:
                        ap_obj.push(score_func(i), False)


def calc_map(ap_data, num_classes):
    """eval.py:1005-1031 (without printing / rounding)."""
    aps = [{'box': [], 'mask': []} for _ in IOU_THRESHOLDS]
    for _class in range(num_classes):
        for iou_idx in range(len(IOU_THRESHOLDS)):
            for iou_type in ('box', 'mask'):
                ap_obj = ap_data[iou_type][iou_idx][_class]
                if not ap_obj.is_empty():
                    aps[iou_idx][iou_type].append(ap_obj.get_ap())
    all_maps = {'box': OrderedDict(), 'mask': OrderedDict()}
    for iou_type in ('box', 'mask'):
        all_maps[iou_type]['all'] = 0
        for i, threshold in enumerate(IOU_THRESHOLDS):
            m = sum(aps[i][iou_type]) / len(aps[i][iou_type]) * 100 if len(aps[i][iou_type]) > 0 else 0
            all_maps[iou_type][int(threshold * 100)] = m
        all_maps[iou_type]['all'] = sum(all_maps[iou_type].values()) / (len(all_maps[iou_type].values()) - 1)
    return all_maps


def pseudo_gt(classes, scores, boxes, masks, w, h, max_gt=10, max_overlap=0.3):
    """SURVEY §8(d) pseudo ground truth: the highest-scoring, mutually non-overlapping detections of the REFERENCE
    (mask IoU <= max_overlap, non-empty mask, non-degenerate box).  Returns gt [g,5] (relative xyxy + class), gt_masks."""
    s = scores[0] if isinstance(scores, (list, tuple)) else scores
    order = torch.argsort(s, descending=True, stable=True).tolist()
    flat = masks.reshape(masks.shape[0], -1).float()
    chosen = []
    for i in order:
        if flat[i].sum() < 4 or (boxes[i, 2] - boxes[i, 0]) * (boxes[i, 3] - boxes[i, 1]) <= 0:
            continue
        if chosen and mask_iou(flat[i:i + 1], flat[chosen]).max().item() > max_overlap:
            continue
        chosen.append(i)
        if len(chosen) == max_gt:
            break
    gt = np.zeros((len(chosen), 5), dtype=np.float64)
    for r, i in enumerate(chosen):
        b = boxes[i].double().numpy()
        gt[r] = [b[0] / w, b[1] / h, b[2] / w, b[3] / h, int(classes[i])]
    return gt, masks[chosen].numpy().astype(np.uint8)


def perturb_gt(gt, gt_masks, w, h, seed=0):
    """Make the pseudo ground truth DISCRIMINATING.  pseudo_gt() returns the reference's own detections, so the
    reference scores IoU = 1 against every GT object and its AP table is flat from IoU .50 to .95: a mask that drifted by 5 % IoU
    would change nothing.  Here every GT object r is displaced so that the detection it came from overlaps it with a chosen IoU
    t_r, the t_r spread quasi-uniformly over [0.52, 0.98] (golden-ratio sequence, deterministic in (seed, r)): boxes are shifted
    along x by s * width with s = (1 - t) / (1 + t) (two equal boxes shifted by s * width overlap with IoU (1 - s) / (1 + s) = t),
    masks by the same fraction of their own extent (zero fill).  The reference's table then FALLS from .50 to .95 as objects drop
    below the threshold one by one, and an evaluated path only reproduces it if its boxes / masks have the reference's IoUs — to
    within the spacing of the t_r — at every threshold.  Returns (gt', gt_masks')."""
    gt = np.array(gt, dtype=np.float64, copy=True)
    gm = np.array(gt_masks, copy=True)
    out = np.zeros_like(gm)
    for r in range(gt.shape[0]):
        t = 0.52 + 0.46 * (((r + 1 + 7 * seed) * 0.6180339887498949) % 1.0)
        s = (1.0 - t) / (1.0 + t)
        sign = 1.0 if (r + seed) % 2 == 0 else -1.0
        bw = gt[r, 2] - gt[r, 0]
        dx = sign * s * bw
        if gt[r, 0] + dx < 0.0 or gt[r, 2] + dx > 1.0:        # keep the box inside the image: shift the other way
            dx = -dx
        gt[r, 0] += dx
        gt[r, 2] += dx
        cols = np.nonzero(gm[r].any(axis=0))[0]
        if cols.size:
            px = int(round(s * (cols[-1] - cols[0] + 1))) * (1 if dx >= 0 else -1)
            if px > 0:
                out[r][:, px:] = gm[r][:, :-px]
            elif px < 0:
                out[r][:, :px] = gm[r][:, -px:]
            else:
                out[r] = gm[r]
    return gt, out


# ----------------------------------------------------------------------------------------------
# prep_display, GPU half — eval.py:135-209,228 with undo_transform=False
COLORS = ((244, 67, 54), (233, 30, 99), (156, 39, 176), (103, 58, 183), (63, 81, 181), (33, 150, 243), (3, 169, 244),
          (0, 188, 212), (0, 150, 136), (76, 175, 80), (139, 195, 74), (205, 220, 57), (255, 235, 59), (255, 193, 7),
          (255, 152, 0), (255, 87, 34), (121, 85, 72), (158, 158, 158), (96, 125, 139))      # data/config.py:6-24


def prep_display_masks(post, img, mask_alpha=0.45, top_k=5, score_threshold=0, class_color=False):
    """post = postprocess() outputs (classes, scores, boxes, masks [n,h,w]); img [h,w,3] float 0..255.
    eval.py:141-142 (img/255), :155-165 (top-k by score, threshold), :169-184 (colours, BGR swap), :189-209 (cumprod
    compositing), :228 (uint8)."""
    img_gpu = img / 255.0
    classes, scores, boxes, masks = post
    idx = scores.argsort(0, descending=True)[:top_k]
    masks = masks[idx]
    classes, scores = classes[idx].numpy(), scores[idx].numpy()
    n = min(top_k, classes.shape[0])
    for j in range(n):
        if scores[j] < score_threshold:
            n = j
            break
    if n > 0:
        masks = masks[:n, :, :, None]
        cols = []
        for j in range(n):
            c = COLORS[(classes[j] * 5 if class_color else j * 5) % len(COLORS)]
            cols.append(torch.Tensor((c[2], c[1], c[0])).float() / 255.)
        colors = torch.cat([c.view(1, 1, 1, 3) for c in cols], dim=0)
        masks_color = masks.repeat(1, 1, 1, 3) * colors * mask_alpha
        inv_alph_masks = masks * (-mask_alpha) + 1
        masks_color_summand = masks_color[0]
        if n > 1:
            inv_alph_cumul = inv_alph_masks[:(n - 1)].cumprod(dim=0)
            masks_color_cumul = masks_color[1:] * inv_alph_cumul
            masks_color_summand += masks_color_cumul.sum(dim=0)
        img_gpu = img_gpu * inv_alph_masks.prod(dim=0) + masks_color_summand
    return (img_gpu * 255).byte()
