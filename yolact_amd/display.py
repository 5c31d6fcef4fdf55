"""`prep_display`'s GPU half on device (eval.py:135-209,228 with undo_transform=False, the
evalvideo path eval.py:692-700): postprocess -> top-k by score -> alpha-composite the instance masks onto the frame ->
uint8.  Boxes / labels are drawn by cv2 on the CPU in the reference and stay there (cv2 is not a dependency here): the
function returns what that code needs.
"""
from __future__ import annotations

import torch

from . import _lib as L
from .config import active_cfg
from .layers.output_utils import postprocess

COLORS = ((244, 67, 54), (233, 30, 99), (156, 39, 176), (103, 58, 183), (63, 81, 181), (33, 150, 243), (3, 169, 244),
          (0, 188, 212), (0, 150, 136), (76, 175, 80), (139, 195, 74), (205, 220, 57), (255, 235, 59), (255, 193, 7),
          (255, 152, 0), (255, 87, 34), (121, 85, 72), (158, 158, 158), (96, 125, 139))      # data/config.py:6-24


def prep_display(dets_out, img, mask_alpha=0.45, top_k=5, score_threshold=0, class_color=False, crop_masks=True,
                 display_masks=True):
    """img: [h, w, 3] float frame, 0..255, on the GPU (the BGR frame evalvideo feeds, eval.py:692-700).
    Returns (img_u8 [h,w,3] uint8 on the GPU, classes, scores, boxes as numpy arrays of the drawn detections)."""
    L.require_cuda(img, 'frame')
    cfg = active_cfg()
    h, w, _ = img.shape
    save = getattr(cfg, 'rescore_bbox', False)
    cfg.rescore_bbox = True                                                    # eval.py:147-152
    try:
        t = postprocess(dets_out, w, h, crop_masks=crop_masks, score_threshold=score_threshold)
    finally:
        cfg.rescore_bbox = save
    img = img.detach().to(torch.float32).contiguous()
    out = torch.empty(h, w, 3, dtype=torch.uint8, device=img.device)
    lib = L.lib()
    if t[0].numel() == 0:
        with torch.cuda.device(img.device):
            L.check(lib.ymi_composite_masks_u8(img.data_ptr(), None, None, 0, h, w, mask_alpha, out.data_ptr(), L.stream_ptr()))
        return out, None, None, None
    idx = t[1].argsort(0, descending=True)[:top_k]                              # eval.py:155
    masks = t[3][idx]
    classes, scores, boxes = [x[idx].cpu().numpy() for x in t[:3]]
    n = min(top_k, classes.shape[0])
    for j in range(n):
        if scores[j] < score_threshold:
            n = j
            break
    if not (display_masks and cfg.eval_mask_branch):
        n_draw = 0
    else:
        n_draw = n
    cols = []
    for j in range(n_draw):
        c = COLORS[(classes[j] * 5 if class_color else j * 5) % len(COLORS)]
        cols.append((c[2], c[1], c[0]))                                        # not undo_transform: swap (eval.py:177-179)
    colors = (torch.tensor(cols, dtype=torch.float32, device=img.device) / 255.) if n_draw else None
    m = masks[:n_draw].contiguous().float() if n_draw else None
    with torch.cuda.device(img.device):
        L.check(lib.ymi_composite_masks_u8(img.data_ptr(), m.data_ptr() if n_draw else None,
                                           colors.data_ptr() if n_draw else None, n_draw, h, w, mask_alpha,
                                           out.data_ptr(), L.stream_ptr()), 'ymi_composite_masks_u8')
    return out, classes[:n], scores[:n], boxes[:n]
