"""The box helpers eval.py imports from layers.box_utils (eval.py:4): jaccard, center_size, mask_iou, plus
crop / sanitize_coordinates.  These run in eval.py's metric code, downstream of the hot path,
on whatever device the caller's tensors live; they are thin torch expressions with the reference's op order
(box_utils.py:20-113, :327-373) — plumbing, not the accelerated path.
"""
import torch


def center_size(boxes):
    return torch.cat(((boxes[:, 2:] + boxes[:, :2]) / 2, boxes[:, 2:] - boxes[:, :2]), 1)


def point_form(boxes):
    return torch.cat((boxes[:, :2] - boxes[:, 2:] / 2, boxes[:, :2] + boxes[:, 2:] / 2), 1)


def intersect(box_a, box_b):
    mx = torch.min(box_a[:, :, None, 2:], box_b[:, None, :, 2:])
    mn = torch.max(box_a[:, :, None, :2], box_b[:, None, :, :2])
    return torch.clamp(mx - mn, min=0).prod(3)


def jaccard(box_a, box_b, iscrowd: bool = False):
    """box_utils.py:54-80.  Two 2-D GPU tensors (what eval.py's prep_metrics passes: predicted boxes vs ground truth,
    eval.py:376-384,438-440) go through the HIP kernel (csrc/metrics.hip jaccard_k, the reference's op order); batched or CPU
    inputs keep the torch expression."""
    if box_a.dim() == 2 and box_b.dim() == 2 and box_a.is_cuda and box_b.is_cuda and box_a.size(0) > 0 and box_b.size(0) > 0:
        from .. import _lib as L
        a, b = box_a.float().contiguous(), box_b.float().contiguous()
        out = torch.empty(a.size(0), b.size(0), dtype=torch.float32, device=a.device)
        with torch.cuda.device(a.device):
            L.check(L.lib().ymi_jaccard_f32(a.data_ptr(), b.data_ptr(), a.size(0), b.size(0), 1 if iscrowd else 0, out.data_ptr(),
                                            L.stream_ptr()), 'ymi_jaccard_f32')
        return out
    use_batch = box_a.dim() == 3
    if not use_batch:
        box_a, box_b = box_a[None], box_b[None]
    inter = intersect(box_a, box_b)
    area_a = ((box_a[:, :, 2] - box_a[:, :, 0]) * (box_a[:, :, 3] - box_a[:, :, 1])).unsqueeze(2).expand_as(inter)
    area_b = ((box_b[:, :, 2] - box_b[:, :, 0]) * (box_b[:, :, 3] - box_b[:, :, 1])).unsqueeze(1).expand_as(inter)
    out = inter / area_a if iscrowd else inter / (area_a + area_b - inter)
    return out if use_batch else out.squeeze(0)


def mask_iou(masks_a, masks_b, iscrowd=False):
    """box_utils.py:98-113.  GPU tensors go through the HIP kernel (csrc/metrics.hip: split-K MFMA intersection + areas
    in one pass over the masks, exact for 0/1 masks); CPU tensors (eval.py --cuda=False bookkeeping) keep the
    reference's torch expression."""
    a = masks_a.reshape(masks_a.size(0), -1)
    b = masks_b.reshape(masks_b.size(0), -1)
    if a.is_cuda and b.is_cuda and a.size(0) > 0 and b.size(0) > 0:
        import ctypes as C
        from .. import _lib as L
        a, b = a.float().contiguous(), b.float().contiguous()
        A, B, n = a.size(0), b.size(0), a.size(1)
        if b.size(1) != n:
            raise RuntimeError('mask_iou: masks have %d and %d pixels' % (n, b.size(1)))
        ws = torch.empty(A * B + A + B, dtype=torch.float32, device=a.device)
        out = torch.empty(A, B, dtype=torch.float32, device=a.device)
        with torch.cuda.device(a.device):
            L.check(L.lib().ymi_mask_iou_f32(a.data_ptr(), b.data_ptr(), A, B, n, 1 if iscrowd else 0, ws.data_ptr(),
                                             out.data_ptr(), L.stream_ptr()), 'ymi_mask_iou_f32')
        return out
    inter = a @ b.t()
    area_a, area_b = a.sum(1).unsqueeze(1), b.sum(1).unsqueeze(0)
    return inter / (area_a + area_b - inter) if not iscrowd else inter / area_a


def mask_bits(masks):
    """0/1 float masks [N, h, w] (or [N, n]) on the GPU -> int64 [N, ceil(n/64)] bit masks (bit i of word j = pixel 64 j + i):
    the ground-truth side of the popcount mask IoU (eval.py:416 builds the float form; this is 1/32 of its bytes)."""
    from .. import _lib as L
    L.require_cuda(masks, 'masks')
    m = masks.reshape(masks.size(0), -1).float().contiguous()
    N, n = m.shape
    bits = torch.empty(N, (n + 63) // 64, dtype=torch.int64, device=m.device)
    if N:
        with torch.cuda.device(m.device):
            L.check(L.lib().ymi_mask_bits_f32(m.data_ptr(), N, n, bits.data_ptr(), L.stream_ptr()), 'ymi_mask_bits_f32')
    return bits


def mask_iou_bits(bits_a, bits_b, iscrowd=False):
    """mask_iou (box_utils.py:98-113) on bit masks (mask_bits / output_utils.postprocess_bits): intersections and areas are
    popcounts — integers below 2^24 — and the division is the reference's fp32 expression, so the result is bit-identical to
    mask_iou on the corresponding 0/1 float masks while reading 1/32 of the bytes."""
    from .. import _lib as L
    L.require_cuda(bits_a, 'bits_a')
    A, B, W64 = bits_a.size(0), bits_b.size(0), bits_a.size(1)
    if bits_b.size(1) != W64:
        raise RuntimeError('mask_iou_bits: masks have %d and %d words' % (W64, bits_b.size(1)))
    out = torch.empty(A, B, dtype=torch.float32, device=bits_a.device)
    if A and B:
        a, b = bits_a.contiguous(), bits_b.contiguous()
        with torch.cuda.device(a.device):
            L.check(L.lib().ymi_mask_iou_bits(a.data_ptr(), b.data_ptr(), A, B, W64, 1 if iscrowd else 0, out.data_ptr(),
                                              L.stream_ptr()), 'ymi_mask_iou_bits')
    return out


def sanitize_coordinates(_x1, _x2, img_size: int, padding: int = 0, cast: bool = True):
    _x1, _x2 = _x1 * img_size, _x2 * img_size
    if cast:
        _x1, _x2 = _x1.long(), _x2.long()
    x1, x2 = torch.min(_x1, _x2), torch.max(_x1, _x2)
    return torch.clamp(x1 - padding, min=0), torch.clamp(x2 + padding, max=img_size)


def crop(masks, boxes, padding: int = 1):
    h, w, n = masks.size()
    x1, x2 = sanitize_coordinates(boxes[:, 0], boxes[:, 2], w, padding, cast=False)
    y1, y2 = sanitize_coordinates(boxes[:, 1], boxes[:, 3], h, padding, cast=False)
    cols = torch.arange(w, device=masks.device, dtype=x1.dtype).view(1, -1, 1)
    rows = torch.arange(h, device=masks.device, dtype=x1.dtype).view(-1, 1, 1)
    inside = (cols >= x1.view(1, 1, -1)) & (cols < x2.view(1, 1, -1)) & (rows >= y1.view(1, 1, -1)) & (rows < y2.view(1, 1, -1))
    return masks * inside.float()
