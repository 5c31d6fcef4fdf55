"""mask_iou (layers/box_utils.py:98-113, consumer eval.py:376-384,435-440).

The restatement (oracle/map_eval.mask_iou) is pinned through tests/test_map_parity.py (it produces the reference's own
mAP tables).  Here the HIP kernel is compared with it: binary masks => bit-exact (integer partial sums < 2^24); soft
masks => 1e-5 relative (fp32 atomics reorder the K-slice sums)."""
import pytest
import torch

from oracle import map_eval as ME


def _masks(n, h, w, seed, density=0.3):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(n, h, w, generator=g) < density).float()


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(100, 20, 120, 160), (7, 3, 137, 99), (33, 33, 64, 64), (1, 1, 5, 7), (100, 12, 550, 550)])
def test_binary_masks_exact(shape):
    from yolact_amd.layers.box_utils import mask_iou
    A, B, h, w = shape
    a, b = _masks(A, h, w, 1), _masks(B, h, w, 2, 0.5)
    a[0] = 0                                   # empty detection mask: 0 / area_b
    if B > 1:
        b[1] = 0                               # empty GT mask
    if A > 1 and B > 1:
        a[1] = b[0]                            # identical masks: IoU exactly 1
    ref = ME.mask_iou(a.view(A, -1), b.view(B, -1))
    got = mask_iou(a.cuda().view(A, -1), b.cuda().view(B, -1)).cpu()
    assert got.shape == ref.shape
    assert torch.equal(torch.isnan(got), torch.isnan(ref))          # 0/0 for two empty masks, like the reference
    assert torch.equal(torch.nan_to_num(got), torch.nan_to_num(ref))
    ref_c = (a.view(A, -1) @ b.view(B, -1).t()) / a.view(A, -1).sum(1, keepdim=True)
    got_c = mask_iou(a.cuda(), b.cuda(), iscrowd=True).cpu()       # 3-D inputs, crowd variant
    assert torch.equal(torch.nan_to_num(got_c), torch.nan_to_num(ref_c))


@pytest.mark.gpu
def test_soft_masks_and_cpu_path():
    from yolact_amd.layers.box_utils import mask_iou
    g = torch.Generator().manual_seed(3)
    a, b = torch.rand(9, 31 * 17, generator=g), torch.rand(4, 31 * 17, generator=g)
    ref = ME.mask_iou(a, b)
    got = mask_iou(a.cuda(), b.cuda()).cpu()
    assert ((got - ref).abs() / ref.abs()).max().item() < 1e-5
    assert torch.allclose(mask_iou(a, b), ref)                      # CPU tensors: the reference's torch expression


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(100, 20, 120, 160), (7, 3, 137, 99), (1, 9, 5, 7), (40, 12, 550, 550)])
def test_bit_masks_popcount_iou_is_bit_identical(shape):
    """mask_bits + mask_iou_bits == mask_iou on the float masks, bit for bit (both the plain and the crowd form), on sizes that
    are not multiples of 64 pixels."""
    from yolact_amd.layers.box_utils import mask_bits, mask_iou, mask_iou_bits
    A, B, h, w = shape
    a, b = _masks(A, h, w, 11), _masks(B, h, w, 12, 0.5)
    a[0] = 0
    if B > 1:
        b[1] = 0
    ab, bb = mask_bits(a.cuda()), mask_bits(b.cuda())
    assert ab.shape == (A, (h * w + 63) // 64) and ab.dtype == torch.int64
    # the packing itself: bit i of word j = pixel 64 j + i
    flat = a.view(A, -1)[min(1, A - 1)]
    word0 = int(ab[min(1, A - 1), 0].item()) & 0xFFFFFFFFFFFFFFFF
    want0 = sum(int(flat[i].item()) << i for i in range(min(64, flat.numel())))
    assert word0 == want0
    for crowd in (False, True):
        ref = mask_iou(a.cuda(), b.cuda(), iscrowd=crowd).cpu()
        got = mask_iou_bits(ab, bb, iscrowd=crowd).cpu()
        assert torch.equal(torch.isnan(got), torch.isnan(ref)) and torch.equal(torch.nan_to_num(got), torch.nan_to_num(ref))


@pytest.mark.gpu
def test_postprocess_bits_equals_postprocess_masks():
    """postprocess_bits (upsample + threshold straight into bits, no [N,h,w] float masks) reproduces postprocess' masks bit for bit
    on a golden case, and the popcount IoU against bit-packed pseudo ground truth equals mask_iou on the float masks."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from gpu_utils import build_net
    from helpers import case_images, load_golden
    from yolact_amd.layers.box_utils import jaccard, mask_bits, mask_iou, mask_iou_bits
    from yolact_amd.layers.output_utils import postprocess, postprocess_bits
    meta, _ = load_golden('r50_dense')
    net = build_net(meta)
    preds = net(case_images(meta).to('cuda:0'))
    h, w = 317, 403                                   # not a multiple of 64 pixels
    copy = lambda: [{'detection': {k: (v.clone() if torch.is_tensor(v) else v) for k, v in preds[0]['detection'].items()}, 'net': net}]
    classes, scores, boxes, masks = postprocess(copy(), w, h)
    c2, s2, b2, bits = postprocess_bits(copy(), w, h)
    assert torch.equal(classes, c2) and torch.equal(scores, s2) and torch.equal(boxes, b2)
    assert torch.equal(mask_bits(masks), bits)
    gt = masks[:7]
    assert torch.equal(torch.nan_to_num(mask_iou_bits(bits, mask_bits(gt))), torch.nan_to_num(mask_iou(masks, gt)))
    # device jaccard (eval.py:438-440) against the torch expression, reference op order
    bf = boxes.float()
    ref = ME.jaccard(bf.cpu(), bf[:7].cpu())
    got = jaccard(bf, bf[:7]).cpu()
    assert torch.equal(torch.nan_to_num(got), torch.nan_to_num(ref))
    crowd = jaccard(bf, bf[:7], iscrowd=True).cpu()
    inter = ref * 0
    assert crowd.shape == ref.shape and bool((torch.nan_to_num(crowd) >= torch.nan_to_num(ref) - 1e-6).all())
