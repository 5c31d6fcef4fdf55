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
