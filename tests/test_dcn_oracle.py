"""Pins the CPU DCNv2 restatement (oracle/yolact_oracle.py::dcn_v2_forward).  The reference has no CPU DCN
(external/DCNv2/src/cpu/dcn_v2_cpu.cpp:23) and its CUDA source does not build against this torch, so the pins are:
  1. the reference's own known-answer test, external/DCNv2/test.py:32-67 (zero offsets, mask = sigmoid(0), identity
     centre-tap weights  =>  2 * DCN(x) == x, tolerance 1e-10 there);
  2. the derived KAT offset = 0, mask = 1  =>  F.conv2d (same test file's premise);
  3. an independent scalar re-derivation of dcn_v2_im2col_cuda.cu:25-54,143-193 (pure Python loops, tiny case)
     with random non-integer offsets that leave the image on every side.
"""
import math

import torch
import torch.nn.functional as F

from oracle.yolact_oracle import dcn_v2_forward


def test_reference_kat_zero_offset_identity():
    torch.manual_seed(0)
    N, C, H, W = 2, 8, 7, 6
    x = torch.randn(N, C, H, W)
    w = torch.zeros(C, C, 3, 3)
    for c in range(C):
        w[c, c, 1, 1] = 1.0                                  # conv_identify, external/DCNv2/test.py:18-29
    offset = torch.zeros(N, 18, H, W)
    mask = torch.sigmoid(torch.zeros(N, 9, H, W))
    out = dcn_v2_forward(x, offset, mask, w, torch.zeros(C), 1, 1, 1) * 2
    assert (x - out).abs().max().item() < 1e-10


def test_zero_offset_unit_mask_is_plain_conv():
    torch.manual_seed(1)
    for stride in (1, 2):
        x = torch.randn(2, 6, 9, 11)
        w = torch.randn(5, 6, 3, 3)
        b = torch.randn(5)
        ref = F.conv2d(x, w, b, stride=stride, padding=1)
        Ho, Wo = ref.shape[2:]
        out = dcn_v2_forward(x, torch.zeros(2, 18, Ho, Wo), torch.ones(2, 9, Ho, Wo), w, b, stride, 1, 1)
        assert out.shape == ref.shape
        assert (out - ref).abs().max().item() < 1e-4


def _bilinear_scalar(img, H, W, h, w):
    """dmcn_im2col_bilinear, dcn_v2_im2col_cuda.cu:25-54 (img[H][W] nested lists)."""
    hl, wl = math.floor(h), math.floor(w)
    hh, wh = hl + 1, wl + 1
    lh, lw = h - hl, w - wl
    uh, uw = 1 - lh, 1 - lw
    v1 = img[hl][wl] if (hl >= 0 and wl >= 0) else 0.0
    v2 = img[hl][wh] if (hl >= 0 and wh <= W - 1) else 0.0
    v3 = img[hh][wl] if (hh <= H - 1 and wl >= 0) else 0.0
    v4 = img[hh][wh] if (hh <= H - 1 and wh <= W - 1) else 0.0
    return uh * uw * v1 + uh * lw * v2 + lh * uw * v3 + lh * lw * v4


def test_against_scalar_rederivation_with_wild_offsets():
    torch.manual_seed(2)
    for stride in (1, 2):
        B, C, H, W, Co = 1, 3, 6, 5, 4
        x = torch.randn(B, C, H, W, dtype=torch.float64)
        wt = torch.randn(Co, C, 3, 3, dtype=torch.float64)
        bias = torch.randn(Co, dtype=torch.float64)
        Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
        offset = torch.randn(B, 18, Ho, Wo, dtype=torch.float64) * 2.5       # many samples fall outside the image
        mask = torch.rand(B, 9, Ho, Wo, dtype=torch.float64)
        got = dcn_v2_forward(x, offset, mask, wt, bias, stride, 1, 1)
        xl = x.tolist()
        for co in range(Co):
            for oy in range(Ho):
                for ox in range(Wo):
                    acc = bias[co].item()
                    for c in range(C):
                        for i in range(3):
                            for j in range(3):
                                k = i * 3 + j
                                h = oy * stride - 1 + i + offset[0, 2 * k, oy, ox].item()
                                w = ox * stride - 1 + j + offset[0, 2 * k + 1, oy, ox].item()
                                v = 0.0
                                if h > -1 and w > -1 and h < H and w < W:   # dcn_v2_im2col_cuda.cu:177-188
                                    v = _bilinear_scalar(xl[0][c], H, W, h, w)
                                acc += wt[co, c, i, j].item() * v * mask[0, k, oy, ox].item()
                    assert abs(acc - got[0, co, oy, ox].item()) < 1e-9, (stride, co, oy, ox)


def test_reference_kernel_recipe_builds_and_exports():
    """oracle/build_ref.sh compiles the reference's own DCNv2 im2col source for gfx950 (cross-compile, no GPU needed) into the
    git-ignored oracle/_ref/; the GPU parity test tests/test_gpu_dcn_reference.py binds `modulated_deformable_im2col_cuda`."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.isdir('/root/reference/external/DCNv2/src/cuda'):
        pytest.skip('/root/reference is not present on this machine')
    so = os.path.join(root, 'oracle', '_ref', 'libdcn_v2_im2col_ref.so')
    if not os.path.exists(so):
        subprocess.run(['bash', os.path.join(root, 'oracle', 'build_ref.sh')], check=True)
    syms = subprocess.run(['nm', '-D', so], capture_output=True, text=True, check=True).stdout
    assert ' T modulated_deformable_im2col_cuda' in syms
    # nothing of the reference is copied into the tree: the recipe reads the source in place, the outputs are git-ignored
    ign = open(os.path.join(root, '.gitignore')).read()
    assert 'oracle/_ref/' in ign
    tracked = subprocess.run(['git', 'ls-files', 'oracle/_ref'], cwd=root, capture_output=True, text=True).stdout
    assert tracked.strip() == ''
