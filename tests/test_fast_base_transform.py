"""FastBaseTransform (utils/augmentations.py:616-658).

CPU: the oracle restatement is pinned to outputs of the reference's own forward(), executed in the build container by
oracle/make_golden_fbt.py (tests/golden/fbt.npz).  GPU: the HIP kernel (both output layouts) against the oracle —
tolerance 1e-4 * max(1, max|ref|), the path's parity bar.  The fp32 source coordinate scale*(dst+0.5)-0.5 is rounded
differently by an FMA-contracting build (torch's CPU kernel) and by ours (-ffp-contract=off): a 1-ulp difference of the
interpolation weight times a pixel difference of up to 255 grey levels (the test images are white noise, the worst
case) is ~5e-3 grey levels = ~6e-5 after the division by std.
"""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'oracle'))
from yolact_amd.config import CONFIGS
from oracle import yolact_oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'fbt.npz')
CASES = [('resnet50', 'yolact_resnet50_config', 2, 37, 53, 11), ('im700', 'yolact_im700_config', 1, 480, 640, 12),
         ('darknet53', 'yolact_darknet53_config', 1, 64, 48, 13)]


def synth_frame(n, h, w, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(n, h, w, 3, generator=g) * 255.0


@pytest.mark.parametrize('case', CASES, ids=[c[0] for c in CASES])
def test_oracle_matches_reference_execution(case):
    name, config, n, h, w, seed = case
    z = np.load(GOLD)
    y = O.fast_base_transform(synth_frame(n, h, w, seed), CONFIGS[config].copy())
    assert tuple(y.shape) == tuple(z[name + '_shape'])
    flat = y.reshape(-1)
    assert np.array_equal(flat[torch.from_numpy(z[name + '_idx'])].numpy(), z[name + '_val'])      # bit-exact
    s = np.array([flat.double().sum().item(), flat.double().abs().sum().item()])
    assert np.allclose(s, z[name + '_sum'], rtol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize('case', CASES, ids=[c[0] for c in CASES])
def test_hip_kernel_matches_oracle(case):
    import yolact_amd
    from yolact_amd.utils.augmentations import FastBaseTransform
    name, config, n, h, w, seed = case
    yolact_amd.set_cfg(config)
    img = synth_frame(n, h, w, seed)
    ref = O.fast_base_transform(img, CONFIGS[config].copy())
    t = FastBaseTransform()
    got = t(img.cuda()).cpu()
    assert got.shape == ref.shape and got.dtype == torch.float32
    tol = 1e-4 * max(1.0, ref.abs().max().item())
    assert (got - ref).abs().max().item() <= tol
    nhwc4 = t.to_nhwc4(img.cuda()).cpu()
    assert nhwc4.shape == (n, ref.shape[2], ref.shape[3], 4)
    assert torch.equal(nhwc4[..., :3].permute(0, 3, 1, 2), got) and nhwc4[..., 3].abs().max().item() == 0


@pytest.mark.gpu
def test_preserve_aspect_ratio_and_errors():
    import yolact_amd
    from yolact_amd.utils.augmentations import FastBaseTransform, calc_size_preserve_ar
    cfg = yolact_amd.set_cfg('yolact_resnet50_config')
    cfg = yolact_amd.active_cfg()
    cfg.preserve_aspect_ratio = True
    try:
        img = synth_frame(1, 90, 160, 5)
        got = FastBaseTransform()(img.cuda()).cpu()
        ow, oh = calc_size_preserve_ar(160, 90, cfg.max_size)
        assert got.shape == (1, 3, oh, ow)
        ref = O.fast_base_transform(img, cfg)
        assert (got - ref).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item())
    finally:
        cfg.preserve_aspect_ratio = False
    with pytest.raises(RuntimeError):
        FastBaseTransform()(synth_frame(1, 8, 8, 1))          # CPU tensor: no fallback
    with pytest.raises(ValueError):
        FastBaseTransform()(torch.zeros(1, 3, 8, 8, device='cuda'))
