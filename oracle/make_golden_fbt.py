"""Golden vectors for FastBaseTransform by EXECUTING THE REFERENCE's forward() (build container only).

The reference class hard-calls .cuda() in __init__ (utils/augmentations.py:626-627), so the instance is created with
object.__new__ and given CPU mean/std tensors; forward() itself (augmentations.py:630-658) runs unmodified on CPU.
Writes tests/golden/fbt.npz: per case the seed/shape/config and 4096 sampled output values + sum / abs-sum digests.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
from make_golden import _shim_reference  # noqa: E402

CASES = [  # name, config, n, h, w, seed
    ('resnet50', 'yolact_resnet50_config', 2, 37, 53, 11),
    ('im700', 'yolact_im700_config', 1, 480, 640, 12),
    ('darknet53', 'yolact_darknet53_config', 1, 64, 48, 13),
]


def synth_frame(n, h, w, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(n, h, w, 3, generator=g) * 255.0


def main():
    _shim_reference()
    sys.path.insert(0, '/root/reference')
    from data import cfg, set_cfg, MEANS, STD
    import utils.augmentations as A
    out = {}
    for name, config, n, h, w, seed in CASES:
        set_cfg(config)
        t = object.__new__(A.FastBaseTransform)
        torch.nn.Module.__init__(t)
        t.mean = torch.Tensor(MEANS).float()[None, :, None, None]
        t.std = torch.Tensor(STD).float()[None, :, None, None]
        t.transform = cfg.backbone.transform
        y = t(synth_frame(n, h, w, seed))
        flat = y.reshape(-1)
        idx = torch.randint(0, flat.numel(), (4096,), generator=torch.Generator().manual_seed(seed + 100))
        out[name + '_shape'] = np.array(y.shape)
        out[name + '_idx'] = idx.numpy()
        out[name + '_val'] = flat[idx].numpy()
        out[name + '_sum'] = np.array([flat.double().sum().item(), flat.double().abs().sum().item()])
        print(name, tuple(y.shape), out[name + '_sum'])
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'fbt.npz'), **out)


if __name__ == '__main__':
    main()
