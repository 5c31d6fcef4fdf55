"""Golden frames for prep_display's mask compositing by EXECUTING THE REFERENCE's prep_display (build container only).

eval.prep_display (eval.py:135-262) is run unmodified with undo_transform=False, --display_bboxes/--display_text off
(those need cv2), eval.postprocess stubbed to return the stored reference postprocess() outputs of a golden case.
get_color() asks the frame tensor for `.device.index` to key its colour cache and returns a bare tuple for CPU tensors
(index None), which the next line cannot .view(): the frame is therefore passed as a Tensor subclass whose `.device`
reports index 'cpu' — the only shim.  Writes tests/golden/display.npz (sampled pixels + digests of the uint8 frames).
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
from make_golden import _shim_reference        # noqa: E402
from make_golden_map import load_post          # noqa: E402

CASES = [('r50_dense', 0, 5), ('r50_dense', 1, 3), ('r50_sparse', 0, 5), ('im700', 0, 1)]   # case, image, top_k


class _Dev:
    index = 'cpu'
    type = 'cpu'


class FrameTensor(torch.Tensor):
    device = property(lambda self: _Dev())


def synth_frame(h, w, seed):
    return torch.rand(h, w, 3, generator=torch.Generator().manual_seed(seed)) * 255.0


def main():
    _shim_reference()
    sys.path.insert(0, '/root/reference')
    torch.Tensor.cuda = lambda self, *a, **k: self
    from data import set_cfg
    import eval as E
    out = {}
    for name, b, top_k in CASES:
        z = np.load(os.path.join(ROOT, 'tests', 'golden', name + '.npz'))
        meta = json.loads(bytes(z['meta']).decode())
        set_cfg(meta['config'])
        w, h = meta['post']
        E.parse_args(['--no_bar', '--cuda=False', '--display_bboxes=False', '--display_text=False', '--top_k=%d' % top_k])
        post = load_post(z, b, w, h)
        E.postprocess = lambda dets, w_, h_, **kw: post
        E.color_cache.clear()
        frame = synth_frame(h, w, 40 + b).as_subclass(FrameTensor)
        img = E.prep_display(None, frame, None, None, undo_transform=False)
        key = '%s_%d_%d' % (name, b, top_k)
        flat = np.asarray(img).reshape(-1)
        idx = np.random.RandomState(7).randint(0, flat.size, 4096)
        out[key + '_shape'] = np.array(img.shape)
        out[key + '_idx'] = idx
        out[key + '_val'] = flat[idx]
        out[key + '_sum'] = np.array([int(flat.astype(np.int64).sum())])
        print(key, img.shape, img.dtype, out[key + '_sum'])
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'display.npz'), **out)


if __name__ == '__main__':
    main()
