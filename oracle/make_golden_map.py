"""Golden mAP tables from the REFERENCE's own evaluator (build container only):  python oracle/make_golden_map.py

For every golden case the reference's stored postprocess() outputs (tests/golden/<case>.npz, produced by executing the
reference model) are scored by the reference's own eval.prep_metrics / eval.calc_map (eval.py:386-470,1005-1031)
against pseudo ground truth built from those same detections (oracle/map_eval.pseudo_gt) and then displaced object by
object (oracle/map_eval.perturb_gt) so that the table falls from IoU .50 to .95 instead of being flat.  eval.postprocess is
replaced by a stub that returns the stored outputs, so exactly the evaluator code is exercised.  Writes
tests/golden/map.npz: per case the GT (boxes, classes, packed masks) and the reference's box / mask mAP table.
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
from make_golden import _shim_reference  # noqa: E402
from oracle import map_eval as ME        # noqa: E402

CASES = ['r50_dense', 'r50_sparse', 'r101_base', 'darknet53', 'im700', 'plus_r50']


def load_post(z, b, w, h):
    classes = torch.from_numpy(z['post%d_class' % b])
    scores = torch.from_numpy(z['post%d_score' % b])
    if ('post%d_score2' % b) in z.files:
        scores = [scores, torch.from_numpy(z['post%d_score2' % b])]
    boxes = torch.from_numpy(z['post%d_box' % b])
    n = classes.shape[0]
    masks = torch.from_numpy(np.unpackbits(z['post%d_maskbits' % b])[: n * h * w].reshape(n, h, w).astype(np.float32))
    return classes, scores, boxes, masks


def main():
    _shim_reference()
    sys.path.insert(0, '/root/reference')
    torch.Tensor.cuda = lambda self, *a, **k: self                 # eval.py:416-417 hard-call .cuda()
    from data import cfg, set_cfg
    import eval as E
    E.parse_args(['--no_bar', '--cuda=False'])
    out = {}
    for name in CASES:
        z = np.load(os.path.join(ROOT, 'tests', 'golden', name + '.npz'))
        meta = json.loads(bytes(z['meta']).decode())
        set_cfg(meta['config'])
        w, h = meta['post']
        ncls = len(cfg.dataset.class_names)
        ap_data = {'box': [[E.APDataObject() for _ in range(ncls)] for _ in E.iou_thresholds],
                   'mask': [[E.APDataObject() for _ in range(ncls)] for _ in E.iou_thresholds]}
        for b, n in enumerate(meta['n']):
            if n == 0:
                continue
            post = load_post(z, b, w, h)
            gt, gt_masks = ME.pseudo_gt(*post, w, h)
            gt, gt_masks = ME.perturb_gt(gt, gt_masks, w, h, seed=CASES.index(name) * 4 + b)     # a table that falls from .50 to .95
            out['%s_gt%d' % (name, b)] = gt
            out['%s_gtmaskbits%d' % (name, b)] = np.packbits(gt_masks.reshape(-1))
            E.postprocess = lambda dets, w_, h_, **kw: post          # the evaluator sees the stored reference outputs
            E.prep_metrics(ap_data, None, None, gt, gt_masks.astype(np.float32), h, w, 0, b, None)
        maps = E.calc_map(ap_data)
        keys = list(maps['box'].keys())
        out[name + '_keys'] = np.array([str(k) for k in keys])
        out[name + '_box'] = np.array([maps['box'][k] for k in keys], dtype=np.float64)
        out[name + '_mask'] = np.array([maps['mask'][k] for k in keys], dtype=np.float64)
        print(name, 'box', maps['box']['all'], 'mask', maps['mask']['all'])
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'map.npz'), **out)


if __name__ == '__main__':
    main()
