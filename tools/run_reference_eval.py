#!/usr/bin/env python
"""Drive the reference's UNMODIFIED eval.py (eval.evaluate, eval.py:870-1003; AP code :386-581) on this machine — once
with the reference's own Yolact and once with the MI355X engine bound through shim/ — on a synthetic dataset with pseudo
ground truth ("run the reference's own evaluator twice").  MEASUREMENT / TEST INFRASTRUCTURE, not product.

    python tools/run_reference_eval.py --who reference|engine --mode gt|map|benchmark|coco --out DIR [--images N] [--cuda 0|1]

`--who reference`: sys.path = [<reference copy>]: the reference model itself (PyTorch-ROCm / MIOpen on the GPU, or CPU).
`--who engine`:    sys.path = [shim/, repo, <reference copy>]: eval.py, data/, utils/ are the reference's files, `yolact`,
                   `layers.*`, `dcn_v2` resolve to yolact_amd (the drop-in of INTEGRATION.md).
Modes: gt        — (reference) detections -> postprocess -> oracle.map_eval.pseudo_gt -> DIR/gt.npz
       map       — eval.evaluate(net, dataset) in mAP mode (prep_metrics / calc_map) -> DIR/map_<who>.json
       benchmark — eval.evaluate with --benchmark (prep_benchmark, the reference's FPS definition) -> DIR/benchmark_<who>.json
       coco      — eval.evaluate with --output_coco_json (Detections.add_bbox / add_mask / dump) -> DIR/*_detections.json
The reference copy is $YOLACT_REFERENCE_DIR, else <repo>/_scratch_reference (staged by tools/stage_reference.sh for ONE
gpurun session and deleted afterwards; never committed), else /root/reference.  One process per (who, mode): the
reference's module names (data, utils, layers, yolact) are global.
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H_IMG, W_IMG = 480, 640          # the "original image size" pull_item reports: postprocess resizes masks to it


def reference_dir():
    for p in (os.environ.get('YOLACT_REFERENCE_DIR'), os.path.join(ROOT, '_scratch_reference'), '/root/reference'):
        if p and os.path.exists(os.path.join(p, 'eval.py')):
            return p
    raise SystemExit('no reference checkout found (tools/stage_reference.sh stages one for a gpurun session)')


def stub(name, **kw):
    m = types.ModuleType(name)
    sys.modules[name] = m
    for k, v in kw.items():
        setattr(m, k, v)
    return m


def install_stubs(who):
    """dependencies of the reference that this image lacks."""
    try:
        import torchvision  # noqa: F401
    except ImportError:
        r = stub('torchvision.models.resnet', Bottleneck=object)
        stub('torchvision', models=stub('torchvision.models', resnet=r), transforms=stub('torchvision.transforms'))
    try:
        import cv2  # noqa: F401
    except ImportError:
        stub('cv2')
    try:
        import pycocotools.mask  # noqa: F401
    except ImportError:
        def encode(arr):
            """pycocotools.mask.encode for eval.py:321 (Detections.add_mask): Fortran uint8 [h,w] -> {'size','counts': bytes}.
            Engine runs use the device RLE codec (byte-identical to pycocotools on 13.5 k reference strings)."""
            from yolact_amd.coco import rle_encode
            m = torch.from_numpy(np.ascontiguousarray(arr)).to('cuda', torch.float32)[None]
            rec = rle_encode(m)[0]
            return {'size': rec['size'], 'counts': rec['counts'].encode('ascii')}
        stub('pycocotools', mask=stub('pycocotools.mask', encode=encode))


class SynthDataset:
    """Duck-typed COCODetection (data/coco.py:100-176 contract, consumer eval.py:936): .ids, __len__, pull_item."""

    def __init__(self, n, size, gt_file=None):
        from yolact_amd.utils.synth import synth_images
        self._synth = synth_images
        self.ids = [1000 + i for i in range(n)]
        self.size = size
        self.gt = np.load(gt_file) if gt_file and os.path.exists(gt_file) else None

    def __len__(self):
        return len(self.ids)

    def pull_item(self, index):
        iid = self.ids[index]
        img = self._synth(1, self.size, self.size, seed=iid)[0].cpu()        # the reference returns a CPU tensor; eval.py:938-940 moves it
        if self.gt is not None and ('gt%d' % iid) in self.gt.files:
            gt = self.gt['gt%d' % iid]
            n = gt.shape[0]
            masks = np.unpackbits(self.gt['bits%d' % iid])[: n * H_IMG * W_IMG].reshape(n, H_IMG, W_IMG).astype(np.float32)
        else:
            gt, masks = np.zeros((1, 5)), np.zeros((1, H_IMG, W_IMG), dtype=np.float32)
            gt[0] = [0.1, 0.1, 0.2, 0.2, 0]
        return img, gt, masks, H_IMG, W_IMG, 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--who', choices=['reference', 'engine'], required=True)
    ap.add_argument('--mode', choices=['gt', 'map', 'benchmark', 'coco'], required=True)
    ap.add_argument('--out', required=True)
    ap.add_argument('--images', type=int, default=16)
    ap.add_argument('--config', default='yolact_resnet50_config')
    ap.add_argument('--cuda', type=int, default=1)
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--conf-gain', type=float, default=0.04)
    ap.add_argument('--perturb', type=int, default=1, help='gt mode: 1 = displaced pseudo-GT (discriminating table), 0 = the flat round-3 form')
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    REF = reference_dir()
    if a.who == 'engine' and not a.cuda:
        raise SystemExit('the engine has no CPU path')
    if a.cuda and not torch.cuda.is_available():
        raise SystemExit('--cuda 1 needs a GPU')

    if a.who == 'reference':
        torch.cuda.current_device = (lambda: 0) if not a.cuda else torch.cuda.current_device      # yolact.py:22 runs at import
        torch.cuda.device_count = lambda: 2          # the reference's own switch that turns the TorchScript FPN off (yolact.py:25-30)
        if not a.cuda:
            torch.Tensor.cuda = lambda self, *x, **k: self                # eval.py:416-417 hard-call .cuda()
            torch.cuda.synchronize = lambda *x, **k: None                 # eval.py:281
    install_stubs(a.who)
    sys.path[:0] = ([os.path.join(ROOT, 'shim'), ROOT, REF] if a.who == 'engine' else [REF, ROOT])

    import eval as E
    assert os.path.realpath(E.__file__).startswith(os.path.realpath(REF)), E.__file__
    import yolact as Y
    if a.who == 'engine':
        import yolact_amd
        assert E.Yolact is yolact_amd.yolact.Yolact, 'shim/ did not bind'
    else:
        assert os.path.realpath(Y.__file__).startswith(os.path.realpath(REF)), Y.__file__
    from data import cfg, set_cfg
    import utils.timer as timer

    argv = ['--no_bar', '--max_images=%d' % a.images, '--cuda=%s' % bool(a.cuda)]
    if a.mode == 'benchmark':
        argv.append('--benchmark')
    if a.mode == 'coco':
        argv += ['--output_coco_json', '--bbox_det_file=%s/bbox_detections.json' % a.out,
                 '--mask_det_file=%s/mask_detections.json' % a.out]
    argv.append('--ap_data_file=%s/ap_data_%s.pkl' % (a.out, a.who))
    E.parse_args(argv)
    set_cfg(a.config)                                # eval.py:1050-1062 (from --config / the checkpoint name)
    if a.mode == 'coco':
        E.prep_coco_cats()                           # eval.py:1048

    from yolact_amd.utils.synth import synth_state_dict
    info = {'who': a.who, 'mode': a.mode, 'images': a.images, 'config': a.config, 'cuda': bool(a.cuda),
            'reference_dir': REF, 'torch': torch.__version__,
            'device': torch.cuda.get_device_name(0) if a.cuda else 'cpu (%d threads)' % torch.get_num_threads()}
    with torch.no_grad():                            # eval.py:1073
        if a.cuda:                                   # eval.py:1077-1081
            import torch.backends.cudnn as cudnn
            cudnn.fastest = True
            torch.set_default_tensor_type('torch.cuda.FloatTensor')
        else:
            torch.set_default_tensor_type('torch.FloatTensor')
            torch.set_num_threads(min(16, torch.get_num_threads()))      # one thread per logical CPU oversubscribes big hosts
        net = E.Yolact()
        sd = synth_state_dict([(k, tuple(v.shape)) for k, v in net.state_dict().items()], seed=a.seed, conf_gain=a.conf_gain)
        net.load_state_dict(sd)                      # stands in for net.load_weights(args.trained_model) (eval.py:1097)
        net.eval()
        if a.cuda:
            net = net.cuda()                         # eval.py:1100-1101
        gt_file = os.path.join(a.out, 'gt.npz')

        if a.mode == 'gt':
            from oracle.map_eval import perturb_gt, pseudo_gt
            net.detect.use_fast_nms = True
            cfg.mask_proto_debug = False
            ds = SynthDataset(a.images, cfg.max_size)
            rec = {}
            for i, iid in enumerate(ds.ids):
                img = ds.pull_item(i)[0].unsqueeze(0)
                preds = net(img.cuda() if a.cuda else img)
                classes, scores, boxes, masks = E.postprocess(preds, W_IMG, H_IMG)
                if classes.numel() == 0:
                    continue
                sc = [s.cpu() for s in scores] if isinstance(scores, list) else scores.cpu()
                gt, gm = pseudo_gt(classes.cpu(), sc, boxes.cpu(), masks.cpu(), W_IMG, H_IMG)
                if a.perturb:        # displace every GT object to a chosen IoU in [0.52, 0.98]: the table then falls from .50 to .95
                    gt, gm = perturb_gt(gt, gm, W_IMG, H_IMG, seed=i)
                rec['gt%d' % iid] = gt
                rec['bits%d' % iid] = np.packbits(gm.reshape(-1))
            np.savez_compressed(gt_file, **rec)
            info['gt_images'] = len(rec) // 2
            info['gt_objects'] = int(sum(v.shape[0] for k, v in rec.items() if k.startswith('gt')))
        else:
            ds = SynthDataset(a.images, cfg.max_size, gt_file)
            buf = io.StringIO()
            t0 = time.time()

            class Tee(io.TextIOBase):
                def write(self, s):
                    buf.write(s)
                    sys.__stdout__.write(s)
                    return len(s)
            with contextlib.redirect_stdout(Tee()):
                maps = E.evaluate(net, ds)           # <- the reference's own loop, timers, prep_* and AP code
            if a.cuda:
                torch.cuda.synchronize()
            info['seconds'] = round(time.time() - t0, 3)
            text = buf.getvalue()
            info['stdout_tail'] = text[-3000:]
            info['timer_sections'] = {k: float(v) for k, v in getattr(timer, '_total_times', {}).items()}
            if a.mode == 'map':
                info['map'] = {t: {str(k): float(v) for k, v in maps[t].items()} for t in ('box', 'mask')}
            if a.mode == 'benchmark':
                for line in text.splitlines():
                    if line.startswith('Average:'):
                        info['eval_py_average_line'] = line.strip()
                        info['fps'] = float(line.split()[1])
            if a.mode == 'coco':
                for nm in ('bbox_detections.json', 'mask_detections.json'):
                    d = json.load(open(os.path.join(a.out, nm)))
                    info[nm] = {'records': len(d), 'first': {k: (v if k != 'segmentation' else {'size': v['size'], 'counts': v['counts'][:40] + '...'})
                                                             for k, v in d[0].items()} if d else None}
    with open(os.path.join(a.out, '%s_%s.json' % (a.mode, a.who)), 'w') as f:
        json.dump(info, f, indent=1)
    print(json.dumps({k: v for k, v in info.items() if k not in ('stdout_tail',)})[:1500])


if __name__ == '__main__':
    main()
