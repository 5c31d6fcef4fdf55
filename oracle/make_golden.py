"""Generate tests/golden/*.npz by EXECUTING THE REFERENCE ITSELF (build container only).

    python oracle/make_golden.py            # needs /root/reference; writes tests/golden/

The reference (dbolya/yolact, Python) is imported unmodified with the four throw-away shims of SURVEY
appendix B (no GPU, no torchvision/cv2/pycocotools in this image), given the deterministic synthetic
parameters of yolact_amd.utils.synth, and run on CPU.  What is recorded, per case:
  * the reference state-dict layout (key, shape)                     -> drop-in checkpoint compatibility
  * digests (shape, sum, abs-sum, 2048 sampled values: round 2 had 64) of C3..C5, P3..P7, proto, loc, conf, mask, priors
  * the complete Detect output (box/mask/class/score per image) with use_fast_nms=True
  * postprocess(out, w, h) results (classes, scores, int boxes, bit-packed masks)
The fixtures are small (a few hundred KB); parameters and inputs are re-derived from seeds at test time.
/root/reference does not exist on the GPU box — nothing here runs there.
"""
from __future__ import annotations

import json
import os
import sys
import types

import numpy as np
import torch

REF = '/root/reference'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from yolact_amd.utils.synth import synth_state_dict, synth_images  # noqa: E402


def _shim_reference(with_dcn_oracle=False):
    torch.cuda.current_device = lambda: 0            # yolact.py:22 runs at import
    torch.cuda.device_count = lambda: 2              # reference's own switch: disables the TorchScript FPN

    def stub(name, **kw):
        m = types.ModuleType(name)
        sys.modules[name] = m
        for k, v in kw.items():
            setattr(m, k, v)
        return m

    r = stub('torchvision.models.resnet', Bottleneck=object)
    stub('torchvision', models=stub('torchvision.models', resnet=r), transforms=stub('torchvision.transforms'))
    stub('cv2')
    stub('pycocotools', mask=stub('pycocotools.mask'))
    if with_dcn_oracle:
        # The reference has no CPU DCN (src/cpu/dcn_v2_cpu.cpp:23): plug the oracle restatement in as `dcn_v2.DCN`
        import torch.nn as nn
        from oracle.yolact_oracle import dcn_v2_forward

        class DCN(nn.Module):
            def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation=1, deformable_groups=1):
                super().__init__()
                self.stride, self.padding, self.dilation = stride, padding, dilation
                self.weight = nn.Parameter(torch.zeros(out_channels, in_channels, kernel_size, kernel_size))
                self.bias = nn.Parameter(torch.zeros(out_channels))
                self.conv_offset_mask = nn.Conv2d(in_channels, 3 * kernel_size * kernel_size, kernel_size,
                                                  stride=stride, padding=padding, bias=True)

            def forward(self, x):
                out = self.conv_offset_mask(x)
                o1, o2, mask = torch.chunk(out, 3, dim=1)
                return dcn_v2_forward(x, torch.cat((o1, o2), 1), torch.sigmoid(mask), self.weight, self.bias,
                                      self.stride, self.padding, self.dilation)
        stub('dcn_v2', DCN=DCN)
    sys.path.insert(0, REF)


def digest(t: torch.Tensor, n=2048):
    t = t.detach().float().contiguous().view(-1)
    g = torch.Generator().manual_seed(t.numel() % 100003 + 7)
    idx = torch.randint(0, t.numel(), (n,), generator=g)
    return dict(numel=int(t.numel()), sum=float(t.double().sum()), abssum=float(t.double().abs().sum()),
                idx=idx.numpy().astype(np.int64), val=t[idx].numpy().astype(np.float32))


CASES = [
    # name, config, B, size, seed, conf_gain, post (w,h)
    ('r50_dense', 'yolact_resnet50_config', 2, 550, 0, 0.04, (160, 120)),
    ('r50_sparse', 'yolact_resnet50_config', 1, 550, 1, 0.03, (96, 128)),
    ('r50_empty', 'yolact_resnet50_config', 1, 550, 2, 0.02, (64, 64)),
    ('r101_base', 'yolact_base_config', 1, 550, 3, 0.04, (80, 60)),
    ('darknet53', 'yolact_darknet53_config', 1, 550, 4, 0.04, (80, 60)),
    ('im700', 'yolact_im700_config', 1, 700, 5, 0.04, (80, 60)),
    ('plus_r50', 'yolact_plus_resnet50_config', 1, 550, 6, 0.04, (80, 60)),
    # cross-class Fast NMS (detection.py:111-135; eval.py --cross_class_nms): the reference's own cc_fast_nms output
    ('r50_cc', 'yolact_resnet50_config', 2, 550, 7, 0.04, (80, 60), dict(cross_class=True)),
    # the "pretrained-like" sparse regime of SURVEY 8(d): ~1 % of the priors over the candidate threshold, a handful of
    # confident detections; postprocess runs with the display threshold (eval.py --score_threshold 0.15)
    ('r50_few', 'yolact_resnet50_config', 2, 550, 8, 0.2, (160, 120), dict(bg_bias=19.5, score_threshold=0.15)),
    # round 5 (VERDICT r4 missing #3-#5): the published YOLACT++ R101 row (dcn_interval 3, data/config.py:772-792), the 400 px
    # config (:706), a YOLACT++ batch of 2 (score2 = scores * maskiou for every image: the batched FastMaskIoUNet), and
    # eval.py --detect (eval.py:1067-1068: cfg.eval_mask_branch = False -> zero coefficients, no prototypes, boxes only)
    ('plus_base', 'yolact_plus_base_config', 1, 550, 9, 0.04, (80, 60)),
    ('im400', 'yolact_im400_config', 1, 400, 10, 0.04, (80, 60)),
    ('plus_r50_b2', 'yolact_plus_resnet50_config', 2, 550, 11, 0.04, (80, 60)),
    ('r50_nomask', 'yolact_resnet50_config', 1, 550, 12, 0.04, (80, 60), dict(eval_mask_branch=False)),
]


def run_case(name, config, B, size, seed, gain, post, outdir, extra=None):
    extra = extra or {}
    from data import cfg, set_cfg
    set_cfg(config)
    cfg.mask_proto_debug = False
    if 'eval_mask_branch' in extra:
        cfg.eval_mask_branch = bool(extra['eval_mask_branch'])
    from yolact import Yolact
    from layers.output_utils import postprocess
    torch.manual_seed(0)
    net = Yolact()
    net.eval()          # the reference's train() override returns None, so no chaining
    net.detect.use_fast_nms = True
    net.detect.use_cross_class_nms = bool(extra.get('cross_class', False))      # eval.py:872
    shapes = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
    sd = synth_state_dict(shapes, seed=seed, conf_gain=gain, bg_bias=extra.get('bg_bias', 0.0))
    net.load_state_dict(sd)
    x = synth_images(B, size, size, seed=1000 + seed)
    rec = {}
    with torch.no_grad():
        outs = net.backbone(x)
        sel = [outs[i] for i in cfg.backbone.selected_layers]
        feats = net.fpn(sel)
        for i, t in enumerate(sel):
            rec['C%d' % (i + 3)] = digest(t.permute(0, 2, 3, 1))
        for i, t in enumerate(feats):
            rec['P%d' % (i + 3)] = digest(t.permute(0, 2, 3, 1))
        # raw head outputs: temporarily stub Detect to capture pred_outs (yolact.py:676 hands them over)
        captured = {}
        real_detect = net.detect

        class Cap:
            def __call__(self, preds, n):
                captured.update(preds)
                return real_detect(preds, n)

            def __getattr__(self, k):
                return getattr(real_detect, k)
        net.detect = Cap()
        dets = net(x)
        net.detect = real_detect
        for k in ('loc', 'conf', 'mask', 'priors', 'proto'):
            if captured.get(k) is not None:           # (no 'proto' without the mask branch, yolact.py:579-580)
                rec[k] = digest(captured[k])
    arrays = {}
    meta = dict(name=name, config=config, B=B, size=size, seed=seed, conf_gain=gain, post=list(post),
                keys=[[k, list(s)] for k, s in shapes], n=[], torch=torch.__version__, **extra)
    meta['n_post'] = []
    for k, d in rec.items():
        arrays['dg_%s_idx' % k] = d['idx']
        arrays['dg_%s_val' % k] = d['val']
        meta['dg_' + k] = dict(numel=d['numel'], sum=d['sum'], abssum=d['abssum'])
    w, h = post
    for b, d in enumerate(dets):
        det = d['detection']
        if det is None:
            meta['n'].append(0)
            meta['n_post'].append(0)
            continue
        meta['n'].append(int(det['score'].shape[0]))
        for k in ('box', 'mask', 'class', 'score'):
            arrays['det%d_%s' % (b, k)] = det[k].numpy()
        with torch.no_grad():
            det_copy = [{'detection': {k: (v.clone() if torch.is_tensor(v) else v) for k, v in det.items()}, 'net': net}]
            classes, scores, boxes, masks = postprocess(det_copy, w, h, score_threshold=extra.get('score_threshold', 0))
        meta['n_post'].append(int(classes.shape[0]))
        if classes.shape[0] == 0:
            continue
        arrays['post%d_class' % b] = classes.numpy()
        if isinstance(scores, list):
            arrays['post%d_score' % b] = scores[0].numpy()
            arrays['post%d_score2' % b] = scores[1].numpy()
        else:
            arrays['post%d_score' % b] = scores.numpy()
        arrays['post%d_box' % b] = boxes.numpy()
        if extra.get('eval_mask_branch', True):
            arrays['post%d_maskbits' % b] = np.packbits(masks.numpy().astype(np.uint8).reshape(-1))
        else:                                          # output_utils.py:58: the 4th value stays the coefficient rows [n, 32]
            arrays['post%d_maskraw' % b] = masks.numpy()
    arrays['meta'] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(outdir, name + '.npz'), **arrays)
    print('%-12s n=%s  K(conf>0.05)=%s' % (name, meta['n'], [
        int((captured['conf'][b, :, 1:].max(1)[0] > 0.05).sum()) for b in range(B)]))


def main():
    only = sys.argv[1:]
    outdir = os.path.join(ROOT, 'tests', 'golden')
    os.makedirs(outdir, exist_ok=True)
    _shim_reference(with_dcn_oracle=True)
    for c in CASES:
        if only and c[0] not in only:
            continue
        run_case(*c[:7], outdir, c[7] if len(c) > 7 else None)


if __name__ == '__main__':
    main()
