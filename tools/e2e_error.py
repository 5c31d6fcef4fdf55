#!/usr/bin/env python
"""End-to-end head error of the HIP path vs the CPU oracle (fp32) on the golden cases, relative to max(1, max|ref|) — the
quantity tests/test_gpu_path.py::test_end_to_end_heads_and_detections bounds by 1e-4.
   YOLACT_AMD_WINOGRAD=0|2|4|1 python tools/e2e_error.py [case ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from helpers import oracle_run, case_images
from gpu_utils import build_net

cases = sys.argv[1:] or ['r50_dense', 'r101_base', 'darknet53', 'im700', 'plus_r50']
mode = {'0': 'direct only', '2': 'F(2x2) + direct', '4': 'F(4x4) + direct', '1': 'autotuned (direct / F(2x2) / F(4x4))'}[
    os.environ.get('YOLACT_AMD_WINOGRAD', '1')]
for name in cases:
    meta, arrays, cfg, sd, raw, dets = oracle_run(name)
    net = build_net(meta)
    got = net.forward_raw(case_images(meta).cuda())
    torch.cuda.synchronize()
    plan = net.plan_for(case_images(meta).cuda())
    nw = sum(1 for op in plan.ops if str(op[2]).endswith('[wino]'))
    errs = []
    for k in ('loc', 'conf_logits', 'mask', 'proto'):
        g, r = got[k].cpu(), raw[k]
        errs.append('%s %.2e' % (k, (g - r).abs().max().item() / max(1.0, r.abs().max().item())))
    print('%-10s %-38s winograd layers %2d   %s' % (name, mode, nw, '  '.join(errs)))
