#!/usr/bin/env python
"""Phase time stamps of the Detect kernels K2 (class_topk_nms_k) / K3 (final_topk_k) on the dense R50 head outputs
(diagnostics build: make -C yolact_amd/csrc clean; make -C yolact_amd/csrc DIAG=1).  PROBE_BATCH=1|8."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device('cuda', 0)
B = int(os.environ.get('PROBE_BATCH', '8'))
with torch.no_grad():
    net, sd = bench.build_model(dev, 550)
    from yolact_amd.utils.synth import synth_images
    x = synth_images(B, 550, 550, seed=1234).to(dev)
    net.forward_device(x)
    plan = net.plan_for(x)
    buf = torch.zeros((4096 + 64) * 16, dtype=torch.int64, device=dev)
    os.environ['YMI_DETECT_TRACE'] = str(buf.data_ptr())
    for _ in range(3):
        net.detect.run_device(plan.loc, plan.conf, plan.coef, plan.priors, True, conf_ld=plan.conf_ld)
    torch.cuda.synchronize()
    tr = buf.cpu().view(-1, 16).double()
    k2 = tr[:80 * B]
    names2 = ['loads', 'bisection', 'take', 'sort', 'decode', 'iou', 'write']
    print('K2 (%d blocks), shader cycles: mean / max per phase' % k2.shape[0])
    for i, nm in enumerate(names2):
        d = k2[:, i + 1] - k2[:, i]
        print('  %-10s %9.0f %9.0f' % (nm, d.mean().item(), d.max().item()))
    print('  %-10s %9.0f %9.0f   (span of all blocks %.0f)' % ('total', (k2[:, 7] - k2[:, 0]).mean().item(), (k2[:, 7] - k2[:, 0]).max().item(),
          (k2[:, 7].max() - k2[:, 0].min()).item()))
    k3 = tr[4096:4096 + B]
    print('K3 (%d blocks)' % B)
    for nm, a, b_ in (('count', 0, 8), ('key loads', 8, 1), ('bisection', 1, 2), ('take', 2, 3), ('sort', 3, 4), ('boxes', 4, 5), ('coef rows', 5, 7)):
        d = k3[:, b_] - k3[:, a]
        print('  %-10s %9.0f %9.0f' % (nm, d.mean().item(), d.max().item()))
    print('  %-10s %9.0f' % ('total', (k3[:, 7] - k3[:, 0]).mean().item()))
