#!/usr/bin/env python
"""Time Detect (3 kernels) on the dense R50 batch-8 head outputs, with ablations of K2 (YMI_DETECT_ABLATE; needs the
diagnostics build: make -C yolact_amd/csrc clean all DIAG=1)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device('cuda', 0)
with torch.no_grad():
    net, sd = bench.build_model(dev, 550)
    from yolact_amd.utils.synth import synth_images
    B = int(os.environ.get('PROBE_BATCH', '8'))
    ABL = [int(v) for v in os.environ.get('PROBE_ABL', '0,1,4,5').split(',')]
    x = synth_images(B, 550, 550, seed=1234).to(dev)
    net.forward_device(x)
    plan = net.plan_for(x)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for abl in ABL:
        os.environ['YMI_DETECT_ABLATE'] = str(abl)
        for _ in range(2):
            net.detect.run_device(plan.loc, plan.conf, plan.coef, plan.priors, True, conf_ld=plan.conf_ld)
        e0.record()
        for _ in range(10):
            o = net.detect.run_device(plan.loc, plan.conf, plan.coef, plan.priors, True, conf_ld=plan.conf_ld)
        e1.record(); e1.synchronize()
        print('ablate=%d  detect %.1f us  counts %s  num_keep %s' % (abl, e0.elapsed_time(e1) * 100, o['count'].tolist(),
              net.detect._ws[next(iter(net.detect._ws))]['num_keep'].tolist()))
