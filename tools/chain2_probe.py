#!/usr/bin/env python
"""The plan's conv3 -> next conv1 fusion decisions (engine.Plan._fuse_pointwise_chains): ms of the two launches it would replace
against ms of the one chained launch (csrc/chain.hip at 64 planes, csrc/chain2.hip at 128 / 256), per pair of the headline plan.

    python tools/chain2_probe.py [--batch 8] [--config yolact_resnet50_config]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--config', default='yolact_resnet50_config')
    args = ap.parse_args()
    import yolact_amd
    from yolact_amd.utils.synth import synth_images, synth_state_dict
    yolact_amd.set_cfg(args.config)
    from yolact_amd.yolact import Yolact
    dev = torch.device('cuda', 0)
    net = Yolact()
    net.load_state_dict_compat(synth_state_dict([(k, tuple(v.shape)) for k, v in net.state_dict().items()], seed=0, conf_gain=0.04))
    net.detect.use_fast_nms = True
    net = net.to(dev)
    size = int(yolact_amd.CONFIGS[args.config].max_size)
    x = synth_images(args.batch, size, size, seed=1234).to(dev)
    with torch.no_grad():
        plan = net.plan_for(x)
    print('%-40s %6s %10s %10s' % ('pair', 'fused', 'plan ms', 'chain ms'))
    for name, on, t_plan, t_one in plan.chain_table:
        print('%-40s %6d %10.4f %10.4f   x%.2f' % (name, on, t_plan, t_one, t_plan / max(t_one, 1e-9)))


if __name__ == '__main__':
    main()
