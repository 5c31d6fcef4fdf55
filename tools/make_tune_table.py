#!/usr/bin/env python
"""Measure the tile / algorithm table the engine ships (yolact_amd/tune/gfx950.json) on the MI355X.

    python tools/make_tune_table.py [--out yolact_amd/tune/gfx950.json] [--fresh] [--only NAME ...]

Every (config, batch, size) below is planned once with on-device measurement of every shape the table does not know yet
(HIP events on the launch stream, engine.Plan.tune); entries accumulate in ONE file, so a shape shared by several plans
is measured once and every plan that contains it runs the same tile (bit-identical results across plans that share
layers).  The default plan of the product then reads the table and measures nothing: deterministic across processes and
boxes (tests/test_gpu_batch_parity.py::test_plan_is_deterministic_across_processes).

Re-run after any change to the conv kernels or tile ids (bump engine.TUNE_GEN so that old tables are ignored)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# (name, config, batch, size): the BASELINE configs at their batch sizes, the golden-fixture shapes of tests/, batch 1
PLANS = [
    ('configs1_r50_b8', 'yolact_resnet50_config', 8, 550),
    ('r50_b1', 'yolact_resnet50_config', 1, 550),
    ('r50_b2', 'yolact_resnet50_config', 2, 550),
    ('r50_b4', 'yolact_resnet50_config', 4, 550),          # round 6: the per-rank batch of the strong-scaling region at 2 GPUs (bench.py --global-batch 8)
    ('configs2_r101_b16', 'yolact_base_config', 16, 550),
    ('r101_b1', 'yolact_base_config', 1, 550),
    ('configs4_im700_b8', 'yolact_im700_config', 8, 700),
    ('im700_b1', 'yolact_im700_config', 1, 700),
    ('configs3_plus_b8', 'yolact_plus_resnet50_config', 8, 550),
    ('plus_b1', 'yolact_plus_resnet50_config', 1, 550),
    ('darknet_b8', 'yolact_darknet53_config', 8, 550),
    ('darknet_b1', 'yolact_darknet53_config', 1, 550),
    # round 5: the published YOLACT++ R101 row and the 400 px config (VERDICT r4 missing #4), the YOLACT++ golden batch of 2
    ('plus_base_b8', 'yolact_plus_base_config', 8, 550),
    ('plus_base_b1', 'yolact_plus_base_config', 1, 550),
    ('im400_b8', 'yolact_im400_config', 8, 400),
    ('im400_b1', 'yolact_im400_config', 1, 400),
    ('plus_b2', 'yolact_plus_resnet50_config', 2, 550),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=os.path.join(ROOT, 'yolact_amd', 'tune', 'gfx950.json'))
    ap.add_argument('--fresh', action='store_true', help='discard the existing table first')
    ap.add_argument('--only', nargs='*', default=None)
    ap.add_argument('--copy-to', default=None, help='also write the finished table here (e.g. under gpurun_out/)')
    args = ap.parse_args()
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    if args.fresh and os.path.exists(args.out):
        os.remove(args.out)
    os.environ['YOLACT_AMD_TUNE_CACHE'] = args.out          # Plan.tune persists new entries here
    os.environ['YOLACT_AMD_AUTOTUNE'] = '1'
    import torch
    import yolact_amd
    from yolact_amd import engine
    from yolact_amd.utils.synth import synth_images, synth_state_dict
    dev = torch.device('cuda', 0)
    log = []
    for name, config, B, size in PLANS:
        if args.only and name not in args.only:
            continue
        engine._table_cache.clear()                          # the shipped table may be the very file being extended
        yolact_amd.set_cfg(config)
        from yolact_amd.yolact import Yolact
        net = Yolact()
        sd = synth_state_dict([(k, tuple(v.shape)) for k, v in net.state_dict().items()], seed=0, conf_gain=0.04)
        net.load_state_dict_compat(sd)
        net.detect.use_fast_nms = True
        net = net.to(dev)
        x = synth_images(B, size, size, seed=1234).to(dev)
        t0 = time.time()
        with torch.no_grad():
            plan = net.plan_for(x)
        torch.cuda.synchronize()
        n_w = sum(1 for op in plan.ops if isinstance(op[2], str) and op[2].endswith('[wino]'))
        log.append({'plan': name, 'config': config, 'batch': B, 'size': size, 'measured_shapes': plan.tune_misses,
                    'winograd_layers': n_w, 'seconds': round(time.time() - t0, 1)})
        print(log[-1], flush=True)
        del net, plan, x
        torch.cuda.empty_cache()
    n = len(engine._read_table_file(args.out))
    print('table %s: %d entries' % (args.out, n))
    if args.copy_to:
        os.makedirs(os.path.dirname(args.copy_to), exist_ok=True)
        with open(args.out) as f, open(args.copy_to, 'w') as g:
            g.write(f.read())
        with open(args.copy_to + '.log', 'w') as g:
            json.dump(log, g, indent=1)


if __name__ == '__main__':
    main()
