#!/usr/bin/env python
"""Whole-step A/B of the direct-vs-Winograd choice (round 5).

engine.Plan._tune_winograd picks Winograd for a 3x3 layer when its three launches, timed back to back on resident buffers, beat the
direct launch by 3 %.  Inside the step the picture differs (per-layer times are 1.2 - 1.7x the isolated ones; a Winograd
layer crosses the memory side twice more than a direct one and pays two more kernel boundaries), so for shapes whose isolated margin is
small the choice can be wrong.  This tool toggles ALL layers of one table key at a time between the two forms and times the WHOLE step
(forward + Detect, the bench's step) — alternating, median of several rounds — and writes `instep|<key>: 0` into the tune table for the
keys where direct wins the step by more than the noise floor.

    python tools/instep_tune.py [--config yolact_resnet50_config --batch 8] [--rounds 7 --steps 40] [--write]
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='yolact_resnet50_config')
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--size', type=int, default=0)
    ap.add_argument('--rounds', type=int, default=7)
    ap.add_argument('--steps', type=int, default=40)
    ap.add_argument('--max-margin', type=float, default=0.70, help='only keys whose isolated wino/direct time ratio exceeds this')
    ap.add_argument('--overlap', type=int, default=1, help='batches in flight (yolact_amd.pipeline.BatchPipeline depth): the regime bench.py times by default is 2')
    ap.add_argument('--write', action='store_true', help='persist the decisions in yolact_amd/tune/gfx950.json')
    args = ap.parse_args()
    import torch
    import yolact_amd
    from yolact_amd import engine
    from yolact_amd.utils.synth import synth_images, synth_state_dict
    yolact_amd.set_cfg(args.config)
    from yolact_amd.yolact import Yolact
    dev = torch.device('cuda', 0)
    net = Yolact()
    net.load_state_dict_compat(synth_state_dict([(k, tuple(v.shape)) for k, v in net.state_dict().items()], seed=0, conf_gain=0.04))
    net.detect.use_fast_nms = True
    net = net.to(dev)
    size = args.size or int(yolact_amd.CONFIGS[args.config].max_size)
    x = synth_images(args.batch, size, size, seed=1234).to(dev)
    with torch.no_grad():
        plan = net.plan_for(x)
        plans = [net.plan_for(x, k) for k in range(args.overlap)]
        from yolact_amd.pipeline import BatchPipeline
        pipe = BatchPipeline(net, args.overlap) if args.overlap > 1 else None

        def set_all(key, on):
            for p_ in plans:
                p_.set_winograd(key, on)

        def step_ms(n):
            if pipe is not None:             # n batches through the pipeline, the host two batches ahead like bench.py
                for _ in range(4):
                    pipe.submit(x)
                pipe.synchronize()
                t0 = time.perf_counter()
                pend = []
                for _ in range(n):
                    pend.append(pipe.submit(x))
                    if len(pend) > args.overlap:
                        pend.pop(0)['done'].synchronize()
                pipe.synchronize()
                return (time.perf_counter() - t0) / n * 1e3
            for _ in range(3):
                net.forward_device(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                net.forward_device(x)['count'].tolist()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e3
        table = engine.load_tune_table(dev)
        out = {}
        base = statistics.median(step_ms(args.steps) for _ in range(3))
        print('plan: %s batch %d, %d tune misses, step %.3f ms' % (args.config, args.batch, plan.tune_misses, base), flush=True)
        for key, lst in sorted(plan.wino_toggle.items()):
            ent = table.get(key)
            if not ent or not ent[1]:
                continue
            ratio = ent[3] / ent[2] if ent[2] else 0.0
            names = [e[1][2] for e in lst]
            uses_wino = all(e[3] for e in lst) and table.get('instep|' + key) != 0
            if not uses_wino or ratio < args.max_margin:
                print('skip %-60s isolated wino/direct %.2f  (%d layers: %s)' % (key[:60], ratio, len(lst), ','.join(names)[:80]))
                continue
            tw, td = [], []
            for r in range(args.rounds):
                set_all(key, True)
                tw.append(step_ms(args.steps))
                set_all(key, False)
                td.append(step_ms(args.steps))
            set_all(key, True)
            mw, md = statistics.median(tw), statistics.median(td)
            # direct must win by more than the spread of the rounds
            spread = max(statistics.pstdev(tw), statistics.pstdev(td))
            win = md < mw - max(2.0 * spread, 0.002 * mw)
            out[key] = {'layers': names, 'isolated_ratio': round(ratio, 3), 'step_ms_winograd': round(mw, 4), 'step_ms_direct': round(md, 4),
                        'spread_ms': round(spread, 4), 'direct_wins_the_step': bool(win)}
            print('%-60s %d layers  isolated %.2f | step: winograd %.4f ms, direct %.4f ms (spread %.4f) -> %s'
                  % (key[:60], len(lst), ratio, mw, md, spread, 'DIRECT' if win else 'winograd'), flush=True)
            if win:
                set_all(key, False)           # later keys are judged on top of the decisions already taken
        final = statistics.median(step_ms(args.steps) for _ in range(3))
        print('step after the decisions: %.3f ms (was %.3f)' % (final, base))
    print(json.dumps({'config': args.config, 'batch': args.batch, 'overlap': args.overlap, 'base_ms': round(base, 4), 'final_ms': round(final, 4), 'keys': out}))
    if args.write and any(v['direct_wins_the_step'] for v in out.values()):
        path = os.path.join(engine.TUNE_DIR, 'gfx950.json')
        entries = engine._read_table_file(path)
        for k, v in out.items():
            if v['direct_wins_the_step']:
                entries['instep|' + k] = 0
        engine._write_table_file(path, entries, dev)
        print('wrote %d instep decisions to %s' % (sum(v['direct_wins_the_step'] for v in out.values()), path))


if __name__ == '__main__':
    main()
