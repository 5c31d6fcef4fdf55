#!/usr/bin/env python
"""Whole-step tile selection with two batches in flight (round 6).

engine.Plan._tune_direct picks the tile of a convolution launch by its ISOLATED time on resident buffers: the objective of a latency
plan.  The regime bench.py times (yolact_amd.pipeline.BatchPipeline, depth 2) is a throughput regime: the launches of the next batch
fill whatever a launch leaves idle, so a tile that occupies fewer CU-cycles can win the step although it loses alone (split-K and
small tiles buy latency with CU time).  This tool measures that directly: per table key (layer shape) the few candidates within
`--window` of the best isolated time are installed in every plan slot and the WHOLE pipelined step is timed, alternating with the
current choice (median of `--rounds`); a candidate is taken when it wins the step by more than the noise floor.  `--write` persists
the winners in yolact_amd/tune/gfx950.json (same keys as the isolated tuner).

    python tools/overlap_tune.py [--config yolact_resnet50_config --batch 8] [--overlap 2] [--window 1.35 --top 3] [--write]
"""
import argparse
import ctypes as C
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='yolact_resnet50_config')
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--size', type=int, default=0)
    ap.add_argument('--overlap', type=int, default=2)
    ap.add_argument('--rounds', type=int, default=5)
    ap.add_argument('--steps', type=int, default=40)
    ap.add_argument('--window', type=float, default=1.35, help='candidates within this factor of the best isolated time')
    ap.add_argument('--top', type=int, default=3, help='at most this many alternatives per key')
    ap.add_argument('--only', default='', help='substring of the layer names to restrict the search to')
    ap.add_argument('--write', action='store_true')
    args = ap.parse_args()
    import torch
    import yolact_amd
    from yolact_amd import engine, _lib as L
    from yolact_amd.pipeline import BatchPipeline
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
    lib = L.lib()
    with torch.no_grad():
        plans = [net.plan_for(x, k) for k in range(args.overlap)]
        pipe = BatchPipeline(net, args.overlap)
        s = L.stream_ptr()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

        def step_ms(n):
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

        def cname(v):
            return L.TILE_NAMES.get(v & 255, '?%d' % (v & 255)) + ('/k%d' % (v >> 8) if v >> 8 else '')

        # the direct convolution launches of slot 0's plan, grouped by table key; the same op index in every slot
        groups = {}
        p0 = plans[0]
        for opi, (fn, dptr, name, where) in enumerate(p0.ops):
            is_dcn = fn is lib.ymi_dcn_v2_forward_f32
            if (fn is not lib.ymi_conv2d_nhwc_f32 and not is_dcn) or opi in p0.wide_ops:
                continue
            d = dptr.contents.conv if is_dcn else dptr.contents
            key = (d.B, d.H, d.W, d.Cin, d.Cout, d.kh, d.kw, d.stride, d.pad, d.res_mode, d.nseg, d.Kpad) + (('dcn',) if is_dcn else ())
            skey = str(key) + ('|x3' if p0.split else '|h2' if p0.h2 else '')
            groups.setdefault(skey, []).append(opi)

        def cur_val(opi):
            fn, dptr, _, _ = p0.ops[opi]
            d = dptr.contents.conv if fn is lib.ymi_dcn_v2_forward_f32 else dptr.contents
            return int(d.tile) + 256 * (int(d.split_k) if d.split_k > 1 else 0)

        def install(opis, val):
            for p_ in plans:
                for opi in opis:
                    fn, dptr, _, where = p_.ops[opi]
                    if p_._apply_choice(fn, dptr, where, val, s) != 0:
                        return False
            return True

        base = statistics.median(step_ms(args.steps) for _ in range(3))
        print('plan: %s batch %d overlap %d, %d tune misses, %d table keys, step %.4f ms' % (
            args.config, args.batch, args.overlap, p0.tune_misses, len(groups), base), flush=True)
        out = {}
        for skey, opis in sorted(groups.items(), key=lambda kv: -len(kv[1])):
            names = [p0.ops[i][2] for i in opis]
            if args.only and not any(args.only in n for n in names):
                continue
            fn, dptr, name, where = p0.ops[opis[0]]
            d = dptr.contents.conv if fn is lib.ymi_dcn_v2_forward_f32 else dptr.contents
            cur = cur_val(opis[0])
            # isolated times of every candidate (the tuner's own measurement), on slot 0's first layer of the key
            iso = {}
            for t in p0.direct_candidates(fn, d):
                if p0._apply_choice(fn, dptr, where, t, s) != 0:
                    continue
                best = 1e30
                for _ in range(2):
                    e0.record()
                    for _ in range(3):
                        fn(dptr, s)
                    e1.record()
                    e1.synchronize()
                    best = min(best, e0.elapsed_time(e1) / 3)
                iso[t] = best
            p0._apply_choice(fn, dptr, where, cur, s)
            torch.cuda.synchronize()
            if cur not in iso:
                print('skip %-70s current choice %s not among the candidates' % (skey[:70], cname(cur)))
                continue
            tmin = min(iso.values())
            alts = sorted((t for t in iso if t != cur and iso[t] <= args.window * tmin), key=iso.get)[:args.top]
            res = {'layers': names, 'current': cname(cur), 'isolated_ms': {cname(t): round(iso[t], 4) for t in [cur] + alts}, 'step_ms': {}}
            best_val, best_ms = cur, None
            for t in alts:
                ta, tb = [], []
                for r in range(args.rounds):
                    install(opis, best_val)
                    ta.append(step_ms(args.steps))
                    if not install(opis, t):
                        break
                    tb.append(step_ms(args.steps))
                if len(tb) < args.rounds:
                    install(opis, best_val)
                    continue
                ma, mb = statistics.median(ta), statistics.median(tb)
                spread = max(statistics.pstdev(ta), statistics.pstdev(tb))
                res['step_ms'][cname(t)] = [round(ma, 4), round(mb, 4), round(spread, 4)]
                if mb < ma - max(2.0 * spread, 0.0015 * ma):
                    best_val, best_ms = t, mb
            install(opis, best_val)
            res['chosen'] = cname(best_val)
            out[skey] = (res, best_val, cur)
            print('%-62s %2d layers  %-22s -> %-22s %s' % (skey[:62], len(opis), cname(cur), cname(best_val),
                  '  '.join('%s iso %.4f step %.4f vs %.4f' % (cname(t), iso[t], res['step_ms'][cname(t)][1], res['step_ms'][cname(t)][0])
                            for t in alts if cname(t) in res['step_ms'])), flush=True)
        final = statistics.median(step_ms(args.steps) for _ in range(3))
        print('step after the decisions: %.4f ms (was %.4f)' % (final, base))
    print(json.dumps({'config': args.config, 'batch': args.batch, 'overlap': args.overlap, 'base_ms': round(base, 4), 'final_ms': round(final, 4),
                      'keys': {k: v[0] for k, v in out.items()}}))
    changed = {k: v[1] for k, v in out.items() if v[1] != v[2]}
    if args.write and changed:
        path = os.path.join(engine.TUNE_DIR, 'gfx950.json')
        entries = engine._read_table_file(path)
        for k, v in changed.items():
            entries[k] = int(v)
        engine._write_table_file(path, entries, dev)
        print('wrote %d entries to %s' % (len(changed), path))


if __name__ == '__main__':
    main()
