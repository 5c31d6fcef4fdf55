#!/usr/bin/env python
"""Would two independent half-batch chains on two streams beat one batch-8 chain?  The backbone is a serial chain of
~50 launches whose tails (last partial round of blocks) leave CUs idle; two B=4 chains in separate hardware queues can
fill each other's tails.  Times plan.run only (no Detect):
    one B=8 plan | two B=4 plans (slots 0/1) on two user streams, each with / without its own side stream.
    python tools/halfbatch_probe.py [--steps 30]"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(fn, n):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=30)
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    import bench
    from yolact_amd.utils.synth import synth_images
    res = {}
    with torch.no_grad():
        net, sd = bench.build_model(dev, 550)
        x = synth_images(8, 550, 550, seed=1234).to(dev)
        xa, xb = x[:4].contiguous(), x[4:].contiguous()
        p8 = net.plan_for(x)
        pa, pb = net.plan_for(xa, slot=0), net.plan_for(xb, slot=1)
        res['tune_misses_b4'] = pa.tune_misses
        s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)

        def one():
            p8.run(x)

        def two():
            cur = torch.cuda.current_stream(dev)
            s1.wait_stream(cur); s2.wait_stream(cur)
            with torch.cuda.stream(s1):
                pa.run(xa)
            with torch.cuda.stream(s2):
                pb.run(xb)
            cur.wait_stream(s1); cur.wait_stream(s2)

        def seq():
            pa.run(xa); pb.run(xb)

        res['b8_ms'] = round(timed(one, args.steps), 3)
        res['2xb4_concurrent_ms'] = round(timed(two, args.steps), 3)
        res['2xb4_sequential_ms'] = round(timed(seq, args.steps), 3)
        for p in (p8, pa, pb):
            p.overlap = False
        res['b8_one_stream_ms'] = round(timed(one, args.steps), 3)
        res['2xb4_concurrent_one_stream_each_ms'] = round(timed(two, args.steps), 3)
    print(json.dumps(res))


if __name__ == '__main__':
    main()
