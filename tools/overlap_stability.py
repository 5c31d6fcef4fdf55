#!/usr/bin/env python
"""Is the two-batches-in-flight gain stable within a process and from process to process?  (round 6)

One process: the bench's model, a BatchPipeline of depth 2; `--rounds` times: serial step (one plan, forward_device) and pipelined step
(pipe.submit), 30 steps each, with an idle pause between the rounds (an idle hardware queue may be unmapped by the scheduler and come
back elsewhere).  Prints one line per round; run it several times to see the process-to-process spread.
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rounds', type=int, default=6)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--pause', type=float, default=0.3)
    ap.add_argument('--batch', type=int, default=8)
    args = ap.parse_args()
    import torch
    import yolact_amd
    from yolact_amd.pipeline import BatchPipeline
    from yolact_amd.utils.synth import synth_images, synth_state_dict
    yolact_amd.set_cfg('yolact_resnet50_config')
    from yolact_amd.yolact import Yolact
    dev = torch.device('cuda', 0)
    net = Yolact()
    net.load_state_dict_compat(synth_state_dict([(k, tuple(v.shape)) for k, v in net.state_dict().items()], seed=0, conf_gain=0.04))
    net.detect.use_fast_nms = True
    net = net.to(dev)
    x = synth_images(args.batch, 550, 550, seed=1234).to(dev)
    with torch.no_grad():
        pipe = BatchPipeline(net, 2)
        pipe.warm(x)

        def serial(n):
            for _ in range(3):
                net.forward_device(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pend = []
            for _ in range(n):
                o = net.forward_device(x)
                ev = torch.cuda.Event(); ev.record()
                pend.append(ev)
                if len(pend) > 2:
                    pend.pop(0).synchronize()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e3

        def piped(n):
            for _ in range(4):
                pipe.submit(x)
            pipe.synchronize()
            t0 = time.perf_counter()
            pend = []
            for _ in range(n):
                pend.append(pipe.submit(x))
                if len(pend) > 2:
                    pend.pop(0)['done'].synchronize()
            pipe.synchronize()
            return (time.perf_counter() - t0) / n * 1e3
        out = []
        for r in range(args.rounds):
            a, b = serial(args.steps), piped(args.steps)
            out.append('%.3f/%.3f' % (a, b))
            time.sleep(args.pause)
        print('pid %d serial/pipelined ms per step: %s' % (os.getpid(), '  '.join(out)), flush=True)


if __name__ == '__main__':
    main()
