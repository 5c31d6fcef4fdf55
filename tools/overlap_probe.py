#!/usr/bin/env python
"""Why did the timed step lose its two-stream overlap once RCCL was initialised (round 2, session 1: forward + Detect
7.08 ms with a world-1 process group vs 6.3 ms in round 1 without one, same kernels)?  Times forward_device + the host read
of the counts under: no process group | group initialised BEFORE the model / its side stream exist | AFTER the plan (and
its side stream) exist; and prices the exchange step itself (dist.gather vs all_gather_into_tensor at world 1).
    python tools/overlap_probe.py --pg none|before|after [--steps 30]
Run each arm in its own process (stream -> hardware-queue assignment is per process)."""
import argparse
import os
import socket
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def init_pg(dev):
    import torch.distributed as dist
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % port, rank=0, world_size=1, device_id=dev)
    t = torch.zeros(4, device=dev)
    dist.all_reduce(t)                      # forces communicator creation
    torch.cuda.synchronize()


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
    ap.add_argument('--pg', default='none')
    ap.add_argument('--steps', type=int, default=30)
    args = ap.parse_args()
    import torch.distributed as dist
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    if args.pg == 'before':
        init_pg(dev)
    import bench
    from yolact_amd import parallel
    from yolact_amd.utils.synth import synth_images
    with torch.no_grad():
        net, sd = bench.build_model(dev, 550)
        x = synth_images(8, 550, 550, seed=1234).to(dev)
        net.forward_device(x)
        torch.cuda.synchronize()
        if args.pg == 'after':
            init_pg(dev)
        res = {'pg': args.pg, 'GPU_MAX_HW_QUEUES': os.environ.get('GPU_MAX_HW_QUEUES')}
        res['fwd_ms'] = round(timed(lambda: net.forward_device(x)['count'].tolist(), args.steps), 3)
        plan = net.plan_for(x)
        plan.overlap = False
        res['fwd_one_stream_ms'] = round(timed(lambda: net.forward_device(x)['count'].tolist(), args.steps), 3)
        plan.overlap = True
        # host cost of one forward's launches alone (no sync inside the loop): the CPU must stay ahead of the GPU
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            net.forward_device(x)
        res['host_issue_ms'] = round((time.perf_counter() - t0) / 10 * 1e3, 3)
        torch.cuda.synchronize()
        out = net.forward_device(x)
        rec = parallel.pack_records(out)
        res['pack_ms'] = round(timed(lambda: parallel.pack_records(out), 50), 3)
        if dist.is_initialized():
            res['gather_forced_ms'] = round(timed(lambda: parallel.gather_records(rec, force_collective=True), 50), 3)
            buf = torch.empty_like(rec)
            res['all_gather_into_tensor_ms'] = round(timed(lambda: dist.all_gather_into_tensor(buf, rec), 50), 3)
            res['step_with_gather_ms'] = round(timed(lambda: parallel.gather_records(parallel.pack_records(
                net.forward_device(x)), force_collective=True)[:, 0].tolist(), args.steps), 3)

            def step_ag():
                r = parallel.pack_records(net.forward_device(x))
                b = torch.empty_like(r)
                dist.all_gather_into_tensor(b, r)
                return b[:, 0].tolist()
            res['step_with_all_gather_ms'] = round(timed(step_ag, args.steps), 3)
        print(res, flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
