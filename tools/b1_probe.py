#!/usr/bin/env python
"""Batch-1 latency anatomy (the reference's published metric is batch-1 FPS, README.md:70 / eval.py:264-281): forward + Detect
eager vs hipGraph replay, launches per step, host launch cost vs device time.
    python tools/b1_probe.py"""
import json
import os
import sys
import time

os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import yolact_amd                                             # noqa: E402
from yolact_amd.utils.synth import synth_images, synth_state_dict   # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    yolact_amd.set_cfg('yolact_resnet50_config')
    from yolact_amd.yolact import Yolact
    from yolact_amd.layers.output_utils import postprocess
    net = Yolact()
    net.load_state_dict_compat(synth_state_dict([(k, tuple(v.shape)) for k, v in net.state_dict().items()], seed=0, conf_gain=0.04))
    net.detect.use_fast_nms = True
    net = net.to(dev)
    x = synth_images(1, 550, 550, seed=1234).to(dev)
    out = {}

    def timed(fn, n=200):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    with torch.no_grad():
        plan = net.plan_for(x)
        nops = sum(1 for op in plan.ops if op[0] not in ('record', 'wait'))
        out['ops_per_step'] = nops
        out['winograd_layers'] = sum(1 for op in plan.ops if isinstance(op[2], str) and op[2].endswith('[wino]'))
        # host cost of issuing one step (no sync inside): launch-rate bound?
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            net.forward_device(x)
        t_issue = (time.perf_counter() - t0) / 50 * 1e3
        torch.cuda.synchronize()
        out['host_issue_ms_per_step'] = round(t_issue, 3)
        out['eager_two_streams_ms'] = round(timed(lambda: net.forward_device(x)['count'].tolist()), 3)
        plan.overlap = False
        out['eager_one_stream_ms'] = round(timed(lambda: net.forward_device(x)['count'].tolist()), 3)
        plan.overlap = True
        os.environ['YOLACT_AMD_GRAPH'] = '1'
        try:
            out['graph_replay_two_streams_ms'] = round(timed(lambda: net.forward_device(x)['count'].tolist()), 3)
            plan.overlap = False
            net._plans = {k: v for k, v in net._plans.items() if not (isinstance(k, tuple) and k and k[0] == 'graph')}
            out['graph_replay_one_stream_ms'] = round(timed(lambda: net.forward_device(x)['count'].tolist()), 3)
        except Exception as e:      # noqa: BLE001
            out['graph_error'] = repr(e)[:300]
        plan.overlap = True
        os.environ['YOLACT_AMD_GRAPH'] = '0'

        def ref_fps():
            preds = net(x)
            t = postprocess(preds, 550, 550, crop_masks=True, score_threshold=0)
            c, s, b, m = [v[:5] for v in t]
            s.cpu().numpy(); c.cpu().numpy(); b.cpu().numpy(); m.cpu().numpy()
            torch.cuda.synchronize()
        out['reference_fps_definition_ms'] = round(timed(ref_fps, 100), 3)
        out['net_call_ms'] = round(timed(lambda: net(x), 100), 3)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
