#!/usr/bin/env python
"""Where a step of the producer / consumer kernel (csrc/pcconv.hip) goes — diagnostics build (`pctrace` stage of tools/gpu_session.sh).

    python tools/pc_trace.py [--layers layer2.1.conv1,...] [--tiles pc128x128] [--flags 0,8,16]

Per (layer, tile, YMI_PC_FLAGS): untraced launch time, then from wave 0 (a consumer) and wave 4 (a producer) of every block: prologue,
main loop, epilogue in us, and inside the loop the share a consumer spends at the step barrier (= waiting for the producers) and the
shares a producer spends waiting for memory (counted vmcnt) and at the barrier (= waiting for the consumers)."""
import argparse
import ctypes as C
import os
import subprocess
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--layers', default='layer1.1.conv1,layer1.1.conv2,layer2.1.conv1,layer2.1.conv3,layer3.0.conv1,layer3.1.conv1')
    ap.add_argument('--tiles', default='pc128x128')
    ap.add_argument('--flags', default='')
    ap.add_argument('--child', default='')
    args = ap.parse_args()
    if args.flags and not args.child:          # YMI_PC_FLAGS is read once per process: one child per value
        for f in args.flags.split(','):
            env = dict(os.environ, YMI_PC_FLAGS=f)
            subprocess.call([sys.executable, os.path.abspath(__file__), '--batch', str(args.batch), '--layers', args.layers, '--tiles', args.tiles,
                             '--child', f], env=env)
        return
    import yolact_amd
    from yolact_amd import _lib as L
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
        plan = net.plan_for(x)
        plan.run(x)
    torch.cuda.synchronize()
    lib = L.lib()
    s = L.stream_ptr()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    NB = 4096
    buf = torch.zeros(NB * 32, dtype=torch.int64, device=dev)
    names = {v: k for k, v in L.TILE_NAMES.items()}
    print('YMI_PC_FLAGS=%s' % os.environ.get('YMI_PC_FLAGS', '0'))
    for fn, arg, name, where in plan.ops:
        if fn is not lib.ymi_conv2d_nhwc_f32 or name not in args.layers.split(','):
            continue
        d = L.ConvDesc.from_buffer_copy(arg.contents)
        M = d.B * d.Ho * d.Wo
        y = torch.empty(M * d.Cout, device=dev)
        d.seg[0].ptr = y.data_ptr()
        d.split_k = 0
        for tn in args.tiles.split(','):
            d.tile = names[tn]
            os.environ.pop('YMI_PIPE_TRACE', None)
            if lib.ymi_conv2d_nhwc_f32(C.byref(d), s) != 0:
                continue
            best = 1e30
            for _ in range(3):
                e0.record()
                for _ in range(10):
                    lib.ymi_conv2d_nhwc_f32(C.byref(d), s)
                e1.record(); e1.synchronize()
                best = min(best, e0.elapsed_time(e1) / 10)
            buf.zero_()
            os.environ['YMI_PIPE_TRACE'] = str(buf.data_ptr())
            for _ in range(3):
                lib.ymi_conv2d_nhwc_f32(C.byref(d), s)
            torch.cuda.synchronize()
            os.environ.pop('YMI_PIPE_TRACE', None)
            tr = buf.cpu().view(NB, 2, 16).double()
            nblk = int((tr[:, 0, 15] == 1).sum())
            t = tr[:nblk]
            ghz = (((t[:, 0, 7] - t[:, 0, 0]) / ((t[:, 0, 11] - t[:, 0, 10]) * 0.01).clamp_min(1e-3)).median().item()) / 1e3
            us = lambda c: (c / (ghz * 1e3)).mean().item()
            nk = int(t[0, 0, 14])
            fl = 2.0 * M * d.Cout * d.kh * d.kw * d.Cin
            print('%-16s %-10s %4d blocks %2d chunks  launch %.1f us (%.0f TF/s)  clock %.2f GHz' % (name, tn, nblk, nk, best * 1e3, fl / best / 1e9, ghz))
            for w, role in ((0, 'consumer'), (1, 'producer')):
                pro, loop, epi = us(t[:, w, 2] - t[:, w, 0]), us(t[:, w, 3] - t[:, w, 2]), us(t[:, w, 7] - t[:, w, 3])
                cyc = ((t[:, w, 3] - t[:, w, 2]) / nk).mean().item()
                bar = (t[:, w, 6] / (t[:, w, 3] - t[:, w, 2])).mean().item()
                vm = (t[:, w, 5] / (t[:, w, 3] - t[:, w, 2])).mean().item()
                print('    %s: prologue %.2f  loop %.2f (%.0f cycles per chunk)  epilogue %.2f us | in the loop: %.0f %% at the barrier%s' % (
                    role, pro, loop, cyc, epi, 100 * bar, '' if w == 0 else ', %.0f %% waiting for memory' % (100 * vm)))


if __name__ == '__main__':
    main()
