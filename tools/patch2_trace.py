#!/usr/bin/env python
"""Where a step of csrc/patch2.hip goes (diagnostics build, `patch2trace` stage of tools/gpu_session.sh): per layer and tile, the
untraced launch, then from wave 0 (a consumer) and wave 4 (a producer) of every block: prologue / loop / epilogue cycles, cycles per
(chunk, tap) step, the share of the loop a consumer spends at the step barrier and a producer spends waiting for memory / at the barrier.

    python tools/patch2_trace.py [--layers proto.8,proto.2] [--tiles patch2p256,patch2p192] [--shape 16x16]"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--layers', default='proto.8,proto.2,layer1.1.conv2')
    ap.add_argument('--tiles', default='patch2p256,patch2p192')
    ap.add_argument('--shape', default='')
    args = ap.parse_args()
    if args.shape:
        os.environ['YMI_PATCH2_TILE'] = args.shape
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
    NB = 8192
    buf = torch.zeros(NB * 32, dtype=torch.int64, device=dev)
    names = {v: k for k, v in L.TILE_NAMES.items()}
    descs = dict(plan.conv_meta)
    for name in args.layers.split(','):
        d0 = descs.get(name)
        if d0 is None:
            continue
        d = L.ConvDesc.from_buffer_copy(d0)
        M = d.B * d.Ho * d.Wo
        y = torch.empty(M * d.Cout, device=dev)
        d.nseg = 1
        d.seg[0] = L.ConvSeg(0, d.Cout, d0.seg[0].act, d.Cout, d.Ho * d.Wo * d.Cout, y.data_ptr())
        d.split_k = 0
        for tn in args.tiles.split(','):
            d.tile = names[tn]
            os.environ.pop('YMI_PATCH2_TRACE', None)
            if lib.ymi_conv2d_nhwc_f32(C.byref(d), s) != 0:
                print(name, tn, 'refused')
                continue
            best = 1e30
            for _ in range(3):
                e0.record()
                for _ in range(5):
                    lib.ymi_conv2d_nhwc_f32(C.byref(d), s)
                e1.record(); e1.synchronize()
                best = min(best, e0.elapsed_time(e1) / 5)
            buf.zero_()
            os.environ['YMI_PATCH2_TRACE'] = str(buf.data_ptr())
            for _ in range(2):
                lib.ymi_conv2d_nhwc_f32(C.byref(d), s)
            torch.cuda.synchronize()
            os.environ.pop('YMI_PATCH2_TRACE', None)
            tr = buf.cpu().view(NB, 2, 16).double()
            nblk = int((tr[:, 0, 15] == 1).sum())
            t = tr[:nblk]
            ns = int(t[0, 0, 14])
            fl = 2.0 * M * d.Cout * 9 * d.Cin
            print('%-16s %-10s %5d blocks (%.2f rounds of 256) %3d steps  launch %.1f us (%.0f TF/s)' % (name, tn, nblk, nblk / 256.0, ns, best * 1e3, fl / best / 1e9))
            for w, role in ((0, 'consumer'), (1, 'producer')):
                pro, loop, epi = (t[:, w, 1] - t[:, w, 0]).mean().item(), ((t[:, w, 4] if w == 0 else t[:, w, 5]) - t[:, w, 1]).mean().item(), (t[:, w, 5] - t[:, w, 4]).mean().item() if w == 0 else 0.0
                lp = (t[:, w, 4] if w == 0 else t[:, w, 5]) - t[:, w, 1]
                print('    %s: prologue %6.0f  loop %7.0f cycles (%4.0f per step)  epilogue %6.0f | in the loop: %2.0f %% at the barrier%s' % (
                    role, pro, loop, loop / ns, epi, 100 * (t[:, w, 3] / lp).mean().item(),
                    '' if w == 0 else ', %2.0f %% waiting for memory' % (100 * (t[:, w, 2] / lp).mean().item())))


if __name__ == '__main__':
    main()
