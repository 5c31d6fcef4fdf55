#!/usr/bin/env python
"""Phase time stamps of pipe_h2_k (csrc/dcn.hip) on the plan's own layers — where does a 30 us launch spend its time?

Diagnostics build only (`make -C yolact_amd/csrc DIAG=1`, or the `pipetrace` stage of tools/gpu_session.sh which rebuilds dcn.hip
alone): wave 0 of every block keeps s_memtime stamps in registers at the kernel's own synchronisation points and writes them behind
its last store (YMI_PIPE_TRACE = device address; YMI_PIPE_TRACE_MODE=2 adds a stamp that waits for the tensor scale).

    python tools/pipe_trace.py [--layers layer2.1.conv1,layer2.1.conv3] [--batch 8] [--reps 6] [--mode 1]

Per layer: the event-timed launch (no tracing), then per phase the mean / median / max over blocks in us, the chip-level view on the
100 MHz constant clock (first block start -> last block end; start spread; per-block residency) and the gap between consecutive
launches of the same kernel (last end of launch k -> first start of launch k + 1)."""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

PHASES = [('entry -> index math done', 0, 2), ('prologue: requests, chunk 0 combined', 2, 3), ('filter DMA wait + first barrier', 3, 4),
          ('MAIN LOOP', 4, 5), ('drain + barrier', 5, 6), ('acc -> LDS tile + sync', 6, 7), ('epilogue math + stores issued', 7, 8),
          ('stores acknowledged', 8, 9)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='yolact_resnet50_config')
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--reps', type=int, default=6)
    ap.add_argument('--mode', type=int, default=1)
    ap.add_argument('--layers', default='layer1.1.conv1,layer1.1.conv2,layer2.1.conv1,layer2.1.conv3,layer3.0.conv1,layer3.1.conv1,layer3.1.conv3')
    args = ap.parse_args()
    import yolact_amd
    from yolact_amd import _lib as L
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
        plan.run(x)
    torch.cuda.synchronize()
    lib = L.lib()
    s = L.stream_ptr()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    NB = 4096
    buf = torch.zeros(args.reps * NB * 16, dtype=torch.int64, device=dev)
    os.environ['YMI_PIPE_TRACE_MODE'] = str(args.mode)
    want = args.layers.split(',')
    for fn, arg, name, where in plan.ops:
        if fn is not lib.ymi_conv2d_nhwc_f32 or not any(name == w or (w.endswith('*') and name.startswith(w[:-1])) for w in want):
            continue
        d = arg.contents
        tile = d.tile
        if not (tile & L.TILE_DCNP):
            print('%s: tile %s is not a pipelined tile, skipped' % (name, L.TILE_NAMES.get(tile & 255, tile)))
            continue
        os.environ.pop('YMI_PIPE_TRACE', None)
        fn(arg, s)
        best = 1e30
        for _ in range(3):
            e0.record()
            for _ in range(10):
                fn(arg, s)
            e1.record(); e1.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10)
        buf.zero_()
        torch.cuda.synchronize()
        for r in range(args.reps):
            os.environ['YMI_PIPE_TRACE'] = str(buf.data_ptr() + 8 * 16 * NB * r)
            fn(arg, s)
        torch.cuda.synchronize()
        os.environ.pop('YMI_PIPE_TRACE', None)
        tr = buf.cpu().view(args.reps, NB, 16)
        nblk = int((tr[-1, :, 15] == 1).sum())
        M = d.B * d.Ho * d.Wo
        fl = 2.0 * M * d.Cout * d.kh * d.kw * d.Cin
        print('\n%s  B%d %dx%d k%d %d>%d  tile %s  %d blocks  untraced launch %.1f us (%.0f TF/s algorithmic)' % (
            name, d.B, d.H, d.W, d.kh, d.Cin, d.Cout, L.TILE_NAMES.get(tile & 255, tile),
            nblk, best * 1e3, fl / best / 1e9))
        t = tr[-1, :nblk].double()                       # the last repetition (steady state: filters / activations cache-warm as in a step)
        cyc = (t[:, 9] - t[:, 0])
        rt = (t[:, 11] - t[:, 10]) * 0.01                # us on the 100 MHz constant clock
        ghz = (cyc / rt.clamp_min(1e-3)).median().item() / 1e3
        print('  shader clock during the launch %.2f GHz; chunks per block %d' % (ghz, int(t[0, 14])))
        us = lambda c: c / (ghz * 1e3)
        print('  %-40s %8s %8s %8s   (us per block)' % ('phase', 'mean', 'median', 'max'))
        for nm, a, b in PHASES:
            dd = us(t[:, b] - t[:, a])
            extra = ''
            if nm == 'MAIN LOOP':
                extra = '   = %.3f us per chunk, %.0f cycles' % (dd.mean().item() / max(int(t[0, 14]), 1), (t[:, b] - t[:, a]).mean().item() / max(int(t[0, 14]), 1))
            print('  %-40s %8.2f %8.2f %8.2f%s' % (nm, dd.mean().item(), dd.median().item(), dd.max().item(), extra))
        if args.mode == 2:
            dd = us(t[:, 1] - t[:, 0])
            print('  %-40s %8.2f %8.2f %8.2f' % ('(entry -> tensor scale known)', dd.mean().item(), dd.median().item(), dd.max().item()))
        dd = us(cyc)
        print('  %-40s %8.2f %8.2f %8.2f' % ('block lifetime', dd.mean().item(), dd.median().item(), dd.max().item()))
        st0 = t[:, 10].min()
        start = (t[:, 10] - st0) * 0.01
        end = (t[:, 11] - st0) * 0.01
        print('  chip view (constant clock): first start 0.00, last start %.2f (median %.2f), first end %.2f, last end %.2f us' % (
            start.max().item(), start.median().item(), end.min().item(), end.max().item()))
        xcc = t[:, 13].long()
        cuid = (xcc << 8) | ((t[:, 12].long() >> 8) & 0xff)      # HW_ID[15:8] = cu_id, sh_id, se_id
        uniq, counts = torch.unique(cuid, return_counts=True)
        print('  placement: %d distinct CUs hold %d blocks (max %d per CU); blocks per XCD %s' % (
            uniq.numel(), nblk, int(counts.max()), torch.bincount(xcc, minlength=8).tolist()))
        gaps = []
        for r in range(1, args.reps):
            a, b = tr[r - 1, :nblk], tr[r, :nblk]
            gaps.append((b[:, 10].min() - a[:, 11].max()).item() * 0.01)
        print('  launch-to-launch gap (last end of k -> first start of k+1): %s us' % ' '.join('%.2f' % g for g in gaps))
        spans = [((tr[r, :nblk, 11].max() - tr[r, :nblk, 10].min()).item() * 0.01) for r in range(args.reps)]
        print('  in-kernel span per repetition: %s us' % ' '.join('%.1f' % g for g in spans))


if __name__ == '__main__':
    main()
