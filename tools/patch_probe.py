#!/usr/bin/env python
"""csrc/patch.hip on the layer it was built for (layer0.x.conv2 at batch 8: 138 x 138 x 64 -> 64, 3x3): time per launch against the
pipelined implicit-GEMM tiles, and — in a diagnostics build (-DYMI_DIAGNOSTICS) — the stall attribution YMI_PATCH_ABLATE selects.
    python tools/patch_probe.py [--ablate 1,2,4,8,16,32,3,63]"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ablate', default='')
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--size', type=int, default=138)
    args = ap.parse_args()
    import torch
    from yolact_amd import _lib as L
    from yolact_amd.engine import Packed
    D_ = 'cuda:0'
    B, S = args.batch, args.size
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, S, S, 64, generator=g).to(D_)
    pk = Packed(torch.randn(64, 64, 3, 3, generator=g) * 0.05, None, None, 1, 1, None, D_)
    y = torch.empty(B, S, S, 64, device=D_)
    amax = torch.zeros(2 * 1024, device=D_)
    L.check(L.lib().ymi_amax_f32(x.data_ptr(), x.numel(), amax.data_ptr(), L.stream_ptr()))
    hp, sc2, winv = pk.h2()

    def timed(tile):
        d = L.ConvDesc()
        d.x, d.w, d.B, d.H, d.W, d.Cin, d.ldx, d.Ho, d.Wo, d.Cout = x.data_ptr(), pk.w.data_ptr(), B, S, S, 64, 64, S, S, 64
        d.kh, d.kw, d.stride, d.pad, d.Kpad, d.nseg, d.tile = 3, 3, 1, 1, pk.Kpad, 1, tile
        d.seg[0] = L.ConvSeg(0, 64, L.ACT_RELU, 64, S * S * 64, y.data_ptr())
        d.w_h2, d.scale_h2, d.winv_h2, d.x_amax, d.y_amax = hp.data_ptr(), sc2.data_ptr(), winv.data_ptr(), amax.data_ptr(), amax.data_ptr() + 4096
        s = L.stream_ptr()
        for _ in range(3):
            L.check(L.lib().ymi_conv2d_nhwc_f32(C.byref(d), s))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            L.check(L.lib().ymi_conv2d_nhwc_f32(C.byref(d), s))
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / 20
    fl = 2 * B * S * S * 64 * 576
    for name, t in (('patch8x16c64', L.DCNP_PATCH_C64), ('dcnp128x64w8', L.DCNP_128x64_W8), ('dcnp64x64', L.DCNP_64x64)):
        ms = timed(t | L.TILE_H2 | L.TILE_DCNP)
        print('%-14s %.4f ms  %.1f TFLOP/s' % (name, ms, fl / ms / 1e9))
    for a in [int(v) for v in args.ablate.split(',') if v]:
        os.environ['YMI_PATCH_ABLATE'] = str(a)
        print('abl=%-3d %.4f ms' % (a, timed(L.DCNP_PATCH_C64 | L.TILE_H2 | L.TILE_DCNP)))
    os.environ.pop('YMI_PATCH_ABLATE', None)


if __name__ == '__main__':
    main()
