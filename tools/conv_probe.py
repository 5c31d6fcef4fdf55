#!/usr/bin/env python
"""Time representative YOLACT conv shapes through the C ABI: every tile shape x ablation mask (YMI_ABLATE; needs the
diagnostics build: make -C yolact_amd/csrc clean all DIAG=1 — the product build ignores the variable).

    python tools/conv_probe.py [--ablate 0,1,3,7] [--reps 20]

Prints one line per (shape, tile, ablation): ms, TF/s, fraction of the 157.3 TF/s fp32 MFMA peak.  Diagnostics
only (ablated runs compute garbage); used to attribute main-loop time to global loads / LDS stores / barriers.
"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolact_amd import _lib as L                      # noqa: E402
from yolact_amd.engine import Packed, out_size        # noqa: E402

# name, B, H, W, Cin, Cout, k, stride, pad, residual
SHAPES = [
    ('proto.8   3x3 256>256 @138', 8, 138, 138, 256, 256, 3, 1, 1, 0),
    ('fpn.pred2 3x3 256>256 @69', 8, 69, 69, 256, 256, 3, 1, 1, 0),
    ('l2.conv2  3x3 256>256 @35', 8, 35, 35, 256, 256, 3, 1, 1, 0),
    ('l3.conv2  3x3 512>512 @18', 8, 18, 18, 512, 512, 3, 1, 1, 0),
    ('l2.conv3  1x1 256>1024 @35', 8, 35, 35, 256, 1024, 1, 1, 0, 1),
    ('l2.conv1  1x1 1024>256 @35', 8, 35, 35, 1024, 256, 1, 1, 0, 0),
    ('l1.conv3  1x1 128>512 @69', 8, 69, 69, 128, 512, 1, 1, 0, 1),
    ('l0.conv3  1x1 64>256 @138', 8, 138, 138, 64, 256, 1, 1, 0, 1),
    ('l0.conv1  1x1 256>64 @138', 8, 138, 138, 256, 64, 1, 1, 0, 0),
    ('l0.conv2  3x3 64>64 @138', 8, 138, 138, 64, 64, 3, 1, 1, 0),
    ('l3.conv1  1x1 2048>512 @18', 8, 18, 18, 2048, 512, 1, 1, 0, 0),
    ('l3.conv3  1x1 512>2048 @18', 8, 18, 18, 512, 2048, 1, 1, 0, 1),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ablate', default='0')
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--tiles', default='')
    ap.add_argument('--shapes', default='')
    ap.add_argument('--batch', type=int, default=0, help='override the batch size of every shape')
    ap.add_argument('--cold', type=int, default=0, help='1: zero the output bound slot before every launch (what a real run sees)')
    ap.add_argument('--amax', type=int, default=1, help='0: launches do not report max|y| (A/B of the epilogue atomic)')
    args = ap.parse_args()
    dev = 'cuda:0'
    lib = L.lib()
    abls = [int(a) for a in args.ablate.split(',')]
    tiles = [int(t) for t in args.tiles.split(',')] if args.tiles else sorted(L.TILE_NAMES)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for si, (name, B, H, W, Cin, Cout, k, st, pad, has_res) in enumerate(SHAPES):
        if args.shapes and str(si) not in args.shapes.split(','):
            continue
        B = args.batch or B
        g = torch.Generator().manual_seed(si)
        w = torch.randn(Cout, Cin, k, k, generator=g) * 0.05
        pk = Packed(w, torch.randn(Cout, generator=g), None, st, pad, None, dev)
        x = torch.randn(B, H, W, Cin, generator=g).to(dev)
        Ho, Wo = out_size(H, k, st, pad), out_size(W, k, st, pad)
        y = torch.empty(B, Ho, Wo, Cout, device=dev)
        res = torch.randn(B, Ho, Wo, Cout, generator=g).to(dev) if has_res else None
        d = L.ConvDesc()
        d.x, d.w, d.bias = x.data_ptr(), pk.w.data_ptr(), pk.bias.data_ptr()
        d.B, d.H, d.W, d.Cin, d.ldx = B, H, W, Cin, Cin
        d.Ho, d.Wo, d.Cout = Ho, Wo, Cout
        d.kh, d.kw, d.stride, d.pad, d.Kpad = k, k, st, pad, pk.Kpad
        if has_res:
            d.res, d.res_ld, d.res_mode = res.data_ptr(), Cout, L.RES_ADD
        d.nseg = 1
        d.seg[0] = L.ConvSeg(0, Cout, L.ACT_RELU, Cout, Ho * Wo * Cout, y.data_ptr())
        if os.environ.get('PROBE_PLANES', '1') == '1':
            d.w_x3 = pk.w3().data_ptr()          # bf16x3 tiles: pre-split filter planes (PROBE_PLANES=0: split both on the fly)
        amax = torch.zeros(2 * 1024, device=dev)        # fp16x2 tiles: filter planes, folded scales, the input's magnitude bound
        L.check(lib.ymi_amax_f32(x.data_ptr(), x.numel(), amax.data_ptr(), L.stream_ptr()))
        hp, sc2, winv = pk.h2()
        d.w_h2, d.scale_h2, d.winv_h2, d.x_amax = hp.data_ptr(), sc2.data_ptr(), winv.data_ptr(), amax.data_ptr()
        d.y_amax = amax.data_ptr() + 4096 if args.amax else None
        fl = lib.ymi_conv_flops(C.byref(d))
        s = L.stream_ptr()
        for t in tiles:
            if (Cout <= 64 and L.TILE_NAMES[t].endswith('x128')):
                continue
            d.tile = t
            for a in abls:
                os.environ['YMI_ABLATE'] = str(a)
                rc = lib.ymi_conv2d_nhwc_f32(C.byref(d), s)
                if rc != 0:
                    print('%-28s %-8s abl=%d rc=%d' % (name, L.TILE_NAMES[t], a, rc))
                    continue
                e0.record()
                for _ in range(args.reps):
                    if args.cold:
                        amax[1024:].zero_()
                    lib.ymi_conv2d_nhwc_f32(C.byref(d), s)
                e1.record()
                e1.synchronize()
                ms = e0.elapsed_time(e1) / args.reps
                print('%-28s %-8s abl=%d %8.4f ms %7.1f TF/s %5.1f%%' % (name, L.TILE_NAMES[t], a, ms, fl / ms / 1e9,
                                                                       fl / ms / 1e9 / 157.3 * 100))
        os.environ['YMI_ABLATE'] = '0'


if __name__ == '__main__':
    main()
