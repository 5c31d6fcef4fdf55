#!/usr/bin/env python
"""Winograd layers through the C ABI, per GEMM arithmetic: whole layer (3 launches) and the grouped GEMM alone (library
profiling records: HIP events on the launch stream), for the exact-fp32 / bf16x3 / fp16x2 tiles and fp16x2 with V written as
fp16 planes by the input transform (ymi_wino_desc.v_planes).  transforms = layer - GEMM.

    python tools/wino_probe.py [--reps 10] [--tiles 1,17,...]
"""
import argparse
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolact_amd import _lib as L                              # noqa: E402
from yolact_amd.engine import Packed, WinoPacked              # noqa: E402

SHAPES = [('proto.8   256>256 @138', 8, 138, 138, 256, 256), ('fpn.pred2 256>256 @69', 8, 69, 69, 256, 256),
          ('head0+p0  256>512 @69', 8, 69, 69, 256, 512), ('l2.conv2  256>256 @35', 8, 35, 35, 256, 256),
          ('l3.conv2  512>512 @18', 8, 18, 18, 512, 512)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--tiles', default='1,17,5')
    ap.add_argument('--m', default='4')
    ap.add_argument('--shapes', default='', help='comma-separated shape indices (default: all)')
    ap.add_argument('--batch', type=int, default=0, help='override the batch size of every shape')
    args = ap.parse_args()
    dev = 'cuda:0'
    lib = L.lib()
    out = {}
    for si, (name, B, H, W, Cin, Cout) in enumerate(SHAPES):
        if args.shapes and str(si) not in args.shapes.split(','):
            continue
        B = args.batch or B
        g = torch.Generator().manual_seed(1)
        w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.02
        pk = Packed(w, torch.randn(Cout, generator=g), None, 1, 1, None, dev)
        x = torch.relu(torch.randn(B, H, W, Cin, generator=g)).to(dev)
        y = torch.empty(B, H, W, Cout, device=dev)
        amax = torch.zeros(2 * 1024, device=dev)
        L.check(lib.ymi_amax_f32(x.data_ptr(), x.numel(), amax.data_ptr(), L.stream_ptr()))
        for m in [int(v) for v in args.m.split(',')]:
            wp = WinoPacked(w, dev, m)
            T = B * ((H + m - 1) // m) * ((W + m - 1) // m)
            G = (m + 2) ** 2
            V = torch.empty(G * T * Cin, device=dev)
            Mw = torch.empty(G * T * Cout, device=dev)
            d = L.WinoDesc()
            d.x, d.u, d.y, d.V, d.M, d.bias = x.data_ptr(), wp.u.data_ptr(), y.data_ptr(), V.data_ptr(), Mw.data_ptr(), pk.bias.data_ptr()
            d.B, d.H, d.W, d.C, d.Cout, d.act, d.m = B, H, W, Cin, Cout, L.ACT_RELU, m
            d.u_x3 = wp.u3().data_ptr()
            up, uinv = wp.h2()
            d.u_h2, d.uinv_h2, d.x_amax, d.y_amax = up.data_ptr(), uinv.data_ptr(), amax.data_ptr(), amax.data_ptr() + 4096
            alg = 2.0 * B * H * W * Cout * 9 * Cin
            for base in [int(t) for t in args.tiles.split(',')]:
                for label, flag, planes in (('fp32', 0, 0), ('x3', L.TILE_X3, 0), ('h2', L.TILE_H2, 0), ('h2+Vplanes', L.TILE_H2, 1)):
                    d.tile, d.v_planes = base | flag, planes
                    if lib.ymi_conv3x3_winograd_f32(C.byref(d), L.stream_ptr()) != 0:
                        continue
                    torch.cuda.synchronize()
                    lib.ymi_prof_reset(); lib.ymi_prof_enable(1)
                    for _ in range(args.reps):
                        lib.ymi_conv3x3_winograd_f32(C.byref(d), L.stream_ptr())
                    torch.cuda.synchronize()
                    lib.ymi_prof_enable(0)
                    ms, fl, tile, kind = C.c_float(), C.c_double(), C.c_int32(), C.c_int32()
                    lay = gem = 0.0
                    gfl = 0.0
                    for i in range(lib.ymi_prof_count()):
                        L.check(lib.ymi_prof_read(i, C.byref(ms), C.byref(fl), C.byref(tile), C.byref(kind)))
                        if kind.value in (3, 4):
                            lay += ms.value
                        else:
                            gem += ms.value; gfl = fl.value
                    lay /= args.reps; gem /= args.reps
                    key = '%s F(%dx%d) %s %s' % (name, m, m, L.TILE_NAMES.get(base, L.TILE_NAMES.get(base | L.TILE_H2, 'tile%d' % base)), label)
                    out[key] = {'layer_ms': round(lay, 4), 'gemm_ms': round(gem, 4), 'transforms_ms': round(lay - gem, 4),
                                'gemm_tflops_executed': round(gfl / gem / 1e9, 1), 'layer_tflops_algorithmic': round(alg / lay / 1e9, 1)}
                    print('%-52s layer %.4f ms  gemm %.4f (%6.1f TF/s executed)  transforms %.4f  alg %6.1f TF/s' % (
                        key, lay, gem, gfl / gem / 1e9, lay - gem, alg / lay / 1e9), flush=True)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
