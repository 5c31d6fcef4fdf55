#!/usr/bin/env python
"""Fused bottleneck launch (csrc/bottleneck.hip, ymi_bottleneck_f32) against (a) an fp64 torch reference on a small case with
ragged borders and (b) the three separate fp16x2 conv launches it replaces, timed on the ResNet layer1 shape (138^2, 256 -> 64 ->
64 -> 256, batch 8).
    python tools/bneck_probe.py [--batch 8] [--size 138]"""
import argparse
import ctypes as C
import json
import os
import sys

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolact_amd import _lib as L                      # noqa: E402
from yolact_amd.engine import Packed                  # noqa: E402

DEV = 'cuda:0'


def make_block(P, g):
    C4 = 4 * P
    def bn(c):
        m = nn.BatchNorm2d(c)
        m.weight.data = torch.rand(c, generator=g) + 0.5
        m.bias.data = torch.randn(c, generator=g) * 0.2
        m.running_mean = torch.randn(c, generator=g) * 0.1
        m.running_var = torch.rand(c, generator=g) + 0.5
        return m.eval()
    w1 = torch.randn(P, C4, 1, 1, generator=g) * (1.0 / C4 ** 0.5)
    w2 = torch.randn(P, P, 3, 3, generator=g) * (1.0 / (9 * P) ** 0.5)
    w3 = torch.randn(C4, P, 1, 1, generator=g) * (1.0 / P ** 0.5)
    return (w1, bn(P)), (w2, bn(P)), (w3, bn(C4))


def reference(x_nhwc, blk):
    x = x_nhwc.permute(0, 3, 1, 2).double()
    t = x
    for i, (w, b) in enumerate(blk):
        t = torch.nn.functional.conv2d(t, w.double(), padding=1 if w.shape[-1] == 3 else 0)
        inv = 1.0 / torch.sqrt(b.running_var.double() + b.eps)
        sc = (b.weight.double() * inv).view(1, -1, 1, 1)
        sh = (b.bias.double() - b.running_mean.double() * b.weight.double() * inv).view(1, -1, 1, 1)
        t = t * sc + sh
        if i < 2:
            t = torch.relu(t)
    return torch.relu(t + x).permute(0, 2, 3, 1).contiguous()


def fused_desc(x, y, pks, amax_x, amax_y, B, H, W, P):
    d = L.BneckDesc()
    d.x, d.y, d.B, d.H, d.W, d.P = x.data_ptr(), y.data_ptr(), B, H, W, P
    keep = []
    for i, pk in enumerate(pks):
        hp, sc2, winv = pk.h2()
        keep.append((hp, sc2))
        setattr(d, 'w%d_h2' % (i + 1), hp.data_ptr())
        setattr(d, 'cout_pad%d' % (i + 1), pk.CoutPad)
        setattr(d, 'scale%d' % (i + 1), sc2.data_ptr())
        setattr(d, 'bias%d' % (i + 1), pk.bias.data_ptr())
    d.x_amax, d.y_amax = amax_x, amax_y
    return d, keep


def conv_desc(x, y, pk, res, ax, ay, B, H, W, Cin, tile):
    d = L.ConvDesc()
    d.x, d.w, d.bias = x.data_ptr(), pk.w.data_ptr(), pk.bias.data_ptr()
    d.scale = pk.scale.data_ptr()
    d.B, d.H, d.W, d.Cin, d.ldx = B, H, W, Cin, Cin
    d.Ho, d.Wo, d.Cout = H, W, pk.Cout
    d.kh, d.kw, d.stride, d.pad, d.Kpad = pk.kh, pk.kw, 1, pk.pad, pk.Kpad
    if res is not None:
        d.res, d.res_ld, d.res_mode = res.data_ptr(), pk.Cout, L.RES_ADD
    d.nseg = 1
    d.seg[0] = L.ConvSeg(0, pk.Cout, L.ACT_RELU, pk.Cout, H * W * pk.Cout, y.data_ptr())
    hp, sc2, winv = pk.h2()
    d.w_h2, d.scale_h2, d.winv_h2 = hp.data_ptr(), sc2.data_ptr(), winv.data_ptr()
    d.x_amax, d.y_amax = ax, ay
    d.tile = tile | L.TILE_H2
    return d


def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--size', type=int, default=138)
    args = ap.parse_args()
    lib = L.lib()
    s = L.stream_ptr()
    P = 64
    g = torch.Generator().manual_seed(5)
    blk = make_block(P, g)
    pks = [Packed(w, None, b, 1, 1 if w.shape[-1] == 3 else 0, None, DEV) for (w, b) in blk]
    out = {}
    # ---- numerics on a small ragged case -----------------------------------------------------------------------------
    B, H, W = 2, 37, 43
    x = (torch.relu(torch.randn(B, H, W, 4 * P, generator=g)) * 1.7)
    x[0, 5, 7, :] *= 40.0                                   # one hot pixel: the per-tile scales differ between tiles
    ref = reference(x, blk)
    xd = x.to(DEV)
    y = torch.zeros(B, H, W, 4 * P, device=DEV)
    amax = torch.zeros(4 * 1024, device=DEV)
    L.check(lib.ymi_amax_f32(xd.data_ptr(), xd.numel(), amax.data_ptr(), s))
    ap_ = [amax.data_ptr() + 4096 * i for i in range(4)]
    d, keep = fused_desc(xd, y, pks, ap_[0], ap_[3], B, H, W, P)
    L.check(lib.ymi_bottleneck_f32(C.byref(d), s), 'bottleneck')
    torch.cuda.synchronize()
    err = (y.cpu().double() - ref).abs().max().item() / ref.abs().max().item()
    # the three separate launches on the same data
    t1 = torch.empty(B, H, W, P, device=DEV); t2 = torch.empty(B, H, W, P, device=DEV); y3 = torch.empty(B, H, W, 4 * P, device=DEV)
    ds = [conv_desc(xd, t1, pks[0], None, ap_[0], ap_[1], B, H, W, 4 * P, L.TILE_64x64),
          conv_desc(t1, t2, pks[1], None, ap_[1], ap_[2], B, H, W, P, L.TILE_64x64),
          conv_desc(t2, y3, pks[2], xd, ap_[2], ap_[3], B, H, W, P, L.TILE_64x64)]
    for dd in ds:
        L.check(lib.ymi_conv2d_nhwc_f32(C.byref(dd), s), 'conv')
    torch.cuda.synchronize()
    err3 = (y3.cpu().double() - ref).abs().max().item() / ref.abs().max().item()
    out['numerics'] = {'fused_err_of_max': err, 'three_launch_err_of_max': err3,
                       'fused_vs_three_launch': (y - y3).abs().max().item() / ref.abs().max().item(),
                       'y_amax_slot': amax[3 * 1024:4 * 1024].max().item(), 'y_max': y.max().item()}
    print(json.dumps(out['numerics']), flush=True)
    # ---- timing on the network's shape ---------------------------------------------------------------------------------
    B, H, W = args.batch, args.size, args.size
    x = torch.relu(torch.randn(B, H, W, 4 * P, generator=g)).to(DEV)
    y = torch.empty_like(x)
    t1 = torch.empty(B, H, W, P, device=DEV); t2 = torch.empty(B, H, W, P, device=DEV); y3 = torch.empty_like(x)
    amax.zero_()
    L.check(lib.ymi_amax_f32(x.data_ptr(), x.numel(), amax.data_ptr(), s))
    d, keep = fused_desc(x, y, pks, ap_[0], ap_[3], B, H, W, P)
    ms_f = timed(lambda: lib.ymi_bottleneck_f32(C.byref(d), s))
    nblk = ((H + 7) // 8) * ((W + 15) // 16) * B
    tr = torch.zeros(nblk * 8, dtype=torch.int64, device=DEV)
    os.environ['YMI_BNECK_TRACE'] = str(tr.data_ptr())
    lib.ymi_bottleneck_f32(C.byref(d), s); torch.cuda.synchronize()
    del os.environ['YMI_BNECK_TRACE']
    tv = tr.cpu().view(-1, 8).double()
    names = ['conv1 K loop', 't1 epilogue', 'conv2 K loop', 't2 epilogue', 'conv3 K loop', 'output']
    out['phases_cycles_mean'] = {n: round((tv[:, i + 1] - tv[:, i]).mean().item()) for i, n in enumerate(names)}
    out['phases_cycles_mean']['block'] = round((tv[:, 6] - tv[:, 0]).mean().item())
    out['phases_cycles_mean']['kernel_span'] = round((tv[:, 6].max() - tv[:, 0].min()).item())
    print(json.dumps(out['phases_cycles_mean']), flush=True)
    best = []
    for i, (src, dst, pk, res, cin) in enumerate(((x, t1, pks[0], None, 4 * P), (t1, t2, pks[1], None, P), (t2, y3, pks[2], x, P))):
        bt, bn_ = 1e9, None
        for tile in (L.TILE_64x64, L.TILE_64x64_S3, L.TILE_128x64, L.TILE_128x64_S3, L.TILE_64x128, L.TILE_64x128_S3, L.TILE_128x128):
            dd = conv_desc(src, dst, pk, res, ap_[i], ap_[i + 1], B, H, W, cin, tile)
            if lib.ymi_conv2d_nhwc_f32(C.byref(dd), s) != 0:
                continue
            ms = timed(lambda: lib.ymi_conv2d_nhwc_f32(C.byref(dd), s), 10)
            if ms < bt:
                bt, bn_ = ms, L.TILE_NAMES[tile | L.TILE_H2]
        best.append((bn_, round(bt, 4)))
    px = B * H * W
    flops = 2.0 * px * (4 * P * P + 9 * P * P + 4 * P * P)
    out['timing'] = {'shape': [B, H, W, 4 * P], 'fused_ms': round(ms_f, 4), 'fused_tflops': round(flops / ms_f / 1e9, 1),
                     'three_launches_ms': round(sum(b[1] for b in best), 4), 'three_launches': best,
                     'fused_io_GBps': round(2 * x.numel() * 4 / ms_f / 1e6, 1)}
    print(json.dumps(out['timing']))


if __name__ == '__main__':
    main()
