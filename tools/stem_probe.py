#!/usr/bin/env python
"""ymi_stem_pool_f32 (csrc/stem.hip: NCHW image -> 7x7/2 conv + BN + ReLU -> 3x3/2 max-pool -> NHWC in one launch) against an fp64
torch reference on a ragged size, and timed against the three launches it replaces (layout change + bound, stem conv, max-pool).
    python tools/stem_probe.py [--batch 8] [--size 550]"""
import argparse
import ctypes as C
import json
import os
import sys

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolact_amd import _lib as L                      # noqa: E402
from yolact_amd.engine import Packed, out_size        # noqa: E402

DEV = 'cuda:0'


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
    ap.add_argument('--size', type=int, default=550)
    args = ap.parse_args()
    lib, s = L.lib(), L.stream_ptr()
    g = torch.Generator().manual_seed(7)
    w = torch.randn(64, 3, 7, 7, generator=g) * 0.08
    bn = nn.BatchNorm2d(64)
    bn.weight.data = torch.rand(64, generator=g) + 0.5
    bn.bias.data = torch.randn(64, generator=g) * 0.2
    bn.running_mean = torch.randn(64, generator=g) * 0.1
    bn.running_var = torch.rand(64, generator=g) + 0.5
    bn.eval()
    pk = Packed(w, None, bn, 2, 3, 4, DEV)
    hp, sc2, winv = pk.h2()
    out = {}

    def desc(x, y, amax_y):
        d = L.StemDesc()
        d.x, d.y, d.B, d.H, d.W, d.cout_pad = x.data_ptr(), y.data_ptr(), x.shape[0], x.shape[2], x.shape[3], pk.CoutPad
        d.w_h2, d.scale_h2, d.bias, d.y_amax, d.kpad = hp.data_ptr(), sc2.data_ptr(), pk.bias.data_ptr(), amax_y, pk.Kpad
        return d
    # ---- numerics -------------------------------------------------------------------------------------------------------
    B, H, W = 2, 61, 77
    x = torch.randn(B, 3, H, W, generator=g) * 1.3
    x[1, :, 40:44, 50:54] *= 30.0
    t = torch.nn.functional.conv2d(x.double(), w.double(), stride=2, padding=3)
    inv = 1.0 / torch.sqrt(bn.running_var.double() + bn.eps)
    t = t * (bn.weight.double() * inv).view(1, -1, 1, 1) + (bn.bias.double() - bn.running_mean.double() * bn.weight.double() * inv).view(1, -1, 1, 1)
    ref = torch.nn.functional.max_pool2d(torch.relu(t), 3, 2, 1).permute(0, 2, 3, 1).contiguous()
    xd = x.to(DEV)
    y = torch.zeros(B, ref.shape[1], ref.shape[2], 64, device=DEV)
    amax = torch.zeros(2 * 1024, device=DEV)
    d = desc(xd, y, amax.data_ptr())
    L.check(lib.ymi_stem_pool_f32(C.byref(d), s), 'stem')
    torch.cuda.synchronize()
    out['numerics'] = {'shape': list(y.shape), 'err_of_max': (y.cpu().double() - ref).abs().max().item() / ref.abs().max().item(),
                       'y_amax_slot': amax[:1024].max().item(), 'y_max': y.max().item()}
    print(json.dumps(out['numerics']), flush=True)
    # ---- timing -----------------------------------------------------------------------------------------------------------
    B, H = args.batch, args.size
    x = torch.randn(B, 3, H, H, generator=g).to(DEV)
    Hs = out_size(H, 7, 2, 3); Hp = out_size(Hs, 3, 2, 1)
    y = torch.empty(B, Hp, Hp, 64, device=DEV)
    d = desc(x, y, amax.data_ptr())
    ms_f = timed(lambda: lib.ymi_stem_pool_f32(C.byref(d), s))
    # the three launches of the plan
    x4 = torch.empty(B, H, H, 4, device=DEV); st = torch.empty(B, Hs, Hs, 64, device=DEV); y3 = torch.empty_like(y)
    cd = L.ConvDesc()
    cd.x, cd.w, cd.bias, cd.scale = x4.data_ptr(), pk.w.data_ptr(), pk.bias.data_ptr(), pk.scale.data_ptr()
    cd.B, cd.H, cd.W, cd.Cin, cd.ldx, cd.Ho, cd.Wo, cd.Cout = B, H, H, 4, 4, Hs, Hs, 64
    cd.kh, cd.kw, cd.stride, cd.pad, cd.Kpad, cd.nseg, cd.cin_alg = 7, 7, 2, 3, pk.Kpad, 1, 3
    cd.seg[0] = L.ConvSeg(0, 64, L.ACT_RELU, 64, Hs * Hs * 64, st.data_ptr())
    cd.w_h2, cd.scale_h2, cd.winv_h2 = hp.data_ptr(), sc2.data_ptr(), winv.data_ptr()
    cd.x_amax, cd.y_amax = amax.data_ptr() + 4096, amax.data_ptr()
    cd.tile = L.TILE_128x64 | L.TILE_H2

    def three():
        lib.ymi_nchw_to_nhwc4_amax_f32(x.data_ptr(), x4.data_ptr(), B, 3, H, H, amax.data_ptr() + 4096, s)
        lib.ymi_conv2d_nhwc_f32(C.byref(cd), s)
        lib.ymi_maxpool3x3s2_nhwc_f32(st.data_ptr(), y3.data_ptr(), B, Hs, Hs, 64, Hp, Hp, s)
    L.check(lib.ymi_conv2d_nhwc_f32(C.byref(cd), s), 'stem conv')
    ms_3 = timed(three)
    torch.cuda.synchronize()
    out['timing'] = {'input': [B, 3, H, H], 'fused_ms': round(ms_f, 4), 'three_launches_ms': round(ms_3, 4),
                     'fused_vs_three_max_diff_of_max': (y - y3).abs().max().item() / y3.abs().max().item()}
    print(json.dumps(out['timing']))


if __name__ == '__main__':
    main()
