#!/usr/bin/env python
"""Per-block timeline of one conv launch (ymi_debug_set_trace): which CU ran each block, when it started, how long
its prologue / K loop / epilogue took.   python tools/conv_trace.py SHAPE_INDEX TILE_ID [ablate]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolact_amd import _lib as L                      # noqa: E402
from yolact_amd.engine import Packed, out_size        # noqa: E402
from conv_probe import SHAPES                         # noqa: E402


def main():
    si, tile = int(sys.argv[1]), int(sys.argv[2])
    os.environ['YMI_ABLATE'] = sys.argv[3] if len(sys.argv) > 3 else '0'
    name, B, H, W, Cin, Cout, k, st, pad, has_res = SHAPES[si]
    dev = 'cuda:0'
    lib = L.lib()
    g = torch.Generator().manual_seed(si)
    pk = Packed(torch.randn(Cout, Cin, k, k, generator=g) * 0.05, torch.randn(Cout, generator=g), None, st, pad, None, dev)
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
    d.nseg, d.tile = 1, tile
    d.seg[0] = L.ConvSeg(0, Cout, L.ACT_RELU, Cout, Ho * Wo * Cout, y.data_ptr())
    s = L.stream_ptr()
    cap = 20000
    tr = torch.zeros(cap * 8, dtype=torch.int64, device=dev)
    for _ in range(3):
        L.check(lib.ymi_conv2d_nhwc_f32(C.byref(d), s))
    torch.cuda.synchronize()
    lib.ymi_debug_set_trace(tr.data_ptr(), cap)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    L.check(lib.ymi_conv2d_nhwc_f32(C.byref(d), s))
    e1.record()
    torch.cuda.synchronize()
    lib.ymi_debug_set_trace(None, 0)
    t = tr.cpu().numpy().reshape(cap, 8)
    t = t[t[:, 4] != 0]
    nb = len(t)
    hw = t[:, 0] & 0xffffffff
    xcc = (t[:, 0] >> 32) & 0xf
    cu, sh, se = (hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 0x7
    cuid = xcc * 1000 + se * 100 + sh * 10 + cu
    t0 = t[:, 1].min()
    st_, lp, ep, en, tp, tc = (t[:, i] - t0 for i in (1, 2, 3, 4, 5, 6))
    print('%s tile=%s blocks=%d  event=%.1f us  span=%d cyc (100MHz ticks? see ratio)' % (name, L.TILE_NAMES[tile], nb, e0.elapsed_time(e1) * 1e3, en.max()))
    print('ticks per us (span/event): %.1f' % (en.max() / (e0.elapsed_time(e1) * 1e3)))
    uniq, cnt = np.unique(cuid, return_counts=True)
    print('distinct CUs used: %d ; blocks per CU histogram: %s' % (len(uniq), dict(zip(*np.unique(cnt, return_counts=True)))))
    for nm, a in (('start', st_), ('prologue', lp - st_), ('kloop', ep - lp), ('epi.lds', tp - ep), ('epi.math', tc - tp), ('epi.store', en - tc), ('epilogue', en - ep), ('total', en - st_), ('end', en)):
        print('%-9s min %8d  p50 %8d  p90 %8d  max %8d' % (nm, a.min(), np.median(a), np.percentile(a, 90), a.max()))
    # per-CU finish time
    fin = {}
    for c, e in zip(cuid, en):
        fin[c] = max(fin.get(c, 0), e)
    f = np.array(list(fin.values()))
    print('per-CU finish: min %d p50 %d max %d' % (f.min(), np.median(f), f.max()))


if __name__ == '__main__':
    main()
