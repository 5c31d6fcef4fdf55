#!/usr/bin/env python
"""Where a chunk step of csrc/wgemm.hip goes (diagnostics build, `wgemmtrace` stage of tools/gpu_session.sh): the Winograd layers of the
headline plan through ymi_conv3x3_winograd_f32 with tile wg128x256h2, stamps from wave 0 (a consumer) and wave 4 (a producer)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolact_amd import _lib as L                              # noqa: E402
from yolact_amd.engine import Packed, WinoPacked              # noqa: E402

SHAPES = [('proto.8   256>256 @138', 8, 138, 138, 256, 256), ('fpn.pred2 256>256 @69', 8, 69, 69, 256, 256),
          ('head0+p0  256>512 @69', 8, 69, 69, 256, 512), ('l3.conv2  512>512 @18', 8, 18, 18, 512, 512)]


def main():
    dev = 'cuda:0'
    lib = L.lib()
    for name, B, H, W, Cin, Cout in SHAPES:
        g = torch.Generator().manual_seed(1)
        w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.02
        pk = Packed(w, torch.randn(Cout, generator=g), None, 1, 1, None, dev)
        x = torch.relu(torch.randn(B, H, W, Cin, generator=g)).to(dev)
        y = torch.empty(B, H, W, Cout, device=dev)
        amax = torch.zeros(2 * 1024, device=dev)
        L.check(lib.ymi_amax_f32(x.data_ptr(), x.numel(), amax.data_ptr(), L.stream_ptr()))
        m = 4
        wp = WinoPacked(w, dev, m)
        T = B * ((H + m - 1) // m) * ((W + m - 1) // m)
        G = (m + 2) ** 2
        V = torch.empty(G * T * Cin, device=dev)
        Mw = torch.empty(G * T * Cout, device=dev)
        d = L.WinoDesc()
        d.x, d.u, d.y, d.V, d.M, d.bias = x.data_ptr(), wp.u.data_ptr(), y.data_ptr(), V.data_ptr(), Mw.data_ptr(), pk.bias.data_ptr()
        d.B, d.H, d.W, d.C, d.Cout, d.act, d.m = B, H, W, Cin, Cout, L.ACT_RELU, m
        up, uinv = wp.h2()
        d.u_h2, d.uinv_h2, d.x_amax, d.y_amax = up.data_ptr(), uinv.data_ptr(), amax.data_ptr(), amax.data_ptr() + 4096
        d.tile, d.v_planes = L.TILE_WG_128x256 | L.TILE_H2, 1
        buf = torch.zeros(512 * 32, dtype=torch.int64, device=dev)
        L.check(lib.ymi_conv3x3_winograd_f32(C.byref(d), L.stream_ptr()))
        os.environ['YMI_WGEMM_TRACE'] = str(buf.data_ptr())
        for _ in range(2):
            L.check(lib.ymi_conv3x3_winograd_f32(C.byref(d), L.stream_ptr()))
        torch.cuda.synchronize()
        os.environ.pop('YMI_WGEMM_TRACE', None)
        # ablation (wrong results by design): event-timed GEMM launch with parts of the traffic / the MFMAs removed
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        row = []
        for a, nm in ((0, 'full'), (16, 'nt stores'), (1, 'no M stores'), (2, 'no V'), (4, 'no U'), (8, 'no MFMA'), (6, 'no V, U'), (7, 'no memory'), (15, 'nothing')):
            os.environ['YMI_WGEMM_ABLATE'] = str(a)
            lib.ymi_prof_reset(); lib.ymi_prof_enable(1)
            for _ in range(5):
                lib.ymi_conv3x3_winograd_f32(C.byref(d), L.stream_ptr())
            torch.cuda.synchronize()
            lib.ymi_prof_enable(0)
            ms, fl, tile, kind = C.c_float(), C.c_double(), C.c_int32(), C.c_int32()
            gem = 0.0
            for i in range(lib.ymi_prof_count()):
                L.check(lib.ymi_prof_read(i, C.byref(ms), C.byref(fl), C.byref(tile), C.byref(kind)))
                if kind.value in (5, 6):
                    gem += ms.value
            row.append('%s %.1f' % (nm, gem / 5 * 1e3))
        os.environ['YMI_WGEMM_ABLATE'] = '0'
        print('    GEMM launch, us: ' + ' | '.join(row))
        tr = buf.cpu().view(512, 2, 16).double()
        nb = int((tr[:, 0, 15] == 1).sum())
        t = tr[:nb]
        ns = t[:, 0, 14]
        items = G * ((T + 127) // 128) * ((Cout + 255) // 256)
        print('%-24s %d items on %d blocks, %d chunk steps per block' % (name, items, nb, int(ns.max())))
        for wv, role in ((0, 'consumer'), (1, 'producer')):
            tot = t[:, wv, 5] - t[:, wv, 0]
            pro = t[:, wv, 1] - t[:, wv, 0]
            print('    %s: total %.0f cycles, prologue %.0f, per chunk step %.0f | at the barrier %.0f %%%s' % (
                role, tot.mean(), pro.mean(), ((tot - pro) / ns).mean(), 100 * (t[:, wv, 3] / (tot - pro)).mean(),
                '' if wv == 0 else ', waiting for memory %.0f %%' % (100 * (t[:, wv, 2] / (tot - pro)).mean())))


if __name__ == '__main__':
    main()
