#!/usr/bin/env python
"""ymi_pointwise_chain_f32 (csrc/chain.hip) at the shape it exists for: the first ResNet stage at 138 x 138 x batch.

    python tools/chain_probe.py [--batch 8] [--reps 10] [--sets 3]

Times the launch in its three forms (conv3 + residual -> conv1; without the second layer; without the residual) on `--sets`
rotating buffer sets (3 x 390 MB: nothing survives in the 256 MB memory-side cache between launches, like in a real step, where
the operands were written by the previous layer and everything else has passed through since), and prints the bytes per second of
the launch's algorithmic traffic.  The two launches of the plan it replaces: layer0.N.conv3 0.083 + layer0.(N+1).conv1 0.055 ms
in the step (profiles/r04_layers.txt), 0.0745 + 0.035 ms alone on resident buffers (profiles/r04_ws_probe.txt).
"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--sets', type=int, default=3)
    args = ap.parse_args()
    from yolact_amd import _lib as L
    from yolact_amd.engine import Packed
    dev = torch.device('cuda', 0)
    M = args.batch * 138 * 138
    g = torch.Generator().manual_seed(1)
    wa = torch.randn(256, 64, 1, 1, generator=g) / 8
    wb = torch.randn(64, 256, 1, 1, generator=g) / 16
    pa, pb = Packed(wa, torch.randn(256, generator=g), None, 1, 0, None, dev), Packed(wb, torch.randn(64, generator=g), None, 1, 0, None, dev)
    pla, sca, _ = pa.h2()
    plb, scb, _ = pb.h2()
    sets = []
    for _ in range(args.sets):
        x = torch.randn(M, 64, device=dev).relu_()
        res = torch.randn(M, 256, device=dev).relu_()
        sets.append((x, res, torch.empty(M, 256, device=dev), torch.empty(M, 64, device=dev)))
    amax = torch.zeros(3 * 1024, device=dev)
    lib, s = L.lib(), L.stream_ptr()
    L.check(lib.ymi_amax_f32(sets[0][0].data_ptr(), sets[0][0].numel(), amax.data_ptr(), s))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def desc(k, two, with_res):
        x, res, y, z = sets[k]
        d = L.ChainDesc()
        d.x, d.y, d.M, d.ldx, d.ldy = x.data_ptr(), y.data_ptr(), M, 64, 256
        d.w_a_h2, d.scale_a_h2, d.bias_a, d.cout_pad_a = pla.data_ptr(), sca.data_ptr(), pa.bias.data_ptr(), pa.CoutPad
        d.k_a, d.n_a, d.n_b, d.act_a, d.act_b = 64, 256, 64, L.ACT_RELU, L.ACT_RELU
        d.x_amax, d.y_amax, d.z_amax = amax.data_ptr(), amax.data_ptr() + 4096, amax.data_ptr() + 8192
        if with_res:
            d.res, d.res_ld = res.data_ptr(), 256
        if two:
            d.z, d.ldz, d.w_b_h2, d.scale_b_h2, d.bias_b, d.cout_pad_b = z.data_ptr(), 64, plb.data_ptr(), scb.data_ptr(), pb.bias.data_ptr(), pb.CoutPad
        return d
    for name, two, with_res in (('conv3 + residual -> conv1', True, True), ('conv3 + residual', False, True), ('conv3 -> conv1, no residual', True, False)):
        ds = [desc(k, two, with_res) for k in range(args.sets)]
        for d in ds:
            L.check(lib.ymi_pointwise_chain_f32(C.byref(d), s))
        best = 1e30
        for _ in range(3):
            e0.record()
            for r in range(args.reps):
                lib.ymi_pointwise_chain_f32(C.byref(ds[r % args.sets]), s)
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1) / args.reps)
        nbytes = 4.0 * M * (64 + 256 + (256 if with_res else 0) + (64 if two else 0))
        print('%-30s M=%d  %.4f ms  %6.1f MB  %.2f TB/s' % (name, M, best, nbytes / 1e6, nbytes / best / 1e9))


if __name__ == '__main__':
    main()
