#!/usr/bin/env python
"""Would the small-map ResNet stages (35^2 / 18^2: few output tiles, long K, launch tails) run faster as TWO half-batch chains on two
streams than as one batch-8 chain?  (Round 2 tried it for the whole network and lost to L2 thrash on the big maps; on the small maps
the working set is a few MB.)  One bottleneck = 1x1 (4P -> P) + ReLU, 3x3 (P -> P) + ReLU, 1x1 (P -> 4P) + residual + ReLU, fp16x2
tiles, each shape with the best of a few candidate tiles (measured here, per batch size).
    python tools/halfchain_probe.py"""
import ctypes as C
import json
import os
import sys

os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolact_amd import _lib as L                      # noqa: E402
from yolact_amd.engine import Packed                  # noqa: E402

DEV = 'cuda:0'
CANDS = [L.TILE_64x64, L.TILE_64x64_S3, L.TILE_64x128, L.TILE_64x128_S3, L.TILE_128x64, L.TILE_128x128, L.TILE_128x128_S3,
         L.TILE_128x128_W8_S3, L.TILE_256x128_W8_S3, L.TILE_32x64_K2, L.TILE_64x32_K2]


def make_desc(x, y, pk, res, amax_x, amax_y, B, H, W, Cin):
    d = L.ConvDesc()
    d.x, d.w, d.bias = x.data_ptr(), pk.w.data_ptr(), pk.bias.data_ptr()
    d.B, d.H, d.W, d.Cin, d.ldx = B, H, W, Cin, Cin
    d.Ho, d.Wo, d.Cout = H, W, pk.Cout
    d.kh, d.kw, d.stride, d.pad, d.Kpad = pk.kh, pk.kw, 1, pk.pad, pk.Kpad
    if res is not None:
        d.res, d.res_ld, d.res_mode = res.data_ptr(), pk.Cout, L.RES_ADD
    d.nseg = 1
    d.seg[0] = L.ConvSeg(0, pk.Cout, L.ACT_RELU, pk.Cout, H * W * pk.Cout, y.data_ptr())
    hp, sc2, winv = pk.h2()
    d.w_h2, d.scale_h2, d.winv_h2 = hp.data_ptr(), sc2.data_ptr(), winv.data_ptr()
    d.x_amax, d.y_amax = amax_x, amax_y
    return d


def best_tile(lib, d, s, reps=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best, bt = None, 1e9
    for t in CANDS:
        d.tile = t | L.TILE_H2
        if lib.ymi_conv2d_nhwc_f32(C.byref(d), s) != 0:
            continue
        e0.record()
        for _ in range(reps):
            lib.ymi_conv2d_nhwc_f32(C.byref(d), s)
        e1.record(); e1.synchronize()
        ms = e0.elapsed_time(e1) / reps
        if ms < bt:
            best, bt = t, ms
    d.tile = best | L.TILE_H2
    return L.TILE_NAMES[best | L.TILE_H2], bt


def main():
    lib = L.lib()
    out = {}
    for name, H, P in (('layer2 35^2 P=256', 35, 256), ('layer3 18^2 P=512', 18, 512)):
        g = torch.Generator().manual_seed(1)
        pk1 = Packed(torch.randn(P, 4 * P, 1, 1, generator=g) * 0.03, torch.randn(P, generator=g) * 0.1, None, 1, 0, None, DEV)
        pk2 = Packed(torch.randn(P, P, 3, 3, generator=g) * 0.02, torch.randn(P, generator=g) * 0.1, None, 1, 1, None, DEV)
        pk3 = Packed(torch.randn(4 * P, P, 1, 1, generator=g) * 0.03, torch.randn(4 * P, generator=g) * 0.1, None, 1, 0, None, DEV)
        x = torch.relu(torch.randn(8, H, H, 4 * P, generator=g)).to(DEV)
        t1 = torch.empty(8, H, H, P, device=DEV); t2 = torch.empty(8, H, H, P, device=DEV); y = torch.empty(8, H, H, 4 * P, device=DEV)
        amax = torch.zeros(4 * 1024, device=DEV)
        L.check(lib.ymi_amax_f32(x.data_ptr(), x.numel(), amax.data_ptr(), L.stream_ptr()))
        ap = [amax.data_ptr() + 4096 * i for i in range(4)]
        sa = torch.cuda.current_stream()
        sb = torch.cuda.Stream()
        s_ptr = C.c_void_p(sa.cuda_stream)

        def chain(B, off):
            """descriptors of one bottleneck over images [off, off + B)"""
            xs, t1s, t2s, ys = (t[off:off + B] for t in (x, t1, t2, y))
            return [make_desc(xs, t1s, pk1, None, ap[0], ap[1], B, H, H, 4 * P), make_desc(t1s, t2s, pk2, None, ap[1], ap[2], B, H, H, P),
                    make_desc(t2s, ys, pk3, xs, ap[2], ap[3], B, H, H, P)]
        full = chain(8, 0)
        tiles8 = [best_tile(lib, d, s_ptr) for d in full]
        halves = [chain(4, 0), chain(4, 4)]
        tiles4 = [best_tile(lib, d, s_ptr) for d in halves[0]]
        for d0, d1 in zip(halves[0], halves[1]):
            d1.tile = d0.tile
        torch.cuda.synchronize()
        N = 6                                            # bottlenecks in a row (the stage), joined after each block like a real plan

        def run_full():
            for _ in range(N):
                for d in full:
                    lib.ymi_conv2d_nhwc_f32(C.byref(d), s_ptr)

        ev_f, ev_j = torch.cuda.Event(), torch.cuda.Event()
        sb_ptr = C.c_void_p(sb.cuda_stream)

        def run_halves():
            for _ in range(N):
                ev_f.record(sa); sb.wait_event(ev_f)
                for d in halves[0]:
                    lib.ymi_conv2d_nhwc_f32(C.byref(d), s_ptr)
                for d in halves[1]:
                    lib.ymi_conv2d_nhwc_f32(C.byref(d), sb_ptr)
                ev_j.record(sb); sa.wait_event(ev_j)

        def run_halves_serial():
            for _ in range(N):
                for d in halves[0] + halves[1]:
                    lib.ymi_conv2d_nhwc_f32(C.byref(d), s_ptr)

        def timed(fn, n=20):
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn()
            e1.record(); e1.synchronize()
            return e0.elapsed_time(e1) / n / N
        out[name] = {'batch8_chain_ms_per_block': round(timed(run_full), 4), 'tiles_b8': tiles8,
                     'two_half_chains_two_streams_ms_per_block': round(timed(run_halves), 4), 'tiles_b4': tiles4,
                     'two_half_chains_one_stream_ms_per_block': round(timed(run_halves_serial), 4)}
        print(name, json.dumps(out[name]), flush=True)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
