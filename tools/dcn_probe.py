#!/usr/bin/env python
"""Every DCNv2 launch of the YOLACT++ plan (BASELINE configs[3]: yolact_plus_resnet50_config, 550x550), every candidate tile:
the register-staged loader of csrc/conv_igemm.hip (exact-fp32 and fp16x2 tiles) against the pipelined gather-GEMM of
csrc/dcn.hip, on the plan's own tensors (offsets / mask logits as the network produces them from the synthetic weights).

    python tools/dcn_probe.py [--batch 8] [--reps 5]

Per layer: ms and TFLOP/s per tile, the maximum deviation of every pipelined tile from the old fp16x2 launch (both fp32-class:
they differ by rounding only), then the sum over the 13 layers of the best old tile and of the best pipelined tile.
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
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--config', default='yolact_plus_resnet50_config')
    ap.add_argument('--tiles', default='', help='comma list of pipelined tile names to time (default: all); old tiles are skipped when given')
    ap.add_argument('--layers', default='', help='comma list of substrings of layer names (default: all DCN layers)')
    ap.add_argument('--ablate', default='', help='comma list of YMI_DCN_ABLATE masks (diagnostics build only): each tile is re-timed per mask')
    args = ap.parse_args()
    os.environ.setdefault('YOLACT_AMD_AUTOTUNE', 'table')
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
    old = [t for t in L.BASIC_TILES if t != L.TILE_128x32]
    old = old + [t | L.TILE_H2 for t in old]
    def tname(v):
        return L.TILE_NAMES[v & 255] + ('/k%d' % (v >> 8) if v >> 8 else '')
    if args.tiles:
        old = [L.TILE_64x64 | L.TILE_H2]
    abls = [int(a) for a in args.ablate.split(',')] if args.ablate else []
    tot_old = tot_new = tot_fl = 0.0
    for fn, dptr, name, where in plan.ops:
        if fn is not lib.ymi_dcn_v2_forward_f32 or (args.layers and not any(k in name for k in args.layers.split(','))):
            continue
        dd = dptr.contents
        d = dd.conv
        fl = lib.ymi_conv_flops(C.byref(d))
        tile0 = d.tile + 256 * max(d.split_k, 0) * (1 if d.split_k > 1 else 0)
        M = d.B * d.Ho * d.Wo
        y = torch.empty(M * d.Cout, device=dev)
        yptr0 = d.seg[0].ptr
        d.seg[0].ptr = y.data_ptr()
        times, ref, dev_max = {}, None, {}
        new = plan.dcnp_candidates(d, dcn=True)
        if args.tiles:
            new = [t for t in new if tname(t) in args.tiles.split(',')]
        for t in old + new:
            if plan._apply_choice(fn, dptr, where, t, s) != 0:
                continue
            torch.cuda.synchronize()
            if t == (L.TILE_64x64 | L.TILE_H2):
                ref = y.clone()
            elif (t & 255) & L.TILE_DCNP and ref is not None:
                dev_max[t] = ((y - ref).abs().max() / ref.abs().max()).item()
            best = 1e30
            for _ in range(2):
                e0.record()
                for _ in range(args.reps):
                    fn(dptr, s)
                e1.record()
                e1.synchronize()
                best = min(best, e0.elapsed_time(e1) / args.reps)
            times[t] = best
            if (t & 255) & L.TILE_DCNP and abls:
                row = []
                for a in abls:
                    os.environ['YMI_DCN_ABLATE'] = str(a)
                    fn(dptr, s)
                    e0.record()
                    for _ in range(args.reps):
                        fn(dptr, s)
                    e1.record()
                    e1.synchronize()
                    row.append('abl=%d %.4f' % (a, e0.elapsed_time(e1) / args.reps))
                os.environ['YMI_DCN_ABLATE'] = '0'
                print('    %-14s %s full %.4f | %s' % (tname(t), name, best, '  '.join(row)))
        d.seg[0].ptr = yptr0
        plan._apply_choice(fn, dptr, where, tile0, s)
        bo = min((times[t], t) for t in old if t in times)
        bn = min((times[t], t) for t in new if t in times) if any(t in times for t in new) else (float('nan'), 0)
        tot_old += bo[0]; tot_new += bn[0]; tot_fl += fl
        print('%-16s B%d %dx%d s%d %d>%d  %.2f GFLOP  | old best %-10s %.4f ms %6.1f TF/s | pipelined best %-14s %.4f ms %6.1f TF/s' % (
            name, d.B, d.H, d.W, d.stride, d.Cin, d.Cout, fl / 1e9, tname(bo[1]), bo[0], fl / bo[0] / 1e9,
            tname(bn[1]) if bn[1] else '-', bn[0], fl / bn[0] / 1e9))
        print('    ' + '  '.join('%s %.4f' % (tname(t), times[t]) for t in old + new if t in times))
        print('    max |pipelined - old fp16x2| / max|y|: %.1e .. %.1e over %d pipelined candidates' % (
            min(dev_max.values()), max(dev_max.values()), len(dev_max)))
    print('TOTAL %d DCN layers: old %.3f ms (%.1f TF/s), pipelined %.3f ms (%.1f TF/s)' % (
        sum(1 for op in plan.ops if op[0] is lib.ymi_dcn_v2_forward_f32), tot_old, tot_fl / tot_old / 1e9, tot_new, tot_fl / tot_new / 1e9))


if __name__ == '__main__':
    main()
