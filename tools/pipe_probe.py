#!/usr/bin/env python
"""The pipelined kernel of csrc/dcn.hip run as an ORDINARY convolution (ymi_conv2d_nhwc_f32 with a YMI_TILE_DCNP tile) against what
the shipped plan runs for the same layer (direct LDS-DMA tile, split-K, or the three Winograd launches), on the plan's own tensors.

    python tools/pipe_probe.py [--config yolact_resnet50_config] [--batch 8] [--reps 5] [--layers substr,substr]

Per eligible layer (3x3 / pad 1 or 1x1 / pad 0, Cin % 32 == 0, one dense output): ms of the plan's op, ms of the best pipelined
candidate (tile, K split), the deviation of its output from the plan's, and at the end the step-level sum of what would change.
"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='yolact_resnet50_config')
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--layers', default='')
    ap.add_argument('--no-ws', action='store_true', help='leave out the weight-stationary streaming kernel (csrc/wstat.hip)')
    ap.add_argument('--all-cands', action='store_true', help='print every candidate, not only the best')
    ap.add_argument('--ablate', default='', help='comma list of YMI_DCN_ABLATE masks (diagnostics build): the best candidate is re-timed per mask')
    ap.add_argument('--tiles', default='', help='restrict the pipelined candidates to these names (e.g. dcnp128x256w16,dcnp160x128w10/k2)')
    ap.add_argument('--plan-only', action='store_true', help='time the plan\'s own op of every eligible layer and nothing else (A/B of a rebuilt library)')
    args = ap.parse_args()
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

    def timed(f, a):
        f(a, s)
        best = 1e30
        for _ in range(2):
            e0.record()
            for _ in range(args.reps):
                f(a, s)
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1) / args.reps)
        return best

    def tname(v):
        return L.TILE_NAMES[v & 255] + ('/k%d' % (v >> 8) if v >> 8 else '')
    descs = dict(plan.conv_meta)
    tot_plan = tot_best = tot_fl = 0.0
    for fn, arg, name, where in plan.ops:
        if fn not in (lib.ymi_conv2d_nhwc_f32, lib.ymi_conv3x3_winograd_f32):
            continue
        base = name.replace('[wino]', '')
        d0 = descs.get(base)
        if d0 is None or (args.layers and not any(k in base for k in args.layers.split(','))):
            continue
        if not (((d0.kh, d0.kw, d0.pad) in ((3, 3, 1), (1, 1, 0))) and d0.Cin % 32 == 0 and d0.nseg == 1 and d0.Cout % 4 == 0
                and d0.seg[0].act <= L.ACT_LEAKY01 and d0.res_mode in (L.RES_NONE, L.RES_ADD) and d0.Kpad // 32 >= 2 and d0.w_h2):
            continue
        fl = lib.ymi_conv_flops(C.byref(d0))
        t_plan = timed(fn, arg)
        if args.plan_only:
            tot_plan += t_plan; tot_best += t_plan; tot_fl += fl
            print('%-22s B%d %3dx%-3d s%d k%d %4d>%-4d %6.2f GF | plan %-34s %.4f ms %6.1f TF/s' % (
                base, d0.B, d0.H, d0.W, d0.stride, d0.kh, d0.Cin, d0.Cout, fl / 1e9, name[-34:], t_plan, fl / t_plan / 1e9), flush=True)
            continue
        # a private copy of the direct descriptor writing to a scratch output
        d = L.ConvDesc.from_buffer_copy(d0)
        M = d.B * d.Ho * d.Wo
        yref = torch.empty(M * d.Cout, device=dev)
        real = d0.seg[0].ptr
        ylive = (C.c_float * (M * d.Cout)).from_address(real) if False else None
        ref = torch.empty(M * d.Cout, device=dev)
        # reference output = what the plan's op wrote (it ran in `timed`): copy it out through a torch view of the arena pointer
        ref_src = None
        for b in plan.arena.all + getattr(plan, 'arena_b', plan.arena).all:
            if b.data_ptr() <= real < b.data_ptr() + b.numel() * 4:
                off = (real - b.data_ptr()) // 4
                ref_src = b[off:off + M * d.Cout]
        if ref_src is not None:
            ref.copy_(ref_src)
        d.seg[0].ptr = yref.data_ptr()
        times, devmax = {}, {}
        for cand in plan.dcnp_candidates(d) + (plan.ws_candidates(d) if not args.no_ws else []) + plan.pc_candidates(d) + plan.patch2_candidates(d):
            if args.tiles and tname(cand) not in args.tiles.split(','):
                continue
            tile, S = cand & 255, cand >> 8
            d.tile = tile
            d.split_k = S if S > 1 else 0
            if S > 1:
                ws = plan._splitk_ws(where, S * M * d.Cout)
                d.split_ws = ws.data_ptr()
            if lib.ymi_conv2d_nhwc_f32(C.byref(d), s) != 0:
                continue
            torch.cuda.synchronize()
            if ref_src is not None:
                devmax[cand] = ((yref - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()
            times[cand] = timed(lib.ymi_conv2d_nhwc_f32, C.pointer(d))
        if not times:
            continue
        best = min(times, key=times.get)
        tot_plan += t_plan; tot_best += min(t_plan, times[best]); tot_fl += fl
        print('%-22s B%d %3dx%-3d s%d k%d %4d>%-4d %6.2f GF | plan %-34s %.4f ms %6.1f TF/s | pipelined %-18s %.4f ms %6.1f TF/s  x%.2f  dev %.1e' % (
            base, d.B, d.H, d.W, d.stride, d.kh, d.Cin, d.Cout, fl / 1e9, name[-34:], t_plan, fl / t_plan / 1e9, tname(best), times[best],
            fl / times[best] / 1e9, t_plan / times[best], devmax.get(best, float('nan'))), flush=True)
        if args.ablate:
            tile, S = best & 255, best >> 8
            d.tile, d.split_k = tile, (S if S > 1 else 0)
            if S > 1:
                d.split_ws = plan._splitk_ws(where, S * M * d.Cout).data_ptr()
            row = []
            for a in args.ablate.split(','):
                os.environ['YMI_DCN_ABLATE'] = a
                row.append('abl=%s %.4f' % (a, timed(lib.ymi_conv2d_nhwc_f32, C.pointer(d))))
            os.environ['YMI_DCN_ABLATE'] = '0'
            print('      %s: %s' % (tname(best), '  '.join(row)), flush=True)
        if args.all_cands:
            print('      ' + '  '.join('%s %.4f' % (tname(c), t) for c, t in sorted(times.items(), key=lambda kv: kv[1])[:10]))
    print('TOTAL eligible layers: plan %.3f ms, with the pipelined kernel where it wins %.3f ms (%.1f -> %.1f TF/s algorithmic)' % (
        tot_plan, tot_best, tot_fl / tot_plan / 1e9, tot_fl / tot_best / 1e9))


if __name__ == '__main__':
    main()
