"""Helpers for the GPU parity tests: drive single kernels through the C ABI from torch tensors."""
from __future__ import annotations

import ctypes as C

import torch

from yolact_amd import _lib as L
from yolact_amd.engine import Packed, out_size

DEV = 'cuda:0'


def nhwc(x_nchw):
    return x_nchw.permute(0, 2, 3, 1).contiguous()


def nchw(x_nhwc):
    return x_nhwc.permute(0, 3, 1, 2).contiguous()


def run_conv(x, weight, bias=None, bn=None, stride=1, pad=0, act=L.ACT_NONE, res=None, res_mode=L.RES_NONE,
             res_after_act=0, tile=L.TILE_AUTO, cin_pad=None, dcn_offmask=None, planes=True, split_k=0, om_layout=0, seg_bounds=None,
             seg_acts=None):
    """x: CPU NCHW tensor. Returns CPU NCHW output of the HIP conv.  `planes`: for bf16x3 tiles also hand the kernel the
    pre-split filter planes (ymi_conv_desc.w_x3); False = both operands are split on the fly."""
    pk = Packed(weight, bias, bn, stride, pad, cin_pad, DEV)
    xn = nhwc(x)
    if cin_pad and cin_pad != xn.shape[-1]:
        xn = torch.nn.functional.pad(xn, (0, cin_pad - xn.shape[-1]))
    xd = xn.to(DEV)
    B, H, W, Cx = xd.shape
    Ho, Wo = out_size(H, pk.kh, stride, pad), out_size(W, pk.kw, stride, pad)
    y = torch.full((B, Ho, Wo, pk.Cout), float('nan'), device=DEV)
    d = L.ConvDesc()
    d.x, d.w = xd.data_ptr(), pk.w.data_ptr()
    d.scale = pk.scale.data_ptr() if pk.scale is not None else None
    d.bias = pk.bias.data_ptr() if pk.bias is not None else None
    d.B, d.H, d.W, d.Cin, d.ldx = B, H, W, pk.Cin, Cx
    d.Ho, d.Wo, d.Cout = Ho, Wo, pk.Cout
    d.kh, d.kw, d.stride, d.pad, d.Kpad = pk.kh, pk.kw, stride, pad, pk.Kpad
    d.res_mode, d.res_after_act = res_mode, res_after_act
    rd = None
    if res is not None:
        rd = nhwc(res).to(DEV)
        d.res, d.res_ld, d.res_H, d.res_W = rd.data_ptr(), rd.shape[3], rd.shape[1], rd.shape[2]
    d.nseg, d.tile = 1, tile
    d.seg[0] = L.ConvSeg(0, pk.Cout, act, pk.Cout, Ho * Wo * pk.Cout, y.data_ptr())
    ysegs = None
    if seg_bounds:            # dense output SEGMENTS [b_i, b_i+1) of the channels, each its own tensor / activation / bound slot
        edges = [0] + list(seg_bounds) + [pk.Cout]
        ysegs = [torch.full((B, Ho, Wo, edges[i + 1] - edges[i]), float('nan'), device=DEV) for i in range(len(edges) - 1)]
        d.nseg = len(ysegs)
        for i, ys in enumerate(ysegs):
            w_ = edges[i + 1] - edges[i]
            d.seg[i] = L.ConvSeg(edges[i], edges[i + 1], (seg_acts or [act] * 3)[i], w_, Ho * Wo * w_, ys.data_ptr())
    if (tile & L.TILE_X3) and planes and dcn_offmask is None:
        d.w_x3 = pk.w3().data_ptr()
    amax = torch.zeros(4 * 1024, device=DEV)          # magnitude-bound slots (16 sub-slots, 64 floats apart): [0] bound of x
                                                      # (ymi_amax_f32), [1] what the launch reports for y ([1 .. 3]: per output segment)
    L.check(L.lib().ymi_amax_f32(xd.data_ptr(), xd.numel(), amax.data_ptr(), L.stream_ptr()), 'amax')
    d.x_amax, d.y_amax = amax.data_ptr(), amax.data_ptr() + 4096
    if tile & L.TILE_H2:
        hp, sc2, winv = pk.h2()
        d.w_h2, d.scale_h2, d.winv_h2 = hp.data_ptr(), sc2.data_ptr(), winv.data_ptr()
    ws = None
    if split_k > 1:
        ws = torch.full((split_k * B * Ho * Wo * pk.Cout,), float('nan'), device=DEV)
        d.split_k, d.split_ws = split_k, ws.data_ptr()
    s = L.stream_ptr()
    if dcn_offmask is not None:
        om = nhwc(dcn_offmask).to(DEV)
        dd = L.DcnDesc()
        dd.conv = d
        dd.offmask, dd.ldo, dd.om_layout = om.data_ptr(), om.shape[3], om_layout
        L.check(L.lib().ymi_dcn_v2_forward_f32(C.byref(dd), s), 'dcn')
    else:
        L.check(L.lib().ymi_conv2d_nhwc_f32(C.byref(d), s), 'conv')
    torch.cuda.synchronize()
    run_conv.last_amax = amax.view(4, 1024).amax(1).cpu().tolist()
    if ysegs is not None:
        return [nchw(t.cpu()) for t in ysegs]
    return nchw(y.cpu())


def rel_err(a, b):
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def build_net(meta, device=DEV):
    import yolact_amd
    from helpers import case_state_dict
    yolact_amd.set_cfg(meta['config'])
    from yolact_amd.yolact import Yolact
    net = Yolact()
    net.load_state_dict_compat(case_state_dict(meta))
    net.detect.use_fast_nms = True          # what eval.py:871 does (the class default is the reference's False)
    net.detect.use_cross_class_nms = bool(meta.get('cross_class', False))     # eval.py:872 (--cross_class_nms)
    return net.to(device)


def run_wino(x, weight, bias=None, bn=None, act=L.ACT_NONE, tile=L.TILE_AUTO, m=2, v_planes=False, up_from=None, up_relu=False,
             proj=None):
    """3x3 / stride 1 / pad 1 conv through ymi_conv3x3_winograd_f32. x: CPU NCHW. Returns CPU NCHW.
    up_from (CPU NCHW, half the size of x): the launch interpolates its input from it (ymi_wino_desc.x_up); `x` only gives the shape."""
    from yolact_amd.engine import WinoPacked
    pk = Packed(weight, bias, bn, 1, 1, None, DEV)          # folded scale / bias
    wp = WinoPacked(weight, DEV, m)
    xd = nhwc(x).to(DEV)
    B, H, W, Cc = xd.shape
    Cout = weight.shape[0]
    T = B * ((H + m - 1) // m) * ((W + m - 1) // m)
    g = (m + 2) ** 2
    V = torch.empty(g * T * Cc, device=DEV)
    Mw = torch.empty(g * T * Cout, device=DEV)
    y = torch.full((B, H, W, Cout), float('nan'), device=DEV)
    d = L.WinoDesc()
    d.x, d.u, d.y, d.V, d.M = xd.data_ptr(), wp.u.data_ptr(), y.data_ptr(), V.data_ptr(), Mw.data_ptr()
    d.scale = pk.scale.data_ptr() if pk.scale is not None else None
    d.bias = pk.bias.data_ptr() if pk.bias is not None else None
    d.B, d.H, d.W, d.C, d.Cout, d.act, d.tile, d.m = B, H, W, Cc, Cout, act, tile, m
    if tile & L.TILE_X3:
        d.u_x3 = wp.u3().data_ptr()
    amax = torch.zeros(2 * 1024, device=DEV)
    L.check(L.lib().ymi_amax_f32(xd.data_ptr(), xd.numel(), amax.data_ptr(), L.stream_ptr()), 'amax')
    d.x_amax, d.y_amax = amax.data_ptr(), amax.data_ptr() + 4096
    if up_from is not None:
        lo = nhwc(up_from).to(DEV)
        d.x, d.x_up, d.up_relu = None, lo.data_ptr(), 1 if up_relu else 0
    if tile & L.TILE_H2:
        up, uinv = wp.h2()
        d.u_h2, d.uinv_h2, d.v_planes = up.data_ptr(), uinv.data_ptr(), 1 if v_planes else 0
    py = None
    if proj is not None:       # (weight [n,256,1,1], bias or None, act): the consuming 1x1 fused into the output transform; y is not written
        pw, pb, pact = proj
        ppk = Packed(pw, pb, None, 1, 0, None, DEV)
        planes, sc2, _ = ppk.h2()
        py = torch.full((B, H, W, pw.shape[0]), float('nan'), device=DEV)
        d.proj_w_h2, d.proj_scale_h2 = planes.data_ptr(), sc2.data_ptr()
        d.proj_bias = ppk.bias.data_ptr() if ppk.bias is not None else None
        d.proj_y, d.proj_y_amax = py.data_ptr(), amax.data_ptr() + 4096
        d.proj_cout, d.proj_ldy, d.proj_act = pw.shape[0], pw.shape[0], pact
        d.y = None
    L.check(L.lib().ymi_conv3x3_winograd_f32(C.byref(d), L.stream_ptr()), 'winograd')
    torch.cuda.synchronize()
    run_wino.last_amax = amax.view(2, 1024).amax(1).cpu().tolist()
    if py is not None:
        assert torch.isnan(y).all()                   # the 3x3's own output was not touched
        return nchw(py.cpu())
    return nchw(y.cpu())


def run_chain(x, wa, ba, res, wb=None, bb=None, act_a=L.ACT_RELU, act_b=L.ACT_RELU):
    """ymi_pointwise_chain_f32 on CPU tensors: x [M,P], wa [4P,P], res [M,4P] or None, wb [P,4P] or None (P = 64: csrc/chain.hip;
    P = 128 / 256: csrc/chain2.hip).  Returns (y [M,4P], z [M,P] or None) on the CPU; run_chain.last_amax = the bounds reported for (y, z)."""
    M, P = x.shape
    pa = Packed(wa.view(4 * P, P, 1, 1), ba, None, 1, 0, None, DEV)
    pla, sca, _ = pa.h2()
    xd = x.contiguous().to(DEV)
    y = torch.full((M, 4 * P), float('nan'), device=DEV)
    amax = torch.zeros(4 * 1024, device=DEV)
    L.check(L.lib().ymi_amax_f32(xd.data_ptr(), xd.numel(), amax.data_ptr(), L.stream_ptr()), 'amax')
    d = L.ChainDesc()
    d.x, d.y, d.M, d.ldx, d.ldy = xd.data_ptr(), y.data_ptr(), M, P, 4 * P
    d.w_a_h2, d.scale_a_h2, d.cout_pad_a = pla.data_ptr(), sca.data_ptr(), pa.CoutPad
    d.bias_a = pa.bias.data_ptr() if pa.bias is not None else None
    d.k_a, d.n_a, d.n_b, d.act_a, d.act_b = P, 4 * P, P, act_a, act_b
    d.x_amax, d.y_amax, d.z_amax = amax.data_ptr(), amax.data_ptr() + 4096, amax.data_ptr() + 8192
    d.gain_a, d.bias_max_a = pa.l1_gain()
    rd = None
    if res is not None:
        rd = res.contiguous().to(DEV)
        d.res, d.res_ld = rd.data_ptr(), 4 * P
        d.res_amax = amax.data_ptr() + 12288
        L.check(L.lib().ymi_amax_f32(rd.data_ptr(), rd.numel(), d.res_amax, L.stream_ptr()), 'amax')
    z = keep = None
    if wb is not None:
        pb = Packed(wb.view(P, 4 * P, 1, 1), bb, None, 1, 0, None, DEV)
        plb, scb, _ = pb.h2()
        keep = (pb, plb, scb)
        z = torch.full((M, P), float('nan'), device=DEV)
        d.z, d.ldz, d.w_b_h2, d.scale_b_h2, d.cout_pad_b = z.data_ptr(), P, plb.data_ptr(), scb.data_ptr(), pb.CoutPad
        d.bias_b = pb.bias.data_ptr() if pb.bias is not None else None
    L.check(L.lib().ymi_pointwise_chain_f32(C.byref(d), L.stream_ptr()), 'chain')
    torch.cuda.synchronize()
    run_chain.last_amax = amax.view(4, 1024).amax(1).cpu().tolist()[1:3]
    return y.cpu(), (z.cpu() if z is not None else None)
