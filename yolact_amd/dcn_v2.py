"""The reference's `dcn_v2` module surface (external/DCNv2/dcn_v2.py) on the HIP kernels.

    dcn_v2_conv(input, offset, mask, weight, bias, stride, padding, dilation, deformable_groups)     dcn_v2.py:16-52
    DCNv2(in_channels, out_channels, kernel_size, stride, padding, dilation=1, deformable_groups=1)  dcn_v2.py:55-96
    DCN(...)  = DCNv2 + conv_offset_mask, forward(input)                                             dcn_v2.py:97-128

Inside a `Yolact` plan these modules are parameter containers (the engine reads weight / bias / conv_offset_mask once and
emits `ymi_dcn_v2_forward_f32` ops, yolact_amd/engine.py).  Called on their own — the reference's own known-answer test does
that (external/DCNv2/test.py:32-67) — they run the same C-ABI entry points: NCHW fp32 in, NCHW fp32 out, like
`_ext.dcn_v2_forward` (src/vision.cpp:3-8, src/dcn_v2.h:9-39).  Forward only (the reference's backward is training code, out of
scope); what YOLACT++ constructs is supported: 3x3, padding 1, dilation 1, one deformable group (backbone.py:22-26).
PyTorch does the layout changes (NCHW <-> NHWC) and owns the memory; all arithmetic is in the HIP library.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn
from torch.nn.modules.utils import _pair

from . import _lib as L


def _ceil(a, b):
    return (a + b - 1) // b * b


from torch.utils.weak import WeakIdKeyDictionary

_pack_cache = WeakIdKeyDictionary()     # weight TENSOR OBJECT -> {(version, bias identity / version, shape, geometry, device): Packed}


def _packed(weight, bias, stride, pad, cin_pad, device):
    """engine.Packed of (weight, bias) (packing splits the filters on the host: worth caching for modules called in a loop).  Keyed
    on the weight tensor OBJECT (identity) through a weak dictionary — an entry dies with its tensor, so a recycled address can never hit a
    stale pack and no dead module's parameters stay pinned on the GPU (round-4 advisor: the key used to be data_ptr + _version) —
    plus its version counter (in-place edits), the bias object / version, the shapes, the geometry and the device."""
    from .engine import Packed
    sub = _pack_cache.get(weight)
    if sub is None:
        sub = _pack_cache[weight] = {}
    key = (weight._version, tuple(weight.shape), None if bias is None else (id(bias), bias._version, tuple(bias.shape)),
           stride, pad, cin_pad, str(device))
    pk = sub.get(key)
    if pk is None:
        sub.clear()                            # one live pack per weight tensor: an older version's pack is garbage
        pk = sub[key] = Packed(weight, bias, None, stride, pad, cin_pad, device)
    return pk


def _nhwc_padded(x, cpad):
    xn = x.detach().to(torch.float32).permute(0, 2, 3, 1)
    if cpad != xn.shape[-1]:
        xn = torch.nn.functional.pad(xn, (0, cpad - xn.shape[-1]))
    return xn.contiguous()


def _fill_desc(d, xd, pk, y, Ho, Wo, amax, h2):
    B, H, W, Cx = xd.shape
    d.x, d.w = xd.data_ptr(), pk.w.data_ptr()
    d.scale = pk.scale.data_ptr() if pk.scale is not None else None
    d.bias = pk.bias.data_ptr() if pk.bias is not None else None
    d.B, d.H, d.W, d.Cin, d.ldx = B, H, W, pk.Cin, Cx
    d.Ho, d.Wo, d.Cout = Ho, Wo, pk.Cout
    d.kh, d.kw, d.stride, d.pad, d.Kpad = pk.kh, pk.kw, pk.stride, pk.pad, pk.Kpad
    d.cin_alg, d.cout_alg = pk.cin_alg, pk.cout_alg
    d.nseg = 1
    d.seg[0] = L.ConvSeg(0, pk.Cout, L.ACT_NONE, pk.Cout, Ho * Wo * pk.Cout, y.data_ptr())
    if h2:
        planes, sc2, winv = pk.h2()
        d.w_h2, d.scale_h2, d.winv_h2 = planes.data_ptr(), sc2.data_ptr(), winv.data_ptr()
        d.x_amax = amax.data_ptr()


def _input_bound(xd):
    """A magnitude-bound slot (ymi_conv_desc.x_amax) raised to max|x| by ymi_amax_f32."""
    from .engine import AMAX_SLOT_FLOATS
    amax = torch.zeros(AMAX_SLOT_FLOATS, dtype=torch.float32, device=xd.device)
    L.check(L.lib().ymi_amax_f32(xd.data_ptr(), xd.numel(), amax.data_ptr(), L.stream_ptr()), 'ymi_amax_f32')
    return amax


def _check_geometry(weight, stride, padding, dilation, deformable_groups):
    st, pd, dl = _pair(stride), _pair(padding), _pair(dilation)
    if tuple(weight.shape[2:4]) != (3, 3) or pd != (1, 1) or dl != (1, 1) or deformable_groups != 1 or st[0] != st[1]:
        raise NotImplementedError('yolact_amd dcn_v2: only the 3x3 / padding 1 / dilation 1 / one-group DCN that YOLACT++ '
                                  'constructs (backbone.py:22-26); got kernel %s stride %s padding %s dilation %s groups %d'
                                  % (tuple(weight.shape[2:4]), st, pd, dl, deformable_groups))
    return st[0]


def _dcn_launch(xd, om, mask_is_prob, pk, Ho, Wo):
    """xd [B,H,W,Cin_pad] NHWC, om [B,Ho,Wo,ldo] NHWC (18 offsets | 9 mask channels) -> y [B,Ho,Wo,Cout] NHWC."""
    y = torch.empty(xd.shape[0], Ho, Wo, pk.Cout, dtype=torch.float32, device=xd.device)
    # the pipelined fp16x2 gather-GEMM (csrc/dcn.hip) where its epilogue applies (Cout % 4 == 0), else the general loader on the
    # exact-fp32 MFMA tiles; both are fp32-class (include/yolact_amd.h, YMI_TILE_H2)
    h2 = pk.Cout % 4 == 0
    amax = _input_bound(xd) if h2 else None
    dd = L.DcnDesc()
    _fill_desc(dd.conv, xd, pk, y, Ho, Wo, amax, h2)
    M = xd.shape[0] * Ho * Wo
    dd.conv.tile = (L.TILE_H2 | L.TILE_DCNP | (L.DCNP_64x128_W8 if M >= 16384 else L.DCNP_32x128)) if h2 else L.TILE_AUTO
    dd.offmask, dd.ldo, dd.mask_is_prob = om.data_ptr(), om.shape[3], 1 if mask_is_prob else 0
    L.check(L.lib().ymi_dcn_v2_forward_f32(C.byref(dd), L.stream_ptr()), 'ymi_dcn_v2_forward_f32')
    return y


def dcn_v2_conv(input, offset, mask, weight, bias, stride, padding, dilation, deformable_groups):
    """`_DCNv2.apply` (dcn_v2.py:16-52), forward only: input [B,Cin,H,W], offset [B,18,Ho,Wo] (channel 2k = dh_k, 2k+1 = dw_k),
    mask [B,9,Ho,Wo] = the modulation itself (callers pass torch.sigmoid(...), dcn_v2.py:122), weight [Cout,Cin,3,3], bias
    [Cout] -> [B,Cout,Ho,Wo]."""
    from .engine import out_size
    L.require_cuda(input, 'dcn_v2_conv input')
    st = _check_geometry(weight, stride, padding, dilation, deformable_groups)
    B, Cin, H, W = input.shape
    Ho, Wo = out_size(H, 3, st, 1), out_size(W, 3, st, 1)
    if tuple(offset.shape) != (B, 18, Ho, Wo) or tuple(mask.shape) != (B, 9, Ho, Wo) or weight.shape[1] != Cin:
        raise ValueError('dcn_v2_conv: offset %s / mask %s / weight %s do not fit input %s at stride %d'
                         % (tuple(offset.shape), tuple(mask.shape), tuple(weight.shape), tuple(input.shape), st))
    dev = input.device
    with torch.cuda.device(dev), torch.no_grad():
        cin_p = _ceil(Cin, 32)                    # zero channels x zero filters: the kernels take Cin % 32 == 0
        xd = _nhwc_padded(input, cin_p)
        om = torch.cat([offset, mask], 1).detach().to(torch.float32).permute(0, 2, 3, 1).contiguous()
        pk = _packed(weight, bias, st, 1, cin_p if cin_p != Cin else None, dev)
        y = _dcn_launch(xd, om, True, pk, Ho, Wo)
        return y.permute(0, 3, 1, 2).contiguous()


class DCNv2(nn.Module):
    """dcn_v2.py:55-96."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation=1, deformable_groups=1):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride = _pair(kernel_size), _pair(stride)
        self.padding, self.dilation = _pair(padding), _pair(dilation)
        self.deformable_groups = deformable_groups
        if (self.kernel_size != (3, 3) or self.padding != (1, 1) or self.dilation != (1, 1) or deformable_groups != 1
                or self.stride[0] != self.stride[1]):            # (DCN.forward launches with stride[0]: an asymmetric pair must not pass)
            raise NotImplementedError('only the 3x3 / pad 1 / one-group / square-stride DCN that YOLACT++ constructs (backbone.py:22-26)')
        self.weight = nn.Parameter(torch.zeros(out_channels, in_channels, *self.kernel_size))
        self.bias = nn.Parameter(torch.zeros(out_channels))
        self.reset_parameters()

    def reset_parameters(self):
        n = self.in_channels * self.kernel_size[0] * self.kernel_size[1]
        with torch.no_grad():
            self.weight.uniform_(-1.0 / n ** 0.5, 1.0 / n ** 0.5)
            self.bias.zero_()

    def forward(self, input, offset, mask):
        k = self.deformable_groups * self.kernel_size[0] * self.kernel_size[1]
        assert 2 * k == offset.shape[1] and k == mask.shape[1]
        return dcn_v2_conv(input, offset, mask, self.weight, self.bias, self.stride, self.padding, self.dilation,
                           self.deformable_groups)


class DCN(DCNv2):
    """dcn_v2.py:97-128: DCNv2 + the 27-channel offset / mask convolution; forward(input)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=1, dilation=1, deformable_groups=1):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, deformable_groups)
        self.conv_offset_mask = nn.Conv2d(in_channels, 27, 3, stride=self.stride, padding=1, bias=True)
        self.init_offset()

    def init_offset(self):
        with torch.no_grad():
            self.conv_offset_mask.weight.zero_()
            self.conv_offset_mask.bias.zero_()

    def forward(self, input):
        """out = conv_offset_mask(input); offset = out[:, :18]; mask = sigmoid(out[:, 18:]); dcn_v2_conv(...) — as two launches
        of the HIP library: the 27-channel 3x3 convolution, then the gather-GEMM reading its NHWC output directly (the sigmoid is
        applied in the kernel, as in the engine's plans)."""
        from .engine import out_size
        L.require_cuda(input, 'DCN input')
        st = self.stride[0]
        B, Cin, H, W = input.shape
        Ho, Wo = out_size(H, 3, st, 1), out_size(W, 3, st, 1)
        dev = input.device
        with torch.cuda.device(dev), torch.no_grad():
            cin_p = _ceil(Cin, 32)
            xd = _nhwc_padded(input, cin_p)
            cp = cin_p if cin_p != Cin else None
            pko = _packed(self.conv_offset_mask.weight, self.conv_offset_mask.bias, st, 1, cp, dev)
            om = torch.empty(B, Ho, Wo, 27, dtype=torch.float32, device=dev)
            d = L.ConvDesc()
            _fill_desc(d, xd, pko, om, Ho, Wo, None, False)
            L.check(L.lib().ymi_conv2d_nhwc_f32(C.byref(d), L.stream_ptr()), 'conv_offset_mask')
            pk = _packed(self.weight, self.bias, st, 1, cp, dev)
            y = _dcn_launch(xd, om, False, pk, Ho, Wo)
            return y.permute(0, 3, 1, 2).contiguous()
