"""Execution plan for the HIP forward pass: weight packing, activation arena, op list.

A plan is built once per (batch, height, width): every convolution becomes one `ymi_conv_desc` with all
pointers/strides resolved, activations live in a small arena of reusable device buffers (NHWC), and the
forward pass is a flat loop of C-ABI calls on the current HIP stream — no Python tensor math, no host
synchronisation.  PyTorch is used for device memory only.
"""
from __future__ import annotations

import ctypes as C
import json
import os
from typing import List, Optional

import torch
import torch.nn as nn

from . import _lib as L
from .config import act_name, make_priors_host
from . import modules as M


def _ceil(a, b):
    return (a + b - 1) // b * b


# ---- shipped tile / algorithm table ------------------------------------------------------------------------------
SPLIT_DEFAULT = '2'   # default of YOLACT_AMD_SPLIT (see Plan.__init__): 0 exact-fp32 MFMA, 1 bf16x3, 2 fp16x2
TUNE_GEN = 5          # bump whenever tile ids or kernel variants change meaning: older tables are ignored (5: pipelined kernel of csrc/dcn.hip)
AMAX_SLOT_FLOATS = 16 * 64   # one magnitude-bound slot: YMI_AMAX_SUB sub-slots, YMI_AMAX_STRIDE floats apart (include/yolact_amd.h)
TUNE_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tune')
_table_cache = {}


def _read_table_file(path):
    try:
        with open(path) as f:
            doc = json.load(f)
    except (OSError, ValueError):
        return {}
    if not isinstance(doc, dict) or doc.get('gen') != TUNE_GEN or not isinstance(doc.get('abi'), int) or doc['abi'] > L.ABI_VERSION:
        return {}       # (gen = meaning of the tile ids; a table written under an OLDER ABI of the same generation stays valid)
    return dict(doc.get('entries', {}))


def _write_table_file(path, entries, device=None, note=None):
    doc = {'gen': TUNE_GEN, 'abi': L.ABI_VERSION,
           'device': torch.cuda.get_device_name(device) if device is not None and device.type == 'cuda' else None,
           'note': note or 'tile id per conv shape (B,H,W,Cin,Cout,kh,kw,stride,pad,res_mode,nseg,Kpad); wino(...) -> '
                           '[m, gemm tile, direct ms, winograd ms, F(2x2) ms, F(4x4) ms]',
           'entries': {k: entries[k] for k in sorted(entries)}}
    tmp = path + '.tmp%d' % os.getpid()
    with open(tmp, 'w') as f:
        json.dump(doc, f, indent=0, separators=(',', ':'))
    os.replace(tmp, path)


def load_tune_table(device):
    """Entries of the shipped table for this device's architecture ({} when there is none)."""
    arch = 'gfx950'
    if device.type == 'cuda':
        arch = getattr(torch.cuda.get_device_properties(device), 'gcnArchName', 'gfx950').split(':')[0]
    if arch not in _table_cache:
        _table_cache[arch] = _read_table_file(os.path.join(TUNE_DIR, arch + '.json'))
    return dict(_table_cache[arch])


def build_meta():
    """yolact_amd/build_meta.json, written by __graft_entry__.build(): {'lint': 'ok' | 'skipped'} (missing file: {})."""
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'build_meta.json')) as f:
            doc = json.load(f)
        return doc if isinstance(doc, dict) else {}
    except (OSError, ValueError):
        return {}


def patch_tile_allowed():
    """csrc/patch.hip issues gfx950's new MFMAs through the plain builtins (no constrained wrapper): what keeps its results right is the
    post-build ISA lint (tools/check_mfma_overlap.py).  A build whose lint could NOT run (build_meta 'lint': 'skipped', only possible
    with YOLACT_AMD_ALLOW_NO_LINT=1) must not select that tile (round-5 advisor); YOLACT_AMD_PATCH=0 is the A/B switch."""
    return os.environ.get('YOLACT_AMD_PATCH', '1') == '1' and build_meta().get('lint', 'ok') != 'skipped'


def split3_planes(w: torch.Tensor) -> torch.Tensor:
    """fp32 [..., R, K] -> bf16 bit patterns int16 [..., 3, R, K]: plane 0 = the top 8 significant bits of every value (its
    upper 16 bits, i.e. truncation to bf16), plane 1 = the top 8 bits of the exact remainder, plane 2 = what is left (at most
    8 bits, exactly a bf16).  plane0 + plane1 + plane2 == w bit for bit (csrc/conv_igemm.hip split8 does the same to the
    activations on the fly); used as `ymi_conv_desc.w_x3` by the bf16x3 tiles."""
    w = w.detach().to(torch.float32).contiguous()
    mask = -65536                                    # 0xFFFF0000 as int32

    def hi(t):                                       # (truncated value as fp32, its upper 16 bits as int16)
        u = t.view(torch.int32) & mask
        return u.view(torch.float32), (u >> 16).to(torch.int16)
    h, hb = hi(w)
    r = w - h
    m, mb = hi(r)
    r2 = r - m
    l, lb = hi(r2)
    # exact for every normal fp32 value; a subnormal input (|w| < 1.2e-38) keeps its top 7 mantissa bits only (error < 1e-40)
    assert bool(((h + m + l - w).abs() <= 1e-38).all()) or not torch.isfinite(w).all()
    return torch.stack([hb, mb, lb], dim=-3).contiguous()


def split2_planes_f16(w: torch.Tensor):
    """fp32 [..., R, K] -> (fp16 bit patterns int16 [..., 2, R, K], inverse row scales fp32 [..., R]) for the fp16x2 tiles
    (csrc/conv_igemm.hip split8h does the same to the activations on the fly): row r is scaled by the power of two s_r that maps
    max|w[r, :]| into [2^13, 2^14) (all-zero rows: 1); plane 0 = fp16(w s) by round to nearest, plane 1 = fp16(w s - plane 0)
    (the difference is exact in fp32).  11 + 11 significant bits + two signs: about two thirds of all fp32 values are represented
    exactly, the rest to one fp32 ulp, unbiased.  The second result is 1 / s_r (exact), folded into the epilogue's per-channel
    scale by the callers."""
    w = w.detach().to(torch.float32).contiguous()
    amax = w.abs().amax(dim=-1)
    _, e = torch.frexp(amax)                                   # amax = m * 2^e with 0.5 <= m < 1, i.e. amax < 2^e
    s = torch.ldexp(torch.ones_like(amax), 14 - e)
    s = torch.where(amax > 0, s, torch.ones_like(s))
    t = w * s.unsqueeze(-1)                                    # exact: a power-of-two scale
    h = t.to(torch.float16)
    l = (t - h.to(torch.float32)).to(torch.float16)
    assert bool(torch.isfinite(h.float()).all()), 'fp16x2 filter planes overflowed'
    planes = torch.stack([h, l], dim=-3).contiguous().view(torch.int16)
    return planes, (1.0 / s).contiguous()


class Packed:
    """A convolution's filters in the engine layout [CoutPad][Kpad] (k = (ky*kw+kx)*Cin + c) + folded epilogue."""

    def __init__(self, weight, bias=None, bn: Optional[nn.BatchNorm2d] = None, stride=1, pad=0, cin_pad=None,
                 device=None):
        weight = weight.detach().to(torch.float32)
        Cout, Cin, kh, kw = weight.shape
        cin_p = cin_pad or Cin
        w = weight.permute(0, 2, 3, 1)
        if cin_p != Cin:
            w = torch.nn.functional.pad(w, (0, cin_p - Cin))
        K = kh * kw * cin_p
        self.Kpad, self.CoutPad = _ceil(K, 32), _ceil(Cout, 128)
        wp = torch.zeros(self.CoutPad, self.Kpad, dtype=torch.float32, device=weight.device)
        wp[:Cout, :K] = w.reshape(Cout, K)
        self.w = wp.to(device).contiguous()
        self._wp_host, self._w3, self._h2 = wp, None, None
        self.weight_oihw = weight if (kh == 3 and kw == 3) else None     # Winograd transform source
        self.Cin, self.Cout, self.kh, self.kw, self.stride, self.pad = cin_p, Cout, kh, kw, stride, pad
        self.cin_alg = Cin
        self.cout_alg = Cout           # real output channels when Cout carries zero-filter padding (FLOP accounting)
        scale = shift = None
        if bn is not None:
            # torch's eval BatchNorm: alpha = gamma * rsqrt(var + eps); y = x*alpha + (beta - mean*alpha), fp32
            invstd = 1.0 / torch.sqrt(bn.running_var.detach().float() + bn.eps)
            scale = bn.weight.detach().float() * invstd
            shift = bn.bias.detach().float() - bn.running_mean.detach().float() * scale
            if bias is not None:
                shift = shift + bias.detach().float() * scale
        elif bias is not None:
            shift = bias.detach().float()
        self.scale = scale.to(device).contiguous() if scale is not None else None
        self.bias = shift.to(device).contiguous() if shift is not None else None

    def tiny_columns(self, binades=10, in_gain=None):
        """True when this layer looks like the CONSUMER OF COMPENSATED OUTLIER CHANNELS: some input channel is weighted >= 2^binades
        less than the typical one (column maximum over filters and taps against the median column maximum; all-zero padding columns
        ignored) AND — `in_gain`, the per-channel gain of the launches that PRODUCE the input tensor (Packed.out_gain, carried on
        engine.T) — that very channel is amplified by its producer >= 2^(binades - 4) over the typical channel.  Both halves are the
        signature of a BN-folded checkpoint with outlier channels (a huge gamma / sigma on the producer, the inverse folded into the
        consumer's filters): the typical channels then sit that many binades below the tensor's magnitude bound, and the fp16x2
        tiles — ONE power-of-two scale per activation tensor — keep the low piece of a value normal only 15 binades below the bound
        (tests/test_gpu_batch_parity.py test_outlier_channels_end_to_end_at_batch8: 2^12 outliers 9e-6, 2^14 3e-5, 2^16
        1.3e-4 of the head tensors).  Such layers run on the bf16x3 tiles instead (bf16 has the fp32 exponent range: no scale, no
        dependence on the bound).
        Tiny columns ALONE are not enough (round-4 advisor): weight decay drives the filters of DEAD input channels to zero in real
        checkpoints — same column signature, ordinary activations, no precision problem — and demoting those layers would silently
        cost them the fp16x2 tiles, Winograd and every fusion.  Without producer information (in_gain None) the weights-only test is
        kept, conservatively."""
        memo = getattr(self, '_tiny_memo', None)
        if memo is None:
            memo = self._tiny_memo = {}
        mkey = (binades, None if in_gain is None else (id(in_gain), int(in_gain.numel())))
        if mkey in memo and (in_gain is None or memo[mkey][1] is in_gain):      # (the gain tensor is kept: its id cannot be reused)
            return memo[mkey][0]
        r = self._tiny_columns(binades, in_gain)
        memo[mkey] = (r, in_gain)
        return r

    def _tiny_columns(self, binades, in_gain):
        w = self._wp_host[:self.Cout, :self.kh * self.kw * self.Cin].abs().float().cpu()
        col = w.view(self.Cout, self.kh * self.kw, self.Cin).amax(dim=(0, 1))
        nzm = col > 0
        if int(nzm.sum()) <= 8:
            return False
        tiny = nzm & (col * float(2 ** binades) < col[nzm].median())
        if not bool(tiny.any()):
            return False
        if in_gain is None:
            return True
        g = in_gain.detach().float().cpu().flatten()
        if g.numel() < self.Cin:                                     # (channel padding of the consumer: padded channels carry no data)
            g = torch.nn.functional.pad(g, (0, self.Cin - g.numel()))
        g = g[:self.Cin]
        pos = g[g > 0]
        if pos.numel() == 0:
            return True
        return bool((tiny & (g >= pos.median() * float(2 ** max(binades - 4, 1)))).any())

    def out_gain(self):
        """[Cout] per-output-channel gain of this layer: max|w[n, :]| x |folded BN scale[n]| (1 without BN) + |folded shift[n]| — how
        large the launch can make each channel it writes relative to the others.  Consumers compare it across channels
        (tiny_columns): an outlier channel of a BN-folded checkpoint shows up as a gain far above the median, whether it comes from a
        huge gamma / sigma or from a huge beta / mean shift (round-5 advisor: the shift used to be ignored; adding it is the
        conservative fold — ordinary checkpoints have |shift| = O(1), which moves no channel 2^6 above the median)."""
        if getattr(self, '_out_gain', None) is None:
            g = self._wp_host[:self.Cout].abs().amax(dim=1).float().cpu()
            if self.scale is not None:
                g = g * self.scale.detach().abs().float().cpu()
            if self.bias is not None:
                g = g + self.bias.detach().abs().float().cpu()[:self.Cout]
            self._out_gain = g
        return self._out_gain

    def _invalidate(self):
        self._w3 = self._h2 = None
        self._out_gain = self._l1_gain = None
        self._tiny_memo = {}

    def scale_out_channels(self, f: torch.Tensor):
        """Multiply output channel n of this layer by f[n] (host tensor [Cout], exact powers of two: Plan._rebalance_outliers): the
        folded scale and shift carry it, the filters are untouched."""
        f = f.detach().float().cpu()
        dev = self.w.device
        sc = self.scale.detach().float().cpu() if self.scale is not None else torch.ones(self.Cout)
        self.scale = (sc * f).to(dev).contiguous()
        if self.bias is not None:
            self.bias = (self.bias.detach().float().cpu() * f).to(dev).contiguous()
        self._invalidate()

    def scale_in_channels(self, f: torch.Tensor):
        """Multiply the filters of INPUT channel c (every tap, every filter) by f[c] (host tensor [Cin], exact powers of two)."""
        f = f.detach().float().cpu()
        taps = self.kh * self.kw
        assert self.Kpad == taps * self.Cin and f.numel() == self.Cin
        wp = self._wp_host.detach().float().cpu().clone()
        wp.view(self.CoutPad, taps, self.Cin).mul_(f.view(1, 1, -1))
        self._wp_host = wp
        self.w = wp.to(self.w.device).contiguous()
        if self.weight_oihw is not None:
            self.weight_oihw = self.weight_oihw.detach().float().cpu() * f.view(1, -1, 1, 1)
        self._invalidate()

    def in_column_max(self):
        """[Cin] max |w| over filters and taps per input channel (host)."""
        w = self._wp_host[:self.Cout, :self.kh * self.kw * self.Cin].detach().abs().float().cpu()
        return w.view(self.Cout, self.kh * self.kw, self.Cin).amax(dim=(0, 1))

    def l1_gain(self):
        """(max_n(|folded BN scale_n| * sum_k |w[n, k]|), max_n |folded shift_n|): |y_n| <= amax(x) * gain + shift_max for every
        output of this layer before the residual / activation — the rigorous bound csrc/chain2.hip scales its y slices by
        (ymi_chain_desc.gain_a / bias_max_a)."""
        if getattr(self, '_l1_gain', None) is None:
            g = self._wp_host[:self.Cout].detach().cpu().abs().double().sum(dim=1)
            if self.scale is not None:
                g = g * self.scale.detach().abs().double().cpu()
            b = float(self.bias.detach().abs().max().cpu()) if self.bias is not None else 0.0
            self._l1_gain = (float(g.max()) * (1.0 + 2.0 ** -20), b)       # (a hair above the fp64 sum: fp32 evaluation on the device)
        return self._l1_gain

    def w3(self):
        """[3][CoutPad][Kpad] bf16 planes of the same filters for the bf16x3 tiles (built on first use)."""
        if self._w3 is None:
            self._w3 = split3_planes(self._wp_host.cpu()).to(self.w.device)
        return self._w3


    def h2(self):
        """(planes [2][CoutPad][Kpad] fp16, scale_h2 [CoutPad] = folded-BN scale / the row's filter scale, winv_h2 [CoutPad] =
        1 / the row's filter scale) for the fp16x2 tiles (ymi_conv_desc.w_h2 / scale_h2 / winv_h2); built on first use."""
        if self._h2 is None:
            planes, winv = split2_planes_f16(self._wp_host.cpu())
            sc = torch.ones(self.CoutPad, dtype=torch.float32, device='cpu')    # (explicit: eval.py:1080 makes CUDA the default tensor type)
            if self.scale is not None:
                sc[:self.Cout] = self.scale.detach().float().cpu()
            dev = self.w.device
            self._h2 = (planes.to(dev), (sc * winv).contiguous().to(dev), winv.to(dev))
        return self._h2


class WinoPacked:
    """Winograd filters U = G g G^T in fp64 -> fp32, layout [(m+2)^2][CoutPad][C] (csrc/winograd.hip); m = 2: F(2x2,3x3),
    m = 4: F(4x4,3x3) (matrices of Lavin & Gray, interpolation points 0, +-1, +-2, inf)."""

    G2 = torch.tensor([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]], dtype=torch.float64)
    G4 = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6],
                       [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=torch.float64)

    def __init__(self, weight, device, m=2):
        w = weight.detach().to(torch.float64).cpu()                 # [Cout, Cin, 3, 3]
        Cout, Cin = w.shape[:2]
        G = self.G4 if m == 4 else self.G2
        a = m + 2
        u = torch.einsum('ai,ncij,bj->abnc', G, w, G)                # [a,a,Cout,Cin]
        self.m = m
        self.CoutPad = _ceil(Cout, 128)
        up = torch.zeros(a * a, self.CoutPad, Cin, dtype=torch.float32)
        up[:, :Cout] = u.reshape(a * a, Cout, Cin).to(torch.float32)
        self.u = up.to(device).contiguous()
        self._up_host, self._u3, self._uh2 = up, None, None

    def u3(self):
        """[G][3][CoutPad][C] bf16 planes of U for the bf16x3 GEMM tiles."""
        if self._u3 is None:
            self._u3 = split3_planes(self._up_host).to(self.u.device)
        return self._u3


def _wino_h2(self):
    """([G][2][CoutPad][C] fp16 planes of U, [G][CoutPad] inverse row scales) for the fp16x2 GEMM tiles."""
    if self._uh2 is None:
        planes, uinv = split2_planes_f16(self._up_host)
        self._uh2 = (planes.to(self.u.device), uinv.to(self.u.device))
    return self._uh2


WinoPacked.h2 = _wino_h2


def wino_eligible(pk, res, segs, act, x_C):
    if not (pk.kh == 3 and pk.kw == 3 and pk.stride == 1 and pk.pad == 1 and pk.Cin % 32 == 0
            and pk.cin_alg == pk.Cin and x_C == pk.Cin and res is None):
        return False
    if segs is not None:                       # segmented scatter (prediction heads): any Cout / activation
        return len(segs) <= 3
    return pk.Cout % 4 == 0 and act in (L.ACT_NONE, L.ACT_RELU, L.ACT_LEAKY01)


def pack_offmask(conv: nn.Conv2d, device, interleave: bool) -> Packed:
    """conv_offset_mask of a DCN layer (dcn_v2.py:107-112: 27 filters = 18 offsets | 9 mask logits) with its filter ROWS zero-padded
    to 32 — a full 128-byte line per pixel, and Cout % 4 == 0 admits the pipelined kernel and its 32-column tiles (YMI_DCNP_*x32) —
    and, `interleave`, permuted to [dh_k, dw_k, mask_k] per tap (ymi_dcn_desc.om_layout = 1: one 12-byte load per tap in the gather
    kernel).  The arithmetic of the 27 real channels is untouched: same filters, same order of accumulation."""
    w, b = conv.weight.detach().float(), conv.bias.detach().float()
    assert w.shape[0] == 27 and conv.stride[0] == conv.stride[1] and conv.padding == (1, 1), 'conv_offset_mask of a 3x3 one-group DCN'
    order = [c for k in range(9) for c in (2 * k, 2 * k + 1, 18 + k)] if interleave else list(range(27))
    wp, bp = w.new_zeros((32,) + tuple(w.shape[1:])), b.new_zeros(32)
    wp[:27], bp[:27] = w[order], b[order]
    pk = Packed(wp, bp, None, conv.stride[0], 1, None, device)
    pk.cout_alg = 27
    return pk


def pack_module(conv: nn.Conv2d, bn=None, device=None, cin_pad=None) -> Packed:
    assert conv.dilation == (1, 1) and conv.groups == 1
    assert conv.stride[0] == conv.stride[1] and conv.padding[0] == conv.padding[1]
    return Packed(conv.weight, conv.bias, bn, conv.stride[0], conv.padding[0], cin_pad, device)


class T:
    """An NHWC activation living in an arena buffer."""
    __slots__ = ('buf', 'B', 'H', 'W', 'C', 'slot', 'gain')

    def __init__(self, buf, B, H, W, C, slot=None, gain=None):
        self.buf, self.B, self.H, self.W, self.C = buf, B, H, W, C
        self.slot = slot          # index of the tensor's magnitude bound in Plan.amax (fp16x2 tiles), None: not tracked
        self.gain = gain          # [C] per-channel gain of the launches that write this tensor (Packed.out_gain; host tensor), None: unknown

    @property
    def ptr(self):
        return self.buf.data_ptr()

    def view(self):
        return self.buf[: self.B * self.H * self.W * self.C].view(self.B, self.H, self.W, self.C)


class Arena:
    """Best-fit reuse of device buffers; ops run in program order on one stream so reuse is race-free.
    Keeps the working set small enough to sit in the 256 MB Infinity Cache between layers."""

    def __init__(self, device):
        self.device = device
        self.free_list: List[torch.Tensor] = []
        self.all: List[torch.Tensor] = []

    def alloc(self, numel) -> torch.Tensor:
        best = None
        for i, b in enumerate(self.free_list):
            if b.numel() >= numel and (best is None or b.numel() < self.free_list[best].numel()):
                best = i
        if best is not None:
            return self.free_list.pop(best)
        b = torch.empty(numel, dtype=torch.float32, device=self.device)
        self.all.append(b)
        return b

    def free(self, t):
        if isinstance(t, _Borrowed):
            return  # owned by someone else (a stage output the FPN still needs)
        buf = t.buf if isinstance(t, T) else t
        assert all(buf is not f for f in self.free_list)
        self.free_list.append(buf)

    def total_bytes(self):
        return sum(b.numel() * 4 for b in self.all)


def out_size(n, k, s, p):
    return (n + 2 * p - k) // s + 1


_SIDE_STREAMS = {}


def _side_stream(device):
    """The side stream of a new plan: one of a FIXED pool of YOLACT_AMD_SIDE_STREAMS (default 2) per device, handed out round robin (two: a
    caller that overlaps batches on a second main stream — bench.py --step-overlap 2 — then uses four streams in all; the FIFTH stream of a
    process was measured on the main stream's hardware queue again, session r6w).
    HIP folds the streams of a process onto GPU_MAX_HW_QUEUES (4) hardware queues in creation order; with one new torch stream per plan
    (rounds 1 - 5) the 4th plan of a process got a side stream on the MAIN stream's queue and its two-stream schedule ran 1.5x slower
    than a single stream (bench.py secondary.outlier_plan, session r6l: 1 180 - 1 260 images/s against 2 050 for the same plan on a
    stream of the pool).  YOLACT_AMD_SIDE_STREAMS=0: one new stream per plan (A/B switch)."""
    n = int(os.environ.get('YOLACT_AMD_SIDE_STREAMS', '2'))
    if n <= 0:
        return torch.cuda.Stream(device=device)
    pool = _side_pool(device, n)
    st = pool['streams'][pool['next'] % n]
    pool['next'] += 1
    return st


def _side_pool(device, n):
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    pool = _SIDE_STREAMS.get(key)
    if pool is None:
        # ALL streams of the pool are created at the first request, back to back: a stream created later — after RCCL or the caller
        # have created theirs — can land on the main stream's hardware queue again (session r6w: the third pool stream, created lazily
        # behind bench.py's second step stream and RCCL's, made the batch-1 plan 2.3x slower)
        pool = _SIDE_STREAMS[key] = {'streams': [torch.cuda.Stream(device=device) for _ in range(n)], 'next': 0}
    return pool


def side_stream_pool(device):
    """The pooled side streams of a device (created on first use; () with YOLACT_AMD_SIDE_STREAMS=0).  yolact_amd.pipeline.BatchPipeline
    runs its un-forked plan slots on them: a plan that does not fork leaves its side stream idle, and a process should not keep more than
    four streams busy (_side_stream)."""
    n = int(os.environ.get('YOLACT_AMD_SIDE_STREAMS', '2'))
    return tuple(_side_pool(device, n)['streams']) if n > 0 else ()


class _OpList(list):
    """The plan's op list: a list that counts its mutations.  The native executor's ymi_plan_op array bakes descriptor addresses and
    op kinds; it is rebuilt whenever `version` moved (round-5 advisor: the earlier key, the ids of the op tuples, could repeat once the
    tuner / the fusions had replaced a tuple and CPython reused the freed one's id)."""
    version = 0

    def _bump(self):
        self.version += 1

    def __setitem__(self, k, v):
        self._bump(); list.__setitem__(self, k, v)

    def __delitem__(self, k):
        self._bump(); list.__delitem__(self, k)

    def append(self, v):
        self._bump(); list.append(self, v)

    def insert(self, k, v):
        self._bump(); list.insert(self, k, v)

    def pop(self, *a):
        self._bump(); return list.pop(self, *a)

    def extend(self, it):
        self._bump(); list.extend(self, it)


class Plan:
    def __init__(self, net, B, H, W, device, dry_two_streams=False):
        """`dry_two_streams`: emit the two-stream op list (fork / join markers, per-stream arenas) without creating HIP
        streams or events — for static checks of the schedule on a machine without a GPU (tests/test_plan_schedule.py);
        such a plan cannot be run."""
        self.net, self.B, self.H, self.W, self.device = net, B, H, W, device
        self.ops = _OpList()   # (callable, args...) executed in order; counts its mutations (native executor cache)
        self.conv_meta = []    # (name, desc) for profiling / roofline accounting
        self.arena = Arena(device)
        self.keepalive = []
        self.lib = L.lib()
        # Two HIP streams: A = the caller's current stream (backbone, P3 branch, protonet), B = a side stream for the
        # small P4..P7 FPN/head convs and Detect, which are latency-bound and leave most CUs idle when run alone
        # (profiles/r01_layers_v4.txt: 0.7 ms of < 60 TF/s layers + 0.35 ms of Detect per batch-8 step).
        self.two_streams = (device.type == 'cuda' and os.environ.get('YOLACT_AMD_STREAMS', '2') != '1') or dry_two_streams
        # ONE side stream per device for every plan of the process (round 6): each torch.cuda.Stream() is another HIP stream, HIP folds
        # streams onto GPU_MAX_HW_QUEUES (4) hardware queues, and the 4th / 5th plan of a process (other batch sizes, a second model)
        # got a side stream that shared a hardware queue with the main stream — its two-stream schedule then ran 1.5x SLOWER than
        # one stream (bench.py's outlier_plan line: 5.7 ms against 4.1 single-stream and 3.8 for the same plan built first)
        self.stream_b = _side_stream(device) if self.two_streams and device.type == 'cuda' else None
        self.overlap = True      # runtime switch: False runs the same op list on ONE stream (serialised kernels), which
                                 # is what per-kernel timing (bench.py's roofline pass) needs
        self.events = {}
        self._cur = 'A'
        # YOLACT_AMD_WINOGRAD: 0 = direct kernels only, 2 = F(2x2,3x3) only, 4 = F(4x4,3x3) only, default both (autotuned)
        wsel = os.environ.get('YOLACT_AMD_WINOGRAD', '1')
        self.use_winograd = device.type == 'cuda' and wsel != '0'
        self.wino_variants = (2,) if wsel == '2' else (4,) if wsel == '4' else (2, 4)
        # YOLACT_AMD_WINOGRAD_FORCE=1: take the Winograd path on every eligible layer even where the direct kernel
        # measured faster (parity tests of the least accurate variant; never the default)
        self.wino_force = os.environ.get('YOLACT_AMD_WINOGRAD_FORCE', '0') == '1'
        # YOLACT_AMD_SPLIT=1: the bf16x3 variants of the GEMM tiles (fp32-class products on the bf16 matrix pipe, 6 bf16
        # MFMAs per product at 16x the fp32-MFMA rate, csrc/conv_igemm.hip split8) join the candidates of every Cin % 32
        # == 0 layer and of the Winograd GEMMs; the measurement decides per shape
        # YOLACT_AMD_SPLIT=2 (default): the fp16x2 variants instead — two fp16 pieces by round to nearest, 3 fp16 MFMAs per product
        # (half the matrix-pipe work of bf16x3, no byte permutes in the split).  Activations are scaled per tensor by a power of
        # two derived from a magnitude bound that every producing launch records on the device (`self.amax`, one float per
        # tensor, zeroed at the start of a run and raised atomically by the epilogues: ymi_conv_desc.y_amax / x_amax)
        smode = os.environ.get('YOLACT_AMD_SPLIT', SPLIT_DEFAULT)
        self.split = smode == '1'
        self.h2 = smode == '2'
        self.mode_key = '|x3' if self.split else '|h2' if self.h2 else ''
        self.amax = torch.zeros(512 * AMAX_SLOT_FLOATS, dtype=torch.float32, device=device)     # 2 MB: 512 slots
        self._nslots = 0
        # YOLACT_AMD_SPLITK=0 keeps every GEMM a single pass (no split-K candidates in the tuner)
        self.splitk = os.environ.get('YOLACT_AMD_SPLITK', '1') == '1'
        # YOLACT_AMD_PIPE=0 keeps the pipelined kernel of csrc/dcn.hip out of the candidates of ORDINARY convolutions (A/B switch)
        self.pipe = os.environ.get('YOLACT_AMD_PIPE', '1') == '1'
        # YOLACT_AMD_WIDE_GUARD=0: keep fp16x2 tiles even on layers whose filters give away outlier input channels (tests of the
        # unguarded behaviour); default: such layers run on bf16x3 tiles (Packed.tiny_columns)
        self.wide_guard = os.environ.get('YOLACT_AMD_WIDE_GUARD', '1') == '1'
        # DCN layers: conv_offset_mask padded to 32 filters / tap-interleaved channel order (pack_offmask); 0 = the reference's 27 / order
        self.om_pad = os.environ.get('YOLACT_AMD_OM_PAD', '1') == '1'
        self.om_interleave = os.environ.get('YOLACT_AMD_OM_INTERLEAVE', '1') == '1'
        # YOLACT_AMD_NATIVE_EXEC=0: issue the op list from the Python loop (one ctypes call per launch) instead of csrc/plan_exec.cpp
        self.native_exec = device.type == 'cuda' and os.environ.get('YOLACT_AMD_NATIVE_EXEC', '1') == '1'
        self._native, self._native_events, self._stem_desc = None, None, None
        self.wide_ops = set()
        self.wide_layers = []       # names of the layers the outlier-channel guard took off the fp16x2 tiles (warned about once, below)
        self._sk_ws = {}
        self.down_on_side_stream = os.environ.get('YOLACT_AMD_DOWN_STREAM', 'B') == 'B'      # measured +1 %
        self.wino_alt, self._wino_packed, self._wino_ws = {}, {}, {}
        self._upsrc, self.wino_up = {}, {}     # 2x bilinear upsampling feeding a 3x3 conv: fused into its F(4x4) input transform
        self._done_event = None
        self._priors_timed = False
        self.param_stamp = None
        self.tune_misses = 0
        self._build()
        if self.wide_layers:
            import warnings
            warnings.warn('yolact_amd: %d layer(s) consume compensated OUTLIER channels (tiny filter columns on channels their producer '
                          'amplifies, Packed.tiny_columns) and run on bf16x3 / exact-fp32 tiles instead of fp16x2, without Winograd and '
                          'the fp16x2-only fusions: %s.  YOLACT_AMD_WIDE_GUARD=0 keeps fp16x2 (at reduced accuracy for such checkpoints).'
                          % (len(self.wide_layers), ', '.join(self.wide_layers[:12]) + (' ...' if len(self.wide_layers) > 12 else '')),
                          RuntimeWarning, stacklevel=2)
        self._bind_wino_workspaces()
        self.sections = [None if op[0] in ('record', 'wait', 'detect', 'nop') else self.section_of(op[2]) for op in self.ops]

    # ---- op emitters ---------------------------------------------------------------------------
    def _arena(self):
        # buffers are recycled only among ops of the SAME stream: program order on one stream makes reuse race-free,
        # across streams it would not be
        if self._cur == 'B':
            if not hasattr(self, 'arena_b'):
                self.arena_b = Arena(self.device)
            return self.arena_b
        return self.arena

    def _new(self, B, H, W, C, slot=None, gain=None) -> T:
        return T(self._arena().alloc(B * H * W * C), B, H, W, C, slot, gain)

    def _slot(self):
        """A fresh magnitude-bound slot (index into self.amax)."""
        self._nslots += 1
        assert self._nslots * AMAX_SLOT_FLOATS <= self.amax.numel()
        return self._nslots - 1

    def _slot_ptr(self, slot):
        return self.amax.data_ptr() + 4 * AMAX_SLOT_FLOATS * slot

    def bound(self, slot):
        """Host read of a slot's value (diagnostics / tests): the maximum over its sub-slots."""
        return float(self.amax[slot * AMAX_SLOT_FLOATS:(slot + 1) * AMAX_SLOT_FLOATS].max())

    def free(self, t):
        self._arena().free(t)

    def conv(self, name, x: T, pk: Packed, act=L.ACT_NONE, res: Optional[T] = None, res_mode=L.RES_NONE,
             res_after_act=0, out: Optional[T] = None, segs=None, dcn_offmask: Optional[T] = None) -> Optional[T]:
        """Emit one fused convolution. `segs`: list of (n0, n1, act, row_stride, batch_stride, ptr) overrides the
        single dense NHWC output."""
        assert x.C == pk.Cin, (name, x.C, pk.Cin)
        Ho, Wo = out_size(x.H, pk.kh, pk.stride, pk.pad), out_size(x.W, pk.kw, pk.stride, pk.pad)
        d = L.ConvDesc()
        d.x, d.w = x.ptr, pk.w.data_ptr()
        d.scale = pk.scale.data_ptr() if pk.scale is not None else None
        d.bias = pk.bias.data_ptr() if pk.bias is not None else None
        d.B, d.H, d.W, d.Cin, d.ldx = x.B, x.H, x.W, pk.Cin, x.C
        d.Ho, d.Wo, d.Cout = Ho, Wo, pk.Cout
        d.kh, d.kw, d.stride, d.pad, d.Kpad = pk.kh, pk.kw, pk.stride, pk.pad, pk.Kpad
        d.res_mode = res_mode
        d.res_after_act = res_after_act
        if res is not None:
            d.res, d.res_ld, d.res_H, d.res_W = res.ptr, res.C, res.H, res.W
            if res_mode == L.RES_ADD:
                assert (res.B, res.H, res.W, res.C) == (x.B, Ho, Wo, pk.Cout), name
        d.tile = L.TILE_AUTO
        d.cin_alg = pk.cin_alg
        d.cout_alg = pk.cout_alg
        # fp16x2 plans: a layer whose filters give away outlier input channels runs on the bf16x3 tiles (Packed.tiny_columns)
        wide = self.h2 and self.wide_guard and pk.Cin % 32 == 0 and pk.tiny_columns(in_gain=x.gain)
        pk.wide = wide
        if wide:
            self.wide_layers.append(name)
        # the gain this launch gives each channel it writes; a shortcut / FPN sum inherits the larger of its two sources
        og = pk.out_gain()
        if res is not None and res.gain is not None and res.gain.numel() == og.numel():
            og = torch.maximum(og, res.gain)
        self.last_out_gain = og
        if (self.split or wide) and dcn_offmask is None:
            d.w_x3 = pk.w3().data_ptr()
        yslot = self._slot()                    # fp16x2 plans: every launch records the magnitude bound of what it writes
        for _ in range(len(segs) - 1 if segs else 0):       # (one slot per output segment, consecutive: ABI 5)
            self._slot()
        if self.h2:                             # (csrc/common.h: one XCD-local atomic per wave at most); consumers read it as
            d.y_amax = self._slot_ptr(yslot)    # x_amax.  The bf16x3 / exact-fp32 plans need no bounds and do not pay for them
        if x.slot is not None:
            d.x_amax = self._slot_ptr(x.slot)
        if self.h2:                             # (DCN layers too: the gathered fp32 tile is split like any other A tile)
            assert x.slot is not None, name
            planes, sc2, winv = pk.h2()
            d.w_h2, d.scale_h2, d.winv_h2 = planes.data_ptr(), sc2.data_ptr(), winv.data_ptr()
        self.last_yslot = yslot
        y = None
        if segs is None:
            y = out if out is not None else self._new(x.B, Ho, Wo, pk.Cout)
            y.slot = yslot
            y.gain = og
            d.nseg = 1
            d.seg[0] = L.ConvSeg(0, pk.Cout, act, pk.Cout, Ho * Wo * pk.Cout, y.ptr)
        else:
            d.nseg = len(segs)
            for i, s in enumerate(segs):
                d.seg[i] = L.ConvSeg(*s)
        self.keepalive.append(pk)
        if not hasattr(self, '_desc_info'):
            self._desc_info = {}
        self._desc_info[C.addressof(d)] = (pk, res.slot if res is not None else None)     # (the chain fusion needs both: l1_gain, res_amax)
        wino = None
        if (self.use_winograd and dcn_offmask is None and out is None and pk.weight_oihw is not None and not wide
                and wino_eligible(pk, res, segs, act, x.C)):
            wino = [w for w in (self._wino_op(x, pk, act, y, name, segs, m) for m in self.wino_variants) if w is not None]
            wino = wino or None
        if dcn_offmask is not None:
            dd = L.DcnDesc()
            dd.conv = d
            dd.offmask, dd.ldo = dcn_offmask.ptr, dcn_offmask.C
            dd.om_layout = 1 if (self.om_pad and self.om_interleave) else 0
            self.ops.append((self.lib.ymi_dcn_v2_forward_f32, C.pointer(dd), name, self._cur))
            self.conv_meta.append((name, dd.conv))
            if wide:
                self.wide_ops.add(len(self.ops) - 1)
        else:
            self.ops.append((self.lib.ymi_conv2d_nhwc_f32, C.pointer(d), name, self._cur))
            self.conv_meta.append((name, d))
            if wide:
                self.wide_ops.add(len(self.ops) - 1)
            if wino is not None:
                self.wino_alt[len(self.ops) - 1] = wino      # op index -> alternative; autotune picks the faster one
                if id(x) in self._upsrc:
                    self.wino_up[len(self.ops) - 1] = self._upsrc[id(x)]
        return y

    def _wino_op(self, x: T, pk: Packed, act, y: Optional[T], name, segs=None, m=2):
        """Winograd alternative of a 3x3 / stride-1 conv: descriptor + per-stream workspaces (V, M)."""
        key = (id(pk), m)
        wp = self._wino_packed.get(key)
        if wp is None:
            wp = self._wino_packed[key] = WinoPacked(pk.weight_oihw, self.device, m)
        th, tw = (x.H + m - 1) // m, (x.W + m - 1) // m
        Tn = x.B * th * tw
        Ng = _ceil(pk.Cout, 4)
        if Tn * max(pk.Cin, Ng) >= (1 << 29):
            return None
        g = (m + 2) * (m + 2)
        need_v, need_m = g * Tn * pk.Cin, g * Tn * Ng
        ws = self._wino_ws.setdefault(self._cur, [None, None])
        if ws[0] is None or ws[0].numel() < need_v:
            ws[0] = torch.empty(need_v, dtype=torch.float32, device=self.device)
        if ws[1] is None or ws[1].numel() < need_m:
            ws[1] = torch.empty(need_m, dtype=torch.float32, device=self.device)
        d = L.WinoDesc()
        d.x, d.u = x.ptr, wp.u.data_ptr()
        if self.split:
            d.u_x3 = wp.u3().data_ptr()
        if self.h2:
            planes, uinv = wp.h2()
            d.u_h2, d.uinv_h2 = planes.data_ptr(), uinv.data_ptr()
        if x.slot is not None:
            d.x_amax = self._slot_ptr(x.slot)
        if self.h2:
            d.y_amax = self._slot_ptr(self.last_yslot)    # same slot as the direct descriptor of this layer
        if segs is None:
            d.y = y.ptr
        else:
            d.nseg = len(segs)
            for i, s in enumerate(segs):
                d.seg[i] = L.ConvSeg(*s)
        d.scale = pk.scale.data_ptr() if pk.scale is not None else None
        d.bias = pk.bias.data_ptr() if pk.bias is not None else None
        d.B, d.H, d.W, d.C, d.Cout, d.act, d.tile, d.m = x.B, x.H, x.W, pk.Cin, pk.Cout, act, L.TILE_AUTO, m
        d.cout_alg = pk.cout_alg
        return d

    def _bind_wino_workspaces(self):
        """Workspaces may have been re-allocated (grown) while the plan was built: point every descriptor at the final ones."""
        for idx, alts in self.wino_alt.items():
            ws = self._wino_ws[self.ops[idx][3]]
            for d in alts:
                d.V, d.M = ws[0].data_ptr(), ws[1].data_ptr()

    def call(self, fn, *args, name=''):
        self.ops.append((fn, args, name, self._cur))

    # ---- FPN laterals of the lower levels, launched EARLY on the side stream (round 5) ------------------------------------------------
    # lat_layers[i](C_j) depends only on backbone stage j (yolact.py:324-334); the top-down sum x_j = up(x_{j+1}) + lat(C_j) needs the
    # level above, i.e. the END of the backbone.  Fused in the lateral's epilogue (YMI_RES_BILINEAR) both sit on the critical path
    # behind C5; split, the lateral GEMMs of C3 / C4 run on stream B beside the later backbone stages — whose 35 x 35 / 18 x 18 launches
    # leave a fifth to a third of the CUs idle — and only ymi_bilinear_add_nhwc_f32 (a 20 us stream) stays behind C5.  Same
    # interpolation, same association (results differ only through the tile the residual-free lateral GEMM runs on).  MEASURED (session r5d, same box, alternating, profiles/r05_ab_runs.txt): 2011 - 2015
    # images/s with it against 2017 - 2028 without — the laterals slow the backbone launches they share the chip with by as much as
    # they save behind C5 (the finding of DESIGN 3.5 again: summed kernel time rises when streams overlap).  Hence OPT-IN:
    # YOLACT_AMD_EARLY_LAT=1; the default keeps the fused epilogue.
    def _stage_done(self, li, t):
        sel = self.net.backbone_selected
        if not self._early_lat_on or li not in sel or sel.index(li) >= len(sel) - 1:
            return
        self._pending_lat.append((li, t))

    def _flush_early_laterals(self):
        sel = self.net.backbone_selected
        n = len(sel)
        for li, t in self._pending_lat:
            j = sel.index(li)
            i = n - 1 - j
            # B may only start behind what A has issued SO FAR: the stage output is complete, and the projection shortcut the
            # side stream computed for the block just finished has been consumed (its buffer went back to B's pool and may be the
            # very one this lateral's output gets: tests/test_plan_schedule.py)
            self.record('latgo%d' % i)
            self.on('B')
            self.wait('latgo%d' % i)
            raw = self.conv('fpn.lat%d' % i, t, self._pack(self.net.fpn.lat_layers[i]))
            self.record('lat%d' % i)
            self.on('A')
            self.early_lat[j] = raw
        self._pending_lat = []

    # stream control (no-ops in single-stream mode): ops emitted after on('B') go to the side stream
    def on(self, which):
        self._cur = which if self.two_streams else 'A'

    def record(self, ev):
        if self.two_streams:
            self.events[ev] = torch.cuda.Event() if self.device.type == 'cuda' else None
            self.ops.append(('record', ev, ev, self._cur))

    def wait(self, ev):
        if self.two_streams:
            self.ops.append(('wait', ev, ev, self._cur))

    # ---- filters, packed once per plan -----------------------------------------------------------------------------------------
    def _pack(self, conv, bn=None, cin_pad=None) -> Packed:
        """pack_module, memoised per (conv, bn): Plan._rebalance_outliers edits the Packed objects of a stage BEFORE the op emitters ask
        for them."""
        key = (id(conv), id(bn), cin_pad)
        pk = self._pk_cache.get(key)
        if pk is None:
            pk = self._pk_cache[key] = pack_module(conv, bn, self.device, cin_pad)
        return pk

    def _rebalance_outliers(self, bb):
        """COMPENSATED OUTLIER CHANNELS, rebalanced at pack time (round 6).  A BN-folded checkpoint can carry channels whose producers
        amplify them 2^k over the typical channel (a huge gamma / sigma) while every consumer's filters hold the inverse — exact in fp32,
        but the fp16x2 tiles give an activation tensor ONE power-of-two scale and a filter row ONE: the typical channels of such a
        tensor sit k binades below its bound.  Round 4 moved the consuming layers to bf16x3 tiles (Packed.tiny_columns: 18 layers, no
        Winograd, no fusions; bench.py secondary.outlier_plan: 6.85 ms instead of 3.94 per batch-8 step).  Here the re-parametrisation
        is UNDONE instead: channel c of a ResNet stage tensor is multiplied by 2^-k in every producer (each block's folded bn3 and the
        projection shortcut's BN: the identity shortcuts carry the same factor) and by 2^k in every consumer's filters (the conv1 of
        the following blocks, the next stage's conv1 / projection shortcut, the FPN lateral); likewise proto_net[0] -> proto_net[2].
        Powers of two commute with fp32 rounding and ReLU, so in fp32 arithmetic every product is unchanged — the network is the same
        function, with balanced tensors.  k = min(producer excess, consumer deficit) per channel, only where both exceed 2^6: a channel
        that is merely large (no tiny consumer column) is left alone — moving ITS range into the filter rows would cost them what it
        saves the tensor.  YOLACT_AMD_REBALANCE=0 keeps the checkpoint's own parametrisation (then the guard of round 4 acts)."""
        self.rebalanced = []
        if not (self.h2 and os.environ.get('YOLACT_AMD_REBALANCE', '1') == '1' and isinstance(bb, M.ResNetBackbone)):
            return

        def exps(gain, cols, thr=6):
            g = gain.double().clamp_min(0)
            pos = g[g > 0]
            if pos.numel() < 8:
                return None
            kg = torch.floor(torch.log2((g / pos.median()).clamp_min(2.0 ** -126)))
            c = cols.double()
            cpos = c[c > 0]
            if cpos.numel() < 8:
                return None
            kc = torch.floor(torch.log2((cpos.median() / c.clamp_min(2.0 ** -126))))
            k = torch.minimum(kg, kc).clamp(min=0, max=40)
            k = torch.where((kg >= thr) & (kc >= thr) & (c > 0), k, torch.zeros_like(k))
            return k if bool((k > 0).any()) else None
        layers = list(bb.layers)
        sel = list(self.net.backbone_selected)
        fpn = self.net.fpn
        for li, layer in enumerate(layers):
            blocks = list(layer)
            if not blocks or blocks[0].downsample is None:
                continue
            prod = [self._pack(b.conv3, b.bn3) for b in blocks] + [self._pack(blocks[0].downsample[0], blocks[0].downsample[1])]
            cons = [self._pack(b.conv1, b.bn1) for b in blocks[1:]]
            if li + 1 < len(layers):
                nb = list(layers[li + 1])[0]
                cons.append(self._pack(nb.conv1, nb.bn1))
                if nb.downsample is not None:
                    cons.append(self._pack(nb.downsample[0], nb.downsample[1]))
            if li in sel:
                cons.append(self._pack(fpn.lat_layers[len(sel) - 1 - sel.index(li)]))
            if not cons or any(pk.Cin != prod[0].Cout or pk.Kpad != pk.kh * pk.kw * pk.Cin for pk in cons):
                continue
            gain = torch.stack([pk.out_gain()[:prod[0].Cout] for pk in prod]).amax(0)
            cols = torch.stack([pk.in_column_max() for pk in cons]).amax(0)
            k = exps(gain, cols)
            if k is None:
                continue
            down, up = torch.pow(2.0, -k).float(), torch.pow(2.0, k).float()
            for pk in prod:
                pk.scale_out_channels(down)
            for pk in cons:
                pk.scale_in_channels(up)
            self.rebalanced.append(('C%d' % (li + 2), int((k > 0).sum()), int(k.max())))
        # proto_net[0] (3x3 + ReLU) -> proto_net[2]: the same signature one level down the P3 branch
        pm_ = list(self.net.proto_net)
        self._proto0_down = None
        if (len(pm_) > 2 and isinstance(pm_[0], nn.Conv2d) and isinstance(pm_[1], nn.ReLU) and isinstance(pm_[2], nn.Conv2d)
                and pm_[0].bias is not None):
            p0, p2 = self._pack(pm_[0]), self._pack(pm_[2])
            if p2.Cin == p0.Cout and p2.Kpad == p2.kh * p2.kw * p2.Cin:
                k = exps(p0.out_gain(), p2.in_column_max())
                if k is not None:
                    down, up = torch.pow(2.0, -k).float(), torch.pow(2.0, k).float()
                    p0.scale_out_channels(down)
                    p2.scale_in_channels(up)
                    self._proto0_down = down
                    self.rebalanced.append(('proto_net.0', int((k > 0).sum()), int(k.max())))

    # ---- graph construction --------------------------------------------------------------------
    def _build(self):
        net, B, H, W, dev = self.net, self.B, self.H, self.W, self.device
        cfg = net.cfg
        lib = self.lib
        ar = self.arena

        self.early_lat, self._pending_lat = {}, []
        self._early_lat_on = self.two_streams and os.environ.get('YOLACT_AMD_EARLY_LAT', '0') == '1'
        bb = net.backbone
        self._pk_cache = {}
        self._rebalance_outliers(bb)
        # ResNet stem (fp16x2 plans): layout change + 7x7/2 conv + BN + ReLU + 3x3/2 max-pool in ONE launch straight from the NCHW
        # input (csrc/stem.hip: 0.107 vs 0.168 ms for the three launches at batch 8, 0.020 vs 0.035 at batch 1, bit-identical (see the test for the exact statement)
        # output; profiles/r03_stem_probe.txt).  YOLACT_AMD_FUSED_STEM=0 keeps the separate launches.
        c1 = getattr(bb, 'conv1', None)
        self.fused_stem = (self.h2 and isinstance(bb, M.ResNetBackbone) and os.environ.get('YOLACT_AMD_FUSED_STEM', '1') == '1'
                           and c1 is not None and tuple(c1.kernel_size) == (7, 7) and tuple(c1.stride) == (2, 2)
                           and tuple(c1.padding) == (3, 3) and c1.in_channels == 3 and c1.out_channels == 64 and H >= 7 and W >= 7)
        x4 = None
        if not self.fused_stem:
            # input: NCHW fp32 -> NHWC with C padded to 4 (pointer patched per call)
            x4 = self._new(B, H, W, 4, slot=self._slot())
            self.in_args = [None, x4.ptr, B, 3, H, W]
            self.in_amax = (x4.ptr, B * H * W * 4, self._slot_ptr(x4.slot))    # the input's magnitude bound: ymi_amax_f32
            self.ops.append(('input', None, 'nchw_to_nhwc4', 'A'))
        if isinstance(bb, M.ResNetBackbone):
            outs = self._resnet(bb, x4)
        else:
            outs = self._darknet(bb, x4)
        self._flush_early_laterals()           # (anything still pending: a stage whose successor has no block to hide behind)
        sel = [outs[i] for i in net.backbone_selected]
        for i, t in enumerate(outs):
            if i not in net.backbone_selected and t is not None:
                ar.free(t)

        # FPN laterals + top-down sums (yolact.py:319-341) on stream A
        fpn = net.fpn
        n = len(sel)
        sums = [None] * n
        prev = None
        for i in range(n):
            j = n - 1 - i
            pk = self._pack(fpn.lat_layers[i])
            if j in self.early_lat and prev is not None:
                raw = self.early_lat[j]                 # lat(C_j), computed on B long ago; the sum happens here, in place
                self.wait('lat%d' % i)
                slot = self._slot()
                self.call(lib.ymi_bilinear_add_nhwc_f32, prev.ptr, raw.ptr, raw.B, prev.H, prev.W, raw.C, raw.H, raw.W,
                          self._slot_ptr(slot) if self.h2 else None, name='fpn.add%d' % i)
                raw.slot = slot
                if raw.gain is not None and prev.gain is not None and raw.gain.numel() == prev.gain.numel():
                    raw.gain = torch.maximum(raw.gain, prev.gain)
                sums[j] = raw
            elif prev is None:
                sums[j] = self.conv('fpn.lat%d' % i, sel[j], pk)
            else:
                sums[j] = self.conv('fpn.lat%d' % i, sel[j], pk, res=prev, res_mode=L.RES_BILINEAR)
            prev = sums[j]
        for t in sel:
            ar.free(t)

        # level geometry (pred convs keep the size, downsample convs are 3x3 / stride 2 / pad 1)
        shapes = [(t.H, t.W) for t in sums]
        for _ in fpn.downsample_layers:
            shapes.append((out_size(shapes[-1][0], 3, 2, 1), out_size(shapes[-1][1], 3, 2, 1)))
        self.feat_shapes = shapes
        nlev = len(shapes)

        # prediction heads (yolact.py:133-212), shared weights, one merged GEMM per level writing straight into
        # the level-concatenated [B,P,k] tensors
        pm = net.prediction_layers[0]
        A = pm.num_priors
        Ccls, D = cfg.num_classes, net.mask_dim
        cells = [h * w for h, w in shapes]
        P = sum(cells) * A
        self.P, self.A, self.D, self.Ccls = P, A, D, Ccls
        # class rows are padded to a multiple of 4 floats (81 -> 84): every head output row is then 16-byte aligned and the
        # head GEMM / Winograd output transform store float4s (the 243-float rows of the dense layout forced scalar
        # stores on 69 % of the head's output channels); Detect reads the rows with stride `conf_ld`
        Cp = _ceil(Ccls, 4) if os.environ.get('YOLACT_AMD_PAD_CONF', '1') == '1' else Ccls
        self.conf_ld = Cp
        self.loc = torch.empty(B, P, 4, device=dev)
        self.conf = torch.zeros(B, P, Cp, device=dev)
        self.coef = torch.empty(B, P, D, device=dev)
        up_pk = [pack_module(m, device=dev) for m in pm.upfeature if isinstance(m, nn.Conv2d)] \
            if hasattr(pm, 'upfeature') else []
        # row order bbox | coef | conf keeps the bbox and coef segments 16-byte aligned (vector stores)
        cw, cb = pm.conf_layer.weight, pm.conf_layer.bias
        if Cp != Ccls:      # zero filters for the padding columns of every anchor's class row
            cw = torch.nn.functional.pad(cw.view(A, Ccls, *cw.shape[1:]), (0, 0, 0, 0, 0, 0, 0, Cp - Ccls)).reshape(A * Cp, *cw.shape[1:])
            cb = torch.nn.functional.pad(cb.view(A, Ccls), (0, Cp - Ccls)).reshape(A * Cp)
        wcat = torch.cat([pm.bbox_layer.weight, pm.mask_layer.weight, cw], 0)
        bcat = torch.cat([pm.bbox_layer.bias, pm.mask_layer.bias, cb], 0)
        hp = pm.bbox_layer
        head_pk = Packed(wcat, bcat, None, hp.stride[0], hp.padding[0], None, dev)
        head_pk.cout_alg = A * (4 + D + Ccls)
        coef_act = {'tanh': L.ACT_TANH, 'sigmoid': L.ACT_SIGMOID, 'relu': L.ACT_RELU, 'none': L.ACT_NONE}[
            act_name(cfg.mask_proto_coeff_activation)]
        n_b, n_c, n_m = A * 4, A * Cp, A * D
        offs = [sum(cells[:l]) * A for l in range(nlev)]
        bbc = cfg.backbone
        pri = []
        for lvl, (fh, fw) in enumerate(shapes):
            pri += make_priors_host(fh, fw, bbc.pred_scales[lvl], bbc.pred_aspect_ratios[lvl], cfg.max_size, bbc)

        # head0.upfeature and proto_net[0] are both 3x3 / pad 1 / 256 -> 256 + ReLU convolutions of the SAME tensor (P3): one
        # launch with their filters concatenated along Cout shares the input tile (direct kernel) or the whole Winograd input
        # transform V (one transform instead of two, one 512-column GEMM), and scatters to two dense outputs
        pmods = list(net.proto_net)
        up_convs = [m for m in pm.upfeature if isinstance(m, nn.Conv2d)] if hasattr(pm, 'upfeature') else []
        self._merged_p3 = None
        merge_p3 = (os.environ.get('YOLACT_AMD_MERGE_P3', '1') == '1' and len(up_convs) == 1 and len(pmods) > 2
                    and isinstance(pmods[0], nn.Conv2d) and isinstance(pmods[1], nn.ReLU)
                    and all(c.kernel_size == (3, 3) and c.padding == (1, 1) and c.stride == (1, 1) and c.bias is not None
                            for c in (up_convs[0], pmods[0]))
                    and up_convs[0].in_channels == pmods[0].in_channels and up_convs[0].out_channels % 4 == 0
                    and pmods[0].out_channels % 4 == 0)

        def head(lvl, f):
            off = offs[lvl]
            u = f
            if lvl == 0 and merge_p3:
                cu, cp = up_convs[0], pmods[0]
                u = self._new(f.B, f.H, f.W, cu.out_channels)
                t0 = self._new(f.B, f.H, f.W, cp.out_channels)
                pkm = Packed(torch.cat([cu.weight, cp.weight], 0), torch.cat([cu.bias, cp.bias], 0), None, 1, 1, None, dev)
                if getattr(self, '_proto0_down', None) is not None:      # (the proto_net[0] half of the merged launch: _rebalance_outliers)
                    pkm.scale_out_channels(torch.cat([torch.ones(cu.out_channels), self._proto0_down]))
                hw = f.H * f.W
                self.conv('head0.up0+proto.0', f, pkm, segs=[
                    (0, cu.out_channels, L.ACT_RELU, cu.out_channels, hw * cu.out_channels, u.ptr),
                    (cu.out_channels, cu.out_channels + cp.out_channels, L.ACT_RELU, cp.out_channels, hw * cp.out_channels,
                     t0.ptr)])
                # one bound PER HALF (segment k raises slot k of consecutive slots): the two halves are different tensors — a large
                # proto_net[0] channel must not coarsen the fp16x2 scale of the head's input (round-4 outlier stress test)
                u.slot, t0.slot = self.last_yslot, self.last_yslot + 1
                u.gain, t0.gain = self.last_out_gain[:cu.out_channels], self.last_out_gain[cu.out_channels:]
                self._merged_p3 = t0
            else:
                for k, pk in enumerate(up_pk):
                    nu = self.conv('head%d.up%d' % (lvl, k), u, pk, act=L.ACT_RELU)
                    if u is not f:
                        self.free(u)
                    u = nu
            segs = [
                (0, n_b, L.ACT_NONE, n_b, P * 4, self.loc.data_ptr() + off * 4 * 4),
                (n_b, n_b + n_m, coef_act, n_m, P * D, self.coef.data_ptr() + off * D * 4),
                (n_b + n_m, n_b + n_m + n_c, L.ACT_NONE, n_c, P * Cp, self.conf.data_ptr() + off * Cp * 4),
            ]
            self.conv('head%d.out' % lvl, u, head_pk, segs=segs)
            if u is not f:
                self.free(u)

        def pred(j):   # pred_layers are stored top-down: index n-1-j belongs to level j (yolact.py:286-289)
            i = n - 1 - j
            return self.conv('fpn.pred%d' % i, sums[j], pack_module(fpn.pred_layers[i], device=dev), act=L.ACT_RELU)

        assert net.proto_src == 0, 'prototypes are taken from P3 in every shipped config'
        # ---- fork: the P3 branch (75 % of the remaining FLOPs) stays on A, P4..P7 + Detect go to the side stream B
        self.record('fork')
        self.on('B')
        self.wait('fork')
        self.on('A')
        p3 = pred(0)
        self.free(sums[0])
        self.on('B')
        feats_b = [pred(j) for j in range(1, n)]
        for j in range(1, n):
            self.free(sums[j])
        for i, m in enumerate(fpn.downsample_layers):
            feats_b.append(self.conv('fpn.down%d' % i, feats_b[-1], pack_module(m, device=dev)))
        # the P3 head stays on A (moving it to B as well measured 862 vs 877 images/s)
        self.on('A')
        head(0, p3)
        self.record('head0')
        self.on('B')
        for lvl in range(1, nlev):
            head(lvl, feats_b[lvl - 1])
        self.wait('head0')
        for f in feats_b:
            self.free(f)
        self.ops.append(('detect', None, 'detect', self._cur))
        self.record('b_done')
        self.on('A')

        # protonet (utils/functions.py:163-213 + yolact.py:588-599) on A; the last conv writes the per-call proto tensor
        t = p3
        mods = list(net.proto_net)
        conv_idx = [i for i, m in enumerate(mods) if isinstance(m, nn.Conv2d)]
        self.proto_patch = None
        self.proto_patch_wino = None     # the F(4x4) descriptor that writes the proto tensor when the last 1x1 is fused into it (wino_proj)
        self.wino_proj = None            # (op index of the 3x3 before the last 1x1, op index of that 1x1): see _tune_winograd
        prev_conv_op = None
        held = None      # low-res source of a fusable 2x upsampling: the consuming conv's F(4x4) input transform may read IT instead
                         # of the upsampled tensor (ymi_wino_desc.x_up), so its buffer stays allocated until that conv's output
                         # has been allocated — freed earlier, the best-fit arena could hand it to the conv's own output
        for i, m in enumerate(mods):
            if i == 0 and self._merged_p3 is not None:      # proto.0 was computed together with head0.up0
                self.free(t)
                t = self._merged_p3
                continue
            if isinstance(m, nn.Conv2d):
                last = i == conv_idx[-1]
                has_relu = (i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU))
                if last:
                    a = {'relu': L.ACT_RELU, 'none': L.ACT_NONE, 'sigmoid': L.ACT_SIGMOID, 'tanh': L.ACT_TANH}[
                        act_name(cfg.mask_proto_prototype_activation)]
                    assert not has_relu
                else:
                    a = L.ACT_RELU if has_relu else L.ACT_NONE
                pk = self._pack(m)
                if last:
                    Ho, Wo = out_size(t.H, pk.kh, pk.stride, pk.pad), out_size(t.W, pk.kw, pk.stride, pk.pad)
                    self.proto_shape = (B, Ho, Wo, pk.Cout)
                    self.conv('proto.%d' % i, t, pk, segs=[(0, pk.Cout, a, pk.Cout, Ho * Wo * pk.Cout, None)])
                    self.proto_patch = self.ops[-1][1].contents  # seg[0].ptr set per call
                    nt = None
                    # conv3x3(256 -> 256) + ReLU -> conv1x1(256 -> <= 32): when the 3x3 runs as F(4x4,3x3) its output transform can
                    # multiply each tile by the 1x1's filters and write the prototypes directly (ymi_wino_desc.proj_*)
                    if (self.h2 and prev_conv_op is not None and prev_conv_op in self.wino_alt and (pk.kh, pk.kw, pk.stride, pk.pad) == (1, 1, 1, 0)
                            and pk.Cin == 256 and pk.Cout <= 32 and pk.Cout % 4 == 0 and a <= L.ACT_LEAKY01 and not getattr(pk, 'wide', False)
                            and os.environ.get('YOLACT_AMD_WINO_PROJ', '1') == '1'):
                        self.wino_proj = (prev_conv_op, len(self.ops) - 1)
                else:
                    nt = self.conv('proto.%d' % i, t, pk, act=a)
                    prev_conv_op = len(self.ops) - 1
                self.free(t)
                if held is not None:
                    self.free(held)
                    held = None
                t = nt
            elif isinstance(m, M.InterpolateModule):
                prev_conv_op = None
                s = int(m.scale_factor)
                has_relu = (i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU))
                y = self._new(t.B, t.H * s, t.W * s, t.C, slot=t.slot, gain=t.gain)   # convex interpolation: the input's bound (and gain) holds
                self.call(lib.ymi_bilinear_nhwc_f32, t.ptr, y.ptr, t.B, t.H, t.W, t.C, y.H, y.W,
                          C.c_float(1.0 / s), C.c_float(1.0 / s), 1 if has_relu else 0, name='proto.interp')
                if s == 2 and os.environ.get('YOLACT_AMD_FUSED_UPSAMPLE', '1') == '1':
                    # (the tuner drops this launch when the consuming conv runs as F(4x4,3x3): csrc/winograd.hip UPS)
                    self._upsrc[id(y)] = (len(self.ops) - 1, t.ptr, 1 if has_relu else 0)
                    held = t
                else:
                    self.free(t)
                t = y
        assert self.proto_patch is not None and held is None
        self.wait('b_done')
        self.priors = torch.tensor(pri, dtype=torch.float32).view(-1, 4).to(dev)
        assert self.priors.shape[0] == P

    def _resnet(self, bb: M.ResNetBackbone, x4: T):
        dev, ar, lib = self.device, self.arena, self.lib
        pk0 = pack_module(bb.conv1, bb.bn1, dev, cin_pad=4)
        if self.fused_stem:
            B, H, W = self.B, self.H, self.W
            Hs, Ws = out_size(H, 7, 2, 3), out_size(W, 7, 2, 3)
            Hp, Wp = out_size(Hs, 3, 2, 1), out_size(Ws, 3, 2, 1)
            x = self._new(B, Hp, Wp, 64, slot=self._slot(), gain=pk0.out_gain())
            planes, sc2, winv = pk0.h2()
            sd = L.StemDesc()
            sd.y, sd.B, sd.H, sd.W, sd.cout_pad, sd.kpad = x.ptr, B, H, W, pk0.CoutPad, pk0.Kpad
            sd.w_h2, sd.scale_h2, sd.bias, sd.y_amax = planes.data_ptr(), sc2.data_ptr(), pk0.bias.data_ptr(), self._slot_ptr(x.slot)
            self.keepalive.append(pk0)
            md = L.ConvDesc()          # accounting only (FLOPs of the 7x7 conv; bench.py takes the bytes of the fused launch)
            md.B, md.H, md.W, md.Cin, md.ldx, md.Ho, md.Wo, md.Cout = B, H, W, 4, 4, Hs, Ws, 64
            md.kh, md.kw, md.stride, md.pad, md.Kpad, md.cin_alg, md.cout_alg = 7, 7, 2, 3, pk0.Kpad, 3, 64
            self.conv_meta.append(('stem', md))
            self.ops.append(('stem', sd, 'stem+maxpool', 'A'))
            self._stem_desc = sd
        else:
            stem = self.conv('stem', x4, pk0, act=L.ACT_RELU)
            ar.free(x4)
            Hp, Wp = out_size(stem.H, 3, 2, 1), out_size(stem.W, 3, 2, 1)
            x = self._new(stem.B, Hp, Wp, stem.C, slot=stem.slot, gain=stem.gain)        # max-pooling cannot raise the magnitude bound
            self.call(lib.ymi_maxpool3x3s2_nhwc_f32, stem.ptr, x.ptr, stem.B, stem.H, stem.W, stem.C, Hp, Wp, name='maxpool')
            ar.free(stem)
        outs = []
        for li, layer in enumerate(bb.layers):
            for bi, blk in enumerate(layer):
                nm = 'layer%d.%d' % (li, bi)
                side_down = blk.downsample is not None and self.two_streams and self.down_on_side_stream
                if side_down:
                    # the projection shortcut only depends on the block input: run it on the side stream while A does
                    # conv1 / conv2 (fills their tails); A picks it up before conv3
                    self.record(nm + '.in')
                    self.on('B')
                    self.wait(nm + '.in')
                    res = self.conv(nm + '.down', x, self._pack(blk.downsample[0], blk.downsample[1]))
                    self.record(nm + '.down')
                    self.on('A')
                o1 = self.conv(nm + '.conv1', x, self._pack(blk.conv1, blk.bn1), act=L.ACT_RELU)
                if blk.use_dcn:
                    dcn = blk.conv2
                    om = self.conv(nm + '.offmask', o1, pack_offmask(dcn.conv_offset_mask, dev, self.om_interleave) if self.om_pad
                                   else pack_module(dcn.conv_offset_mask, None, dev))
                    pk = Packed(dcn.weight, dcn.bias, blk.bn2, dcn.stride[0], 1, None, dev)     # (stride is a pair, dcn_v2.py:62)
                    o2 = self.conv(nm + '.dcn', o1, pk, act=L.ACT_RELU, dcn_offmask=om)
                    ar.free(om)
                else:
                    o2 = self.conv(nm + '.conv2', o1, pack_module(blk.conv2, blk.bn2, dev), act=L.ACT_RELU)
                ar.free(o1)
                if side_down:
                    self.wait(nm + '.down')
                elif blk.downsample is not None:
                    res = self.conv(nm + '.down', x, self._pack(blk.downsample[0], blk.downsample[1]))
                else:
                    res = x
                y = self.conv(nm + '.conv3', o2, self._pack(blk.conv3, blk.bn3), act=L.ACT_RELU, res=res,
                              res_mode=L.RES_ADD)
                ar.free(o2)
                if side_down:
                    self.on('B')          # back to the side stream's pool: its next op waits for a later event of A
                    self.free(res)
                    self.on('A')
                elif res is not x:
                    ar.free(res)
                ar.free(x)
                x = y
                if bi == 0:                     # behind this stage's projection shortcut on B: the previous stages' FPN laterals
                    self._flush_early_laterals()
            # stage output: still needed by the FPN, so the next stage only borrows it
            outs.append(x)
            self._stage_done(li, x)
            x = _Borrowed(x)
        return outs

    def _darknet(self, bb: M.DarkNetBackbone, x4: T):
        dev, ar = self.device, self.arena

        def unit(name, t, seq, cin_pad=None, res=None):
            pk = pack_module(seq[0], seq[1], dev, cin_pad=cin_pad)
            if res is None:
                return self.conv(name, t, pk, act=L.ACT_LEAKY01)
            return self.conv(name, t, pk, act=L.ACT_LEAKY01, res=res, res_mode=L.RES_ADD, res_after_act=1)

        x = unit('preconv', x4, bb._preconv, cin_pad=4)
        ar.free(x4)
        outs = []
        for li, layer in enumerate(bb.layers):
            mods = list(layer)
            y = unit('dark%d.down' % li, x, mods[0])
            ar.free(x)
            x = y
            for bi, blk in enumerate(mods[1:]):
                a = unit('dark%d.%d.conv1' % (li, bi), x, blk.conv1)
                y = unit('dark%d.%d.conv2' % (li, bi), a, blk.conv2, res=x)
                ar.free(a)
                ar.free(x)
                x = y
                if bi == 0:
                    self._flush_early_laterals()
            outs.append(x)
            self._stage_done(li, x)
            x = _Borrowed(x)
        return outs

    # ---- execution -------------------------------------------------------------------------------
    @staticmethod
    def section_of(name):
        """The reference's timer section an op belongs to (yolact.py:570 'backbone', :574 'fpn', :581 'proto',
        :607 'pred_heads'); 'Detect' (detection.py:63) is timed by Detect.finish."""
        if name.startswith('fpn.'):
            return 'fpn'
        if name.startswith('proto.'):
            return 'proto'
        if name.startswith('head'):
            return 'pred_heads'
        return 'backbone'

    def mark_done(self):
        """Record, on the caller's current stream, that everything reading this plan's persistent buffers has been
        enqueued; the next run() — possibly on another stream — waits for it."""
        if self.device.type == 'cuda' and not torch.cuda.is_current_stream_capturing():
            if self._done_event is None:
                self._done_event = torch.cuda.Event()
            self._done_event.record(torch.cuda.current_stream(self.device))

    def run(self, x: torch.Tensor, detect=None, timer=None, skip_proto=False):
        """x [B,3,H,W] fp32 contiguous on the plan's device. Returns (proto, detect_result): the fresh proto tensor
        and whatever `detect(stream_ptr)` returned (None without a callback).  loc/conf/coef are the plan's persistent
        head buffers.  `detect` is invoked at the point of the op list where every head has been written; its
        kernels must be launched on the stream it is handed (the side stream in two-stream mode) while its output
        tensors are allocated by the caller's ambient stream — they are only consumed after the final join.
        `timer`: the reference's utils.timer module (or None): the op ranges are bracketed with its section names.
        The arena, the head buffers and the Winograd workspaces are shared by every run of this plan: callers serialise
        run() on the host (Yolact._run_lock_for(device)) and consecutive runs are ordered on the device by an event, so two user
        streams cannot overlap on the same buffers."""
        lib = self.lib
        cur = torch.cuda.current_stream(self.device)
        capturing = torch.cuda.is_current_stream_capturing()
        if self._done_event is not None and not capturing:
            cur.wait_event(self._done_event)
        sa = C.c_void_p(cur.cuda_stream)
        two = self.two_streams and self.overlap
        sb = C.c_void_p(self.stream_b.cuda_stream) if two else sa
        # skip_proto: cfg.eval_mask_branch is False at call time (eval.py --detect, eval.py:1067-1068): the reference computes no
        # prototypes (yolact.py:579-580) — the protonet's launches are left out of the op loop and None is returned for them
        proto = None
        if not skip_proto:
            proto = torch.empty(self.proto_shape, dtype=torch.float32, device=self.device)
            self.proto_patch.seg[0].ptr = proto.data_ptr()
            if self.proto_patch_wino is not None:
                self.proto_patch_wino.proj_y = proto.data_ptr()
        # only a module with the reference's timer API (utils/timer.py: start / stop / env) is driven
        if timer is not None and not all(hasattr(timer, a) for a in ('start', 'stop', 'env')):
            timer = None
        if timer is None and self.native_exec:
            nat = self._native_plan()
            if nat is not None:          # the whole op list in two native calls (csrc/plan_exec.cpp) around the Detect callback
                det = self._run_native(nat, x, sa, sb, two, detect, skip_proto)
                self.mark_done()
                return proto, det
        if self.h2:                  # magnitude bounds are re-derived by every run (on the caller's stream, ahead of every op)
            self.amax[:self._nslots * AMAX_SLOT_FLOATS].zero_()
        self._sec = None
        try:
            det = self._dispatch(x, cur, sa, sb, two, detect, timer, skip_proto)
        finally:
            if timer is not None and self._sec is not None:   # a failed launch must not leave a reference timer running
                timer.stop(self._sec)
        self.mark_done()
        return proto, det

    # ---- native executor (csrc/plan_exec.cpp, ABI 7) ------------------------------------------------------------------------------
    _SEC_ID = {'backbone': L.SEC_BACKBONE, 'fpn': L.SEC_FPN, 'proto': L.SEC_PROTO, 'pred_heads': L.SEC_HEADS}

    def _native_plan(self):
        """The op list as a ymi_plan_op array (rebuilt whenever an op was replaced: tuner, set_winograd), the index of the Detect
        marker, the event handles and the index of the input-layout op.  None when the list holds a call the executor does not know
        (the Python loop then runs it)."""
        key = (self.ops.version, len(self.ops)) if isinstance(self.ops, _OpList) else tuple(map(id, self.ops))
        if self._native is not None and self._native[0] == key:
            return self._native[1]
        lib = self.lib
        n = len(self.ops)
        arr = (L.PlanOp * (n + 1))()
        arr[0].kind, arr[0].stream = L.OP_MEMSET, 0
        arr[0].p[0], arr[0].i[0] = self.amax.data_ptr(), 4 * self._nslots * AMAX_SLOT_FLOATS
        ev_index, det_idx, in_idx, ok = {}, None, None, True
        by_fn = {id(lib.ymi_conv2d_nhwc_f32): L.OP_CONV, id(lib.ymi_conv3x3_winograd_f32): L.OP_WINO,
                 id(lib.ymi_dcn_v2_forward_f32): L.OP_DCN, id(lib.ymi_pointwise_chain_f32): L.OP_CHAIN}
        for k, (fn, args, name, where) in enumerate(self.ops):
            o = arr[k + 1]
            o.stream = 1 if where == 'B' else 0
            o.section = self._SEC_ID.get(self.sections[k], 0) if self.sections[k] else 0
            if fn == 'input':
                a = self.in_args
                o.kind, in_idx = L.OP_INPUT, k + 1
                o.p[1] = a[1]
                o.p[2] = self.in_amax[2] if self.h2 else None
                o.i[0], o.i[1], o.i[2], o.i[3] = a[2], a[3], a[4], a[5]
            elif fn == 'stem':
                o.kind, o.desc = L.OP_STEM, C.addressof(args)
            elif fn == 'nop':
                o.kind = L.OP_NOP
            elif fn in ('record', 'wait'):
                o.kind = L.OP_RECORD if fn == 'record' else L.OP_WAIT
                o.i[0] = ev_index.setdefault(args, len(ev_index))
            elif fn == 'detect':
                o.kind, det_idx = L.OP_NOP, k + 1
            elif id(fn) in by_fn:
                o.kind, o.desc = by_fn[id(fn)], C.addressof(args.contents)
            elif fn is lib.ymi_bilinear_nhwc_f32:
                o.kind = L.OP_BILINEAR
                o.p[0], o.p[1] = args[0], args[1]
                for q in range(6):
                    o.i[q] = args[2 + q]
                o.f[0], o.f[1], o.i[6] = args[8].value, args[9].value, args[10]
            elif fn is lib.ymi_maxpool3x3s2_nhwc_f32:
                o.kind = L.OP_MAXPOOL
                o.p[0], o.p[1] = args[0], args[1]
                for q in range(6):
                    o.i[q] = args[2 + q]
            elif fn is lib.ymi_bilinear_add_nhwc_f32:
                o.kind = L.OP_BILINEAR_ADD
                o.p[0], o.p[1], o.p[2] = args[0], args[1], args[8]
                for q in range(6):
                    o.i[q] = args[2 + q]
            else:
                ok = False
                break
        if not ok or det_idx is None:
            self._native = (key, None)
            return None
        if self._native_events is None or len(self._native_events) < len(ev_index):
            for h in (self._native_events or ()):       # regrown: the old handles are destroyed, not leaked
                if h:
                    lib.ymi_event_destroy(C.c_void_p(h))
            evs = (C.c_void_p * max(len(ev_index), 1))()
            for q in range(len(ev_index)):
                h = C.c_void_p()
                L.check(lib.ymi_event_create(C.byref(h)), 'ymi_event_create')
                evs[q] = h.value
            self._native_events = evs
        nat = (arr, det_idx, self._native_events, in_idx, n + 1)
        self._native = (key, nat)
        return nat

    def _run_native(self, nat, x, sa, sb, two, detect, skip_proto):
        arr, det_idx, evs, in_idx, n = nat
        lib = self.lib
        if in_idx is not None:
            arr[in_idx].p[0] = x.data_ptr()
        if self._stem_desc is not None:
            self._stem_desc.x = x.data_ptr()
        skip = (1 << L.SEC_PROTO) if skip_proto else 0
        failed = C.c_int32(-1)
        ov = 1 if two else 0

        def go(first, last):
            rc = lib.ymi_plan_run(arr, first, last, sa, sb, evs, ov, skip, C.byref(failed))
            if rc != 0:
                L.check(rc, self.ops[failed.value - 1][2] if failed.value > 0 else 'plan op')
        go(0 if self.h2 else 1, det_idx)
        det = None
        if detect is not None:
            det = detect(sb if self.ops[det_idx - 1][3] == 'B' else sa)
        go(det_idx + 1, n)
        return det

    def __del__(self):
        try:
            if getattr(self, '_native_events', None) is not None:
                for h in self._native_events:
                    if h:
                        self.lib.ymi_event_destroy(h)
        except Exception:
            pass

    def _dispatch(self, x, cur, sa, sb, two, detect, timer, skip_proto=False):
        """The flat op loop of run(): C-ABI launches on the two streams, event records / waits, the Detect callback."""
        lib = self.lib
        det = None
        for (fn, args, name, where), nsec in zip(self.ops, self.sections):
            s = sb if where == 'B' else sa
            if timer is not None and nsec is not None and nsec != self._sec:
                if self._sec is not None:
                    timer.stop(self._sec)
                timer.start(nsec)
                self._sec = nsec
                if nsec == 'pred_heads' and not self._priors_timed:
                    with timer.env('makepriors'):       # yolact.py:219: priors are host constants of the plan, built once
                        self._priors_timed = True
            if fn == 'input':
                a = self.in_args
                if self.h2:          # layout change + magnitude bound of the input in one launch
                    rc = lib.ymi_nchw_to_nhwc4_amax_f32(x.data_ptr(), a[1], a[2], a[3], a[4], a[5], self.in_amax[2], s)
                else:
                    rc = lib.ymi_nchw_to_nhwc4_f32(x.data_ptr(), a[1], a[2], a[3], a[4], a[5], s)
            elif fn == 'nop' or (skip_proto and nsec == 'proto'):
                continue
            elif fn == 'stem':
                args.x = x.data_ptr()
                rc = lib.ymi_stem_pool_f32(C.byref(args), s)
            elif fn == 'record':
                if two:
                    self.events[args].record(self.stream_b if where == 'B' else cur)
                continue
            elif fn == 'wait':
                if two:
                    (self.stream_b if where == 'B' else cur).wait_event(self.events[args])
                continue
            elif fn == 'detect':
                if detect is not None:
                    det = detect(s)
                continue
            elif isinstance(args, tuple):
                rc = fn(*args, s)
            else:
                rc = fn(args, s)
            if rc != 0:
                L.check(rc, name)
        return det

    # ---- tile / algorithm selection ------------------------------------------------------------------
    def tune(self, x: torch.Tensor, reps: int = 3):
        """Pick the tile of every conv launch and direct-vs-Winograd per 3x3 layer.

        Deterministic by default: the choices come from the SHIPPED table `yolact_amd/tune/gfx950.json` (measured on
        MI355X by tools/make_tune_table.py, keyed by layer shape), so two processes on two boxes run the same kernels
        in the same summation order and produce the same bits.  A shape the table does not know is measured on the
        device (HIP events on the launch stream, `self.tune_misses` counts them); such a plan is only reproducible
        within its process unless YOLACT_AMD_TUNE_CACHE names a file to persist the new entries in.

        YOLACT_AMD_AUTOTUNE: '1' (default) table + measure on miss | '0' no tuning at all (library heuristic tiles,
        direct kernels only) | 'table' table only, a miss falls back to the heuristic | 'force' ignore the tables and
        measure everything (what tools/make_tune_table.py uses)."""
        mode = os.environ.get('YOLACT_AMD_AUTOTUNE', '1')
        self.tune_table, self.wino_table, self.tune_misses = [], [], 0
        if mode == '0':
            return []
        cache_path = os.environ.get('YOLACT_AMD_TUNE_CACHE')
        disk = {}
        if mode != 'force':
            disk.update(load_tune_table(self.device))
            if cache_path and os.path.exists(cache_path):
                disk.update(_read_table_file(cache_path))
        n0 = len(disk)
        self.run(x)                      # realistic values in every buffer
        s = L.stream_ptr()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        measure = mode != 'table'
        self._tune_direct(e0, e1, s, reps, disk, measure)
        self._tune_winograd(e0, e1, s, reps, disk, measure)
        self._fuse_pointwise_chains(e0, e1, s, reps, disk, measure)
        torch.cuda.synchronize(self.device)
        if cache_path and (len(disk) != n0 or mode == 'force'):
            _write_table_file(cache_path, disk, self.device)
        self.tune_entries = disk
        return self.tune_table

    autotune = tune      # round-1 name

    def _splitk_ws(self, where, numel):
        """Per-stream workspace of the split-K partial sums (grown on demand; every layer of a stream shares it)."""
        ws = self._sk_ws.get(where)
        if ws is None or ws.numel() < numel:
            ws = self._sk_ws[where] = torch.empty(numel, dtype=torch.float32, device=self.device)
            for fn, dptr, _n, w in self.ops:          # re-point descriptors that already use the old buffer
                if fn is self.lib.ymi_conv2d_nhwc_f32 and w == where and dptr.contents.split_k > 1:
                    dptr.contents.split_ws = ws.data_ptr()
                elif fn is self.lib.ymi_dcn_v2_forward_f32 and w == where and dptr.contents.conv.split_k > 1:
                    dptr.contents.conv.split_ws = ws.data_ptr()
        return ws

    @staticmethod
    def _splitk_ok(d):
        """Shapes the split-K path takes (csrc/conv_igemm.hip run_splitk): small-map 1x1 convolutions with a long K."""
        return (d.kh == 1 and d.kw == 1 and d.pad == 0 and d.Cin == d.Kpad and d.Cin % 32 == 0 and d.nseg == 1
                and d.Cout % 4 == 0 and d.seg[0].act <= L.ACT_LEAKY01 and d.res_mode in (L.RES_NONE, L.RES_ADD)
                and d.B * d.Ho * d.Wo <= 16384 and d.Kpad >= 256)

    def _fuse_pointwise_chains(self, e0, e1, s, reps, disk, measure):
        """conv3 (1x1, 64 -> 256, + shortcut, ReLU) of a first-stage bottleneck directly followed by conv1 (1x1, 256 -> 64, ReLU) of
        the next one -> ONE ymi_pointwise_chain_f32 launch (csrc/chain.hip: the 256-channel tensor is written once and never read
        back); a conv3 of that shape WITHOUT such a successor (the last block of the stage) -> the same streaming kernel with its
        second layer switched off.  Either only where measured faster than the launches it replaces: the decision is a table entry
        like every other choice ('chain(B, H, W)' / 'chain1(B, H, W)': [1 | 0, ms of the plan's launches, ms of the chain launch]);
        YOLACT_AMD_CHAIN=0 keeps the plan's launches."""
        self.chain_table = []
        if not self.h2 or os.environ.get('YOLACT_AMD_CHAIN', '1') != '1':
            return
        lib = self.lib

        def timed(run):
            run()
            best = 1e30
            for _ in range(2):
                e0.record()
                for _ in range(reps):
                    run()
                e1.record()
                e1.synchronize()
                best = min(best, e0.elapsed_time(e1) / reps)
            return best
        i = 0
        while i < len(self.ops):
            f3, p3, n3, w3 = self.ops[i]
            i += 1
            if f3 is not lib.ymi_conv2d_nhwc_f32 or (i - 1) in self.wide_ops:
                continue
            d3 = p3.contents
            M = d3.B * d3.Ho * d3.Wo
            P_ = d3.Cin
            if not ((d3.kh, d3.kw, d3.stride, d3.pad, d3.nseg) == (1, 1, 1, 0, 1) and P_ in (64, 128, 256) and d3.Cout == 4 * P_
                    and d3.res_mode == L.RES_ADD and not d3.res_after_act and d3.seg[0].n0 == 0 and d3.seg[0].act <= L.ACT_LEAKY01
                    and d3.w_h2 and d3.x_amax and M * max(d3.seg[0].row_stride, d3.res_ld) < (1 << 29)):
                continue
            if P_ > 64 and os.environ.get('YOLACT_AMD_CHAIN2', '0') not in ('1', 'force'):       # csrc/chain2.hip is OPT-IN: measured 0.80x of the two launches it replaces (profiles/r06_chain2_probe.txt)
                continue
            pair = False
            if i < len(self.ops) and self.ops[i][0] is lib.ymi_conv2d_nhwc_f32 and self.ops[i][3] == w3 and i not in self.wide_ops:
                d1 = self.ops[i][1].contents
                pair = ((d1.kh, d1.kw, d1.stride, d1.pad, d1.Cin, d1.Cout, d1.nseg) == (1, 1, 1, 0, 4 * P_, P_, 1)
                        and d1.res_mode == L.RES_NONE and d1.x == d3.seg[0].ptr and d1.ldx == d3.seg[0].row_stride
                        and d1.seg[0].n0 == 0 and d1.seg[0].act <= L.ACT_LEAKY01 and bool(d1.w_h2))
            if P_ > 64 and not pair:          # csrc/chain2.hip without its second layer has nothing over the pipelined tiles
                continue
            cd = L.ChainDesc()
            cd.x, cd.res, cd.y = d3.x, d3.res, d3.seg[0].ptr
            cd.w_a_h2, cd.scale_a_h2, cd.bias_a = d3.w_h2, d3.scale_h2, d3.bias
            cd.x_amax, cd.y_amax = d3.x_amax, d3.y_amax
            cd.M, cd.ldx, cd.res_ld, cd.ldy = M, d3.ldx, d3.res_ld, d3.seg[0].row_stride
            cd.k_a, cd.n_a, cd.n_b, cd.cout_pad_a, cd.cout_pad_b = P_, 4 * P_, P_, _ceil(4 * P_, 128), _ceil(P_, 128)
            cd.act_a = d3.seg[0].act
            if P_ > 64:                       # the rigorous bound of y (include/yolact_amd.h, ABI 8)
                info = getattr(self, '_desc_info', {}).get(C.addressof(d3))
                if info is None or info[1] is None:
                    continue
                cd.gain_a, cd.bias_max_a = info[0].l1_gain()
                cd.res_amax = self._slot_ptr(info[1])
            name = n3
            if pair:
                f1, p1, n1, w1 = self.ops[i]
                cd.z, cd.ldz, cd.z_amax, cd.act_b = d1.seg[0].ptr, d1.seg[0].row_stride, d1.y_amax, d1.seg[0].act
                cd.w_b_h2, cd.scale_b_h2, cd.bias_b = d1.w_h2, d1.scale_h2, d1.bias
                name = n3 + '+' + n1
            if pair:
                # aliasing contract of ymi_pointwise_chain_f32 (include/yolact_amd.h): z must not overlap y or the residual, and may
                # alias the INPUT x only exactly in place (same base, same row stride): every block reads x tile T before it writes
                # z tile T and nobody else touches those rows.  The arena normally hands conv1's output the buffer conv3's input
                # just freed, which is that case; any other overlap (a z that landed in a buffer of another layout) is not fused.
                def span(ptr, ld):
                    return (int(ptr), int(ptr) + 4 * M * int(ld))

                def overlap(a, b):
                    return a[0] < b[1] and b[0] < a[1]
                zs, xs = span(cd.z, cd.ldz), span(cd.x, cd.ldx)
                in_place = int(cd.z) == int(cd.x) and cd.ldz == cd.ldx
                if overlap(zs, span(cd.y, cd.ldy)) or overlap(zs, span(cd.res, cd.res_ld)) or (overlap(zs, xs) and not in_place):
                    continue
            cptr = C.pointer(cd)
            key = ('chain' if pair else 'chain1') + str((d3.B, d3.Ho, d3.Wo) + ((P_,) if P_ > 64 else ())) + self.mode_key
            ent = disk.get(key)
            if P_ > 64 and os.environ.get('YOLACT_AMD_CHAIN2') == 'force':     # (A/B switch: install it whatever its isolated time says)
                ent = [1, 0.0, 0.0]
            if ent is None:
                self.tune_misses += 1
                if not measure or lib.ymi_pointwise_chain_f32(cptr, s) != 0:
                    continue
                t_plan = timed((lambda: (f3(p3, s), f1(p1, s))) if pair else (lambda: f3(p3, s)))
                t_one = timed(lambda: lib.ymi_pointwise_chain_f32(cptr, s))
                ent = disk[key] = [1 if t_one < 0.97 * t_plan else 0, round(t_plan, 4), round(t_one, 4)]
            self.chain_table.append((name, ent[0], ent[1], ent[2]))
            if ent[0] and lib.ymi_pointwise_chain_f32(cptr, s) == 0:
                self.keepalive.append(cd)
                self.ops[i - 1] = (lib.ymi_pointwise_chain_f32, cptr, name, w3)
                if pair:
                    self.ops[i] = ('nop', None, n1 + '[fused into ' + n3 + ']', w1)
                    i += 1

    def _apply_choice(self, fn, dptr, where, val, s):
        """Install a table value (tile id + 256 * split_k) in a descriptor; returns the launch status of one run."""
        d = dptr.contents.conv if fn is self.lib.ymi_dcn_v2_forward_f32 else dptr.contents
        tile, S = int(val) & 255, int(val) >> 8
        d.tile = tile
        if S > 1:
            is_dcn = fn is self.lib.ymi_dcn_v2_forward_f32
            if tile & L.TILE_DCNP:
                nk_ = d.Kpad // 32
                if -(-nk_ // S) * (S - 1) >= nk_:
                    return -1
            elif is_dcn or not self._splitk_ok(d) or (d.Kpad // 32) % S:
                return -1
            d.split_k = S
            d.split_ws = self._splitk_ws(where, S * d.B * d.Ho * d.Wo * d.Cout).data_ptr()
        else:
            d.split_k = 0
        return fn(dptr, s)

    @staticmethod
    def _pipe_ok(d):
        """Ordinary convolutions the pipelined kernel of csrc/dcn.hip takes (run_pipe): 3x3 / pad 1 or 1x1 / pad 0, Cin % 32 == 0, one
        dense output, activation none / ReLU / LeakyReLU, residual none / add."""
        return ((d.kh, d.kw, d.pad) in ((3, 3, 1), (1, 1, 0)) and d.Cin % 32 == 0 and d.Kpad == d.kh * d.kw * d.Cin and d.Kpad >= 64
                and d.nseg == 1 and d.Cout % 4 == 0 and d.seg[0].n0 == 0 and d.seg[0].act <= L.ACT_LEAKY01
                and d.seg[0].row_stride % 4 == 0 and d.seg[0].batch_stride == d.Ho * d.Wo * d.seg[0].row_stride
                and d.res_mode in (L.RES_NONE, L.RES_ADD))

    @staticmethod
    def dcnp_candidates(d, dcn=False):
        """(tile + 256 * split_k) candidates of the pipelined DCN kernel for a descriptor: every block tile unsplit, and — where a
        tile's grid alone leaves CUs idle (the 35x35 / 18x18 maps) — chunk-aligned K splits that bring the block count to 0.5 .. 2
        blocks per CU."""
        M, nk = d.B * d.Ho * d.Wo, d.Kpad // 32
        out = []
        for t, name in sorted(L.DCNP_TILES.items()):
            bm, bn = (int(v) for v in name[4:].split('w')[0].split('x'))
            if (bn > 128 and d.Cout < 256) or (dcn and t in L.DCNP_PLAIN_ONLY) or (bn == 32) != (d.Cout <= 32):
                continue                                  # (32-column tiles: the Cout <= 32 layers, and nothing else for those)
            tid = t | L.TILE_H2 | L.TILE_DCNP
            out.append(tid)
            blocks = -(-M // bm) * -(-d.Cout // bn)
            if blocks < 400 and d.Cout % 4 == 0:
                for S in (2, 3, 4, 5, 6, 8, 9, 12, 16):
                    per = -(-nk // S)                     # chunks per range (the last range may be shorter, never empty)
                    if per >= 4 and per * (S - 1) < nk and 128 <= blocks * S <= 1100:
                        out.append(tid + 256 * S)
        return out

    @staticmethod
    def patch2_candidates(d):
        """Tile ids of csrc/patch2.hip (3x3 / stride 1 / pad 1 with the input patch of a pixel tile in LDS, filters streamed) for a
        descriptor it takes: Cin % 32 == 0, Cout >= 64 and % 4 == 0, no residual, 1 .. 3 dense segments with boundaries at multiples of 128
        channels, activation none / ReLU / LeakyReLU.  YOLACT_AMD_PATCH2=0 removes them (A/B switch)."""
        if os.environ.get('YOLACT_AMD_PATCH2', '1') != '1':
            return []
        if not ((d.kh, d.kw, d.stride, d.pad) == (3, 3, 1, 1) and d.Cin % 32 == 0 and d.Kpad == 9 * d.Cin and d.Cout >= 64 and d.Cout % 4 == 0
                and d.res_mode == L.RES_NONE and 1 <= d.nseg <= 3 and d.w_h2):
            return []
        cov = 0
        for i in range(d.nseg):
            g = d.seg[i]
            if (g.n0 != cov or g.n0 % 128 or g.act > L.ACT_LEAKY01 or g.act < 0 or g.row_stride % 4 or g.row_stride < g.n1 - g.n0
                    or g.batch_stride != d.Ho * d.Wo * g.row_stride):
                return []
            cov = g.n1
        if cov < d.Cout:
            return []
        return [t | L.TILE_H2 | L.TILE_DCNP for t in sorted(L.PATCH2_TILES)]

    @staticmethod
    def pc_candidates(d):
        """(tile + 256 * split_k) candidates of the producer / consumer kernel (csrc/pcconv.hip) for a descriptor the pipelined kernel
        takes (_pipe_ok) with more than 32 output channels: the block tile unsplit, and the chunk-aligned K splits that bring a
        short grid to 0.5 .. 4 blocks per CU.  OPT-IN (YOLACT_AMD_PC=1): measured in sessions r6b / r6c, never in front of the pipelined
        tiles (profiles/r06_pc_probe.txt) — both are bound by the same global -> LDS rates, not by how the waves share the work."""
        if os.environ.get('YOLACT_AMD_PC', '0') != '1' or d.Cout <= 32:
            return []
        M, nk = d.B * d.Ho * d.Wo, d.Kpad // 32
        out = []
        for t, name in sorted(L.PC_TILES.items()):
            bm, bn = (int(v) for v in os.environ.get('YOLACT_AMD_PC_SHAPE', name[2:]).split('x'))     # (YMI_PC_FLAGS 32 / 64 experiments: the
            if bn > 128 and d.Cout < 256:                                                              #  block behind the id is 32 / 64 x 128)
                continue
            tid = t | L.TILE_H2 | L.TILE_DCNP
            out.append(tid)
            blocks = -(-M // bm) * -(-d.Cout // bn)
            if blocks < 400 and d.Cout % 4 == 0:
                for S in (2, 3, 4, 5, 6, 8, 9, 12, 16):
                    per = -(-nk // S)
                    if per >= 4 and per * (S - 1) < nk and 128 <= blocks * S <= 1100:
                        out.append(tid + 256 * S)
        return out

    @staticmethod
    def ws_candidates(d):
        """(tile + 256 * split_k) candidates of the weight-stationary streaming kernel (csrc/wstat.hip) for a descriptor it takes
        (_pipe_ok, no residual, Cout <= 64): every block shape of the column width that covers Cout, with the K ranges that make a
        block's filters fit its 64 KB of LDS — the fewest ranges, and a few more where that leaves the grid short of the chip."""
        if d.res_mode != L.RES_NONE or d.Cout > 64:
            return []
        M, nk = d.B * d.Ho * d.Wo, d.Kpad // 32
        out = []
        for t, name in sorted(L.WS_TILES.items()):
            bm, bn = (int(v) for v in name[2:].split('w')[0].split('x'))
            if (bn == 32) != (d.Cout <= 32):
                continue
            cap = (64 * 1024) // (bn * 128)                # chunks of filters (bn columns x 32 k x 2 planes x 2 bytes) in 64 KB
            blocks, n = -(-M // bm), 0
            for S in (1, 2, 3, 4, 5, 6, 8, 9, 12, 16):
                per = -(-nk // S)
                if per > cap or (S > 1 and per * (S - 1) >= nk):
                    continue
                if n and blocks * S > 1200:                # beyond the fewest ranges: only while the grid is short of ~4 blocks per CU
                    break
                out.append((t | L.TILE_H2 | L.TILE_DCNP) + 256 * (S if S > 1 else 0))
                n += 1
                if n == 3:
                    break
        return out

    def direct_candidates(self, fn, d, wide=False):
        """(tile + 256 * split_k) values _tune_direct measures for one convolution descriptor (also tools/overlap_tune.py)."""
        is_dcn = fn is self.lib.ymi_dcn_v2_forward_f32
        h2_, split_ = self.h2 and not wide, self.split or (wide and not is_dcn)
        if is_dcn:                   # DCN gather loader: basic tiles; the bf16x3 arithmetic does not exist for it
            cands = [t for t in L.BASIC_TILES if t != L.TILE_128x32]
        elif d.Cin % 32 != 0:        # stem loader: basic tiles only
            cands = [L.TILE_128x64, L.TILE_64x64] if d.Cout <= 64 else list(L.BASIC_TILES)
        elif d.Cout <= 32:
            cands = [L.TILE_128x32, L.TILE_64x64, L.TILE_64x64_S3, L.TILE_32x32_K4, L.TILE_32x32_K4_S4,
                     L.TILE_64x32_K2, L.TILE_64x32_K2_S3]
        elif d.Cout <= 64:
            cands = [L.TILE_128x64, L.TILE_128x64_S3, L.TILE_64x64, L.TILE_64x64_S3, L.TILE_64x64_S4,
                     L.TILE_32x32_K4, L.TILE_32x32_K4_S4, L.TILE_64x32_K2, L.TILE_64x32_K2_S3,
                     L.TILE_32x64_K2, L.TILE_32x64_K2_S3]
        else:
            cands = [t for t in sorted(L.TILE_NAMES) if t != L.TILE_128x32 and not (t & (L.TILE_X3 | L.TILE_H2))]
            if d.Cout < 256:
                cands = [t for t in cands if t != L.TILE_128x256_W8]
        spflag = (L.TILE_X3 if split_ else L.TILE_H2 if h2_ else 0) if not (is_dcn and split_) else 0
        if spflag:      # (the Cin = 4 stem loader has the basic tiles only)
            base_ok = L.H2_BASE_TILES if spflag == L.TILE_H2 else L.X3_BASE_TILES
            cands = cands + [t | spflag for t in cands if t in base_ok
                             and (d.Cin % 32 == 0 or t in L.BASIC_TILES)]
        if is_dcn and h2_:           # the pipelined gather-GEMM of csrc/dcn.hip (fp16x2 plans)
            cands = cands + self.dcnp_candidates(d, dcn=True)
        if not is_dcn and h2_ and self.pipe and self._pipe_ok(d):   # the same pipelined kernel as an ordinary convolution
            cands = cands + self.dcnp_candidates(d) + self.ws_candidates(d)     # (+ the streaming kernel for narrow outputs)
            cands = cands + self.pc_candidates(d)                               # (+ round 6: producer / consumer blocks, opt-in)
        if not is_dcn and h2_ and self.pipe:
            cands = cands + self.patch2_candidates(d)                           # (+ round 6: 3x3 with the input patch in LDS)
            if ((d.kh, d.kw, d.stride, d.pad, d.Cin, d.Cout) == (3, 3, 1, 1, 64, 64) and d.res_mode == L.RES_NONE
                    and patch_tile_allowed()):
                cands = cands + [L.DCNP_PATCH_C64 | L.TILE_H2 | L.TILE_DCNP]    # csrc/patch.hip: the input patch in LDS, filters in registers
        if not is_dcn and self._splitk_ok(d) and self.splitk:
            # split-K candidates: big tiles whose grid alone cannot fill the chip, K cut 2 / 4 ways
            x3 = spflag
            for S in (2, 4):
                if (d.Kpad // 32) % S == 0 and d.Kpad // S >= 128:
                    cands += [(t | x3) + 256 * S for t in (L.TILE_128x128, L.TILE_64x128, L.TILE_128x64, L.TILE_64x64,
                                                          L.TILE_256x128_W8)]

        return cands

    def _tune_direct(self, e0, e1, s, reps, disk, measure):
        cache = {}
        for opi, (fn, dptr, name, where) in enumerate(self.ops):
            is_dcn = fn is self.lib.ymi_dcn_v2_forward_f32
            if fn is not self.lib.ymi_conv2d_nhwc_f32 and not is_dcn:
                continue
            d = dptr.contents.conv if is_dcn else dptr.contents
            key = (d.B, d.H, d.W, d.Cin, d.Cout, d.kh, d.kw, d.stride, d.pad, d.res_mode, d.nseg, d.Kpad) + (('dcn',) if is_dcn else ())
            wide = opi in self.wide_ops            # outlier-channel guard: bf16x3 tiles (DCN: exact-fp32 tiles) for this layer only
            h2_, split_ = self.h2 and not wide, self.split or (wide and not is_dcn)
            skey = str(key) + ('|x3' if split_ else '|h2' if h2_ else '')
            key = key + (('wide',) if wide else ())
            if key not in cache and skey in disk:
                v_ = int(disk[skey])
                patch_off = ((v_ & 255) == (L.DCNP_PATCH_C64 | L.TILE_H2 | L.TILE_DCNP) and not patch_tile_allowed())
                # (YOLACT_AMD_PATCH=0: the A/B switch of csrc/patch.hip — a table entry that names it counts as a miss)
                if not patch_off and self._apply_choice(fn, dptr, where, disk[skey], s) == 0:   # a stale / foreign entry must not make every forward raise
                    cache[key] = v_
            if key not in cache:
                self.tune_misses += 1
                if not measure:
                    cache[key] = L.TILE_AUTO
                    d.tile, d.split_k = L.TILE_AUTO, 0
                    continue
                cands = self.direct_candidates(fn, d, wide)
                def cname(v):
                    return L.TILE_NAMES[v & 255] + ('/k%d' % (v >> 8) if v >> 8 else '')
                best, best_ms, times = None, 1e30, {}
                ok_cands = []
                for t in cands:                       # warm every candidate once (code fetch, clocks); skip the
                    if self._apply_choice(fn, dptr, where, t, s) == 0:        # ones this layer cannot use
                        ok_cands.append(t)
                # the launch's own magnitude-bound slot is zeroed before every timed launch, as at the start of a real run:
                # a slot that already holds the maximum hides the atomics of the first residency round
                yslot = self.amax[(d.y_amax - self.amax.data_ptr()) // 4:][:AMAX_SLOT_FLOATS] if d.y_amax else None
                for rnd in range(2):                  # two interleaved rounds, keep each tile's best time: a single
                    for t in ok_cands:                # noisy sample used to flip near-ties and move bench by +-3 %
                        self._apply_choice(fn, dptr, where, t, s)
                        e0.record()
                        for _ in range(reps):
                            if yslot is not None:
                                yslot.zero_()
                            fn(dptr, s)
                        e1.record()
                        e1.synchronize()
                        ms = e0.elapsed_time(e1) / reps
                        times[cname(t)] = round(min(ms, times.get(cname(t), 1e30)), 4)
                for t in ok_cands:
                    if times[cname(t)] < best_ms:
                        best, best_ms = t, times[cname(t)]
                assert best is not None, name
                cache[key] = best
                disk[skey] = best
                self.tune_table.append((name, cname(best), times))
            self._apply_choice(fn, dptr, where, cache[key], s)

    def _tune_winograd(self, e0, e1, s, reps, disk, measure):
        """Per eligible layer: best GEMM tile of the Winograd path, then Winograd vs the (already tuned) direct kernel.
        The winner replaces the op in the list."""
        lib = self.lib
        self.wino_toggle = {}       # table key -> [(op index, direct op, Winograd op, isolated timing favours Winograd)]: tools/instep_tune.py
        wtiles = [L.TILE_64x64, L.TILE_64x128, L.TILE_128x64, L.TILE_128x128_W8, L.TILE_64x128_S3, L.TILE_32x64_K2,
                  L.TILE_128x128, L.TILE_128x128_S3, L.TILE_128x128_W8_S3, L.TILE_256x128_W8, L.TILE_256x128_W8_S3]
        if self.split:
            wtiles = wtiles + [t | L.TILE_X3 for t in wtiles]
        if self.h2:     # fp16x2 GEMM tiles, V either fp32 (split on the fly) or written as fp16 planes by the input transform
            hb = wtiles + [L.TILE_128x256_W8]
            wtiles = wtiles + [t | L.TILE_H2 for t in hb] + [t | L.TILE_H2 | L.WINO_PLANES for t in hb]
            if os.environ.get('YOLACT_AMD_WGEMM', '1') == '1':       # round 6: the persistent producer / consumer grouped GEMM (planes only)
                wtiles = wtiles + [L.TILE_WG_128x256 | L.TILE_H2 | L.WINO_PLANES]
        memo = {}
        for idx, alts in sorted(self.wino_alt.items()):
            fn, dptr, name, where = self.ops[idx]
            if fn is not lib.ymi_conv2d_nhwc_f32:
                continue
            w0 = alts[0]
            key = 'wino' + str((w0.B, w0.H, w0.W, w0.C, w0.Cout, w0.act, w0.nseg, tuple(a.m for a in alts))) + (
                self.mode_key)
            if key not in memo and key in disk:
                ent = tuple(disk[key])
                ok = True
                if ent[1]:                                 # validate the stored (m, tile) with one launch
                    wd = [a for a in alts if a.m == ent[0]]
                    if wd:
                        wd[0].tile, wd[0].v_planes = int(ent[1]) & 255, 1 if int(ent[1]) & L.WINO_PLANES else 0
                        ok = lib.ymi_conv3x3_winograd_f32(C.byref(wd[0]), s) == 0
                    else:
                        ok = False
                if ok:
                    memo[key] = ent
            if key not in memo:
                self.tune_misses += 1
                if not measure:
                    memo[key] = (0, 0, 0.0, 0.0, 0.0, 0.0)
                else:
                    def timed(f, arg):
                        f(arg, s)
                        best = 1e30
                        for _ in range(2):
                            e0.record()
                            for _ in range(reps):
                                f(arg, s)
                            e1.record()
                            e1.synchronize()
                            best = min(best, e0.elapsed_time(e1) / reps)
                        return best
                    t_direct = timed(fn, dptr)
                    best_m, best_t, best_ms, per_m = 0, 0, 1e30, {}
                    for wd in alts:
                        for t in wtiles:
                            wd.tile, wd.v_planes = t & 255, 1 if t & L.WINO_PLANES else 0
                            if lib.ymi_conv3x3_winograd_f32(C.byref(wd), s) != 0:
                                continue
                            ms = timed(lib.ymi_conv3x3_winograd_f32, C.pointer(wd))
                            per_m[wd.m] = min(ms, per_m.get(wd.m, 1e30))
                            if ms < best_ms:
                                best_m, best_t, best_ms = wd.m, t, ms
                    memo[key] = (best_m, best_t, round(t_direct, 4), round(best_ms, 4),
                                 round(per_m.get(2, 0.0), 4), round(per_m.get(4, 0.0), 4))
                    disk[key] = list(memo[key])
            best_m, best_t, t_direct, t_wino, t_f2, t_f4 = memo[key]
            self.wino_table.append((name, 'F%d/%s%s' % (best_m, L.TILE_NAMES.get(int(best_t) & 255, '-'),
                                                         'p' if int(best_t) & L.WINO_PLANES else ''), t_direct, t_wino, t_f2, t_f4))
            # 'instep|<key>' = 0 (tools/instep_tune.py, round 5): the isolated timings above favour Winograd, but measured INSIDE the step
            # — three launches and their V / M tensors crossing the memory side between them, against one launch — the direct kernel
            # is the faster choice for this shape: keep it.  (Decided on whole-step time with every layer of the shape toggled.)
            instep_direct = disk.get('instep|' + key) == 0 and not self.wino_force
            if best_t and not (self.wino_up.get(idx) is not None or (self.wino_proj is not None and self.wino_proj[0] == idx)):
                wd_ = [a for a in alts if a.m == best_m][0]
                wd_.tile, wd_.v_planes = int(best_t) & 255, 1 if int(best_t) & L.WINO_PLANES else 0
                self.wino_toggle.setdefault(key, []).append(
                    (idx, (fn, dptr, name, where), (lib.ymi_conv3x3_winograd_f32, C.pointer(wd_), name + '[wino]', where),
                     bool(t_wino < 0.97 * t_direct)))
            if best_t and (self.wino_force or (t_wino < 0.97 * t_direct and not instep_direct)):
                wd = [a for a in alts if a.m == best_m][0]
                wd.tile, wd.v_planes = int(best_t) & 255, 1 if int(best_t) & L.WINO_PLANES else 0
                self.ops[idx] = (lib.ymi_conv3x3_winograd_f32, C.pointer(wd), name + '[wino]', where)
                up = self.wino_up.get(idx)
                if up is not None and best_m == 4 and wd.H % 2 == 0 and wd.W % 2 == 0:
                    bidx, lo_ptr, relu = up        # the upsampled tensor is never materialised: the input transform interpolates
                    wd.x_up, wd.up_relu = lo_ptr, relu
                    bfn, bargs, bname, bwhere = self.ops[bidx]
                    self.ops[bidx] = ('nop', None, bname + '[fused into ' + name + ']', bwhere)
                if self.wino_proj is not None and self.wino_proj[0] == idx and best_m == 4 and wd.Cout == 256 and wd.nseg == 0:
                    pidx = self.wino_proj[1]
                    pfn, pdptr, pname, pwhere = self.ops[pidx]
                    if pfn is lib.ymi_conv2d_nhwc_f32 and pwhere == where:
                        pd = pdptr.contents          # the 1x1's own descriptor: filters, epilogue, output, bound slot
                        wd.proj_w_h2, wd.proj_scale_h2, wd.proj_bias = pd.w_h2, pd.scale_h2, pd.bias
                        wd.proj_y, wd.proj_y_amax = pd.seg[0].ptr, pd.y_amax
                        wd.proj_cout, wd.proj_ldy, wd.proj_act = pd.Cout, pd.seg[0].row_stride, pd.seg[0].act
                        # proj_y is patched per call (NULL here): validate the fused descriptor with ONE launch into a scratch
                        # prototype tensor; if the library rejects it the two separate launches stay (round-4 advisor: every other
                        # fusion is validated before it is installed — an unvalidated one would make every forward raise)
                        scratch = torch.empty(self.proto_shape, dtype=torch.float32, device=self.device)
                        wd.proj_y = scratch.data_ptr()
                        rc = lib.ymi_conv3x3_winograd_f32(C.byref(wd), s)
                        torch.cuda.synchronize(self.device)
                        wd.proj_y = None
                        if rc != 0:
                            wd.proj_w_h2 = None
                            self.wino_proj_rejected = rc
                        else:
                            self.ops[pidx] = ('nop', None, pname + '[fused into ' + name + ']', pwhere)
                            self.proto_patch_wino = wd

    def set_winograd(self, key, on: bool):
        """Switch every layer of one Winograd table key between its direct launch and its three Winograd launches (tools/instep_tune.py:
        whole-step A/B of the choice the isolated timings made).  Only layers without a fused upsampling / projection are listed."""
        for idx, direct_op, wino_op, _ in self.wino_toggle.get(key, []):
            self.ops[idx] = wino_op if on else direct_op

    def conv_flops(self):
        return sum(self.lib.ymi_conv_flops(C.byref(d)) for _, d in self.conv_meta)


class _Borrowed(T):
    """A stage output that is still needed later (FPN input): the consuming block must not free it."""
    __slots__ = ()

    def __init__(self, t: T):
        super().__init__(t.buf, t.B, t.H, t.W, t.C, t.slot, t.gain)
