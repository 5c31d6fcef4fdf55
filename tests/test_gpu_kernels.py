"""GPU parity of the individual HIP kernels (through the C ABI) against plain torch fp32 CPU references."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from yolact_amd import _lib as L  # noqa: E402


def _g(seed):
    return torch.Generator().manual_seed(seed)


TILES = sorted(L.TILE_NAMES)


@pytest.mark.parametrize('tile', TILES)
@pytest.mark.parametrize('shape', [
    # B, Cin, H, W, Cout, k, stride, pad
    (2, 64, 19, 23, 96, 1, 1, 0),
    (1, 32, 17, 13, 40, 3, 1, 1),
    (2, 64, 21, 18, 130, 3, 2, 1),
    (1, 128, 9, 9, 64, 1, 2, 0),
    (3, 96, 5, 7, 351, 3, 1, 1),
])
def test_conv_plain(shape, tile):
    from gpu_utils import run_conv, rel_err
    B, Cin, H, W, Cout, k, s, p = shape
    if (tile & L.TILE_DCNP) and Cout % 4:
        pytest.skip('the pipelined kernel (csrc/dcn.hip) stores float4 rows: Cout % 4 == 0 (an explicit request is refused, see '
                    'test_dcn_pipelined_rejects_what_it_cannot_run)')
    if not (tile & L.TILE_DCNP) and (tile & 31) == L.TILE_WG_128x256:
        pytest.skip('csrc/wgemm.hip is the grouped GEMM of the Winograd path only (ymi_conv3x3_winograd_f32; refused by the conv engine: '
                    'test_winograd_persistent_grouped_gemm)')
    if (tile & L.TILE_DCNP) and (tile & 31) in L.PATCH2_TILES and (k, s, p, Cout >= 64) != (3, 1, 1, True):
        pytest.skip('csrc/patch2.hip takes 3x3 / s1 / p1 with Cout >= 64 only (refused with YMI_EARG: test_patch2_kernel_output_segments)')
    if (tile & L.TILE_DCNP) and (tile & 31) == L.DCNP_PATCH_C64:
        pytest.skip('csrc/patch.hip takes exactly one shape (3x3 / s1 / p1, 64 -> 64): tests/test_gpu_round5.py::test_patch_kernel_matches_torch; '
                    'anything else is refused with YMI_EARG (asserted there)')
    if tile & L.TILE_DCNP:
        base = tile & 31
        cols = 32 if L.DCNP_128x32_W4 <= base <= L.DCNP_64x32_W2 else int(L.WS_TILES[base].split('x')[1].split('w')[0]) if base in L.WS_TILES else None
        if cols is not None and (Cout > cols or (base in L.WS_TILES and k * k * Cin // 32 * cols * 128 > 65536)):
            pytest.skip('32- / 64-column tiles: narrower than this layer, or (weight-stationary) its filters exceed 64 KB unsplit — refused '
                        'with YMI_EARG (test_weight_stationary_kernel_rejects_what_it_cannot_run)')
    g = _g(B * 1000 + Cin + Cout + k)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    b = torch.randn(Cout, generator=g)
    y = run_conv(x, w, b, None, s, p, tile=tile)
    ref = F.conv2d(x, w, b, s, p)
    assert y.shape == ref.shape
    assert rel_err(y, ref) < 2e-5
    if tile & L.TILE_X3:                     # the same tile with both operands split on the fly (no filter planes)
        assert rel_err(run_conv(x, w, b, None, s, p, tile=tile, planes=False), ref) < 2e-5


def test_conv_asymmetric_filter_detects_transposes():
    """A = identity-like checks miss swapped roles; use a one-hot filter tap so every (ky,kx,c)->n route is unique."""
    from gpu_utils import run_conv
    Cin, Cout = 32, 64
    x = torch.randn(1, Cin, 6, 5, generator=_g(5))
    w = torch.zeros(Cout, Cin, 3, 3)
    for n in range(Cout):
        w[n, (n * 7) % Cin, (n // 3) % 3, n % 3] = 1.0 + n
    y = run_conv(x, w, None, None, 1, 1)
    assert torch.equal(y, F.conv2d(x, w, None, 1, 1))   # one product per output: exact


def test_conv_bn_relu_residual():
    from gpu_utils import run_conv, rel_err
    import torch.nn as nn
    g = _g(7)
    x = torch.randn(2, 64, 14, 14, generator=g)
    w = torch.randn(256, 64, 1, 1, generator=g) / 8
    bn = nn.BatchNorm2d(256).eval()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(256, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(256, generator=g) * 0.1)
        bn.running_mean.copy_(torch.randn(256, generator=g) * 0.1)
        bn.running_var.copy_(torch.rand(256, generator=g) + 0.5)
    res = torch.randn(2, 256, 14, 14, generator=g)
    y = run_conv(x, w, None, bn, 1, 0, act=L.ACT_RELU, res=res, res_mode=L.RES_ADD)
    with torch.no_grad():
        ref = F.relu(bn(F.conv2d(x, w)) + res)
    assert rel_err(y, ref) < 2e-5
    # darknet form: act(conv) + res
    y2 = run_conv(x, w, None, bn, 1, 0, act=L.ACT_LEAKY01, res=res, res_mode=L.RES_ADD, res_after_act=1)
    with torch.no_grad():
        ref2 = F.leaky_relu(bn(F.conv2d(x, w)), 0.1) + res
    assert rel_err(y2, ref2) < 2e-5


@pytest.mark.parametrize('sizes', [((18, 18), (35, 35)), ((35, 35), (69, 69)), ((5, 7), (9, 13))])
def test_conv_fpn_bilinear_residual(sizes):
    """lat conv + F.interpolate(prev, size) fused in the epilogue (yolact.py:331-335)."""
    from gpu_utils import run_conv, rel_err
    (hs, ws), (h, w_) = sizes
    g = _g(hs * 100 + h)
    x = torch.randn(2, 64, h, w_, generator=g)
    w = torch.randn(256, 64, 1, 1, generator=g) / 8
    b = torch.randn(256, generator=g)
    prev = torch.randn(2, 256, hs, ws, generator=g)
    y = run_conv(x, w, b, None, 1, 0, res=prev, res_mode=L.RES_BILINEAR)
    ref = F.interpolate(prev, size=(h, w_), mode='bilinear', align_corners=False) + F.conv2d(x, w, b)
    assert rel_err(y, ref) < 2e-5


def test_conv_stem_7x7_c4_loader():
    from gpu_utils import run_conv, rel_err
    g = _g(11)
    x = torch.randn(2, 3, 61, 47, generator=g)
    w = torch.randn(64, 3, 7, 7, generator=g) / 12
    y = run_conv(x, w, None, None, 2, 3, act=L.ACT_RELU, cin_pad=4)
    ref = F.relu(F.conv2d(x, w, None, 2, 3))
    assert rel_err(y, ref) < 2e-5
    for tile in (L.TILE_64x64, L.TILE_128x64):                 # the stem on the bf16x3 tiles, with and without filter planes
        for planes in (True, False):
            yx = run_conv(x, w, None, None, 2, 3, act=L.ACT_RELU, cin_pad=4, tile=tile | L.TILE_X3, planes=planes)
            assert rel_err(yx, ref) < 2e-5
        yh = run_conv(x, w, None, None, 2, 3, act=L.ACT_RELU, cin_pad=4, tile=tile | L.TILE_H2)      # ... and on fp16x2
        assert rel_err(yh, ref) < 2e-5
    # darknet pre-conv: 3x3 / s1 / p1 on 3 channels
    w3 = torch.randn(32, 3, 3, 3, generator=g) / 5
    y3 = run_conv(x, w3, None, None, 1, 1, cin_pad=4)
    assert rel_err(y3, F.conv2d(x, w3, None, 1, 1)) < 2e-5


def test_conv_head_segments_and_tanh():
    """One GEMM scattering to loc / conf / coef tensors with per-level offsets (yolact.py:169-173,633-634)."""
    from gpu_utils import nhwc, DEV
    from yolact_amd.engine import Packed
    g = _g(13)
    B, Cin, H, W, A, Ccls, D = 2, 64, 6, 5, 3, 81, 32
    x = torch.randn(B, Cin, H, W, generator=g)
    wb, wc, wm = (torch.randn(A * k, Cin, 3, 3, generator=g) / 24 for k in (4, Ccls, D))
    bb, bc, bm = (torch.randn(A * k, generator=g) for k in (4, Ccls, D))
    pk = Packed(torch.cat([wb, wc, wm]), torch.cat([bb, bc, bm]), None, 1, 1, None, DEV)
    P, off = H * W * A + 17, 17
    loc = torch.zeros(B, P, 4, device=DEV)
    conf = torch.zeros(B, P, Ccls, device=DEV)
    coef = torch.zeros(B, P, D, device=DEV)
    xd = nhwc(x).to(DEV)
    d = L.ConvDesc()
    d.x, d.w, d.bias = xd.data_ptr(), pk.w.data_ptr(), pk.bias.data_ptr()
    d.B, d.H, d.W, d.Cin, d.ldx, d.Ho, d.Wo, d.Cout = B, H, W, Cin, Cin, H, W, pk.Cout
    d.kh, d.kw, d.stride, d.pad, d.Kpad = 3, 3, 1, 1, pk.Kpad
    d.nseg = 3
    n_b, n_c, n_m = A * 4, A * Ccls, A * D
    d.seg[0] = L.ConvSeg(0, n_b, L.ACT_NONE, n_b, P * 4, loc.data_ptr() + off * 4 * 4)
    d.seg[1] = L.ConvSeg(n_b, n_b + n_c, L.ACT_NONE, n_c, P * Ccls, conf.data_ptr() + off * Ccls * 4)
    d.seg[2] = L.ConvSeg(n_b + n_c, n_b + n_c + n_m, L.ACT_TANH, n_m, P * D, coef.data_ptr() + off * D * 4)
    L.check(L.lib().ymi_conv2d_nhwc_f32(C.byref(d), L.stream_ptr()), 'conv')
    torch.cuda.synchronize()
    rl = F.conv2d(x, wb, bb, 1, 1).permute(0, 2, 3, 1).reshape(B, -1, 4)
    rc = F.conv2d(x, wc, bc, 1, 1).permute(0, 2, 3, 1).reshape(B, -1, Ccls)
    rm = torch.tanh(F.conv2d(x, wm, bm, 1, 1).permute(0, 2, 3, 1).reshape(B, -1, D))
    assert torch.allclose(loc.cpu()[:, off:], rl, atol=2e-5)
    assert torch.allclose(conf.cpu()[:, off:], rc, atol=2e-5)
    assert torch.allclose(coef.cpu()[:, off:], rm, atol=2e-5)
    assert loc.cpu()[:, :off].abs().max() == 0      # nothing written before the level offset


def test_conv_rejects_bad_arguments():
    d = L.ConvDesc()
    assert L.lib().ymi_conv2d_nhwc_f32(C.byref(d), None) == -3        # null pointers
    assert L.lib().ymi_conv2d_nhwc_f32(None, None) == -3
    with pytest.raises(RuntimeError, match='null'):
        L.check(-3, 'x')


def test_layout_pool_resize():
    from gpu_utils import DEV, nhwc, nchw
    lib, s = L.lib(), L.stream_ptr()
    g = _g(17)
    x = torch.randn(2, 3, 37, 29, generator=g)
    xd = x.to(DEV)
    y = torch.empty(2, 37, 29, 4, device=DEV)
    L.check(lib.ymi_nchw_to_nhwc4_f32(xd.data_ptr(), y.data_ptr(), 2, 3, 37, 29, s))
    assert torch.equal(y.cpu()[..., :3], nhwc(x)) and y.cpu()[..., 3].abs().max() == 0
    # the fused variant: same layout change + the magnitude bound of the input in its slot (16 sub-slots, 64 floats apart)
    y2 = torch.empty(2, 37, 29, 4, device=DEV)
    slot = torch.zeros(1024, device=DEV)
    L.check(lib.ymi_nchw_to_nhwc4_amax_f32(xd.data_ptr(), y2.data_ptr(), 2, 3, 37, 29, slot.data_ptr(), s))
    assert torch.equal(y2, y) and slot.max().item() == x.abs().max().item()
    assert slot.view(16, 64)[:, 1:].abs().max().item() == 0
    a = torch.randn(2, 40, 21, 19, generator=g)
    ad = nhwc(a).to(DEV)
    back = torch.empty(2, 40, 21, 19, device=DEV)
    L.check(lib.ymi_nhwc_to_nchw_f32(ad.data_ptr(), back.data_ptr(), 2, 40, 21, 19, s))
    assert torch.equal(back.cpu(), a)
    # maxpool 3x3/2/1 with -inf padding (all-negative input exercises the padding value)
    m = -torch.rand(2, 64, 23, 31, generator=g) - 1
    md = nhwc(m).to(DEV)
    Ho, Wo = (23 + 2 - 3) // 2 + 1, (31 + 2 - 3) // 2 + 1
    mo = torch.empty(2, Ho, Wo, 64, device=DEV)
    L.check(lib.ymi_maxpool3x3s2_nhwc_f32(md.data_ptr(), mo.data_ptr(), 2, 23, 31, 64, Ho, Wo, s))
    assert torch.equal(nchw(mo.cpu()), F.max_pool2d(m, 3, 2, 1))
    # bilinear x2 (scale_factor form) + relu, and to-size form
    u = torch.randn(2, 32, 13, 11, generator=g)
    ud = nhwc(u).to(DEV)
    uo = torch.empty(2, 26, 22, 32, device=DEV)
    L.check(lib.ymi_bilinear_nhwc_f32(ud.data_ptr(), uo.data_ptr(), 2, 13, 11, 32, 26, 22, C.c_float(0.5), C.c_float(0.5), 1, s))
    ref = F.relu(F.interpolate(u, scale_factor=2, mode='bilinear', align_corners=False))
    assert torch.allclose(nchw(uo.cpu()), ref, atol=1e-5)
    uo2 = torch.empty(2, 30, 17, 32, device=DEV)
    L.check(lib.ymi_bilinear_nhwc_f32(ud.data_ptr(), uo2.data_ptr(), 2, 13, 11, 32, 30, 17, C.c_float(0), C.c_float(0), 0, s))
    ref2 = F.interpolate(u, size=(30, 17), mode='bilinear', align_corners=False)
    assert torch.allclose(nchw(uo2.cpu()), ref2, atol=1e-5)   # fp32 coordinate math, no FMA contraction


# ---------------------------------------------------------------------------------------------------
def _dcn_inputs(seed, B=2, Cc=32, H=9, W=8, Co=48, stride=1):
    g = _g(seed)
    x = torch.randn(B, Cc, H, W, generator=g)
    w = torch.randn(Co, Cc, 3, 3, generator=g) / 17
    b = torch.randn(Co, generator=g)
    Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
    om = torch.randn(B, 27, Ho, Wo, generator=g) * 1.5
    return x, w, b, om


@pytest.mark.parametrize('tile', [L.TILE_AUTO, L.TILE_64x64 | L.TILE_H2, L.TILE_128x64 | L.TILE_H2, L.TILE_128x128 | L.TILE_H2,
                                  L.TILE_64x128 | L.TILE_H2])
@pytest.mark.parametrize('stride', [1, 2])
def test_dcn_matches_oracle(stride, tile):
    """The gather-fused DCN GEMM on the exact-fp32 tiles and on the fp16x2 tiles (the gathered, modulated fp32 tile in LDS is split
    like any other activation tile; its magnitude is bounded by the input's: convex bilinear weights x a sigmoid)."""
    from gpu_utils import run_conv, rel_err
    from oracle.yolact_oracle import dcn_v2_forward
    x, w, b, om = _dcn_inputs(23 + stride, stride=stride)
    y = run_conv(x, w, b, None, stride, 1, dcn_offmask=om, tile=tile)
    ref = dcn_v2_forward(x, om[:, :18], torch.sigmoid(om[:, 18:]), w, b, stride, 1, 1)
    assert rel_err(y, ref) < 2e-5


PIPE_ALL = [t | L.TILE_H2 | L.TILE_DCNP for t in sorted(L.DCNP_TILES)]                 # every block tile of csrc/dcn.hip
DCNP_ALL = [t for t in PIPE_ALL if (t & 31) not in L.DCNP_PLAIN_ONLY]                     # ... that the DCN gather can use
PC_ALL = [t | L.TILE_H2 | L.TILE_DCNP for t in sorted(L.PC_TILES)]                     # producer / consumer blocks of csrc/pcconv.hip (round 6)


@pytest.mark.parametrize('tile', DCNP_ALL)
@pytest.mark.parametrize('case', [(2, 32, 9, 8, 48, 1), (2, 32, 9, 8, 48, 2), (1, 64, 23, 19, 132, 1), (3, 128, 13, 11, 128, 2),
                                  (1, 96, 37, 41, 256, 1)])
def test_dcn_pipelined_matches_oracle(case, tile):
    """csrc/dcn.hip (the software-pipelined gather-GEMM: corner loads three chunks ahead in a register ring, samples stored to LDS as
    fp16x2 planes) against the CPU oracle: odd chunk counts (Cin = 32, 96), ragged row / column tiles, stride 2, batch > 1."""
    from gpu_utils import run_conv, rel_err
    from oracle.yolact_oracle import dcn_v2_forward
    B, Cc, H, W, Co, stride = case
    x, w, b, om = _dcn_inputs(41 + Cc + stride, B, Cc, H, W, Co, stride)
    om[:, :18] *= 2.0                       # |offsets| of several pixels: sample points leave the image
    y = run_conv(x, w, b, None, stride, 1, dcn_offmask=om, tile=tile, act=L.ACT_RELU)
    ref = torch.relu(dcn_v2_forward(x, om[:, :18], torch.sigmoid(om[:, 18:]), w, b, stride, 1, 1))
    assert rel_err(y, ref) < 2e-5
    assert abs(run_conv.last_amax[1] - ref.abs().max().item()) <= 2e-5 * ref.abs().max().item()     # the bound it reports for y


@pytest.mark.parametrize('tile', [L.TILE_AUTO, L.TILE_64x64 | L.TILE_H2] + DCNP_ALL)
def test_dcn_tap_interleaved_offmask_is_the_same_launch(tile):
    """ymi_dcn_desc.om_layout = 1 ([dh_k, dw_k, mask_k] per tap, padded to 32 channels — what engine.pack_offmask makes
    conv_offset_mask write) against layout 0 (the reference's 18 offsets | 9 masks, dcn_v2.py:118-122) on the same values: the same
    samples in the same order, so bit-identical — on the register-staged loader (exact fp32, fp16x2) and on every pipelined tile."""
    from gpu_utils import run_conv
    x, w, b, om = _dcn_inputs(77, 2, 64, 13, 11, 132, 1)
    om[:, :18] *= 2.0
    order = [c for k in range(9) for c in (2 * k, 2 * k + 1, 18 + k)]
    om1 = torch.cat([om[:, order], torch.full((2, 5, 13, 11), float('nan'))], 1)         # the 5 padding channels are never read
    y0 = run_conv(x, w, b, None, 1, 1, dcn_offmask=om, tile=tile, act=L.ACT_RELU)
    y1 = run_conv(x, w, b, None, 1, 1, dcn_offmask=om1, tile=tile, act=L.ACT_RELU, om_layout=1)
    assert torch.equal(y0, y1)
    with pytest.raises(RuntimeError):
        run_conv(x, w, b, None, 1, 1, dcn_offmask=om1, tile=tile, om_layout=2)


@pytest.mark.parametrize('tile', [L.DCNP_64x128_W8, L.DCNP_96x128_W6, 11, 12, 13])
@pytest.mark.parametrize('split', [2, 3, 4, 5, 9])
def test_dcn_pipelined_split_k(tile, split):
    """ymi_conv_desc.split_k on the pipelined DCN tiles: chunk-aligned K ranges (4 ranges of 9 chunks start INSIDE a tap when a tap is
    4 chunks), partial sums through split_ws, deterministic second pass with scale / bias / ReLU and the magnitude bound."""
    from gpu_utils import run_conv, rel_err
    from oracle.yolact_oracle import dcn_v2_forward
    x, w, b, om = _dcn_inputs(61 + split, 2, 128, 13, 11, 260, 1)
    om[:, :18] *= 2.0
    y = run_conv(x, w, b, None, 1, 1, dcn_offmask=om, tile=tile | L.TILE_H2 | L.TILE_DCNP, act=L.ACT_RELU, split_k=split)
    ref = torch.relu(dcn_v2_forward(x, om[:, :18], torch.sigmoid(om[:, 18:]), w, b, 1, 1, 1))
    assert rel_err(y, ref) < 2e-5
    assert abs(run_conv.last_amax[1] - ref.abs().max().item()) <= 2e-5 * ref.abs().max().item()
    y2 = run_conv(x, w, b, None, 1, 1, dcn_offmask=om, tile=tile | L.TILE_H2 | L.TILE_DCNP, act=L.ACT_RELU, split_k=split)
    assert torch.equal(y, y2)                                               # fixed summation order: bit-reproducible


@pytest.mark.parametrize('tile', PIPE_ALL + PC_ALL)
@pytest.mark.parametrize('case', [(2, 64, 19, 17, 72, 3, 1, False, L.ACT_RELU), (1, 128, 23, 21, 260, 3, 2, False, L.ACT_NONE),
                                  (2, 256, 14, 15, 64, 1, 1, True, L.ACT_RELU), (3, 64, 9, 10, 128, 1, 2, False, L.ACT_LEAKY01),
                                  (1, 32, 31, 29, 36, 3, 1, False, L.ACT_RELU), (2, 96, 12, 13, 512, 1, 1, True, L.ACT_LEAKY01),
                                  (2, 64, 19, 17, 32, 3, 1, False, L.ACT_NONE), (1, 128, 23, 21, 28, 3, 2, False, L.ACT_RELU),
                                  (3, 256, 14, 15, 32, 1, 1, True, L.ACT_RELU)])
def test_pipelined_kernel_as_ordinary_convolution(case, tile):
    """ymi_conv2d_nhwc_f32 with a YMI_TILE_DCNP tile = the pipelined kernel of csrc/dcn.hip in PLAIN mode (one load per sample, integer
    tap geometry): 3x3 / pad 1 and 1x1 / pad 0, stride 1 / 2, folded BN, bias, residual before or after the activation, ragged row
    and column tiles, odd chunk counts — against torch fp32."""
    from gpu_utils import run_conv, rel_err
    B, Cin, H, W, Cout, k, stride, has_res, act = case
    if L.DCNP_128x32_W4 <= (tile & 31) <= L.DCNP_64x32_W2 and Cout > 32:
        pytest.skip('32-column tiles take Cout <= 32 only (rejected with YMI_EARG: test_dcn_pipelined_rejects_what_it_cannot_run)')
    g = _g(300 + Cin + Cout + k)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    b = torch.randn(Cout, generator=g) * 0.1
    pad = 1 if k == 3 else 0
    ref = F.conv2d(x, w, b, stride, pad)
    res = torch.randn(ref.shape, generator=g) if has_res else None
    after = 1 if act == L.ACT_LEAKY01 else 0
    y = run_conv(x, w, b, None, stride, pad, act=act, res=res, res_mode=L.RES_ADD if has_res else L.RES_NONE, res_after_act=after, tile=tile)

    def a_(t):
        return torch.relu(t) if act == L.ACT_RELU else F.leaky_relu(t, 0.1) if act == L.ACT_LEAKY01 else t
    if has_res:
        ref = a_(ref) + res if after else a_(ref + res)
    else:
        ref = a_(ref)
    assert rel_err(y, ref) < 2e-5
    assert abs(run_conv.last_amax[1] - ref.abs().max().item()) <= 2e-5 * ref.abs().max().item()


@pytest.mark.parametrize('tile', [L.DCNP_64x128_W8, L.DCNP_128x128_W8_R1, L.DCNP_96x256_W12, L.DCNP_128x256_W16, L.DCNP_32x128, 14, 15, 16, 29])
@pytest.mark.parametrize('split', [2, 3, 4, 8])
def test_pipelined_ordinary_convolution_split_k(tile, split):
    """K ranges on the PLAIN path (1x1, K = 512 -> 16 chunks; 3x3 with a residual): partial sums + the deterministic second pass."""
    from gpu_utils import run_conv, rel_err
    g = _g(400 + split)
    x = torch.randn(2, 512, 9, 11, generator=g)
    w = torch.randn(260, 512, 1, 1, generator=g) / 512 ** 0.5
    b = torch.randn(260, generator=g) * 0.1
    res = torch.randn(2, 260, 9, 11, generator=g)
    t = tile | L.TILE_H2 | L.TILE_DCNP
    y = run_conv(x, w, b, None, 1, 0, act=L.ACT_RELU, res=res, res_mode=L.RES_ADD, tile=t, split_k=split)
    ref = torch.relu(F.conv2d(x, w, b) + res)
    assert rel_err(y, ref) < 2e-5
    assert torch.equal(y, run_conv(x, w, b, None, 1, 0, act=L.ACT_RELU, res=res, res_mode=L.RES_ADD, tile=t, split_k=split))
    x3 = torch.randn(1, 64, 13, 12, generator=g)
    w3 = torch.randn(68, 64, 3, 3, generator=g) / 24
    per = -(-18 // split)                                       # 9 * 64 / 32 = 18 chunks: ranges of `per`, the last one non-empty
    if per >= 2 and per * (split - 1) < 18:
        y3 = run_conv(x3, w3, None, None, 2, 1, tile=t, split_k=split)
        assert rel_err(y3, F.conv2d(x3, w3, None, 2, 1)) < 2e-5


@pytest.mark.parametrize('tile', PC_ALL)
@pytest.mark.parametrize('case', [(8, 1024, 35, 35, 256, 1, 1, False), (2, 128, 69, 69, 128, 3, 1, False), (8, 256, 35, 35, 1024, 1, 1, True),
                                  (2, 512, 35, 35, 256, 3, 2, False), (1, 64, 7, 5, 68, 3, 1, True), (1, 96, 3, 3, 36, 1, 1, False)])
def test_producer_consumer_kernel_is_bit_identical_to_the_pipelined_tile(case, tile):
    """csrc/pcconv.hip against the pipelined kernel of csrc/dcn.hip on the same descriptor: same products, same order per accumulator,
    same epilogue — the outputs and the recorded magnitude bound are equal bit for bit (backbone-sized and ragged shapes, residual,
    stride 2, odd chunk counts); torch fp32 as the outside reference."""
    from gpu_utils import run_conv, rel_err
    B, Cin, H, W, Cout, k, stride, has_res = case
    g = _g(900 + Cin + Cout + k)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    b = torch.randn(Cout, generator=g) * 0.1
    pad = 1 if k == 3 else 0
    ref = F.conv2d(x, w, b, stride, pad)
    res = torch.randn(ref.shape, generator=g) if has_res else None
    kw = dict(act=L.ACT_RELU, res=res, res_mode=L.RES_ADD if has_res else L.RES_NONE)
    y = run_conv(x, w, b, None, stride, pad, tile=tile, **kw)
    am = run_conv.last_amax[1]
    y0 = run_conv(x, w, b, None, stride, pad, tile=L.DCNP_128x128_W8_R1 | L.TILE_H2 | L.TILE_DCNP, **kw)
    assert torch.equal(y, y0) and am == run_conv.last_amax[1]
    assert rel_err(y, torch.relu(ref + res) if has_res else torch.relu(ref)) < 2e-5


@pytest.mark.parametrize('tile', [L.DCNP_128x32_W4, L.DCNP_256x32_W8, L.DCNP_64x32_W2])
@pytest.mark.parametrize('split', [1, 2, 4, 9])
def test_offset_mask_convolution_on_the_32_column_tiles(tile, split):
    """The shape the 32-column tiles exist for: a 3x3 / pad 1 convolution to 27 channels zero-padded to 32 (engine.pack_offmask), K
    ranges included (35x35 / 18x18 maps have few row tiles) — against torch fp32, the padding channels exactly zero."""
    from gpu_utils import run_conv, rel_err
    g = _g(500 + split)
    x = torch.randn(2, 128, 18, 19, generator=g)
    w = torch.zeros(32, 128, 3, 3)
    w[:27] = torch.randn(27, 128, 3, 3, generator=g) / 34
    b = torch.zeros(32)
    b[:27] = torch.randn(27, generator=g)
    t = tile | L.TILE_H2 | L.TILE_DCNP
    y = run_conv(x, w, b, None, 1, 1, tile=t, split_k=split if split > 1 else 0)
    assert rel_err(y, F.conv2d(x, w, b, 1, 1)) < 2e-5
    assert y[:, 27:].abs().max().item() == 0
    assert torch.equal(y, run_conv(x, w, b, None, 1, 1, tile=t, split_k=split if split > 1 else 0))


@pytest.mark.parametrize('case', [(2, 26, 22, 32, L.ACT_RELU, L.ACT_RELU, True), (1, 17, 23, 28, L.ACT_RELU, L.ACT_NONE, False),
                                  (3, 8, 12, 4, L.ACT_NONE, L.ACT_LEAKY01, True)])
@pytest.mark.parametrize('tile', [L.TILE_128x128 | L.TILE_H2, L.TILE_64x64])
def test_winograd_output_transform_fused_with_the_consuming_1x1(case, tile):
    """ymi_wino_desc.proj_*: conv3x3(256 -> 256) + act -> conv1x1(256 -> n <= 32) + act2 with the 3x3's output never written
    (protonet's last two layers, utils/functions.py:163-213) against torch fp32 and against the two separate launches: ragged edge
    tiles (H, W not multiples of 4), batch > 1, n < 32, with / without a bias on the 1x1; the bound reported for the projected tensor."""
    from gpu_utils import run_wino, run_conv, rel_err
    B, H, W, n, act, act2, pbias = case
    g = _g(800 + H + n)
    x = torch.randn(B, 64, H, W, generator=g)
    w = torch.randn(256, 64, 3, 3, generator=g) / 24
    b = torch.randn(256, generator=g) * 0.2
    pw = torch.randn(n, 256, 1, 1, generator=g) / 16
    pb = torch.randn(n, generator=g) * 0.1 if pbias else None

    def a_(t, a):
        return torch.relu(t) if a == L.ACT_RELU else F.leaky_relu(t, 0.1) if a == L.ACT_LEAKY01 else t
    mid = a_(F.conv2d(x, w, b, 1, 1), act)
    ref = a_(F.conv2d(mid, pw, pb), act2)
    y = run_wino(x, w, b, None, act=act, tile=tile, m=4, proj=(pw, pb, act2))
    assert rel_err(y, ref) < 3e-5
    assert abs(run_wino.last_amax[1] - ref.abs().max().item()) <= 3e-5 * ref.abs().max().item()
    two = run_conv(run_wino(x, w, b, None, act=act, tile=tile, m=4), pw, pb, None, 1, 0, act=act2, tile=L.TILE_64x64 | L.TILE_H2)
    assert rel_err(y, two) < 3e-5
    assert torch.equal(y, run_wino(x, w, b, None, act=act, tile=tile, m=4, proj=(pw, pb, act2)))


def test_winograd_fused_projection_rejects_what_it_cannot_run():
    from gpu_utils import run_wino
    g = _g(3)
    x = torch.randn(1, 32, 8, 8, generator=g)
    pw = torch.randn(32, 256, 1, 1, generator=g)
    with pytest.raises(RuntimeError):                                     # F(2x2): the fused kernel is the F(4x4) output transform
        run_wino(x, torch.randn(256, 32, 3, 3, generator=g), None, None, m=2, proj=(pw, None, L.ACT_NONE))
    with pytest.raises(RuntimeError):                                     # 128 dense channels, not 256
        run_wino(x, torch.randn(128, 32, 3, 3, generator=g), None, None, m=4, proj=(pw[:, :128], None, L.ACT_NONE))
    with pytest.raises(RuntimeError):                                     # 48 projected channels
        run_wino(x, torch.randn(256, 32, 3, 3, generator=g), None, None, m=4, proj=(torch.randn(48, 256, 1, 1, generator=g), None, L.ACT_NONE))


@pytest.mark.parametrize('M', [16, 1000, 4097, 16 * 256 * 3 + 5])
@pytest.mark.parametrize('mode', ['chain', 'chain_nores_leaky', 'tail_only'])
def test_pointwise_chain_matches_fp64(M, mode):
    """ymi_pointwise_chain_f32 (csrc/chain.hip): y = act(W_a x + b_a + res) [64 -> 256], z = act(W_b y + b_b) [256 -> 64] in one
    streaming launch, y taken from LDS for the second GEMM — against fp64 on the CPU: fewer tiles than blocks, ragged last tile,
    several tiles per block; with / without the residual and the second layer; the magnitude bounds of y and z."""
    from gpu_utils import run_chain
    g = _g(900 + M)
    x = torch.randn(M, 64, generator=g) * 3
    wa = torch.randn(256, 64, generator=g) / 8
    ba = torch.randn(256, generator=g) * 0.3
    res = torch.randn(M, 256, generator=g) * 2 if mode != 'chain_nores_leaky' else None
    wb = torch.randn(64, 256, generator=g) / 16 if mode != 'tail_only' else None
    bb = torch.randn(64, generator=g) * 0.1 if wb is not None else None
    act = L.ACT_LEAKY01 if mode == 'chain_nores_leaky' else L.ACT_RELU

    def a_(t):
        return torch.relu(t) if act == L.ACT_RELU else F.leaky_relu(t, 0.1)
    yr = x.double() @ wa.double().t() + ba.double()
    yr = a_(yr + res.double() if res is not None else yr)
    y, z = run_chain(x, wa, ba, res, wb, bb, act, act)
    assert ((y.double() - yr).abs().max() / yr.abs().max()).item() < 5e-7
    assert abs(run_chain.last_amax[0] - yr.abs().max().item()) <= 1e-6 * yr.abs().max().item()
    if wb is not None:
        zr = a_(yr @ wb.double().t() + bb.double())
        assert ((z.double() - zr).abs().max() / zr.abs().max()).item() < 1e-6
        assert abs(run_chain.last_amax[1] - zr.abs().max().item()) <= 2e-6 * zr.abs().max().item()
        y2, z2 = run_chain(x, wa, ba, res, wb, bb, act, act)
        assert torch.equal(y, y2) and torch.equal(z, z2)
    else:
        assert z is None


PATCH2_ALL = [t | L.TILE_H2 | L.TILE_DCNP for t in sorted(L.PATCH2_TILES)]          # csrc/patch2.hip: 256- / 192-pixel tiles


@pytest.mark.parametrize('tile', PATCH2_ALL)
@pytest.mark.parametrize('case', [(2, 256, 69, 69, 256, L.ACT_RELU, None), (1, 128, 35, 35, 128, L.ACT_NONE, None), (8, 64, 18, 18, 72, L.ACT_LEAKY01, None),
                                  (1, 32, 5, 7, 64, L.ACT_RELU, None), (1, 96, 138, 138, 260, L.ACT_RELU, None), (3, 64, 23, 40, 132, L.ACT_NONE, '5x31'),
                                  (1, 64, 33, 30, 128, L.ACT_RELU, '23x8'), (2, 160, 17, 19, 384, L.ACT_RELU, '4x4')])
def test_patch2_kernel_matches_torch(case, tile, monkeypatch):
    """csrc/patch2.hip (3x3 / s1 / p1, the input patch of a pixel tile in LDS one 32-channel chunk at a time, filters streamed through a
    ring, producer / consumer waves) against torch fp32: the maps of the headline plan and ragged ones, channel counts that are no
    multiple of the block's 128, one chunk .. eight, every activation, host-picked and forced tile shapes (tiles wider than the map,
    tiny tiles, tiles that do not fill the block's pixel slots); the magnitude bound; bit-reproducible."""
    from gpu_utils import run_conv, rel_err
    B, Cin, H, W, Cout, act, shape = case
    if shape:
        th, tw = (int(v) for v in shape.split('x'))
        if th * tw > (192 if (tile & 31) == L.DCNP_PATCH2_192 else 256):
            pytest.skip('forced tile shape larger than this block')
        monkeypatch.setenv('YMI_PATCH2_TILE', shape)
    else:
        monkeypatch.delenv('YMI_PATCH2_TILE', raising=False)
    g = _g(1200 + Cin + Cout + H)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    b = torch.randn(Cout, generator=g) * 0.1
    ref = F.conv2d(x, w, b, 1, 1)
    ref = torch.relu(ref) if act == L.ACT_RELU else F.leaky_relu(ref, 0.1) if act == L.ACT_LEAKY01 else ref
    y = run_conv(x, w, b, None, 1, 1, act=act, tile=tile)
    assert rel_err(y, ref) < 2e-5
    assert abs(run_conv.last_amax[1] - ref.abs().max().item()) <= 2e-5 * ref.abs().max().item()
    assert torch.equal(y, run_conv(x, w, b, None, 1, 1, act=act, tile=tile))


@pytest.mark.parametrize('tile', PATCH2_ALL)
def test_patch2_kernel_output_segments(tile):
    """Two and three dense output segments with boundaries at multiples of 128 channels (head0.upfeature + proto_net[0] in one launch:
    256 | 256), each with its own tensor, activation and magnitude-bound slot; a boundary off the 128 grid is refused."""
    from gpu_utils import run_conv, rel_err
    g = _g(1300)
    x = torch.randn(2, 64, 21, 26, generator=g)
    w = torch.randn(448, 64, 3, 3, generator=g) / 24
    b = torch.randn(448, generator=g) * 0.1
    ref = F.conv2d(x, w, b, 1, 1)
    ys = run_conv(x, w, b, None, 1, 1, tile=tile, seg_bounds=[256], seg_acts=[L.ACT_RELU, L.ACT_NONE, 0])
    assert rel_err(ys[0], torch.relu(ref[:, :256])) < 2e-5 and rel_err(ys[1], ref[:, 256:]) < 2e-5
    am = run_conv.last_amax
    assert abs(am[1] - torch.relu(ref[:, :256]).max().item()) <= 2e-5 * am[1] and abs(am[2] - ref[:, 256:].abs().max().item()) <= 2e-5 * am[2]
    ys = run_conv(x, w, b, None, 1, 1, tile=tile, seg_bounds=[128, 384], seg_acts=[L.ACT_NONE, L.ACT_RELU, L.ACT_LEAKY01])
    assert rel_err(ys[0], ref[:, :128]) < 2e-5 and rel_err(ys[1], torch.relu(ref[:, 128:384])) < 2e-5
    assert rel_err(ys[2], F.leaky_relu(ref[:, 384:], 0.1)) < 2e-5
    with pytest.raises(RuntimeError):
        run_conv(x, w, b, None, 1, 1, tile=tile, seg_bounds=[200])
    with pytest.raises(RuntimeError):                                     # stride 2 / 1x1 / residual: not this kernel's
        run_conv(x, w, b, None, 2, 1, tile=tile)


@pytest.mark.parametrize('P', [128, 256])
@pytest.mark.parametrize('M,mode', [(64, 'chain'), (37, 'chain'), (9800, 'chain'), (1225, 'chain_nores_leaky'), (200, 'chain_big_res'),
                                    (130, 'chain_outlier_row')])
def test_pointwise_chain2_matches_fp64(M, mode, P):
    """csrc/chain2.hip through ymi_pointwise_chain_f32 (P = 128 / 256 planes: y = act(W_a x + b_a + res) [P -> 4P] written once, each
    128-channel slice fed from LDS into z = act(W_b y + b_b) [4P -> P]) against fp64: one block, a ragged block, the 35 x 35 x 8 map,
    no residual + LeakyReLU, a residual 2^6 larger than the conv term and a row 2^8 larger than the rest (both stretch the y slices
    below their shared bound-derived scale); the magnitude bounds of y and z; bit-reproducible."""
    from gpu_utils import run_chain
    g = _g(950 + M + P)
    x = torch.randn(M, P, generator=g).abs() * 3
    if mode == 'chain_outlier_row':
        x[7] *= 256.0
    wa = torch.randn(4 * P, P, generator=g) / P ** 0.5
    ba = torch.randn(4 * P, generator=g) * 0.3
    res = None if mode == 'chain_nores_leaky' else torch.randn(M, 4 * P, generator=g) * (128.0 if mode == 'chain_big_res' else 2.0)
    wb = torch.randn(P, 4 * P, generator=g) / (4 * P) ** 0.5
    bb = torch.randn(P, generator=g) * 0.1
    act = L.ACT_LEAKY01 if mode == 'chain_nores_leaky' else L.ACT_RELU

    def a_(t):
        return torch.relu(t) if act == L.ACT_RELU else F.leaky_relu(t, 0.1)
    yr = x.double() @ wa.double().t() + ba.double()
    yr = a_(yr + res.double() if res is not None else yr)
    zr = a_(yr @ wb.double().t() + bb.double())
    y, z = run_chain(x, wa, ba, res, wb, bb, act, act)
    assert ((y.double() - yr).abs().max() / yr.abs().max()).item() < 1e-6
    assert abs(run_chain.last_amax[0] - yr.abs().max().item()) <= 2e-6 * yr.abs().max().item()
    # per ROW for z: an outlier row must not hide the error of the ordinary ones behind its own scale
    rel = (z.double() - zr).abs().amax(dim=1) / zr.abs().amax(dim=1).clamp_min(1e-30)
    assert rel.max().item() < 2e-6, rel.max().item()
    assert abs(run_chain.last_amax[1] - zr.abs().max().item()) <= 3e-6 * zr.abs().max().item()
    y2, z2 = run_chain(x, wa, ba, res, wb, bb, act, act)
    assert torch.equal(y, y2) and torch.equal(z, z2)


WS_ALL = [t | L.TILE_H2 | L.TILE_DCNP for t in sorted(L.WS_TILES)]                     # every block shape of csrc/wstat.hip


def _ws_splits(tile, nk):
    """K-range counts a weight-stationary tile accepts for nk chunks: its filters must fit 64 KB (16 chunks at 32 columns, 8 at 64)."""
    cap = 16 if L.WS_TILES[tile & 31].split('x')[1].startswith('32') else 8
    return [S for S in (1, 2, 3, 4, 5, 6, 8, 9, 12, 16) if -(-nk // S) <= cap and (S == 1 or -(-nk // S) * (S - 1) < nk)]


@pytest.mark.parametrize('tile', WS_ALL)
@pytest.mark.parametrize('case', [(2, 64, 19, 17, 32, 3, 1, L.ACT_RELU), (1, 128, 23, 21, 28, 3, 2, L.ACT_NONE),
                                  (3, 256, 14, 15, 32, 1, 1, L.ACT_LEAKY01), (2, 64, 31, 29, 64, 1, 1, L.ACT_RELU),
                                  (1, 32, 9, 10, 36, 3, 1, L.ACT_NONE), (2, 96, 12, 13, 4, 1, 2, L.ACT_RELU),
                                  (1, 160, 40, 37, 32, 3, 1, L.ACT_NONE)])
def test_weight_stationary_streaming_kernel(case, tile):
    """csrc/wstat.hip (filters of the block's K range resident in LDS, activations streamed from global memory straight into MFMA
    operand registers, no barrier in the loop) against torch fp32: 3x3 / pad 1 and 1x1, stride 1 / 2, ragged row tiles and waves
    past the last row, odd chunk counts, K ranges that start inside a tap, Cout < the tile's columns; the fewest K ranges the
    tile accepts and the most; bit-reproducible; the magnitude bound it reports."""
    from gpu_utils import run_conv, rel_err
    B, Cin, H, W, Cout, k, stride, act = case
    if Cout > int(L.WS_TILES[tile & 31].split('x')[1].split('w')[0]):
        pytest.skip('more output channels than the tile has columns (rejected: test_weight_stationary_kernel_rejects...)')
    g = _g(700 + Cin + Cout + k)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    b = torch.randn(Cout, generator=g) * 0.1
    pad = 1 if k == 3 else 0
    ref = F.conv2d(x, w, b, stride, pad)
    ref = torch.relu(ref) if act == L.ACT_RELU else F.leaky_relu(ref, 0.1) if act == L.ACT_LEAKY01 else ref
    splits = _ws_splits(tile, k * k * Cin // 32)
    for S in sorted({splits[0], splits[-1]}):
        y = run_conv(x, w, b, None, stride, pad, act=act, tile=tile, split_k=S if S > 1 else 0)
        assert rel_err(y, ref) < 2e-5, S
        assert abs(run_conv.last_amax[1] - ref.abs().max().item()) <= 2e-5 * ref.abs().max().item()
        assert torch.equal(y, run_conv(x, w, b, None, stride, pad, act=act, tile=tile, split_k=S if S > 1 else 0))


def test_weight_stationary_kernel_rejects_what_it_cannot_run():
    from gpu_utils import run_conv
    g = _g(9)
    x = torch.randn(1, 64, 9, 9, generator=g)
    t32, t64 = (L.TILE_H2 | L.TILE_DCNP | v for v in (L.DCNP_WS_128x32_W4, L.DCNP_WS_512x64_W8))
    with pytest.raises(RuntimeError):                                     # 48 output channels on a 32-column tile
        run_conv(x, torch.randn(48, 64, 1, 1, generator=g), None, None, 1, 0, tile=t32)
    with pytest.raises(RuntimeError):                                     # 18 chunks of 64 columns do not fit 64 KB unsplit
        run_conv(x, torch.randn(64, 64, 3, 3, generator=g), None, None, 1, 1, tile=t64)
    with pytest.raises(RuntimeError):                                     # residual: not this kernel's epilogue
        run_conv(x, torch.randn(32, 64, 1, 1, generator=g), None, None, 1, 0, tile=t32, res=torch.randn(1, 32, 9, 9, generator=g),
                 res_mode=L.RES_ADD)
    with pytest.raises(RuntimeError):                                     # Cout % 4 != 0
        run_conv(x, torch.randn(30, 64, 1, 1, generator=g), None, None, 1, 0, tile=t32)


def test_dcn_pipelined_rejects_what_it_cannot_run():
    """An explicit YMI_TILE_DCNP request outside the kernel's envelope is an error code, never a silent other kernel."""
    from gpu_utils import run_conv
    x, w, b, om = _dcn_inputs(5, Co=50)                                  # Cout % 4 != 0
    with pytest.raises(RuntimeError):
        run_conv(x, w, b, None, 1, 1, dcn_offmask=om, tile=L.TILE_H2 | L.TILE_DCNP | L.DCNP_64x128)
    x, w, b, om = _dcn_inputs(5)
    with pytest.raises(RuntimeError):                                     # no fp16x2 flag
        run_conv(x, w, b, None, 1, 1, dcn_offmask=om, tile=L.TILE_DCNP | L.DCNP_64x128)
    with pytest.raises(RuntimeError):                                     # unknown block tile
        run_conv(x, w, b, None, 1, 1, dcn_offmask=om, tile=L.TILE_H2 | L.TILE_DCNP | 31)
    with pytest.raises(RuntimeError):                                     # a 32-column tile for 48 output channels
        run_conv(x, w, b, None, 1, 1, tile=L.TILE_H2 | L.TILE_DCNP | L.DCNP_128x32_W4)


def test_dcn_v2_module_reference_kat_through_the_shim():
    """The reference's own known-answer test (external/DCNv2/test.py:32-67 check_zero_offset), restated line by line against the
    module the reference imports (`from dcn_v2 import dcn_v2_conv, DCNv2, DCN` resolves to shim/dcn_v2.py): zero offset conv,
    mask = sigmoid(0), identity centre-tap weights => 2 * dcn_v2(input, offset, mask) == input.  Then DCN.forward(input) and
    dcn_v2_conv with random offsets against the CPU oracle."""
    import importlib
    import os
    import sys
    import torch.nn as nn
    from oracle.yolact_oracle import dcn_v2_forward
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'shim'))
    try:
        dcn_v2 = importlib.import_module('dcn_v2')
    finally:
        sys.path.pop(0)
    assert dcn_v2.__file__.replace(os.sep, '/').endswith('shim/dcn_v2.py')
    N, inC, inH, inW, outC, kH, kW, dg = 2, 2, 4, 4, 2, 3, 3, 1             # test.py:14-18
    dev = 'cuda:0'
    conv_offset = nn.Conv2d(inC, dg * 2 * kH * kW, (kH, kW), (1, 1), (1, 1), bias=True).to(dev)
    conv_mask = nn.Conv2d(inC, dg * kH * kW, (kH, kW), (1, 1), (1, 1), bias=True).to(dev)
    m = dcn_v2.DCNv2(inC, outC, (kH, kW), stride=1, padding=1, dilation=1, deformable_groups=dg).to(dev)
    for c in (conv_offset, conv_mask):
        c.weight.data.zero_(); c.bias.data.zero_()
    m.weight.data.zero_(); m.bias.data.zero_()
    for q in range(outC):                                                     # conv_identify, test.py:21-30
        m.weight.data[q, q, kH // 2, kW // 2] = 1.0
    inp = torch.randn(N, inC, inH, inW, generator=_g(77)).to(dev)
    with torch.no_grad():
        offset, mask = conv_offset(inp), torch.sigmoid(conv_mask(inp))
    out = m(inp, offset, mask) * 2
    assert (inp - out).abs().max().item() < 1e-6                              # (the reference asserts 1e-10 on exact 0.5 products)
    # DCN.forward: conv_offset_mask + sigmoid + gather-GEMM, non-trivial offsets, both kernel families (Cout % 4 == 0 -> pipelined)
    for Cin, Cout, stride in ((48, 64, 1), (32, 18, 2)):
        g = _g(100 + Cin)
        d = dcn_v2.DCN(Cin, Cout, 3, stride=stride, padding=1).to(dev)
        with torch.no_grad():
            d.conv_offset_mask.weight.copy_(torch.randn(27, Cin, 3, 3, generator=g) * 0.05)
            d.conv_offset_mask.bias.copy_(torch.randn(27, generator=g) * 0.5)
            d.bias.copy_(torch.randn(Cout, generator=g) * 0.1)
        x = torch.randn(2, Cin, 17, 13, generator=g)
        y = d(x.to(dev)).cpu()
        om = F.conv2d(x, d.conv_offset_mask.weight.cpu(), d.conv_offset_mask.bias.cpu(), stride, 1)
        ref = dcn_v2_forward(x, om[:, :18], torch.sigmoid(om[:, 18:]), d.weight.detach().cpu(), d.bias.detach().cpu(), stride, 1, 1)
        assert y.shape == ref.shape and (y - ref).abs().max().item() < 5e-5 * max(1.0, ref.abs().max().item())
        y2 = dcn_v2.dcn_v2_conv(x.to(dev), om[:, :18].contiguous().to(dev), torch.sigmoid(om[:, 18:]).to(dev), d.weight, d.bias,
                                stride, 1, 1, 1).cpu()
        assert (y2 - ref).abs().max().item() < 5e-5 * max(1.0, ref.abs().max().item())
    with pytest.raises(NotImplementedError):
        dcn_v2.DCN(8, 8, 5, stride=1, padding=2)


def test_dcn_known_answers():
    """external/DCNv2/test.py:32-67: zero offsets, mask logit 0 (sigmoid = 0.5), identity 3x3 centre weights
    => 2*DCN(x) == x; and offset 0 with mask -> 1 reduces DCN to F.conv2d."""
    from gpu_utils import run_conv
    Cc = 32
    x = torch.randn(2, Cc, 10, 7, generator=_g(31))
    w = torch.zeros(Cc, Cc, 3, 3)
    for i in range(Cc):
        w[i, i, 1, 1] = 1.0
    om = torch.zeros(2, 27, 10, 7)
    y = run_conv(x, w, torch.zeros(Cc), None, 1, 1, dcn_offmask=om)
    assert (2 * y - x).abs().max().item() < 1e-6
    w2 = torch.randn(40, Cc, 3, 3, generator=_g(32)) / 17
    om2 = torch.zeros(2, 27, 10, 7)
    om2[:, 18:] = 40.0      # sigmoid(40) == 1 in fp32
    y2 = run_conv(x, w2, None, None, 1, 1, dcn_offmask=om2)
    assert torch.allclose(y2, F.conv2d(x, w2, None, 1, 1), atol=2e-5)


@pytest.mark.parametrize('shape', [(2, 64, 9, 11, 96), (1, 32, 69, 69, 64), (3, 128, 6, 5, 132), (1, 256, 18, 18, 256)])
@pytest.mark.parametrize('tile', [L.TILE_AUTO, L.TILE_64x64, L.TILE_64x128, L.TILE_128x128_W8, L.TILE_32x64_K2,
                                  L.TILE_64x64 | L.TILE_X3, L.TILE_128x128 | L.TILE_X3, L.TILE_64x128 | L.TILE_X3,
                                  L.TILE_128x128_S3 | L.TILE_X3, L.TILE_256x128_W8 | L.TILE_X3])
@pytest.mark.parametrize('m', [2, 4])
def test_winograd_matches_direct(shape, tile, m):
    """Winograd F(2x2,3x3) / F(4x4,3x3) paths (csrc/winograd.hip) vs torch's conv and vs the direct implicit-GEMM kernel:
    sizes that are not multiples of the tile (partial last tile row / column), BN fold + ReLU epilogue, every GEMM tile
    the plan may pick.  Tolerance: F(2x2) only adds a few fp32 additions (2e-5 of max|ref| like the direct kernel's
    test); F(4x4)'s transform coefficients (up to 8) amplify fp32 rounding to ~1e-5 on white-noise inputs (CPU fp32
    emulation of the same algorithm: 0.5-1.4e-5), bar 5e-5."""
    from gpu_utils import run_conv, run_wino, rel_err
    import torch.nn as nn
    B, Cin, H, W, Cout = shape
    g = _g(B * 100 + Cin + Cout + H)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    bn = nn.BatchNorm2d(Cout).eval()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(Cout, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(Cout, generator=g) * 0.1)
        bn.running_mean.copy_(torch.randn(Cout, generator=g) * 0.1)
        bn.running_var.copy_(torch.rand(Cout, generator=g) + 0.5)
        ref = F.relu(bn(F.conv2d(x, w, None, 1, 1)))
    y = run_wino(x, w, None, bn, L.ACT_RELU, tile, m)
    assert y.shape == ref.shape
    assert rel_err(y, ref) < (2e-5 if m == 2 else 5e-5)
    direct = run_conv(x, w, None, bn, 1, 1, act=L.ACT_RELU)
    assert rel_err(y, direct) < (1e-5 if m == 2 else 5e-5)


@pytest.mark.parametrize('m', [2, 4])
def test_winograd_head_segments_and_tanh(m):
    """Segmented Winograd output transform: the shared prediction-head conv (Cout = A*(4+81+32) = 351, not a multiple
    of 4) scattering to loc / conf / coef with a level offset, tanh on the coefficients (yolact.py:169-193)."""
    from gpu_utils import nhwc, DEV
    from yolact_amd.engine import Packed, WinoPacked
    g = _g(29)
    B, Cin, H, W, A, Ccls, D = 2, 64, 7, 5, 3, 81, 32
    x = torch.randn(B, Cin, H, W, generator=g)
    wb, wc, wm = (torch.randn(A * k, Cin, 3, 3, generator=g) / 24 for k in (4, Ccls, D))
    bb, bc, bm = (torch.randn(A * k, generator=g) for k in (4, Ccls, D))
    wcat = torch.cat([wb, wc, wm])
    pk = Packed(wcat, torch.cat([bb, bc, bm]), None, 1, 1, None, DEV)
    wp = WinoPacked(wcat, DEV, m)
    P, off = H * W * A + 17, 17
    loc = torch.zeros(B, P, 4, device=DEV)
    conf = torch.zeros(B, P, Ccls, device=DEV)
    coef = torch.zeros(B, P, D, device=DEV)
    xd = nhwc(x).to(DEV)
    T = B * ((H + m - 1) // m) * ((W + m - 1) // m)
    Ng = (pk.Cout + 3) // 4 * 4
    V = torch.empty((m + 2) ** 2 * T * Cin, device=DEV)
    Mw = torch.empty((m + 2) ** 2 * T * Ng, device=DEV)
    d = L.WinoDesc()
    d.x, d.u, d.bias, d.V, d.M = xd.data_ptr(), wp.u.data_ptr(), pk.bias.data_ptr(), V.data_ptr(), Mw.data_ptr()
    d.B, d.H, d.W, d.C, d.Cout, d.m = B, H, W, Cin, pk.Cout, m
    d.nseg = 3
    n_b, n_c, n_m = A * 4, A * Ccls, A * D
    d.seg[0] = L.ConvSeg(0, n_b, L.ACT_NONE, n_b, P * 4, loc.data_ptr() + off * 4 * 4)
    d.seg[1] = L.ConvSeg(n_b, n_b + n_c, L.ACT_NONE, n_c, P * Ccls, conf.data_ptr() + off * Ccls * 4)
    d.seg[2] = L.ConvSeg(n_b + n_c, n_b + n_c + n_m, L.ACT_TANH, n_m, P * D, coef.data_ptr() + off * D * 4)
    L.check(L.lib().ymi_conv3x3_winograd_f32(C.byref(d), L.stream_ptr()), 'winograd')
    torch.cuda.synchronize()
    rl = F.conv2d(x, wb, bb, 1, 1).permute(0, 2, 3, 1).reshape(B, -1, 4)
    rc = F.conv2d(x, wc, bc, 1, 1).permute(0, 2, 3, 1).reshape(B, -1, Ccls)
    rm = torch.tanh(F.conv2d(x, wm, bm, 1, 1).permute(0, 2, 3, 1).reshape(B, -1, D))
    tol = 2e-5 if m == 2 else 1e-4          # outputs are O(4): F(4x4) rounding ~1e-5 relative (see test_winograd_matches_direct)
    assert torch.allclose(loc.cpu()[:, off:], rl, atol=tol)
    assert torch.allclose(conf.cpu()[:, off:], rc, atol=tol)
    assert torch.allclose(coef.cpu()[:, off:], rm, atol=tol)
    assert loc.cpu()[:, :off].abs().max() == 0 and conf.cpu()[:, :off].abs().max() == 0


def test_winograd_rejects_unsupported():
    d = L.WinoDesc()
    assert L.lib().ymi_conv3x3_winograd_f32(C.byref(d), L.stream_ptr()) != 0


@pytest.mark.parametrize('base', [3, 5, 1, 8, 16, 17, 19, 21])
def test_conv_bf16x3_is_fp32_class(base):
    """tile | TILE_X3: every operand split exactly into three bf16 pieces, 6 piece products on the bf16 matrix pipe.  The error
    against an fp64 reference must be of the exact-fp32 kernel's own class (both ~1e-7 of sum|a b|), over a K = 2304
    reduction with operands of mixed magnitude (incl. values far below / above the bf16-friendly range)."""
    from gpu_utils import run_conv
    g = _g(90 + base)
    B, Cin, H, W, Cout, k = 2, 256, 13, 11, 192, 3
    x = torch.randn(B, Cin, H, W, generator=g) * torch.exp(torch.randn(B, Cin, 1, 1, generator=g) * 2.0)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    x[0, 0, 0, 0], x[0, 1, 2, 3], w[0, 0, 0, 0], w[5, 7, 1, 1] = 1e-30, 3.0e4, -2.5e-22, 40.0
    ref = F.conv2d(x.double(), w.double(), None, 1, 1)
    mag = F.conv2d(x.double().abs(), w.double().abs(), None, 1, 1)
    y32 = run_conv(x, w, None, None, 1, 1, tile=base).double()
    e32 = ((y32 - ref).abs() / mag).max().item()
    for planes in (True, False):
        yx3 = run_conv(x, w, None, None, 1, 1, tile=base | L.TILE_X3, planes=planes).double()
        ex3 = ((yx3 - ref).abs() / mag).max().item()
        print('tile %s: fp32 MFMA err %.2e, bf16x3 (%s) err %.2e (of sum|ab|)' % (
            L.TILE_NAMES[base], e32, 'filter planes' if planes else 'both split on the fly', ex3))
        assert e32 < 1e-5 and ex3 < 1e-5      # K = 2304 products of mixed magnitude: both are fp32-rounding class
        assert ex3 < 2 * e32 + 1e-7           # measured: the split path is the MORE accurate one (2.0e-6 vs 4.0e-6, 64x64 tile)


@pytest.mark.parametrize('tile', [L.TILE_128x128, L.TILE_64x128 | L.TILE_X3, L.TILE_128x128 | L.TILE_X3, L.TILE_256x128_W8 | L.TILE_X3,
                                  L.TILE_64x64 | L.TILE_X3, L.TILE_64x128 | L.TILE_H2, L.TILE_128x128 | L.TILE_H2,
                                  L.TILE_256x128_W8 | L.TILE_H2, L.TILE_64x64 | L.TILE_H2])
@pytest.mark.parametrize('S', [2, 4])
def test_conv_1x1_split_k(tile, S):
    """split_k: the K reduction of a 1x1 convolution cut into S ranges (S x the blocks), partial sums added in a fixed
    order by the second pass together with folded BN + residual + ReLU / LeakyReLU (darknet order too); stride 2 as well."""
    from gpu_utils import run_conv, rel_err
    import torch.nn as nn
    g = _g(400 + S)
    for (B, Cin, H, W, Cout, stride) in ((2, 512, 13, 11, 256, 1), (1, 1024, 9, 9, 2048, 2), (3, 256, 7, 5, 132, 1)):
        x = torch.randn(B, Cin, H, W, generator=g)
        w = torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5
        bn = nn.BatchNorm2d(Cout).eval()
        with torch.no_grad():
            bn.weight.copy_(torch.rand(Cout, generator=g) + 0.5)
            bn.bias.copy_(torch.randn(Cout, generator=g) * 0.1)
            bn.running_mean.copy_(torch.randn(Cout, generator=g) * 0.1)
            bn.running_var.copy_(torch.rand(Cout, generator=g) + 0.5)
            conv = bn(F.conv2d(x, w, None, stride))
        res = torch.randn(conv.shape, generator=g)
        y = run_conv(x, w, None, bn, stride, 0, act=L.ACT_RELU, res=res, res_mode=L.RES_ADD, tile=tile, split_k=S)
        assert rel_err(y, F.relu(conv + res)) < 2e-5
        y2 = run_conv(x, w, None, bn, stride, 0, act=L.ACT_LEAKY01, res=res, res_mode=L.RES_ADD, res_after_act=1, tile=tile, split_k=S)
        assert rel_err(y2, F.leaky_relu(conv, 0.1) + res) < 2e-5
        y3 = run_conv(x, w, torch.randn(Cout, generator=_g(5)), None, stride, 0, tile=tile, split_k=S)
        assert rel_err(y3, F.conv2d(x, w, torch.randn(Cout, generator=_g(5)), stride)) < 2e-5
        # unsplit launch of the same tile: same sums up to the fp32 association of the S partials
        y0 = run_conv(x, w, None, bn, stride, 0, act=L.ACT_RELU, res=res, res_mode=L.RES_ADD, tile=tile)
        assert rel_err(y, y0) < 5e-6


def test_conv_split_k_rejects_unsupported():
    from gpu_utils import run_conv
    x = torch.randn(1, 64, 8, 8, generator=_g(1))
    w3 = torch.randn(64, 64, 3, 3, generator=_g(2)) / 24
    with pytest.raises(RuntimeError):
        run_conv(x, w3, None, None, 1, 1, split_k=2)                     # 3x3: not a pointwise layer
    w1 = torch.randn(64, 64, 1, 1, generator=_g(3)) / 8
    with pytest.raises(RuntimeError):
        run_conv(x, w1, None, None, 1, 0, split_k=4)                     # K = 64 = 2 chunks: not divisible into 4 ranges
    with pytest.raises(RuntimeError):
        run_conv(x, w1, None, None, 1, 0, act=L.ACT_TANH, split_k=2)     # epilogue outside the second pass's repertoire


@pytest.mark.parametrize('mode', ['direct', 'direct_x3', 'wino2', 'wino4', 'wino4_x3'])
def test_merged_two_output_conv(mode):
    """The merged head0.upfeature + proto_net[0] launch (engine.Plan: one 3x3 conv of P3 with the two filter banks
    concatenated along Cout, scattering to two dense NHWC tensors with ReLU): direct kernel (segment epilogue) and the
    segmented Winograd output transform (16-byte store path), against two separate torch convolutions."""
    from gpu_utils import nhwc, nchw, DEV
    from yolact_amd.engine import Packed, WinoPacked
    g = _g(77)
    B, Cin, H, W, C1, C2 = 2, 64, 11, 9, 64, 32
    x = torch.randn(B, Cin, H, W, generator=g)
    w1, w2 = torch.randn(C1, Cin, 3, 3, generator=g) / 24, torch.randn(C2, Cin, 3, 3, generator=g) * 40      # the second tensor is ~1000x larger
    b1, b2 = torch.randn(C1, generator=g), torch.randn(C2, generator=g)
    wcat, bcat = torch.cat([w1, w2]), torch.cat([b1, b2])
    amax = torch.zeros(3 * 1024, device=DEV)       # two consecutive magnitude-bound slots (+ a guard slot): one per segment (ABI 5)
    pk = Packed(wcat, bcat, None, 1, 1, None, DEV)
    xd = nhwc(x).to(DEV)
    y1 = torch.full((B, H, W, C1), float('nan'), device=DEV)
    y2 = torch.full((B, H, W, C2), float('nan'), device=DEV)
    segs = [L.ConvSeg(0, C1, L.ACT_RELU, C1, H * W * C1, y1.data_ptr()), L.ConvSeg(C1, C1 + C2, L.ACT_RELU, C2, H * W * C2, y2.data_ptr())]
    x3 = L.TILE_X3 if mode.endswith('x3') else 0
    if mode.startswith('direct'):
        d = L.ConvDesc()
        d.x, d.w, d.bias = xd.data_ptr(), pk.w.data_ptr(), pk.bias.data_ptr()
        d.B, d.H, d.W, d.Cin, d.ldx, d.Ho, d.Wo, d.Cout = B, H, W, Cin, Cin, H, W, C1 + C2
        d.kh, d.kw, d.stride, d.pad, d.Kpad = 3, 3, 1, 1, pk.Kpad
        d.nseg, d.tile = 2, L.TILE_64x64 | x3
        d.seg[0], d.seg[1] = segs
        d.y_amax = amax.data_ptr()
        if x3:
            d.w_x3 = pk.w3().data_ptr()
        L.check(L.lib().ymi_conv2d_nhwc_f32(C.byref(d), L.stream_ptr()), 'merged direct')
        tol = 2e-5
    else:
        m = 2 if mode == 'wino2' else 4
        wp = WinoPacked(wcat, DEV, m)
        T = B * ((H + m - 1) // m) * ((W + m - 1) // m)
        V = torch.empty((m + 2) ** 2 * T * Cin, device=DEV)
        Mw = torch.empty((m + 2) ** 2 * T * (C1 + C2), device=DEV)
        d = L.WinoDesc()
        d.x, d.u, d.bias, d.V, d.M = xd.data_ptr(), wp.u.data_ptr(), pk.bias.data_ptr(), V.data_ptr(), Mw.data_ptr()
        d.B, d.H, d.W, d.C, d.Cout, d.m, d.nseg = B, H, W, Cin, C1 + C2, m, 2
        d.tile = L.TILE_64x64 | x3
        d.seg[0], d.seg[1] = segs
        d.y_amax = amax.data_ptr()
        if x3:
            d.u_x3 = wp.u3().data_ptr()
        L.check(L.lib().ymi_conv3x3_winograd_f32(C.byref(d), L.stream_ptr()), 'merged winograd')
        tol = 2e-5 if m == 2 else 5e-5
    torch.cuda.synchronize()
    r1, r2 = F.relu(F.conv2d(x, w1, b1, 1, 1)), F.relu(F.conv2d(x, w2, b2, 1, 1))
    from gpu_utils import rel_err
    assert rel_err(nchw(y1.cpu()), r1) < tol and rel_err(nchw(y2.cpu()), r2) < tol
    # each segment raises ITS slot to max|y| of that tensor (a shared bound would report the larger one for both)
    bounds = amax.view(3, 1024).amax(1).cpu().tolist()
    assert bounds[0] == y1.abs().max().item() and bounds[1] == y2.abs().max().item() and bounds[2] == 0.0
    assert bounds[1] > 100 * bounds[0]


@pytest.mark.parametrize('base', [3, 5, 1, 8, 6, 16, 17, 19, 21])
def test_conv_fp16x2_is_fp32_class(base):
    """tile | TILE_H2: x * s = h + l, two fp16 pieces by round to nearest (s: a power of two per tensor from the magnitude bound
    the producer recorded, per filter row for the weights), 3 piece products on the fp16 matrix pipe, fp32 accumulate.  Same bar
    as the bf16x3 tiles: the error against an fp64 reference over a K = 2304 reduction with operands of mixed magnitude (values
    far below / far above the fp16 range before scaling: 1e-30, 3e4, a weight of 40 next to weights of 1e-2) must be of the
    exact-fp32 kernel's own class, and the launch must report max|y| exactly."""
    from gpu_utils import run_conv
    g = _g(90 + base)
    B, Cin, H, W, Cout, k = 2, 256, 13, 11, 192, 3
    x = torch.randn(B, Cin, H, W, generator=g) * torch.exp(torch.randn(B, Cin, 1, 1, generator=g) * 2.0)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    x[0, 0, 0, 0], x[0, 1, 2, 3], w[0, 0, 0, 0], w[5, 7, 1, 1] = 1e-30, 3.0e4, -2.5e-22, 40.0
    ref = F.conv2d(x.double(), w.double(), None, 1, 1)
    mag = F.conv2d(x.double().abs(), w.double().abs(), None, 1, 1)
    y32 = run_conv(x, w, None, None, 1, 1, tile=base).double()
    e32 = ((y32 - ref).abs() / mag).max().item()
    yh = run_conv(x, w, None, None, 1, 1, tile=base | L.TILE_H2)
    amax = run_conv.last_amax
    eh = ((yh.double() - ref).abs() / mag).max().item()
    print('tile %s: fp32 MFMA err %.2e, fp16x2 err %.2e (of sum|ab|)' % (L.TILE_NAMES[base], e32, eh))
    assert e32 < 1e-5 and eh < 1e-5
    assert eh < 2 * e32 + 1e-7
    assert amax[0] == x.abs().max().item() and amax[1] == yh.abs().max().item()


def test_conv_fp16x2_scale_follows_the_magnitude_bound():
    """The activation scale is derived on the device from x_amax: tensors far outside the fp16 range (1e6, 1e-9) go through
    unchanged in accuracy, an all-zero tensor is handled (bound 0 -> scale 1), and a bound that is LARGER than the true maximum
    (what a max-pool / interpolation consumer hands on) only costs headroom, not correctness."""
    from gpu_utils import run_conv, rel_err
    g = _g(321)
    x = torch.randn(2, 64, 9, 7, generator=g)
    w = torch.randn(96, 64, 3, 3, generator=g) / 24
    b = torch.randn(96, generator=g)
    ref = F.conv2d(x, w, b, 1, 1)
    for gain in (1e6, 1e-9, 1.0):
        y = run_conv(x * gain, w, b * gain, None, 1, 1, tile=L.TILE_64x64 | L.TILE_H2)
        assert rel_err(y / gain, ref) < 2e-5, gain
    y0 = run_conv(torch.zeros_like(x), w, b, None, 1, 1, tile=L.TILE_64x64 | L.TILE_H2)
    assert torch.equal(y0, b.view(1, -1, 1, 1).expand_as(y0))


@pytest.mark.parametrize('shape', [(2, 64, 9, 11, 96), (1, 256, 18, 18, 256), (1, 32, 37, 35, 64)])
@pytest.mark.parametrize('tile', [L.TILE_64x64, L.TILE_128x128, L.TILE_64x128, L.TILE_256x128_W8, L.TILE_32x64_K2, L.TILE_128x128_S3])
@pytest.mark.parametrize('m', [2, 4])
@pytest.mark.parametrize('v_planes', [False, True])
def test_winograd_fp16x2(shape, tile, m, v_planes):
    """The Winograd path on fp16x2 GEMM tiles: U as fp16 planes with a scale per (component, filter row); V either fp32 and split
    on the fly (scale from the input's magnitude bound times the transform's gain bound 4 / 100) or written by the input
    transform directly as two fp16 planes (`v_planes`: no operand split in the GEMM loop at all)."""
    from gpu_utils import run_wino, rel_err
    import torch.nn as nn
    B, Cin, H, W, Cout = shape
    g = _g(B * 100 + Cin + Cout + H + m)
    x = torch.randn(B, Cin, H, W, generator=g) * torch.exp(torch.randn(B, Cin, 1, 1, generator=g))
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    bn = nn.BatchNorm2d(Cout).eval()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(Cout, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(Cout, generator=g) * 0.1)
        bn.running_mean.copy_(torch.randn(Cout, generator=g) * 0.1)
        bn.running_var.copy_(torch.rand(Cout, generator=g) + 0.5)
        ref = F.relu(bn(F.conv2d(x, w, None, 1, 1)))
    y = run_wino(x, w, None, bn, L.ACT_RELU, tile | L.TILE_H2, m, v_planes=v_planes)
    y_fp32 = run_wino(x, w, None, bn, L.ACT_RELU, tile, m)
    assert rel_err(y, ref) < (2e-5 if m == 2 else 5e-5)
    assert rel_err(y, y_fp32) < (1e-5 if m == 2 else 3e-5)          # same algorithm, exact-fp32 MFMA GEMM
    assert run_wino.last_amax[1] == y_fp32.abs().max().item() or abs(run_wino.last_amax[1] - y_fp32.abs().max().item()) < 1e-3


@pytest.mark.parametrize('m', [2, 4])
@pytest.mark.parametrize('shape', [(8, 256, 69, 69, 256), (2, 256, 35, 35, 360), (1, 64, 21, 30, 128), (3, 512, 18, 18, 512), (1, 128, 9, 7, 36),
                                   (2, 96, 40, 33, 260)])
def test_winograd_persistent_grouped_gemm(shape, m):
    """csrc/wgemm.hip (tile YMI_TILE_WG_128x256 | YMI_TILE_H2, V as fp16 planes): the grouped GEMM of the Winograd path as one persistent
    producer / consumer launch — work items that outnumber the CUs and fewer items than CUs, ragged row tiles, column counts that are
    no multiple of 256 / 128 / 32, K = 64 .. 512, F(2x2) and F(4x4) — BIT-IDENTICAL to the 128 x 128 fp16x2 tile on the same planes
    (same products, same K order), and against torch fp32; an ordinary convolution or fp32 V is refused."""
    from gpu_utils import run_wino, run_conv, rel_err
    B, Cin, H, W, Cout = shape
    g = _g(B * 100 + Cin + Cout + H + m + 7)
    x = torch.randn(B, Cin, H, W, generator=g) * torch.exp(torch.randn(B, Cin, 1, 1, generator=g))
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    b = torch.randn(Cout, generator=g) * 0.1
    ref = F.relu(F.conv2d(x, w, b, 1, 1))
    t = L.TILE_WG_128x256 | L.TILE_H2
    y = run_wino(x, w, b, None, L.ACT_RELU, t, m, v_planes=True)
    y0 = run_wino(x, w, b, None, L.ACT_RELU, L.TILE_128x128 | L.TILE_H2, m, v_planes=True)
    assert torch.equal(y, y0)
    assert rel_err(y, ref) < (2e-5 if m == 2 else 5e-5)
    assert torch.equal(y, run_wino(x, w, b, None, L.ACT_RELU, t, m, v_planes=True))
    with pytest.raises(RuntimeError):                                     # V not written as planes
        run_wino(x, w, b, None, L.ACT_RELU, t, m, v_planes=False)
    with pytest.raises(RuntimeError):                                     # not a tile of the conv engine
        run_conv(x, w, b, None, 1, 1, tile=t)


@pytest.mark.parametrize('B,H,W', [(2, 61, 77), (1, 550, 550)])
def test_fused_stem_matches_reference_and_the_three_launches(B, H, W):
    """ymi_stem_pool_f32 (csrc/stem.hip): NCHW image -> conv 7x7/2 + BN + ReLU -> max-pool 3x3/2 -> NHWC in one launch, per-tile input
    scales.  fp32-class against an fp64 torch reference (ragged size, a hot patch so that tiles pick different scales), and
    BIT-IDENTICAL to the three launches it replaces in the plan as long as no fp16 piece goes subnormal (a power-of-two scale does
    not change which mantissa bits the two pieces keep), including the magnitude bound it reports."""
    from gpu_utils import DEV
    import ctypes as C
    import torch.nn as nn
    from yolact_amd.engine import Packed, out_size
    lib, s = L.lib(), L.stream_ptr()
    g = _g(7)
    w = torch.randn(64, 3, 7, 7, generator=g) * 0.08
    bn = nn.BatchNorm2d(64)
    bn.weight.data = torch.rand(64, generator=g) + 0.5
    bn.bias.data = torch.randn(64, generator=g) * 0.2
    bn.running_mean = torch.randn(64, generator=g) * 0.1
    bn.running_var = torch.rand(64, generator=g) + 0.5
    bn.eval()
    pk = Packed(w, None, bn, 2, 3, 4, DEV)
    hp, sc2, winv = pk.h2()
    x = torch.randn(B, 3, H, W, generator=g) * 1.3
    if H < 100:                                      # the hot patch: neighbouring tiles pick different input scales
        x[-1, :, H // 2:H // 2 + 4, W // 2:W // 2 + 4] *= 30.0
    xd = x.to(DEV)
    Hs, Ws = out_size(H, 7, 2, 3), out_size(W, 7, 2, 3)
    Hp, Wp = out_size(Hs, 3, 2, 1), out_size(Ws, 3, 2, 1)
    y = torch.zeros(B, Hp, Wp, 64, device=DEV)
    amax = torch.zeros(3 * 1024, device=DEV)
    d = L.StemDesc()
    d.x, d.y, d.B, d.H, d.W, d.cout_pad, d.kpad = xd.data_ptr(), y.data_ptr(), B, H, W, pk.CoutPad, pk.Kpad
    d.w_h2, d.scale_h2, d.bias, d.y_amax = hp.data_ptr(), sc2.data_ptr(), pk.bias.data_ptr(), amax.data_ptr()
    L.check(lib.ymi_stem_pool_f32(C.byref(d), s), 'stem')
    # the plan's three launches
    x4 = torch.empty(B, H, W, 4, device=DEV); st = torch.empty(B, Hs, Ws, 64, device=DEV); y3 = torch.empty_like(y)
    L.check(lib.ymi_nchw_to_nhwc4_amax_f32(xd.data_ptr(), x4.data_ptr(), B, 3, H, W, amax.data_ptr() + 4096, s))
    cd = L.ConvDesc()
    cd.x, cd.w, cd.bias, cd.scale = x4.data_ptr(), pk.w.data_ptr(), pk.bias.data_ptr(), pk.scale.data_ptr()
    cd.B, cd.H, cd.W, cd.Cin, cd.ldx, cd.Ho, cd.Wo, cd.Cout = B, H, W, 4, 4, Hs, Ws, 64
    cd.kh, cd.kw, cd.stride, cd.pad, cd.Kpad, cd.nseg, cd.cin_alg = 7, 7, 2, 3, pk.Kpad, 1, 3
    cd.seg[0] = L.ConvSeg(0, 64, L.ACT_RELU, 64, Hs * Ws * 64, st.data_ptr())
    cd.w_h2, cd.scale_h2, cd.winv_h2 = hp.data_ptr(), sc2.data_ptr(), winv.data_ptr()
    cd.x_amax, cd.y_amax = amax.data_ptr() + 4096, amax.data_ptr() + 8192
    cd.tile = L.TILE_128x64 | L.TILE_H2
    L.check(lib.ymi_conv2d_nhwc_f32(C.byref(cd), s), 'stem conv')
    L.check(lib.ymi_maxpool3x3s2_nhwc_f32(st.data_ptr(), y3.data_ptr(), B, Hs, Ws, 64, Hp, Wp, s))
    torch.cuda.synchronize()
    assert amax[:1024].max().item() == y.max().item()
    if H >= 100:
        # same dynamic range everywhere: no piece of either path leaves the normal fp16 range, and then the split is scale-invariant
        assert torch.equal(y, y3), 'fused stem differs from conv + max-pool'
    else:
        # under the image-wide scale of the separate launches the small values far from the hot patch lose low bits (subnormal
        # low pieces); the per-tile scale keeps them: equal to rounding, the fused launch at least as close to fp64
        assert (y - y3).abs().max().item() <= 1e-6 * y3.abs().max().item()
    if H < 100:
        t = torch.nn.functional.conv2d(x.double(), w.double(), stride=2, padding=3)
        inv = 1.0 / torch.sqrt(bn.running_var.double() + bn.eps)
        t = t * (bn.weight.double() * inv).view(1, -1, 1, 1) + (bn.bias.double() - bn.running_mean.double() * bn.weight.double() * inv).view(1, -1, 1, 1)
        ref = torch.nn.functional.max_pool2d(torch.relu(t), 3, 2, 1).permute(0, 2, 3, 1).contiguous()
        err = (y.cpu().double() - ref).abs().max().item() / ref.abs().max().item()
        print('fused stem: max error %.2e of max|y|' % err)
        assert err < 1e-6


@pytest.mark.parametrize('tile,v_planes,relu', [(L.TILE_64x64, False, True), (L.TILE_64x64 | L.TILE_H2, True, True),
                                                (L.TILE_128x128 | L.TILE_H2, False, False)])
def test_winograd_fused_upsample_is_bit_identical(tile, v_planes, relu):
    """ymi_wino_desc.x_up: the F(4x4,3x3) input transform interpolates its 6 x 6 patches from the half-size tensor (protonet's
    interpolate -> conv, utils/functions.py:187-206) instead of reading a materialised upsampling: same bits as
    ymi_bilinear_nhwc_f32 followed by the plain launch, including the image borders where the reference clamps its source rows."""
    from gpu_utils import DEV, run_wino, nhwc, nchw
    g = _g(23)
    B, Cc, Hl, Wl = 2, 32, 11, 13
    lo = torch.randn(B, Cc, Hl, Wl, generator=g)
    w = torch.randn(32, Cc, 3, 3, generator=g) * 0.1
    bias = torch.randn(32, generator=g) * 0.1
    lod = nhwc(lo).to(DEV)
    up = torch.empty(B, 2 * Hl, 2 * Wl, Cc, device=DEV)
    L.check(L.lib().ymi_bilinear_nhwc_f32(lod.data_ptr(), up.data_ptr(), B, Hl, Wl, Cc, 2 * Hl, 2 * Wl, 0.5, 0.5, 1 if relu else 0,
                                         L.stream_ptr()))
    torch.cuda.synchronize()
    x_up = nchw(up.cpu())
    ref = torch.nn.functional.interpolate(lo, scale_factor=2, mode='bilinear', align_corners=False)
    if relu:
        ref = torch.relu(ref)
    assert (x_up - ref).abs().max().item() < 1e-6          # the materialised upsampling itself is torch's
    y_sep = run_wino(x_up, w, bias, None, L.ACT_RELU, tile, 4, v_planes)
    y_fused = run_wino(x_up, w, bias, None, L.ACT_RELU, tile, 4, v_planes, up_from=lo, up_relu=relu)
    assert torch.equal(y_fused, y_sep)
    with pytest.raises(RuntimeError):                      # F(2x2) has no fused form
        run_wino(x_up, w, bias, None, L.ACT_RELU, tile, 2, v_planes, up_from=lo, up_relu=relu)
