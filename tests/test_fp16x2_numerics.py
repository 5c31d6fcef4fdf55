"""CPU checks of the fp16x2 arithmetic the default GEMM tiles use (csrc/conv_igemm.hip PREC 3 / 4, engine.split2_planes_f16):
what the two-piece fp16 representation loses, what the three kept piece products lose, and that the power-of-two scaling
helper of csrc/common.h does what the kernels assume.  The GPU tests (tests/test_gpu_kernels.py::test_conv_fp16x2_is_fp32_class)
measure the kernels themselves against fp64; this file pins the scheme without a GPU."""
import os
import subprocess

import numpy as np
import pytest
import torch

from yolact_amd.engine import split2_planes_f16

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _split(x, s):
    """numpy model of split8h: t = x * s, h = fp16(t) (round to nearest even), l = fp16(t - h)."""
    t = (x * s).astype(np.float32)
    h = t.astype(np.float16)
    l = (t - h.astype(np.float32)).astype(np.float16)
    return h, l


def test_two_fp16_pieces_represent_fp32_to_one_ulp():
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(400000) * np.exp(rng.standard_normal(400000) * 2)).astype(np.float32)
    s = np.float32(2.0 ** (13 - np.ceil(np.log2(np.abs(x).max()))))          # amax * s in [2^12, 2^13]: inside the kernels' window
    h, l = _split(x, s)
    rec = (h.astype(np.float64) + l.astype(np.float64)) / float(s)
    err = np.abs(rec - x.astype(np.float64))
    ulp = np.spacing(np.abs(x)).astype(np.float64)
    big = np.abs(x) * float(s) >= 2.0 ** -2          # 15 binades below the maximum and up: both pieces are normal fp16 numbers
    assert (err[big] <= ulp[big]).all()              # never worse than one fp32 ulp
    exact = (err[big] == 0).mean()
    assert 0.60 < exact < 0.80, exact                # about two thirds of all values exactly, the rest to one ulp
    signed = ((rec - x.astype(np.float64)) / ulp)[big]
    assert abs(signed.mean()) < 0.01                 # round to nearest: unbiased (the bf16x3 truncation split is one-sided)
    # below the window the low piece goes subnormal: absolute error stays <= 2^-25 of the scaled unit, i.e. 2^-38 of the maximum
    assert (err[~big] * float(s) <= 2.0 ** -24).all()


def test_three_piece_products_are_fp32_class():
    """K = 2304 dot products: fp16x2 (hh + hl + lh, exact products, sum in fp64 here to isolate the PRODUCT error) against the
    exact result; the error must be far below what fp32 accumulation alone costs (~2^-24 sqrt(K) of sum|ab|)."""
    rng = np.random.default_rng(1)
    K, N = 2304, 512
    a = (rng.standard_normal((N, K)) * np.exp(rng.standard_normal((N, 1)))).astype(np.float32)
    b = (rng.standard_normal((N, K)) / 48).astype(np.float32)
    sa = np.float32(2.0 ** (13 - np.ceil(np.log2(np.abs(a).max()))))
    sb = np.float32(2.0 ** (13 - np.ceil(np.log2(np.abs(b).max()))))
    ah, al = (v.astype(np.float64) for v in _split(a, sa))
    bh, bl = (v.astype(np.float64) for v in _split(b, sb))
    got = ((ah * bh + ah * bl + al * bh).sum(1)) / (float(sa) * float(sb))
    ref = (a.astype(np.float64) * b.astype(np.float64)).sum(1)
    mag = np.abs(a.astype(np.float64) * b.astype(np.float64)).sum(1)
    e_h2 = (np.abs(got - ref) / mag).max()
    # exact-fp32-product model: every product rounded to fp32 once (what v_mfma_f32_32x32x2_f32 does before accumulating)
    prod32 = (a.astype(np.float64) * b.astype(np.float64)).astype(np.float32).astype(np.float64).sum(1)
    e_p32 = (np.abs(prod32 - ref) / mag).max()
    acc32 = 2.0 ** -24 * np.sqrt(K)                  # scale of the fp32 ACCUMULATION error both kernels share
    print('product error of sum|ab|: fp16x2 %.2e, one fp32 rounding per product %.2e; fp32 accumulation ~%.1e' % (e_h2, e_p32, acc32))
    assert e_h2 < 2e-8 and e_h2 < acc32 / 50         # two orders of magnitude below the accumulation error
    assert e_h2 < 4 * e_p32                          # same class as rounding each product to fp32


def test_filter_planes_reconstruct_and_scale_rows_independently():
    g = torch.Generator().manual_seed(3)
    w = torch.randn(6, 64, generator=g)
    w[0] *= 1e-12; w[1] *= 3e5; w[2] = 0; w[3, 5] = 40.0
    planes, winv = split2_planes_f16(w)
    assert planes.shape == (2, 6, 64) and planes.dtype == torch.int16 and winv.shape == (6,)
    h, l = planes[0].view(torch.float16).double(), planes[1].view(torch.float16).double()
    rec = (h + l) * winv.double()[:, None]
    ulp = torch.from_numpy(np.spacing(w.abs().numpy())).double()
    assert bool(((rec - w.double()).abs() <= ulp).all())
    amax_scaled = (w.abs().amax(1) / winv)
    assert bool(((amax_scaled[[0, 1, 3, 4, 5]] >= 2 ** 13) & (amax_scaled[[0, 1, 3, 4, 5]] < 2 ** 14)).all())
    assert winv[2] == 1.0                            # an all-zero (padding) row keeps scale 1
    fr, ex = torch.frexp(winv)
    assert bool((fr == 0.5).all())                   # powers of two: folding 1/s into the epilogue scale is exact


def test_h2_scale_helper(tmp_path):
    """ymi_h2_scale (csrc/common.h): amax * s lands in [2^13, 2^14), s and 1/s are exact powers of two, degenerate bounds give 1."""
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    if not os.path.exists(hipcc):
        pytest.skip('no hipcc')
    src = tmp_path / 't.hip'
    src.write_text('#include "%s"\n#include <cstdio>\n#include <cmath>\nint main(){ const float v[] = {1.f, 0.9999f, 3.0e4f, 5.5f, 1e-30f, 1e30f, 0.f, INFINITY, 16384.f, '
                   '16383.9f, 1e-44f};\n for (float a : v) { float s, i; ymi_h2_scale(a, s, i); printf("%%a %%a %%a\\n", a, s, i); } return 0; }\n'
                   % os.path.join(ROOT, 'yolact_amd', 'csrc', 'common.h'))
    exe = tmp_path / 't'
    subprocess.run([hipcc, '--offload-arch=gfx950', '-O1', str(src), '-o', str(exe)], check=True, capture_output=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split('\n')
    for line in out:
        if not line.strip():
            continue
        a, s, i = (float.fromhex(t) if t not in ('inf', '-inf', 'nan') else float(t) for t in line.split())
        assert s * i == 1.0 and np.frexp(s)[0] == 0.5
        if np.isfinite(a) and a >= 1.2e-38:
            if 1e-20 < a < 1e20:
                assert 2.0 ** 13 <= a * s < 2.0 ** 14, (a, s)
            else:
                assert a * s < 2.0 ** 14         # clamped scale: never above the window
        else:
            assert s == 1.0, (a, s)


def test_split_is_invariant_under_the_power_of_two_scale():
    """Why the per-tile scales of csrc/stem.hip / csrc/bottleneck.hip reproduce the per-tensor scale of the conv engine bit for
    bit: a power of two only shifts exponents, so h and l keep the same mantissa bits as long as neither leaves the normal fp16
    range — the pieces of two scales differ by exactly that power of two, the represented value not at all."""
    rng = np.random.default_rng(5)
    x = rng.standard_normal(200000).astype(np.float32)
    x = x[np.abs(x) > 2.0 ** -6]                                  # 6 binades of dynamic range below ~4
    h1, l1 = _split(x, np.float32(2.0 ** 11))                     # |x| s in [2^5, 2^13]: both pieces normal
    h2, l2 = _split(x, np.float32(2.0 ** 8))                      # a tile whose maximum is 8x larger
    assert np.array_equal(h1.astype(np.float32), h2.astype(np.float32) * np.float32(8))
    assert np.array_equal(l1.astype(np.float32), l2.astype(np.float32) * np.float32(8))
    # ... and where the low piece does go subnormal the coarser scale loses bits (the hot-patch case of the stem test)
    small = (rng.standard_normal(20000) * 2.0 ** -16).astype(np.float32)
    ha, la = _split(small, np.float32(2.0 ** 11))
    hb, lb = _split(small, np.float32(2.0 ** 4))
    ra = (ha.astype(np.float64) + la.astype(np.float64)) / 2.0 ** 11
    rb = (hb.astype(np.float64) + lb.astype(np.float64)) / 2.0 ** 4
    assert np.abs(ra - small).max() <= np.abs(rb - small).max() and np.abs(rb - small).max() > 0
