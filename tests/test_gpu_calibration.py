"""The box-calibration kernels of csrc/calib.hip (ABI 7) through the C ABI: what bench.py's `box_calibration` block times.
They are not on the product path; the test pins their ACCOUNTING (the FLOPs / bytes they report are the ones they execute)
and that the rates they deliver on an MI355X are in the range the block's reader will assume."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = 'cuda:0'


def test_calibration_kernels_report_what_they_execute():
    from yolact_amd import _lib as L
    lib = L.lib()
    s = L.stream_ptr()
    out = torch.zeros(64, device=DEV)
    fl = C.c_double()
    L.check(lib.ymi_calib_mfma_f16(out.data_ptr(), 512, 1000, C.byref(fl), s), 'mfma')
    assert fl.value == 512 * 4 * 1000 * 4 * 2.0 * 32 * 32 * 16
    assert float(out.abs().sum()) == 0.0                      # the guard store never fires
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    L.check(lib.ymi_calib_mfma_f16(out.data_ptr(), 512, 20000, C.byref(fl), s), 'mfma')
    torch.cuda.synchronize()
    e0.record()
    for _ in range(4):
        L.check(lib.ymi_calib_mfma_f16(out.data_ptr(), 512, 20000, C.byref(fl), s), 'mfma')
    e1.record()
    e1.synchronize()
    tf = 4 * fl.value / (e0.elapsed_time(e1) * 1e-3) / 1e12
    assert 1200.0 < tf < 2600.0, tf                           # dense fp16 MFMA: 2500 spec, 1.7 - 2.5 PF under DVFS with random operands
    n = 1 << 26                                               # 256 MB
    src = torch.randn(n, device=DEV)
    dst = torch.zeros(n, device=DEV)
    by = C.c_double()
    L.check(lib.ymi_calib_hbm_copy(src.data_ptr(), dst.data_ptr(), n, C.byref(by), s), 'copy')
    assert by.value == 8.0 * n and torch.equal(src, dst)
    assert lib.ymi_calib_hbm_copy(src.data_ptr(), dst.data_ptr(), n - 1, C.byref(by), s) == -2
    assert lib.ymi_calib_hbm_copy(None, dst.data_ptr(), n, C.byref(by), s) == -3
    assert lib.ymi_calib_mfma_f16(out.data_ptr(), 0, 10, C.byref(fl), s) == -1


def test_latency_chain_ends_where_the_permutation_says():
    """ymi_calib_latency follows i = chain[i]: after `hops` steps along a known cycle the lane is at a known place; a warm 1 MB footprint
    answers at L2 latency, a 1 GiB one at memory latency (well apart on an MI355X)."""
    from yolact_amd import _lib as L
    lib = L.lib()
    s = L.stream_ptr()
    g = torch.Generator(device='cpu').manual_seed(5)
    out = torch.zeros(4, dtype=torch.int32, device=DEV)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ns = {}
    for mb, hops in ((1, 10000), (1024, 5000)):
        lines = mb * (1 << 20) // 128
        perm = torch.randperm(lines, generator=g)
        chain = torch.zeros(lines * 32, dtype=torch.int32)
        chain[perm * 32] = (torch.roll(perm, -1) * 32).to(torch.int32)
        chain_d = chain.to(DEV)
        for r in range(2):           # the second pass continues on the cycle: warm for 1 MB (every line seen), untouched lines for 1 GiB
            e0.record()
            L.check(lib.ymi_calib_latency(chain_d.data_ptr(), chain_d.numel(), int(perm[r * hops % lines]) * 32, hops, out.data_ptr(), s),
                    'latency')
            e1.record()
            e1.synchronize()
            assert int(out[0]) == int(perm[(r + 1) * hops % lines]) * 32
        ns[mb] = e0.elapsed_time(e1) * 1e6 / hops
    print('load-to-use latency: 1 MB %.0f ns, 1 GiB %.0f ns' % (ns[1], ns[1024]))
    assert 40.0 < ns[1] < 1500.0 and ns[1024] > 1.3 * ns[1], ns       # measured: 99 - 105 ns (L2), 385 - 390 ns (1 - 2 GiB)
    assert lib.ymi_calib_latency(None, 10, 0, 1, out.data_ptr(), s) == -3
    assert lib.ymi_calib_latency(chain_d.data_ptr(), 10, 10, 1, out.data_ptr(), s) == -1
    assert lib.ymi_calib_latency(chain_d.data_ptr(), 10, 0, 0, out.data_ptr(), s) == -1
