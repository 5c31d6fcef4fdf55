"""The DCNv2 forward path against the REFERENCE'S OWN kernel, compiled for gfx950 (oracle/build_ref.sh ->
oracle/_ref/libdcn_v2_im2col_ref.so from /root/reference/external/DCNv2/src/cuda/dcn_v2_im2col_cuda.cu, no source copied).

The reference ships no CPU DCN (src/cpu/dcn_v2_cpu.cpp:23), so until round 3 the DCN restatement in oracle/yolact_oracle.py
was pinned only by zero-offset KATs and the `plus_r50` golden was circular for the DCN layers.  Here both the CPU oracle and
the HIP kernel are compared with  columns = modulated_deformable_im2col_cuda(...)  (dcn_v2_im2col_cuda.cu:125-195, :329-352)
followed by  out[b] = weight.view(Co, -1) @ columns[b] + bias  (dcn_v2_cuda.cu:123-163; the GEMM in fp64), with random
NON-INTEGER offsets, many of them leaving the image, stride 1 and 2, batch > 1.
"""
import ctypes as C
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, 'oracle', '_ref', 'libdcn_v2_im2col_ref.so')
DEV = 'cuda:0'


def _ref_lib():
    if not os.path.exists(REF_SO):
        pytest.skip('oracle/_ref/libdcn_v2_im2col_ref.so not built (oracle/build_ref.sh needs /root/reference; the GPU box '
                    'only carries the prebuilt file)')
    lib = C.CDLL(REF_SO)
    lib.modulated_deformable_im2col_cuda.restype = None
    lib.modulated_deformable_im2col_cuda.argtypes = [C.c_void_p] * 4 + [C.c_int] * 15 + [C.c_void_p]
    return lib


def reference_dcn_forward(lib, x, offset, mask, weight, bias, stride, pad):
    """dcn_v2_cuda_forward (dcn_v2_cuda.cu:42-172) with the reference's own im2col kernel; x/offset/mask NCHW on the GPU."""
    B, Cin, H, W = x.shape
    Co, _, kh, kw = weight.shape
    Ho, Wo = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
    cols = torch.full((B, Cin * kh * kw, Ho * Wo), float('nan'), device=x.device)
    x, offset, mask = x.contiguous(), offset.contiguous(), mask.contiguous()
    lib.modulated_deformable_im2col_cuda(C.c_void_p(torch.cuda.current_stream().cuda_stream), x.data_ptr(), offset.data_ptr(),
                                         mask.data_ptr(), B, Cin, H, W, Ho, Wo, kh, kw, pad, pad, stride, stride, 1, 1, 1,
                                         cols.data_ptr())
    torch.cuda.synchronize()
    assert torch.isfinite(cols).all()
    out = weight.view(Co, -1).double() @ cols.double() + bias.double()[None, :, None]
    return out.view(B, Co, Ho, Wo), cols


def _case(seed, B, Cin, H, W, Co, stride, spread):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Co, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    b = torch.randn(Co, generator=g) * 0.1
    Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
    om = torch.randn(B, 27, Ho, Wo, generator=g)
    om[:, :18] *= spread                      # offsets of several pixels: many sample points leave the image
    om[0, :18, 0, 0] = torch.tensor([-1.0, -1.0, -0.999, 0.5, float(H), 0.25, 0.0, float(W)] + [0.0] * 10)   # the :155 edge tests
    return x, w, b, om


@pytest.mark.parametrize('stride', [1, 2])
@pytest.mark.parametrize('shape', [(2, 64, 19, 23, 64, 2.5), (1, 128, 35, 35, 128, 6.0), (3, 32, 9, 7, 96, 1.0)])
def test_dcn_hip_and_oracle_match_reference_kernel(shape, stride):
    from gpu_utils import run_conv
    from oracle.yolact_oracle import dcn_v2_forward
    from yolact_amd import _lib as L
    lib = _ref_lib()
    B, Cin, H, W, Co, spread = shape
    x, w, b, om = _case(100 + 7 * stride + Cin, B, Cin, H, W, Co, stride, spread)
    offset, mask = om[:, :18].contiguous(), torch.sigmoid(om[:, 18:]).contiguous()         # dcn_v2.py:119-122
    ref, cols = reference_dcn_forward(lib, x.to(DEV), offset.to(DEV), mask.to(DEV), w.to(DEV), b.to(DEV), stride, 1)
    ref = ref.cpu()
    outside = ((cols == 0).float().mean().item())
    scale = ref.abs().max().item()
    # (1) the CPU oracle restatement == the reference kernel (this is the oracle's non-circular DCN pin)
    orc = dcn_v2_forward(x, offset, mask, w, b, stride, 1, 1)
    e_or = (orc.double() - ref).abs().max().item()
    # (2) the HIP path (gather fused into the implicit-GEMM loader, ymi_dcn_v2_forward_f32) == the reference kernel
    hip = run_conv(x, w, b, None, stride, 1, dcn_offmask=om)
    e_hip = (hip.double() - ref).abs().max().item()
    # (3) the same launch on the fp16x2 tiles (what the default plan runs for the DCN layers)
    hip2 = run_conv(x, w, b, None, stride, 1, dcn_offmask=om, tile=L.TILE_64x64 | L.TILE_H2)
    e_h2 = (hip2.double() - ref).abs().max().item()
    # (4) the pipelined gather-GEMM of csrc/dcn.hip (round 4: what the default plan runs for the DCN layers), every block tile
    e_p = {}
    if Co % 4 == 0:
        for t in sorted(L.DCNP_TILES):
            if t in L.DCNP_PLAIN_ONLY:               # 64x64 wave tiles: ordinary convolutions only
                continue
            hp = run_conv(x, w, b, None, stride, 1, dcn_offmask=om, tile=t | L.TILE_H2 | L.TILE_DCNP)
            e_p[L.DCNP_TILES[t]] = (hp.double() - ref).abs().max().item()
        print('    pipelined: ' + '  '.join('%s %.2e' % kv for kv in e_p.items()))
        assert max(e_p.values()) <= 1e-4 * max(1.0, scale), e_p
    print('DCN %s stride %d: %.1f%% zero columns, |ref|max %.3f, oracle err %.2e, HIP err %.2e (exact-fp32 tile) %.2e (fp16x2 tile)' % (
        shape[:5], stride, 100 * outside, scale, e_or, e_hip, e_h2))
    assert e_or <= 2e-5 * max(1.0, scale), e_or
    assert e_hip <= 1e-4 * max(1.0, scale), e_hip
    assert e_h2 <= 1e-4 * max(1.0, scale), e_h2


def test_reference_kernel_zero_offset_identity():
    """The reference's own KAT (external/DCNv2/test.py:32-67): zero offsets, mask 1 => a plain convolution."""
    lib = _ref_lib()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 16, 11, 13, generator=g)
    w = torch.randn(8, 16, 3, 3, generator=g) * 0.1
    b = torch.zeros(8)
    off = torch.zeros(2, 18, 11, 13)
    ref, _ = reference_dcn_forward(lib, x.to(DEV), off.to(DEV), torch.ones(2, 9, 11, 13, device=DEV), w.to(DEV), b.to(DEV), 1, 1)
    want = torch.nn.functional.conv2d(x.double(), w.double(), None, 1, 1)
    assert (ref.cpu() - want).abs().max().item() < 1e-5
