"""Host side of the Winograd path (engine.WinoPacked): the pre-transformed filters U = G g G^T, laid out
[(m+2)^2][CoutPad][C], must reproduce a 3x3 / stride-1 / pad-1 convolution when combined with the published input /
output transforms (Lavin & Gray: F(2x2,3x3) and F(4x4,3x3), interpolation points 0, +-1, +-2, inf) — the same algebra
csrc/winograd.hip runs on the device (B^T d B -> per-group GEMM -> A^T M A).  CPU-only: checks the matrices, the group
index e = i*(m+2) + j and the zero padding of CoutPad; the kernels themselves are tested in test_gpu_kernels.py."""
import pytest
import torch
import torch.nn.functional as F

from yolact_amd.engine import WinoPacked, wino_eligible, Packed

BT = {2: torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64),
      4: torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0],
                       [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=torch.float64)}
AT = {2: torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64),
      4: torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]],
                      dtype=torch.float64)}


def winograd_conv(x, wp, cout, m):
    """x [B,C,H,W] fp64; wp: WinoPacked.  Tiles of m x m outputs, (m+2)^2 independent [T x C] x [C x Cout] products."""
    B, C, H, W = x.shape
    a = m + 2
    th, tw = -(-H // m), -(-W // m)
    xp = F.pad(x, (1, tw * m + 1 - W, 1, th * m + 1 - H))
    d = xp.unfold(2, a, m).unfold(3, a, m)                                     # [B,C,th,tw,a,a]
    V = torch.einsum('ij,bcyxjk,lk->ilbyxc', BT[m], d, BT[m]).reshape(a * a, B * th * tw, C)
    U = wp.u.double()                                                          # [a*a, CoutPad, C]
    Mg = torch.einsum('etc,enc->etn', V, U)[:, :, :cout]                       # the grouped GEMM
    Mt = Mg.reshape(a, a, B, th, tw, cout)
    Y = torch.einsum('ij,jkbyxn,lk->bnyixl', AT[m], Mt, AT[m]).reshape(B, cout, th * m, tw * m)
    return Y[:, :, :H, :W]


@pytest.mark.parametrize('m', [2, 4])
@pytest.mark.parametrize('shape', [(2, 32, 7, 9, 5), (1, 64, 12, 8, 130), (1, 32, 1, 1, 3)])
def test_packed_filters_reproduce_conv(m, shape):
    B, C, H, W, N = shape
    g = torch.Generator().manual_seed(m * 100 + H)
    x = torch.randn(B, C, H, W, generator=g, dtype=torch.float64)
    w = torch.randn(N, C, 3, 3, generator=g) / (9 * C) ** 0.5
    wp = WinoPacked(w, torch.device('cpu'), m)
    a = m + 2
    assert wp.u.shape == (a * a, (N + 127) // 128 * 128, C) and wp.u.dtype == torch.float32 and wp.m == m
    assert wp.u[:, N:].abs().max() == 0                                       # zero filter rows up to CoutPad
    ref = F.conv2d(x, w.double(), None, 1, 1)
    got = winograd_conv(x, wp, N, m)
    # only U is rounded to fp32 here: ~1e-7 relative, times the transform's amplification (<= ~50 for F(4x4))
    assert (got - ref).abs().max().item() < 2e-5 * ref.abs().max().item()


def test_eligibility_rules():
    w3 = torch.randn(64, 32, 3, 3)
    pk = Packed(w3, None, None, 1, 1, None, torch.device('cpu'))
    assert wino_eligible(pk, None, None, 1, 32)
    assert wino_eligible(pk, None, [(0, 64, 0, 64, 0, 1)], 0, 32)             # segmented heads: any Cout / activation
    assert not wino_eligible(pk, object(), None, 1, 32)                       # residual epilogues stay direct
    assert not wino_eligible(Packed(w3, None, None, 2, 1, None, torch.device('cpu')), None, None, 1, 32)   # stride 2
    assert not wino_eligible(Packed(torch.randn(64, 32, 1, 1), None, None, 1, 0, None, torch.device('cpu')), None, None, 1, 32)
    assert not wino_eligible(Packed(torch.randn(62, 32, 3, 3), None, None, 1, 1, None, torch.device('cpu')), None, None, 1, 32)
    assert not wino_eligible(pk, None, None, 3, 32)                           # tanh only through segments
