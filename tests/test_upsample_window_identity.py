"""CPU proof of the border argument behind ymi_wino_desc.x_up (csrc/winograd.hip, wino43_in_k<.., UPS = true>): a tile's 6 x 6
hi-res patch is built from the 4 x 4 low-res window starting at (2 ty - 1, 2 tx - 1) with CLAMPED window addresses, hi-res offset r
using window rows r >> 1 and (r >> 1) + 1, the weights of bl_coord and the bracket order of bilinear_nhwc_k.  Both algorithms are
restated here in numpy fp32, operation by operation, and compared BIT FOR BIT on every hi-res pixel of every tile for sizes that
exercise all the clamping cases (1-pixel-wide inputs, odd sizes, the last tile's overhang).  The GPU test
tests/test_gpu_kernels.py::test_winograd_fused_upsample_is_bit_identical checks the kernels themselves."""
import numpy as np
import pytest

F = np.float32


def bl_coord(dst, in_size):
    """csrc/layout.hip bl_coord with scale 0.5 (torch's area_pixel_compute_source_index, align_corners=False), fp32."""
    src = F(0.5) * (F(dst) + F(0.5)) - F(0.5)
    src = F(0.0) if src < 0 else src
    i0 = int(src)
    if i0 > in_size - 1:
        i0 = in_size - 1
    i1 = i0 + (1 if i0 < in_size - 1 else 0)
    return i0, i1, F(src - F(i0))


def upsample_reference(lo):
    """bilinear_nhwc_k: v = hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11), one pixel at a time."""
    Hl, Wl = lo.shape
    out = np.zeros((2 * Hl, 2 * Wl), F)
    for oy in range(2 * Hl):
        y0, y1, ly = bl_coord(oy, Hl)
        for ox in range(2 * Wl):
            x0, x1, lx = bl_coord(ox, Wl)
            hy, hx = F(F(1) - ly), F(F(1) - lx)
            top = F(F(hx * lo[y0, x0]) + F(lx * lo[y0, x1]))
            bot = F(F(hx * lo[y1, x0]) + F(lx * lo[y1, x1]))
            out[oy, ox] = F(F(hy * top) + F(ly * bot))
    return out


def upsample_windowed(lo):
    """The fused input transform's patch construction, tile by tile (F(4x4,3x3): tiles of 4 x 4 outputs, patch origin 4 t - 1)."""
    Hl, Wl = lo.shape
    H, W = 2 * Hl, 2 * Wl
    out = np.full((H, W), np.nan, F)
    clamp = lambda v, n: 0 if v < 0 else (n - 1 if v > n - 1 else v)
    for ty in range((H + 3) // 4):
        for tx in range((W + 3) // 4):
            y0, x0 = 4 * ty - 1, 4 * tx - 1
            wy = [clamp(2 * ty - 1 + k, Hl) for k in range(4)]
            wx = [clamp(2 * tx - 1 + k, Wl) for k in range(4)]
            lx = [bl_coord(x0 + j, Wl)[2] for j in range(6)]
            ly = [bl_coord(y0 + j, Hl)[2] for j in range(6)]
            hrow = [[F(F(F(F(1) - lx[j]) * lo[wy[k], wx[j >> 1]]) + F(lx[j] * lo[wy[k], wx[(j >> 1) + 1]])) for j in range(6)]
                    for k in range(4)]
            for r in range(6):
                for j in range(6):
                    yy, xx = y0 + r, x0 + j
                    if 0 <= yy < H and 0 <= xx < W:
                        v = F(F(F(F(1) - ly[r]) * hrow[r >> 1][j]) + F(ly[r] * hrow[(r >> 1) + 1][j]))
                        if not np.isnan(out[yy, xx]):
                            assert out[yy, xx].tobytes() == v.tobytes()       # overlapping patches agree
                        out[yy, xx] = v
    return out


@pytest.mark.parametrize('Hl,Wl', [(1, 1), (1, 2), (2, 1), (2, 3), (3, 3), (4, 5), (5, 4), (7, 9), (11, 13), (16, 17)])
def test_windowed_patch_equals_pixelwise_bilinear(Hl, Wl):
    rng = np.random.default_rng(100 * Hl + Wl)
    lo = (rng.standard_normal((Hl, Wl)) * np.exp(rng.standard_normal((Hl, Wl)))).astype(F)
    ref = upsample_reference(lo)
    got = upsample_windowed(lo)
    assert not np.isnan(got).any()
    assert got.tobytes() == ref.tobytes()


def test_reference_restatement_is_torch_bilinear():
    """the numpy restatement of bilinear_nhwc_k is F.interpolate(scale_factor=2, mode='bilinear', align_corners=False) to fp32 rounding"""
    import torch
    rng = np.random.default_rng(3)
    lo = rng.standard_normal((9, 7)).astype(F)
    ref = torch.nn.functional.interpolate(torch.from_numpy(lo)[None, None], scale_factor=2, mode='bilinear', align_corners=False)[0, 0]
    assert np.abs(upsample_reference(lo) - ref.numpy()).max() < 1e-6
