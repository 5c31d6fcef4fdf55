"""Geometry invariants of the fused stem's tiling (csrc/stem.hip), checked on the CPU from the constants in the source: every
pooled pixel's 3x3 / stride-2 window lies inside its tile's stem pixels, every stem pixel's 7x7 / stride-2 receptive field inside
the tile's input patch, the patch fits the two slots per thread the kernel fetches, and the tiles cover the pooled map exactly once.
(The arithmetic itself is tested on the GPU: tests/test_gpu_kernels.py::test_fused_stem_matches_reference_and_the_three_launches.)"""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = open(os.path.join(ROOT, 'yolact_amd', 'csrc', 'stem.hip')).read()


def _const(name):
    m = re.search(r'\b%s = (\d+)' % name, SRC)
    assert m, name
    return int(m.group(1))


PH, PW, NTHR, IPITCH = _const('PH'), _const('PW'), _const('NTHR'), _const('IPITCH')
SH, SW = 2 * PH + 1, 2 * PW + 1
IH, IW = 2 * SH + 5, 2 * SW + 5


def test_constants_are_consistent():
    assert SH * SW <= 128, 'stem pixels of a tile are the rows of ONE 128-row GEMM (4 m-tiles of 32)'
    assert IW <= IPITCH and IH * IPITCH <= 2 * NTHR, 'the patch is fetched as two slots per thread'
    assert _const('KPAD') == 224 and _const('COUT') == 64


@pytest.mark.parametrize('H,W', [(550, 550), (700, 700), (61, 77), (7, 7), (9, 300)])
def test_tiles_cover_and_contain(H, W):
    Hs, Ws = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    Hp, Wp = (Hs + 2 - 3) // 2 + 1, (Ws + 2 - 3) // 2 + 1
    tiles_y, tiles_x = -(-Hp // PH), -(-Wp // PW)
    seen = set()
    for ty in range(tiles_y):
        for tx in range(tiles_x):
            sy0, sx0 = 2 * ty * PH - 1, 2 * tx * PW - 1            # first stem pixel of the tile
            iy0, ix0 = 4 * ty * PH - 5, 4 * tx * PW - 5            # first input pixel of the patch
            for qy in range(PH):
                for qx in range(PW):
                    py, px = ty * PH + qy, tx * PW + qx
                    if py >= Hp or px >= Wp:
                        continue
                    assert (py, px) not in seen
                    seen.add((py, px))
                    for dy in range(3):
                        for dx in range(3):
                            sy, sx = 2 * py - 1 + dy, 2 * px - 1 + dx       # MaxPool2d(3, 2, 1) window
                            i, j = sy - sy0, sx - sx0
                            assert 0 <= i < SH and 0 <= j < SW and (i, j) == (2 * qy + dy, 2 * qx + dx)
            # receptive fields of the tile's corner stem pixels stay inside the patch
            for i, j in ((0, 0), (SH - 1, SW - 1)):
                sy, sx = sy0 + i, sx0 + j
                for ky, kx in ((0, 0), (6, 6)):
                    iy, ix = 2 * sy - 3 + ky, 2 * sx - 3 + kx               # Conv2d(7, stride 2, pad 3)
                    assert 0 <= iy - iy0 < IH and 0 <= ix - ix0 < IW
                    assert (iy - iy0, ix - ix0) == (2 * i + ky, 2 * j + kx)
    assert len(seen) == Hp * Wp
