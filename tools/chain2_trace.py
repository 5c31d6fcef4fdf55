#!/usr/bin/env python
"""Where a block of csrc/chain2.hip spends its time (diagnostics build, `chain2trace` stage of tools/gpu_session.sh): synthetic
35 x 35 x 8 (P = 256) and 69 x 69 x 8 (P = 128) problems through ymi_pointwise_chain_f32 with YMI_CHAIN2_TRACE set."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))


def main():
    import gpu_utils as G
    from yolact_amd import _lib as L
    dev = torch.device('cuda', 0)
    for P, M in ((256, 9800), (128, 38088)):
        g = torch.Generator().manual_seed(1)
        x = torch.randn(M, P, generator=g).abs()
        wa = torch.randn(4 * P, P, generator=g) / P ** 0.5
        ba = torch.randn(4 * P, generator=g) * 0.1
        res = torch.randn(M, 4 * P, generator=g)
        wb = torch.randn(P, 4 * P, generator=g) / (4 * P) ** 0.5
        bb = torch.randn(P, generator=g) * 0.1
        nb = (M + 63) // 64
        buf = torch.zeros(nb * 16, dtype=torch.int64, device=dev)
        os.environ.pop('YMI_CHAIN2_TRACE', None)
        G.run_chain(x, wa, ba, res, wb, bb)
        os.environ['YMI_CHAIN2_TRACE'] = str(buf.data_ptr())
        for _ in range(2):
            G.run_chain(x, wa, ba, res, wb, bb)
        os.environ.pop('YMI_CHAIN2_TRACE', None)
        t = buf.cpu().view(nb, 16).double()
        t = t[t[:, 15] == 1]
        tot = t[:, 6] - t[:, 0]
        steps = 8 * (P // 32)           # chunk steps per block: NS * CPS
        print('P = %d, M = %d: %d blocks; per block (shader cycles, mean over blocks): total %.0f | prologue %.0f | loop %.0f | epilogue 2 + drain %.0f' % (
            P, M, t.shape[0], tot.mean(), (t[:, 1] - t[:, 0]).mean(), (t[:, 5] - t[:, 1]).mean(), (t[:, 6] - t[:, 5]).mean()))
        print('    in the loop: %d chunk steps, %.0f cycles per step; waiting for the next chunk %.0f %%, at the step barrier %.0f %%, epilogue 1 (incl. its barrier) %.0f %%' % (
            steps, (t[:, 5] - t[:, 1]).mean() / steps, 100 * (t[:, 2] / (t[:, 5] - t[:, 1])).mean(), 100 * (t[:, 3] / (t[:, 5] - t[:, 1])).mean(),
            100 * (t[:, 4] / (t[:, 5] - t[:, 1])).mean()))


if __name__ == '__main__':
    main()
