import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'tests'))
from gpu_utils import run_chain
from yolact_amd import _lib as L
for M in (1000, 12293, 4097):
    g = torch.Generator().manual_seed(900 + M)
    x = torch.randn(M, 64, generator=g) * 3
    wa = torch.randn(256, 64, generator=g) / 8
    ba = torch.randn(256, generator=g) * 0.3
    res = torch.randn(M, 256, generator=g) * 2
    wb = torch.randn(64, 256, generator=g) / 16
    bb = torch.randn(64, generator=g) * 0.1
    yr = torch.relu(x.double() @ wa.double().t() + ba.double() + res.double())
    zr = torch.relu(yr @ wb.double().t() + bb.double())
    for rep in range(3):
        y, z = run_chain(x, wa, ba, res, wb, bb, L.ACT_RELU, L.ACT_RELU)
        e = (z.double() - zr).abs()
        bad = (e > 1e-4 * zr.abs().max()).nonzero()
        print('M', M, 'rep', rep, 'max err', e.max().item() / zr.abs().max().item(), 'bad elements', bad.shape[0])
        if bad.shape[0]:
            rows = sorted(set(bad[:, 0].tolist()))
            cols = sorted(set(bad[:, 1].tolist()))
            print('  rows', rows[:40], '... n', len(rows))
            print('  rows %32', sorted(set(r % 32 for r in rows)))
            print('  tiles', sorted(set(r // 32 for r in rows))[:40])
            print('  cols', cols)
            # which slice explains the error: recompute z leaving out / rescaling one 32-channel slice
            r0 = rows[0]
            for c in range(8):
                yy = yr[r0].clone(); 
                part = (yy[32*c:32*c+32] @ wb.double()[:, 32*c:32*c+32].t())
                print('   row', r0, 'slice', c, 'contribution norm %.3f' % part.norm().item(), ' err vec norm %.3f' % (z[r0].double() - zr[r0]).norm().item())
