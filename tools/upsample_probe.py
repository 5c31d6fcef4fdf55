#!/usr/bin/env python
"""`postprocess`'s bilinear upsample + threshold (output_utils.py:91-94) for a fixed-capacity batch: B*cap prototype-resolution masks ->
[B*cap, h, w] fp32, the 968 MB write stream of a batch-8 step.  Times ymi_mask_upsample_batch_f32 under the kernel variant
selected by YOLACT_AMD_UPSAMPLE (rows = default | rowsnt | band; read once per process, so run the script once per variant) and
prints a digest of the output bytes: the variants must print the SAME digest (bit-identical masks).

    YOLACT_AMD_UPSAMPLE=band python tools/upsample_probe.py [--batch 8] [--cap 100] [--size 550] [--proto 138]
"""
import argparse
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolact_amd import _lib as L        # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--cap', type=int, default=100)
    ap.add_argument('--size', type=int, default=550)
    ap.add_argument('--width', type=int, default=0)
    ap.add_argument('--proto', type=int, default=138)
    ap.add_argument('--reps', type=int, default=20)
    args = ap.parse_args()
    dev = 'cuda:0'
    lib = L.lib()
    h, w = args.size, args.width or args.size
    g = torch.Generator().manual_seed(3)
    lo = torch.rand(args.batch * args.cap, args.proto, args.proto, generator=g).to(dev)
    count = torch.full((args.batch,), args.cap, dtype=torch.int32, device=dev)
    out = torch.empty(args.batch * args.cap, h, w, device=dev)
    s = L.stream_ptr()

    def run(thresh):
        L.check(lib.ymi_mask_upsample_batch_f32(lo.data_ptr(), count.data_ptr(), out.data_ptr(), args.batch, args.cap, args.proto,
                                                args.proto, h, w, thresh, s))
    digs = []
    for thresh in (0.5, -1.0):
        out.fill_(float('nan'))
        run(thresh)
        torch.cuda.synchronize()
        digs.append(hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e30
    for _ in range(3):
        e0.record()
        for _ in range(args.reps):
            run(0.5)
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / args.reps)
    nbytes = out.numel() * 4
    fill = 1e30
    for _ in range(3):                      # reference point: torch's fill kernel over the same bytes (a pure write stream)
        e0.record()
        for _ in range(args.reps):
            out.zero_()
        e1.record()
        e1.synchronize()
        fill = min(fill, e0.elapsed_time(e1) / args.reps)
    print('   (torch zero_ of the same %d MB: %.4f ms  %.2f TB/s)' % (nbytes // 10 ** 6, fill, nbytes / fill / 1e9))
    print('variant %-7s %d masks %dx%d -> %dx%d: %.4f ms  %.2f TB/s  digest(binarised) %s  digest(soft) %s' % (
        os.environ.get('YOLACT_AMD_UPSAMPLE', 'rows'), args.batch * args.cap, args.proto, args.proto, h, w, best,
        nbytes / best / 1e9, digs[0], digs[1]))


if __name__ == '__main__':
    main()
