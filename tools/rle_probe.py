#!/usr/bin/env python
"""Throughput of the device RLE encoder (csrc/rle.hip) on postprocess-shaped masks: N x 550 x 550 float32 blobs.
   python tools/rle_probe.py [N] [h] [w]"""
import sys, time
import torch
sys.path.insert(0, '.')
from yolact_amd import _lib as L
from yolact_amd.coco import rle_encode

N, h, w = (int(a) for a in (sys.argv[1:4] + ['800', '550', '550'][len(sys.argv) - 1:]))
g = torch.Generator(device='cuda').manual_seed(0)
yy, xx = torch.meshgrid(torch.arange(h, device='cuda'), torch.arange(w, device='cuda'), indexing='ij')
cx, cy = torch.rand(N, generator=g, device='cuda') * w, torch.rand(N, generator=g, device='cuda') * h
r = 20 + torch.rand(N, generator=g, device='cuda') * 120
masks = (((xx[None] - cx[:, None, None]) ** 2 + (yy[None] - cy[:, None, None]) ** 2) < r[:, None, None] ** 2).float().contiguous()
lib = L.lib()
cap = 4096
counts = torch.empty(N, cap, dtype=torch.int32, device='cuda'); nruns = torch.empty(N, dtype=torch.int32, device='cuda')
text = torch.empty(N, 7 * cap, dtype=torch.uint8, device='cuda'); nchars = torch.empty(N, dtype=torch.int32, device='cuda')
s = L.stream_ptr()
e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
for it in range(3):
    e0.record()
    for _ in range(10):
        lib.ymi_mask_rle_f32(masks.data_ptr(), N, h, w, counts.data_ptr(), nruns.data_ptr(), cap, s)
    e1.record()
    for _ in range(10):
        lib.ymi_rle_to_string(counts.data_ptr(), nruns.data_ptr(), N, cap, text.data_ptr(), nchars.data_ptr(), 7 * cap, s)
    e2.record(); e2.synchronize()
ms_c, ms_s = e0.elapsed_time(e1) / 10, e1.elapsed_time(e2) / 10
byts = N * h * w * 4
print('rle_counts_k : %d masks %dx%d  %.3f ms  %.2f TB/s (algorithmic %d MB)  mean runs %.0f' % (N, h, w, ms_c, byts / ms_c / 1e9, byts // 10**6, nruns.float().mean()))
print('rle_string_k : %.3f ms   mean chars %.0f' % (ms_s, nchars.float().mean()))
torch.cuda.synchronize(); t = time.time(); out = rle_encode(masks); dt = time.time() - t
print('rle_encode (device + string copy + python dicts): %.2f ms for %d masks = %.0f masks/s' % (dt * 1e3, N, N / dt))
t = time.time(); mh = masks[:100].cpu().numpy(); dt_copy = time.time() - t
sys.path.insert(0, '.')
from oracle import coco_rle as R
t = time.time(); ref = [R.encode(m) for m in mh[:20]]; dt_cpu = (time.time() - t) / 20
assert ref == out[:20]
print('reference-style host path: copy %.2f ms/mask + numpy oracle encode %.2f ms/mask' % (dt_copy * 1e3 / 100, dt_cpu * 1e3))
