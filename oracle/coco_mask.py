"""CPU restatement of `COCO.annToMask` (pycocotools) as used by COCODetection.pull_item (data/coco.py:144-148).
*** TEST INFRASTRUCTURE ONLY ***

pycocotools is not in /root/reference (an unpinned pip dependency, environment.yml:30) and is not installed in this
image, so this restates its published algorithm (cocoapi `common/maskApi.c`, `PythonAPI/pycocotools/coco.py`,
`_mask.pyx`):

  annToRLE   segmentation is a list  -> one rleFrPoly per polygon, merged with rleMerge(intersect=0) = union
             segmentation['counts'] is a list -> uncompressed RLE (frUncompressedRLE)
             otherwise             -> compressed RLE string (rleFrString)
  annToMask  rleDecode -> uint8 [h,w] (pycocotools returns it Fortran-ordered; values identical)

  fr_poly    maskApi.c rleFrPoly: vertices scaled by 5 and rounded, every edge walked densely along its major axis,
             the crossings of x-grid lines collected as (column, first row) boundary points, sorted in column-major order
             and turned into run lengths (zero-length runs merged).

PARITY UNPINNED for fr_poly against pycocotools itself: the reference ships no polygon -> mask vectors and pycocotools
cannot be executed here.  What IS pinned: the RLE string codec and the decode (oracle/coco_rle.py against 13.5 k strings
the reference wrote, tests/golden/rle_web.json), and fr_poly's output against exact geometric properties
(tests/test_coco_dataset.py: axis-aligned rectangles and triangles with integer vertices have closed-form masks).
"""
from __future__ import annotations

import math
from typing import Dict, List

import numpy as np

from . import coco_rle


def fr_poly(xy: List[float], h: int, w: int) -> List[int]:
    """maskApi.c rleFrPoly -> run-length counts (column-major, starting with zeros)."""
    k = len(xy) // 2
    scale = 5.0
    x = [int(scale * xy[2 * j] + .5) for j in range(k)]
    y = [int(scale * xy[2 * j + 1] + .5) for j in range(k)]
    x.append(x[0])
    y.append(y[0])
    u, v = [], []
    for j in range(k):
        xs, xe, ys, ye = x[j], x[j + 1], y[j], y[j + 1]
        dx, dy = abs(xe - xs), abs(ys - ye)
        flip = (dx >= dy and xs > xe) or (dx < dy and ys > ye)
        if flip:
            xs, xe, ys, ye = xe, xs, ye, ys
        if dx >= dy:
            s = (ye - ys) / dx if dx else 0.0          # C: 0/0 -> nan, only multiplied by t = 0 when dx == 0 ... see note
            for d in range(dx + 1):
                t = dx - d if flip else d
                u.append(t + xs)
                v.append(int(ys + s * t + .5))
        else:
            s = (xe - xs) / dy
            for d in range(dy + 1):
                t = dy - d if flip else d
                v.append(t + ys)
                u.append(int(xs + s * t + .5))
    pts = []
    for j in range(1, len(u)):
        if u[j] != u[j - 1]:
            xd = float(u[j] if u[j] < u[j - 1] else u[j] - 1)
            xd = (xd + .5) / scale - .5
            if math.floor(xd) != xd or xd < 0 or xd > w - 1:
                continue
            yd = float(v[j] if v[j] < v[j - 1] else v[j - 1])
            yd = (yd + .5) / scale - .5
            if yd < 0:
                yd = 0.0
            elif yd > h:
                yd = float(h)
            yd = math.ceil(yd)
            pts.append(int(xd) * h + int(yd))
    pts.append(h * w)
    pts.sort()
    a = []
    p = 0
    for t in pts:
        a.append(t - p)
        p = t
    b = [a[0]]
    j = 1
    while j < len(a):
        if a[j] > 0:
            b.append(a[j])
            j += 1
        else:
            j += 1
            if j < len(a):
                b[-1] += a[j]
                j += 1
    return b


def ann_to_mask(ann: Dict, h: int, w: int) -> np.ndarray:
    seg = ann['segmentation']
    if isinstance(seg, list):
        m = np.zeros((h, w), dtype=np.uint8)
        for poly in seg:
            m |= coco_rle.rle_decode(fr_poly(poly, h, w), h, w).astype(np.uint8)
        return m
    counts = seg['counts']
    if isinstance(counts, list):
        return coco_rle.rle_decode(counts, h, w).astype(np.uint8)
    if isinstance(counts, bytes):
        counts = counts.decode('ascii')
    return coco_rle.rle_decode(coco_rle.rle_from_string(counts), h, w).astype(np.uint8)
