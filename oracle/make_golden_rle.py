"""Samples real pycocotools RLE strings written by the reference's own pipeline into tests/golden/rle_web.json.

web/dets/*.json in the reference were produced by `eval.py --output_web_json` (eval.py:342-371): every detection carries
the `segmentation` dict that Detections.add_mask built with pycocotools.mask.encode (eval.py:320-324) and the box that
add_bbox rounded (eval.py:306-318).  They are the only outputs of that third-party encoder available offline, so they
pin oracle/coco_rle.py.  Run in the build container:  python oracle/make_golden_rle.py
Checks EVERY string of the sampled files (decode -> size, inside-box, re-encode identical) and commits a small sample.
"""
import ast
import json
import os
import re
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import coco_rle as R  # noqa: E402

REF = '/root/reference/web/dets'
FILES = ['yolact_resnet50.json', 'yolact_base.json', 'yolact_im700.json', 'yolact_darknet53.json']


def check(det):
    h, w = det['mask']['size']
    s = det['mask']['counts']
    counts = R.rle_from_string(s)
    assert sum(counts) == h * w and all(c >= 0 for c in counts), (sum(counts), h, w)
    assert all(c > 0 for c in counts[1:]), 'only the first run may be empty'
    assert R.rle_to_string(counts) == s
    m = R.rle_decode(counts, h, w)
    assert R.rle_encode_counts(m) == counts
    ys, xs = np.nonzero(m)
    if len(xs):
        x, y, bw, bh = det['bbox']
        # masks are cropped to the box (+1 px) at PROTOTYPE resolution and then upsampled (box_utils.py:327-373 via
        # output_utils.py:69-99), so they may leak ~2 prototype pixels (138 per image side) past the rounded box
        t = 2.0 * max(h, w) / 138 + 2
        assert xs.min() >= x - t and xs.max() <= x + bw + t and ys.min() >= y - t and ys.max() <= y + bh + t, det['bbox']
    return len(counts), int(m.sum())


def main():
    rng = np.random.RandomState(7)
    sample, total = [], 0
    for f in FILES:
        d = json.load(open(os.path.join(REF, f)))
        dets = [(im['image_id'], det) for im in d['images'] for det in im['dets']]
        for _, det in dets:
            check(det)
        total += len(dets)
        for i in rng.choice(len(dets), 40, replace=False):
            image_id, det = dets[int(i)]
            nruns, area = check(det)
            sample.append({'file': f, 'image_id': image_id, 'bbox': det['bbox'], 'score': det['score'],
                           'size': det['mask']['size'], 'counts': det['mask']['counts'], 'nruns': nruns, 'area': area})
    # the category-id table get_coco_cat() inverts (data/config.py:46-55), read as data
    src = open('/root/reference/data/config.py').read()
    label_map = ast.literal_eval(re.search(r'COCO_LABEL_MAP\s*=\s*(\{.*?\})', src, re.S).group(1))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'rle_web.json')
    json.dump({'source': 'reference web/dets/*.json (eval.py --output_web_json with the official weights)',
               'checked_total': total, 'label_map': {str(k): v for k, v in label_map.items()}, 'sample': sample}, open(out, 'w'))
    print('checked %d reference RLE strings, wrote %d samples -> %s (%d bytes)' % (total, len(sample), out,
                                                                                os.path.getsize(out)))


if __name__ == '__main__':
    main()
