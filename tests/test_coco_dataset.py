"""COCODetection's annotation half (data/coco.py:13-176, pycocotools annToMask) — SURVEY §8(f) rank 4, host code only.

pycocotools is neither in /root/reference nor installed: its maskApi.c is restated twice, independently — plain Python in
oracle/coco_mask.py and C++ in csrc/coco_host.cpp — and compared bit for bit on random and degenerate polygons; closed-form
known answers (integer rectangles cover x0 <= x < x1, y0 <= y < y1 = pycocotools' `area == w*h` behaviour; the full-image
polygon; empty / outside polygons) pin the conventions; the RLE forms reuse the codec that IS pinned to 13.5 k
reference-written strings (tests/test_coco_rle.py).  The index class is compared with the reference's access pattern on a
synthetic annotation file (tests/coco_synth.py).  The image half + the end-to-end pull_item: tests/test_gpu_jpeg.py.
"""
import json

import numpy as np
import pytest

from oracle import coco_mask, coco_rle
from oracle import coco_dataset as OD
from tests import coco_synth
from yolact_amd.data import COCOAnnotationTransform, COCOIndex, ann_to_mask
from yolact_amd.coco import COCO_LABEL_MAP


def poly_mask(poly, h, w):
    return ann_to_mask({'segmentation': [poly]}, h, w)


def test_rectangles_have_the_closed_form_mask():
    h, w = 23, 31
    for (x0, y0, x1, y1) in [(2, 1, 7, 5), (0, 0, 31, 23), (30, 22, 31, 23), (5, 0, 6, 23), (0, 7, 31, 8)]:
        want = np.zeros((h, w), dtype=np.uint8)
        want[y0:y1, x0:x1] = 1
        for poly in ([x0, y0, x1, y0, x1, y1, x0, y1], [x0, y0, x0, y1, x1, y1, x1, y0]):      # both orientations
            assert np.array_equal(poly_mask(poly, h, w), want), (x0, y0, x1, y1)
            assert np.array_equal(coco_rle.rle_decode(coco_mask.fr_poly(poly, h, w), h, w), want)
    assert poly_mask([40, 40, 50, 40, 50, 50], h, w).sum() == 0                                  # entirely outside
    assert poly_mask([3, 3], h, w).sum() == 0                                                    # a single point


def test_native_rasteriser_matches_the_python_restatement():
    rng = np.random.default_rng(11)
    for trial in range(300):
        h, w = int(rng.integers(1, 70)), int(rng.integers(1, 70))
        k = int(rng.integers(1, 12))
        if trial % 3 == 0:      # integer / half-integer vertices, some outside the image
            poly = (rng.integers(-6, max(h, w) + 6, 2 * k) / rng.choice([1, 2])).tolist()
        else:
            poly = (rng.random(2 * k) * (max(h, w) + 10) - 5).round(2).tolist()
        if trial % 7 == 0 and k > 1:
            poly[2:4] = poly[0:2]                                                                # repeated vertex
        want = coco_rle.rle_decode(coco_mask.fr_poly(poly, h, w), h, w)
        assert np.array_equal(poly_mask(poly, h, w), want), (trial, h, w, poly)


def test_annotation_forms_union_rle_list_rle_string():
    h, w = 40, 30
    a = [3, 3, 20, 4, 18, 30], [10, 10, 29, 12, 25, 39, 8, 35]
    both = ann_to_mask({'segmentation': list(a)}, h, w)
    assert np.array_equal(both, poly_mask(a[0], h, w) | poly_mask(a[1], h, w))
    assert np.array_equal(both, coco_mask.ann_to_mask({'segmentation': list(a)}, h, w))
    counts = coco_rle.rle_encode_counts(both)
    for seg in ({'size': [h, w], 'counts': [int(c) for c in counts]}, {'size': [h, w], 'counts': coco_rle.rle_to_string(counts)},
                {'size': [h, w], 'counts': coco_rle.rle_to_string(counts).encode('ascii')}):
        assert np.array_equal(ann_to_mask({'segmentation': seg}, h, w), both)
    with pytest.raises(RuntimeError):
        ann_to_mask({'segmentation': {'size': [h, w], 'counts': [5, 5]}}, h, w)                  # runs do not cover h*w
    with pytest.raises(ValueError):
        ann_to_mask({'segmentation': {'size': [h + 1, w], 'counts': [h * w]}}, h, w)
    with pytest.raises(ValueError):
        ann_to_mask({'segmentation': [[1, 2, 3]]}, h, w)


def test_index_and_target_transform_follow_the_reference(tmp_path):
    info = coco_synth.write_dataset(str(tmp_path))
    coco = COCOIndex(info)
    ds = json.load(open(info))
    assert list(coco.imgToAnns.keys()) == [139, 285, 632, 724]              # insertion order = first annotation of each image
    assert set(coco.imgs.keys()) == {139, 285, 632, 724, 785}
    ids = coco.getAnnIds(imgIds=139)
    assert ids == [a['id'] for a in ds['annotations'] if a['image_id'] == 139]
    assert [a['id'] for a in coco.loadAnns(ids)] == ids and coco.loadImgs(285)[0]['file_name'].startswith('COCO_val2014')
    assert coco.getAnnIds(imgIds=785) == []
    for a in ds['annotations']:
        t = coco.imgs[a['image_id']]
        assert np.array_equal(coco.annToMask(a), coco_mask.ann_to_mask(a, t['height'], t['width'])), a['id']
    tt = COCOAnnotationTransform()
    target = [dict(a) for a in ds['annotations'] if a['image_id'] == 139]
    target[0]['category_id'] = -1                                             # what pull_item does to crowds
    got = tt(target, 35, 50)
    want = OD.annotation_transform(target, 35, 50, COCO_LABEL_MAP)
    assert got == want and got[0][4] == -1 and got[1][4] == COCO_LABEL_MAP[18] - 1
    assert got[1][:4] == [2.5 / 35, 3.25 / 50, 22.5 / 35, 33.75 / 50]


def test_cv2_resize_restatement_is_the_half_pixel_bilinear():
    """The oracle's cv2.resize restatement against torch's F.interpolate(align_corners=False): same formula, the sample
    coordinate rounded differently -> a few 1e-5 of the 0..255 range at most."""
    import torch
    rng = np.random.default_rng(3)
    for (h, w, s) in [(37, 53, 64), (64, 48, 550), (5, 7, 20), (480, 640, 550)]:
        img = rng.integers(0, 256, (h, w, 3)).astype(np.float32)
        got = OD.cv2_resize_linear_f32(img, s, s)
        ref = torch.nn.functional.interpolate(torch.from_numpy(img).permute(2, 0, 1)[None], (s, s), mode='bilinear',
                                              align_corners=False)[0].permute(1, 2, 0).numpy()
        assert got.shape == (s, s, 3)
        # white-noise input = the worst case: |d coord| <= ~6e-5 px times a full-scale step between neighbours
        assert np.abs(got - ref).max() < 5e-2, np.abs(got - ref).max()
        assert np.abs(got - ref).mean() < 5e-3
