"""COCO result wire format (eval.py:300-340 Detections, pycocotools RLE).

CPU: the restated codec (oracle/coco_rle.py) against REAL pycocotools strings the reference's pipeline wrote
(tests/golden/rle_web.json, sampled from /root/reference/web/dets/*.json by oracle/make_golden_rle.py; all 13 525 strings
of those files were checked when the fixture was generated), hand-derived known answers and edge cases.
GPU: ymi_mask_rle_f32 + ymi_rle_to_string through the C ABI: byte-identical strings for the golden masks, random and
degenerate masks, capacity overflow, and the Detections records of a full postprocess output.  Integer / byte work:
the bar is bit-exact.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import coco_rle as R

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, 'golden', 'rle_web.json')))


# ---------------------------------------------------------------------------------------------- CPU: the oracle
def test_oracle_roundtrips_reference_strings():
    assert GOLD['checked_total'] >= 10000 and len(GOLD['sample']) == 160
    for s in GOLD['sample']:
        h, w = s['size']
        counts = R.rle_from_string(s['counts'])
        assert len(counts) == s['nruns'] and sum(counts) == h * w
        assert R.rle_to_string(counts) == s['counts']
        m = R.rle_decode(counts, h, w)
        assert int(m.sum()) == s['area']
        assert R.encode(m) == {'size': [h, w], 'counts': s['counts']}


def test_oracle_known_answers():
    # column-major flattening of [[0,1],[1,1]] is 0,1,1,1 -> runs 1,3
    assert R.rle_encode_counts(np.array([[0, 1], [1, 1]])) == [1, 3]
    assert R.rle_to_string([1, 3]) == '13'
    # a mask that starts with foreground gets an empty first run
    assert R.rle_encode_counts(np.array([[1, 0], [1, 0]])) == [0, 2, 2]
    assert R.rle_encode_counts(np.zeros((3, 5))) == [15]
    assert R.rle_encode_counts(np.ones((3, 5))) == [0, 15]
    # 5-bit groups: 31 = 0b11111 needs a second group because bit 0x10 of the first would read as a sign
    assert R.rle_to_string([31]) == chr(48 + (31 | 0x20)) + chr(48)
    assert R.rle_to_string([16]) == chr(48 + (16 | 0x20)) + chr(48)
    assert R.rle_to_string([15]) == chr(48 + 15)
    # deltas against counts[i-2] from the fourth count on, negative deltas sign-extend
    s = R.rle_to_string([5, 40, 7, 3, 100])
    assert R.rle_from_string(s) == [5, 40, 7, 3, 100]
    assert s[-1] != R.rle_to_string([100])[-1] or len(s) > 0


@pytest.mark.parametrize('shape', [(1, 1), (1, 7), (7, 1), (5, 3), (64, 64), (37, 129)])
def test_oracle_random_roundtrip(shape):
    g = np.random.RandomState(shape[0] * 131 + shape[1])
    for p in (0.02, 0.5, 0.98):
        m = (g.rand(*shape) < p).astype(np.float32)
        c = R.rle_encode_counts(m)
        assert sum(c) == m.size and all(v > 0 for v in c[1:])
        assert np.array_equal(R.rle_decode(R.rle_from_string(R.rle_to_string(c)), *shape), m.astype(np.uint8))


def test_detections_bbox_records_match_oracle():
    from yolact_amd.coco import COCO_LABEL_MAP, Detections
    assert {int(k): v for k, v in GOLD['label_map'].items()} == COCO_LABEL_MAP       # data/config.py:46-55
    d = Detections(label_map=COCO_LABEL_MAP)
    inv = {v - 1: k for k, v in COCO_LABEL_MAP.items()}
    g = np.random.RandomState(3)
    for _ in range(50):
        x1, y1 = g.rand(2) * 300
        box = np.array([x1, y1, x1 + g.rand() * 200, y1 + g.rand() * 200], dtype=np.float32)
        cls, sc = int(g.randint(80)), float(np.float32(g.rand()))
        d.add_bbox(17, cls, box, sc)
        assert d.bbox_data[-1] == R.bbox_record(17, cls, box, sc, inv)
    assert d.bbox_data[0]['category_id'] in COCO_LABEL_MAP
    with pytest.raises(TypeError):
        d.add_mask(1, 0, np.zeros((4, 4)), 0.5)


# ---------------------------------------------------------------------------------------------- GPU
def _gpu_encode(masks_np, cap=4096):
    from yolact_amd.coco import rle_counts, rle_encode
    m = torch.from_numpy(np.ascontiguousarray(masks_np, dtype=np.float32)).cuda()
    return rle_encode(m, cap), rle_counts(m, cap)


@pytest.mark.gpu
def test_gpu_rle_matches_reference_strings():
    by_size = {}
    for s in GOLD['sample']:
        by_size.setdefault(tuple(s['size']), []).append(s)
    assert len(by_size) > 5
    for (h, w), items in by_size.items():
        masks = np.stack([R.rle_decode(R.rle_from_string(s['counts']), h, w) for s in items])
        enc, cnt = _gpu_encode(masks)
        for s, e, c in zip(items, enc, cnt):
            assert e == {'size': [h, w], 'counts': s['counts']}
            assert c == R.rle_from_string(s['counts'])


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(1, 1), (1, 9), (9, 1), (2, 70), (3, 3), (5, 64), (63, 65), (138, 138), (550, 550),
                                   (427, 640), (700, 700), (33, 2048)])
def test_gpu_rle_random_and_degenerate(shape):
    h, w = shape
    g = np.random.RandomState(h * 7 + w)
    ms = [np.zeros(shape), np.ones(shape)]
    one = np.zeros(shape); one[h - 1, w - 1] = 1; ms.append(one)
    first = np.zeros(shape); first[0, 0] = 1; ms.append(first)
    for p in (0.001, 0.3, 0.999):
        ms.append(g.rand(*shape) < p)
    blob = np.zeros(shape); blob[h // 4:h // 4 + max(1, h // 2), w // 3:w // 3 + max(1, w // 2)] = 1; ms.append(blob)
    stripes = np.zeros(shape); stripes[::2, :] = 1; ms.append(stripes)           # a transition at every element
    masks = np.stack(ms).astype(np.float32)
    masks[6] *= 3.5                                                              # any nonzero value is foreground
    enc, cnt = _gpu_encode(masks, cap=64)                                        # forces the capacity retry
    for m, e, c in zip(masks, enc, cnt):
        assert c == R.rle_encode_counts(m)
        assert e == R.encode(m)


@pytest.mark.gpu
def test_gpu_rle_rejects_bad_arguments():
    from yolact_amd import _lib as L
    lib = L.lib()
    assert lib.ymi_mask_rle_f32(None, 1, 4, 4, None, None, 16, None) == -3
    assert lib.ymi_mask_rle_f32(None, 0, 4, 4, None, None, 16, None) == 0          # empty batch: nothing to do
    assert lib.ymi_mask_rle_f32(None, 1, 4, 4096, None, None, 16, None) in (-2, -3)
    assert lib.ymi_rle_to_string(None, None, 1, 16, None, None, 16, None) == -3
    from yolact_amd.coco import rle_encode
    assert rle_encode(torch.zeros(0, 5, 5, device='cuda')) == []
    with pytest.raises(RuntimeError):
        rle_encode(torch.zeros(1, 5, 5))                                          # CPU tensor: no fallback


@pytest.mark.gpu
@pytest.mark.parametrize('name,b', [('r50_dense', 0), ('r50_dense', 1), ('im700', 0), ('plus_r50', 0)])
def test_gpu_detections_records_of_reference_postprocess_outputs(name, b):
    """Detections.add_image fed the EXECUTED reference's postprocess output (tests/golden/<case>.npz) == the restated
    add_bbox / add_mask on the same arrays on the host (what eval.py:420-429 does)."""
    from helpers import load_golden
    from yolact_amd.coco import COCO_LABEL_MAP, Detections
    meta, arrays = load_golden(name)
    w, h = meta['post']
    cl = arrays['post%d_class' % b]
    sc = arrays['post%d_score' % b]
    bx = arrays['post%d_box' % b]
    n = cl.shape[0]
    assert n > 0
    mk = np.unpackbits(arrays['post%d_maskbits' % b])[: n * h * w].reshape(n, h, w).astype(np.float32)
    box_sc = sc
    mask_sc = arrays['post%d_score2' % b] if ('post%d_score2' % b) in arrays else sc      # YOLACT++: rescored masks
    d = Detections(label_map=COCO_LABEL_MAP)
    d.add_image(42, torch.from_numpy(cl), torch.from_numpy(bx), torch.from_numpy(box_sc), torch.from_numpy(mk).cuda(),
                torch.from_numpy(mask_sc))
    inv = {v - 1: k for k, v in COCO_LABEL_MAP.items()}
    keep = [i for i in range(n) if (bx[i, 3] - bx[i, 1]) * (bx[i, 2] - bx[i, 0]) > 0]
    assert len(d.bbox_data) == len(d.mask_data) == len(keep) > 0
    for k, i in enumerate(keep):
        assert d.bbox_data[k] == R.bbox_record(42, cl[i], bx[i], float(box_sc[i]), inv)
        assert d.mask_data[k] == {'image_id': 42, 'category_id': inv[int(cl[i])], 'segmentation': R.encode(mk[i]),
                                  'score': float(mask_sc[i])}
    json.dumps(d.mask_data)                                                       # json.dump-able like the reference's


# ------------------------------------------------------------------------------- GPU: upsample + threshold + RLE fused
def _smooth_lowres(n, ph, pw, seed):
    """Sigmoid-like low-resolution masks with structure at every scale (blobs, thin lines, values close to 0.5), cropped
    to boxes like postprocess does (zeros outside)."""
    g = torch.Generator().manual_seed(seed)
    m = torch.rand(n, ph, pw, generator=g)
    m = torch.nn.functional.avg_pool2d(m[:, None], 5, 1, 2)[:, 0] * 2.2 - 0.6          # smooth field around the threshold
    m = torch.sigmoid(6 * (m - 0.5))
    for i in range(n):
        y0, x0 = int(torch.randint(0, ph // 2 + 1, (1,), generator=g)), int(torch.randint(0, pw // 2 + 1, (1,), generator=g))
        y1, x1 = min(ph, y0 + 1 + int(torch.randint(0, ph, (1,), generator=g))), min(pw, x0 + 1 + int(torch.randint(0, pw, (1,), generator=g)))
        crop = torch.zeros(ph, pw)
        crop[y0:y1, x0:x1] = 1
        m[i] *= crop
    m[0] = 0.0                                                                         # an empty mask
    if n > 1:
        m[1] = 1.0                                                                     # a full one
    return m


@pytest.mark.gpu
@pytest.mark.parametrize('lo,hi', [((138, 138), (550, 550)), ((138, 138), (480, 640)), ((176, 176), (700, 700)),
                                   ((7, 5), (3, 9)), ((20, 30), (20, 30)), ((138, 138), (97, 2000)), ((3, 3), (1, 1))])
def test_gpu_fused_upsample_rle_equals_upsample_then_rle(lo, hi):
    """ymi_mask_rle_upsampled_f32 (masks never materialised) == ymi_mask_upsample_f32 + ymi_mask_rle_f32, byte for byte, and
    == the host codec applied to the device's upsampled masks."""
    import ctypes as C
    from yolact_amd import _lib as L
    from yolact_amd.coco import rle_encode, rle_encode_lowres
    (ph, pw), (h, w) = lo, hi
    n = 12
    mlo = _smooth_lowres(n, ph, pw, seed=ph * 1000 + w).cuda()
    full = torch.empty(n, h, w, device='cuda')
    L.check(L.lib().ymi_mask_upsample_f32(mlo.data_ptr(), full.data_ptr(), n, ph, pw, h, w, C.c_float(0.5), L.stream_ptr()))
    want = rle_encode(full)
    got = rle_encode_lowres(mlo, h, w, 0.5, cap=8)             # tiny initial capacity: exercises the retry path too
    assert got == want
    host = full.cpu().numpy()
    for i in (0, 1, n - 1):
        assert got[i] == R.encode(host[i])
    assert got[0]['counts'] == R.rle_to_string([h * w]) and got[1]['counts'] == R.rle_to_string([0, h * w])


@pytest.mark.gpu
def test_gpu_fused_rle_rejects_bad_arguments():
    from yolact_amd import _lib as L
    lib = L.lib()
    m = torch.zeros(2, 8, 8, device='cuda')
    c = torch.zeros(2, 16, dtype=torch.int32, device='cuda')
    nr = torch.zeros(2, dtype=torch.int32, device='cuda')
    import ctypes as C
    half = C.c_float(0.5)
    assert lib.ymi_mask_rle_upsampled_f32(m.data_ptr(), 2, 8, 8, 0, 16, half, c.data_ptr(), nr.data_ptr(), 16, None) == -1
    assert lib.ymi_mask_rle_upsampled_f32(None, 2, 8, 8, 16, 16, half, c.data_ptr(), nr.data_ptr(), 16, None) == -3
    assert lib.ymi_mask_rle_upsampled_f32(m.data_ptr(), 2, 8, 8, 16, 4096, half, c.data_ptr(), nr.data_ptr(), 16, None) == -2
    assert lib.ymi_mask_rle_upsampled_f32(m.data_ptr(), 0, 8, 8, 16, 16, half, c.data_ptr(), nr.data_ptr(), 16, None) == 0


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['r50_dense', 'plus_r50'])
def test_gpu_postprocess_rle_equals_postprocess_then_encode(name):
    """postprocess_rle (the COCO result path without full-resolution masks) == rle_encode(postprocess(...)) on the
    detections of a golden case (incl. YOLACT++ re-scoring); Detections.add_records == Detections.add_image."""
    import yolact_amd
    from gpu_utils import build_net
    from helpers import oracle_run
    from yolact_amd.coco import COCO_LABEL_MAP, Detections, rle_encode
    from yolact_amd.layers.output_utils import postprocess, postprocess_rle
    meta, arrays, cfg, sd, raw, dets = oracle_run(name)
    yolact_amd.set_cfg(meta['config'])
    net = build_net(meta)
    w, h = meta['post']
    done = 0
    for ref in dets:
        if ref is None:
            continue
        mk = lambda: [{'detection': {k: ref[k].cuda().clone() for k in ('box', 'mask', 'class', 'score', 'proto')}, 'net': net}]  # noqa: E731
        classes, scores, boxes, masks = postprocess(mk(), w, h, score_threshold=0.1)
        c2, s2, b2, rles = postprocess_rle(mk(), w, h, score_threshold=0.1, fused=True)
        assert postprocess_rle(mk(), w, h, score_threshold=0.1)[3] == rles          # the default (two kernels) gives the same records
        assert torch.equal(classes, c2) and torch.equal(boxes, b2)
        if isinstance(scores, list):
            assert all(torch.equal(a, b_) for a, b_ in zip(scores, s2))
            box_sc, mask_sc = scores
        else:
            assert torch.equal(scores, s2)
            box_sc = mask_sc = scores
        assert len(rles) == masks.shape[0] > 0 and rles == rle_encode(masks)
        d1, d2 = Detections(label_map=COCO_LABEL_MAP), Detections(label_map=COCO_LABEL_MAP)
        d1.add_image(7, classes, boxes, box_sc, masks, mask_sc)
        d2.add_records(7, c2, b2, box_sc, rles, mask_sc)
        assert d1.bbox_data == d2.bbox_data and d1.mask_data == d2.mask_data and len(d1.mask_data) > 0
        done += 1
    assert done > 0
    assert postprocess_rle([{'detection': None, 'net': net}], w, h) == ([], [], [], [])
