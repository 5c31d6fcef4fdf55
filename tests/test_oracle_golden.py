"""Pins the CPU oracle (oracle/yolact_oracle.py) against outputs of the reference itself (tests/golden/, written
by oracle/make_golden.py from /root/reference).  CPU only."""
import pytest
import torch

from helpers import ALL_CASES, NOMASK_CASE, check_digest, load_golden, match_detections, oracle_run, unpack_masks


@pytest.mark.parametrize('name', ALL_CASES)
def test_forward_stages_match_reference(name):
    meta, arrays, cfg, sd, raw, dets = oracle_run(name)
    for k, t in raw['stages'].items():
        check_digest(t.permute(0, 2, 3, 1), meta, arrays, k, rtol=2e-5, atol=2e-5)
    for k in ('loc', 'conf', 'mask', 'priors', 'proto'):
        check_digest(raw[k], meta, arrays, k, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize('name', ALL_CASES)
def test_detect_matches_reference(name):
    meta, arrays, cfg, sd, raw, dets = oracle_run(name)
    for b in range(meta['B']):
        n = meta['n'][b]
        if n == 0:
            assert dets[b] is None
            continue
        ref = {k: torch.from_numpy(arrays['det%d_%s' % (b, k)]) for k in ('box', 'mask', 'class', 'score')}
        problems = match_detections(dets[b], ref, score_tol=1e-6, box_tol=1e-6, coef_tol=1e-6)
        assert not problems, problems[:5]


@pytest.mark.parametrize('name', ALL_CASES)
def test_postprocess_matches_reference(name):
    from oracle import yolact_oracle as O
    meta, arrays, cfg, sd, raw, dets = oracle_run(name)
    w, h = meta['post']
    thr = meta.get('score_threshold', 0.0)          # r50_few: the display threshold (output_utils.py:42-50)
    for b in range(meta['B']):
        n_post = meta.get('n_post', meta['n'])[b]
        if n_post == 0:
            assert O.postprocess(dets[b], w, h, cfg, sd, score_threshold=thr) is None
            continue
        classes, scores, boxes, masks, soft = O.postprocess(dets[b], w, h, cfg, sd, score_threshold=thr, return_soft=True)
        assert classes.shape[0] == n_post
        assert torch.equal(classes, torch.from_numpy(arrays['post%d_class' % b]))
        assert torch.equal(boxes, torch.from_numpy(arrays['post%d_box' % b]))
        if isinstance(scores, list):
            assert torch.allclose(scores[0], torch.from_numpy(arrays['post%d_score' % b]), atol=1e-6)
            assert torch.allclose(scores[1], torch.from_numpy(arrays['post%d_score2' % b]), atol=1e-5)
        else:
            assert torch.allclose(scores, torch.from_numpy(arrays['post%d_score' % b]), atol=1e-6)
        ref = unpack_masks(arrays, b, n_post, h, w)
        bad = (masks != ref)
        # a binarised pixel may only flip where the soft value sits on the 0.5 threshold
        assert (soft[bad] - 0.5).abs().max().item() < 1e-5 if bad.any() else True
        assert bad.float().mean().item() < 1e-5


def test_sparse_case_is_sparse():
    """r50_few is the 'pretrained-like' regime SURVEY 8(d) asks for: ~1 % of the priors pass the candidate threshold and only
    a handful of detections clear the display threshold (the reference's own postprocess count is in the fixture)."""
    meta, arrays, cfg, sd, raw, dets = oracle_run('r50_few')
    for b in range(meta['B']):
        k = int((raw['conf'][b][:, 1:].max(1)[0] > cfg.nms_conf_thresh).sum())
        assert 0.003 * raw['conf'].shape[1] < k < 0.02 * raw['conf'].shape[1], k
        assert 1 <= meta['n_post'][b] <= 20, meta['n_post']
        assert int((dets[b]['score'] > meta['score_threshold']).sum()) == meta['n_post'][b]


def test_cross_class_case_is_the_references_cc_fast_nms():
    """r50_cc was produced by the reference with detect.use_cross_class_nms = True (eval.py:872): more than max_num_detections
    rows come back (cc_fast_nms does not truncate beyond top_k, detection.py:111-135) and the class column is not sorted by
    class-major order."""
    meta, arrays, cfg, sd, raw, dets = oracle_run('r50_cc')
    assert meta['cross_class'] is True
    for b in range(meta['B']):
        assert cfg.max_num_detections < meta['n'][b] <= cfg.nms_top_k
        assert dets[b]['score'].shape[0] == meta['n'][b]


def test_detect_only_mode_matches_reference():
    """r50_nomask: the reference run with cfg.eval_mask_branch = False (what eval.py --detect sets, eval.py:1067-1068): the heads
    return ZERO coefficients (yolact.py:172-175), no prototypes are computed (:579-580), the detection dicts carry no 'proto'
    (detection.py:73-74) and postprocess returns classes / scores / integer boxes with the coefficient rows as its 4th value
    (output_utils.py:58,97-122).  Pins that branch of the oracle."""
    from oracle import yolact_oracle as O
    meta, arrays, cfg, sd, raw, dets = oracle_run(NOMASK_CASE)
    assert meta['eval_mask_branch'] is False and cfg.eval_mask_branch is False
    assert raw['proto'] is None and 'dg_proto' not in meta
    for k, t in raw['stages'].items():
        check_digest(t.permute(0, 2, 3, 1), meta, arrays, k, rtol=2e-5, atol=2e-5)
    for k in ('loc', 'conf', 'priors'):
        check_digest(raw[k], meta, arrays, k, rtol=2e-5, atol=2e-5)
    assert meta['dg_mask']['abssum'] == 0.0 and float(raw['mask'].abs().sum()) == 0.0
    w, h = meta['post']
    for b in range(meta['B']):
        assert 'proto' not in dets[b]
        ref = {k: torch.from_numpy(arrays['det%d_%s' % (b, k)]) for k in ('box', 'mask', 'class', 'score')}
        assert not match_detections(dets[b], ref, score_tol=1e-6, box_tol=1e-6, coef_tol=1e-6)
        classes, scores, boxes, masks = O.postprocess(dets[b], w, h, cfg, sd)
        assert torch.equal(classes, torch.from_numpy(arrays['post%d_class' % b]))
        assert torch.equal(boxes, torch.from_numpy(arrays['post%d_box' % b]))
        assert torch.allclose(scores, torch.from_numpy(arrays['post%d_score' % b]), atol=1e-6)
        assert torch.equal(masks, torch.from_numpy(arrays['post%d_maskraw' % b])) and masks.shape == (meta['n'][b], 32)
