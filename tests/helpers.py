"""Shared test plumbing: golden fixtures, synthetic parameters, oracle runs, comparison helpers."""
from __future__ import annotations

import functools
import json
import os

import numpy as np
import torch

from yolact_amd.config import CONFIGS
from yolact_amd.utils.synth import synth_images, synth_state_dict

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
ALL_CASES = ['r50_dense', 'r50_sparse', 'r50_empty', 'r101_base', 'darknet53', 'im700', 'plus_r50', 'r50_cc', 'r50_few',
             'plus_base', 'im400', 'plus_r50_b2']      # round 5: yolact_plus_base (R101++, dcn_interval 3), yolact_im400, a YOLACT++ batch of 2
NOMASK_CASE = 'r50_nomask'                             # eval.py --detect: cfg.eval_mask_branch = False (zero coefficients, no prototypes)


@functools.lru_cache(maxsize=None)
def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + '.npz'))
    arrays = {k: z[k] for k in z.files}
    meta = json.loads(bytes(arrays.pop('meta')).decode())
    return meta, arrays


def case_cfg(meta):
    cfg = CONFIGS[meta['config']].copy()
    if 'eval_mask_branch' in meta:
        cfg.eval_mask_branch = bool(meta['eval_mask_branch'])
    return cfg


def case_state_dict(meta):
    shapes = [(k, tuple(s)) for k, s in meta['keys']]
    return synth_state_dict(shapes, seed=meta['seed'], conf_gain=meta['conf_gain'], bg_bias=meta.get('bg_bias', 0.0))


def case_images(meta):
    return synth_images(meta['B'], meta['size'], meta['size'], seed=1000 + meta['seed'])


@functools.lru_cache(maxsize=None)
def oracle_run(name):
    """(meta, arrays, raw head outputs, detections) of the CPU oracle on a golden case."""
    from oracle import yolact_oracle as O
    meta, arrays = load_golden(name)
    cfg = case_cfg(meta)
    sd = case_state_dict(meta)
    with torch.no_grad():
        raw = O.forward_raw(case_images(meta), sd, cfg)
        dets = O.detect(raw, cfg, cross_class=bool(meta.get('cross_class', False)))
    return meta, arrays, cfg, sd, raw, dets


def check_digest(t: torch.Tensor, meta, arrays, key, rtol=1e-4, atol=1e-4):
    """Compare a tensor against a golden digest: numel, sampled values, global sums."""
    d = meta['dg_' + key]
    flat = t.detach().float().cpu().contiguous().view(-1)
    assert flat.numel() == d['numel'], (key, flat.numel(), d['numel'])
    idx = torch.from_numpy(arrays['dg_%s_idx' % key])
    ref = torch.from_numpy(arrays['dg_%s_val' % key])
    got = flat[idx]
    scale = max(1.0, d['abssum'] / d['numel'])
    err = (got - ref).abs().max().item()
    assert err <= atol * scale + rtol * ref.abs().max().item(), '%s: sampled max err %g (scale %g)' % (key, err, scale)
    s, a = float(flat.double().sum()), float(flat.double().abs().sum())
    assert abs(a - d['abssum']) <= rtol * d['abssum'] + atol, '%s: abssum %r vs %r' % (key, a, d['abssum'])
    assert abs(s - d['sum']) <= rtol * d['abssum'] + atol, '%s: sum %r vs %r' % (key, s, d['sum'])


def unpack_masks(arrays, b, n, h, w):
    bits = np.unpackbits(arrays['post%d_maskbits' % b])[: n * h * w]
    return torch.from_numpy(bits.reshape(n, h, w).astype(np.float32))


def match_detections(got, ref, score_tol=1e-4, box_tol=1e-4, coef_tol=1e-4):
    """Compare two detection dicts (box/mask/class/score[/prior]).  Exact order is required except inside groups
    of *exactly tied* reference scores (the reference's sort is unstable; SURVEY hard part 3(iv)).
    Returns the list of mismatch descriptions (empty = match)."""
    problems = []
    n = ref['score'].shape[0]
    if got['score'].shape[0] != n:
        return ['count %d vs %d' % (got['score'].shape[0], n)]
    gs, rs = got['score'].float().cpu(), ref['score'].float().cpu()
    if (gs - rs).abs().max().item() > score_tol:
        problems.append('scores differ by %g' % (gs - rs).abs().max().item())
    gc, rc = got['class'].cpu().long(), ref['class'].cpu().long()
    gb, rb = got['box'].float().cpu(), ref['box'].float().cpu()
    gm, rm = got['mask'].float().cpu(), ref['mask'].float().cpu()
    i = 0
    while i < n:
        j = i + 1
        while j < n and rs[j] == rs[i]:
            j += 1
        if j - i == 1:
            if gc[i] != rc[i]:
                problems.append('class[%d] %d vs %d' % (i, gc[i], rc[i]))
            if (gb[i] - rb[i]).abs().max().item() > box_tol * max(1.0, rb[i].abs().max().item()):
                problems.append('box[%d] %s vs %s' % (i, gb[i].tolist(), rb[i].tolist()))
            if (gm[i] - rm[i]).abs().max().item() > coef_tol:
                problems.append('coef[%d] err %g' % (i, (gm[i] - rm[i]).abs().max().item()))
        else:   # tied group: compare as multisets of (class, box)
            a = sorted((int(gc[k]), tuple(round(v, 4) for v in gb[k].tolist())) for k in range(i, j))
            b = sorted((int(rc[k]), tuple(round(v, 4) for v in rb[k].tolist())) for k in range(i, j))
            if a != b:
                problems.append('tied group [%d,%d) differs' % (i, j))
        i = j
    return problems


def assert_margin_match(prior_idx, out, raw, dets, cfg, delta=1e-3, value_tol=1e-4):
    """End-to-end detection parity, margin-aware (SURVEY 7 hard part 3(ii), oracle/margins.py).

    `out`: the product's list of {'detection': ..}; `prior_idx`: Detect.last_prior_idx (prior index per returned
    detection); `raw` / `dets`: the oracle's head tensors and detections.  For every image
      * every (prior, class) decision of the reference whose margin exceeds `delta` is reproduced exactly
        (sure <= device output <= possible),
      * detections present on both sides agree in score / box / coefficients to `value_tol`,
      * the device scores are sorted descending.
    Returns a per-image summary (sure / possible / common counts) for the test log."""
    from oracle import margins as MG
    summary = []
    for b in range(len(out)):
        m = MG.detect_margins(raw['conf'][b], raw['loc'][b], raw['priors'], cfg.nms_conf_thresh, cfg.nms_thresh,
                              cfg.nms_top_k, cfg.max_num_detections, delta=delta, delta_iou=delta)
        MG.check_against_oracle(m, dets[b])
        g = out[b]['detection']
        if g is None:
            assert not m['sure'], 'image %d: device returned nothing, %d sure detections expected' % (b, len(m['sure']))
            summary.append((b, 0, len(m['possible']), 0))
            continue
        gp, gc = prior_idx[b].cpu().tolist(), g['class'].cpu().tolist()
        problems = MG.margin_match(gp, gc, m)
        assert not problems, 'image %d: %s\n%s' % (b, problems, MG.summarize(m))
        sc = g['score'].float().cpu()
        assert bool((sc[:-1] >= sc[1:]).all()), 'scores must be sorted descending'
        r = dets[b]
        ncommon = 0
        if r is not None:
            ridx = {pc: i for i, pc in enumerate(zip(r['prior'].tolist(), r['class'].tolist()))}
            pairs = [(i, ridx[pc]) for i, pc in enumerate(zip(gp, gc)) if pc in ridx]
            ncommon = len(pairs)
            if pairs:
                gi = torch.tensor([i for i, _ in pairs]); ri = torch.tensor([j for _, j in pairs])
                assert (sc[gi] - r['score'][ri]).abs().max().item() <= value_tol
                bscale = max(1.0, r['box'].abs().max().item())
                assert (g['box'].float().cpu()[gi] - r['box'][ri]).abs().max().item() <= value_tol * bscale
                assert (g['mask'].float().cpu()[gi] - r['mask'][ri]).abs().max().item() <= value_tol
            if len(m['possible']) == len(m['sure']):          # nothing undecidable: the outputs must be the same set
                assert ncommon == r['score'].shape[0] == len(gp)
        summary.append((b, len(m['sure']), len(m['possible']), ncommon))
    return summary
