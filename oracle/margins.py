"""Margin analysis of the reference's Detect decisions.  *** TEST INFRASTRUCTURE ONLY ***

`Detect` (layers/functions/detection.py:81-180) is a chain of hard decisions on fp32 scores and IoUs:

  (1) candidate filter   max_c score[c,p] > conf_thresh                      detection.py:83-86
  (2) per-class top-k    rank of score[c,p] among the candidates < top_k     detection.py:138-141
  (3) Fast NMS           max_{i ranked above j} IoU(i,j) <= nms_thresh       detection.py:143-158
  (4) final cut          rank of the survivor's score over all classes < max_num_detections   detection.py:172-174

A second implementation whose head tensors differ from the reference's by 1e-6..1e-4 cannot reproduce decisions whose
margin is smaller than that drift (hard part 3: on the dense synthetic recipe the smallest margins are 3e-7),
and it MUST reproduce every decision whose margin is larger.  This module makes that statement executable: from the
ORACLE's scores and boxes it derives, per image,

  sure      (prior, class) pairs that end up in the output under EVERY perturbation of the scores by < delta and of the
            IoUs by < delta_iou;
  possible  pairs that end up in the output under SOME such perturbation.

By construction  sure  <=  reference output  <=  possible  (checked here against the oracle's own detect_image), and
the parity tests require  sure <= device output <= possible.  Everything in possible - sure is reported with the margin
that makes it undecidable.

The bounds are conservative interval arithmetic over the decision chain (a pair is "sure" only if every decision on its
path is sure, using the widest set of rivals that could be ranked above it; "possible" uses the narrowest).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch

from . import yolact_oracle as O

Tensor = torch.Tensor


def _iou_matrix(b: np.ndarray) -> np.ndarray:
    """box_utils.py:47-51,72-79 in float64 (the margins absorb the fp32 / fp64 difference)."""
    x1, y1, x2, y2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    iw = np.clip(np.minimum(x2[:, None], x2[None, :]) - np.maximum(x1[:, None], x1[None, :]), 0, None)
    ih = np.clip(np.minimum(y2[:, None], y2[None, :]) - np.maximum(y1[:, None], y1[None, :]), 0, None)
    inter = iw * ih
    area = (x2 - x1) * (y2 - y1)
    return inter / (area[:, None] + area[None, :] - inter)


def _count_greater(sorted_desc: np.ndarray, v: np.ndarray) -> np.ndarray:
    """#elements of `sorted_desc` (descending) that are > v, elementwise."""
    asc = sorted_desc[::-1]
    return len(asc) - np.searchsorted(asc, v, side='right')


def detect_margins(conf: Tensor, loc: Tensor, priors: Tensor, conf_thresh=0.05, nms_thresh=0.5, top_k=200, max_det=100,
                   delta=1e-3, delta_iou=1e-3) -> Dict:
    """conf [P,C] post-softmax, loc [P,4], priors [P,4] of ONE image (the oracle's tensors).
    Returns dict(sure=set[(prior, class)], possible=set[...], unsure=list of (prior, class, score, reason))."""
    boxes_all = O.decode(loc, priors).double().numpy()
    cur = conf[:, 1:].t().contiguous().double().numpy()          # [C-1, P]
    ncls, P = cur.shape
    maxsc = cur.max(0)
    pres_sure = maxsc > conf_thresh + delta
    pres_poss = maxsc > conf_thresh - delta
    poss_idx = np.nonzero(pres_poss)[0]
    why = {}

    surv_sure, surv_poss = [], []        # (score, prior, class)
    for c in range(ncls):
        sc_all = cur[c]
        s_poss = sc_all[poss_idx]
        order = np.argsort(-s_poss, kind='stable')
        # only the head of the list can matter: rank < top_k needs fewer than top_k sure rivals above score + delta
        sp_sorted = s_poss[order]
        sure_mask_sorted = pres_sure[poss_idx][order]
        ss_sorted = sp_sorted[sure_mask_sorted]
        # candidates that can possibly be in the class list
        above_sure = _count_greater(ss_sorted, sp_sorted + delta)
        in_poss = above_sure < top_k
        n_head = int(in_poss.sum())
        if n_head == 0:
            continue
        head = np.nonzero(in_poss)[0]                      # positions in the sorted order (a prefix-like set)
        hp = poss_idx[order[head]]                         # prior ids
        hs = sp_sorted[head]
        above_poss = _count_greater(sp_sorted, hs - delta) - 1           # minus self
        in_sure = pres_sure[hp] & (above_poss < top_k)
        iou = _iou_matrix(boxes_all[hp])
        np.fill_diagonal(iou, 0.0)
        # possible suppressors of j: anything possibly in the list that could be ranked above j
        could_be_above = hs[:, None] > (hs[None, :] - delta)             # [i, j]
        np.fill_diagonal(could_be_above, False)
        worst = np.where(could_be_above, iou, 0.0).max(0) if n_head > 1 else np.zeros(n_head)
        kept_sure = worst <= nms_thresh - delta_iou
        # sure suppressors of j: surely in the list and surely ranked above j
        surely_above = in_sure[:, None] & (hs[:, None] > (hs[None, :] + delta))
        best = np.where(surely_above, iou, 0.0).max(0) if n_head > 1 else np.zeros(n_head)
        supp_sure = best > nms_thresh + delta_iou
        for k in range(n_head):
            key = (int(hp[k]), c)
            if not supp_sure[k]:
                surv_poss.append((hs[k], key))
            if in_sure[k] and kept_sure[k]:
                surv_sure.append((hs[k], key))
            else:
                r = []
                if not pres_sure[hp[k]]:
                    r.append('threshold margin %.2e' % abs(maxsc[hp[k]] - conf_thresh))
                if pres_sure[hp[k]] and not in_sure[k]:
                    r.append('top-%d rank margin < %.0e' % (top_k, delta))
                if not kept_sure[k] and not supp_sure[k]:
                    r.append('nms margin (worst rival IoU %.4f)' % worst[k])
                why[key] = '; '.join(r)

    sure_scores = np.sort(np.array([s for s, _ in surv_sure]))[::-1] if surv_sure else np.zeros(0)
    poss_scores = np.sort(np.array([s for s, _ in surv_poss]))[::-1] if surv_poss else np.zeros(0)
    sure_keys = {k for _, k in surv_sure}
    sure, possible, unsure = set(), set(), []
    for s, key in surv_poss:
        n_above_sure = int(_count_greater(sure_scores, np.array([s + delta]))[0])
        if n_above_sure >= max_det:
            continue                                        # surely cut by the final top-k
        possible.add(key)
        n_above_poss = int(_count_greater(poss_scores, np.array([s - delta]))[0]) - 1
        if key in sure_keys and n_above_poss < max_det:
            sure.add(key)
        else:
            reason = why.get(key, '')
            if key in sure_keys:
                reason = 'final top-%d cut margin < %.0e' % (max_det, delta)
            unsure.append((key[0], key[1], float(s), reason))
    return {'sure': sure, 'possible': possible, 'unsure': unsure}


def check_against_oracle(marg: Dict, ref: Optional[Dict[str, Tensor]]):
    """Self-check: sure <= the oracle's own output <= possible."""
    rset = set() if ref is None else set(zip(ref['prior'].tolist(), ref['class'].tolist()))
    assert marg['sure'] <= rset, ('margin analysis claims sure detections the oracle does not output',
                                  sorted(marg['sure'] - rset)[:5])
    assert rset <= marg['possible'], ('oracle outputs detections the margin analysis excludes',
                                      sorted(rset - marg['possible'])[:5])


def margin_match(got_prior: List[int], got_class: List[int], marg: Dict) -> List[str]:
    """Device output of one image vs the margin analysis.  Returns the list of violations (empty = every decision whose
    margin exceeds delta was reproduced)."""
    g = set(zip(got_prior, got_class))
    problems = []
    miss = marg['sure'] - g
    extra = g - marg['possible']
    if len(g) != len(got_prior):
        problems.append('duplicate (prior, class) pairs in the device output')
    for k in sorted(miss)[:10]:
        problems.append('missing sure detection prior=%d class=%d' % k)
    for k in sorted(extra)[:10]:
        problems.append('impossible detection prior=%d class=%d' % k)
    return problems


def summarize(marg: Dict) -> str:
    n_s, n_p = len(marg['sure']), len(marg['possible'])
    lines = ['%d sure, %d possible (%d undecidable at this delta)' % (n_s, n_p, n_p - n_s)]
    for p, c, s, r in marg['unsure'][:8]:
        lines.append('  prior %d class %d score %.6f: %s' % (p, c, s, r))
    return '\n'.join(lines)
