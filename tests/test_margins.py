"""The margin analysis of Detect's decisions (oracle/margins.py) is itself checked on the CPU:
  * sure <= the oracle's output <= possible on every golden case,
  * a perturbed re-run of the oracle (scores moved by < delta/2, boxes by 1e-5) still satisfies sure <= out <= possible —
    i.e. the analysis really is conservative, which is what lets the GPU parity tests demand it of the device path."""
import pytest
import torch

from helpers import oracle_run

CASES = ['r50_dense', 'r50_sparse', 'r50_empty', 'r101_base', 'plus_r50']


def _marg(raw, cfg, b, delta):
    from oracle import margins as MG
    return MG.detect_margins(raw['conf'][b], raw['loc'][b], raw['priors'], cfg.nms_conf_thresh, cfg.nms_thresh,
                             cfg.nms_top_k, cfg.max_num_detections, delta=delta, delta_iou=delta)


@pytest.mark.parametrize('name', CASES)
def test_margins_bracket_the_oracle(name):
    from oracle import margins as MG
    meta, arrays, cfg, sd, raw, dets = oracle_run(name)
    for b in range(meta['B']):
        for delta in (1e-3, 1e-4):
            m = _marg(raw, cfg, b, delta)
            MG.check_against_oracle(m, dets[b])
            if dets[b] is not None:
                assert len(m['sure']) >= 0.4 * dets[b]['score'].shape[0], MG.summarize(m)   # the analysis is not vacuous


@pytest.mark.parametrize('name', ['r50_dense', 'r50_sparse', 'r101_base'])
def test_margins_hold_under_perturbation(name):
    from oracle import margins as MG
    from oracle import yolact_oracle as O
    meta, arrays, cfg, sd, raw, dets = oracle_run(name)
    delta = 1e-3
    g = torch.Generator().manual_seed(11)
    for b in range(meta['B']):
        m = _marg(raw, cfg, b, delta)
        for trial in range(3):
            conf = raw['conf'][b] + (torch.rand(raw['conf'][b].shape, generator=g) - 0.5) * (0.9 * delta)
            loc = raw['loc'][b] + (torch.rand(raw['loc'][b].shape, generator=g) - 0.5) * 2e-4
            out = O.detect_image(conf, loc, raw['mask'][b], raw['priors'], cfg.nms_conf_thresh, cfg.nms_thresh,
                                 cfg.nms_top_k, cfg.max_num_detections)
            problems = MG.margin_match(out['prior'].tolist(), out['class'].tolist(), m)
            assert not problems, problems
