"""Oracle parity of the plans that are actually TIMED: the BASELINE configurations at their batch sizes, with the default
(shipped-table) plan, plus one run with F(4x4,3x3) Winograd forced onto every eligible layer.

Every golden fixture runs at batch 1-2; which layers go Winograd, and with which tile, depends on the batch (the tune
table is keyed by layer shape incl. B), so parity at batch 1 says nothing about the batch-8 plan bench.py times.  Here the
CPU oracle (oracle/yolact_oracle.py: torch-CPU restatement pinned to the executed reference, tests/test_oracle_golden.py)
runs on the GPU box's host cores at the bench shapes and is compared with `net.forward_raw` and `net(x)`:

  heads   |loc|, |coef| <= 1e-4 absolute; softmax scores <= 1e-4 absolute — these are what the north star's "masks / scores
          within 1e-4 fp32" names.  Conf LOGITS and raw PROTOTYPES are held to 1e-4 * max(1, max|ref|): that bar is LOOSER than
          an absolute 1e-4 whenever the tensor's scale exceeds 1 (max|ref| is 10 - 60 on the synthetic weights), and it is
          stated here because it is a choice, not an accident: both are unbounded intermediate tensors whose error the softmax /
          sigmoid that follows contracts, and the quantities a user sees (scores, boxes, coefficients, binarised masks) carry the
          absolute bar.  Measured errors are 1.3e-6 .. 1.8e-5 of the scale (profiles/r0*_gpu_tests_summary.txt), i.e. <= 1.1e-3
          absolute on logits of magnitude 60 and far below 1e-4 absolute after the softmax (<= 5e-7 measured)
  Detect  margin-aware matching (oracle/margins.py): every reference decision whose margin exceeds 1e-3 is reproduced
          exactly, common detections agree to 1e-4; undecidable ones are listed
  masks   postprocess() of two images: binary masks differ from the oracle's only where its soft value is within 1e-4 of 0.5
"""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import CONFIGS, assert_margin_match  # noqa: E402
from yolact_amd.utils.synth import synth_images, synth_state_dict  # noqa: E402

DEV = 'cuda:0'

# (id, config, batch, size, weight seed, conf gain, image seed)
CASES = [
    ('configs1_r50_b8', 'yolact_resnet50_config', 8, 550, 0, 0.04, 1234),     # == bench.py's default workload, same tensors
    ('configs2_r101_b16', 'yolact_base_config', 16, 550, 3, 0.04, 2234),
    ('configs4_im700_b8', 'yolact_im700_config', 8, 700, 5, 0.04, 3234),       # per-GPU share of configs[4]
    ('configs3_plus_r50_b8', 'yolact_plus_resnet50_config', 8, 550, 6, 0.04, 4234),
    ('darknet53_b8', 'yolact_darknet53_config', 8, 550, 4, 0.04, 5234),        # named in north_star
    ('plus_base_b8', 'yolact_plus_base_config', 8, 550, 9, 0.04, 6234),        # the published YOLACT++ R101 row (README.md:80; data/config.py:772-792)
    ('im400_b8', 'yolact_im400_config', 8, 400, 10, 0.04, 7234),               # data/config.py:706
]


_oracle_cache = {}      # (batch, size, image seed, config, weight seed, gain) -> the oracle's head tensors


def _build(config, seed, gain):
    import yolact_amd
    yolact_amd.set_cfg(config)
    from yolact_amd.yolact import Yolact
    net = Yolact()
    sd = synth_state_dict([(k, tuple(v.shape)) for k, v in net.state_dict().items()], seed=seed, conf_gain=gain)
    net.load_state_dict_compat(sd)
    net.detect.use_fast_nms = True
    net._synth_key = (config, seed, gain)      # two builds of the same synthetic model share the oracle's results
    return net.to(DEV), sd


def _compare(net, sd, config, B, size, img_seed, tag):
    from oracle import yolact_oracle as O
    from yolact_amd.layers.output_utils import postprocess
    cfg = CONFIGS[config].copy()
    x = synth_images(B, size, size, seed=img_seed)
    key = (B, size, img_seed) + net._synth_key
    if key not in _oracle_cache:      # the CPU oracle at the bench batch is the expensive part: one run per (model, input)
        with torch.no_grad():
            raw = O.forward_raw(x, sd, cfg)
            _oracle_cache[key] = raw
    raw = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in _oracle_cache[key].items()}     # consumers may write in place
    with torch.no_grad():
        dets = O.detect(raw, cfg)
    xd = x.to(DEV)
    got = net.forward_raw(xd)
    torch.cuda.synchronize()
    plan = net.plan_for(xd)
    errs = {}
    for k, absolute in (('loc', True), ('mask', True), ('conf_logits', False), ('proto', False)):
        g, r = got[k].cpu(), raw[k]
        assert g.shape == r.shape, (k, g.shape, r.shape)
        err = (g - r).abs().max().item()
        scale = 1.0 if absolute else max(1.0, r.abs().max().item())
        errs[k] = err / scale
        assert err <= 1e-4 * scale, '%s %s: max err %g (scale %g)' % (tag, k, err, scale)
    conf_err = (torch.softmax(got['conf_logits'], -1).cpu() - raw['conf']).abs().max().item()
    assert conf_err <= 1e-4, conf_err
    assert torch.equal(got['priors'].cpu(), raw['priors'])
    out = net(xd)
    assert len(out) == B
    summary = assert_margin_match(net.detect.last_prior_idx, out, raw, dets, cfg, delta=1e-3)
    n_w2 = sum(1 for op in plan.ops if isinstance(op[2], str) and op[2].endswith('[wino]') and op[1].contents.m == 2)
    n_w4 = sum(1 for op in plan.ops if isinstance(op[2], str) and op[2].endswith('[wino]') and op[1].contents.m == 4)
    print('%s: plan F(2x2) x%d, F(4x4) x%d, tune misses %d; head errors / scale %s; softmax err %.2e; per image '
          '(sure, possible, common) %s' % (tag, n_w2, n_w4, plan.tune_misses, {k: '%.2e' % v for k, v in errs.items()},
                                           conf_err, [(s, p, c) for _, s, p, c in summary]))
    for b in (0, B - 1):
        if dets[b] is None:
            continue
        # stage the ORACLE's detections through the device postprocess: identical inputs, so the masks may differ only
        # at pixels whose soft value sits within 1e-4 of the threshold
        d = {k: dets[b][k].to(DEV).clone() for k in ('box', 'mask', 'class', 'score', 'proto')}
        classes, scores, boxes, masks = postprocess([{'detection': d, 'net': net}], size, size)
        rc, rs, rb, rm, soft = O.postprocess(dets[b], size, size, cfg, sd, return_soft=True)
        assert torch.equal(classes.cpu(), rc) and torch.equal(boxes.cpu(), rb)
        bad = masks.cpu() != rm
        assert bad.float().mean().item() < 1e-4
        if bad.any():
            assert (soft[bad] - 0.5).abs().max().item() < 1e-4
    return plan, n_w2, n_w4


@pytest.mark.parametrize('case', CASES, ids=[c[0] for c in CASES])
def test_timed_plan_matches_oracle_at_batch(case):
    tag, config, B, size, seed, gain, img_seed = case
    net, sd = _build(config, seed, gain)
    _compare(net, sd, config, B, size, img_seed, tag)


def test_forced_f4x4_plan_matches_oracle_at_batch8():
    """YOLACT_AMD_WINOGRAD=4 + YOLACT_AMD_WINOGRAD_FORCE=1: every eligible 3x3 / stride-1 layer runs F(4x4,3x3) whether or
    not it is the fastest choice, so the whole-network error of the least accurate variant is what gets checked."""
    old = {k: os.environ.get(k) for k in ('YOLACT_AMD_WINOGRAD', 'YOLACT_AMD_WINOGRAD_FORCE')}
    os.environ['YOLACT_AMD_WINOGRAD'], os.environ['YOLACT_AMD_WINOGRAD_FORCE'] = '4', '1'
    try:
        tag, config, B, size, seed, gain, img_seed = CASES[0]
        net, sd = _build(config, seed, gain)
        plan, n_w2, n_w4 = _compare(net, sd, config, B, size, img_seed, 'forced F(4x4) ' + tag)
        assert n_w2 == 0 and n_w4 >= 25, (n_w2, n_w4)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


from yolact_amd.utils.synth import plant_outlier_channels as _plant_outlier_channels  # noqa: E402  (shared with bench.py's outlier_plan line)


@pytest.mark.parametrize('k_exp', [12, 16, 20])
def test_outlier_channels_are_rebalanced_end_to_end_at_batch8(k_exp):
    """Round 6, the default path for such checkpoints: Plan._rebalance_outliers undoes the compensated re-parametrisation at pack time
    (producers' folded BN scales x 2^-k, consumers' filters x 2^k: exact), so NOTHING leaves the fp16x2 tiles — same launches, same
    Winograd / fusion decisions, zero tune misses, the timed plan's speed — and the heads meet the same bars as the clean network."""
    tag, config, B, size, seed, gain, img_seed = CASES[0]
    net, sd0 = _build(config, seed, gain)
    sd, planted = _plant_outlier_channels({k: v.cpu() for k, v in sd0.items()}, k_exp)
    net.load_state_dict_compat(sd)
    net._synth_key = (config, seed, gain, 'outliers-rebalanced', k_exp)
    plan, _, _ = _compare(net, sd, config, B, size, img_seed, 'outlier channels 2^%d %s, rebalanced' % (k_exp, planted))
    assert not plan.wide_ops and plan.tune_misses == 0
    assert [r[0] for r in plan.rebalanced] == ['C3', 'C4', 'C5', 'proto_net.0'] and all(k_exp - 2 <= r[2] <= k_exp for r in plan.rebalanced)
    x = synth_images(B, size, size, seed=img_seed).to(DEV)
    net.forward_raw(x)
    bounds = sorted(plan.bound(sl) for sl in range(plan._nslots))
    print('magnitude bounds of the rebalanced plan\'s tensors: median %.3g, max %.3g' % (bounds[len(bounds) // 2], bounds[-1]))
    assert bounds[-1] < 2 ** 10 * bounds[len(bounds) // 2]         # no tensor carries the planted range any more


@pytest.mark.parametrize('k_exp', [12, 16, 20])
def test_outlier_channels_end_to_end_at_batch8(k_exp, monkeypatch):
    """fp16x2 carries one power-of-two scale per activation TENSOR (h stays a normal fp16 27 binades below the bound,
    l 15) and one per FILTER ROW; a single-layer test covered 2^14 outliers, nothing did end to end.  Here the R50 network is
    re-parametrised (exactly, see _plant_outlier_channels) so that C3 / C4 / C5 and a protonet tensor carry channels 2^12 .. 2^20 times
    larger than the rest, and the default batch-8 plan is held to the same bars as the timed plan: heads within 1e-4, every
    decidable detection reproduced.  Measured WITHOUT the guard (YOLACT_AMD_WIDE_GUARD=0, profiles/r04_outlier_stress.txt): head
    errors 9e-6 at 2^12, 3.4e-5 at 2^14, 1.27e-4 at 2^16 — the low fp16 piece of a typical value goes subnormal 15 binades below the
    tensor's bound.  The engine therefore recognises such layers from their FILTERS (input channels weighted >= 2^10 less than the
    typical one: engine.Packed.tiny_columns) and runs them on the bf16x3 tiles, which have no scale."""
    monkeypatch.setenv('YOLACT_AMD_REBALANCE', '0')      # the guard's path (the default rebalances: test above)
    tag, config, B, size, seed, gain, img_seed = CASES[0]
    net, sd0 = _build(config, seed, gain)
    sd, planted = _plant_outlier_channels({k: v.cpu() for k, v in sd0.items()}, k_exp)
    net.load_state_dict_compat(sd)
    net._synth_key = (config, seed, gain, 'outliers', k_exp)
    plan, _, _ = _compare(net, sd, config, B, size, img_seed, 'outlier channels 2^%d %s' % (k_exp, planted))
    wide = sorted(plan.ops[i][2] for i in plan.wide_ops)
    print('layers moved to bf16x3 tiles by the outlier-channel guard (%d): %s' % (len(wide), wide))
    assert len(wide) >= 10 and any(n.startswith('fpn.lat') for n in wide) and any(n.startswith('proto.') for n in wide)
    # the planted channels really are outliers of their tensors on the device: the bound of C3's slot vs a typical channel
    x = synth_images(B, size, size, seed=img_seed).to(DEV)
    net.forward_raw(x)
    bounds = sorted(plan.bound(sl) for sl in range(plan._nslots))
    print('magnitude bounds of the plan\'s tensors with 2^%d outliers: median %.3g, max %.3g' % (k_exp, bounds[len(bounds) // 2], bounds[-1]))
    assert bounds[-1] >= 2 ** (k_exp - 4) * bounds[len(bounds) // 2]


def test_outlier_guard_is_silent_on_the_timed_plan():
    """The synthetic (and any ordinary) checkpoint has no compensated outlier channels: no layer leaves the fp16x2 tiles."""
    tag, config, B, size, seed, gain, img_seed = CASES[0]
    net, sd = _build(config, seed, gain)
    plan = net.plan_for(synth_images(B, size, size, seed=img_seed).to(DEV))
    assert not plan.wide_ops and plan.tune_misses == 0


def test_timed_plan_fuses_the_last_1x1_into_the_output_transform():
    """The batch-8 plan of configs[1] runs proto.8 as F(4x4,3x3), so proto.10 (1x1, 256 -> 32) is computed inside its output
    transform (ymi_wino_desc.proj_*; the 156 MB tensor between them is never written): the op is gone from the launch list, and the
    prototypes equal those of the two separate launches to fp32 rounding of a 256-term sum (YOLACT_AMD_WINO_PROJ=0)."""
    tag, config, B, size, seed, gain, img_seed = CASES[0]
    x = synth_images(B, size, size, seed=img_seed).to(DEV)
    net, sd = _build(config, seed, gain)
    plan = net.plan_for(x)
    assert any(op[0] == 'nop' and op[2].startswith('proto.10[fused into proto.8') for op in plan.ops), [op[2] for op in plan.ops if 'proto' in op[2]]
    with torch.no_grad():
        fused = net.forward_raw(x)['proto'].clone()
    os.environ['YOLACT_AMD_WINO_PROJ'] = '0'
    try:
        net2, _ = _build(config, seed, gain)
        plan2 = net2.plan_for(x)
        assert not any(op[0] == 'nop' and op[2].startswith('proto.10') for op in plan2.ops)
        with torch.no_grad():
            two = net2.forward_raw(x)['proto']
    finally:
        del os.environ['YOLACT_AMD_WINO_PROJ']
    err = ((fused - two).abs().max() / two.abs().max()).item()
    print('fused projection vs two launches: %.2e of max|proto|' % err)
    assert err < 2e-5


def test_plan_is_deterministic_across_processes():
    """The default plan comes from the shipped tune table, so two fresh processes produce identical bits (the round-1
    plan timed its tiles per process: K-split tiles change the summation order, i.e. the bits)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, hashlib, torch; sys.path.insert(0, %r); sys.path.insert(0, %r + '/tests')\n"
        "from test_gpu_batch_parity import _build\n"
        "from yolact_amd.utils.synth import synth_images\n"
        "net, sd = _build('yolact_resnet50_config', 0, 0.04)\n"
        "x = synth_images(8, 550, 550, seed=1234).to('cuda:0')\n"
        "r = net.forward_raw(x); torch.cuda.synchronize()\n"
        "h = hashlib.sha256()\n"
        "[h.update(r[k].cpu().numpy().tobytes()) for k in ('loc', 'conf_logits', 'mask', 'proto')]\n"
        "print('DIGEST', h.hexdigest(), net.plan_for(x).tune_misses)\n" % (root, root))
    outs = []
    for _ in range(2):
        p = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-2000:]
        outs.append([l for l in p.stdout.splitlines() if l.startswith('DIGEST')][-1].split())
    assert outs[0][1] == outs[1][1], 'two processes produced different head tensors'
    assert outs[0][2] == '0' and outs[1][2] == '0', 'BASELINE configs[1] must be fully covered by the shipped tune table'


def test_timed_plan_chains_the_first_stage_pointwise_layers():
    """The batch-8 plan of configs[1] runs conv3 (+ shortcut, ReLU) of layer0.0 / layer0.1 and conv1 of the next block as ONE
    ymi_pointwise_chain_f32 launch each (the shipped table says it is faster); with YOLACT_AMD_CHAIN=0 the same layers are two
    launches, and the network's outputs agree to fp32 rounding (the chain splits the 256-channel tensor with per-slice scales
    instead of the tensor-wide one: a different, not a larger, rounding)."""
    tag, config, B, size, seed, gain, img_seed = CASES[0]
    x = synth_images(B, size, size, seed=img_seed).to(DEV)
    net, sd = _build(config, seed, gain)
    plan = net.plan_for(x)
    names = [op[2] for op in plan.ops if op[0] is plan.lib.ymi_pointwise_chain_f32]
    assert names[:2] == ['layer0.0.conv3+layer0.1.conv1', 'layer0.1.conv3+layer0.2.conv1'], (names, getattr(plan, 'chain_table', None))
    assert names[2:] in ([], ['layer0.2.conv3'])           # (the last block's conv3 alone on the same kernel, where the table prefers it)
    with torch.no_grad():
        a = {k: v.clone() for k, v in net.forward_raw(x).items() if k in ('loc', 'conf_logits', 'mask', 'proto')}
    os.environ['YOLACT_AMD_CHAIN'] = '0'
    try:
        net2, _ = _build(config, seed, gain)
        plan2 = net2.plan_for(x)
        assert not any(op[0] is plan2.lib.ymi_pointwise_chain_f32 for op in plan2.ops)
        with torch.no_grad():
            b = net2.forward_raw(x)
    finally:
        del os.environ['YOLACT_AMD_CHAIN']
    for k in a:
        err = ((a[k] - b[k]).abs().max() / b[k].abs().max()).item()
        print('chain vs two launches, %s: %.2e of max' % (k, err))
        assert err < 2e-5, k
