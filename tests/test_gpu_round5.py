"""Round-5 additions on the GPU (VERDICT r4 "missing" #3 - #5): YOLACT++ batched postprocess (FastMaskIoUNet once per batch),
the published `yolact_plus_base_config` / `yolact_im400_config` (covered by the parametrised golden tests through
helpers.ALL_CASES) at their batch-8 plans, and eval.py's --detect mode (cfg.eval_mask_branch False)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import NOMASK_CASE, assert_margin_match, case_images, check_digest, oracle_run  # noqa: E402

DEV = 'cuda:0'


def test_yolact_plus_batched_postprocess_equals_per_image_and_reference():
    """postprocess_batch with cfg.use_maskiou (output_utils.py:79-88 per image in the reference): ONE FastMaskIoUNet chain over all
    B * cap masks.  Rows of live detections equal the per-image postprocess() bit for bit, scores2 equals what the EXECUTED
    reference wrote for both images of the golden batch (plus_r50_b2) when fed the oracle's detections."""
    import yolact_amd
    from gpu_utils import build_net
    from oracle import yolact_oracle as O
    from yolact_amd.layers.output_utils import postprocess, postprocess_batch
    meta, arrays, cfg, sd, raw, dets = oracle_run('plus_r50_b2')
    net = build_net(meta)
    x = case_images(meta).to(DEV)
    w, h = meta['post']
    dev_out = net.forward_device(x)
    assert dev_out['net'] is net
    preds = net.detect.finish(dev_out, dev_out['proto'], net)
    bat = postprocess_batch(dev_out, w, h)
    assert isinstance(bat['scores'], list) and len(bat['scores']) == 2
    counts = bat['count'].tolist()
    for b in range(meta['B']):
        classes, scores, boxes, masks = postprocess(preds, w, h, batch_idx=b)
        n = counts[b]
        assert n == classes.shape[0] and n > 0 and isinstance(scores, list)
        assert torch.equal(bat['masks'][b, :n], masks) and torch.equal(bat['boxes'][b, :n], boxes)
        assert torch.equal(bat['scores'][0][b, :n], scores[0]) and torch.equal(bat['scores'][1][b, :n], scores[1])
    # the reference's own score2 for BOTH images: stage the oracle's detections (== the reference's, tests/test_oracle_golden.py)
    # as one fixed-capacity batch
    cap = int(cfg.max_num_detections)
    D = raw['mask'].shape[-1]
    fake = {'count': torch.tensor([d['score'].shape[0] for d in dets], dtype=torch.int32, device=DEV),
            'box': torch.zeros(meta['B'], cap, 4, device=DEV), 'score': torch.zeros(meta['B'], cap, device=DEV),
            'cls': torch.zeros(meta['B'], cap, dtype=torch.int64, device=DEV), 'coef': torch.zeros(meta['B'], cap, D, device=DEV),
            'proto': raw['proto'].to(DEV).contiguous(), 'net': net}
    for b, d in enumerate(dets):
        n = d['score'].shape[0]
        fake['box'][b, :n], fake['score'][b, :n] = d['box'].to(DEV), d['score'].to(DEV)
        fake['cls'][b, :n], fake['coef'][b, :n] = d['class'].to(DEV), d['mask'].to(DEV)
    bat = postprocess_batch(fake, w, h)
    for b, d in enumerate(dets):
        n = meta['n_post'][b]
        ref2 = torch.from_numpy(arrays['post%d_score2' % b])
        # (the staged detections are the ORACLE's: its scores equal the reference's to ~1e-7, not bit for bit)
        assert (bat['scores'][0][b, :n].cpu() - torch.from_numpy(arrays['post%d_score' % b])).abs().max().item() < 1e-6
        assert (bat['scores'][1][b, :n].cpu() - ref2).abs().max().item() < 1e-4 * max(1.0, float(ref2.abs().max()))
        rc, rs, rb, rm = O.postprocess(d, w, h, cfg, sd)
        assert torch.equal(bat['boxes'][b, :n].cpu(), rb) and torch.equal(bat['scores'][0][b, :n].cpu(), rs[0])
        assert (bat['scores'][1][b, :n].cpu() - rs[1]).abs().max().item() < 1e-4 * max(1.0, float(rs[1].abs().max()))
        assert (bat['masks'][b, :n].cpu() != rm).float().mean().item() < 1e-4
    # rescore_bbox = True (eval.py:147-152 prep_display) -> the product alone
    yolact_amd.active_cfg().rescore_bbox = True
    try:
        one = postprocess_batch(fake, w, h)['scores']
        assert torch.is_tensor(one) and torch.equal(one, bat['scores'][1])
    finally:
        yolact_amd.active_cfg().rescore_bbox = False


def test_detect_only_mode_end_to_end():
    """eval.py --detect (eval.py:1067-1068: cfg.eval_mask_branch = False) through the engine: zero coefficients (yolact.py:172-175), no
    'proto' on the detection dicts (detection.py:73-74), the protonet's launches skipped, postprocess returns boxes / classes /
    scores and the coefficient rows as its 4th value (output_utils.py:58,97-122) — against the reference-executed golden."""
    import yolact_amd
    from gpu_utils import build_net
    from yolact_amd.layers.output_utils import postprocess, postprocess_batch
    meta, arrays, cfg, sd, raw, dets = oracle_run(NOMASK_CASE)
    net = build_net(meta)
    x = case_images(meta).to(DEV)
    acfg = yolact_amd.active_cfg()
    assert acfg.eval_mask_branch is True
    full = net(x)                                              # the same model WITH the mask branch first: has prototypes
    assert 'proto' in full[0]['detection'] and float(full[0]['detection']['mask'].abs().sum()) > 0
    acfg.eval_mask_branch = False
    try:
        out = net(x)
        w, h = meta['post']
        # every decision of the reference whose margin exceeds 1e-3 is reproduced (oracle/margins.py), common detections to 1e-4
        print(NOMASK_CASE, assert_margin_match(net.detect.last_prior_idx, out, raw, dets, cfg, delta=1e-3))
        for b in range(meta['B']):
            g = out[b]['detection']
            assert set(g) == {'box', 'mask', 'class', 'score'} and float(g['mask'].abs().sum()) == 0.0
            ref = {k: torch.from_numpy(arrays['det%d_%s' % (b, k)]) for k in ('box', 'mask', 'class', 'score')}
            # boxes stay equal to what the run WITH the mask branch returned: the branch does not touch them
            assert torch.equal(g['box'], full[b]['detection']['box']) and torch.equal(g['score'], full[b]['detection']['score'])
            classes, scores, boxes, masks = postprocess(out, w, h, batch_idx=b)
            assert boxes.dtype == torch.int64 and masks.shape == (g['score'].shape[0], 32) and float(masks.abs().sum()) == 0.0
            # staged with the reference's own detections: integer boxes exact
            d = {kk: ref[kk].to(DEV).clone() for kk in ('box', 'mask', 'class', 'score')}
            c2, s2, b2, m2 = postprocess([{'detection': d, 'net': net}], w, h)
            assert torch.equal(b2.cpu(), torch.from_numpy(arrays['post%d_box' % b]))
            assert torch.equal(c2.cpu(), torch.from_numpy(arrays['post%d_class' % b]))
            assert torch.equal(m2.cpu(), torch.from_numpy(arrays['post%d_maskraw' % b]))
        dev_out = net.forward_device(x)
        assert dev_out['proto'] is None
        bat = postprocess_batch(dev_out, w, h)
        assert bat['masks'] is None and bat['boxes'].dtype == torch.int64
        n = int(bat['count'][0])
        assert torch.equal(bat['boxes'][0, :n], postprocess(out, w, h)[2])
    finally:
        acfg.eval_mask_branch = True
    again = net(x)                                             # and back: the plan is shared, nothing was left switched off
    assert torch.equal(again[0]['detection']['proto'], full[0]['detection']['proto'])
    assert torch.equal(again[0]['detection']['mask'], full[0]['detection']['mask'])


def test_head_digests_of_detect_only_mode():
    from gpu_utils import build_net
    meta, arrays, cfg, sd, raw, dets = oracle_run(NOMASK_CASE)
    net = build_net(meta)
    got = net.forward_raw(case_images(meta).to(DEV))
    for k in ('loc', 'priors'):
        check_digest(got[k], meta, arrays, k, rtol=1e-4, atol=1e-4)
    check_digest(torch.softmax(got['conf_logits'], -1), meta, arrays, 'conf', rtol=1e-4, atol=1e-4)


def test_data_parallel_replicas_run_the_engine():
    """nn.DataParallel's replicate() (eval.py:630-634,661) on the real module: two replicas (both on cuda:0 — the box has one GPU) are
    shallow copies with broadcast parameter copies; each must build / find its plan and return what the module itself returns."""
    from gpu_utils import build_net
    from helpers import load_golden
    meta, _ = load_golden('r50_dense')
    net = build_net(meta)
    x = case_images(meta).to(DEV)
    ref = net(x)
    try:
        reps = torch.nn.parallel.replicate(net, [0, 0])
    except Exception as e:            # a torch build that refuses duplicate device ids: the shallow-copy half alone
        print('replicate([0, 0]) not available (%s): using _replicate_for_data_parallel' % e)
        reps = [net._replicate_for_data_parallel() for _ in range(2)]
        for r in reps:
            for k, v in net._modules.items():
                r._modules[k] = v
    outs = torch.nn.parallel.parallel_apply(reps, [(x,), (x,)], devices=[0, 0])
    for out in outs:
        assert len(out) == len(ref)
        for a, r in zip(out, ref):
            assert a['net'] in reps
            for k in ('box', 'score', 'class', 'mask', 'proto'):
                assert torch.equal(a['detection'][k], r['detection'][k]), k
    dp = torch.nn.DataParallel(net, device_ids=[0])
    one = dp(x)
    assert torch.equal(one[0]['detection']['box'], ref[0]['detection']['box'])


def test_early_fpn_laterals_equal_the_fused_epilogue(monkeypatch):
    """YOLACT_AMD_EARLY_LAT=1: lateral convolution without residual on the side stream + ymi_bilinear_add_nhwc_f32 against the fused
    YMI_RES_BILINEAR epilogue of the default plan.  The pass itself reproduces the epilogue's interpolation exactly
    (test_bilinear_add_kernel_matches_torch); the lateral GEMM, freed of the bilinear epilogue, runs on another tile of the table
    (another fp32 summation order), so the head tensors agree to rounding, not bit for bit."""
    from gpu_utils import build_net
    from helpers import load_golden
    meta, _ = load_golden('r50_dense')
    x = case_images(meta).to(DEV)
    ref = build_net(meta).forward_raw(x)
    monkeypatch.setenv('YOLACT_AMD_EARLY_LAT', '1')
    net = build_net(meta)
    got = net.forward_raw(x)
    plan = net.plan_for(x)
    assert any(op[2] == 'fpn.add1' for op in plan.ops) and any(op[2] == 'fpn.add2' for op in plan.ops)
    for k in ('loc', 'conf_logits', 'mask', 'proto'):
        assert (got[k] - ref[k]).abs().max().item() <= 4e-6 * max(1.0, ref[k].abs().max().item()), k


def test_bilinear_add_kernel_matches_torch():
    import ctypes as C
    from yolact_amd import _lib as L
    g = torch.Generator().manual_seed(3)
    for (B, Hi, Wi, Cc, Ho, Wo) in ((2, 18, 18, 256, 35, 35), (1, 35, 35, 64, 69, 69), (3, 5, 7, 8, 9, 13)):
        x = torch.randn(B, Hi, Wi, Cc, generator=g).to(DEV)
        y = torch.randn(B, Ho, Wo, Cc, generator=g).to(DEV)
        want = y + torch.nn.functional.interpolate(x.permute(0, 3, 1, 2), (Ho, Wo), mode='bilinear', align_corners=False).permute(0, 2, 3, 1)
        amax = torch.zeros(16 * 64, device=DEV)
        L.check(L.lib().ymi_bilinear_add_nhwc_f32(x.data_ptr(), y.data_ptr(), B, Hi, Wi, Cc, Ho, Wo, amax.data_ptr(), L.stream_ptr()))
        assert (y - want).abs().max().item() < 2e-6 * max(1.0, want.abs().max().item())
        assert abs(float(amax.max()) - float(y.abs().max())) == 0.0
    assert L.lib().ymi_bilinear_add_nhwc_f32(x.data_ptr(), y.data_ptr(), 1, 5, 7, 6, 9, 13, None, L.stream_ptr()) == -2


def test_native_executor_equals_the_python_loop(monkeypatch):
    """csrc/plan_exec.cpp walks the op list in two native calls (around the Detect callback); YOLACT_AMD_NATIVE_EXEC=0 issues the same
    list from the Python loop.  Same launches, same streams, same events: every output bit-identical; the native path is the default
    and the one this test must find active."""
    import time
    from gpu_utils import build_net
    from helpers import load_golden
    meta, _ = load_golden('r50_dense')
    x = case_images(meta).to(DEV)
    net = build_net(meta)
    plan = net.plan_for(x)
    assert plan.native_exec
    a = net.forward_device(x)
    assert plan._native is not None and plan._native[1] is not None
    raw = net.forward_raw(x)
    monkeypatch.setenv('YOLACT_AMD_NATIVE_EXEC', '0')
    net2 = build_net(meta)
    plan2 = net2.plan_for(x)
    assert not plan2.native_exec
    b = net2.forward_device(x)
    raw2 = net2.forward_raw(x)
    for k in ('count', 'box', 'score', 'cls', 'coef', 'proto'):
        nb = int(a['count'].min())
        assert torch.equal(a[k][:, :nb] if a[k].dim() > 1 and k != 'proto' else a[k], b[k][:, :nb] if b[k].dim() > 1 and k != 'proto' else b[k]), k
    for k in ('loc', 'conf_logits', 'mask', 'proto'):
        assert torch.equal(raw[k], raw2[k]), k

    def issue_ms(n_, reps=30):
        torch.cuda.synchronize()
        t = []
        for _ in range(reps):
            t0 = time.perf_counter()
            n_.forward_device(x)
            t.append(time.perf_counter() - t0)
            torch.cuda.synchronize()
        return sorted(t)[len(t) // 2] * 1e3
    t_nat, t_py = issue_ms(net), issue_ms(net2)
    print('host time to issue one batch-%d step: native executor %.3f ms, Python loop %.3f ms' % (meta['B'], t_nat, t_py))
    assert t_nat < 1.15 * t_py          # measured 0.40 vs 0.43 and 0.62 vs 0.66 ms; a timing on a shared host gets a margin, the equalities above do not


@pytest.mark.parametrize('case', [(2, 19, 23, 'bn_relu'), (1, 138, 138, 'bn_relu'), (3, 8, 16, 'bias'), (1, 5, 7, 'leaky'), (2, 40, 33, 'bn_relu')])
def test_patch_kernel_matches_torch(case):
    """csrc/patch.hip (YMI_DCNP_PATCH_C64): 3x3 / stride 1 / pad 1, 64 -> 64, the input patch of an 8 x 16 tile in LDS and the filters in
    registers — against torch's fp32 convolution (ragged edge tiles, a map smaller than a tile, exactly one tile, more tiles than
    blocks), with the magnitude bound it reports, and bit-reproducible."""
    import torch.nn as nn
    from gpu_utils import run_conv
    from yolact_amd import _lib as L
    B, H, W, mode = case
    g = torch.Generator().manual_seed(B * 1000 + H)
    x = torch.randn(B, 64, H, W, generator=g) * 3.0
    w = torch.randn(64, 64, 3, 3, generator=g) * 0.05
    bias, bn, act = None, None, L.ACT_RELU
    if mode == 'bn_relu':
        bn = nn.BatchNorm2d(64).eval()
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5, generator=g); bn.bias.normal_(0, 0.1, generator=g)
            bn.running_mean.normal_(0, 0.1, generator=g); bn.running_var.uniform_(0.5, 1.5, generator=g)
    elif mode == 'bias':
        bias, act = torch.randn(64, generator=g), L.ACT_NONE
    else:
        bias, act = torch.randn(64, generator=g), L.ACT_LEAKY01
    tile = L.DCNP_PATCH_C64 | L.TILE_H2 | L.TILE_DCNP
    got = run_conv(x, w, bias, bn, 1, 1, act, tile=tile)
    amax = run_conv.last_amax[1]
    with torch.no_grad():
        ref = torch.nn.functional.conv2d(x.double(), w.double(), bias.double() if bias is not None else None, padding=1)
        if bn is not None:
            ref = bn.double()(ref)
            bn.float()
        ref = torch.relu(ref) if act == L.ACT_RELU else torch.nn.functional.leaky_relu(ref, 0.1) if act == L.ACT_LEAKY01 else ref
    err = (got.double() - ref).abs().max().item() / max(1.0, ref.abs().max().item())
    assert err < 2e-6, err
    assert abs(amax - got.abs().max().item()) <= 1e-6 * max(1.0, amax)
    again = run_conv(x, w, bias, bn, 1, 1, act, tile=tile)
    assert torch.equal(again, got)
    # the tile is refused for anything but its shape (an explicit request a kernel cannot honour is an error, never another kernel)
    with pytest.raises(RuntimeError):
        run_conv(torch.randn(1, 64, 9, 9), torch.randn(64, 64, 3, 3), None, None, 2, 1, L.ACT_NONE, tile=tile)
    with pytest.raises(RuntimeError):
        run_conv(torch.randn(1, 128, 9, 9), torch.randn(64, 128, 3, 3), None, None, 1, 1, L.ACT_NONE, tile=tile)


def test_patch_kernel_speed_on_the_layer_it_was_built_for():
    """layer0.x.conv2 at batch 8 (138 x 138 x 64 -> 64): the patch kernel against the pipelined implicit-GEMM tile the shipped table
    held for this shape (printed; the tuner decides what the plan runs)."""
    import ctypes as C
    from gpu_utils import DEV as D_
    from yolact_amd import _lib as L
    from yolact_amd.engine import Packed
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(8, 138, 138, 64, generator=g)).to(D_)
    pk = Packed(torch.randn(64, 64, 3, 3, generator=g) * 0.05, None, None, 1, 1, None, D_)
    y = torch.empty(8, 138, 138, 64, device=D_)
    amax = torch.zeros(2 * 1024, device=D_)
    L.check(L.lib().ymi_amax_f32(x.data_ptr(), x.numel(), amax.data_ptr(), L.stream_ptr()))
    hp, sc2, winv = pk.h2()
    out = {}
    for name, tile in (('patch8x16c64', L.DCNP_PATCH_C64 | L.TILE_H2 | L.TILE_DCNP), ('dcnp128x64w8', L.DCNP_128x64_W8 | L.TILE_H2 | L.TILE_DCNP)):
        d = L.ConvDesc()
        d.x, d.w, d.B, d.H, d.W, d.Cin, d.ldx, d.Ho, d.Wo, d.Cout = x.data_ptr(), pk.w.data_ptr(), 8, 138, 138, 64, 64, 138, 138, 64
        d.kh, d.kw, d.stride, d.pad, d.Kpad, d.nseg, d.tile = 3, 3, 1, 1, pk.Kpad, 1, tile
        d.seg[0] = L.ConvSeg(0, 64, L.ACT_RELU, 64, 138 * 138 * 64, y.data_ptr())
        d.w_h2, d.scale_h2, d.winv_h2, d.x_amax, d.y_amax = hp.data_ptr(), sc2.data_ptr(), winv.data_ptr(), amax.data_ptr(), amax.data_ptr() + 4096
        s = L.stream_ptr()
        for _ in range(3):
            L.check(L.lib().ymi_conv2d_nhwc_f32(C.byref(d), s))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            L.check(L.lib().ymi_conv2d_nhwc_f32(C.byref(d), s))
        e1.record(); e1.synchronize()
        out[name] = (e0.elapsed_time(e1) / 20, y.clone())
        print('%-14s %.4f ms  %.1f TFLOP/s' % (name, out[name][0], 2 * 8 * 138 * 138 * 64 * 576 / out[name][0] / 1e9))
    a, b = out['patch8x16c64'][1], out['dcnp128x64w8'][1]
    assert (a - b).abs().max().item() < 2e-6 * max(1.0, b.abs().max().item())


@pytest.mark.parametrize('case', [(5, 1, 8, 138, 138, 2, 0), (3, 8, 16, 68, 68, 2, 0), (4, 16, 32, 33, 33, 2, 0), (2, 8, 16, 9, 11, 1, 1),
                                  (2, 32, 64, 16, 16, 2, 0), (1, 4, 12, 7, 9, 1, 1)])
def test_small_direct_convolution_matches_torch(case):
    """ymi_conv2d_direct_nhwc_f32: the pixel-per-thread kernel with LDS-resident filters that FastMaskIoUNet's narrow layers take at
    batch scale (1 -> 8, 8 -> 16, 16 -> 32; yolact.py:363-375, data/config.py:785-791: 3x3 / stride 2 / unpadded, also checked padded /
    stride 1), and the general kernel every other shape still takes — against torch's fp32 convolution."""
    import ctypes as C
    from yolact_amd import _lib as L
    N, Cin, Cout, H, W, stride, pad = case
    g = torch.Generator().manual_seed(Cin * 100 + Cout)
    x = torch.randn(N, Cin, H, W, generator=g)
    wt = torch.randn(Cout, Cin, 3, 3, generator=g) / (3.0 * Cin ** 0.5)
    b = torch.randn(Cout, generator=g)
    Ho, Wo = (H + 2 * pad - 3) // stride + 1, (W + 2 * pad - 3) // stride + 1
    co4 = (Cout + 3) // 4 * 4
    wp = torch.zeros(9 * Cin, co4)
    wp[:, :Cout] = wt.permute(2, 3, 1, 0).reshape(9 * Cin, Cout)
    xd, wd, bd = x.permute(0, 2, 3, 1).contiguous().to(DEV), wp.to(DEV), b.to(DEV)
    y = torch.full((N, Ho, Wo, Cout), float('nan'), device=DEV)
    L.check(L.lib().ymi_conv2d_direct_nhwc_f32(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), y.data_ptr(), N, H, W, Cin, Ho, Wo, Cout, 3, 3,
                                               stride, pad, 1, L.stream_ptr()))
    ref = torch.relu(torch.nn.functional.conv2d(x.double(), wt.double(), b.double(), stride, pad)).permute(0, 2, 3, 1)
    err = (y.cpu().double() - ref).abs().max().item() / max(1.0, ref.abs().max().item())
    assert err < 2e-6 and not torch.isnan(y).any(), err


def test_after_detect_hook_runs_on_detects_stream_and_changes_nothing():
    """Yolact.forward_device(x, after_detect=f): f sees Detect's outputs with Detect's stream current (the side stream of the two-stream
    plan), before the prototypes exist; what it enqueues there (the data-parallel record gather) is covered by the plan's final join.
    The outputs are the ones a plain forward_device returns."""
    from gpu_utils import build_net
    from helpers import case_images, load_golden
    from yolact_amd import parallel
    meta, _ = load_golden('r50_dense')
    x = case_images(meta).to(DEV)
    net = build_net(meta)
    ref = net.forward_device(x)
    ref_rec = parallel.pack_records(ref).clone()
    plan = net.plan_for(x)
    seen = {}

    def hook(out):
        seen['stream'] = torch.cuda.current_stream().cuda_stream
        seen['keys'] = sorted(out.keys())
        return parallel.pack_records(out).clone()          # a consumer of the records, enqueued on Detect's stream

    got = net.forward_device(x, after_detect=hook)
    rec = got.pop('after_detect')
    torch.cuda.synchronize()
    assert 'proto' not in seen['keys'] and 'rec' in seen['keys']
    if plan.stream_b is not None and plan.two_streams:
        assert seen['stream'] == plan.stream_b.cuda_stream != torch.cuda.current_stream().cuda_stream
    assert torch.equal(rec, ref_rec)
    for k in ('count', 'box', 'score', 'cls', 'coef', 'proto'):
        assert torch.equal(got[k], ref[k]), k
    # and through the sharded entry point (world 1: the gather is a passthrough enqueued from inside the forward)
    sh = net.forward_sharded(x)
    base = net.detect.finish(ref, ref['proto'], net)
    for a, r in zip(sh, base):
        assert (a['detection'] is None) == (r['detection'] is None)
        if r['detection'] is not None:
            for k in ('box', 'score', 'class'):
                assert torch.equal(a['detection'][k], r['detection'][k]), k
