"""CPU-side checks of the host layer: checkpoint layout, C-ABI symbols, config parity, error behaviour."""
import ctypes
import os
import re

import pytest
import torch

from helpers import ALL_CASES, load_golden, case_state_dict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _make_net(config):
    import yolact_amd
    yolact_amd.set_cfg(config)
    from yolact_amd.yolact import Yolact
    return Yolact()


@pytest.mark.parametrize('name', ['r50_dense', 'r101_base', 'darknet53', 'im700', 'plus_r50', 'plus_base', 'im400'])
def test_state_dict_layout_equals_reference(name):
    """Our parameter containers expose exactly the reference's keys and shapes, so reference
    checkpoints load unchanged."""
    meta, _ = load_golden(name)
    net = _make_net(meta['config'])
    ours = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    ref = {k: tuple(s) for k, s in meta['keys']}
    assert set(ours) == set(ref), (sorted(set(ours) - set(ref))[:5], sorted(set(ref) - set(ours))[:5])
    assert ours == ref
    net.load_state_dict_compat(case_state_dict(meta))     # strict load works


def test_load_weights_drops_legacy_keys(tmp_path):
    meta, _ = load_golden('r50_dense')
    net = _make_net(meta['config'])
    sd = case_state_dict(meta)
    sd['backbone.layer1.0.conv1.weight'] = torch.zeros(1)            # legacy name (yolact.py:481-483)
    sd['fpn.downsample_layers.2.weight'] = torch.zeros(256, 256, 3, 3)  # surplus v1.0 layer (yolact.py:486-489)
    p = tmp_path / 'yolact_resnet50_54_800000.pth'
    torch.save(sd, p)
    net.load_weights(str(p))


def test_priors_match_reference_digest():
    """make_priors_host (python doubles -> fp32) reproduces the reference's prior boxes bit for bit."""
    from helpers import check_digest, case_cfg
    from yolact_amd.config import make_priors_host
    for name, shapes in (('r50_dense', [69, 35, 18, 9, 5]), ('im700', [88, 44, 22, 11, 6]),
                         ('plus_r50', [69, 35, 18, 9, 5])):
        meta, arrays = load_golden(name)
        cfg = case_cfg(meta)
        bb = cfg.backbone
        data = []
        for lvl, s in enumerate(shapes):
            data += make_priors_host(s, s, bb.pred_scales[lvl], bb.pred_aspect_ratios[lvl], cfg.max_size, bb)
        pri = torch.tensor(data, dtype=torch.float32).view(-1, 4)
        check_digest(pri, meta, arrays, 'priors', rtol=0, atol=0)


def test_cabi_library_exports_every_declared_symbol():
    """The shared library loads and exports every function include/yolact_amd.h declares (no compute here)."""
    from yolact_amd import _lib as L
    hdr = open(os.path.join(ROOT, 'include', 'yolact_amd.h')).read()
    declared = set(re.findall(r'\b(ymi_[a-z0-9_]+)\s*\(', hdr))
    declared -= {'ymi_conv_seg', 'ymi_conv_desc', 'ymi_detect_desc', 'ymi_dcn_desc'}
    assert os.path.exists(L.LIB_PATH), 'build first: python -c "import __graft_entry__ as g; g.build()"'
    lib = ctypes.CDLL(L.LIB_PATH)
    for sym in sorted(declared):
        assert hasattr(lib, sym), 'missing symbol ' + sym
    assert declared == {s for s, _, _ in L.SYMBOLS}, declared ^ {s for s, _, _ in L.SYMBOLS}
    assert L.lib().ymi_abi_version() == L.ABI_VERSION
    assert L.lib().ymi_strerror(-2).decode().startswith('shape')


def test_struct_sizes_match_header():
    """ctypes mirrors have the C layout (checked against sizes computed from the header's field lists)."""
    from yolact_amd import _lib as L
    assert ctypes.sizeof(L.ConvSeg) == 32
    assert ctypes.sizeof(L.ConvDesc) == 5 * 8 + 22 * 4 + 3 * 32 + 16 + 8 + 5 * 8 + 8      # + fp16x2 planes / scales / amax slots
    assert ctypes.sizeof(L.DcnDesc) == ctypes.sizeof(L.ConvDesc) + 24        # offmask, ldo, mask_is_prob, om_layout, pad
    assert ctypes.sizeof(L.DetectDesc) == 4 * 8 + 7 * 4 + 2 * 4 + 4 + 2 * 4 + 14 * 8


def test_product_path_rejects_cpu_tensors():
    """No CPU fallback: CPU inputs raise instead of silently computing somewhere else."""
    net = _make_net('yolact_resnet50_config')
    from yolact_amd.layers.detection import Detect
    d = Detect(81, 0, 200, 0.05, 0.5)
    assert d.use_fast_nms is False and d.use_cross_class_nms is False       # the reference's defaults (detection.py:29-30)
    # traditional NMS (the reference's constructor default) is CPU/Cython and off the hot path: the engine says so ONCE and
    # runs Fast NMS (so a plain Yolact()(x) works); YOLACT_AMD_STRICT_NMS=1 makes it a hard error instead
    preds = {'loc': torch.zeros(1, 8, 4), 'conf': torch.zeros(1, 8, 81), 'mask': torch.zeros(1, 8, 32), 'priors': torch.zeros(8, 4)}
    Detect._warned_traditional = False
    with pytest.warns(UserWarning, match='use_fast_nms'):
        with pytest.raises(RuntimeError, match='GPU'):
            d(preds, None)
    os.environ['YOLACT_AMD_STRICT_NMS'] = '1'
    try:
        with pytest.raises(NotImplementedError, match='use_fast_nms'):
            d(preds, None)
        with pytest.raises(NotImplementedError, match='use_fast_nms'):
            net(torch.zeros(1, 3, 550, 550))
    finally:
        del os.environ['YOLACT_AMD_STRICT_NMS']
    d.use_fast_nms = True                                                    # eval.py:871
    net.detect.use_fast_nms = True
    with pytest.raises(RuntimeError, match='GPU'):
        net(torch.zeros(1, 3, 550, 550))
    with pytest.raises(RuntimeError, match='GPU'):
        d({'loc': torch.zeros(1, 8, 4), 'conf': torch.zeros(1, 8, 81), 'mask': torch.zeros(1, 8, 32),
           'priors': torch.zeros(8, 4)}, None)
    with pytest.raises(ValueError):
        Detect(81, 0, 200, 0.05, 0.0)      # detection.py:25-26
    with pytest.raises(NotImplementedError):
        net.train()


def test_product_does_not_import_oracle():
    """The product package never references oracle/ (judge rule: oracle is test infrastructure only)."""
    pkg = os.path.join(ROOT, 'yolact_amd')
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith('.py'):
                src = open(os.path.join(dp, fn)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle', src, re.M), fn


def test_struct_layout_matches_c_compiler(tmp_path):
    """Compile the header with gcc and compare sizeof/offsetof with the ctypes mirrors."""
    import subprocess
    from yolact_amd import _lib as L
    src = tmp_path / 'sz.c'
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "%s"\nint main(){printf("%%zu %%zu %%zu %%zu %%zu %%zu %%zu %%zu %%zu %%zu %%zu %%zu %%zu '
                   '%%zu %%zu %%zu %%zu %%zu %%zu %%zu %%zu %%zu %%zu\\n",'
                   'sizeof(ymi_conv_seg),sizeof(ymi_conv_desc),sizeof(ymi_dcn_desc),sizeof(ymi_detect_desc),'
                   'offsetof(ymi_conv_desc,seg),offsetof(ymi_conv_desc,B),offsetof(ymi_detect_desc,scores_t),'
                   'offsetof(ymi_dcn_desc,offmask),sizeof(ymi_wino_desc),offsetof(ymi_wino_desc,u_x3),'
                   'sizeof(ymi_jpeg_info),offsetof(ymi_jpeg_info,coef_count),offsetof(ymi_jpeg_info,dw),'
                   'offsetof(ymi_wino_desc,x_up),offsetof(ymi_wino_desc,up_relu),'
                   'sizeof(ymi_stem_desc),offsetof(ymi_stem_desc,kpad),'
                   'offsetof(ymi_dcn_desc,om_layout),offsetof(ymi_wino_desc,proj_w_h2),offsetof(ymi_wino_desc,proj_cout),'
                   'sizeof(ymi_chain_desc),offsetof(ymi_chain_desc,M),offsetof(ymi_chain_desc,act_a));return 0;}'
                   % os.path.join(ROOT, 'include', 'yolact_amd.h'))
    exe = tmp_path / 'sz'
    subprocess.run(['gcc', str(src), '-o', str(exe)], check=True)
    got = [int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    want = [ctypes.sizeof(L.ConvSeg), ctypes.sizeof(L.ConvDesc), ctypes.sizeof(L.DcnDesc), ctypes.sizeof(L.DetectDesc),
            L.ConvDesc.seg.offset, L.ConvDesc.B.offset, L.DetectDesc.scores_t.offset, L.DcnDesc.offmask.offset,
            ctypes.sizeof(L.WinoDesc), L.WinoDesc.u_x3.offset, ctypes.sizeof(L.JpegInfo), L.JpegInfo.coef_count.offset,
            L.JpegInfo.dw.offset, L.WinoDesc.x_up.offset, L.WinoDesc.up_relu.offset,
            ctypes.sizeof(L.StemDesc), L.StemDesc.kpad.offset,
            L.DcnDesc.om_layout.offset, L.WinoDesc.proj_w_h2.offset, L.WinoDesc.proj_cout.offset,
            ctypes.sizeof(L.ChainDesc), L.ChainDesc.M.offset, L.ChainDesc.act_a.offset]
    assert got == want, (got, want)


def test_new_entries_validate_their_descriptors_before_any_launch():
    """ymi_stem_pool_f32 / the fused-upsampling form of ymi_conv3x3_winograd_f32 reject bad descriptors with
    the documented codes (-3 null, -1 argument, -2 shape) — checked without a GPU: the validation comes before any HIP call."""
    from yolact_amd import _lib as L
    lib = L.lib()
    assert lib.ymi_stem_pool_f32(None, None) == -3
    buf = (ctypes.c_float * 64)()
    p = ctypes.addressof(buf)
    p16 = (p + 15) & ~15
    d = L.StemDesc()
    assert lib.ymi_stem_pool_f32(ctypes.byref(d), None) == -3           # every pointer NULL
    d.x, d.y, d.w_h2, d.scale_h2, d.bias = p16, p16, p16, p16, p16
    d.B, d.H, d.W, d.cout_pad, d.kpad = 1, 5, 64, 128, 224
    assert lib.ymi_stem_pool_f32(ctypes.byref(d), None) == -1           # smaller than the 7x7 filter
    d.H, d.kpad = 64, 196
    assert lib.ymi_stem_pool_f32(ctypes.byref(d), None) == -2           # filters not in the Kpad-224 plane layout
    w = L.WinoDesc()
    w.u, w.V, w.M, w.y, w.x_up = p16, p16, p16, p16, p16
    w.B, w.H, w.W, w.C, w.Cout, w.m = 1, 8, 8, 32, 32, 2
    assert lib.ymi_conv3x3_winograd_f32(ctypes.byref(w), None) == -2    # fused upsampling: F(4x4) only
    w.m, w.H = 4, 9
    assert lib.ymi_conv3x3_winograd_f32(ctypes.byref(w), None) == -2    # ... and an even output size


@pytest.mark.parametrize('config,size,gflop,P', [
    ('yolact_resnet50_config', 550, 118.28, 19248), ('yolact_base_config', 550, 164.68, 19248),
    ('yolact_darknet53_config', 550, 154.71, 19248), ('yolact_im700_config', 700, 262.93, 30963),
    ('yolact_plus_resnet50_config', 550, 141.38, 57744)])
def test_plan_graph_flops_match_reference_hooks(config, size, gflop, P):
    """Dry-run plan construction (no launches): the conv graph the engine builds has exactly the conv FLOPs that
    forward hooks measured on the reference model and the reference's prior count."""
    from yolact_amd.engine import Plan
    net = _make_net(config)
    plan = Plan(net, 1, size, size, torch.device('cpu'))
    assert abs(plan.conv_flops() / 1e9 - gflop) < 0.02, plan.conv_flops() / 1e9
    assert plan.P == P
    assert plan.arena.total_bytes() < 400e6       # per-image working set stays inside the 256 MB-class cache budget


def _gloo_worker(rank, world, port, q):
    import torch.distributed as dist
    from yolact_amd import parallel
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank, world_size=world)
    B, cap, D = 3, 5, 4
    g = torch.Generator().manual_seed(100 + rank)
    out = dict(count=torch.tensor([2, 0, 5], dtype=torch.int32) if rank == 0 else torch.tensor([1, 3, 0], dtype=torch.int32),
               box=torch.rand(B, cap, 4, generator=g), score=torch.rand(B, cap, generator=g),
               cls=torch.randint(0, 80, (B, cap), generator=g), coef=torch.rand(B, cap, D, generator=g))
    rec = parallel.pack_records(out)
    allrec = parallel.gather_records(rec, dst=0)
    if rank == 0:
        dets = parallel.unpack_records(allrec, D)
        ok = len(dets) == world * B and dets[1] is None and dets[5] is None
        ok = ok and dets[0]['box'].shape == (2, 4) and dets[3]['score'].shape == (1,) and dets[4]['mask'].shape == (3, D)
        ok = ok and torch.equal(dets[0]['box'], out['box'][0, :2]) and torch.equal(dets[2]['class'], out['cls'][2, :5])
        q.put(bool(ok))
    else:
        assert allrec is None
    lo, hi = parallel.shard_range(13, rank, world)
    assert (lo, hi) == ((0, 7) if rank == 0 else (7, 13))
    # uneven shards (13 images over 2 ranks = 7 + 6): every rank pads to `per` rows, dst trims to the global batch
    per = 7
    mine = torch.arange((hi - lo) * rec.shape[1], dtype=torch.float32).view(hi - lo, rec.shape[1]) + 1000 * rank
    allr = parallel.gather_records(mine, dst=0, rows_per_rank=per, n_items=13)
    if rank == 0:
        assert allr.shape == (13, rec.shape[1]) and torch.equal(allr[:7], mine) and float(allr[7, 0]) == 1000.0
    # mismatched sizes without rows_per_rank: an error on every rank, not undefined behaviour inside the collective
    try:
        parallel.gather_records(mine, dst=0)
        raise SystemExit('size mismatch not detected')
    except RuntimeError as e:
        assert 'different numbers of records' in str(e)
    # Yolact.forward_sharded's two halves with the persistent-buffer gatherer, two steps into the same receive buffer: the
    # first step's results must survive the second (round-3 advisor: they used to be views of the buffer), local images carry
    # their prototypes, remote ones an explicit None that postprocess() refuses
    gat = parallel.RecordGatherer(0)
    steps = []
    for step in range(2):
        gs = torch.Generator().manual_seed(7 + 10 * step + rank)
        outs = dict(count=torch.tensor([2, 1], dtype=torch.int32), box=torch.rand(2, cap, 4, generator=gs),
                    score=torch.rand(2, cap, generator=gs), cls=torch.randint(0, 80, (2, cap), generator=gs),
                    coef=torch.rand(2, cap, D, generator=gs), proto=torch.rand(2, 6, 6, D, generator=gs))
        allr = gat(parallel.pack_records(outs), 2, n_items=4, force_collective=True)
        if rank == 0:
            res = parallel.assemble_sharded(allr, outs, 0, 2, D, net='net')
            steps.append((res, [r['detection']['box'].clone() for r in res], outs))
    if rank == 0:
        (res0, boxes0, outs0), (res1, _, _) = steps
        buf = next(iter(gat._out.values()))
        for r, b0 in zip(res0, boxes0):
            assert torch.equal(r['detection']['box'], b0), 'step-0 results changed when step 1 reused the receive buffer'
            assert r['detection']['box'].untyped_storage().data_ptr() != buf.untyped_storage().data_ptr()
        assert torch.equal(res0[0]['detection']['box'], outs0['box'][0, :2])
        assert res0[0]['detection']['proto'] is outs0['proto'][0] or torch.equal(res0[0]['detection']['proto'], outs0['proto'][0])
        assert res0[2]['detection']['proto'] is None and res0[3]['detection']['proto'] is None and res0[2]['net'] == 'net'
        from yolact_amd.layers.output_utils import postprocess
        import yolact_amd
        yolact_amd.set_cfg('yolact_resnet50_config')
        try:
            postprocess(res0, 50, 50, batch_idx=2)
            raise SystemExit('postprocess accepted a detection without prototypes')
        except RuntimeError as e:
            assert 'another rank' in str(e)
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_gather_world2_gloo():
    """The one collective of the path (gather of fixed-size detection records), world size 2 on CPU/gloo."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


@pytest.mark.parametrize('config,anchors', [('yolact_resnet50_config', 3), ('yolact_plus_resnet50_config', 9)])
def test_head_gemm_is_padded_for_vector_stores_but_accounted_algorithmically(config, anchors):
    """Class rows are stored with a stride of 84 floats (81 logits + 3 zero-filter columns per anchor) so that every segment
    of the head GEMM starts 16-byte aligned; the descriptor's cout_alg keeps the FLOP accounting on the real columns and
    Detect is handed the stride."""
    from yolact_amd.engine import Plan
    net = _make_net(config)
    plan = Plan(net, 2, 550, 550, torch.device('cpu'))
    assert plan.Ccls == 81 and plan.conf_ld == 84 and tuple(plan.conf.shape) == (2, plan.P, 84)
    heads = [d for n, d in plan.conv_meta if n.endswith('.out')]
    assert len(heads) == 5
    for d in heads:
        assert d.Cout == anchors * (4 + 32 + 84) and d.cout_alg == anchors * (4 + 32 + 81)
        assert d.nseg == 3
        segs = [d.seg[i] for i in range(3)]
        assert [s.n0 for s in segs] == [0, anchors * 4, anchors * (4 + 84)] or \
               [s.n1 - s.n0 for s in segs].count(anchors * 84) == 1            # loc | conf (padded) | coef in some order
        for s in segs:
            assert s.n0 % 4 == 0 and (s.n1 - s.n0) % 4 == 0, 'segments must start and end on float4 boundaries'


def _gloo8_worker(rank, world, port, q):
    import torch.distributed as dist
    from yolact_amd import parallel
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank, world_size=world)
    B, cap, D = 13, 5, 4                     # 13 images over 8 ranks: shards of 2,2,2,2,2,2,1,0 images
    L_ = 1 + cap * (6 + D)
    g = torch.Generator().manual_seed(7)
    full = dict(count=torch.randint(0, cap + 1, (B,), generator=g).to(torch.int32), box=torch.rand(B, cap, 4, generator=g),
                score=torch.rand(B, cap, generator=g), cls=torch.randint(0, 80, (B, cap), generator=g),
                coef=torch.rand(B, cap, D, generator=g), proto=torch.rand(B, 3, 3, D, generator=g))
    calls = []

    class FakeNet:                            # stands in for Yolact: the device forward is the GPU part, not under test here
        class detect:
            top_k, use_cross_class_nms = 200, False

        def forward_device(self, xs, after_detect=None):
            # like Yolact.forward_device: the caller's consumer of the records (sharded_forward's gather) is invoked from INSIDE the
            # forward, behind Detect and before the prototypes exist; ranks with an empty shard enter the same collective from outside
            lo = int(xs[0, 0, 0, 0])
            calls.append((lo, xs.shape[0]))
            out = {k: v[lo:lo + xs.shape[0]] for k, v in full.items() if k != 'proto'}
            assert after_detect is not None, 'sharded_forward must hand its gather to a forward that accepts after_detect'
            out['after_detect'] = after_detect(out)
            out['proto'] = full['proto'][lo:lo + xs.shape[0]]
            return out
    import yolact_amd
    yolact_amd.set_cfg('yolact_resnet50_config')
    yolact_amd.cfg.max_num_detections = cap
    x = torch.arange(B, dtype=torch.float32).view(B, 1, 1, 1).expand(B, 3, 4, 4).contiguous()
    gat = parallel.RecordGatherer(0)
    net = FakeNet()
    for step in range(3):                     # persistent buffers: the same storage every step
        rec, mine = parallel.sharded_forward(net.forward_device, x, D, gat)
        if rank == 0:
            assert rec.shape == (B, L_)
            if step == 0:
                ptr = rec.data_ptr()
            assert rec.data_ptr() == ptr, 'gather buffers must be allocated once'
    lo, hi = parallel.shard_range(B, rank, world)
    assert (hi - lo) == [2, 2, 2, 2, 2, 2, 1, 0][rank]
    assert calls == ([(lo, hi - lo)] * 3 if hi > lo else [])
    # round 6: the PER-RANK SHARD form (n_global): a rank holds only its own images (rank 7: none) — same records on the root, same
    # persistent buffer; a shard of the wrong length is refused before any collective is entered
    rec2, mine2 = parallel.sharded_forward(net.forward_device, x[lo:hi].contiguous(), D, gat, n_global=B)
    if rank == 0:
        assert rec2.data_ptr() == ptr and torch.equal(rec2, rec)
    else:
        assert rec2 is None
    assert (mine2 is None) == (hi == lo)
    try:
        parallel.sharded_forward(net.forward_device, x[:hi - lo + 1].contiguous(), D, gat, n_global=B)
        raise SystemExit('a shard of the wrong length was accepted')
    except ValueError as e:
        assert 'owns images' in str(e)
    # strong scaling as BASELINE configs[1] runs on 8 GPUs: a FIXED global batch of 8 -> one image per rank, 8 records on the root
    x8 = torch.arange(8, dtype=torch.float32).view(8, 1, 1, 1).expand(8, 3, 4, 4).contiguous()
    lo8, hi8 = parallel.shard_range(8, rank, world)
    assert hi8 - lo8 == 1
    rec8, _ = parallel.sharded_forward(net.forward_device, x8[lo8:hi8].contiguous(), D, parallel.RecordGatherer(0), n_global=8)
    if rank == 0:
        assert rec8.shape == (8, L_) and torch.equal(rec8[:, 0], full['count'][:8].to(torch.float32))
    if rank == 0:
        dets = parallel.unpack_records(rec, D)
        ok = len(dets) == B
        for b in range(B):
            n = int(full['count'][b])
            if n == 0:
                ok = ok and dets[b] is None
            else:
                ok = ok and torch.equal(dets[b]['box'], full['box'][b, :n]) and torch.equal(dets[b]['score'], full['score'][b, :n])
                ok = ok and torch.equal(dets[b]['class'], full['cls'][b, :n]) and torch.equal(dets[b]['mask'], full['coef'][b, :n])
        q.put(bool(ok))
    else:
        assert rec is None
    cpus = parallel.pin_rank_affinity(rank, world)
    assert cpus is None or len(cpus) >= 1
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_world8_uneven_shards_gloo():
    """BASELINE's 8-GPU layout on CPU/gloo: 13 images over 8 ranks (shards 2,2,2,2,2,2,1,0 — one rank idle), the sharded
    forward + ONE gather into persistent buffers, three steps, per-rank CPU affinity.  The device forward is faked: what is under
    test is everything of the N > 1 path that does not need a GPU."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_gloo8_worker, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_round4_entries_validate_their_descriptors_before_any_launch():
    """ymi_pointwise_chain_f32, the fused-projection form of ymi_conv3x3_winograd_f32, ymi_dcn_desc.om_layout and the 32-column /
    weight-stationary tile requests: bad descriptors get the documented codes without a GPU (validation precedes every HIP call)."""
    from yolact_amd import _lib as L
    lib = L.lib()
    assert lib.ymi_pointwise_chain_f32(None, None) == -3
    buf = (ctypes.c_float * 64)()
    p16 = (ctypes.addressof(buf) + 15) & ~15
    c = L.ChainDesc()
    assert lib.ymi_pointwise_chain_f32(ctypes.byref(c), None) == -3      # every pointer NULL
    c.x, c.y, c.w_a_h2, c.scale_a_h2, c.x_amax = p16, p16, p16, p16, p16
    c.M, c.ldx, c.ldy, c.k_a, c.n_a, c.n_b, c.cout_pad_a = 100, 64, 256, 64, 128, 64, 256
    assert lib.ymi_pointwise_chain_f32(ctypes.byref(c), None) == -1      # only 64 -> 256 (-> 64) is instantiated
    c.n_a, c.ldy = 256, 250
    assert lib.ymi_pointwise_chain_f32(ctypes.byref(c), None) == -2      # row pitch below the channel count / not a multiple of 4
    c.ldy, c.z, c.ldz = 256, p16, 64
    assert lib.ymi_pointwise_chain_f32(ctypes.byref(c), None) == -3      # second layer requested without its filters
    w = L.WinoDesc()
    w.u, w.V, w.M, w.x, w.proj_w_h2 = p16, p16, p16, p16, p16
    w.B, w.H, w.W, w.C, w.Cout, w.m = 1, 8, 8, 32, 256, 2
    assert lib.ymi_conv3x3_winograd_f32(ctypes.byref(w), None) == -2    # fused projection: F(4x4) only
    w.m, w.Cout = 4, 128
    assert lib.ymi_conv3x3_winograd_f32(ctypes.byref(w), None) == -2    # ... of a 256-channel layer
    w.Cout, w.proj_cout, w.proj_ldy = 256, 48, 48
    assert lib.ymi_conv3x3_winograd_f32(ctypes.byref(w), None) == -2    # ... to at most 32 channels
    w.proj_cout, w.proj_ldy = 32, 32
    assert lib.ymi_conv3x3_winograd_f32(ctypes.byref(w), None) == -3    # ... with its scale vector and output
    dd = L.DcnDesc()
    dd.offmask, dd.ldo, dd.om_layout = p16, 32, 2
    assert lib.ymi_dcn_v2_forward_f32(ctypes.byref(dd), None) == -1     # unknown channel order of the offset / mask tensor
    dd.ldo, dd.om_layout = 20, 0
    assert lib.ymi_dcn_v2_forward_f32(ctypes.byref(dd), None) == -2     # fewer than 27 channels per pixel


def test_mfma_overlap_lint_flags_what_it_should(tmp_path):
    """tools/check_mfma_overlap.py (the post-build ISA lint): a destination over part of SrcB / SrcC or on SrcA is
    flagged; an accumulator updated in place (vDst == SrcC) and disjoint operands are not; llvm-objdump's trailing comments are
    ignored."""
    import subprocess
    import sys
    asm = tmp_path / 'k.s'
    asm.write_text('\n'.join([
        '0000000000001900 <_Z6kernelv>:',
        '\tv_mfma_f32_16x16x32_f16 v[220:223], v[90:93], v[218:221], v[222:225]   // 000000001904: D3D50000',
        '\tv_mfma_f32_16x16x32_f16 v[134:137], v[50:53], v[138:141], v[134:137]',
        '\tv_mfma_f32_32x32x16_f16 v[0:15], v[118:121], v[0:3], 0',
        '\tv_mfma_f32_16x16x32_f16 v[4:7], v[4:7], v[8:11], v[12:15]',
        '\tv_mfma_f32_32x32x16_f16 a[0:15], v[118:121], v[0:3], a[0:15]',
        '\tv_mfma_f32_16x16x32_f16 v[20:23], v[30:33], v[40:43], 0', '']))
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'check_mfma_overlap.py'), str(asm)], capture_output=True, text=True)
    assert r.returncode == 1 and '4 offending MFMA instruction(s)' in r.stdout, r.stdout
    assert r.stdout.count('srcB') == 2 and r.stdout.count('srcC') == 1 and r.stdout.count('srcA') == 1
    ok = tmp_path / 'ok.s'
    ok.write_text('\tv_mfma_f32_16x16x32_f16 v[134:137], v[50:53], v[138:141], v[134:137]\n')
    assert subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'check_mfma_overlap.py'), str(ok)]).returncode == 0


def test_workspace_bytes_query_matches_what_the_engine_allocates():
    """ymi_workspace_bytes (ABI 7; SURVEY 8(b) "ownership: a ymi_workspace_bytes(op, shape...) query per op"): host code, no GPU.
    Every selector against the formula the Python engine itself allocates with (engine.Plan._wino_op, _apply_choice,
    layers/box_utils.mask_iou, Detect._workspace, parallel records), the error codes, and the JPEG sizes against ymi_jpeg_parse."""
    from yolact_amd import _lib as L
    lib = L.lib()
    wb = lambda what, d: lib.ymi_workspace_bytes(what, ctypes.byref(d) if d is not None else None)
    for m, B, H, W, Cin, Cout in ((4, 8, 138, 138, 256, 256), (2, 8, 18, 18, 512, 512), (4, 1, 69, 69, 256, 351), (0, 2, 5, 5, 256, 30)):
        d = L.WinoDesc()
        d.m, d.B, d.H, d.W, d.C, d.Cout = m, B, H, W, Cin, Cout
        mm = m or 2
        T, G = B * -(-H // mm) * -(-W // mm), (mm + 2) ** 2
        assert wb(L.WS_WINO_V, d) == 4 * G * T * Cin
        assert wb(L.WS_WINO_M, d) == 4 * G * T * (-(-Cout // 4) * 4)
    d = L.WinoDesc()
    d.m, d.B, d.H, d.W, d.C, d.Cout = 3, 1, 8, 8, 32, 32
    assert wb(L.WS_WINO_V, d) == -1
    c = L.ConvDesc()
    c.B, c.Ho, c.Wo, c.Cout, c.split_k = 8, 35, 35, 1024, 4
    assert wb(L.WS_SPLITK, c) == 4 * 4 * 8 * 35 * 35 * 1024
    c.split_k = 1
    assert wb(L.WS_SPLITK, c) == 0
    s = L.MaskIouShape(100, 7, 550 * 550)
    assert wb(L.WS_MASK_IOU, s) == 4 * (100 * 7 + 100 + 7)
    dd = L.DetectDesc()
    dd.B, dd.P, dd.C, dd.D, dd.top_k, dd.max_det, dd.cross_class = 8, 19248, 81, 32, 200, 100, 0
    assert wb(L.WS_DETECT_SCORES_T, dd) == 4 * 8 * 80 * 19248
    assert wb(L.WS_DETECT_PER_PRIOR, dd) == 4 * 8 * 19248
    assert wb(L.WS_DETECT_CAND, dd) == 4 * 8 * 80 * 200
    assert wb(L.WS_DETECT_REC, dd) == 4 * 8 * (1 + 100 * 38)        # 15.2 KB per image: the payload of the one gather
    dd.cross_class = 1
    assert wb(L.WS_DETECT_REC, dd) == 4 * 8 * (1 + 200 * 38)
    assert wb(L.WS_AMAX_SLOT, None) == 16 * 64 * 4
    r = L.RleShape(100, 550, 550, 0)
    assert wb(L.WS_RLE_COUNTS, r) == 4 * 100 * (550 * 550 + 1)
    r.cap = 4096
    assert wb(L.WS_RLE_COUNTS, r) == 4 * 100 * 4096
    assert wb(L.WS_WINO_V, None) == -3 and wb(99, r) == -1
    # JPEG: the sizes ymi_jpeg_parse reports are the ones the query returns
    import numpy as np
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'jpeg.npz'), allow_pickle=True)
    key = [k for k in z.files if k.startswith('jpg_')]
    assert key
    for k in key[:5]:
        data = bytes(z[k].tobytes())
        info = L.JpegInfo()
        assert lib.ymi_jpeg_parse(data, len(data), ctypes.byref(info)) == 0
        assert wb(L.WS_JPEG_COEFS, info) == 2 * info.coef_count and wb(L.WS_JPEG_PLANES, info) == info.plane_bytes


def _gloo_mask_worker(rank, world, port, q):
    """World-2 data-parallel step WITH the mask gather (round 5): every rank 'computes' its shard (deterministic stand-ins for
    forward_device / postprocess_bits_batch — there is no GPU here), the records and the bit-packed masks reach rank 0 through the
    two fixed-capacity gathers, and rank 0 feeds EVERY image of the global batch through prep_metrics-shaped code
    (eval.py:376-440: mask IoU of each image's detections against its ground truth) — local and remote images alike."""
    import torch.distributed as dist
    from yolact_amd import parallel
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank, world_size=world)
    B, cap, D, h, w = 5, 6, 4, 9, 13                       # 5 images over 2 ranks = 3 + 2 (uneven), 117 pixels = 2 words per mask
    W64 = (h * w + 63) // 64

    def image_masks(b):                                    # the "true" final masks of global image b: [n_b, h, w] float {0,1}
        g = torch.Generator().manual_seed(500 + b)
        n = 1 + (b * 2) % cap
        return (torch.rand(n, h, w, generator=g) > 0.5).float()

    def pack(m):                                           # float masks -> bits, the layout of ymi_mask_upsample_bits
        n = m.shape[0]
        flat = torch.zeros(n, W64 * 64, dtype=torch.int64)
        flat[:, :h * w] = m.reshape(n, -1).to(torch.int64)
        return (flat.view(n, W64, 64) << torch.arange(64, dtype=torch.int64)).sum(-1)

    class FakeNet:
        class detect:
            top_k, use_cross_class_nms = 200, False

        def forward_device(self, x):
            lo = int(x[0, 0, 0, 0])                        # the global index of the shard's first image rides in the pixels
            b = x.shape[0]
            g = torch.Generator().manual_seed(900 + lo)
            out = dict(count=torch.tensor([image_masks(lo + i).shape[0] for i in range(b)], dtype=torch.int32),
                       box=torch.rand(b, cap, 4, generator=g), score=torch.rand(b, cap, generator=g),
                       cls=torch.randint(0, 80, (b, cap), generator=g), coef=torch.rand(b, cap, D, generator=g),
                       proto=torch.rand(b, 3, 3, D, generator=g), lo=lo)
            return out
    net = FakeNet()
    x = torch.arange(B, dtype=torch.float32).view(B, 1, 1, 1).expand(B, 3, 4, 4).contiguous()

    def masks_fn(out):
        if out is None:
            return torch.zeros(0, cap, W64, dtype=torch.int64)
        bits = torch.zeros(out['count'].shape[0], cap, W64, dtype=torch.int64)
        for i in range(bits.shape[0]):
            m = image_masks(out['lo'] + i)
            bits[i, :m.shape[0]] = pack(m)
        return bits
    import yolact_amd.parallel as P
    P._cap_of = lambda fd: cap                              # (the stand-in has no config)
    g1, g2 = parallel.RecordGatherer(0), parallel.RecordGatherer(0)
    for step in range(2):                                   # twice into the same persistent buffers
        rec, mine, bits = parallel.sharded_forward(net.forward_device, x, D, g1, 0, masks_fn=masks_fn, mask_gatherer=g2)
        lo, hi = parallel.shard_range(B, rank, world)
        if rank != 0:
            assert rec is None and bits is None
            continue
        assert rec.shape[0] == B and bits.shape == (B, cap, W64) and bits.dtype == torch.int64
        res = parallel.assemble_sharded(rec, mine, lo, hi, D, net='net', bits=bits, mask_size=(h, w))
        assert len(res) == B
        for b in range(B):                                  # prep_metrics-shaped: IoU of every image's masks against its ground truth
            det = res[b]['detection']
            truth = image_masks(b)
            assert res[b]['mask_size'] == (h, w) and det['mask_bits'].shape == (truth.shape[0], W64)
            assert (det['proto'] is None) == (b >= hi)      # local images keep their prototypes, remote ones carry only the bits
            masks = parallel.unpack_mask_bits(det['mask_bits'], h, w)
            assert torch.equal(masks, truth), 'image %d: gathered masks differ' % b
            gt = truth[:1]
            inter = masks.view(masks.shape[0], -1) @ gt.view(1, -1).t()
            iou = inter / (masks.view(masks.shape[0], -1).sum(1, keepdim=True) + gt.sum() - inter)
            assert abs(float(iou[0, 0]) - 1.0) < 1e-6 and iou.shape == (truth.shape[0], 1)
        q.put(True)
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_mask_gather_world2_gloo():
    """forward_sharded's masks='bits' half on CPU / gloo, world 2, uneven shards: the root ends with a USABLE global batch."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_gloo_mask_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True and q.get(timeout=5) is True


def test_yolact_replicates_for_data_parallel():
    """eval.py:661 wraps the net in nn.DataParallel (CustomDataParallel, eval.py:630-634); replicate() makes one shallow copy of the
    module per device (nn.Module._replicate_for_data_parallel).  No GPU here, so this is the CPU half: the copy carries what
    forward() needs, shares the plan cache and the Detect object, and owns NO lock that would serialise replicas of different
    devices (tests/test_gpu_round5.py runs real replicas on the MI355X)."""
    import threading
    net = _make_net('yolact_resnet50_config')
    rep = net._replicate_for_data_parallel()
    assert type(rep) is type(net) and rep is not net
    assert rep._plans is net._plans and rep.detect is net.detect and rep._run_locks is net._run_locks
    a, b = torch.device('cuda', 0), torch.device('cuda', 1)
    assert rep._run_lock_for(a) is net._run_lock_for(a) and net._run_lock_for(a) is not net._run_lock_for(b)
    assert isinstance(net._run_lock_for(b), type(threading.Lock()))
    # a replica whose children were replaced by per-device copies (what replicate() does next) still resolves its structure
    for name in ('backbone', 'fpn', 'proto_net', 'prediction_layers'):
        assert name in rep._modules
    assert rep.mask_dim == net.mask_dim and rep.backbone_selected == net.backbone_selected and rep.cfg is net.cfg
    wrapped = torch.nn.DataParallel(net)                     # no GPU: DataParallel degrades to calling the module itself
    assert wrapped.module is net
    with pytest.raises(RuntimeError, match='must live on the GPU'):
        wrapped(torch.zeros(1, 3, 64, 64))


def test_outlier_guard_needs_an_amplifying_producer_not_just_tiny_columns():
    """Packed.tiny_columns (round-4 advisor, medium): a layer leaves the fp16x2 tiles only when a tiny filter column sits on a channel
    that its PRODUCER amplifies — the BN-folded outlier signature the stress test plants — and no longer on tiny columns alone, which
    is also what dead (weight-decayed) input channels of real checkpoints look like.  CPU plans, no launches."""
    import warnings
    from yolact_amd.engine import Plan
    from yolact_amd.utils.synth import plant_outlier_channels as _plant_outlier_channels, synth_state_dict
    net = _make_net('yolact_resnet50_config')
    sd = synth_state_dict([(k, tuple(v.shape)) for k, v in net.state_dict().items()], seed=0, conf_gain=0.04)
    old = os.environ.get('YOLACT_AMD_SPLIT')
    old_rb = os.environ.get('YOLACT_AMD_REBALANCE')
    os.environ['YOLACT_AMD_SPLIT'] = '2'
    os.environ['YOLACT_AMD_REBALANCE'] = '0'          # the GUARD's behaviour (round 4); the default rebalances instead: test below
    try:
        def wide_layers(state):
            net.load_state_dict_compat(state)
            with warnings.catch_warnings(record=True) as w:
                warnings.simplefilter('always')
                plan = Plan(net, 2, 550, 550, torch.device('cpu'))
            return plan.wide_layers, [x for x in w if 'OUTLIER' in str(x.message)]
        clean, warned = wide_layers(sd)
        assert clean == [] and not warned
        planted, _ = _plant_outlier_channels({k: v.clone() for k, v in sd.items()}, 12)
        wide, warned = wide_layers(planted)
        assert len(wide) >= 10 and any(n.startswith('fpn.lat') for n in wide) and 'proto.2' in wide
        assert len(warned) == 1 and 'layer1.1.conv1' in str(warned[0].message)  # the demotion is SAID, once per plan
        dead = {k: v.clone() for k, v in sd.items()}
        for key in ('backbone.layers.1.1.conv1.weight', 'fpn.lat_layers.1.weight', 'proto_net.2.weight', 'backbone.layers.2.3.conv2.weight'):
            dead[key][:, [3, 17, 40]] *= 2.0 ** -16                              # dead input channels: tiny columns, ordinary producers
        wide, warned = wide_layers(dead)
        assert wide == [] and not warned
        # round 6, the default: compensated outlier channels are REBALANCED at pack time (Plan._rebalance_outliers) — the producers'
        # folded BN scales and the consumers' filters exchange the power of two back, nothing leaves the fp16x2 tiles, nothing is
        # said; a clean checkpoint and one with dead input channels (no amplifying producer) are left exactly as they are
        os.environ.pop('YOLACT_AMD_REBALANCE', None)

        def rebalanced(state):
            net.load_state_dict_compat(state)
            with warnings.catch_warnings(record=True) as w:
                warnings.simplefilter('always')
                plan = Plan(net, 2, 550, 550, torch.device('cpu'))
            return plan, [x for x in w if 'OUTLIER' in str(x.message)]
        plan, warned = rebalanced(planted)
        assert plan.wide_layers == [] and not warned
        # (k = min(producer excess, consumer deficit), floored: the planted 2^12 comes back as 2^10 .. 2^12 per channel — a channel may
        #  keep a factor <= 4 of its own ordinary spread; what matters is that no tensor carries the planted range any more)
        assert [r[0] for r in plan.rebalanced] == ['C3', 'C4', 'C5', 'proto_net.0'] and all(r[1] == 4 and 10 <= r[2] <= 12 for r in plan.rebalanced)
        clean_plan, _ = rebalanced(sd)
        assert clean_plan.rebalanced == [] and clean_plan.wide_layers == []
        assert rebalanced(dead)[0].rebalanced == []
        # the rebalanced producers' gains are back inside the clean checkpoint's spread (x4), channel by channel
        for pk_p, pk_c in zip(plan.keepalive, clean_plan.keepalive):
            if hasattr(pk_p, 'out_gain') and pk_p.Cout == pk_c.Cout:
                gp, gc = pk_p.out_gain(), pk_c.out_gain()
                ok = gc > 0
                assert float((gp[ok] / gc[ok]).max()) <= 4.0 + 1e-6 and float((gp[ok] / gc[ok]).min()) >= 0.25 - 1e-6
    finally:
        for k_, v_ in (('YOLACT_AMD_SPLIT', old), ('YOLACT_AMD_REBALANCE', old_rb)):
            if v_ is None:
                os.environ.pop(k_, None)
            else:
                os.environ[k_] = v_


def test_native_plan_executor_host_side(tmp_path):
    """csrc/plan_exec.cpp (ABI 7) without a GPU: the ymi_plan_op layout equals the C compiler's, a list of NOP / skipped ops runs to
    completion without touching HIP, an unknown op kind is reported with its index, and a CPU-built plan translates into the op array
    (every call of the op list is one the executor knows)."""
    import subprocess
    from yolact_amd import _lib as L
    from yolact_amd.engine import Plan
    lib = L.lib()
    src = tmp_path / 'po.c'
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "%s"\nint main(){printf("%%zu %%zu %%zu %%zu %%zu\\n",sizeof(ymi_plan_op),'
                   'offsetof(ymi_plan_op,desc),offsetof(ymi_plan_op,p),offsetof(ymi_plan_op,i),offsetof(ymi_plan_op,f));return 0;}'
                   % os.path.join(ROOT, 'include', 'yolact_amd.h'))
    exe = tmp_path / 'po'
    subprocess.run(['gcc', str(src), '-o', str(exe)], check=True)
    got = [int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert got == [ctypes.sizeof(L.PlanOp), L.PlanOp.desc.offset, L.PlanOp.p.offset, L.PlanOp.i.offset, L.PlanOp.f.offset]
    arr = (L.PlanOp * 4)()
    arr[1].kind, arr[1].section = L.OP_CONV, L.SEC_PROTO          # would dereference a NULL descriptor — but its section is skipped
    failed = ctypes.c_int32(-1)
    assert lib.ymi_plan_run(arr, 0, 4, None, None, None, 0, 1 << L.SEC_PROTO, ctypes.byref(failed)) == 0
    arr[2].kind = 99
    assert lib.ymi_plan_run(arr, 0, 4, None, None, None, 0, 1 << L.SEC_PROTO, ctypes.byref(failed)) == -1 and failed.value == 2
    assert lib.ymi_plan_run(None, 0, 1, None, None, None, 0, 0, None) == -1
    assert lib.ymi_plan_run(arr, 3, 2, None, None, None, 0, 0, None) == -1
    arr[1].section = 0
    assert lib.ymi_plan_run(arr, 1, 2, None, None, None, 0, 0, ctypes.byref(failed)) == -3 and failed.value == 1     # NULL descriptor: the call's own check
    plan = Plan(_make_net('yolact_plus_resnet50_config'), 1, 550, 550, torch.device('cpu'))
    nat = plan._native_plan()
    assert nat is not None
    ops, det_idx, evs, in_idx, n = nat
    assert n == len(plan.ops) + 1 and ops[0].kind == L.OP_MEMSET and plan.ops[det_idx - 1][0] == 'detect'
    kinds = [ops[k].kind for k in range(n)]
    assert kinds.count(L.OP_DCN) == 13 and L.OP_CONV in kinds and (L.OP_STEM in kinds or L.OP_INPUT in kinds)
    assert all(ops[k + 1].section == L.SEC_PROTO for k, op in enumerate(plan.ops) if isinstance(op[2], str) and op[2].startswith('proto.'))
    assert plan._native_plan() is nat                              # cached until an op is replaced
    plan.ops[5] = ('nop', None, 'x', 'A')
    assert plan._native_plan() is not nat
