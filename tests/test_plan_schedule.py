"""Static race check of the two-stream execution plan (engine.Plan) — no GPU needed.

The plan is a flat op list with fork / join markers ('record' / 'wait' of named events) and two buffer pools.  Two HIP
streams only order their work through those events, so every pair of ops on DIFFERENT streams that touch the same
device buffer (at least one writing) must be ordered by a record -> wait chain, including the recycled arena buffers and
the hand-placed pool transfers (the projection shortcut computed on the side stream and consumed on the main one).
The check replays the op list with vector clocks and reports unordered conflicting pairs; it also checks the invariants
that make consecutive iterations safe: the side stream's first op waits on an event of the main stream, and the main
stream has joined ALL side-stream work when the plan ends.
"""
import pytest
import torch

from yolact_amd import _lib as L


def _make_net(config):
    import yolact_amd
    yolact_amd.set_cfg(config)
    from yolact_amd.yolact import Yolact
    return Yolact()


def _buffers(plan):
    """(base, bytes, label) of every allocation the ops may touch."""
    out = []
    for label, ar in (('arenaA', plan.arena), ('arenaB', getattr(plan, 'arena_b', None))):
        if ar is not None:
            out += [(b.data_ptr(), b.numel() * 4, '%s[%d]' % (label, i)) for i, b in enumerate(ar.all)]
    for label in ('loc', 'conf', 'coef'):
        t = getattr(plan, label)
        out.append((t.data_ptr(), t.numel() * 4, label))
    return out


def _owner(bufs, ptr):
    """(label, sub): arena buffers are used whole (sub None); the persistent head tensors loc / conf / coef are written
    level by level — [B, P, k] rows starting at the level's prior offset — so a write is keyed by that offset and only
    conflicts with the same level or with a reader of the whole tensor (Detect)."""
    if not ptr:
        return None                                   # the per-call proto output: fresh memory every run
    hits = [(lab, base) for base, size, lab in bufs if base <= ptr < base + max(size, 1)]
    assert len(hits) == 1, (ptr, hits)
    lab, base = hits[0]
    return (lab, ptr - base) if lab in ('loc', 'conf', 'coef') else (lab, None)


def _conflict(a, b):
    return a[0] == b[0] and (a[1] is None or b[1] is None or a[1] == b[1])


def _accesses(plan, op, bufs):
    """(reads, writes) as buffer labels."""
    fn, args, name, where = op
    lib = plan.lib
    if fn == 'input':
        return set(), {_owner(bufs, plan.in_args[1])}
    if fn == 'stem':                                   # fused stem: reads the caller's NCHW image, writes the pooled map
        return set(), {_owner(bufs, args.y)}
    if fn == 'detect':
        return {('loc', None), ('conf', None), ('coef', None)}, set()
    if fn is lib.ymi_dcn_v2_forward_f32:
        dd = args.contents
        d, extra = dd.conv, [dd.offmask]
    elif fn is lib.ymi_conv2d_nhwc_f32:
        d, extra = args.contents, []
    else:                                             # layout / pool / resize calls: (src, dst, ...)
        assert isinstance(args, tuple), name
        return {_owner(bufs, args[0])}, {_owner(bufs, args[1])}
    reads = {_owner(bufs, d.x)} | {_owner(bufs, e) for e in extra}
    if d.res_mode != L.RES_NONE:
        reads.add(_owner(bufs, d.res))
    writes = {_owner(bufs, d.seg[i].ptr) for i in range(d.nseg)}
    return reads - {None}, writes - {None}


def check_schedule(plan):
    bufs = _buffers(plan)
    pos = {'A': 0, 'B': 0}                             # ops issued so far per stream
    know = {'A': 0, 'B': 0}                            # ops of the OTHER stream known complete before the next op starts
    events, trace, problems = {}, [], []
    b_synced = False                                   # has B waited on an event of A yet?
    last_b_burst_ok = True
    for op in plan.ops:
        fn, args, name, where = op
        other = 'B' if where == 'A' else 'A'
        if fn == 'record':
            events[args] = (where, pos[where])
            continue
        if fn == 'wait':
            src, p = events[args]                      # KeyError = wait before record: a bug in itself
            if src == other:
                know[where] = max(know[where], p)
                if where == 'B':
                    b_synced = True
            continue
        if where == 'B' and pos['B'] == 0 and not b_synced:
            # iteration n+1's side-stream work must start behind a point of the main stream that itself follows the
            # join of iteration n (the main stream's program order provides the rest)
            last_b_burst_ok = False
            problems.append('the first side-stream op (%s) does not wait on the main stream' % name)
        pos[where] += 1
        r, w = _accesses(plan, op, bufs)
        trace.append((where, pos[where], know[where], name, r, w))
    for i, (s1, i1, k1, n1, r1, w1) in enumerate(trace):
        for (s2, i2, k2, n2, r2, w2) in trace[i + 1:]:
            if s1 == s2:
                continue
            shared = [x for x in w1 for y in (r2 | w2) if _conflict(x, y)] + [x for x in w2 for y in r1 if _conflict(x, y)]
            if shared and not (k2 >= i1 or k1 >= i2):
                problems.append('%s (%s#%d) and %s (%s#%d) both touch %s without an event between them'
                                % (n1, s1, i1, n2, s2, i2, sorted(set(x[0] for x in shared))))
    if know['A'] != pos['B']:
        problems.append('plan ends with %d of %d side-stream ops joined' % (know['A'], pos['B']))
    return problems, pos, last_b_burst_ok


@pytest.mark.parametrize('config,size', [('yolact_resnet50_config', 550), ('yolact_base_config', 550),
                                         ('yolact_darknet53_config', 550), ('yolact_im700_config', 700),
                                         ('yolact_plus_resnet50_config', 550)])
def test_two_stream_plan_is_race_free(config, size):
    from yolact_amd.engine import Plan
    net = _make_net(config)
    plan = Plan(net, 2, size, size, torch.device('cpu'), dry_two_streams=True)
    problems, pos, _ = check_schedule(plan)
    assert not problems, problems[:5]
    assert pos['B'] > 8 and pos['A'] > 40               # both streams carry work (P4..P7 + Detect vs the rest)
    if 'resnet' in config or 'base' in config or 'im700' in config:
        assert any(n.endswith('.down') and w == 'B' for _, _, n, w in plan.ops), 'projection shortcuts on the side stream'


def test_checker_catches_a_missing_wait():
    """Remove the wait that orders conv3 behind the side-stream shortcut: the checker must flag that pair."""
    from yolact_amd.engine import Plan
    plan = Plan(_make_net('yolact_resnet50_config'), 1, 550, 550, torch.device('cpu'), dry_two_streams=True)
    idx = [i for i, op in enumerate(plan.ops) if op[0] == 'wait' and op[1] == 'layer1.0.down']
    assert len(idx) == 1
    del plan.ops[idx[0]]
    problems, _, _ = check_schedule(plan)
    assert any('layer1.0.down' in p and 'layer1.0.conv3' in p for p in problems), problems[:3]


def test_single_stream_plan_has_no_markers():
    from yolact_amd.engine import Plan
    plan = Plan(_make_net('yolact_resnet50_config'), 1, 550, 550, torch.device('cpu'))
    assert all(op[0] not in ('record', 'wait') and op[3] == 'A' for op in plan.ops)


def test_fused_stem_and_upsample_hooks_in_the_plan(monkeypatch):
    """fp16x2 ResNet plans start with ONE 'stem' op (image -> stem -> max-pool, csrc/stem.hip) instead of the layout change, the
    7x7 conv and the max-pool, with unchanged FLOP accounting; the protonet's 2x upsampling is registered for fusion into the
    consuming conv's Winograd input transform; Darknet plans and YOLACT_AMD_FUSED_STEM=0 keep the separate launches."""
    from yolact_amd.engine import Plan
    dev = torch.device('cpu')
    plan = Plan(_make_net('yolact_resnet50_config'), 2, 550, 550, dev)
    names = [op[2] for op in plan.ops]
    assert plan.fused_stem and plan.ops[0][0] == 'stem' and 'maxpool' not in names
    assert all(op[0] != 'input' for op in plan.ops)
    sd = plan.ops[0][1]
    assert (sd.B, sd.H, sd.W, sd.kpad) == (2, 550, 550, 224)
    assert plan.conv_meta[0][0] == 'stem'
    flops_fused = plan.conv_flops()
    # the upsampling in front of proto.8: one registered (bilinear op, low-res tensor, relu) for the conv that consumes it
    assert len(plan._upsrc) == 1
    (bidx, lo_ptr, relu), = plan._upsrc.values()
    assert plan.ops[bidx][2] == 'proto.interp' and lo_ptr and relu == 1
    if plan.wino_alt:                                  # (Winograd alternatives are only built on a GPU)
        (cidx, up), = plan.wino_up.items()
        assert up == (bidx, lo_ptr, relu) and plan.ops[cidx][2].startswith('proto.')
    monkeypatch.setenv('YOLACT_AMD_FUSED_STEM', '0')
    plain = Plan(_make_net('yolact_resnet50_config'), 2, 550, 550, dev)
    pn = [op[2] for op in plain.ops]
    assert not plain.fused_stem and plain.ops[0][0] == 'input' and 'maxpool' in pn and pn[1] == 'stem'
    assert plain.conv_flops() == flops_fused
    monkeypatch.delenv('YOLACT_AMD_FUSED_STEM')
    dk = Plan(_make_net('yolact_darknet53_config'), 1, 550, 550, dev)
    assert not dk.fused_stem and dk.ops[0][0] == 'input'


@pytest.mark.parametrize('config,size,B', [('yolact_resnet50_config', 550, 1), ('yolact_resnet50_config', 550, 8),
                                           ('yolact_base_config', 550, 2), ('yolact_im700_config', 700, 1),
                                           ('yolact_plus_resnet50_config', 550, 2), ('yolact_darknet53_config', 550, 2),
                                           ('yolact_resnet50_config', 256, 3), ('yolact_resnet50_config', 1024, 1)])
def test_fused_upsample_source_outlives_its_consumer(config, size, B):
    """Round-3 advisor: the conv behind the protonet's 2x upsampling may read the LOW-RES tensor inside its Winograd input
    transform (ymi_wino_desc.x_up) — so that buffer must stay untouched from the bilinear op up to and including the consuming
    conv: no op in between may write it, and the conv's own output must not alias it (the arena used to free it right after
    the bilinear op, and only the buffer sizes of the shipped shapes kept the best-fit pool from recycling it)."""
    from yolact_amd.engine import Plan
    plan = Plan(_make_net(config), B, size, size, torch.device('cpu'), dry_two_streams=True)
    bufs = _buffers(plan)
    assert len(plan._upsrc) == 1
    (bidx, lo_ptr, _relu), = plan._upsrc.values()
    lo = _owner(bufs, lo_ptr)
    up_dst = plan.ops[bidx][1][1]                      # the bilinear op's destination = the consuming conv's input
    cidx = next(i for i in range(bidx + 1, len(plan.ops))
                if plan.ops[i][0] is plan.lib.ymi_conv2d_nhwc_f32 and plan.ops[i][1].contents.x == up_dst)
    for i in range(bidx, cidx + 1):
        op = plan.ops[i]
        if op[0] in ('record', 'wait', 'detect', 'nop'):
            continue
        _r, w = _accesses(plan, op, bufs)
        assert not any(_conflict(lo, x) for x in w), (plan.ops[i][2], 'writes the low-res source of the fused upsampling')


def test_offset_mask_filters_are_padded_and_tap_interleaved():
    """engine.pack_offmask: conv_offset_mask's 27 filters (dcn_v2.py:107-112: 18 offsets, then 9 mask logits) become 32 rows — row
    3k = dh_k, 3k + 1 = dw_k, 3k + 2 = mask_k (ymi_dcn_desc.om_layout = 1), rows 27 .. 31 zero — with the bias permuted the same way;
    without interleaving the reference's order is kept.  The filters' VALUES are untouched (same arithmetic per channel)."""
    import torch.nn as nn
    from yolact_amd.engine import pack_offmask
    torch.manual_seed(3)
    conv = nn.Conv2d(64, 27, 3, stride=2, padding=1)
    w = conv.weight.detach().permute(0, 2, 3, 1).reshape(27, -1)         # the engine's K order: (ky, kx, c)
    for inter in (True, False):
        pk = pack_offmask(conv, None, inter)
        assert (pk.Cout, pk.cout_alg, pk.stride, pk.pad, pk.kh, pk.kw) == (32, 27, 2, 1, 3, 3)
        rows = pk._wp_host[:32, :w.shape[1]]
        order = [c for k in range(9) for c in (2 * k, 2 * k + 1, 18 + k)] if inter else list(range(27))
        assert torch.equal(rows[:27], w[order]) and rows[27:].abs().max() == 0
        assert torch.equal(pk.bias.cpu()[:27], conv.bias.detach()[order]) and pk.bias.cpu()[27:].abs().max() == 0


def test_narrow_tile_candidates_follow_their_envelopes():
    """Plan.dcnp_candidates / Plan.ws_candidates (pure host logic): 32-column tiles are offered to Cout <= 32 layers and to nothing
    else; a weight-stationary candidate always carries enough K ranges for its filters to fit 64 KB of LDS, every range non-empty;
    layers with a residual or more than 64 channels get no weight-stationary candidate."""
    from yolact_amd import _lib as L
    from yolact_amd.engine import Plan

    def desc(B, H, Cin, Cout, k, res=L.RES_NONE):
        d = L.ConvDesc()
        d.B, d.H, d.W, d.Ho, d.Wo, d.Cin, d.Cout = B, H, H, H, H, Cin, Cout
        d.kh = d.kw = k
        d.stride, d.pad, d.Kpad, d.nseg, d.res_mode = 1, k // 2, k * k * Cin, 1, res
        return d
    names = lambda cands: {L.TILE_NAMES[c & 255] for c in cands}          # noqa: E731
    narrow = {'dcnp128x32w4', 'dcnp256x32w8', 'dcnp64x32w2'}
    assert names(Plan.dcnp_candidates(desc(8, 69, 128, 32, 3))) == narrow
    assert not (names(Plan.dcnp_candidates(desc(8, 69, 128, 128, 3))) & narrow)
    for d in (desc(8, 69, 128, 32, 3), desc(8, 35, 256, 32, 3), desc(8, 18, 512, 32, 3), desc(8, 138, 256, 32, 1), desc(8, 138, 64, 64, 1),
              desc(1, 138, 256, 64, 1)):
        cands = Plan.ws_candidates(d)
        assert cands, (d.Cin, d.Cout)
        for c in cands:
            name, S = L.TILE_NAMES[c & 255], max(c >> 8, 1)
            cols = int(name[2:].split('x')[1].split('w')[0])
            nk = d.Kpad // 32
            per = -(-nk // S)
            assert (cols == 32) == (d.Cout <= 32)
            assert per * cols * 128 <= 65536 and (S == 1 or per * (S - 1) < nk), (name, S, nk)
    assert Plan.ws_candidates(desc(8, 138, 64, 256, 1)) == [] and Plan.ws_candidates(desc(8, 138, 256, 64, 1, L.RES_ADD)) == []


@pytest.mark.parametrize('config,size', [('yolact_resnet50_config', 550), ('yolact_darknet53_config', 550), ('yolact_plus_base_config', 550)])
def test_early_lateral_plan_is_race_free(config, size, monkeypatch):
    """YOLACT_AMD_EARLY_LAT=1 (round 5, opt-in): the FPN laterals of the lower levels run on the side stream beside the later backbone
    stages and the top-down sum becomes an in-place pass on the main stream.  The first version of this schedule had a real race —
    the lateral's output landed in the pool buffer of the projection shortcut that the main stream's conv3 was still reading — and
    this checker is what found it."""
    from yolact_amd.engine import Plan
    monkeypatch.setenv('YOLACT_AMD_EARLY_LAT', '1')
    plan = Plan(_make_net(config), 2, size, size, torch.device('cpu'), dry_two_streams=True)
    names = [op[2] for op in plan.ops]
    assert 'fpn.add1' in names and 'fpn.add2' in names
    lat = {op[2]: op[3] for op in plan.ops if isinstance(op[2], str) and op[2].startswith('fpn.lat')}
    assert lat['fpn.lat0'] == 'A' and lat['fpn.lat1'] == 'B' and lat['fpn.lat2'] == 'B'
    assert names.index('fpn.lat2') < names.index('fpn.lat1') < names.index('fpn.lat0')     # launched as their stages finish
    problems, pos, _ = check_schedule(plan)
    assert not problems, problems[:5]
