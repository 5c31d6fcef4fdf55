"""Round 6 on the MI355X: plan slots / overlapping batches, the side-stream pool."""
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = 'cuda:0'


def _net():
    from test_gpu_batch_parity import _build
    return _build('yolact_resnet50_config', 0, 0.04)


def test_two_plan_slots_overlap_on_two_streams_and_agree():
    """Yolact.forward_device(slot=): two plan instances of one model (own arenas, head buffers, Winograd and Detect workspaces) driven
    alternately on two HIP streams — what bench.py --step-overlap 2 does — produce, batch after batch, the records a single plan on one
    stream produces (bit for bit: same kernels, same table), also when the two streams run ahead of each other."""
    from yolact_amd import parallel
    from yolact_amd.utils.synth import synth_images
    net, _ = _build_cached()
    xs = [synth_images(2, 550, 550, seed=500 + i).to(DEV) for i in range(4)]
    with torch.no_grad():
        ref = [parallel.pack_records(net.forward_device(x)).clone() for x in xs]
        torch.cuda.synchronize()
        streams = [torch.cuda.current_stream(), torch.cuda.Stream()]
        got = [None] * 8
        for rep in range(2):
            for i, x in enumerate(xs):
                slot = i & 1
                with torch.cuda.stream(streams[slot]):
                    got[4 * rep + i] = parallel.pack_records(net.forward_device(x, slot=slot)).clone()
        torch.cuda.synchronize()
    for i in range(8):
        assert torch.equal(got[i], ref[i % 4]), i
    p0, p1 = net.plan_for(xs[0], 0), net.plan_for(xs[0], 1)
    assert p0 is not p1 and p0.loc.data_ptr() != p1.loc.data_ptr()
    assert p0.stream_b is not p1.stream_b                      # the pool hands different side streams to consecutive plans ...


def test_batch_pipeline_matches_the_single_plan_path():
    """yolact_amd.pipeline.BatchPipeline (what bench.py times by default): submit() rotates plan slots / streams; every batch's records
    equal the single-plan path's, `done` events order the host against each batch, after_detect runs once per batch with the slot set."""
    from yolact_amd import parallel
    from yolact_amd.pipeline import BatchPipeline
    from yolact_amd.utils.synth import synth_images
    net, _ = _build_cached()
    xs = [synth_images(2, 550, 550, seed=500 + i).to(DEV) for i in range(5)]
    with torch.no_grad():
        ref = [parallel.pack_records(net.forward_device(x)).clone() for x in xs]
        torch.cuda.synchronize()
        for depth, fork, slots in ((2, True, [0, 1, 0, 1, 0]), (4, False, [0, 1, 2, 3, 0]), (3, False, [0, 1, 2, 0, 1])):
            pipe = BatchPipeline(net, depth)        # depth 2: every plan forks its side stream; 3, 4: un-forked plans, one stream each
            assert pipe.fork is fork and len(set(s.cuda_stream for s in pipe.streams)) == depth
            pipe.warm(xs[0])
            seen = []
            outs = [pipe.submit(x, after_detect=lambda o: seen.append(pipe.current_slot)) for x in xs]
            recs = []
            for o in outs:
                o['done'].synchronize()
                recs.append(parallel.pack_records(o).clone())
            pipe.synchronize()
            assert seen == slots and [o['slot'] for o in outs] == seen
            for r, q in zip(recs, ref):
                assert torch.equal(r, q)
            assert all(net.plan_for(xs[0], k).overlap for k in range(depth))       # the fork switch is restored after every submit
        assert BatchPipeline(net).depth == 4 and BatchPipeline(net).fork is False
    for bad in ((0, None), (5, None), (3, True)):     # more than four busy streams: measured slower than depth 1, refused
        with pytest.raises(ValueError):
            BatchPipeline(net, bad[0], fork=bad[1])


def test_side_stream_pool_is_bounded():
    """engine._side_stream: at most YOLACT_AMD_SIDE_STREAMS (2) side streams per device however many plans a process builds (the 4th
    plan of a process used to get a stream on the main stream's hardware queue and ran 1.5x slower than one stream)."""
    from yolact_amd import engine
    dev = torch.device(DEV)
    got = {id(engine._side_stream(dev)) for _ in range(12)}
    assert 1 <= len(got) <= 2


_cache = {}


def _build_cached():
    if 'n' not in _cache:
        _cache['n'] = _net()
    return _cache['n']
