"""The multi-GPU exchange step on real RCCL: a one-GPU box still runs the `nccl` backend with world size 1 (SURVEY 8(e):
"World-size-1 RCCL (degenerate gather) still exercises the code path").  pack -> dist.gather -> unpack must reproduce what
Detect.finish returns for the same device outputs."""
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = 'cuda:0'


def test_world1_nccl_gather_roundtrip_equals_detect_finish():
    import torch.distributed as dist
    from gpu_utils import build_net
    from helpers import case_images, load_golden
    from yolact_amd import parallel
    meta, _ = load_golden('r50_dense')
    net = build_net(meta)
    x = case_images(meta).to(DEV)
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    assert not dist.is_initialized()
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % port, rank=0, world_size=1,
                            device_id=torch.device(DEV))
    try:
        assert dist.get_backend() == 'nccl' and dist.get_world_size() == 1
        dev_out = net.forward_device(x)
        rec = parallel.pack_records(dev_out)
        # the record tensor is what the Detect selection kernel wrote itself (ymi_detect_desc.out_rec): valid rows equal the
        # cat / cast form built from the separate outputs
        assert rec is dev_out['rec']
        legacy = parallel.pack_records({k: v for k, v in dev_out.items() if k != 'rec'})
        cnt = dev_out['count'].tolist()
        L_ = 6 + dev_out['coef'].shape[2]
        for b, n in enumerate(cnt):
            assert torch.equal(rec[b, :1 + n * L_], legacy[b, :1 + n * L_]), b
        # persistent-buffer gatherer (what bench.py and Yolact.forward_sharded use): same bytes, same storage every step
        gat = parallel.RecordGatherer(0)
        g1 = gat(rec, rec.shape[0], force_collective=True)
        p1 = g1.data_ptr()
        g2 = gat(rec, rec.shape[0], force_collective=True)
        torch.cuda.synchronize()
        assert g2.data_ptr() == p1 and torch.equal(g2, rec)
        sharded = net.forward_sharded(x)
        ref0 = net.detect.finish(dev_out, dev_out['proto'], net)
        assert len(sharded) == len(ref0)
        for a, r in zip(sharded, ref0):
            assert (a['detection'] is None) == (r['detection'] is None)
            if a['detection'] is not None:
                assert torch.equal(a['detection']['box'], r['detection']['box'])
                assert torch.equal(a['detection']['class'], r['detection']['class'])
                assert a['detection']['proto'].shape == r['detection']['proto'].shape
        # masks='bits' (round 5): the owners' bit-packed masks ride a second gather (forced through the real collective here);
        # every detection of the global batch carries its final masks, local (prototypes present) or remote (prototypes None)
        import os
        from yolact_amd.layers.output_utils import postprocess, postprocess_bits
        os.environ['YOLACT_AMD_FORCE_GATHER'] = '1'
        try:
            shm = net.forward_sharded(x, masks='bits', mask_size=(97, 131))
        finally:
            os.environ.pop('YOLACT_AMD_FORCE_GATHER', None)
        torch.cuda.synchronize()
        for b, (a, r) in enumerate(zip(shm, ref0)):
            if r['detection'] is None:
                assert a['detection'] is None
                continue
            assert a['mask_size'] == (97, 131)
            classes, scores, boxes, masks = postprocess(ref0, 131, 97, batch_idx=b)
            c1, s1, b1, bits1 = postprocess_bits(shm, 131, 97, batch_idx=b)                 # local form: recomputed from the prototypes
            assert torch.equal(a['detection']['mask_bits'], bits1) and a['detection']['mask_bits'].dtype == torch.int64
            assert torch.equal(parallel.unpack_mask_bits(bits1, 97, 131), masks)
            a['detection']['proto'] = None                                                  # as if computed on another rank
            c2, s2, b2, bits2 = postprocess_bits(shm, 131, 97, batch_idx=b)
            assert torch.equal(bits2, bits1) and torch.equal(b2, boxes) and torch.equal(c2, classes) and torch.equal(s2, scores)
            with pytest.raises(RuntimeError, match='assembled for'):
                postprocess_bits(shm, 550, 550, batch_idx=b)
        same = parallel.gather_records(rec, dst=0)                         # world 1, not forced: passthrough
        assert same is rec
        got = parallel.gather_records(rec, dst=0, force_collective=True)   # the real collective, one rank
        torch.cuda.synchronize()
        assert got is not rec and got.shape == rec.shape and torch.equal(got, rec)
        # uneven-shard form: pad to 4 rows per rank, trim to the 2 real images
        got4 = parallel.gather_records(rec, dst=0, rows_per_rank=4, n_items=meta['B'], force_collective=True)
        assert torch.equal(got4, rec)
        dets = parallel.unpack_records(got, dev_out['coef'].shape[2])
        ref = net.detect.finish(dev_out, dev_out['proto'], net)
        assert len(dets) == len(ref) == meta['B']
        for d, r in zip(dets, ref):
            r = r['detection']
            assert (d is None) == (r is None)
            if d is not None:
                for k in ('box', 'score', 'class', 'mask'):
                    assert torch.equal(d[k], r[k]), k
                assert d['class'].dtype == torch.int64
        dist.barrier()
    finally:
        dist.destroy_process_group()
