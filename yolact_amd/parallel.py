"""Data-parallel inference across the GPUs of one node: one process per GPU, batch sharded on dim 0, and ONE
gather of fixed-size detection records per batch over RCCL/xGMI.

The reference's only multi-GPU inference path is nn.DataParallel batch splitting with a no-op gather
(eval.py:630-634,661).  In eval mode images are independent (BN running stats, per-image Detect), so there is
no collective inside the network; weights are replicated by each rank loading the same checkpoint.
Record per image: count + cap x (box 4, score 1, class 1, coef D) fp32  (15.2 KB at cap=100, D=32) — latency-bound,
so a single direct gather (every sender on its own xGMI link to the root) is the right collective, not a ring.
The count and the class ids travel AS fp32 inside the record (one dtype, one buffer, written by the Detect kernel itself):
exact for every integer below 2^24, and a count is <= 200, a class id <= 80.

Masks (round 5).  The prototypes stay on the rank that computed them; when the caller on the gather root needs the masks of the
WHOLE batch (eval.py's prep_metrics / COCO dump over a global batch), every owner assembles its images' masks where the
prototypes live and ships them as BITS (output_utils.postprocess_bits_batch: int64 [cap, ceil(h*w/64)] per image, 3.8 MB at
550 x 550) through a second fixed-capacity gather — `sharded_forward(..., masks_fn=...)`, `Yolact.forward_sharded(masks='bits')`.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous block partition of a global batch (same rule on every rank)."""
    per = (n_items + world - 1) // world
    lo = min(rank * per, n_items)
    return lo, min(lo + per, n_items)


def pack_records(dev_out: Dict[str, torch.Tensor]) -> torch.Tensor:
    """Fixed-capacity Detect outputs -> one [B, 1 + cap*(6+D)] fp32 tensor (count first). Device-side, no sync.
    The Detect selection kernel writes this record itself (ymi_detect_desc.out_rec -> dev_out['rec']): no torch op at all in
    the step; the cat / cast form below only serves outputs that come without it (tests with hand-made tensors)."""
    if 'rec' in dev_out:
        return dev_out['rec']
    B, cap, D = dev_out['coef'].shape
    body = torch.cat([dev_out['box'], dev_out['score'].unsqueeze(-1), dev_out['cls'].to(torch.float32).unsqueeze(-1),
                      dev_out['coef']], dim=-1).reshape(B, cap * (6 + D))
    return torch.cat([dev_out['count'].to(torch.float32).unsqueeze(-1), body], dim=1).contiguous()


def unpack_records(rec: torch.Tensor, D: int) -> List[Optional[Dict[str, torch.Tensor]]]:
    """Inverse of pack_records on the gathered tensor; slices to the per-image counts (one host read)."""
    B = rec.shape[0]
    cap = (rec.shape[1] - 1) // (6 + D)
    counts = rec[:, 0].to(torch.int64).tolist()
    body = rec[:, 1:].reshape(B, cap, 6 + D)
    out = []
    for b, n in enumerate(counts):
        if n == 0:
            out.append(None)
            continue
        r = body[b, :n]
        out.append({'box': r[:, :4], 'score': r[:, 4], 'class': r[:, 5].to(torch.int64), 'mask': r[:, 6:]})
    return out


class RecordGatherer:
    """The gather with PERSISTENT buffers: one [world * rows, L] tensor per (rows, L, device) allocated once; the per-rank
    receive buffers are views of it, so a step allocates nothing and needs no torch.cat afterwards (the collective itself
    assembles the global batch).  `rows` = records every rank contributes (ceil(global batch / world)); shorter shards are
    padded into a persistent staging buffer."""

    def __init__(self, dst: int = 0):
        self.dst = dst
        self._out = {}
        self._pad = {}

    def __call__(self, rec: torch.Tensor, rows: int, n_items: Optional[int] = None, force_collective: bool = False):
        import os
        if not (dist.is_available() and dist.is_initialized()):
            return rec
        world = dist.get_world_size()
        if world == 1 and not (force_collective or os.environ.get('YOLACT_AMD_FORCE_GATHER', '0') == '1'):
            return rec
        if rec.shape[0] > rows:
            raise ValueError('shard has %d records, expected at most %d' % (rec.shape[0], rows))
        key = (rows, rec.shape[1], rec.device, rec.dtype)
        if rec.shape[0] < rows:                               # uneven last shard: count-0 records up to `rows`
            pad = self._pad.get(key)
            if pad is None:
                pad = self._pad[key] = torch.zeros(rows, rec.shape[1], dtype=rec.dtype, device=rec.device)
            pad[:rec.shape[0]].copy_(rec)
            pad[rec.shape[0]:, 0].zero_()
            rec = pad
        rec = rec.contiguous()
        if dist.get_rank() != self.dst:
            dist.gather(rec, None, dst=self.dst)
            return None
        out = self._out.get(key)
        if out is None:
            out = self._out[key] = torch.empty(world * rows, rec.shape[1], dtype=rec.dtype, device=rec.device)
        dist.gather(rec, [out[r * rows:(r + 1) * rows] for r in range(world)], dst=self.dst)
        return out if n_items is None else out[:n_items]


def pin_rank_affinity(local_rank: int, local_world: int):
    """Give every rank of a node its own contiguous slice of the host's CPUs (one process per GPU: eight Python launch loops
    sharing all cores migrate and contend; the launch path is latency-sensitive).  Returns the CPU set, or None when the
    platform has no sched_setaffinity."""
    import os
    if not hasattr(os, 'sched_setaffinity') or local_world < 1:
        return None
    cpus = sorted(os.sched_getaffinity(0))
    per = max(1, len(cpus) // local_world)
    mine = cpus[local_rank * per:(local_rank + 1) * per] or cpus
    os.sched_setaffinity(0, mine)
    return mine


def sharded_forward(forward_device, x_global: torch.Tensor, D: int, gatherer: Optional[RecordGatherer] = None, dst: int = 0,
                    masks_fn=None, mask_gatherer: Optional[RecordGatherer] = None, n_global: Optional[int] = None):
    """Data-parallel forward of one GLOBAL batch.  Two calling forms:
      * n_global None (default): every rank passes the same `x_global` [B, ...] and slices its own share out of it;
      * n_global = B (round 6): every rank passes ONLY ITS SHARD — images shard_range(B, rank, world) of the global batch, possibly
        zero of them — so that no rank ever holds the other ranks' images (BASELINE configs[4]: 64 x 700 x 700 images are 376 MB;
        eval.py:630-634 scatters frames the same way).  The shard's length is checked against shard_range.
    In both forms
    rank r runs `forward_device` (Yolact.forward_device: forward + Detect, no host sync) on images shard_range(B, r, world)
    and the fixed-size detection records of all images are gathered on `dst` with the ONE collective of the path.  Returns
    (records [B, L] on dst | None elsewhere, this rank's device outputs or None for an empty shard).  The prototypes stay on
    the rank that computed them (masks are assembled where the prototypes live).

    masks_fn (optional): `masks_fn(device_outputs) -> int64 [b, cap, W64]` — the bit-packed masks of this rank's images
    (output_utils.postprocess_bits_batch(...)['bits']; an empty shard calls `masks_fn(None)` for a [0, cap, W64] tensor).  They ride
    a SECOND fixed-capacity gather (same collective, same padding rule, persistent buffers of `mask_gatherer`) and the function
    returns a third value: the gathered bits [B, cap, W64] on dst, None elsewhere."""
    world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
    rank = dist.get_rank() if world > 1 else 0
    B = int(x_global.shape[0]) if n_global is None else int(n_global)
    lo, hi = shard_range(B, rank, world)
    if n_global is None:
        x_mine = x_global[lo:hi]
    else:
        if int(x_global.shape[0]) != hi - lo:
            raise ValueError('sharded_forward(n_global=%d): rank %d of %d owns images [%d, %d) = %d, got a shard of %d'
                             % (B, rank, world, lo, hi, hi - lo, int(x_global.shape[0])))
        x_mine = x_global
    rows = (B + world - 1) // world
    g = gatherer or RecordGatherer(dst)
    out, allrec, hooked = None, None, False
    if hi > lo:
        if _accepts_after_detect(forward_device):
            # the gather is enqueued from INSIDE the forward, right behind Detect on the stream Detect runs on (Yolact.forward_device):
            # the records do not depend on the prototypes, so the collective overlaps the protonet instead of queueing behind it
            out = forward_device(x_mine.contiguous(),
                                 after_detect=lambda o: g(pack_records(o), rows, n_items=B, force_collective=world > 1))
            allrec, hooked = out.pop('after_detect'), True
        else:
            out = forward_device(x_mine.contiguous())
    if not hooked:
        if out is not None:
            rec = pack_records(out)
        else:                                         # more ranks than images: contribute count-0 records only
            L_ = 1 + int(_cap_of(forward_device)) * (6 + D)
            rec = torch.zeros(0, L_, dtype=torch.float32, device=x_global.device)
        allrec = g(rec, rows, n_items=B, force_collective=world > 1)
    if masks_fn is None:
        return allrec, out
    bits = masks_fn(out)                                       # [b, cap, W64] int64
    if bits.dtype != torch.int64 or bits.dim() != 3 or bits.shape[0] != hi - lo:
        raise ValueError('masks_fn must return int64 [%d, cap, W64] bit masks, got %s %s' % (hi - lo, bits.dtype, tuple(bits.shape)))
    cap, W64 = int(bits.shape[1]), int(bits.shape[2])
    mg = mask_gatherer or RecordGatherer(dst)
    allbits = mg(bits.reshape(hi - lo, cap * W64), rows, n_items=B, force_collective=world > 1)
    return allrec, out, (allbits.view(-1, cap, W64) if allbits is not None else None)


def assemble_sharded(rec: torch.Tensor, mine, lo: int, hi: int, D: int, net=None, bits: Optional[torch.Tensor] = None,
                     mask_size=None):
    """dst side of Yolact.forward_sharded: the gathered records [B, L] -> the list the reference's Detect returns for the global
    batch ({'detection': {...} | None, 'net': net} per image).  `rec` is first DETACHED from the gatherer's persistent receive
    buffer (clone; 15 KB per image): unpack_records slices without copying, and the next step overwrites that buffer in place
    while callers may still hold — or asynchronously postprocess — this step's results.  `proto` is this rank's own prototype
    tensor for images lo .. hi-1 and None for detections computed elsewhere (masks are assembled where the prototypes live,
    SURVEY 8(e); postprocess() refuses a None with a clear error).

    bits [B, cap, W64] (the second gather of sharded_forward) + mask_size (h, w): EVERY detection also carries
    `mask_bits` int64 [n, W64] — its final binary masks at (h, w), assembled by its owner — and the per-image record carries
    `mask_size`; output_utils.postprocess_bits(out, w, h, batch_idx=b) then works for every image of the global batch, local or
    remote, and layers.box_utils.mask_iou_bits scores them against bit-packed ground truth."""
    rec = rec.clone()
    if bits is not None:
        bits = bits.clone()                       # detached from the gatherer's persistent receive buffer, like the records
    out = []
    for b, det in enumerate(unpack_records(rec, D)):
        if det is not None:
            det['proto'] = mine['proto'][b - lo] if (mine is not None and lo <= b < hi) else None
            if bits is not None:
                det['mask_bits'] = bits[b, :det['score'].shape[0]]
        r = {'detection': det, 'net': net}
        if bits is not None:
            r['mask_size'] = tuple(mask_size)
        out.append(r)
    return out


def unpack_mask_bits(bits: torch.Tensor, h: int, w: int) -> torch.Tensor:
    """int64 [n, ceil(h*w/64)] bit masks (bit i of word j = pixel 64 j + i) -> float32 {0,1} masks [n, h, w]: the layout eval.py's
    prep_metrics / prep_display consume (torch ops, any device) — for callers that need the float form of a gathered mask."""
    n = bits.shape[0]
    shifts = torch.arange(64, device=bits.device, dtype=torch.int64)
    flat = ((bits.unsqueeze(-1) >> shifts) & 1).reshape(n, -1)[:, :h * w]
    return flat.to(torch.float32).view(n, h, w)


def _accepts_after_detect(forward_device) -> bool:
    import inspect
    try:
        return 'after_detect' in inspect.signature(forward_device).parameters
    except (TypeError, ValueError):
        return False


def _cap_of(forward_device):
    net = getattr(forward_device, '__self__', None)
    det = getattr(net, 'detect', None)
    if det is None:
        raise RuntimeError('sharded_forward: an empty shard needs the record length; pass a bound Yolact.forward_device')
    from .config import active_cfg
    return det.top_k if det.use_cross_class_nms else int(active_cfg().max_num_detections)


def gather_records(rec: torch.Tensor, dst: int = 0, rows_per_rank: Optional[int] = None, n_items: Optional[int] = None,
                   force_collective: bool = False) -> Optional[torch.Tensor]:
    """The single collective of the path: ONE dist.gather (RCCL over xGMI under the `nccl` backend) of the fixed-size
    records to rank `dst`.  Returns [n, L] on dst (n = world * rows, trimmed to `n_items` when given), None elsewhere.

    rows_per_rank: records every rank contributes (= ceil(global batch / world), the `per` of shard_range); a shorter
    (last) shard is padded with count-0 records, which the trim removes.  Default: this rank's own row count, i.e. all
    shards equal — checked, because a size mismatch inside an RCCL gather is undefined behaviour, not an error.
    Without an initialised process group the records are returned as they are.  At world size 1 the gather is skipped
    unless `force_collective` (or YOLACT_AMD_FORCE_GATHER=1): then the real collective runs with one rank, which is how
    the single-GPU box exercises the RCCL code path."""
    import os
    if not (dist.is_available() and dist.is_initialized()):
        return rec
    world = dist.get_world_size()
    force = force_collective or os.environ.get('YOLACT_AMD_FORCE_GATHER', '0') == '1'
    if world == 1 and not force:
        return rec
    rows = int(rows_per_rank) if rows_per_rank is not None else int(rec.shape[0])
    if rows_per_rank is None and world > 1:
        # cheap consistency check (one tiny all_reduce of the row count) instead of undefined behaviour on mismatch
        t = torch.tensor([rows, -rows], device=rec.device, dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if int(t[0]) != -int(t[1]):
            raise RuntimeError('gather_records: ranks hold different numbers of records (%d..%d); pass rows_per_rank'
                               % (-int(t[1]), int(t[0])))
    # one code path for the collective: the persistent-buffer gatherer (round 4 kept an allocating empty_like x world + cat here)
    g = _default_gatherers.get(dst)
    if g is None:
        g = _default_gatherers[dst] = RecordGatherer(dst)
    out = g(rec.contiguous(), rows, n_items=n_items, force_collective=True)
    return out.clone() if out is not None else None       # (callers of this function own their result; the gatherer reuses its buffer)


_default_gatherers: Dict[int, RecordGatherer] = {}
