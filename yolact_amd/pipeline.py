"""Throughput mode: consecutive batches overlapped on the device.

The reference's throughput path keeps several frames in flight (eval.py evalvideo, eval.py:592-790: a thread pool runs
`prep_frame` / `eval_network` / `print` of different frames concurrently; eval.py:630-634 splits a batch over devices).  On one
MI355X the engine's analogue is `BatchPipeline`: `depth` plan instances of one model (Yolact.forward_device(slot=k): own activation
arena, head buffers, Winograd and Detect workspaces; the filters are packed per plan) rotate over `depth` HIP streams, so that the
launches of batch i + 1 fill the CUs that batch i's under-filled launches (the 35 x 35 / 18 x 18 stages, Detect, every launch boundary)
leave idle.  Measured (profiles/r06_step_overlap.txt): +15 - 18 % images/s at batch 8, 2.1x at batch 1, records bit-identical to the
single-plan path (tests/test_gpu_round6.py).

    pipe = BatchPipeline(net)                      # net: yolact_amd.yolact.Yolact on the GPU; four batches in flight, one stream each
    for x in batches:
        out = pipe.submit(x)                       # enqueue only: device tensors (fixed capacity) + out['done'] (a torch.cuda.Event)
        ...                                        # consume `out` on pipe.stream_of(out), or after out['done'].synchronize()
    pipe.synchronize()
"""
from __future__ import annotations

import torch


class BatchPipeline:
    """depth / fork: how many batches are in flight and whether each plan also forks its own side stream (engine.Plan: the P4..P7 heads
    and Detect next to the P3 branch).  A process has FOUR usable hardware queues for this (the fifth concurrently active stream was
    measured on the main stream's queue: 0.4 - 0.6x, profiles/r06_step_overlap.txt), so the two shapes that fit are
      depth 2, fork=True    two plans x (main + side stream)                      batch 8: 2 424 images/s   batch 1: 1 055
      depth 4, fork=False   four plans, one stream each (the default)             batch 8: 2 435 - 2 465    batch 1: 1 360   batch 2: 1 846 vs 1 599
    (same box, profiles/r06_pipeline_depth.txt).  fork=False runs the plans' own op lists un-forked (Plan.overlap = False for the
    duration of the submit) on [the caller's stream, the engine's two pooled side streams — idle while no plan forks —, one new stream]."""

    def __init__(self, net, depth: int = 4, device=None, fork=None):
        if depth < 1:
            raise ValueError('depth must be >= 1')
        fork = (depth <= 2) if fork is None else bool(fork)
        if (fork and depth > 2) or depth > 4:
            raise ValueError('BatchPipeline: depth %d with fork=%s needs more than four concurrently active HIP streams; measured slower '
                             'than depth 1 on MI355X (profiles/r06_step_overlap.txt)' % (depth, fork))
        self.net, self.depth, self.fork = net, int(depth), fork
        dev = torch.device(device) if device is not None else next(net.parameters()).device
        if dev.type != 'cuda':
            raise RuntimeError('BatchPipeline: the model must live on the GPU (no CPU fallback)')
        self.device = dev
        # slot 0 runs on the caller's current stream at construction time, the others on streams of their own
        if fork:
            extra = [torch.cuda.Stream(device=dev) for _ in range(self.depth - 1)]
        else:
            from .engine import side_stream_pool
            extra = list(side_stream_pool(dev))[:self.depth - 1]
            extra += [torch.cuda.Stream(device=dev) for _ in range(self.depth - 1 - len(extra))]
        self.streams = [torch.cuda.current_stream(dev)] + extra
        self._n = 0
        self.current_slot = 0

    def warm(self, x):
        """Build (pack, look up the tile table for) every slot's plan for x's shape — set-up, not a step."""
        for k in range(self.depth):
            self.net.plan_for(x, k)

    def run_in_slot(self, x, fn):
        """fn(slot) with the next slot's stream current and that slot's plan in this pipeline's fork mode; returns (slot, fn's result)."""
        slot = self._n % self.depth
        self._n += 1
        self.current_slot = slot
        plan = self.net.plan_for(x, slot)
        prev = plan.overlap
        plan.overlap = self.fork
        try:
            with torch.cuda.stream(self.streams[slot]):
                return slot, fn(slot)
        finally:
            plan.overlap = prev

    def submit(self, x, after_detect=None):
        """Enqueue forward + Detect of batch x on the next slot's stream; returns Yolact.forward_device's dict + 'slot', 'done'.
        `after_detect(out)` is called like forward_device's (behind Detect, on the stream Detect runs on); `self.current_slot` tells it
        which slot is being issued (per-slot receive buffers of a gather, for instance)."""
        def go(slot):
            out = self.net.forward_device(x, after_detect=after_detect, slot=slot)
            ev = torch.cuda.Event()
            ev.record(self.streams[slot])
            out['done'] = ev
            return out
        slot, out = self.run_in_slot(x, go)
        out['slot'] = slot
        return out

    def stream_of(self, out):
        return self.streams[out['slot']]

    def synchronize(self):
        for st in self.streams:
            st.synchronize()
