"""Throughput mode: consecutive batches overlapped on the device.

The reference's throughput path keeps several frames in flight (eval.py evalvideo, eval.py:592-790: a thread pool runs
`prep_frame` / `eval_network` / `print` of different frames concurrently; eval.py:630-634 splits a batch over devices).  On one
MI355X the engine's analogue is `BatchPipeline`: `depth` plan instances of one model (Yolact.forward_device(slot=k): own activation
arena, head buffers, Winograd and Detect workspaces; the filters are packed per plan) rotate over `depth` HIP streams, so that the
launches of batch i + 1 fill the CUs that batch i's under-filled launches (the 35 x 35 / 18 x 18 stages, Detect, every launch boundary)
leave idle.  Measured (profiles/r06_step_overlap.txt): +15 - 18 % images/s at batch 8, 2.1x at batch 1, records bit-identical to the
single-plan path (tests/test_gpu_round6.py).  depth 2 is the default: with the two streams a plan forks itself that makes four HIP
streams — the fifth stream of a process was measured back on the main stream's hardware queue (engine._side_stream).

    pipe = BatchPipeline(net)                      # net: yolact_amd.yolact.Yolact on the GPU
    for x in batches:
        out = pipe.submit(x)                       # enqueue only: device tensors (fixed capacity) + out['done'] (a torch.cuda.Event)
        ...                                        # consume `out` on pipe.stream_of(out), or after out['done'].synchronize()
    pipe.synchronize()
"""
from __future__ import annotations

import torch


class BatchPipeline:
    def __init__(self, net, depth: int = 2, device=None):
        if depth < 1:
            raise ValueError('depth must be >= 1')
        self.net, self.depth = net, int(depth)
        dev = torch.device(device) if device is not None else next(net.parameters()).device
        if dev.type != 'cuda':
            raise RuntimeError('BatchPipeline: the model must live on the GPU (no CPU fallback)')
        self.device = dev
        # slot 0 runs on the caller's current stream at construction time, the others on streams of their own
        self.streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(device=dev) for _ in range(self.depth - 1)]
        self._n = 0
        self.current_slot = 0

    def warm(self, x):
        """Build (pack, look up the tile table for) every slot's plan for x's shape — set-up, not a step."""
        for k in range(self.depth):
            self.net.plan_for(x, k)

    def submit(self, x, after_detect=None):
        """Enqueue forward + Detect of batch x on the next slot's stream; returns Yolact.forward_device's dict + 'slot', 'done'.
        `after_detect(out)` is called like forward_device's (behind Detect, on the stream Detect runs on); `self.current_slot` tells it
        which slot is being issued (per-slot receive buffers of a gather, for instance)."""
        slot = self._n % self.depth
        self._n += 1
        self.current_slot = slot
        st = self.streams[slot]
        with torch.cuda.stream(st):
            out = self.net.forward_device(x, after_detect=after_detect, slot=slot)
            ev = torch.cuda.Event()
            ev.record(st)
        out['slot'], out['done'] = slot, ev
        return out

    def stream_of(self, out):
        return self.streams[out['slot']]

    def synchronize(self):
        for st in self.streams:
            st.synchronize()
