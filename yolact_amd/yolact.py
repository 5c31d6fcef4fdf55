"""`Yolact` — the reference's model class (yolact.py:379-676) re-hosted on the MI355X HIP engine.

API kept: `Yolact()` reads the active config, `.load_weights/.save_weights`, `.forward(x[B,3,H,W])` in eval mode
returns `[{'detection': {...}|None, 'net': self}] * B` exactly like the reference's `self.detect(pred_outs, self)`
(yolact.py:676), `.detect.use_fast_nms / .use_cross_class_nms`, `.maskiou_net`, `.prediction_layers`, `.backbone`.
State-dict keys/shapes equal the reference's, so its checkpoints load unchanged.

Compute: none here.  forward() builds (once per input shape) an execution plan of C-ABI calls into
libyolact_amd.so and replays it on the current HIP stream.  CPU tensors are rejected: there is no fallback.
"""
from __future__ import annotations

import ctypes as C_
import os
import sys
import threading

import torch
import torch.nn as nn

from . import _lib as L
from . import modules as M
from .config import active_cfg, backbone_kind, is_lincomb
from .engine import Plan
from .layers.detection import Detect


class Yolact(nn.Module):
    def __init__(self):
        super().__init__()
        cfg = active_cfg()
        if not is_lincomb(cfg):
            raise NotImplementedError('only mask_type.lincomb configs are on the hot path (SURVEY §8)')
        for flag in ('use_prediction_module', 'use_yolo_regressors', 'use_mask_scoring', 'use_instance_coeff',
                     'use_focal_loss', 'use_objectness_score', 'mask_proto_use_grid', 'mask_proto_bias',
                     'mask_proto_prototypes_as_features', 'mask_proto_split_prototypes_by_head',
                     'mask_proto_coeff_gate'):
            if cfg.get(flag, False) if isinstance(cfg, dict) else getattr(cfg, flag, False):
                raise NotImplementedError('cfg.%s is not used by any shipped YOLACT config and is out of scope' % flag)
        bb = cfg.backbone
        kind = backbone_kind(bb)
        args = list(bb.args)
        if kind == 'resnet':
            self.backbone = M.ResNetBackbone(*args)
        else:
            self.backbone = M.DarkNetBackbone(*args)
        if cfg.freeze_bn:
            pass  # eval-only engine: BatchNorm always uses running statistics

        # protonet (yolact.py:408-425); writes cfg.mask_dim back like the reference
        self.proto_src = cfg.mask_proto_src
        if cfg.fpn is None or self.proto_src is None:
            raise NotImplementedError('configs without an FPN / with image-sourced prototypes are out of scope')
        self.proto_net, mask_dim = M.make_net(cfg.fpn.num_features, cfg.mask_proto_net, include_last_relu=False)
        cfg.mask_dim = mask_dim
        self.mask_dim = mask_dim

        self.backbone_selected = list(bb.selected_layers)
        src_channels = self.backbone.channels
        if cfg.use_maskiou:
            self.maskiou_net = M.FastMaskIoUNet(cfg.maskiou_net, cfg.num_classes)
        self.fpn = M.FPN([src_channels[i] for i in self.backbone_selected], cfg.fpn.num_features,
                         cfg.fpn.num_downsample, cfg.fpn.pad)
        if not (cfg.fpn.use_conv_downsample and cfg.fpn.relu_pred_layers and not cfg.fpn.relu_downsample_layers
                and cfg.fpn.interpolation_mode == 'bilinear'):
            raise NotImplementedError('FPN variant outside the shipped configs')
        self.selected_layers = list(range(len(self.backbone_selected) + cfg.fpn.num_downsample))
        nf = cfg.fpn.num_features

        self.prediction_layers = nn.ModuleList()
        cfg.num_heads = len(self.selected_layers)
        if not cfg.share_prediction_module:
            raise NotImplementedError('unshared prediction modules are not used by any shipped config')
        for idx in self.selected_layers:
            ars, scales = bb.pred_aspect_ratios[idx], bb.pred_scales[idx]
            num_priors = sum(len(a) * len(scales) for a in ars)
            parent = self.prediction_layers[0] if idx > 0 else None
            self.prediction_layers.append(M.PredictionModule(
                nf, num_priors, cfg.num_classes, mask_dim, cfg.extra_head_net, dict(cfg.head_layer_params),
                ars, scales, parent=parent, index=idx))
        if cfg.use_semantic_segmentation_loss:
            self.semantic_seg_conv = nn.Conv2d(nf, cfg.num_classes - 1, kernel_size=1)  # train-only, in checkpoints

        self.detect = Detect(cfg.num_classes, bkg_label=0, top_k=cfg.nms_top_k, conf_thresh=cfg.nms_conf_thresh,
                             nms_thresh=cfg.nms_thresh)
        self._plans = {}
        self._plan_lock = threading.Lock()
        # one launch lock PER DEVICE: a plan's arena / head buffers are shared by the forwards of one device; nn.DataParallel
        # replicas (eval.py:630-634,661) are shallow copies that share these dictionaries and run on different devices from
        # different threads — they must not serialise each other's launch loops
        self._run_locks = {}
        self._ptensors = None
        # any route that rewrites parameters wholesale (nn.Module.load_state_dict included) drops the packed / BN-folded
        # copies the plans hold; in-place edits are caught by the version stamps checked in plan_for()
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.invalidate_plans())
        self.eval()

    @property
    def cfg(self):
        """The config the forward pass reads — at CALL time, like the reference's global `cfg` (yolact.py:566-674): the
        reference's own `data.config.cfg` when that module is loaded (eval.py), else yolact_amd.config.cfg."""
        return active_cfg()

    # ---- weights ---------------------------------------------------------------------------------------
    def save_weights(self, path):
        torch.save(self.state_dict(), path)

    def load_weights(self, path):
        """yolact.py:477-490: drop legacy `backbone.layer*` keys and surplus fpn.downsample_layers."""
        sd = torch.load(path, map_location='cpu')
        self.load_state_dict_compat(sd)

    def load_state_dict_compat(self, sd):
        sd = dict(sd)
        for key in list(sd.keys()):
            if key.startswith('backbone.layer') and not key.startswith('backbone.layers'):
                del sd[key]
            elif key.startswith('fpn.downsample_layers.'):
                if int(key.split('.')[2]) >= self.cfg.fpn.num_downsample:
                    del sd[key]
        self.load_state_dict(sd)
        self.invalidate_plans()

    def _run_lock_for(self, device):
        lk = self._run_locks.get(device)
        if lk is None:
            with self._plan_lock:
                lk = self._run_locks.setdefault(device, threading.Lock())
        return lk

    def invalidate_plans(self):
        with self._plan_lock:
            self._plans.clear()
            self._ptensors = None

    def _param_stamp(self):
        """Sum of the autograd version counters of every parameter / buffer: changes on any in-place edit."""
        pt = self._ptensors
        if pt is None:
            pt = self._ptensors = [t for t in list(self.parameters()) + list(self.buffers())]
            if not any(True for _ in self.parameters()):      # a DataParallel replica: its weights are plain attributes
                pt += [w for m in self.modules() for w in (getattr(m, 'weight', None), getattr(m, 'bias', None)) if torch.is_tensor(w)]
        return sum(t._version for t in pt)

    def _weights_device(self):
        """Where the weights live.  An nn.DataParallel replica (torch.nn.parallel.replicate, eval.py:630-634,661) keeps its broadcast
        parameter copies as PLAIN tensor attributes — replica.parameters() is empty — so fall back to the first conv's weight."""
        for p in self.parameters():
            return p.device
        for m in self.modules():
            w = getattr(m, 'weight', None)
            if torch.is_tensor(w):
                return w.device
        raise RuntimeError('Yolact: no weights found')

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)   # .cuda()/.to(): packed filters must be rebuilt on the new device
        if hasattr(self, '_plans'):
            self._plans.clear()
            self._ptensors = None
        return r

    def init_weights(self, backbone_path):
        raise NotImplementedError('training initialisation is out of scope (inference hot path only)')

    def train(self, mode=True):
        if mode:
            raise NotImplementedError('yolact_amd.Yolact is inference-only; training returns raw preds in the reference '
                                      '(yolact.py:639-647) and is out of scope for this tier')
        return super().train(False)

    # ---- forward ---------------------------------------------------------------------------------------
    def plan_for(self, x, slot=0) -> Plan:
        key = (tuple(x.shape), x.device, slot)
        p = self._plans.get(key)
        stamp = self._param_stamp()
        if p is not None and p.param_stamp != stamp:
            # a parameter was edited in place after the plan packed it (BN fold, Winograd filters): rebuild everything
            self.invalidate_plans()
            p, stamp = None, self._param_stamp()
        if p is None:
            with self._plan_lock:   # one-time setup is not thread-safe in the reference either (eval.py:793-796)
                p = self._plans.get(key)
                if p is None:
                    B, _, H, W = x.shape
                    with torch.no_grad():
                        p = Plan(self, B, H, W, x.device)
                        p.param_stamp = stamp
                        if x.is_cuda:
                            p.tune(x)
                    self._plans[key] = p
        return p

    def forward(self, x):
        """x: float32 [B,3,H,W], normalised RGB (resnet_transform, data/config.py:181-186)."""
        self.detect._require_fast_nms()      # traditional NMS (the reference's Detect default) is not on the hot path: Fast NMS runs
                                             # instead with one UserWarning per Detect object (YOLACT_AMD_STRICT_NMS=1: raise)
        L.require_cuda(x, 'input batch')
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError('expected [B,3,H,W], got %s' % (tuple(x.shape),))
        if self._weights_device() != x.device:
            raise RuntimeError('model and input live on different devices')
        cfg = self.cfg                    # read at call time, like the reference (yolact.py:566-568)
        cfg._tmp_img_h, cfg._tmp_img_w = int(x.shape[2]), int(x.shape[3])
        x = x.detach().to(torch.float32).contiguous()
        with torch.cuda.device(x.device):
            plan = self.plan_for(x)
            nomask = not bool(getattr(cfg, 'eval_mask_branch', True))
            # the op list carries the reference's timer sections (backbone / fpn / proto / pred_heads, yolact.py:570-607)
            with self._run_lock_for(x.device):
                proto, dev_out = plan.run(x, detect=lambda s: self.detect.run_device(
                    plan.loc, plan.conf, self._coef_for(plan, nomask), plan.priors, True, stream=s, conf_ld=plan.conf_ld),
                    timer=sys.modules.get('utils.timer'), skip_proto=nomask)
            return self.detect.finish(dev_out, proto, self)

    def _coef_for(self, plan, nomask):
        """The coefficient tensor Detect gathers from: the head GEMM's output, or — cfg.eval_mask_branch False at call time, i.e.
        eval.py --detect (eval.py:1067-1068) — the all-zero tensor the reference substitutes (yolact.py:172-175)."""
        if not nomask:
            return plan.coef
        if getattr(plan, 'zero_coef', None) is None:
            plan.zero_coef = torch.zeros_like(plan.coef)
            # the fill runs on torch's current stream, Detect on the plan's side stream with no event between them (round-5
            # advisor): a one-time host wait at the first --detect forward of a plan orders them for good
            torch.cuda.current_stream(plan.coef.device).synchronize()
        return plan.zero_coef

    def maskiou_forward(self, masks_lo):
        """FastMaskIoUNet.forward (yolact.py:363-375) on cropped prototype-resolution masks [N,ph,pw] -> [N,80]:
        5 x (3x3 stride-2 unpadded conv + ReLU), 1x1 -> 80 + ReLU, global max-pool. Called by postprocess()
        for YOLACT++ configs (output_utils.py:79-88)."""
        L.require_cuda(masks_lo, 'masks')
        dev = masks_lo.device
        lib = L.lib()
        key = ('maskiou', dev)
        packed = self._plans.get(key)
        if packed is None:
            from .engine import Packed
            packed = []
            for m in self.maskiou_net.maskiou_net:
                if isinstance(m, nn.Conv2d):
                    Cout, Cin, kh, kw = m.weight.shape
                    co4 = (Cout + 3) // 4 * 4
                    w = torch.zeros(kh * kw * Cin, co4, device=dev)
                    w[:, :Cout] = m.weight.detach().float().permute(2, 3, 1, 0).reshape(kh * kw * Cin, Cout)
                    # the wide layers (Cin % 32 == 0: 32 -> 64, 64 -> 128, the 1x1 128 -> 80) also get the conv engine's packing:
                    # at batch scale (postprocess_batch: B x cap masks at once) they run as exact-fp32 MFMA GEMMs
                    # (ymi_conv2d_nhwc_f32, heuristic tile) instead of the thread-per-pixel kernel — round 5: the six direct
                    # launches averaged 150 us each for 800 masks (profiles/r05_kernel_stats_plus_with_postprocess.txt)
                    pk = Packed(m.weight, m.bias, None, m.stride[0], m.padding[0], None, dev) if (Cin % 32 == 0 and Cout % 4 == 0) else None
                    packed.append((w.contiguous(), m.bias.detach().float().contiguous(), Cin, Cout, kh, kw,
                                   m.stride[0], m.padding[0], pk))
            self._plans[key] = packed
        N, H, W = masks_lo.shape
        x = masks_lo.contiguous()          # [N,H,W,1] NHWC with one channel
        with torch.cuda.device(dev):
            s = L.stream_ptr()
            for (w, b, Cin, Cout, kh, kw, stride, pad, pk) in packed:
                Ho, Wo = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
                y = torch.empty(N, Ho, Wo, Cout, device=dev)
                if pk is not None:       # (ONE fixed unsplit tile whatever N is: the K summation order — hence every bit — is the same
                                         #  for one image's masks and for a whole batch's: postprocess_batch == postprocess row by row)
                    d = L.ConvDesc()
                    d.x, d.w, d.bias = x.data_ptr(), pk.w.data_ptr(), pk.bias.data_ptr()
                    d.B, d.H, d.W, d.Cin, d.ldx, d.Ho, d.Wo, d.Cout = N, H, W, Cin, Cin, Ho, Wo, Cout
                    d.kh, d.kw, d.stride, d.pad, d.Kpad, d.nseg, d.tile = kh, kw, stride, pad, pk.Kpad, 1, L.TILE_64x64
                    d.seg[0] = L.ConvSeg(0, Cout, L.ACT_RELU, Cout, Ho * Wo * Cout, y.data_ptr())
                    L.check(lib.ymi_conv2d_nhwc_f32(C_.byref(d), s), 'maskiou conv (engine)')
                else:
                    L.check(lib.ymi_conv2d_direct_nhwc_f32(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), N, H, W,
                                                           Cin, Ho, Wo, Cout, kh, kw, stride, pad, 1, s), 'maskiou conv')
                x, H, W = y, Ho, Wo
            out = torch.empty(N, x.shape[3], device=dev)
            L.check(lib.ymi_global_maxpool_nhwc_f32(x.data_ptr(), out.data_ptr(), N, H * W, x.shape[3], s), 'maskiou max')
        return out

    def _forward_device_one(self, x, slot=0, after_detect=None):
        plan = self.plan_for(x, slot)
        nomask = not bool(getattr(self.cfg, 'eval_mask_branch', True))
        if os.environ.get('YOLACT_AMD_GRAPH', '0') == '1':
            out = self._forward_device_graph(plan, x, slot, nomask)
            if after_detect is not None:       # (a replayed graph has no point "after Detect": the hook runs behind the whole replay)
                out['after_detect'] = after_detect(out)
            return out

        def detect_cb(s):
            out = self.detect.run_device(plan.loc, plan.conf, self._coef_for(plan, nomask), plan.priors, True, stream=s, slot=slot,
                                         conf_ld=plan.conf_ld)
            if after_detect is not None:
                # the caller's consumer of the detection records (the record gather of the data-parallel path, a host read of the
                # counts) is enqueued HERE, on the stream Detect just ran on — in two-stream mode the side stream, where it overlaps the
                # protonet's tail on the main stream instead of queueing behind it (the final join of the plan covers it)
                ptr = s.value if hasattr(s, 'value') else s
                side = plan.stream_b if (plan.stream_b is not None and ptr == plan.stream_b.cuda_stream) else None
                if side is not None:
                    with torch.cuda.stream(side):
                        out['after_detect'] = after_detect(out)
                else:
                    out['after_detect'] = after_detect(out)
            return out
        with self._run_lock_for(x.device):    # a plan's arena / head buffers are shared state: one forward at a time per model
            proto, out = plan.run(x, detect=detect_cb, skip_proto=nomask)
        out['proto'] = proto
        out['net'] = self                    # postprocess_batch's FastMaskIoUNet (YOLACT++) lives on the model, like dets['net']
        return out

    def _forward_device_graph(self, plan, x, slot, nomask=False):
        """YOLACT_AMD_GRAPH=1: the whole op list of the plan (both streams, ~190 launches incl. Detect) is captured once
        into a hipGraph and replayed per batch — one submission instead of ~190 launches, which is what bounds small
        batches (batch 1: the GPU idles between 10-20 us kernels while Python issues the next one).  The graph owns a
        static input and static outputs; every call copies x in and clones the results out, so returned tensors keep
        the eager path's lifetime rules."""
        key = ('graph', tuple(x.shape), x.device, slot, bool(nomask))   # cfg.eval_mask_branch changes the captured launches
        # capture and replay touch the plan's shared arena / head buffers exactly like an eager run: same host lock, and
        # the device-side ordering against the previous run (possibly on another stream) through the plan's done-event
        with self._run_lock_for(x.device):
            rec = self._plans.get(key)
            cur = torch.cuda.current_stream(x.device)
            if plan._done_event is not None:
                cur.wait_event(plan._done_event)
            if rec is None:
                def body(inp):
                    proto, out = plan.run(inp, detect=lambda s: self.detect.run_device(
                        plan.loc, plan.conf, self._coef_for(plan, nomask), plan.priors, True, stream=s, slot=slot, conf_ld=plan.conf_ld),
                        skip_proto=nomask)
                    out['proto'] = proto
                    out['net'] = self
                    return out
                static_x = x.clone()
                body(static_x)                                   # eager warm-up: workspaces, lazy module state
                torch.cuda.synchronize(x.device)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    static_out = body(static_x)
                rec = self._plans[key] = (graph, static_x, static_out)
            graph, static_x, static_out = rec
            static_x.copy_(x)
            graph.replay()
            res = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in static_out.items()}
            plan.mark_done()                                     # the clones read the graph's static outputs
            return res

    def forward_device(self, x, after_detect=None, slot=0):
        """Forward + Detect with NO host synchronisation: fixed-capacity device tensors
        (count [B], box [B,cap,4], score, cls, coef, prior) + 'proto'. Used by the data-parallel path
        (yolact_amd.parallel) and by throughput runs that keep results on the GPU.

        after_detect (optional): `after_detect(out)` is called with Detect's outputs (everything but 'proto') as soon as Detect's
        kernels are enqueued, with the stream they run on as torch's current stream; what it returns is out['after_detect'].  The
        data-parallel path hands its record gather in here: the records do not depend on the prototypes, and the protonet is the
        tail of the step.

        slot (round 6): which of the model's plan instances for this input shape runs the batch.  Every slot owns its activation arena,
        head buffers, Winograd workspaces and Detect workspaces, so two batches issued on DIFFERENT HIP streams with different slots may
        overlap on the device — a throughput server alternating slot 0 / 1 on two streams fills the CUs a batch's under-filled launches
        (the 35 x 35 / 18 x 18 stages, Detect, the launch boundaries) leave idle with the next batch's work (bench.py --step-overlap)."""
        L.require_cuda(x, 'input batch')
        x = x.detach().to(torch.float32).contiguous()
        with torch.cuda.device(x.device):
            return self._forward_device_one(x, slot=int(slot), after_detect=after_detect)

    def forward_sharded(self, x_global, dst=0, masks=None, mask_size=None, n_global=None):
        """Data-parallel inference of one global batch across the ranks of torch.distributed (one process per GPU, RCCL): this
        rank computes its contiguous share of the images (parallel.shard_range), then ONE gather of the fixed-size detection
        records brings every image's detections to `dst` (eval.py:630-634,661 is batch splitting with a no-op gather; there is
        no collective inside the network).  Returns, on dst, the list the reference's Detect would return for the global
        batch — {'detection': {...}|None, 'net': self} per image; `proto` is the prototype tensor for the images this rank
        computed itself and None for detections gathered from other ranks (assemble those masks on their own rank: every
        rank can run postprocess_batch on its forward_device output) — and None on the other ranks.  The returned tensors
        are fresh copies: they stay valid across later calls.

        n_global (round 6): when given, `x_global` is THIS RANK'S SHARD only — images parallel.shard_range(n_global, rank, world) of a
        global batch of n_global images (an empty [0, 3, H, W] tensor on a rank that owns none) — instead of the whole batch.

        masks='bits' (round 5; eval.py:630-634,791 moves the frame to the device that holds a detection's prototypes — here the
        MASKS move instead): every rank assembles the final binary masks of its own images at `mask_size` = (h, w) (default: the
        input size) where their prototypes live (output_utils.postprocess_bits_batch: one lincomb and one upsample-into-bits
        launch per shard) and a second fixed-capacity gather brings them to `dst` as bits (3.8 MB per image at 550 x 550).  Every
        detection of the returned list then also carries `mask_bits` int64 [n, ceil(h*w/64)], the per-image record `mask_size`, and
        `postprocess_bits(out, w, h, batch_idx=b)` / `layers.box_utils.mask_iou_bits` work for EVERY image of the global batch
        (parallel.unpack_mask_bits gives the float form)."""
        from . import parallel
        L.require_cuda(x_global, 'input batch')
        if masks not in (None, 'bits'):
            raise ValueError("forward_sharded: masks must be None or 'bits', got %r" % (masks,))
        if not hasattr(self, '_gatherer') or self._gatherer.dst != dst:
            self._gatherer = parallel.RecordGatherer(dst)
            self._mask_gatherer = parallel.RecordGatherer(dst)
        bits = None
        if masks == 'bits':
            from .layers.output_utils import postprocess_bits_batch
            hh, ww = (int(mask_size[0]), int(mask_size[1])) if mask_size is not None else (int(x_global.shape[2]), int(x_global.shape[3]))
            cap = parallel._cap_of(self.forward_device)

            def masks_fn(out):
                if out is None:
                    return torch.zeros(0, cap, (hh * ww + 63) // 64, dtype=torch.int64, device=x_global.device)
                return postprocess_bits_batch(out, ww, hh)['bits']
            rec, mine, bits = parallel.sharded_forward(self.forward_device, x_global, self.mask_dim, self._gatherer, dst,
                                                       masks_fn=masks_fn, mask_gatherer=self._mask_gatherer, n_global=n_global)
        else:
            rec, mine = parallel.sharded_forward(self.forward_device, x_global, self.mask_dim, self._gatherer, dst, n_global=n_global)
        if rec is None:
            return None
        import torch.distributed as dist
        world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        rank = dist.get_rank() if world > 1 else 0
        lo, hi = parallel.shard_range(int(x_global.shape[0]) if n_global is None else int(n_global), rank, world)
        return parallel.assemble_sharded(rec, mine, lo, hi, self.mask_dim, self, bits=bits,
                                         mask_size=(hh, ww) if masks == 'bits' else None)

    def forward_raw(self, x):
        """Head outputs before Detect (for parity tests): loc, conf (logits), mask, priors, proto — clones."""
        L.require_cuda(x, 'input batch')
        x = x.detach().to(torch.float32).contiguous()
        with torch.cuda.device(x.device):
            plan = self.plan_for(x)
            with self._run_lock_for(x.device):
                proto, _ = plan.run(x)
                out = {'loc': plan.loc.clone(), 'conf_logits': plan.conf[..., :plan.Ccls].contiguous().clone(),
                       'mask': plan.coef.clone(),
                       'priors': plan.priors, 'proto': proto}
                plan.mark_done()      # the clones read the plan's persistent head buffers
            return out
