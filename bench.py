#!/usr/bin/env python
"""Throughput of the YOLACT inference hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py ...)

Metric (BASELINE.json): images/s on synthetic 550x550 batches, ResNet50-FPN (`yolact_resnet50_config`,
BASELINE configs[1]), batch 8 PER GPU (weak scaling: global batch 8*N).  One step = one pass of the hot path
over one batch already resident in HBM: NCHW->NHWC, backbone, FPN, protonet, heads, softmax, decode, Fast NMS,
then the single RCCL gather of detection records to rank 0 and the host read of the per-image counts
(what `Yolact.forward` must do to return the reference's dynamically-sized outputs).
`--with-postprocess` adds postprocess() to 550x550 for every image (the reference's FPS definition).

Launching: `python bench.py --gpus N` starts its N ranks ITSELF (re-exec under torch.distributed.run on 127.0.0.1) when it
is not already running under a launcher; it refuses to run when the box has fewer than N GPUs or WORLD_SIZE disagrees
with --gpus.  Every run — N = 1 included — initialises the `nccl` (= RCCL) process group and sends the detection records
through the real `dist.gather`; the JSON carries `rccl_ranks` and `gathered_records` (= global batch) as the proof.
`secondary` (N = 1): the same batch with postprocess() to 550x550 in the step, and the reference's own published-FPS
definition (eval.py:264-281 prep_benchmark: batch 1, postprocess, top-k `.cpu().numpy()` copies, sync).
Other BASELINE configs: `--config yolact_im700_config --batch 8` (configs[4], per-GPU share; size 700 is implied),
`--config yolact_base_config --batch 16` (configs[2]), `--config yolact_plus_resnet50_config` (configs[3]).

One JSON line on rank 0, with `roofline`, `box_calibration` and, at N=1, `cpu_baseline` (the CPU oracle = port of the reference's
path, timed on this host's cores).

`roofline` (round 5 layout): the TOP LEVEL is SURVEY 8(d)'s whole-path figure — `achieved` = algorithmic conv FLOPs of a step (118.28
GFLOP per image) / the TIMED step on one GPU, `peak` = the matrix pipe at the precision used (fp16x2: 2500 / 3 = 833.3 TFLOP/s),
`frac` = their ratio: the number BASELINE.json's ">= 60 % of the conv roofline" target refers to.  `kernel` / `dominant_kernel` = the
instantiation with the largest summed duration on ITS matrix-pipe fraction (a direct launch priced with its algorithmic conv FLOPs, a
grouped Winograd GEMM with the FLOPs it executes: `flops_basis`), measured live with HIP events on the launch stream;
`dominant_kernel.hbm_view` = its algorithmic bytes per launch over the same duration (for a Winograd GEMM these are Winograd-domain
bytes: `layer_view` puts them next to what the convolution needs); `traffic` = PMC HBM bytes per launch of that kernel (static: the
committed summary of separate rocprofv3 --pmc passes, reported only under the tile table they were measured with).  `engine` = all
GEMM launches of a step, `all_conv` = algorithmic FLOPs over the summed launch durations, `bound_sum` = sum over launches of
max(FLOPs / tile peak, algorithmic bytes / 8 TB/s), `per_kernel` = every instantiation.
`box_calibration`: three fixed micro-workloads timed right before the warm-up steps (csrc/calib.hip: fp16 MFMA loop, 1 GiB HBM copy,
L2-resident read stream), the host's cost per C-ABI call / kernel launch, and what rocm-smi reports — so that two runs can be
attributed to the box or to the code.  `host_issue_ms_per_step`: host time to issue one step (far below ms_per_step = GPU-bound).
DESIGN.md 3.5 / 6.
Round 6: `--step-overlap N` (default 4; `config.step_overlap`): consecutive steps rotate over N plan instances of the model
(yolact_amd.pipeline.BatchPipeline over Yolact.forward_device(slot=): own arena, head buffers, workspaces) on N HIP streams — N = 2: each
plan also forks its side stream; N = 3, 4: one stream per plan (four busy streams is what a process has hardware queues for) —, so that
batch i + 1 starts while batch i is in its tail — the reference's own throughput mode pipelines frames the same way (eval.py evalvideo: a thread pool keeps several frames in
flight).  Every step is still one full pass over one batch, every step's counts still reach the host inside the timed region;
`strong_scaling` (one plan, one stream) is the serial figure in the same line.
"""
import argparse
import ctypes as C
import json
import os
import statistics
import sys
import time

# The plan overlaps two HIP streams.  ROCm maps streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues in creation
# order; once RCCL has created its own streams, the plan's side stream lands on the SAME hardware queue as the main one
# and the overlap is gone (measured: forward 6.69 ms instead of 6.25, profiles/r02_overlap_probe.txt).  Must be set before
# the HIP runtime initialises.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CU x 2.4 GHz (spec)
BF16_MFMA_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense bf16 MFMA (spec; 2470-2495 measured)
X3_PEAK_TFLOPS = BF16_MFMA_PEAK_TFLOPS / 6   # bf16x3 tiles: 6 bf16 MFMA products per fp32-class product -> 416.7 TFLOP/s of
                                             # fp32-equivalent work is what the matrix pipe can deliver in that mode
H2_PEAK_TFLOPS = BF16_MFMA_PEAK_TFLOPS / 3   # fp16x2 tiles: 3 fp16 MFMA products per fp32-class product (fp16 MFMA rate = bf16's)
HBM_PEAK_GBPS = 8000.0             # MI355X_MICROARCH.md: HBM3E spec (~6.3 TB/s achievable read, 4.7 TB/s copy measured)


def kernel_peak(name):
    """Matrix-pipe peak, in fp32(-equivalent) TFLOP/s, of a conv_igemm_f32 instantiation: the exact-fp32 MFMA tiles are
    bounded by the fp32 MFMA peak, the `...x3` tiles (fp32-class products as 6 bf16 MFMAs) by bf16 peak / 6, the `...h2`
    tiles (3 fp16 MFMAs) by fp16 peak / 3."""
    tile = name.split('<', 1)[1].split(',', 1)[0] if '<' in name else ''
    return X3_PEAK_TFLOPS if tile.endswith('x3') else H2_PEAK_TFLOPS if (tile.endswith('h2') or tile.startswith('dcnp') or tile.startswith('ws') or tile.startswith('patch')) \
        else FP32_MFMA_PEAK_TFLOPS


def conv_alg_bytes(d):
    """ALGORITHMIC HBM bytes of one convolution launch (every conv reads its input once, its filters once, a
    residual once, and writes its output once; fp32)."""
    b = d.B * d.H * d.W * (d.cin_alg or d.Cin) + d.Cout * d.kh * d.kw * (d.cin_alg or d.Cin) + d.B * d.Ho * d.Wo * d.Cout
    if d.res_mode == 1:            # YMI_RES_ADD: the shortcut tensor
        b += d.B * d.Ho * d.Wo * d.Cout
    elif d.res_mode == 2:          # YMI_RES_BILINEAR: the coarser FPN level
        b += d.B * d.res_H * d.res_W * d.Cout
    return 4.0 * b


def wino_bytes(d, m):
    """Bytes the three launches of a Winograd layer must move: input transform (x in, V out), grouped GEMM (V, U in, M out),
    output transform (M in, y out); V / M hold (m+2)^2 components per (tile, channel)."""
    g = (m + 2) * (m + 2)
    T = d.B * ((d.H + m - 1) // m) * ((d.W + m - 1) // m)
    x, y = 4.0 * d.B * d.H * d.W * d.Cin, 4.0 * d.B * d.Ho * d.Wo * d.Cout
    V, M, U = 4.0 * g * T * d.Cin, 4.0 * g * T * d.Cout, 4.0 * g * d.Cout * d.Cin
    return {'in': x + V, 'gemm': V + U + M, 'out': M + y}
CONFIG = 'yolact_resnet50_config'


def build_model(device, size, config=CONFIG):
    import yolact_amd
    from yolact_amd.utils.synth import synth_state_dict
    yolact_amd.set_cfg(config)
    from yolact_amd.yolact import Yolact
    net = Yolact()
    shapes = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
    sd = synth_state_dict(shapes, seed=0, conf_gain=0.04)      # ~70 % of priors over threshold
    net.load_state_dict_compat(sd)
    net.detect.use_fast_nms = True                              # eval.py:871 (default --fast_nms); the class default is False
    return net.to(device), sd


def median_over_passes(recs, reps):
    """recs: the [ms, flops, tile, kind] records of `reps` identical passes over the op list, pass after pass.  Replaces every record's
    duration by the median over the passes at its position — in place, and only when the passes really are the same sequence (same
    count, same kinds); returns whether it did."""
    import statistics
    n = len(recs)
    nper = n // reps if reps > 0 and n and n % reps == 0 else 0
    if not nper or not all(recs[r * nper + i][3] == recs[i][3] for r in range(reps) for i in range(nper)):
        return False
    for i in range(nper):
        med = statistics.median(recs[r * nper + i][0] for r in range(reps))
        for r in range(reps):
            recs[r * nper + i][0] = med
    return True


def roofline(net, x, reps=5):
    """Per-launch HIP-event timing of every conv launch (events recorded on the launch stream by the library)."""
    from yolact_amd import _lib as L
    lib = L.lib()
    graph = os.environ.pop('YOLACT_AMD_GRAPH', None)  # ... launched eagerly (a replayed hipGraph records no events)
    plan = net.plan_for(x)
    names = [n for n, _ in plan.conv_meta]
    # The timed region overlaps the small P4..P7 / Detect kernels with the P3 branch on a second HIP stream; kernels
    # that share the GPU have inflated individual durations, so the per-kernel pass runs the SAME op list serialised
    # on one stream (plan.overlap = False): a kernel's own rate, events recorded on the stream it is launched on.
    plan.overlap = False
    lib.ymi_prof_reset()
    lib.ymi_prof_enable(1)
    for _ in range(reps):
        net.forward_device(x)
    torch.cuda.synchronize()
    lib.ymi_prof_enable(0)
    plan.overlap = True
    if graph is not None:
        os.environ['YOLACT_AMD_GRAPH'] = graph
    n = lib.ymi_prof_count()
    ms, fl, tile, kind = C.c_float(), C.c_double(), C.c_int32(), C.c_int32()
    # record kinds: 0/1/2 = one direct conv launch (loader id), 7 = a direct 1x1 launch on the pointwise loader (kernel
    # template LOADER 3); 3 / 4 = a whole Winograd F(2x2) / F(4x4) layer (input transform + 16- / 36-group GEMM + output
    # transform, ALGORITHMIC conv FLOPs); 5 / 6 = the Winograd GEMM launch alone (the FLOPs it executes).
    # 14 = csrc/patch.hip (3x3 64 -> 64 from an LDS-resident input patch).
    # 11 = the weight-stationary streaming kernel (csrc/wstat.hip); 12 = a 1x1 layer fused into the previous layer's launch (its FLOPs, no
    # duration of its own); 13 = ymi_pointwise_chain_f32 (csrc/chain.hip: conv3 + shortcut + ReLU -> the next block's conv1).
    # 9 = the pipelined DCNv2 gather-GEMM (csrc/dcn.hip: pipe_h2_k<..., PLAIN = false>), 10 = the same kernel as an ordinary 3x3 / 1x1
    # convolution (PLAIN = true).
    # 8 = the fused ResNet stem launch (layout change + 7x7 conv + BN + ReLU + max-pool; the conv's algorithmic FLOPs).
    # Layer table / all_conv: kinds 0-4, 7 and 8.  Single-kernel roofline: kinds 0-2, 7, 8, 5 and 6.
    by_kernel, layers = {}, {}
    tot_ms = tot_fl = 0.0
    li = -1
    nl = len(names)
    descs = [d for _, d in plan.conv_meta]
    bound_ms = {'mfma': 0.0, 'hbm': 0.0}          # step-level sum of max(F / peak of the tile used, bytes / HBM peak) per launch
    wino_pending = None
    lv = {'n': 0, 'ms': 0.0, 'fl': 0.0, 'conv_bytes': 0.0, 'wd_bytes': 0.0, 'gemm_bytes': 0.0}
    # every record of every pass first: a record's duration is the MEDIAN over the passes at its position (the events bracket a
    # host-issued launch: a host thread descheduled between the start event and the launch shows up as a 26 ms "kernel" — seen once on
    # a loaded host, session r5fz — and a mean over three passes keeps a third of it)
    recs = []
    for i in range(n):
        L.check(lib.ymi_prof_read(i, C.byref(ms), C.byref(fl), C.byref(tile), C.byref(kind)))
        recs.append([ms.value, fl.value, tile.value, kind.value])
    median_over_passes(recs, reps)
    for i in range(n):
        ms.value, fl.value, tile.value, kind.value = recs[i]
        tname = L.TILE_NAMES.get(tile.value, '?')
        if kind.value == 12:      # a 1x1 layer computed INSIDE the previous layer's F(4x4) output transform (ymi_wino_desc.proj_*): its
            li += 1               # algorithmic FLOPs count for the step, its time is part of that layer's record
            la = layers.setdefault(names[li % nl], [0.0, fl.value, ''])
            la[0] += ms.value / reps
            la[2] = 'computed inside the launch of the layer above (wino43_out_proj_k / chain_h2_k)'
            tot_ms += ms.value; tot_fl += fl.value
            continue
        # algorithmic bytes of this record and its lower-bound time (sum over layers of max(F/peak, bytes/BW))
        if kind.value in (3, 4):                     # a whole Winograd layer: remember its geometry for the GEMM record
            wino_pending = wino_bytes(descs[(li + 1) % nl], 2 * kind.value - 4)
            nbytes = 0.0
            # layer view (VERDICT r3 #2e): the three launches of the layer against what the CONVOLUTION needs — algorithmic
            # FLOPs over the summed duration, conv-algorithmic bytes next to the Winograd-domain bytes the launches move
            lv['n'] += 1; lv['ms'] += ms.value; lv['fl'] += fl.value
            lv['conv_bytes'] += conv_alg_bytes(descs[(li + 1) % nl])
            lv['wd_bytes'] += wino_pending['in'] + wino_pending['gemm'] + wino_pending['out']
            lv['gemm_bytes'] += wino_pending['gemm']
        elif kind.value in (5, 6):
            nbytes = wino_pending['gemm']
        elif kind.value == 8:                        # fused stem (csrc/stem.hip): NCHW image in, filters, POOLED map out
            d8 = descs[(li + 1) % nl]
            nbytes = 4.0 * (d8.B * d8.H * d8.W * 3 + 64 * 147 + d8.B * ((d8.Ho - 1) // 2 + 1) * ((d8.Wo - 1) // 2 + 1) * 64)
        else:
            nbytes = conv_alg_bytes(descs[(li + 1) % nl])
            if kind.value in (2, 9):                 # DCNv2: + the 27-channel offset / mask-logit tensor the gather reads
                dd_ = descs[(li + 1) % nl]
                nbytes += 4.0 * dd_.B * dd_.Ho * dd_.Wo * 27
            if kind.value == 13:                     # the chain launch also writes the next block's conv1 output (64 channels)
                dd_ = descs[(li + 1) % nl]
                nbytes += 4.0 * dd_.B * dd_.Ho * dd_.Wo * 64
        if kind.value not in (3, 4):
            pk_ = (X3_PEAK_TFLOPS if tname.endswith('x3') else H2_PEAK_TFLOPS if (tname.endswith('h2') or tname.startswith('dcnp') or tname.startswith('ws') or tname.startswith('patch'))
                   else FP32_MFMA_PEAK_TFLOPS)
            t_m, t_h = fl.value / (pk_ * 1e12) * 1e3, nbytes / (HBM_PEAK_GBPS * 1e9) * 1e3
            bound_ms['mfma' if t_m >= t_h else 'hbm'] += max(t_m, t_h) / reps
            if kind.value in (5, 6):                 # + the two transform launches of that layer: pure HBM streams
                bound_ms['hbm'] += (wino_pending['in'] + wino_pending['out']) / (HBM_PEAK_GBPS * 1e9) * 1e3 / reps
        if kind.value not in (5, 6):
            li += 1
            lkey = ('winograd F(%dx%d,3x3) <gemm %s> (3 launches)' % (2 * kind.value - 4, 2 * kind.value - 4, tname)) \
                if kind.value in (3, 4) else 'stem_pool_k<%s,conv 7x7/2 + BN + ReLU + max-pool 3x3/2 fused>' % tname if kind.value == 8 \
                else 'pipe_h2_k<%s,DCNv2 gather>' % tname if kind.value == 9 \
                else 'pipe_h2_k<%s,convolution>' % tname if kind.value == 10 \
                else 'ws_h2_k<%s,convolution>' % tname if kind.value == 11 \
                else 'patch3x3_c64_k<%s,convolution>' % tname if kind.value == 14 \
                else 'chain_h2_k<conv3 + shortcut + ReLU -> next conv1, one launch>' if kind.value == 13 \
                else 'conv_igemm_f32<%s,loader%d>' % (tname, 3 if kind.value == 7 else kind.value)
            la = layers.setdefault(names[li % nl], [0.0, fl.value, lkey])
            la[0] += ms.value / reps
            la[2] = lkey
            tot_ms += ms.value; tot_fl += fl.value
        if kind.value not in (3, 4):
            key = ('conv_igemm_f32<%s,winograd grouped GEMM>' % tname) if kind.value in (5, 6) else \
                ('stem_pool_k<%s,fused stem>' % tname) if kind.value == 8 else \
                ('pipe_h2_k<%s,DCNv2 gather>' % tname) if kind.value == 9 else \
                ('pipe_h2_k<%s,convolution>' % tname) if kind.value == 10 else \
                ('ws_h2_k<%s,convolution>' % tname) if kind.value == 11 else \
                ('patch3x3_c64_k<%s,convolution>' % tname) if kind.value == 14 else \
                'chain_h2_k<pointwise chain>' if kind.value == 13 else \
                'conv_igemm_f32<%s,loader%d>' % (tname, 3 if kind.value == 7 else kind.value)
            a = by_kernel.setdefault(key, [0.0, 0.0, 0, 0.0, 0.0, 0.0])
            a[0] += ms.value; a[1] += fl.value; a[2] += 1; a[3] += nbytes
            a[4] += t_m; a[5] += t_h                 # lower-bound time of these launches on the matrix pipe / on HBM
    lib.ymi_prof_reset()
    dom = max(by_kernel.items(), key=lambda kv: kv[1][0])
    name, (dms, dfl, dn, dby, d_tm, d_th) = dom
    ach = dfl / (dms * 1e-3) / 1e12
    ach_gbps = dby / (dms * 1e-3) / 1e9
    # which roof bounds a kernel: the larger of its launches' summed matrix-pipe time (at the peak of the tile it runs) and
    # their summed HBM time (algorithmic bytes at 8 TB/s)
    detail = {k: {'ms_per_step': v[0] / reps, 'tflops': v[1] / (v[0] * 1e-3) / 1e12, 'launches_per_step': v[2] // reps,
                  'peak': round(kernel_peak(k), 1), 'frac': round(v[1] / (v[0] * 1e-3) / 1e12 / kernel_peak(k), 4),
                  'bound': 'mfma' if v[4] >= v[5] else 'hbm', 'alg_MB_per_launch': round(v[3] / v[2] / 1e6, 2),
                  'alg_GBps': round(v[3] / (v[0] * 1e-3) / 1e9, 1), 'hbm_frac': round(v[3] / (v[0] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                  'bound_frac': round(max(v[4], v[5]) / v[0], 4)}
              for k, v in by_kernel.items()}
    peak = kernel_peak(name)
    dom_bound = 'mfma' if d_tm >= d_th else 'hbm'
    # matrix-pipe utilisation of the whole engine: time the pipe would need at each launch's own peak / time taken
    eng_ms = sum(v[0] for v in by_kernel.values())
    eng_ideal_ms = sum(v[1] / (kernel_peak(k) * 1e12) * 1e3 for k, v in by_kernel.items())
    x3_ms = sum(v[0] for k, v in by_kernel.items() if kernel_peak(k) == X3_PEAK_TFLOPS)
    h2_ms = sum(v[0] for k, v in by_kernel.items() if kernel_peak(k) == H2_PEAK_TFLOPS)
    wino2 = sum(1 for v in layers.values() if v[2].startswith('winograd F(2x2'))
    wino4 = sum(1 for v in layers.values() if v[2].startswith('winograd F(4x4'))
    tr, tr_src = traffic_from_profiles(name)
    head = ({'bound': 'mfma', 'kernel': name, 'achieved': round(ach, 2), 'peak': round(peak, 1), 'unit': 'TFLOP/s',
             'frac': round(ach / peak, 4)} if dom_bound == 'mfma' else
            {'bound': 'hbm', 'kernel': name, 'achieved': round(ach_gbps, 1), 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
             'frac': round(ach_gbps / HBM_PEAK_GBPS, 4)})
    head.update({
        'traffic': tr, 'traffic_source': tr_src,
        'bound_basis': 'per launch max(FLOPs / peak of the tile it runs, algorithmic bytes / 8 TB/s), summed over the kernel\'s '
                       'launches: matrix pipe %.3f ms vs HBM %.3f ms per step -> %s-bound' % (d_tm / reps, d_th / reps, dom_bound),
        'mfma': {'achieved': round(ach, 2), 'peak': round(peak, 1), 'unit': 'TFLOP/s', 'frac': round(ach / peak, 4)},
        'hbm': {'achieved': round(ach_gbps, 1), 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s', 'frac': round(ach_gbps / HBM_PEAK_GBPS, 4),
                'alg_bytes_per_launch': dby / dn},
        # the honest step-level bound SURVEY 8(d) asks for when the peak is a bf16-class one: sum over every conv-layer launch
        # of max(F / peak of the tile actually used, algorithmic bytes / HBM peak) (+ the Winograd transforms as HBM streams)
        'bound_sum_ms': round(bound_ms['mfma'] + bound_ms['hbm'], 3),
        'bound_sum': {'mfma_bound_launches_ms': round(bound_ms['mfma'], 3), 'hbm_bound_launches_ms': round(bound_ms['hbm'], 3),
                      'measured_ms': round(tot_ms / reps, 3),
                      'frac': round((bound_ms['mfma'] + bound_ms['hbm']) / (tot_ms / reps), 4)},
    })
    head.update({
        # continuity with round 1, where every tile ran on the exact-fp32 MFMA: the same achieved rate against THAT peak
        'achieved_vs_fp32_mfma_peak': round(ach / FP32_MFMA_PEAK_TFLOPS, 4),
        'peak_basis': ('fp32 MFMA (v_mfma_f32_32x32x2_f32), 157.3 TFLOP/s' if peak == FP32_MFMA_PEAK_TFLOPS else
                       'fp16x2 tile: every fp32 operand carried as two fp16 pieces (round to nearest, power-of-two scale per '
                       'tensor / filter row), 3 piece products per fp32-class product on v_mfma_f32_32x32x16_f16 with fp32 '
                       'accumulate -> peak = 2500 / 3 = 833.3 TFLOP/s of fp32-equivalent work (executed fp16 FLOPs = 3 x '
                       'achieved)' if peak == H2_PEAK_TFLOPS else
                       'bf16x3 tile: every fp32 operand split exactly into 3 bf16 pieces, 6 piece products per fp32-class '
                       'product on v_mfma_f32_32x32x16_bf16 with fp32 accumulate -> peak = 2500 / 6 = 416.7 TFLOP/s of '
                       'fp32-equivalent work (executed bf16 FLOPs = 6 x achieved)'),
        'measured': 'HIP events on the launch stream, %d serialised passes right after the timed region' % reps,
        'flops_basis': 'FLOPs the launch executes on the matrix cores (for a direct conv launch = the algorithmic conv '
                       'FLOPs; for the Winograd GEMM launch = 16 (F(2x2,3x3)) or 36 (F(4x4,3x3)) GEMMs [T x C] x [C x Cout], '
                       'i.e. the layer\'s algorithmic FLOPs / 2.25 resp. / 4)',
        'avg_launch_ms': round(dms / dn, 4), 'flops_per_launch': dfl / dn,
        'all_conv': {'ms_per_step': round(tot_ms / reps, 3), 'tflops': round(tot_fl / (tot_ms * 1e-3) / 1e12, 2),
                     'gflop_per_step': round(tot_fl / reps / 1e9, 2),
                     'frac': round(tot_fl / (tot_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
                     'basis': 'ALGORITHMIC conv FLOPs (SURVEY 8(d): 118.28 GFLOP/image) over the summed durations of '
                              'every conv-layer launch incl. the Winograd transforms; of %d layers %d run Winograd '
                              'F(2x2,3x3) (2.25x fewer multiplications) and %d F(4x4,3x3) (4x fewer), so this figure '
                              'can exceed what the matrix cores execute' % (len(layers), wino2, wino4)},
        'engine': {'kernel': 'conv_igemm_f32<*> + pipe_h2_k<*> (every instantiation: direct loaders, pipelined kernel, grouped Winograd GEMM)',
                   'ms_per_step': round(eng_ms / reps, 3),
                   'executed_tflops': round(sum(v[1] for v in by_kernel.values()) / (eng_ms * 1e-3) / 1e12, 2),
                   'frac': round(eng_ideal_ms / eng_ms, 4),
                   'executed_vs_fp32_mfma_peak': round(sum(v[1] for v in by_kernel.values()) / (eng_ms * 1e-3) / 1e12
                                                       / FP32_MFMA_PEAK_TFLOPS, 4),
                   'x3_share_of_time': round(x3_ms / eng_ms, 3), 'h2_share_of_time': round(h2_ms / eng_ms, 3),
                   'basis': 'fp32(-equivalent) FLOPs executed by all GEMM launches of a step / their summed durations; frac '
                            '= matrix-pipe time at each launch\'s own peak (157.3 exact-fp32 tiles, 416.7 bf16x3 tiles, 833.3 '
                            'fp16x2 tiles) / time taken'},
        'per_kernel': detail,
    })
    if lv['n']:
        lms = lv['ms'] / reps
        head['layer_view'] = {
            'what': 'the Winograd layers as LAYERS (input transform + grouped GEMM + output transform = 3 launches each): the '
                    'headline roofline above describes the GEMM launch alone on Winograd-domain bytes (V + U + M), which is not '
                    'what the convolution needs',
            'winograd_layers': lv['n'] // reps, 'ms_per_step': round(lms, 3),
            'alg_tflops': round(lv['fl'] / reps / (lms * 1e-3) / 1e12, 1),
            'frac_of_fp16x2_peak': round(lv['fl'] / reps / (lms * 1e-3) / 1e12 / H2_PEAK_TFLOPS, 4),
            'conv_alg_MB_per_step': round(lv['conv_bytes'] / reps / 1e6, 1),
            'winograd_domain_MB_per_step': round(lv['wd_bytes'] / reps / 1e6, 1),
            'gemm_launch_MB_per_step': round(lv['gemm_bytes'] / reps / 1e6, 1),
            'domain_over_conv_bytes': round(lv['wd_bytes'] / lv['conv_bytes'], 2),
            'conv_alg_GBps': round(lv['conv_bytes'] / reps / (lms * 1e-3) / 1e9, 1),
            'hbm_frac_on_conv_bytes': round(lv['conv_bytes'] / reps / (lms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
            'bound_ms_at_peaks': round(max(lv['fl'] / reps / (H2_PEAK_TFLOPS * 1e12), lv['conv_bytes'] / reps / (HBM_PEAK_GBPS * 1e9)) * 1e3, 3),
        }
        head['layer_view']['frac_of_layer_bound'] = round(head['layer_view']['bound_ms_at_peaks'] / lms, 4)
    return head, layers


def whole_path_roofline(rk, step_ms, batch):
    """The top-level `roofline` object (VERDICT r4 #1b): SURVEY 8(d)'s figure for the WHOLE path — algorithmic conv GFLOP per image
    x images/s of one GPU / the matrix-pipe peak of the precision actually used — with the dominant kernel (largest summed
    duration) on ITS matrix-pipe fraction under `dominant_kernel`; the Winograd-domain GB/s view of that kernel, which rounds 3 - 4
    showed at the top level, is `dominant_kernel.hbm_view`."""
    eng = rk['engine']
    peak = H2_PEAK_TFLOPS if eng['h2_share_of_time'] >= 0.5 else X3_PEAK_TFLOPS if eng['x3_share_of_time'] >= 0.5 else FP32_MFMA_PEAK_TFLOPS
    gflop = rk['all_conv']['gflop_per_step']
    ach = gflop / step_ms                                   # GFLOP / ms = TFLOP/s, one GPU
    dom = {k: rk[k] for k in ('kernel', 'avg_launch_ms', 'flops_per_launch', 'flops_basis', 'traffic', 'traffic_source', 'bound_basis',
                              'peak_basis', 'achieved_vs_fp32_mfma_peak')}
    dom.update({'bound': 'mfma', 'achieved': rk['mfma']['achieved'], 'peak': rk['mfma']['peak'], 'unit': 'TFLOP/s',
                'frac': rk['mfma']['frac'], 'hbm_view': dict(rk['hbm'], classified_bound=rk['bound'],
                                                             note='algorithmic bytes of THIS launch (for a grouped Winograd GEMM: V + U + M, '
                                                                  'Winograd-domain bytes, not what the convolution needs: see layer_view)')})
    head = {'bound': 'mfma', 'achieved': round(ach, 2), 'peak': round(peak, 1), 'unit': 'TFLOP/s', 'frac': round(ach / peak, 4),
            'traffic': rk['traffic'], 'traffic_source': rk['traffic_source'],
            'kernel': rk['kernel'],
            'basis': 'SURVEY 8(d): achieved = algorithmic conv FLOPs of a step (%.2f GFLOP = %d images x %.2f GFLOP) / the TIMED step '
                     '(%.3f ms, everything in it: convolutions, Winograd transforms, Detect, the gather, the host read) on one GPU; peak = '
                     'the matrix pipe at the precision used by %.0f %% of the GEMM time (fp16x2: 2500 / 3 = 833.3 TFLOP/s of fp32-class '
                     'products; bf16x3: 416.7; exact fp32 MFMA: 157.3).  `traffic` = PMC HBM bytes per launch of `kernel`, the dominant '
                     'kernel, whose own matrix-pipe fraction is dominant_kernel.frac'
                     % (gflop, batch, gflop / batch, step_ms, 100 * max(eng['h2_share_of_time'], eng['x3_share_of_time'], 0.0)
                        if peak != FP32_MFMA_PEAK_TFLOPS else 100.0),
            'frac_of_fp32_mfma_peak': round(ach / FP32_MFMA_PEAK_TFLOPS, 4),
            # algorithmic fp32-class products per second against the RAW 16-bit matrix pipe (2500 TFLOP/s: what a network that could
            # run every product as ONE fp16 MFMA would be priced against) - i.e. frac / 3 for an fp16x2 plan
            'frac_of_raw_fp16_pipe': round(ach / BF16_MFMA_PEAK_TFLOPS, 4),
            'dominant_kernel': dom}
    for k in ('bound_sum_ms', 'bound_sum', 'measured', 'all_conv', 'engine', 'per_kernel', 'layer_view'):
        if k in rk:
            head[k] = rk[k]
    return head


def tune_table_sha():
    import hashlib
    p = os.path.join(ROOT, 'yolact_amd', 'tune', 'gfx950.json')
    return hashlib.sha256(open(p, 'rb').read()).hexdigest()[:16] if os.path.exists(p) else None


def traffic_from_profiles(kernel):
    """(HBM bytes per launch of the dominant kernel, where the figure comes from).  PMC counters cannot be collected from
    inside this process, so the figure is STATIC: the committed summary of separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
    passes over this very command (FETCH_SIZE doubled per MI355X_MICROARCH.md; `tools/gpu_session.sh <name> traffic`).
    It is only reported when that summary was measured with the tile table this run uses (`tune_sha`), else null."""
    for fn in ('r06_traffic.json', 'r05_traffic.json', 'r04_traffic.json', 'r03_traffic.json', 'r02_traffic.json'):
        path = os.path.join(ROOT, 'profiles', fn)
        if os.path.exists(path):
            with open(path) as f:
                doc = json.load(f)
            rec = doc.get(kernel)
            sha = doc.get('tune_sha')
            if rec and sha is not None and sha == tune_table_sha():
                return rec['bytes_per_launch'], 'static: profiles/%s (separate PMC passes of the same command and tile table)' % fn
            if rec:
                return None, 'profiles/%s holds a PMC record for this kernel, but measured under another tile table: not reported' % fn
    return None, 'no PMC record for this kernel under profiles/'


def gpu_power_state():
    """What the box says about its own clocks / power limits (rocm-smi, when it answers within a few seconds): recorded next to
    the calibration so that a slow run can be told from a slow box."""
    import shutil
    import subprocess
    exe = shutil.which('rocm-smi') or '/opt/rocm/bin/rocm-smi'
    if not os.path.exists(exe):
        return {'source': 'none (rocm-smi not found)'}
    try:
        r = subprocess.run([exe, '-d', '0', '--showclocks', '--showpower', '--showmaxpower', '--showperflevel', '--showtemp', '--json'],
                           capture_output=True, text=True, timeout=15)
        doc = json.loads(r.stdout[r.stdout.index('{'):])
        card = doc.get('card0') or next(iter(doc.values()))
        keep = {k: v for k, v in card.items() if any(t in k.lower() for t in ('sclk', 'mclk', 'fclk', 'power', 'performance level',
                                                                            'temperature (sensor junction)', 'temperature (sensor memory)'))}
        return {'source': 'rocm-smi -d 0 --showclocks --showpower --showmaxpower --showperflevel --showtemp', **keep}
    except Exception as e:          # the calibration kernels below are the evidence; this is context
        return {'source': 'rocm-smi failed: %s: %s' % (type(e).__name__, str(e)[:120])}


def box_calibration(dev, seconds=0.25):
    """Two fixed micro-workloads timed on THIS box right before the timed region (csrc/calib.hip): the fp16 matrix
    pipe the fp16x2 tiles run on (register-resident v_mfma_f32_32x32x16_f16, random-mantissa operands, two waves per SIMD) and a
    float4 HBM copy of 1 GiB (far beyond the 256 MB Infinity Cache).  The MFMA loop runs for ~`seconds` so that the chip settles
    at the clock its power budget allows under matrix load — the figure is the rate over the LAST half of that interval, the
    first launches are reported separately (a box that starts cold shows a higher number there)."""
    from yolact_amd import _lib as L
    lib = L.lib()
    s = L.stream_ptr()
    out = torch.zeros(64, device=dev)
    fl = C.c_double()
    n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
    blocks, iters = 2 * n_cu, 20000

    def mfma():
        L.check(lib.ymi_calib_mfma_f16(out.data_ptr(), blocks, iters, C.byref(fl), s), 'ymi_calib_mfma_f16')
    mfma()
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    rates = []
    t_end = time.perf_counter() + seconds
    while time.perf_counter() < t_end or len(rates) < 8:
        evs[0].record()
        for _ in range(4):
            mfma()
        evs[1].record()
        evs[1].synchronize()
        rates.append(4 * fl.value / (evs[0].elapsed_time(evs[1]) * 1e-3) / 1e12)
    half = rates[len(rates) // 2:]
    n = (1 << 30) // 4
    src = torch.empty(n, dtype=torch.float32, device=dev).normal_()
    dst = torch.empty_like(src)
    by = C.c_double()

    def copy():
        L.check(lib.ymi_calib_hbm_copy(src.data_ptr(), dst.data_ptr(), n, C.byref(by), s), 'ymi_calib_hbm_copy')
    copy()
    torch.cuda.synchronize()
    evs[0].record()
    for _ in range(20):
        copy()
    evs[1].record()
    evs[1].synchronize()
    gbps = 20 * by.value / (evs[0].elapsed_time(evs[1]) * 1e-3) / 1e9
    # the same copy over footprints that FIT the 256 MB memory-side cache (source + destination = 16 / 64 / 192 MB, repeated): what a
    # tensor written by one launch and read by the next one sees.  The 1 GiB copy and the MFMA loop did not separate the ~1.8k from
    # the ~2.05k images/s boxes of this pool (sessions r5a ... r5x) while every HBM-side kernel of the step was 20 - 60 % slower
    # on the former: this is the axis those two numbers do not cover.
    cache_copy = {}
    for mb in (8, 32, 96):
        nn = mb * (1 << 20) // 4
        rep = max(4, 2048 // (2 * mb))

        def ccopy():
            L.check(lib.ymi_calib_hbm_copy(src.data_ptr(), dst.data_ptr(), nn, C.byref(by), s), 'ymi_calib_hbm_copy')
        for _ in range(2):
            ccopy()
        torch.cuda.synchronize()
        evs[0].record()
        for _ in range(rep):
            ccopy()
        evs[1].record()
        evs[1].synchronize()
        cache_copy['%dMB' % (2 * mb)] = round(rep * by.value / (evs[0].elapsed_time(evs[1]) * 1e-3) / 1e9, 1)
    # GPU-side cost of one DEPENDENT launch: 400 one-block launches back to back on the stream, start-to-end by HIP events (the
    # step is ~133 dependent launches; = max(host issue rate, the command processor's launch-to-launch time))
    evs[0].record()
    for _ in range(400):
        lib.ymi_calib_mfma_f16(out.data_ptr(), 1, 1, None, s)
    evs[1].record()
    evs[1].synchronize()
    dep_us = evs[0].elapsed_time(evs[1]) * 1e3 / 400
    # load-to-use LATENCY (ymi_calib_latency: one lane follows a random single-cycle permutation, one entry per 128-byte line) over
    # 1 MB (L2 after the warm pass), 64 MB and 2 GiB (memory + address translation: 16 M lines on distinct pages)
    latency = {}
    i32 = torch.int32
    res = torch.zeros(4, dtype=i32, device=dev)
    for label, mb, hops in (('1MB', 1, 20000), ('64MB', 64, 20000), ('2048MB', 2048, 10000)):
        lines = mb * (1 << 20) // 128
        perm = torch.randperm(lines, device=dev)
        chain = torch.zeros(lines * 32, dtype=i32, device=dev)
        chain[perm * 32] = (torch.roll(perm, -1) * 32).to(i32)
        starts = [int(v) * 32 for v in perm[torch.arange(3, device=dev) * hops % lines]]
        del perm
        best = None
        for r in range(3):          # every pass starts where the previous one stopped on the cycle: a footprint larger than L2 is
            torch.cuda.synchronize()    # never re-visited, the 1 MB one is warm from the second pass on
            evs[0].record()
            L.check(lib.ymi_calib_latency(chain.data_ptr(), chain.numel(), starts[r], hops, res.data_ptr(), s), 'ymi_calib_latency')
            evs[1].record()
            evs[1].synchronize()
            ns = evs[0].elapsed_time(evs[1]) * 1e6 / hops
            best = ns if best is None else min(best, ns)
        latency[label] = round(best, 1)
        del chain
    # L2-resident read stream (1 MB swept by 8 blocks per CU): the global -> CU path the GEMM tiles are bound by
    l2src = torch.randn(1 << 18, device=dev)
    l2b = C.c_double()

    def l2():
        L.check(lib.ymi_calib_l2_read(l2src.data_ptr(), l2src.numel(), 8 * n_cu, 16, out.data_ptr(), C.byref(l2b), s), 'ymi_calib_l2_read')
    l2()
    torch.cuda.synchronize()
    evs[0].record()
    for _ in range(10):
        l2()
    evs[1].record()
    evs[1].synchronize()
    l2_gbps = 10 * l2b.value / (evs[0].elapsed_time(evs[1]) * 1e-3) / 1e9
    # the HOST half: what one C-ABI call and one asynchronous kernel launch cost from this Python process on this box (the step is
    # ~190 such calls; slow hosts have measured 20 % lower batch-1 numbers with identical kernels)
    t0 = time.perf_counter()
    for _ in range(2000):
        lib.ymi_abi_version()
    call_us = (time.perf_counter() - t0) / 2000 * 1e6
    small = torch.zeros(1024, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300):
        lib.ymi_calib_hbm_copy(small.data_ptr(), small.data_ptr() + 2048, 512, None, s)
    launch_us = (time.perf_counter() - t0) / 300 * 1e6
    torch.cuda.synchronize()
    del src, dst
    host = {'ctypes_call_us': round(call_us, 2), 'async_kernel_launch_us': round(launch_us, 2)}
    try:
        with open('/proc/cpuinfo') as f:
            names = [ln.split(':', 1)[1].strip() for ln in f if ln.startswith('model name')]
        host['cpu_model'] = names[0] if names else None
        host['logical_cpus'] = len(names)
        host['kernel'] = os.uname().release
        with open('/proc/loadavg') as f:
            host['loadavg_1m'] = float(f.read().split()[0])
    except OSError:
        pass
    # which physical GPU this was (kfd's unique_id): the pool's boxes differ, and two runs on the same id are directly comparable
    gpu_ids = []
    import glob
    for pth in sorted(glob.glob('/sys/class/kfd/kfd/topology/nodes/*/properties')):
        try:
            with open(pth) as f:
                props = dict(ln.split(None, 1) for ln in f.read().splitlines() if len(ln.split(None, 1)) == 2)
            if int(props.get('simd_count', '0')) > 0:
                gpu_ids.append(props.get('unique_id', '').strip())
        except (OSError, ValueError):
            continue
    host['gpu_unique_ids'] = gpu_ids
    return {'host': host, 'mfma_f16_tflops': round(sum(half) / len(half), 1), 'mfma_f16_tflops_first_launches': round(rates[0], 1),
            'mfma_f16_frac_of_2500': round(sum(half) / len(half) / BF16_MFMA_PEAK_TFLOPS, 4),
            'hbm_copy_GBps': round(gbps, 1), 'hbm_copy_frac_of_8000': round(gbps / HBM_PEAK_GBPS, 4),
            'l2_read_GBps': round(l2_gbps, 1), 'cache_copy_GBps_by_footprint': cache_copy, 'dependent_launch_us': round(dep_us, 2),
            'load_latency_ns_by_footprint': latency,
            'device': torch.cuda.get_device_name(dev), 'compute_units': n_cu, 'power_state': gpu_power_state(),
            'what': 'csrc/calib.hip, timed with HIP events on the launch stream right before the warm-up steps: %d x 4 waves of '
                    'register-resident v_mfma_f32_32x32x16_f16 (random mantissas) for %.2f s, rate of the last half of the interval; '
                    'float4 copy of 1 GiB (read + write bytes) x 20; the same copy over 16 / 64 / 192 MB footprints (inside the 256 MB '
                    'memory-side cache); 400 dependent one-block launches; a one-lane dependent-load chain over 1 MB / 64 MB / 2 GiB (best of 3)' % (blocks, seconds)}


def cpu_baseline(sd, size, batch=8, budget_s=14.0):
    """The CPU oracle (port of the reference's forward + Detect) on this host's cores, bounded sample of the TIMED workload.
    The intra-op thread count is swept ON THE BATCH-8 WORKLOAD ITSELF (round 3 swept on batch 4 and then timed batch 8: the
    "best" count was picked on another workload): one batch per candidate (4 ... 128 threads, ascending, stopping once more
    threads clearly hurt — torch's default of one thread per logical CPU oversubscribes a 256-CPU host: 0.05 images/s); the two
    best settings are then timed three batches each and the better MEDIAN is reported with its thread count."""
    import statistics
    import yolact_amd
    from oracle import yolact_oracle as O
    from yolact_amd.utils.synth import synth_images
    cfg = yolact_amd.CONFIGS[CONFIG].copy()
    ncpu = os.cpu_count() or 1
    try:
        naff = len(os.sched_getaffinity(0))
    except AttributeError:
        naff = ncpu
    default_threads = torch.get_num_threads()
    x = synth_images(batch, size, size, seed=1234)           # the very batch the GPU path is timed on
    cands = sorted({t for t in (4, 8, 16, 32, 64, 128) if 1 <= t <= naff} | {max(1, min(naff, 4))})   # (never empty: a 1 - 3 CPU mask gets its own size)

    def one_batch():
        t0 = time.perf_counter()
        O.detect(O.forward_raw(x, sd, cfg), cfg)
        return time.perf_counter() - t0
    sweep, runs = {}, {}
    t_all = time.perf_counter()
    with torch.no_grad():
        for t in cands:
            torch.set_num_threads(t)
            O.detect(O.forward_raw(x[:1], sd, cfg), cfg)     # warm-up (thread pool, oneDNN primitive cache)
            sweep[t] = round(batch / one_batch(), 3)
            if sweep[t] < 0.7 * max(sweep.values()):
                break
        top = sorted(sweep, key=sweep.get, reverse=True)[:2]
        for t in top:
            torch.set_num_threads(t)
            O.detect(O.forward_raw(x[:1], sd, cfg), cfg)
            runs[t] = [round(batch / one_batch(), 3) for _ in range(3)]
    torch.set_num_threads(default_threads)
    med = {t: statistics.median(v) for t, v in runs.items()}
    best = max(med, key=med.get)
    dt = time.perf_counter() - t_all
    return {'value': round(med[best], 3), 'unit': 'images/s', 'cores': best, 'threads': best, 'host_logical_cpus': ncpu,
            'cpus_in_affinity_mask': naff, 'kind': 'port',
            'thread_sweep_images_per_s': {str(k): v for k, v in sweep.items()},
            'scaling_note': ('torch-CPU (oneDNN) does not scale past %d threads on this workload on this host: the sweep %s stops where more '
                             'threads hurt; `cores` is the thread count of the best MEDIAN, not the host\'s %d logical CPUs'
                             % (best, {k: v for k, v in sweep.items()}, ncpu)),
            'timed_runs_images_per_s': {str(k): v for k, v in runs.items()},
            'sample': 'median of 3 batches of %d (= the timed workload, %dx%d) forward+Detect through oracle/yolact_oracle.py at '
                      'each of the two best thread counts of a sweep over the same batch (%d + %d batches, %.0f s of host time '
                      'in all); the reference checkout does not exist on the GPU box, so this is the port, not eval.py itself '
                      '(tests/test_reference_timing.py times both side by side where the reference exists); torch %s CPU fp32'
                      % (batch, size, size, len(sweep), 3 * len(runs), dt, torch.__version__)}


def secondary_lines(net, x, size, steps=10, config=CONFIG):
    """The reference's FPS definitions next to the headline (SURVEY 8(d) "Secondary"): (a) the same batch
    with postprocess() to size x size inside the step (one lincomb + one upsample launch for the whole batch), (b) batch 1
    end to end exactly like eval.py:264-281 prep_benchmark — forward, Detect, postprocess, the top-k (5) `.cpu().numpy()`
    copies, synchronize — which is what the reference's published Titan Xp numbers measure."""
    from yolact_amd.layers.output_utils import postprocess, postprocess_batch
    out = {}

    def timed(fn, n):
        # median of three groups of n: the pool's boxes stall for 15 - 40 ms now and then (profiles/r06_overlap_stability.txt), which is
        # +1.5 ms per step on a single group of 10
        for _ in range(3):
            fn()
        groups = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            groups.append((time.perf_counter() - t0) / n)
        return statistics.median(groups)

    B = x.shape[0]

    def step_post():
        o = net.forward_device(x)
        r = postprocess_batch(o, size, size)
        r['count'].tolist()

    def step_fwd():
        net.forward_device(x)['count'].tolist()
    t_post = timed(step_post, steps)
    t_fwd = timed(step_fwd, steps)
    mask_bytes = float(net.forward_device(x)['count'].sum().item()) * size * size * 4
    out['batch_with_postprocess'] = {
        'value': round(B / t_post, 2), 'unit': 'images/s', 'ms_per_step': round(t_post * 1e3, 3),
        'postprocess_ms_per_step': round((t_post - t_fwd) * 1e3, 3),
        'mask_write_GBps': round(mask_bytes / max(t_post - t_fwd, 1e-9) / 1e9, 1),
        'what': 'batch %d forward + Detect + postprocess to %dx%d (float32 {0,1} masks for every detection, %.0f MB '
                'written per step) + host read of the counts' % (B, size, size, mask_bytes / 1e6)}
    x1 = x[:1].contiguous()

    def step_ref_fps():
        preds = net(x1)
        t = postprocess(preds, size, size, crop_masks=True, score_threshold=0)
        classes, scores, boxes, masks = [v[:5] for v in t]
        if isinstance(scores, list):                 # YOLACT++ (eval.py:270-272): [box scores, box scores * maskiou]
            scores[0].cpu().numpy(); scores[1].cpu().numpy()
        else:
            scores.cpu().numpy()
        classes.cpu().numpy(); boxes.cpu().numpy(); masks.cpu().numpy()
        torch.cuda.synchronize()
    t1 = timed(step_ref_fps, 4 * steps)
    # where that time goes (each stage timed on its own, synchronised on both sides; the stages overlap a little in the real step)
    def only_net():
        net(x1)
        torch.cuda.synchronize()
    preds1 = net(x1)

    def only_post():
        postprocess([{'detection': dict(preds1[0]['detection']), 'net': net}], size, size, crop_masks=True, score_threshold=0)
        torch.cuda.synchronize()
    tpost = postprocess([{'detection': dict(preds1[0]['detection']), 'net': net}], size, size, crop_masks=True, score_threshold=0)

    def only_copy():
        for v in tpost:
            for u in (v if isinstance(v, list) else [v]):
                u[:5].cpu().numpy()
        torch.cuda.synchronize()
    brk = {'net(x) + sync': round(timed(only_net, 2 * steps) * 1e3, 3), 'postprocess + sync': round(timed(only_post, 2 * steps) * 1e3, 3),
           'top-5 .cpu().numpy() copies + sync': round(timed(only_copy, 2 * steps) * 1e3, 3),
           'bytes_copied': int(sum(u[:5].numel() * u.element_size() for v in tpost for u in (v if isinstance(v, list) else [v])))}
    out['reference_fps_definition_batch1'] = {
        'value': round(1.0 / t1, 2), 'unit': 'images/s (FPS)', 'ms_per_image': round(t1 * 1e3, 3), 'breakdown_ms': brk,
        'what': 'eval.py:264-281 prep_benchmark: batch 1 net(x) + postprocess to %dx%d + top-5 .cpu().numpy() copies + '
                'sync (the definition behind the reference README FPS column)' % (size, size)}
    # (c) the "pretrained-like" sparse regime SURVEY 8(d) asks to time next to the dense worst case: same network, class
    # logits shaped so that ~1 % of the priors pass the 0.05 candidate threshold (tests/golden/r50_few is this recipe)
    from yolact_amd.utils.synth import synth_state_dict
    from yolact_amd.yolact import Yolact
    net_s = Yolact()
    net_s.load_state_dict_compat(synth_state_dict([(k, tuple(v.shape)) for k, v in net_s.state_dict().items()], seed=8,
                                                  conf_gain=0.2, bg_bias=19.5))
    net_s.detect.use_fast_nms = True
    net_s = net_s.to(x.device)

    def step_sparse():
        net_s.forward_device(x)['count'].tolist()
    t_s = timed(step_sparse, steps)
    o = net_s.forward_device(x)
    conf = net_s.plan_for(x).conf[..., :net_s.plan_for(x).Ccls]
    kept = (torch.softmax(conf, -1)[..., 1:].amax(-1) > 0.05).float().sum(1)
    cnt = o['count'].tolist()
    over = [int((o['score'][b, :cnt[b]] > 0.15).sum()) for b in range(B)]
    out['sparse_regime'] = {
        'value': round(B / t_s, 2), 'unit': 'images/s', 'ms_per_step': round(t_s * 1e3, 3),
        'candidate_priors_per_image': round(float(kept.mean()), 1), 'detections_over_0.15_per_image': over,
        'what': 'batch %d forward + Detect with ~1 %% of the %d priors over the candidate threshold (dense headline: ~70 %%): '
                'Detect does less selection work, the convolutions are unchanged' % (B, conf.shape[1])}
    return out


def outlier_plan_line(dev, x, size, config, k_exp=16, steps=10, rebalance=True):
    """VERDICT r5 #3 / #5b: what the step costs when a checkpoint TRIPS the outlier-channel guard.  The same network re-parametrised
    exactly (utils.synth.plant_outlier_channels: channels of C3 .. C5 and of proto_net[0] scaled by 2^16, their consumers' filters by
    2^-16 — what BN-folded checkpoints with outlier channels look like): engine.Packed.tiny_columns moves the consuming layers to the
    bf16x3 tiles, without Winograd and without the fp16x2-only fusions.  Tiles of shapes the shipped table does not hold for that
    arithmetic are measured in this process (tune_misses says how many)."""
    import warnings
    from yolact_amd.utils.synth import plant_outlier_channels, synth_state_dict
    from yolact_amd.yolact import Yolact
    import yolact_amd
    yolact_amd.set_cfg(config)
    net_o = Yolact()
    sd0 = synth_state_dict([(k, tuple(v.shape)) for k, v in net_o.state_dict().items()], seed=0, conf_gain=0.04)
    sd, planted = plant_outlier_channels({k: v.cpu() for k, v in sd0.items()}, k_exp)
    net_o.load_state_dict_compat(sd)
    net_o.detect.use_fast_nms = True
    net_o = net_o.to(dev)
    old_rb = os.environ.get('YOLACT_AMD_REBALANCE')
    os.environ['YOLACT_AMD_REBALANCE'] = '1' if rebalance else '0'
    try:
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            plan = net_o.plan_for(x)
    finally:
        if old_rb is None:
            os.environ.pop('YOLACT_AMD_REBALANCE', None)
        else:
            os.environ['YOLACT_AMD_REBALANCE'] = old_rb

    def step():
        net_o.forward_device(x)['count'].tolist()
    for _ in range(3):
        step()
    per, groups = [], []
    for _ in range(3):                      # median of three groups (see secondary_lines.timed)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            t1 = time.perf_counter()
            step()
            per.append(round((time.perf_counter() - t1) * 1e3, 3))
        torch.cuda.synchronize()
        groups.append((time.perf_counter() - t0) / steps)
    dt = statistics.median(groups)
    if os.environ.get('BENCH_DEBUG'):
        print('outlier_plan_line rebalance=%s per-step ms %s' % (rebalance, per), file=sys.stderr)
    wide = sorted(plan.ops[i][2] for i in plan.wide_ops)
    return {'value': round(x.shape[0] / dt, 2), 'unit': 'images/s', 'ms_per_step': round(dt * 1e3, 3),
            'layers_on_bf16x3': len(wide), 'first_layers': wide[:6], 'tune_misses': plan.tune_misses,
            'rebalanced': [list(r) for r in getattr(plan, 'rebalanced', [])],
            'what': 'configs[1] step with the 2^%d outlier checkpoint (exact re-parametrisation of the timed network: %s): %s'
                    % (k_exp, ', '.join(n for n, _ in planted),
                       'compensated outlier channels rebalanced at pack time (Plan._rebalance_outliers: [tensor, channels, max exponent]), '
                       'every layer stays on its fp16x2 tile' if rebalance else
                       'YOLACT_AMD_REBALANCE=0: the outlier-channel guard demotes %d layers to bf16x3 tiles (no Winograd, no fp16x2-only '
                       'fusions there)' % len(wide))}


def relaunch_under_torchrun(args):
    """`python bench.py --gpus N` outside a launcher: start the N ranks ourselves (one process per GPU, RCCL rendezvous on
    127.0.0.1) and exit with their status."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=8, help='images per GPU')
    ap.add_argument('--size', type=int, default=0, help='input size (default: the config\'s max_size: 550, or 700 for im700)')
    ap.add_argument('--config', default=CONFIG, help='other BASELINE configs (parity-test cases; the metric is quoted on '
                    'the default yolact_resnet50_config)')
    ap.add_argument('--with-postprocess', action='store_true')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-secondary', action='store_true')
    ap.add_argument('--no-calibration', action='store_true', help='skip the box calibration block (csrc/calib.hip)')
    ap.add_argument('--no-pipeline', action='store_true', help='block on the host read of every step before launching the next')
    ap.add_argument('--exchange-after-join', action='store_true', help='A/B: enqueue the record gather + count read behind the whole '
                    'forward (round 4 behaviour) instead of right behind Detect on its stream')
    ap.add_argument('--prealloc-gb', type=int, default=0, help='experiment: reserve ONE allocation of this size in torch\'s caching '
                    'allocator before anything else is allocated, so that weights / activations / workspaces are carved from one mapping')
    ap.add_argument('--layers', action='store_true', help='also print the per-layer conv table to stderr')
    ap.add_argument('--global-batch', type=int, default=0, help='strong scaling: a FIXED global batch split over the ranks '
                    '(parallel.shard_range; eval.py:630-634 splits one batch the same way).  Default: --batch, i.e. BASELINE '
                    'configs[1] at 8 GPUs = one image per GPU.  The weak-scaling region (--batch images per GPU) stays the headline value; '
                    'this second timed region is reported as `strong_scaling`')
    ap.add_argument('--no-strong', action='store_true', help='skip the strong-scaling region')
    ap.add_argument('--step-overlap', type=int, default=int(os.environ.get('YOLACT_AMD_STEP_OVERLAP', '4')), choices=(1, 2, 3, 4),
                    help='N > 1: consecutive steps rotate over N plan instances (yolact_amd.pipeline.BatchPipeline) on N HIP streams, so '
                         'that batch i + 1 starts while batch i is still in its tail (2: every plan also forks its side stream; 3, 4: one '
                         'stream per plan); 1: one plan (every step behind the last)')
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit('--gpus must be >= 1')

    launched = 'WORLD_SIZE' in os.environ and 'RANK' in os.environ
    n_dev = torch.cuda.device_count()
    if n_dev < args.gpus:
        raise SystemExit('bench.py: --gpus %d requested but this box has %d visible GPU(s); refusing to report a '
                         '%d-GPU number measured on fewer devices' % (args.gpus, n_dev, args.gpus))
    if not launched and args.gpus > 1:
        relaunch_under_torchrun(args)
    # stdout carries exactly ONE line — the result JSON.  RCCL prints a version banner through C stdio at start-up /
    # tear-down, and buffered library output can land after Python's own line; so file descriptor 1 is pointed at stderr
    # for the whole run and the JSON is written to the saved original descriptor at the very end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if args.gpus != world:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    import torch.distributed as dist
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:                                     # one launch loop per GPU: give each rank its own slice of the host's CPUs
        from yolact_amd import parallel as _par
        _par.pin_rank_affinity(local_rank, int(os.environ.get('LOCAL_WORLD_SIZE', world)))
    rccl_error = None
    if launched:
        dist.init_process_group('nccl', device_id=dev)                                 # RCCL over xGMI
    else:
        # single process: still bring RCCL up (world 1) so the gather the path performs is the real collective
        import socket
        s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
        try:
            dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % port, rank=0, world_size=1, device_id=dev)
        except Exception as e:      # an RCCL-less box must still produce the single-GPU line; the JSON says so
            rccl_error = '%s: %s' % (type(e).__name__, e)

    import yolact_amd
    from yolact_amd import parallel
    from yolact_amd.layers.output_utils import postprocess_batch
    from yolact_amd.utils.synth import synth_images
    size = args.size or int(yolact_amd.CONFIGS[args.config].max_size)
    with torch.no_grad():
        if args.prealloc_gb > 0:
            _pre = torch.empty(args.prealloc_gb << 30, dtype=torch.uint8, device=dev)
            del _pre                                   # stays cached: later allocations are split from this block
        net, sd = build_model(dev, size, args.config)
        x = synth_images(args.batch, size, size, seed=1234 + rank).to(dev)   # resident in HBM
        have_pg = dist.is_initialized()
        got = {'n': 0}

        # One step = forward + Detect + the RCCL gather of the records + the host read of the per-image counts (rank 0).
        # The host read of step i is issued as an asynchronous device-to-host copy right behind that step's kernels and
        # COLLECTED after step i+1 has been launched (depth-2 software pipeline, two pinned buffers), so the GPU does not
        # idle while the host turns around; every step's counts still reach the host, in order, inside the timed region.
        # --no-pipeline restores the launch / blocking read / launch sequence.
        host_counts = [torch.empty(args.batch * world, dtype=torch.float32, pin_memory=True) for _ in range(2 * max(1, args.step_overlap))]
        turn = {'i': 0}
        gatherer = parallel.RecordGatherer(0)        # persistent receive buffers: no allocation, no torch.cat per step
        # --step-overlap 2: a second plan instance (slot 1) with its own gather buffers, driven on its own stream
        NOV = args.step_overlap
        gatherers = [gatherer] + [parallel.RecordGatherer(0) for _ in range(NOV - 1)]
        from yolact_amd.pipeline import BatchPipeline
        pipeline = BatchPipeline(net, NOV, dev) if NOV > 1 else None      # the product's throughput mode (yolact_amd/pipeline.py)
        lane = {'k': 0}

        def exchange(out):
            gatherer = gatherers[lane['k']]
            # (called by forward_device right behind Detect, on the stream Detect runs on — the way Yolact.forward_sharded /
            #  parallel.sharded_forward enqueue it: the records do not depend on the prototypes)
            # pack_records = the record tensor the Detect selection kernel wrote itself (no torch op)
            rec = gatherer(parallel.pack_records(out), args.batch, force_collective=have_pg)
            if rec is None:
                return None
            buf = host_counts[turn['i'] % len(host_counts)]
            turn['i'] += 1
            buf[:rec.shape[0]].copy_(rec[:, 0], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            return (buf, ev, int(rec.shape[0]))

        def launch_on(slot):
            lane['k'] = slot
            if args.exchange_after_join:
                out = net.forward_device(x, slot=slot)
                handle = exchange(out)
            else:
                out = net.forward_device(x, after_detect=exchange, slot=slot)
                handle = out.pop('after_detect')
            if args.with_postprocess:
                postprocess_batch(out, size, size)
            return handle

        def launch():
            if pipeline is None:
                return launch_on(0)
            if args.exchange_after_join or args.with_postprocess:       # (these variants consume the outputs on the slot's stream)
                return pipeline.run_in_slot(x, launch_on)[1]
            lane['k'] = pipeline._n % NOV
            return pipeline.submit(x, after_detect=exchange).pop('after_detect')

        def collect(handle):
            if handle is not None:
                buf, ev, n = handle
                ev.synchronize()
                counts = buf[:n].tolist()                           # the per-image detection counts, on the host
                got['n'] = len(counts)

        issue_s = []                                    # host time to ISSUE one step (launch loop + gather + copy request), per step

        def run_steps(k):
            # the host reads of the last `depth - 1` steps are still outstanding while the next step is issued
            # (depth 2 with one plan; with N plan instances in rotation N steps may be in flight on the device)
            depth, pend = max(2, args.step_overlap), []
            for _ in range(k):
                t_is = time.perf_counter()
                cur = launch()
                issue_s.append(time.perf_counter() - t_is)
                if args.no_pipeline:
                    collect(cur)
                else:
                    pend.append(cur)
                    if len(pend) >= depth:
                        collect(pend.pop(0))
            for h in pend:
                collect(h)

        net.plan_for(x)                          # plan build (weight packing, table look-ups) is set-up, not a step
        if pipeline is not None:
            pipeline.warm(x)
        calib = box_calibration(dev) if (rank == 0 and not args.no_calibration) else None
        run_steps(args.warmup)
        if have_pg:
            dist.barrier()
        torch.cuda.synchronize()
        del issue_s[:]
        t0 = time.perf_counter()
        run_steps(args.steps)
        if have_pg:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        per_rank_ms, gather_us = None, None
        if have_pg:
            mine_t = torch.tensor([dt], device=dev, dtype=torch.float64)
            every = [torch.zeros_like(mine_t) for _ in range(world)]
            dist.all_gather(every, mine_t)                       # every rank's own wall time of the timed region
            per_rank_ms = [round(float(v.item()) / args.steps * 1e3, 3) for v in every]
            t = mine_t.clone()
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
            # the collective alone: the record gather of the last step repeated 50 times, host-paired wall time
            # per call (launch + RCCL kernel + completion), after the timed region
            last = parallel.pack_records(net.forward_device(x))
            torch.cuda.synchronize()
            dist.barrier()
            tg = time.perf_counter()
            for _ in range(50):
                gatherer(last, args.batch, force_collective=True)
            torch.cuda.synchronize()
            gather_us = round((time.perf_counter() - tg) / 50 * 1e6, 1)

        # ---- strong scaling: a FIXED global batch G split over the ranks, the reference's own multi-GPU form
        # (eval.py:630-634,661).  Rank r computes images shard_range(G, r, world) — an empty share enters the same gather — through
        # parallel.sharded_forward's per-rank-shard form; same barrier / synchronize / MAX-over-ranks timing as the weak region.
        strong = None
        if not args.no_strong:
            G = args.global_batch or args.batch
            lo_s, hi_s = parallel.shard_range(G, rank, world)
            xs = synth_images(max(hi_s - lo_s, 1), size, size, seed=4321 + rank).to(dev)[:hi_s - lo_s]
            sg = parallel.RecordGatherer(0)
            pinned = [torch.empty(G, dtype=torch.float32, pin_memory=True) for _ in range(2)]
            st = {'i': 0, 'n': 0}

            def s_launch():
                # (world 1: the RCCL gather is forced like in the weak region, so both regions pay the same collective)
                rec, _ = parallel.sharded_forward(net.forward_device, xs, net.mask_dim, sg, 0, n_global=G) if world > 1 else (
                    sg(parallel.pack_records(net.forward_device(xs)), G, force_collective=have_pg), None)
                if rec is None:
                    return None
                buf = pinned[st['i'] & 1]
                st['i'] += 1
                buf[:rec.shape[0]].copy_(rec[:, 0], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                return (buf, ev, int(rec.shape[0]))

            def s_collect(h):
                if h is not None:
                    h[1].synchronize()
                    st['n'] = len(h[0][:h[2]].tolist())

            def s_run(k):
                prev = None
                for _ in range(k):
                    cur = s_launch()
                    s_collect(prev)
                    prev = cur
                s_collect(prev)
            s_run(args.warmup)
            if have_pg:
                dist.barrier()
            torch.cuda.synchronize()
            ts0 = time.perf_counter()
            s_run(args.steps)
            if have_pg:
                dist.barrier()
            torch.cuda.synchronize()
            dts = time.perf_counter() - ts0
            if have_pg:
                tt = torch.tensor([dts], device=dev, dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                dts = float(tt.item())
            if rank == 0:
                if st['n'] != G:
                    raise SystemExit('bench.py: strong-scaling region gathered %d records, expected %d' % (st['n'], G))
                strong = {'global_batch': G, 'images_per_rank': [parallel.shard_range(G, r, world)[1] - parallel.shard_range(G, r, world)[0] for r in range(world)],
                          'value': round(G * args.steps / dts, 2), 'unit': 'images/s', 'ms_per_step': round(dts / args.steps * 1e3, 3),
                          'what': 'FIXED global batch of %d images split over %d rank(s) (eval.py:630-634), one gather of the records; the '
                                  'driver divides the N-GPU value by the 1-GPU value of the SAME key for strong-scaling efficiency' % (G, world)}

        result = None
        if rank == 0:
            if got['n'] != args.batch * world:
                raise SystemExit('bench.py: rank 0 gathered %d records, expected %d' % (got['n'], args.batch * world))
            plan = net.plan_for(x)
            rk, layers = roofline(net, x)
            rf = whole_path_roofline(rk, dt / args.steps * 1e3, args.batch)
            imgs = args.batch * world * args.steps
            is_headline = args.config == CONFIG and size == 550 and args.batch == 8
            result = {
                'metric': 'images/sec (%dx%d, batch %d per GPU), YOLACT %s forward + Detect (Fast NMS)'
                          % (size, size, args.batch, args.config.replace('_config', '')),
                'value': round(imgs / dt, 2), 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps,
                'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True,
                'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
                'config': {'workload': '%s%s, %dx%d, batch %d per GPU, random-init '
                                       'weights (no checkpoint offline), inputs resident in HBM'
                                       % ('configs[1]: ' if is_headline else '', args.config, size, size, args.batch),
                           'global_batch': args.batch * world, 'parallelism': 'dp%d' % world,
                           'postprocess_in_step': bool(args.with_postprocess),
                           'step_overlap': ('%d: consecutive batches rotate over %d plan instances on %d HIP streams (yolact_amd.pipeline.BatchPipeline, %s)'
                                            % (args.step_overlap, args.step_overlap, args.step_overlap,
                                               'every plan forks its side stream' if pipeline.fork else 'one stream per plan') if args.step_overlap > 1
                                            else '1: every step is issued behind the previous one on one stream'),
                           'host_read': ('blocking, every step' if args.no_pipeline else
                                         'every step, asynchronous D2H copy collected after the next step(s) are launched (depth %d)' % max(2, args.step_overlap)),
                           'plan': {'source': 'shipped tune table yolact_amd/tune/gfx950.json' if plan.tune_misses == 0
                                    else 'tune table + %d shapes measured in this process' % plan.tune_misses,
                                    'tune_misses': plan.tune_misses}},
                'rccl_ranks': dist.get_world_size() if have_pg else 0,
                'gathered_records': got['n'],
                'collective': ('dist.gather over the nccl (RCCL) backend, %d rank(s)' % dist.get_world_size()) if have_pg
                              else 'none (RCCL init failed: %s)' % rccl_error,
                # host side of the timed region: how long the Python launch loop needs to ISSUE one step (median / max over the timed
                # steps).  Far below ms_per_step = the GPU sets the pace; close to it = this box's host does (box_calibration.host)
                'host_issue_ms_per_step': {'median': round(sorted(issue_s)[len(issue_s) // 2] * 1e3, 3), 'max': round(max(issue_s) * 1e3, 3)},
                'per_rank_ms_per_step': per_rank_ms,            # each rank's own clock over the timed region (value uses the MAX)
                'gather_us': gather_us,                         # the record gather alone, host-paired, 50 back-to-back calls
                'record_bytes_per_image': 4 * int(parallel.pack_records(net.forward_device(x)).shape[1]),
                'roofline': rf,
                # both scaling modes in one line: `value` / `scaling` above = weak (--batch images per GPU);
                # strong_scaling = a fixed global batch split over the ranks
                'strong_scaling': strong,
            }
            if rf['engine']['h2_share_of_time'] > 0:
                result['dtype'] = ('f32 (fp32 in / fp32 accumulate / fp32 out; %.0f %% of the GEMM time on fp16x2 tiles = every fp32 '
                                   'operand as two fp16 pieces by round to nearest (22 significant bits + 2 signs: exact for ~2/3 of '
                                   'all fp32 values, one fp32 ulp otherwise), 3 exact piece products per product on the fp16 matrix '
                                   'pipe; measured against fp64 the error is below the exact-fp32 MFMA kernel\'s own (fp32 accumulation '
                                   'dominates both); the rest on exact-fp32 MFMA)' % (100 * rf['engine']['h2_share_of_time']))
            elif rf['engine']['x3_share_of_time'] > 0:
                result['dtype'] = ('f32 (fp32 in / fp32 accumulate; %.0f %% of the GEMM time on bf16x3 tiles = every fp32 product as 6 '
                                   'exact bf16 piece products on the bf16 matrix pipe, error class of one fp32 rounding; the rest '
                                   'on exact-fp32 MFMA)' % (100 * rf['engine']['x3_share_of_time']))
            result['roofline']['all_conv']['sustained_tflops_in_timed_region'] = round(
                rf['all_conv']['gflop_per_step'] / (dt / args.steps * 1e3), 2)
            if calib is not None:
                result['box_calibration'] = calib
                # the same value had this box delivered 2000 TFLOP/s on the calibration loop (a typical MI355X under fp16 MFMA
                # load): comparable across boxes for the matrix-bound share of the step, NOT a substitute for `value`
                calib['value_if_box_delivered_2000_tflops'] = round(result['value'] * 2000.0 / max(calib['mfma_f16_tflops'], 1.0), 1)
            if args.layers:
                for name, best, times in getattr(plan, 'tune_table', []):
                    print('tune %-20s -> %-8s %s' % (name, best, times), file=sys.stderr)
                for name, best, t_dir, t_win, t_f2, t_f4 in getattr(plan, 'wino_table', []):
                    print('wino %-20s -> %-14s direct %.4f ms  F(2x2) %.4f  F(4x4) %.4f%s' % (
                        name, best, t_dir, t_f2, t_f4, '' if t_win < 0.97 * t_dir else '   (kept direct)'), file=sys.stderr)
                for k, (ms, fl, kern) in layers.items():
                    print('%-22s %8.3f ms %8.2f GFLOP %7.1f TF/s  %s' % (k, ms, fl / 1e9, fl / ms / 1e9, kern),
                          file=sys.stderr)
            if world == 1 and not args.no_secondary:
                result['secondary'] = secondary_lines(net, x, size, config=args.config)
                if is_headline:
                    result['secondary']['outlier_plan'] = outlier_plan_line(dev, x, size, args.config)
                    result['secondary']['outlier_plan_guard_only'] = outlier_plan_line(dev, x, size, args.config, rebalance=False)
        if rank == 0 and world == 1 and not args.no_cpu_baseline and args.config == CONFIG:
            result['cpu_baseline'] = cpu_baseline(sd, size, args.batch)
    if have_pg:
        dist.barrier()
        dist.destroy_process_group()
    sys.stdout.flush()
    if rank == 0 and result is not None:
        os.write(real_stdout, (json.dumps(result) + '\n').encode())
    os.close(real_stdout)


if __name__ == '__main__':
    main()
