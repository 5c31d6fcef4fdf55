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

One JSON line on rank 0, with `roofline` (dominant conv kernel, live HIP-event timing on the launch stream) and,
at N=1, `cpu_baseline` (the CPU oracle = port of the reference's path, timed on this host's cores).

`roofline` fields: `kernel` / `achieved` / `frac` / `avg_launch_ms` / `traffic` = the instantiation of the conv engine
with the largest summed duration; a direct launch is priced with its algorithmic conv FLOPs, a grouped Winograd GEMM
launch with the FLOPs it executes (`flops_basis`), so `frac` is a matrix-core utilisation.  `engine` = the same over ALL
GEMM launches of a step.  `all_conv` = ALGORITHMIC conv FLOPs (118.28 GFLOP/image) over the summed
durations of every conv-layer launch incl. the Winograd transforms — the figure BASELINE.json's ">= 60 % of the conv
roofline" target refers to; with Winograd layers it can exceed what the matrix cores execute.  `per_kernel` = every
instantiation.  DESIGN.md 3.5 / 6.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CU x 2.4 GHz (spec)
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
    return net.to(device), sd


def roofline(net, x, reps=3):
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
    # record kinds: 0/1/2 = one direct conv launch (loader id); 3 / 4 = a whole Winograd F(2x2) / F(4x4) layer (input
    # transform + 16- / 36-group GEMM + output transform, ALGORITHMIC conv FLOPs); 5 / 6 = the Winograd GEMM launch alone
    # (the FLOPs it executes).  Layer table / all_conv: kinds 0-4.  Single-kernel roofline: kinds 0-2, 5 and 6.
    by_kernel, layers = {}, {}
    tot_ms = tot_fl = 0.0
    li = -1
    nl = len(names)
    for i in range(n):
        L.check(lib.ymi_prof_read(i, C.byref(ms), C.byref(fl), C.byref(tile), C.byref(kind)))
        tname = L.TILE_NAMES.get(tile.value, '?')
        if kind.value not in (5, 6):
            li += 1
            lkey = ('winograd F(%dx%d,3x3) <gemm %s> (3 launches)' % (2 * kind.value - 4, 2 * kind.value - 4, tname)) \
                if kind.value in (3, 4) else 'conv_igemm_f32<%s,loader%d>' % (tname, kind.value)
            la = layers.setdefault(names[li % nl], [0.0, fl.value, lkey])
            la[0] += ms.value / reps
            la[2] = lkey
            tot_ms += ms.value; tot_fl += fl.value
        if kind.value not in (3, 4):
            key = ('conv_igemm_f32<%s,winograd grouped GEMM>' % tname) if kind.value in (5, 6) else \
                'conv_igemm_f32<%s,loader%d>' % (tname, kind.value)
            a = by_kernel.setdefault(key, [0.0, 0.0, 0])
            a[0] += ms.value; a[1] += fl.value; a[2] += 1
    lib.ymi_prof_reset()
    dom = max(by_kernel.items(), key=lambda kv: kv[1][0])
    name, (dms, dfl, dn) = dom
    ach = dfl / (dms * 1e-3) / 1e12
    detail = {k: {'ms_per_step': v[0] / reps, 'tflops': v[1] / (v[0] * 1e-3) / 1e12, 'launches_per_step': v[2] // reps}
              for k, v in by_kernel.items()}
    wino2 = sum(1 for v in layers.values() if v[2].startswith('winograd F(2x2'))
    wino4 = sum(1 for v in layers.values() if v[2].startswith('winograd F(4x4'))
    return {
        'bound': 'mfma', 'kernel': name, 'achieved': round(ach, 2), 'peak': FP32_MFMA_PEAK_TFLOPS,
        'unit': 'TFLOP/s', 'frac': round(ach / FP32_MFMA_PEAK_TFLOPS, 4), 'traffic': traffic_from_profiles(name),
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
        'engine': {'kernel': 'conv_igemm_f32<*> (every instantiation: direct loaders + grouped Winograd GEMM)',
                   'ms_per_step': round(sum(v[0] for v in by_kernel.values()) / reps, 3),
                   'executed_tflops': round(sum(v[1] for v in by_kernel.values()) / (sum(v[0] for v in by_kernel.values()) * 1e-3) / 1e12, 2),
                   'frac': round(sum(v[1] for v in by_kernel.values()) / (sum(v[0] for v in by_kernel.values()) * 1e-3) / 1e12
                                 / FP32_MFMA_PEAK_TFLOPS, 4),
                   'basis': 'FLOPs executed on the matrix cores by all GEMM launches of a step / their summed durations'},
        'per_kernel': detail,
    }, layers


def traffic_from_profiles(kernel):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE in separate runs, FETCH_SIZE doubled per MI355X_MICROARCH.md; tools/gpu_session_traffic.sh writes
    profiles/r01_traffic.json).  null when no PMC record exists for that kernel."""
    path = os.path.join(ROOT, 'profiles', 'r01_traffic.json')
    if not os.path.exists(path):
        return None
    with open(path) as f:
        rec = json.load(f).get(kernel)
    return rec['bytes_per_launch'] if rec else None


def cpu_baseline(sd, size, budget_s=20.0):
    """The CPU oracle (port of the reference's forward + Detect) on this host's cores, bounded sample."""
    import yolact_amd
    from oracle import yolact_oracle as O
    from yolact_amd.utils.synth import synth_images
    cfg = yolact_amd.CONFIGS[CONFIG].copy()
    threads = torch.get_num_threads()
    x = synth_images(2, size, size, seed=4321)
    with torch.no_grad():
        O.detect(O.forward_raw(x, sd, cfg), cfg)            # warm-up
        t0 = time.perf_counter()
        n = 0
        while True:
            O.detect(O.forward_raw(x, sd, cfg), cfg)
            n += x.shape[0]
            dt = time.perf_counter() - t0
            if dt > budget_s or n >= 64:
                break
    return {'value': round(n / dt, 3), 'unit': 'images/s', 'cores': threads, 'kind': 'port',
            'sample': '%d images (batches of 2, %dx%d) forward+Detect through oracle/yolact_oracle.py, %.1f s, '
                      'torch %s CPU fp32, os.cpu_count()=%s' % (n, size, size, dt, torch.__version__, os.cpu_count())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=8, help='images per GPU')
    ap.add_argument('--size', type=int, default=550)
    ap.add_argument('--config', default=CONFIG, help='other BASELINE configs (parity-test cases; the metric is quoted on '
                    'the default yolact_resnet50_config)')
    ap.add_argument('--with-postprocess', action='store_true')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--layers', action='store_true', help='also print the per-layer conv table to stderr')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if args.gpus != world and world > 1:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))   # RCCL over xGMI
    dev = torch.device('cuda', local_rank)
    torch.cuda.set_device(dev)

    from yolact_amd import parallel
    from yolact_amd.layers.output_utils import postprocess
    from yolact_amd.utils.synth import synth_images
    with torch.no_grad():
        net, sd = build_model(dev, args.size, args.config)
        x = synth_images(args.batch, args.size, args.size, seed=1234 + rank).to(dev)   # resident in HBM

        def step():
            out = net.forward_device(x)
            rec = parallel.gather_records(parallel.pack_records(out), dst=0)
            if rec is not None:
                counts = rec[:, 0].tolist()                         # host read of the per-image counts (rank 0)
            if args.with_postprocess:
                n_local = out['count'].tolist()
                for b, n in enumerate(n_local):
                    if n:
                        det = {'box': out['box'][b, :n], 'mask': out['coef'][b, :n], 'class': out['cls'][b, :n],
                               'score': out['score'][b, :n], 'proto': out['proto'][b]}
                        postprocess([{'detection': det, 'net': net}], args.size, args.size)
            return rec

        for _ in range(args.warmup):
            step()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())

        result = None
        if rank == 0:
            rf, layers = roofline(net, x)
            imgs = args.batch * world * args.steps
            result = {
                'metric': 'images/sec (550x550, batch 8 per GPU), YOLACT ResNet50-FPN forward + Detect (Fast NMS)',
                'value': round(imgs / dt, 2), 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps,
                'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True,
                'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
                'config': {'workload': '%s%s, %dx%d, batch %d per GPU, random-init '
                                       'weights (no checkpoint offline), inputs resident in HBM'
                                       % ('configs[1]: ' if args.config == CONFIG else '', args.config, args.size, args.size,
                                          args.batch),
                           'global_batch': args.batch * world, 'parallelism': 'dp%d' % world,
                           'postprocess_in_step': bool(args.with_postprocess)},
                'roofline': rf,
            }
            result['roofline']['all_conv']['sustained_tflops_in_timed_region'] = round(
                rf['all_conv']['gflop_per_step'] / (dt / args.steps * 1e3), 2)
            if args.layers:
                for name, best, times in getattr(net.plan_for(x), 'tune_table', []):
                    print('tune %-20s -> %-8s %s' % (name, best, times), file=sys.stderr)
                for name, best, t_dir, t_win, t_f2, t_f4 in getattr(net.plan_for(x), 'wino_table', []):
                    print('wino %-20s -> %-14s direct %.4f ms  F(2x2) %.4f  F(4x4) %.4f%s' % (
                        name, best, t_dir, t_f2, t_f4, '' if t_win < 0.97 * t_dir else '   (kept direct)'), file=sys.stderr)
                for k, (ms, fl, kern) in layers.items():
                    print('%-22s %8.3f ms %8.2f GFLOP %7.1f TF/s  %s' % (k, ms, fl / 1e9, fl / ms / 1e9, kern),
                          file=sys.stderr)
        if rank == 0 and world == 1 and not args.no_cpu_baseline and args.config == CONFIG:
            result['cpu_baseline'] = cpu_baseline(sd, args.size)
        if rank == 0:
            print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
