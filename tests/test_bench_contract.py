"""The bench.py output contract, checked on the committed result of the last GPU session (profiles/r06_bench.json): the
keys the driver and the judge read, their types, and the internal consistency of the roofline and calibration blocks."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load():
    with open(os.path.join(ROOT, 'profiles', 'r06_bench.json')) as f:
        return json.loads(f.read())


def test_top_level_fields():
    b = _load()
    for k, typ in (('metric', str), ('value', float), ('unit', str), ('n_gpus', int), ('steps', int), ('warmup', int),
                   ('ms_per_step', float), ('higher_is_better', bool), ('scaling', str), ('dtype', str), ('data', str),
                   ('config', dict), ('roofline', dict), ('cpu_baseline', dict)):
        assert isinstance(b[k], typ), (k, type(b[k]))
    assert b['vs_baseline'] is None                      # BASELINE.md publishes no number for this metric on MI355X
    assert b['unit'] == 'images/s' and b['higher_is_better'] is True and b['scaling'] == 'weak' and b['dtype'].startswith('f32')
    assert b['data'] == 'synthetic' and 'workload' in b['config'] and 'configs[1]' in b['config']['workload']
    assert 'model' not in b['config']
    # value = images of the whole job / wall time of the timed region
    imgs = b['config']['global_batch'] * b['steps']
    assert abs(b['value'] - imgs / (b['ms_per_step'] * b['steps'] / 1e3)) / b['value'] < 1e-3
    # the exchange step really ran on RCCL and every image's record arrived
    assert b['rccl_ranks'] == b['n_gpus'] and b['gathered_records'] == b['config']['global_batch']
    assert b['config']['plan']['tune_misses'] == 0           # the timed plan is the shipped (deterministic) one
    # round 6: both scaling modes in one line — `value` = weak (batch per GPU, consecutive batches overlapped on four plan instances),
    # strong_scaling = a fixed global batch split over the ranks on ONE plan / stream (= the serial figure at one GPU)
    ss = b['strong_scaling']
    assert ss['global_batch'] == 8 and sum(ss['images_per_rank']) == 8 and ss['unit'] == 'images/s' and ss['value'] > 0
    assert b['config']['step_overlap'].startswith('4:') and ss['value'] < b['value']
    sec = b['secondary']
    assert sec['outlier_plan']['layers_on_bf16x3'] == 0 and sec['outlier_plan']['rebalanced'] and sec['outlier_plan_guard_only']['layers_on_bf16x3'] >= 10
    assert sec['outlier_plan']['value'] > sec['outlier_plan_guard_only']['value']
    assert sec['batch_with_postprocess']['value'] > 0 and sec['reference_fps_definition_batch1']['value'] > 0
    assert sec['sparse_regime']['value'] > 0 and 50 < sec['sparse_regime']['candidate_priors_per_image'] < 600


def test_roofline_block():
    b = _load()
    r = b['roofline']
    # top level (VERDICT r4 #1b): SURVEY 8(d)'s whole-path figure — algorithmic conv GFLOP of a step / the TIMED step, against the
    # matrix-pipe peak of the precision used
    assert r['bound'] == 'mfma' and r['unit'] == 'TFLOP/s' and r['peak'] in (157.3, 416.7, 833.3)
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3 and 0 < r['frac'] <= 1.0
    ac = r['all_conv']
    assert abs(ac['gflop_per_step'] - 8 * 118.28) < 1.0  # 118.28 GFLOP per image
    assert abs(r['achieved'] - ac['gflop_per_step'] / b['ms_per_step']) / r['achieved'] < 2e-3
    assert abs(r['achieved'] - 118.28 * b['value'] / 1e3) / r['achieved'] < 2e-3        # = GFLOP per image x images/s (one GPU)
    assert r['traffic'] is None or r['traffic'] > 0
    assert isinstance(r['traffic_source'], str) and ('static' in r['traffic_source'] or r['traffic'] is None)
    # the dominant kernel (largest summed duration) on ITS matrix-pipe fraction; its algorithmic-bytes view is a sub-key
    d = r['dominant_kernel']
    assert d['kernel'] == r['kernel'] and d['bound'] == 'mfma' and d['unit'] == 'TFLOP/s'
    assert d['peak'] in (157.3, 416.7, 833.3) and (('h2' in d['kernel']) or ('dcnp' in d['kernel'])) == (d['peak'] == 833.3)
    assert abs(d['frac'] - d['achieved'] / d['peak']) < 1e-3 and 0 < d['frac'] <= 1.0
    assert abs(d['achieved'] - d['flops_per_launch'] / (d['avg_launch_ms'] * 1e-3) / 1e12) / d['achieved'] < 0.01
    hb = d['hbm_view']
    assert hb['peak'] == 8000.0 and abs(hb['frac'] - hb['achieved'] / 8000.0) < 1e-3
    assert abs(hb['achieved'] - hb['alg_bytes_per_launch'] / (d['avg_launch_ms'] * 1e-3) / 1e9) / hb['achieved'] < 0.01
    # step-level bound: sum over launches of max(FLOPs / tile peak, algorithmic bytes / 8 TB/s) <= the measured serialised time
    bs = r['bound_sum']
    assert abs(r['bound_sum_ms'] - (bs['mfma_bound_launches_ms'] + bs['hbm_bound_launches_ms'])) < 2e-3
    assert 0 < r['bound_sum_ms'] < bs['measured_ms'] and abs(bs['frac'] - r['bound_sum_ms'] / bs['measured_ms']) < 2e-3
    assert r['kernel'] in r['per_kernel']
    dom = max(r['per_kernel'].items(), key=lambda kv: kv[1]['ms_per_step'])[0]
    assert dom == r['kernel']                            # "dominant" = largest summed duration
    assert abs(ac['tflops'] - ac['gflop_per_step'] / ac['ms_per_step']) < 0.5
    assert 0 < r['engine']['frac'] <= 1.0
    for k, v in r['per_kernel'].items():
        assert 0 < v['frac'] <= 1.0 and abs(v['frac'] - v['tflops'] / v['peak']) < 2e-3, k
        assert v['bound'] in ('hbm', 'mfma') and 0 < v['bound_frac'] <= 1.0


def test_box_calibration_block():
    """VERDICT r4 #1a: a fixed fp16-MFMA loop and an HBM copy timed right before the timed region, next to what the box reports about
    its clocks — so that two runs can be told apart as 'the box' or 'the code'."""
    b = _load()
    c = b['box_calibration']
    assert 1000.0 < c['mfma_f16_tflops'] <= 2600.0 and abs(c['mfma_f16_frac_of_2500'] - c['mfma_f16_tflops'] / 2500.0) < 1e-3
    assert 2000.0 < c['hbm_copy_GBps'] <= 8000.0 and abs(c['hbm_copy_frac_of_8000'] - c['hbm_copy_GBps'] / 8000.0) < 1e-3
    assert c['compute_units'] == 256 and isinstance(c['power_state'], dict) and 'source' in c['power_state']
    assert abs(c['value_if_box_delivered_2000_tflops'] - b['value'] * 2000.0 / c['mfma_f16_tflops']) < 0.5
    assert b['per_rank_ms_per_step'] is not None and len(b['per_rank_ms_per_step']) == b['n_gpus']
    assert b['gather_us'] > 0 and b['record_bytes_per_image'] == 4 * (1 + 100 * 38)
    brk = b['secondary']['reference_fps_definition_batch1']['breakdown_ms']
    assert brk['bytes_copied'] >= 5 * 550 * 550 * 4 and all(brk[k] > 0 for k in brk)


def test_cpu_baseline_block():
    import statistics
    c = _load()['cpu_baseline']
    assert c['kind'] in ('port', 'reference') and c['unit'] == 'images/s' and c['value'] > 0 and c['cores'] >= 1
    assert c['threads'] == c['cores'] <= c['cpus_in_affinity_mask'] <= c['host_logical_cpus']
    sweep, runs = c['thread_sweep_images_per_s'], c['timed_runs_images_per_s']
    # the thread count is swept ON the timed workload, the two best settings are timed three batches each, the better median is reported
    top2 = sorted(sweep, key=sweep.get, reverse=True)[:2]
    assert sorted(runs) == sorted(top2) and all(len(v) == 3 for v in runs.values())
    assert str(c['cores']) in runs and c['value'] == round(statistics.median(runs[str(c['cores'])]), 3)
    assert c['value'] == max(round(statistics.median(v), 3) for v in runs.values())
    assert isinstance(c['sample'], str) and 'median of 3 batches of 8' in c['sample']


def test_layer_view_block():
    """roofline.layer_view (VERDICT r3 #2e): the Winograd layers as layers — algorithmic FLOPs over all three launches, the bytes
    the convolution needs next to the Winograd-domain bytes the launches move."""
    v = _load()['roofline']['layer_view']
    assert v['winograd_layers'] >= 10 and v['ms_per_step'] > 0
    assert abs(v['domain_over_conv_bytes'] - v['winograd_domain_MB_per_step'] / v['conv_alg_MB_per_step']) < 0.02
    assert v['winograd_domain_MB_per_step'] > v['gemm_launch_MB_per_step'] > v['conv_alg_MB_per_step']
    assert 0 < v['frac_of_fp16x2_peak'] < 1 and 0 < v['frac_of_layer_bound'] < 1
    assert abs(v['frac_of_layer_bound'] - v['bound_ms_at_peaks'] / v['ms_per_step']) < 2e-3


def test_cli_defaults():
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--help'], capture_output=True, text=True,
                         timeout=120)
    assert out.returncode == 0
    for flag in ('--gpus', '--steps', '--warmup'):
        assert flag in out.stdout


def test_refuses_more_gpus_than_the_box_has():
    """`python bench.py --gpus N` must never print an N-GPU line measured on fewer devices (round-1 verdict, Weak 4):
    this container has no GPU, so any N is refused with a non-zero exit and a message that says why."""
    for n in ('1', '2', '8'):
        out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', n], capture_output=True, text=True,
                             timeout=300)
        assert out.returncode != 0
        assert 'visible GPU' in (out.stderr + out.stdout) and '--gpus %s' % n in (out.stderr + out.stdout)
        assert '"value"' not in out.stdout
    env = dict(os.environ, WORLD_SIZE='4', RANK='0')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '0'], capture_output=True, text=True,
                         timeout=300, env=env)
    assert out.returncode != 0


def test_per_kernel_pass_takes_the_median_over_passes():
    """A host thread descheduled between a start event and its launch shows up as a 26 ms "kernel" in one pass (session r5fz): the
    per-record duration is the median over the serialised passes, applied only when the passes are the same sequence."""
    sys.path.insert(0, ROOT)
    import bench
    one = [[0.05, 1e9, 3, 10], [0.10, 2e9, 5, 7], [0.02, 5e8, 3, 10]]
    recs = [list(r) for _ in range(5) for r in one]
    recs[3 + 1][0] = 26.0                                   # pass 1, record 1: the hiccup
    recs[9 + 2][0] = 0.021
    assert bench.median_over_passes(recs, 5) is True
    assert [r[0] for r in recs[:3]] == [0.05, 0.10, 0.02] and all(recs[3 * p + 1][0] == 0.10 for p in range(5))
    assert abs(sum(r[0] for r in recs) / 5 - sum(r[0] for r in recs[:3])) < 1e-12          # what the caller's "/ reps" sums to
    ragged = [list(r) for r in one] + [[0.05, 1e9, 3, 10]]
    before = [r[0] for r in ragged]
    assert bench.median_over_passes(ragged, 2) is False and [r[0] for r in ragged] == before      # not the same sequence: untouched
    other = [list(r) for r in one] + [[0.05, 1e9, 3, 10], [0.10, 2e9, 5, 9], [0.02, 5e8, 3, 10]]
    assert bench.median_over_passes(other, 2) is False
    assert bench.median_over_passes([], 5) is False
