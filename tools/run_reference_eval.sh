#!/bin/bash
# On the GPU box (stage `evalpy` of tools/gpu_session.sh): the reference's own evaluator, twice.
#   1. reference Yolact on the MI355X through PyTorch-ROCm/MIOpen  -> pseudo-GT, its mAP table, its --benchmark FPS
#   2. the engine bound through shim/                              -> the same evaluator, same dataset: mAP, FPS, COCO json
# $1 = output dir.  Needs _scratch_reference/ (tools/stage_reference.sh).
O=$1; N=${2:-24}
export GPU_MAX_HW_QUEUES=8
PY="python tools/run_reference_eval.py --out $O --images $N"
export MIOPEN_FIND_MODE=FAST MIOPEN_USER_DB_PATH=/tmp/miopen_db MIOPEN_CUSTOM_CACHE_DIR=/tmp/miopen_cache
timeout 420 $PY --who reference --mode gt || { echo "reference on GPU failed or too slow: falling back to the reference on the host CPU"; REFCPU="--cuda 0"; timeout 600 $PY --who reference --mode gt $REFCPU; }
timeout 600 $PY --who reference --mode map $REFCPU
timeout 600 $PY --who reference --mode benchmark $REFCPU
$PY --who engine --mode map
$PY --who engine --mode benchmark
$PY --who engine --mode coco
python - $O <<'PY'
import json, sys
o = sys.argv[1]
r, e = (json.load(open('%s/map_%s.json' % (o, w))) for w in ('reference', 'engine'))
print('mAP table, reference vs engine (the reference\'s own prep_metrics / calc_map):')
worst = 0.0
for t in ('box', 'mask'):
    for k in r['map'][t]:
        worst = max(worst, abs(r['map'][t][k] - e['map'][t][k]))
    print('  %-4s reference %s' % (t, ' '.join('%6.2f' % v for v in r['map'][t].values())))
    print('  %-4s engine    %s' % (t, ' '.join('%6.2f' % v for v in e['map'][t].values())))
rb, eb = (json.load(open('%s/benchmark_%s.json' % (o, w))) for w in ('reference', 'engine'))
print('largest |delta| over the table: %.3f points' % worst)
print('eval.py --benchmark: reference (%s) %s | engine %s' % (rb['device'], rb.get('eval_py_average_line'), eb.get('eval_py_average_line')))
json.dump({'max_abs_map_delta': worst, 'reference_fps': rb.get('fps'), 'engine_fps': eb.get('fps'),
           'reference_device': rb['device'], 'images': r['images']}, open('%s/summary.json' % o, 'w'), indent=1)
PY
