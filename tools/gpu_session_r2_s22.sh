#!/bin/bash
# Round 2, session 22: fused RLE kernel with cached source rows: its tests + the A/B probe of session 21.
O=gpurun_out/r2s22; mkdir -p $O
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_coco_rle.py -m gpu -q --timeout 300 > $O/pytest.log 2>&1; tail -2 $O/pytest.log
sed -n '/^python - <<.PY./,/^PY$/p' tools/gpu_session_r2_s21.sh | sed '1d;$d' > /tmp/probe.py
python /tmp/probe.py > $O/rle_fused_probe.json 2> $O/probe.err; cat $O/rle_fused_probe.json
