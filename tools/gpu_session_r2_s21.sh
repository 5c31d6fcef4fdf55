#!/bin/bash
# Round 2, session 21: fused upsample + threshold + RLE kernel (ymi_mask_rle_upsampled_f32), postprocess_rle; the mask
# kernels were recompiled against the shared upsample_math.h, so the postprocess / mAP / display tests run too.
O=gpurun_out/r2s21; mkdir -p $O
export TMPDIR=/tmp
timeout 800 python -m pytest tests/test_coco_rle.py tests/test_gpu_path.py tests/test_map_parity.py tests/test_prep_display.py tests/test_mask_iou.py -m gpu -q --timeout 500 -rA --durations=5 > $O/pytest.log 2>&1; grep -E "passed|failed|s call" $O/pytest.log | tail -8; grep -E "^FAILED|^ERROR" $O/pytest.log | head
python - <<'PY' > $O/rle_fused_probe.json 2> $O/probe.err
import json, time, ctypes as C, torch, sys
sys.path.insert(0, '.')
from yolact_amd import _lib as L
from yolact_amd.coco import rle_encode, rle_encode_lowres
torch.manual_seed(0)
n, ph, pw, h, w = 100, 138, 138, 550, 550
lo = torch.sigmoid(8 * (torch.nn.functional.avg_pool2d(torch.rand(n, 1, ph, pw), 9, 1, 4)[:, 0] - 0.5)).cuda()
full = torch.empty(n, h, w, device='cuda')
def two_step():
    L.check(L.lib().ymi_mask_upsample_f32(lo.data_ptr(), full.data_ptr(), n, ph, pw, h, w, C.c_float(0.5), L.stream_ptr()))
    return rle_encode(full)
def fused():
    return rle_encode_lowres(lo, h, w, 0.5)
assert two_step() == fused()
res = {}
for name, fn in (('upsample_then_rle_ms', two_step), ('fused_ms', fused)):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): fn()
    torch.cuda.synchronize(); res[name] = round((time.perf_counter() - t0) / 20 * 1e3, 3)
res['what'] = '100 masks 138x138 -> 550x550, incl. the string kernel and the host copies of the strings'
print(json.dumps(res))
PY
cat $O/rle_fused_probe.json; tail -2 $O/probe.err | cut -c1-200
