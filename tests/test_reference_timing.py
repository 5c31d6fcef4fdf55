"""bench.py's `cpu_baseline` is `kind: "port"` (the GPU box has no /root/reference): this test shows, where the reference
exists, that the port is a fair stand-in — the reference's own Yolact.forward + Detect and oracle/yolact_oracle.py are timed
side by side on the same tensors and thread count and must be within 2x of each other (measured: within ~20 %).  Runs in a
subprocess (the reference's module names are global).  Build container only."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import json, sys, time
ROOT = sys.argv[1]
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/oracle')
import torch
from make_golden import _shim_reference
_shim_reference()
from data import cfg, set_cfg
set_cfg('yolact_resnet50_config'); cfg.mask_proto_debug = False
from yolact import Yolact
from yolact_amd.utils.synth import synth_state_dict, synth_images
import yolact_amd
from oracle import yolact_oracle as O
torch.set_num_threads(min(8, torch.get_num_threads()))
net = Yolact(); net.eval(); net.detect.use_fast_nms = True
sd = synth_state_dict([(k, tuple(v.shape)) for k, v in net.state_dict().items()], seed=0, conf_gain=0.04)
net.load_state_dict(sd)
x = synth_images(2, 550, 550, seed=1234)
ocfg = yolact_amd.CONFIGS['yolact_resnet50_config'].copy()
def best(fn, n=3):
    fn(); ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return min(ts)
with torch.no_grad():
    t_ref = best(lambda: net(x))
    t_port = best(lambda: O.detect(O.forward_raw(x, sd, ocfg), ocfg))
print(json.dumps({'reference_images_per_s': 2 / t_ref, 'port_images_per_s': 2 / t_port, 'threads': torch.get_num_threads()}))
'''


@pytest.mark.skipif(not os.path.isdir('/root/reference'), reason='/root/reference is not present on this machine')
def test_port_is_as_fast_as_the_reference_on_cpu():
    out = subprocess.run([sys.executable, '-c', SCRIPT, ROOT], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads(out.stdout.strip().splitlines()[-1])
    print('reference %.2f images/s, port %.2f images/s on %d threads' % (r['reference_images_per_s'], r['port_images_per_s'],
                                                                         r['threads']))
    ratio = r['port_images_per_s'] / r['reference_images_per_s']
    assert 0.5 < ratio < 2.0, r
