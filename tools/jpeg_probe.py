#!/usr/bin/env python
"""Throughput of the pull_item image read: host entropy decode, H2D copy + device reconstruction, and the whole
pull_item-style chain (imread -> BaseTransform) on a photo-sized 4:2:0 file (641x427, restart interval 7)."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    import yolact_amd
    from tests.test_gpu_jpeg import _big_jpeg
    from yolact_amd.data import jpeg
    from yolact_amd.utils.augmentations import BaseTransform
    yolact_amd.set_cfg('yolact_resnet50_config')
    data = _big_jpeg()
    res = {'file_bytes': len(data), 'image': '641x427 4:2:0'}
    n = 200
    t0 = time.perf_counter()
    for _ in range(n):
        jpeg.decode_coefficients(data)
    res['host_entropy_decode_ms'] = round((time.perf_counter() - t0) / n * 1e3, 3)
    tr = BaseTransform()
    for fn, key in ((lambda: jpeg.imread(data), 'imread_ms'), (lambda: tr(jpeg.imread(data))[0], 'imread_plus_base_transform_ms')):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        res[key] = round((time.perf_counter() - t0) / n * 1e3, 3)
    res['images_per_s_one_host_thread'] = round(1e3 / res['imread_plus_base_transform_ms'], 1)
    print(json.dumps(res))


if __name__ == '__main__':
    main()
