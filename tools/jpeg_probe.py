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


def photo_sized_jpeg():
    """641x427 4:2:0 baseline file with restart markers, written by Pillow (no oracle / test code in a measurement tool)."""
    import io
    import numpy as np
    from PIL import Image
    rng = np.random.default_rng(8)
    h, w = 427, 641
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([128 + 100 * np.sin(xx / 17.0 + yy / 23.0), 128 + 90 * np.cos(xx / 11.0) * np.sin(yy / 19.0),
                    (xx + yy) * 255.0 / (w + h)], -1) + rng.normal(0, 10, (h, w, 3))
    buf = io.BytesIO()
    Image.fromarray(np.clip(img, 0, 255).astype(np.uint8)).save(buf, 'JPEG', quality=85, subsampling=2, restart_marker_blocks=7)
    return buf.getvalue()


def main():
    import yolact_amd
    from yolact_amd.data import jpeg
    from yolact_amd.utils.augmentations import BaseTransform
    yolact_amd.set_cfg('yolact_resnet50_config')
    data = photo_sized_jpeg()
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
