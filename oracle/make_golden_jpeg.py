#!/usr/bin/env python
"""Golden vectors for the JPEG path: file bytes + the pixels libjpeg-turbo decodes them to.  *** TEST INFRASTRUCTURE ***

    python oracle/make_golden_jpeg.py            # needs Pillow (libjpeg-turbo inside); writes tests/golden/jpeg.npz

cv2.imread (data/coco.py:138) wraps libjpeg-turbo with the library defaults; OpenCV is not installed here, Pillow is and
wraps the same library with the same defaults (Pillow %s / libjpeg-turbo %s when this file was last run), so Pillow's
decode (RGB -> BGR, + ImageOps.exif_transpose for the orientation cv2 applies) is the reference output.  Cases: Pillow-written
4:4:4 / 4:2:2 / 4:2:0 at odd sizes, baseline / optimised / progressive / restart intervals / grayscale / EXIF orientations
1-8, and files written by oracle/jpeg_encode.py with the sampling ratios Pillow cannot write (4:4:0, 4:1:1, 4:1:0, mixed
chroma factors) to reach libjpeg's other upsampling paths.
"""
import io
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def synth(rng, h, w):
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([128 + 100 * np.sin(xx / 7.0 + yy / 13.0), 128 + 90 * np.cos(xx / 5.0) * np.sin(yy / 9.0),
                    xx * 255.0 / max(w - 1, 1)], -1)
    img += rng.normal(0, 12, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


def main():
    from PIL import Image, ImageOps, features
    from oracle import jpeg_encode as E
    rng = np.random.default_rng(20260923)
    cases = {}

    def add(name, data):
        im = Image.open(io.BytesIO(data))
        im = ImageOps.exif_transpose(im)
        ref = np.ascontiguousarray(np.array(im.convert('RGB'))[..., ::-1])
        cases[name] = (np.frombuffer(data, dtype=np.uint8).copy(), ref)

    for (h, w) in [(33, 47), (64, 64), (17, 3), (8, 2), (1, 1), (50, 35)]:
        a = synth(rng, h, w)
        for ss, sn in [(0, '444'), (1, '422'), (2, '420')]:
            for kw, kn in [({}, 'base'), ({'progressive': True}, 'prog'), ({'restart_marker_blocks': 3}, 'rst'),
                           ({'optimize': True}, 'opt')]:
                buf = io.BytesIO()
                Image.fromarray(a).save(buf, 'JPEG', quality=int(rng.integers(30, 98)), subsampling=ss, **kw)
                add('pil_%dx%d_%s_%s' % (h, w, sn, kn), buf.getvalue())
        for kw, kn in [({}, 'base'), ({'progressive': True}, 'prog')]:
            buf = io.BytesIO()
            Image.fromarray(a[..., 0], 'L').save(buf, 'JPEG', quality=80, **kw)
            add('pil_%dx%d_gray_%s' % (h, w, kn), buf.getvalue())
    a = synth(rng, 37, 53)
    for o in range(1, 9):
        im = Image.fromarray(a)
        ex = im.getexif()
        ex[0x0112] = o
        buf = io.BytesIO()
        im.save(buf, 'JPEG', quality=90, subsampling=2, exif=ex.tobytes())
        add('pil_exif%d' % o, buf.getvalue())
    for (h, w) in [(40, 52), (17, 23), (9, 5), (3, 3)]:
        for samp in [[(1, 2), (1, 1), (1, 1)], [(4, 1), (1, 1), (1, 1)], [(4, 2), (1, 1), (1, 1)], [(2, 2), (2, 1), (1, 2)],
                     [(2, 2), (1, 2), (2, 1)], [(2, 4), (1, 1), (1, 1)], [(1, 4), (1, 1), (1, 1)], [(2, 1), (1, 1), (1, 1)]]:
            for ri in (0, 2):
                d = E.encode_rgb(synth(rng, h, w), samp, quality=int(rng.integers(40, 95)), restart_interval=ri)
                add('enc_%dx%d_%s_r%d' % (h, w, ''.join('%d%d' % s for s in samp), ri), d)
    out = {}
    for k, (d, ref) in cases.items():
        out['jpg_' + k] = d
        out['bgr_' + k] = ref
    out['versions'] = np.array(['Pillow %s' % Image.__version__ if hasattr(Image, '__version__') else 'Pillow',
                                'libjpeg-turbo %s' % features.version_feature('libjpeg_turbo')])
    path = os.path.join(ROOT, 'tests', 'golden', 'jpeg.npz')
    np.savez_compressed(path, **out)
    print('%d cases -> %s (%d KB)' % (len(cases), path, os.path.getsize(path) // 1024))


if __name__ == '__main__':
    main()
