"""GPU: the image read and the whole `COCODetection.pull_item` (data/coco.py:100-176) through the C ABI.

  * `yolact_amd.data.jpeg.imread` (host entropy decode -> ymi_jpeg_reconstruct_bgr_u8 on the device) on all 156 golden files:
    bit-identical to libjpeg-turbo's pixels (tests/golden/jpeg.npz);
  * a 640x480-class photograph-sized synthetic file (the kernels' launch geometry beyond the tiny fixtures), checked
    against the host emulation of the same arithmetic (tests/jpeg_emul.cpp) — and, if Pillow is present, against it;
  * pull_item on the synthetic COCO dataset (tests/coco_synth.py) against the oracle's restatement: masks / targets /
    sizes / crowd counts exact, the transformed image within 2e-3 of the normalised range (cv2.resize's and the kernel's
    bilinear differ only in how the sample coordinate is rounded), every output form of the reference's return tuple.
"""
import ctypes as C
import io
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle import coco_dataset as OD
from tests import coco_synth
from yolact_amd import _lib as L
from yolact_amd.coco import COCO_LABEL_MAP

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, 'golden', 'jpeg.npz'))
NAMES = sorted(k[4:] for k in GOLD.files if k.startswith('jpg_'))


def test_imread_is_bit_identical_to_libjpeg_turbo_on_every_golden_file():
    from yolact_amd.data import jpeg
    bad = []
    for n in NAMES:
        data, ref = GOLD['jpg_' + n].tobytes(), GOLD['bgr_' + n]
        got = jpeg.imread(data)
        assert got.is_cuda and got.dtype == torch.uint8 and tuple(got.shape) == ref.shape, n
        if not np.array_equal(got.cpu().numpy(), ref):
            bad.append((n, int((got.cpu().numpy() != ref).sum())))
    assert not bad, bad[:10]


def _big_jpeg():
    from oracle import jpeg_encode as E
    rng = np.random.default_rng(8)
    h, w = 427, 641
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([128 + 100 * np.sin(xx / 17.0 + yy / 23.0), 128 + 90 * np.cos(xx / 11.0) * np.sin(yy / 19.0),
                    (xx + yy) * 255.0 / (w + h)], -1) + rng.normal(0, 10, (h, w, 3))
    return E.encode_rgb(np.clip(img, 0, 255).astype(np.uint8), [(2, 2), (1, 1), (1, 1)], quality=85, restart_interval=7)


def test_imread_photo_sized_matches_host_emulation(tmp_path):
    from yolact_amd.data import jpeg
    data = _big_jpeg()
    got = jpeg.imread(data).cpu().numpy()
    so = str(tmp_path / 'libjpeg_emul.so')
    subprocess.run(['g++', '-O2', '-shared', '-fPIC', '-o', so, os.path.join(HERE, 'jpeg_emul.cpp')], check=True)
    emul = C.CDLL(so)
    emul.emul_jpeg_reconstruct_bgr_u8.argtypes = [C.POINTER(L.JpegInfo), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    info, coefs, qt = jpeg.decode_coefficients(data)
    coefs, qt = coefs.numpy().copy(), qt.numpy().copy()
    planes = np.zeros(int(info.plane_bytes), dtype=np.uint8)
    want = np.zeros((info.out_height, info.out_width, 3), dtype=np.uint8)
    emul.emul_jpeg_reconstruct_bgr_u8(C.byref(info.raw), coefs.ctypes.data, qt.ctypes.data, planes.ctypes.data, want.ctypes.data)
    assert got.shape == (427, 641, 3) and np.array_equal(got, want)
    try:
        from PIL import Image
    except ImportError:
        return
    ref = np.array(Image.open(io.BytesIO(data)).convert('RGB'))[..., ::-1]
    assert np.array_equal(got, ref)


def test_imread_rejects_what_it_cannot_decode():
    from yolact_amd.data import jpeg
    with pytest.raises(ValueError):
        jpeg.imread(b'\x89PNG\r\n\x1a\n' + b'\x00' * 64)
    data = GOLD['jpg_pil_33x47_420_base'].tobytes()
    i = data.index(b'\xff\xc0')
    with pytest.raises(RuntimeError, match='unsupported|subset'):
        jpeg.imread(data[:i + 1] + b'\xc9' + data[i + 2:])
    with pytest.raises(TypeError):
        jpeg.imread(12345)


def test_pull_item_matches_the_oracle(tmp_path):
    import yolact_amd
    from yolact_amd.data import COCODetection
    from yolact_amd.utils.augmentations import BaseTransform
    yolact_amd.set_cfg('yolact_resnet50_config')
    info_file = coco_synth.write_dataset(str(tmp_path))
    ds = COCODetection(str(tmp_path), info_file, transform=BaseTransform())
    assert len(ds) == 4 and ds.ids == [139, 285, 632, 724]
    for idx in range(len(ds)):
        img, target, masks, h, w, num_crowds = ds.pull_item(idx)
        o_img, o_target, o_masks, o_h, o_w, o_crowds = OD.pull_item(str(tmp_path), info_file, idx, COCO_LABEL_MAP)
        assert img.is_cuda and img.dtype == torch.float32 and tuple(img.shape) == (3, 550, 550)
        assert (h, w, int(num_crowds)) == (o_h, o_w, int(o_crowds)), idx
        assert masks.dtype == np.uint8 and np.array_equal(masks, o_masks), idx
        assert target.shape == o_target.shape and np.array_equal(target, o_target), idx
        err = np.abs(img.cpu().numpy() - o_img).max()
        assert err < 2e-3, (idx, err)
    # image 139: the crowd annotation (first in the file) is last, the sliver was discarded
    img, target, masks, h, w, num_crowds = ds.pull_item(0)
    assert target.shape == (3, 5) and target[-1, 4] == -1 and int(num_crowds) == 1 and masks.shape == (3, 50, 35)
    # image 632: only a crowd; decoded upright (EXIF orientation 6)
    img, target, masks, h, w, num_crowds = ds.pull_item(2)
    assert (h, w) == (53, 37) and int(num_crowds) == 1 and target[0, 4] == -1
    # __getitem__ shape of the return value (data/coco.py:88-98)
    im, (gt, mk, nc) = ds[1]
    assert tuple(im.shape) == (3, 550, 550) and gt.shape[1] == 5 and mk.shape[0] == gt.shape[0]
    # has_gt=False: every image, no targets (the reference's evalimages-style use)
    ds2 = COCODetection(str(tmp_path), info_file, transform=BaseTransform(), has_gt=False)
    assert len(ds2) == 5
    img, target, masks, h, w, num_crowds = ds2.pull_item(4)
    o_img = OD.pull_item(str(tmp_path), info_file, 4, COCO_LABEL_MAP, has_gt=False)[0]
    assert target is None and masks is None and (h, w) == (17, 3) and np.abs(img.cpu().numpy() - o_img).max() < 2e-3
    # pull_image / pull_anno
    raw = ds.pull_image(0)
    assert raw.dtype == torch.uint8 and tuple(raw.shape) == (50, 35, 3)
    assert np.array_equal(raw.cpu().numpy(), GOLD['bgr_pil_50x35_420_base'])
    assert [a['image_id'] for a in ds.pull_anno(0)] == [139] * 4


def test_pulled_item_feeds_the_network(tmp_path):
    """pull_item -> unsqueeze -> Yolact.forward, the eval.py:936-944 sequence."""
    import bench
    from yolact_amd.data import COCODetection
    from yolact_amd.utils.augmentations import BaseTransform
    info_file = coco_synth.write_dataset(str(tmp_path))
    dev = torch.device('cuda', 0)
    with torch.no_grad():
        net, _ = bench.build_model(dev, 550)
        ds = COCODetection(str(tmp_path), info_file, transform=BaseTransform())
        img, gt, gt_masks, h, w, num_crowd = ds.pull_item(0)
        preds = net(img.unsqueeze(0))
    assert isinstance(preds, list) and len(preds) == 1 and 'detection' in preds[0]
