"""The image read of COCODetection.pull_item (data/coco.py:138-141: `cv2.imread`) — SURVEY §8(f) rank 4.

cv2.imread on a JPEG = libjpeg-turbo (library defaults) + EXIF orientation + BGR.  tests/golden/jpeg.npz holds 156 files
with the pixels libjpeg-turbo itself produced for them (Pillow's decoder, oracle/make_golden_jpeg.py): Pillow-written
4:4:4 / 4:2:2 / 4:2:0 at odd sizes (down to 1x1), baseline / optimised Huffman / progressive / restart intervals /
grayscale / EXIF orientations 1-8, and files with the sampling ratios only oracle/jpeg_encode.py can write (4:4:0, 4:1:1,
4:1:0, mixed chroma factors: libjpeg's h1v2 and replication upsamplers).  Integer / byte work: the bar is bit-exact.

CPU (this file, no GPU):
  * the oracle (oracle/jpeg_oracle.py) against every golden file, and live against Pillow on photographs when Pillow is
    importable (it is in the build image);
  * the product's HOST half (csrc/jpeg_host.cpp through the C ABI): quantised coefficients and latched quantisation
    tables equal to the oracle's for every golden file; error codes for non-JPEG / unsupported / corrupt input;
  * the product's DEVICE arithmetic, executed on the host: tests/jpeg_emul.cpp (built here with g++) loops over blocks
    and pixels calling the very inline functions the kernels call (csrc/jpeg_math.h) -> pixels equal to the golden ones.
GPU: tests/test_gpu_jpeg.py runs the kernels themselves through ymi_jpeg_reconstruct_bgr_u8 on the same files.
"""
import ctypes as C
import glob
import io
import os
import subprocess

import numpy as np
import pytest

from oracle import jpeg_oracle as J
from yolact_amd import _lib as L

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = np.load(os.path.join(HERE, 'golden', 'jpeg.npz'))
NAMES = sorted(k[4:] for k in GOLD.files if k.startswith('jpg_'))
# the pure-Python entropy decoder of the oracle is slow: the big Pillow cases go through it in one test only
SMALL = [n for n in NAMES if GOLD['bgr_' + n].size <= 40 * 52 * 3]


def gold(name):
    return GOLD['jpg_' + name].tobytes(), GOLD['bgr_' + name]


def host_decode(data):
    lib = L.lib()
    info = L.JpegInfo()
    L.check(lib.ymi_jpeg_parse(data, len(data), C.byref(info)), 'ymi_jpeg_parse')
    coefs = np.full(int(info.coef_count), 12345, dtype=np.int16)        # the call must zero what no scan writes
    qt = np.zeros(192, dtype=np.uint16)
    L.check(lib.ymi_jpeg_decode_coefs(data, len(data), coefs.ctypes.data, coefs.size, qt.ctypes.data, C.byref(info)),
            'ymi_jpeg_decode_coefs')
    return info, coefs, qt


@pytest.fixture(scope='module')
def emul(tmp_path_factory):
    so = str(tmp_path_factory.mktemp('emul') / 'libjpeg_emul.so')
    subprocess.run(['g++', '-O2', '-shared', '-fPIC', '-o', so, os.path.join(HERE, 'jpeg_emul.cpp')], check=True)
    lib = C.CDLL(so)
    lib.emul_jpeg_reconstruct_bgr_u8.restype = C.c_int
    lib.emul_jpeg_reconstruct_bgr_u8.argtypes = [C.POINTER(L.JpegInfo), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    return lib


def test_fixture_inventory():
    assert len(NAMES) == 156
    kinds = {n.split('_')[0] for n in NAMES}
    assert kinds == {'pil', 'enc'}
    assert sum('prog' in n for n in NAMES) >= 20 and sum('rst' in n for n in NAMES) >= 15
    assert sum(n.startswith('pil_exif') for n in NAMES) == 8
    assert 'libjpeg-turbo' in str(GOLD['versions'][1])


def test_oracle_matches_libjpeg_turbo_on_every_golden_file():
    for n in NAMES:
        data, ref = gold(n)
        got = J.imread_bgr(data)
        assert got.shape == ref.shape and got.dtype == np.uint8, n
        assert np.array_equal(got, ref), (n, int((got != ref).sum()))


def test_oracle_matches_pillow_live_on_photographs():
    Image = pytest.importorskip('PIL.Image')
    files = []
    for pat in ('/usr/local/lib/python3*/dist-packages/matplotlib/mpl-data/sample_data/grace_hopper.jpg',
                '/usr/share/javascript/highlight.js/styles/*.jpg'):
        files += sorted(glob.glob(pat))
    if not files:
        pytest.skip('no photographs on this machine')
    for f in files[:3]:
        data = open(f, 'rb').read()
        ref = np.array(Image.open(io.BytesIO(data)).convert('RGB'))[..., ::-1]
        assert np.array_equal(J.imread_bgr(data, honour_exif=False), ref), f


def test_host_entropy_decoder_matches_oracle_coefficients():
    for n in NAMES:
        data, ref = gold(n)
        info, coefs, qt = host_decode(data)
        oi = J.parse(data)
        oc = J.decode_coefficients(data, oi)
        assert (info.width, info.height, info.ncomp) == (oi['width'], oi['height'], len(oi['comps'])), n
        assert bool(info.progressive) == oi['progressive'] and info.orientation == oi['orientation'], n
        assert (info.out_height, info.out_width) == ref.shape[:2], n
        want = np.concatenate([c.reshape(-1) for c in oc]).astype(np.int16)
        assert coefs.size == want.size and np.array_equal(coefs, want), n
        for i, c in enumerate(oi['comps']):
            assert (info.bw[i], info.bh[i], info.dw[i], info.dh[i]) == (c['bw'], c['bh'], c['dw'], c['dh']), n
            assert np.array_equal(qt[64 * i:64 * i + 64].astype(np.int64), oi['scans'][0]['qt'][c['tq']]), n


def test_device_arithmetic_on_host_matches_libjpeg_turbo(emul):
    """jpeg_math.h (what the kernels execute) through the g++-built emulation == the golden pixels, every file."""
    for n in NAMES:
        data, ref = gold(n)
        info, coefs, qt = host_decode(data)
        planes = np.zeros(int(info.plane_bytes), dtype=np.uint8)
        out = np.zeros((info.out_height, info.out_width, 3), dtype=np.uint8)
        assert emul.emul_jpeg_reconstruct_bgr_u8(C.byref(info), coefs.ctypes.data, qt.ctypes.data, planes.ctypes.data,
                                                 out.ctypes.data) == 0
        assert np.array_equal(out, ref), (n, int((out != ref).sum()))


def test_idct_extremes_agree_between_oracle_and_device_arithmetic(emul):
    """Saturating blocks (all coefficients at the 8-bit-precision extremes): the range-limit table wraps identically."""
    rng = np.random.default_rng(5)
    data, _ = gold('pil_64x64_444_base')
    info, coefs, qt = host_decode(data)
    oi = J.parse(data)
    for trial in range(4):
        c = rng.integers(-1024, 1024, coefs.size).astype(np.int16) if trial else np.full(coefs.size, 1023, np.int16)
        planes = np.zeros(int(info.plane_bytes), dtype=np.uint8)
        out = np.zeros((info.out_height, info.out_width, 3), dtype=np.uint8)
        emul.emul_jpeg_reconstruct_bgr_u8(C.byref(info), c.ctypes.data, qt.ctypes.data, planes.ctypes.data, out.ctypes.data)
        off = 0
        for i, comp in enumerate(oi['comps']):
            nb = info.bw[i] * info.bh[i]
            blk = c[off:off + nb * 64].reshape(info.bh[i], info.bw[i], 64).astype(np.int32)
            want = J.idct_islow(blk, qt[64 * i:64 * i + 64].astype(np.int64))
            want = want.transpose(0, 2, 1, 3).reshape(info.bh[i] * 8, info.bw[i] * 8)
            got = planes[off:off + nb * 64].reshape(info.bh[i] * 8, info.bw[i] * 8)
            assert np.array_equal(got, want), (trial, i)
            off += nb * 64


def test_error_codes():
    lib = L.lib()
    info = L.JpegInfo()
    png = b'\x89PNG\r\n\x1a\n' + b'\x00' * 32
    assert lib.ymi_jpeg_parse(png, len(png), C.byref(info)) == L.EFORMAT
    assert lib.ymi_jpeg_parse(None, 0, C.byref(info)) == -3
    data, _ = gold('pil_33x47_420_base')
    # SOF0 -> SOF3 (lossless) / SOF9 (arithmetic): valid JPEG processes outside the subset
    i = data.index(b'\xff\xc0')
    for m in (0xC3, 0xC9):
        bad = data[:i + 1] + bytes([m]) + data[i + 2:]
        assert lib.ymi_jpeg_parse(bad, len(bad), C.byref(info)) == L.EUNSUPPORTED
    # 12-bit precision
    bad = data[:i + 4] + bytes([12]) + data[i + 5:]
    assert lib.ymi_jpeg_parse(bad, len(bad), C.byref(info)) == L.EUNSUPPORTED
    # a segment length that runs past the end of the file
    bad = data[:i + 2] + b'\xff\xff' + data[i + 4:]
    assert lib.ymi_jpeg_parse(bad, len(bad), C.byref(info)) == L.EFORMAT
    # capacity check of the coefficient buffer
    assert lib.ymi_jpeg_parse(data, len(data), C.byref(info)) == 0
    coefs = np.zeros(16, dtype=np.int16)
    qt = np.zeros(192, dtype=np.uint16)
    assert lib.ymi_jpeg_decode_coefs(data, len(data), coefs.ctypes.data, coefs.size, qt.ctypes.data, C.byref(info)) == -2
    # a missing Huffman table
    j = data.index(b'\xff\xc4')
    ln = int.from_bytes(data[j + 2:j + 4], 'big')
    bad = data[:j] + data[j + 2 + ln:]
    coefs = np.zeros(int(info.coef_count), dtype=np.int16)
    rc = lib.ymi_jpeg_decode_coefs(bad, len(bad), coefs.ctypes.data, coefs.size, qt.ctypes.data, C.byref(info))
    assert rc == L.EFORMAT
    with pytest.raises(J.JpegError):
        J.parse(png)


def test_truncated_stream_decodes_like_libjpeg_pads():
    """libjpeg treats a premature end of the entropy-coded data as zero bits (a warning, not an error); so do both halves
    here, and they agree with each other on what comes out."""
    data, _ = gold('pil_64x64_420_base')
    cut = data[:len(data) * 2 // 3]
    info, coefs, _ = host_decode(cut)
    oi = J.parse(cut)
    oc = J.decode_coefficients(cut, oi)
    assert np.array_equal(coefs, np.concatenate([c.reshape(-1) for c in oc]).astype(np.int16))


def test_python_wrapper_host_side():
    from yolact_amd.data import jpeg
    data, ref = gold('pil_exif6')
    info = jpeg.parse(data)
    assert (info.out_height, info.out_width) == ref.shape[:2] and info.orientation == 6 and info.sampling[0] == (2, 2)
    with pytest.raises(ValueError):
        jpeg.parse(b'GIF89a....')
    inf2, coefs, qt = jpeg.decode_coefficients(data)
    assert coefs.numel() == info.coef_count and coefs.dtype.is_floating_point is False
