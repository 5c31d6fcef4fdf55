"""`cv2.imread` for JPEG files with the pixel work on the GPU (COCODetection.pull_item, data/coco.py:138-141).

cv2.imread(path) on a JPEG = libjpeg-turbo with its defaults + EXIF orientation -> uint8 BGR [h,w,3].  Here the serial
half (markers, Huffman decoding; csrc/jpeg_host.cpp) runs on the host and hands the quantised coefficient blocks to the
GPU, which does everything data parallel (dequantise, ISLOW IDCT, fancy chroma upsampling, YCbCr -> BGR, orientation;
csrc/jpeg.hip).  The result is bit-identical to libjpeg-turbo's (tests/test_jpeg.py, tests/test_gpu_jpeg.py) and stays
on the device for the transform that follows.  Not a JPEG -> ValueError (cv2 would try its other codecs; COCO is JPEG
only); corrupt / unsupported streams -> RuntimeError with the library's message.  There is no CPU decode path.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import torch

from .. import _lib as L


MAX_COEFS = 1 << 29        # 512 M coefficients (a 16k x 16k 4:4:4 image); a corrupt header must not allocate 26 GB


class JpegInfo:
    """Plain-Python view of ymi_jpeg_info."""

    def __init__(self, raw: L.JpegInfo):
        self.raw = raw
        self.width, self.height = raw.width, raw.height
        self.out_width, self.out_height = raw.out_width, raw.out_height
        self.ncomp, self.progressive, self.orientation, self.color = raw.ncomp, bool(raw.progressive), raw.orientation, raw.color
        self.sampling = [(raw.hs[i], raw.vs[i]) for i in range(raw.ncomp)]
        self.coef_count, self.plane_bytes = raw.coef_count, raw.plane_bytes


def _as_bytes(src) -> bytes:
    if isinstance(src, (bytes, bytearray, memoryview)):
        return bytes(src)
    if isinstance(src, (str, os.PathLike)):
        with open(src, 'rb') as f:
            return f.read()
    raise TypeError('imread: a path or the file bytes, got %r' % type(src))


def parse(src) -> JpegInfo:
    """Header only (host): sizes, sampling, progressive flag, EXIF orientation."""
    data = _as_bytes(src)
    if data[:2] != b'\xff\xd8':
        raise ValueError('not a JPEG stream (no SOI marker)')
    raw = L.JpegInfo()
    L.check(L.lib().ymi_jpeg_parse(data, len(data), C.byref(raw)), 'ymi_jpeg_parse')
    if raw.coef_count > MAX_COEFS:
        raise ValueError('JPEG header declares %d x %d pixels: larger than this reader accepts' % (raw.width, raw.height))
    return JpegInfo(raw)


class _Staging(threading.local):
    """Per-thread pinned staging buffer for the coefficients (grown on demand); the event orders its reuse behind the
    previous image's host-to-device copy."""

    def __init__(self):
        self.buf = None
        self.qt = None
        self.event = None


_staging = _Staging()


def decode_coefficients(src):
    """Host half only: (JpegInfo, coefs int16 [coef_count] pinned CPU tensor view, qt int16-typed [192] CPU tensor holding
    uint16 values).  Exposed for the CPU tests; `imread` is the product entry."""
    data = _as_bytes(src)
    info = parse(data)
    st = _staging
    if st.event is not None:
        st.event.synchronize()
    if st.buf is None or st.buf.numel() < info.coef_count:
        pin = torch.cuda.is_available()
        st.buf = torch.empty(max(int(info.coef_count), 1 << 20), dtype=torch.int16, pin_memory=pin)
        st.qt = torch.empty(192, dtype=torch.int16, pin_memory=pin)
    L.check(L.lib().ymi_jpeg_decode_coefs(data, len(data), st.buf.data_ptr(), st.buf.numel(), st.qt.data_ptr(),
                                          C.byref(info.raw)), 'ymi_jpeg_decode_coefs')
    return info, st.buf[:info.coef_count], st.qt


def imread(src, device=None) -> torch.Tensor:
    """path | bytes -> uint8 BGR [h, w, 3] on `device` (default: the current CUDA device), EXIF orientation applied."""
    if not torch.cuda.is_available():
        raise RuntimeError('yolact_amd.data.jpeg.imread: no GPU — the pixel reconstruction runs on the device only '
                           '(the CPU oracle lives under oracle/ and is test-only)')
    device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    info, coefs, qt = decode_coefficients(src)
    with torch.cuda.device(device):
        coefs_d = coefs.to(device, non_blocking=True)
        qt_d = qt.to(device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        _staging.event = ev
        planes = torch.empty(int(info.plane_bytes), dtype=torch.uint8, device=device)
        out = torch.empty((info.out_height, info.out_width, 3), dtype=torch.uint8, device=device)
        L.check(L.lib().ymi_jpeg_reconstruct_bgr_u8(C.byref(info.raw), coefs_d.data_ptr(), qt_d.data_ptr(), planes.data_ptr(),
                                                    out.data_ptr(), L.stream_ptr()), 'ymi_jpeg_reconstruct_bgr_u8')
    return out
