"""A minimal baseline JPEG *encoder* for test vectors.  *** TEST INFRASTRUCTURE ONLY ***

Pillow can only write 4:4:4 / 4:2:2 / 4:2:0 files.  libjpeg decodes any integer sampling ratio, and its upsampler has
separate code paths for them (jdsample.c: h1v2 fancy upsampling for 4:4:0, box replication for 4:1:1 / 4:1:0 and for
components narrower than three samples).  This writes syntactically plain baseline files (T.81: SOF0, Annex K Huffman
tables, scaled Annex K quantisation tables, optional restart intervals) with ARBITRARY sampling factors so that those
paths can be pinned against libjpeg-turbo itself (Pillow decodes them) — see oracle/make_golden_jpeg.py.  Image quality is
irrelevant; only the syntax has to be valid.
"""
from __future__ import annotations

import struct

import numpy as np

from .jpeg_oracle import ZIGZAG

# T.81 Annex K.1 quantisation tables (natural order) and K.3 typical Huffman tables
_QL = [16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87,
       80, 62, 18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72,
       92, 95, 98, 112, 100, 103, 99]
_QC = [17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99,
       99, 99] + [99] * 32
_DC_L = ([0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0], list(range(12)))
_DC_C = ([0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0], list(range(12)))
_AC_L = ([0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d],
         [0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14,
          0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09,
          0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a,
          0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65,
          0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88,
          0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9,
          0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca,
          0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea,
          0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa])
_AC_C = ([0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77],
         [0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32,
          0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16,
          0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39,
          0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64,
          0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86,
          0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7,
          0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8,
          0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9,
          0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa])


def _codes(bits, vals):
    out, code, k = {}, 0, 0
    for length in range(1, 17):
        for _ in range(bits[length - 1]):
            out[vals[k]] = (code, length)
            k += 1
            code += 1
        code <<= 1
    return out


class _BitWriter:
    def __init__(self):
        self.out = bytearray()
        self.acc, self.n = 0, 0

    def put(self, code, length):
        self.acc = (self.acc << length) | code
        self.n += length
        while self.n >= 8:
            b = (self.acc >> (self.n - 8)) & 0xFF
            self.out.append(b)
            if b == 0xFF:
                self.out.append(0)
            self.n -= 8
        self.acc &= (1 << self.n) - 1

    def flush(self):
        if self.n:
            self.put((1 << (8 - self.n)) - 1, 8 - self.n)


def _dct_matrix():
    m = np.zeros((8, 8))
    for k in range(8):
        for n in range(8):
            m[k, n] = (np.sqrt(1 / 8) if k == 0 else np.sqrt(2 / 8)) * np.cos((2 * n + 1) * k * np.pi / 16)
    return m


def _magnitude(v):
    a = abs(v)
    s = a.bit_length()
    return s, (v if v >= 0 else v + (1 << s) - 1)


def encode(planes, sampling, width, height, quality=75, restart_interval=0, jfif=True):
    """planes: one uint8 array per component at its DOWNSAMPLED size ceil(width*h/hmax) x ceil(height*v/vmax);
    sampling: [(h, v), ...] (1 or 3 components: Y, Cb, Cr).  Returns the file bytes."""
    nc = len(planes)
    hmax, vmax = max(h for h, _ in sampling), max(v for _, v in sampling)
    sf = 5000 / quality if quality < 50 else 200 - 2 * quality
    qts = [np.clip((np.array(q) * sf + 50) // 100, 1, 255).astype(np.int64) for q in (_QL, _QC)]
    mcux, mcuy = -(-width // (8 * hmax)), -(-height // (8 * vmax))
    D = _dct_matrix()
    coefs = []
    for ci, (pl, (h, v)) in enumerate(zip(planes, sampling)):
        bw, bh = mcux * h, mcuy * v
        dh, dw = pl.shape
        full = np.zeros((bh * 8, bw * 8))
        full[:dh, :dw] = pl
        full[:dh, dw:] = pl[:, -1:]                      # replicate the edge like an encoder would
        full[dh:, :] = full[dh - 1:dh, :]
        blocks = full.reshape(bh, 8, bw, 8).transpose(0, 2, 1, 3) - 128.0
        f = np.einsum('ij,abjk,lk->abil', D, blocks, D)
        q = qts[0 if ci == 0 else 1].reshape(8, 8)
        coefs.append(np.rint(f / q).astype(np.int64).reshape(bh, bw, 64))
    out = bytearray(b'\xff\xd8')
    if jfif:
        out += b'\xff\xe0' + struct.pack('>H', 16) + b'JFIF\x00\x01\x01\x00\x00\x01\x00\x01\x00\x00'
    for t in range(1 if nc == 1 else 2):
        zz = np.zeros(64, dtype=np.int64)
        zz[:] = qts[t][ZIGZAG]
        out += b'\xff\xdb' + struct.pack('>HB', 67, t) + bytes(int(x) for x in zz)
    out += b'\xff\xc0' + struct.pack('>HBHHB', 8 + 3 * nc, 8, height, width, nc)
    for ci, (h, v) in enumerate(sampling):
        out += bytes([ci + 1, (h << 4) | v, 0 if ci == 0 else 1])
    tables = [(0x00, _DC_L), (0x10, _AC_L)] + ([(0x01, _DC_C), (0x11, _AC_C)] if nc > 1 else [])
    for tid, (bits, vals) in tables:
        out += b'\xff\xc4' + struct.pack('>HB', 19 + len(vals), tid) + bytes(bits) + bytes(vals)
    if restart_interval:
        out += b'\xff\xdd' + struct.pack('>HH', 4, restart_interval)
    out += b'\xff\xda' + struct.pack('>HB', 6 + 2 * nc, nc)
    for ci in range(nc):
        out += bytes([ci + 1, 0x00 if ci == 0 else 0x11])
    out += bytes([0, 63, 0])
    dc_codes = [_codes(*_DC_L), _codes(*_DC_C)]
    ac_codes = [_codes(*_AC_L), _codes(*_AC_C)]
    bw_ = _BitWriter()
    pred = [0] * nc
    n = 0
    for my in range(mcuy):
        for mx in range(mcux):
            if restart_interval and n and n % restart_interval == 0:
                bw_.flush()
                bw_.out += bytes([0xFF, 0xD0 + ((n // restart_interval - 1) & 7)])
                pred = [0] * nc
            n += 1
            for ci, (h, v) in enumerate(sampling):
                t = 0 if ci == 0 else 1
                for by in range(v):
                    for bx in range(h):
                        blk = coefs[ci][my * v + by, mx * h + bx]
                        diff = int(blk[0]) - pred[ci]
                        pred[ci] = int(blk[0])
                        s, bits = _magnitude(diff)
                        bw_.put(*dc_codes[t][s])
                        if s:
                            bw_.put(bits, s)
                        run = 0
                        for k in range(1, 64):
                            c = int(blk[ZIGZAG[k]])
                            if c == 0:
                                run += 1
                                continue
                            while run > 15:
                                bw_.put(*ac_codes[t][0xF0])
                                run -= 16
                            s, bits = _magnitude(c)
                            bw_.put(*ac_codes[t][(run << 4) | s])
                            bw_.put(bits, s)
                            run = 0
                        if run:
                            bw_.put(*ac_codes[t][0x00])
    bw_.flush()
    out += bw_.out + b'\xff\xd9'
    return bytes(out)


def encode_rgb(rgb: np.ndarray, sampling, quality=75, restart_interval=0) -> bytes:
    """uint8 [h,w,3] RGB -> JPEG with the given per-component sampling factors (box-average downsampling)."""
    h, w, _ = rgb.shape
    r, g, b = [rgb[..., i].astype(np.float64) for i in range(3)]
    y = 0.299 * r + 0.587 * g + 0.114 * b
    cb = -0.168736 * r - 0.331264 * g + 0.5 * b + 128
    cr = 0.5 * r - 0.418688 * g - 0.081312 * b + 128
    hmax, vmax = max(s[0] for s in sampling), max(s[1] for s in sampling)
    planes = []
    for pl, (sh, sv) in zip((y, cb, cr), sampling):
        fh, fv = hmax // sh, vmax // sv
        dw, dh = -(-w * sh // hmax), -(-h * sv // vmax)
        pad = np.pad(pl, ((0, dh * fv - h), (0, dw * fh - w)), mode='edge')
        ds = pad.reshape(dh, fv, dw, fh).mean(axis=(1, 3))
        planes.append(np.clip(np.rint(ds), 0, 255).astype(np.uint8))
    return encode(planes, sampling, w, h, quality, restart_interval)
