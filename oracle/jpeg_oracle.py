"""CPU restatement of `cv2.imread(path)` for JPEG files.  *** TEST INFRASTRUCTURE ONLY ***

`COCODetection.pull_item` (data/coco.py:138-141) reads every image with `cv2.imread` -> uint8 BGR [h,w,3].  OpenCV is not
in /root/reference (an unpinned pip dependency, environment.yml:27 `opencv-python`); its JPEG reader is a thin wrapper over
the bundled **libjpeg-turbo** run with the library defaults (ISLOW integer IDCT, "fancy" triangle-filter chroma
upsampling, no merged upsampling, YCbCr -> RGB by the fixed-point tables) followed by the EXIF orientation (IMREAD_COLOR
honours it since OpenCV 3.1).  This file restates that published algorithm (ITU T.81 entropy decoding + libjpeg's
jidctint.c / jdsample.c / jdcolor.c arithmetic) in plain Python / numpy:

  parse()              markers: SOF0/1/2, DHT, DQT, DRI, SOS, APP1 (EXIF orientation), APP14 (Adobe transform)
  decode_coefficients  Huffman decoding, baseline and progressive (spectral selection + successive approximation)
  idct_islow           jidctint.c jpeg_idct_islow: CONST_BITS 13, PASS1_BITS 2, the 12 FIX constants, range limit
  upsample             jdsample.c: h2v1 / h2v2 fancy upsampling (downsampled_width > 2), replication otherwise; the
                       context rows at the top / bottom edge duplicate the first / last real row (jdmainct.c)
  ycc_to_bgr           jdcolor.c build_ycc_rgb_table / ycc_rgb_convert
  imread_bgr           the whole thing, + EXIF orientation

PINNED: tests/test_jpeg.py checks it bit for bit against libjpeg-turbo itself — Pillow's decoder (libjpeg-turbo
3.1.4, the same library and defaults OpenCV wraps; `PIL.features.version_feature('libjpeg_turbo')`) — on photographs
found in this image and on synthetic files written at every chroma subsampling, odd sizes, restart intervals, progressive
mode, grayscale; the committed fixtures (tests/golden/jpeg.npz, made by oracle/make_golden_jpeg.py) carry the same
bytes + pixels to the GPU box.  cv2 itself is absent here, so "cv2.imread == libjpeg-turbo defaults + EXIF rotate + BGR" is
taken from OpenCV's published grfmt_jpeg.cpp, not executed.
"""
from __future__ import annotations

import struct
from typing import Dict, List, Optional

import numpy as np

ZIGZAG = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7,
                   14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39,
                   46, 53, 60, 61, 54, 47, 55, 62, 63], dtype=np.int64)      # zigzag index -> natural (row-major) index


class JpegError(ValueError):
    pass


class _Huff:
    """T.81 Annex C canonical code; decode by code length (F.2.2.3)."""

    def __init__(self, bits: List[int], vals: List[int]):
        self.lookup: Dict = {}
        code = 0
        k = 0
        for length in range(1, 17):
            for _ in range(bits[length - 1]):
                self.lookup[(length, code)] = vals[k]
                k += 1
                code += 1
            code <<= 1


class _Bits:
    """Entropy-coded segment reader: 0xFF00 unstuffing; a marker ends the data (further bits read as 0)."""

    def __init__(self, data: bytes, pos: int):
        self.d, self.p = data, pos
        self.acc, self.n = 0, 0
        self.marker: Optional[int] = None

    def _fill(self):
        if self.marker is None and self.p < len(self.d):
            b = self.d[self.p]
            if b == 0xFF:
                b2 = self.d[self.p + 1] if self.p + 1 < len(self.d) else 0xD9
                if b2 == 0:
                    self.p += 2
                else:
                    self.marker = b2
                    b = 0
            else:
                self.p += 1
        else:
            b = 0
        self.acc = (self.acc << 8) | b
        self.n += 8

    def bit(self) -> int:
        if self.n == 0:
            self._fill()
        self.n -= 1
        return (self.acc >> self.n) & 1

    def bits(self, k: int) -> int:
        v = 0
        for _ in range(k):
            v = (v << 1) | self.bit()
        return v

    def decode(self, h: _Huff) -> int:
        code = 0
        for length in range(1, 17):
            code = (code << 1) | self.bit()
            v = h.lookup.get((length, code))
            if v is not None:
                return v
        raise JpegError('bad Huffman code')

    def restart(self):
        """Align to a byte boundary and consume the RSTn marker."""
        self.acc, self.n = 0, 0
        if self.marker is None:
            # scan forward to the marker (padding bits were already consumed)
            while self.p + 1 < len(self.d) and not (self.d[self.p] == 0xFF and self.d[self.p + 1] not in (0, 0xFF)):
                self.p += 1
            self.marker = self.d[self.p + 1]
        if not (0xD0 <= self.marker <= 0xD7):
            raise JpegError('expected RSTn, found 0x%02x' % self.marker)
        self.p += 2
        self.marker = None


def _extend(v: int, s: int) -> int:
    """F.2.2.1 EXTEND."""
    return v if v >= (1 << (s - 1)) else v - (1 << s) + 1


def _exif_orientation(seg: bytes) -> int:
    if not seg.startswith(b'Exif\x00\x00'):
        return 1
    t = seg[6:]
    if len(t) < 8:
        return 1
    e = '<' if t[:2] == b'II' else '>'
    if t[:2] not in (b'II', b'MM'):
        return 1
    off = struct.unpack(e + 'I', t[4:8])[0]
    if off + 2 > len(t):
        return 1
    n = struct.unpack(e + 'H', t[off:off + 2])[0]
    for i in range(n):
        ent = t[off + 2 + 12 * i: off + 14 + 12 * i]
        if len(ent) < 12:
            break
        tag, typ, cnt = struct.unpack(e + 'HHI', ent[:8])
        if tag == 0x0112 and typ == 3:
            v = struct.unpack(e + 'H', ent[8:10])[0]
            return v if 1 <= v <= 8 else 1
    return 1


def parse(data: bytes) -> Dict:
    if data[:2] != b'\xff\xd8':
        raise JpegError('not a JPEG (no SOI)')
    info = {'qt': {}, 'dc': {}, 'ac': {}, 'scans': [], 'restart': 0, 'orientation': 1, 'adobe': None, 'jfif': False}
    p = 2
    while p < len(data):
        if data[p] != 0xFF:
            raise JpegError('marker expected at %d' % p)
        while data[p] == 0xFF:
            p += 1
        m = data[p]
        p += 1
        if m == 0xD9:
            break
        if m == 0x01 or 0xD0 <= m <= 0xD7:
            continue
        ln = struct.unpack('>H', data[p:p + 2])[0]
        seg = data[p + 2:p + ln]
        if m in (0xC0, 0xC1, 0xC2):
            prec, h, w, nc = struct.unpack('>BHHB', seg[:6])
            if prec != 8:
                raise JpegError('only 8-bit samples')
            comps = []
            for i in range(nc):
                cid, hv, tq = seg[6 + 3 * i: 9 + 3 * i]
                comps.append({'id': cid, 'h': hv >> 4, 'v': hv & 15, 'tq': tq})
            info.update(width=w, height=h, comps=comps, progressive=(m == 0xC2))
        elif m in (0xC3, 0xC5, 0xC6, 0xC7, 0xC9, 0xCA, 0xCB, 0xCD, 0xCE, 0xCF):
            raise JpegError('unsupported JPEG process SOF%d' % (m - 0xC0))
        elif m == 0xC4:
            q = 0
            while q < len(seg):
                tc, th = seg[q] >> 4, seg[q] & 15
                bits = list(seg[q + 1:q + 17])
                n = sum(bits)
                vals = list(seg[q + 17:q + 17 + n])
                (info['ac'] if tc else info['dc'])[th] = _Huff(bits, vals)
                q += 17 + n
            # tables defined between scans apply to the later scans: remember them per scan below
        elif m == 0xDB:
            q = 0
            while q < len(seg):
                pq, tq = seg[q] >> 4, seg[q] & 15
                if pq:
                    t = np.array(struct.unpack('>64H', seg[q + 1:q + 129]), dtype=np.int64); q += 129
                else:
                    t = np.array(list(seg[q + 1:q + 65]), dtype=np.int64); q += 65
                nat = np.zeros(64, dtype=np.int64)
                nat[ZIGZAG] = t
                info['qt'][tq] = nat
        elif m == 0xDD:
            info['restart'] = struct.unpack('>H', seg[:2])[0]
        elif m == 0xE0 and seg[:5] == b'JFIF\x00':
            info['jfif'] = True
        elif m == 0xE1:
            o = _exif_orientation(seg)
            if o != 1:
                info['orientation'] = o
        elif m == 0xEE and seg[:5] == b'Adobe' and len(seg) >= 12:
            info['adobe'] = seg[11]
        elif m == 0xDA:
            ns = seg[0]
            sc = {'comps': [], 'dc': dict(info['dc']), 'ac': dict(info['ac']), 'restart': info['restart'],
                  'qt': {k: v.copy() for k, v in info['qt'].items()}}
            for i in range(ns):
                cs, t = seg[1 + 2 * i], seg[2 + 2 * i]
                sc['comps'].append((cs, t >> 4, t & 15))
            sc['Ss'], sc['Se'], ahal = seg[1 + 2 * ns], seg[2 + 2 * ns], seg[3 + 2 * ns]
            sc['Ah'], sc['Al'] = ahal >> 4, ahal & 15
            sc['start'] = p + ln
            info['scans'].append(sc)
            # skip the entropy-coded data
            q = p + ln
            while q + 1 < len(data):
                if data[q] == 0xFF and data[q + 1] != 0 and not (0xD0 <= data[q + 1] <= 0xD7):
                    break
                q += 1
            else:
                q = len(data)          # truncated file: no marker after the entropy-coded data
            p = q
            continue
        p += ln
    if 'comps' not in info:
        raise JpegError('no frame header')
    return info


def decode_coefficients(data: bytes, info: Dict) -> List[np.ndarray]:
    """-> per component int32 [blocks_h, blocks_w, 64] quantised coefficients in natural order (blocks padded to whole MCUs)."""
    comps = info['comps']
    hmax, vmax = max(c['h'] for c in comps), max(c['v'] for c in comps)
    W, H = info['width'], info['height']
    mcux, mcuy = -(-W // (8 * hmax)), -(-H // (8 * vmax))
    coef = []
    for c in comps:
        c['bw'], c['bh'] = mcux * c['h'], mcuy * c['v']                       # padded block grid
        c['dw'], c['dh'] = -(-W * c['h'] // hmax), -(-H * c['v'] // vmax)     # downsampled_width / height
        coef.append(np.zeros((c['bh'], c['bw'], 64), dtype=np.int32))
    idx = {c['id']: i for i, c in enumerate(comps)}
    for sc in info['scans']:
        br = _Bits(data, sc['start'])
        sel = [(idx[cs], td, ta) for cs, td, ta in sc['comps']]
        Ss, Se, Ah, Al = sc['Ss'], sc['Se'], sc['Ah'], sc['Al']
        prog = info['progressive']
        pred = [0] * len(comps)
        eobrun = 0
        if len(sel) > 1:
            if sum(comps[ci]['h'] * comps[ci]['v'] for ci, _, _ in sel) > 10:
                raise JpegError('more than 10 blocks per MCU (libjpeg D_MAX_BLOCKS_IN_MCU)')
            units = [(mx, my) for my in range(mcuy) for mx in range(mcux)]
        else:
            c = comps[sel[0][0]]
            units = [(bx, by) for by in range(-(-c['dh'] // 8)) for bx in range(-(-c['dw'] // 8))]
        ri = sc['restart']
        for n, (ux, uy) in enumerate(units):
            if ri and n and n % ri == 0:
                br.restart()
                pred = [0] * len(comps)
                eobrun = 0
            if len(sel) > 1:
                blocks = [(ci, td, ta, ux * comps[ci]['h'] + bx, uy * comps[ci]['v'] + by)
                          for ci, td, ta in sel for by in range(comps[ci]['v']) for bx in range(comps[ci]['h'])]
            else:
                blocks = [(sel[0][0], sel[0][1], sel[0][2], ux, uy)]
            for ci, td, ta, bx, by in blocks:
                blk = coef[ci][by, bx]
                if not prog:
                    t = br.decode(sc['dc'][td])
                    pred[ci] += _extend(br.bits(t), t) if t else 0
                    blk[0] = pred[ci]
                    k = 1
                    h = sc['ac'][ta]
                    while k < 64:
                        rs = br.decode(h)
                        r, s = rs >> 4, rs & 15
                        if s == 0:
                            if r == 15:
                                k += 16
                                continue
                            break
                        k += r
                        blk[ZIGZAG[k]] = _extend(br.bits(s), s)
                        k += 1
                elif Ss == 0:
                    if Ah == 0:
                        t = br.decode(sc['dc'][td])
                        pred[ci] += _extend(br.bits(t), t) if t else 0
                        blk[0] = pred[ci] * (1 << Al)
                    elif br.bit():
                        blk[0] |= (1 << Al)
                elif Ah == 0:
                    if eobrun > 0:
                        eobrun -= 1
                        continue
                    h = sc['ac'][ta]
                    k = Ss
                    while k <= Se:
                        rs = br.decode(h)
                        r, s = rs >> 4, rs & 15
                        if s == 0:
                            if r < 15:
                                eobrun = (1 << r) - 1
                                if r:
                                    eobrun += br.bits(r)
                                break
                            k += 16
                            continue
                        k += r
                        blk[ZIGZAG[k]] = _extend(br.bits(s), s) * (1 << Al)
                        k += 1
                else:
                    p1, m1 = 1 << Al, -(1 << Al)
                    h = sc['ac'][ta]
                    k = Ss
                    if eobrun == 0:
                        while k <= Se:
                            rs = br.decode(h)
                            r, s = rs >> 4, rs & 15
                            if s:
                                s = p1 if br.bit() else m1
                            elif r != 15:
                                eobrun = 1 << r
                                if r:
                                    eobrun += br.bits(r)
                                break
                            while k <= Se:
                                z = ZIGZAG[k]
                                if blk[z] != 0:
                                    if br.bit() and (blk[z] & p1) == 0:
                                        blk[z] += p1 if blk[z] >= 0 else m1
                                else:
                                    r -= 1
                                    if r < 0:
                                        break
                                k += 1
                            if s and k <= Se:
                                blk[ZIGZAG[k]] = s
                            k += 1
                    if eobrun > 0:
                        while k <= Se:
                            z = ZIGZAG[k]
                            if blk[z] != 0 and br.bit() and (blk[z] & p1) == 0:
                                blk[z] += p1 if blk[z] >= 0 else m1
                            k += 1
                        eobrun -= 1
    return coef


# jidctint.c
_C = dict(F0_298=2446, F0_390=3196, F0_541=4433, F0_765=6270, F0_899=7373, F1_175=9633, F1_501=12299, F1_847=15137,
          F1_961=16069, F2_053=16819, F2_562=20995, F3_072=25172)


def _idct_1d(i0, i1, i2, i3, i4, i5, i6, i7):
    """One jidctint.c pass over 8 int64 arrays; returns the 8 un-descaled sums (tmp10+tmp3, ...)."""
    c = _C
    z2, z3 = i2, i6
    z1 = (z2 + z3) * c['F0_541']
    tmp2 = z1 + z3 * (-c['F1_847'])
    tmp3 = z1 + z2 * c['F0_765']
    tmp0 = (i0 + i4) << 13
    tmp1 = (i0 - i4) << 13
    tmp10, tmp13, tmp11, tmp12 = tmp0 + tmp3, tmp0 - tmp3, tmp1 + tmp2, tmp1 - tmp2
    tmp0, tmp1, tmp2, tmp3 = i7, i5, i3, i1
    z1, z2, z3, z4 = tmp0 + tmp3, tmp1 + tmp2, tmp0 + tmp2, tmp1 + tmp3
    z5 = (z3 + z4) * c['F1_175']
    tmp0, tmp1, tmp2, tmp3 = tmp0 * c['F0_298'], tmp1 * c['F2_053'], tmp2 * c['F3_072'], tmp3 * c['F1_501']
    z1, z2, z3, z4 = z1 * (-c['F0_899']), z2 * (-c['F2_562']), z3 * (-c['F1_961']), z4 * (-c['F0_390'])
    z3 = z3 + z5
    z4 = z4 + z5
    tmp0, tmp1, tmp2, tmp3 = tmp0 + z1 + z3, tmp1 + z2 + z4, tmp2 + z2 + z3, tmp3 + z1 + z4
    return [tmp10 + tmp3, tmp11 + tmp2, tmp12 + tmp1, tmp13 + tmp0, tmp13 - tmp0, tmp12 - tmp1, tmp11 - tmp2, tmp10 - tmp3]


def _descale(x, n):
    return (x + (1 << (n - 1))) >> n


def idct_islow(coef: np.ndarray, qt: np.ndarray) -> np.ndarray:
    """coef [...,64] quantised natural order, qt [64] -> uint8 samples [...,8,8] (jidctint.c jpeg_idct_islow)."""
    d = (coef.astype(np.int64) * qt.astype(np.int64)).reshape(coef.shape[:-1] + (8, 8))     # [.., row, col]
    cols = _idct_1d(*[d[..., r, :] for r in range(8)])                       # pass 1: down the columns
    ws = np.stack([_descale(v, 13 - 2) for v in cols], axis=-2)                              # [.., row, col]
    rows = _idct_1d(*[ws[..., :, c] for c in range(8)])                      # pass 2: along the rows
    out = np.stack([_descale(v, 13 + 2 + 3) for v in rows], axis=-1)
    # IDCT_range_limit: sample_range_limit + CENTERJSAMPLE indexed by (x & RANGE_MASK), RANGE_MASK = 1023
    x = out & 1023
    res = np.where(x < 128, x + 128, np.where(x < 512, 255, np.where(x < 896, 0, x - 896)))
    return res.astype(np.uint8)


def _planes(coef: List[np.ndarray], info: Dict) -> List[np.ndarray]:
    planes = []
    for c, cf in zip(info['comps'], coef):
        qt = info['scans'][0]['qt'].get(c['tq'])
        for sc in info['scans']:            # the table in force when the component's first scan starts (single-table files: trivial)
            if any(cs == c['id'] for cs, _, _ in sc['comps']):
                qt = sc['qt'][c['tq']]
                break
        px = idct_islow(cf, qt)             # [bh, bw, 8, 8]
        bh, bw = cf.shape[:2]
        planes.append(px.transpose(0, 2, 1, 3).reshape(bh * 8, bw * 8))
    return planes


def upsample(plane: np.ndarray, dw: int, dh: int, hf: int, vf: int) -> np.ndarray:
    """plane: IDCT output (padded); (dw, dh) the real downsampled size; expand by hf x vf (jdsample.c)."""
    p = plane[:dh, :dw].astype(np.int64)
    if hf == 1 and vf == 1:
        return p
    if hf == 2 and vf == 1 and dw > 2:                       # h2v1_fancy_upsample
        out = np.empty((dh, 2 * dw), dtype=np.int64)
        prev = np.concatenate([p[:, :1], p[:, :-1]], axis=1)
        nxt = np.concatenate([p[:, 1:], p[:, -1:]], axis=1)
        out[:, 0::2] = (p * 3 + prev + 1) >> 2
        out[:, 1::2] = (p * 3 + nxt + 2) >> 2
        out[:, 0] = p[:, 0]
        out[:, -1] = p[:, -1]
        return out
    if hf == 2 and vf == 2 and dw > 2:                       # h2v2_fancy_upsample
        above = np.concatenate([p[:1], p[:-1]], axis=0)      # context rows: the edge rows are duplicated (jdmainct.c)
        below = np.concatenate([p[1:], p[-1:]], axis=0)
        out = np.empty((2 * dh, 2 * dw), dtype=np.int64)
        for v, other in ((0, above), (1, below)):
            cs = p * 3 + other                               # thiscolsum
            last = np.concatenate([cs[:, :1], cs[:, :-1]], axis=1)
            nxt = np.concatenate([cs[:, 1:], cs[:, -1:]], axis=1)
            even = (cs * 3 + last + 8) >> 4
            odd = (cs * 3 + nxt + 7) >> 4
            even[:, 0] = (cs[:, 0] * 4 + 8) >> 4
            odd[:, -1] = (cs[:, -1] * 4 + 7) >> 4
            out[v::2, 0::2] = even
            out[v::2, 1::2] = odd
        return out
    if hf == 1 and vf == 2:                                  # h1v2_fancy_upsample (libjpeg-turbo >= 1.5.1)
        above = np.concatenate([p[:1], p[:-1]], axis=0)
        below = np.concatenate([p[1:], p[-1:]], axis=0)
        out = np.empty((2 * dh, dw), dtype=np.int64)
        out[0::2] = (p * 3 + above + 1) >> 2
        out[1::2] = (p * 3 + below + 2) >> 2
        return out
    return np.repeat(np.repeat(p, vf, axis=0), hf, axis=1)   # h2v1 / h2v2 / int_upsample: replication


def ycc_to_bgr(y, cb, cr) -> np.ndarray:
    """jdcolor.c build_ycc_rgb_table + ycc_rgb_convert (SCALEBITS 16)."""
    x_cb, x_cr = cb - 128, cr - 128
    r = y + ((91881 * x_cr + 32768) >> 16)
    b = y + ((116130 * x_cb + 32768) >> 16)
    g = y + ((-22554 * x_cb + 32768 + (-46802) * x_cr) >> 16)
    return np.clip(np.stack([b, g, r], axis=-1), 0, 255).astype(np.uint8)


def apply_orientation(img: np.ndarray, o: int) -> np.ndarray:
    """EXIF orientation 1..8 -> upright image (OpenCV ExifTransform / PIL ImageOps.exif_transpose semantics)."""
    if o == 2:
        return img[:, ::-1]
    if o == 3:
        return img[::-1, ::-1]
    if o == 4:
        return img[::-1]
    if o == 5:
        return img.transpose(1, 0, 2)
    if o == 6:
        return img.transpose(1, 0, 2)[:, ::-1]
    if o == 7:
        return img.transpose(1, 0, 2)[::-1, ::-1]
    if o == 8:
        return img.transpose(1, 0, 2)[::-1]
    return img


def imread_bgr(data: bytes, honour_exif: bool = True) -> np.ndarray:
    info = parse(data)
    coef = decode_coefficients(data, info)
    comps = info['comps']
    W, H = info['width'], info['height']
    hmax, vmax = max(c['h'] for c in comps), max(c['v'] for c in comps)
    planes = _planes(coef, info)
    full = []
    for c, pl in zip(comps, planes):
        if hmax % c['h'] or vmax % c['v']:
            raise JpegError('fractional sampling ratio')
        full.append(upsample(pl, c['dw'], c['dh'], hmax // c['h'], vmax // c['v'])[:H, :W])
    if len(comps) == 1:
        img = np.repeat(full[0].astype(np.uint8)[..., None], 3, axis=-1)
    elif len(comps) == 3:
        ids = [c['id'] for c in comps]
        rgb = (info['adobe'] == 0) or (info['adobe'] is None and not info['jfif'] and ids == [82, 71, 66])
        if rgb:
            img = np.stack([full[2], full[1], full[0]], axis=-1).astype(np.uint8)
        else:
            img = ycc_to_bgr(full[0], full[1], full[2])
    else:
        raise JpegError('%d-component JPEG (CMYK / YCCK) is not supported' % len(comps))
    if honour_exif:
        img = apply_orientation(img, info['orientation'])
    return np.ascontiguousarray(img)
