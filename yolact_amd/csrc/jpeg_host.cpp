// JPEG entropy decoding on the host for COCODetection.pull_item's image read (data/coco.py:138-141: cv2.imread).
//
// cv2.imread == libjpeg-turbo with the library defaults.  The split here is the one a GPU wants: the serial part (marker
// parsing + Huffman decoding, ITU T.81) runs on the host and produces the QUANTISED coefficient blocks; everything that is
// data parallel (dequantisation, the ISLOW integer IDCT, fancy chroma upsampling, YCbCr -> BGR, EXIF orientation) runs on
// the GPU (jpeg.hip).  Baseline / extended sequential and progressive Huffman files, 8-bit, 1 or 3 components, restart
// intervals.  Arithmetic-coded, lossless, hierarchical, 12-bit and 4-component (CMYK / YCCK) files return YMI_EUNSUPPORTED.
// No device code in this file; no allocation (the caller owns the coefficient buffer).
#include "../../include/yolact_amd.h"
#include <string.h>

#define YMI_OK 0
#define YMI_EARG (-1)
#define YMI_ESHAPE (-2)
#define YMI_ENULL (-3)

namespace {

const uint8_t kZigzag[64 + 16] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                  41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                  30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
                                  // a corrupt run can step past 63: land on the last coefficient instead of outside the block
                                  63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};

struct Huff {
  bool defined = false;
  // canonical code (T.81 Annex C / F.2.2.3): per length the largest code, and the index of the first value
  int32_t maxcode[18];
  int32_t valptr[17];
  int32_t mincode[17];
  uint8_t vals[256];
  // 9-bit lookahead: (length << 8) | value, 0 = longer than 9 bits
  uint16_t look[512];
};

int build_huff(Huff &h, const uint8_t *bits, const uint8_t *vals, int nvals) {
  int code = 0, k = 0;
  memset(h.look, 0, sizeof(h.look));
  for (int l = 1; l <= 16; ++l) {
    h.valptr[l] = k;
    h.mincode[l] = code;
    for (int i = 0; i < bits[l - 1]; ++i) {
      if (k >= nvals || code >= (1 << l)) return YMI_EFORMAT;
      h.vals[k] = vals[k];
      if (l <= 9) {
        const int lo = code << (9 - l), n = 1 << (9 - l);
        for (int j = 0; j < n; ++j) h.look[lo + j] = (uint16_t)((l << 8) | vals[k]);
      }
      ++k;
      ++code;
    }
    h.maxcode[l] = bits[l - 1] ? code - 1 : -1;
    code <<= 1;
  }
  h.maxcode[17] = 0x7fffffff;
  h.defined = true;
  return YMI_OK;
}

struct Comp {
  int id, h, v, tq;
  int bw, bh, dw, dh;     // padded block grid; real downsampled size
  int16_t *coef;          // [bh][bw][64]
  int dc_tbl, ac_tbl;     // of the current scan
  int pred;
  bool qt_latched;
};

struct Frame {
  int width = 0, height = 0, ncomp = 0;
  bool progressive = false, seen = false;
  Comp comp[3];
  int hmax = 1, vmax = 1, mcux = 0, mcuy = 0;
};

// Entropy-coded segment reader: 0xFF00 unstuffing; at a marker the stream reads as zero bits (libjpeg's behaviour for
// truncated data) and `marker` remembers it.
struct Bits {
  const uint8_t *d;
  size_t n, p;
  uint64_t acc = 0;
  int cnt = 0;
  int marker = 0;

  inline void fill() {
    while (cnt <= 56) {
      uint32_t b = 0;
      if (!marker && p < n) {
        b = d[p];
        if (b == 0xFF) {
          const uint32_t b2 = p + 1 < n ? d[p + 1] : 0xD9;
          if (b2 == 0) p += 2;
          else if (b2 == 0xFF) { ++p; continue; }       // fill byte
          else { marker = (int)b2; b = 0; }
        } else {
          ++p;
        }
      } else if (!marker) {
        marker = 0xD9;
      }
      acc |= (uint64_t)b << (56 - cnt);
      cnt += 8;
    }
  }
  inline uint32_t peek(int k) { if (cnt < k) fill(); return (uint32_t)(acc >> (64 - k)); }
  inline void skip(int k) { acc <<= k; cnt -= k; }
  inline uint32_t get(int k) { if (k == 0) return 0; const uint32_t v = peek(k); skip(k); return v; }
  inline int decode(const Huff &h) {
    if (cnt < 16) fill();
    const uint32_t top = (uint32_t)(acc >> (64 - 9));
    const uint16_t e = h.look[top];
    if (e) { skip(e >> 8); return e & 0xFF; }
    uint32_t code = top;
    int l = 9;
    for (;;) {
      ++l;
      if (l > 16) return -1;
      code = (uint32_t)(acc >> (64 - l));
      if ((int32_t)code <= h.maxcode[l]) break;
    }
    skip(l);
    const int idx = h.valptr[l] + (int)code - h.mincode[l];
    return (idx >= 0 && idx < 256) ? h.vals[idx] : -1;
  }
  // byte-align, consume the RSTn marker
  int restart() {
    acc = 0; cnt = 0;
    if (!marker) {
      while (p + 1 < n && !(d[p] == 0xFF && d[p + 1] != 0 && d[p + 1] != 0xFF)) ++p;
      if (p + 1 >= n) return YMI_EFORMAT;
      marker = d[p + 1];
    }
    if (marker < 0xD0 || marker > 0xD7) return YMI_EFORMAT;
    p += 2;
    marker = 0;
    return YMI_OK;
  }
};

inline int extend(uint32_t v, int s) { return (int)v < (1 << (s - 1)) ? (int)v - (1 << s) + 1 : (int)v; }

struct Decoder {
  const uint8_t *d;
  size_t n;
  Frame f;
  Huff dc[4], ac[4];
  uint16_t qt[4][64];       // natural order
  bool qt_def[4] = {false, false, false, false};
  uint16_t *qt_out = nullptr;    // [3][64] latched per component
  int restart_interval = 0;
  int orientation = 1, adobe = -1;
  bool jfif = false;
  int nscans = 0;

  static int be16(const uint8_t *p) { return (p[0] << 8) | p[1]; }

  static int exif_orientation(const uint8_t *s, size_t len) {
    if (len < 14 || memcmp(s, "Exif\0\0", 6) != 0) return 1;
    const uint8_t *t = s + 6;
    const size_t tl = len - 6;
    const bool le = t[0] == 'I' && t[1] == 'I';
    if (!le && !(t[0] == 'M' && t[1] == 'M')) return 1;
    auto u16 = [&](size_t o) -> uint32_t { return le ? (t[o] | (t[o + 1] << 8)) : ((t[o] << 8) | t[o + 1]); };
    auto u32 = [&](size_t o) -> uint32_t {
      return le ? (t[o] | (t[o + 1] << 8) | (t[o + 2] << 16) | ((uint32_t)t[o + 3] << 24))
                : (((uint32_t)t[o] << 24) | (t[o + 1] << 16) | (t[o + 2] << 8) | t[o + 3]);
    };
    const size_t off = u32(4);
    if (off + 2 > tl) return 1;
    const uint32_t cnt = u16(off);
    for (uint32_t i = 0; i < cnt; ++i) {
      const size_t e = off + 2 + 12 * (size_t)i;
      if (e + 12 > tl) break;
      if (u16(e) == 0x0112 && u16(e + 2) == 3) {
        const uint32_t v = u16(e + 8);
        return (v >= 1 && v <= 8) ? (int)v : 1;
      }
    }
    return 1;
  }

  // One pass over the markers.  coefs == nullptr: headers only (stops at the first SOS).
  int run(int16_t *coefs) {
    if (n < 4 || d[0] != 0xFF || d[1] != 0xD8) return YMI_EFORMAT;
    size_t p = 2;
    while (p + 1 < n) {
      if (d[p] != 0xFF) return YMI_EFORMAT;
      while (p < n && d[p] == 0xFF) ++p;
      if (p >= n) break;
      const int m = d[p++];
      if (m == 0xD9) break;
      if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
      if (p + 2 > n) return YMI_EFORMAT;
      const size_t ln = (size_t)be16(d + p);
      if (ln < 2 || p + ln > n) return YMI_EFORMAT;
      const uint8_t *s = d + p + 2;
      const size_t sl = ln - 2;
      if (m == 0xC0 || m == 0xC1 || m == 0xC2) {
        if (f.seen || sl < 6) return YMI_EFORMAT;
        if (s[0] != 8) return YMI_EUNSUPPORTED;
        f.height = be16(s + 1); f.width = be16(s + 3); f.ncomp = s[5];
        if (f.ncomp != 1 && f.ncomp != 3) return YMI_EUNSUPPORTED;
        if (f.width <= 0 || f.height <= 0 || sl < 6 + 3 * (size_t)f.ncomp) return YMI_EFORMAT;
        f.progressive = (m == 0xC2);
        for (int i = 0; i < f.ncomp; ++i) {
          Comp &c = f.comp[i];
          c.id = s[6 + 3 * i]; c.h = s[7 + 3 * i] >> 4; c.v = s[7 + 3 * i] & 15; c.tq = s[8 + 3 * i];
          if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) return YMI_EFORMAT;
          c.qt_latched = false;
          if (c.h > f.hmax) f.hmax = c.h;
          if (c.v > f.vmax) f.vmax = c.v;
        }
        if (f.ncomp == 1) { f.comp[0].h = f.comp[0].v = 1; f.hmax = f.vmax = 1; }   // a single component is never subsampled
        f.mcux = (f.width + 8 * f.hmax - 1) / (8 * f.hmax);
        f.mcuy = (f.height + 8 * f.vmax - 1) / (8 * f.vmax);
        int16_t *q = coefs;
        for (int i = 0; i < f.ncomp; ++i) {
          Comp &c = f.comp[i];
          if (f.hmax % c.h || f.vmax % c.v) return YMI_EUNSUPPORTED;     // fractional sampling ratios
          c.bw = f.mcux * c.h; c.bh = f.mcuy * c.v;
          c.dw = (f.width * c.h + f.hmax - 1) / f.hmax; c.dh = (f.height * c.v + f.vmax - 1) / f.vmax;
          c.coef = q;
          if (q) q += (size_t)c.bw * c.bh * 64;
        }
        f.seen = true;
      } else if (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
        return YMI_EUNSUPPORTED;        // lossless / hierarchical / arithmetic
      } else if (m == 0xCC) {
        return YMI_EUNSUPPORTED;
      } else if (m == 0xC4) {
        size_t q = 0;
        while (q + 17 <= sl) {
          const int tc = s[q] >> 4, th = s[q] & 15;
          if (tc > 1 || th > 3) return YMI_EFORMAT;
          int nv = 0;
          for (int i = 0; i < 16; ++i) nv += s[q + 1 + i];
          if (nv > 256 || q + 17 + nv > sl) return YMI_EFORMAT;
          const int rc = build_huff(tc ? ac[th] : dc[th], s + q + 1, s + q + 17, nv);
          if (rc) return rc;
          q += 17 + nv;
        }
      } else if (m == 0xDB) {
        size_t q = 0;
        while (q < sl) {
          const int pq = s[q] >> 4, tq = s[q] & 15;
          if (tq > 3 || pq > 1 || q + 1 + (pq ? 128 : 64) > sl) return YMI_EFORMAT;
          for (int i = 0; i < 64; ++i)
            qt[tq][kZigzag[i]] = pq ? (uint16_t)be16(s + q + 1 + 2 * i) : s[q + 1 + i];
          qt_def[tq] = true;
          q += 1 + (pq ? 128 : 64);
        }
      } else if (m == 0xDD) {
        if (sl < 2) return YMI_EFORMAT;
        restart_interval = be16(s);
      } else if (m == 0xE0) {
        if (sl >= 5 && memcmp(s, "JFIF\0", 5) == 0) jfif = true;
      } else if (m == 0xE1) {
        const int o = exif_orientation(s, sl);
        if (o != 1) orientation = o;
      } else if (m == 0xEE) {
        if (sl >= 12 && memcmp(s, "Adobe", 5) == 0) adobe = s[11];
      } else if (m == 0xDA) {
        if (!f.seen) return YMI_EFORMAT;
        if (!coefs) return YMI_OK;          // header pass
        size_t end = 0;
        const int rc = scan(s, sl, p + ln, &end);
        if (rc) return rc;
        ++nscans;
        p = end;
        continue;
      }
      p += ln;
    }
    if (!f.seen) return YMI_EFORMAT;
    if (coefs && nscans == 0) return YMI_EFORMAT;
    return YMI_OK;
  }

  int scan(const uint8_t *s, size_t sl, size_t start, size_t *end) {
    if (sl < 1) return YMI_EFORMAT;
    const int ns = s[0];
    if (ns < 1 || ns > f.ncomp || sl < 1 + 2 * (size_t)ns + 3) return YMI_EFORMAT;
    Comp *sel[3];
    for (int i = 0; i < ns; ++i) {
      sel[i] = nullptr;
      for (int c = 0; c < f.ncomp; ++c) if (f.comp[c].id == s[1 + 2 * i]) sel[i] = &f.comp[c];
      if (!sel[i]) return YMI_EFORMAT;
      sel[i]->dc_tbl = s[2 + 2 * i] >> 4;
      sel[i]->ac_tbl = s[2 + 2 * i] & 15;
      if (sel[i]->dc_tbl > 3 || sel[i]->ac_tbl > 3) return YMI_EFORMAT;
      if (!sel[i]->qt_latched) {          // libjpeg latches the table when the component's first scan starts
        if (!qt_def[sel[i]->tq]) return YMI_EFORMAT;
        memcpy(qt_out + 64 * (sel[i] - f.comp), qt[sel[i]->tq], 128);
        sel[i]->qt_latched = true;
      }
    }
    if (ns > 1) {      // libjpeg: D_MAX_BLOCKS_IN_MCU = 10 (JERR_BAD_MCU_SIZE)
      int nb = 0;
      for (int i = 0; i < ns; ++i) nb += sel[i]->h * sel[i]->v;
      if (nb > 10) return YMI_EUNSUPPORTED;
    }
    const int Ss = s[1 + 2 * ns], Se = s[2 + 2 * ns], Ah = s[3 + 2 * ns] >> 4, Al = s[3 + 2 * ns] & 15;
    if (f.progressive) {
      if (Ss > Se || Se > 63 || Al > 13 || (Ss == 0 && Se != 0) || (Ss > 0 && ns != 1)) return YMI_EFORMAT;
    } else if (Ss != 0 || Se != 63 || Ah != 0 || Al != 0) {
      return YMI_EFORMAT;
    }
    for (int i = 0; i < ns; ++i) {
      const bool need_dc = !f.progressive || Ss == 0, need_ac = !f.progressive || Ss > 0;
      if (need_dc && !(f.progressive && Ah) && !dc[sel[i]->dc_tbl].defined) return YMI_EFORMAT;
      if (need_ac && !ac[sel[i]->ac_tbl].defined) return YMI_EFORMAT;
      sel[i]->pred = 0;
    }
    Bits br{d, n, start};
    int eobrun = 0;
    long units;
    int ux_n;
    if (ns > 1) { ux_n = f.mcux; units = (long)f.mcux * f.mcuy; }
    else { ux_n = (sel[0]->dw + 7) / 8; units = (long)ux_n * ((sel[0]->dh + 7) / 8); }
    for (long u = 0; u < units; ++u) {
      if (restart_interval && u && u % restart_interval == 0) {
        const int rc = br.restart();
        if (rc) return rc;
        for (int i = 0; i < ns; ++i) sel[i]->pred = 0;
        eobrun = 0;
      }
      const int ux = (int)(u % ux_n), uy = (int)(u / ux_n);
      for (int i = 0; i < ns; ++i) {
        Comp &c = *sel[i];
        const int nbx = ns > 1 ? c.h : 1, nby = ns > 1 ? c.v : 1;
        for (int by = 0; by < nby; ++by)
          for (int bx = 0; bx < nbx; ++bx) {
            const int X = ns > 1 ? ux * c.h + bx : ux, Y = ns > 1 ? uy * c.v + by : uy;
            int16_t *blk = c.coef + ((size_t)Y * c.bw + X) * 64;
            int rc;
            if (!f.progressive) rc = block_baseline(br, c, blk);
            else if (Ss == 0) rc = block_dc(br, c, blk, Ah, Al);
            else if (Ah == 0) rc = block_ac_first(br, c, blk, Ss, Se, Al, eobrun);
            else rc = block_ac_refine(br, c, blk, Ss, Se, Al, eobrun);
            if (rc) return rc;
          }
      }
    }
    // the next marker: where the reader stopped, or the next one in the stream
    size_t q = br.p;
    while (q + 1 < n && !(d[q] == 0xFF && d[q + 1] != 0 && d[q + 1] != 0xFF && !(d[q + 1] >= 0xD0 && d[q + 1] <= 0xD7))) ++q;
    *end = q;
    return YMI_OK;
  }

  int block_baseline(Bits &br, Comp &c, int16_t *blk) {
    int t = br.decode(dc[c.dc_tbl]);
    if (t < 0 || t > 15) return YMI_EFORMAT;
    if (t) c.pred += extend(br.get(t), t);
    blk[0] = (int16_t)c.pred;
    const Huff &h = ac[c.ac_tbl];
    for (int k = 1; k < 64;) {
      const int rs = br.decode(h);
      if (rs < 0) return YMI_EFORMAT;
      const int r = rs >> 4, s = rs & 15;
      if (s == 0) {
        if (r != 15) break;
        k += 16;
        continue;
      }
      k += r;
      blk[kZigzag[k]] = (int16_t)extend(br.get(s), s);
      ++k;
    }
    return YMI_OK;
  }

  int block_dc(Bits &br, Comp &c, int16_t *blk, int Ah, int Al) {
    if (Ah == 0) {
      const int t = br.decode(dc[c.dc_tbl]);
      if (t < 0 || t > 15) return YMI_EFORMAT;
      if (t) c.pred += extend(br.get(t), t);
      blk[0] = (int16_t)(c.pred * (1 << Al));
    } else if (br.get(1)) {
      blk[0] |= (int16_t)(1 << Al);
    }
    return YMI_OK;
  }

  int block_ac_first(Bits &br, Comp &c, int16_t *blk, int Ss, int Se, int Al, int &eobrun) {
    if (eobrun > 0) { --eobrun; return YMI_OK; }
    const Huff &h = ac[c.ac_tbl];
    for (int k = Ss; k <= Se;) {
      const int rs = br.decode(h);
      if (rs < 0) return YMI_EFORMAT;
      const int r = rs >> 4, s = rs & 15;
      if (s == 0) {
        if (r < 15) {
          eobrun = (1 << r) - 1;
          if (r) eobrun += (int)br.get(r);
          break;
        }
        k += 16;
        continue;
      }
      k += r;
      blk[kZigzag[k]] = (int16_t)(extend(br.get(s), s) * (1 << Al));
      ++k;
    }
    return YMI_OK;
  }

  int block_ac_refine(Bits &br, Comp &c, int16_t *blk, int Ss, int Se, int Al, int &eobrun) {
    const int p1 = 1 << Al, m1 = -(1 << Al);
    const Huff &h = ac[c.ac_tbl];
    int k = Ss;
    if (eobrun == 0) {
      for (; k <= Se; ++k) {
        const int rs = br.decode(h);
        if (rs < 0) return YMI_EFORMAT;
        int r = rs >> 4, s = rs & 15;
        if (s) {
          s = br.get(1) ? p1 : m1;
        } else if (r != 15) {
          eobrun = 1 << r;
          if (r) eobrun += (int)br.get(r);
          break;
        }
        do {
          int16_t *co = blk + kZigzag[k];
          if (*co != 0) {
            if (br.get(1) && (*co & p1) == 0) *co = (int16_t)(*co + (*co >= 0 ? p1 : m1));
          } else if (--r < 0) {
            break;
          }
          ++k;
        } while (k <= Se);
        if (s && k <= Se) blk[kZigzag[k]] = (int16_t)s;
      }
    }
    if (eobrun > 0) {
      for (; k <= Se; ++k) {
        int16_t *co = blk + kZigzag[k];
        if (*co != 0 && br.get(1) && (*co & p1) == 0) *co = (int16_t)(*co + (*co >= 0 ? p1 : m1));
      }
      --eobrun;
    }
    return YMI_OK;
  }
};

void fill_info(const Decoder &dec, ymi_jpeg_info *info) {
  const Frame &f = dec.f;
  memset(info, 0, sizeof(*info));
  info->width = f.width; info->height = f.height; info->ncomp = f.ncomp;
  info->progressive = f.progressive ? 1 : 0;
  info->orientation = dec.orientation;
  if (f.ncomp == 1) info->color = YMI_JPEG_GRAY;
  else {
    const bool rgb_ids = f.comp[0].id == 'R' && f.comp[1].id == 'G' && f.comp[2].id == 'B';
    // jdapimin.c default_decompress_parms: JFIF => YCbCr; Adobe transform 0 => RGB, 1 => YCbCr; neither: by component ids
    info->color = (dec.adobe == 0 || (dec.adobe < 0 && !dec.jfif && rgb_ids)) ? YMI_JPEG_RGB : YMI_JPEG_YCBCR;
  }
  int64_t ncoef = 0, plane = 0;
  for (int i = 0; i < f.ncomp; ++i) {
    const Comp &c = f.comp[i];
    info->hs[i] = c.h; info->vs[i] = c.v; info->bw[i] = c.bw; info->bh[i] = c.bh; info->dw[i] = c.dw; info->dh[i] = c.dh;
    info->hf[i] = f.hmax / c.h; info->vf[i] = f.vmax / c.v;
    ncoef += (int64_t)c.bw * c.bh * 64;
    plane += (int64_t)c.bw * c.bh * 64;
  }
  info->coef_count = ncoef;
  info->plane_bytes = plane;
  const bool swap = dec.orientation >= 5;
  info->out_width = swap ? f.height : f.width;
  info->out_height = swap ? f.width : f.height;
}

}  // namespace

extern "C" {

int ymi_jpeg_parse(const uint8_t *data, size_t n, ymi_jpeg_info *info) {
  if (!data || !info) return YMI_ENULL;
  Decoder dec;
  dec.d = data; dec.n = n;
  const int rc = dec.run(nullptr);
  if (rc) return rc;
  fill_info(dec, info);
  return YMI_OK;
}

int ymi_jpeg_decode_coefs(const uint8_t *data, size_t n, int16_t *coefs, int64_t coef_capacity, uint16_t *qt,
                          ymi_jpeg_info *info) {
  if (!data || !coefs || !qt || !info) return YMI_ENULL;
  {
    Decoder hdr;
    hdr.d = data; hdr.n = n;
    const int rc = hdr.run(nullptr);
    if (rc) return rc;
    fill_info(hdr, info);
    if (info->coef_count > coef_capacity) return YMI_ESHAPE;
  }
  memset(coefs, 0, (size_t)info->coef_count * sizeof(int16_t));
  memset(qt, 0, 3 * 64 * sizeof(uint16_t));
  Decoder dec;
  dec.d = data; dec.n = n; dec.qt_out = qt;
  const int rc = dec.run(coefs);
  if (rc) return rc;
  for (int i = 0; i < dec.f.ncomp; ++i) if (!dec.f.comp[i].qt_latched) return YMI_EFORMAT;   // a component no scan covered
  fill_info(dec, info);
  return YMI_OK;
}

}
