// Ground-truth mask rasterisation for COCODetection.pull_item (data/coco.py:144-148: `self.coco.annToMask(obj)`): host code.
// pycocotools' maskApi.c restated: rleFrPoly (polygon -> column-major run lengths), rleFrString (compressed counts string),
// rleDecode; a polygon list is the UNION of its polygons (rleMerge with intersect = 0), so every piece is OR-ed into the
// caller's [h,w] row-major uint8 mask.  No device code; the only allocation is host scratch inside a call.
#include "../../include/yolact_amd.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

#define YMI_OK 0
#define YMI_EARG (-1)
#define YMI_ESHAPE (-2)
#define YMI_ENULL (-3)

namespace {

// runs of a column-major flattening, starting with zeros, OR-ed into a row-major mask
int fill_runs(const uint32_t *cnt, long n, int h, int w, uint8_t *mask) {
  const long total = (long)h * w;
  if (total > (1L << 31) - 1) return YMI_EARG;
  long pos = 0;
  int v = 0;
  for (long i = 0; i < n; ++i) {
    const long c = cnt[i];
    if (pos + c > total) return YMI_EFORMAT;
    if (v) {
      for (long q = pos; q < pos + c; ++q) mask[(q % h) * (long)w + q / h] = 1;
    }
    pos += c;
    v ^= 1;
  }
  return pos == total ? YMI_OK : YMI_EFORMAT;
}

}  // namespace

extern "C" {

int ymi_coco_poly_fill_u8(const double *xy, int k, int h, int w, uint8_t *mask) {
  if (!xy || !mask) return YMI_ENULL;
  if (k < 1 || h <= 0 || w <= 0 || (long)h * w > (1L << 31) - 1) return YMI_EARG;
  // maskApi.c scales by 5 into int: coordinates that cannot be image coordinates (or NaN) are an argument error here
  // (and bounds the boundary walk below: with |coordinate| <= 3 * max(w, h) + 1024 an edge contributes at most
  // 5 * (6 * max(w, h) + 2048) points, so memory stays proportional to the input — coordinates up to 1e5 used to let one
  // hostile polygon ask for gigabytes; pycocotools itself has no such guard)
  const double lim = 3.0 * (w > h ? w : h) + 1024.0;
  for (int j = 0; j < 2 * k; ++j) if (!(std::fabs(xy[j]) <= lim)) return YMI_EARG;
  const double scale = 5;
  std::vector<int> x(k + 1), y(k + 1);
  for (int j = 0; j < k; ++j) x[j] = (int)(scale * xy[j * 2 + 0] + .5);
  x[k] = x[0];
  for (int j = 0; j < k; ++j) y[j] = (int)(scale * xy[j * 2 + 1] + .5);
  y[k] = y[0];
  size_t m = 0;
  for (int j = 0; j < k; ++j) m += (size_t)std::max(std::abs(x[j] - x[j + 1]), std::abs(y[j] - y[j + 1])) + 1;
  std::vector<int> u, v;
  u.reserve(m);
  v.reserve(m);
  // upsample and get discrete points densely along the entire boundary
  for (int j = 0; j < k; ++j) {
    int xs = x[j], xe = x[j + 1], ys = y[j], ye = y[j + 1];
    const int dx = std::abs(xe - xs), dy = std::abs(ys - ye);
    const bool flip = (dx >= dy && xs > xe) || (dx < dy && ys > ye);
    if (flip) { std::swap(xs, xe); std::swap(ys, ye); }
    if (dx >= dy) {
      const double s = dx ? (double)(ye - ys) / dx : 0.0;     // dx == dy == 0: a repeated vertex, one point
      for (int d = 0; d <= dx; ++d) {
        const int t = flip ? dx - d : d;
        u.push_back(t + xs);
        v.push_back((int)(ys + s * t + .5));
      }
    } else {
      const double s = (double)(xe - xs) / dy;
      for (int d = 0; d <= dy; ++d) {
        const int t = flip ? dy - d : d;
        v.push_back(t + ys);
        u.push_back((int)(xs + s * t + .5));
      }
    }
  }
  // get points along the y-boundary and downsample
  std::vector<uint32_t> a;
  for (size_t j = 1; j < u.size(); ++j)
    if (u[j] != u[j - 1]) {
      double xd = (double)(u[j] < u[j - 1] ? u[j] : u[j] - 1);
      xd = (xd + .5) / scale - .5;
      if (std::floor(xd) != xd || xd < 0 || xd > w - 1) continue;
      double yd = (double)(v[j] < v[j - 1] ? v[j] : v[j - 1]);
      yd = (yd + .5) / scale - .5;
      if (yd < 0) yd = 0;
      else if (yd > h) yd = h;
      yd = std::ceil(yd);
      a.push_back((uint32_t)((int)xd * h + (int)yd));
    }
  // run lengths from the sorted boundary points
  a.push_back((uint32_t)((long)h * w));
  std::sort(a.begin(), a.end());
  uint32_t p = 0;
  for (size_t j = 0; j < a.size(); ++j) { const uint32_t t = a[j]; a[j] -= p; p = t; }
  std::vector<uint32_t> b;
  size_t j = 0;
  b.push_back(a[j++]);
  while (j < a.size()) {
    if (a[j] > 0) b.push_back(a[j++]);
    else {
      ++j;
      if (j < a.size()) b.back() += a[j++];
    }
  }
  return fill_runs(b.data(), (long)b.size(), h, w, mask);
}

int ymi_coco_rle_fill_u8(const uint32_t *counts, long n, int h, int w, uint8_t *mask) {
  if (!counts || !mask) return YMI_ENULL;
  if (n < 0 || h <= 0 || w <= 0) return YMI_EARG;
  return fill_runs(counts, n, h, w, mask);
}

int ymi_coco_rle_string_fill_u8(const char *s, long len, int h, int w, uint8_t *mask) {
  if (!s || !mask) return YMI_ENULL;
  if (len < 0 || h <= 0 || w <= 0) return YMI_EARG;
  // maskApi.c rleFrString
  std::vector<uint32_t> cnts;
  long p = 0;
  while (p < len) {
    long x = 0;
    int k = 0, more = 1;
    while (more) {
      if (p >= len) return YMI_EFORMAT;
      const long c = (long)(unsigned char)s[p] - 48;
      x |= (c & 0x1f) << (5 * k);
      more = (int)(c & 0x20);
      ++p;
      ++k;
      if (k > 12) return YMI_EFORMAT;                           // before the shift below: 5 * 13 = 65 bits would be undefined
      if (!more && (c & 0x10)) x |= (long)(~0UL << (5 * k));   // sign extension (maskApi.c: x |= -1 << 5*k)
    }
    if (cnts.size() > 2) x += (long)cnts[cnts.size() - 2];
    if (x < 0) return YMI_EFORMAT;
    cnts.push_back((uint32_t)x);
  }
  return fill_runs(cnts.data(), (long)cnts.size(), h, w, mask);
}

}
