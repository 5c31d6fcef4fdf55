// Arithmetic of the JPEG reconstruction kernels (jpeg.hip), shared verbatim with the host emulation that the CPU tests
// build with g++ (tests/test_jpeg.py): the SAME inline functions run per block / per pixel on both sides, so the integer
// arithmetic is checked against the oracle without a GPU and the GPU test only has to prove the launch geometry.
//   idct8 / descale / idct_range_limit   jidctint.c jpeg_idct_islow (CONST_BITS 13, PASS1_BITS 2)
//   sample_at                            jdsample.c h2v1 / h2v2 / h1v2 fancy upsampling, replication otherwise
//   ycc_to_bgr                           jdcolor.c build_ycc_rgb_table / ycc_rgb_convert (SCALEBITS 16)
//   orient                               EXIF orientation 1..8 -> destination pixel
#pragma once
#include <stddef.h>
#include <stdint.h>
#if defined(__HIPCC__)
#define YMI_HD __host__ __device__ __forceinline__
#else
#define YMI_HD static inline
#endif

namespace ymi_jpeg {

constexpr int F0_298 = 2446, F0_390 = 3196, F0_541 = 4433, F0_765 = 6270, F0_899 = 7373, F1_175 = 9633, F1_501 = 12299,
              F1_847 = 15137, F1_961 = 16069, F2_053 = 16819, F2_562 = 20995, F3_072 = 25172;

// one 1-D pass of jidctint.c on 8 values; out[k] are the un-descaled sums
YMI_HD void idct8(const long in[8], long out[8]) {
  long z2 = in[2], z3 = in[6];
  long z1 = (z2 + z3) * F0_541;
  long tmp2 = z1 + z3 * (-(long)F1_847);
  long tmp3 = z1 + z2 * F0_765;
  long tmp0 = (in[0] + in[4]) * 8192;          // << CONST_BITS (a multiply: the operand may be negative)
  long tmp1 = (in[0] - in[4]) * 8192;
  const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
  tmp0 = in[7]; tmp1 = in[5]; tmp2 = in[3]; tmp3 = in[1];
  z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
  long z4 = tmp1 + tmp3;
  const long z5 = (z3 + z4) * F1_175;
  tmp0 *= F0_298; tmp1 *= F2_053; tmp2 *= F3_072; tmp3 *= F1_501;
  z1 *= -(long)F0_899; z2 *= -(long)F2_562; z3 *= -(long)F1_961; z4 *= -(long)F0_390;
  z3 += z5; z4 += z5;
  tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
  out[0] = tmp10 + tmp3; out[7] = tmp10 - tmp3;
  out[1] = tmp11 + tmp2; out[6] = tmp11 - tmp2;
  out[2] = tmp12 + tmp1; out[5] = tmp12 - tmp1;
  out[3] = tmp13 + tmp0; out[4] = tmp13 - tmp0;
}

YMI_HD long descale(long x, int n) { return (x + (1L << (n - 1))) >> n; }

struct CompPlane {
  const uint8_t *p;
  int stride, dw, dh, hf, vf, mode;      // mode: 0 full size, 1 h2v1 fancy, 2 h2v2 fancy, 3 h1v2 fancy, 4 replicate
};
struct ColorArgs {
  CompPlane c[3];
  int ncomp, color, W, H, orientation, out_w;
};

YMI_HD int sample_at(const CompPlane &c, int x, int y) {
  if (c.mode == 0) return c.p[(size_t)y * c.stride + x];
  if (c.mode == 4) return c.p[(size_t)(y / c.vf) * c.stride + x / c.hf];
  if (c.mode == 1) {           // h2v1_fancy_upsample
    const int i = x >> 1;
    const uint8_t *r = c.p + (size_t)y * c.stride;
    const int v = r[i];
    if (x & 1) return i == c.dw - 1 ? v : (v * 3 + r[i + 1] + 2) >> 2;
    return i == 0 ? v : (v * 3 + r[i - 1] + 1) >> 2;
  }
  const int j = y >> 1;
  // the nearer row is j; the further one is above for the upper output row, below for the lower; the image edge
  // duplicates the first / last real row (jdmainct.c context rows)
  int jo = (y & 1) ? j + 1 : j - 1;
  jo = jo < 0 ? 0 : (jo > c.dh - 1 ? c.dh - 1 : jo);
  const uint8_t *r0 = c.p + (size_t)j * c.stride, *r1 = c.p + (size_t)jo * c.stride;
  if (c.mode == 3) return (r0[x] * 3 + r1[x] + ((y & 1) ? 2 : 1)) >> 2;      // h1v2_fancy_upsample
  const int i = x >> 1;                                                      // h2v2_fancy_upsample
  const int cs = r0[i] * 3 + r1[i];
  if (x & 1) {
    if (i == c.dw - 1) return (cs * 4 + 7) >> 4;
    return (cs * 3 + (r0[i + 1] * 3 + r1[i + 1]) + 7) >> 4;
  }
  if (i == 0) return (cs * 4 + 8) >> 4;
  return (cs * 3 + (r0[i - 1] * 3 + r1[i - 1]) + 8) >> 4;
}

YMI_HD int clamp255(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }


// IDCT_range_limit = sample_range_limit + CENTERJSAMPLE, indexed by (x & RANGE_MASK), RANGE_MASK = 1023
YMI_HD int idct_range_limit(long v) {
  const int x = (int)(v & 1023);
  return x < 128 ? x + 128 : (x < 512 ? 255 : (x < 896 ? 0 : x - 896));
}

// one pixel: upsample every component, convert; returns b | g << 8 | r << 16
YMI_HD uint32_t pixel_bgr(const ColorArgs &a, int x, int y) {
  int b, g, r;
  if (a.ncomp == 1) {
    b = g = r = sample_at(a.c[0], x, y);
  } else {
    const int c0 = sample_at(a.c[0], x, y), c1 = sample_at(a.c[1], x, y), c2 = sample_at(a.c[2], x, y);
    if (a.color == 1 /* YMI_JPEG_RGB */) {
      r = c0; g = c1; b = c2;
    } else {   // jdcolor.c: Cr_r_tab, Cb_b_tab, Cb_g_tab + Cr_g_tab, SCALEBITS 16, ONE_HALF folded into the Cb table
      const int cb = c1 - 128, cr = c2 - 128;
      r = clamp255(c0 + ((91881 * cr + 32768) >> 16));
      b = clamp255(c0 + ((116130 * cb + 32768) >> 16));
      g = clamp255(c0 + ((-22554 * cb + 32768 - 46802 * cr) >> 16));
    }
  }
  return (uint32_t)b | ((uint32_t)g << 8) | ((uint32_t)r << 16);
}

// source pixel (x, y) of a W x H image -> destination (ox, oy) after the EXIF orientation
YMI_HD void orient(int o, int W, int H, int x, int y, int &ox, int &oy) {
  switch (o) {
    case 2: ox = W - 1 - x; oy = y; break;
    case 3: ox = W - 1 - x; oy = H - 1 - y; break;
    case 4: ox = x; oy = H - 1 - y; break;
    case 5: ox = y; oy = x; break;
    case 6: ox = H - 1 - y; oy = x; break;
    case 7: ox = H - 1 - y; oy = W - 1 - x; break;
    case 8: ox = y; oy = W - 1 - x; break;
    default: ox = x; oy = y; break;
  }
}

// upsampling mode of a component (jdsample.c jinit_upsampler: fancy h2v1 / h2v2 only when downsampled_width > 2)
YMI_HD int upsample_mode(int hf, int vf, int dw) {
  if (hf == 1 && vf == 1) return 0;
  if (hf == 2 && vf == 1 && dw > 2) return 1;
  if (hf == 2 && vf == 2 && dw > 2) return 2;
  if (hf == 1 && vf == 2) return 3;
  return 4;
}

// one 8x8 block: quantised coefficients (natural order) -> 64 samples, row-major (the two passes of jpeg_idct_islow)
YMI_HD void idct_block(const int16_t *coef, const uint16_t *qt, uint8_t *out64) {
  int ws[64];
  for (int c = 0; c < 8; ++c) {
    long in[8], o[8];
    for (int r = 0; r < 8; ++r) in[r] = (int)coef[r * 8 + c] * (int)qt[r * 8 + c];
    idct8(in, o);
    for (int r = 0; r < 8; ++r) ws[r * 8 + c] = (int)descale(o[r], 13 - 2);
  }
  for (int r = 0; r < 8; ++r) {
    long in[8], o[8];
    for (int c = 0; c < 8; ++c) in[c] = ws[r * 8 + c];
    idct8(in, o);
    for (int c = 0; c < 8; ++c) out64[r * 8 + c] = (uint8_t)idct_range_limit(descale(o[c], 13 + 2 + 3));
  }
}

}  // namespace ymi_jpeg
