// JPEG reconstruction on the GPU: the data-parallel half of cv2.imread (COCODetection.pull_item, data/coco.py:138-141).
// Input: the quantised coefficient blocks the host entropy decoder produced (jpeg_host.cpp).  Two kernels:
//   jpeg_idct_k   dequantise + libjpeg's ISLOW integer IDCT (jidctint.c jpeg_idct_islow: CONST_BITS 13, PASS1_BITS 2) +
//                 range limit -> one uint8 plane per component;
//   jpeg_color_k  "fancy" triangle-filter chroma upsampling (jdsample.c h2v1 / h2v2 / h1v2, replication otherwise),
//                 YCbCr -> RGB by the SCALEBITS-16 fixed-point tables (jdcolor.c), EXIF orientation, store as BGR HWC.
// Integer / byte work, bit-exact against libjpeg-turbo (tests pin it through the oracle).  HBM-bound and tiny: a 640x480
// image is 0.9 MB of coefficients in and 0.9 MB of pixels out.
#include "common.h"
#include "../../include/yolact_amd.h"

namespace {

constexpr int F0_298 = 2446, F0_390 = 3196, F0_541 = 4433, F0_765 = 6270, F0_899 = 7373, F1_175 = 9633, F1_501 = 12299,
              F1_847 = 15137, F1_961 = 16069, F2_053 = 16819, F2_562 = 20995, F3_072 = 25172;

// one 1-D pass of jidctint.c on 8 values; out[k] are the un-descaled sums
__device__ __forceinline__ void idct8(const long in[8], long out[8]) {
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

__device__ __forceinline__ long descale(long x, int n) { return (x + (1L << (n - 1))) >> n; }

// 256 threads = 32 JPEG blocks x 8 threads.  coef [nblk][64] int16 natural order, qt [64], plane rows of bw*8 bytes.
__global__ __launch_bounds__(256) void jpeg_idct_k(const int16_t *__restrict__ coef, const uint16_t *__restrict__ qt,
                                                   uint8_t *__restrict__ plane, int bw, int nblk) {
  __shared__ int s_in[32][8][9];      // [block][row][col], padded
  __shared__ int s_ws[32][8][9];
  const int lb = threadIdx.x >> 3, t = threadIdx.x & 7;
  const int blk = blockIdx.x * 32 + lb;
  const bool live = blk < nblk;
  if (live) {   // row t of the block: 8 int16 = 16 bytes
    const uint4 raw = *reinterpret_cast<const uint4 *>(coef + (size_t)blk * 64 + t * 8);
    const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int v = (int)(int16_t)((w[c >> 1] >> (16 * (c & 1))) & 0xFFFF);
      s_in[lb][t][c] = v * (int)qt[t * 8 + c];
    }
  }
  __syncthreads();
  if (live) {   // pass 1: column t
    long in[8], out[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) in[r] = s_in[lb][r][t];
    idct8(in, out);
#pragma unroll
    for (int r = 0; r < 8; ++r) s_ws[lb][r][t] = (int)descale(out[r], 13 - 2);
  }
  __syncthreads();
  if (live) {   // pass 2: row t
    long in[8], out[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) in[c] = s_ws[lb][t][c];
    idct8(in, out);
    uint32_t pk[2] = {0, 0};
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      // IDCT_range_limit = sample_range_limit + CENTERJSAMPLE, indexed by (x & RANGE_MASK), RANGE_MASK = 1023
      const int x = (int)(descale(out[c], 13 + 2 + 3) & 1023);
      const int v = x < 128 ? x + 128 : (x < 512 ? 255 : (x < 896 ? 0 : x - 896));
      pk[c >> 2] |= (uint32_t)v << (8 * (c & 3));
    }
    const int by = blk / bw, bx = blk - by * bw;
    *reinterpret_cast<uint2 *>(plane + ((size_t)(by * 8 + t) * bw + bx) * 8) = make_uint2(pk[0], pk[1]);
  }
}

struct CompPlane {
  const uint8_t *p;
  int stride, dw, dh, hf, vf, mode;      // mode: 0 full size, 1 h2v1 fancy, 2 h2v2 fancy, 3 h1v2 fancy, 4 replicate
};
struct ColorArgs {
  CompPlane c[3];
  int ncomp, color, W, H, orientation, out_w;
};

__device__ __forceinline__ int sample_at(const CompPlane &c, int x, int y) {
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

__device__ __forceinline__ int clamp255(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

__global__ __launch_bounds__(256) void jpeg_color_k(const ColorArgs a, uint8_t *__restrict__ out) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= a.W || y >= a.H) return;
  int b, g, r;
  if (a.ncomp == 1) {
    b = g = r = sample_at(a.c[0], x, y);
  } else {
    const int c0 = sample_at(a.c[0], x, y), c1 = sample_at(a.c[1], x, y), c2 = sample_at(a.c[2], x, y);
    if (a.color == YMI_JPEG_RGB) {
      r = c0; g = c1; b = c2;
    } else {   // jdcolor.c: Cr_r_tab, Cb_b_tab, Cb_g_tab + Cr_g_tab, SCALEBITS 16, ONE_HALF folded into the Cb table
      const int cb = c1 - 128, cr = c2 - 128;
      r = clamp255(c0 + ((91881 * cr + 32768) >> 16));
      b = clamp255(c0 + ((116130 * cb + 32768) >> 16));
      g = clamp255(c0 + ((-22554 * cb + 32768 - 46802 * cr) >> 16));
    }
  }
  int ox, oy;
  switch (a.orientation) {
    case 2: ox = a.W - 1 - x; oy = y; break;
    case 3: ox = a.W - 1 - x; oy = a.H - 1 - y; break;
    case 4: ox = x; oy = a.H - 1 - y; break;
    case 5: ox = y; oy = x; break;
    case 6: ox = a.H - 1 - y; oy = x; break;
    case 7: ox = a.H - 1 - y; oy = a.W - 1 - x; break;
    case 8: ox = y; oy = a.W - 1 - x; break;
    default: ox = x; oy = y; break;
  }
  uint8_t *o = out + ((size_t)oy * a.out_w + ox) * 3;
  o[0] = (uint8_t)b; o[1] = (uint8_t)g; o[2] = (uint8_t)r;
}

}  // namespace

extern "C" int ymi_jpeg_reconstruct_bgr_u8(const ymi_jpeg_info *info, const int16_t *coefs, const uint16_t *qt,
                                           uint8_t *planes_ws, uint8_t *out, void *stream) {
  if (!info || !coefs || !qt || !planes_ws || !out) return YMI_ENULL;
  if ((info->ncomp != 1 && info->ncomp != 3) || info->width <= 0 || info->height <= 0) return YMI_EARG;
  if (info->orientation < 1 || info->orientation > 8) return YMI_EARG;
  hipStream_t s = (hipStream_t)stream;
  ColorArgs a;
  a.ncomp = info->ncomp; a.color = info->color; a.W = info->width; a.H = info->height;
  a.orientation = info->orientation;
  a.out_w = info->orientation >= 5 ? info->height : info->width;
  if (info->out_width != a.out_w) return YMI_EARG;
  size_t off = 0;
  for (int i = 0; i < info->ncomp; ++i) {
    const int bw = info->bw[i], bh = info->bh[i], hf = info->hf[i], vf = info->vf[i];
    if (bw <= 0 || bh <= 0 || hf < 1 || hf > 4 || vf < 1 || vf > 4) return YMI_EARG;
    if (info->dw[i] <= 0 || info->dh[i] <= 0 || info->dw[i] > bw * 8 || info->dh[i] > bh * 8) return YMI_EARG;
    if ((long)info->dw[i] * hf < info->width || (long)info->dh[i] * vf < info->height) return YMI_EARG;
    const int nblk = bw * bh;
    hipLaunchKernelGGL(jpeg_idct_k, dim3((nblk + 31) / 32), dim3(256), 0, s, coefs + off, qt + 64 * i, planes_ws + off, bw,
                       nblk);
    CompPlane &c = a.c[i];
    c.p = planes_ws + off; c.stride = bw * 8; c.dw = info->dw[i]; c.dh = info->dh[i]; c.hf = hf; c.vf = vf;
    // jdsample.c jinit_upsampler: fancy h2v1 / h2v2 only when downsampled_width > 2
    if (hf == 1 && vf == 1) c.mode = 0;
    else if (hf == 2 && vf == 1 && c.dw > 2) c.mode = 1;
    else if (hf == 2 && vf == 2 && c.dw > 2) c.mode = 2;
    else if (hf == 1 && vf == 2) c.mode = 3;
    else c.mode = 4;
    off += (size_t)nblk * 64;
  }
  if ((int64_t)off != info->coef_count) return YMI_EARG;
  hipLaunchKernelGGL(jpeg_color_k, dim3((a.W + 63) / 64, (a.H + 3) / 4), dim3(256), 0, s, a, out);
  return ymi_launch_status();
}
