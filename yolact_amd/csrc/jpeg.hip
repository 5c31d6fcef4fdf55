// JPEG reconstruction on the GPU: the data-parallel half of cv2.imread (COCODetection.pull_item, data/coco.py:138-141).
// Input: the quantised coefficient blocks the host entropy decoder produced (jpeg_host.cpp).  Two kernels:
//   jpeg_idct_k   dequantise + libjpeg's ISLOW integer IDCT (jidctint.c jpeg_idct_islow: CONST_BITS 13, PASS1_BITS 2) +
//                 range limit -> one uint8 plane per component;
//   jpeg_color_k  "fancy" triangle-filter chroma upsampling (jdsample.c h2v1 / h2v2 / h1v2, replication otherwise),
//                 YCbCr -> RGB by the SCALEBITS-16 fixed-point tables (jdcolor.c), EXIF orientation, store as BGR HWC.
// The arithmetic lives in jpeg_math.h (shared with the g++-built host emulation of the CPU tests).  Integer / byte work,
// bit-exact against libjpeg-turbo.  HBM-bound and tiny: a 640x480 image is 0.9 MB of coefficients in, 0.9 MB of pixels out.
#include "common.h"
#include "../../include/yolact_amd.h"
#include "jpeg_math.h"

namespace {

using namespace ymi_jpeg;

// 256 threads = 32 JPEG blocks x 8 threads.  coef [nblk][64] int16 natural order, qt [64], plane rows of bw*8 bytes.
// Thread t of a block: loads row t (one 16-byte load), runs column t of pass 1 and row t of pass 2, stores 8 bytes.
__global__ __launch_bounds__(256) void jpeg_idct_k(const int16_t *__restrict__ coef, const uint16_t *__restrict__ qt,
                                                   uint8_t *__restrict__ plane, int bw, int nblk) {
  __shared__ int s_in[32][8][9];      // [block][row][col], padded
  __shared__ int s_ws[32][8][9];
  const int lb = threadIdx.x >> 3, t = threadIdx.x & 7;
  const int blk = blockIdx.x * 32 + lb;
  const bool live = blk < nblk;
  if (live) {
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
    for (int c = 0; c < 8; ++c)
      pk[c >> 2] |= (uint32_t)idct_range_limit(descale(out[c], 13 + 2 + 3)) << (8 * (c & 3));
    const int by = blk / bw, bx = blk - by * bw;
    *reinterpret_cast<uint2 *>(plane + ((size_t)(by * 8 + t) * bw + bx) * 8) = make_uint2(pk[0], pk[1]);
  }
}

// one thread per source pixel; a block = 64 x 4 pixels
__global__ __launch_bounds__(256) void jpeg_color_k(const ColorArgs a, uint8_t *__restrict__ out) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= a.W || y >= a.H) return;
  const uint32_t v = pixel_bgr(a, x, y);
  int ox, oy;
  orient(a.orientation, a.W, a.H, x, y, ox, oy);
  uint8_t *o = out + ((size_t)oy * a.out_w + ox) * 3;
  o[0] = (uint8_t)v; o[1] = (uint8_t)(v >> 8); o[2] = (uint8_t)(v >> 16);
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
    off += (size_t)bw * bh * 64;
  }
  if ((int64_t)off != info->coef_count || info->plane_bytes < (int64_t)off) return YMI_EARG;
  off = 0;
  for (int i = 0; i < info->ncomp; ++i) {
    const int bw = info->bw[i], nblk = bw * info->bh[i];
    hipLaunchKernelGGL(jpeg_idct_k, dim3((nblk + 31) / 32), dim3(256), 0, s, coefs + off, qt + 64 * i, planes_ws + off, bw,
                       nblk);
    CompPlane &c = a.c[i];
    c.p = planes_ws + off; c.stride = bw * 8; c.dw = info->dw[i]; c.dh = info->dh[i]; c.hf = info->hf[i]; c.vf = info->vf[i];
    c.mode = upsample_mode(c.hf, c.vf, c.dw);
    off += (size_t)nblk * 64;
  }
  hipLaunchKernelGGL(jpeg_color_k, dim3((a.W + 63) / 64, (a.H + 3) / 4), dim3(256), 0, s, a, out);
  return ymi_launch_status();
}
