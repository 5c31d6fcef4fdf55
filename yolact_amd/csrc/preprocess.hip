// FastBaseTransform on device: the step immediately before Yolact.forward in evalimage /
// evalvideo (eval.py:596-597,692-695).  Reference: utils/augmentations.py:616-658 —
//   img [N,H,W,3] float BGR  ->  permute to NCHW  ->  F.interpolate(bilinear, align_corners=False) to (oh, ow)
//   ->  (x - mean) / std | x - mean | x / 255  (backbone.transform, data/config.py:181-202; MEANS/STD are BGR-ordered,
//   data/config.py:28-29, applied BEFORE the channel swap)  ->  BGR -> RGB.
// One kernel, one pass: the source is read once (4 taps x 12 bytes per output pixel), the result is written once either
// as the reference's NCHW [N,3,oh,ow] or directly as the engine's NHWC4 input (skipping ymi_nchw_to_nhwc4_f32).
#include "common.h"
#include "../../include/yolact_amd.h"

namespace {

__device__ __forceinline__ void src_coord(int dst, float scale, int in_size, int &i0, int &i1, float &l1) {
  // torch's area_pixel_compute_source_index, align_corners=False, fp32
  float src = scale * ((float)dst + 0.5f) - 0.5f;
  src = src < 0.f ? 0.f : src;
  i0 = (int)src;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + ((i0 < in_size - 1) ? 1 : 0);
  l1 = src - (float)i0;
}

struct FbtParams {
  const float *img;
  float *out;
  int H, W, oh, ow;
  float sh, sw;
  float mean[3], stdv[3];   // BGR order
  int mode;                 // 0 normalize, 1 subtract means, 2 to_float (/255), 3 none
  int nhwc4;
  long total;
};

__global__ __launch_bounds__(256) void fast_base_transform_k(const FbtParams p) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < p.total; i += (long)gridDim.x * 256L) {
    const int x = (int)(i % p.ow);
    long r = i / p.ow;
    const int y = (int)(r % p.oh);
    const long n = r / p.oh;
    int y0, y1, x0, x1; float ly, lx;
    src_coord(y, p.sh, p.H, y0, y1, ly);
    src_coord(x, p.sw, p.W, x0, x1, lx);
    const float *b = p.img + n * (long)p.H * p.W * 3;
    const float *p00 = b + ((long)y0 * p.W + x0) * 3, *p01 = b + ((long)y0 * p.W + x1) * 3;
    const float *p10 = b + ((long)y1 * p.W + x0) * 3, *p11 = b + ((long)y1 * p.W + x1) * 3;
    float v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      // upsample_bilinear2d: h0 * (w0 * v00 + w1 * v01) + h1 * (w0 * v10 + w1 * v11)
      float t = (1.f - ly) * ((1.f - lx) * p00[c] + lx * p01[c]) + ly * ((1.f - lx) * p10[c] + lx * p11[c]);
      if (p.mode == 0) t = (t - p.mean[c]) / p.stdv[c];
      else if (p.mode == 1) t = t - p.mean[c];
      else if (p.mode == 2) t = t / 255.f;
      v[c] = t;
    }
    // BGR -> RGB: output channel k takes input channel 2 - k
    if (p.nhwc4) {
      const f32x4 o = {v[2], v[1], v[0], 0.f};
      *reinterpret_cast<f32x4 *>(p.out + i * 4) = o;
    } else {
      const long plane = (long)p.oh * p.ow, pix = (long)y * p.ow + x;
      float *o = p.out + n * 3 * plane + pix;
      o[0] = v[2]; o[plane] = v[1]; o[2 * plane] = v[0];
    }
  }
}

}  // namespace

extern "C" int ymi_fast_base_transform_f32(const float *img, float *out, int N, int H, int W, int oh, int ow,
                                           const float *mean_bgr, const float *std_bgr, int mode, int out_nhwc4,
                                           void *stream) {
  if (!img || !out) return YMI_ENULL;
  if (N <= 0 || H <= 0 || W <= 0 || oh <= 0 || ow <= 0 || mode < 0 || mode > 3) return YMI_EARG;
  if ((mode == 0 && (!mean_bgr || !std_bgr)) || (mode == 1 && !mean_bgr)) return YMI_ENULL;
  FbtParams p;
  p.img = img; p.out = out; p.H = H; p.W = W; p.oh = oh; p.ow = ow;
  p.sh = (float)H / (float)oh; p.sw = (float)W / (float)ow;
  for (int c = 0; c < 3; ++c) { p.mean[c] = mean_bgr ? mean_bgr[c] : 0.f; p.stdv[c] = std_bgr ? std_bgr[c] : 1.f; }
  p.mode = mode; p.nhwc4 = out_nhwc4;
  p.total = (long)N * oh * ow;
  long g = (p.total + 255) / 256;
  if (g > 256L * 32) g = 256L * 32;
  hipLaunchKernelGGL(fast_base_transform_k, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, p);
  return ymi_launch_status();
}
