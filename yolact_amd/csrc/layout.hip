// Bandwidth kernels around the conv engine: layout change, max-pool, bilinear resize (NHWC, float4 lanes).
#include "common.h"
#include "../../include/yolact_amd.h"

namespace {

// x [B,C,H,W] (C <= 4) -> y [B,H,W,4]; one thread per pixel, plane reads coalesced, 16-byte stores.
// AMAX: also raise *amax to max |element| (the magnitude bound of the network input, ymi_conv_desc.x_amax of the stem) — the
// values pass through registers here anyway, a separate ymi_amax_f32 launch re-reads them (10 us of a 1.6 ms batch-1 step)
template <bool AMAX>
__global__ __launch_bounds__(256) void nchw_to_nhwc4_k(const float *__restrict__ x, float *__restrict__ y,
                                                        int C, int HW, long total, float *__restrict__ amax) {
  float am = 0.f;
  ymi_amax_pre apre = {};
  if (AMAX) apre = ymi_amax_prefetch(amax);
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
    const long b = i / HW, pix = i - b * HW;
    const float *src = x + b * C * HW + pix;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (C > 0) v[0] = src[0];
    if (C > 1) v[1] = src[HW];
    if (C > 2) v[2] = src[2L * HW];
    if (C > 3) v[3] = src[3L * HW];
    *reinterpret_cast<f32x4 *>(y + i * 4) = v;
    if (AMAX) am = fmaxf(am, ymi_absmax4(v));
  }
  if (AMAX) ymi_amax_finish(apre, am);
}

// x [B,H,W,C] -> y [B,C,H,W] through a 32x33 LDS tile (pixels x channels) so both sides coalesce.
__global__ __launch_bounds__(256) void nhwc_to_nchw_k(const float *__restrict__ x, float *__restrict__ y,
                                                       int C, int HW) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int p = p0 + r, c = c0 + tx;
    tile[r][tx] = (p < HW && c < C) ? x[((long)b * HW + p) * C + c] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r, p = p0 + tx;
    if (p < HW && c < C) y[((long)b * C + c) * HW + p] = tile[tx][r];
  }
}

// MaxPool 3x3 / stride 2 / pad 1, -inf padding (nn.MaxPool2d semantics). One thread per (pixel, 4 channels).
__global__ __launch_bounds__(256) void maxpool3x3s2_k(const float *__restrict__ x, float *__restrict__ y,
                                                       int H, int W, int C4, int Ho, int Wo, long total) {
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < (unsigned)total; i += gridDim.x * 256u) {   // 32-bit index math (total <
    const unsigned pu = i / (unsigned)C4, ru = pu / (unsigned)Wo, bu = ru / (unsigned)Ho;          // 2^31, host-checked): the 64-bit
    const int c4 = (int)(i - pu * (unsigned)C4), ox = (int)(pu - ru * (unsigned)Wo), oy = (int)(ru - bu * (unsigned)Ho);   // div / mod
    const long b = bu;                                                                          // chain cost more than the pooling
    const float ninf = -__builtin_inff();
    f32x4 m = {ninf, ninf, ninf, ninf};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = oy * 2 - 1 + ky;
      if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = ox * 2 - 1 + kx;
        if ((unsigned)ix >= (unsigned)W) continue;
        const f32x4 v = *reinterpret_cast<const f32x4 *>(x + (((b * H + iy) * W + ix) * C4 + c4) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) m[e] = v[e] > m[e] ? v[e] : m[e];
      }
    }
    *reinterpret_cast<f32x4 *>(y + (long)i * 4) = m;
  }
}

__device__ __forceinline__ void bl_coord(int dst, float scale, int in_size, int &i0, int &i1, float &l1) {
  float src = scale * ((float)dst + 0.5f) - 0.5f;
  src = src < 0.f ? 0.f : src;
  i0 = (int)src;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + ((i0 < in_size - 1) ? 1 : 0);
  l1 = src - (float)i0;
}

// F.interpolate(bilinear, align_corners=False) on NHWC; thread per (out pixel, 4 channels).
__global__ __launch_bounds__(256) void bilinear_nhwc_k(const float *__restrict__ x, float *__restrict__ y,
                                                        int Hi, int Wi, int C4, int Ho, int Wo, float sh, float sw,
                                                        int relu, long total) {
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < (unsigned)total; i += gridDim.x * 256u) {   // 32-bit index math, see above
    const unsigned pu = i / (unsigned)C4, ru = pu / (unsigned)Wo, bu = ru / (unsigned)Ho;
    const int c4 = (int)(i - pu * (unsigned)C4), ox = (int)(pu - ru * (unsigned)Wo), oy = (int)(ru - bu * (unsigned)Ho);
    const long b = bu;
    int y0, y1, x0, x1; float ly, lx;
    bl_coord(oy, sh, Hi, y0, y1, ly);
    bl_coord(ox, sw, Wi, x0, x1, lx);
    const float *img = x + (b * Hi * Wi * C4 + c4) * 4;
    const f32x4 v00 = *reinterpret_cast<const f32x4 *>(img + (long)(y0 * Wi + x0) * C4 * 4);
    const f32x4 v01 = *reinterpret_cast<const f32x4 *>(img + (long)(y0 * Wi + x1) * C4 * 4);
    const f32x4 v10 = *reinterpret_cast<const f32x4 *>(img + (long)(y1 * Wi + x0) * C4 * 4);
    const f32x4 v11 = *reinterpret_cast<const f32x4 *>(img + (long)(y1 * Wi + x1) * C4 * 4);
    const float hy = 1.f - ly, hx = 1.f - lx;
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v = hy * (hx * v00[e] + lx * v01[e]) + ly * (hx * v10[e] + lx * v11[e]);
      o[e] = (relu && v < 0.f) ? 0.f : v;
    }
    *reinterpret_cast<f32x4 *>(y + (long)i * 4) = o;
  }
}

// FPN top-down sum as its own pass (yolact.py:332-334: x = F.interpolate(x, size=(h, w), mode=bilinear) + lat_layer(convout)):
// y[b,oy,ox,:] += bilinear(x -> Ho x Wo)[b,oy,ox,:] IN PLACE, the interpolation in the epilogue form of csrc/conv_igemm.hip
// (YMI_RES_BILINEAR: same coordinates, same expression), so lateral-conv launch + this pass == the fused launch.  Raises
// the magnitude-bound slot of the SUM (fp16x2 consumers scale by it).
__global__ __launch_bounds__(256) void bilinear_add_k(const float *__restrict__ x, float *__restrict__ y, int Hi, int Wi, int C4, int Ho,
                                                       int Wo, float sh, float sw, long total, float *__restrict__ amax) {
  const ymi_amax_pre apre = ymi_amax_prefetch(amax);
  float am = 0.f;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < (unsigned)total; i += gridDim.x * 256u) {
    const unsigned pu = i / (unsigned)C4, ru = pu / (unsigned)Wo, bu = ru / (unsigned)Ho;
    const int c4 = (int)(i - pu * (unsigned)C4), ox = (int)(pu - ru * (unsigned)Wo), oy = (int)(ru - bu * (unsigned)Ho);
    const long b = bu;
    int y0, y1, x0, x1; float ly, lx;
    bl_coord(oy, sh, Hi, y0, y1, ly);
    bl_coord(ox, sw, Wi, x0, x1, lx);
    const float *img = x + (b * Hi * Wi * C4 + c4) * 4;
    const f32x4 v00 = *reinterpret_cast<const f32x4 *>(img + (long)(y0 * Wi + x0) * C4 * 4);
    const f32x4 v01 = *reinterpret_cast<const f32x4 *>(img + (long)(y0 * Wi + x1) * C4 * 4);
    const f32x4 v10 = *reinterpret_cast<const f32x4 *>(img + (long)(y1 * Wi + x0) * C4 * 4);
    const f32x4 v11 = *reinterpret_cast<const f32x4 *>(img + (long)(y1 * Wi + x1) * C4 * 4);
    const f32x4 rv = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
    f32x4 *dst = reinterpret_cast<f32x4 *>(y + (long)i * 4);
    const f32x4 o = *dst + rv;
    *dst = o;
    am = fmaxf(am, ymi_absmax4(o));
  }
  if (amax) ymi_amax_finish(apre, am);
}

inline int grid_for(long total) {
  long g = (total + 255) / 256;
  const long cap = 256L * 8;  // 8 blocks per CU, grid-stride the rest
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

namespace {
// *out = max(*out, max |x|): one float4 per thread and trip, a wave reduction and one integer atomic per wave (non-negative
// floats order like their bit patterns).  Feeds ymi_conv_desc.x_amax for tensors no conv launch produced (the network input).
__global__ __launch_bounds__(256) void amax_k(const float *__restrict__ x, long n4, float *__restrict__ out) {
  float am = 0.f;
  const ymi_amax_pre apre = ymi_amax_prefetch(out);
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256L)
    am = fmaxf(am, ymi_absmax4(*reinterpret_cast<const f32x4 *>(x + 4 * i)));
  ymi_amax_finish(apre, am);
}
}  // namespace

extern "C" {

int ymi_amax_f32(const float *x, long n, float *out, void *stream) {
  if (!x || !out) return YMI_ENULL;
  if (n <= 0 || (n & 3) || (((uintptr_t)x) & 15)) return YMI_ESHAPE;
  long g = (n / 4 + 255) / 256;
  hipLaunchKernelGGL(amax_k, dim3((unsigned)(g > 2048 ? 2048 : g)), dim3(256), 0, (hipStream_t)stream, x, n / 4, out);
  return ymi_launch_status();
}

int ymi_nchw_to_nhwc4_f32(const float *x, float *y, int B, int C, int H, int W, void *stream) {
  if (!x || !y) return YMI_ENULL;
  if (B <= 0 || C <= 0 || C > 4 || H <= 0 || W <= 0) return YMI_EARG;
  const long total = (long)B * H * W;
  hipLaunchKernelGGL(nchw_to_nhwc4_k<false>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, y, C, H * W, total,
                     (float *)nullptr);
  return ymi_launch_status();
}

int ymi_nchw_to_nhwc4_amax_f32(const float *x, float *y, int B, int C, int H, int W, float *amax, void *stream) {
  if (!x || !y || !amax) return YMI_ENULL;
  if (B <= 0 || C <= 0 || C > 4 || H <= 0 || W <= 0) return YMI_EARG;
  const long total = (long)B * H * W;
  hipLaunchKernelGGL(nchw_to_nhwc4_k<true>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, y, C, H * W, total, amax);
  return ymi_launch_status();
}

int ymi_nhwc_to_nchw_f32(const float *x, float *y, int B, int C, int H, int W, void *stream) {
  if (!x || !y) return YMI_ENULL;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || B > 65535) return YMI_EARG;
  const int HW = H * W;
  dim3 grid((HW + 31) / 32, (C + 31) / 32, B);
  if (grid.y > 65535) return YMI_EARG;
  hipLaunchKernelGGL(nhwc_to_nchw_k, grid, dim3(256), 0, (hipStream_t)stream, x, y, C, HW);
  return ymi_launch_status();
}

int ymi_maxpool3x3s2_nhwc_f32(const float *x, float *y, int B, int H, int W, int C, int Ho, int Wo, void *stream) {
  if (!x || !y) return YMI_ENULL;
  if (C % 4 != 0 || B <= 0) return YMI_ESHAPE;
  if (Ho != (H + 2 - 3) / 2 + 1 || Wo != (W + 2 - 3) / 2 + 1) return YMI_ESHAPE;
  const long total = (long)B * Ho * Wo * (C / 4);
  if (total >= (1L << 31)) return YMI_ESHAPE;
  hipLaunchKernelGGL(maxpool3x3s2_k, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, y, H, W, C / 4, Ho,
                     Wo, total);
  return ymi_launch_status();
}

int ymi_bilinear_nhwc_f32(const float *x, float *y, int B, int Hi, int Wi, int C, int Ho, int Wo, float scale_h,
                          float scale_w, int relu, void *stream) {
  if (!x || !y) return YMI_ENULL;
  if (C % 4 != 0 || B <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0) return YMI_ESHAPE;
  const float sh = scale_h > 0.f ? scale_h : (float)Hi / (float)Ho;
  const float sw = scale_w > 0.f ? scale_w : (float)Wi / (float)Wo;
  const long total = (long)B * Ho * Wo * (C / 4);
  if (total >= (1L << 31)) return YMI_ESHAPE;
  hipLaunchKernelGGL(bilinear_nhwc_k, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, y, Hi, Wi, C / 4,
                     Ho, Wo, sh, sw, relu, total);
  return ymi_launch_status();
}

int ymi_bilinear_add_nhwc_f32(const float *x, float *y, int B, int Hi, int Wi, int C, int Ho, int Wo, float *y_amax, void *stream) {
  if (!x || !y) return YMI_ENULL;
  if (C % 4 != 0 || B <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0 || ((((uintptr_t)x) | ((uintptr_t)y)) & 15)) return YMI_ESHAPE;
  const long total = (long)B * Ho * Wo * (C / 4);
  if (total >= (1L << 31)) return YMI_ESHAPE;
  hipLaunchKernelGGL(bilinear_add_k, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, y, Hi, Wi, C / 4, Ho, Wo,
                     (float)Hi / (float)Ho, (float)Wi / (float)Wo, total, y_amax);
  return ymi_launch_status();
}

}  // extern "C"

// ---- small direct convolution (FastMaskIoUNet, yolact.py:363-375: Cin = 1..64, tiny spatial sizes) ----------
namespace {

// One thread per (pixel, 4 output channels); weights [K][CoutPad4] with k = (ky*kw+kx)*Cin + c so the four
// output channels of a k are one float4.  ~9 MFLOP per detection: bandwidth/latency, not matrix-core, work.
__global__ __launch_bounds__(256) void conv_direct_k(const float *__restrict__ x, const float *__restrict__ w,
                                                      const float *__restrict__ bias, float *__restrict__ y, int H,
                                                      int W, int Cin, int Ho, int Wo, int Cout, int Co4, int kh, int kw,
                                                      int stride, int pad, int relu, long total) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
    const int c4 = (int)(i % Co4);
    long r = i / Co4;
    const int ox = (int)(r % Wo); r /= Wo;
    const int oy = (int)(r % Ho);
    const long b = r / Ho;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int ky = 0; ky < kh; ++ky) {
      const int iy = oy * stride - pad + ky;
      if ((unsigned)iy >= (unsigned)H) continue;
      for (int kx = 0; kx < kw; ++kx) {
        const int ix = ox * stride - pad + kx;
        if ((unsigned)ix >= (unsigned)W) continue;
        const float *xp = x + ((b * H + iy) * W + ix) * Cin;
        const float *wp = w + ((long)(ky * kw + kx) * Cin) * (Co4 * 4) + c4 * 4;
        for (int c = 0; c < Cin; ++c) {
          const f32x4 wv = *reinterpret_cast<const f32x4 *>(wp + (long)c * (Co4 * 4));
          acc += xp[c] * wv;
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int co = c4 * 4 + e;
      if (co < Cout) {
        float v = acc[e] + (bias ? bias[co] : 0.f);
        if (relu && v < 0.f) v = 0.f;
        y[((b * Ho + oy) * Wo + ox) * Cout + co] = v;
      }
    }
  }
}

// The first layers of FastMaskIoUNet at BATCH scale (round 5: postprocess_batch runs the net once over all B x cap masks — 800 masks of
// 138 x 138 at batch 8): Cin x Cout = 1 x 8, 8 x 16, 16 x 32, 3x3 / stride 2 / unpadded.  conv_direct_k above spends a thread per
// (pixel, 4 output channels) with scalar activation loads and run-time loops: 150 us per layer.  Here a thread owns ONE output
// pixel and ALL its output channels (<= 32 accumulators in registers), the filters sit in LDS as [tap][cin][cout] and are read as
// wave-uniform broadcasts, the pixel's input channels arrive as float4s: the layers become the bandwidth streams they are.
template <int CIN, int COUT>
__global__ __launch_bounds__(256) void conv_direct_small_k(const float *__restrict__ x, const float *__restrict__ w,
                                                            const float *__restrict__ bias, float *__restrict__ y, int H, int W, int Ho,
                                                            int Wo, int kh, int kw, int stride, int pad, int relu, unsigned total) {
  extern __shared__ __attribute__((aligned(16))) float wl[];
  const int nw = kh * kw * CIN * COUT;
  for (int i = threadIdx.x; i < nw; i += 256) wl[i] = w[i];
  __syncthreads();
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const unsigned r = i / (unsigned)Wo, bu = r / (unsigned)Ho;
    const int ox = (int)(i - r * (unsigned)Wo), oy = (int)(r - bu * (unsigned)Ho);
    f32x4 acc[COUT / 4];
#pragma unroll
    for (int n = 0; n < COUT / 4; ++n)
      acc[n] = bias ? *reinterpret_cast<const f32x4 *>(bias + 4 * n) : f32x4{0.f, 0.f, 0.f, 0.f};
    for (int ky = 0; ky < kh; ++ky) {
      const int iy = oy * stride - pad + ky;
      if ((unsigned)iy >= (unsigned)H) continue;
      for (int kx = 0; kx < kw; ++kx) {
        const int ix = ox * stride - pad + kx;
        if ((unsigned)ix >= (unsigned)W) continue;
        const float *xp = x + ((size_t)(bu * (unsigned)H + (unsigned)iy) * (unsigned)W + (unsigned)ix) * CIN;
        float xv[CIN];
        if constexpr (CIN % 4 == 0) {
#pragma unroll
          for (int c = 0; c < CIN / 4; ++c) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(xp + 4 * c);
            xv[4 * c] = v[0]; xv[4 * c + 1] = v[1]; xv[4 * c + 2] = v[2]; xv[4 * c + 3] = v[3];
          }
        } else {
#pragma unroll
          for (int c = 0; c < CIN; ++c) xv[c] = xp[c];
        }
        const float *wt = wl + (ky * kw + kx) * CIN * COUT;
#pragma unroll
        for (int c = 0; c < CIN; ++c)
#pragma unroll
          for (int n = 0; n < COUT / 4; ++n) {
            const f32x4 wv = *reinterpret_cast<const f32x4 *>(wt + c * COUT + 4 * n);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[n][e] = __builtin_fmaf(xv[c], wv[e], acc[n][e]);
          }
      }
    }
    float *yp = y + (size_t)i * COUT;
#pragma unroll
    for (int n = 0; n < COUT / 4; ++n) {
      f32x4 v = acc[n];
      if (relu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] < 0.f ? 0.f : v[e];
      }
      *reinterpret_cast<f32x4 *>(yp + 4 * n) = v;
    }
  }
}

// y[b,c] = max over the HW positions of x[b,:,c]   (F.max_pool2d with kernel = full map)
__global__ void global_max_k(const float *__restrict__ x, float *__restrict__ y, int HW, int C, long total) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= total) return;
  const long b = i / C; const int c = (int)(i - b * C);
  float m = -__builtin_inff();
  for (int p = 0; p < HW; ++p) { const float v = x[(b * HW + p) * C + c]; m = v > m ? v : m; }
  y[i] = m;
}

}  // namespace

extern "C" {

int ymi_conv2d_direct_nhwc_f32(const float *x, const float *w, const float *bias, float *y, int B, int H, int W, int Cin,
                               int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad, int relu, void *stream) {
  if (!x || !w || !y) return YMI_ENULL;
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || kh <= 0 || kw <= 0 || stride <= 0) return YMI_EARG;
  if (Ho != (H + 2 * pad - kh) / stride + 1 || Wo != (W + 2 * pad - kw) / stride + 1) return YMI_ESHAPE;
  const int Co4 = (Cout + 3) / 4;
  {   // the pixel-per-thread kernel for the narrow first layers of FastMaskIoUNet (filters in LDS; everything 16-byte aligned)
    const long px = (long)B * Ho * Wo;
    const size_t lds = (size_t)kh * kw * Cin * Cout * sizeof(float);
    const bool aligned = !((((uintptr_t)x) | ((uintptr_t)w) | ((uintptr_t)y) | ((uintptr_t)bias)) & 15);
    if (aligned && px < (1L << 31) && lds <= 48 * 1024 && (long)B * H * W * Cin < (1L << 32)) {
      const int g = grid_for(px);
#define YMI_SMALL(CI, CO)                                                                                                        \
  if (Cin == CI && Cout == CO) {                                                                                                 \
    hipLaunchKernelGGL((conv_direct_small_k<CI, CO>), dim3(g), dim3(256), lds, (hipStream_t)stream, x, w, bias, y, H, W, Ho, Wo, kh, \
                       kw, stride, pad, relu, (unsigned)px);                                                                     \
    return ymi_launch_status();                                                                                                  \
  }
      YMI_SMALL(1, 8)
      YMI_SMALL(8, 16)
      YMI_SMALL(16, 32)
#undef YMI_SMALL
    }
  }
  const long total = (long)B * Ho * Wo * Co4;
  hipLaunchKernelGGL(conv_direct_k, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, w, bias, y, H, W, Cin,
                     Ho, Wo, Cout, Co4, kh, kw, stride, pad, relu, total);
  return ymi_launch_status();
}

int ymi_global_maxpool_nhwc_f32(const float *x, float *y, int B, int HW, int C, void *stream) {
  if (!x || !y) return YMI_ENULL;
  if (B <= 0 || HW <= 0 || C <= 0) return YMI_EARG;
  const long total = (long)B * C;
  hipLaunchKernelGGL(global_max_k, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, HW, C,
                     total);
  return ymi_launch_status();
}

}  // extern "C"
