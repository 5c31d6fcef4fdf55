// mask_iou on device: the mAP-gate consumer of the masks in eval.py's prep_metrics
// (eval.py:376-384,435-440).  Reference: layers/box_utils.py:98-113 —
//   intersection = masks_a @ masks_b.t();  area_a = masks_a.sum(1);  area_b = masks_b.sum(1)
//   iou = intersection / (area_a + area_b - intersection)        (iscrowd: intersection / area_a)
// masks_a [A, n], masks_b [B, n] float32 (n = h*w, values 0/1 after postprocess' threshold).
// K1: split-K GEMM on the fp32 matrix cores: grid (n / KS, A / 32, B / 32); each of a block's 4 waves multiplies a 32 x 32
//     (a, b) tile over its own quarter of the K slice (free-K-order 16-byte fragments, as in the conv engine), the four
//     partial tiles are summed in LDS and added to inter[A,B] with fp32 atomics; row sums (areas) ride along.
//     For 0/1 masks every partial sum is an integer < 2^24, so the atomic accumulation is exact and order-independent.
// K2: iou = inter / (area_a + area_b - inter).
// HBM-bound: (A + B) * n * 4 bytes read once when B <= 32 (145 MB for 100 detections x 20 GT at 550 x 550).
#include "common.h"
#include "upsample_math.h"
#include "../../include/yolact_amd.h"

namespace {

constexpr int KS = 4096;   // pixels per block (1024 per wave)

__global__ __launch_bounds__(256) void mask_inter_k(const float *__restrict__ ma, const float *__restrict__ mb, int A, int B,
                                                    long n, float *__restrict__ inter, float *__restrict__ area_a,
                                                    float *__restrict__ area_b) {
  __shared__ float red[4][32][33];
  __shared__ float rs[2][4][32];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int row = lane & 31, half = lane >> 5;
  const int a0 = blockIdx.y * 32, b0 = blockIdx.z * 32;
  const long k0 = (long)blockIdx.x * KS + wave * (KS / 4);
  const bool a_ok = a0 + row < A, b_ok = b0 + row < B;
  const float *pa = ma + (size_t)(a_ok ? a0 + row : 0) * n;
  const float *pb = mb + (size_t)(b_ok ? b0 + row : 0) * n;
  const bool vec = (n & 3) == 0 && ((((uintptr_t)ma) | ((uintptr_t)mb)) & 15) == 0;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float sa = 0.f, sb = 0.f;
  for (int kk = 0; kk < KS / 4; kk += 8) {
    const long k = k0 + kk + 4 * half;            // this lane-half's 4 consecutive pixels of the 8-pixel group
    f32x4 va = {0.f, 0.f, 0.f, 0.f}, vb = va;
    if (k + 3 < n && vec) {
      if (a_ok) va = *reinterpret_cast<const f32x4 *>(pa + k);
      if (b_ok) vb = *reinterpret_cast<const f32x4 *>(pb + k);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (k + e < n) { if (a_ok) va[e] = pa[k + e]; if (b_ok) vb[e] = pb[k + e]; }
      }
    }
    sa += (va[0] + va[1]) + (va[2] + va[3]);
    sb += (vb[0] + vb[1]) + (vb[2] + vb[3]);
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(va[s], vb[s], acc, 0, 0, 0);
  }
  // C layout of the 32x32 MFMA: col = lane & 31 (b), row = (r & 3) + 8*(r >> 2) + 4*half (a)
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave][(r & 3) + 8 * (r >> 2) + 4 * half][row] = acc[r];
  // row sums: the two lane halves of a row hold disjoint pixels
  sa += __shfl_xor(sa, 32);
  sb += __shfl_xor(sb, 32);
  if (half == 0) { rs[0][wave][row] = sa; rs[1][wave][row] = sb; }
  __syncthreads();
  for (int i = t; i < 32 * 32; i += 256) {
    const int ar = i >> 5, bc = i & 31;
    const float v = (red[0][ar][bc] + red[1][ar][bc]) + (red[2][ar][bc] + red[3][ar][bc]);
    if (a0 + ar < A && b0 + bc < B && v != 0.f) atomicAdd(&inter[(size_t)(a0 + ar) * B + b0 + bc], v);
  }
  if (t < 32) {
    if (blockIdx.z == 0 && a0 + t < A) {
      const float v = (rs[0][0][t] + rs[0][1][t]) + (rs[0][2][t] + rs[0][3][t]);
      if (v != 0.f) atomicAdd(&area_a[a0 + t], v);
    }
    if (blockIdx.y == 0 && b0 + t < B) {
      const float v = (rs[1][0][t] + rs[1][1][t]) + (rs[1][2][t] + rs[1][3][t]);
      if (v != 0.f) atomicAdd(&area_b[b0 + t], v);
    }
  }
}

__global__ void mask_iou_final_k(const float *__restrict__ inter, const float *__restrict__ area_a,
                                 const float *__restrict__ area_b, float *__restrict__ iou, int A, int B, int iscrowd) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= A * B) return;
  const int a = i / B, b = i - a * B;
  const float in = inter[i];
  const float uni = iscrowd ? area_a[a] : (area_a[a] + area_b[b]) - in;   // box_utils.py:110-112
  iou[i] = in / uni;
}

}  // namespace

extern "C" int ymi_mask_iou_f32(const float *masks_a, const float *masks_b, int A, int B, long n, int iscrowd, float *ws,
                                float *iou, void *stream) {
  if (!masks_a || !masks_b || !ws || !iou) return YMI_ENULL;
  if (A <= 0 || B <= 0 || n <= 0) return YMI_EARG;
  if ((A + 31) / 32 > 65535 || (B + 31) / 32 > 65535) return YMI_EARG;
  hipStream_t s = (hipStream_t)stream;
  float *inter = ws, *area_a = ws + (size_t)A * B, *area_b = area_a + A;     // ws: A*B + A + B floats
  hipError_t e = hipMemsetAsync(ws, 0, ((size_t)A * B + A + B) * sizeof(float), s);
  if (e != hipSuccess) return (int)e;
  const long kb = (n + KS - 1) / KS;
  hipLaunchKernelGGL(mask_inter_k, dim3((unsigned)kb, (A + 31) / 32, (B + 31) / 32), dim3(256), 0, s, masks_a, masks_b, A, B,
                     n, inter, area_a, area_b);
  int rc = ymi_launch_status();
  if (rc) return rc;
  hipLaunchKernelGGL(mask_iou_final_k, dim3((A * B + 255) / 256), dim3(256), 0, s, inter, area_a, area_b, iou, A, B, iscrowd);
  return ymi_launch_status();
}

// ------------------------------------------------------------------------------------------------------------------
// prep_display's GPU half: alpha-composite the top-k instance masks onto the frame.
// Reference: eval.py:186-209 (+ :228 the uint8 conversion) —
//   img = frame / 255;  mc_j = (m_j * color_j) * alpha;  inv_j = m_j * (-alpha) + 1
//   out = img * prod_j inv_j + mc_0 + sum_{j>=1} mc_j * prod_{i<j} inv_i ;  result = uint8(out * 255)
// (= painting detection n-1 first and detection 0 last).  One pass: the frame and the n masks are read once, the
// uint8 frame is written once.
namespace {

__global__ __launch_bounds__(256) void composite_masks_k(const float *__restrict__ img, const float *__restrict__ masks,
                                                         const float *__restrict__ colors, int n, long hw, float alpha,
                                                         unsigned char *__restrict__ out) {
  for (long p = blockIdx.x * 256L + threadIdx.x; p < hw; p += (long)gridDim.x * 256L) {
    float px[3] = {img[p * 3 + 0] / 255.f, img[p * 3 + 1] / 255.f, img[p * 3 + 2] / 255.f};
    float cum = 1.f;                       // prod_{i<j} inv_i
    float first[3] = {0.f, 0.f, 0.f}, rest[3] = {0.f, 0.f, 0.f};
    for (int j = 0; j < n; ++j) {
      const float m = masks[(long)j * hw + p];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float mc = (m * colors[j * 3 + c]) * alpha;
        if (j == 0) first[c] = mc;
        else rest[c] += mc * cum;
      }
      cum = cum * (m * (-alpha) + 1.f);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float v = (px[c] * cum + (n > 1 ? first[c] + rest[c] : first[c])) * 255.f;
      out[p * 3 + c] = (unsigned char)(int)v;      // torch .byte(): truncation toward zero (values are in [0, 255])
    }
  }
}

}  // namespace

extern "C" int ymi_composite_masks_u8(const float *img, const float *masks, const float *colors, int n, int h, int w,
                                      float alpha, unsigned char *out, void *stream) {
  if (!img || !out) return YMI_ENULL;
  if (n < 0 || h <= 0 || w <= 0) return YMI_EARG;
  if (n > 0 && (!masks || !colors)) return YMI_ENULL;
  const long hw = (long)h * w;
  long g = (hw + 255) / 256;
  if (g > 256L * 16) g = 256L * 16;
  hipLaunchKernelGGL(composite_masks_k, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, img, masks, colors, n, hw,
                     alpha, out);
  return ymi_launch_status();
}

// ------------------------------------------------------------------------------------------------------------------
// prep_metrics without the 121 MB / image float masks (second half; eval.py:376-384,416-440):
//   * jaccard of predicted boxes vs ground truth on the device, reference op order (layers/box_utils.py:33-80);
//   * masks as BITS: one uint64 word per 64 consecutive pixels of the flat [h*w] mask (bit i of word j = pixel 64 j + i);
//     a wave's ballot IS the word.  ymi_mask_bits_f32 packs existing 0/1 float masks (ground truth);
//     ymi_mask_upsample_bits upsamples + thresholds the low-resolution masks of postprocess (output_utils.py:91-94) straight
//     into bits — the same up_coord / up_lerp2 arithmetic as the float kernel, so every bit equals the float path's pixel —
//     3.8 MB instead of 121 MB per image at 550 x 550;
//   * mask_iou from bits: intersection = popcount(a & b), areas = popcounts; all three are integers < 2^24, so
//     inter / (area_a + area_b - inter) evaluated in fp32 is bit-identical to box_utils.py:98-113 on the float masks.
namespace {

__global__ void jaccard_k(const float *__restrict__ a, const float *__restrict__ b, int A, int B, int iscrowd,
                          float *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= A * B) return;
  const int ia = i / B, ib = i - ia * B;
  const float ax1 = a[ia * 4 + 0], ay1 = a[ia * 4 + 1], ax2 = a[ia * 4 + 2], ay2 = a[ia * 4 + 3];
  const float bx1 = b[ib * 4 + 0], by1 = b[ib * 4 + 1], bx2 = b[ib * 4 + 2], by2 = b[ib * 4 + 3];
  // box_utils.py:47-51: inter = clamp(min(a.xy2, b.xy2) - max(a.xy1, b.xy1), 0).prod
  float iw = fminf(ax2, bx2) - fmaxf(ax1, bx1), ih = fminf(ay2, by2) - fmaxf(ay1, by1);
  iw = iw < 0.f ? 0.f : iw; ih = ih < 0.f ? 0.f : ih;
  const float inter = iw * ih;
  const float area_a = (ax2 - ax1) * (ay2 - ay1), area_b = (bx2 - bx1) * (by2 - by1);      // :72-75
  out[i] = iscrowd ? inter / area_a : inter / ((area_a + area_b) - inter);                 // :77-79
}

__global__ __launch_bounds__(256) void mask_bits_k(const float *__restrict__ m, long n, long W64, unsigned long long *__restrict__ bits) {
  const long mask = blockIdx.y;
  const float *src = m + mask * n;
  for (long wv = (long)blockIdx.x * 4 + (threadIdx.x >> 6); wv < W64; wv += (long)gridDim.x * 4) {
    const long p = wv * 64 + (threadIdx.x & 63);
    const float v = p < n ? src[p] : 0.f;
    const unsigned long long word = __ballot(v > 0.5f);
    if ((threadIdx.x & 63) == 0) bits[mask * W64 + wv] = word;
  }
}

__global__ __launch_bounds__(256) void mask_upsample_bits_k(const float *__restrict__ lo, int ph, int pw, int h, int w, float sh,
                                                            float sw, float thresh, long W64,
                                                            unsigned long long *__restrict__ bits) {
  const long mask = blockIdx.y;
  const float *img = lo + mask * (long)ph * pw;
  const long n = (long)h * w;
  for (long wv = (long)blockIdx.x * 4 + (threadIdx.x >> 6); wv < W64; wv += (long)gridDim.x * 4) {
    const long p = wv * 64 + (threadIdx.x & 63);
    bool on = false;
    if (p < n) {
      const int y = (int)(p / w), x = (int)(p - (long)y * w);
      int y0, y1, x0, x1; float ly, lx;
      up_coord(y, sh, ph, y0, y1, ly);
      up_coord(x, sw, pw, x0, x1, lx);
      const float v = up_lerp2(img[y0 * pw + x0], img[y0 * pw + x1], img[y1 * pw + x0], img[y1 * pw + x1], lx, ly);
      on = v > thresh;
    }
    const unsigned long long word = __ballot(on);
    if ((threadIdx.x & 63) == 0) bits[mask * W64 + wv] = word;
  }
}

// grid (A, ceil(B / 8)): a block intersects mask a with 8 masks b; areas ride along
__global__ __launch_bounds__(256) void mask_iou_bits_k(const unsigned long long *__restrict__ ba, const unsigned long long *__restrict__ bb,
                                                       int A, int B, long W64, int iscrowd, float *__restrict__ iou) {
  __shared__ unsigned red[4][17];
  const int a = blockIdx.x, b0 = blockIdx.y * 8;
  const unsigned long long *pa = ba + (long)a * W64;
  unsigned cnt[17];
#pragma unroll
  for (int k = 0; k < 17; ++k) cnt[k] = 0;
  for (long j = threadIdx.x; j < W64; j += 256) {
    const unsigned long long wa = pa[j];
    cnt[16] += (unsigned)__popcll(wa);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (b0 + k < B) {
        const unsigned long long wb = bb[(long)(b0 + k) * W64 + j];
        cnt[k] += (unsigned)__popcll(wa & wb);
        cnt[8 + k] += (unsigned)__popcll(wb);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 17; ++k) {
    unsigned v = cnt[k];
#pragma unroll
    for (int off = 32; off; off >>= 1) v += __shfl_xor(v, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < 8 && b0 + threadIdx.x < B) {
    const int k = threadIdx.x;
    const float inter = (float)(red[0][k] + red[1][k] + red[2][k] + red[3][k]);
    const float area_b = (float)(red[0][8 + k] + red[1][8 + k] + red[2][8 + k] + red[3][8 + k]);
    const float area_a = (float)(red[0][16] + red[1][16] + red[2][16] + red[3][16]);
    iou[(long)a * B + b0 + k] = iscrowd ? inter / area_a : inter / ((area_a + area_b) - inter);
  }
}

}  // namespace

extern "C" {

int ymi_jaccard_f32(const float *box_a, const float *box_b, int A, int B, int iscrowd, float *out, void *stream) {
  if (!box_a || !box_b || !out) return YMI_ENULL;
  if (A <= 0 || B <= 0 || (long)A * B > (1L << 30)) return YMI_EARG;
  hipLaunchKernelGGL(jaccard_k, dim3((A * B + 255) / 256), dim3(256), 0, (hipStream_t)stream, box_a, box_b, A, B, iscrowd, out);
  return ymi_launch_status();
}

int ymi_mask_bits_f32(const float *masks, int N, long n, uint64_t *bits, void *stream) {
  if (!masks || !bits) return YMI_ENULL;
  if (N <= 0 || N > 65535 || n <= 0) return YMI_EARG;
  const long W64 = (n + 63) / 64;
  long g = (W64 + 3) / 4;
  hipLaunchKernelGGL(mask_bits_k, dim3((unsigned)(g > 1024 ? 1024 : g), N), dim3(256), 0, (hipStream_t)stream, masks, n, W64,
                     (unsigned long long *)bits);
  return ymi_launch_status();
}

int ymi_mask_upsample_bits(const float *masks_lo, int N, int ph, int pw, int h, int w, float thresh, uint64_t *bits, void *stream) {
  if (!masks_lo || !bits) return YMI_ENULL;
  if (N <= 0 || N > 65535 || ph <= 0 || pw <= 0 || h <= 0 || w <= 0) return YMI_EARG;
  const long W64 = ((long)h * w + 63) / 64;
  long g = (W64 + 3) / 4;
  hipLaunchKernelGGL(mask_upsample_bits_k, dim3((unsigned)(g > 1024 ? 1024 : g), N), dim3(256), 0, (hipStream_t)stream, masks_lo, ph,
                     pw, h, w, (float)ph / (float)h, (float)pw / (float)w, thresh, W64, (unsigned long long *)bits);
  return ymi_launch_status();
}

int ymi_mask_iou_bits(const uint64_t *bits_a, const uint64_t *bits_b, int A, int B, long W64, int iscrowd, float *iou, void *stream) {
  if (!bits_a || !bits_b || !iou) return YMI_ENULL;
  if (A <= 0 || B <= 0 || W64 <= 0 || A > 65535 || (B + 7) / 8 > 65535) return YMI_EARG;
  hipLaunchKernelGGL(mask_iou_bits_k, dim3(A, (B + 7) / 8), dim3(256), 0, (hipStream_t)stream, (const unsigned long long *)bits_a,
                     (const unsigned long long *)bits_b, A, B, W64, iscrowd, iou);
  return ymi_launch_status();
}

}  // extern "C"
