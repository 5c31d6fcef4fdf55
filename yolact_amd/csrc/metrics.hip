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
