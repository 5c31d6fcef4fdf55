// Winograd F(2x2, 3x3) for the stride-1 / pad-1 3x3 convolutions (Lavin & Gray, "Fast Algorithms for Convolutional
// Neural Networks"): 16 multiplications per 2x2 output tile and channel pair instead of 36, i.e. 2.25x fewer MFMA
// FLOPs than the direct implicit GEMM, at the price of two bandwidth passes:
//   K1 wino_in_k    V[e][t][c]  = (B^T d B)[e]     d = 4x4 input patch of tile t (rows 2ty-1..2ty+2, zero padded)
//   K2 grouped GEMM M[e][t][n]  = sum_c V[e][t][c] * U[e][n][c]      16 independent [T x C] x [C x Cout] GEMMs on the
//                                 fp32 matrix cores (csrc/conv_igemm.hip, one launch, gridDim.y = 16)
//   K3 wino_out_k   y[b,2ty+i,2tx+j,n] = act(scale[n] * (A^T M A)[i][j] + bias[n])
// U = G g G^T is computed once on the host in fp64 (engine.py).  All transform matrices have entries in {0, +-1, +-1/2},
// so the only extra rounding is a handful of fp32 additions: measured error vs the direct kernel <= 2e-6 relative
// (tests/test_gpu_kernels.py::test_winograd_matches_direct), far inside the 1e-4 parity budget.
// The execution plan times this path against the direct kernel per layer and keeps the faster one.
#include "common.h"
#include "../../include/yolact_amd.h"

int ymi_internal_grouped_gemm(const ymi_conv_desc *d, int groups, long x_gs, long w_gs, long y_gs, double prof_flops,
                              int prof_kind, hipStream_t s, const void *a2 = nullptr, unsigned a2_plane = 0, long a2_gs = 0,
                              unsigned sc_gs = 0);
int ymi_internal_wgemm(const void *v, const void *u, const float *uinv, const float *x_amax, float amax_mul, float *m, int G, long T,
                       int C, int Ng, int cout_pad, double prof_flops, int prof_kind, hipStream_t s);      // csrc/wgemm.hip
int ymi_internal_prof_begin(double flops, int tile, int kind, hipStream_t s);
void ymi_internal_prof_end(int idx, hipStream_t s);

namespace {

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// fp16x2 planes of V (ymi_wino_desc.v_planes): component e of (tile t, channels c .. c+3) as  v * s = h + l  (round to nearest),
// layout [G][2][T][C] fp16 — the same bytes as the fp32 V it replaces; the grouped GEMM then stages both operands as planes and
// its K loop contains no operand split at all (csrc/conv_igemm.hip PREC 4).  `o` points at plane 0 of component 0.
__device__ __forceinline__ void store_v_planes(_Float16 *o, long e, long stride_e2, long plane, const f32x4 v, float s) {
  f16x4 h, l;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float t = v[k] * s;
    h[k] = (_Float16)t;
    l[k] = (_Float16)(t - (float)h[k]);
  }
  *reinterpret_cast<f16x4 *>(o + e * stride_e2) = h;
  *reinterpret_cast<f16x4 *>(o + e * stride_e2 + plane) = l;
}

__device__ __forceinline__ f32x4 ld4(const float *p, bool ok) {
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  return ok ? *reinterpret_cast<const f32x4 *>(p) : z;
}

// one thread = one tile x 4 channels.  x [B,H,W,C] NHWC, V [16][T][C]  (PLANES: fp16x2 planes [16][2][T][C], scaled by the power
// of two derived from *x_amax * 4 — |B^T d B| <= 4 max|d| for F(2x2,3x3))
template <bool PLANES>
__global__ __launch_bounds__(256) void wino_in_k(const float *__restrict__ x, float *__restrict__ V, int H, int W, int C4,
                                                 int th, int tw, long T, long total, const float *__restrict__ x_amax) {
  float vs = 1.f, vinv = 1.f;
  if (PLANES) ymi_h2_scale(ymi_amax_read(x_amax) * 4.f, vs, vinv);
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < (unsigned)total; i += gridDim.x * 256u) {   // 32-bit index math
    const unsigned tu = i / (unsigned)C4, ru = tu / (unsigned)tw, bu = ru / (unsigned)th;          // (total < 2^31, host-checked;
    const int c4 = (int)(i - tu * (unsigned)C4), tx = (int)(tu - ru * (unsigned)tw), ty = (int)(ru - bu * (unsigned)th);   // 64-bit
    const long t = tu, b = bu;                                                                  // div / mod cost ~3x the transform)
    const int y0 = 2 * ty - 1, x0 = 2 * tx - 1;
    const float *base = x + ((b * H) * (long)W) * (C4 * 4L) + c4 * 4;
    f32x4 d[4][4];
#pragma unroll
    for (int iy = 0; iy < 4; ++iy) {
      const int yy = y0 + iy;
      const bool yok = (unsigned)yy < (unsigned)H;
#pragma unroll
      for (int ix = 0; ix < 4; ++ix) {
        const int xx = x0 + ix;
        const bool ok = yok && (unsigned)xx < (unsigned)W;
        d[iy][ix] = ld4(base + ((long)(ok ? yy : 0) * W + (ok ? xx : 0)) * (C4 * 4L), ok);
      }
    }
    // B^T d B,  B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
    f32x4 u[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {          // rows: u = B^T d
      u[0][j] = d[0][j] - d[2][j];
      u[1][j] = d[1][j] + d[2][j];
      u[2][j] = d[2][j] - d[1][j];
      u[3][j] = d[1][j] - d[3][j];
    }
    const long stride_e = T * (C4 * 4L);
    float *o = V + t * (C4 * 4L) + c4 * 4;
    _Float16 *o2 = reinterpret_cast<_Float16 *>(V) + t * (C4 * 4L) + c4 * 4;
#pragma unroll
    for (int iy = 0; iy < 4; ++iy) {       // columns: v = u B
      const f32x4 v0 = u[iy][0] - u[iy][2], v1 = u[iy][1] + u[iy][2], v2 = u[iy][2] - u[iy][1], v3 = u[iy][1] - u[iy][3];
      if (PLANES) {
        store_v_planes(o2, iy * 4 + 0, 2 * stride_e, stride_e, v0, vs);
        store_v_planes(o2, iy * 4 + 1, 2 * stride_e, stride_e, v1, vs);
        store_v_planes(o2, iy * 4 + 2, 2 * stride_e, stride_e, v2, vs);
        store_v_planes(o2, iy * 4 + 3, 2 * stride_e, stride_e, v3, vs);
      } else {
        *reinterpret_cast<f32x4 *>(o + (iy * 4 + 0) * stride_e) = v0;
        *reinterpret_cast<f32x4 *>(o + (iy * 4 + 1) * stride_e) = v1;
        *reinterpret_cast<f32x4 *>(o + (iy * 4 + 2) * stride_e) = v2;
        *reinterpret_cast<f32x4 *>(o + (iy * 4 + 3) * stride_e) = v3;
      }
    }
  }
}

// one thread = one tile x 4 output channels.  M [16][T][N], y [B,Ho,Wo,N]
__global__ __launch_bounds__(256) void wino_out_k(const float *__restrict__ Mm, float *__restrict__ y,
                                                  const float *__restrict__ scale, const float *__restrict__ bias, int Ho,
                                                  int Wo, int N4, int th, int tw, long T, int act, long total,
                                                  float *__restrict__ y_amax) {
  float am = 0.f;
  const ymi_amax_pre apre = ymi_amax_prefetch(y_amax);
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < (unsigned)total; i += gridDim.x * 256u) {   // 32-bit index math
    const unsigned tu = i / (unsigned)N4, ru = tu / (unsigned)tw, bu = ru / (unsigned)th;          // (total < 2^31, host-checked;
    const int n4 = (int)(i - tu * (unsigned)N4), tx = (int)(tu - ru * (unsigned)tw), ty = (int)(ru - bu * (unsigned)th);   // 64-bit
    const long t = tu, b = bu;                                                                  // div / mod cost ~3x the transform)
    const long stride_e = T * (N4 * 4L);
    const float *src = Mm + t * (N4 * 4L) + n4 * 4;
    f32x4 m[4][4];
#pragma unroll
    for (int e = 0; e < 16; ++e) m[e >> 2][e & 3] = *reinterpret_cast<const f32x4 *>(src + e * stride_e);
    // A^T m A,  A^T = [1 1 1 0; 0 1 -1 -1]
    f32x4 s[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s[0][j] = (m[0][j] + m[1][j]) + m[2][j];
      s[1][j] = (m[1][j] - m[2][j]) - m[3][j];
    }
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, bi = {0.f, 0.f, 0.f, 0.f};
    if (scale) sc = *reinterpret_cast<const f32x4 *>(scale + n4 * 4);
    if (bias) bi = *reinterpret_cast<const f32x4 *>(bias + n4 * 4);
    const float slope = act == YMI_ACT_RELU ? 0.f : (act == YMI_ACT_LEAKY01 ? 0.1f : 1.f);
    // Every value (scale, bias, activation) BEFORE the first store (round 5, from the ISA): a load result first used inside a
    // conditional store block made the compiler wait with vmcnt(0) in EVERY such block, i.e. for the previous block's store — one
    // memory round trip per pixel.  Measured on the plan's launches: 8.4 -> 7.6 us here, 27.8 -> 25.0 us (69 x 69 x 256, batch 8) and
    // 10.4 -> 9.4 us (35 x 35) for wino43_out_k.  The same restructuring of the INPUT transforms (36 unconditional loads up front
    // instead of six branchy columns) and of the segmented output kernels measured 2 - 10 % SLOWER and was not kept: those launches
    // already run at 3.3 - 5 TB/s of tensor bytes, the dependent round trips were not what bounded them.
    f32x4 vv[2][2];
#pragma unroll
    for (int iy = 0; iy < 2; ++iy) {
      const f32x4 o0 = (s[iy][0] + s[iy][1]) + s[iy][2], o1 = (s[iy][1] - s[iy][2]) - s[iy][3];
#pragma unroll
      for (int ix = 0; ix < 2; ++ix) {
        f32x4 v = (ix == 0 ? o0 : o1) * sc + bi;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], slope * v[e]);
        vv[iy][ix] = v;
        const bool ok = 2 * ty + iy < Ho && 2 * tx + ix < Wo;
        am = fmaxf(am, ok ? ymi_absmax4(v) : 0.f);
      }
    }
#pragma unroll
    for (int iy = 0; iy < 2; ++iy)
#pragma unroll
      for (int ix = 0; ix < 2; ++ix) {
        const int oy = 2 * ty + iy, ox = 2 * tx + ix;
        if (oy < Ho && ox < Wo) *reinterpret_cast<f32x4 *>(y + ((b * Ho + oy) * (long)Wo + ox) * (N4 * 4L) + n4 * 4) = vv[iy][ix];
      }
  }
  if (y_amax) ymi_amax_finish(apre, am);
}

// Same output transform, scattering to up to 3 output segments with their own strides / activations (the shared
// prediction-head conv: loc | coef (tanh) | conf rows of the level-concatenated [B,P,k] tensors, yolact.py:169-193).
// Rows of a segment may be only 4-byte aligned (conf: 243 floats), so the stores are scalar.
struct SegTab { ymi_conv_seg seg[3]; int nseg; };

__device__ __forceinline__ float wino_act(float v, int act) {
  switch (act) {
    case YMI_ACT_RELU: return v > 0.f ? v : 0.f;
    case YMI_ACT_LEAKY01: return v > 0.f ? v : 0.1f * v;
    case YMI_ACT_TANH: return tanhf(v);
    case YMI_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
    default: return v;
  }
}

// The 4 channels n .. n+3 of a thread lie in ONE segment whose rows can take 16-byte stores (the two dense halves of the
// merged head0.upfeature + proto_net[0] layer, the loc and coefficient rows of the prediction heads; NOT the 243-float
// class rows): returns the segment index or -1.
__device__ __forceinline__ int seg_vec4(const SegTab &st, int n, int Cout) {
  if (n + 3 >= Cout) return -1;
#pragma unroll
  for (int k = 0; k < 3; ++k)
    if (k < st.nseg && n >= st.seg[k].n0 && n + 3 < st.seg[k].n1 && ((n - st.seg[k].n0) & 3) == 0 &&
        (st.seg[k].row_stride & 3) == 0 && (st.seg[k].batch_stride & 3) == 0 && (((uintptr_t)st.seg[k].ptr) & 15) == 0)
      return k;
  return -1;
}

// per-thread constants of the 16-byte store path: resolved ONCE per (tile, channel group), not per pixel
struct SegVec {
  float *ptr; long bs; int rs, act;
  f32x4 sc, bi;
};
__device__ __forceinline__ SegVec seg_vec_setup(const SegTab &st, int k, const float *__restrict__ scale,
                                                const float *__restrict__ bias, int n) {
  ymi_conv_seg sg = st.seg[0];
  if (k == 1) sg = st.seg[1];
  if (k == 2) sg = st.seg[2];
  SegVec v;
  v.ptr = sg.ptr + (n - sg.n0); v.bs = sg.batch_stride; v.rs = sg.row_stride; v.act = sg.act;
#pragma unroll
  for (int e = 0; e < 4; ++e) { v.sc[e] = scale ? scale[n + e] : 1.f; v.bi[e] = bias ? bias[n + e] : 0.f; }
  return v;
}

// Per-segment magnitude bounds (ABI 5): a launch with nseg > 1 segments raises nseg CONSECUTIVE slots (YMI_AMAX_SUB * YMI_AMAX_STRIDE
// floats apart), segment k -> slot k.  One shared bound was wrong for the merged head0.upfeature + proto_net[0] launch: its two
// halves are different tensors with different consumers, and a large proto_net[0] channel inflated the bound — hence coarsened the
// fp16x2 scale — of the head's input (tests/test_gpu_batch_parity.py test_outlier_channels_end_to_end_at_batch8).
constexpr int YMI_AMAX_SLOT = YMI_AMAX_SUB * YMI_AMAX_STRIDE;
__device__ __forceinline__ void seg_amax_add(float (&am)[3], int k, float v) {
  am[0] = k == 0 ? fmaxf(am[0], v) : am[0];
  am[1] = k == 1 ? fmaxf(am[1], v) : am[1];
  am[2] = k == 2 ? fmaxf(am[2], v) : am[2];
}
__device__ __forceinline__ void seg_amax_commit(float *y_amax, int nseg, const ymi_amax_pre &apre0, const float (&am)[3]) {
  if (!y_amax) return;
  ymi_amax_finish(apre0, am[0]);
  if (nseg > 1) ymi_amax_finish(ymi_amax_prefetch(y_amax + YMI_AMAX_SLOT), am[1]);
  if (nseg > 2) ymi_amax_finish(ymi_amax_prefetch(y_amax + 2 * YMI_AMAX_SLOT), am[2]);
}

__device__ __forceinline__ float seg_vec_store(const SegVec &sv, f32x4 v, long b, long pix) {
  v = v * sv.sc + sv.bi;
  if (sv.act <= YMI_ACT_LEAKY01) {      // none / ReLU / LeakyReLU: max(x, slope x)
    const float slope = sv.act == YMI_ACT_RELU ? 0.f : (sv.act == YMI_ACT_LEAKY01 ? 0.1f : 1.f);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], slope * v[e]);
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = wino_act(v[e], sv.act);
  }
  *reinterpret_cast<f32x4 *>(sv.ptr + b * sv.bs + pix * sv.rs) = v;
  return ymi_absmax4(v);
}

__global__ __launch_bounds__(256) void wino_out_seg_k(const float *__restrict__ Mm, const SegTab st,
                                                      const float *__restrict__ scale, const float *__restrict__ bias,
                                                      int Ho, int Wo, int N4, int Cout, int th, int tw, long T, long total, float *__restrict__ y_amax) {
  float am[3] = {0.f, 0.f, 0.f};
  const ymi_amax_pre apre = ymi_amax_prefetch(y_amax);
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < (unsigned)total; i += gridDim.x * 256u) {   // 32-bit index math
    const unsigned tu = i / (unsigned)N4, ru = tu / (unsigned)tw, bu = ru / (unsigned)th;          // (total < 2^31, host-checked;
    const int n4 = (int)(i - tu * (unsigned)N4), tx = (int)(tu - ru * (unsigned)tw), ty = (int)(ru - bu * (unsigned)th);   // 64-bit
    const long t = tu, b = bu;                                                                  // div / mod cost ~3x the transform)
    const long stride_e = T * (N4 * 4L);
    const float *src = Mm + t * (N4 * 4L) + n4 * 4;
    f32x4 m[4][4];
#pragma unroll
    for (int e = 0; e < 16; ++e) m[e >> 2][e & 3] = *reinterpret_cast<const f32x4 *>(src + e * stride_e);
    f32x4 s[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s[0][j] = (m[0][j] + m[1][j]) + m[2][j];
      s[1][j] = (m[1][j] - m[2][j]) - m[3][j];
    }
    f32x4 o[2][2];
#pragma unroll
    for (int iy = 0; iy < 2; ++iy) {
      o[iy][0] = (s[iy][0] + s[iy][1]) + s[iy][2];
      o[iy][1] = (s[iy][1] - s[iy][2]) - s[iy][3];
    }
    const int kv = seg_vec4(st, n4 * 4, Cout);
    if (kv >= 0) {
      const SegVec sv = seg_vec_setup(st, kv, scale, bias, n4 * 4);
#pragma unroll
      for (int iy = 0; iy < 2; ++iy)
#pragma unroll
        for (int ix = 0; ix < 2; ++ix) {
          const int oy = 2 * ty + iy, ox = 2 * tx + ix;
          if (oy < Ho && ox < Wo) seg_amax_add(am, kv, seg_vec_store(sv, o[iy][ix], b, (long)oy * Wo + ox));
        }
      continue;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int n = n4 * 4 + e;
      if (n >= Cout) continue;
      // segment of channel n: compile-time indices + selects (the table lives in SGPRs)
      float *ptr = nullptr; long bs = 0; int rs = 0, act = 0, n0 = 0, ks = 0;
#pragma unroll
      for (int k = 0; k < 3; ++k)
        if (k < st.nseg && n >= st.seg[k].n0 && n < st.seg[k].n1) {
          ptr = st.seg[k].ptr; bs = st.seg[k].batch_stride; rs = st.seg[k].row_stride; act = st.seg[k].act; n0 = st.seg[k].n0; ks = k;
        }
      if (!ptr) continue;
      const float sc = scale ? scale[n] : 1.f, bi = bias ? bias[n] : 0.f;
#pragma unroll
      for (int iy = 0; iy < 2; ++iy) {
        const int oy = 2 * ty + iy;
        if (oy >= Ho) continue;
#pragma unroll
        for (int ix = 0; ix < 2; ++ix) {
          const int ox = 2 * tx + ix;
          if (ox >= Wo) continue;
          const float val = wino_act(o[iy][ix][e] * sc + bi, act);
          seg_amax_add(am, ks, fabsf(val));
          ptr[b * bs + ((long)oy * Wo + ox) * rs + (n - n0)] = val;
        }
      }
    }
  }
  seg_amax_commit(y_amax, st.nseg, apre, am);
}

// ---- F(4x4, 3x3): 36 multiplications per 4x4 output tile instead of 144 (4x fewer MFMA FLOPs than direct, 1.78x fewer
// than F(2x2)); V / M hold 36 values per 16 pixels (2.25x the activation bytes instead of 4x).  Transform matrices of
// Lavin & Gray (interpolation points 0, +-1, +-2, inf); the filter transform G g G^T is done on the host in fp64.
// Rounding: coefficients up to 8 amplify fp32 rounding ~4x over F(2x2); end to end on the golden cases the heads move by
// <= 8e-6 relative (direct kernels: 2e-6), an order of magnitude inside the 1e-4 budget.

// t = B^T v for one column / row of six values;  B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0;
//                                                      0 2 -1 -2 1 0; 0 4 0 -5 0 1]
__device__ __forceinline__ void bt6(const f32x4 v0, const f32x4 v1, const f32x4 v2, const f32x4 v3, const f32x4 v4, const f32x4 v5,
                                    f32x4 &t0, f32x4 &t1, f32x4 &t2, f32x4 &t3, f32x4 &t4, f32x4 &t5) {
  const f32x4 a = v4 - v2 * 4.f, b = v3 - v1 * 4.f;       // shared terms
  const f32x4 c = v4 - v2, e = (v3 - v1) * 2.f;
  t0 = (v0 * 4.f - v2 * 5.f) + v4;
  t1 = a + b;
  t2 = a - b;
  t3 = c + e;
  t4 = c - e;
  t5 = (v1 * 4.f - v3 * 5.f) + v5;
}

// one thread = one 6x6 input patch (tile) x 4 channels.  x [B,H,W,C] NHWC, V [36][T][C]  (PLANES: fp16x2 planes [36][2][T][C];
// |B^T d B| <= 100 max|d|: every row of B^T has an absolute sum of at most 10)
// torch's area_pixel_compute_source_index for align_corners = False, exactly as csrc/layout.hip bl_coord evaluates it
__device__ __forceinline__ void up2_coord(int dst, int in_size, float &l1) {
  float src = 0.5f * ((float)dst + 0.5f) - 0.5f;
  src = src < 0.f ? 0.f : src;
  int i0 = (int)src;
  if (i0 > in_size - 1) i0 = in_size - 1;
  l1 = src - (float)i0;
}

// UPS: the layer's input is the 2x bilinear upsampling (+ ReLU) of xl [B,H/2,W/2,C]: the 6 x 6 patch of a tile (hi-res rows
// 4 ty - 1 .. 4 ty + 4) only touches the 4 x 4 low-res window starting at (2 ty - 1, 2 tx - 1): hi-res offset r uses window rows
// r >> 1 and (r >> 1) + 1 (window addresses clamped to the image: where the reference clamps i0 / i1 the weights it derives are
// such that the duplicated row reproduces its arithmetic), with the weights of bl_coord and the operation order of
// bilinear_nhwc_k: hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11).  16 loads instead of 36, no upsampled tensor in HBM.
template <bool PLANES, bool UPS = false>
__global__ __launch_bounds__(256) void wino43_in_k(const float *__restrict__ x, float *__restrict__ V, int H, int W, int C4,
                                                   int th, int tw, long T, long total, const float *__restrict__ x_amax,
                                                   int up_relu = 0) {
  float vs = 1.f, vinv = 1.f;
  if (PLANES) ymi_h2_scale(ymi_amax_read(x_amax) * 100.f, vs, vinv);
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < (unsigned)total; i += gridDim.x * 256u) {   // 32-bit index math
    const unsigned tu = i / (unsigned)C4, ru = tu / (unsigned)tw, bu = ru / (unsigned)th;          // (total < 2^31, host-checked;
    const int c4 = (int)(i - tu * (unsigned)C4), tx = (int)(tu - ru * (unsigned)tw), ty = (int)(ru - bu * (unsigned)th);   // 64-bit
    const long t = tu, b = bu;                                                                  // div / mod cost ~3x the transform)
    const int y0 = 4 * ty - 1, x0 = 4 * tx - 1;
    f32x4 u[6][6];                         // u = B^T d, built column by column
    if constexpr (UPS) {
      const int Hl = H >> 1, Wl = W >> 1;
      const float *lbase = x + ((b * Hl) * (long)Wl) * (C4 * 4L) + c4 * 4;
      f32x4 lo[4][4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        int wy = 2 * ty - 1 + k; wy = wy < 0 ? 0 : (wy > Hl - 1 ? Hl - 1 : wy);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          int wx = 2 * tx - 1 + q; wx = wx < 0 ? 0 : (wx > Wl - 1 ? Wl - 1 : wx);
          lo[k][q] = *reinterpret_cast<const f32x4 *>(lbase + ((long)wy * Wl + wx) * (C4 * 4L));
        }
      }
      float lx[6], ly[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) { up2_coord(x0 + j, Wl, lx[j]); up2_coord(y0 + j, Hl, ly[j]); }
      f32x4 hrow[4][6];                    // horizontal pass first (the reference's inner brackets)
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < 6; ++j) hrow[k][j] = (1.f - lx[j]) * lo[k][j >> 1] + lx[j] * lo[k][(j >> 1) + 1];
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const bool xok = (unsigned)(x0 + j) < (unsigned)W;
        f32x4 dcol[6];
#pragma unroll
        for (int iy = 0; iy < 6; ++iy) {
          const bool ok = xok && (unsigned)(y0 + iy) < (unsigned)H;
          f32x4 v = (1.f - ly[iy]) * hrow[iy >> 1][j] + ly[iy] * hrow[(iy >> 1) + 1][j];
          if (up_relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] < 0.f ? 0.f : v[e];
          }
          const f32x4 z = {0.f, 0.f, 0.f, 0.f};
          dcol[iy] = ok ? v : z;
        }
        bt6(dcol[0], dcol[1], dcol[2], dcol[3], dcol[4], dcol[5], u[0][j], u[1][j], u[2][j], u[3][j], u[4][j], u[5][j]);
      }
    } else {
    const float *base = x + ((b * H) * (long)W) * (C4 * 4L) + c4 * 4;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int xx = x0 + j;
      const bool xok = (unsigned)xx < (unsigned)W;
      f32x4 dcol[6];
#pragma unroll
      for (int iy = 0; iy < 6; ++iy) {
        const int yy = y0 + iy;
        const bool ok = xok && (unsigned)yy < (unsigned)H;
        dcol[iy] = ld4(base + ((long)(ok ? yy : 0) * W + (ok ? xx : 0)) * (C4 * 4L), ok);
      }
      bt6(dcol[0], dcol[1], dcol[2], dcol[3], dcol[4], dcol[5], u[0][j], u[1][j], u[2][j], u[3][j], u[4][j], u[5][j]);
    }
    }
    const long stride_e = T * (C4 * 4L);
    float *o = V + t * (C4 * 4L) + c4 * 4;
    _Float16 *o2 = reinterpret_cast<_Float16 *>(V) + t * (C4 * 4L) + c4 * 4;
#pragma unroll
    for (int iy = 0; iy < 6; ++iy) {       // rows: v = u B
      f32x4 v[6];
      bt6(u[iy][0], u[iy][1], u[iy][2], u[iy][3], u[iy][4], u[iy][5], v[0], v[1], v[2], v[3], v[4], v[5]);
#pragma unroll
      for (int l = 0; l < 6; ++l) {
        if (PLANES) store_v_planes(o2, iy * 6 + l, 2 * stride_e, stride_e, v[l], vs);
        else *reinterpret_cast<f32x4 *>(o + (iy * 6 + l) * stride_e) = v[l];
      }
    }
  }
}

// s = A^T v for six values;  A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
__device__ __forceinline__ void at6(const f32x4 v0, const f32x4 v1, const f32x4 v2, const f32x4 v3, const f32x4 v4, const f32x4 v5,
                                    f32x4 &s0, f32x4 &s1, f32x4 &s2, f32x4 &s3) {
  const f32x4 p12 = v1 + v2, m12 = v1 - v2, p34 = v3 + v4, m34 = v3 - v4;
  s0 = (v0 + p12) + p34;
  s1 = m12 + m34 * 2.f;
  s2 = p12 + p34 * 4.f;
  s3 = (m12 + m34 * 8.f) + v5;
}

// o[4][4] = A^T M A of tile t, 4 output channels.  M [36][T][N]
__device__ __forceinline__ void wino43_out_tile(const float *__restrict__ src, long stride_e, f32x4 (&o)[4][4]) {
  f32x4 s[4][6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    f32x4 mc[6];
#pragma unroll
    for (int iy = 0; iy < 6; ++iy) mc[iy] = *reinterpret_cast<const f32x4 *>(src + (iy * 6 + j) * stride_e);
    at6(mc[0], mc[1], mc[2], mc[3], mc[4], mc[5], s[0][j], s[1][j], s[2][j], s[3][j]);
  }
#pragma unroll
  for (int iy = 0; iy < 4; ++iy) at6(s[iy][0], s[iy][1], s[iy][2], s[iy][3], s[iy][4], s[iy][5], o[iy][0], o[iy][1], o[iy][2], o[iy][3]);
}

__global__ __launch_bounds__(256) void wino43_out_k(const float *__restrict__ Mm, float *__restrict__ y,
                                                    const float *__restrict__ scale, const float *__restrict__ bias, int Ho,
                                                    int Wo, int N4, int th, int tw, long T, int act, long total,
                                                    float *__restrict__ y_amax) {
  float am = 0.f;
  const ymi_amax_pre apre = ymi_amax_prefetch(y_amax);
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < (unsigned)total; i += gridDim.x * 256u) {   // 32-bit index math
    const unsigned tu = i / (unsigned)N4, ru = tu / (unsigned)tw, bu = ru / (unsigned)th;          // (total < 2^31, host-checked;
    const int n4 = (int)(i - tu * (unsigned)N4), tx = (int)(tu - ru * (unsigned)tw), ty = (int)(ru - bu * (unsigned)th);   // 64-bit
    const long t = tu, b = bu;                                                                  // div / mod cost ~3x the transform)
    f32x4 o[4][4];
    wino43_out_tile(Mm + t * (N4 * 4L) + n4 * 4, T * (N4 * 4L), o);
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, bi = {0.f, 0.f, 0.f, 0.f};
    if (scale) sc = *reinterpret_cast<const f32x4 *>(scale + n4 * 4);
    if (bias) bi = *reinterpret_cast<const f32x4 *>(bias + n4 * 4);
    const float slope = act == YMI_ACT_RELU ? 0.f : (act == YMI_ACT_LEAKY01 ? 0.1f : 1.f);
#pragma unroll
    for (int iy = 0; iy < 4; ++iy)           // values first, stores last (see wino_out_k)
#pragma unroll
      for (int ix = 0; ix < 4; ++ix) {
        f32x4 v = o[iy][ix] * sc + bi;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], slope * v[e]);
        o[iy][ix] = v;
        const bool ok = 4 * ty + iy < Ho && 4 * tx + ix < Wo;
        am = fmaxf(am, ok ? ymi_absmax4(v) : 0.f);
      }
#pragma unroll
    for (int iy = 0; iy < 4; ++iy)
#pragma unroll
      for (int ix = 0; ix < 4; ++ix) {
        const int oy = 4 * ty + iy, ox = 4 * tx + ix;
        if (oy < Ho && ox < Wo) *reinterpret_cast<f32x4 *>(y + ((b * Ho + oy) * (long)Wo + ox) * (N4 * 4L) + n4 * 4) = o[iy][ix];
      }
  }
  if (y_amax) ymi_amax_finish(apre, am);
}

// ---- F(4x4,3x3) output transform + the 1x1 convolution that consumes it, in one launch (ymi_wino_desc.proj_*) ------------------
// protonet ends conv3x3(256 -> 256) + ReLU -> conv1x1(256 -> 32) (utils/functions.py:163-213, yolact.py:588-599): the 3x3's output
// at 138 x 138 x 8 is 156 MB written by the output transform and read back by a layer of 2.5 GFLOP.  Here a wave owns one 4x4 tile
// x all 256 channels (lane = 4 channels, as in wino43_out_k), keeps scale / bias / ReLU'd values in registers, writes them to LDS as
// the two fp16 planes of the fp16x2 arithmetic (power-of-two scale from the TILE's own maximum: exact, no tensor-wide bound
// needed), and multiplies the [16 pixels x 256] tile by the 1x1 filters [256 x 32] (resident in LDS for the block's lifetime) on
// v_mfma_f32_16x16x32_f16 — issued as W Y^T so that a lane ends with four consecutive output channels of one pixel.  The 3x3's
// output never exists in memory.  LDS rows are 512 + 16 bytes: lanes of a 16-lane read phase hit 16 different bank groups.
struct ProjArgs {
  const void *w_h2; const float *scale_h2, *bias; float *y, *y_amax;
  int cout, ldy, act; unsigned w_plane;
};
constexpr int PJ_ROW = 528, PJ_YPLANE = 16 * PJ_ROW, PJ_WPLANE = 32 * PJ_ROW, PJ_LDS = 2 * PJ_WPLANE + 4 * 2 * PJ_YPLANE;

__global__ __launch_bounds__(256) void wino43_out_proj_k(const float *__restrict__ Mm, const float *__restrict__ scale,
                                                         const float *__restrict__ bias, int Ho, int Wo, int th, int tw, long T,
                                                         int act, const ProjArgs pp) {
#if defined(__HIP_DEVICE_COMPILE__)
  __shared__ __attribute__((aligned(16))) char lds[PJ_LDS];
  typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
  typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lr = lane & 15, g = lane >> 4;
  const ymi_amax_pre apre = ymi_amax_prefetch(pp.y_amax);
  // the 1x1 filters: planes [2][CoutPad][256] fp16 -> LDS [2][32][528 B] (rows past cout are the zero padding rows of the pack)
  for (int u = t; u < 2 * 32 * 32; u += 256) {          // 16-byte pieces: (plane, row, piece of 32)
    const int plane = u >> 10, row = (u >> 5) & 31, pc = u & 31;
    const f32x4 v = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(pp.w_h2) + (size_t)plane * pp.w_plane + (size_t)row * 512 + pc * 16);
    *reinterpret_cast<f32x4 *>(lds + plane * PJ_WPLANE + row * PJ_ROW + pc * 16) = v;
  }
  __syncthreads();
  char *ytile = lds + 2 * PJ_WPLANE + wave * (2 * PJ_YPLANE);
  f32x4 sc = {1.f, 1.f, 1.f, 1.f}, bi = {0.f, 0.f, 0.f, 0.f};
  if (scale) sc = *reinterpret_cast<const f32x4 *>(scale + lane * 4);
  if (bias) bi = *reinterpret_cast<const f32x4 *>(bias + lane * 4);
  const float slope = act == YMI_ACT_RELU ? 0.f : (act == YMI_ACT_LEAKY01 ? 0.1f : 1.f);
  const float slope2 = pp.act == YMI_ACT_RELU ? 0.f : (pp.act == YMI_ACT_LEAKY01 ? 0.1f : 1.f);
  float am = 0.f;
  for (long tl = (long)blockIdx.x * 4 + wave; tl < T; tl += (long)gridDim.x * 4) {
    const unsigned tu = (unsigned)tl, ru = tu / (unsigned)tw, bu = ru / (unsigned)th;
    const int tx = (int)(tu - ru * (unsigned)tw), ty = (int)(ru - bu * (unsigned)th);
    f32x4 o[4][4];
    wino43_out_tile(Mm + tl * 256L + lane * 4, T * 256L, o);
    float tm = 0.f;
#pragma unroll
    for (int iy = 0; iy < 4; ++iy)
#pragma unroll
      for (int ix = 0; ix < 4; ++ix) {
        f32x4 v = o[iy][ix] * sc + bi;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], slope * v[e]);
        o[iy][ix] = v;
        tm = fmaxf(tm, ymi_absmax4(v));
      }
    const unsigned tmb = ymi_wave_umax63(__builtin_bit_cast(unsigned, tm));
    float sT, invT;
    ymi_h2_scale(__builtin_bit_cast(float, (unsigned)__builtin_amdgcn_readlane((int)tmb, 63)), sT, invT);
#pragma unroll
    for (int iy = 0; iy < 4; ++iy)
#pragma unroll
      for (int ix = 0; ix < 4; ++ix) {
        const f32x4 v = o[iy][ix] * sT;
        f16x4 h4, l4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const _Float16 h = (_Float16)v[e];
          h4[e] = h;
          l4[e] = (_Float16)(v[e] - (float)h);
        }
        char *dst = ytile + (iy * 4 + ix) * PJ_ROW + lane * 8;
        *reinterpret_cast<f16x4 *>(dst) = h4;
        *reinterpret_cast<f16x4 *>(dst + PJ_YPLANE) = l4;
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    f32x4 acc[2];
    acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
    acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const f16x8 xh = *reinterpret_cast<const f16x8 *>(ytile + lr * PJ_ROW + c * 64 + g * 16);
      const f16x8 xl = *reinterpret_cast<const f16x8 *>(ytile + PJ_YPLANE + lr * PJ_ROW + c * 64 + g * 16);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const f16x8 wh = *reinterpret_cast<const f16x8 *>(lds + (j * 16 + lr) * PJ_ROW + c * 64 + g * 16);
        const f16x8 wl = *reinterpret_cast<const f16x8 *>(lds + PJ_WPLANE + (j * 16 + lr) * PJ_ROW + c * 64 + g * 16);
        acc[j] = ymi_mfma16(wl, xh, acc[j]);      // D^T[n][pixel] += W Y^T: h*l, l*h, h*h
        acc[j] = ymi_mfma16(wh, xl, acc[j]);
        acc[j] = ymi_mfma16(wh, xh, acc[j]);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();                    // (the next tile's plane writes stay behind these reads)
    // lane: output channels 16 j + 4 g .. + 3 of pixel lr = (iy, ix) = (lr >> 2, lr & 3) of the tile
    const int oy = 4 * ty + (lr >> 2), ox = 4 * tx + (lr & 3);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = j * 16 + 4 * g;
      if (n < pp.cout && oy < Ho && ox < Wo) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float r = acc[j][e] * (pp.scale_h2[n + e] * invT) + (pp.bias ? pp.bias[n + e] : 0.f);
          v[e] = fmaxf(r, slope2 * r);
        }
        am = fmaxf(am, ymi_absmax4(v));
        *reinterpret_cast<f32x4 *>(pp.y + (((long)bu * Ho + oy) * (long)Wo + ox) * pp.ldy + n) = v;
      }
    }
  }
  if (pp.y_amax) ymi_amax_finish(apre, am);
#endif
}

__global__ __launch_bounds__(256) void wino43_out_seg_k(const float *__restrict__ Mm, const SegTab st,
                                                        const float *__restrict__ scale, const float *__restrict__ bias,
                                                        int Ho, int Wo, int N4, int Cout, int th, int tw, long T, long total, float *__restrict__ y_amax) {
  float am[3] = {0.f, 0.f, 0.f};
  const ymi_amax_pre apre = ymi_amax_prefetch(y_amax);
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < (unsigned)total; i += gridDim.x * 256u) {   // 32-bit index math
    const unsigned tu = i / (unsigned)N4, ru = tu / (unsigned)tw, bu = ru / (unsigned)th;          // (total < 2^31, host-checked;
    const int n4 = (int)(i - tu * (unsigned)N4), tx = (int)(tu - ru * (unsigned)tw), ty = (int)(ru - bu * (unsigned)th);   // 64-bit
    const long t = tu, b = bu;                                                                  // div / mod cost ~3x the transform)
    f32x4 o[4][4];
    wino43_out_tile(Mm + t * (N4 * 4L) + n4 * 4, T * (N4 * 4L), o);
    const int kv = seg_vec4(st, n4 * 4, Cout);
    if (kv >= 0) {
      const SegVec sv = seg_vec_setup(st, kv, scale, bias, n4 * 4);
#pragma unroll
      for (int iy = 0; iy < 4; ++iy)
#pragma unroll
        for (int ix = 0; ix < 4; ++ix) {
          const int oy = 4 * ty + iy, ox = 4 * tx + ix;
          if (oy < Ho && ox < Wo) seg_amax_add(am, kv, seg_vec_store(sv, o[iy][ix], b, (long)oy * Wo + ox));
        }
      continue;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int n = n4 * 4 + e;
      if (n >= Cout) continue;
      float *ptr = nullptr; long bs = 0; int rs = 0, act = 0, n0 = 0, ks = 0;
#pragma unroll
      for (int k = 0; k < 3; ++k)
        if (k < st.nseg && n >= st.seg[k].n0 && n < st.seg[k].n1) {
          ptr = st.seg[k].ptr; bs = st.seg[k].batch_stride; rs = st.seg[k].row_stride; act = st.seg[k].act; n0 = st.seg[k].n0; ks = k;
        }
      if (!ptr) continue;
      const float sc = scale ? scale[n] : 1.f, bi = bias ? bias[n] : 0.f;
#pragma unroll
      for (int iy = 0; iy < 4; ++iy) {
        const int oy = 4 * ty + iy;
        if (oy >= Ho) continue;
#pragma unroll
        for (int ix = 0; ix < 4; ++ix) {
          const int ox = 4 * tx + ix;
          if (ox >= Wo) continue;
          const float val = wino_act(o[iy][ix][e] * sc + bi, act);
          seg_amax_add(am, ks, fabsf(val));
          ptr[b * bs + ((long)oy * Wo + ox) * rs + (n - n0)] = val;
        }
      }
    }
  }
  seg_amax_commit(y_amax, st.nseg, apre, am);
}

unsigned grid_for(long total) {
  long g = (total + 255) / 256;
  const long cap = 256L * 64;
  return (unsigned)(g > cap ? cap : g);
}

}  // namespace

extern "C" int ymi_conv3x3_winograd_f32(const ymi_wino_desc *d, void *stream) {
  if (!d || (!d->x && !d->x_up) || !d->u || !d->V || !d->M) return YMI_ENULL;
  if (d->x_up && (d->m != 4 || (d->H & 1) || (d->W & 1) || (((uintptr_t)d->x_up) & 15))) return YMI_ESHAPE;   // fused 2x upsampling: F(4x4) only
  if (d->B <= 0 || d->H <= 0 || d->W <= 0 || d->C <= 0 || d->Cout <= 0 || d->nseg < 0 || d->nseg > 3) return YMI_EARG;
  const bool proj = d->proj_w_h2 != nullptr;     // fused 1x1 consumer: F(4x4), 256 dense output channels, <= 32 projected ones
  if (proj && (d->m != 4 || d->nseg != 0 || d->Cout != 256 || d->proj_cout <= 0 || d->proj_cout > 32 || (d->proj_cout & 3) ||
               (d->proj_ldy & 3) || d->proj_ldy < d->proj_cout || d->proj_act > YMI_ACT_LEAKY01 || d->proj_act < 0))
    return YMI_ESHAPE;
  if (proj && (!d->proj_scale_h2 || !d->proj_y || (((uintptr_t)d->proj_w_h2) & 15) || (((uintptr_t)d->proj_y) & 15))) return YMI_ENULL;
  if (d->nseg == 0 && ((!d->y && !proj) || (d->Cout & 3) || d->act > YMI_ACT_LEAKY01 || d->act < 0)) return YMI_ESHAPE;
  for (int k = 0; k < d->nseg; ++k) if (!d->seg[k].ptr) return YMI_ENULL;
  if (d->C & 31) return YMI_ESHAPE;
  if (d->m != 0 && d->m != 2 && d->m != 4) return YMI_EARG;
  const int mt = d->m == 4 ? 4 : 2;           // output tile edge: F(2x2,3x3) or F(4x4,3x3)
  const int ng = (mt + 2) * (mt + 2);         // independent GEMMs (16 or 36)
  const int Ng = (d->Cout + 3) / 4 * 4;       // GEMM width: zero filter rows up to a multiple of 4
  hipStream_t s = (hipStream_t)stream;
  const int th = (d->H + mt - 1) / mt, tw = (d->W + mt - 1) / mt;      // output size == input size (3x3, stride 1, pad 1)
  const long T = (long)d->B * th * tw;
  if (T * (long)(d->C > Ng ? d->C : Ng) >= (1L << 29)) return YMI_ESHAPE;   // per-group tensors < 2 GiB
  const int C4 = d->C / 4, N4 = Ng / 4;
  // profiling: record kind 3 (F(2x2)) / 4 (F(4x4)) = the whole layer (3 launches) with the layer's ALGORITHMIC FLOPs (2*9*C*Cout per output
  // pixel, like the direct kernel); record kind 5 / 6 (inside the GEMM launch) = the 16- / 36-group GEMM alone with the FLOPs
  // it executes (2*ng*T*C*Cout = algorithmic / 2.25 resp. / 4 for sizes that are multiples of the tile)
  const double alg = 2.0 * d->B * d->H * d->W * (double)(d->cout_alg > 0 ? d->cout_alg : d->Cout) * 9.0 * d->C;
  const double exe = 2.0 * ng * (double)T * d->C * Ng;
  const int outer = ymi_internal_prof_begin(alg, d->tile ? d->tile : YMI_TILE_64x64, mt == 4 ? 4 : 3, s);
  const bool h2 = (d->tile & YMI_TILE_H2) != 0;
  if (h2 && (!d->u_h2 || !d->uinv_h2 || !d->x_amax)) return YMI_ENULL;
  const bool planes = h2 && d->v_planes != 0;
#ifdef YMI_DIAGNOSTICS   // diagnostics build only (env YMI_WINO_ABLATE bit0): no input-transform launch — V keeps the previous run's values, every
                         // other launch of the step is unchanged: the measured ceiling of fusing the input transform away (wrong results by design)
  static const int wabl = [] { const char *e = getenv("YMI_WINO_ABLATE"); return e ? atoi(e) : 0; }();
  if (!(wabl & 1)) {
#endif
  if (mt == 4) {
    if (d->x_up) {
      if (planes) hipLaunchKernelGGL((wino43_in_k<true, true>), dim3(grid_for(T * C4)), dim3(256), 0, s, d->x_up, d->V, d->H, d->W, C4, th, tw, T, T * C4, d->x_amax, d->up_relu);
      else hipLaunchKernelGGL((wino43_in_k<false, true>), dim3(grid_for(T * C4)), dim3(256), 0, s, d->x_up, d->V, d->H, d->W, C4, th, tw, T, T * C4, d->x_amax, d->up_relu);
    } else if (planes) hipLaunchKernelGGL((wino43_in_k<true, false>), dim3(grid_for(T * C4)), dim3(256), 0, s, d->x, d->V, d->H, d->W, C4, th, tw, T, T * C4, d->x_amax, 0);
    else hipLaunchKernelGGL((wino43_in_k<false, false>), dim3(grid_for(T * C4)), dim3(256), 0, s, d->x, d->V, d->H, d->W, C4, th, tw, T, T * C4, d->x_amax, 0);
  } else {
    if (planes) hipLaunchKernelGGL(wino_in_k<true>, dim3(grid_for(T * C4)), dim3(256), 0, s, d->x, d->V, d->H, d->W, C4, th, tw, T, T * C4, d->x_amax);
    else hipLaunchKernelGGL(wino_in_k<false>, dim3(grid_for(T * C4)), dim3(256), 0, s, d->x, d->V, d->H, d->W, C4, th, tw, T, T * C4, d->x_amax);
  }
#ifdef YMI_DIAGNOSTICS
  }
#endif
  int rc = ymi_launch_status();
  if (rc) return rc;
  ymi_conv_desc g = {};
  g.x = d->V; g.w = d->u;
  g.B = 1; g.H = (int)T; g.W = 1; g.Cin = d->C; g.ldx = d->C;
  g.Ho = (int)T; g.Wo = 1; g.Cout = Ng;
  g.kh = g.kw = 1; g.stride = 1; g.pad = 0; g.Kpad = d->C;
  g.nseg = 1; g.tile = d->tile;
  g.w_x3 = d->u_x3;                    // [G][3][CoutPad][C] bf16 planes (optional)
  // fp16x2: U as [G][2][CoutPad][C] fp16 planes with a scale per (component, filter row); V scaled by the power of two derived
  // from the input tensor's magnitude bound times the transform's gain bound (|B^T d B| <= 4 resp. 100 max|d|)
  g.w_h2 = d->u_h2; g.scale_h2 = d->uinv_h2; g.winv_h2 = d->uinv_h2;
  g.x_amax = d->x_amax; g.x_amax_mul = mt == 4 ? 100.f : 4.f;
  g.seg[0].n0 = 0; g.seg[0].n1 = Ng; g.seg[0].act = YMI_ACT_NONE; g.seg[0].row_stride = Ng;
  g.seg[0].batch_stride = T * Ng; g.seg[0].ptr = d->M;
  const long cout_pad = ((long)d->Cout + 127) / 128 * 128;
  if ((d->tile & 31) == YMI_TILE_WG_128x256) {     // the persistent producer / consumer grouped GEMM (csrc/wgemm.hip): planes only
    if (!planes || (d->tile & YMI_TILE_X3)) return YMI_EARG;
    rc = ymi_internal_wgemm(d->V, d->u_h2, d->uinv_h2, d->x_amax, mt == 4 ? 100.f : 4.f, d->M, ng, T, d->C, Ng, (int)cout_pad, exe,
                            mt == 4 ? 6 : 5, s);
  } else if (planes)      // V = [G][2][T][C] fp16: both GEMM operands arrive pre-split (conv_igemm PREC 4)
    rc = ymi_internal_grouped_gemm(&g, ng, T * d->C, cout_pad * d->C, T * Ng, exe, mt == 4 ? 6 : 5, s, d->V,
                                   (unsigned)(T * d->C * 2), 2L * T * d->C * 2, (unsigned)cout_pad);
  else
    rc = ymi_internal_grouped_gemm(&g, ng, T * d->C, cout_pad * d->C, T * Ng, exe, mt == 4 ? 6 : 5 /* kind: winograd GEMM */, s,
                                   nullptr, 0, 0, (unsigned)cout_pad);
  if (rc) return rc;
  if (d->nseg > 0) {
    SegTab st;
    st.nseg = d->nseg;
    for (int k = 0; k < 3; ++k) st.seg[k] = d->seg[k];
    if (mt == 4)
      hipLaunchKernelGGL(wino43_out_seg_k, dim3(grid_for(T * N4)), dim3(256), 0, s, d->M, st, d->scale, d->bias, d->H, d->W,
                         N4, d->Cout, th, tw, T, T * N4, d->y_amax);
    else
      hipLaunchKernelGGL(wino_out_seg_k, dim3(grid_for(T * N4)), dim3(256), 0, s, d->M, st, d->scale, d->bias, d->H, d->W, N4,
                         d->Cout, th, tw, T, T * N4, d->y_amax);
    rc = ymi_launch_status();
    ymi_internal_prof_end(outer, s);
    return rc;
  }
  if (proj) {          // output transform + the consuming 1x1 convolution in one launch; y (the 3x3's output) is not written
    ProjArgs pa;
    pa.w_h2 = d->proj_w_h2; pa.scale_h2 = d->proj_scale_h2; pa.bias = d->proj_bias; pa.y = d->proj_y; pa.y_amax = d->proj_y_amax;
    pa.cout = d->proj_cout; pa.ldy = d->proj_ldy; pa.act = d->proj_act; pa.w_plane = (unsigned)(128L * 256 * 2);
    const long blocks = (T + 3) / 4;
    hipLaunchKernelGGL(wino43_out_proj_k, dim3((unsigned)(blocks < 256 ? blocks : 256)), dim3(256), 0, s, d->M, d->scale, d->bias, d->H,
                       d->W, th, tw, T, d->act, pa);
    rc = ymi_launch_status();
    ymi_internal_prof_end(outer, s);
    if (rc == YMI_OK) {      // the 1x1 layer keeps its place in the profile: a record of its algorithmic FLOPs and no duration of its own
      const int pr = ymi_internal_prof_begin(2.0 * d->B * d->H * d->W * 256.0 * d->proj_cout, d->tile ? d->tile : YMI_TILE_64x64, 12, s);
      ymi_internal_prof_end(pr, s);
    }
    return rc;
  }
  if (mt == 4)
    hipLaunchKernelGGL(wino43_out_k, dim3(grid_for(T * N4)), dim3(256), 0, s, d->M, d->y, d->scale, d->bias, d->H, d->W, N4, th,
                       tw, T, d->act, T * N4, d->y_amax);
  else
    hipLaunchKernelGGL(wino_out_k, dim3(grid_for(T * N4)), dim3(256), 0, s, d->M, d->y, d->scale, d->bias, d->H, d->W, N4, th,
                       tw, T, d->act, T * N4, d->y_amax);
  rc = ymi_launch_status();
  ymi_internal_prof_end(outer, s);
  return rc;
}
