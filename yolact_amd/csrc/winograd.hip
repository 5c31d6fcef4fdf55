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
                              int prof_kind, hipStream_t s);
int ymi_internal_prof_begin(double flops, int tile, int kind, hipStream_t s);
void ymi_internal_prof_end(int idx, hipStream_t s);

namespace {

__device__ __forceinline__ f32x4 ld4(const float *p, bool ok) {
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  return ok ? *reinterpret_cast<const f32x4 *>(p) : z;
}

// one thread = one tile x 4 channels.  x [B,H,W,C] NHWC, V [16][T][C]
__global__ __launch_bounds__(256) void wino_in_k(const float *__restrict__ x, float *__restrict__ V, int H, int W, int C4,
                                                 int th, int tw, long T, long total) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
    const int c4 = (int)(i % C4);
    const long t = i / C4;
    const int tx = (int)(t % tw);
    const long r = t / tw;
    const int ty = (int)(r % th);
    const long b = r / th;
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
#pragma unroll
    for (int iy = 0; iy < 4; ++iy) {       // columns: v = u B
      const f32x4 v0 = u[iy][0] - u[iy][2], v1 = u[iy][1] + u[iy][2], v2 = u[iy][2] - u[iy][1], v3 = u[iy][1] - u[iy][3];
      *reinterpret_cast<f32x4 *>(o + (iy * 4 + 0) * stride_e) = v0;
      *reinterpret_cast<f32x4 *>(o + (iy * 4 + 1) * stride_e) = v1;
      *reinterpret_cast<f32x4 *>(o + (iy * 4 + 2) * stride_e) = v2;
      *reinterpret_cast<f32x4 *>(o + (iy * 4 + 3) * stride_e) = v3;
    }
  }
}

// one thread = one tile x 4 output channels.  M [16][T][N], y [B,Ho,Wo,N]
__global__ __launch_bounds__(256) void wino_out_k(const float *__restrict__ Mm, float *__restrict__ y,
                                                  const float *__restrict__ scale, const float *__restrict__ bias, int Ho,
                                                  int Wo, int N4, int th, int tw, long T, int act, long total) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
    const int n4 = (int)(i % N4);
    const long t = i / N4;
    const int tx = (int)(t % tw);
    const long r = t / tw;
    const int ty = (int)(r % th);
    const long b = r / th;
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
#pragma unroll
    for (int iy = 0; iy < 2; ++iy) {
      const f32x4 o0 = (s[iy][0] + s[iy][1]) + s[iy][2], o1 = (s[iy][1] - s[iy][2]) - s[iy][3];
      const int oy = 2 * ty + iy;
      if (oy >= Ho) continue;
#pragma unroll
      for (int ix = 0; ix < 2; ++ix) {
        const int ox = 2 * tx + ix;
        if (ox >= Wo) continue;
        f32x4 v = (ix == 0 ? o0 : o1) * sc + bi;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], slope * v[e]);
        *reinterpret_cast<f32x4 *>(y + ((b * Ho + oy) * (long)Wo + ox) * (N4 * 4L) + n4 * 4) = v;
      }
    }
  }
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

__global__ __launch_bounds__(256) void wino_out_seg_k(const float *__restrict__ Mm, const SegTab st,
                                                      const float *__restrict__ scale, const float *__restrict__ bias,
                                                      int Ho, int Wo, int N4, int Cout, int th, int tw, long T, long total) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
    const int n4 = (int)(i % N4);
    const long t = i / N4;
    const int tx = (int)(t % tw);
    const long r = t / tw;
    const int ty = (int)(r % th);
    const long b = r / th;
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
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int n = n4 * 4 + e;
      if (n >= Cout) continue;
      // segment of channel n: compile-time indices + selects (the table lives in SGPRs)
      float *ptr = nullptr; long bs = 0; int rs = 0, act = 0, n0 = 0;
#pragma unroll
      for (int k = 0; k < 3; ++k)
        if (k < st.nseg && n >= st.seg[k].n0 && n < st.seg[k].n1) {
          ptr = st.seg[k].ptr; bs = st.seg[k].batch_stride; rs = st.seg[k].row_stride; act = st.seg[k].act; n0 = st.seg[k].n0;
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
          ptr[b * bs + ((long)oy * Wo + ox) * rs + (n - n0)] = wino_act(o[iy][ix][e] * sc + bi, act);
        }
      }
    }
  }
}

unsigned grid_for(long total) {
  long g = (total + 255) / 256;
  const long cap = 256L * 64;
  return (unsigned)(g > cap ? cap : g);
}

}  // namespace

extern "C" int ymi_conv3x3_winograd_f32(const ymi_wino_desc *d, void *stream) {
  if (!d || !d->x || !d->u || !d->V || !d->M) return YMI_ENULL;
  if (d->B <= 0 || d->H <= 0 || d->W <= 0 || d->C <= 0 || d->Cout <= 0 || d->nseg < 0 || d->nseg > 3) return YMI_EARG;
  if (d->nseg == 0 && (!d->y || (d->Cout & 3) || d->act > YMI_ACT_LEAKY01 || d->act < 0)) return YMI_ESHAPE;
  for (int k = 0; k < d->nseg; ++k) if (!d->seg[k].ptr) return YMI_ENULL;
  if (d->C & 31) return YMI_ESHAPE;
  const int Ng = (d->Cout + 3) / 4 * 4;       // GEMM width: zero filter rows up to a multiple of 4
  hipStream_t s = (hipStream_t)stream;
  const int th = (d->H + 1) / 2, tw = (d->W + 1) / 2;      // output size == input size (3x3, stride 1, pad 1)
  const long T = (long)d->B * th * tw;
  if (T * (long)(d->C > Ng ? d->C : Ng) >= (1L << 29)) return YMI_ESHAPE;   // per-group tensors < 2 GiB
  const int C4 = d->C / 4, N4 = Ng / 4;
  // profiling: record kind 3 = the whole layer (3 launches) with the layer's ALGORITHMIC FLOPs (2*9*C*Cout per output
  // pixel, like the direct kernel); record kind 5 (inside the GEMM launch) = the 16-group GEMM alone with the FLOPs it
  // executes (2*16*T*C*Cout = algorithmic / 2.25 for even sizes)
  const double alg = 2.0 * d->B * d->H * d->W * (double)d->Cout * 9.0 * d->C;
  const double exe = 2.0 * 16.0 * (double)T * d->C * Ng;
  const int outer = ymi_internal_prof_begin(alg, d->tile ? d->tile : YMI_TILE_64x64, 3, s);
  hipLaunchKernelGGL(wino_in_k, dim3(grid_for(T * C4)), dim3(256), 0, s, d->x, d->V, d->H, d->W, C4, th, tw, T, T * C4);
  int rc = ymi_launch_status();
  if (rc) return rc;
  ymi_conv_desc g = {};
  g.x = d->V; g.w = d->u;
  g.B = 1; g.H = (int)T; g.W = 1; g.Cin = d->C; g.ldx = d->C;
  g.Ho = (int)T; g.Wo = 1; g.Cout = Ng;
  g.kh = g.kw = 1; g.stride = 1; g.pad = 0; g.Kpad = d->C;
  g.nseg = 1; g.tile = d->tile;
  g.seg[0].n0 = 0; g.seg[0].n1 = Ng; g.seg[0].act = YMI_ACT_NONE; g.seg[0].row_stride = Ng;
  g.seg[0].batch_stride = T * Ng; g.seg[0].ptr = d->M;
  const long cout_pad = ((long)d->Cout + 127) / 128 * 128;
  rc = ymi_internal_grouped_gemm(&g, 16, T * d->C, cout_pad * d->C, T * Ng, exe, 5 /* kind: winograd GEMM */, s);
  if (rc) return rc;
  if (d->nseg > 0) {
    SegTab st;
    st.nseg = d->nseg;
    for (int k = 0; k < 3; ++k) st.seg[k] = d->seg[k];
    hipLaunchKernelGGL(wino_out_seg_k, dim3(grid_for(T * N4)), dim3(256), 0, s, d->M, st, d->scale, d->bias, d->H, d->W, N4,
                       d->Cout, th, tw, T, T * N4);
    rc = ymi_launch_status();
    ymi_internal_prof_end(outer, s);
    return rc;
  }
  hipLaunchKernelGGL(wino_out_k, dim3(grid_for(T * N4)), dim3(256), 0, s, d->M, d->y, d->scale, d->bias, d->H, d->W, N4, th,
                     tw, T, d->act, T * N4);
  rc = ymi_launch_status();
  ymi_internal_prof_end(outer, s);
  return rc;
}
