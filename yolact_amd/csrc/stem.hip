// ResNet stem in ONE launch: the NCHW fp32 image -> conv1 7x7 / stride 2 / pad 3 (3 -> 64) -> eval BatchNorm -> ReLU -> max-pool
// 3x3 / stride 2 / pad 1 -> NHWC fp32 (backbone.py:126-133 + the layout change at the entry of Yolact.forward, yolact.py:564).
// The separate launches (layout change + magnitude bound, implicit-GEMM stem with a per-lane tap gather, max-pool) move the
// 64-channel 275 x 275 stem output twice through HBM (155 MB written, 155 MB read at batch 8) and spend 0.18 ms of a 4.5 ms step
// (r03 kernel stats: 22 + 120 + 40 us); here the stem output never leaves the CU.
//
// A persistent workgroup (512 threads, one per CU) keeps the filters in LDS as the conv engine's two fp16 planes (fp16x2
// arithmetic: three v_mfma_f32_32x32x16_f16 per product, fp32 accumulate) and loops over tiles of 4 x 6 POOLED pixels:
//   * the 23 x 31 x 3 input patch of the tile is fetched straight from the NCHW image (coalesced rows), split into two fp16
//     planes of [pixel][4 channels] with a power-of-two scale from the PATCH's own maximum (exact to undo per tile; no separate
//     magnitude-bound pass over the input), zero outside the image (the convolution's padding);
//   * implicit GEMM [128 (9 x 13 stem pixels) x 224] x [224 x 64], k = (7 ky + kx) * 4 + c exactly as engine.Packed lays the stem
//     filters out: an A fragment = two taps x 4 channels = two 8-byte LDS reads per plane at compile-time tap offsets;
//   * folded BN + ReLU into an fp32 tile in LDS (-inf for stem pixels outside the 275 x 275 stem image: the pool's padding),
//     3 x 3 / stride 2 maximum, NHWC store, magnitude bound of the output for the next layer.
// The next tile's patch is requested before the current tile's MFMAs.
#include "common.h"
#include <stdlib.h>
#include "../../include/yolact_amd.h"

int ymi_internal_prof_begin(double flops, int tile, int kind, hipStream_t s);
void ymi_internal_prof_end(int idx, hipStream_t s);

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int NTHR = 512, NWAVE = 8;
constexpr int PH = 4, PW = 6;                         // pooled pixels per tile
constexpr int SH = 2 * PH + 1, SW = 2 * PW + 1;       // stem pixels per tile: 9 x 13 = 117 (rows of the GEMM, padded to 128)
constexpr int IH = 2 * SH + 5, IW = 2 * SW + 5;       // input patch: 23 x 31
constexpr int IPITCH = 32;                            // patch row pitch in pixels
constexpr int NPATCH = IH * IPITCH;                   // 736 patch slots
constexpr int KPAD = 224, KSTEPS = KPAD / 16;         // k = tap * 4 + c, 49 taps -> 196, padded like engine.Packed (Kpad 224)
constexpr int WPITCH = KPAD * 2 + 16;                 // bytes per filter row in LDS (464: conflict-free 16-byte fragment reads)
constexpr int COUT = 64;
constexpr int EPITCH = COUT + 1;                      // floats per row of the stem-output tile

struct StemParams {
  const float *x; float *y;
  int B, H, W, Hs, Ws, Hp, Wp, tiles_x, tiles_y, ntiles;
  const unsigned short *w;      // fp16 planes [2][cout_pad][KPAD]
  int cout_pad;
  const float *scale, *bias;    // folded BN scale / the filter row's power-of-two scale; folded bias
  float *y_amax;
};

__device__ __forceinline__ int crow(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

__global__ __launch_bounds__(NTHR) void stem_pool_k(const StemParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int WB = 0, WPLANE = COUT * WPITCH;                    // filter planes
  constexpr int PB = 2 * WPLANE, PPLANE = NPATCH * 8;              // patch planes: 8 bytes (4 fp16) per pixel
  constexpr int EB = PB + 2 * PPLANE;                              // stem-output tile [128][EPITCH] fp32
  constexpr int RED = EB + 128 * EPITCH * 4;
  constexpr int LDS_BYTES = RED + 64;
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];

  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int hh = lane >> 5;
  const ymi_amax_pre apre = ymi_amax_prefetch(p.y_amax);

  // ---- filters -> LDS, once per workgroup -----------------------------------------------------------------------------
  for (int u = t; u < 2 * COUT * (KPAD / 8); u += NTHR) {           // unit = (plane, row, 8 fp16)
    const int plane = u / (COUT * (KPAD / 8)), r = u - plane * (COUT * (KPAD / 8));
    const int row = r / (KPAD / 8), g8 = r - row * (KPAD / 8);
    const f16x8 v = *reinterpret_cast<const f16x8 *>(p.w + ((size_t)plane * p.cout_pad + row) * KPAD + 8 * g8);
    *reinterpret_cast<f16x8 *>(lds + WB + plane * WPLANE + row * WPITCH + g8 * 16) = v;
  }
  const int mt = wave & 3, nt = wave >> 2;                          // this wave's 32 x 32 output tile
  const int ncol = nt * 32 + (lane & 31);
  const float sc = p.scale[ncol], bi = p.bias[ncol];
  // A fragment geometry of this lane: GEMM row m = stem pixel (i, j) of the tile
  const int m = mt * 32 + (lane & 31);
  const int mi = m < SH * SW ? m / SW : 0, mj = m < SH * SW ? m - (m / SW) * SW : 0;
  const int abase = ((2 * mi) * IPITCH + 2 * mj) * 8;               // byte offset of the patch pixel under tap (0, 0)
  float *red = reinterpret_cast<float *>(lds + RED);
  float *et = reinterpret_cast<float *>(lds + EB);

  // patch slots of this thread: slot = r * 32 + c (c < 31), two per thread
  auto fetch = [&](int tile, float (&v)[2][3]) {
    const int b = tile / (p.tiles_x * p.tiles_y), tt = tile - b * (p.tiles_x * p.tiles_y);
    const int ty = tt / p.tiles_x, tx = tt - ty * p.tiles_x;
    const int iy0 = 4 * (ty * PH) - 5, ix0 = 4 * (tx * PW) - 5;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int slot = t + NTHR * k, r = slot >> 5, c = slot & 31;
      const int iy = iy0 + r, ix = ix0 + c;
      const bool ok = slot < NPATCH && c < IW && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      const float *src = p.x + ((size_t)b * 3 * p.H + (ok ? iy : 0)) * p.W + (ok ? ix : 0);
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) v[k][ch] = ok ? src[(size_t)ch * p.H * p.W] : 0.f;
    }
  };
  float cur[2][3], nxt[2][3];
  int tile = blockIdx.x;
  if (tile < p.ntiles) fetch(tile, cur);
  float amy = 0.f;
  __syncthreads();                                                  // filters in place
  for (; tile < p.ntiles; tile += gridDim.x) {
    const int b = tile / (p.tiles_x * p.tiles_y), tt = tile - b * (p.tiles_x * p.tiles_y);
    const int ty = tt / p.tiles_x, tx = tt - ty * p.tiles_x;
    // ---- patch: local maximum -> power-of-two scale -> fp16 planes ------------------------------------------------------
    float am = 0.f;
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) am = fmaxf(am, fabsf(cur[k][ch]));
    {
      const unsigned mx = ymi_wave_umax63(__float_as_uint(am));
      if (lane == 63) red[wave] = __uint_as_float(mx);
    }
    __syncthreads();                                                // (also: the previous tile's pooling is done with `et`)
    float pm = 0.f;
#pragma unroll
    for (int w = 0; w < NWAVE; ++w) pm = fmaxf(pm, red[w]);
    float s, inv;
    ymi_h2_scale(pm, s, inv);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int slot = t + NTHR * k;
      if (slot < NPATCH) {
        f16x4 h, l;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
          const float tv = cur[k][ch] * s;
          h[ch] = (_Float16)tv;
          l[ch] = (_Float16)(tv - (float)h[ch]);
        }
        h[3] = (_Float16)0.f; l[3] = (_Float16)0.f;
        *reinterpret_cast<f16x4 *>(lds + PB + slot * 8) = h;
        *reinterpret_cast<f16x4 *>(lds + PB + PPLANE + slot * 8) = l;
      }
    }
    const int next = tile + gridDim.x;
    if (next < p.ntiles) fetch(next, nxt);                          // in flight under the MFMAs
    __syncthreads();                                                // patch planes published
    // ---- implicit GEMM: 14 k-steps of (2 taps x 4 channels per lane half) ------------------------------------------------
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const unsigned char *wrow = lds + WB + ncol * WPITCH;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      // taps of this k-step: lane half 0 -> 4 ks, 4 ks + 1; half 1 -> 4 ks + 2, 4 ks + 3 (taps >= 49 are zero filters: clamp)
      constexpr int NT = 49;
      const int ta0 = 4 * ks < NT ? 4 * ks : NT - 1, ta1 = 4 * ks + 1 < NT ? 4 * ks + 1 : NT - 1;
      const int tb0 = 4 * ks + 2 < NT ? 4 * ks + 2 : NT - 1, tb1 = 4 * ks + 3 < NT ? 4 * ks + 3 : NT - 1;
      const int oa0 = ((ta0 / 7) * IPITCH + ta0 % 7) * 8, oa1 = ((ta1 / 7) * IPITCH + ta1 % 7) * 8;
      const int ob0 = ((tb0 / 7) * IPITCH + tb0 % 7) * 8, ob1 = ((tb1 / 7) * IPITCH + tb1 % 7) * 8;
      const int o0 = abase + (hh ? ob0 : oa0), o1 = abase + (hh ? ob1 : oa1);
      const f16x4 h0 = *reinterpret_cast<const f16x4 *>(lds + PB + o0), h1 = *reinterpret_cast<const f16x4 *>(lds + PB + o1);
      const f16x4 l0 = *reinterpret_cast<const f16x4 *>(lds + PB + PPLANE + o0), l1 = *reinterpret_cast<const f16x4 *>(lds + PB + PPLANE + o1);
      const f16x8 ah = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
      const f16x8 al = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
      const f16x8 bh = *reinterpret_cast<const f16x8 *>(wrow + (16 * ks + 8 * hh) * 2);
      const f16x8 bl = *reinterpret_cast<const f16x8 *>(wrow + WPLANE + (16 * ks + 8 * hh) * 2);
      acc = ymi_mfma32(ah, bl, acc);
      acc = ymi_mfma32(al, bh, acc);
      acc = ymi_mfma32(ah, bh, acc);
    }
    // ---- BN + ReLU -> stem tile (fp32); stem pixels outside the stem image = -inf (the max-pool's padding) ----------------
    const int sy0 = 2 * (ty * PH) - 1, sx0 = 2 * (tx * PW) - 1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = mt * 32 + crow(r, lane);
      const int i = row / SW, j = row - i * SW;
      const bool ok = row < SH * SW && (unsigned)(sy0 + i) < (unsigned)p.Hs && (unsigned)(sx0 + j) < (unsigned)p.Ws;
      float v = (acc[r] * inv) * sc + bi;
      v = v > 0.f ? v : 0.f;
      et[row * EPITCH + ncol] = ok ? v : -__builtin_inff();
    }
    __syncthreads();
    // ---- 3 x 3 / stride 2 maximum: thread = channel, pooled pixels q = (t >> 6) + 8 k -------------------------------------
    const int c = t & 63;
#pragma unroll
    for (int k = 0; k < (PH * PW + 7) / 8; ++k) {
      const int q = (t >> 6) + 8 * k;
      if (q < PH * PW) {
        const int qy = q / PW, qx = q - qy * PW;
        const int py = ty * PH + qy, px = tx * PW + qx;
        float mx = -__builtin_inff();
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) mx = fmaxf(mx, et[((2 * qy + dy) * SW + 2 * qx + dx) * EPITCH + c]);
        if (py < p.Hp && px < p.Wp) {
          p.y[(((size_t)b * p.Hp + py) * p.Wp + px) * COUT + c] = mx;
          amy = fmaxf(amy, fabsf(mx));
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) cur[k][ch] = nxt[k][ch];
  }
  if (p.y_amax) ymi_amax_finish(apre, amy);
#endif
}

}  // namespace

extern "C" int ymi_stem_pool_f32(const ymi_stem_desc *d, void *stream) {
  if (!d) return YMI_ENULL;
  if (!d->x || !d->y || !d->w_h2 || !d->scale_h2 || !d->bias) return YMI_ENULL;
  if (d->B <= 0 || d->H < 7 || d->W < 7) return YMI_EARG;
  if (d->cout_pad < COUT || d->kpad != KPAD) return YMI_ESHAPE;
  if ((((uintptr_t)d->w_h2) | ((uintptr_t)d->y)) & 15) return YMI_ESHAPE;
  StemParams p;
  p.x = d->x; p.y = d->y; p.B = d->B; p.H = d->H; p.W = d->W;
  p.Hs = (d->H + 6 - 7) / 2 + 1; p.Ws = (d->W + 6 - 7) / 2 + 1;
  p.Hp = (p.Hs + 2 - 3) / 2 + 1; p.Wp = (p.Ws + 2 - 3) / 2 + 1;
  if ((long)d->B * 3 * d->H * d->W >= (1L << 31) || (long)d->B * p.Hp * p.Wp * COUT >= (1L << 31)) return YMI_ESHAPE;
  p.tiles_x = (p.Wp + PW - 1) / PW; p.tiles_y = (p.Hp + PH - 1) / PH;
  p.ntiles = p.tiles_x * p.tiles_y * d->B;
  p.w = (const unsigned short *)d->w_h2; p.cout_pad = d->cout_pad;
  p.scale = d->scale_h2; p.bias = d->bias; p.y_amax = d->y_amax;
  hipStream_t s = (hipStream_t)stream;
  const double flops = 2.0 * d->B * p.Hs * p.Ws * 64.0 * 147.0;      // the stem's algorithmic FLOPs (Cin = 3)
  const int pr = ymi_internal_prof_begin(flops, YMI_TILE_H2 | YMI_TILE_64x64, 8, s);
  int dev = 0, cus = 256;
  hipGetDevice(&dev);
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const int grid = p.ntiles < cus ? p.ntiles : cus;
  hipLaunchKernelGGL(stem_pool_k, dim3(grid), dim3(NTHR), 0, s, p);
  const int rc = ymi_launch_status();
  ymi_internal_prof_end(pr, s);
  return rc;
}
