// 3x3 / stride 1 / pad 1 convolution, 64 -> 64 channels, with the INPUT PATCH of a spatial tile held in LDS and the FILTERS held in
// registers (round 5) — conv2 of the first ResNet stage (backbone.py:44-46: 3x3 + bn2 + ReLU at 138 x 138 x 64, three launches per
// batch-8 step, the same shape in every ResNet config).
//
// Why this layer needs its own kernel.  As an implicit GEMM (csrc/dcn.hip pipe_h2_k, 128 x 64 tiles) its A operand is re-read from
// L2 once per filter tap: 9 x 39 MB + the filters once per row tile = 526 MB through the global -> LDS path for 78 MB of tensor and
// 11.2 GFLOP — 62 us, i.e. the 8.5 TB/s that path delivers, against 16 us of HBM time and 19 us of matrix-pipe time.  Here a block
// owns an 8 x 16 pixel output tile: the 10 x 18 x 64 input patch is loaded ONCE (46 KB), split into the two fp16 planes of the
// fp16x2 arithmetic and kept in LDS; the nine taps are nine shifted views of it.  The filters — 64 x 576 x 2 planes = 147 KB, more
// than LDS can hold next to the patch — live in REGISTERS for the block's lifetime (persistent blocks, one per CU): wave (q, i) owns
// output channels 16 q .. 16 q + 15 (18 K-chunks x 2 planes x 4 VGPRs = 144 VGPRs of MFMA operand fragments, the scheme of
// csrc/chain.hip) and the four 16-pixel rows 4 i .. 4 i + 3 of the tile.  v_mfma_f32_16x16x32_f16 issued as W X^T: a lane ends with
// four consecutive output channels of one pixel (float4 epilogue, no transposition).  The next tile's patch is requested before the
// current tile's MFMAs and written to the other LDS buffer behind them: one barrier per tile.
// Arithmetic: the fp16x2 scheme of the engine (tensor scale from x_amax, h*l + l*h + h*h on the fp16 pipe, fp32 accumulate), K order
// tap-major like engine.Packed — the filter planes of Packed.h2() are used unchanged.
#include "common.h"
#include <type_traits>
#include "../../include/yolact_amd.h"

int ymi_internal_prof_begin(double flops, int tile, int kind, hipStream_t s);
void ymi_internal_prof_end(int idx, hipStream_t s);

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int C = 64, NW = 8, NT = 64 * NW;
constexpr int TH = 8, TW = 16, PH = TH + 2, PW = TW + 2, PPX = PH * PW;      // output tile, input patch (180 pixels)
constexpr int RS = 2 * C + 16;              // bytes per patch pixel in a plane: 128 + 16 (the 16 lanes of a read phase hit 16 bank groups)
constexpr int PLANE = PPX * RS, BUF = 2 * PLANE;                             // 25 920 B per plane, two planes per buffer
constexpr int NLOAD = (PPX * (C / 4) + NT - 1) / NT;                          // float4 loads per thread per patch: 6 (2880 / 512) ...
constexpr int NHALF = NLOAD / 2;                                              // ... requested and published in two halves of 3 (registers)
static_assert(NLOAD == 2 * NHALF, "patch loads split in two halves");
constexpr int OFF_EP = 2 * BUF;             // per-(wave, lane group) epilogue constants behind the two patch buffers: 8 x 4 x 32 B
constexpr int NCH = 9 * (C / 32);           // K chunks of 32: (tap, channel half) = 18

struct PatchParams {
  const float *x, *scale_h2, *bias, *x_amax;
  const void *w_h2;
  float *y, *y_amax;
  int B, H, W, ldx, ldy, act, tiles_x, tiles_y, ntiles;
  unsigned w_plane, x_bytes, y_bytes;
};

__global__ __launch_bounds__(NT) void patch3x3_c64_k(const PatchParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  __shared__ __attribute__((aligned(16))) char lds[2 * BUF + NW * 4 * 32];
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lr = lane & 15, g = lane >> 4;
  const int q = wave & 3, ih = wave >> 2;              // this wave's 16 output channels (16 q ..) and its four tile rows (4 ih ..)
  constexpr unsigned OOB = 0x80000000u;
  float sA, invA;
  ymi_h2_scale(ymi_amax_read(p.x_amax), sA, invA);
  const ymi_amax_pre apre = ymi_amax_prefetch(p.y_amax);

  // ---- this wave's filters -> registers, once: A-operand fragments (row = output channel 16 q + lr, k = 32 chunk + 8 g .. + 7) ------
  f16x8 wh[NCH], wl[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const char *src = reinterpret_cast<const char *>(p.w_h2) + (size_t)(16 * q + lr) * (2 * 9 * C) + c * 64 + g * 16;
    wh[c] = *reinterpret_cast<const f16x8 *>(src);
    wl[c] = *reinterpret_cast<const f16x8 *>(src + p.w_plane);
  }
  // epilogue constants of this lane's four channels (16 q + 4 g ..): parked in LDS, re-read per tile (8 registers the filters need)
  if (lr == 0) {
    f32x4 sc, bi = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      sc[e] = p.scale_h2[16 * q + 4 * g + e] * invA;   // folded BN scale / filter-row scale, times the exact 1 / sA
      if (p.bias) bi[e] = p.bias[16 * q + 4 * g + e];
    }
    *reinterpret_cast<f32x4 *>(lds + OFF_EP + (wave * 4 + g) * 32) = sc;
    *reinterpret_cast<f32x4 *>(lds + OFF_EP + (wave * 4 + g) * 32 + 16) = bi;
  }
  const float slope = p.act == YMI_ACT_RELU ? 0.f : (p.act == YMI_ACT_LEAKY01 ? 0.1f : 1.f);

  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.x, 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.y, 0, (int)p.y_bytes, 0x00020000);
  const int per_img = p.tiles_x * p.tiles_y;

  // patch loads: thread t takes float4 number t + NT * i of the patch (pixel (t + NT i) / 16, channels 4 ((t + NT i) % 16) ..);
  // pixels outside the image (the convolution's zero padding) and slots past the patch are out-of-bounds buffer offsets: zeros
  auto request = [&](int tile, auto half_c, f32x4 (&v)[NHALF]) {
    constexpr int HALF = decltype(half_c)::value;
    const bool live = tile < p.ntiles;
    const int b = tile / per_img, r = tile - b * per_img;
    const int ty = r / p.tiles_x, tx = r - ty * p.tiles_x;
    const int y0 = ty * TH - 1, x0 = tx * TW - 1;
#pragma unroll
    for (int i = 0; i < NHALF; ++i) {
      const int idx = t + NT * (i + HALF * NHALF), px = idx >> 4, cg = idx & 15;
      const int py = px / PW, pxx = px - py * PW;
      const int yy = y0 + py, xx = x0 + pxx;
      const bool in = live && px < PPX && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
      const unsigned off = in ? (unsigned)(((b * p.H + yy) * p.W + xx) * p.ldx + 4 * cg) * 4u : OOB;
      v[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, off, 0, 0));
    }
  };
  auto publish = [&](const f32x4 (&v)[NHALF], auto half_c, char *buf) {
    constexpr int HALF = decltype(half_c)::value;
#pragma unroll
    for (int i = 0; i < NHALF; ++i) {
      const int idx = t + NT * (i + HALF * NHALF), px = idx >> 4, cg = idx & 15;
      if (px < PPX) {
        const f32x4 s = v[i] * sA;
        f16x4 h4, l4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const _Float16 h = (_Float16)s[e];
          h4[e] = h;
          l4[e] = (_Float16)(s[e] - (float)h);
        }
        char *dst = buf + px * RS + cg * 8;
        *reinterpret_cast<f16x4 *>(dst) = h4;
        *reinterpret_cast<f16x4 *>(dst + PLANE) = l4;
      }
    }
  };
#define PATCH_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

  float am = 0.f;
  const int grid = (int)gridDim.x;
  using H0 = std::integral_constant<int, 0>;
  using H1 = std::integral_constant<int, 1>;
  f32x4 ld[NHALF];
  int tile = blockIdx.x;
  request(tile, H0{}, ld);
  publish(ld, H0{}, lds);
  request(tile, H1{}, ld);
  publish(ld, H1{}, lds);
  PATCH_BARRIER();
  for (int k = 0; tile < p.ntiles; tile += grid, ++k) {
    const char *const cur = lds + (k & 1) * BUF;
    char *const nxt = lds + ((k + 1) & 1) * BUF;          // last read one iteration ago: free for the whole of this one
    request(tile + grid, H0{}, ld);                      // first half of the next patch: in flight during taps 0 .. 3
    // ---- 18 K chunks x 4 tile rows: B operand = patch pixel (4 ih + rb + ky, lr + kx), channels 32 c + 8 g .. + 7 -----------------
    f32x4 acc[4];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) acc[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto taps = [&](auto t0_c, auto t1_c) {
#pragma unroll
      for (int tap = decltype(t0_c)::value; tap < decltype(t1_c)::value; ++tap) {
        const int ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int rp = 0; rp < 2; ++rp) {              // two tile rows at a time (fragment registers), three products each
            f16x8 xh[2], xl[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              const char *src = cur + ((4 * ih + 2 * rp + u + ky) * PW + lr + kx) * RS + c * 64 + g * 16;
              xh[u] = *reinterpret_cast<const f16x8 *>(src);
              xl[u] = *reinterpret_cast<const f16x8 *>(src + PLANE);
            }
#pragma unroll
            for (int pr = 0; pr < 3; ++pr)              // product-major: consecutive MFMAs belong to different accumulators
#pragma unroll
              for (int u = 0; u < 2; ++u)
                acc[2 * rp + u] = ymi_mfma16(pr == 0 ? wl[2 * tap + c] : wh[2 * tap + c], pr == 1 ? xl[u] : xh[u], acc[2 * rp + u]);
          }
      }
    };
    taps(std::integral_constant<int, 0>{}, std::integral_constant<int, 4>{});
    publish(ld, H0{}, nxt);
    request(tile + grid, H1{}, ld);                      // second half: in flight during taps 4 .. 8
    taps(std::integral_constant<int, 4>{}, std::integral_constant<int, 9>{});
    // ---- epilogue: lane = pixel (row 4 ih + rb, column lr) x channels 16 q + 4 g .. + 3 -----------------------------------------------
    {
      const int b = tile / per_img, r = tile - b * per_img;
      const int ty = r / p.tiles_x, tx = r - ty * p.tiles_x;
      const int ox = tx * TW + lr;
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) {
        const int oy = ty * TH + 4 * ih + rb;
        const bool ok = oy < p.H && ox < p.W;
        const f32x4 sc = *reinterpret_cast<const f32x4 *>(lds + OFF_EP + (wave * 4 + g) * 32);
        const f32x4 bi = *reinterpret_cast<const f32x4 *>(lds + OFF_EP + (wave * 4 + g) * 32 + 16);
        f32x4 v = acc[rb] * sc + bi;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], slope * v[e]);
        am = fmaxf(am, ok ? ymi_absmax4(v) : 0.f);
        const unsigned off = ok ? (unsigned)(((b * p.H + oy) * p.W + ox) * p.ldy + 16 * q + 4 * g) * 4u : OOB;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yrs, off, 0, 0);
      }
    }
    publish(ld, H1{}, nxt);
    PATCH_BARRIER();
  }
#undef PATCH_BARRIER
  if (p.y_amax) ymi_amax_finish(apre, am);
#endif
}

}  // namespace

// internal (called by ymi_conv2d_nhwc_f32 for tile YMI_TILE_H2 | YMI_TILE_DCNP | YMI_DCNP_PATCH_C64): 3x3 / stride 1 / pad 1, 64 -> 64,
// one dense output, no residual, activation none / ReLU / LeakyReLU.  YMI_EARG for anything else.  Profiling record kind 14.
int ymi_internal_patch_conv(const ymi_conv_desc *d, hipStream_t s) {
  const ymi_conv_seg &g0 = d->seg[0];
  if (d->kh != 3 || d->kw != 3 || d->stride != 1 || d->pad != 1 || d->Cin != C || d->Cout != C || d->Kpad != 9 * C) return YMI_EARG;
  if (d->Ho != d->H || d->Wo != d->W || d->nseg != 1 || g0.n0 != 0 || g0.n1 < C || g0.act < 0 || g0.act > YMI_ACT_LEAKY01) return YMI_EARG;
  if (d->res_mode != YMI_RES_NONE || d->split_k > 1) return YMI_EARG;
  if (!d->x || !g0.ptr || !d->w_h2 || !d->scale_h2 || !d->x_amax) return YMI_ENULL;
  if ((d->ldx & 3) || (g0.row_stride & 3) || d->ldx < C || g0.row_stride < C || g0.batch_stride != (int64_t)d->Ho * d->Wo * g0.row_stride)
    return YMI_ESHAPE;
  if ((((uintptr_t)d->x) | ((uintptr_t)g0.ptr) | ((uintptr_t)d->w_h2)) & 15) return YMI_ESHAPE;
  const long px = (long)d->B * d->H * d->W;
  if (px * d->ldx >= (1L << 29) || px * g0.row_stride >= (1L << 29)) return YMI_ESHAPE;       // 32-bit buffer offsets
  PatchParams p;
  p.x = d->x; p.scale_h2 = d->scale_h2; p.bias = d->bias; p.x_amax = d->x_amax; p.w_h2 = d->w_h2;
  p.y = g0.ptr; p.y_amax = d->y_amax;
  p.B = d->B; p.H = d->H; p.W = d->W; p.ldx = d->ldx; p.ldy = g0.row_stride; p.act = g0.act;
  p.tiles_x = (d->W + TW - 1) / TW; p.tiles_y = (d->H + TH - 1) / TH; p.ntiles = d->B * p.tiles_x * p.tiles_y;
  p.w_plane = (unsigned)((((long)d->Cout + 127) / 128 * 128) * d->Kpad * 2L);
  p.x_bytes = (unsigned)(px * d->ldx * 4); p.y_bytes = (unsigned)(px * g0.row_stride * 4);
  int dev = 0, cus = 256;
  hipGetDevice(&dev);
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const int pr = ymi_internal_prof_begin(2.0 * (double)px * C * 9.0 * C, YMI_TILE_H2 | YMI_TILE_DCNP | YMI_DCNP_PATCH_C64, 14, s);
  hipLaunchKernelGGL(patch3x3_c64_k, dim3((unsigned)(p.ntiles < cus ? p.ntiles : cus)), dim3(NT), 0, s, p);
  const int rc = ymi_launch_status();
  ymi_internal_prof_end(pr, s);
  return rc;
}
