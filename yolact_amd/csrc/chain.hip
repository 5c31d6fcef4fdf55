// The tail of one ResNet bottleneck and the head of the next in ONE streaming launch (backbone.py:37-57, first stage: planes = 64):
//     y = relu(bn3(conv3 1x1 (64 -> 256)(t)) + residual)          Bottleneck.forward of block b, lines 3..5 from the end
//     z = relu(bn1(conv1 1x1 (256 -> 64)(y)))                      Bottleneck.forward of block b + 1, first line
// At 138 x 138 x 8 both layers are pure streams (5 GFLOP each against 351 / 195 MB): conv3 reads t (39 MB) and the residual
// (156 MB) and writes y (156 MB); conv1 reads y back and writes z (39 MB).  In the plan they took 0.083 + 0.055 ms, i.e. 4.2 and
// 3.5 TB/s; here y is read back from LDS, not from memory, and the launch is one pass of 390 MB.
//
// One block per CU, persistent, 8 waves.  Every wave keeps ITS filters in registers for the launch's lifetime, as MFMA operand
// fragments (96 VGPRs: 32 channels of conv3, 16 channels x K = 256 of conv1) — the first version kept both filter sets in LDS
// (141 KB) and spent its time on 52 ds_read_b128 per wave per 16 pixels (profiles/r04_chain_probe.txt, first table: 0.100 ms).
// Per iteration the block takes 32 pixels:
//   P1  t tile (32 x 64 fp32, requested one iteration ahead) -> two fp16 planes in LDS (tensor scale from x_amax)        | barrier
//   P2  GEMM 1 on v_mfma_f32_16x16x32_f16, issued as W X^T: wave w owns output channels 32 w .. 32 w + 31 of both 16-pixel
//       halves; a lane ends with four consecutive channels of one pixel
//   P3  epilogue 1 in that layout: scale / bias / + residual (float4, requested one iteration ahead) / ReLU / float4 store of y;
//       the wave's two 16 x 32 slices of y -> fp16 planes in LDS, each with the power-of-two scale of ITS OWN maximum     | barrier
//   P4  GEMM 2: wave (q, i) owns z channels 16 q .. + 15 of half i; the K = 256 sum is taken slice by slice (32 channels = one
//       producing wave), each partial sum divided by its slice's scale (exact): no block- or tensor-wide bound of y is needed
//   P5  epilogue 2: scale / bias / ReLU / float4 store of z
// ONE LDS-only barrier per 32 pixels (operand tiles double-buffered; P4 / P5 of tile k - 1 run next to P2 / P3 of tile k); the
// global loads of iteration i + 1 are in flight during iteration i; all memory
// operations are unconditional buffer operations (out-of-range offsets for masked rows), so the compiler's vmcnt is exact.
// Magnitude bounds of y and z are reported like every other launch (ymi_amax_*).  z == nullptr: GEMM 2 is skipped (conv3 alone:
// the last block of the stage, whose consumer is not a 64-channel 1x1).
#include "common.h"
#include <type_traits>
#include "../../include/yolact_amd.h"

int ymi_internal_prof_begin(double flops, int tile, int kind, hipStream_t s);
void ymi_internal_prof_end(int idx, hipStream_t s);

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int K1 = 64, N1 = 256, N2 = 64, PX = 32, NW = 8;
constexpr int RS1 = 2 * K1 + 16;          // bytes per LDS row of a K = 64 plane (128 + 16: the 16 lanes of a read phase hit 16 bank groups)
constexpr int RS2 = 2 * N1 + 16;          // ... of a K = 256 plane
constexpr int A1_PLANE = PX * RS1, A2_PLANE = PX * RS2;
// LDS: the two operand tiles and the slice scales are DOUBLE-buffered (iteration k works in buffer k & 1), which is what lets one
// barrier per iteration do (see the kernel): 2 x 9.2 KB + 2 x 33.8 KB + scales + conv3's per-lane epilogue constants
constexpr int A1_BUF = 2 * A1_PLANE, A2_BUF = 2 * A2_PLANE;
constexpr int OFF_A1 = 0, OFF_A2 = OFF_A1 + 2 * A1_BUF, OFF_SC = OFF_A2 + 2 * A2_BUF, OFF_EP = OFF_SC + 2 * (2 * NW * 4),
              CH_LDS = OFF_EP + NW * 2 * 4 * 32;   // OFF_EP: conv3's per-lane epilogue constants (scale, bias) x (wave, j, g)

struct ChainParams {
  const float *x, *res, *x_amax;
  const void *wa, *wb;                    // fp16 planes [2][cpa][64] / [2][cpb][256]
  const float *sa, *ba, *sb, *bb;         // scale_h2 / bias of the two layers
  float *y, *z, *y_amax, *z_amax;
  int M, ldx, res_ld, ldy, ldz, act_a, act_b;
  unsigned wa_plane, wb_plane;            // bytes per plane
};

__global__ __launch_bounds__(64 * NW) void chain_h2_k(const ChainParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  __shared__ __attribute__((aligned(16))) char lds[CH_LDS];
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lr = lane & 15, g = lane >> 4;
  const int q2 = wave & 3, i2 = wave >> 2;             // GEMM 2: this wave's 16 z channels (16 q2 ..) and 16-pixel half of the tile
  float sA, invA;
  ymi_h2_scale(ymi_amax_read(p.x_amax), sA, invA);
  const ymi_amax_pre apre_y = ymi_amax_prefetch(p.y_amax);
  const ymi_amax_pre apre_z = ymi_amax_prefetch(p.z_amax);
  const bool two = p.z != nullptr;

  // ---- this wave's filters -> REGISTERS, once: as MFMA A-operand fragments (row = output channel lr of a 16-channel tile, 8
  // consecutive k at 8 g of a 32-deep chunk).  GEMM 1: channels 32 wave + 16 j + lr, chunks 0 / 1; GEMM 2: channels 16 q2 + lr, chunks 0 .. 7
  f16x8 w3h[2][2], w3l[2][2], w1h[8], w1l[8];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const char *src = reinterpret_cast<const char *>(p.wa) + (size_t)(32 * wave + 16 * j + lr) * (2 * K1) + c * 64 + g * 16;
      w3h[j][c] = *reinterpret_cast<const f16x8 *>(src);
      w3l[j][c] = *reinterpret_cast<const f16x8 *>(src + p.wa_plane);
    }
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const f16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
    const char *src = reinterpret_cast<const char *>(p.wb) + (size_t)(16 * q2 + lr) * (2 * N1) + c * 64 + g * 16;
    w1h[c] = two ? *reinterpret_cast<const f16x8 *>(src) : z8;
    w1l[c] = two ? *reinterpret_cast<const f16x8 *>(src + p.wb_plane) : z8;
  }
  // per-lane epilogue constants: channels 32 wave + 16 j + 4 g .. + 3 of conv3, 16 q2 + 4 g .. + 3 of conv1
  // (conv3's live in LDS, 32 bytes per (wave, j, g), and are re-read in every epilogue: 16 registers the filter fragments need)
  if (lr == 0) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = 32 * wave + 16 * j + 4 * g;
      f32x4 sc, bi;
#pragma unroll
      for (int e = 0; e < 4; ++e) { sc[e] = p.sa[n + e] * invA; bi[e] = p.ba ? p.ba[n + e] : 0.f; }
      *reinterpret_cast<f32x4 *>(lds + OFF_EP + ((wave * 2 + j) * 4 + g) * 32) = sc;
      *reinterpret_cast<f32x4 *>(lds + OFF_EP + ((wave * 2 + j) * 4 + g) * 32 + 16) = bi;
    }
  }
  f32x4 sc1 = {1.f, 1.f, 1.f, 1.f}, bi1 = {0.f, 0.f, 0.f, 0.f};
  if (two) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { sc1[e] = p.sb[16 * q2 + 4 * g + e]; bi1[e] = p.bb ? p.bb[16 * q2 + 4 * g + e] : 0.f; }
  }
  const float slope_a = p.act_a == YMI_ACT_RELU ? 0.f : (p.act_a == YMI_ACT_LEAKY01 ? 0.1f : 1.f);
  const float slope_b = p.act_b == YMI_ACT_RELU ? 0.f : (p.act_b == YMI_ACT_LEAKY01 ? 0.1f : 1.f);

  const int ntiles = (p.M + PX - 1) / PX;
  // loads of a tile: this thread's 16 bytes of t (pixel t >> 4 of 32, channels 4 (t & 15) ..), this lane's residual vectors (pixel
  // 16 i + lr, channels 32 wave + 16 j + 4 g ..).  Buffer loads / stores with an out-of-range offset for rows past M (and tiles
  // past the last): zeros back, nothing written, and NO branch — with conditional memory operations the compiler cannot count what
  // is outstanding and falls back to vmcnt(0) between iterations, i.e. it waits for the stores of y it has just issued
  constexpr unsigned OOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.x, 0, (int)((unsigned)p.M * (unsigned)p.ldx * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc((void *)(p.res ? p.res : p.x), 0, p.res ? (int)((unsigned)p.M * (unsigned)p.res_ld * 4u) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.y, 0, (int)((unsigned)p.M * (unsigned)p.ldy * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t zrs = __builtin_amdgcn_make_buffer_rsrc((void *)(p.z ? p.z : p.y), 0, p.z ? (int)((unsigned)p.M * (unsigned)p.ldz * 4u) : 0, 0x00020000);
  const int xp = t >> 4, xc = t & 15;
  auto load_x = [&](int tile) {
    const int m = tile * PX + xp;
    const unsigned off = (tile < ntiles && m < p.M) ? (unsigned)(m * p.ldx + 4 * xc) * 4u : OOB;
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, off, 0, 0));
  };
  auto load_res = [&](int tile, f32x4 (&r)[2][2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = tile * PX + 16 * i + lr;
      const unsigned off = (tile < ntiles && m < p.M) ? (unsigned)(m * p.res_ld + 32 * wave + 4 * g) * 4u : OOB;
#pragma unroll
      for (int j = 0; j < 2; ++j) r[i][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rrs, off + 64u * j, 0, 0));
    }
  };

  // LDS-only barrier: __syncthreads() would also wait for the global loads of the NEXT iteration (vmcnt(0)), i.e. expose one memory
  // round trip per tile
#define CHAIN_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
  // One iteration = 32 pixels, ONE barrier, software-pipelined over two tiles: after the barrier of iteration k the waves run
  // GEMM 2 + epilogue 2 of tile k - 1 (operands in buffer (k - 1) & 1, published by that barrier) and GEMM 1 + epilogue 1 of tile k
  // (t tile in buffer k & 1, written before the barrier) — two independent instruction streams for the scheduler, and half the
  // barriers of the first version.  Buffer k & 1 is next written in iteration k + 2, after every wave has passed barrier k + 1,
  // i.e. finished reading it.
  // `rv`: this tile's residual vectors (requested an iteration ago), `rn`: the registers the next tile's are requested into.  The
  // loop runs two iterations per trip with the two register sets (and the two LDS buffers) swapped: a copy rn -> rv at the end of
  // an iteration would wait for every outstanding memory operation, the stores of y just issued included (seen in the ISA as
  // s_waitcnt vmcnt(0) on the back edge).  Iterations past the last tile load nothing and store nothing but the last tile's z.
  float am_y = 0.f, am_z = 0.f;
  f32x4 xv;
  const int grid = (int)gridDim.x;
  auto iteration = [&](const int tile, auto buf_c, f32x4 (&rv)[2][2], f32x4 (&rn)[2][2]) {
    constexpr int BUF = decltype(buf_c)::value;
    char *const a1 = lds + OFF_A1 + BUF * A1_BUF;
    char *const a2w = lds + OFF_A2 + BUF * A2_BUF;
    const char *const a2r = lds + OFF_A2 + (BUF ^ 1) * A2_BUF;
    float *const scw = reinterpret_cast<float *>(lds + OFF_SC) + BUF * (2 * NW);
    const float *const scr = reinterpret_cast<const float *>(lds + OFF_SC) + (BUF ^ 1) * (2 * NW);
    // P1: t tile -> planes
    {
      const f32x4 v = xv * sA;
      f16x4 h4, l4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const _Float16 h = (_Float16)v[e];
        h4[e] = h;
        l4[e] = (_Float16)(v[e] - (float)h);
      }
      char *dst = a1 + xp * RS1 + xc * 8;
      *reinterpret_cast<f16x4 *>(dst) = h4;
      *reinterpret_cast<f16x4 *>(dst + A1_PLANE) = l4;
    }
    xv = load_x(tile + grid);                           // next iteration's operands: in flight until its P1 / P3
    load_res(tile + grid, rn);
    CHAIN_BARRIER();
    if (two) {
      // P4 (tile k - 1): GEMM 2 for pixels 16 i2 + lr, z channels 16 q2 ..: the K = 256 sum one 32-channel slice (one producing
      // wave, one scale) at a time
      f32x4 tot = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c0 = 0; c0 < 8; c0 += 2) {               // two slices at a time: two independent accumulators interleaved
        f16x8 xh[2], xl[2];
        f32x4 a2[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          xh[u] = *reinterpret_cast<const f16x8 *>(a2r + (16 * i2 + lr) * RS2 + (c0 + u) * 64 + g * 16);
          xl[u] = *reinterpret_cast<const f16x8 *>(a2r + A2_PLANE + (16 * i2 + lr) * RS2 + (c0 + u) * 64 + g * 16);
          a2[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int pr = 0; pr < 3; ++pr)
#pragma unroll
          for (int u = 0; u < 2; ++u)
            a2[u] = ymi_mfma16(pr == 0 ? w1l[c0 + u] : w1h[c0 + u], pr == 1 ? xl[u] : xh[u], a2[u]);
#pragma unroll
        for (int u = 0; u < 2; ++u) tot += a2[u] * scr[i2 * NW + c0 + u];
      }
      // P5: epilogue 2
      const int tp = tile - grid, m = tp * PX + 16 * i2 + lr;
      const bool ok = tp >= 0 && m < p.M;
      f32x4 v = tot * sc1 + bi1;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], slope_b * v[e]);
      am_z = fmaxf(am_z, ok ? ymi_absmax4(v) : 0.f);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), zrs, ok ? (unsigned)(m * p.ldz + 16 * q2 + 4 * g) * 4u : OOB, 0, 0);
    }
    // P2 (tile k): GEMM 1 — both 16-pixel halves x this wave's two 16-channel tiles
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      f16x8 xh[2], xl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        xh[i] = *reinterpret_cast<const f16x8 *>(a1 + (16 * i + lr) * RS1 + c * 64 + g * 16);
        xl[i] = *reinterpret_cast<const f16x8 *>(a1 + A1_PLANE + (16 * i + lr) * RS1 + c * 64 + g * 16);
      }
#pragma unroll
      for (int pr = 0; pr < 3; ++pr)                    // (product-major: consecutive MFMAs belong to different accumulators)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = ymi_mfma16(pr == 0 ? w3l[j][c] : w3h[j][c], pr == 1 ? xl[i] : xh[i], acc[i][j]);
    }
    // P3 (tile k): epilogue 1 + this wave's 16 x 32 slices of y -> planes (one power-of-two scale per slice)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = tile * PX + 16 * i + lr;
      const bool ok = m < p.M;
      const unsigned yoff = ok ? (unsigned)(m * p.ldy + 32 * wave + 4 * g) * 4u : OOB;
      float tm = 0.f;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const f32x4 sc3 = *reinterpret_cast<const f32x4 *>(lds + OFF_EP + ((wave * 2 + j) * 4 + g) * 32);
        const f32x4 bi3 = *reinterpret_cast<const f32x4 *>(lds + OFF_EP + ((wave * 2 + j) * 4 + g) * 32 + 16);
        f32x4 v = acc[i][j] * sc3 + bi3 + rv[i][j];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], slope_a * v[e]);
        if (!ok) v = f32x4{0.f, 0.f, 0.f, 0.f};
        acc[i][j] = v;
        tm = fmaxf(tm, ymi_absmax4(v));
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yrs, yoff + 64u * j, 0, 0);
      }
      am_y = fmaxf(am_y, tm);
      if (two) {
        const unsigned tmb = ymi_wave_umax63(__builtin_bit_cast(unsigned, tm));
        float sT, invT;
        ymi_h2_scale(__builtin_bit_cast(float, (unsigned)__builtin_amdgcn_readlane((int)tmb, 63)), sT, invT);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const f32x4 v = acc[i][j] * sT;
          f16x4 h4, l4;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const _Float16 h = (_Float16)v[e];
            h4[e] = h;
            l4[e] = (_Float16)(v[e] - (float)h);
          }
          char *dst = a2w + (16 * i + lr) * RS2 + (32 * wave + 16 * j + 4 * g) * 2;
          *reinterpret_cast<f16x4 *>(dst) = h4;
          *reinterpret_cast<f16x4 *>(dst + A2_PLANE) = l4;
        }
        if (lane == 0) scw[i * NW + wave] = invT;
      }
    }
  };
  int tile = blockIdx.x;
  f32x4 ra[2][2], rb[2][2];
  xv = load_x(tile);
  load_res(tile, ra);
  for (; tile < ntiles + (two ? grid : 0); tile += 2 * grid) {      // (+ one iteration for the last tile's second layer)
    iteration(tile, std::integral_constant<int, 0>{}, ra, rb);
    iteration(tile + grid, std::integral_constant<int, 1>{}, rb, ra);
  }
#undef CHAIN_BARRIER
  if (p.y_amax) ymi_amax_finish(apre_y, am_y);
  if (two && p.z_amax) ymi_amax_finish(apre_z, am_z);
#endif
}

}  // namespace

// y = act_a(scale_a * (W_a x) + bias_a + res),  z = act_b(scale_b * (W_b y) + bias_b): two chained 1x1 convolutions, 64 -> 256 -> 64
// (include/yolact_amd.h, ymi_chain_desc).  Profiling: record kind 13 = the launch with conv A's algorithmic FLOPs, then — when z
// is computed — a kind-12 record (no duration of its own) carrying conv B's.
int ymi_internal_chain2(const ymi_chain_desc *d, hipStream_t s);    // csrc/chain2.hip: 128 / 256 planes, filters streamed through LDS

extern "C" int ymi_pointwise_chain_f32(const ymi_chain_desc *d, void *stream) {
  if (!d) return YMI_ENULL;
  if (!d->x || !d->w_a_h2 || !d->scale_a_h2 || !d->y || !d->x_amax) return YMI_ENULL;
  if (d->z && (!d->w_b_h2 || !d->scale_b_h2)) return YMI_ENULL;
  const bool wide = d->k_a == 128 || d->k_a == 256;
  if (d->M <= 0 || (!wide && (d->k_a != K1 || d->n_a != N1 || (d->z && d->n_b != N2)))) return YMI_EARG;
  if (wide && (d->n_a != 4 * d->k_a || (d->z && d->n_b != d->k_a))) return YMI_EARG;
  if (d->act_a < 0 || d->act_a > YMI_ACT_LEAKY01 || d->act_b < 0 || d->act_b > YMI_ACT_LEAKY01) return YMI_EARG;
  if (d->ldx < d->k_a || d->ldy < d->n_a || (d->z && d->ldz < d->n_b) || (d->res && d->res_ld < d->n_a)) return YMI_ESHAPE;
  if ((d->ldx | d->ldy | d->ldz | d->res_ld) & 3) return YMI_ESHAPE;
  if ((((uintptr_t)d->x) | ((uintptr_t)d->y) | ((uintptr_t)d->z) | ((uintptr_t)d->res) | ((uintptr_t)d->w_a_h2) | ((uintptr_t)d->w_b_h2)) & 15)
    return YMI_ESHAPE;
  if (d->M * (int64_t)(d->ldy > d->res_ld ? d->ldy : d->res_ld) >= (1LL << 29) || d->M * (int64_t)d->ldx >= (1LL << 29)) return YMI_ESHAPE;   // 32-bit buffer offsets
  if (wide) return ymi_internal_chain2(d, (hipStream_t)stream);
  if (d->cout_pad_a < N1 || (d->z && d->cout_pad_b < N2)) return YMI_ESHAPE;
  ChainParams p;
  p.x = d->x; p.res = d->res; p.x_amax = d->x_amax; p.wa = d->w_a_h2; p.wb = d->w_b_h2;
  p.sa = d->scale_a_h2; p.ba = d->bias_a; p.sb = d->scale_b_h2; p.bb = d->bias_b;
  p.y = d->y; p.z = d->z; p.y_amax = d->y_amax; p.z_amax = d->z_amax;
  p.M = (int)d->M; p.ldx = d->ldx; p.res_ld = d->res_ld; p.ldy = d->ldy; p.ldz = d->ldz; p.act_a = d->act_a; p.act_b = d->act_b;
  p.wa_plane = (unsigned)((long)d->cout_pad_a * K1 * 2); p.wb_plane = (unsigned)((long)d->cout_pad_b * N1 * 2);
  hipStream_t s = (hipStream_t)stream;
  const int pr = ymi_internal_prof_begin(2.0 * (double)d->M * K1 * N1, YMI_TILE_H2 | YMI_TILE_64x64, 13, s);
  int dev = 0, cus = 256;
  hipGetDevice(&dev);
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const long ntiles = (d->M + PX - 1) / PX;
  hipLaunchKernelGGL(chain_h2_k, dim3((unsigned)(ntiles < cus ? ntiles : cus)), dim3(64 * NW), 0, s, p);
  const int rc = ymi_launch_status();
  ymi_internal_prof_end(pr, s);
  if (rc == YMI_OK && d->z) {
    const int p2 = ymi_internal_prof_begin(2.0 * (double)d->M * N1 * N2, YMI_TILE_H2 | YMI_TILE_64x64, 12, s);
    ymi_internal_prof_end(p2, s);
  }
  return rc;
}
