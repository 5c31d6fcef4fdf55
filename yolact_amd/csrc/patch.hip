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
// MEASURED (tools/patch_probe.py, profiles/r05_patch_probe.txt), batch 8 / batch 16, against 0.0575 / 0.111 ms for the implicit-GEMM
// tile of the round-4 table:
//   8 waves x 4 rows, two per SIMD, no room to prefetch fragments (ds_read -> wait -> MFMA in the ISA)      0.0522
//   4 waves x 8 rows, ONE per SIMD, fragments double-buffered in registers (patch3x3_c64_k below)           0.0526 / 0.090
//     — clean ISA, yet request -> 432 MFMAs -> epilogue -> convert + publish -> barrier serialise inside the one wave a SIMD holds
//   the same with stores deferred by one tile                                                              0.0555
//   PRODUCER / CONSUMER waves (patch3x3_c64_pc_k, the default): four waves keep the filters and multiply, four load / convert /
//   publish the next patch beside them, everything inside 256 registers                                    0.0421 / 0.0695 (323 TFLOP/s)
// = 6.3 us per tile of which 3.5 are MFMA issue; the rest is tile quantisation (1296 tiles on 256 CUs = 5.06 rounds run as 6) and the
// 8.5 % of masked rows / columns of 138 = 8 x 17.25.  The tuner takes the kernel where it wins.
// Arithmetic: the fp16x2 scheme of the engine (tensor scale from x_amax, h*l + l*h + h*h on the fp16 pipe, fp32 accumulate), K order
// tap-major like engine.Packed — the filter planes of Packed.h2() are used unchanged.
#include "common.h"
#include <stdlib.h>
#include <type_traits>
#include "../../include/yolact_amd.h"

int ymi_internal_prof_begin(double flops, int tile, int kind, hipStream_t s);
void ymi_internal_prof_end(int idx, hipStream_t s);

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int C = 64, NW = 4, NT = 64 * NW;
constexpr int TH = 8, TW = 16, PH = TH + 2, PW = TW + 2, PPX = PH * PW;      // output tile, input patch (180 pixels)
// bytes per patch pixel in a plane: 128 + 32.  A ds_read_b128 is served in four groups of 16 lanes that MIX the lane's pixel (lr) and
// its k group (g): {0-3, 12-15, 20-27}, ...; with a pitch of 160 bytes the 16-byte slot index (10 lr + g) mod 16 is a permutation of
// 0 .. 15 inside every group (conflict-free); with the 144 bytes of csrc/chain.hip's K = 64 plane seven of the eight g = 1 lanes of
// a group land on the banks of a g = 0 lane (2-way: first version of this kernel)
constexpr int RS = 2 * C + 32;
constexpr int PLANE = PPX * RS, BUF = 2 * PLANE;                             // 28 800 B per plane, two planes per buffer
constexpr int NLOAD = (PPX * (C / 4) + NT - 1) / NT;                          // float4 loads per thread per patch: 12 (2880 / 256)
constexpr int NCH = 9 * (C / 32);           // K chunks of 32: (tap, channel half) = 18

struct PatchParams {
  const float *x, *scale_h2, *bias, *x_amax;
  const void *w_h2;
  float *y, *y_amax;
  int B, H, W, ldx, ldy, act, tiles_x, tiles_y, ntiles;
  unsigned w_plane, x_bytes, y_bytes;
};

struct Frag { f16x8 h, l; };

// Four waves, ONE per SIMD (up to 512 registers each): wave q owns output channels 16 q .. 16 q + 15 — 144 VGPRs of filter fragments —
// and ALL eight rows of the tile: eight independent accumulators, so consecutive MFMAs never depend on each other, and room for a
// second set of activation fragments: the LDS reads of K chunk s + 1 are issued before the 24 MFMAs of chunk s (explicit double
// buffering in registers).  The first version ran eight waves at two per SIMD inside 256 registers: no room to prefetch, every
// chunk was ds_read -> s_waitcnt -> MFMA (seen in the ISA), 0.052 ms for the 0.058 of the implicit-GEMM tile.
__global__ __launch_bounds__(NT, 1) void patch3x3_c64_k(const PatchParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  __shared__ __attribute__((aligned(16))) char lds[2 * BUF];
  const int t = threadIdx.x, lane = t & 63, q = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lr = lane & 15, g = lane >> 4;
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
  f32x4 sc, bi = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    sc[e] = p.scale_h2[16 * q + 4 * g + e] * invA;     // folded BN scale / filter-row scale, times the exact 1 / sA
    if (p.bias) bi[e] = p.bias[16 * q + 4 * g + e];
  }
  const float slope = p.act == YMI_ACT_RELU ? 0.f : (p.act == YMI_ACT_LEAKY01 ? 0.1f : 1.f);

  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.x, 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.y, 0, (int)p.y_bytes, 0x00020000);
  const int per_img = p.tiles_x * p.tiles_y;

  // patch loads: thread t takes float4 number t + NT * i of the patch (pixel (t + NT i) / 16, channels 4 ((t + NT i) % 16) ..);
  // pixels outside the image (the convolution's zero padding) and slots past the patch are out-of-bounds buffer offsets: zeros
  auto request = [&](int tile, f32x4 (&v)[NLOAD]) {
    const bool live = tile < p.ntiles;
    const int b = tile / per_img, r = tile - b * per_img;
    const int ty = r / p.tiles_x, tx = r - ty * p.tiles_x;
    const int y0 = ty * TH - 1, x0 = tx * TW - 1;
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
      const int idx = t + NT * i, px = idx >> 4, cg = idx & 15;
      const int py = px / PW, pxx = px - py * PW;
      const int yy = y0 + py, xx = x0 + pxx;
      const bool in = live && px < PPX && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
      const unsigned off = in ? (unsigned)(((b * p.H + yy) * p.W + xx) * p.ldx + 4 * cg) * 4u : OOB;
      v[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, off, 0, 0));
    }
  };
  auto publish = [&](const f32x4 (&v)[NLOAD], char *buf) {
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
      const int idx = t + NT * i, px = idx >> 4, cg = idx & 15;
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
  f32x4 ld[NLOAD];
  int tile = blockIdx.x;
  request(tile, ld);
  publish(ld, lds);
  PATCH_BARRIER();
  for (int k = 0; tile < p.ntiles; tile += grid, ++k) {
    const char *const cur = lds + (k & 1) * BUF;
    char *const nxt = lds + ((k + 1) & 1) * BUF;          // last read one iteration ago: free for the whole of this one
    request(tile + grid, ld);                            // the next patch: in flight during this tile's MFMAs
    // ---- 18 K chunks x 8 tile rows: B operand = patch pixel (rb + ky, lr + kx), channels 32 c + 8 g .. + 7 ---------------------------
    f32x4 acc[TH];
#pragma unroll
    for (int rb = 0; rb < TH; ++rb) {
      acc[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
      asm volatile("" : "+v"(acc[rb]));                   // (own register group per accumulator: see patch3x3_c64_pc_k)
    }
    const char *const lane_base = cur + lr * RS + g * 16;
    auto load_chunk = [&](int ch, Frag (&f)[TH]) {        // chunk = 2 tap + c
      const int tap = ch >> 1, c = ch & 1, ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
      for (int rb = 0; rb < TH; ++rb) {
        const char *src = lane_base + ((rb + ky) * PW + kx) * RS + c * 64;
        f[rb].h = *reinterpret_cast<const f16x8 *>(src);
        f[rb].l = *reinterpret_cast<const f16x8 *>(src + PLANE);
      }
    };
    // MFMAs through the plain builtin, accumulating IN PLACE (vDst == SrcC is the one overlap the hardware handles).  The hazard of
    // csrc/common.h — hipcc placing an MFMA's destination over a source operand that dies at the instruction — is closed differently
    // from ymi_mfma16 (whose per-instruction register constraints made the compiler shuttle every accumulator through a[0:3] here:
    // 4 000 v_accvgpr moves, seen in the ISA): the filter fragments live for the whole kernel, and the activation fragments of a
    // chunk are kept alive until its last MFMA has been issued (the empty asm below), so no destination can land on them.
    // tools/check_mfma_overlap.py verifies the built object.
    auto mma_chunk = [&](int ch, const Frag (&f)[TH]) {
#pragma unroll
      for (int pr = 0; pr < 3; ++pr)                      // product-major: eight independent accumulators between dependent MFMAs
#pragma unroll
        for (int rb = 0; rb < TH; ++rb)
          acc[rb] = (ch == NCH - 1 && pr == 2) ? ymi_mfma16(wh[ch], f[rb].h, acc[rb])      // (the chain's last link: see patch3x3_c64_pc_k)
                                               : __builtin_amdgcn_mfma_f32_16x16x32_f16(pr == 0 ? wl[ch] : wh[ch], pr == 1 ? f[rb].l : f[rb].h, acc[rb], 0, 0, 0);
#pragma unroll
      for (int rb = 0; rb < TH; ++rb) asm volatile("" ::"v"(f[rb].h), "v"(f[rb].l));
    };
    Frag fa[TH], fb[TH];
    load_chunk(0, fa);
#pragma unroll
    for (int ch = 0; ch < NCH; ch += 2) {
      load_chunk(ch + 1, fb);
      mma_chunk(ch, fa);
      if (ch + 2 < NCH) load_chunk(ch + 2, fa);
      mma_chunk(ch + 1, fb);
    }
    // ---- epilogue: lane = pixel (row rb, column lr) x channels 16 q + 4 g .. + 3 ------------------------------------------------------
    // (tried: forming the values here and storing them during the NEXT tile's MFMAs, so that the next tile's fragment loads do not
    // wait for these stores to complete — the s_waitcnt vmcnt at the top of the loop in the ISA; measured 0.0555 against 0.0526 ms)
    {
      const int b = tile / per_img, r = tile - b * per_img;
      const int ty = r / p.tiles_x, tx = r - ty * p.tiles_x;
      const int ox = tx * TW + lr;
#pragma unroll
      for (int rb = 0; rb < TH; ++rb) {
        const int oy = ty * TH + rb;
        const bool ok = oy < p.H && ox < p.W;
        f32x4 v = acc[rb] * sc + bi;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], slope * v[e]);
        am = fmaxf(am, ok ? ymi_absmax4(v) : 0.f);
        const unsigned off = ok ? (unsigned)(((b * p.H + oy) * p.W + ox) * p.ldy + 16 * q + 4 * g) * 4u : OOB;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yrs, off, 0, 0);
      }
    }
    publish(ld, nxt);                                    // the next tile's patch: its buffer was last read one iteration ago
    PATCH_BARRIER();
  }
#undef PATCH_BARRIER
  if (p.y_amax) ymi_amax_finish(apre, am);
#endif
}

// ---- producer / consumer form (8 waves, two per SIMD, inside 256 registers each) ----------------------------------------------------
// Waves 0 .. 3 are CONSUMERS: wave q keeps the filters of output channels 16 q .. + 15 in 144 registers and runs the tile's MFMAs in
// two passes of four rows (4 accumulators, fragments double-buffered: 144 + 16 + 64 registers).  Waves 4 .. 7 are PRODUCERS: they
// request the next tile's patch, split it into the fp16 planes and write it to the other LDS buffer WHILE the consumers multiply —
// the load / convert / publish phases that serialise inside the one wave per SIMD of patch3x3_c64_k run beside the MFMAs here.  One
// barrier per tile for all eight waves.
constexpr int PC_NT = 512, PC_NLOAD = (PPX * (C / 4) + 255) / 256;             // 12 float4 per producer thread per patch

__global__ __launch_bounds__(PC_NT) void patch3x3_c64_pc_k(const PatchParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  __shared__ __attribute__((aligned(16))) char lds[2 * BUF];
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lr = lane & 15, g = lane >> 4;
  constexpr unsigned OOB = 0x80000000u;
  float sA, invA;
  ymi_h2_scale(ymi_amax_read(p.x_amax), sA, invA);
  const ymi_amax_pre apre = ymi_amax_prefetch(p.y_amax);
  const int per_img = p.tiles_x * p.tiles_y;
  const int grid = (int)gridDim.x;
#define PATCH_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

  if (wave >= 4) {
    // =============================== producers: 256 threads, patch of tile k + 1 -> LDS buffer (k + 1) & 1 ===========================
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.x, 0, (int)p.x_bytes, 0x00020000);
    const int tp = t - 256;
    auto request = [&](int tile, f32x4 (&v)[PC_NLOAD]) {
      const bool live = tile < p.ntiles;
      const int b = tile / per_img, r = tile - b * per_img;
      const int ty = r / p.tiles_x, tx = r - ty * p.tiles_x;
      const int y0 = ty * TH - 1, x0 = tx * TW - 1;
#pragma unroll
      for (int i = 0; i < PC_NLOAD; ++i) {
        const int idx = tp + 256 * i, px = idx >> 4, cg = idx & 15;
        const int py = px / PW, pxx = px - py * PW;
        const int yy = y0 + py, xx = x0 + pxx;
        const bool in = live && px < PPX && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
        const unsigned off = in ? (unsigned)(((b * p.H + yy) * p.W + xx) * p.ldx + 4 * cg) * 4u : OOB;
        v[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, off, 0, 0));
      }
    };
    auto publish = [&](const f32x4 (&v)[PC_NLOAD], char *buf) {
#pragma unroll
      for (int i = 0; i < PC_NLOAD; ++i) {
        const int idx = tp + 256 * i, px = idx >> 4, cg = idx & 15;
        if (px < PPX) {
          const f32x4 sv = v[i] * sA;
          f16x4 h4, l4;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const _Float16 h = (_Float16)sv[e];
            h4[e] = h;
            l4[e] = (_Float16)(sv[e] - (float)h);
          }
          char *dst = buf + px * RS + cg * 8;
          *reinterpret_cast<f16x4 *>(dst) = h4;
          *reinterpret_cast<f16x4 *>(dst + PLANE) = l4;
        }
      }
    };
    // the patch of tile k + 2 is in flight while that of tile k + 1 is converted and written: two register sets, swapped per trip
    int tile = blockIdx.x;
    f32x4 va[PC_NLOAD], vb[PC_NLOAD];
    request(tile, va);
    request(tile + grid, vb);
    publish(va, lds);
    PATCH_BARRIER();
    for (int k = 0; tile < p.ntiles; tile += 2 * grid, k += 2) {
      request(tile + 2 * grid, va);
      publish(vb, lds + ((k + 1) & 1) * BUF);            // tile + grid
      PATCH_BARRIER();
      if (tile + grid < p.ntiles) {
        request(tile + 3 * grid, vb);
        publish(va, lds + (k & 1) * BUF);                // tile + 2 grid
        PATCH_BARRIER();
      }
    }
    return;
  }
  // ================================= consumers: wave q = output channels 16 q .. 16 q + 15 ==============================================
  const int q = wave;
  f16x8 wh[NCH], wl[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const char *src = reinterpret_cast<const char *>(p.w_h2) + (size_t)(16 * q + lr) * (2 * 9 * C) + c * 64 + g * 16;
    wh[c] = *reinterpret_cast<const f16x8 *>(src);
    wl[c] = *reinterpret_cast<const f16x8 *>(src + p.w_plane);
  }
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.y, 0, (int)p.y_bytes, 0x00020000);
  const float slope = p.act == YMI_ACT_RELU ? 0.f : (p.act == YMI_ACT_LEAKY01 ? 0.1f : 1.f);
  f32x4 sc, bi = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    sc[e] = p.scale_h2[16 * q + 4 * g + e] * invA;       // folded BN scale / filter-row scale, times the exact 1 / sA
    if (p.bias) bi[e] = p.bias[16 * q + 4 * g + e];
  }
  float am = 0.f;
  int tile = blockIdx.x;
  PATCH_BARRIER();
  for (int k = 0; tile < p.ntiles; tile += grid, ++k) {
    const char *const cur = lds + (k & 1) * BUF;
    const char *const lane_base = cur + lr * RS + g * 16;
    const int b = tile / per_img, r = tile - b * per_img;
    const int ty = r / p.tiles_x, tx = r - ty * p.tiles_x;
    const int ox = tx * TW + lr;
#pragma unroll
    for (int half = 0; half < 2; ++half) {                // rows 4 half .. 4 half + 3 of the tile
      f32x4 acc[4];
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) {
        acc[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
        // every accumulator its OWN live register group from the start: a zero shared between them is a SrcC that dies at the
        // first MFMA, and hipcc then placed that MFMA's destination half over it (v[166:169] <- ..., v[168:171]: caught by
        // tools/check_mfma_overlap.py in the first build of this kernel)
        asm volatile("" : "+v"(acc[rb]));
      }
      // Per chunk the products run (w_h, x_l) | (w_l, x_h) | (w_h, x_h) — both small terms first, as everywhere.
      auto src_of = [&](int ch, int rb) {
        const int tap = ch >> 1, c = ch & 1, ky = tap / 3, kx = tap - 3 * ky;
        return lane_base + ((4 * half + rb + ky) * PW + kx) * RS + c * 64;
      };
      // x_l single-buffered (free after the first product, re-read behind it: eight MFMAs of cover); x_h double-buffered (its last
      // use is the chunk's last product and its first the next chunk's second: one buffer left 4 MFMAs of cover — measured)
      f16x8 xl[4], xh0[4], xh1[4];
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) {
        xl[rb] = *reinterpret_cast<const f16x8 *>(src_of(0, rb) + PLANE);
        xh0[rb] = *reinterpret_cast<const f16x8 *>(src_of(0, rb));
      }
      auto chunk = [&](int ch, f16x8 (&xh)[4], f16x8 (&xhn)[4]) {
        if (ch + 1 < NCH) {
#pragma unroll
          for (int rb = 0; rb < 4; ++rb) xhn[rb] = *reinterpret_cast<const f16x8 *>(src_of(ch + 1, rb));
        }
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[ch], xl[rb], acc[rb], 0, 0, 0);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) asm volatile("" ::"v"(xl[rb]));       // (plain builtin + keep-alive: see patch3x3_c64_k)
        if (ch + 1 < NCH) {
#pragma unroll
          for (int rb = 0; rb < 4; ++rb) xl[rb] = *reinterpret_cast<const f16x8 *>(src_of(ch + 1, rb) + PLANE);
        }
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[ch], xh[rb], acc[rb], 0, 0, 0);
        // the LAST product of the chain through ymi_mfma16 (destination disjoint from every source): that is where the compiler
        // retargets the accumulators for the epilogue, and it placed one HALF over its own SrcC (v[174:177] <- ..., v[176:179]:
        // caught by the lint in this kernel's first full build).  Everywhere else the chain stays in place.
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
          acc[rb] = ch == NCH - 1 ? ymi_mfma16(wh[ch], xh[rb], acc[rb]) : __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[ch], xh[rb], acc[rb], 0, 0, 0);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) asm volatile("" ::"v"(xh[rb]));
      };
#pragma unroll
      for (int ch = 0; ch < NCH; ch += 2) {
        chunk(ch, xh0, xh1);
        chunk(ch + 1, xh1, xh0);
      }
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) {
        const int oy = ty * TH + 4 * half + rb;
        const bool ok = oy < p.H && ox < p.W;
        f32x4 v = acc[rb] * sc + bi;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], slope * v[e]);
        am = fmaxf(am, ok ? ymi_absmax4(v) : 0.f);
        const unsigned off = ok ? (unsigned)(((b * p.H + oy) * p.W + ox) * p.ldy + 16 * q + 4 * g) * 4u : OOB;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yrs, off, 0, 0);
      }
    }
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
  static const int variant = [] { const char *e = getenv("YMI_PATCH_VARIANT"); return e ? atoi(e) : 1; }();   // 1 (default) = producer / consumer, 0 = one wave per SIMD
  if (variant == 1)
    hipLaunchKernelGGL(patch3x3_c64_pc_k, dim3((unsigned)(p.ntiles < cus ? p.ntiles : cus)), dim3(PC_NT), 0, s, p);
  else
    hipLaunchKernelGGL(patch3x3_c64_k, dim3((unsigned)(p.ntiles < cus ? p.ntiles : cus)), dim3(NT), 0, s, p);
  const int rc = ymi_launch_status();
  ymi_internal_prof_end(pr, s);
  return rc;
}
