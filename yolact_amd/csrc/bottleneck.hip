// Fused ResNet bottleneck with an identity shortcut (backbone.py:37-57, stride 1, no downsample):
//   y = relu( bn3(conv3_1x1( relu(bn2(conv2_3x3( relu(bn1(conv1_1x1(x))) ))) )) + x )
// as ONE launch per block of the network.  The three separate launches move x twice (conv1 input, conv3 residual) and the two
// P-channel intermediates four times through HBM: 624 MB for a layer-0 block at batch 8 against 312 MB for x in + y out; the
// 138^2 / 69^2 blocks are HBM-bound (profiles/r03_layers_v2.txt).  Here a workgroup owns an 8 x 16 pixel tile:
//   phase 1  t1 = relu(bn1(conv1(x))) on the tile + 1 pixel halo (10 x 18 = 180 rows, padded to 192): GEMM [192 x 4P] x [4P x P],
//            x streamed global -> LDS by LDS-DMA in 32-channel chunks (3 stages), split to fp16x2 on the fly like the conv
//            engine's PREC 3 tiles; t1 leaves the accumulators as two fp16 planes in LDS (zero outside the image: conv2's padding),
//            scaled by a power of two from the TILE's own maximum (a per-tile scale is exact to undo: the whole accumulator of the
//            consuming GEMM carries it);
//   phase 2  t2 = relu(bn2(conv2(t1))): implicit GEMM [128 x 9P] x [9P x P], A fragments read straight from the t1 planes (tap
//            offsets on a swizzled 128-byte-row image), filter planes streamed by LDS-DMA (4 stages); t2 -> fp16 planes in LDS;
//   phase 3  y = relu(bn3(conv3(t2)) + x): GEMM [128 x P] x [P x 4P], filter planes prefetched during phase 2, residual read
//            from global (L2: the tile's x rows were streamed a few microseconds earlier), output + magnitude bound.
// Arithmetic: fp16x2 (three v_mfma_f32_32x32x16_f16 per product, fp32 accumulate) exactly as csrc/conv_igemm.hip PREC 3 / 4; the
// filters come as the same two fp16 planes + folded scales (engine.Packed.h2()).  512 threads, one workgroup per CU (144 KB LDS).
#include "common.h"
#include <stdlib.h>
#include "../../include/yolact_amd.h"

int ymi_internal_prof_begin(double flops, int tile, int kind, hipStream_t s);
void ymi_internal_prof_end(int idx, hipStream_t s);

namespace {

constexpr unsigned OOB = 0x80000000u;
typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr int TH = 8, TW = 16, HW2 = TW + 2, NHALO = (TH + 2) * (TW + 2), M1 = 192, NPIX = TH * TW;
constexpr int NWAVE = 8, NTHR = 64 * NWAVE;

struct BnParams {
  const float *x; float *y;
  int B, H, W, tiles_x, tiles_y;
  const void *w1, *w2, *w3;                 // fp16 planes [2][CoutPad][Kpad]
  unsigned w1_plane, w2_plane, w3_plane;    // bytes between the two planes
  unsigned w1_bytes, w2_bytes, w3_bytes;    // buffer sizes
  const float *sc1, *bi1, *sc2, *bi2, *sc3, *bi3;
  const float *x_amax; float *y_amax;
  unsigned x_bytes;
  unsigned long long *trace;     // diagnostics (env YMI_BNECK_TRACE = device address, 8 u64 per block): phase time stamps
};

struct Split2 { f16x8 h, l; };
__device__ __forceinline__ Split2 split8h(const f32x4 x0, const f32x4 x1, const float s) {
  const float x[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
  Split2 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float t = x[e] * s;
    const _Float16 h = (_Float16)t;
    o.h[e] = h;
    o.l[e] = (_Float16)(t - (float)h);
  }
  return o;
}

// C/D layout of the 32x32 MFMA: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
__device__ __forceinline__ int crow(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

#define BN_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
#define BN_STAMP(i) do { if (p.trace && threadIdx.x == 0) p.trace[(size_t)blockIdx.x * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
#define BN_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

template <int P>
__global__ __launch_bounds__(NTHR) void bottleneck_k(const BnParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int C = 4 * P;
  constexpr int ROWB = 2 * P;                         // bytes per row of a t plane (P fp16)
  constexpr int SLOTS = P / 8;                        // 16-byte slots per row
  constexpr int T1H = 0, T1L = M1 * ROWB, T1_END = 2 * M1 * ROWB;
  constexpr int T2H = T1_END, T2L = T2H + NPIX * ROWB, T2_END = T2H + 2 * NPIX * ROWB;
  constexpr int WCH = 2 * P * 64;                     // bytes of one 32-deep chunk of the w1 / w2 planes
  constexpr int W3CH = 2 * C * 64;                    // ... of the w3 planes
  constexpr int ACH = M1 * 128;                       // bytes of one 32-channel chunk of the x halo tile (fp32)
  constexpr int NS1 = 3, P1S = T1_END, P1STAGE = ACH + WCH;
  constexpr int NS2 = 3, TAPB = (P / 32) * WCH;          // phase 2 stages one filter TAP (P / 32 chunks) at a time, 3 deep
  constexpr int W2S = T2_END, W3C0 = W2S + NS2 * TAPB;
  constexpr int RED = T1H + NHALO * ROWB;                // reduction scratch: the 12 padding rows of the t1 h-plane (never read)
  constexpr int LDS_BYTES = W3C0 + W3CH;
  static_assert(NPIX * (C + 4) * 4 <= LDS_BYTES, "output tile");
  static_assert(P1S + NS1 * P1STAGE <= LDS_BYTES, "phase-1 staging");
  static_assert(W3CH <= T1_END, "the second w3 chunk reuses the t1 planes");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
  static_assert(P == 64, "tile / wave mapping written for P = 64");
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];

  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int hh = lane >> 5;
  const int logical = ymi_xcd_remap(blockIdx.x, gridDim.x);
  const int tiles = p.tiles_x * p.tiles_y;
  const int b = logical / tiles, tt = logical - b * tiles;
  const int ty = tt / p.tiles_x, tx = tt - ty * p.tiles_x;
  const int y0 = ty * TH, x0 = tx * TW;

  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.x, 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t w1rs = __builtin_amdgcn_make_buffer_rsrc((void *)p.w1, 0, (int)p.w1_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t w2rs = __builtin_amdgcn_make_buffer_rsrc((void *)p.w2, 0, (int)p.w2_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t w3rs = __builtin_amdgcn_make_buffer_rsrc((void *)p.w3, 0, (int)p.w3_bytes, 0x00020000);

  float sA, invA;
  ymi_h2_scale(ymi_amax_read(p.x_amax), sA, invA);
  const ymi_amax_pre apre = ymi_amax_prefetch(p.y_amax);
  BN_STAMP(0);
  // folded BN scale / bias of conv1 and conv2 for this lane's output columns: fetched now, not behind the filter DMAs
  float sc1r[2], bi1r[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) { sc1r[j] = p.sc1[32 * j + (lane & 31)]; bi1r[j] = p.bi1[32 * j + (lane & 31)]; }
  const float sc2r = p.sc2[32 * (wave >> 2) + (lane & 31)], bi2r = p.bi2[32 * (wave >> 2) + (lane & 31)];

  // ---- staging addresses ----------------------------------------------------------------------------------------------
  // x chunk: [192 rows][32 floats], 16-byte slots swizzled by (row >> 1) & 7; one wave instruction = 8 rows x 128 bytes
  const int kq = t & 7, r0 = t >> 3;                       // this lane's physical slot / row within a 64-row pass
  const int sl = kq ^ ((r0 >> 1) & 7);                     // logical slot living at that position
  unsigned a_voff[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int r = r0 + 64 * i;
    const int hy = r / HW2, hx = r - hy * HW2;
    const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
    const bool ok = r < NHALO && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
    a_voff[i] = ok ? (unsigned)((((b * p.H + iy) * p.W + ix) * C + 4 * sl) * 4) : OOB;
  }
  // filter planes: unit u = (plane, 16-row group); lane l fills row l >> 2, physical slot l & 3 of a 64-byte row
  auto w_off = [&](int u, int rows, unsigned plane_bytes, int kpad) -> unsigned {
    const int plane = u / (rows / 16), rg = u - plane * (rows / 16);
    const int row = rg * 16 + (lane >> 2), lsl = (lane & 3) ^ ((row >> 2) & 3);
    return (unsigned)plane * plane_bytes + (unsigned)((row * kpad + 8 * lsl) * 2);
  };
  const unsigned w1_voff = w_off(wave, P, p.w1_plane, C);           // 2 planes x P / 16 groups = 8 units: one per wave
  const unsigned w2_voff = w_off(wave, P, p.w2_plane, 9 * P);
  unsigned w3_voff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) w3_voff[i] = w_off(wave + NWAVE * i, C, p.w3_plane, P);   // 2 x C / 16 = 32 units: four per wave

  auto issue_p1 = [&](int kc, int st) {                    // x chunk kc + w1 chunk kc -> stage st
    unsigned char *As = lds + P1S + st * P1STAGE;
#pragma unroll
    for (int i = 0; i < 3; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_ptr_t)(As + (wave * 8 + 64 * i) * 128), 16, a_voff[i], kc * 128, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(w1rs, (lds_ptr_t)(As + ACH + wave * 1024), 16, w1_voff, kc * 64, 0, 0);
  };
  auto issue_w2 = [&](int tap, int st) {                   // the P / 32 chunks of one filter tap -> stage st
#pragma unroll
    for (int c = 0; c < P / 32; ++c)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w2rs, (lds_ptr_t)(lds + W2S + st * TAPB + c * WCH + wave * 1024), 16, w2_voff,
                                               (tap * (P / 32) + c) * 64, 0, 0);
  };
  auto issue_w3 = [&](int kc, int base) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w3rs, (lds_ptr_t)(lds + base + (wave + NWAVE * i) * 1024), 16, w3_voff[i], kc * 64, 0, 0);
  };
  // B fragment (filter planes image [2][rows][64 bytes]) of n-tile nt, k half s2
  auto load_b = [&](const unsigned char *Bp, int rows, int nt, int s2) -> Split2 {
    const int row = nt * 32 + (lane & 31);
    const unsigned char *q = Bp + row * 64 + (((2 * s2 + hh) ^ ((row >> 2) & 3)) * 16);
    Split2 o;
    o.h = *reinterpret_cast<const f16x8 *>(q);
    o.l = *reinterpret_cast<const f16x8 *>(q + rows * 64);
    return o;
  };
  // A fragment from a t-plane image: row `row`, logical 16-byte slot `slot`
  auto load_t = [&](int base_h, int base_l, int row, int slot) -> Split2 {
    const int off = row * ROWB + ((slot ^ (row & (SLOTS - 1))) * 16);
    Split2 o;
    o.h = *reinterpret_cast<const f16x8 *>(lds + base_h + off);
    o.l = *reinterpret_cast<const f16x8 *>(lds + base_l + off);
    return o;
  };
  auto mfma3 = [&](f32x16 &acc, const Split2 &a, const Split2 &bq) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.h, bq.l, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.l, bq.h, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.h, bq.h, acc, 0, 0, 0);
  };
  // block-wide maximum of a non-negative value (every thread calls; one barrier)
  float *red = reinterpret_cast<float *>(lds + RED);
  auto block_max = [&](float v) -> float {
    const unsigned m = ymi_wave_umax63(__float_as_uint(v));
    if (lane == 63) red[wave] = __uint_as_float(m);
    __syncthreads();
    float r = 0.f;
#pragma unroll
    for (int w = 0; w < NWAVE; ++w) r = fmaxf(r, red[w]);
    return r;
  };
  // one value -> its two fp16 pieces in a t-plane image at (row, channel n)
  auto store_t = [&](int base_h, int base_l, int row, int n, float v, float s) {
    const float tv = v * s;
    const _Float16 h = (_Float16)tv;
    const _Float16 l = (_Float16)(tv - (float)h);
    const int off = row * ROWB + (((n >> 3) ^ (row & (SLOTS - 1))) * 16) + (n & 7) * 2;
    *reinterpret_cast<_Float16 *>(lds + base_h + off) = h;
    *reinterpret_cast<_Float16 *>(lds + base_l + off) = l;
  };

  // ===== phase 1: t1 = relu(bn1(conv1(x))) on the halo tile ==========================================================
  f32x16 acc1[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc1[j][r] = 0.f;
  constexpr int NK1 = C / 32;
  issue_p1(0, 0);
  issue_p1(1, 1);
  asm volatile("" ::"v"(sc1r[0]), "v"(sc1r[1]), "v"(bi1r[0]), "v"(bi1r[1]), "v"(sc2r), "v"(bi2r));   // (keeps those loads up here)
  const int fsw = (lane >> 1) & 7;
  for (int kc = 0; kc < NK1; ++kc) {
    if (kc + 1 < NK1) BN_WAIT_VM(4); else BN_WAIT_VM(0);
    BN_BARRIER();
    if (kc + 2 < NK1) issue_p1(kc + 2, (kc + 2) % NS1);
    if (wave < M1 / 32) {                                 // waves 0..5: one 32-row m-tile x both n-tiles; 6, 7 only stage
      const unsigned char *As = lds + P1S + (kc % NS1) * P1STAGE;
      const float *Ar = reinterpret_cast<const float *>(As) + (wave * 32 + (lane & 31)) * 32;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const f32x4 v0 = *reinterpret_cast<const f32x4 *>(Ar + 4 * ((4 * s2 + 2 * hh) ^ fsw));
        const f32x4 v1 = *reinterpret_cast<const f32x4 *>(Ar + 4 * ((4 * s2 + 2 * hh + 1) ^ fsw));
        const Split2 a = split8h(v0, v1, sA);
#pragma unroll
        for (int j = 0; j < 2; ++j) mfma3(acc1[j], a, load_b(As + ACH, P, j, s2));
      }
    }
  }
  BN_STAMP(1);
  // the first w3 chunk and the first w2 chunks start now; the staging area of phase 1 is dead after this barrier
  __syncthreads();
  issue_w3(0, W3C0);
  issue_w2(0, 0);
  issue_w2(1, 1);
  {
    float v1[2][16];
    float am = 0.f;
    if (wave < M1 / 32) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float sc = sc1r[j], bi = bi1r[j];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wave * 32 + crow(r, lane);
          const int hy = row / HW2, hx = row - hy * HW2;
          const bool ok = row < NHALO && (unsigned)(y0 - 1 + hy) < (unsigned)p.H && (unsigned)(x0 - 1 + hx) < (unsigned)p.W;
          float v = (acc1[j][r] * invA) * sc + bi;
          v = v > 0.f ? v : 0.f;
          v = ok ? v : 0.f;
          v1[j][r] = v;
          am = fmaxf(am, v);
        }
      }
    }
    float s1, inv1;
    ymi_h2_scale(block_max(am), s1, inv1);
    if (wave < M1 / 32) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) store_t(T1H, T1L, wave * 32 + crow(r, lane), 32 * j + (lane & 31), v1[j][r], s1);
    }
    BN_STAMP(2);
    // ===== phase 2: t2 = relu(bn2(conv2(t1))), 128 pixels x P ==========================================================
    const int mt = wave & 3, nt = wave >> 2;
    const int pix = mt * 32 + (lane & 31), py = pix >> 4, px = pix & 15;
    f32x16 acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
    constexpr int CPT = P / 32;                            // chunks per filter tap
    for (int tap = 0; tap < 9; ++tap) {                    // one barrier per tap (2 chunks, 12 MFMAs per wave)
      if (tap + 1 < 9) BN_WAIT_VM(CPT); else BN_WAIT_VM(0);
      BN_BARRIER();                                        // (the first one also publishes the t1 planes)
      if (tap + 2 < 9) issue_w2(tap + 2, (tap + 2) % NS2);
      const int ky = tap / 3, kx = tap - ky * 3;
      const int hp = (py + ky) * HW2 + (px + kx);
#pragma unroll
      for (int cc = 0; cc < CPT; ++cc) {
        const unsigned char *Bp = lds + W2S + (tap % NS2) * TAPB + cc * WCH;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
          mfma3(acc2, load_t(T1H, T1L, hp, 4 * cc + 2 * s2 + hh), load_b(Bp, P, nt, s2));
      }
    }
    BN_STAMP(3);
    float v2[16];
    float am2 = 0.f;
    {
      const float sc = sc2r, bi = bi2r;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = (acc2[r] * inv1) * sc + bi;
        v = v > 0.f ? v : 0.f;
        v2[r] = v;
        am2 = fmaxf(am2, v);
      }
    }
    float s2_, inv2;
    ymi_h2_scale(block_max(am2), s2_, inv2);               // (its barrier: every wave is done reading t1)
#pragma unroll
    for (int r = 0; r < 16; ++r) store_t(T2H, T2L, mt * 32 + crow(r, lane), 32 * nt + (lane & 31), v2[r], s2_);
    BN_STAMP(4);
    issue_w3(1, 0);                                        // second w3 chunk -> the (dead) t1 planes
    // ===== phase 3: y = relu(bn3(conv3(t2)) + x), 128 pixels x 4P =======================================================
    f32x16 acc3[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc3[j][r] = 0.f;
    const int nb = 4 * (wave >> 2);                        // first of this wave's four n-tiles
    __syncthreads();                                       // t2 planes published (the w3 chunk 0 landed long ago)
    BN_WAIT_VM(4);                                         // chunk 0 (issued before the four pieces of chunk 1)
    BN_BARRIER();
    // output mapping: thread = (float4 column c4, row group): 16 rows each; the residual rows are requested NOW, so that their
    // latency runs under the conv3 MFMAs (scalar loads + stores per accumulator register cost 44k cycles per block)
    constexpr int ELD = C + 4;
    const int c4 = t & 63, rg = t >> 6;
    f32x4 res[16];
    const f32x4 sc4 = *reinterpret_cast<const f32x4 *>(p.sc3 + 4 * c4), bi4 = *reinterpret_cast<const f32x4 *>(p.bi3 + 4 * c4);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int q = rg + 8 * i, oy = y0 + (q >> 4), ox = x0 + (q & 15);
      const bool ok = oy < p.H && ox < p.W;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      res[i] = ok ? *reinterpret_cast<const f32x4 *>(p.x + ((size_t)(b * p.H + oy) * p.W + ox) * C + 4 * c4) : z;
    }
#pragma unroll
    for (int kc = 0; kc < P / 32; ++kc) {
      if (kc == 1) { BN_WAIT_VM(18); BN_BARRIER(); }       // chunk 1 landed (16 residual + 2 scale loads may still be in flight)
      const unsigned char *Bp = lds + (kc == 0 ? W3C0 : 0);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const Split2 a = load_t(T2H, T2L, pix, 4 * kc + 2 * s2 + hh);
#pragma unroll
        for (int j = 0; j < 4; ++j) mfma3(acc3[j], a, load_b(Bp, C, nb + j, s2));
      }
    }
    BN_STAMP(5);
    __syncthreads();                                       // every wave is done with the planes: LDS becomes the output tile
    float *es = reinterpret_cast<float *>(lds);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) es[(mt * 32 + crow(r, lane)) * ELD + 32 * (nb + j) + (lane & 31)] = acc3[j][r] * inv2;
    __syncthreads();
    float amy = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int q = rg + 8 * i, oy = y0 + (q >> 4), ox = x0 + (q & 15);
      if (oy < p.H && ox < p.W) {
        f32x4 v = *reinterpret_cast<const f32x4 *>(es + q * ELD + 4 * c4);
        v = v * sc4 + bi4;
        v = v + res[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
        *reinterpret_cast<f32x4 *>(p.y + ((size_t)(b * p.H + oy) * p.W + ox) * C + 4 * c4) = v;
        amy = fmaxf(amy, ymi_absmax4(v));
      }
    }
    if (p.y_amax) ymi_amax_finish(apre, amy);
    BN_STAMP(6);
  }
#endif
}

}  // namespace

extern "C" int ymi_bottleneck_f32(const ymi_bneck_desc *d, void *stream) {
  if (!d) return YMI_ENULL;
  if (!d->x || !d->y || !d->w1_h2 || !d->w2_h2 || !d->w3_h2 || !d->scale1 || !d->scale2 || !d->scale3 || !d->bias1 || !d->bias2 ||
      !d->bias3 || !d->x_amax)
    return YMI_ENULL;
  if (d->B <= 0 || d->H <= 0 || d->W <= 0) return YMI_EARG;
  if (d->P != 64) return YMI_ESHAPE;                       // instantiated for the 64-channel bottlenecks (ResNet layer1)
  const int C = 4 * d->P;
  if ((long)d->B * d->H * d->W * C >= (1L << 29)) return YMI_ESHAPE;
  if ((((uintptr_t)d->x) | ((uintptr_t)d->y) | ((uintptr_t)d->w1_h2) | ((uintptr_t)d->w2_h2) | ((uintptr_t)d->w3_h2)) & 15) return YMI_ESHAPE;
  if (d->cout_pad1 < d->P || d->cout_pad2 < d->P || d->cout_pad3 < C) return YMI_ESHAPE;
  BnParams p;
  p.x = d->x; p.y = d->y; p.B = d->B; p.H = d->H; p.W = d->W;
  p.tiles_x = (d->W + TW - 1) / TW; p.tiles_y = (d->H + TH - 1) / TH;
  p.w1 = d->w1_h2; p.w2 = d->w2_h2; p.w3 = d->w3_h2;
  p.w1_plane = (unsigned)d->cout_pad1 * C * 2u; p.w2_plane = (unsigned)d->cout_pad2 * 9u * d->P * 2u; p.w3_plane = (unsigned)d->cout_pad3 * d->P * 2u;
  p.w1_bytes = 2u * p.w1_plane; p.w2_bytes = 2u * p.w2_plane; p.w3_bytes = 2u * p.w3_plane;
  p.sc1 = d->scale1; p.bi1 = d->bias1; p.sc2 = d->scale2; p.bi2 = d->bias2; p.sc3 = d->scale3; p.bi3 = d->bias3;
  p.x_amax = d->x_amax; p.y_amax = d->y_amax;
  p.x_bytes = (unsigned)((long)d->B * d->H * d->W * C * 4);
  { const char *e = getenv("YMI_BNECK_TRACE"); p.trace = e ? (unsigned long long *)strtoull(e, nullptr, 0) : nullptr; }
  hipStream_t s = (hipStream_t)stream;
  const double px = (double)d->B * d->H * d->W;
  const int pr = ymi_internal_prof_begin(2.0 * px * (C * (double)d->P + 9.0 * d->P * d->P + (double)d->P * C), YMI_TILE_H2 | YMI_TILE_64x64, 7, s);
  const long grid = (long)p.tiles_x * p.tiles_y * d->B;
  hipLaunchKernelGGL(bottleneck_k<64>, dim3((unsigned)grid), dim3(NTHR), 0, s, p);
  const int rc = ymi_launch_status();
  ymi_internal_prof_end(pr, s);
  return rc;
}
