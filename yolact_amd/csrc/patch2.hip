// 3x3 / stride 1 / pad 1 convolution, Cin % 32 == 0, Cout >= 64, with the INPUT PATCH of a spatial tile held in LDS (one 32-channel
// chunk at a time) and the FILTERS streamed through an LDS ring — round 6's answer to "a direct 3x3 above the 190 - 380 TFLOP/s
// plateau" (proto_net, fpn.pred, head.upfeature: yolact.py:579-605, data/config.py:691; conv2 of the 128-plane ResNet stage).
//
// What bounds the implicit-GEMM tiles (profiles/r06_pipe_phase_trace.txt, r06_pc_trace.txt, r06_chain2_probe.txt): a CU moves
// ~27 bytes per clock from L2 into LDS / registers (~9.5 from beyond L2) whatever the kernel structure, and a 128 x 128 tile of the
// fp16x2 arithmetic needs 32 KB per 32-deep K chunk = 1 200 cycles of that path for 770 cycles of MFMAs; the A operand of a 3x3
// convolution is the SAME pixels nine times.  Here a block owns TH x TW output pixels (<= 256) x 128 output channels:
//   * the (TH + 2) x (TW + 2) x 32-channel input patch of the current channel chunk lives in LDS as the two fp16 planes of the fp16x2
//     arithmetic (tensor scale from x_amax) — loaded and split ONCE per chunk, double buffered; the nine taps are nine shifted views;
//   * per (chunk, tap) step only the filters move: 128 rows x 32 k x 2 planes = 16 KB by LDS-DMA through a three-unit ring;
//     16 KB + 41 KB / 9 per step against 1 536 cycles of MFMAs: the global -> LDS path is loaded to half of what it delivers;
//   * 8 waves: four CONSUMERS (one per SIMD: fragment reads + 48 v_mfma_f32_32x32x16_f16 per step, 2 channel tiles x NPT / 2 pixel
//     tiles each) and four PRODUCERS (filter DMAs two steps ahead; the next chunk's patch: requests at tap 0, split + ds_write at
//     tap 3), one s_barrier per step — a producer stalled on the memory pipe holds no MFMA back (csrc/pcconv.hip has the same roles);
//   * orientation W X^T: a lane ends with 4 x 4 consecutive output channels of ONE pixel: float4 stores straight from the accumulators.
// The tile shape (TH, TW) is a RUN-TIME parameter (the lane -> pixel map is computed, not wired): the host picks, per map size, the
// shape that wastes the fewest pixels and residency rounds (138 -> 23 x 11, 69 -> 8 x 24 ...).  Output: up to three dense segments
// whose boundaries are multiples of 128 channels (head0.upfeature + proto_net[0] share one launch).  K order tap-major like
// engine.Packed: the filter planes / scale_h2 of Packed.h2() are used unchanged.  Same products as every fp16x2 tile (h*l, l*h, h*h,
// fp32 accumulate); the K summation order per accumulator is (chunk, tap) instead of (tap, chunk): results agree with the other
// kernels to fp32 rounding, not bit for bit.
#include "common.h"
#include <stdlib.h>
#include <type_traits>
#include "../../include/yolact_amd.h"

int ymi_internal_prof_begin(double flops, int tile, int kind, hipStream_t s);
void ymi_internal_prof_end(int idx, hipStream_t s);

namespace {

typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0x80000000u;

constexpr int BN = 128;                    // output channels per block
constexpr int MAXP = 336;                  // patch pixels an LDS buffer holds ((TH + 2) * (TW + 2) <= MAXP)
constexpr int PPITCH = 80;                 // bytes per patch pixel in a plane: 64 (32 channels) + 16 — a tap is then a SCALAR byte offset (no XOR
                                           // swizzle to redo per tap), and consecutive pixels start 20 banks apart
constexpr int PPLANE = MAXP * PPITCH;      // bytes per patch plane
constexpr int PBUF = 2 * PPLANE;
constexpr int WUNIT = 2 * BN * 64;         // bytes per filter unit: two planes of 128 rows x 64 bytes
constexpr int NWU = 3;
constexpr int OFF_P = 0, OFF_W = 2 * PBUF, OFF_C = OFF_W + NWU * WUNIT, P2_LDS = OFF_C + 2 * BN * 4;
constexpr int NLP = (MAXP * 8 + 255) / 256;   // float4 patch loads per producer thread per chunk (8 lanes per pixel)
constexpr int NDMA = 4;                    // filter DMA pieces per producer wave per step (16 pieces of 16 rows x 64 bytes)

struct P2Seg { float *ptr; float *amax; int n0, n1, ld, act; };
struct Patch2Params {
  const float *x, *scale_h2, *bias, *x_amax;
  const void *w_h2;
  P2Seg seg[3];
  int nseg;
  int B, H, W, Cin, ldx, Cout, Kpad;
  int TH, TW, PW, PPX, TPX;                // tile, patch width, patch pixels, tile pixels
  int tiles_x, tiles_y, ntiles, tiles_n;
  unsigned w_plane, x_bytes;
  unsigned long long *trace;               // diagnostics build (env YMI_PATCH2_TRACE): 16 u64 per block from waves 0 and 4
};

template <int NPT>                         // pixel tiles of 32 per block (even): block tile = 32 NPT pixels x 128 channels
__global__ __launch_bounds__(512, 2) void patch2_k(const Patch2Params p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int NPJ = NPT / 2;             // pixel tiles per consumer wave
  __shared__ __attribute__((aligned(16))) char lds[P2_LDS];
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const bool producer = wave >= 4;
  const int lr = lane & 31, hh = lane >> 5;
#ifdef YMI_DIAGNOSTICS
  unsigned long long tr_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const bool tracing = p.trace != nullptr;
  tr_[0] = __builtin_amdgcn_s_memtime();
#endif

  const int logical = ymi_xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = logical % p.tiles_n, tile = logical / p.tiles_n;
  const int n0 = tile_n * BN;
  const int per_img = p.tiles_x * p.tiles_y;
  const int b = tile / per_img, r_ = tile - b * per_img;
  const int ty = r_ / p.tiles_x, tx = r_ - ty * p.tiles_x;
  const int oy0 = ty * p.TH, ox0 = tx * p.TW;          // first output pixel of the tile

  float sA, invA;
  ymi_h2_scale(ymi_amax_read(p.x_amax), sA, invA);
  // the block's output segment (boundaries are multiples of BN: one segment per block)
  P2Seg sg = p.seg[0];
  if (p.nseg > 1 && n0 >= p.seg[1].n0) sg = p.seg[1];
  if (p.nseg > 2 && n0 >= p.seg[2].n0) sg = p.seg[2];
  const ymi_amax_pre apre = ymi_amax_prefetch(sg.amax);

  const int nch = p.Cin >> 5, nsteps = 9 * nch;
#define P2_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
#define P2_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

  // epilogue constants of the block's 128 channels -> LDS (scale_h2 / sA, bias)
  if (t < BN) {
    float *cs = reinterpret_cast<float *>(lds + OFF_C);
    const int n = n0 + t;
    cs[t] = n < p.Cout ? p.scale_h2[n] * invA : 0.f;
    cs[BN + t] = (n < p.Cout && p.bias) ? p.bias[n] : 0.f;
  }

  float am = 0.f;
  if (producer) {
    // =========================================== PRODUCERS ========================================================================
    const int pt = t - 256, pw = wave - 4;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.x, 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.w_h2, 0, (int)(2 * p.w_plane), 0x00020000);
    // patch: float4 number pt + 256 i = (patch pixel, 4-channel group cg of the chunk's 32)
    unsigned poff[NLP];
    int pdst[NLP];
#pragma unroll
    for (int i = 0; i < NLP; ++i) {
      const int idx = pt + 256 * i, px = idx >> 3, cg = idx & 7;
      const int py = px / p.PW, pxx = px - py * p.PW;
      const int yy = oy0 - 1 + py, xx = ox0 - 1 + pxx;
      const bool in = px < p.PPX && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
      poff[i] = in ? (unsigned)((((b * p.H + yy) * p.W + xx) * p.ldx + 4 * cg) * 4) : OOB;
      pdst[i] = px < MAXP ? px * PPITCH + 16 * (cg >> 1) + (cg & 1) * 8 : -1;
    }
    f32x4 pv[NLP];
    auto patch_request = [&](int c) {                   // chunk c's patch (past the last chunk: zeros, no access)
      const bool live = c < nch;
#pragma unroll
      for (int i = 0; i < NLP; ++i)
        pv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, live ? poff[i] : OOB, live ? (unsigned)(c * 128) : 0u, 0));
    };
    auto patch_publish = [&](char *buf) {
#pragma unroll
      for (int i = 0; i < NLP; ++i) {
        if (pdst[i] >= 0) {
          const f32x4 v = pv[i] * sA;
          f16x4 h4, l4;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const _Float16 h = (_Float16)v[e];
            h4[e] = h;
            l4[e] = (_Float16)(v[e] - (float)h);
          }
          *reinterpret_cast<f16x4 *>(buf + pdst[i]) = h4;
          *reinterpret_cast<f16x4 *>(buf + PPLANE + pdst[i]) = l4;
        }
      }
    };
    // filters: piece q of 16 = (plane, 16-row group); wave pw issues pieces pw, pw + 4, pw + 8, pw + 12
    unsigned woff[NDMA];
    int wdst[NDMA];
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
      const int q = pw + 4 * i, plane = q >> 3, rg = q & 7;
      const int row = rg * 16 + (lane >> 2), lsl = (lane & 3) ^ ((row >> 2) & 3);
      const bool ok = n0 + row < ((p.Cout + 127) & ~127);                        // (rows of the padded filter planes exist up to CoutPad)
      woff[i] = ok ? (unsigned)plane * p.w_plane + (unsigned)(((n0 + row) * p.Kpad + 8 * lsl) * 2) : OOB;
      wdst[i] = plane * (BN * 64) + rg * 1024;
    }
    auto w_request = [&](int s, int unit) {             // step s = 9 chunk + tap -> k = tap * Cin + 32 chunk
      const bool live = s < nsteps;
      const int c = (int)(((unsigned)s * 7282u) >> 16), tap = s - 9 * c;       // s / 9 for s < 16384 (scalar: no integer division)
      const unsigned so = live ? (unsigned)((tap * p.Cin + 32 * c) * 2) : 0u;
#pragma unroll
      for (int i = 0; i < NDMA; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_ptr_t)(lds + OFF_W + unit * WUNIT + wdst[i]), 16, live ? woff[i] : OOB, so, 0, 0);
    };
    // prologue: patch of chunk 0, filters of steps 0 and 1
    patch_request(0);
    w_request(0, 0);
    w_request(1, 1);
    P2_WAIT_VM(2 * NDMA);
    patch_publish(lds + OFF_P);
    P2_WAIT_VM(0);
#ifdef YMI_DIAGNOSTICS
    tr_[1] = __builtin_amdgcn_s_memtime();
#endif
    P2_BARRIER();
    // chunk by chunk, the nine taps unrolled: the patch loads of tap 0 and their use at tap 3 are straight-line code, so the compiler
    // counts what is outstanding instead of falling back to vmcnt(0) (which would wait for the filter DMAs just issued)
    int wu = 2;                                         // unit of step s + 2
    for (int c = 0; c < nch; ++c) {
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        w_request(9 * c + tap + 2, wu);
        wu = wu == 2 ? 0 : wu + 1;
        if (tap == 0) patch_request(c + 1);             // behind this step's filter DMAs: in flight for two steps
        if (tap == 3) patch_publish(lds + OFF_P + ((c + 1) & 1) * PBUF);     // landed by the wait of tap 2
#ifdef YMI_DIAGNOSTICS
        const unsigned long long a_ = tracing ? __builtin_amdgcn_s_memtime() : 0ull;
#endif
        if (tap <= 1) P2_WAIT_VM(NDMA + NLP); else P2_WAIT_VM(NDMA);         // the next step's filters have landed
#ifdef YMI_DIAGNOSTICS
        const unsigned long long b_ = tracing ? __builtin_amdgcn_s_memtime() : 0ull;
#endif
        P2_BARRIER();
#ifdef YMI_DIAGNOSTICS
        if (tracing) { tr_[2] += b_ - a_; tr_[3] += __builtin_amdgcn_s_memtime() - b_; }
#endif
      }
    }
  } else {
    // =========================================== CONSUMERS ========================================================================
    // (the accumulators live in THIS branch only: live across the producers' code they cost it 128 registers and spilled)
    f32x16 acc[2][NPJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NPJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int wc = wave & 1, wp = wave >> 1;
    const int psw = (lr >> 2) & 3;
    // running LDS byte addresses (advanced by SCALAR deltas from tap to tap — recurrences the compiler cannot hoist out of the chunk
    // loop: the hoisted form, 9 taps x 4 tiles of lane addresses, spilled): xcur[j] = this lane's patch pixel of pixel tile j at the
    // current tap, k half hh; wcur = this lane's filter row in the current ring unit
    int xcur[NPJ];
#pragma unroll
    for (int j = 0; j < NPJ; ++j) {
      const int q = 32 * (wp * NPJ + j) + lr;
      const int qy = q / p.TW, qx = q - qy * p.TW;
      xcur[j] = OFF_P + (q < p.TPX ? qy * p.PW + qx : 0) * PPITCH + 16 * hh;
    }
    int wcur0 = OFF_W + (32 * (2 * wc) + lr) * 64 + 16 * ((0 + hh) ^ psw);     // k step 0 (slots 0 / 1), channel tile 2 wc; + 2048: tile 2 wc + 1
    int wcur1 = OFF_W + (32 * (2 * wc) + lr) * 64 + 16 * ((2 + hh) ^ psw);     // k step 1 (slots 2 / 3)
    const int d_col = PPITCH, d_row = (p.PW - 2) * PPITCH, d_back = -(2 * p.PW + 2) * PPITCH;
#ifdef YMI_DIAGNOSTICS
    tr_[1] = __builtin_amdgcn_s_memtime();
#endif
    P2_BARRIER();
    // Software pipeline over GROUPS = (k step s2, pair of pixel tiles): the pixel fragments of group g + 1 are requested before the
    // MFMAs of group g — also ACROSS the step barrier: the patch is static for the whole chunk (and the next chunk's was published at
    // tap 3), so only the four filter fragments of a step have to be read behind its barrier (first version: every group was
    // ds_read -> wait -> 12 MFMAs, 2 295 cycles per step for 1 536 of MFMAs: profiles/r06_patch2_trace.txt)
    constexpr int NG = (NPJ + 1) / 2;                   // pixel-tile pairs per k step
    f16x8 xh[2][2], xl[2][2];                           // [buffer][tile of the pair]
    f16x8 wh[2][2], wl[2][2];                           // [k step][channel tile]
    auto load_x = [&](auto bufc, int j0, int s2) {
      constexpr int BF = decltype(bufc)::value;
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
        if (j0 + jj < NPJ) {
          xh[BF][jj] = *reinterpret_cast<const f16x8 *>(lds + xcur[j0 + jj] + 32 * s2);
          xl[BF][jj] = *reinterpret_cast<const f16x8 *>(lds + xcur[j0 + jj] + 32 * s2 + PPLANE);
        }
    };
    auto load_w = [&]() {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const char *wa = lds + (s2 ? wcur1 : wcur0) + i * 2048;
          wh[s2][i] = *reinterpret_cast<const f16x8 *>(wa);
          wl[s2][i] = *reinterpret_cast<const f16x8 *>(wa + BN * 64);
        }
    };
    auto mfma_group = [&](auto bufc, int j0, int s2) {
      constexpr int BF = decltype(bufc)::value;
#pragma unroll
      for (int pr = 0; pr < 3; ++pr)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int jj = 0; jj < 2; ++jj)
            if (j0 + jj < NPJ)
              acc[i][j0 + jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pr == 0 ? wl[s2][i] : wh[s2][i], pr == 1 ? xl[BF][jj] : xh[BF][jj], acc[i][j0 + jj], 0, 0, 0);
    };
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    static_assert((2 * NG) % 2 == 0, "groups per step");
    __builtin_amdgcn_s_setprio(1);                      // the matrix waves go first when a SIMD's two waves compete for issue
    load_x(B0{}, 0, 0);                                 // group 0 of step 0
    int wu = 0;
    for (int c = 0; c < nch; ++c) {
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        load_w();
        // groups g = s2 * NG + pair; buffer = g & 1 (2 NG is even: a step starts and ends on the same buffer parity)
#pragma unroll
        for (int g = 0; g < 2 * NG; ++g) {
          const int s2 = g / NG, j0 = 2 * (g % NG);
          if (g + 1 < 2 * NG) {
            const int s2n = (g + 1) / NG, j0n = 2 * ((g + 1) % NG);
            if ((g + 1) & 1) load_x(B1{}, j0n, s2n); else load_x(B0{}, j0n, s2n);
          } else {
            // last group of the step: advance to the next tap / chunk / ring unit, then request ITS first group (patch data: valid)
            const int dw = wu == 2 ? -2 * WUNIT : WUNIT;
            wu = wu == 2 ? 0 : wu + 1;
            wcur0 += dw; wcur1 += dw;
            const int dx = tap == 8 ? d_back + ((c & 1) ? -PBUF : PBUF) : (tap % 3 == 2 ? d_row : d_col);
#pragma unroll
            for (int j = 0; j < NPJ; ++j) xcur[j] += dx;
            load_x(B0{}, 0, 0);
          }
          if (g & 1) mfma_group(B1{}, j0, s2); else mfma_group(B0{}, j0, s2);
        }
#ifdef YMI_DIAGNOSTICS
        const unsigned long long a_ = tracing ? __builtin_amdgcn_s_memtime() : 0ull;
#endif
        P2_BARRIER();
#ifdef YMI_DIAGNOSTICS
        if (tracing) tr_[3] += __builtin_amdgcn_s_memtime() - a_;
#endif
      }
    }
    __builtin_amdgcn_s_setprio(0);
#ifdef YMI_DIAGNOSTICS
    tr_[4] = __builtin_amdgcn_s_memtime();
#endif
    // ---- epilogue (consumers): scale / bias / activation, float4 stores of 4 consecutive channels of one pixel ---------------------
    {
      const float *cs = reinterpret_cast<const float *>(lds + OFF_C);
      const float slope = sg.act == YMI_ACT_RELU ? 0.f : (sg.act == YMI_ACT_LEAKY01 ? 0.1f : 1.f);
      const unsigned ybytes = (unsigned)((size_t)p.B * p.H * p.W * sg.ld * 4);
      const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void *)sg.ptr, 0, (int)ybytes, 0x00020000);
#pragma unroll
      for (int j = 0; j < NPJ; ++j) {
        const int q = 32 * (wp * NPJ + j) + lr;
        const int qy = q / p.TW, qx = q - qy * p.TW;
        const int oy = oy0 + qy, ox = ox0 + qx;
        const bool ok = q < p.TPX && oy < p.H && ox < p.W;
        const unsigned rowoff = ok ? (unsigned)(((b * p.H + oy) * p.W + ox) * sg.ld) : 0u;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int nl = 32 * (2 * wc + i) + 8 * g + 4 * hh;             // channel within the block
            const f32x4 sc = *reinterpret_cast<const f32x4 *>(cs + nl), bi = *reinterpret_cast<const f32x4 *>(cs + BN + nl);
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e];
            v = v * sc + bi;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], slope * v[e]);
            const bool okn = ok && n0 + nl < sg.n1;                         // (Cout % 4 == 0: the four channels exist together)
            if (!okn) v = f32x4{0.f, 0.f, 0.f, 0.f};
            am = fmaxf(am, ymi_absmax4(v));
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yrs, okn ? (rowoff + (unsigned)(n0 + nl - sg.n0)) * 4u : OOB, 0, 0);
          }
      }
    }
  }
#undef P2_WAIT_VM
#undef P2_BARRIER

  if (sg.amax) ymi_amax_finish(apre, am);
#ifdef YMI_DIAGNOSTICS
  if (tracing) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tr_[5] = __builtin_amdgcn_s_memtime();
    if (lane == 0 && (wave == 0 || wave == 4)) {
      unsigned long long *o_ = p.trace + 32 * (size_t)blockIdx.x + (wave == 0 ? 0 : 16);
#pragma unroll
      for (int i = 0; i < 8; ++i) o_[i] = tr_[i];
      o_[14] = (unsigned long long)nsteps;
      o_[15] = 1;
    }
  }
#endif
#endif
}

// tile shape for an H x W map and a block of `cap` pixels (patch <= MAXP pixels): the (TH, TW) that needs the fewest residency rounds
// of 256 CUs (x a work factor for masked pixels), ties to the smaller patch
void pick_tile(int B, int H, int W, int tiles_n, int cap, int &TH, int &TW) {
  double best = 1e30;
  TH = 8; TW = cap / 8;
  for (int th = 4; th <= 32; ++th) {
    const int tw = cap / th;
    if (tw < 4 || (th + 2) * (tw + 2) > MAXP) continue;
    for (int tw2 = tw; tw2 >= tw - 3 && tw2 >= 4; --tw2) {        // (a slightly narrower tile can fit the map better)
      const long nt = (long)B * ((H + th - 1) / th) * ((W + tw2 - 1) / tw2) * tiles_n;
      const double rounds = nt <= 256 ? 1.0 : (double)((nt + 255) / 256);
      const double cost = rounds * 1000.0 + (double)(th + 2) * (tw2 + 2) * 0.05 + (nt <= 256 ? (256 - nt) * 0.5 : 0.0);
      if (cost < best) { best = cost; TH = th; TW = tw2; }
    }
  }
}

}  // namespace

// internal (called by ymi_conv2d_nhwc_f32 for tile YMI_TILE_H2 | YMI_TILE_DCNP | YMI_DCNP_PATCH2_*): 3x3 / stride 1 / pad 1,
// Cin % 32 == 0, no residual, up to three dense output segments whose boundaries are multiples of 128 channels, activation none /
// ReLU / LeakyReLU per segment.  Profiling record kind 15.
int ymi_internal_patch2_conv(const ymi_conv_desc *d, int base_tile, hipStream_t s) {
  if (d->kh != 3 || d->kw != 3 || d->stride != 1 || d->pad != 1 || d->Cin % 32 != 0 || d->Kpad != 9 * d->Cin) return YMI_EARG;
  if (d->res_mode != YMI_RES_NONE || d->nseg < 1 || d->nseg > 3 || (d->Cout & 3) || d->Cout < 64) return YMI_EARG;
  if (!d->w_h2 || !d->scale_h2 || !d->x_amax || (((uintptr_t)d->w_h2) & 15)) return YMI_ENULL;
  if (d->Ho != d->H || d->Wo != d->W) return YMI_EARG;
  const long HW = (long)d->H * d->W;
  if ((long)d->B * HW * d->ldx >= (1L << 29)) return YMI_ESHAPE;
  Patch2Params p;
  int covered = 0;
  for (int i = 0; i < d->nseg; ++i) {
    const ymi_conv_seg &g = d->seg[i];
    if (g.n0 != covered || g.n1 <= g.n0 || (g.n0 % BN) || g.act < 0 || g.act > YMI_ACT_LEAKY01 || (g.row_stride & 3) || (((uintptr_t)g.ptr) & 15) ||
        g.batch_stride != HW * g.row_stride || g.row_stride < g.n1 - g.n0 || (long)d->B * HW * g.row_stride >= (1L << 29))
      return YMI_EARG;
    covered = g.n1;
    p.seg[i].ptr = g.ptr; p.seg[i].n0 = g.n0; p.seg[i].n1 = g.n1; p.seg[i].ld = g.row_stride; p.seg[i].act = g.act;
    p.seg[i].amax = d->y_amax ? d->y_amax + (size_t)i * (YMI_AMAX_SUB * YMI_AMAX_STRIDE) : nullptr;   // (consecutive slots: ABI 5)
  }
  if (covered < d->Cout) return YMI_EARG;
  for (int i = d->nseg; i < 3; ++i) p.seg[i] = p.seg[0];
  p.nseg = d->nseg;
  p.x = d->x; p.scale_h2 = d->scale_h2; p.bias = d->bias; p.x_amax = d->x_amax; p.w_h2 = d->w_h2;
  p.B = d->B; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.ldx = d->ldx; p.Cout = d->Cout; p.Kpad = d->Kpad;
  p.tiles_n = (d->Cout + BN - 1) / BN;
  const int npt = base_tile == YMI_DCNP_PATCH2_192 ? 6 : 8;
  pick_tile(d->B, d->H, d->W, p.tiles_n, 32 * npt, p.TH, p.TW);
  { const char *e = getenv("YMI_PATCH2_TILE"); int th, tw; if (e && sscanf(e, "%dx%d", &th, &tw) == 2 && th * tw <= 32 * npt && (th + 2) * (tw + 2) <= MAXP) { p.TH = th; p.TW = tw; } }
  p.PW = p.TW + 2; p.PPX = (p.TH + 2) * p.PW; p.TPX = p.TH * p.TW;
  p.tiles_x = (d->W + p.TW - 1) / p.TW; p.tiles_y = (d->H + p.TH - 1) / p.TH; p.ntiles = d->B * p.tiles_x * p.tiles_y;
  p.w_plane = (unsigned)((((long)d->Cout + 127) / 128 * 128) * d->Kpad * 2L);
  p.x_bytes = (unsigned)((size_t)d->B * HW * d->ldx * sizeof(float));
  p.trace = nullptr;
#ifdef YMI_DIAGNOSTICS
  { const char *e = getenv("YMI_PATCH2_TRACE"); p.trace = e ? (unsigned long long *)strtoull(e, nullptr, 0) : nullptr; }
#endif
  const double flops = 2.0 * (double)d->B * HW * (double)(d->cout_alg > 0 ? d->cout_alg : d->Cout) * 9.0 * (double)(d->cin_alg > 0 ? d->cin_alg : d->Cin);
  const int pr = ymi_internal_prof_begin(flops, base_tile | YMI_TILE_H2 | YMI_TILE_DCNP, 15, s);
  const unsigned grid = (unsigned)(p.ntiles * p.tiles_n);
  if (npt == 6) hipLaunchKernelGGL(patch2_k<6>, dim3(grid), dim3(512), 0, s, p);
  else hipLaunchKernelGGL(patch2_k<8>, dim3(grid), dim3(512), 0, s, p);
  const int rc = ymi_launch_status();
  ymi_internal_prof_end(pr, s);
  return rc;
}
