// The tail of one ResNet bottleneck and the head of the next in ONE launch for the 128- and 256-plane stages (backbone.py:37-57):
//     y = relu(bn3(conv3 1x1 (P -> 4P)(t)) + residual)          Bottleneck.forward of block b, lines 3..5 from the end
//     z = relu(bn1(conv1 1x1 (4P -> P)(y)))                      Bottleneck.forward of block b + 1, first line
// csrc/chain.hip does this for P = 64 with both filter sets in registers; at P = 128 / 256 the filters are 0.5 / 2 MB and are
// streamed through LDS instead.
//
// Why (round 6, profiles/r06_pipe_phase_trace.txt, r06_pc_trace.txt): the 35 x 35 / 69 x 69 stages are bound by what a CU can pull
// from BEYOND its L2 — ~9.5 bytes per clock per CU, whatever the kernel structure (the producer / consumer split of csrc/pcconv.hip
// measured the same 1 725 cycles per 16 KB A chunk as the all-in-one waves of csrc/dcn.hip) — and conv1's whole A operand is the
// 4P-channel tensor y that conv3 wrote one launch earlier: 40 MB (P = 256) / 78 MB (P = 128) at batch 8, read once or twice.  Here a
// block owns 64 pixels, computes y for them 128 channels at a time, writes it ONCE and feeds each 128-channel slice straight into the
// K-sum of z from LDS: y is never read back.  The filters come from L2 (every block streams all of them: 2 MB per 64 pixels at
// P = 256 — an L2-hit stream, the fast path), the activations from memory.
//
// One block = 4 waves (one per SIMD, up to 512 registers each), 64 pixels.  Orientation W X^T (as csrc/chain.hip): the filter rows are
// the MFMA row operand, the pixels the column operand, so a lane ends with 4 x 4 consecutive channels of ONE pixel — residual
// loads, y / z stores and the fp16-plane writes of the y slice are 16-byte operations straight from the accumulators, no LDS
// transposition.  Wave (wm, wn): pixels 32 wm .. + 31, channel tiles 2 wn, 2 wn + 1 of whatever 128 channels are being produced.
//   * t (64 x P fp32) is read once, split into the two fp16 planes of the fp16x2 arithmetic (tensor scale from x_amax) and KEPT IN
//     REGISTERS as MFMA column-operand fragments for the whole block (128 VGPRs at P = 256);
//   * filters arrive by LDS-DMA in chunks of 128 rows x 64 k x 2 planes = 32 KB through a three-unit ring, two chunk steps ahead;
//     per slice: P / 64 chunks of conv3's filters, then (P / 128) x 2 chunks of conv1's (the columns of W1 that belong to the slice);
//   * the y slice goes to LDS as fp16 planes scaled by a RIGOROUS bound of the tensor, known when the kernel starts:
//         |y| <= amax(t) * max_n(|scale_n| sum_k |w3[n, k]|) + max_n |bias_n| + amax(residual)        (ReLU cannot raise it)
//     — csrc/chain.hip gives every 16 x 32 slice the scale of its own maximum; here slices of one pixel tile accumulate into ONE
//     set of z accumulators, so they must share a scale, and a bound that is 2^4 .. 2^7 loose only moves the low plane that many
//     binades towards the fp16 subnormals (absolute error <= 2^-25 of the scaled bound per element, far below the fp32
//     accumulation's own rounding): DESIGN 3.15;
//   * one counted s_waitcnt vmcnt + s_barrier per chunk step (24 MFMAs per wave), residual loads of a slice issued at its first step.
// Magnitude bounds of y and z are reported like every other launch (ymi_amax_*).
#include "common.h"
#include <stdlib.h>
#include <type_traits>
#include <utility>
#include "../../include/yolact_amd.h"

int ymi_internal_prof_begin(double flops, int tile, int kind, hipStream_t s);
void ymi_internal_prof_end(int idx, hipStream_t s);

namespace {

typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0x80000000u;

struct Chain2Params {
  const float *x, *res, *x_amax, *res_amax;
  const void *wa, *wb;                    // fp16 planes [2][cpa][P] / [2][cpb][4P]
  const float *sa, *ba, *sb, *bb;         // scale_h2 / bias of the two layers
  float *y, *z, *y_amax, *z_amax;
  int M, ldx, res_ld, ldy, ldz, act_a, act_b;
  unsigned wa_plane, wb_plane;            // bytes per plane
  float gain_a, bias_max_a;               // max_n(|bn scale_n| sum_k |w_a[n, k]|), max_n |bias_a[n]|: the bound of y (see the header)
  unsigned long long *trace;              // diagnostics build (env YMI_CHAIN2_TRACE): 16 u64 per block from wave 0
};

constexpr int BM = 64, SN = 128, UNIT = 32 * 1024, NUNIT = 3;
constexpr int OFF_W = 0, OFF_A2 = NUNIT * UNIT, A2_BYTES = BM * SN * 4, OFF_EP = OFF_A2 + A2_BYTES;

template <int P> constexpr int chain2_lds() { return OFF_EP + (4 * P * 2 + P * 2) * 4; }

template <int P>
__global__ __launch_bounds__(256, 1) void chain2_k(const Chain2Params p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int K1 = P, N1 = 4 * P, N2 = P;
  constexpr int NS = N1 / SN;             // slices of y
  constexpr int NG1 = K1 / 64;            // chunk steps of GEMM 1 per slice (64 k each)
  constexpr int NH = N2 / 128;            // 128-channel halves of z
  constexpr int NG2 = NH * 2;             // chunk steps of GEMM 2 per slice: (half, 64-k chunk of the slice)
  constexpr int CPS = NG1 + NG2;          // chunk steps per slice
  constexpr int KS1 = K1 / 16;            // k steps of GEMM 1 = register fragments of t per plane
  constexpr int NDMA = 8;                 // LDS-DMA pieces per wave per chunk (32 pieces of 16 rows x 64 bytes)
  __shared__ __attribute__((aligned(16))) char lds[chain2_lds<P>()];
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave & 1, wn = wave >> 1;
  const int lr = lane & 31, hh = lane >> 5;
  const int m0 = blockIdx.x * BM;

  float sX, invX, sY, invY;
  const float ax = ymi_amax_read(p.x_amax);
  ymi_h2_scale(ax, sX, invX);
  {
    const float ar = p.res_amax ? ymi_amax_read(p.res_amax) : 0.f;
    ymi_h2_scale(ax * p.gain_a + p.bias_max_a + ar, sY, invY);
  }
  const ymi_amax_pre apre_y = ymi_amax_prefetch(p.y_amax);
  const ymi_amax_pre apre_z = ymi_amax_prefetch(p.z_amax);

  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.x, 0, (int)((unsigned)p.M * (unsigned)p.ldx * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc((void *)(p.res ? p.res : p.x), 0, p.res ? (int)((unsigned)p.M * (unsigned)p.res_ld * 4u) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.y, 0, (int)((unsigned)p.M * (unsigned)p.ldy * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t zrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.z, 0, (int)((unsigned)p.M * (unsigned)p.ldz * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t wars = __builtin_amdgcn_make_buffer_rsrc((void *)p.wa, 0, (int)(2 * p.wa_plane), 0x00020000);
  const __amdgpu_buffer_rsrc_t wbrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.wb, 0, (int)(2 * p.wb_plane), 0x00020000);

  // ---- filter chunks: 128 rows x 64 k x 2 planes.  LDS image of a unit: [sub = k / 32][plane][128 rows][64 bytes], 16-byte slot s
  // of row r at s ^ ((r >> 2) & 3) (the plane image of csrc/dcn.hip).  Piece q of 32 = (sub, plane, 16-row group); wave w issues
  // pieces w, w + 4, .. (8 per chunk)
  unsigned dma_a[NDMA], dma_b[NDMA];      // byte offset of this lane's 16 bytes relative to the chunk origin, for K = K1 / K = N1 rows
  int dma_l[NDMA];
#pragma unroll
  for (int i = 0; i < NDMA; ++i) {
    const int q = wave + 4 * i, sub = q >> 4, plane = (q >> 3) & 1, rg = q & 7;
    const int row = rg * 16 + (lane >> 2), lsl = (lane & 3) ^ ((row >> 2) & 3);
    dma_a[i] = (unsigned)plane * p.wa_plane + (unsigned)((row * K1 + 32 * sub + 8 * lsl) * 2);
    dma_b[i] = (unsigned)plane * p.wb_plane + (unsigned)((row * N1 + 32 * sub + 8 * lsl) * 2);
    dma_l[i] = (sub * 2 + plane) * (128 * 64) + rg * 1024;
  }
  // chunk cc of the block's stream (cc = slice * CPS + step) -> LDS unit `unit`; past the end: out-of-bounds requests (zeros, no access)
  auto issue_chunk = [&](int sl, int step, int unit) {
    char *dst = lds + OFF_W + unit * UNIT;
    if (step < NG1) {                     // conv3's filters: rows 128 sl .., k 64 step ..
      const unsigned so = sl < NS ? (unsigned)((sl * SN * K1 + 64 * step) * 2) : 0u;
#pragma unroll
      for (int i = 0; i < NDMA; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wars, (lds_ptr_t)(dst + dma_l[i]), 16, sl < NS ? dma_a[i] : OOB, so, 0, 0);
    } else {                              // conv1's filters: rows 128 half .., k = the slice's channels 128 sl + 64 j ..
      const int j2 = step - NG1, half = j2 >> 1, j = j2 & 1;
      const unsigned so = sl < NS ? (unsigned)((half * 128 * N1 + sl * SN + 64 * j) * 2) : 0u;
#pragma unroll
      for (int i = 0; i < NDMA; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wbrs, (lds_ptr_t)(dst + dma_l[i]), 16, (sl < NS && p.z) ? dma_b[i] : OOB, so, 0, 0);
    }
  };
#define C2_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
#define C2_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#ifdef YMI_DIAGNOSTICS
  unsigned long long tr_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const bool tracing = p.trace != nullptr;
  tr_[0] = __builtin_amdgcn_s_memtime();
  // a step's tail: [a] counted wait for the next chunk [b] barrier [c]; tr_[2] += b - a (memory), tr_[3] += c - b (other waves)
#define C2_STEP_END(N) do { if (tracing) { const unsigned long long a_ = __builtin_amdgcn_s_memtime(); C2_WAIT_VM(N); \
    const unsigned long long b_ = __builtin_amdgcn_s_memtime(); C2_BARRIER(); tr_[2] += b_ - a_; tr_[3] += __builtin_amdgcn_s_memtime() - b_; } \
    else { C2_WAIT_VM(N); C2_BARRIER(); } } while (0)
#else
#define C2_STEP_END(N) do { C2_WAIT_VM(N); C2_BARRIER(); } while (0)
#endif

  issue_chunk(0, 0, 0);
  issue_chunk(0, 1, 1);

  // ---- epilogue constants -> LDS: conv3's (scale_h2 / sX, bias) per channel of y, conv1's (scale_h2 / sY, bias) per channel of z
  float *ep = reinterpret_cast<float *>(lds + OFF_EP);
  for (int n = t; n < N1; n += 256) { ep[n] = p.sa[n] * invX; ep[N1 + n] = p.ba ? p.ba[n] : 0.f; }
  if (p.z)
    for (int n = t; n < N2; n += 256) { ep[2 * N1 + n] = p.sb[n] * invY; ep[2 * N1 + N2 + n] = p.bb ? p.bb[n] : 0.f; }

  // ---- t tile -> registers, as column-operand fragments: pixel 32 wm + lr, k = 16 s + 8 hh .. + 7 of step s
  f16x8 xh[KS1], xl[KS1];
  {
    const int m = m0 + 32 * wm + lr;
    const unsigned base = m < p.M ? (unsigned)(m * p.ldx + 8 * hh) * 4u : OOB;
#pragma unroll
    for (int s0 = 0; s0 < KS1; s0 += 4) {              // four steps at a time: 8 loads in flight, 32 raw registers
      f32x4 raw[4][2];
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int q = 0; q < 2; ++q)
          raw[s][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, base, (unsigned)((16 * (s0 + s) + 4 * q) * 4), 0));
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float v = raw[s][e >> 2][e & 3] * sX;
          const _Float16 h = (_Float16)v;
          xh[s0 + s][e] = h;
          xl[s0 + s][e] = (_Float16)(v - (float)h);
        }
    }
  }
  C2_WAIT_VM(0);
  C2_BARRIER();                           // chunks 0 and 1 landed, constants written
#ifdef YMI_DIAGNOSTICS
  tr_[1] = __builtin_amdgcn_s_memtime();
#endif

  // fragment addresses inside a unit / inside the y-slice planes (bytes)
  const int psw = (lr >> 2) & 3;
  auto w_frag = [&](const char *unit, int ks, int tile, int plane) {          // filter rows 32 tile + lr, k step ks (0 .. 3) of the chunk
    return *reinterpret_cast<const f16x8 *>(unit + ((ks >> 1) * 2 + plane) * (128 * 64) + (32 * tile + lr) * 64 + 16 * ((2 * (ks & 1) + hh) ^ psw));
  };
  // y-slice planes: [sub = k / 32 (0 .. 3)][plane][64 pixels][64 bytes]
  auto a2_addr = [&](int sub, int plane, int row, int slot) { return OFF_A2 + (sub * 2 + plane) * (BM * 64) + row * 64 + 16 * (slot ^ ((row >> 2) & 3)); };

  f32x16 acc2[NH][2];
#pragma unroll
  for (int h = 0; h < NH; ++h)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[h][j][r] = 0.f;
  const float slope_a = p.act_a == YMI_ACT_RELU ? 0.f : (p.act_a == YMI_ACT_LEAKY01 ? 0.1f : 1.f);
  const float slope_b = p.act_b == YMI_ACT_RELU ? 0.f : (p.act_b == YMI_ACT_LEAKY01 ? 0.1f : 1.f);
  const int mrow = m0 + 32 * wm + lr;     // this lane's pixel
  const bool row_ok = mrow < p.M;
  float am_y = 0.f, am_z = 0.f;
  int ru = 0;                             // ring unit of the current chunk step

  for (int sl = 0; sl < NS; ++sl) {
    f32x16 acc1[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc1[j][r] = 0.f;
    f32x4 rv[2][4];                       // residual: channels 128 sl + 32 (2 wn + j) + 8 g + 4 hh .. + 3 of this lane's pixel
    // ---- GEMM 1: NG1 chunk steps ------------------------------------------------------------------------------------------------
#pragma unroll
    for (int st = 0; st < NG1; ++st) {
      const char *unit = lds + OFF_W + ru * UNIT;
      if (st == 0) {                      // this slice's residual vectors: landed long before the epilogue (the step waits are in order)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const unsigned off = (row_ok && p.res) ? (unsigned)(mrow * p.res_ld + sl * SN + 32 * (2 * wn + j) + 8 * g + 4 * hh) * 4u : OOB;
            rv[j][g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rrs, off, 0, 0));
          }
      }
      {                                   // request the chunk two steps ahead into the unit freed by the last barrier
        const int nst = st + 2, nsl = sl + (nst >= CPS ? 1 : 0), nstep = nst >= CPS ? nst - CPS : nst;
        issue_chunk(nsl, nstep, ru == 0 ? 2 : ru - 1);
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        f16x8 wh[2], wl[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) { wh[j] = w_frag(unit, ks, 2 * wn + j, 0); wl[j] = w_frag(unit, ks, 2 * wn + j, 1); }
        const int s = 4 * st + ks;
#pragma unroll
        for (int pr = 0; pr < 3; ++pr)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pr == 0 ? wl[j] : wh[j], pr == 1 ? xl[s] : xh[s], acc1[j], 0, 0, 0);
      }
      ru = ru == 2 ? 0 : ru + 1;
      if (st == 0) C2_STEP_END(NDMA + 8); else C2_STEP_END(NDMA);     // the next step's chunk has landed (behind it: this step's requests)
    }
    // ---- epilogue 1: y slice -> global (float4 per (tile, g)), -> fp16 planes in LDS ---------------------------------------------
#ifdef YMI_DIAGNOSTICS
    const unsigned long long e_t0 = tracing ? __builtin_amdgcn_s_memtime() : 0ull;
#endif
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int ct = 2 * wn + j;          // channel tile of the slice
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = sl * SN + 32 * ct + 8 * g + 4 * hh;
        const f32x4 sc = *reinterpret_cast<const f32x4 *>(ep + n), bi = *reinterpret_cast<const f32x4 *>(ep + N1 + n);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc1[j][4 * g + e];
        v = v * sc + bi + rv[j][g];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], slope_a * v[e]);
        if (!row_ok) v = f32x4{0.f, 0.f, 0.f, 0.f};
        am_y = fmaxf(am_y, ymi_absmax4(v));
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yrs, row_ok ? (unsigned)(mrow * p.ldy + n) * 4u : OOB, 0, 0);
        const f32x4 ts = v * sY;
        f16x4 h4, l4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const _Float16 h = (_Float16)ts[e];
          h4[e] = h;
          l4[e] = (_Float16)(ts[e] - (float)h);
        }
        // k of the slice = 32 ct + 8 g + 4 hh .. + 3 -> sub ct, 16-byte slot g, byte 8 hh
        *reinterpret_cast<f16x4 *>(lds + a2_addr(ct, 0, 32 * wm + lr, g) + 8 * hh) = h4;
        *reinterpret_cast<f16x4 *>(lds + a2_addr(ct, 1, 32 * wm + lr, g) + 8 * hh) = l4;
      }
    }
    C2_BARRIER();                         // the y slice is in LDS
#ifdef YMI_DIAGNOSTICS
    if (tracing) tr_[4] += __builtin_amdgcn_s_memtime() - e_t0;
#endif
    // ---- GEMM 2: (half, 64-k chunk of the slice) -----------------------------------------------------------------------------------
#pragma unroll
    for (int st = 0; st < NG2; ++st) {
      const char *unit = lds + OFF_W + ru * UNIT;
      const int half = st >> 1, jk = st & 1;
      {
        const int nst = NG1 + st + 2, nsl = sl + (nst >= CPS ? 1 : 0), nstep = nst >= CPS ? nst - CPS : nst;
        issue_chunk(nsl, nstep, ru == 0 ? 2 : ru - 1);
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        f16x8 wh[2], wl[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) { wh[j] = w_frag(unit, ks, 2 * wn + j, 0); wl[j] = w_frag(unit, ks, 2 * wn + j, 1); }
        const int sub = 2 * jk + (ks >> 1), slot = 2 * (ks & 1) + hh;
        const f16x8 yh = *reinterpret_cast<const f16x8 *>(lds + a2_addr(sub, 0, 32 * wm + lr, slot));
        const f16x8 yl = *reinterpret_cast<const f16x8 *>(lds + a2_addr(sub, 1, 32 * wm + lr, slot));
#pragma unroll
        for (int pr = 0; pr < 3; ++pr)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc2[half][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pr == 0 ? wl[j] : wh[j], pr == 1 ? yl : yh, acc2[half][j], 0, 0, 0);
      }
      ru = ru == 2 ? 0 : ru + 1;
      if (st == 0) C2_STEP_END(NDMA + 8); else C2_STEP_END(NDMA);     // (st == 0: the eight y stores of epilogue 1 sit behind the chunk waited for)
    }
  }
#ifdef YMI_DIAGNOSTICS
  tr_[5] = __builtin_amdgcn_s_memtime();
#endif
#undef C2_STEP_END
#undef C2_WAIT_VM
#undef C2_BARRIER
  // ---- epilogue 2: z -----------------------------------------------------------------------------------------------------------------
  if (p.z) {
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = 128 * h + 32 * (2 * wn + j) + 8 * g + 4 * hh;
          const f32x4 sc = *reinterpret_cast<const f32x4 *>(ep + 2 * N1 + n), bi = *reinterpret_cast<const f32x4 *>(ep + 2 * N1 + N2 + n);
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc2[h][j][4 * g + e];
          v = v * sc + bi;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], slope_b * v[e]);
          if (!row_ok) v = f32x4{0.f, 0.f, 0.f, 0.f};
          am_z = fmaxf(am_z, ymi_absmax4(v));
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), zrs, row_ok ? (unsigned)(mrow * p.ldz + n) * 4u : OOB, 0, 0);
        }
  }
  if (p.y_amax) ymi_amax_finish(apre_y, am_y);
  if (p.z && p.z_amax) ymi_amax_finish(apre_z, am_z);
#ifdef YMI_DIAGNOSTICS
  if (tracing) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tr_[6] = __builtin_amdgcn_s_memtime();
    if (t == 0) {
      unsigned long long *o_ = p.trace + 16 * (size_t)blockIdx.x;
#pragma unroll
      for (int i = 0; i < 8; ++i) o_[i] = tr_[i];
      o_[15] = 1;
    }
  }
#endif
#endif
}

}  // namespace

// internal (called by ymi_pointwise_chain_f32 of csrc/chain.hip for k_a = 128 / 256): validated shapes only.  Profiling: kind 13 with
// conv A's FLOPs, then (z computed) a kind-12 record with conv B's.
int ymi_internal_chain2(const ymi_chain_desc *d, hipStream_t s) {
  const int P = d->k_a;
  if ((P != 128 && P != 256) || d->n_a != 4 * P || (d->z && d->n_b != P)) return YMI_EARG;
  if (d->cout_pad_a < 4 * P || (d->z && d->cout_pad_b < P)) return YMI_ESHAPE;
  if (!(d->gain_a > 0.f) || !(d->bias_max_a >= 0.f)) return YMI_EARG;     // the bound of y is part of the contract (and NaN-proof)
  if (d->res && !d->res_amax) return YMI_ENULL;
  Chain2Params p;
  p.x = d->x; p.res = d->res; p.x_amax = d->x_amax; p.res_amax = d->res ? d->res_amax : nullptr; p.wa = d->w_a_h2; p.wb = d->w_b_h2;
  p.sa = d->scale_a_h2; p.ba = d->bias_a; p.sb = d->scale_b_h2; p.bb = d->bias_b;
  p.y = d->y; p.z = d->z; p.y_amax = d->y_amax; p.z_amax = d->z_amax;
  p.M = (int)d->M; p.ldx = d->ldx; p.res_ld = d->res_ld; p.ldy = d->ldy; p.ldz = d->ldz; p.act_a = d->act_a; p.act_b = d->act_b;
  p.wa_plane = (unsigned)((long)d->cout_pad_a * P * 2); p.wb_plane = (unsigned)((long)d->cout_pad_b * 4 * P * 2);
  p.gain_a = d->gain_a; p.bias_max_a = d->bias_max_a;
  p.trace = nullptr;
#ifdef YMI_DIAGNOSTICS
  { const char *e = getenv("YMI_CHAIN2_TRACE"); p.trace = e ? (unsigned long long *)strtoull(e, nullptr, 0) : nullptr; }
#endif
  const int pr = ymi_internal_prof_begin(2.0 * (double)d->M * P * 4 * P, YMI_TILE_H2 | YMI_TILE_64x64, 13, s);
  const unsigned grid = (unsigned)((d->M + BM - 1) / BM);
  if (P == 128) hipLaunchKernelGGL(chain2_k<128>, dim3(grid), dim3(256), 0, s, p);
  else hipLaunchKernelGGL(chain2_k<256>, dim3(grid), dim3(256), 0, s, p);
  const int rc = ymi_launch_status();
  ymi_internal_prof_end(pr, s);
  if (rc == YMI_OK && d->z) {
    const int p2 = ymi_internal_prof_begin(2.0 * (double)d->M * 4 * P * P, YMI_TILE_H2 | YMI_TILE_64x64, 12, s);
    ymi_internal_prof_end(p2, s);
  }
  return rc;
}
