// NHWC implicit-GEMM convolution on the fp32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32).
//
// GEMM view:  M = B*Ho*Wo output pixels, N = Cout, K = kh*kw*Cin with k = (ky*kw+kx)*Cin + c.
//   A[m,k]  gathered on the fly from the NHWC input (zero for padding / m >= M / k >= K)
//   B[k,n]  packed weights, stored [n][k] (k contiguous) so A and B tiles have the same LDS image
// Block = 256 threads = 4 wave64; block tile BM x BN, K step 32.  Each wave owns TM x TN MFMA tiles of 32x32.
//
// Staging: LDS-DMA.  Both operand tiles go global -> LDS directly with `buffer_load_dwordx4 ... lds`
// (no VGPR round trip, no ds_write, no select): one wave instruction fills 8 rows x 128 bytes.  Rows that fall in the
// convolution's zero padding (or past M) are fetched with an out-of-range buffer offset, for which the hardware's
// buffer bounds check returns zeros — the padding costs no branch and no extra instruction.  The round-1 register
// staged version lost 14 % (global loads) + 11 % (ds_write) of the K loop to staging (profiles/r01_probe_v3.txt).
//
// LDS image: [row][32 floats] = 128-byte rows, no padding (an LDS-DMA destination is lane-linear), 16-byte slots
// XOR-swizzled: physical slot = logical slot ^ ((row >> 1) & 7).  The swizzle is applied on the SOURCE side (each lane
// fetches the k-slot that belongs at its physical position — still the same 128-byte global segment per row) and on
// the ds_read_b128 fragment addresses; every 16-lane group of a fragment read then covers all 64 banks (rows are
// distinct mod 16 within a group), i.e. conflict-free like the padded layout it replaces.
//
// Fragment trick: the MFMA consumes k in pairs {lanes 0-31: k0, lanes 32-63: k1}.  The K order of a dot product is
// free as long as A and B agree, so lane-half h loads k = 8g+4h .. 8g+4h+3 with ONE 16-byte LDS read and MFMA step j
// pairs (8g+j, 8g+4+j).  4 MFMAs (256 cycles/SIMD) per ds_read_b128 pair.
//
// Pipeline: LDS double buffer; the DMA of chunk kc+1 is issued before the MFMAs of chunk kc, one barrier per K step.
// Tap/channel bookkeeping is incremental (no integer division in the loop).
//
// Epilogue (fused): folded-BN scale/bias, residual add (plain or bilinear-upsampled source = FPN top-down path),
// activation, scatter to up to 3 output segments with independent strides.
#include "common.h"
#include <stdlib.h>
#include <mutex>
#include <type_traits>
#include "../../include/yolact_amd.h"

namespace {

constexpr int BK = 32;
constexpr unsigned OOB = 0x80000000u;  // buffer offset >= num_records (< 2^31, validated) -> the load returns zeros

typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// ---- PREC = 1: fp32-class products on the bf16 matrix pipe ("bf16x3") -------------------------------------------------
// x = h + m + l exactly, three bf16 pieces taken by TRUNCATION (8 + 8 + 8 = 24 significant bits; bf16 has the fp32
// exponent range, so no scaling and no overflow / underflow cases beyond fp32's own).  a*b = sum of 9 piece products, each
// EXACT in the fp32 accumulate of v_mfma_f32_32x32x16_bf16 (8 x 8 bits); the three smallest (m*l, l*m, l*l <= 3 * 2^-24 |ab|)
// are dropped, which is the size of ONE fp32 rounding of the product — the error class of the exact-fp32 MFMA itself
// (tools/split_probe.hip measures both against fp64).  6 bf16 MFMAs at 16x the fp32-MFMA rate = 0.375x the matrix-pipe
// time; the split costs 4 VALU + 1.5 v_perm per element on the separate VALU pipe.
struct Split3 { bf16x8 h, m, l; };

__device__ __forceinline__ Split3 split8(const f32x4 x0, const f32x4 x1) {
  const float x[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
  unsigned r1[8], r2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float h = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x[e]) & 0xFFFF0000u);
    const float r = x[e] - h;                                   // exact: the low 16 mantissa bits
    r1[e] = __builtin_bit_cast(unsigned, r);
    const float m = __builtin_bit_cast(float, r1[e] & 0xFFFF0000u);
    r2[e] = __builtin_bit_cast(unsigned, r - m);               // exact, <= 8 significant bits: already a bf16 value
  }
  u32x4 ph, pm, pl;
#pragma unroll
  for (int q = 0; q < 4; ++q) {                                 // {odd[31:16], even[31:16]}
    ph[q] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, x[2 * q + 1]), __builtin_bit_cast(unsigned, x[2 * q]), 0x07060302u);
    pm[q] = __builtin_amdgcn_perm(r1[2 * q + 1], r1[2 * q], 0x07060302u);
    pl[q] = __builtin_amdgcn_perm(r2[2 * q + 1], r2[2 * q], 0x07060302u);
  }
  Split3 s;
  s.h = __builtin_bit_cast(bf16x8, ph); s.m = __builtin_bit_cast(bf16x8, pm); s.l = __builtin_bit_cast(bf16x8, pl);
  return s;
}


// ---- PREC = 3 / 4: fp32-class products on the fp16 matrix pipe ("fp16x2") ----------------------------------------------
// x * s = h + l, two fp16 pieces by ROUND TO NEAREST (v_cvt_pk_f16_f32): h = fp16(x s), l = fp16(x s - h) with x s - h exact
// in fp32.  11 + 11 significant bits + two signs represent about two thirds of all fp32 values exactly and the rest to one fp32 ulp,
// unbiased.  s = a power of two per tensor (ymi_h2_scale: the producer's magnitude bound -> [2^13, 2^14)), so h never
// overflows and stays a normal fp16 for 27 binades below the tensor's maximum.  a*b = hh + hl + lh (+ ll dropped,
// <= 2^-22 |ab|): 3 MFMAs (v_mfma_f32_32x32x16_f16, exact products, fp32 accumulate) instead of bf16x3's 6, and the split
// is 3 VALU per element with no byte permutes.  tools/split_probe.hip: 574 TFLOP/s fp32-equivalent on random data (bf16x3
// 313, exact fp32 154), error against fp64 2.6e-7 of sum|ab| (bf16x3 3.3e-7, fp32 MFMA 5.2e-7).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
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

struct KParams {
  ymi_conv_desc d;
  int M, HoWo, tiles_n, nk;
  unsigned x_bytes, w_bytes;      // buffer-resource sizes
  const float *offmask;           // DCN only
  int ldo, mask_is_prob, om_layout;
  long x_gs, w_gs, y_gs;          // grouped GEMM (gridDim.y groups, Winograd): element strides of x / w / seg[0].ptr per group
  const void *w3;                 // PREC == 2: filters pre-split into three bf16 planes [groups][3][CoutPad][Kpad]
  unsigned w3_plane;              // PREC == 2: bytes between planes (CoutPad * Kpad * 2)
  unsigned w3_gs;                 // PREC == 2: bytes between groups (3 planes for a Winograd component, K range * 2 for split-K)
                                  // (PREC >= 3: the same three fields describe the TWO fp16 planes of ymi_conv_desc.w_h2)
  const void *a2;                 // PREC == 4: the A operand pre-split into two fp16 planes [groups][2][rows][ldx] (Winograd V)
  unsigned a2_plane;              // PREC == 4: bytes between the two planes
  long a2_gs;                     // PREC == 4: bytes between groups
  unsigned sc_gs;                 // PREC >= 3: floats between the groups' scale_h2 arrays (0: shared)
  unsigned long long *trace;      // diagnostics only (ymi_debug_set_trace): per block {hw id, t0, t_loop, t_epi, t1, t_transposed}
  int abl;                        // diagnostics only (env YMI_ABLATE): bit0 skip the staging of chunks > 0,
                                  // bit2 skip barriers in the K loop — wrong results, used to attribute stall time;
                                  // bit3 disable the residency cap (results unaffected)
};

__device__ __forceinline__ float act_apply(float v, int act) {
  switch (act) {
    case YMI_ACT_RELU: return v > 0.f ? v : 0.f;
    case YMI_ACT_LEAKY01: return v > 0.f ? v : 0.1f * v;
    case YMI_ACT_TANH: return tanhf(v);
    case YMI_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
    default: return v;
  }
}

// torch's area_pixel_compute_source_index for align_corners=False (fp32 arithmetic on purpose,
// see SURVEY appendix A5): src = max(scale*(dst+0.5)-0.5, 0)
__device__ __forceinline__ void bilin_coord(int dst, float scale, int in_size, int &i0, int &i1, float &l1) {
  float src = scale * ((float)dst + 0.5f) - 0.5f;
  src = src < 0.f ? 0.f : src;
  i0 = (int)src;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + ((i0 < in_size - 1) ? 1 : 0);
  l1 = src - (float)i0;
}

// General epilogue (multi-segment scatter, unaligned rows, bilinear FPN residual, tanh / sigmoid): one row per loop
// iteration, NOT unrolled — compact code matters more than ILP here (see the fast path's comment).  Inlined: a real
// call would impose the callee's register budget and a scratch stack on the whole kernel (occupancy 5 -> 2).
// Returns the magnitude bound (max |value written|) of this thread; the caller commits it at a CONVERGED point of the wave
// (the reduction uses cross-lane shuffles, so it must not sit behind the early exit of the lanes past Cout).
template <int BM, int BN, int WK, int RPT, int RSTEP, bool RES_PREFETCH>
__device__ __forceinline__ void epilogue_general_rows(const KParams &p, const float *es, f32x4 sc, f32x4 bi, const f32x4 *rpre,
                                                   int m0, int n, int c4, int rbase, bool vec_res, float invA, float (&amax)[3]) {
  constexpr int ELD = BN + 4;
  const ymi_conv_desc &d = p.d;
  if (n >= d.Cout) return;
  // The segment table lives in the kernel arguments; resolve it with compile-time indices + selects (indexing
  // d.seg[] with a per-lane value turns into dependent per-lane global loads).
  struct SegR { float *ptr; int64_t bs; int rs, act, n0, idx; };
  auto seg_of = [&](int nn) -> SegR {   // segment holding output channel nn (ptr == nullptr: none)
    SegR r = {nullptr, 0, 0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < 3; ++s)
      if (s < d.nseg && nn >= d.seg[s].n0 && nn < d.seg[s].n1 && nn < d.Cout)
        r = SegR{d.seg[s].ptr, d.seg[s].batch_stride, d.seg[s].row_stride, d.seg[s].act, d.seg[s].n0, s};
    return r;
  };
  SegR se[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) se[e] = seg_of(n + e);
  // vector store only if all 4 channels live in one aligned segment
  const bool vec_out = se[0].ptr != nullptr && se[0].ptr == se[3].ptr && ((n - se[0].n0) & 3) == 0 && (se[0].rs & 3) == 0 &&
                       (se[0].bs & 3) == 0 && (((uintptr_t)se[0].ptr) & 15) == 0;
  // magnitude bounds of what this thread writes, one per segment (ymi_conv_desc.y_amax: nseg consecutive slots)
  float rscale_h = 0.f, rscale_w = 0.f;
  if (d.res_mode == YMI_RES_BILINEAR) {
    rscale_h = (float)d.res_H / (float)d.Ho;
    rscale_w = (float)d.res_W / (float)d.Wo;
  }
#pragma unroll 1
  for (int i = 0; i < RPT; ++i) {
    const int row = rbase + RSTEP * i;
    const int m = m0 + row;
    if (m >= p.M) break;
    const int b = m / p.HoWo, pix = m - b * p.HoWo;
    f32x4 v = *reinterpret_cast<const f32x4 *>(es + row * ELD + 4 * c4);
#pragma unroll
    for (int q = 1; q < WK; ++q) v += *reinterpret_cast<const f32x4 *>(es + q * (BM * ELD) + row * ELD + 4 * c4);
    v = (v * invA) * sc + bi;           // invA: fp16x2 activation scale (a power of two: exact), 1 otherwise
    f32x4 rv = {0.f, 0.f, 0.f, 0.f};
    if (d.res_mode == YMI_RES_ADD) {
      const float *rp = d.res + (size_t)m * d.res_ld + n;
      if (vec_res) rv = RES_PREFETCH ? rpre[i] : *reinterpret_cast<const f32x4 *>(rp);
      else {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (n + e < d.Cout) rv[e] = rp[e];
      }
    } else if (d.res_mode == YMI_RES_BILINEAR) {
      const int oy = pix / d.Wo, ox = pix - oy * d.Wo;
      int y0, y1, x0, x1; float ly, lx;
      bilin_coord(oy, rscale_h, d.res_H, y0, y1, ly);
      bilin_coord(ox, rscale_w, d.res_W, x0, x1, lx);
      const float *rb_ = d.res + (size_t)b * d.res_H * d.res_W * d.res_ld + n;
      const float *p00 = rb_ + (size_t)(y0 * d.res_W + x0) * d.res_ld, *p01 = rb_ + (size_t)(y0 * d.res_W + x1) * d.res_ld;
      const float *p10 = rb_ + (size_t)(y1 * d.res_W + x0) * d.res_ld, *p11 = rb_ + (size_t)(y1 * d.res_W + x1) * d.res_ld;
      f32x4 v00 = rv, v01 = rv, v10 = rv, v11 = rv;
      if (vec_res) {
        v00 = *reinterpret_cast<const f32x4 *>(p00); v01 = *reinterpret_cast<const f32x4 *>(p01);
        v10 = *reinterpret_cast<const f32x4 *>(p10); v11 = *reinterpret_cast<const f32x4 *>(p11);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n + e < d.Cout) { v00[e] = p00[e]; v01[e] = p01[e]; v10[e] = p10[e]; v11[e] = p11[e]; }
      }
      rv = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
    }
    f32x4 o = d.res_after_act ? v : v + rv;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = act_apply(o[e], se[e].act);
    if (d.res_after_act) o += rv;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (se[e].ptr != nullptr) {
        const float a = fabsf(o[e]);
        amax[0] = se[e].idx == 0 ? fmaxf(amax[0], a) : amax[0];
        amax[1] = se[e].idx == 1 ? fmaxf(amax[1], a) : amax[1];
        amax[2] = se[e].idx == 2 ? fmaxf(amax[2], a) : amax[2];
      }
    if (vec_out) {
      *reinterpret_cast<f32x4 *>(se[0].ptr + (size_t)b * se[0].bs + (size_t)pix * se[0].rs + (n - se[0].n0)) = o;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (se[e].ptr != nullptr) se[e].ptr[(size_t)b * se[e].bs + (size_t)pix * se[e].rs + (n + e - se[e].n0)] = o[e];
    }
  }
}

// (per-segment bounds, ABI 5: segment k of a multi-segment launch raises slot k of nseg consecutive slots — see csrc/winograd.hip)
template <int BM, int BN, int WK, int RPT, int RSTEP, bool RES_PREFETCH>
__device__ __forceinline__ void epilogue_general(const KParams &p, const float *es, f32x4 sc, f32x4 bi, const f32x4 *rpre,
                                              int m0, int n, int c4, int rbase, bool vec_res, float invA, const ymi_amax_pre &apre) {
  float am[3] = {0.f, 0.f, 0.f};
  epilogue_general_rows<BM, BN, WK, RPT, RSTEP, RES_PREFETCH>(p, es, sc, bi, rpre, m0, n, c4, rbase, vec_res, invA, am);
  if (p.d.y_amax) {
    constexpr int SLOT = YMI_AMAX_SUB * YMI_AMAX_STRIDE;
    ymi_amax_finish(apre, am[0]);
    if (p.d.nseg > 1) ymi_amax_finish(ymi_amax_prefetch(p.d.y_amax + SLOT), am[1]);
    if (p.d.nseg > 2) ymi_amax_finish(ymi_amax_prefetch(p.d.y_amax + 2 * SLOT), am[2]);
  }
}

// LOADER: 0 = Cin % 32 == 0 (a K chunk lies inside one filter tap; tap is block-uniform)
//         1 = Cin == 4 (stem; a K chunk = 8 taps x 4 channels; tap is per-lane)
//         2 = DCNv2 modulated deformable gather (Cin % 32 == 0, 3x3, pad 1): A through registers, B by DMA (WK == 1)
//         3 = POINTWISE: kh = kw = 1, pad = 0 (any stride) — every 1x1 convolution and the grouped Winograd GEMMs
//             (gridDim.y = groups).  A row's source offset is fixed for the whole K loop (one tap, always in range), so a
//             staging piece is ONE buffer_load..lds whose only varying operand is the scalar K-chunk offset: no per-piece
//             predicate / address VALU (the generic loader spends ~10 VALU + SALU per piece on tap bookkeeping, which the
//             bf16x3 tiles — half the matrix-pipe time per chunk — can no longer hide: ablation in profiles/r02_*)
// WK:     waves along K.  WM*WN*WK == 4.  With WK > 1 a pipeline stage holds WK consecutive 32-deep chunks and wave
//         (wm, wn, wk) multiplies chunk wk of every stage; the WK partial tiles are summed (fixed order) in the
//         epilogue's LDS tile.  This quarters the block tile (32x32 with WK = 4) without an inter-block reduction:
//         4x the blocks and 1/4 of the serial K chain for the small-M layers (18x18 ... 5x5 maps) that otherwise
//         leave most CUs idle behind one long K loop.
// NSTAGE: LDS pipeline depth.  An LDS-DMA takes ~1 us from issue to landed under load (MI355X_MICROARCH.md,
//         "ldsdma-fill"), i.e. longer than one K step of MFMA work (0.43 us at 16 MFMAs per wave), so with a prefetch
//         distance of one step a CU needs >= 3 co-resident blocks to keep its matrix pipes fed.  NSTAGE = 3/4 keeps
//         2/3 steps in flight per block (counted s_waitcnt vmcnt(N) + raw s_barrier, never a full drain), which is
//         what the layers whose grid gives each CU only 1-3 blocks need.
// blocks per CU the LDS footprint allows (= waves per SIMD for 256-thread blocks): the register budget handed to the
// compiler, so that the epilogue's prefetch registers never cost a resident block
// floats of LDS per K chunk: A tile [BM][32] fp32 + B tile [BN][32] fp32, or (PREC == 2) B as three bf16 planes [3][BN][32]
template <int BM, int BN, int PREC>
constexpr int conv_sub_floats() { return BM * BK + (PREC == 2 ? BN * BK * 3 / 2 : BN * BK); }   // (fp16x2: 2 planes = BN * BK)

template <int WM, int WN, int WK, int TM, int TN, int NSTAGE, int LOADER, int PREC>
constexpr int conv_occupancy() {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int ns = (LOADER == 2) ? 2 : NSTAGE;
  constexpr int stage_b = ns * conv_sub_floats<BM, BN, PREC>() * WK * 4, epi_b = WK * BM * (BN + 4) * 4;
  constexpr int lds_b = stage_b > epi_b ? stage_b : epi_b;
  constexpr int occ = (160 * 1024) / lds_b;
  // the DCN gather keeps per-tap geometry in registers; the bf16x3 path holds 12 registers of split pieces per 32-row
  // fragment on top of the raw fp32 fragment: budget registers (= blocks per CU) so that neither spills
  constexpr int cap = (LOADER == 2) ? (PREC == 0 ? 3 : 2)
                      : (PREC >= 1 ? (WM * WN * WK == 8 ? 1 : (TM * TN >= 4 ? 2 : (TM * TN == 2 ? (PREC == 1 ? 2 : 3) : 4))) : 5);
  return occ > cap ? cap : (occ < 1 ? 1 : occ);
}

template <int WM, int WN, int WK, int TM, int TN, int NSTAGE, int LOADER, int PREC>
__global__ __launch_bounds__(64 * WM * WN * WK, (conv_occupancy<WM, WN, WK, TM, TN, NSTAGE, LOADER, PREC>() * (WM * WN * WK) / 4))
void conv_igemm_f32(const KParams p) {
#if defined(__HIP_DEVICE_COMPILE__)   // host pass: empty body.  hipcc (ROCm 7.2) silently drops the host launch stub of a
                                      // templated kernel whose body uses the buffer-resource LDS-DMA builtins.
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int NWAVE = WM * WN * WK, NTHR = 64 * NWAVE;   // 4 waves (256 threads) or 8 waves (512 threads)
  constexpr int RPP = NTHR / 8;               // tile rows staged by one pass of the block (8 lanes per row)
  // DMA pieces per wave per chunk: A tile 8 rows x 128 bytes each; B tile likewise, or (PREC == 2, three bf16 planes of
  // 64-byte rows) 16 rows x 64 bytes of one plane each
  constexpr int NPL = (PREC == 2) ? 3 : 2;      // planes of a pre-split operand: bf16x3 three, fp16x2 two
  constexpr bool BPL = PREC >= 2;               // B (filters) staged as 16-bit planes
  constexpr bool APL = PREC == 4;               // A staged as fp16 planes too (no operand split in the loop at all)
  constexpr bool H2 = PREC >= 3;                // fp16x2 arithmetic
  constexpr int RA = APL ? (2 * BM) / (16 * NWAVE) : BM / RPP, RB = BPL ? (NPL * BN) / (16 * NWAVE) : BN / RPP;
  static_assert((APL ? (2 * BM) % (16 * NWAVE) == 0 : BM % RPP == 0) && (BPL ? (NPL * BN) % (16 * NWAVE) == 0 : BN % RPP == 0),
                "tile rows must be a multiple of the rows per staging pass");
  static_assert(!APL || LOADER == 3, "pre-split A planes: pointwise loader only");
  constexpr int SUB = conv_sub_floats<BM, BN, PREC>();        // floats per chunk image
  constexpr int STAGE = SUB * WK;            // floats per pipeline stage
  constexpr int NS = (LOADER == 2) ? 2 : NSTAGE;   // the register-staged DCN gather keeps the simple 2-stage drain
  constexpr int DMA_PER_STEP = (RA + RB) * WK;     // LDS-DMA instructions every wave issues per step
  constexpr int ELD = BN + 4;                // epilogue tile row stride
  constexpr int LDS_FLOATS = (NS * STAGE > WK * BM * ELD) ? NS * STAGE : WK * BM * ELD;
  static_assert(DMA_PER_STEP * (NS - 2) <= 63, "vmcnt is a 6-bit counter");
  static_assert(WM * WN * WK == 4 || WM * WN * WK == 8, "4 or 8 waves per block");
  static_assert(LOADER != 2 || WK == 1, "DCN gather runs without the K split");
  static_assert(PREC == 0 || PREC == 3 || LOADER != 2, "the DCN gather: exact-fp32 or fp16x2 (A split on the fly from the fp32 LDS image)");
  static_assert(!BPL || BN % 16 == 0, "16-bit planes: 16-row DMA pieces");
  __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];

  const ymi_conv_desc &d = p.d;
  const int t = threadIdx.x;
  const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);   // wave-uniform -> SGPR (LDS-DMA bases)
  const int wk = wave % WK, wmn = wave / WK;
  const int wm = wmn / WN, wn = wmn % WN;
  const int kq = t & 7, r0 = t >> 3;          // DMA: physical 16-byte slot and row (within a 32-row group) of this lane
  const int sl = kq ^ ((r0 >> 1) & 7);        // logical k-slot that lives at this lane's physical position

  unsigned long long tr_t0 = 0, tr_loop = 0, tr_epi = 0, tr_tp = 0, tr_c = 0;
  if (p.trace) tr_t0 = __builtin_readcyclecounter();
  const int logical = ymi_xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = logical % p.tiles_n, tile_m = logical / p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int grp = blockIdx.y;    // group of a grouped GEMM (the 16 Winograd components); 0 otherwise
  const __amdgpu_buffer_rsrc_t xrs = APL
      ? __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)p.a2 + (size_t)grp * p.a2_gs), 0, (int)(2 * p.a2_plane), 0x00020000)
      : __builtin_amdgcn_make_buffer_rsrc((void *)(d.x + (size_t)grp * p.x_gs), 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrs = BPL
      ? __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)p.w3 + (size_t)grp * p.w3_gs), 0, (int)(NPL * p.w3_plane), 0x00020000)
      : __builtin_amdgcn_make_buffer_rsrc((void *)(d.w + (size_t)grp * p.w_gs), 0, (int)p.w_bytes, 0x00020000);
  // fp16x2: the power-of-two scale of the activation operand, from the producer's magnitude bound (ymi_h2_scale)
  float sA = 1.f, invA = 1.f;
  if (H2) {
    float xam = d.x_amax ? ymi_amax_read(d.x_amax) : 0.f;
    if (d.x_amax_mul != 0.f) xam *= d.x_amax_mul;
    ymi_h2_scale(xam, sA, invA);
  }
  // magnitude bound of the output (y_amax): what the slot holds NOW, fetched here so that the epilogue need not wait for it
  const ymi_amax_pre apre = ymi_amax_prefetch(d.y_amax);

  // ---- epilogue thread mapping + residual prefetch -----------------------------------------------
  // Each thread owns 4 consecutive output channels of RPT rows.  A plain residual (bottleneck shortcut) is fetched
  // NOW, so its HBM latency overlaps the whole K loop instead of serialising behind it in the epilogue (the
  // K <= 128 1x1 layers at 138x138 are HBM-bound: 2.3 TB/s before this, profiles/r01_*).
  constexpr int C4 = BN / 4;        // float4 columns per tile row
  constexpr int RSTEP = NTHR / C4;  // rows covered by one pass of the block
  constexpr int RPT = BM / RSTEP;   // rows per thread
  constexpr bool RES_PREFETCH = RPT <= 8;
  const int c4 = t % C4, rbase = t / C4;
  const int n = n0 + 4 * c4;
  const bool vec_res = (d.res_ld & 3) == 0 && ((((uintptr_t)d.res) & 15) == 0) && (n + 3 < d.Cout);
  f32x4 rpre[RES_PREFETCH ? RPT : 1];
  if (RES_PREFETCH && d.res_mode == YMI_RES_ADD && vec_res) {
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const int m = m0 + rbase + RSTEP * i;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      rpre[i] = (m < p.M) ? *reinterpret_cast<const f32x4 *>(d.res + (size_t)m * d.res_ld + n) : z;
    }
  }

  // Folded-BN scale / bias are fetched NOW as well: epilogue loads sit behind the other blocks' DMA traffic in the
  // CU's memory queue.
  // (fp16x2 tiles: scale_h2 = scale / the filter row's power-of-two scale, one array per group of a grouped launch)
  const float *scp = H2 ? d.scale_h2 + (size_t)grp * p.sc_gs : d.scale;
  f32x4 sc = {1.f, 1.f, 1.f, 1.f}, bi = {0.f, 0.f, 0.f, 0.f};
  if (n + 3 < d.Cout && (((uintptr_t)scp | (uintptr_t)d.bias) & 15) == 0) {
    if (scp) sc = *reinterpret_cast<const f32x4 *>(scp + n);
    if (d.bias) bi = *reinterpret_cast<const f32x4 *>(d.bias + n);
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (n + e < d.Cout) {
        if (scp) sc[e] = scp[n + e];
        if (d.bias) bi[e] = d.bias[n + e];
      }
    }
  }
  // ---- per-thread A-row bookkeeping (fixed over the K loop) ---------------------------------
  int a_iy0[RA], a_ix0[RA], a_base[RA];        // a_base: byte offset of (pixel, channel 4*sl) for tap (0,0)
#pragma unroll
  for (int i = 0; i < RA; ++i) {
    // pre-split A planes (PREC 4): piece i of this wave = unit u = wave + NWAVE * i of the 2 * BM / 16 (plane, 16-row group)
    // units; lane l fills row l >> 2, physical 16-byte slot l & 3 of a 64-byte row; a_base = byte offset of (plane, pixel,
    // logical slot) for K chunk 0
    const int au = wave + NWAVE * i, apl = APL ? au / (BM / 16) : 0, arow = APL ? (au - apl * (BM / 16)) * 16 + (lane >> 2) : 0;
    const int m = APL ? m0 + arow : m0 + r0 + RPP * i;
    if (m < p.M) {
      const int b = m / p.HoWo, pix = m - b * p.HoWo;
      const int oy = pix / d.Wo, ox = pix - oy * d.Wo;
      a_iy0[i] = oy * d.stride - d.pad;
      a_ix0[i] = ox * d.stride - d.pad;
      a_base[i] = (((b * d.H + a_iy0[i]) * d.W + a_ix0[i]) * d.ldx + 4 * sl) * 4;
      if (APL) a_base[i] = (int)(apl * p.a2_plane) + (((b * d.H + a_iy0[i]) * d.W + a_ix0[i]) * d.ldx + 8 * ((lane & 3) ^ ((arow >> 2) & 3))) * 2;
      if (LOADER == 2) a_base[i] = m;  // DCN: remember the output pixel, geometry recomputed per tap
    } else {
      a_iy0[i] = -(1 << 28);  // never valid
      a_ix0[i] = -(1 << 28);
      a_base[i] = 0;
    }
  }
  unsigned a_voff[LOADER == 3 ? RA : 1];       // POINTWISE: the row's byte offset (or OOB past M), fixed over the K loop
  if (LOADER == 3) {
#pragma unroll
    for (int i = 0; i < RA; ++i) a_voff[i] = (a_iy0[i] > -(1 << 27)) ? (unsigned)a_base[i] : OOB;
  }
  unsigned b_off[RB];                           // byte offset of (filter row, k-slot sl) for chunk 0
  if (BPL) {
    // piece i of this wave = unit u = wave + NWAVE * i of the NPL * BN / 16 (plane, 16-row group) units; lane l fills row
    // l >> 2, physical 16-byte slot l & 3 (64-byte rows); logical slot = physical ^ ((row >> 2) & 3)
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const int u = wave + NWAVE * i, plane = u / (BN / 16), rg = u - plane * (BN / 16);
      const int row = rg * 16 + (lane >> 2), lsl = (lane & 3) ^ ((row >> 2) & 3);
      b_off[i] = (unsigned)plane * p.w3_plane + (unsigned)(((n0 + row) * d.Kpad + 8 * lsl) * 2);
    }
  } else {
#pragma unroll
    for (int i = 0; i < RB; ++i) b_off[i] = (unsigned)(((n0 + r0 + RPP * i) * d.Kpad + 4 * sl) * 4);
  }
  // LDS destination (floats from the chunk's B base) and per-chunk byte advance of a B piece
  auto a_lds = [&](int i) -> int {              // LDS destination (floats from the chunk's A base) of A piece i
    if (APL) {
      const int u = wave + NWAVE * i, plane = u / (BM / 16), rg = u - plane * (BM / 16);
      return plane * (BM * 16) + rg * 256;
    }
    return (wave * 8 + RPP * i) * BK;
  };
  constexpr int A_CHUNK_BYTES = APL ? BK * 2 : BK * 4;
  auto b_lds = [&](int i) -> int {
    if (BPL) {
      const int u = wave + NWAVE * i, plane = u / (BN / 16), rg = u - plane * (BN / 16);
      return plane * (BN * 16) + rg * 256;
    }
    return (wave * 8 + RPP * i) * BK;
  };
  constexpr int B_CHUNK_BYTES = BPL ? BK * 2 : BK * 4;

  // incremental (tap, channel-chunk) state of the next chunk to stage for each of the WK chunk slots (LOADER 0 / 2)
  int nx_c[WK], nx_ky[WK], nx_kx[WK];
  auto advance = [&](int j) {
    nx_c[j] += BK;
    if (nx_c[j] == d.Cin) { nx_c[j] = 0; if (++nx_kx[j] == d.kw) { nx_kx[j] = 0; ++nx_ky[j]; } }
  };
#pragma unroll
  for (int j = 0; j < WK; ++j) {
    nx_c[j] = 0; nx_ky[j] = 0; nx_kx[j] = 0;
    if (LOADER != 1) for (int a = 0; a < j; ++a) advance(j);
  }

  f32x4 ra[LOADER == 2 ? RA : 1];
  int dc_o[LOADER == 2 ? RA : 1][4];      // DCN: corner offsets (elements) and weights (+ modulation) of the current tap
  float dc_w[LOADER == 2 ? RA : 1][5];

  // stage step `st` (chunks st*WK .. st*WK + WK - 1) into LDS stage `buf`
  auto issue_tile = [&](int st, int buf) {
#pragma unroll
    for (int j = 0; j < WK; ++j) {
      const int kc = st * WK + j;
      const bool live = (WK == 1) || (kc < p.nk);   // ragged last step: the slot is zero-filled (OOB) so that every
                                                     // step issues the same number of DMAs (vmcnt accounting)
      float *As = lds + buf * STAGE + j * SUB;
      float *Bs = As + BM * BK;
      if (LOADER == 3) {
#pragma unroll
        for (int i = 0; i < RA; ++i)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_ptr_t)(As + a_lds(i)), 16,
                                                   live ? a_voff[i] : OOB, live ? kc * A_CHUNK_BYTES : 0, 0, 0);
      } else if (LOADER == 0) {
        const int koff = ((nx_ky[j] * d.W + nx_kx[j]) * d.ldx + nx_c[j]) * 4;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
          const bool ok = live && (unsigned)(a_iy0[i] + nx_ky[j]) < (unsigned)d.H &&
                          (unsigned)(a_ix0[i] + nx_kx[j]) < (unsigned)d.W;
          const unsigned voff = ok ? (unsigned)(a_base[i] + koff) : OOB;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_ptr_t)(As + (wave * 8 + RPP * i) * BK), 16, voff, 0, 0, 0);
        }
      } else if (LOADER == 1) {
        const int tap = kc * 8 + sl;
        const int ky = tap / d.kw, kx = tap - ky * d.kw;
        const bool tap_ok = tap < d.kh * d.kw;
        const int koff = ((ky * d.W + kx) * d.ldx - 4 * sl) * 4;   // a_base carries +4*sl channels; Cin == 4 -> channel 0
#pragma unroll
        for (int i = 0; i < RA; ++i) {
          const bool ok = live && tap_ok && (unsigned)(a_iy0[i] + ky) < (unsigned)d.H &&
                          (unsigned)(a_ix0[i] + kx) < (unsigned)d.W;
          const unsigned voff = ok ? (unsigned)(a_base[i] + koff) : OOB;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_ptr_t)(As + (wave * 8 + RPP * i) * BK), 16, voff, 0, 0, 0);
        }
      } else {
        // DCNv2 (dcn_v2_im2col_cuda.cu:143-193): sample point = (oy*s - p + ky + dh, ox*s - p + kx + dw),
        // zero unless -1 < h < H and -1 < w < W; zero-padded bilinear; times sigmoid(mask logit).
        // Thread (kq, r0) produces LOGICAL slot kq of its rows and stores it at the swizzled position.
        // The sampling geometry of a row depends on the tap only, so offsets / sigmoid / corner addresses are
        // resolved once per tap (first channel chunk) and the Cin/32 chunks of the tap issue nothing but their
        // four independent corner loads: no dependent offmask -> address -> data round trip per K step.
        if (nx_c[j] == 0) {
          const int tap = nx_ky[j] * 3 + nx_kx[j];
#pragma unroll
          for (int i = 0; i < RA; ++i) {
            int o1 = 0, o2 = 0, o3 = 0, o4 = 0;
            float w1 = 0.f, w2 = 0.f, w3 = 0.f, w4 = 0.f, mk = 0.f;
            if (a_iy0[i] > -(1 << 27)) {
              const int m = a_base[i];
              const float *om = p.offmask + (size_t)m * p.ldo;
              const float dh = om[p.om_layout ? 3 * tap : 2 * tap], dw = om[p.om_layout ? 3 * tap + 1 : 2 * tap + 1];
              const float mraw = om[p.om_layout ? 3 * tap + 2 : 18 + tap];
              mk = p.mask_is_prob ? mraw : 1.f / (1.f + expf(-mraw));
              const float h = (float)(a_iy0[i] + nx_ky[j]) + dh, w = (float)(a_ix0[i] + nx_kx[j]) + dw;
              if (h > -1.f && w > -1.f && h < (float)d.H && w < (float)d.W) {
                const int hl = (int)floorf(h), wl = (int)floorf(w);
                const int hh = hl + 1, wh = wl + 1;
                const float lh = h - (float)hl, lw = w - (float)wl, uh = 1.f - lh, uw = 1.f - lw;
                const int ib = (m / p.HoWo) * d.H * d.W;
                // an out-of-range corner contributes 0 (dmcn_im2col_bilinear): weight 0 on a safe address
                if (hl >= 0 && wl >= 0) { o1 = (ib + hl * d.W + wl) * d.ldx; w1 = uh * uw; }
                if (hl >= 0 && wh <= d.W - 1) { o2 = (ib + hl * d.W + wh) * d.ldx; w2 = uh * lw; }
                if (hh <= d.H - 1 && wl >= 0) { o3 = (ib + hh * d.W + wl) * d.ldx; w3 = lh * uw; }
                if (hh <= d.H - 1 && wh <= d.W - 1) { o4 = (ib + hh * d.W + wh) * d.ldx; w4 = lh * lw; }
              }
            }
            dc_o[i][0] = o1; dc_o[i][1] = o2; dc_o[i][2] = o3; dc_o[i][3] = o4;
            dc_w[i][0] = w1; dc_w[i][1] = w2; dc_w[i][2] = w3; dc_w[i][3] = w4; dc_w[i][4] = mk;
          }
        }
        const float *xc = d.x + nx_c[j] + 4 * kq;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
          const f32x4 v1 = *reinterpret_cast<const f32x4 *>(xc + dc_o[i][0]);
          const f32x4 v2 = *reinterpret_cast<const f32x4 *>(xc + dc_o[i][1]);
          const f32x4 v3 = *reinterpret_cast<const f32x4 *>(xc + dc_o[i][2]);
          const f32x4 v4 = *reinterpret_cast<const f32x4 *>(xc + dc_o[i][3]);
          ra[i] = (dc_w[i][0] * v1 + dc_w[i][1] * v2 + dc_w[i][2] * v3 + dc_w[i][3] * v4) * dc_w[i][4];
        }
      }
#pragma unroll
      for (int i = 0; i < RB; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_ptr_t)(Bs + b_lds(i)), 16,
                                                 live ? b_off[i] : OOB, live ? kc * B_CHUNK_BYTES : 0, 0, 0);
      if (LOADER != 1 && LOADER != 3) {  // this slot's next chunk is WK chunks further
#pragma unroll
        for (int a = 0; a < WK; ++a) advance(j);
      }
    }
  };
  // DCN only: registers -> swizzled LDS image
  auto store_a_regs = [&](int buf) {
    if (LOADER == 2) {
      float *As = lds + buf * STAGE;
#pragma unroll
      for (int i = 0; i < RA; ++i) *reinterpret_cast<f32x4 *>(As + (r0 + RPP * i) * BK + 4 * sl) = ra[i];
    }
  };

  // The same staging, one 8-row piece (one DMA instruction + its address math) at a time: the main loop drops these
  // between the MFMA groups of the current step, so the ~60 staging instructions of a step issue in the shadow of
  // this wave's own MFMAs instead of ahead of them (ablation: the staging block cost 9 % of the big layers,
  // 130 -> 143 TFLOP/s with it removed).  q enumerates slot j = q / (RA+RB), then the A pieces, then the B pieces.
  constexpr int NP = WK * (RA + RB);
  auto issue_piece = [&](int st, int buf, int q) {
    const int j = q / (RA + RB), r = q - j * (RA + RB);
    const int kc = st * WK + j;
    const bool live = (WK == 1) || (kc < p.nk);
    float *As = lds + buf * STAGE + j * SUB;
    float *Bs = As + BM * BK;
    if (r < RA && LOADER == 3) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_ptr_t)(As + a_lds(r)), 16,
                                               live ? a_voff[r] : OOB, live ? kc * A_CHUNK_BYTES : 0, 0, 0);
    } else if (r < RA) {
      const int i = r;
      bool ok;
      int koff;
      if (LOADER == 0) {
        koff = ((nx_ky[j] * d.W + nx_kx[j]) * d.ldx + nx_c[j]) * 4;
        ok = live && (unsigned)(a_iy0[i] + nx_ky[j]) < (unsigned)d.H && (unsigned)(a_ix0[i] + nx_kx[j]) < (unsigned)d.W;
      } else {   // LOADER == 1
        const int tap = kc * 8 + sl;
        const int ky = tap / d.kw, kx = tap - ky * d.kw;
        koff = ((ky * d.W + kx) * d.ldx - 4 * sl) * 4;
        ok = live && tap < d.kh * d.kw && (unsigned)(a_iy0[i] + ky) < (unsigned)d.H && (unsigned)(a_ix0[i] + kx) < (unsigned)d.W;
      }
      const unsigned voff = ok ? (unsigned)(a_base[i] + koff) : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_ptr_t)(As + (wave * 8 + RPP * i) * BK), 16, voff, 0, 0, 0);
    } else {
      const int i = r - RA;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_ptr_t)(Bs + b_lds(i)), 16,
                                               live ? b_off[i] : OOB, live ? kc * B_CHUNK_BYTES : 0, 0, 0);
      if (r == RA + RB - 1 && LOADER != 1 && LOADER != 3) {   // last piece of slot j: its next chunk is WK chunks further
#pragma unroll
        for (int a = 0; a < WK; ++a) advance(j);
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment addressing: row = lane & 31 (+ tile offsets), logical slot 2g + h at physical (2g + h) ^ f
  const int fsw = (lane >> 1) & 7, hh_ = lane >> 5;
  const int frag_row = (lane & 31) * BK;
  int fo[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) fo[g] = frag_row + 4 * ((2 * g + hh_) ^ fsw);

  // Fragments are double-buffered in registers: the ds_read_b128s of k-group g+1 are issued before the 4*TM*TN
  // MFMAs of group g, so LDS latency hides behind the matrix pipe even with a single wave on the SIMD.  The first
  // two groups of a stage are requested BEFORE the next chunk's DMA is issued, so their latency overlaps the
  // staging address math instead of stalling the first MFMA.
  f32x4 fa[2][TM], fb[2][TN];
  auto load_frag = [&](int buf, int g, int slot) {
    const float *As = lds + buf * STAGE + wk * SUB + (wm * TM * 32) * BK;
    const float *Bs = lds + buf * STAGE + wk * SUB + BM * BK + (wn * TN * 32) * BK;
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[slot][i] = *reinterpret_cast<const f32x4 *>(As + i * 32 * BK + fo[g]);
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[slot][j] = *reinterpret_cast<const f32x4 *>(Bs + j * 32 * BK + fo[g]);
  };
  // hook h (0..15) sits behind the TM*TN MFMAs of k-pair step h = 4*g + s; piece q goes to hook (16*q) / NP
  auto compute = [&](int buf, bool stage_next, int nst, int nbuf) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int slot = g & 1;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[slot][i][s], fb[slot][j][s], acc[i][j], 0, 0, 0);
        if (LOADER != 2) {
#pragma unroll
          for (int q = 0; q < NP; ++q) {
            if ((16 * q) / NP == 4 * g + s) {
              __builtin_amdgcn_sched_barrier(0);
              if (stage_next) issue_piece(nst, nbuf, q);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
      }
      if (g + 2 < 4) {
        __builtin_amdgcn_sched_barrier(0);   // keep the prefetch right behind the MFMAs that free its registers
        load_frag(buf, g + 2, slot);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  // ---- PREC == 1: the same chunk as 2 steps of 16 k; lane half h of step s holds k = 16 s + 8 h .. + 7 = logical 16-byte
  // slots 4s + 2h and 4s + 2h + 1 of its row (A and B agree on the K order, which is all a dot product needs).
  // Raw fp32 fragments of step s + 1 are requested right after step s has been split, so their LDS latency and the
  // splitting VALU work of the next step overlap this step's 6 * TM * TN bf16 MFMAs (VALU and matrix pipes are separate).
  // raw fp32 fragments / filter-plane fragments, one set per step of the chunk (compile-time indexed: no register copies)
  f32x4 rwa[2][TM][2], rwb[2][TN][2];     // (dead, hence register-free, when PREC == 0)
  Split3 pbn[2][TN];                      // PREC == 2: the B pieces, read straight from the LDS planes
  Split2 pbh[2][TN], pah[2][TM];          // PREC >= 3: fp16x2 B pieces (and, PREC == 4, A pieces) from the LDS planes
  int fo2[2][2];
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
    for (int q = 0; q < 2; ++q) fo2[s2][q] = frag_row + 4 * ((4 * s2 + 2 * hh_ + q) ^ fsw);
  const int psw = ((lane & 31) >> 2) & 3;   // plane image: 64-byte rows, 16-byte slot s of row n lives at slot s ^ ((n >> 2) & 3)
  auto load_raw = [&](int buf, auto s2c) {
    constexpr int s2 = decltype(s2c)::value;
    const float *As = lds + buf * STAGE + wk * SUB + (wm * TM * 32) * BK;
    if constexpr (APL) {       // plane image: 64-byte rows (16 floats), plane stride BM * 16 floats
      const float *Ap = lds + buf * STAGE + wk * SUB + ((wm * TM * 32 + (lane & 31)) * 16 + 4 * ((2 * s2 + hh_) ^ psw));
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        pah[s2][i].h = *reinterpret_cast<const f16x8 *>(Ap + i * 32 * 16);
        pah[s2][i].l = *reinterpret_cast<const f16x8 *>(Ap + i * 32 * 16 + BM * 16);
      }
    } else {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        rwa[s2][i][0] = *reinterpret_cast<const f32x4 *>(As + i * 32 * BK + fo2[s2][0]);
        rwa[s2][i][1] = *reinterpret_cast<const f32x4 *>(As + i * 32 * BK + fo2[s2][1]);
      }
    }
    if constexpr (H2) {
      const float *Bp = lds + buf * STAGE + wk * SUB + BM * BK + ((wn * TN * 32 + (lane & 31)) * 16 + 4 * ((2 * s2 + hh_) ^ psw));
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        pbh[s2][j].h = *reinterpret_cast<const f16x8 *>(Bp + j * 32 * 16);
        pbh[s2][j].l = *reinterpret_cast<const f16x8 *>(Bp + j * 32 * 16 + BN * 16);
      }
    } else if constexpr (PREC == 2) {
      const float *Bp = lds + buf * STAGE + wk * SUB + BM * BK + ((wn * TN * 32 + (lane & 31)) * 16 + 4 * ((2 * s2 + hh_) ^ psw));
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        pbn[s2][j].h = *reinterpret_cast<const bf16x8 *>(Bp + j * 32 * 16);
        pbn[s2][j].m = *reinterpret_cast<const bf16x8 *>(Bp + j * 32 * 16 + BN * 16);
        pbn[s2][j].l = *reinterpret_cast<const bf16x8 *>(Bp + j * 32 * 16 + 2 * BN * 16);
      }
    } else {
      const float *Bs = lds + buf * STAGE + wk * SUB + BM * BK + (wn * TN * 32) * BK;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        rwb[s2][j][0] = *reinterpret_cast<const f32x4 *>(Bs + j * 32 * BK + fo2[s2][0]);
        rwb[s2][j][1] = *reinterpret_cast<const f32x4 *>(Bs + j * 32 * BK + fo2[s2][1]);
      }
    }
  };
  // STAGE_NEXT is a compile-time flag: the staging pieces sit between the MFMAs unconditionally, and the (rare) steps that
  // stage nothing run a second copy of the loop body without them — instead of one scalar branch per piece per chunk
  auto compute_x3 = [&](int buf, auto stage_c, int nst, int nbuf) {
    constexpr bool STAGE_NEXT = decltype(stage_c)::value;
    constexpr int NPOS = 12 * TM * TN;         // hook positions: behind every MFMA
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      Split3 xa[TM], xbs[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) xa[i] = split8(rwa[s2][i][0], rwa[s2][i][1]);
      if constexpr (PREC != 2) {
#pragma unroll
        for (int j = 0; j < TN; ++j) xbs[j] = split8(rwb[s2][j][0], rwb[s2][j][1]);
      }
      if (s2 == 0) load_raw(buf, std::integral_constant<int, 1>{});
      // product-major order: consecutive MFMAs hit different accumulators (no back-to-back dependent issue); per
      // accumulator the small cross terms still come first and the dominant h*h product last
#pragma unroll
      for (int pr = 0; pr < 6; ++pr)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const Split3 &xb = (PREC == 2) ? pbn[s2][j] : xbs[j];
            const bf16x8 fa_ = pr == 0 ? xa[i].h : pr == 1 ? xa[i].l : pr == 2 ? xa[i].m : pr == 3 ? xa[i].h : pr == 4 ? xa[i].m : xa[i].h;
            const bf16x8 fb_ = pr == 0 ? xb.l : pr == 1 ? xb.h : pr == 2 ? xb.m : pr == 3 ? xb.m : pr == 4 ? xb.h : xb.h;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa_, fb_, acc[i][j], 0, 0, 0);
            if constexpr (STAGE_NEXT) {
              const int pos = ((s2 * 6 + pr) * TM + i) * TN + j;
#pragma unroll
              for (int q = 0; q < NP; ++q) {
                if ((NPOS * q) / NP == pos) {
                  __builtin_amdgcn_sched_barrier(0);
                  issue_piece(nst, nbuf, q);
                  __builtin_amdgcn_sched_barrier(0);
                }
              }
            }
          }
    }
  };
  // fp16x2: the same chunk structure with 3 products per (A, B) fragment pair — h*l, l*h first, the dominant h*h last
  auto compute_h2 = [&](int buf, auto stage_c, int nst, int nbuf) {
    constexpr bool STAGE_NEXT = decltype(stage_c)::value;
    constexpr int NPOS = 6 * TM * TN;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      Split2 xa[TM];
      if constexpr (!APL) {
#pragma unroll
        for (int i = 0; i < TM; ++i) xa[i] = split8h(rwa[s2][i][0], rwa[s2][i][1], sA);
      }
      if (s2 == 0) load_raw(buf, std::integral_constant<int, 1>{});
#pragma unroll
      for (int pr = 0; pr < 3; ++pr)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const Split2 &a_ = APL ? pah[s2][i] : xa[i];
            const f16x8 fa_ = pr == 1 ? a_.l : a_.h;
            const f16x8 fb_ = pr == 0 ? pbh[s2][j].l : pbh[s2][j].h;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa_, fb_, acc[i][j], 0, 0, 0);
            if constexpr (STAGE_NEXT) {
              const int pos = ((s2 * 3 + pr) * TM + i) * TN + j;
#pragma unroll
              for (int q = 0; q < NP; ++q) {
                if ((NPOS * q) / NP == pos) {
                  __builtin_amdgcn_sched_barrier(0);
                  issue_piece(nst, nbuf, q);
                  __builtin_amdgcn_sched_barrier(0);
                }
              }
            }
          }
    }
  };
  // split-precision chunk, bf16x3 or fp16x2 by the template's PREC
  auto compute_sp = [&](int buf, auto stage_c, int nst, int nbuf) {
    if constexpr (H2) compute_h2(buf, stage_c, nst, nbuf);
    else compute_x3(buf, stage_c, nst, nbuf);
  };
  // precision-independent entry points of the main loop
  auto prefetch_frags = [&](int buf) {
    if constexpr (PREC >= 1) {
      load_raw(buf, std::integral_constant<int, 0>{});
    } else {
      load_frag(buf, 0, 0);
      load_frag(buf, 1, 1);
    }
  };
  auto compute_chunk = [&](int buf, bool stage_next, int nst, int nbuf) {
    if constexpr (PREC >= 1) {
      if (stage_next) compute_sp(buf, std::true_type{}, nst, nbuf);
      else compute_sp(buf, std::false_type{}, nst, nbuf);
    } else {
      compute(buf, stage_next, nst, nbuf);
    }
  };
  // a wave without a chunk of its own in a ragged K-split step still stages its share
  auto stage_only = [&](int nst, int nbuf) {
#pragma unroll
    for (int q = 0; q < NP; ++q) issue_piece(nst, nbuf, q);
  };

  // ---- main loop ---------------------------------------------------------------------------
  // s_waitcnt vmcnt(N) only (expcnt / lgkmcnt fields at "no wait"): gfx9 encoding vmcnt[3:0] | expcnt[6:4] |
  // lgkmcnt[11:8] | vmcnt_hi[15:14].  The asm memory clobber keeps the compiler from moving LDS accesses across.
#define YMI_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
#define YMI_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
  const int nsteps = (p.nk + WK - 1) / WK;
  if (p.trace) tr_loop = __builtin_readcyclecounter();
  // bf16x3 tiles without an intra-block K split: the loop is peeled into "every step stages the next one" + the final
  // NS-1 steps that stage nothing, so the body has no runtime branch around the staging pieces and the accumulators flow
  // through straight-line code (a join of two body variants inside the loop costs 16 v_mov per accumulator per chunk)
  constexpr bool PEELED = PREC >= 1 && WK == 1 && LOADER != 2;
#ifdef YMI_DIAGNOSTICS
  const bool no_ablation = p.abl == 0;     // the ablation switches live in the generic loop below
#else
  constexpr bool no_ablation = true;       // product build: the generic loop is not even compiled into the peeled kernels
#endif
  if (PEELED && no_ablation) {
    if (NS == 2) {
      issue_tile(0, 0);
      __syncthreads();   // drains the DMA (vmcnt(0)) then barrier
      // two steps per trip with compile-time buffer indices: the compiler copies every accumulator once per loop trip
      // (16 v_mov per 32x32 tile; it does not coalesce the loop-carried MFMA accumulators), so a longer trip halves that
      // (only where the longer trip fits the register budget: the 1x2 wave tiles; 1x1 and 2x2 tiles spill with it)
      constexpr bool UNROLL2 = TM * TN == 2;
      int st = 0;
      for (; UNROLL2 && st + 2 < nsteps; st += 2) {
        prefetch_frags(0);
        __builtin_amdgcn_sched_barrier(0);
        compute_sp(0, std::true_type{}, st + 1, 1);
        __syncthreads();
        prefetch_frags(1);
        __builtin_amdgcn_sched_barrier(0);
        compute_sp(1, std::true_type{}, st + 2, 0);
        __syncthreads();
      }
      for (; st + 1 < nsteps; ++st) {              // remaining staged steps (all of them without UNROLL2)
        const int cur = st & 1;
        prefetch_frags(cur);
        __builtin_amdgcn_sched_barrier(0);
        compute_sp(cur, std::true_type{}, st + 1, cur ^ 1);
        __syncthreads();
      }
      prefetch_frags(st & 1);
      __builtin_amdgcn_sched_barrier(0);
      compute_sp(st & 1, std::false_type{}, 0, 0);
      __syncthreads();
    } else {
#pragma unroll
      for (int s0 = 0; s0 < NS - 1; ++s0)
        if (s0 < nsteps) issue_tile(s0, s0);
      {
        const int fl = (nsteps < NS - 1 ? nsteps : NS - 1) - 1;
        if (NS >= 4 && fl >= 2) YMI_WAIT_VM(2 * DMA_PER_STEP);
        else if (fl >= 1) YMI_WAIT_VM(DMA_PER_STEP);
        else YMI_WAIT_VM(0);
      }
      YMI_BARRIER();
      int cur = 0, nxt = NS - 1, st = 0;
      for (; st + NS - 1 < nsteps; ++st) {          // steady state: NS - 2 later steps stay in flight behind step st + 1
        prefetch_frags(cur);
        __builtin_amdgcn_sched_barrier(0);
        compute_sp(cur, std::true_type{}, st + NS - 1, nxt);
        if (NS >= 4) YMI_WAIT_VM(2 * DMA_PER_STEP); else YMI_WAIT_VM(DMA_PER_STEP);
        YMI_BARRIER();
        cur = (cur + 1 == NS) ? 0 : cur + 1;
        nxt = (nxt + 1 == NS) ? 0 : nxt + 1;
      }
      for (; st < nsteps; ++st) {                   // drain: nothing left to stage
        prefetch_frags(cur);
        __builtin_amdgcn_sched_barrier(0);
        compute_sp(cur, std::false_type{}, 0, 0);
        if (st + 1 < nsteps) {
          const int rem = nsteps - 2 - st;
          if (NS >= 4 && rem >= 2) YMI_WAIT_VM(2 * DMA_PER_STEP);
          else if (rem >= 1) YMI_WAIT_VM(DMA_PER_STEP);
          else YMI_WAIT_VM(0);
        }
        YMI_BARRIER();
        cur = (cur + 1 == NS) ? 0 : cur + 1;
      }
    }
  } else
  if (NS == 2) {
    issue_tile(0, 0);
    store_a_regs(0);
    __syncthreads();   // drains the DMA (vmcnt(0)) then barrier
    for (int st = 0; st < nsteps; ++st) {
      const int cur = st & 1;
      const bool more = (st + 1) < nsteps;
      const bool mine = (WK == 1) || (st * WK + wk < p.nk);   // wave-uniform: does this wave's chunk exist?
      if (mine) prefetch_frags(cur);
      __builtin_amdgcn_sched_barrier(0);
      const bool stage = more && !(p.abl & 1);
      if (LOADER == 2) {
        if (stage) issue_tile(st + 1, cur ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        compute_chunk(cur, false, 0, 0);          // exact-fp32 MFMA, or fp16x2 on the gathered fp32 tile (PREC 3)
      } else if (mine) {
        compute_chunk(cur, stage, st + 1, cur ^ 1);
      } else if (stage) {
        stage_only(st + 1, cur ^ 1);
      }
      if (more) store_a_regs(cur ^ 1);
      if (!(p.abl & 4)) __syncthreads();
    }
  } else {
    // NS-1 steps in flight.  Stage of step s = s % NS.  At the end of iteration st the data of step st+1 must have
    // landed: every wave waits until at most (steps issued beyond st+1) * DMA_PER_STEP of ITS DMAs are outstanding
    // (they complete in order), then the barrier makes all waves' pieces visible.  Re-staging slot (st-1) % NS in
    // iteration st is safe: its last readers passed the barrier that ended iteration st-1.
#pragma unroll
    for (int s0 = 0; s0 < NS - 1; ++s0)
      if (s0 < nsteps) issue_tile(s0, s0);
    {
      const int fl = (nsteps < NS - 1 ? nsteps : NS - 1) - 1;   // steps allowed to stay in flight behind step 0
      if (NS >= 4 && fl >= 2) YMI_WAIT_VM(2 * DMA_PER_STEP);
      else if (fl >= 1) YMI_WAIT_VM(DMA_PER_STEP);
      else YMI_WAIT_VM(0);
    }
    YMI_BARRIER();
    int cur = 0, nxt = NS - 1;   // stage of step st, stage to refill (= stage of step st + NS - 1)
    for (int st = 0; st < nsteps; ++st) {
      const bool mine = (WK == 1) || (st * WK + wk < p.nk);
      if (mine) prefetch_frags(cur);
      __builtin_amdgcn_sched_barrier(0);
      const bool stage = st + NS - 1 < nsteps && !(p.abl & 1);
      if (mine) compute_chunk(cur, stage, st + NS - 1, nxt);
      else if (stage) stage_only(st + NS - 1, nxt);
      if (st + 1 < nsteps) {
        const int rem = nsteps - 2 - st;                        // steps issued beyond st+1
        const int fl = rem < NS - 2 ? rem : NS - 2;
        if (NS >= 4 && fl >= 2) YMI_WAIT_VM(2 * DMA_PER_STEP);
        else if (fl >= 1) YMI_WAIT_VM(DMA_PER_STEP);
        else YMI_WAIT_VM(0);
      }
      if (!(p.abl & 4)) YMI_BARRIER();
      cur = (cur + 1 == NS) ? 0 : cur + 1;
      nxt = (nxt + 1 == NS) ? 0 : nxt + 1;
    }
  }
#undef YMI_WAIT_VM
#undef YMI_BARRIER

  // ---- epilogue ----------------------------------------------------------------------------
  // Accumulators -> LDS tile [WK][BM][BN+4] -> each thread owns 4 consecutive output channels of a row, so
  // residual loads and output stores are 16 bytes per lane and 16*C4 contiguous bytes per row
  // (the first version stored one dword per lane per accumulator register and was store-issue bound on
  // the K <= 128 1x1 layers: 18-35 TF/s; see profiles/).
  // C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8*(r >> 2) + 4*(lane >> 5).
  if (p.trace) tr_epi = __builtin_readcyclecounter();
  float *es = lds;  // the main loop ended with a barrier: LDS is free
  {
    const int ncol = lane & 31, half = lane >> 5;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          es[wk * (BM * ELD) + ((wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * ELD + (wn * TN + j) * 32 + ncol] =
              acc[i][j][r];
  }
  __syncthreads();

  if (p.trace) tr_tp = __builtin_readcyclecounter();   // accumulators transposed through LDS
  // ---- FAST PATH (block-uniform test): one dense, 16-byte aligned output segment, Cout % 4 == 0, residual none or
  // prefetched, activation none / ReLU / LeakyReLU.  Covers every backbone / FPN-pred / protonet / upfeature layer.
  // Kept tiny on purpose: the general path below is ~6000 instructions of divergent code and an epilogue wave
  // streaming through it cold took 30k-70k cycles per block in steady state (instruction fetch behind the other
  // blocks' memory traffic; block traces in profiles/), 8x longer than the same epilogue on an idle CU.
  {
    const ymi_conv_seg &g0 = d.seg[0];
    const bool fast = d.nseg == 1 && g0.n0 == 0 && g0.n1 >= d.Cout && (d.Cout & 3) == 0 && (g0.row_stride & 3) == 0 &&
                      (((uintptr_t)g0.ptr) & 15) == 0 && g0.batch_stride == (int64_t)p.HoWo * g0.row_stride &&
                      g0.act <= YMI_ACT_LEAKY01 &&
                      (d.res_mode == YMI_RES_NONE ||
                       (RES_PREFETCH && d.res_mode == YMI_RES_ADD && (d.res_ld & 3) == 0 && ((((uintptr_t)d.res) & 15) == 0)));
    if (fast) {
      // act(x) = max(x, slope * x): slope 0 -> ReLU, 0.1 -> LeakyReLU(0.1), 1 -> identity (exact: one mul + one max)
      const float slope = g0.act == YMI_ACT_RELU ? 0.f : (g0.act == YMI_ACT_LEAKY01 ? 0.1f : 1.f);
      const bool has_res = d.res_mode == YMI_RES_ADD, after = d.res_after_act != 0;
      f32x4 o[RPT];
#pragma unroll
      for (int i = 0; i < RPT; ++i) {
        const int row = rbase + RSTEP * i;
        f32x4 v = *reinterpret_cast<const f32x4 *>(es + row * ELD + 4 * c4);
#pragma unroll
        for (int q = 1; q < WK; ++q) v += *reinterpret_cast<const f32x4 *>(es + q * (BM * ELD) + row * ELD + 4 * c4);
        if (H2) v = v * invA;             // fp16x2 activation scale: a power of two, exact
        v = v * sc + bi;
        f32x4 rv = {0.f, 0.f, 0.f, 0.f};
        if (RES_PREFETCH && has_res) rv = rpre[i];
        if (!after) v += rv;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], slope * v[e]);
        if (after) v += rv;
        o[i] = v;
      }
      if (n < d.Cout) {
        float *base = g0.ptr + (size_t)grp * p.y_gs + (size_t)(m0 + rbase) * g0.row_stride + n;
#pragma unroll
        for (int i = 0; i < RPT; ++i)
          if (m0 + rbase + RSTEP * i < p.M) *reinterpret_cast<f32x4 *>(base + (size_t)(RSTEP * i) * g0.row_stride) = o[i];
      }
      if (d.y_amax) {                     // magnitude bound of the tile: block-uniform condition, every lane takes part
        float am = 0.f;
        if (n < d.Cout) {
#pragma unroll
          for (int i = 0; i < RPT; ++i)
            if (m0 + rbase + RSTEP * i < p.M) am = fmaxf(am, ymi_absmax4(o[i]));
        }
        ymi_amax_finish(apre, am);
      }
    } else {
      epilogue_general<BM, BN, WK, RPT, RSTEP, RES_PREFETCH>(p, es, sc, bi, rpre, m0, n, c4, rbase, vec_res, invA, apre);
    }
  }
  if (p.trace && t == 0) {
    // HW_REG_HW_ID (4): wave[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13]...; HW_REG_XCC_ID (20): xcc[3:0]
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
    unsigned long long *o = p.trace + (size_t)blockIdx.x * 8;
    o[0] = ((unsigned long long)xcc << 32) | hw;
    o[1] = tr_t0; o[2] = tr_loop; o[3] = tr_epi; o[4] = __builtin_readcyclecounter(); o[5] = tr_tp; o[6] = tr_c;
  }
#endif  // __HIP_DEVICE_COMPILE__
}

// ---- host side --------------------------------------------------------------------------------
struct ProfRec { hipEvent_t e0, e1; double flops; int tile; int kind; };
constexpr int PROF_MAX = 4096;
ProfRec g_prof[PROF_MAX];
unsigned long long *g_trace = nullptr;
long g_trace_cap = 0;
int g_prof_n = 0, g_prof_alloc = 0, g_prof_on = 0;
std::mutex g_prof_mu;   // the opt-in profiling records are process-global: threaded callers (eval.py's evalvideo pool) may
                        // launch concurrently, so slot allocation is serialised; with profiling off nothing is locked

// The workgroup dispatcher does not balance a grid that fits in one residency round: it packs up to `occupancy`
// blocks on a CU while others hold fewer (a 616-block layer ran as if its busiest CU held 4+ blocks, not 3).  When
// the grid is at most occ*256 blocks we therefore cap residency at k = ceil(grid / 256) blocks per CU by padding
// the block's LDS allocation with unused dynamic LDS, so no CU can take more than its share.
constexpr int LDS_PER_CU = 160 * 1024, NUM_CU = 256;

template <int WM, int WN, int WK, int TM, int TN, int NS, bool ALL_LOADERS, int PREC>
int launch_cfg(const KParams &kp, int loader, hipStream_t s, int groups) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int ns_eff_bytes = NS * conv_sub_floats<BM, BN, PREC>() * WK * 4, epi_bytes = WK * BM * (BN + 4) * 4;
  constexpr int static_lds = ns_eff_bytes > epi_bytes ? ns_eff_bytes : epi_bytes;
  KParams p = kp;
  const int tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.d.Cout + BN - 1) / BN;
  const int grid = tiles_m * p.tiles_n;
  if (p.trace && (grid > g_trace_cap || groups > 1)) p.trace = nullptr;
  int dyn = 0;
  {
    int occ = LDS_PER_CU / static_lds;                     // LDS-limited residency (VGPRs allow >= this for every fp32 tile)
    if (PREC >= 1) {
      constexpr int occ_regs = conv_occupancy<WM, WN, WK, TM, TN, NS, 0, PREC>();
      occ = occ < occ_regs ? occ : occ_regs;
    }
    const int k = (grid * groups + NUM_CU - 1) / NUM_CU;    // blocks per CU if perfectly spread
    if (k < occ && !(p.abl & 8)) {
      const int want = LDS_PER_CU / (k + 1) + 1024;         // > 160K/(k+1)  =>  at most k blocks fit
      if (want > static_lds && want <= LDS_PER_CU / k) dyn = want - static_lds;
    }
  }
  const bool pointwise = p.d.kh == 1 && p.d.kw == 1 && p.d.pad == 0;
  if constexpr (PREC == 4) {             // pre-split A planes exist for the pointwise loader only (the grouped Winograd GEMMs)
    if (loader != 0 || !pointwise) return YMI_EARG;
    hipLaunchKernelGGL((conv_igemm_f32<WM, WN, WK, TM, TN, NS, 3, 4>), dim3(grid, groups), dim3(64 * WM * WN * WK), dyn, s, p);
    return ymi_launch_status();
  } else
  if (loader == 0 && (groups > 1 || pointwise)) {
    hipLaunchKernelGGL((conv_igemm_f32<WM, WN, WK, TM, TN, NS, 3, PREC>), dim3(grid, groups), dim3(64 * WM * WN * WK), dyn, s, p);
  } else if (loader == 0) {
    hipLaunchKernelGGL((conv_igemm_f32<WM, WN, WK, TM, TN, NS, 0, PREC>), dim3(grid, groups), dim3(64 * WM * WN * WK), dyn, s, p);
  } else if constexpr (ALL_LOADERS) {   // stem (Cin = 4) and DCN gather loaders: basic tiles only; DCN: exact-fp32 MFMA only
    if (loader == 1) hipLaunchKernelGGL((conv_igemm_f32<WM, WN, WK, TM, TN, NS, 1, PREC>), dim3(grid, groups), dim3(64 * WM * WN * WK), dyn, s, p);
    else if constexpr (PREC == 0) hipLaunchKernelGGL((conv_igemm_f32<WM, WN, WK, TM, TN, NS, 2, 0>), dim3(grid, groups), dim3(64 * WM * WN * WK), dyn, s, p);
    else if constexpr (PREC == 3) hipLaunchKernelGGL((conv_igemm_f32<WM, WN, WK, TM, TN, NS, 2, 3>), dim3(grid, groups), dim3(64 * WM * WN * WK), dyn, s, p);
    else return YMI_EARG;
  } else {
    return YMI_EARG;
  }
  return ymi_launch_status();
}

// bf16x3 tiles: with pre-split filter planes (d->w_x3) only the activations are split on the fly (PREC 2), else both (PREC 1)
// fp16x2 tiles (table PREC 3): filter planes always; with pre-split activation planes (kp.a2) nothing is split in the loop (PREC 4)
template <int WM, int WN, int WK, int TM, int TN, int NS, bool ALL_LOADERS, int PREC>
int launch_prec(const KParams &kp, int loader, hipStream_t s, int groups) {
  if constexpr (PREC == 3) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NWAVE = WM * WN * WK;
    if constexpr ((2 * BN) % (16 * NWAVE) != 0) return YMI_EARG;
    else {
      if (kp.a2 != nullptr) {
        if constexpr ((2 * BM) % (16 * NWAVE) == 0) return launch_cfg<WM, WN, WK, TM, TN, NS, ALL_LOADERS, 4>(kp, loader, s, groups);
        else return YMI_EARG;
      }
      return launch_cfg<WM, WN, WK, TM, TN, NS, ALL_LOADERS, 3>(kp, loader, s, groups);
    }
  } else
  if constexpr (PREC == 1) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NWAVE = WM * WN * WK;
    if constexpr (BN % 16 == 0 && (3 * BN) % (16 * NWAVE) == 0 && NS * conv_sub_floats<BM, BN, 2>() * WK * 4 <= 160 * 1024) {
      if (kp.w3 != nullptr) return launch_cfg<WM, WN, WK, TM, TN, NS, ALL_LOADERS, 2>(kp, loader, s, groups);
    }
  }
  return launch_cfg<WM, WN, WK, TM, TN, NS, ALL_LOADERS, PREC>(kp, loader, s, groups);
}

// tile id -> (WM, WN, WK, TM, TN, NSTAGE, all loaders?, PREC).  ids | YMI_TILE_X3 (32): the same block tile computed as
// bf16x3 (fp32-class products on the bf16 matrix pipe, see split8()).
#define YMI_TILE_TABLE(X)                                      \
  X(YMI_TILE_128x128, 2, 2, 1, 2, 2, 2, true, 0)               \
  X(YMI_TILE_128x64, 2, 2, 1, 2, 1, 2, true, 0)                \
  X(YMI_TILE_64x64, 2, 2, 1, 1, 1, 2, true, 0)                 \
  X(YMI_TILE_128x32, 4, 1, 1, 1, 1, 2, true, 0)                \
  X(YMI_TILE_64x128, 2, 2, 1, 1, 2, 2, true, 0)                \
  X(YMI_TILE_32x32_K4, 1, 1, 4, 1, 1, 2, false, 0)             \
  X(YMI_TILE_64x32_K2, 2, 1, 2, 1, 1, 2, false, 0)             \
  X(YMI_TILE_32x64_K2, 1, 2, 2, 1, 1, 2, false, 0)             \
  X(YMI_TILE_64x64_S3, 2, 2, 1, 1, 1, 3, false, 0)             \
  X(YMI_TILE_64x64_S4, 2, 2, 1, 1, 1, 4, false, 0)             \
  X(YMI_TILE_64x128_S3, 2, 2, 1, 1, 2, 3, false, 0)            \
  X(YMI_TILE_128x64_S3, 2, 2, 1, 2, 1, 3, false, 0)            \
  X(YMI_TILE_32x32_K4_S4, 1, 1, 4, 1, 1, 4, false, 0)          \
  X(YMI_TILE_64x32_K2_S3, 2, 1, 2, 1, 1, 3, false, 0)          \
  X(YMI_TILE_32x64_K2_S3, 1, 2, 2, 1, 1, 3, false, 0)          \
  X(YMI_TILE_128x128_W8, 4, 2, 1, 1, 2, 2, false, 0)           \
  X(YMI_TILE_256x128_W8, 4, 2, 1, 2, 2, 2, false, 0)           \
  X(YMI_TILE_128x256_W8, 4, 2, 1, 1, 4, 2, false, 0)           \
  X(YMI_TILE_128x128_S3, 2, 2, 1, 2, 2, 3, false, 0)           \
  X(YMI_TILE_128x128_W8_S3, 4, 2, 1, 1, 2, 3, false, 0)        \
  X(YMI_TILE_256x128_W8_S3, 4, 2, 1, 2, 2, 3, false, 0)        \
  X(YMI_TILE_128x128_W8_S4, 4, 2, 1, 1, 2, 4, false, 0)        \
  X(YMI_TILE_X3 | YMI_TILE_128x128, 2, 2, 1, 2, 2, 2, true, 1)      \
  X(YMI_TILE_X3 | YMI_TILE_128x64, 2, 2, 1, 2, 1, 2, true, 1)       \
  X(YMI_TILE_X3 | YMI_TILE_64x64, 2, 2, 1, 1, 1, 2, true, 1)        \
  X(YMI_TILE_X3 | YMI_TILE_64x128, 2, 2, 1, 1, 2, 2, true, 1)       \
  X(YMI_TILE_X3 | YMI_TILE_32x32_K4, 1, 1, 4, 1, 1, 2, false, 1)    \
  X(YMI_TILE_X3 | YMI_TILE_64x32_K2, 2, 1, 2, 1, 1, 2, false, 1)    \
  X(YMI_TILE_X3 | YMI_TILE_32x64_K2, 1, 2, 2, 1, 1, 2, false, 1)    \
  X(YMI_TILE_X3 | YMI_TILE_64x64_S3, 2, 2, 1, 1, 1, 3, false, 1)    \
  X(YMI_TILE_X3 | YMI_TILE_64x128_S3, 2, 2, 1, 1, 2, 3, false, 1)   \
  X(YMI_TILE_X3 | YMI_TILE_128x64_S3, 2, 2, 1, 2, 1, 3, false, 1)   \
  X(YMI_TILE_X3 | YMI_TILE_128x128_W8, 4, 2, 1, 1, 2, 2, false, 1)  \
  X(YMI_TILE_X3 | YMI_TILE_256x128_W8, 4, 2, 1, 2, 2, 2, false, 1)  \
  X(YMI_TILE_X3 | YMI_TILE_128x128_S3, 2, 2, 1, 2, 2, 3, false, 1)  \
  X(YMI_TILE_X3 | YMI_TILE_128x128_W8_S3, 4, 2, 1, 1, 2, 3, false, 1) \
  X(YMI_TILE_X3 | YMI_TILE_256x128_W8_S3, 4, 2, 1, 2, 2, 3, false, 1) \
  X(YMI_TILE_X3 | YMI_TILE_128x128_W8_S4, 4, 2, 1, 1, 2, 4, false, 1) \
  X(YMI_TILE_H2 | YMI_TILE_128x128, 2, 2, 1, 2, 2, 2, true, 3)      \
  X(YMI_TILE_H2 | YMI_TILE_128x64, 2, 2, 1, 2, 1, 2, true, 3)       \
  X(YMI_TILE_H2 | YMI_TILE_64x64, 2, 2, 1, 1, 1, 2, true, 3)        \
  X(YMI_TILE_H2 | YMI_TILE_64x128, 2, 2, 1, 1, 2, 2, true, 3)       \
  X(YMI_TILE_H2 | YMI_TILE_32x32_K4, 1, 1, 4, 1, 1, 2, false, 3)    \
  X(YMI_TILE_H2 | YMI_TILE_64x32_K2, 2, 1, 2, 1, 1, 2, false, 3)    \
  X(YMI_TILE_H2 | YMI_TILE_32x64_K2, 1, 2, 2, 1, 1, 2, false, 3)    \
  X(YMI_TILE_H2 | YMI_TILE_64x64_S3, 2, 2, 1, 1, 1, 3, false, 3)    \
  X(YMI_TILE_H2 | YMI_TILE_64x128_S3, 2, 2, 1, 1, 2, 3, false, 3)   \
  X(YMI_TILE_H2 | YMI_TILE_128x64_S3, 2, 2, 1, 2, 1, 3, false, 3)   \
  X(YMI_TILE_H2 | YMI_TILE_128x128_W8, 4, 2, 1, 1, 2, 2, false, 3)  \
  X(YMI_TILE_H2 | YMI_TILE_256x128_W8, 4, 2, 1, 2, 2, 2, false, 3)  \
  X(YMI_TILE_H2 | YMI_TILE_128x256_W8, 4, 2, 1, 1, 4, 2, false, 3)  \
  X(YMI_TILE_H2 | YMI_TILE_128x128_S3, 2, 2, 1, 2, 2, 3, false, 3)  \
  X(YMI_TILE_H2 | YMI_TILE_128x128_W8_S3, 4, 2, 1, 1, 2, 3, false, 3) \
  X(YMI_TILE_H2 | YMI_TILE_256x128_W8_S3, 4, 2, 1, 2, 2, 3, false, 3) \
  X(YMI_TILE_H2 | YMI_TILE_128x128_W8_S4, 4, 2, 1, 1, 2, 4, false, 3) \
  X(YMI_TILE_H2 | YMI_TILE_32x32_K4_S4, 1, 1, 4, 1, 1, 4, false, 3)   \
  X(YMI_TILE_H2 | YMI_TILE_64x32_K2_S3, 2, 1, 2, 1, 1, 3, false, 3)   \
  X(YMI_TILE_H2 | YMI_TILE_32x64_K2_S3, 1, 2, 2, 1, 1, 3, false, 3)

int tile_dims(int tile, int &bm, int &bn) {
  switch (tile) {
#define X(id, wm, wn, wk, tm, tn, ns, all, prec) case id: bm = wm * tm * 32; bn = wn * tn * 32; return 0;
    YMI_TILE_TABLE(X)
#undef X
  }
  return -1;
}

bool tile_all_loaders(int tile) {
  switch (tile) {
#define X(id, wm, wn, wk, tm, tn, ns, all, prec) case id: return all;
    YMI_TILE_TABLE(X)
#undef X
  }
  return false;
}

int pick_tile(const ymi_conv_desc *d) {
  // Fallback heuristic (the Python plan autotunes on the device and overrides this).  Cost model: the GEMMs are
  // MFMA-bound, a CU runs its ceil(tiles/256) blocks `occ` at a time, and the matrix-pipe utilisation u(c) with c
  // co-resident blocks was measured in round 1 (profiles/r01_layers_v1.txt): one block per CU leaves the pipe
  // idle during its LDS-store/barrier/ds_read bubbles (u ~0.45), two or more overlap (u ~0.7-0.76).
  const long M = (long)d->B * d->Ho * d->Wo;
  const int N = d->Cout;
  // few output tiles: quarter the tile and split K across the block's waves so every CU gets work and the serial
  // K chain is 4x shorter (18x18 ... 5x5 maps)
  if (((M + 63) / 64) * ((N + 63) / 64) < 512) return YMI_TILE_32x32_K4;
  if (N <= 32) return YMI_TILE_128x32;
  const int cand[4] = {YMI_TILE_128x128, YMI_TILE_128x64, YMI_TILE_64x128, YMI_TILE_64x64};
  const int occ[4] = {2, 2, 2, 4};
  const double u[4][4] = {{0.50, 0.76, 0.76, 0.76}, {0.46, 0.72, 0.72, 0.72}, {0.46, 0.72, 0.72, 0.72},
                          {0.41, 0.69, 0.66, 0.64}};
  int best = YMI_TILE_64x64;
  double best_cost = 1e300;
  for (int i = 0; i < 4; ++i) {
    int bm, bn; tile_dims(cand[i], bm, bn);
    if (bn > 64 && N <= 64) continue;  // don't waste half the N tile
    const long tiles = ((M + bm - 1) / bm) * ((N + bn - 1) / bn);
    const long n = (tiles + 255) / 256;           // blocks on the busiest CU
    const long full = n / occ[i], rem = n % occ[i];
    double cost = full * occ[i] / u[i][occ[i] - 1];
    if (rem) cost += rem / u[i][rem - 1];
    cost *= (double)bm * bn;
    if (cost < best_cost) { best_cost = cost; best = cand[i]; }
  }
  return best;
}

int validate(const ymi_conv_desc *d, int loader) {
  if (!d || !d->x || !d->w) return YMI_ENULL;
  if (d->nseg < 1 || d->nseg > 3) return YMI_EARG;
  for (int s = 0; s < d->nseg; ++s) if (!d->seg[s].ptr) return YMI_ENULL;
  if (d->B <= 0 || d->H <= 0 || d->W <= 0 || d->Cout <= 0 || d->Ho <= 0 || d->Wo <= 0) return YMI_EARG;
  if (d->stride <= 0 || d->kh <= 0 || d->kw <= 0 || d->pad < 0 || d->Cin <= 0) return YMI_EARG;   // (a C caller must get
                                                                                 // an error code, never a SIGFPE)
  // output segments: ascending, non-overlapping channel ranges inside [0, Cout]; a row must hold the segment
  for (int s = 0; s < d->nseg; ++s) {
    const ymi_conv_seg &g = d->seg[s];
    if (g.n0 < 0 || g.n1 <= g.n0 || g.n0 >= d->Cout) return YMI_EARG;
    if (s > 0 && g.n0 < d->seg[s - 1].n1) return YMI_EARG;
    const int width = (g.n1 < d->Cout ? g.n1 : d->Cout) - g.n0;
    if (g.row_stride < width || g.batch_stride < 0) return YMI_ESHAPE;
  }
  if (d->res_mode != YMI_RES_NONE && d->res_mode != YMI_RES_ADD && d->res_mode != YMI_RES_BILINEAR) return YMI_EARG;
  if (d->res_mode != YMI_RES_NONE && d->res_ld < d->Cout) return YMI_ESHAPE;
  if (d->res_mode == YMI_RES_BILINEAR && (d->res_H <= 0 || d->res_W <= 0)) return YMI_EARG;
  if (d->ldx % 4 != 0 || d->Kpad % 32 != 0 || d->Kpad < d->kh * d->kw * d->Cin) return YMI_ESHAPE;
  if (loader == 1) { if (d->Cin != 4) return YMI_ESHAPE; }
  else if (d->Cin % 32 != 0) return YMI_ESHAPE;
  if (loader == 2 && (d->kh != 3 || d->kw != 3 || d->pad != 1)) return YMI_ESHAPE;
  if (d->Ho != (d->H + 2 * d->pad - d->kh) / d->stride + 1) return YMI_ESHAPE;
  if (d->Wo != (d->W + 2 * d->pad - d->kw) / d->stride + 1) return YMI_ESHAPE;
  // buffer-resource addressing: activation and filter tensors must each stay below 2 GiB
  if ((long)d->B * d->H * d->W * d->ldx >= (1L << 29)) return YMI_ESHAPE;
  if ((((long)d->Cout + 127) / 128 * 128) * d->Kpad >= (1L << 29)) return YMI_ESHAPE;
  if ((((uintptr_t)d->x) & 15) || (((uintptr_t)d->w) & 15)) return YMI_ESHAPE;
  if (d->res_mode != YMI_RES_NONE && !d->res) return YMI_ENULL;
  return YMI_OK;
}

// split_k > 1 (internal, ymi_conv2d_nhwc_f32 with desc->split_k): the `groups` of the launch are K ranges of ONE pointwise
// GEMM — group g multiplies channels [g*K/S, (g+1)*K/S) and writes its raw partial sums to ws + g*M*Cout (fast-path
// epilogue, no scale / bias / activation); splitk_fixup_k then adds the partials in a fixed order and applies the epilogue.
struct H2Group { const void *a2 = nullptr; unsigned a2_plane = 0; long a2_gs = 0; unsigned sc_gs = 0; };

int run_conv(const ymi_conv_desc *d, int loader, const float *offmask, int ldo, hipStream_t s, int groups = 1, long x_gs = 0,
             long w_gs = 0, long y_gs = 0, double prof_flops = -1.0, int prof_kind = -1, int split_k = 1,
             const H2Group &h2g = H2Group(), int mask_is_prob = 0, int om_layout = 0) {
  int rc = validate(d, loader);
  // grouped launches take the fast-path epilogue only (it applies the group's output offset)
  if (rc == YMI_OK && groups > 1 &&
      (loader != 0 || d->nseg != 1 || groups > 65535 || (d->Cout & 3) || d->res_mode != YMI_RES_NONE ||
       d->seg[0].act > YMI_ACT_LEAKY01 || d->seg[0].n0 != 0 || (d->seg[0].row_stride & 3) ||
       d->seg[0].batch_stride != (int64_t)d->Ho * d->Wo * d->seg[0].row_stride))
    rc = YMI_EARG;
  if (rc) return rc;
  KParams kp;
  kp.d = *d;
  kp.HoWo = d->Ho * d->Wo;
  kp.M = d->B * kp.HoWo;
  kp.nk = d->Kpad / BK / split_k;
  kp.tiles_n = 0;
  kp.x_bytes = (unsigned)((size_t)d->B * d->H * d->W * d->ldx * sizeof(float));
  {
    const long cout_pad = ((long)d->Cout + 127) / 128 * 128;   // packing contract: CoutPad % 128 == 0
    kp.w_bytes = (unsigned)(cout_pad * d->Kpad * (long)sizeof(float));
  }
  kp.offmask = offmask;
  kp.ldo = ldo;
  kp.mask_is_prob = mask_is_prob;
  kp.om_layout = om_layout;
  kp.x_gs = x_gs; kp.w_gs = w_gs; kp.y_gs = y_gs;
  const bool h2 = (d->tile & YMI_TILE_H2) != 0;
  if (h2 && (d->tile & YMI_TILE_X3)) return YMI_EARG;
  // fp16x2 needs the filter planes, the folded per-row scales and a magnitude bound for the activations (or pre-split ones)
  if (h2 && (!d->w_h2 || !d->scale_h2 || (!d->x_amax && !h2g.a2))) return YMI_ENULL;
  kp.w3 = h2 ? d->w_h2 : d->w_x3;
  kp.w3_plane = (unsigned)((((long)d->Cout + 127) / 128 * 128) * d->Kpad * 2L);
  kp.w3_gs = split_k > 1 ? (unsigned)(d->Kpad / split_k * 2) : (h2 ? 2u : 3u) * kp.w3_plane;
  if (kp.w3 && (((uintptr_t)kp.w3) & 15)) return YMI_ESHAPE;
  kp.a2 = h2 ? h2g.a2 : nullptr; kp.a2_plane = h2g.a2_plane; kp.a2_gs = h2g.a2_gs; kp.sc_gs = h2g.sc_gs;
  if (kp.a2 && ((((uintptr_t)kp.a2) & 15) || (d->ldx & 7))) return YMI_ESHAPE;
  kp.abl = 0;
#ifdef YMI_DIAGNOSTICS   // `make DIAG=1`: ablation switches for tools/conv_probe.py — they produce WRONG results by design
  { const char *e = getenv("YMI_ABLATE"); kp.abl = e ? atoi(e) : 0; }
#endif
  kp.trace = g_trace;
  int tile = d->tile ? d->tile : pick_tile(d);
  if (loader != 0 && (!tile_all_loaders(tile) || (loader == 2 && (tile & YMI_TILE_X3)))) {
    if (d->tile) return YMI_EARG;   // explicit request the stem / DCN loaders cannot honour
    tile = YMI_TILE_64x64;
  }
  ProfRec *pr = nullptr;
  std::unique_lock<std::mutex> prof_lock(g_prof_mu, std::defer_lock);
  if (g_prof_on && prof_kind != -2) {       // (-2: the caller brackets several launches with its own record)
    prof_lock.lock();            // held across the launch so e0 / launch / e1 of one record stay together on the stream
    if (g_prof_n < PROF_MAX) {
      pr = &g_prof[g_prof_n];
      if (g_prof_n >= g_prof_alloc) { hipEventCreate(&pr->e0); hipEventCreate(&pr->e1); g_prof_alloc = g_prof_n + 1; }
      // record kind: 0 / 1 / 2 = the loader of a direct launch, 7 = a direct 1x1 launch on the pointwise loader (template
      // LOADER 3); 3 .. 6 are the Winograd records of csrc/winograd.hip
      const bool pw = loader == 0 && d->kh == 1 && d->kw == 1 && d->pad == 0;
      pr->flops = prof_flops >= 0 ? prof_flops : ymi_conv_flops(d); pr->tile = tile; pr->kind = prof_kind >= 0 ? prof_kind : (pw ? 7 : loader);
      hipEventRecord(pr->e0, s);
    }
  }
  switch (tile) {
#define X(id, wm, wn, wk, tm, tn, ns, all, prec) case id: rc = launch_prec<wm, wn, wk, tm, tn, ns, all, prec>(kp, loader, s, groups); break;
    YMI_TILE_TABLE(X)
#undef X
    default: return YMI_EARG;
  }
  if (pr) { hipEventRecord(pr->e1, s); ++g_prof_n; }
  return rc;
}

// ---- split-K second pass: y = act(scale * sum_g partial[g] + bias (+ res)) ------------------------------------------
// One float4 of output channels per thread; partials are added in the order g = 0 .. S-1 (bit-reproducible).
__global__ __launch_bounds__(256) void splitk_fixup_k(const float *__restrict__ part, long gstride, int S, long M, int N4, int ldy,
                                                      float *__restrict__ y, const float *__restrict__ scale,
                                                      const float *__restrict__ bias, const float *__restrict__ res, int res_ld,
                                                      int act, int res_after_act, float *__restrict__ y_amax) {
  const long total = M * N4;
  float am = 0.f;
  const ymi_amax_pre apre = ymi_amax_prefetch(y_amax);
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
    const long m = i / N4;
    const int n = (int)(i - m * N4) * 4;
    f32x4 v = *reinterpret_cast<const f32x4 *>(part + m * (N4 * 4L) + n);
    for (int g = 1; g < S; ++g) v += *reinterpret_cast<const f32x4 *>(part + g * gstride + m * (N4 * 4L) + n);
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, bi = {0.f, 0.f, 0.f, 0.f}, rv = {0.f, 0.f, 0.f, 0.f};
    if (scale) sc = *reinterpret_cast<const f32x4 *>(scale + n);
    if (bias) bi = *reinterpret_cast<const f32x4 *>(bias + n);
    if (res) rv = *reinterpret_cast<const f32x4 *>(res + m * res_ld + n);
    v = v * sc + bi;
    const float slope = act == YMI_ACT_RELU ? 0.f : (act == YMI_ACT_LEAKY01 ? 0.1f : 1.f);
    if (!res_after_act) v += rv;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], slope * v[e]);
    if (res_after_act) v += rv;
    am = fmaxf(am, ymi_absmax4(v));
    *reinterpret_cast<f32x4 *>(y + m * ldy + n) = v;
  }
  if (y_amax) ymi_amax_finish(apre, am);
}

int ymi_internal_prof_begin_fwd(double flops, int tile, int kind, hipStream_t s);
void ymi_internal_prof_end_fwd(int idx, hipStream_t s);

int run_splitk(const ymi_conv_desc *d, hipStream_t s) {
  const int S = d->split_k;
  int rc = validate(d, 0);
  if (rc) return rc;
  const ymi_conv_seg &g0 = d->seg[0];
  const long HoWo = (long)d->Ho * d->Wo, M = (long)d->B * HoWo;
  if (d->kh != 1 || d->kw != 1 || d->pad != 0 || d->Cin != d->Kpad || S < 2 || S > 16 || (d->Kpad / BK) % S != 0) return YMI_EARG;
  if (d->nseg != 1 || g0.n0 != 0 || g0.n1 < d->Cout || (d->Cout & 3) || (g0.row_stride & 3) || (((uintptr_t)g0.ptr) & 15) ||
      g0.batch_stride != HoWo * g0.row_stride || g0.act > YMI_ACT_LEAKY01 || g0.act < 0)
    return YMI_ESHAPE;
  if (d->res_mode != YMI_RES_NONE && (d->res_mode != YMI_RES_ADD || (d->res_ld & 3) || (((uintptr_t)d->res) & 15))) return YMI_ESHAPE;
  if ((((uintptr_t)d->scale) | ((uintptr_t)d->bias)) & 15) return YMI_ESHAPE;
  if (!d->split_ws) return YMI_ENULL;
  if (((uintptr_t)d->split_ws) & 15) return YMI_ESHAPE;
  if (M * d->Cout >= (1L << 29)) return YMI_ESHAPE;
  ymi_conv_desc pd = *d;
  pd.split_k = 0;
  pd.scale = nullptr; pd.bias = nullptr; pd.res = nullptr; pd.res_mode = YMI_RES_NONE; pd.res_after_act = 0;
  pd.scale_h2 = d->winv_h2;            // fp16x2: the partial launches undo the operand scales only -> true partial sums
  pd.y_amax = nullptr;                 // (the second pass reports the magnitude of the finished tensor)
  if ((d->tile & YMI_TILE_H2) && !d->winv_h2) return YMI_ENULL;
  pd.seg[0].n0 = 0; pd.seg[0].n1 = d->Cout; pd.seg[0].act = YMI_ACT_NONE; pd.seg[0].row_stride = d->Cout;
  pd.seg[0].batch_stride = HoWo * d->Cout; pd.seg[0].ptr = d->split_ws;
  const int outer = ymi_internal_prof_begin_fwd(ymi_conv_flops(d), d->tile ? d->tile : pick_tile(d), 7, s);
  const long ksub = d->Kpad / S;
  rc = run_conv(&pd, 0, nullptr, 0, s, S, ksub, ksub, M * d->Cout, -1.0, -2, S);
  if (rc) return rc;
  const long total = M * (d->Cout / 4);
  long gsz = (total + 255) / 256;
  const long cap = 256L * 32;
  hipLaunchKernelGGL(splitk_fixup_k, dim3((unsigned)(gsz > cap ? cap : gsz)), dim3(256), 0, s, d->split_ws, M * (long)d->Cout, S, M,
                     d->Cout / 4, g0.row_stride, g0.ptr, d->scale, d->bias, d->res_mode == YMI_RES_ADD ? d->res : nullptr,
                     d->res_ld, g0.act, d->res_after_act, d->y_amax);
  rc = ymi_launch_status();
  ymi_internal_prof_end_fwd(outer, s);
  return rc;
}

}  // namespace

// internal (not part of the C ABI): profiling brackets for composite ops (csrc/winograd.hip): a record that spans
// several launches.  Returns the record index or -1 when profiling is off.
int ymi_internal_prof_begin(double flops, int tile, int kind, hipStream_t s) {
  if (!g_prof_on) return -1;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (g_prof_n >= PROF_MAX) return -1;
  ProfRec *pr = &g_prof[g_prof_n];
  if (g_prof_n >= g_prof_alloc) { hipEventCreate(&pr->e0); hipEventCreate(&pr->e1); g_prof_alloc = g_prof_n + 1; }
  pr->flops = flops; pr->tile = tile; pr->kind = kind;
  hipEventRecord(pr->e0, s);
  return g_prof_n++;
}
void ymi_internal_prof_end(int idx, hipStream_t s) {
  if (idx >= 0) hipEventRecord(g_prof[idx].e1, s);
}
namespace {
int ymi_internal_prof_begin_fwd(double flops, int tile, int kind, hipStream_t s) { return ymi_internal_prof_begin(flops, tile, kind, s); }
void ymi_internal_prof_end_fwd(int idx, hipStream_t s) { ymi_internal_prof_end(idx, s); }
}  // namespace

// internal (not part of the C ABI): grouped GEMM for csrc/winograd.hip — `groups` independent 1x1 GEMMs that share the
// descriptor's shape; group g reads x + g*x_gs, w + g*w_gs and writes seg[0].ptr + g*y_gs.
int ymi_internal_grouped_gemm(const ymi_conv_desc *d, int groups, long x_gs, long w_gs, long y_gs, double prof_flops,
                              int prof_kind, hipStream_t s, const void *a2, unsigned a2_plane, long a2_gs, unsigned sc_gs) {
  H2Group g;
  g.a2 = a2; g.a2_plane = a2_plane; g.a2_gs = a2_gs; g.sc_gs = sc_gs;
  return run_conv(d, 0, nullptr, 0, s, groups, x_gs, w_gs, y_gs, prof_flops, prof_kind, 1, g);
}

int ymi_internal_dcn_h2(const ymi_dcn_desc *dd, int base_tile, hipStream_t s);   // csrc/dcn.hip (C++ linkage: internal)

int ymi_internal_pipe_conv(const ymi_conv_desc *d, int base_tile, hipStream_t s); // csrc/dcn.hip: the same pipeline, ordinary convolution
int ymi_internal_ws_conv(const ymi_conv_desc *d, int base_tile, hipStream_t s);   // csrc/wstat.hip: weight-stationary streaming kernel (Cout <= 64)
int ymi_internal_patch_conv(const ymi_conv_desc *d, hipStream_t s);               // csrc/patch.hip: 3x3 64 -> 64 from an LDS-resident input patch
int ymi_internal_pc_conv(const ymi_conv_desc *d, int base_tile, hipStream_t s);   // csrc/pcconv.hip: producer / consumer waves
int ymi_internal_patch2_conv(const ymi_conv_desc *d, int base_tile, hipStream_t s);   // csrc/patch2.hip: 3x3, input patch in LDS, filters streamed

// internal (csrc/dcn.hip): the second pass of a split-K launch for a dense [M, Cout] output (Cout % 4 == 0, 16-byte aligned rows)
int ymi_internal_splitk_fixup(const float *part, long gstride, int S, long M, int Cout, int ldy, float *y, const float *scale,
                              const float *bias, const float *res, int res_ld, int act, int res_after_act, float *y_amax,
                              hipStream_t s) {
  const long total = M * (Cout / 4);
  long gsz = (total + 255) / 256;
  const long cap = 256L * 32;
  hipLaunchKernelGGL(splitk_fixup_k, dim3((unsigned)(gsz > cap ? cap : gsz)), dim3(256), 0, s, part, gstride, S, M, Cout / 4, ldy, y,
                     scale, bias, res, res_ld, act, res_after_act, y_amax);
  return ymi_launch_status();
}

extern "C" {

double ymi_conv_flops(const ymi_conv_desc *d) {
  return 2.0 * d->B * d->Ho * d->Wo * (double)(d->cout_alg > 0 ? d->cout_alg : d->Cout) * d->kh * d->kw *
         (double)(d->cin_alg > 0 ? d->cin_alg : d->Cin);
}

int ymi_conv_pick_tile(const ymi_conv_desc *d) { return d ? pick_tile(d) : YMI_ENULL; }

int ymi_conv2d_nhwc_f32(const ymi_conv_desc *d, void *stream) {
  if (!d) return YMI_ENULL;
  if (d->tile & YMI_TILE_DCNP) {             // the pipelined kernel of csrc/dcn.hip as an ordinary convolution (fp16x2 only)
    if (!(d->tile & YMI_TILE_H2) || (d->tile & YMI_TILE_X3)) return YMI_EARG;
    const int rc = validate(d, 0);
    if (rc) return rc;
    if ((d->tile & 31) == YMI_DCNP_PATCH_C64) return ymi_internal_patch_conv(d, (hipStream_t)stream);
    if ((d->tile & 31) >= YMI_DCNP_PATCH2_256) return ymi_internal_patch2_conv(d, d->tile & 31, (hipStream_t)stream);
    if ((d->tile & 31) == YMI_DCNP_PC_128x128) return ymi_internal_pc_conv(d, d->tile & 31, (hipStream_t)stream);
    if ((d->tile & 31) >= YMI_DCNP_WS_128x32_W4) return ymi_internal_ws_conv(d, d->tile & 31, (hipStream_t)stream);
    return ymi_internal_pipe_conv(d, d->tile & 31, (hipStream_t)stream);
  }
  if (d->split_k > 1) return run_splitk(d, (hipStream_t)stream);
  return run_conv(d, d->Cin == 4 ? 1 : 0, nullptr, 0, (hipStream_t)stream);
}

int ymi_dcn_v2_forward_f32(const ymi_dcn_desc *d, void *stream) {
  if (!d || !d->offmask) return YMI_ENULL;
  if (d->ldo < 27) return YMI_ESHAPE;
  if (d->om_layout != 0 && d->om_layout != 1) return YMI_EARG;
  if (d->conv.tile & YMI_TILE_DCNP) {        // the pipelined gather-GEMM (fp16x2 only); an explicit request it cannot honour fails
    if (!(d->conv.tile & YMI_TILE_H2) || (d->conv.tile & YMI_TILE_X3)) return YMI_EARG;
    const int rc = validate(&d->conv, 2);
    if (rc) return rc;
    return ymi_internal_dcn_h2(d, d->conv.tile & 31, (hipStream_t)stream);
  }
  return run_conv(&d->conv, 2, d->offmask, d->ldo, (hipStream_t)stream, 1, 0, 0, 0, -1.0, -1, 1, H2Group(), d->mask_is_prob, d->om_layout);
}

int ymi_debug_set_trace(void *buf, long cap_blocks) {
  g_trace = (unsigned long long *)buf;
  g_trace_cap = cap_blocks;
  return YMI_OK;
}

int ymi_prof_enable(int on) { g_prof_on = on; return YMI_OK; }
int ymi_prof_count(void) { return g_prof_n; }
int ymi_prof_reset(void) { std::lock_guard<std::mutex> lk(g_prof_mu); g_prof_n = 0; return YMI_OK; }
int ymi_prof_read(int i, float *ms, double *flops, int32_t *tile, int32_t *kind) {
  if (i < 0 || i >= g_prof_n) return YMI_EARG;
  hipError_t e = hipEventSynchronize(g_prof[i].e1);
  if (e != hipSuccess) return (int)e;
  float t = 0.f;
  e = hipEventElapsedTime(&t, g_prof[i].e0, g_prof[i].e1);
  if (e != hipSuccess) return (int)e;
  if (ms) *ms = t;
  if (flops) *flops = g_prof[i].flops;
  if (tile) *tile = g_prof[i].tile;
  if (kind) *kind = g_prof[i].kind;
  return YMI_OK;
}

}  // extern "C"
