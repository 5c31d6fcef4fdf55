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
#include "../../include/yolact_amd.h"

namespace {

constexpr int BK = 32;
constexpr unsigned OOB = 0x80000000u;  // buffer offset >= num_records (< 2^31, validated) -> the load returns zeros

typedef __attribute__((address_space(3))) void *lds_ptr_t;

struct KParams {
  ymi_conv_desc d;
  int M, HoWo, tiles_n, nk;
  unsigned x_bytes, w_bytes;      // buffer-resource sizes
  const float *offmask;           // DCN only
  int ldo;
  int abl;                        // diagnostics only (env YMI_ABLATE): bit0 skip the staging of chunks > 0,
                                  // bit2 skip barriers in the K loop — wrong results, used to attribute stall time
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

// LOADER: 0 = Cin % 32 == 0 (a K chunk lies inside one filter tap; tap is block-uniform)
//         1 = Cin == 4 (stem; a K chunk = 8 taps x 4 channels; tap is per-lane)
//         2 = DCNv2 modulated deformable gather (Cin % 32 == 0, 3x3, pad 1): A through registers, B by DMA (WK == 1)
// WK:     waves along K.  WM*WN*WK == 4.  With WK > 1 a pipeline stage holds WK consecutive 32-deep chunks and wave
//         (wm, wn, wk) multiplies chunk wk of every stage; the WK partial tiles are summed (fixed order) in the
//         epilogue's LDS tile.  This quarters the block tile (32x32 with WK = 4) without an inter-block reduction:
//         4x the blocks and 1/4 of the serial K chain for the small-M layers (18x18 ... 5x5 maps) that otherwise
//         leave most CUs idle behind one long K loop.
template <int WM, int WN, int WK, int TM, int TN, int LOADER>
__global__ __launch_bounds__(256) void conv_igemm_f32(const KParams p) {
#if defined(__HIP_DEVICE_COMPILE__)   // host pass: empty body.  hipcc (ROCm 7.2) silently drops the host launch stub of a
                                      // templated kernel whose body uses the buffer-resource LDS-DMA builtins.
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int RA = BM / 32, RB = BN / 32;  // 8-row DMA pieces per wave per chunk for the A / B tile
  constexpr int SUB = (BM + BN) * BK;        // floats per chunk image [BM + BN rows][32]
  constexpr int STAGE = SUB * WK;            // floats per pipeline stage
  constexpr int ELD = BN + 4;                // epilogue tile row stride
  constexpr int LDS_FLOATS = (2 * STAGE > WK * BM * ELD) ? 2 * STAGE : WK * BM * ELD;
  static_assert(WM * WN * WK == 4, "4 waves per block");
  static_assert(LOADER != 2 || WK == 1, "DCN gather runs without the K split");
  __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];

  const ymi_conv_desc &d = p.d;
  const int t = threadIdx.x;
  const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);   // wave-uniform -> SGPR (LDS-DMA bases)
  const int wk = wave % WK, wmn = wave / WK;
  const int wm = wmn / WN, wn = wmn % WN;
  const int kq = t & 7, r0 = t >> 3;          // DMA: physical 16-byte slot and row (within a 32-row group) of this lane
  const int sl = kq ^ ((r0 >> 1) & 7);        // logical k-slot that lives at this lane's physical position

  const int logical = ymi_xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = logical % p.tiles_n, tile_m = logical / p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)d.x, 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void *)d.w, 0, (int)p.w_bytes, 0x00020000);

  // ---- epilogue thread mapping + residual prefetch -----------------------------------------------
  // Each thread owns 4 consecutive output channels of RPT rows.  A plain residual (bottleneck shortcut) is fetched
  // NOW, so its HBM latency overlaps the whole K loop instead of serialising behind it in the epilogue (the
  // K <= 128 1x1 layers at 138x138 are HBM-bound: 2.3 TB/s before this, profiles/r01_*).
  constexpr int C4 = BN / 4;        // float4 columns per tile row
  constexpr int RSTEP = 256 / C4;   // rows covered by one pass of the block
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

  // ---- per-thread A-row bookkeeping (fixed over the K loop) ---------------------------------
  int a_iy0[RA], a_ix0[RA], a_base[RA];        // a_base: byte offset of (pixel, channel 4*sl) for tap (0,0)
#pragma unroll
  for (int i = 0; i < RA; ++i) {
    const int m = m0 + r0 + 32 * i;
    if (m < p.M) {
      const int b = m / p.HoWo, pix = m - b * p.HoWo;
      const int oy = pix / d.Wo, ox = pix - oy * d.Wo;
      a_iy0[i] = oy * d.stride - d.pad;
      a_ix0[i] = ox * d.stride - d.pad;
      a_base[i] = (((b * d.H + a_iy0[i]) * d.W + a_ix0[i]) * d.ldx + 4 * sl) * 4;
      if (LOADER == 2) a_base[i] = m;  // DCN: remember the output pixel, geometry recomputed per tap
    } else {
      a_iy0[i] = -(1 << 28);  // never valid
      a_ix0[i] = -(1 << 28);
      a_base[i] = 0;
    }
  }
  unsigned b_off[RB];                           // byte offset of (filter row, k-slot sl) for chunk 0
#pragma unroll
  for (int i = 0; i < RB; ++i) b_off[i] = (unsigned)(((n0 + r0 + 32 * i) * d.Kpad + 4 * sl) * 4);

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

  // stage step `st` (chunks st*WK .. st*WK + WK - 1) into LDS stage `buf`
  auto issue_tile = [&](int st, int buf) {
#pragma unroll
    for (int j = 0; j < WK; ++j) {
      const int kc = st * WK + j;
      if (WK > 1 && kc >= p.nk) break;      // ragged last step: that chunk slot is simply not multiplied
      float *As = lds + buf * STAGE + j * SUB;
      float *Bs = As + BM * BK;
      if (LOADER == 0) {
        const int koff = ((nx_ky[j] * d.W + nx_kx[j]) * d.ldx + nx_c[j]) * 4;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
          const bool ok = (unsigned)(a_iy0[i] + nx_ky[j]) < (unsigned)d.H && (unsigned)(a_ix0[i] + nx_kx[j]) < (unsigned)d.W;
          const unsigned voff = ok ? (unsigned)(a_base[i] + koff) : OOB;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_ptr_t)(As + (wave * 8 + 32 * i) * BK), 16, voff, 0, 0, 0);
        }
      } else if (LOADER == 1) {
        const int tap = kc * 8 + sl;
        const int ky = tap / d.kw, kx = tap - ky * d.kw;
        const bool tap_ok = tap < d.kh * d.kw;
        const int koff = ((ky * d.W + kx) * d.ldx - 4 * sl) * 4;   // a_base carries +4*sl channels; Cin == 4 -> channel 0
#pragma unroll
        for (int i = 0; i < RA; ++i) {
          const bool ok = tap_ok && (unsigned)(a_iy0[i] + ky) < (unsigned)d.H && (unsigned)(a_ix0[i] + kx) < (unsigned)d.W;
          const unsigned voff = ok ? (unsigned)(a_base[i] + koff) : OOB;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_ptr_t)(As + (wave * 8 + 32 * i) * BK), 16, voff, 0, 0, 0);
        }
      } else {
        // DCNv2 (dcn_v2_im2col_cuda.cu:143-193): sample point = (oy*s - p + ky + dh, ox*s - p + kx + dw),
        // zero unless -1 < h < H and -1 < w < W; zero-padded bilinear; times sigmoid(mask logit).
        // Thread (kq, r0) produces LOGICAL slot kq of its rows and stores it at the swizzled position.
        const int tap = nx_ky[j] * 3 + nx_kx[j], c = nx_c[j] + 4 * kq;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          if (a_iy0[i] > -(1 << 27)) {
            const int m = a_base[i];
            const float *om = p.offmask + (size_t)m * p.ldo;
            const float dh = om[2 * tap], dw = om[2 * tap + 1];
            const float mk = 1.f / (1.f + expf(-om[18 + tap]));
            const float h = (float)(a_iy0[i] + nx_ky[j]) + dh, w = (float)(a_ix0[i] + nx_kx[j]) + dw;
            if (h > -1.f && w > -1.f && h < (float)d.H && w < (float)d.W) {
              const int hl = (int)floorf(h), wl = (int)floorf(w);
              const int hh = hl + 1, wh = wl + 1;
              const float lh = h - (float)hl, lw = w - (float)wl, uh = 1.f - lh, uw = 1.f - lw;
              const int b = m / p.HoWo;
              const float *img = d.x + (size_t)b * d.H * d.W * d.ldx + c;
              f32x4 v1 = {0.f, 0.f, 0.f, 0.f}, v2 = v1, v3 = v1, v4 = v1;
              if (hl >= 0 && wl >= 0) v1 = *reinterpret_cast<const f32x4 *>(img + (hl * d.W + wl) * d.ldx);
              if (hl >= 0 && wh <= d.W - 1) v2 = *reinterpret_cast<const f32x4 *>(img + (hl * d.W + wh) * d.ldx);
              if (hh <= d.H - 1 && wl >= 0) v3 = *reinterpret_cast<const f32x4 *>(img + (hh * d.W + wl) * d.ldx);
              if (hh <= d.H - 1 && wh <= d.W - 1) v4 = *reinterpret_cast<const f32x4 *>(img + (hh * d.W + wh) * d.ldx);
              const float w1 = uh * uw, w2 = uh * lw, w3 = lh * uw, w4 = lh * lw;
              v = (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4) * mk;
            }
          }
          ra[i] = v;
        }
      }
#pragma unroll
      for (int i = 0; i < RB; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_ptr_t)(Bs + (wave * 8 + 32 * i) * BK), 16, b_off[i], kc * (BK * 4), 0, 0);
      if (LOADER != 1) {  // this slot's next chunk is WK chunks further
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
      for (int i = 0; i < RA; ++i) *reinterpret_cast<f32x4 *>(As + (r0 + 32 * i) * BK + 4 * sl) = ra[i];
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
  auto compute = [&](int buf) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int slot = g & 1;
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[slot][i][s], fb[slot][j][s], acc[i][j], 0, 0, 0);
      if (g + 2 < 4) {
        __builtin_amdgcn_sched_barrier(0);   // keep the prefetch right behind the MFMAs that free its registers
        load_frag(buf, g + 2, slot);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  // ---- main loop ---------------------------------------------------------------------------
  const int nsteps = (p.nk + WK - 1) / WK;
  issue_tile(0, 0);
  store_a_regs(0);
  __syncthreads();   // drains the DMA (vmcnt(0)) then barrier
  for (int st = 0; st < nsteps; ++st) {
    const int cur = st & 1;
    const bool more = (st + 1) < nsteps;
    const bool mine = (WK == 1) || (st * WK + wk < p.nk);   // wave-uniform: does this wave's chunk exist?
    if (mine) {
      load_frag(cur, 0, 0);
      load_frag(cur, 1, 1);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (more && !(p.abl & 1)) issue_tile(st + 1, cur ^ 1);
    __builtin_amdgcn_sched_barrier(0);
    if (mine) compute(cur);
    if (more) store_a_regs(cur ^ 1);
    if (!(p.abl & 4)) __syncthreads();
  }

  // ---- epilogue ----------------------------------------------------------------------------
  // Accumulators -> LDS tile [WK][BM][BN+4] -> each thread owns 4 consecutive output channels of a row, so
  // residual loads and output stores are 16 bytes per lane and 16*C4 contiguous bytes per row
  // (the first version stored one dword per lane per accumulator register and was store-issue bound on
  // the K <= 128 1x1 layers: 18-35 TF/s; see profiles/).
  // C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8*(r >> 2) + 4*(lane >> 5).
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

  if (n >= d.Cout) return;
  f32x4 sc = {1.f, 1.f, 1.f, 1.f}, bi = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (n + e < d.Cout) {
      if (d.scale) sc[e] = d.scale[n + e];
      if (d.bias) bi[e] = d.bias[n + e];
    }
  }
  // segment of each of the 4 channels; vector store only if all 4 live in one aligned segment
  int e_seg[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    e_seg[e] = -1;
#pragma unroll
    for (int s = 0; s < 3; ++s)
      if (s < d.nseg && n + e >= d.seg[s].n0 && n + e < d.seg[s].n1 && n + e < d.Cout) e_seg[e] = s;
  }
  bool vec_out = e_seg[0] >= 0 && e_seg[0] == e_seg[1] && e_seg[0] == e_seg[2] && e_seg[0] == e_seg[3];
  if (vec_out) {
    const ymi_conv_seg &sg = d.seg[e_seg[0]];
    vec_out = ((n - sg.n0) & 3) == 0 && (sg.row_stride & 3) == 0 && (sg.batch_stride & 3) == 0 &&
              (((uintptr_t)sg.ptr) & 15) == 0;
  }
  float rscale_h = 0.f, rscale_w = 0.f;
  if (d.res_mode == YMI_RES_BILINEAR) {
    rscale_h = (float)d.res_H / (float)d.Ho;
    rscale_w = (float)d.res_W / (float)d.Wo;
  }
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const int row = rbase + RSTEP * i;
    const int m = m0 + row;
    if (m >= p.M) continue;
    const int b = m / p.HoWo, pix = m - b * p.HoWo;
    f32x4 v = *reinterpret_cast<const f32x4 *>(es + row * ELD + 4 * c4);
#pragma unroll
    for (int q = 1; q < WK; ++q) v += *reinterpret_cast<const f32x4 *>(es + q * (BM * ELD) + row * ELD + 4 * c4);
    v = v * sc + bi;
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
    if (vec_out) {
      const ymi_conv_seg &sg = d.seg[e_seg[0]];
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        o[e] = d.res_after_act ? act_apply(v[e], sg.act) + rv[e] : act_apply(v[e] + rv[e], sg.act);
      *reinterpret_cast<f32x4 *>(sg.ptr + (size_t)b * sg.batch_stride + (size_t)pix * sg.row_stride + (n - sg.n0)) = o;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (e_seg[e] < 0) continue;
        const ymi_conv_seg &sg = d.seg[e_seg[e]];
        const float o = d.res_after_act ? act_apply(v[e], sg.act) + rv[e] : act_apply(v[e] + rv[e], sg.act);
        sg.ptr[(size_t)b * sg.batch_stride + (size_t)pix * sg.row_stride + (n + e - sg.n0)] = o;
      }
    }
  }
#endif  // __HIP_DEVICE_COMPILE__
}

// ---- host side --------------------------------------------------------------------------------
struct ProfRec { hipEvent_t e0, e1; double flops; int tile; int kind; };
constexpr int PROF_MAX = 4096;
ProfRec g_prof[PROF_MAX];
int g_prof_n = 0, g_prof_alloc = 0, g_prof_on = 0;

template <int WM, int WN, int WK, int TM, int TN>
int launch_cfg(const KParams &kp, int loader, hipStream_t s) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  KParams p = kp;
  const int tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.d.Cout + BN - 1) / BN;
  const int grid = tiles_m * p.tiles_n;
  if (loader == 0) hipLaunchKernelGGL((conv_igemm_f32<WM, WN, WK, TM, TN, 0>), dim3(grid), dim3(256), 0, s, p);
  else if (loader == 1) hipLaunchKernelGGL((conv_igemm_f32<WM, WN, WK, TM, TN, 1>), dim3(grid), dim3(256), 0, s, p);
  else if constexpr (WK == 1) hipLaunchKernelGGL((conv_igemm_f32<WM, WN, WK, TM, TN, 2>), dim3(grid), dim3(256), 0, s, p);
  else return YMI_EARG;   // the DCN gather has no K-split variant
  return ymi_launch_status();
}

int tile_dims(int tile, int &bm, int &bn) {
  switch (tile) {
    case YMI_TILE_128x128: bm = 128; bn = 128; return 0;
    case YMI_TILE_128x64: bm = 128; bn = 64; return 0;
    case YMI_TILE_64x64: bm = 64; bn = 64; return 0;
    case YMI_TILE_128x32: bm = 128; bn = 32; return 0;
    case YMI_TILE_64x128: bm = 64; bn = 128; return 0;
    case YMI_TILE_32x32_K4: bm = 32; bn = 32; return 0;
    case YMI_TILE_64x32_K2: bm = 64; bn = 32; return 0;
    case YMI_TILE_32x64_K2: bm = 32; bn = 64; return 0;
  }
  return -1;
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

int run_conv(const ymi_conv_desc *d, int loader, const float *offmask, int ldo, hipStream_t s) {
  int rc = validate(d, loader);
  if (rc) return rc;
  KParams kp;
  kp.d = *d;
  kp.HoWo = d->Ho * d->Wo;
  kp.M = d->B * kp.HoWo;
  kp.nk = d->Kpad / BK;
  kp.tiles_n = 0;
  kp.x_bytes = (unsigned)((size_t)d->B * d->H * d->W * d->ldx * sizeof(float));
  {
    const long cout_pad = ((long)d->Cout + 127) / 128 * 128;   // packing contract: CoutPad % 128 == 0
    kp.w_bytes = (unsigned)(cout_pad * d->Kpad * (long)sizeof(float));
  }
  kp.offmask = offmask;
  kp.ldo = ldo;
  { const char *e = getenv("YMI_ABLATE"); kp.abl = e ? atoi(e) : 0; }
  int tile = d->tile ? d->tile : pick_tile(d);
  if (loader == 2 && (tile == YMI_TILE_32x32_K4 || tile == YMI_TILE_64x32_K2 || tile == YMI_TILE_32x64_K2)) {
    if (d->tile) return YMI_EARG;   // explicit request the DCN gather cannot honour
    tile = YMI_TILE_64x64;
  }
  ProfRec *pr = nullptr;
  if (g_prof_on && g_prof_n < PROF_MAX) {
    pr = &g_prof[g_prof_n];
    if (g_prof_n >= g_prof_alloc) { hipEventCreate(&pr->e0); hipEventCreate(&pr->e1); g_prof_alloc = g_prof_n + 1; }
    pr->flops = ymi_conv_flops(d); pr->tile = tile; pr->kind = loader;
    hipEventRecord(pr->e0, s);
  }
  switch (tile) {
    case YMI_TILE_128x128: rc = launch_cfg<2, 2, 1, 2, 2>(kp, loader, s); break;
    case YMI_TILE_128x64: rc = launch_cfg<2, 2, 1, 2, 1>(kp, loader, s); break;
    case YMI_TILE_64x128: rc = launch_cfg<2, 2, 1, 1, 2>(kp, loader, s); break;
    case YMI_TILE_64x64: rc = launch_cfg<2, 2, 1, 1, 1>(kp, loader, s); break;
    case YMI_TILE_128x32: rc = launch_cfg<4, 1, 1, 1, 1>(kp, loader, s); break;
    case YMI_TILE_32x32_K4: rc = launch_cfg<1, 1, 4, 1, 1>(kp, loader, s); break;
    case YMI_TILE_64x32_K2: rc = launch_cfg<2, 1, 2, 1, 1>(kp, loader, s); break;
    case YMI_TILE_32x64_K2: rc = launch_cfg<1, 2, 2, 1, 1>(kp, loader, s); break;
    default: return YMI_EARG;
  }
  if (pr) { hipEventRecord(pr->e1, s); ++g_prof_n; }
  return rc;
}

}  // namespace

extern "C" {

double ymi_conv_flops(const ymi_conv_desc *d) {
  return 2.0 * d->B * d->Ho * d->Wo * (double)d->Cout * d->kh * d->kw * (double)(d->cin_alg > 0 ? d->cin_alg : d->Cin);
}

int ymi_conv_pick_tile(const ymi_conv_desc *d) { return d ? pick_tile(d) : YMI_ENULL; }

int ymi_conv2d_nhwc_f32(const ymi_conv_desc *d, void *stream) {
  if (!d) return YMI_ENULL;
  return run_conv(d, d->Cin == 4 ? 1 : 0, nullptr, 0, (hipStream_t)stream);
}

int ymi_dcn_v2_forward_f32(const ymi_dcn_desc *d, void *stream) {
  if (!d || !d->offmask) return YMI_ENULL;
  if (d->ldo < 27) return YMI_ESHAPE;
  return run_conv(&d->conv, 2, d->offmask, d->ldo, (hipStream_t)stream);
}

int ymi_prof_enable(int on) { g_prof_on = on; return YMI_OK; }
int ymi_prof_count(void) { return g_prof_n; }
int ymi_prof_reset(void) { g_prof_n = 0; return YMI_OK; }
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
