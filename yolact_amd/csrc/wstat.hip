// Weight-stationary streaming convolution for NARROW outputs (Cout <= 64) on the fp16x2 matrix path of gfx950:
// ymi_conv2d_nhwc_f32 with a YMI_TILE_DCNP | YMI_DCNP_WS_* tile.  What it is for: the 27 (-> 32) channel conv_offset_mask of every
// DCN layer (dcn_v2.py:107-112), proto.10 (yolact.py mask_proto_net: 256 -> 32, 1x1), the 64-channel conv1 of the first ResNet stage
// (backbone.py:37-57).  Such a layer multiplies little and reads a lot: its cost is getting the input to the matrix cores.
//
// The LDS-staged tiles (conv_igemm.hip, dcn.hip) stage the A operand through LDS because every wave of a block multiplies every
// row of the block tile by ITS columns.  With <= 64 output channels one wave takes all the columns of its rows, so nothing is
// shared between waves but the filters:
//   * the filters of the block's K range (fp16 planes of engine.Packed.h2()) are copied to LDS ONCE per block by LDS-DMA and stay
//     there ("weight-stationary"); a K that does not fit (3x3 x 128 channels = 36 chunks of 4 KB) is cut into split_k ranges,
//     partial sums through split_ws and the deterministic second pass of the other split-K paths;
//   * the activations never touch LDS: lane (row r, k-group g) of v_mfma_f32_16x16x32_f16 needs A[r][8g .. 8g+7], i.e. 32
//     contiguous bytes of the pixel's channel vector — two buffer_load_dwordx4 straight into the registers the MFMA reads (after
//     the fp16 split); 4 lanes cover the pixel's 128-byte line of the chunk, a wave instruction touches 16 lines;
//   * no barrier after the filter copy: every wave streams its own rows, D chunks of loads in flight in a register ring, at its
//     own pace (the pipelined kernel's chunk barrier couples 2 .. 16 waves per 32-deep step, 6 .. 12 MFMAs each);
//   * the MFMA is issued as D^T = W X^T (filters as the A operand): a lane ends up with FOUR CONSECUTIVE output channels of one
//     pixel, so the epilogue is float4 scale / bias / activation / store with no transposition.
// Measured (profiles/r04_ws_probe.txt): a draw with the 32-column tiles of the pipelined kernel — these launches take 16 .. 30 us, of
// which launch, ramp, tail and the split-K second pass are most; the tuner picks per shape.
// Arithmetic: the fp16x2 scheme of the other tiles (x * s = h + l, s a power of two from the tensor's magnitude bound; products
// h*l, l*h, h*h; fp32 accumulation), same filter planes, same scale_h2.
#include "common.h"
#include <type_traits>
#include "../../include/yolact_amd.h"

int ymi_internal_prof_begin(double flops, int tile, int kind, hipStream_t s);
void ymi_internal_prof_end(int idx, hipStream_t s);
int ymi_internal_splitk_fixup(const float *part, long gstride, int S, long M, int Cout, int ldy, float *y, const float *scale,
                              const float *bias, const float *res, int res_ld, int act, int res_after_act, float *y_amax,
                              hipStream_t s);

namespace {

constexpr int BK = 32;
constexpr unsigned OOB = 0x80000000u;   // buffer offset >= num_records: the load returns zeros
constexpr int WS_LDS_MAX = 64 * 1024;   // filter bytes of a block's K range: two blocks per CU at the maximum

typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
struct Split2 { f16x8 h, l; };

struct WsParams {
  const float *x, *scale_h2, *bias, *x_amax;
  const void *w_h2;
  float *y, *y_amax;
  int B, H, W, Cin, ldx, Ho, Wo, Cout, stride, Kpad, ldy, act;
  int taps, kw, pad;
  int M, HoWo, nk, nk_split;
  long y_gs;
  unsigned x_bytes, w_plane;
};

// MT: 16-row tiles per wave; NT: 16-column tiles (2: Cout <= 32, 4: Cout <= 64); NW: waves per block (they share the filters,
// nothing else); D: chunks of loads in flight per wave.
template <int MT, int NT, int NW, int D>
__global__ __launch_bounds__(64 * NW)
void ws_h2_k(const WsParams p) {
#if defined(__HIP_DEVICE_COMPILE__)   // (host pass: empty body, see conv_igemm.hip)
  extern __shared__ __attribute__((aligned(16))) char ws_lds[];
  constexpr int PLANE = NT * 16 * 64, CH = 2 * PLANE;   // bytes: one fp16 plane of a chunk (64-byte rows), one chunk
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lr = lane & 15, g = lane >> 4;
  const int tile_m = ymi_xcd_remap(blockIdx.x, gridDim.x);
  const int m_wave = (tile_m * NW + wave) * (MT * 16);

  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.x, 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.w_h2, 0, (int)(2 * p.w_plane), 0x00020000);

  float sA, invA;
  ymi_h2_scale(ymi_amax_read(p.x_amax), sA, invA);
  const ymi_amax_pre apre = ymi_amax_prefetch(p.y_amax);

  // K range of this block (split-K: gridDim.y ranges of nk_split chunks)
  const int kc0 = blockIdx.y * p.nk_split;
  const int my_nk = (p.nk - kc0) < p.nk_split ? (p.nk - kc0) : p.nk_split;

  // ---- the filters of the range -> LDS, once.  Unit u = (chunk, plane, 16-row group): one DMA of 16 rows x 64 bytes; lane (row
  // lane >> 2, 16-byte slot lane & 3) fetches the piece that belongs at that slot of the XOR-swizzled image (slot s of row r holds
  // piece s ^ ((r >> 2) & 3): the fragment reads below are conflict-free)
  {
    const int row16 = lane >> 2;
    const int units = my_nk * 2 * NT;
    for (int u = wave; u < units; u += NW) {
      const int kc = u / (2 * NT), rem = u - kc * (2 * NT), plane = rem / NT, rg = rem - plane * NT;
      const int row = rg * 16 + row16, piece = (lane & 3) ^ ((row >> 2) & 3);
      const unsigned off = (unsigned)plane * p.w_plane + (unsigned)((row * p.Kpad + (kc0 + kc) * BK + 8 * piece) * 2);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_ptr_t)(ws_lds + kc * CH + plane * PLANE + rg * 1024), 16, off, 0, 0, 0);
    }
  }

  // ---- this lane's rows: pixel lr of each of the wave's MT row tiles, channels 8g .. 8g+7 of a chunk -------------------------
  int g_iy0[MT], g_ix0[MT], g_ib[MT];
  bool g_ok[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int m = m_wave + i * 16 + lr;
    g_ok[i] = m < p.M;
    const int mm = g_ok[i] ? m : 0;
    const int b = mm / p.HoWo, pix = mm - b * p.HoWo;
    const int oy = pix / p.Wo, ox = pix - oy * p.Wo;
    g_iy0[i] = oy * p.stride - p.pad;
    g_ix0[i] = ox * p.stride - p.pad;
    g_ib[i] = b * p.H * p.W;
  }
  unsigned gq[MT];                                      // byte offset of the lane's 32 bytes at the current tap (OOB: padding / no row)
  const int cpt = p.Cin / BK;                           // chunks per tap
  int g_tap = kc0 / cpt, g_c = (kc0 - g_tap * cpt) * BK;
  int g_left = my_nk;
  bool g_first = true;
  auto geom = [&](int tap) {
    const int ky = tap / p.kw, kx = tap - p.kw * ky;
    const bool live = tap < p.taps;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int h = g_iy0[i] + ky, w = g_ix0[i] + kx;
      const bool in = live && g_ok[i] && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
      gq[i] = in ? (unsigned)(((g_ib[i] + h * p.W + w) * p.ldx + 8 * g) * 4) : OOB;
    }
  };
  f32x4 ring[D][MT][2];
  auto request = [&](auto slot_c) {                     // the loads of the next chunk of the range (past its end: all out of bounds)
    constexpr int S = decltype(slot_c)::value;
    if (g_c == 0 || g_first) {
      g_first = false;
      geom(g_left > 0 ? g_tap : p.taps);
    }
    const int so = g_c * 4;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      ring[S][i][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, gq[i], so, 0));
      ring[S][i][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, gq[i] + 16u, so, 0));
    }
    g_c += BK;
    if (g_c == p.Cin) { g_c = 0; ++g_tap; }
    if (--g_left == 0) {                                // past the end of the range (it may end inside a tap): nothing but zeros
#pragma unroll
      for (int i = 0; i < MT; ++i) gq[i] = OOB;
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  asm volatile("" ::: "memory");                        // (the requests below stay behind the filter copy in program order)
  // prologue: D chunks requested behind the filter copy; the copy (older, vmcnt is in order) has landed when at most those
  // 2 * MT * D loads are outstanding
  request(std::integral_constant<int, 0>{});
  if constexpr (D > 1) request(std::integral_constant<int, 1>{});
  if constexpr (D > 2) request(std::integral_constant<int, 2>{});
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * MT * D) : "memory");
  __syncthreads();
  if (m_wave >= p.M) return;                            // a wave past the last row (ragged last block): nothing to multiply

  const int b_fro = lr * 64 + ((g ^ ((lr >> 2) & 3)) * 16);     // this lane's 16 bytes of a 16-row filter tile: row lr, piece g
  auto step = [&](int st, auto slot_c) {
    constexpr int S = decltype(slot_c)::value;
    Split2 fa[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const f32x4 v = ring[S][i][q] * sA;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const _Float16 h = (_Float16)v[e];
          fa[i].h[4 * q + e] = h;
          fa[i].l[4 * q + e] = (_Float16)(v[e] - (float)h);
        }
      }
    }
    request(slot_c);                                    // chunk st + D into the registers just freed
    const int kc = st < my_nk ? st : my_nk - 1;         // (a padding step multiplies zeros by the last chunk's filters)
    const char *Bc = ws_lds + kc * CH + b_fro;
    Split2 fb[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      fb[j].h = *reinterpret_cast<const f16x8 *>(Bc + j * 1024);
      fb[j].l = *reinterpret_cast<const f16x8 *>(Bc + PLANE + j * 1024);
    }
#pragma unroll
    for (int pr = 0; pr < 3; ++pr)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const f16x8 fx = pr == 1 ? fa[i].l : fa[i].h;
          const f16x8 fw = pr == 0 ? fb[j].l : fb[j].h;
          acc[i][j] = ymi_mfma16(fw, fx, acc[i][j]);      // D^T[cout][pixel] += W X^T
        }
  };
  const int nk_pad = ((my_nk + D - 1) / D) * D;
  for (int st = 0; st < nk_pad; st += D) {
    step(st, std::integral_constant<int, 0>{});
    if constexpr (D > 1) step(st + 1, std::integral_constant<int, 1>{});
    if constexpr (D > 2) step(st + 2, std::integral_constant<int, 2>{});
  }

  // ---- epilogue: lane holds channels 16 j + 4 g .. + 3 of pixel lr of row tile i ---------------------------------------------
  const float slope = p.act == YMI_ACT_RELU ? 0.f : (p.act == YMI_ACT_LEAKY01 ? 0.1f : 1.f);
  float am = 0.f;
  float *ybase = p.y + (size_t)blockIdx.y * p.y_gs;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = j * 16 + 4 * g;
    if (n < p.Cout) {                                   // Cout % 4 == 0 (host check): the four channels exist together
      f32x4 sc, bi = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 4; ++e) { sc[e] = p.scale_h2[n + e] * invA; if (p.bias) bi[e] = p.bias[n + e]; }
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int m = m_wave + i * 16 + lr;
        f32x4 v = acc[i][j] * sc + bi;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], slope * v[e]);
        if (m < p.M) {
          am = fmaxf(am, ymi_absmax4(v));
          *reinterpret_cast<f32x4 *>(ybase + (size_t)m * p.ldy + n) = v;
        }
      }
    }
  }
  if (p.y_amax) ymi_amax_finish(apre, am);
#endif
}

template <int MT, int NT, int NW, int D>
int launch_ws(const WsParams &p, int splits, hipStream_t s) {
  constexpr int BM = NW * MT * 16;
  const int lds = p.nk_split * NT * 2048;
  hipLaunchKernelGGL((ws_h2_k<MT, NT, NW, D>), dim3((p.M + BM - 1) / BM, splits), dim3(64 * NW), lds, s, p);
  return ymi_launch_status();
}

}  // namespace

// internal (called by ymi_conv2d_nhwc_f32 for YMI_TILE_H2 | YMI_TILE_DCNP | YMI_DCNP_WS_*): the weight-stationary streaming kernel
// for a validated descriptor.  3x3 / pad 1 or 1x1 / pad 0, any stride, Cin % 32 == 0, one dense output of Cout <= 32 (.._x32 tiles)
// or <= 64 (.._x64), Cout % 4 == 0, activation none / ReLU / LeakyReLU, no residual; the filters of a K range must fit 64 KB of LDS
// (ceil(nk / split_k) * 4 KB per 32 columns): YMI_EARG otherwise.  Profiling record kind 11.
int ymi_internal_ws_conv(const ymi_conv_desc *d, int base_tile, hipStream_t s) {
  const ymi_conv_seg &g0 = d->seg[0];
  const long HoWo = (long)d->Ho * d->Wo, M = (long)d->B * HoWo;
  if (base_tile < YMI_DCNP_WS_128x32_W4 || base_tile > YMI_DCNP_WS_512x64_W8) return YMI_EARG;
  const int NT = base_tile >= YMI_DCNP_WS_128x64_W4 ? 4 : 2;
  if (!((d->kh == 3 && d->kw == 3 && d->pad == 1) || (d->kh == 1 && d->kw == 1 && d->pad == 0))) return YMI_EARG;
  if (d->Cin % 32 != 0 || d->Kpad != d->kh * d->kw * d->Cin) return YMI_EARG;
  if (d->nseg != 1 || g0.n0 != 0 || g0.n1 < d->Cout || (d->Cout & 3) || d->Cout > 16 * NT || (g0.row_stride & 3) ||
      (((uintptr_t)g0.ptr) & 15) || g0.batch_stride != HoWo * g0.row_stride || g0.act > YMI_ACT_LEAKY01 || g0.act < 0)
    return YMI_EARG;
  if (d->res_mode != YMI_RES_NONE) return YMI_EARG;
  if (!d->w_h2 || !d->scale_h2 || !d->x_amax || (((uintptr_t)d->w_h2) & 15)) return YMI_ENULL;
  if (M >= (1L << 29) || M * (long)g0.row_stride >= (1L << 31) || (long)d->B * d->H * d->W * d->ldx >= (1L << 29)) return YMI_ESHAPE;
  const int S = d->split_k > 1 ? d->split_k : 1;
  const int nk = d->Kpad / BK;
  const int per = (nk + S - 1) / S;
  if (per * NT * 2048 > WS_LDS_MAX) return YMI_EARG;   // the block's filters have to fit its LDS: ask for more K ranges
  if (S > 1) {
    if (S > 16 || per * (S - 1) >= nk) return YMI_EARG;                 // every range non-empty
    if (!d->split_ws || !d->winv_h2) return YMI_ENULL;
    if ((((uintptr_t)d->split_ws) & 15) || M * (long)d->Cout >= (1L << 29)) return YMI_ESHAPE;
  }
  WsParams p;
  p.x = d->x; p.scale_h2 = d->scale_h2; p.bias = d->bias; p.x_amax = d->x_amax; p.w_h2 = d->w_h2;
  p.y = g0.ptr; p.y_amax = d->y_amax;
  p.B = d->B; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.ldx = d->ldx; p.Ho = d->Ho; p.Wo = d->Wo; p.Cout = d->Cout;
  p.stride = d->stride; p.Kpad = d->Kpad; p.ldy = g0.row_stride; p.act = g0.act;
  p.taps = d->kh * d->kw; p.kw = d->kw; p.pad = d->pad;
  p.M = (int)M; p.HoWo = (int)HoWo; p.nk = nk; p.nk_split = per; p.y_gs = 0;
  p.x_bytes = (unsigned)((size_t)d->B * d->H * d->W * d->ldx * sizeof(float));
  p.w_plane = (unsigned)((((long)d->Cout + 127) / 128 * 128) * d->Kpad * 2L);
  if (S > 1) {           // partial launches undo the operand scales only (true partial sums), the second pass does the rest
    p.scale_h2 = d->winv_h2; p.bias = nullptr; p.act = YMI_ACT_NONE; p.y_amax = nullptr;
    p.y = d->split_ws; p.ldy = d->Cout; p.y_gs = M * (long)d->Cout;
  }
  const double flops = 2.0 * (double)M * (double)(d->cout_alg > 0 ? d->cout_alg : d->Cout) * (double)(d->kh * d->kw) *
                       (double)(d->cin_alg > 0 ? d->cin_alg : d->Cin);
  const int pr = ymi_internal_prof_begin(flops, base_tile | YMI_TILE_H2 | YMI_TILE_DCNP, 11, s);
  int rc;
  switch (base_tile) {                                   // <16-row tiles per wave, 16-column tiles, waves, chunks in flight>
    case YMI_DCNP_WS_128x32_W4: rc = launch_ws<2, 2, 4, 3>(p, S, s); break;
    case YMI_DCNP_WS_256x32_W8: rc = launch_ws<2, 2, 8, 3>(p, S, s); break;
    case YMI_DCNP_WS_256x32_W4: rc = launch_ws<4, 2, 4, 2>(p, S, s); break;
    case YMI_DCNP_WS_512x32_W8: rc = launch_ws<4, 2, 8, 2>(p, S, s); break;
    case YMI_DCNP_WS_128x64_W4: rc = launch_ws<2, 4, 4, 3>(p, S, s); break;
    case YMI_DCNP_WS_256x64_W8: rc = launch_ws<2, 4, 8, 3>(p, S, s); break;
    case YMI_DCNP_WS_256x64_W4: rc = launch_ws<4, 4, 4, 2>(p, S, s); break;
    default: rc = launch_ws<4, 4, 8, 2>(p, S, s); break;   // YMI_DCNP_WS_512x64_W8
  }
  if (rc == YMI_OK && S > 1)
    rc = ymi_internal_splitk_fixup(d->split_ws, M * (long)d->Cout, S, M, d->Cout, g0.row_stride, g0.ptr, d->scale, d->bias, nullptr, 0,
                                   g0.act, 0, d->y_amax, s);
  ymi_internal_prof_end(pr, s);
  return rc;
}
