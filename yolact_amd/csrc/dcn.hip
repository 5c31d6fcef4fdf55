// DCNv2 forward (external/DCNv2/src/cuda/dcn_v2_cuda.cu:42-172, dcn_v2_im2col_cuda.cu:25-54,125-195) as ONE software-pipelined
// gather-GEMM on the fp16x2 matrix path of gfx950 — round 4's replacement for the register-staged `LOADER 2` of
// csrc/conv_igemm.hip on the fp16x2 plans (that loader stays for the exact-fp32 plans and for descriptors this kernel does
// not take).
//
// What was wrong with the old loader (profiles/r03_bench_v3_plus_b8.json: 93 TFLOP/s, 0.11 of the fp16x2 peak, 125 us per layer
// against 53 us for the plain 3x3 of the same shape): every 32-deep K chunk was ONE serial round trip — issue the four bilinear
// corner loads of the next chunk, 12 MFMAs (0.16 us), wait for the loads (vmcnt(0): 1 - 1.7 us under load), ds_write, barrier —
// and the first chunk of every tap added a second dependent round trip (offset / mask logits -> addresses -> data).
//
// This kernel keeps the idea (no `columns` tensor: the modulated bilinear sample is formed in registers and goes straight into
// the A tile) and rebuilds the schedule around the latency:
//   * the corner loads of chunk c+3 are issued while chunk c is multiplied: a two-slot register ring, i.e. two to three K
//     steps of latency cover; a sample is `buffer_load_dwordx4` x 4 corners, 8 lanes = 128 contiguous bytes (32 channels) per
//     corner, the only per-chunk address work is ONE scalar offset (the channel chunk) — corner offsets live in registers per tap;
//   * the offset / mask logits of tap t+1 are fetched when tap t starts, so a tap change costs arithmetic, not a round trip;
//   * every (pixel, tap, channel) sample is formed ONCE per row block and written to LDS already split into the two fp16 planes
//     of the fp16x2 arithmetic (scaled by the tensor's power-of-two scale: |sample| <= max|x| because the bilinear weights are
//     convex and the modulation is a sigmoid) — the consuming waves run the MFMA loop with NO operand split, like the
//     pre-split Winograd GEMM (PREC 4 of conv_igemm.hip);
//   * filters arrive as fp16 planes by LDS-DMA three chunks ahead (four B stages), the A planes are double-buffered; vmcnt is
//     in-order, so the filter DMA has to run as far ahead as the gather or waiting for it would drain the gather too;
//   * counted `s_waitcnt vmcnt(N)` + one raw `s_barrier` per chunk; the loop body is ONE straight-line step (requests past the
//     end of K are out-of-bounds buffer loads, i.e. zeros without a memory access) so its vmcnt is a constant.
// K order, LDS images, fragment layout, MFMA order (h*l, l*h, h*h) and the fast-path epilogue are those of the fp16x2 tiles of
// conv_igemm.hip, so the filter planes / scale_h2 of engine.Packed.h2() are used unchanged.
#include "common.h"
#include <stdlib.h>
#include <type_traits>
#include "../../include/yolact_amd.h"

int ymi_internal_prof_begin(double flops, int tile, int kind, hipStream_t s);
void ymi_internal_prof_end(int idx, hipStream_t s);

// cache policy of the A-operand requests (the activation rows, read once or twice per launch from beyond L2): 0 = default, 2 = nt
// (non-temporal).  A/B'd in session r6r (stage `auxab` of tools/gpu_session.sh): see profiles/r06_aux_ab.txt
#ifndef YMI_A_AUX
#define YMI_A_AUX 0
#endif

namespace {

constexpr int BK = 32;
constexpr unsigned OOB = 0x80000000u;   // buffer offset >= num_records: the load returns zeros

typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
struct Split2 { f16x8 h, l; };

struct DcnParams {
  const float *x, *offmask, *scale_h2, *bias, *x_amax;
  const void *w_h2;
  float *y, *y_amax;
  int B, H, W, Cin, ldx, Ho, Wo, Cout, stride, Kpad, ldo, ldy, act, mask_is_prob;
  int taps, kw, pad;             // PLAIN (an ordinary convolution): kh * kw (9 or 1), kw, padding; the DCN path is 9 / 3 / 1
  const float *res; int res_ld, res_after_act;   // PLAIN: optional residual [M, res_ld] added before (or after) the activation
  int M, HoWo, tiles_n, nk;
  int nk_split;  // chunks per K range (gridDim.y ranges; == nk without split-K); range y writes raw partial sums to y + y * y_gs
  long y_gs;
  unsigned x_bytes, om_bytes, w_plane;
  int om_layout;
  int abl;      // diagnostics build only (env YMI_DCN_ABLATE): bit0 corner loads -> OOB (no memory access), bit1 filter DMAs -> OOB,
                // bit2 no combine / LDS store, bit3 no MFMAs, bit4 no barrier, bit5 no vmcnt wait — wrong results by design; bit6 no residency cap
  unsigned long long *trace;    // diagnostics build only (env YMI_PIPE_TRACE = device address of a u64 buffer, 16 slots per block): phase
                                // time stamps of wave 0 (tools/pipe_trace.py).  Stamps are kept in registers and written behind the
                                // kernel's last store (a store inside the pipeline would change its counted vmcnt waits)
  int trace_mode;               // 1: stamps at the kernel's own synchronisation points only; 2: + a stamp that WAITS for the tensor scale
};

template <int WM, int WN, int TM, int TN, int RING>
constexpr int dcn_lds_floats() {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int pipe = 2 * (2 * BM * 16) + (RING + 2) * (2 * BN * 16), epi = BM * (BN + 4);
  return pipe > epi ? pipe : epi;
}
template <int WM, int WN, int TM, int TN, int RING, bool PLAIN>
constexpr int dcn_occupancy() {     // blocks per CU: LDS-limited (at most two), and — DCN gather only — capped by its register ring (16
                                    // registers per row and ring slot): 4-wave blocks and 8-wave blocks with one row per thread run two
                                    // per CU, anything larger one.  An ordinary convolution (PLAIN) loads one sample, not four corners,
                                    // and has no such cap (68 - 110 registers)
  constexpr int occ = (160 * 1024) / (dcn_lds_floats<WM, WN, TM, TN, RING>() * 4);
  constexpr int nw = WM * WN, ra = (WM * TM * 32) / (8 * nw);
  constexpr int cap = (!PLAIN && (nw > 8 || (nw == 8 && ra > 1))) ? 1 : (WN * TN == 1 && nw <= 4) ? 4 : 2;   // (32-column tiles: small blocks)
  return occ > cap ? cap : (occ < 1 ? 1 : occ);
}

// RING: corner loads run RING + 1 chunks ahead of the MFMAs in a ring of RING register slots (RING + 2 filter stages).  2 for the
// small tiles (a step is shorter than a load round trip); 1 for the 96 .. 192-row tiles, whose step is >= 1000 TA cycles.
// PLAIN: the same pipeline as an ORDINARY convolution (3x3 / pad 1 or 1x1 / pad 0, any stride, Cin % 32 == 0): one load per sample
// instead of four corners, integer tap geometry, optional residual — ymi_conv2d_nhwc_f32 with a YMI_TILE_DCNP tile.  What it has
// that the LDS-DMA tiles of conv_igemm.hip do not: the A operand is split into its fp16 planes ONCE by the loading thread (the
// consuming waves run a split-free MFMA loop), 6 .. 16-wave blocks of 32 x 64 wave tiles, 256-column tiles.
template <int WM, int WN, int TM, int TN, int RING, bool PLAIN>
__global__ __launch_bounds__(64 * WM * WN, (dcn_occupancy<WM, WN, TM, TN, RING, PLAIN>() * (WM * WN) + 3) / 4)
void pipe_h2_k(const DcnParams p) {
#if defined(__HIP_DEVICE_COMPILE__)   // (host pass: empty body, see conv_igemm.hip)
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NW = WM * WN, NT = 64 * NW;
  constexpr int RPP = NT / 8, RA = BM / RPP;            // gather: 8 lanes (32 channels) per row, RA rows per thread per chunk
  constexpr int BUNITS = (2 * BN) / 16;                 // filter-plane DMA pieces of a chunk: (plane, 16-row group), 16 rows x 64 bytes each
  constexpr int RB = (BUNITS + NW - 1) / NW;            // pieces per wave per chunk; when NW does not divide BUNITS the surplus pieces
                                                        // repeat the last unit (same bytes to the same place: every wave issues the
                                                        // same number of DMAs, so the vmcnt below is one constant)
  static_assert(BM % RPP == 0 && RA >= 1, "tile rows vs gather passes");
  static_assert(RING == 1 || RING == 2, "ring depth");
  constexpr int NSB = RING + 2;                         // filter stages: chunk st (multiplied), st+1 .. st+RING (landed / in flight), st+RING+1 (requested)
  constexpr int A_STAGE = 2 * BM * 16, B_STAGE = 2 * BN * 16;    // floats: two fp16 planes of 64-byte rows
  constexpr int NC = PLAIN ? 1 : 4;                     // loads per sample: the four bilinear corners, or the pixel itself
  constexpr int NG = NC * RA, NB = RB;                  // VMEM operations of one chunk: corner loads, filter DMAs
  constexpr int N_STEADY = RING * (NG + NB);            // operations issued behind the filter DMA of chunk st+1 at the end of step st
  static_assert(N_STEADY <= 63, "vmcnt is a 6-bit counter");
  constexpr int ELD = BN + 4;
  constexpr int LDS_FLOATS = dcn_lds_floats<WM, WN, TM, TN, RING>();
  __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
  float *const Abase = lds, *const Bbase = lds + 2 * A_STAGE;

  const int t = threadIdx.x;
  const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int kq = t & 7, r0 = t >> 3;
#ifdef YMI_DIAGNOSTICS
  unsigned long long tr_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tr_rt0 = 0;
  const bool tracing = p.trace != nullptr;
#define YMI_STAMP(i) do { if (tracing) tr_[i] = __builtin_amdgcn_s_memtime(); } while (0)
  if (tracing) tr_rt0 = __builtin_amdgcn_s_memrealtime();
  YMI_STAMP(0);
#else
#define YMI_STAMP(i) do { } while (0)
#endif

  const int logical = ymi_xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = logical % p.tiles_n, tile_m = logical / p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.x, 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc((void *)p.offmask, 0, (int)p.om_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.w_h2, 0, (int)(2 * p.w_plane), 0x00020000);

  float sA, invA;
  ymi_h2_scale(ymi_amax_read(p.x_amax), sA, invA);
  const ymi_amax_pre apre = ymi_amax_prefetch(p.y_amax);
#ifdef YMI_DIAGNOSTICS
  if (tracing && p.trace_mode == 2) {      // wait for the magnitude bound (the stamp takes the scale as an operand)
    unsigned long long tt;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt) : "s"(__builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sA))) : "memory");
    tr_[1] = tt;
  }
#endif

  // ---- epilogue mapping (conv_igemm.hip's fast path) ------------------------------------------------------------------------
  constexpr int C4 = BN / 4, RSTEP = NT / C4, RPT = BM / RSTEP;
  const int c4 = t % C4, rbase = t / C4;
  const int n = n0 + 4 * c4;
  // ---- gather bookkeeping: rows r0 + RPP*i of the tile, channels 4*kq .. 4*kq+3 of the chunk ------------------------------
  int g_iy0[RA], g_ix0[RA], g_ib[RA];
  unsigned g_om[RA];                                    // byte offset of the row's offset / mask-logit vector (OOB past M)
  int a_st[RA];                                         // byte offset of this thread's 8-byte piece inside an A plane
#pragma unroll
  for (int i = 0; i < RA; ++i) {
    const int row = r0 + RPP * i, m = m0 + row;
    a_st[i] = row * 64 + (((kq >> 1) ^ ((row >> 2) & 3)) * 16) + (kq & 1) * 8;
    if (m < p.M) {
      const int b = m / p.HoWo, pix = m - b * p.HoWo;
      const int oy = pix / p.Wo, ox = pix - oy * p.Wo;
      g_iy0[i] = oy * p.stride - (PLAIN ? p.pad : 1);
      g_ix0[i] = ox * p.stride - (PLAIN ? p.pad : 1);
      g_ib[i] = b * p.H * p.W;
      g_om[i] = (unsigned)m * (unsigned)p.ldo * 4u;
    } else {
      g_iy0[i] = 0; g_ix0[i] = 0; g_ib[i] = 0;
      g_om[i] = OOB;
    }
  }
  unsigned b_off[RB];
  int b_lds[RB];
#pragma unroll
  for (int i = 0; i < RB; ++i) {
    const int u0 = wave + NW * i, u = u0 < BUNITS ? u0 : BUNITS - 1;
    const int plane = u / (BN / 16), rg = u - plane * (BN / 16);
    const int row = rg * 16 + (lane >> 2), lsl = (lane & 3) ^ ((row >> 2) & 3);
    b_off[i] = (unsigned)plane * p.w_plane + (unsigned)(((n0 + row) * p.Kpad + 8 * lsl) * 2);
    b_lds[i] = plane * (BN * 16) + rg * 256;
  }

  // per-tap sampling geometry of this thread's rows (dcn_v2_im2col_cuda.cu:143-193): byte offsets of the four corners (+ the
  // thread's channel slot; OOB where the corner contributes nothing) and the bilinear weights + the modulation.  Branch-free
  // (selects): the code is expanded once per ring slot.  tap >= 9 (requests past the end of K, see the main loop): all OOB.
  unsigned gq[RA][4];
  float gwt[RA][5];
  float raw[RA][3];                                     // dh, dw, mask logit of the next tap to resolve (fetched one tap ahead)
  auto raw_fetch = [&](int tap) {
    if constexpr (PLAIN) return;
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      if (p.om_layout) {           // [dh_k, dw_k, mask_k] per tap: one 12-byte load (ymi_dcn_desc.om_layout = 1)
        const unsigned o = g_om[i] != OOB ? g_om[i] + 12u * tap : OOB;
        const u32x3 v = __builtin_bit_cast(u32x3, __builtin_amdgcn_raw_buffer_load_b96(ors, o, 0, 0));
        const unsigned v0 = v[0], v1 = v[1], v2 = v[2];       // (scalar copies: hipcc 7.2 mis-indexes bit casts of vector elements)
        raw[i][0] = __uint_as_float(v0); raw[i][1] = __uint_as_float(v1); raw[i][2] = __uint_as_float(v2);
      } else {
        const unsigned o = g_om[i] != OOB ? g_om[i] + 8u * tap : OOB;
        raw[i][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ors, o, 0, 0));
        raw[i][1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ors, o, 4, 0));
        raw[i][2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ors, o, 4 * 18 - 4 * tap, 0));
      }
    }
  };
  const int ntaps = PLAIN ? p.taps : 9;
  auto geom = [&](int tap) {
    if constexpr (PLAIN) {       // an ordinary convolution: the tap's pixel, or nothing where it falls into the padding
      const int ky = tap / p.kw, kx = tap - p.kw * ky;
      const bool live = tap < ntaps;
#pragma unroll
      for (int i = 0; i < RA; ++i) {
        const int h = g_iy0[i] + ky, w = g_ix0[i] + kx;
        const bool in = live && g_om[i] != OOB && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
        gq[i][0] = in ? (unsigned)(((g_ib[i] + h * p.W + w) * p.ldx + 4 * kq) * 4) : OOB;
      }
      return;
    }
    const int ky = tap / 3, kx = tap - 3 * ky;
    const bool live = tap < 9;
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      const float dh = raw[i][0], dw = raw[i][1];
      const float mk = p.mask_is_prob ? raw[i][2] : 1.f / (1.f + expf(-raw[i][2]));
      const float h = (float)(g_iy0[i] + ky) + dh, w = (float)(g_ix0[i] + kx) + dw;
      const bool in = live && g_om[i] != OOB && h > -1.f && w > -1.f && h < (float)p.H && w < (float)p.W;
      const int hl = (int)floorf(h), wl = (int)floorf(w);
      const int hh = hl + 1, wh = wl + 1;
      const float lh = h - (float)hl, lw = w - (float)wl, uh = 1.f - lh, uw = 1.f - lw;
      const unsigned o1 = (unsigned)(((g_ib[i] + hl * p.W + wl) * p.ldx + 4 * kq) * 4);
      const unsigned dx = (unsigned)(p.ldx * 4), dy = (unsigned)(p.W * p.ldx * 4);
      // an out-of-range corner contributes 0 (dmcn_im2col_bilinear): no read at all (the buffer bounds check returns zeros)
      const bool t_ = in && hl >= 0, b_ = in && hh <= p.H - 1, l_ = wl >= 0, r_ = wh <= p.W - 1;
      gq[i][0] = (t_ && l_) ? o1 : OOB;
      gq[i][1] = (t_ && r_) ? o1 + dx : OOB;
      gq[i][2] = (b_ && l_) ? o1 + dy : OOB;
      gq[i][3] = (b_ && r_) ? o1 + dy + dx : OOB;
      gwt[i][0] = (t_ && l_) ? uh * uw : 0.f;
      gwt[i][1] = (t_ && r_) ? uh * lw : 0.f;
      gwt[i][2] = (b_ && l_) ? lh * uw : 0.f;
      gwt[i][3] = (b_ && r_) ? lh * lw : 0.f;
      gwt[i][4] = mk;
    }
  };

  // ---- the register ring of the gather: two chunks in flight ---------------------------------------------------------------
  f32x4 ring[RING][RA][NC];
  float ringw[RING][RA][5];
  // K range of this block (split-K: gridDim.y ranges of nk_split chunks; chunk-aligned, not necessarily tap-aligned)
  const int kc0 = blockIdx.y * p.nk_split;
  const int cpt = p.Cin / BK;                           // chunks per tap
  int g_tap = kc0 / cpt, g_c = (kc0 - g_tap * cpt) * BK;   // (tap, first channel) of the next chunk to request
  const int my_nk = (p.nk - kc0) < p.nk_split ? (p.nk - kc0) : p.nk_split;   // (the last range may be shorter)
  int g_left = my_nk;                                   // chunks of the range still to request
  bool g_first = true;
  auto tap_step = [&]() {                               // start of a chunk's requests: resolve the geometry at a tap boundary
    if (g_c == 0 || g_first) {                          // (or at the start of a range that begins inside a tap)
      g_first = false;
      geom(g_left > 0 ? g_tap : ntaps);
      if (g_tap + 1 < ntaps) raw_fetch(g_tap + 1);
    }
  };
  auto gather_row = [&](auto slot_c, int i) {           // the four corner loads of row i of the chunk at (g_tap, g_c)
    constexpr int S = decltype(slot_c)::value;
    const int so = g_c * 4;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
#ifdef YMI_DIAGNOSTICS
      ring[S][i][c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, (p.abl & 1) ? OOB : gq[i][c], so, YMI_A_AUX));
#else
      ring[S][i][c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, gq[i][c], so, YMI_A_AUX));
#endif
    }
    if constexpr (!PLAIN) {
#pragma unroll
      for (int e = 0; e < 5; ++e) ringw[S][i][e] = gwt[i][e];
    }
  };
  auto chunk_advance = [&]() {
    g_c += BK;
    if (g_c == p.Cin) { g_c = 0; ++g_tap; }
    if (--g_left == 0) {                                // past the end of the range: every further request is an out-of-bounds load
#pragma unroll
      for (int i = 0; i < RA; ++i)
#pragma unroll
        for (int c = 0; c < NC; ++c) gq[i][c] = OOB;
    }
  };
  // sample -> two fp16 planes -> LDS (row i of the chunk held in ring slot S)
  auto combine_row = [&](auto slot_c, int i, float *As) {
    constexpr int S = decltype(slot_c)::value;
#ifdef YMI_DIAGNOSTICS
    if (p.abl & 4) return;
#endif
    f32x4 v;
    if constexpr (PLAIN) {
      v = ring[S][i][0] * sA;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float a = ringw[S][i][0] * ring[S][i][0][e];
        a = __builtin_fmaf(ringw[S][i][1], ring[S][i][1][e], a);
        a = __builtin_fmaf(ringw[S][i][2], ring[S][i][2][e], a);
        a = __builtin_fmaf(ringw[S][i][3], ring[S][i][3][e], a);
        v[e] = (a * ringw[S][i][4]) * sA;
      }
    }
    f16x4 h4, l4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const _Float16 h = (_Float16)v[e];
      h4[e] = h;
      l4[e] = (_Float16)(v[e] - (float)h);
    }
    char *dst = reinterpret_cast<char *>(As) + a_st[i];
    *reinterpret_cast<f16x4 *>(dst) = h4;
    *reinterpret_cast<f16x4 *>(dst + BM * 64) = l4;
  };
  // (a chunk index past the end of K reads the following filter rows — or zeros past the buffer — into a stage nobody multiplies)
  auto issue_b_piece = [&](int kc, int stage, int i) {
#ifdef YMI_DIAGNOSTICS
    if (p.abl & 2) { __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_ptr_t)(Bbase + stage * B_STAGE + b_lds[i]), 16, OOB, 0, 0, 0); return; }
#endif
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_ptr_t)(Bbase + stage * B_STAGE + b_lds[i]), 16, b_off[i], (kc0 + kc) * (BK * 2), 0, 0);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragments: plane images of 64-byte rows, 16-byte slot s of row r at s ^ ((r >> 2) & 3); lane half h of step s2 holds
  // k = 16 s2 + 8 h .. + 7 of row lane & 31
  Split2 pa[TM], pb[TN];
  const int psw = ((lane & 31) >> 2) & 3, hh_ = lane >> 5;
  const int fro[2] = {(lane & 31) * 16 + 4 * ((0 + hh_) ^ psw), (lane & 31) * 16 + 4 * ((2 + hh_) ^ psw)};
  auto load_frag = [&](const float *As, const float *Bs, int s2) {
    const float *Ap = As + (wm * TM * 32) * 16 + fro[s2];
    const float *Bp = Bs + (wn * TN * 32) * 16 + fro[s2];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      pa[i].h = *reinterpret_cast<const f16x8 *>(Ap + i * 32 * 16);
      pa[i].l = *reinterpret_cast<const f16x8 *>(Ap + i * 32 * 16 + BM * 16);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      pb[j].h = *reinterpret_cast<const f16x8 *>(Bp + j * 32 * 16);
      pb[j].l = *reinterpret_cast<const f16x8 *>(Bp + j * 32 * 16 + BN * 16);
    }
  };

#define YMI_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
#define YMI_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

  // One K step: the MFMAs of chunk st, and between them the pieces of the producer side — combine row q of chunk st+1 (ring
  // slot SLOT), request the same row of chunk st+3 into the registers just freed, one filter DMA of chunk st+3.  EVERY step does
  // all of it: the requests of the last three steps go past the end of K (tap >= 9: all-OOB corner loads, i.e. zeros without a
  // memory access; filter DMAs into stages nobody multiplies), which keeps the step a single straight-line body with one
  // constant vmcnt.
  constexpr int NPIECE = RA + RB, NPOS = 6 * TM * TN;
  int bst = 0;                                          // filter stage of the chunk being multiplied (= st mod NSB)
  auto step = [&](int st, auto slot_c) {
    const float *As = Abase + (st & 1) * A_STAGE, *Bs = Bbase + bst * B_STAGE;
    float *An = Abase + ((st + 1) & 1) * A_STAGE;
    const int bnx = bst == 0 ? NSB - 1 : bst - 1;      // stage of chunk st + RING + 1 = (st - 1) mod NSB: free since the last barrier
    load_frag(As, Bs, 0);
    tap_step();
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      if (s2 == 1) load_frag(As, Bs, 1);
#pragma unroll
      for (int pr = 0; pr < 3; ++pr)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const f16x8 fa_ = pr == 1 ? pa[i].l : pa[i].h;
            const f16x8 fb_ = pr == 0 ? pb[j].l : pb[j].h;
#ifdef YMI_DIAGNOSTICS
            if (!(p.abl & 8))
#endif
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa_, fb_, acc[i][j], 0, 0, 0);
            const int pos = ((s2 * 3 + pr) * TM + i) * TN + j;
#pragma unroll
            for (int q = 0; q < NPIECE; ++q) {
              if ((NPOS * q) / NPIECE == pos) {
                if (q < RA) {
                  combine_row(slot_c, q, An);
                  gather_row(slot_c, q);
                } else {
                  issue_b_piece(st + RING + 1, bnx, q - RA);
                }
              }
            }
          }
    }
    chunk_advance();
    bst = bst + 1 == NSB ? 0 : bst + 1;
#ifdef YMI_DIAGNOSTICS
    if (!(p.abl & 32)) YMI_WAIT_VM(N_STEADY);
    if (!(p.abl & 16)) YMI_BARRIER();
#else
    YMI_WAIT_VM(N_STEADY);        // the filter DMA of chunk st+1 has the RING * (NG + NB) operations of the last RING steps behind it
    YMI_BARRIER();
#endif
  };

  // ---- prologue: chunks 0 .. RING requested, chunk 0 combined ---------------------------------------------------------------
  YMI_STAMP(2);                 // index arithmetic done
  raw_fetch(g_tap);
  tap_step();
#pragma unroll
  for (int i = 0; i < RA; ++i) gather_row(std::integral_constant<int, 0>{}, i);
#pragma unroll
  for (int i = 0; i < RB; ++i) issue_b_piece(0, 0, i);
  chunk_advance();
  if constexpr (RING == 2) {
    tap_step();
#pragma unroll
    for (int i = 0; i < RA; ++i) gather_row(std::integral_constant<int, 1>{}, i);
#pragma unroll
    for (int i = 0; i < RB; ++i) issue_b_piece(1, 1, i);
    chunk_advance();
  }
#pragma unroll
  for (int i = 0; i < RA; ++i) combine_row(std::integral_constant<int, 0>{}, i, Abase);
  tap_step();
#pragma unroll
  for (int i = 0; i < RA; ++i) gather_row(std::integral_constant<int, 0>{}, i);
#pragma unroll
  for (int i = 0; i < RB; ++i) issue_b_piece(RING, RING, i);
  chunk_advance();
  YMI_STAMP(3);                 // prologue requests issued, chunk 0 combined (its loads have arrived)
  YMI_WAIT_VM(N_STEADY);        // the filter DMA of chunk 0 has RING * (NG + NB) younger operations behind it
  YMI_BARRIER();
  YMI_STAMP(4);                 // first barrier passed: the main loop starts

  // ---- main loop: RING steps per trip (the ring slot is a compile-time index) -----------------------------------------------
  // (RING == 2: both steps unconditionally inside the trip: with `if (st + 1 < nk)` around the second one the CFG has a path from
  // the first step straight back to itself, and the compiler's vmcnt for the ring registers drops from ~16 to 3 — seen in the ISA)
  const int nk = my_nk;
  if constexpr (RING == 2) {
    int st = 0;
    for (; st + 1 < nk; st += 2) {
      step(st, std::integral_constant<int, 1>{});
      step(st + 1, std::integral_constant<int, 0>{});
    }
    if (st < nk) step(st, std::integral_constant<int, 1>{});      // odd number of chunks (Cin = 32 * odd)
  } else {
    for (int st = 0; st < nk; ++st) step(st, std::integral_constant<int, 0>{});
  }
  YMI_STAMP(5);                 // main loop done
  YMI_WAIT_VM(0);               // the run-ahead filter DMAs target LDS the epilogue is about to reuse
  YMI_BARRIER();
  YMI_STAMP(6);
#undef YMI_WAIT_VM
#undef YMI_BARRIER

  // ---- epilogue: accumulators -> LDS tile -> 16-byte stores (the fast path of conv_igemm.hip) -------------------------------
  // folded-BN scale (times the filter row's inverse scale) and bias: requested before the transposition, consumed (scale * the
  // exact power of two 1 / sA) before the row loop — a pending load inside the per-row branches would make the compiler wait
  // vmcnt(0), i.e. for the previous row's STORE, in every row
  f32x4 sc = {1.f, 1.f, 1.f, 1.f}, bi = {0.f, 0.f, 0.f, 0.f};
  if (n < p.Cout) {                                     // Cout % 4 == 0 (host check): the four channels exist together
    if (((((uintptr_t)p.scale_h2) | ((uintptr_t)p.bias)) & 15) == 0) {
      sc = *reinterpret_cast<const f32x4 *>(p.scale_h2 + n);
      if (p.bias) bi = *reinterpret_cast<const f32x4 *>(p.bias + n);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) { sc[e] = p.scale_h2[n + e]; if (p.bias) bi[e] = p.bias[n + e]; }
    }
  }
  // PLAIN: the residual rows of this thread (bottleneck shortcut), requested before the transposition as well
  f32x4 rv[PLAIN ? RPT : 1];
  const bool has_res = PLAIN && p.res != nullptr;
  if constexpr (PLAIN) {
    if (has_res && n < p.Cout) {
#pragma unroll
      for (int i = 0; i < RPT; ++i) {
        const int m = m0 + rbase + RSTEP * i;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        rv[i] = m < p.M ? *reinterpret_cast<const f32x4 *>(p.res + (size_t)m * p.res_ld + n) : z;
      }
    }
  }
  float *es = lds;
  {
    const int ncol = lane & 31, half = lane >> 5;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          es[((wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * ELD + (wn * TN + j) * 32 + ncol] = acc[i][j][r];
  }
  sc = sc * invA;               // exact (a power of two); (v * invA) * sc == v * (invA * sc)
  __syncthreads();
  YMI_STAMP(7);                 // accumulators transposed through LDS (and scale / bias / residual loads back)
  const float slope = p.act == YMI_ACT_RELU ? 0.f : (p.act == YMI_ACT_LEAKY01 ? 0.1f : 1.f);
  float am = 0.f;
  f32x4 o[RPT];
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    f32x4 v = *reinterpret_cast<const f32x4 *>(es + (rbase + RSTEP * i) * ELD + 4 * c4);
    v = v * sc + bi;
    if constexpr (PLAIN) { if (has_res && !p.res_after_act) v += rv[i]; }
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], slope * v[e]);
    if constexpr (PLAIN) { if (has_res && p.res_after_act) v += rv[i]; }
    o[i] = v;
  }
  if (n < p.Cout) {
    float *base = p.y + (size_t)blockIdx.y * p.y_gs + (size_t)(m0 + rbase) * p.ldy + n;
#pragma unroll
    for (int i = 0; i < RPT; ++i)
      if (m0 + rbase + RSTEP * i < p.M) {
        am = fmaxf(am, ymi_absmax4(o[i]));
        *reinterpret_cast<f32x4 *>(base + (size_t)(RSTEP * i) * p.ldy) = o[i];
      }
  }
  if (p.y_amax) ymi_amax_finish(apre, am);
#ifdef YMI_DIAGNOSTICS
  if (tracing) {
    YMI_STAMP(8);               // stores and the bound's atomic issued
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    YMI_STAMP(9);               // ... and acknowledged
    if (t == 0) {
      unsigned long long *o = p.trace + 16 * (size_t)(blockIdx.x + gridDim.x * blockIdx.y);
#pragma unroll
      for (int i = 0; i < 10; ++i) o[i] = tr_[i];
      o[10] = tr_rt0;
      o[11] = __builtin_amdgcn_s_memrealtime();
      o[12] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);                                   // HW_REG_HW_ID
      o[13] = (unsigned long long)(__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u);                            // XCC_ID
      o[14] = (unsigned long long)my_nk;
      o[15] = 1;
    }
  }
#endif
#undef YMI_STAMP
#endif
}

template <int WM, int WN, int TM, int TN, int RING, bool PLAIN>
int launch_dcn_k(DcnParams p, hipStream_t s) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  p.tiles_n = (p.Cout + BN - 1) / BN;
  const int grid = ((p.M + BM - 1) / BM) * p.tiles_n;
  // The workgroup dispatcher does not balance a grid that fits in one residency round: it packs up to `occupancy` blocks on a CU
  // while others hold none (csrc/conv_igemm.hip launch_cfg).  When the grid is at most occ * 256 blocks, cap the residency at
  // k = ceil(grid / 256) blocks per CU by padding the block's LDS allocation with unused dynamic LDS.
  int dyn = 0;
  {
    constexpr int LDS_PER_CU = 160 * 1024, static_lds = dcn_lds_floats<WM, WN, TM, TN, RING>() * 4;
    const int occ = dcn_occupancy<WM, WN, TM, TN, RING, PLAIN>(), k = (grid * ((p.nk + p.nk_split - 1) / p.nk_split) + 255) / 256;
#ifdef YMI_DIAGNOSTICS
    const bool cap_on = !(p.abl & 64);
#else
    const bool cap_on = true;
#endif
    if (k < occ && cap_on) {
      const int want = LDS_PER_CU / (k + 1) + 1024;
      if (want > static_lds && want <= LDS_PER_CU / k) dyn = want - static_lds;
    }
  }
  const int splits = (p.nk + p.nk_split - 1) / p.nk_split;
  hipLaunchKernelGGL((pipe_h2_k<WM, WN, TM, TN, RING, PLAIN>), dim3(grid, splits), dim3(64 * WM * WN), dyn, s, p);
  return ymi_launch_status();
}

template <int WM, int WN, int TM, int TN, int RING = 2>
int launch_dcn(const DcnParams &p, bool plain, hipStream_t s) {
  return plain ? launch_dcn_k<WM, WN, TM, TN, RING, true>(p, s) : launch_dcn_k<WM, WN, TM, TN, RING, false>(p, s);
}

}  // namespace

// internal (csrc/conv_igemm.hip): second pass of a split-K launch — y = act(scale * sum_g part[g] + bias (+ res)), partials added in a
// fixed order
int ymi_internal_splitk_fixup(const float *part, long gstride, int S, long M, int Cout, int ldy, float *y, const float *scale,
                              const float *bias, const float *res, int res_ld, int act, int res_after_act, float *y_amax,
                              hipStream_t s);

namespace {

// Shared host side of the two entries below.  `offmask` != nullptr: DCNv2; nullptr: an ordinary convolution (PLAIN).
// desc->split_k = S > 1 (with split_ws: S * M * Cout floats): the K reduction is cut into S chunk-aligned ranges computed by S times as
// many blocks — the small maps (35x35, 18x18 at batch 8) have too few row tiles for 256 CUs once a tile covers 256 output channels
// (every sample gathered once for all of them) — and a second launch adds the partial sums in a fixed order and applies scale / bias /
// residual / activation (splitk_fixup_k of csrc/conv_igemm.hip: deterministic).
int run_pipe(const ymi_conv_desc *d, const float *offmask, int ldo, int mask_is_prob, int om_layout, int base_tile, int prof_kind,
             hipStream_t s) {
  const bool plain = offmask == nullptr;
  const ymi_conv_seg &g0 = d->seg[0];
  const long HoWo = (long)d->Ho * d->Wo, M = (long)d->B * HoWo;
  if (plain) {
    if (!((d->kh == 3 && d->kw == 3 && d->pad == 1) || (d->kh == 1 && d->kw == 1 && d->pad == 0))) return YMI_EARG;
  } else if (d->kh != 3 || d->kw != 3 || d->pad != 1) return YMI_EARG;
  if (d->Cin % 32 != 0 || d->Kpad != d->kh * d->kw * d->Cin) return YMI_EARG;
  if (d->nseg != 1 || g0.n0 != 0 || g0.n1 < d->Cout || (d->Cout & 3) || (g0.row_stride & 3) || (((uintptr_t)g0.ptr) & 15) ||
      g0.batch_stride != HoWo * g0.row_stride || g0.act > YMI_ACT_LEAKY01 || g0.act < 0)
    return YMI_EARG;
  if (d->res_mode != YMI_RES_NONE && (!plain || d->res_mode != YMI_RES_ADD || (d->res_ld & 3) || (((uintptr_t)d->res) & 15))) return YMI_EARG;
  if (!d->w_h2 || !d->scale_h2 || !d->x_amax || (((uintptr_t)d->w_h2) & 15)) return YMI_ENULL;
  if (M * (long)(plain ? 1 : ldo) >= (1L << 29) || M * (long)g0.row_stride >= (1L << 31)) return YMI_ESHAPE;
  const int S = d->split_k > 1 ? d->split_k : 1;
  const int nk = d->Kpad / BK;
  if (nk < 2) return YMI_EARG;
  if (S > 1) {
    // (ranges of ceil(nk / S) chunks, the last one shorter when S does not divide nk; every range non-empty)
    if (S > 16 || (nk + S - 1) / S < 2 || ((nk + S - 1) / S) * (S - 1) >= nk) return YMI_EARG;
    if (!d->split_ws || !d->winv_h2) return YMI_ENULL;
    if ((((uintptr_t)d->split_ws) & 15) || M * (long)d->Cout >= (1L << 29)) return YMI_ESHAPE;
  }
  DcnParams p;
  p.x = d->x; p.offmask = offmask; p.scale_h2 = d->scale_h2; p.bias = d->bias; p.x_amax = d->x_amax;
  p.w_h2 = d->w_h2; p.y = g0.ptr; p.y_amax = d->y_amax;
  p.B = d->B; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.ldx = d->ldx; p.Ho = d->Ho; p.Wo = d->Wo; p.Cout = d->Cout;
  p.stride = d->stride; p.Kpad = d->Kpad; p.ldo = ldo; p.ldy = g0.row_stride; p.act = g0.act; p.mask_is_prob = mask_is_prob; p.om_layout = om_layout;
  p.taps = d->kh * d->kw; p.kw = d->kw; p.pad = d->pad;
  p.res = d->res_mode == YMI_RES_ADD ? d->res : nullptr; p.res_ld = d->res_ld; p.res_after_act = d->res_after_act;
  p.M = (int)M; p.HoWo = (int)HoWo; p.tiles_n = 0; p.nk = nk; p.nk_split = (nk + S - 1) / S; p.y_gs = 0;
  p.x_bytes = (unsigned)((size_t)d->B * d->H * d->W * d->ldx * sizeof(float));
  p.om_bytes = plain ? 0u : (unsigned)((size_t)M * ldo * sizeof(float));
  p.w_plane = (unsigned)((((long)d->Cout + 127) / 128 * 128) * d->Kpad * 2L);
  p.abl = 0; p.trace = nullptr; p.trace_mode = 0;
#ifdef YMI_DIAGNOSTICS   // `make DIAG=1`: stall attribution for tools/dcn_probe.py — wrong results by design
  { const char *e = getenv("YMI_DCN_ABLATE"); p.abl = e ? atoi(e) : 0; }
  { const char *e = getenv("YMI_PIPE_TRACE"); p.trace = e ? (unsigned long long *)strtoull(e, nullptr, 0) : nullptr; }
  { const char *e = getenv("YMI_PIPE_TRACE_MODE"); p.trace_mode = e ? atoi(e) : 1; }
#endif
  if (S > 1) {           // partial launches undo the operand scales only (true partial sums), the second pass does the rest
    p.scale_h2 = d->winv_h2; p.bias = nullptr; p.act = YMI_ACT_NONE; p.y_amax = nullptr; p.res = nullptr;
    p.y = d->split_ws; p.ldy = d->Cout; p.y_gs = M * (long)d->Cout;
  }
  const int tile_id = base_tile | YMI_TILE_H2 | YMI_TILE_DCNP;
  const double flops = 2.0 * (double)M * (double)(d->cout_alg > 0 ? d->cout_alg : d->Cout) * (double)(d->kh * d->kw) *
                       (double)(d->cin_alg > 0 ? d->cin_alg : d->Cin);
  if (base_tile < YMI_DCNP_64x128 || base_tile > YMI_DCNP_64x32_W2) return YMI_EARG;
  if (base_tile >= YMI_DCNP_128x32_W4 && d->Cout > 32) return YMI_EARG;      // one column tile: nothing to gain from re-staging the rows per 32 columns
  int rc;
  const int pr = ymi_internal_prof_begin(flops, tile_id, prof_kind, s);
  switch (base_tile) {                                   // <waves along M, waves along N, 32x32 tiles per wave along M, along N, ring>
    case YMI_DCNP_64x128: rc = launch_dcn<2, 2, 1, 2>(p, plain, s); break;
    case YMI_DCNP_64x128_W8: rc = launch_dcn<2, 4, 1, 1>(p, plain, s); break;
    case YMI_DCNP_64x64: rc = launch_dcn<2, 2, 1, 1>(p, plain, s); break;
    case YMI_DCNP_128x128_W8: rc = launch_dcn<4, 2, 1, 2>(p, plain, s); break;
    case YMI_DCNP_128x64_W8: rc = launch_dcn<4, 2, 1, 1>(p, plain, s); break;
    case YMI_DCNP_32x128: rc = launch_dcn<1, 4, 1, 1>(p, plain, s); break;
    // one block per CU sized to M / 256 rows: the kernel is bound by the vector-memory pipe (a 1 KB wave load or DMA occupies it for
    // ~27 cycles whether it hits, misses or is out of bounds: profiles/r04_dcn_ablation.txt), so what counts is (i) every CU busy for
    // the whole launch — no second residency round, no CUs with twice the blocks of others — (ii) the filter DMAs (16 per chunk per
    // 128 columns whatever the row count) amortised over more rows, (iii) every sample gathered once for as many columns as possible
    case YMI_DCNP_96x128_W6: rc = launch_dcn<3, 2, 1, 2, 1>(p, plain, s); break;
    case YMI_DCNP_128x128_W8_R1: rc = launch_dcn<4, 2, 1, 2, 1>(p, plain, s); break;
    case YMI_DCNP_160x128_W10: rc = launch_dcn<5, 2, 1, 2, 1>(p, plain, s); break;
    case YMI_DCNP_192x128_W12: rc = launch_dcn<6, 2, 1, 2, 1>(p, plain, s); break;
    case YMI_DCNP_64x256_W8: rc = launch_dcn<2, 4, 1, 2, 1>(p, plain, s); break;
    case YMI_DCNP_96x256_W12: rc = launch_dcn<3, 4, 1, 2, 1>(p, plain, s); break;
    case YMI_DCNP_128x256_W16: rc = launch_dcn<4, 4, 1, 2, 1>(p, plain, s); break;
    // 64 x 64 wave tiles (a third fewer LDS fragment reads per MFMA, half the barriers per FLOP of a block of equal wave count):
    // ordinary convolutions only — four gathered rows per thread would not fit the DCN path's register ring
    case YMI_DCNP_128x256_W8T: rc = plain ? launch_dcn_k<2, 4, 2, 2, 1, true>(p, s) : YMI_EARG; break;
    case YMI_DCNP_128x128_W4T: rc = plain ? launch_dcn_k<2, 2, 2, 2, 1, true>(p, s) : YMI_EARG; break;
    case YMI_DCNP_256x128_W8T: rc = plain ? launch_dcn_k<4, 2, 2, 2, 1, true>(p, s) : YMI_EARG; break;
    // 32 columns: every wave a 32 x 32 tile of its own rows (Cout <= 32: the offset / mask convolution of a DCN layer)
    case YMI_DCNP_128x32_W4: rc = plain ? launch_dcn_k<4, 1, 1, 1, 2, true>(p, s) : YMI_EARG; break;
    case YMI_DCNP_256x32_W8: rc = plain ? launch_dcn_k<8, 1, 1, 1, 1, true>(p, s) : YMI_EARG; break;
    default: rc = plain ? launch_dcn_k<2, 1, 1, 1, 2, true>(p, s) : YMI_EARG; break;   // YMI_DCNP_64x32_W2
  }
  if (rc == YMI_OK && S > 1)
    rc = ymi_internal_splitk_fixup(d->split_ws, M * (long)d->Cout, S, M, d->Cout, g0.row_stride, g0.ptr, d->scale, d->bias,
                                   d->res_mode == YMI_RES_ADD ? d->res : nullptr, d->res_ld, g0.act, d->res_after_act, d->y_amax, s);
  ymi_internal_prof_end(pr, s);
  return rc;
}

}  // namespace

// internal (not part of the C ABI; called by ymi_dcn_v2_forward_f32 in csrc/conv_igemm.hip): the pipelined gather-GEMM for a
// validated descriptor whose tile is YMI_TILE_H2 | YMI_TILE_DCNP | YMI_DCNP_* (base_tile = the YMI_DCNP_* part).  YMI_EARG when the
// descriptor is outside what the kernel takes.  Profiling record kind 9.
int ymi_internal_dcn_h2(const ymi_dcn_desc *dd, int base_tile, hipStream_t s) {
  return run_pipe(&dd->conv, dd->offmask, dd->ldo, dd->mask_is_prob, dd->om_layout, base_tile, 9, s);
}

// internal (called by ymi_conv2d_nhwc_f32): the same pipeline as an ordinary 3x3 / pad 1 or 1x1 / pad 0 convolution.  Kind 10.
int ymi_internal_pipe_conv(const ymi_conv_desc *d, int base_tile, hipStream_t s) {
  return run_pipe(d, nullptr, 0, 0, 0, base_tile, 10, s);
}
