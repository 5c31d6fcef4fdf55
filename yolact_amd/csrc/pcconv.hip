// Ordinary 3x3 / pad 1 and 1x1 convolutions on the fp16x2 matrix path as a PRODUCER / CONSUMER gather-GEMM (round 6).
//
// Why (profiles/r06_pipe_phase_trace.txt, profiles/r04_pipe_ablation.txt): in pipe_h2_k (csrc/dcn.hip) every wave does
// everything — requests its rows, converts them to fp16 planes, issues its share of the filter DMAs, reads fragments, multiplies.
// 85 % of such a launch is the main loop, and a K chunk of the 96 x 128 tile takes 1 690 cycles where its MFMAs need 770 on the
// busiest SIMD and its 28 KB of global -> LDS traffic ~760 cycles of the CU's vector-memory pipe (one 1 KB wave load or DMA occupies
// it for ~27 cycles, hit, miss or out of bounds): the two ADD UP.  All waves run the same instruction stream in phase; a wave whose
// next instruction is a vector-memory request stalls at issue while the pipe is backed up, and the MFMAs behind that request wait
// with it (in-order issue).  Removing the MFMAs from the loop saves 13 %, removing the memory traffic 2 %: neither is "the" bound,
// their serialisation is.
//
// Here the two streams live in different waves.  A block is 8 waves: waves 0 .. 3 — one per SIMD — are CONSUMERS (ds_read_b128
// fragments + v_mfma_f32_32x32x16_f16 only, a 64 x 64 tile each; 256 x 128 / 128 x 256 blocks were measured in session r6c and dropped); waves 4 .. 7 — again one per SIMD — are PRODUCERS:
// they request the fp32 rows of chunk c + 3 into a two-slot register ring, split the rows of chunk c + 1 into the two fp16 planes of
// the fp16x2 arithmetic (tensor scale from x_amax, exactly as pipe_h2_k) and ds_write them, and issue the LDS-DMAs of the filter planes
// of chunk c + 2.  A producer that stalls on the memory pipe no longer holds any MFMA back, and the consumers' stream carries half an
// LDS read per MFMA instead of 12 other instructions.  One s_barrier per 32-deep K chunk joins the two sides (A planes double
// buffered, three filter stages); vmcnt is counted (one constant per step: every request past the end of K is an out-of-bounds buffer
// load).  K order, LDS images, fragment layout, MFMA order (h*l, l*h, h*h) and the epilogue are pipe_h2_k's, so the filter planes /
// scale_h2 of engine.Packed.h2() are used unchanged and the results of a layer agree with the pipelined tile's to the last bit
// (same products, same summation order per accumulator).
#include "common.h"
#include <stdlib.h>
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

typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
struct Split2 { f16x8 h, l; };

struct PcParams {
  const float *x, *scale_h2, *bias, *x_amax;
  const void *w_h2;
  float *y, *y_amax;
  int B, H, W, Cin, ldx, Ho, Wo, Cout, stride, Kpad, ldy, act;
  int taps, kw, pad;
  const float *res; int res_ld, res_after_act;
  int M, HoWo, tiles_n, nk;
  int nk_split;  // chunks per K range (gridDim.y ranges; == nk without split-K); range y writes raw partial sums to y + y * y_gs
  long y_gs;
  unsigned x_bytes, w_plane;
  int flags;     // env YMI_PC_FLAGS (A/B switches): bit0 consumers raise their priority (s_setprio 1) for the main loop; bit2 no residency cap; bit3 / bit4: ring depth 2 / 3 instead of 4 (128-column tiles)
  unsigned long long *trace;   // diagnostics build: phase stamps of wave 0 (consumer) and wave NCW (producer), 32 u64 per block
};

constexpr int NSA = 2;                  // LDS stages of the A planes (written by the producers)
constexpr int NPW = 4;                  // producer waves

// R = depth of the producers' register ring = how far requests run ahead: the rows AND the filter planes of chunk c + R + 1 are
// requested in step c and must have landed by the end of step c + R - 1 (rows: combined in step c + R) — R - 0.5 steps of cover for a
// memory round trip that takes 1 - 1.7 us under load (csrc/dcn.hip) against a step of 0.35 - 0.7 us; R + 2 filter stages.
template <int CM, int CN, int TM, int TN, int R>
constexpr int pc_lds_floats() {
  constexpr int BM = CM * TM * 32, BN = CN * TN * 32;
  constexpr int pipe = NSA * (2 * BM * 16) + (R + 2) * (2 * BN * 16), epi = BM * (BN + 4);
  return pipe > epi ? pipe : epi;
}

// CM x CN consumer waves (CM * CN == 4), each TM x TN MFMA tiles of 32 x 32
template <int CM, int CN, int TM, int TN, int R>
__global__ __launch_bounds__(64 * (CM * CN + NPW), 2)
void pc_conv_k(const PcParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BM = CM * TM * 32, BN = CN * TN * 32, NCW = CM * CN, NT = 64 * (NCW + NPW), NPT = 64 * NPW;
  static_assert(NCW == 4, "one consumer wave per SIMD");
  constexpr int RPP = NPT / 8, RA = BM / RPP;           // producers: 8 lanes (32 channels) per row, RA rows per thread per chunk
  constexpr int BUNITS = (2 * BN) / 16;                 // filter-plane DMA pieces of a chunk: (plane, 16-row group), 16 rows x 64 bytes each
  constexpr int RB = BUNITS / NPW;                      // pieces per producer wave per chunk
  static_assert(BM % RPP == 0 && RA >= 1 && BUNITS % NPW == 0, "tile vs producer passes");
  constexpr int A_STAGE = 2 * BM * 16, B_STAGE = 2 * BN * 16;    // floats: two fp16 planes of 64-byte rows
  constexpr int NVM = RA + RB;                          // vector-memory operations a producer issues per step
  constexpr int NSB = R + 2;                            // filter stages
  static_assert(R >= 2 && R <= 4, "ring depth");
  static_assert(R * NVM <= 63, "vmcnt is a 6-bit counter");
  constexpr int ELD = BN + 4;
  constexpr int LDS_FLOATS = pc_lds_floats<CM, CN, TM, TN, R>();
  __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
  float *const Abase = lds, *const Bbase = lds + NSA * A_STAGE;

  const int t = threadIdx.x;
  const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const bool producer = wave >= NCW;
#ifdef YMI_DIAGNOSTICS
  unsigned long long tr_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tr_rt0 = 0;
  const bool tracing = p.trace != nullptr;
#define PC_STAMP(i) do { if (tracing) tr_[i] = __builtin_amdgcn_s_memtime(); } while (0)
  if (tracing) tr_rt0 = __builtin_amdgcn_s_memrealtime();
  PC_STAMP(0);
#else
#define PC_STAMP(i) do { } while (0)
#endif

  const int logical = ymi_xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = logical % p.tiles_n, tile_m = logical / p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  float sA, invA;
  ymi_h2_scale(ymi_amax_read(p.x_amax), sA, invA);
  const ymi_amax_pre apre = ymi_amax_prefetch(p.y_amax);

  const int kc0 = blockIdx.y * p.nk_split;
  const int nk = (p.nk - kc0) < p.nk_split ? (p.nk - kc0) : p.nk_split;   // chunks of this block's K range

#define PC_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
#define PC_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (producer) {
    // =========================================== PRODUCERS ========================================================================
    const int pt = t - 64 * NCW, pw = wave - NCW;
    const int kq = pt & 7, r0 = pt >> 3;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.x, 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.w_h2, 0, (int)(2 * p.w_plane), 0x00020000);
    // rows r0 + RPP * i of the tile, channels 4 kq .. 4 kq + 3 of the chunk
    int g_iy0[RA], g_ix0[RA], g_ib[RA], a_st[RA];
    bool g_ok[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      const int row = r0 + RPP * i, m = m0 + row;
      a_st[i] = row * 64 + (((kq >> 1) ^ ((row >> 2) & 3)) * 16) + (kq & 1) * 8;
      g_ok[i] = m < p.M;
      const int mm = g_ok[i] ? m : 0;
      const int b = mm / p.HoWo, pix = mm - b * p.HoWo;
      const int oy = pix / p.Wo, ox = pix - oy * p.Wo;
      g_iy0[i] = oy * p.stride - p.pad;
      g_ix0[i] = ox * p.stride - p.pad;
      g_ib[i] = b * p.H * p.W;
    }
    unsigned b_off[RB];
    int b_lds[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const int u = pw + NPW * i;
      const int plane = u / (BN / 16), rg = u - plane * (BN / 16);
      const int row = rg * 16 + (lane >> 2), lsl = (lane & 3) ^ ((row >> 2) & 3);
      b_off[i] = (unsigned)plane * p.w_plane + (unsigned)(((n0 + row) * p.Kpad + 8 * lsl) * 2);
      b_lds[i] = plane * (BN * 16) + rg * 256;
    }
    const int cpt = p.Cin / BK;                         // chunks per tap
    int g_tap = kc0 / cpt, g_c = (kc0 - g_tap * cpt) * BK;   // (tap, first channel) of the next chunk to request
    int g_left = nk;                                    // chunks of the range still to request
    unsigned gq[RA];                                    // byte offset of the current tap's pixel per row (OOB: padding / past M / past K)
    bool g_first = true;
    auto tap_step = [&]() {                             // at a tap boundary (or the start of a range inside a tap): the tap's pixels
      if (g_c == 0 || g_first) {
        g_first = false;
        const bool live = g_left > 0;
        const int ky = g_tap / p.kw, kx = g_tap - p.kw * ky;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
          const int h = g_iy0[i] + ky, w = g_ix0[i] + kx;
          const bool in = live && g_ok[i] && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
          gq[i] = in ? (unsigned)(((g_ib[i] + h * p.W + w) * p.ldx + 4 * kq) * 4) : OOB;
        }
      }
    };
    auto chunk_advance = [&]() {
      g_c += BK;
      if (g_c == p.Cin) { g_c = 0; ++g_tap; }
      if (--g_left == 0) {
#pragma unroll
        for (int i = 0; i < RA; ++i) gq[i] = OOB;
      }
    };
    f32x4 ring[R][RA];
    auto request_a = [&](auto slot_c) {                 // the rows of the chunk at (g_tap, g_c) -> ring slot
      constexpr int S = decltype(slot_c)::value;
      tap_step();
      const int so = g_c * 4;
#pragma unroll
      for (int i = 0; i < RA; ++i) ring[S][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, gq[i], so, 0));
      chunk_advance();
    };
    auto request_b = [&](int kc, int stage) {           // filter planes of chunk kc of the range -> stage (past the range: zeros)
      const bool live = kc < nk;
#pragma unroll
      for (int i = 0; i < RB; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_ptr_t)(Bbase + stage * B_STAGE + b_lds[i]), 16, live ? b_off[i] : OOB,
                                                 live ? (kc0 + kc) * (BK * 2) : 0, 0, 0);
    };
    auto combine = [&](auto slot_c, float *As) {        // ring slot -> two fp16 planes -> LDS
      constexpr int S = decltype(slot_c)::value;
#pragma unroll
      for (int i = 0; i < RA; ++i) {
        const f32x4 v = ring[S][i] * sA;
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
      }
    };
    // prologue — the issue order of the steady state ([B(c + R + 1), A(c + R + 1)] in step c) run for "steps -(R + 1) .. -1";
    // A(j) lives in ring slot j % R, B(j) in filter stage j % (R + 2)
    [&]<int... J>(std::integer_sequence<int, J...>) {
      ((request_b(J, J), request_a(std::integral_constant<int, J>{})), ...);          // B(j), A(j) for j = 0 .. R - 1
    }(std::make_integer_sequence<int, R>{});
    PC_WAIT_VM((R - 1) * NVM);                          // B(0), A(0) have landed
    combine(std::integral_constant<int, 0>{}, Abase);
    request_b(R, R);
    request_a(std::integral_constant<int, 0>{});        // A(R) -> slot 0
    PC_WAIT_VM((R - 1) * NVM);                          // B(1), A(1) have landed
    PC_STAMP(1);
    PC_BARRIER();
    PC_STAMP(2);
    // step c: combine A(c + 1) (landed: last step's wait), request B(c + R + 1) and A(c + R + 1), wait for B(c + 2) and A(c + 2)
    int bnx = R + 1;                                    // stage of chunk c + R + 1
    auto pstep = [&](int c, auto slot_c) {              // slot_c: ring slot of A(c + 1) = (c + 1) % R
      combine(slot_c, Abase + ((c + 1) & 1) * A_STAGE);
      request_b(c + R + 1, bnx);
      request_a(slot_c);
      bnx = bnx + 1 == NSB ? 0 : bnx + 1;
#ifdef YMI_DIAGNOSTICS
      if (tracing) {                                    // where a producer step goes: issue + conversion | memory wait | barrier wait
        const unsigned long long a = __builtin_amdgcn_s_memtime();
        PC_WAIT_VM((R - 1) * NVM);
        const unsigned long long b = __builtin_amdgcn_s_memtime();
        PC_BARRIER();
        const unsigned long long e = __builtin_amdgcn_s_memtime();
        tr_[5] += b - a; tr_[6] += e - b;
        return;
      }
#endif
      PC_WAIT_VM((R - 1) * NVM);
      PC_BARRIER();
    };
    int c = 0;
    for (; c + R <= nk; c += R) {                       // c % R == 0 here: step c + i combines slot (i + 1) % R
      [&]<int... I>(std::integer_sequence<int, I...>) {
        (pstep(c + I, std::integral_constant<int, (I + 1) % R>{}), ...);
      }(std::make_integer_sequence<int, R>{});
    }
    [&]<int... I>(std::integer_sequence<int, I...>) {   // the last nk % R steps
      ((c + I < nk ? pstep(c + I, std::integral_constant<int, (I + 1) % R>{}) : (void)0), ...);
    }(std::make_integer_sequence<int, R - 1>{});
    PC_STAMP(3);
  } else {
    // =========================================== CONSUMERS ========================================================================
    const int wm = wave / CN, wn = wave % CN;
    // fragments: plane images of 64-byte rows, 16-byte slot s of row r at s ^ ((r >> 2) & 3); lane half h of step s2 holds
    // k = 16 s2 + 8 h .. + 7 of row lane & 31
    const int psw = ((lane & 31) >> 2) & 3, hh_ = lane >> 5;
    const int fro[2] = {(lane & 31) * 16 + 4 * ((0 + hh_) ^ psw), (lane & 31) * 16 + 4 * ((2 + hh_) ^ psw)};
    Split2 pa[2][TM], pb[2][TN];
    auto load_frag = [&](const float *As, const float *Bs, auto s2c) {
      constexpr int s2 = decltype(s2c)::value;
      const float *Ap = As + (wm * TM * 32) * 16 + fro[s2];
      const float *Bp = Bs + (wn * TN * 32) * 16 + fro[s2];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        pa[s2][i].h = *reinterpret_cast<const f16x8 *>(Ap + i * 32 * 16);
        pa[s2][i].l = *reinterpret_cast<const f16x8 *>(Ap + i * 32 * 16 + BM * 16);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        pb[s2][j].h = *reinterpret_cast<const f16x8 *>(Bp + j * 32 * 16);
        pb[s2][j].l = *reinterpret_cast<const f16x8 *>(Bp + j * 32 * 16 + BN * 16);
      }
    };
    auto mfmas = [&](auto s2c) {
      constexpr int s2 = decltype(s2c)::value;
#pragma unroll
      for (int pr = 0; pr < 3; ++pr)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const f16x8 fa_ = pr == 1 ? pa[s2][i].l : pa[s2][i].h;
            const f16x8 fb_ = pr == 0 ? pb[s2][j].l : pb[s2][j].h;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa_, fb_, acc[i][j], 0, 0, 0);
          }
    };
    using K0 = std::integral_constant<int, 0>;
    using K1 = std::integral_constant<int, 1>;
    PC_STAMP(1);
    PC_BARRIER();                                       // chunk 0: A planes written, filter planes landed
    PC_STAMP(2);
    if (p.flags & 1) __builtin_amdgcn_s_setprio(1);
    int bst = 0;
    for (int c = 0; c < nk; ++c) {
      const float *As = Abase + (c & 1) * A_STAGE, *Bs = Bbase + bst * B_STAGE;
      load_frag(As, Bs, K0{});
      load_frag(As, Bs, K1{});
      mfmas(K0{});
      mfmas(K1{});
      bst = bst + 1 == NSB ? 0 : bst + 1;
#ifdef YMI_DIAGNOSTICS
      if (tracing) {                                    // time a consumer spends at the step barrier (= waiting for the producers)
        const unsigned long long a = __builtin_amdgcn_s_memtime();
        PC_BARRIER();
        tr_[6] += __builtin_amdgcn_s_memtime() - a;
        continue;
      }
#endif
      PC_BARRIER();
    }
    if (p.flags & 1) __builtin_amdgcn_s_setprio(0);
    PC_STAMP(3);
  }
  PC_WAIT_VM(0);                // (producers: the run-ahead filter DMAs target LDS the epilogue is about to reuse)
  PC_BARRIER();
  PC_STAMP(4);
#undef PC_WAIT_VM
#undef PC_BARRIER

  // ---- epilogue: accumulators -> LDS tile -> 16-byte stores by all eight waves (pipe_h2_k's) -------------------------------------
  constexpr int C4 = BN / 4, RSTEP = NT / C4, RPT = BM / RSTEP;
  static_assert(BM % RSTEP == 0, "epilogue rows");
  const int c4 = t % C4, rbase = t / C4;
  const int n = n0 + 4 * c4;
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
  f32x4 rv[RPT];
  const bool has_res = p.res != nullptr;
  if (has_res && n < p.Cout) {
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const int m = m0 + rbase + RSTEP * i;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      rv[i] = m < p.M ? *reinterpret_cast<const f32x4 *>(p.res + (size_t)m * p.res_ld + n) : z;
    }
  }
  float *es = lds;
  if (!producer) {
    const int wm = wave / CN, wn = wave % CN;
    const int ncol = lane & 31, half = lane >> 5;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          es[((wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * ELD + (wn * TN + j) * 32 + ncol] = acc[i][j][r];
  }
  sc = sc * invA;               // exact (a power of two)
  __syncthreads();
  const float slope = p.act == YMI_ACT_RELU ? 0.f : (p.act == YMI_ACT_LEAKY01 ? 0.1f : 1.f);
  float am = 0.f;
  f32x4 o[RPT];
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    f32x4 v = *reinterpret_cast<const f32x4 *>(es + (rbase + RSTEP * i) * ELD + 4 * c4);
    v = v * sc + bi;
    if (has_res && !p.res_after_act) v += rv[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], slope * v[e]);
    if (has_res && p.res_after_act) v += rv[i];
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
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PC_STAMP(7);
    if (lane == 0 && (wave == 0 || wave == NCW)) {
      unsigned long long *o_ = p.trace + 32 * (size_t)(blockIdx.x + gridDim.x * blockIdx.y) + (wave == 0 ? 0 : 16);
#pragma unroll
      for (int i = 0; i < 8; ++i) o_[i] = tr_[i];
      o_[10] = tr_rt0;
      o_[11] = __builtin_amdgcn_s_memrealtime();
      o_[12] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
      o_[13] = (unsigned long long)(__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u);
      o_[14] = (unsigned long long)nk;
      o_[15] = 1;
    }
  }
#endif
#undef PC_STAMP
#endif
}

template <int CM, int CN, int TM, int TN, int R>
int launch_pc(PcParams p, hipStream_t s) {
  constexpr int BM = CM * TM * 32, BN = CN * TN * 32;
  p.tiles_n = (p.Cout + BN - 1) / BN;
  const int grid = ((p.M + BM - 1) / BM) * p.tiles_n;
  const int splits = (p.nk + p.nk_split - 1) / p.nk_split;
  // The dispatcher does not balance a grid that fits in one residency round (csrc/dcn.hip launch_dcn_k): when the whole grid is at
  // most 256 k blocks, cap the residency at k blocks per CU by padding the block's LDS allocation with unused dynamic LDS.
  int dyn = 0;
  {
    constexpr int LDS_PER_CU = 160 * 1024, static_lds = pc_lds_floats<CM, CN, TM, TN, R>() * 4;
    const int occ = LDS_PER_CU / static_lds, k = (grid * splits + 255) / 256;
    if (k < occ && !(p.flags & 4)) {
      const int want = LDS_PER_CU / (k + 1) + 1024;
      if (want > static_lds && want <= LDS_PER_CU / k) dyn = want - static_lds;
    }
  }
  hipLaunchKernelGGL((pc_conv_k<CM, CN, TM, TN, R>), dim3(grid, splits), dim3(64 * (CM * CN + NPW)), dyn, s, p);
  return ymi_launch_status();
}

}  // namespace

// internal (called by ymi_conv2d_nhwc_f32 for tile YMI_TILE_H2 | YMI_TILE_DCNP | YMI_DCNP_PC_*): the producer / consumer kernel for a
// validated descriptor (3x3 / pad 1 or 1x1 / pad 0, any stride, Cin % 32 == 0, one dense output, optional residual, optional split-K
// with the deterministic second pass of csrc/conv_igemm.hip).  Profiling record kind 14.
int ymi_internal_pc_conv(const ymi_conv_desc *d, int base_tile, hipStream_t s) {
  const ymi_conv_seg &g0 = d->seg[0];
  const long HoWo = (long)d->Ho * d->Wo, M = (long)d->B * HoWo;
  if (!((d->kh == 3 && d->kw == 3 && d->pad == 1) || (d->kh == 1 && d->kw == 1 && d->pad == 0))) return YMI_EARG;
  if (d->Cin % 32 != 0 || d->Kpad != d->kh * d->kw * d->Cin) return YMI_EARG;
  if (d->nseg != 1 || g0.n0 != 0 || g0.n1 < d->Cout || (d->Cout & 3) || (g0.row_stride & 3) || (((uintptr_t)g0.ptr) & 15) ||
      g0.batch_stride != HoWo * g0.row_stride || g0.act > YMI_ACT_LEAKY01 || g0.act < 0)
    return YMI_EARG;
  if (d->res_mode != YMI_RES_NONE && (d->res_mode != YMI_RES_ADD || (d->res_ld & 3) || (((uintptr_t)d->res) & 15))) return YMI_EARG;
  if (!d->w_h2 || !d->scale_h2 || !d->x_amax || (((uintptr_t)d->w_h2) & 15)) return YMI_ENULL;
  if (M * (long)g0.row_stride >= (1L << 31) || (long)d->B * d->H * d->W * d->ldx >= (1L << 29)) return YMI_ESHAPE;
  const int S = d->split_k > 1 ? d->split_k : 1;
  const int nk = d->Kpad / BK;
  if (nk < 2) return YMI_EARG;
  if (S > 1) {
    if (S > 16 || (nk + S - 1) / S < 2 || ((nk + S - 1) / S) * (S - 1) >= nk) return YMI_EARG;
    if (!d->split_ws || !d->winv_h2) return YMI_ENULL;
    if ((((uintptr_t)d->split_ws) & 15) || M * (long)d->Cout >= (1L << 29)) return YMI_ESHAPE;
  }
  PcParams p;
  p.x = d->x; p.scale_h2 = d->scale_h2; p.bias = d->bias; p.x_amax = d->x_amax;
  p.w_h2 = d->w_h2; p.y = g0.ptr; p.y_amax = d->y_amax;
  p.B = d->B; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.ldx = d->ldx; p.Ho = d->Ho; p.Wo = d->Wo; p.Cout = d->Cout;
  p.stride = d->stride; p.Kpad = d->Kpad; p.ldy = g0.row_stride; p.act = g0.act;
  p.taps = d->kh * d->kw; p.kw = d->kw; p.pad = d->pad;
  p.res = d->res_mode == YMI_RES_ADD ? d->res : nullptr; p.res_ld = d->res_ld; p.res_after_act = d->res_after_act;
  p.M = (int)M; p.HoWo = (int)HoWo; p.tiles_n = 0; p.nk = nk; p.nk_split = (nk + S - 1) / S; p.y_gs = 0;
  p.x_bytes = (unsigned)((size_t)d->B * d->H * d->W * d->ldx * sizeof(float));
  p.w_plane = (unsigned)((((long)d->Cout + 127) / 128 * 128) * d->Kpad * 2L);
  static const int flags = [] { const char *e = getenv("YMI_PC_FLAGS"); return e ? atoi(e) : 0; }();
  p.flags = flags;
  p.trace = nullptr;
#ifdef YMI_DIAGNOSTICS
  { const char *e = getenv("YMI_PIPE_TRACE"); p.trace = e ? (unsigned long long *)strtoull(e, nullptr, 0) : nullptr; }
#endif
  if (S > 1) {           // partial launches undo the operand scales only (true partial sums), the second pass does the rest
    p.scale_h2 = d->winv_h2; p.bias = nullptr; p.act = YMI_ACT_NONE; p.y_amax = nullptr; p.res = nullptr;
    p.y = d->split_ws; p.ldy = d->Cout; p.y_gs = M * (long)d->Cout;
  }
  const double flops = 2.0 * (double)M * (double)(d->cout_alg > 0 ? d->cout_alg : d->Cout) * (double)(d->kh * d->kw) *
                       (double)(d->cin_alg > 0 ? d->cin_alg : d->Cin);
  int rc;
  const int pr = ymi_internal_prof_begin(flops, base_tile | YMI_TILE_H2 | YMI_TILE_DCNP, 14, s);
  switch (base_tile) {                                   // <consumer waves along M, along N, 32x32 tiles per consumer along M, along N>
    case YMI_DCNP_PC_128x128:
      if (p.flags & 32) { rc = (p.flags & 8) ? launch_pc<1, 4, 1, 1, 2>(p, s) : (p.flags & 16) ? launch_pc<1, 4, 1, 1, 3>(p, s) : launch_pc<1, 4, 1, 1, 4>(p, s); break; }   // experiment (session r6q): a 32 x 128 block behind the same id
      if (p.flags & 64) { rc = (p.flags & 8) ? launch_pc<2, 2, 1, 2, 2>(p, s) : launch_pc<2, 2, 1, 2, 4>(p, s); break; }                                                      // ... and 64 x 128
      rc = (p.flags & 8) ? launch_pc<2, 2, 2, 2, 2>(p, s) : (p.flags & 16) ? launch_pc<2, 2, 2, 2, 3>(p, s) : launch_pc<2, 2, 2, 2, 4>(p, s); break;
    default: rc = YMI_EARG; break;
  }
  if (rc == YMI_OK && S > 1)
    rc = ymi_internal_splitk_fixup(d->split_ws, M * (long)d->Cout, S, M, d->Cout, g0.row_stride, g0.ptr, d->scale, d->bias,
                                   d->res_mode == YMI_RES_ADD ? d->res : nullptr, d->res_ld, g0.act, d->res_after_act, d->y_amax, s);
  ymi_internal_prof_end(pr, s);
  return rc;
}
