// The grouped GEMM of the Winograd path (M_g = V_g U_g^T for the 16 / 36 components g, csrc/winograd.hip) as ONE persistent
// producer / consumer launch — round 6, for the layers whose GEMM is the step's dominant kernel (proto_net, fpn.pred, head.upfeature:
// yolact.py:579-605, 133-212; profiles/r05_kernel_stats_*: conv_igemm_f32<128x128h2, winograd grouped GEMM> 0.64 ms per step).
//
// What the 128 x 128 tile of csrc/conv_igemm.hip spends its time on there (K = C = 256: eight 32-deep chunks per block, 5 544 blocks for
// proto_net's last 3x3): a prologue and an epilogue per 2.9 us of MFMAs, the V rows of a tile fetched from beyond L2 once per COLUMN tile
// (~9.5 B/clk/CU) and 32 KB through the global -> LDS path per 770 cycles of MFMAs.  This kernel:
//   * 128 rows x 256 columns per work item: V crosses the path ONCE for all of a layer's 256 output channels, and a chunk carries
//     48 KB for 1 536 cycles of MFMAs instead of 32 KB for 768;
//   * persistent blocks (one per CU) walk the items (component, row tile, column block) round robin; the chunk stream does not stop at an
//     item boundary: the producers request chunk c + 2 — of this item or the next — while chunk c is multiplied, so an item has no
//     prologue of its own;
//   * four PRODUCER waves only issue LDS-DMAs (both operands arrive as fp16 planes: V from the input transform, U from the pack: there is
//     nothing to convert), 12 pieces of 1 KB per wave per chunk, + one piece per item with the item's 256 inverse filter scales; four
//     CONSUMER waves (one per SIMD, 2 x 4 MFMA tiles each, 128 accumulator registers) read fragments and multiply; one s_barrier per chunk;
//   * orientation U V^T: a lane ends with 4 x 4 consecutive columns of ONE row of M: float4 stores straight from the accumulators.
// Same products and the same K order as the tile it replaces (h*l, l*h, h*h per 16-deep step, chunks ascending): M is bit-identical.
#include "common.h"
#include <stdlib.h>
#include <type_traits>
#include "../../include/yolact_amd.h"

int ymi_internal_prof_begin(double flops, int tile, int kind, hipStream_t s);
void ymi_internal_prof_end(int idx, hipStream_t s);

namespace {

typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0x80000000u;

constexpr int BM = 128, BN = 256;
constexpr int A_BYTES = 2 * BM * 64, B_BYTES = 2 * BN * 64, STAGE = A_BYTES + B_BYTES, NST = 3;     // 16 KB + 32 KB per 32-deep chunk
constexpr int OFF_SC = NST * STAGE, WG_LDS = OFF_SC + 2 * BN * 4;
constexpr int NPC = 12;                    // DMA pieces per producer wave per chunk: 4 of V, 8 of U

struct WgParams {
  const char *v, *u;                       // V planes [G][2][T][C] fp16, U planes [G][2][CoutPad][C] fp16
  const float *uinv, *x_amax;              // [G][cout_pad] inverse filter-row scales; the input tensor's magnitude-bound slot
  float *m;                                // M [G][T][Ng] fp32
  long v_gs, u_gs, m_gs;                   // bytes / bytes / floats between components
  unsigned v_plane, u_plane;               // bytes between the two planes
  int G, T, C, Ng, cout_pad, tiles_m, tiles_n, nitems;
  float amax_mul;
  unsigned long long *trace;
  int abl;                                 // diagnostics build (env YMI_WGEMM_ABLATE): bit0 no M stores, bit1 V requests out of bounds (no access), bit2 U requests
                                           // out of bounds, bit3 no MFMAs, bit4 non-temporal M stores — wrong results by design (except bit4)
};

__global__ __launch_bounds__(512, 2) void wgemm_k(const WgParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  __shared__ __attribute__((aligned(16))) char lds[WG_LDS];
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const bool producer = wave >= 4;
  const int lr = lane & 31, hh = lane >> 5;
  const int nk = p.C >> 5;
  const int nb = (int)gridDim.x, b0 = (int)blockIdx.x;
  const int my_items = (p.nitems - b0 + nb - 1) / nb;                     // items b0, b0 + nb, ...
  const int nsteps = my_items * nk;
#ifdef YMI_DIAGNOSTICS
  unsigned long long tr_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const bool tracing = p.trace != nullptr;
  tr_[0] = __builtin_amdgcn_s_memtime();
#endif
#define WG_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
#define WG_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
  // item -> (component, row tile, column block): the column blocks of a row tile are consecutive items (they share V)
  auto decode = [&](int it, int &g, int &tm, int &tn) {
    tn = it % p.tiles_n;
    const int r = it / p.tiles_n;
    tm = r % p.tiles_m;
    g = r / p.tiles_m;
  };

  if (producer) {
    // =========================================== PRODUCERS: LDS-DMA only ========================================================
    const int pw = wave - 4;
    // piece pw + 4 i: i < 4 -> V piece (plane, 16-row group of 128 rows); i >= 4 -> U piece (plane, 16-row group of 256 rows)
    int a_row[4], a_dst[4], b_row[8], b_dst[8];
    unsigned a_ko[4], b_ko[8];             // byte offset inside a row: the lane's 16-byte k slot (swizzled) of the chunk
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = pw + 4 * i, plane = q >> 3, rg = q & 7;
      const int row = rg * 16 + (lane >> 2), lsl = (lane & 3) ^ ((row >> 2) & 3);
      a_row[i] = row; a_ko[i] = (unsigned)plane * p.v_plane + 16u * lsl; a_dst[i] = plane * (BM * 64) + rg * 1024;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int q = pw + 4 * i, plane = q >> 4, rg = q & 15;
      const int row = rg * 16 + (lane >> 2), lsl = (lane & 3) ^ ((row >> 2) & 3);
      b_row[i] = row; b_ko[i] = (unsigned)plane * p.u_plane + 16u * lsl; b_dst[i] = A_BYTES + plane * (BN * 64) + rg * 1024;
    }
    // request cursor: chunk (ritem, rkc); per item: buffer resources of the component, this lane's row offsets
    int ridx = 0, rkc = 0;                 // index among MY items, chunk of it
    __amdgpu_buffer_rsrc_t vrs, urs, srs;
    unsigned a_vo[4], b_vo[8];
    int sc_par = 0;
    auto open_item = [&](int idx) {
      const bool live = idx < my_items;
      int g = 0, tm = 0, tn = 0;
      decode(live ? b0 + idx * nb : 0, g, tm, tn);
      vrs = __builtin_amdgcn_make_buffer_rsrc((void *)(p.v + (size_t)g * p.v_gs), 0, live ? (int)p.v_gs : 0, 0x00020000);
      urs = __builtin_amdgcn_make_buffer_rsrc((void *)(p.u + (size_t)g * p.u_gs), 0, live ? (int)p.u_gs : 0, 0x00020000);
      srs = __builtin_amdgcn_make_buffer_rsrc((void *)(p.uinv + (size_t)g * p.cout_pad), 0, live ? p.cout_pad * 4 : 0, 0x00020000);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = tm * BM + a_row[i];
        a_vo[i] = (live && row < p.T && !(p.abl & 2)) ? a_ko[i] + (unsigned)(row * p.C) * 2u : OOB;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = tn * BN + b_row[i];
        b_vo[i] = (live && row < p.cout_pad && !(p.abl & 4)) ? b_ko[i] + (unsigned)(row * p.C) * 2u : OOB;
      }
      // the item's 256 inverse filter scales: ONE 1 KB piece (every producer wave issues it — same bytes, same place — so that the
      // waves' vmcnt stay in step); columns past cout_pad: zeros
      const int n = tn * BN + 4 * lane;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srs, (lds_ptr_t)(lds + OFF_SC + sc_par * (BN * 4)), 16, (live && n < p.cout_pad) ? (unsigned)n * 4u : OOB, 0, 0, 0);
      sc_par ^= 1;
    };
    auto request = [&](int stage) {        // the chunk at the cursor -> ring stage; then advance the cursor
      if (rkc == 0) open_item(ridx);
      char *dst = lds + stage * STAGE;
      const unsigned so = (unsigned)rkc * 64u;
#pragma unroll
      for (int i = 0; i < 4; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(vrs, (lds_ptr_t)(dst + a_dst[i]), 16, a_vo[i], so, 0, 0);
#pragma unroll
      for (int i = 0; i < 8; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(urs, (lds_ptr_t)(dst + b_dst[i]), 16, b_vo[i], so, 0, 0);
      if (++rkc == nk) { rkc = 0; ++ridx; }
    };
    request(0);
    request(1);
    WG_WAIT_VM(NPC);                       // chunk 0 (and its item's scales) landed
#ifdef YMI_DIAGNOSTICS
    tr_[1] = __builtin_amdgcn_s_memtime();
#endif
    WG_BARRIER();
    int st = 2;
    for (int s = 0; s < nsteps; ++s) {
      request(st);                         // chunk s + 2 into the stage freed by the last barrier (past the last item: out-of-bounds pieces)
      st = st == 2 ? 0 : st + 1;
#ifdef YMI_DIAGNOSTICS
      const unsigned long long a_ = tracing ? __builtin_amdgcn_s_memtime() : 0ull;
#endif
      WG_WAIT_VM(NPC);                     // chunk s + 1 landed (a scale piece, if any, is older than the 12 pieces that may remain)
#ifdef YMI_DIAGNOSTICS
      const unsigned long long b_ = tracing ? __builtin_amdgcn_s_memtime() : 0ull;
#endif
      WG_BARRIER();
#ifdef YMI_DIAGNOSTICS
      if (tracing) { tr_[2] += b_ - a_; tr_[3] += __builtin_amdgcn_s_memtime() - b_; }
#endif
    }
    WG_WAIT_VM(0);
  } else {
    // =========================================== CONSUMERS ========================================================================
    const int wc = wave & 1, wp = wave >> 1;                              // column half (4 tiles of 32), row half (2 tiles of 32)
    const int psw = (lr >> 2) & 3;
    float sA, invA;
    ymi_h2_scale(ymi_amax_read(p.x_amax) * p.amax_mul, sA, invA);
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // this lane's fragment offsets inside a stage: U rows 32 (4 wc + i) + lr, V rows 32 (2 wp + j) + lr; k step s2 -> slot 2 s2 + hh
    const int uo0 = A_BYTES + (32 * (4 * wc) + lr) * 64 + 16 * ((0 + hh) ^ psw), uo1 = A_BYTES + (32 * (4 * wc) + lr) * 64 + 16 * ((2 + hh) ^ psw);
    const int vo0 = (32 * (2 * wp) + lr) * 64 + 16 * ((0 + hh) ^ psw), vo1 = (32 * (2 * wp) + lr) * 64 + 16 * ((2 + hh) ^ psw);
#ifdef YMI_DIAGNOSTICS
    tr_[1] = __builtin_amdgcn_s_memtime();
#endif
    WG_BARRIER();
    __builtin_amdgcn_s_setprio(1);
    int st = 0, kc = 0, idx = 0, sc_par = 0;
    for (int s = 0; s < nsteps; ++s) {
      const char *sb = lds + st * STAGE;
      // groups = (k step, pair of column tiles): 12 MFMAs each; the fragments of group g + 1 are requested before the MFMAs of group g
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const char *ub = sb + (s2 ? uo1 : uo0), *vb = sb + (s2 ? vo1 : vo0);
        f16x8 vh[2], vl[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          vh[j] = *reinterpret_cast<const f16x8 *>(vb + j * 2048);
          vl[j] = *reinterpret_cast<const f16x8 *>(vb + BM * 64 + j * 2048);
        }
#pragma unroll
        for (int ip = 0; ip < 2; ++ip) {
          f16x8 uh[2], ul[2];
#pragma unroll
          for (int ii = 0; ii < 2; ++ii) {
            uh[ii] = *reinterpret_cast<const f16x8 *>(ub + (2 * ip + ii) * 2048);
            ul[ii] = *reinterpret_cast<const f16x8 *>(ub + BN * 64 + (2 * ip + ii) * 2048);
          }
#pragma unroll
          for (int pr = 0; pr < 3; ++pr)
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
              for (int j = 0; j < 2; ++j)
#ifdef YMI_DIAGNOSTICS
                if (!(p.abl & 8))
#endif
                acc[2 * ip + ii][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pr == 0 ? ul[ii] : uh[ii], pr == 1 ? vl[j] : vh[j], acc[2 * ip + ii][j], 0, 0, 0);
        }
      }
      st = st == 2 ? 0 : st + 1;
      if (++kc == nk) {
        // ---- the item is complete: scale, store (float4 = 4 consecutive columns of one row of M), clear ------------------------
        kc = 0;
        int g, tm, tn;
        decode(b0 + idx * nb, g, tm, tn);
        ++idx;
        const float *sc = reinterpret_cast<const float *>(lds + OFF_SC + sc_par * (BN * 4));
        sc_par ^= 1;
        float *mg = p.m + (size_t)g * p.m_gs;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int row = tm * BM + 32 * (2 * wp + j) + lr;
          const bool rok = row < p.T;
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
              const int nl = 32 * (4 * wc + i) + 8 * gq + 4 * hh, n = tn * BN + nl;
              const f32x4 s4 = *reinterpret_cast<const f32x4 *>(sc + nl);
              f32x4 v;
#pragma unroll
              for (int e = 0; e < 4; ++e) { v[e] = acc[i][j][4 * gq + e] * (s4[e] * invA); acc[i][j][4 * gq + e] = 0.f; }
#ifdef YMI_DIAGNOSTICS
              if (p.abl & 1) continue;
              if (p.abl & 16) { if (rok && n < p.Ng) __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(mg + (size_t)row * p.Ng + n)); continue; }
#endif
              if (rok && n < p.Ng) *reinterpret_cast<f32x4 *>(mg + (size_t)row * p.Ng + n) = v;
            }
        }
      }
#ifdef YMI_DIAGNOSTICS
      const unsigned long long a_ = tracing ? __builtin_amdgcn_s_memtime() : 0ull;
#endif
      WG_BARRIER();
#ifdef YMI_DIAGNOSTICS
      if (tracing) tr_[3] += __builtin_amdgcn_s_memtime() - a_;
#endif
    }
    __builtin_amdgcn_s_setprio(0);
  }
#undef WG_WAIT_VM
#undef WG_BARRIER
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

}  // namespace

// internal (called by ymi_conv3x3_winograd_f32 for tile YMI_TILE_H2 | YMI_TILE_WG_128x256 with V written as fp16 planes): the grouped
// GEMM M_g = V_g U_g^T of a Winograd layer.  v: [G][2][T][C] fp16 planes, u: [G][2][cout_pad][C], uinv: [G][cout_pad], m: [G][T][Ng].
// Profiling record kind `prof_kind` (5 / 6 like the tile it replaces).
int ymi_internal_wgemm(const void *v, const void *u, const float *uinv, const float *x_amax, float amax_mul, float *m, int G, long T,
                       int C, int Ng, int cout_pad, double prof_flops, int prof_kind, hipStream_t s) {
  if (!v || !u || !uinv || !x_amax || !m) return YMI_ENULL;
  if (C % 32 != 0 || C < 64 || (Ng & 3) || G <= 0 || T <= 0) return YMI_ESHAPE;
  if (T * (long)C * 4 >= (1L << 31) || (long)cout_pad * C * 4 >= (1L << 31) || T * (long)Ng >= (1L << 29)) return YMI_ESHAPE;   // 32-bit buffer offsets per component
  if ((((uintptr_t)v) | ((uintptr_t)u) | ((uintptr_t)m) | ((uintptr_t)uinv)) & 15) return YMI_ESHAPE;
  WgParams p;
  p.v = (const char *)v; p.u = (const char *)u; p.uinv = uinv; p.x_amax = x_amax; p.m = m;
  p.v_plane = (unsigned)(T * C * 2); p.v_gs = 2L * T * C * 2;
  p.u_plane = (unsigned)((long)cout_pad * C * 2); p.u_gs = 2L * cout_pad * C * 2;
  p.m_gs = T * (long)Ng;
  p.G = G; p.T = (int)T; p.C = C; p.Ng = Ng; p.cout_pad = cout_pad;
  p.tiles_m = (int)((T + BM - 1) / BM); p.tiles_n = (Ng + BN - 1) / BN; p.nitems = G * p.tiles_m * p.tiles_n;
  p.amax_mul = amax_mul;
  p.trace = nullptr; p.abl = 0;
#ifdef YMI_DIAGNOSTICS
  { const char *e = getenv("YMI_WGEMM_ABLATE"); p.abl = e ? atoi(e) : 0; }
  { const char *e = getenv("YMI_WGEMM_TRACE"); p.trace = e ? (unsigned long long *)strtoull(e, nullptr, 0) : nullptr; }
#endif
  int dev = 0, cus = 256;
  hipGetDevice(&dev);
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const int pr = ymi_internal_prof_begin(prof_flops, YMI_TILE_H2 | YMI_TILE_WG_128x256, prof_kind, s);
  hipLaunchKernelGGL(wgemm_k, dim3((unsigned)(p.nitems < cus ? p.nitems : cus)), dim3(512), 0, s, p);
  const int rc = ymi_launch_status();
  ymi_internal_prof_end(pr, s);
  return rc;
}
