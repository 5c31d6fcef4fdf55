// Shared helpers for the yolact_amd gfx950 kernels. No torch, no compatibility layers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define YMI_OK 0
#define YMI_EARG (-1)      // bad argument / unsupported shape
#define YMI_ESHAPE (-2)    // shape constraint violated (alignment, size)
#define YMI_ENULL (-3)     // null pointer

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static inline int ymi_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? YMI_OK : (int)e;
}

__device__ __forceinline__ int ymi_lane() { return threadIdx.x & 63; }

// XCD-aware bijective remap of a 1-D grid: block b runs on XCD (b % 8); give every XCD a
// contiguous run of logical tiles so neighbouring tiles (which share operand panels) hit
// the same per-XCD L2 (guide T1, bijective form).
__device__ __forceinline__ int ymi_xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, local = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + local;
}

// ---- fp16x2 ("h2") operand scaling ----------------------------------------------------------------------------------------
// An activation tensor is carried through the fp16 matrix pipe as  x * s = h + l  (two fp16 pieces, round to nearest);  s is
// the power of two that maps the tensor's magnitude bound `amax` (written by its producer, csrc/conv_igemm.hip epilogue) into
// [2^13, 2^14): far below the fp16 maximum 65504 and with 27 binades of normal range below it for h.  Producer-side
// pre-splitting (Winograd input transform) and consumer-side un-scaling MUST derive s from the same slot with the same
// formula, hence one helper.  amax == 0 (an all-zero tensor), inf or NaN: s = 1.
__host__ __device__ __forceinline__ void ymi_h2_scale(float amax, float &s, float &inv) {
  const unsigned u = __builtin_bit_cast(unsigned, amax);
  const int eb = (int)((u >> 23) & 0xff);            // amax in [2^(eb-127), 2^(eb-126))
  int es = 267 - eb;                                 // s = 2^(14 - (eb - 126))  ->  amax * s in [2^13, 2^14)
  if (eb == 0 || eb == 255) es = 127;
  es = es < 2 ? 2 : (es > 252 ? 252 : es);           // s and 1/s both stay normal fp32 numbers
  s = __builtin_bit_cast(float, (unsigned)es << 23);
  inv = __builtin_bit_cast(float, (unsigned)(254 - es) << 23);
}

// Non-negative floats order like their bit patterns: a device-wide running maximum is one integer atomic.  Every wave of a
// launch ends here, so (session r3s3 / r3s4 measurements: unconditional atomics on one address cost a 38 000-wave launch
// 100 us, a load-and-compare per wave on the LDS-crossbar shuffle path still 10 %):
//   * the wave maximum is formed on the DPP data path (6 VALU steps, no LDS crossbar), the result lands in lane 63;
//   * lane 63 loads the slot (relaxed, device scope) and issues the atomic only when it would RAISE it — the expected number
//     of atomics per launch is ~ln(waves); the slot only grows within a run, so a stale read at worst costs a redundant atomic;
//   * begin / end are separate so that a caller can put its output stores between the load and the compare: the load's
//     latency then overlaps the store issue instead of extending the wave's lifetime.
//   * a slot is YMI_AMAX_SUB sub-slots, YMI_AMAX_STRIDE floats (one cache line and more) apart; a wave commits to the
//     sub-slot of its XCD (below) and a reader takes the maximum over all of them (ymi_amax_read).  Every run
//     starts from zeroed slots, so the first residency round of a launch — thousands of waves finishing together, all seeing
//     0 — does raise the value with atomics; on ONE address that burst cost the plan 1 ms per step (session r3s6: a probe that
//     re-launches a layer never sees it, the slot already holds the maximum), spread over 16 lines it is ~1 us.
#define YMI_AMAX_SUB 16
#define YMI_AMAX_STRIDE 64
struct ymi_amax_ticket { unsigned bits, cur; float *sub; };
__device__ __forceinline__ unsigned ymi_wave_umax63(unsigned v) {
  auto step = [](unsigned x, unsigned y) { return x > y ? x : y; };
  v = step(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xb1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
  v = step(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4e, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
  v = step(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false));   // row_shr:4
  v = step(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false));   // row_shr:8
  v = step(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));   // row_bcast:15 -> rows 1, 3
  v = step(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));   // row_bcast:31 -> rows 2, 3
  return v;                                                                                 // valid in lane 63
}
__device__ __forceinline__ ymi_amax_ticket ymi_amax_begin(float m, float *slot) {
  ymi_amax_ticket t;
  t.bits = ymi_wave_umax63(__builtin_bit_cast(unsigned, fmaxf(m, 0.f)));
  t.cur = 0xffffffffu;
  // sub-slot = 2 * (the XCD this wave runs on) + one grid bit: a sub-slot is only ever touched from ONE XCD, so both the
  // load and the atomic can be XCD-local operations on that XCD's L2 (workgroup scope: no sc1, i.e. not forced out to the
  // memory-side coherence point the way device-scope accesses are on a multi-XCD part — 38 000 device-scope LOADS of 16 lines
  // still cost an 80 us launch 18 us, session r3s7).  The kernel-end release writes the L2 lines back, the consumer's
  // kernel-start acquire re-reads them: the ordinary path of every global store.
  const unsigned xcc = (unsigned)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u;     // HW_REG_XCC_ID[3:0]
  t.sub = slot + (2 * xcc + ((blockIdx.x + blockIdx.y) & 1)) * YMI_AMAX_STRIDE;
  if ((threadIdx.x & 63) == 63 && t.bits != 0)
    t.cur = __hip_atomic_load(reinterpret_cast<const unsigned *>(t.sub), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  return t;
}
__device__ __forceinline__ void ymi_amax_end(const ymi_amax_ticket &t, float *) {
  if ((threadIdx.x & 63) == 63 && t.bits > t.cur)
    __hip_atomic_fetch_max(reinterpret_cast<unsigned *>(t.sub), t.bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// The bound a consumer reads: the maximum over the sub-slots (lanes 0 .. YMI_AMAX_SUB-1 load one each; wave-uniform result).
__device__ __forceinline__ float ymi_amax_read(const float *slot) {
  unsigned v = 0;
  if ((threadIdx.x & 63) < YMI_AMAX_SUB) v = reinterpret_cast<const unsigned *>(slot)[(threadIdx.x & 63) * YMI_AMAX_STRIDE];
  v = ymi_wave_umax63(v);
  return __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_readlane((int)v, 63));
}
__device__ __forceinline__ void ymi_amax_commit(float m, float *slot) {
  const ymi_amax_ticket t = ymi_amax_begin(m, slot);
  ymi_amax_end(t, slot);
}
__device__ __forceinline__ float ymi_absmax4(const f32x4 v) {
  return fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
}
