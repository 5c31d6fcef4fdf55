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

// Magnitude-bound slots (ymi_conv_desc.x_amax / y_amax).  Non-negative floats order like their bit patterns, so a running
// maximum is one integer atomic; but EVERY wave of a producing launch ends with one, and every run starts from zeroed slots.
// What the measurements of round 3 forced (sessions r3s3 .. r3s8):
//   * unconditional device-scope atomics on one address: a 38 000-wave launch went from 80 to 190 us;
//   * compare first, atomic only when it would RAISE the value: fine for a probe that re-launches a layer (the slot already
//     holds the maximum), but in a real run the first residency round of every launch sees 0 and bursts: +1 ms per step;
//   * device-scope LOADS for that compare are not free either on a multi-XCD part (they go to the memory-side coherence
//     point): 38 000 of them on 16 lines cost 18 us;
// hence: a slot is YMI_AMAX_SUB sub-slots, YMI_AMAX_STRIDE floats (their own cache lines) apart; a wave only touches the
// sub-slot of the XCD it runs on, with XCD-local (workgroup-scope) operations on that XCD's L2; the current value is loaded at
// the START of the kernel (latency hidden), the wave maximum is formed on the DPP data path (6 VALU steps, no LDS crossbar),
// and the atomic — fire and forget — is issued only if it raises what the prefetch saw.  A reader takes the maximum over the
// sub-slots (ymi_amax_read).
#define YMI_AMAX_SUB 16
#define YMI_AMAX_STRIDE 64
struct ymi_amax_pre { float *sub; unsigned cur; };
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
// At the START of a kernel: pick the sub-slot and load what it holds (lane 63 only; the latency hides behind the kernel's
// own prologue / main loop).  slot == nullptr: the launch reports nothing.
//   sub-slot = 2 * (the XCD this wave runs on) + one grid bit: a sub-slot is only ever touched from ONE XCD, so both the load
//   and the atomic are XCD-local operations on that XCD's L2 (workgroup scope: not forced out to the memory-side coherence
//   point the way device-scope accesses are on a multi-XCD part — 38 000 device-scope LOADS of 16 lines still cost an 80 us
//   launch 18 us, session r3s7).  The kernel-end release writes the L2 lines back, the consumer's kernel-start acquire
//   re-reads them: the ordinary path of every global store.
__device__ __forceinline__ ymi_amax_pre ymi_amax_prefetch(float *slot) {
  ymi_amax_pre p;
  const unsigned xcc = (unsigned)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u;     // HW_REG_XCC_ID[3:0]
  p.sub = slot ? slot + (2 * xcc + ((blockIdx.x + blockIdx.y) & 1)) * YMI_AMAX_STRIDE : nullptr;
  p.cur = 0xffffffffu;
  if (slot && (threadIdx.x & 63) == 63)
    p.cur = __hip_atomic_load(reinterpret_cast<const unsigned *>(p.sub), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  return p;
}
// At the END (a converged point of the wave): wave maximum on the DPP path, then the atomic — fire and forget, no returned
// value to wait for — only if it RAISES what the prefetch saw.  The value seen may be stale (lower): that costs a redundant
// atomic, never a wrong bound; blocks that start after the first residency round see an almost final value and stay silent.
__device__ __forceinline__ void ymi_amax_finish(const ymi_amax_pre &p, float m) {
  const unsigned bits = ymi_wave_umax63(__builtin_bit_cast(unsigned, fmaxf(m, 0.f)));
  if ((threadIdx.x & 63) == 63 && bits > p.cur)
    __hip_atomic_fetch_max(reinterpret_cast<unsigned *>(p.sub), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// The bound a consumer reads: the maximum over the sub-slots (lanes 0 .. YMI_AMAX_SUB-1 load one each; wave-uniform result).
__device__ __forceinline__ float ymi_amax_read(const float *slot) {
  unsigned v = 0;
  if ((threadIdx.x & 63) < YMI_AMAX_SUB) v = reinterpret_cast<const unsigned *>(slot)[(threadIdx.x & 63) * YMI_AMAX_STRIDE];
  v = ymi_wave_umax63(v);
  return __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_readlane((int)v, 63));
}
__device__ __forceinline__ float ymi_absmax4(const f32x4 v) {
  return fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
}

// ---- gfx950's new MFMAs through wrappers that keep the destination off the source operands -------------------------------------
// hipcc 7.2 (ROCm 7.2.0) gives v_mfma_f32_16x16x32_f16 / v_mfma_f32_32x32x16_f16 no early-clobber and no tie on vDst: whenever a
// source operand dies at the instruction the allocator may place the result on (part of) its registers —
//     v_mfma_f32_16x16x32_f16 v[220:223], v[90:93], v[218:221], v[222:225]          (first version of csrc/chain.hip)
// and on the MI355X that instruction returns wrong values in lanes 48..63 of the overlapping registers, not on every execution
// (profiles/r04_mfma_overlap.txt: 250 - 340 wrong values of 786 752, different ones per run; none once the overlap is gone).
// The wrappers pass the result and all three sources through an empty asm statement after the instruction: they are live at the
// same point, hence in disjoint registers.  A chain `acc = ymi_mfma16(a, b, acc)` therefore alternates between two register
// groups; dependent MFMAs with different vDst cost wait states, so callers interleave independent accumulators.
// tools/check_mfma_overlap.py disassembles the built objects and fails the build on any MFMA whose destination overlaps a source
// (`make lint`, run by __graft_entry__.build()): the older kernels (conv_igemm.hip, dcn.hip: long in-place chains) are clean without
// the wrappers and stay as they are.
typedef _Float16 ymi_f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x4 ymi_mfma16(const ymi_f16x8 a, const ymi_f16x8 b, const f32x4 c) {
  f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  asm volatile("" : "+v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ f32x16 ymi_mfma32(const ymi_f16x8 a, const ymi_f16x8 b, const f32x16 c) {
  f32x16 d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  asm volatile("" : "+v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
