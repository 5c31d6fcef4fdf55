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
// launch ends here, so the atomic is issued only when the wave's maximum EXCEEDS what the slot already holds (a relaxed
// device-scope load first): a 38 000-wave launch hammering one address with atomics cost 100 us (session r3s3: the 1x1
// convolutions at 138^2 went from 80 to 190 us); with the check the expected number of atomics per launch is ~ln(waves).
// The slot only grows within a run, so a stale (smaller) read can at worst issue a redundant atomic.
__device__ __forceinline__ void ymi_amax_commit(float m, float *slot) {
#pragma unroll
  for (int off = 32; off; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  if ((threadIdx.x & 63) == 0 && m > 0.f) {
    const unsigned bits = __builtin_bit_cast(unsigned, m);
    const unsigned cur = __hip_atomic_load(reinterpret_cast<unsigned *>(slot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (bits > cur) atomicMax(reinterpret_cast<unsigned *>(slot), bits);
  }
}
__device__ __forceinline__ float ymi_absmax4(const f32x4 v) {
  return fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
}
