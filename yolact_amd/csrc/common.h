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
