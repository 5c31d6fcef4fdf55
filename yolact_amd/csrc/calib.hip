// Box calibration: two fixed micro-workloads that say what THIS box delivers right now, so that a throughput number
// measured on it can be compared with one measured on another box (the driver's 1763 images/s against the
// builder's 2058 could not be attributed to the box or to the code).  bench.py times them right before the timed region:
//   ymi_calib_mfma_f16   — the matrix pipe the fp16x2 tiles run on: register-resident v_mfma_f32_32x32x16_f16, four
//                          independent accumulators per wave, operands with random mantissas (the chip clocks to its power
//                          budget: zero operands would run ~19 % faster, MI355X_MICROARCH.md "DVFS give-back");
//   ymi_calib_hbm_copy   — a float4 copy of a buffer far larger than the 256 MB Infinity Cache (read + write stream).
// Neither is on the product path; nothing reads their results but the caller's clock.
#include "common.h"
#include "../../include/yolact_amd.h"

namespace {

__global__ __launch_bounds__(256) void calib_mfma_k(float *out, int iters, unsigned seed) {
  // fragments: 8 fp16 per lane for A and B, pseudo-random mantissas, magnitudes in [2^-6, 2^-5) with alternating signs so the
  // accumulators neither overflow nor collapse to zero
  unsigned h = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + seed;
  ymi_f16x8 a[2], b[2];
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      h = h * 1664525u + 1013904223u;
      const unsigned short ba = (unsigned short)(0x2400u | ((h >> 9) & 0x3ffu) | ((h >> 3) & 0x8000u));   // +-[2^-6, 2^-5)
      h = h * 1664525u + 1013904223u;
      const unsigned short bb = (unsigned short)(0x2400u | ((h >> 9) & 0x3ffu) | ((h >> 3) & 0x8000u));
      a[f][e] = __builtin_bit_cast(_Float16, ba);
      b[f][e] = __builtin_bit_cast(_Float16, bb);
    }
  f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[0], c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[0], c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[1], c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[1], c3, 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) s += c0[e] + c1[e] + c2[e] + c3[e];
  if (s == 12345.678f) out[0] = s;      // keeps the loop alive; (practically) never taken
}

__global__ __launch_bounds__(256) void calib_copy_k(const f32x4 *__restrict__ src, f32x4 *__restrict__ dst, long n) {
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

// L2-resident read stream: every block sweeps the same small buffer (1 MB: resident in each XCD's 4 MB L2 after the first sweep)
// `iters` times with 16-byte loads, 4 independent accumulators per lane.  The GEMM tiles of this engine are bound by exactly this
// path (global -> CU at L2-hit latency), which neither the MFMA loop nor the HBM copy above exercises.
__global__ __launch_bounds__(256) void calib_l2_k(const f32x4 *__restrict__ src, int n4, int iters, float *out) {
  f32x4 a0 = {}, a1 = {}, a2 = {}, a3 = {};
  const int t = threadIdx.x, step = 256 * 4;
  for (int it = 0; it < iters; ++it) {
    // (the start offset rotates with the block so that the CUs of an XCD do not hit the same channel at the same time)
    int i = (t + ((blockIdx.x * 37 + it) & 63) * 256) % n4;
    for (int k = 0; k < n4 / step; ++k) {
      a0 += src[i]; i += 256; if (i >= n4) i -= n4;
      a1 += src[i]; i += 256; if (i >= n4) i -= n4;
      a2 += src[i]; i += 256; if (i >= n4) i -= n4;
      a3 += src[i]; i += 256; if (i >= n4) i -= n4;
    }
  }
  const f32x4 s = (a0 + a1) + (a2 + a3);
  if (s[0] + s[1] + s[2] + s[3] == 12345.678f) out[0] = s[0];
}

// Dependent-load chain: ONE lane follows chain[i] -> i for `hops` steps (the caller lays a random single-cycle permutation over the
// footprint it wants to probe, one entry per 128-byte line).  Time / hops = the load-to-use latency of that footprint: L2 for 1 MB,
// memory + address translation for gigabytes.  The axis the bandwidth probes above do not see: the pool's slow boxes run every
// streaming probe at the fast boxes' rate while every short or scatter-heavy kernel of the step is 30 - 60 % slower.
__global__ __launch_bounds__(64) void calib_chase_k(const int *__restrict__ chain, int start, int hops, int *out) {
  if (threadIdx.x != 0) return;
  int i = start;
  for (int h = 0; h < hops; ++h) {
    // (a VECTOR-memory load on purpose: the compiler would follow a uniform index through the scalar cache)
    const int *p = chain + i;
    asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(i) : "v"(p) : "memory");
  }
  out[0] = i;
}

}  // namespace

extern "C" {

int ymi_calib_latency(const int32_t *chain, long n, int start, int hops, int32_t *out, void *stream) {
  if (!chain || !out) return YMI_ENULL;
  if (n < 1 || n > 0x7fffffffL || start < 0 || start >= n || hops < 1) return YMI_EARG;
  hipLaunchKernelGGL(calib_chase_k, dim3(1), dim3(64), 0, (hipStream_t)stream, (const int *)chain, start, hops, (int *)out);
  return ymi_launch_status();
}

int ymi_calib_l2_read(const float *src, long n_floats, int blocks, int iters, float *out, double *bytes, void *stream) {
  if (!src || !out) return YMI_ENULL;
  if (n_floats < 4096 || (n_floats % 4096) || blocks < 1 || iters < 1 || (((uintptr_t)src) & 15)) return YMI_ESHAPE;
  hipLaunchKernelGGL(calib_l2_k, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const f32x4 *)src, (int)(n_floats / 4), iters, out);
  if (bytes) *bytes = 4.0 * (double)n_floats * iters * blocks;
  return ymi_launch_status();
}


int ymi_calib_mfma_f16(float *out, int blocks, int iters, double *flops, void *stream) {
  if (!out) return YMI_ENULL;
  if (blocks < 1 || iters < 1) return YMI_EARG;
  hipLaunchKernelGGL(calib_mfma_k, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters, 0x9e3779b9u);
  if (flops) *flops = (double)blocks * 4.0 /* waves */ * iters * 4.0 /* MFMAs */ * (2.0 * 32 * 32 * 16);
  return ymi_launch_status();
}

int ymi_calib_hbm_copy(const float *src, float *dst, long n_floats, double *bytes, void *stream) {
  if (!src || !dst) return YMI_ENULL;
  if (n_floats < 4 || (n_floats & 3) || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) return YMI_ESHAPE;
  hipLaunchKernelGGL(calib_copy_k, dim3(256 * 8), dim3(256), 0, (hipStream_t)stream, (const f32x4 *)src, (f32x4 *)dst,
                     n_floats / 4);
  if (bytes) *bytes = 8.0 * (double)n_floats;
  return ymi_launch_status();
}

}
