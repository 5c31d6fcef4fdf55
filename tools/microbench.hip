// Measured peaks on the gpurun box (SURVEY §8(d) asks for measured MFMA and HBM peaks next to the datasheet ones).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench.hip -o tools/microbench.bin && tools/microbench.bin
// Prints one JSON object: fp32 MFMA TF/s (32x32x2), bf16 MFMA TF/s (32x32x16), HBM copy / read GB/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int NACC>
__global__ __launch_bounds__(256) void mfma_f32_k(float *out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 2e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void mfma_bf16_k(float *out, int iters, float a0) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(a0 + threadIdx.x * 1e-3f + e); b[e] = (__bf16)(a0 * 0.5f + e); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void copy_k(const f32x4 *__restrict__ in, f32x4 *__restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
  for (; i < n; i += stride) out[i] = in[i];
}

__global__ __launch_bounds__(256) void read_k(const f32x4 *__restrict__ in, float *out, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (; i < n; i += stride) s += in[i];
  if (s[0] + s[1] + s[2] + s[3] == 123.456f) out[0] = 1.f;
}

template <class F>
float time_ms(F f, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  float *out; CK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
  printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d", prop.gcnArchName, cus, prop.clockRate / 1000);
  const int iters = 20000;
  for (int wps = 1; wps <= 4; wps *= 2) {   // blocks per CU = waves per SIMD
    float ms = time_ms([&] { hipLaunchKernelGGL(mfma_f32_k<4>, dim3(cus * wps), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, 3);
    double fl = (double)cus * wps * 4 * iters * 4 * 4096.0;
    printf(", \"mfma_f32_32x32x2_tflops_wps%d\": %.1f", wps, fl / ms / 1e9);
  }
  {
    float ms = time_ms([&] { hipLaunchKernelGGL(mfma_f32_k<1>, dim3(cus * 4), dim3(256), 0, 0, out, iters * 4, 1.f, 2.f); }, 3);
    double fl = (double)cus * 4 * 4 * iters * 4 * 4096.0;
    printf(", \"mfma_f32_1acc_wps4_tflops\": %.1f", fl / ms / 1e9);
  }
  for (int wps = 1; wps <= 2; wps *= 2) {
    float ms = time_ms([&] { hipLaunchKernelGGL(mfma_bf16_k<4>, dim3(cus * wps), dim3(256), 0, 0, out, iters * 4, 1.f); }, 3);
    double fl = (double)cus * wps * 4 * iters * 4 * 4 * 32768.0;
    printf(", \"mfma_bf16_32x32x16_tflops_wps%d\": %.1f", wps, fl / ms / 1e9);
  }
  const size_t bytes = (size_t)2 << 30;   // 2 GiB each, far beyond the 256 MiB Infinity Cache
  f32x4 *a, *b;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
  CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
  {
    float ms = time_ms([&] { hipLaunchKernelGGL(copy_k, dim3(cus * 8), dim3(256), 0, 0, a, b, bytes / 16); }, 5);
    printf(", \"hbm_copy_gbps\": %.0f", 2.0 * bytes / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL(read_k, dim3(cus * 8), dim3(256), 0, 0, a, out, bytes / 16); }, 5);
    printf(", \"hbm_read_gbps\": %.0f", 1.0 * bytes / ms / 1e6);
  }
  printf("}\n");
  return 0;
}
