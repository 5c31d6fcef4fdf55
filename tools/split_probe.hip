// Inner-loop probe: exact-fp32 MFMA (v_mfma_f32_32x32x2_f32) vs fp32-class emulation on the bf16 matrix pipe
// (x = h + m + l, three bf16 pieces by truncation = 24 mantissa bits; 6 of the 9 piece products kept: hh hm mh mm hl lh;
// the dropped ml, lm, ll terms are <= 3 * 2^-24 |a b|, i.e. fp32-rounding class).  bf16 MFMA runs 16x the fp32-MFMA rate,
// so 6 products cost 6/16 = 0.375 of the fp32 time IF the splitting VALU work and the LDS reads hide behind the MFMAs.
// This probe measures exactly that, at the register/LDS level (operands resident in LDS, the block loops over one K chunk),
// and checks the numerics of both paths against an fp64 product on the host.
//   hipcc --offload-arch=gfx950 -O3 tools/split_probe.hip -o tools/split_probe.bin && tools/split_probe.bin
//   MODE 0  fp32 MFMA 32x32x2 (the engine's current inner loop: ds_read_b128 fragments, free-K-order trick)
//   MODE 1  A and B split on the fly after the ds_read (no change to the LDS image / the LDS-DMA staging)
//   MODE 2  A split on the fly, B (weights) pre-split on the host into three bf16 planes in LDS
//   MODE 3  both pre-split (upper bound: no VALU at all; would need producers to write split activations)
//   MODE 4  "fp16x2": x*s = h + l, two fp16 pieces by round-to-nearest (11 + 11 significant bits + signs: ~2/3 of all fp32
//           values exactly, the rest off by one fp32 ulp), s = a power-of-two scale that maps the tensor's amax below
//           65504; 3 of the 4 piece products (hl, lh, hh) on v_mfma_f32_32x32x16_f16, fp32 accumulate, ONE accumulator;
//           the dropped l*l term is <= 2^-22 |a b|.  A split on the fly (3 VALU per element, no v_perm), B pre-split
//           into two fp16 planes in LDS (same bytes as the fp32 tile)
//   MODE 5  fp16x2, both operands pre-split (no VALU in the loop)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int BK = 32;

struct Split3 { bf16x8 h, m, l; };

// 8 fp32 -> three bf16x8 by truncation: 4 VALU per element (and, sub, and, sub) + 1.5 v_perm per element pair
__device__ __forceinline__ Split3 split8(const f32x4 x0, const f32x4 x1) {
  float x[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
  unsigned r1[8], r2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float h = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x[e]) & 0xFFFF0000u);
    const float r = x[e] - h;                                   // exact: the low 16 mantissa bits
    r1[e] = __builtin_bit_cast(unsigned, r);
    const float m = __builtin_bit_cast(float, r1[e] & 0xFFFF0000u);
    r2[e] = __builtin_bit_cast(unsigned, r - m);               // exact, <= 8 significant bits -> a bf16 value
  }
  u32x4 ph, pm, pl;
#pragma unroll
  for (int q = 0; q < 4; ++q) {                                 // {odd[31:16], even[31:16]}
    ph[q] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, x[2 * q + 1]), __builtin_bit_cast(unsigned, x[2 * q]), 0x07060302u);
    pm[q] = __builtin_amdgcn_perm(r1[2 * q + 1], r1[2 * q], 0x07060302u);
    pl[q] = __builtin_amdgcn_perm(r2[2 * q + 1], r2[2 * q], 0x07060302u);
  }
  Split3 s;
  s.h = __builtin_bit_cast(bf16x8, ph); s.m = __builtin_bit_cast(bf16x8, pm); s.l = __builtin_bit_cast(bf16x8, pl);
  return s;
}

struct Split2 { f16x8 h, l; };

// 8 fp32 -> two fp16x8 by round-to-nearest: t = x * s (exact), h = f16(t), l = f16(t - h)  (t - h is exact in fp32)
__device__ __forceinline__ Split2 split8h(const f32x4 x0, const f32x4 x1, const float s) {
  const float x[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
  Split2 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float t = x[e] * s;
    const _Float16 h = (_Float16)t;
    o.h[e] = h;
    o.l[e] = (_Float16)(t - (float)h);
  }
  return o;
}

__device__ __forceinline__ f32x16 mma6(const Split3 &a, const Split3 &b, f32x16 c) {
  // small terms first, the dominant product last
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.l, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.l, b.h, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b.m, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.m, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b.h, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.h, c, 0, 0, 0);
  return c;
}

// A, B: [rows][32] fp32 in global (row-major, k contiguous).  planes: [3][rows][32] bf16 bit patterns (u16).
// LDS fp32 image: [row][32], 16-byte slots XOR-swizzled by (row>>1)&7 (the engine's layout).
// LDS plane image: [3][row][32 bf16] = 64-byte rows, 16-byte slots XOR-swizzled by (row>>2)&3.
template <int MODE, int TM, int TN, int ORDER>
__global__ __launch_bounds__(256) void probe_k(const float *__restrict__ A, const float *__restrict__ B,
                                               const unsigned short *__restrict__ Ap, const unsigned short *__restrict__ Bp,
                                               float *__restrict__ Cout, int iters, int write_c, float sA = 1.f, float sB = 1.f) {
  constexpr int BM = 2 * TM * 32, BN = 2 * TN * 32;
  __shared__ __attribute__((aligned(16))) float As[BM * BK];
  __shared__ __attribute__((aligned(16))) float Bs[BN * BK];
  __shared__ __attribute__((aligned(16))) unsigned short Aps[(MODE == 3 || MODE == 5) ? 3 * BM * BK : 8];
  __shared__ __attribute__((aligned(16))) unsigned short Bps[(MODE >= 2) ? 3 * BN * BK : 8];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
  for (int i = t; i < BM * 8; i += 256) {                       // 16-byte slots
    const int row = i >> 3, sl = i & 7;
    *reinterpret_cast<f32x4 *>(As + row * BK + 4 * (sl ^ ((row >> 1) & 7))) = *reinterpret_cast<const f32x4 *>(A + row * BK + 4 * sl);
  }
  for (int i = t; i < BN * 8; i += 256) {
    const int row = i >> 3, sl = i & 7;
    *reinterpret_cast<f32x4 *>(Bs + row * BK + 4 * (sl ^ ((row >> 1) & 7))) = *reinterpret_cast<const f32x4 *>(B + row * BK + 4 * sl);
  }
  if (MODE == 3 || MODE == 5)
    for (int i = t; i < 3 * BM * 4; i += 256) {
      const int p = i / (BM * 4), rem = i - p * BM * 4, row = rem >> 2, sl = rem & 3;
      *reinterpret_cast<u32x4 *>(Aps + (p * BM + row) * BK + 8 * (sl ^ ((row >> 2) & 3))) =
          *reinterpret_cast<const u32x4 *>(Ap + (p * BM + row) * BK + 8 * sl);
    }
  if (MODE >= 2)
    for (int i = t; i < 3 * BN * 4; i += 256) {
      const int p = i / (BN * 4), rem = i - p * BN * 4, row = rem >> 2, sl = rem & 3;
      *reinterpret_cast<u32x4 *>(Bps + (p * BN + row) * BK + 8 * (sl ^ ((row >> 2) & 3))) =
          *reinterpret_cast<const u32x4 *>(Bp + (p * BN + row) * BK + 8 * sl);
    }
  __syncthreads();

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int l31 = lane & 31, hh = lane >> 5;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
      const int fsw = (l31 >> 1) & 7;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 fa[TM], fb[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
          fa[i] = *reinterpret_cast<const f32x4 *>(As + ((wm * TM + i) * 32 + l31) * BK + 4 * ((2 * g + hh) ^ fsw));
#pragma unroll
        for (int j = 0; j < TN; ++j)
          fb[j] = *reinterpret_cast<const f32x4 *>(Bs + ((wn * TN + j) * 32 + l31) * BK + 4 * ((2 * g + hh) ^ fsw));
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][s], fb[j][s], acc[i][j], 0, 0, 0);
      }
    } else if (MODE >= 4) {
      const int fsw = (l31 >> 1) & 7, psw = (l31 >> 2) & 3;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        Split2 a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int row = (wm * TM + i) * 32 + l31;
          if (MODE == 5) {
            const int o = row * BK + 8 * ((2 * s + hh) ^ psw);
            a[i].h = *reinterpret_cast<const f16x8 *>(Aps + 0 * BM * BK + o);
            a[i].l = *reinterpret_cast<const f16x8 *>(Aps + 1 * BM * BK + o);
          } else {
            const float *p = As + row * BK;
            a[i] = split8h(*reinterpret_cast<const f32x4 *>(p + 4 * ((4 * s + 2 * hh) ^ fsw)),
                           *reinterpret_cast<const f32x4 *>(p + 4 * ((4 * s + 2 * hh + 1) ^ fsw)), sA);
          }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int row = (wn * TN + j) * 32 + l31;
          const int o = row * BK + 8 * ((2 * s + hh) ^ psw);
          b[j].h = *reinterpret_cast<const f16x8 *>(Bps + 0 * BN * BK + o);
          b[j].l = *reinterpret_cast<const f16x8 *>(Bps + 1 * BN * BK + o);
        }
#pragma unroll
        for (int pr = 0; pr < 3; ++pr)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
              const f16x8 fa_ = pr == 0 ? a[i].h : pr == 1 ? a[i].l : a[i].h;
              const f16x8 fb_ = pr == 0 ? b[j].l : pr == 1 ? b[j].h : b[j].h;
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa_, fb_, acc[i][j], 0, 0, 0);
            }
      }
    } else {
      const int fsw = (l31 >> 1) & 7, psw = (l31 >> 2) & 3;
#pragma unroll
      for (int s = 0; s < 2; ++s) {                             // lane half hh holds k = 16 s + 8 hh .. + 7
        Split3 a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int row = (wm * TM + i) * 32 + l31;
          if (MODE == 3) {
            const int o = row * BK + 8 * ((2 * s + hh) ^ psw);
            a[i].h = *reinterpret_cast<const bf16x8 *>(Aps + 0 * BM * BK + o);
            a[i].m = *reinterpret_cast<const bf16x8 *>(Aps + 1 * BM * BK + o);
            a[i].l = *reinterpret_cast<const bf16x8 *>(Aps + 2 * BM * BK + o);
          } else {
            const float *p = As + row * BK;
            a[i] = split8(*reinterpret_cast<const f32x4 *>(p + 4 * ((4 * s + 2 * hh) ^ fsw)),
                          *reinterpret_cast<const f32x4 *>(p + 4 * ((4 * s + 2 * hh + 1) ^ fsw)));
          }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int row = (wn * TN + j) * 32 + l31;
          if (MODE >= 2) {
            const int o = row * BK + 8 * ((2 * s + hh) ^ psw);
            b[j].h = *reinterpret_cast<const bf16x8 *>(Bps + 0 * BN * BK + o);
            b[j].m = *reinterpret_cast<const bf16x8 *>(Bps + 1 * BN * BK + o);
            b[j].l = *reinterpret_cast<const bf16x8 *>(Bps + 2 * BN * BK + o);
          } else {
            const float *p = Bs + row * BK;
            b[j] = split8(*reinterpret_cast<const f32x4 *>(p + 4 * ((4 * s + 2 * hh) ^ fsw)),
                          *reinterpret_cast<const f32x4 *>(p + 4 * ((4 * s + 2 * hh + 1) ^ fsw)));
          }
        }
        if (ORDER == 0) {
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = mma6(a[i], b[j], acc[i][j]);
        } else {        // product-major: consecutive MFMAs on different accumulators
#pragma unroll
          for (int pr = 0; pr < 6; ++pr)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
              for (int j = 0; j < TN; ++j) {
                const bf16x8 fa_ = pr == 0 ? a[i].h : pr == 1 ? a[i].l : pr == 2 ? a[i].m : pr == 3 ? a[i].h : pr == 4 ? a[i].m : a[i].h;
                const bf16x8 fb_ = pr == 0 ? b[j].l : pr == 1 ? b[j].h : pr == 2 ? b[j].m : pr == 3 ? b[j].m : pr == 4 ? b[j].h : b[j].h;
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa_, fb_, acc[i][j], 0, 0, 0);
              }
        }
      }
    }
  }
  if (write_c) {       // C[m][n], m = A row, n = B row; C/D layout: col = lane&31, row = (r&3) + 8(r>>2) + 4(lane>>5)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh, n = (wn * TN + j) * 32 + l31;
          Cout[(size_t)blockIdx.x * BM * BN + m * BN + n] = (MODE >= 4) ? acc[i][j][r] * (1.f / (sA * sB)) : acc[i][j][r];
        }
  } else {
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
    if (sum == 12345.678f) Cout[0] = sum;
  }
}

static unsigned short trunc_bf16(float x) { unsigned u; memcpy(&u, &x, 4); return (unsigned short)(u >> 16); }
static float from_bf16(unsigned short b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; }

static void make_planes(const std::vector<float> &X, int rows, std::vector<unsigned short> &P) {
  P.assign((size_t)3 * rows * BK, 0);
  for (int i = 0; i < rows * BK; ++i) {
    const float x = X[i];
    const unsigned short h = trunc_bf16(x);
    const float r = x - from_bf16(h);
    const unsigned short m = trunc_bf16(r);
    const float r2 = r - from_bf16(m);
    P[i] = h; P[(size_t)rows * BK + i] = m; P[(size_t)2 * rows * BK + i] = trunc_bf16(r2);
  }
}

static float pow2_scale(const std::vector<float> &X) {     // largest power of two s with amax * s <= 32768 (< 65504)
  float amax = 0.f;
  for (float v : X) amax = fmaxf(amax, fabsf(v));
  int e; frexpf(amax, &e);                                   // amax = m * 2^e, 0.5 <= m < 1
  return ldexpf(1.f, 15 - e);
}
static void make_planes_f16(const std::vector<float> &X, int rows, float s, std::vector<unsigned short> &P) {
  P.assign((size_t)3 * rows * BK, 0);
  for (int i = 0; i < rows * BK; ++i) {
    const float t = X[i] * s;
    const _Float16 h = (_Float16)t;
    const _Float16 l = (_Float16)(t - (float)h);
    memcpy(&P[i], &h, 2); memcpy(&P[(size_t)rows * BK + i], &l, 2);
  }
}

template <int MODE, int TM, int TN, int ORDER = 0>
void run(const char *name, int cus, bool first) {
  constexpr int BM = 2 * TM * 32, BN = 2 * TN * 32;
  std::vector<float> A(BM * BK), B(BN * BK);
  srand(7);
  const bool zeros = getenv("SPLIT_PROBE_ZEROS") != nullptr;   // zero operands: same instruction stream at the un-throttled clock
  for (auto &v : A) v = zeros ? 0.f : (float)rand() / RAND_MAX * 2.f - 1.f;
  for (auto &v : B) v = zeros ? 0.f : ((float)rand() / RAND_MAX * 2.f - 1.f) * 0.05f;
  // a few tiny / large magnitudes to exercise the exponent range of the pieces
  if (!zeros) { A[3] = 1e-20f; A[40] = 3.0e4f; B[5] = -2.5e-12f; B[77] = 17.f; }
  std::vector<unsigned short> Ap, Bp;
  float sA = 1.f, sB = 1.f;
  if (MODE >= 4) { sA = pow2_scale(A); sB = pow2_scale(B); make_planes_f16(A, BM, sA, Ap); make_planes_f16(B, BN, sB, Bp); }
  else { make_planes(A, BM, Ap); make_planes(B, BN, Bp); }
  float *dA, *dB, *dC; unsigned short *dAp, *dBp;
  const int max_blocks = cus * 4;
  CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dC, (size_t)max_blocks * BM * BN * 4));
  CK(hipMalloc(&dAp, Ap.size() * 2)); CK(hipMalloc(&dBp, Bp.size() * 2));
  CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dAp, Ap.data(), Ap.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dBp, Bp.data(), Bp.size() * 2, hipMemcpyHostToDevice));
  // numerics: one K chunk, compare with fp64
  hipLaunchKernelGGL((probe_k<MODE, TM, TN, ORDER>), dim3(1), dim3(256), 0, 0, dA, dB, dAp, dBp, dC, 1, 1, sA, sB);
  CK(hipDeviceSynchronize());
  std::vector<float> C(BM * BN);
  CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
  double worst = 0.0;
  for (int m = 0; m < BM; ++m)
    for (int n = 0; n < BN; ++n) {
      double ref = 0.0, mag = 0.0;
      for (int k = 0; k < BK; ++k) { ref += (double)A[m * BK + k] * B[n * BK + k]; mag += fabs((double)A[m * BK + k] * B[n * BK + k]); }
      const double e = fabs((double)C[m * BN + n] - ref) / (mag + 1e-300);
      if (e > worst) worst = e;
    }
  printf("%s\"%s\": {\"tile\": \"%dx%d\", \"err_over_sum_abs\": %.3e", first ? "" : ", ", name, BM, BN, worst);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 4000;
  for (int bpc = 1; bpc <= 3; ++bpc) {
    const int blocks = cus * bpc;
    hipLaunchKernelGGL((probe_k<MODE, TM, TN, ORDER>), dim3(blocks), dim3(256), 0, 0, dA, dB, dAp, dBp, dC, 200, 0, sA, sB);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((probe_k<MODE, TM, TN, ORDER>), dim3(blocks), dim3(256), 0, 0, dA, dB, dAp, dBp, dC, iters, 0, sA, sB);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double fl = 2.0 * BM * BN * BK * (double)iters * blocks;     // fp32-equivalent FLOPs
    printf(", \"tflops_equiv_%dblk_per_cu\": %.1f", bpc, fl / ms / 1e9);
  }
  printf("}");
  CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC)); CK(hipFree(dAp)); CK(hipFree(dBp));
}

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("{\"device\": \"%s\", \"cus\": %d, ", prop.gcnArchName, cus);
  run<0, 2, 2>("fp32_mfma_128x128", cus, true);
  run<0, 1, 2>("fp32_mfma_64x128", cus, false);
  run<1, 2, 2>("bf16x3_fly_128x128", cus, false);
  run<1, 1, 2>("bf16x3_fly_64x128", cus, false);
  run<2, 2, 2>("bf16x3_Bplanes_128x128", cus, false);
  run<2, 1, 2>("bf16x3_Bplanes_64x128", cus, false);
  run<3, 2, 2>("bf16x3_ABplanes_128x128", cus, false);
  run<3, 1, 2>("bf16x3_ABplanes_64x128", cus, false);
  run<1, 2, 2, 1>("bf16x3_fly_128x128_prodmajor", cus, false);
  run<1, 1, 2, 1>("bf16x3_fly_64x128_prodmajor", cus, false);
  run<3, 2, 2, 1>("bf16x3_ABplanes_128x128_prodmajor", cus, false);
  run<4, 2, 2>("fp16x2_Bplanes_128x128", cus, false);
  run<4, 1, 2>("fp16x2_Bplanes_64x128", cus, false);
  run<4, 1, 1>("fp16x2_Bplanes_64x64", cus, false);
  run<5, 2, 2>("fp16x2_ABplanes_128x128", cus, false);
  run<5, 1, 2>("fp16x2_ABplanes_64x128", cus, false);
  printf("}\n");
  return 0;
}
