// Detect on device: softmax + candidate filter, per-class top-k + Fast NMS, final top-N merge.
// Reference: layers/functions/detection.py:32-180 (Detect.__call__/detect/fast_nms/cc_fast_nms),
//            layers/box_utils.py:33-80 (intersect/jaccard), :267-312 (decode).
//
// Whole batch in three launches, no host synchronisation, fixed-capacity outputs + counts:
//   K1  grid (P/64, B)      softmax over C classes, fg max/argmax, keep = max > conf_thresh, class-major scores whose SIGN
//                           BIT carries the keep flag
//   K2  grid (nclass, B)    1024 threads: select the top_k kept priors of a class (keys in registers by 16-byte loads,
//                           bisection on the key bits with scalar-unit popcounts, prefix-sum compaction, in-wave bitonic
//                           sort), decode their boxes, Fast-NMS column tests spread evenly over all threads (reciprocal
//                           pre-test, exact division only near the threshold), keep iou_max <= nms_thresh
//   K3  grid (B)            1024 threads: select + sort the best max_det survivors over all classes, gather outputs
// r03 phase traces (tools/detect_trace.py, dense synthetic heads: 13 214 kept priors per image): K2 217k -> 106k shader
// cycles per block (loads 82k -> 2.6k, IoU 50k -> 22k), K3 119 -> 36 us at batch 1.
// Tie rule: the reference sorts with the unstable torch.sort; we define stable order (lowest index first),
// SURVEY §7 hard part 3(iv).  All float math mirrors the reference's op order; build with -ffp-contract=off.
#include "common.h"
#include <stdlib.h>
#include "../../include/yolact_amd.h"

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int NT = 1024;         // threads per block of K2 / K3: 16 waves.  These kernels are instruction-ISSUE bound (one wave
                                 // issues ~1 instruction per 5 cycles) and a (class, image) block has a CU to itself at batch 1,
                                 // so the keys are spread over as many waves as a block can have
constexpr int NT_MAX = 1024;
constexpr int SORT_N = 256;      // bitonic capacity (top_k, max_det <= 256)

// diagnostics (`make DIAG=1`, env YMI_DETECT_TRACE = device address of a u64 buffer, 16 slots per block): phase time stamps of
// K2 / K3 by thread 0 (tools/detect_probe.py).  Product builds pass nullptr: one scalar branch per stamp.
#define YMI_STAMP(i) do { if (trace && threadIdx.x == 0) trace[(size_t)tblk * 16 + (i)] = __builtin_readcyclecounter(); } while (0)

__device__ __forceinline__ unsigned f2key(float f) {
  // order-preserving float -> uint (larger float => larger key); never 0 for finite inputs
  unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// decode(loc, prior) exactly as box_utils.py:304-310 evaluates it (left to right, fp32):
//   c = p.xy + (loc.xy * 0.1) * p.wh ; s = p.wh * exp(loc.wh * 0.2) ; xy1 = c - s/2 ; xy2 = s + xy1
__device__ __forceinline__ f32x4 decode_box(const float *loc, const float *pr) {
  const float cx = pr[0] + (loc[0] * 0.1f) * pr[2];
  const float cy = pr[1] + (loc[1] * 0.1f) * pr[3];
  const float w = pr[2] * expf(loc[2] * 0.2f);
  const float h = pr[3] * expf(loc[3] * 0.2f);
  f32x4 b;
  b[0] = cx - w / 2.f;
  b[1] = cy - h / 2.f;
  b[2] = w + b[0];
  b[3] = h + b[1];
  return b;
}

// ------------------------------------------------------------------------------------------------
// K1: softmax + keep.  Block = 64 priors x C classes staged in LDS (row stride C, C odd => conflict free).
__global__ __launch_bounds__(256) void softmax_keep_k(const float *__restrict__ conf, int P, int C, int ld, int is_logits,
                                                      float thresh, float *__restrict__ scores_t,
                                                      int *__restrict__ keep, int *__restrict__ num_keep,
                                                      float *__restrict__ maxsc, int *__restrict__ argmax) {
  extern __shared__ float s[];  // 64 * C
  __shared__ int blk_cnt;
  __shared__ unsigned char kflag[64];
  const int b = blockIdx.y, p0 = blockIdx.x * 64;
  const int np = min(64, P - p0);
  const int t = threadIdx.x;
  if (t == 0) blk_cnt = 0;
  const float *src = conf + ((size_t)b * P + p0) * ld;
  if (ld == C) {
    for (int i = t; i < np * C; i += 256) s[i] = src[i];
  } else {                                       // padded class rows (ld > C): drop the padding while staging
    for (int i = t; i < np * C; i += 256) { const int jj = i / C, c = i - jj * C; s[i] = src[jj * ld + c]; }
  }
  __syncthreads();

  const int j = t >> 2, sub = t & 3;  // 4 lanes per prior
  float *row = s + j * C;
  const bool live = j < np;
  if (is_logits) {
    float mx = -__builtin_inff();
    if (live) for (int c = sub; c < C; c += 4) mx = fmaxf(mx, row[c]);
    mx = fmaxf(mx, __shfl_xor(mx, 1));
    mx = fmaxf(mx, __shfl_xor(mx, 2));
    float sum = 0.f;
    if (live) for (int c = sub; c < C; c += 4) { const float e = expf(row[c] - mx); row[c] = e; sum += e; }
    sum += __shfl_xor(sum, 1);
    sum += __shfl_xor(sum, 2);
    if (live) for (int c = sub; c < C; c += 4) row[c] = row[c] / sum;
  }
  // foreground max / argmax (first index on ties, like a left-to-right scan)
  float best = -__builtin_inff();
  int bi = 0x7fffffff;
  if (live) for (int c = 1 + sub; c < C; c += 4) { const float v = row[c]; if (v > best) { best = v; bi = c - 1; } }
#pragma unroll
  for (int off = 1; off <= 2; off <<= 1) {
    const float ob = __shfl_xor(best, off);
    const int oi = __shfl_xor(bi, off);
    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  if (live && sub == 0) {
    const int kp = best > thresh ? 1 : 0;
    keep[(size_t)b * P + p0 + j] = kp;
    kflag[j] = (unsigned char)kp;
    maxsc[(size_t)b * P + p0 + j] = best;
    argmax[(size_t)b * P + p0 + j] = bi;
    if (kp) atomicAdd(&blk_cnt, 1);
  }
  __syncthreads();
  if (t == 0 && blk_cnt) atomicAdd(&num_keep[b], blk_cnt);
  // class-major store: scores_t[b][c-1][p0 + j]; the SIGN BIT marks the priors below the confidence threshold (softmax
  // scores are >= 0), so that K2 reads one array instead of two
  const int nfg = C - 1;
  for (int i = t; i < nfg * 64; i += 256) {
    const int c = i >> 6, jj = i & 63;
    if (jj < np) {
      const float v = s[jj * C + c + 1];
      scores_t[((size_t)b * nfg + c) * P + p0 + jj] = kflag[jj] ? v : __uint_as_float(__float_as_uint(v) | 0x80000000u);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Block-wide top-k selection + sort.  Keys: order-preserving uint of the score, 0 = not a candidate.  Selects the k
// largest keys, ties broken by lowest index, and leaves them sorted (key desc, index asc) in sh.comp[0..k).
struct SelShared {
  unsigned hist[2 * (NT_MAX / 64)];   // [2][waves] per-wave counts of the bisection passes (alternating slots)
  unsigned long long comp[SORT_N];
  unsigned cnt_gt;
  unsigned wave_tot[2][NT_MAX / 64];
};

// Wave64 sum on the DPP data path (quad swaps, row_shr 4/8, row_bcast 15/31); the total lands in lane 63.
__device__ __forceinline__ unsigned wave_sum_to_lane63(unsigned v) {
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xb1, 0xf, 0xf, false);    // quad_perm [1,0,3,2]
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4e, 0xf, 0xf, false);    // quad_perm [2,3,0,1]
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1,3
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2,3
  return v;
}

// inclusive prefix sum over the lanes of a wave (used once per selection, not per pass)
__device__ __forceinline__ unsigned wave_incl_scan(unsigned v, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned o = (unsigned)__shfl_up((int)v, off);
    if (lane >= off) v += o;
  }
  return v;
}

// exclusive prefix of `v` over the threads of the block (thread order) and the block total; ONE barrier; `slot`
// alternates between consecutive calls so that a second call needs no barrier of its own before writing
template <int NTH>
__device__ __forceinline__ unsigned block_excl_scan(unsigned v, SelShared &sh, int slot, unsigned &total) {
  constexpr int NW = NTH / 64;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const unsigned inc = wave_incl_scan(v, lane);
  if (lane == 63) sh.wave_tot[slot][w] = inc;
  __syncthreads();
  unsigned before = 0, tot = 0;
#pragma unroll
  for (int ww = 0; ww < NW; ++ww) { const unsigned x = sh.wave_tot[slot][ww]; if (ww < w) before += x; tot += x; }
  total = tot;
  return before + inc - v;
}

// Bitonic sort of sh.comp[0..SORT_N), descending on the 64-bit composite (key desc, index asc): thread t < SORT_N owns
// element t in registers; partners inside a wave (stride < 64) are exchanged with lane shuffles and no barrier, only the
// strides 64 / 128 (3 of the 36 stages) go through LDS.  (36 LDS stages with a barrier each cost 10.7k cycles of the 217k
// of a K2 block, r03 phase trace.)
__device__ __forceinline__ void block_bitonic_desc(SelShared &sh) {
  const int t = threadIdx.x;
  const bool mine = t < SORT_N;
  unsigned long long v = mine ? sh.comp[t] : 0ull;
#pragma unroll
  for (int size = 2; size <= SORT_N; size <<= 1) {
#pragma unroll
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      unsigned long long o;
      if (stride >= 64) {
        __syncthreads();
        if (mine) sh.comp[t] = v;
        __syncthreads();
        o = mine ? sh.comp[t ^ stride] : 0ull;
      } else {
        const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)v, stride);
        const unsigned hi = (unsigned)__shfl_xor((int)(unsigned)(v >> 32), stride);
        o = ((unsigned long long)hi << 32) | lo;
      }
      const bool want_max = ((t & size) == 0) == ((t & stride) == 0);   // descending block: the lower index keeps the larger
      const unsigned long long mx = v > o ? v : o, mn = v > o ? o : v;
      v = want_max ? mx : mn;
    }
  }
  __syncthreads();
  if (mine) sh.comp[t] = v;
  __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// Register-resident top-k: every thread holds EPT keys in VGPRs and the k-th largest key is found by bisection on its 32
// bits.  One pass = EPT compares whose 64-lane masks are counted on the SCALAR unit (s_bcnt1 + s_add run beside the VALU:
// one vector instruction per key; the per-lane compare + add-with-carry + DPP reduction this replaces took 1600 cycles per
// pass, 55k of the 217k cycles of a K2 block), one LDS word per wave and ONE barrier.  No memory traffic, no atomics.
// Register slot j of thread t holds element 4 * (t + NTH * (j / 4)) + j % 4: four consecutive elements per thread, so that
// the keys arrive as 16-byte loads (160 dword loads per thread took 82k cycles).
// The winners are compacted with one block-wide prefix sum (no LDS atomics) and sorted (key desc, index asc) into
// sh.comp[0..k).  EPT == 0 (n too large for the register file): element t + NTH * j is re-evaluated from memory
// (L2-resident) on every pass.
// keys[j] > cand (STRICT) / >= cand over the whole wave: the 64-lane mask of each compare goes to an SGPR pair and is counted
// by s_bcnt1 + s_add on the scalar unit.  Eight keys per asm statement: written in C++ (ballot + popcount), the compiler moves
// all EPT counts behind all EPT compares (into the lane-0 store's block) and spills the masks through v_writelane /
// v_readlane (1600 of them for EPT = 80).
#define YMI_CMP8(OP)                                                                                                  \
  asm volatile(OP " %[m0], %[k0], %[cd]\n\t" OP " %[m1], %[k1], %[cd]\n\t" OP " %[m2], %[k2], %[cd]\n\t"            \
               OP " %[m3], %[k3], %[cd]\n\t" OP " %[m4], %[k4], %[cd]\n\t" OP " %[m5], %[k5], %[cd]\n\t"            \
               OP " %[m6], %[k6], %[cd]\n\t" OP " %[m7], %[k7], %[cd]\n\t"                                          \
               "s_bcnt1_i32_b64 %[n0], %[m0]\n\ts_bcnt1_i32_b64 %[n1], %[m1]\n\ts_bcnt1_i32_b64 %[n2], %[m2]\n\t"   \
               "s_bcnt1_i32_b64 %[n3], %[m3]\n\ts_bcnt1_i32_b64 %[n4], %[m4]\n\ts_bcnt1_i32_b64 %[n5], %[m5]\n\t"   \
               "s_bcnt1_i32_b64 %[n6], %[m6]\n\ts_bcnt1_i32_b64 %[n7], %[m7]\n\t"                                  \
               "s_add_u32 %[n0], %[n0], %[n1]\n\ts_add_u32 %[n2], %[n2], %[n3]\n\ts_add_u32 %[n4], %[n4], %[n5]\n\t" \
               "s_add_u32 %[n6], %[n6], %[n7]\n\ts_add_u32 %[n0], %[n0], %[n2]\n\ts_add_u32 %[n4], %[n4], %[n6]\n\t" \
               "s_add_u32 %[n0], %[n0], %[n4]\n\ts_add_u32 %[c], %[c], %[n0]"                                        \
               : [c] "+s"(c), [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2), [m3] "=&s"(m3), [m4] "=&s"(m4),         \
                 [m5] "=&s"(m5), [m6] "=&s"(m6), [m7] "=&s"(m7), [n0] "=&s"(n0), [n1] "=&s"(n1), [n2] "=&s"(n2),      \
                 [n3] "=&s"(n3), [n4] "=&s"(n4), [n5] "=&s"(n5), [n6] "=&s"(n6), [n7] "=&s"(n7)                       \
               : [k0] "v"(k[0]), [k1] "v"(k[1]), [k2] "v"(k[2]), [k3] "v"(k[3]), [k4] "v"(k[4]), [k5] "v"(k[5]),      \
                 [k6] "v"(k[6]), [k7] "v"(k[7]), [cd] "s"(cand)                                                       \
               : "scc")
template <int EPT, bool STRICT>
__device__ __forceinline__ unsigned wave_count_keys(const unsigned (&keys)[EPT > 0 ? EPT : 1], unsigned cand_any) {
  static_assert(EPT % 8 == 0, "keys per thread: a multiple of 8");
  const unsigned cand = (unsigned)__builtin_amdgcn_readfirstlane((int)cand_any);
  unsigned c = 0;
#pragma unroll
  for (int g = 0; g < EPT / 8; ++g) {
    const unsigned *k = &keys[8 * g];
    unsigned long long m0, m1, m2, m3, m4, m5, m6, m7;
    unsigned n0, n1, n2, n3, n4, n5, n6, n7;
    if (STRICT) YMI_CMP8("v_cmp_gt_u32_e64");
    else YMI_CMP8("v_cmp_ge_u32_e64");
  }
  return c;
}
#undef YMI_CMP8

template <int NTH>
__device__ __forceinline__ int reg_index(int t, int j) { return 4 * (t + NTH * (j >> 2)) + (j & 3); }

template <int EPT, int NTH, typename KeyFn>
__device__ __forceinline__ void block_topk_regs(const unsigned (&keys)[EPT > 0 ? EPT : 1], KeyFn key, int n, int k, SelShared &sh,
                                unsigned long long *trace = nullptr, long tblk = 0) {
  constexpr int NW = NTH / 64;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int rounds = EPT > 0 ? EPT : (n + NTH - 1) / NTH;
  auto key_at = [&](int j) -> unsigned {          // EPT == 0 only
    const int i = t + NTH * j;
    return i < n ? key(i) : 0u;
  };
  auto index_of = [&](int j) -> int { return EPT > 0 ? reg_index<NTH>(t, j) : t + NTH * j; };
  auto block_count = [&](unsigned cand, bool strict, int slot) -> unsigned {
    unsigned c = 0;
    if (EPT > 0) {
      c = strict ? wave_count_keys<EPT, true>(keys, cand) : wave_count_keys<EPT, false>(keys, cand);   // wave-uniform (SALU)
      if (lane == 0) sh.hist[slot * NW + w] = c;
    } else {
      for (int j = 0; j < rounds; ++j) {
        const unsigned kk = key_at(j);
        c += (strict ? kk > cand : kk >= cand) ? 1u : 0u;
      }
      c = wave_sum_to_lane63(c);
      if (lane == 63) sh.hist[slot * NW + w] = c;
    }
    __syncthreads();
    unsigned tot = 0;
#pragma unroll
    for (int ww = 0; ww < NW; ++ww) tot += sh.hist[slot * NW + ww];
    return tot;
  };
  unsigned T = 0;
  for (int bit = 31; bit >= 0; --bit) {
    const unsigned cand = T | (1u << bit);
    if (block_count(cand, false, bit & 1) >= (unsigned)k) T = cand;   // alternating slots: one barrier per bit
  }
  __syncthreads();
  YMI_STAMP(2);
  const unsigned n_gt = block_count(T, true, 0);      // keys strictly above the threshold: all taken
  const unsigned n_ge = block_count(T, false, 1);
  const unsigned krem = (unsigned)k - n_gt;            // how many of the keys == T are taken (lowest indices first)
  if (t < SORT_N) sh.comp[t] = 0ull;
  if (t == 0) sh.cnt_gt = 0;
  const bool all_eq = n_ge - n_gt == krem;             // common case: every key equal to T is taken
  // winners that need no tie-breaking (all keys >= T, or only those > T): compacted by a prefix sum; their order is
  // irrelevant (sorted below)
  auto wins = [&](unsigned kj) -> bool { return kj != 0 && (all_eq ? kj >= T : kj > T); };
  unsigned mine = 0;
  if constexpr (EPT > 0) {
#pragma unroll
    for (int j = 0; j < EPT; ++j) mine += wins(keys[j]) ? 1u : 0u;
  } else {
    for (int j = 0; j < rounds; ++j) mine += wins(key_at(j)) ? 1u : 0u;
  }
  unsigned total;
  unsigned pos = block_excl_scan<NTH>(mine, sh, 0, total);      // (its barrier also publishes the zeroed comp[])
  if (mine) {
    if constexpr (EPT > 0) {
#pragma unroll
      for (int j = 0; j < EPT; ++j)
        if (wins(keys[j])) {
          sh.comp[pos] = ((unsigned long long)keys[j] << 32) | (unsigned)(0xffffffffu - (unsigned)index_of(j));
          ++pos;
        }
    } else {
      for (int j = 0; j < rounds; ++j) {
        const unsigned kj = key_at(j);
        if (wins(kj)) {
          sh.comp[pos] = ((unsigned long long)kj << 32) | (unsigned)(0xffffffffu - (unsigned)index_of(j));
          ++pos;
        }
      }
    }
  }
  if (!all_eq) {
    // tie on the threshold with more equals than slots (rare): the krem equals with the LOWEST element indices are taken.
    // Index order = (group of 4 registers, thread, register) for the register layout, (round, thread) for the memory one;
    // one prefix sum per group.
    unsigned base_eq = 0;
    if constexpr (EPT > 0) {
#pragma unroll
      for (int g = 0; g < EPT / 4; ++g) {
        unsigned ec = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) ec += (keys[4 * g + e] != 0 && keys[4 * g + e] == T) ? 1u : 0u;
        unsigned tot;
        unsigned rank = base_eq + block_excl_scan<NTH>(ec, sh, 1 - (g & 1), tot);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (keys[4 * g + e] != 0 && keys[4 * g + e] == T) {
            if (rank < krem)
              sh.comp[n_gt + rank] = ((unsigned long long)T << 32) | (unsigned)(0xffffffffu - (unsigned)index_of(4 * g + e));
            ++rank;
          }
        base_eq += tot;
      }
    } else {
      for (int j = 0; j < rounds; ++j) {
        const unsigned kk = key_at(j);
        const bool eq = kk != 0 && kk == T;
        unsigned tot;
        const unsigned rank = base_eq + block_excl_scan<NTH>(eq ? 1u : 0u, sh, 1 - (j & 1), tot);
        if (eq && rank < krem) sh.comp[n_gt + rank] = ((unsigned long long)T << 32) | (unsigned)(0xffffffffu - (unsigned)index_of(j));
        base_eq += tot;
      }
    }
  }
  __syncthreads();
  YMI_STAMP(3);
  block_bitonic_desc(sh);
}

// Column test of Fast NMS: is max_{i in [r0,r1)} IoU(box i, box j) <= thresh FALSE, IoU exactly as jaccard() evaluates it
// (box_utils.py:47-51,72-79): inter / (area_i + area_j - inter); a NaN IoU suppresses the column (the reference's
// `iou_max <= thresh` is False for NaN).  Only the comparison is needed, so the IEEE division (~25 instructions) runs only
// when some quotient by v_rcp_f32 (1 ulp) lies within 2^-20 of the threshold or a union is not a normal number; everywhere
// else `inter * rcp(union)` decides, identically.  (The division loop was 50k of the 217k cycles of a K2 block.)
__device__ __forceinline__ bool iou_col_suppressed(const f32x4 *bx, int j, int r0, int r1, float thresh, float lo, float hi) {
  const f32x4 bj = bx[j];
  const float area_j = (bj[2] - bj[0]) * (bj[3] - bj[1]);
  bool sup = false, unsure = false;
#pragma unroll 4
  for (int i = r0; i < r1; ++i) {                      // straight-line body: no divergence
    const f32x4 a = bx[i];
    float iw = fminf(a[2], bj[2]) - fmaxf(a[0], bj[0]);
    float ih = fminf(a[3], bj[3]) - fmaxf(a[1], bj[1]);
    iw = iw < 0.f ? 0.f : iw;
    ih = ih < 0.f ? 0.f : ih;
    const float inter = iw * ih;
    const float uni = ((a[2] - a[0]) * (a[3] - a[1]) + area_j) - inter;
    const float q = inter * __builtin_amdgcn_rcpf(uni);
    const bool normal = uni > 1e-30f && uni < 1e30f;   // (false for NaN)
    const bool above = normal && q > hi, below = normal && q < lo;
    sup |= above;
    unsure |= !(above || below);
  }
  if (unsure) {                                        // rare: some row needs the correctly rounded quotient
    for (int i = r0; i < r1; ++i) {
      const f32x4 a = bx[i];
      float iw = fminf(a[2], bj[2]) - fmaxf(a[0], bj[0]);
      float ih = fminf(a[3], bj[3]) - fmaxf(a[1], bj[1]);
      iw = iw < 0.f ? 0.f : iw;
      ih = ih < 0.f ? 0.f : ih;
      const float inter = iw * ih;
      const float iou = inter / (((a[2] - a[0]) * (a[3] - a[1]) + area_j) - inter);
      if (!(iou <= thresh)) sup = true;                // above, or NaN
    }
  }
  return sup;
}

// K2: one block per (class, image).  EPT = keys per thread (0: keys stay in memory).  SIGNED: the scores carry the keep flag
// in their sign bit (scores_t as K1 writes it); otherwise (cross-class: the per-prior fg max) the keep array is read too.
template <int EPT, int NTH, bool SIGNED>
__global__ __launch_bounds__(NTH) void class_topk_nms_k(const float *__restrict__ scores,  // [B,nclass,P]
                                                       const int *__restrict__ keep, const int *__restrict__ num_keep,
                                                       const float *__restrict__ loc, const float *__restrict__ priors,
                                                       int P, int nclass, int top_k, float nms_thresh,
                                                       float *__restrict__ cand_score, int *__restrict__ cand_prior,
                                                       int dbg, unsigned long long *trace) {
  const long tblk = blockIdx.y * gridDim.x + blockIdx.x;
  YMI_STAMP(0);
  __shared__ SelShared sh;
  __shared__ f32x4 bx[SORT_N];
  __shared__ unsigned char supf[SORT_N];
  const int c = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
  const int K = num_keep[b];
  float *cs = cand_score + ((size_t)b * nclass + c) * top_k;
  int *cp = cand_prior + ((size_t)b * nclass + c) * top_k;
  if (K == 0) {
    for (int i = t; i < top_k; i += NTH) { cs[i] = -1.f; cp[i] = -1; }
    return;
  }
  const int k = K < top_k ? K : top_k;
  const float *sc = scores + ((size_t)b * nclass + c) * P;
  const int *kp = keep + (size_t)b * P;
  auto key_of = [&](float sv, int kf) -> unsigned {
    if (SIGNED) return (__float_as_uint(sv) & 0x80000000u) ? 0u : f2key(sv);
    return kf ? f2key(sv) : 0u;
  };
  if constexpr (EPT > 0) {
    unsigned keys[EPT];
    const bool vec = (P & 3) == 0 && ((((uintptr_t)sc) | ((uintptr_t)kp)) & 15) == 0;   // block-uniform
    if (vec) {
#pragma unroll
      for (int g = 0; g < EPT / 4; ++g) {
        const int i = 4 * (t + NTH * g), ii = i < P ? i : 0;   // unconditional, independent 16-byte loads
        const f32x4 sv = *reinterpret_cast<const f32x4 *>(sc + ii);
        i32x4 kf = {1, 1, 1, 1};
        if (!SIGNED) kf = *reinterpret_cast<const i32x4 *>(kp + ii);
#pragma unroll
        for (int e = 0; e < 4; ++e) keys[4 * g + e] = i < P ? key_of(sv[e], kf[e]) : 0u;
      }
    } else {
#pragma unroll
      for (int j = 0; j < EPT; ++j) {
        const int i = reg_index<NTH>(t, j), ii = i < P ? i : 0;
        const float sv = sc[ii];
        const int kf = SIGNED ? 1 : kp[ii];
        keys[j] = i < P ? key_of(sv, kf) : 0u;
      }
    }
    if (trace) { unsigned o = 0;
#pragma unroll
      for (int j = 0; j < EPT; ++j) o |= keys[j];
      if (o == 0xdeadbeefu) trace[0] = 0; }            // (forces the loads to have landed before stamp 1)
    YMI_STAMP(1);
    if (dbg & 1) { sh.comp[t] = ((unsigned long long)keys[0] << 32) | (unsigned)(0xffffffffu - (unsigned)reg_index<NTH>(t, 0)); __syncthreads(); }
    else block_topk_regs<EPT, NTH>(keys, [](int) -> unsigned { return 0u; }, P, k, sh, trace, tblk);
  } else {
    const unsigned none[1] = {0u};
    block_topk_regs<0, NTH>(none, [&](int i) -> unsigned { return key_of(sc[i], SIGNED ? 1 : kp[i]); }, P, k, sh);
  }

  YMI_STAMP(4);
  // rank t -> prior index, score, decoded box
  int prior = -1;
  float score = -1.f;
  if (t < k) {
    prior = (int)(0xffffffffu - (unsigned)(sh.comp[t] & 0xffffffffull));
    score = sc[prior];
    const f32x4 bb = decode_box(loc + ((size_t)b * P + prior) * 4, priors + (size_t)prior * 4);
    bx[t] = bb;
  }
  __syncthreads();
  YMI_STAMP(5);
  // iou_max[j] = max_{i<j} IoU(i, j) (0 for j = 0); keep = iou_max <= nms_thresh.  Column j costs j IoUs: column u < k/2 is
  // paired with column k-1-u into a "virtual column" of k-1 rows (the middle column of an odd k stands alone), and every
  // virtual column is cut into PARTS equal row ranges, one thread each: (k/2) * PARTS threads of ~(k-1)/PARTS IoUs, all the
  // same length.  A thread that finds a suppressing row raises its column's flag (plain store of 1: no atomics needed).
  constexpr int PARTS = NTH / (SORT_N / 2);           // 8 for 1024 threads
  const float eps = fabsf(nms_thresh) * 9.5367431640625e-07f;   // 2^-20
  const float lo = nms_thresh - eps, hi = nms_thresh + eps;
  if (t < SORT_N) supf[t] = 0;
  __syncthreads();
  if (!(dbg & 4)) {
    const int u = t & (SORT_N / 2 - 1), part = t / (SORT_N / 2);
    const int nvc = (k + 1) / 2;                       // virtual columns
    if (u < nvc) {
      const int jl = k - 1 - u;                        // the long column of the pair (== u for the middle column)
      const int L = jl, rows = (jl == u) ? L : k - 1;  // rows [0, L) belong to column jl, rows [L, k-1) to column u
      const int ra = (int)(((long)rows * part) / PARTS), rb = (int)(((long)rows * (part + 1)) / PARTS);
      const int a1 = ra < L ? ra : L, b1 = rb < L ? rb : L;           // rows of column jl: [a1, b1)
      if (a1 < b1 && iou_col_suppressed(bx, jl, a1, b1, nms_thresh, lo, hi)) supf[jl] = 1;
      const int a2 = (ra > L ? ra : L) - L, b2 = (rb > L ? rb : L) - L;   // rows of column u: [a2, b2)
      if (a2 < b2 && iou_col_suppressed(bx, u, a2, b2, nms_thresh, lo, hi)) supf[u] = 1;
    }
  }
  __syncthreads();
  YMI_STAMP(6);
  if (t < top_k) {
    bool kept = false;
    if (t < k) kept = (dbg & 4) ? true : (supf[t] == 0 && 0.f <= nms_thresh);   // (the maximum also runs over the zeroed
                                                                                 // lower triangle: >= 0)
    cs[t] = kept ? score : -1.f;
    cp[t] = kept ? prior : -1;
  }
  YMI_STAMP(7);
}

// K3: one block per image: best max_det over all per-class survivors (flattened class-major, rank-minor)
template <int EPT>
__global__ __launch_bounds__(NT) void final_topk_k(const float *__restrict__ cand_score, const int *__restrict__ cand_prior,
                                                   const float *__restrict__ loc, const float *__restrict__ priors,
                                                   const float *__restrict__ coef, const int *__restrict__ argmax,
                                                   int P, int D, int nclass, int top_k, int cap, int cross_class,
                                                   int *__restrict__ out_count, float *__restrict__ out_box,
                                                   float *__restrict__ out_score, long long *__restrict__ out_class,
                                                   float *__restrict__ out_coef, int *__restrict__ out_prior,
                                                   float *__restrict__ out_rec, unsigned long long *trace) {
  const long tblk = 4096 + blockIdx.x;
  YMI_STAMP(0);
  __shared__ SelShared sh;
  __shared__ unsigned nvalid;
  __shared__ int sel_f[SORT_N], sel_prior[SORT_N];
  const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int n = nclass * top_k;
  const float *cs = cand_score + (size_t)b * n;
  const int *cp = cand_prior + (size_t)b * n;
  // packed record of image b (ymi_detect_desc.out_rec): count | cap x (box 4, score, class, coef D), all fp32
  const int RL = 6 + D;
  float *rec = out_rec ? out_rec + (size_t)b * (1 + (size_t)cap * RL) : nullptr;
  int k = 0;
  if constexpr (EPT > 0) {
    unsigned keys[EPT];
    const bool vec = (n & 3) == 0 && ((((uintptr_t)cs) | ((uintptr_t)cp)) & 15) == 0;
    if (vec) {
#pragma unroll
      for (int g = 0; g < EPT / 4; ++g) {
        const int i = 4 * (t + NT * g), ii = i < n ? i : 0;
        const f32x4 sv = *reinterpret_cast<const f32x4 *>(cs + ii);
        const i32x4 pf = *reinterpret_cast<const i32x4 *>(cp + ii);
#pragma unroll
        for (int e = 0; e < 4; ++e) keys[4 * g + e] = (i < n && pf[e] >= 0) ? f2key(sv[e]) : 0u;
      }
    } else {
#pragma unroll
      for (int j = 0; j < EPT; ++j) {
        const int i = reg_index<NT>(t, j), ii = i < n ? i : 0;
        const int pf = cp[ii];
        const float sv = cs[ii];
        keys[j] = (i < n && pf >= 0) ? f2key(sv) : 0u;
      }
    }
    // survivors = non-zero keys: counted on the scalar unit like a bisection pass (the separate counting loop over the
    // candidate arrays took 26k cycles)
    const unsigned c = wave_count_keys<EPT, true>(keys, 0u);
    if (lane == 0) sh.hist[w] = c;
    __syncthreads();
    unsigned nv = 0;
#pragma unroll
    for (int ww = 0; ww < NT / 64; ++ww) nv += sh.hist[ww];
    __syncthreads();
    YMI_STAMP(1);
    k = (int)nv < cap ? (int)nv : cap;
    if (t == 0) { out_count[b] = k; if (rec) rec[0] = (float)k; }
    if (k == 0) return;
    block_topk_regs<EPT, NT>(keys, [](int) -> unsigned { return 0u; }, n, k, sh, trace, tblk);
  } else {
    if (t == 0) nvalid = 0;
    __syncthreads();
    unsigned local = 0;
    for (int i = t; i < n; i += NT) local += cp[i] >= 0 ? 1u : 0u;
    if (local) atomicAdd(&nvalid, local);
    __syncthreads();
    const int nv = (int)nvalid;
    k = nv < cap ? nv : cap;
    if (t == 0) { out_count[b] = k; if (rec) rec[0] = (float)k; }
    if (k == 0) return;
    const unsigned none[1] = {0u};
    block_topk_regs<0, NT>(none, [&](int i) -> unsigned { return cp[i] >= 0 ? f2key(cs[i]) : 0u; }, n, k, sh);
  }
  YMI_STAMP(4);
  for (int j = t; j < k; j += NT) {
    const int f = (int)(0xffffffffu - (unsigned)(sh.comp[j] & 0xffffffffull));
    const int prior = cp[f];
    sel_f[j] = f; sel_prior[j] = prior;
    const f32x4 bb = decode_box(loc + ((size_t)b * P + prior) * 4, priors + (size_t)prior * 4);
    float *ob = out_box + ((size_t)b * cap + j) * 4;
    ob[0] = bb[0]; ob[1] = bb[1]; ob[2] = bb[2]; ob[3] = bb[3];
    const float sv = cs[f];
    out_score[(size_t)b * cap + j] = sv;
    const long long cls = cross_class ? (long long)argmax[(size_t)b * P + prior] : (long long)(f / top_k);
    out_class[(size_t)b * cap + j] = cls;
    out_prior[(size_t)b * cap + j] = prior;
    if (rec) {
      float *r = rec + 1 + (size_t)j * RL;
      r[0] = bb[0]; r[1] = bb[1]; r[2] = bb[2]; r[3] = bb[3]; r[4] = sv; r[5] = (float)cls;
    }
  }
  __syncthreads();
  YMI_STAMP(5);
  // coefficient rows: D floats each, copied by all threads (prior indices from LDS: one dependent load per element, not two)
  for (int i = t; i < k * D; i += NT) {
    const int j = i / D, e = i - j * D;
    const float cv = coef[((size_t)b * P + sel_prior[j]) * D + e];
    out_coef[((size_t)b * cap + j) * D + e] = cv;
    if (rec) rec[1 + (size_t)j * RL + 6 + e] = cv;
  }
  YMI_STAMP(7);
}

}  // namespace

extern "C" int ymi_detect_f32(const ymi_detect_desc *d, void *stream) {
  if (!d) return YMI_ENULL;
  if (!d->conf || !d->loc || !d->coef || !d->priors || !d->scores_t || !d->keep || !d->num_keep || !d->cand_score ||
      !d->cand_prior || !d->out_count || !d->out_box || !d->out_score || !d->out_class || !d->out_coef ||
      !d->out_prior || !d->maxsc || !d->argmax)
    return YMI_ENULL;
  if (d->B <= 0 || d->P <= 0 || d->C < 2 || d->D <= 0 || (d->conf_ld != 0 && d->conf_ld < d->C)) return YMI_EARG;
  if (d->top_k <= 0 || d->top_k > SORT_N || d->max_det <= 0 || d->max_det > SORT_N) return YMI_EARG;
  if (d->B > 65535) return YMI_EARG;
  if ((size_t)64 * d->C * sizeof(float) > 60000) return YMI_ESHAPE;
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(d->num_keep, 0, sizeof(int32_t) * d->B, s);
  if (e != hipSuccess) return (int)e;
  const int nfg = d->C - 1;
  hipLaunchKernelGGL(softmax_keep_k, dim3((d->P + 63) / 64, d->B), dim3(256), 64 * d->C * sizeof(float), s, d->conf,
                     d->P, d->C, d->conf_ld > 0 ? d->conf_ld : d->C, d->conf_is_logits, d->conf_thresh, d->scores_t, d->keep, d->num_keep, d->maxsc,
                     d->argmax);
  int rc = ymi_launch_status();
  if (rc) return rc;
  const int nclass = d->cross_class ? 1 : nfg;
  unsigned long long *trace = nullptr;
  int dbg = 0;   // diagnostics only (env YMI_DETECT_ABLATE): bit0 skip selection, bit2 skip the IoU triangle
#ifdef YMI_DIAGNOSTICS   // `make DIAG=1` only (tools/detect_probe.py): wrong results by design
  { const char *e = getenv("YMI_DETECT_ABLATE"); if (e) dbg = atoi(e); }
  { const char *e = getenv("YMI_DETECT_TRACE"); if (e) trace = (unsigned long long *)strtoull(e, nullptr, 0); }
#endif
  const float *sc = d->cross_class ? d->maxsc : d->scores_t;
#define YMI_K2S(EPT, NTH, SG)                                                                                        \
  hipLaunchKernelGGL((class_topk_nms_k<EPT, NTH, SG>), dim3(nclass, d->B), dim3(NTH), 0, s, sc, d->keep, d->num_keep, d->loc, \
                     d->priors, d->P, nclass, d->top_k, d->nms_thresh, d->cand_score, d->cand_prior, dbg, trace)
#define YMI_K2(EPT, NTH) do { if (d->cross_class) YMI_K2S(EPT, NTH, false); else YMI_K2S(EPT, NTH, true); } while (0)
  // keys per thread x 1024 threads: 19 248 priors (550 px) -> 24, 30 963 (700 px) / 57 744 (YOLACT++) -> 64; beyond 65 536 the
  // keys stay in memory and every bisection pass re-reads them from L2
  if (d->P <= NT * 24) YMI_K2(24, NT);
  else if (d->P <= NT * 64) YMI_K2(64, NT);
  else YMI_K2(0, NT);
#undef YMI_K2
#undef YMI_K2S
  rc = ymi_launch_status();
  if (rc) return rc;
  const int cap = d->cross_class ? d->top_k : d->max_det;
#define YMI_K3(EPT)                                                                                                   \
  hipLaunchKernelGGL(final_topk_k<EPT>, dim3(d->B), dim3(NT), 0, s, d->cand_score, d->cand_prior, d->loc, d->priors, \
                     d->coef, d->argmax, d->P, d->D, nclass, d->top_k, cap, d->cross_class, d->out_count, d->out_box, \
                     d->out_score, (long long *)d->out_class, d->out_coef, d->out_prior, d->out_rec, trace)
  const long ncand = (long)nclass * d->top_k;
  if (ncand <= NT * 16) YMI_K3(16);
  else if (ncand <= NT * 64) YMI_K3(64);
  else YMI_K3(0);
#undef YMI_K3
  return ymi_launch_status();
}
