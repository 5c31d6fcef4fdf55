// Detect on device: softmax + candidate filter, per-class top-k + Fast NMS, final top-N merge.
// Reference: layers/functions/detection.py:32-180 (Detect.__call__/detect/fast_nms/cc_fast_nms),
//            layers/box_utils.py:33-80 (intersect/jaccard), :267-312 (decode).
//
// Whole batch in three launches, no host synchronisation, fixed-capacity outputs + counts:
//   K1  grid (P/64, B)      softmax over C classes, fg max/argmax, keep = max > conf_thresh, class-major scores
//   K2  grid (nclass, B)    select the top_k kept priors of a class (register-resident bisection; ties: lowest prior
//                           index first), bitonic sort, decode their boxes, folded IoU upper triangle,
//                           keep iou_max <= nms_thresh
//   K3  grid (B)            select + sort the best max_det survivors over all classes, gather outputs
// Tie rule: the reference sorts with the unstable torch.sort; we define stable order (lowest index first),
// SURVEY §7 hard part 3(iv).  All float math mirrors the reference's op order; build with -ffp-contract=off.
#include "common.h"
#include <stdlib.h>
#include "../../include/yolact_amd.h"

namespace {

constexpr int NT = 256;          // threads per block for K3 and the default K2
constexpr int NT_MAX = 512;      // K2 with 57 744 priors (YOLACT++): 512 threads x 128 keys
constexpr int SORT_N = 256;      // bitonic capacity (top_k, max_det <= 256)

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
    maxsc[(size_t)b * P + p0 + j] = best;
    argmax[(size_t)b * P + p0 + j] = bi;
    if (kp) atomicAdd(&blk_cnt, 1);
  }
  __syncthreads();
  if (t == 0 && blk_cnt) atomicAdd(&num_keep[b], blk_cnt);
  // class-major store: scores_t[b][c-1][p0 + j]
  const int nfg = C - 1;
  for (int i = t; i < nfg * 64; i += 256) {
    const int c = i >> 6, jj = i & 63;
    if (jj < np) scores_t[((size_t)b * nfg + c) * P + p0 + jj] = s[jj * C + c + 1];
  }
}

// ------------------------------------------------------------------------------------------------
// Block-wide top-k selection + sort.  Keys: order-preserving uint of the score, 0 = not a candidate.  Selects the k
// largest keys, ties broken by lowest index, and leaves them sorted (key desc, index asc) in sh.comp[0..k).
struct SelShared {
  unsigned hist[256];             // [2][NT_MAX / 64] per-wave partial counts of the bisection passes
  unsigned long long comp[SORT_N];
  unsigned prefix, krem, cnt_gt, cnt_eq, sel_eq;
  unsigned wave_tot[NT_MAX / 64];
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

// ------------------------------------------------------------------------------------------------
// Register-resident top-k: every thread holds EPT keys (element i = t + NT*j) in VGPRs and the k-th largest key is
// found by bisection on its 32 bits (one block-wide count per bit: compares on registers, a wave reduction and ONE
// barrier) — no memory traffic and no atomics in the search.  The LDS-histogram radix select above needs 4 passes
// over global memory and serialises on a handful of hot bins (softmax scores share their top byte): 282 us for
// the 640 (class, image) blocks of a batch-8 step, vs ~25 us this way.
// Result: the selected elements sorted (key desc, index asc) in sh.comp[0..k) like block_topk_sorted.
// EPT > 0: keys[] holds element t + NT*j in registers.  EPT == 0 (n too large for the register file): the same
// algorithm re-evaluates key(i) from memory (L2-resident) on every pass.
template <int EPT, int NTH, typename KeyFn>
__device__ void block_topk_regs(const unsigned (&keys)[EPT > 0 ? EPT : 1], KeyFn key, int n, int k, SelShared &sh) {
  constexpr int NW = NTH / 64;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int rounds = EPT > 0 ? EPT : (n + NTH - 1) / NTH;
  auto key_at = [&](int j) -> unsigned {          // EPT == 0 only
    const int i = t + NTH * j;
    return i < n ? key(i) : 0u;
  };
  auto block_count = [&](unsigned cand, bool strict, int slot) -> unsigned {
    // per-lane counts (compare + add-with-carry on the VALU), then ONE DPP wave reduction: no LDS crossbar shuffles,
    // no SGPR pressure (80 ballots per pass made the compiler spill SGPRs through v_writelane)
    unsigned c = 0;
    if (EPT > 0) {
      unsigned c4[4] = {0u, 0u, 0u, 0u};     // four independent carry chains
#pragma unroll
      for (int j = 0; j < (EPT > 0 ? EPT : 1); ++j) c4[j & 3] += (strict ? keys[j] > cand : keys[j] >= cand) ? 1u : 0u;
      c = (c4[0] + c4[1]) + (c4[2] + c4[3]);
    } else {
      for (int j = 0; j < rounds; ++j) {
        const unsigned kk = key_at(j);
        c += (strict ? kk > cand : kk >= cand) ? 1u : 0u;
      }
    }
    c = wave_sum_to_lane63(c);
    if (lane == 63) sh.hist[slot * NW + w] = c;
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
  const unsigned n_gt = block_count(T, true, 0);      // keys strictly above the threshold: all taken
  const unsigned n_ge = block_count(T, false, 1);
  const unsigned krem = (unsigned)k - n_gt;            // how many of the keys == T are taken (lowest indices first)
  if (t < SORT_N) sh.comp[t] = 0ull;
  if (t == 0) { sh.cnt_gt = 0; }
  __syncthreads();
  if (n_ge - n_gt == krem) {
    // common case: every key equal to T is taken; slot order is irrelevant (sorted below)
    auto take = [&](unsigned kj, int j) {
      const bool win = kj != 0 && kj >= T;
      const unsigned long long bal = __ballot(win);
      if (bal) {                                       // wave-uniform
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(&sh.cnt_gt, (unsigned)__popcll(bal));
        base = __shfl(base, 0);
        if (win) {
          const unsigned slot = base + (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
          sh.comp[slot] = ((unsigned long long)kj << 32) | (unsigned)(0xffffffffu - (unsigned)(t + NTH * j));
        }
      }
    };
    if constexpr (EPT > 0) {
#pragma unroll
      for (int j = 0; j < EPT; ++j) take(keys[j], j);
    } else {
      for (int j = 0; j < rounds; ++j) take(key_at(j), j);
    }
  } else {
    // tie on the threshold with more equals than slots: index order = (j, wave, lane); rare, one barrier per round
    unsigned base_eq = 0;
    for (int j = 0; j < rounds; ++j) {
      unsigned kk = 0;
      if (EPT > 0) {
#pragma unroll
        for (int jj = 0; jj < (EPT > 0 ? EPT : 1); ++jj) if (jj == j) kk = keys[jj];   // register array: no dynamic indexing
      } else {
        kk = key_at(j);
      }
      const bool gt = kk != 0 && kk > T, eq = kk != 0 && kk == T;
      const unsigned long long bal = __ballot(eq);
      if (lane == 0) sh.wave_tot[w] = (unsigned)__popcll(bal);
      __syncthreads();
      unsigned before = base_eq, tot = 0;
      for (int ww = 0; ww < NW; ++ww) { if (ww < w) before += sh.wave_tot[ww]; tot += sh.wave_tot[ww]; }
      before += (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
      const unsigned long long comp = ((unsigned long long)kk << 32) | (unsigned)(0xffffffffu - (unsigned)(t + NTH * j));
      if (gt) sh.comp[atomicAdd(&sh.cnt_gt, 1u)] = comp;
      else if (eq && before < krem) sh.comp[n_gt + before] = comp;
      base_eq += tot;
      __syncthreads();
    }
  }
  __syncthreads();
  // bitonic sort, descending on the 64-bit composite (key desc, index asc)
  for (int size = 2; size <= SORT_N; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      const int partner = t ^ stride;
      if (partner > t && partner < SORT_N) {
        const unsigned long long a = sh.comp[t], b = sh.comp[partner];
        const bool desc = (t & size) == 0;
        if (desc ? (a < b) : (a > b)) { sh.comp[t] = b; sh.comp[partner] = a; }
      }
      __syncthreads();
    }
  }
}

// max_{i in [r0,r1)} IoU(box i, box j) exactly as jaccard() evaluates it (box_utils.py:47-51,72-79):
// inter / (area_i + area_j - inter).  A NaN IoU poisons the column with +inf (the reference's `iou_max <= thresh`
// is False for NaN).
__device__ __forceinline__ float iou_colmax(const float (*bx)[4], int j, int r0, int r1) {
  const float x1 = bx[j][0], y1 = bx[j][1], x2 = bx[j][2], y2 = bx[j][3];
  const float area_j = (x2 - x1) * (y2 - y1);
  float m = 0.f;
  for (int i = r0; i < r1; ++i) {
    const float ax1 = bx[i][0], ay1 = bx[i][1], ax2 = bx[i][2], ay2 = bx[i][3];
    float iw = fminf(ax2, x2) - fmaxf(ax1, x1);
    float ih = fminf(ay2, y2) - fmaxf(ay1, y1);
    iw = iw < 0.f ? 0.f : iw;
    ih = ih < 0.f ? 0.f : ih;
    const float inter = iw * ih;
    const float area_i = (ax2 - ax1) * (ay2 - ay1);
    const float iou = inter / ((area_i + area_j) - inter);
    m = (iou != iou) ? __builtin_inff() : (iou > m ? iou : m);
  }
  return m;
}

// K2: one block per (class, image).  EPT = keys per thread (0: keys stay in memory)
template <int EPT, int NTH>
__global__ __launch_bounds__(NTH, (NTH / 256) * (EPT > 80 ? 1 : 2)) void class_topk_nms_k(const float *__restrict__ scores,  // [B,nclass,P]
                                                       const int *__restrict__ keep, const int *__restrict__ num_keep,
                                                       const float *__restrict__ loc, const float *__restrict__ priors,
                                                       int P, int nclass, int top_k, float nms_thresh,
                                                       float *__restrict__ cand_score, int *__restrict__ cand_prior,
                                                       int dbg) {
  __shared__ SelShared sh;
  __shared__ float bx[SORT_N][4];
  __shared__ float pm_long[SORT_N], pm_short[SORT_N];
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
  if constexpr (EPT > 0) {
    unsigned keys[EPT];
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
      const int i = t + NTH * j, ii = i < P ? i : 0;   // unconditional, independent loads: all 2*EPT stay in flight
      const int kf = kp[ii];
      const float sv = sc[ii];
      keys[j] = (i < P && kf) ? f2key(sv) : 0u;
    }
    if (dbg & 1) { sh.comp[t] = ((unsigned long long)keys[0] << 32) | (unsigned)(0xffffffffu - (unsigned)t); __syncthreads(); }
    else block_topk_regs<EPT, NTH>(keys, [](int) -> unsigned { return 0u; }, P, k, sh);
  } else {
    const unsigned none[1] = {0u};
    block_topk_regs<0, NTH>(none, [&](int i) -> unsigned { return kp[i] ? f2key(sc[i]) : 0u; }, P, k, sh);
  }

  // rank t -> prior index, score, decoded box
  int prior = -1;
  float score = -1.f;
  if (t < k) {
    prior = (int)(0xffffffffu - (unsigned)(sh.comp[t] & 0xffffffffull));
    score = sc[prior];
    const f32x4 bb = decode_box(loc + ((size_t)b * P + prior) * 4, priors + (size_t)prior * 4);
    bx[t][0] = bb[0]; bx[t][1] = bb[1]; bx[t][2] = bb[2]; bx[t][3] = bb[3];
  }
  __syncthreads();
  // iou_max[j] = max_{i<j} IoU(i, j) (0 for j = 0); keep = iou_max <= nms_thresh.  Column j costs j IoUs, so the long
  // columns j in [k-h, k) are folded with the short columns k-1-j in [0, h), h = k/2, onto two threads of ~h IoUs
  // each: helper u < h does rows [0, h) of column k-1-u; owner j does its rows [h, j) and the whole column k-1-j.
  const int h = k / 2;
  float m_own = 0.f;
  if (dbg & 4) {
  } else if (t < h) {
    pm_long[k - 1 - t] = iou_colmax(bx, k - 1 - t, 0, h);
  } else if (t < k) {
    if (t >= k - h) {
      m_own = iou_colmax(bx, t, h, t);
      pm_short[k - 1 - t] = iou_colmax(bx, k - 1 - t, 0, k - 1 - t);
    } else {
      m_own = iou_colmax(bx, t, 0, t);    // middle column (k odd)
    }
  }
  __syncthreads();
  if (t < top_k) {
    bool kept = false;
    if (t < k) {
      const float m = t < h ? pm_short[t] : (t >= k - h ? fmaxf(m_own, pm_long[t]) : m_own);
      kept = m <= nms_thresh;
    }
    cs[t] = kept ? score : -1.f;
    cp[t] = kept ? prior : -1;
  }
}

// K3: one block per image: best max_det over all per-class survivors (flattened class-major, rank-minor)
template <int EPT>
__global__ __launch_bounds__(NT, (EPT > 80 ? 1 : 2)) void final_topk_k(const float *__restrict__ cand_score, const int *__restrict__ cand_prior,
                                                   const float *__restrict__ loc, const float *__restrict__ priors,
                                                   const float *__restrict__ coef, const int *__restrict__ argmax,
                                                   int P, int D, int nclass, int top_k, int cap, int cross_class,
                                                   int *__restrict__ out_count, float *__restrict__ out_box,
                                                   float *__restrict__ out_score, long long *__restrict__ out_class,
                                                   float *__restrict__ out_coef, int *__restrict__ out_prior,
                                                   float *__restrict__ out_rec) {
  __shared__ SelShared sh;
  __shared__ unsigned nvalid;
  const int b = blockIdx.x, t = threadIdx.x;
  const int n = nclass * top_k;
  const float *cs = cand_score + (size_t)b * n;
  const int *cp = cand_prior + (size_t)b * n;
  if (t == 0) nvalid = 0;
  __syncthreads();
  unsigned local = 0;
  for (int i = t; i < n; i += NT) local += cp[i] >= 0 ? 1u : 0u;
  if (local) atomicAdd(&nvalid, local);
  __syncthreads();
  const int nv = (int)nvalid;
  const int k = nv < cap ? nv : cap;
  if (t == 0) out_count[b] = k;
  // packed record of image b (ymi_detect_desc.out_rec): count | cap x (box 4, score, class, coef D), all fp32
  const int RL = 6 + D;
  float *rec = out_rec ? out_rec + (size_t)b * (1 + (size_t)cap * RL) : nullptr;
  if (rec && t == 0) rec[0] = (float)k;
  if (k == 0) return;
  if constexpr (EPT > 0) {
    unsigned keys[EPT];
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
      const int i = t + NT * j, ii = i < n ? i : 0;
      const int pf = cp[ii];
      const float sv = cs[ii];
      keys[j] = (i < n && pf >= 0) ? f2key(sv) : 0u;
    }
    block_topk_regs<EPT, NT>(keys, [](int) -> unsigned { return 0u; }, n, k, sh);
  } else {
    const unsigned none[1] = {0u};
    block_topk_regs<0, NT>(none, [&](int i) -> unsigned { return cp[i] >= 0 ? f2key(cs[i]) : 0u; }, n, k, sh);
  }
  for (int j = t; j < k; j += NT) {
    const int f = (int)(0xffffffffu - (unsigned)(sh.comp[j] & 0xffffffffull));
    const int prior = cp[f];
    const f32x4 bb = decode_box(loc + ((size_t)b * P + prior) * 4, priors + (size_t)prior * 4);
    float *ob = out_box + ((size_t)b * cap + j) * 4;
    ob[0] = bb[0]; ob[1] = bb[1]; ob[2] = bb[2]; ob[3] = bb[3];
    out_score[(size_t)b * cap + j] = cs[f];
    const long long cls = cross_class ? (long long)argmax[(size_t)b * P + prior] : (long long)(f / top_k);
    out_class[(size_t)b * cap + j] = cls;
    out_prior[(size_t)b * cap + j] = prior;
    if (rec) {
      float *r = rec + 1 + (size_t)j * RL;
      r[0] = bb[0]; r[1] = bb[1]; r[2] = bb[2]; r[3] = bb[3]; r[4] = cs[f]; r[5] = (float)cls;
    }
  }
  // coefficient rows: D floats each, copied by all threads
  for (int i = t; i < k * D; i += NT) {
    const int j = i / D, e = i - j * D;
    const int f = (int)(0xffffffffu - (unsigned)(sh.comp[j] & 0xffffffffull));
    const float cv = coef[((size_t)b * P + cp[f]) * D + e];
    out_coef[((size_t)b * cap + j) * D + e] = cv;
    if (rec) rec[1 + (size_t)j * RL + 6 + e] = cv;
  }
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
  int dbg = 0;   // diagnostics only (env YMI_DETECT_ABLATE): bit0 skip selection, bit2 skip the IoU triangle
#ifdef YMI_DIAGNOSTICS   // `make DIAG=1` only (tools/detect_probe.py): wrong results by design
  { const char *e = getenv("YMI_DETECT_ABLATE"); if (e) dbg = atoi(e); }
#endif
  const float *sc = d->cross_class ? d->maxsc : d->scores_t;
#define YMI_K2(EPT, NTH)                                                                                             \
  hipLaunchKernelGGL((class_topk_nms_k<EPT, NTH>), dim3(nclass, d->B), dim3(NTH), 0, s, sc, d->keep, d->num_keep, d->loc, \
                     d->priors, d->P, nclass, d->top_k, d->nms_thresh, d->cand_score, d->cand_prior, dbg)
  // keys per thread x threads: 19 248 priors (550 px) -> 80 x 256, 30 963 (700 px) -> 128 x 256, 57 744 (YOLACT++) ->
  // 128 x 512; beyond 65 536 the keys stay in memory and every bisection pass re-reads them from L2
  if (d->P <= 256 * 80) YMI_K2(80, 256);
  else if (d->P <= 256 * 128) YMI_K2(128, 256);
  else if (d->P <= 512 * 128) YMI_K2(128, 512);
  else YMI_K2(0, 256);
#undef YMI_K2
  rc = ymi_launch_status();
  if (rc) return rc;
  const int cap = d->cross_class ? d->top_k : d->max_det;
#define YMI_K3(EPT)                                                                                                   \
  hipLaunchKernelGGL(final_topk_k<EPT>, dim3(d->B), dim3(NT), 0, s, d->cand_score, d->cand_prior, d->loc, d->priors, \
                     d->coef, d->argmax, d->P, d->D, nclass, d->top_k, cap, d->cross_class, d->out_count, d->out_box, \
                     d->out_score, (long long *)d->out_class, d->out_coef, d->out_prior, d->out_rec)
  const long ncand = (long)nclass * d->top_k;
  if (ncand <= NT * 64) YMI_K3(64);
  else if (ncand <= NT * 128) YMI_K3(128);
  else YMI_K3(0);
#undef YMI_K3
  return ymi_launch_status();
}
