// Detect on device: softmax + candidate filter, per-class top-k + Fast NMS, final top-N merge.
// Reference: layers/functions/detection.py:32-180 (Detect.__call__/detect/fast_nms/cc_fast_nms),
//            layers/box_utils.py:33-80 (intersect/jaccard), :267-312 (decode).
//
// Whole batch in three launches, no host synchronisation, fixed-capacity outputs + counts:
//   K1  grid (P/64, B)      softmax over C classes, fg max/argmax, keep = max > conf_thresh, class-major scores
//   K2  grid (nclass, B)    radix-select the top_k kept priors of a class (ties: lowest prior index first),
//                           bitonic sort, decode their boxes, IoU upper triangle, keep iou_max <= nms_thresh
//   K3  grid (B)            radix-select + sort the best max_det survivors over all classes, gather outputs
// Tie rule: the reference sorts with the unstable torch.sort; we define stable order (lowest index first),
// SURVEY §7 hard part 3(iv).  All float math mirrors the reference's op order; build with -ffp-contract=off.
#include "common.h"
#include "../../include/yolact_amd.h"

namespace {

constexpr int NT = 256;          // threads per block for K2/K3
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
__global__ __launch_bounds__(256) void softmax_keep_k(const float *__restrict__ conf, int P, int C, int is_logits,
                                                      float thresh, float *__restrict__ scores_t,
                                                      int *__restrict__ keep, int *__restrict__ num_keep,
                                                      float *__restrict__ maxsc, int *__restrict__ argmax) {
  extern __shared__ float s[];  // 64 * C
  __shared__ int blk_cnt;
  const int b = blockIdx.y, p0 = blockIdx.x * 64;
  const int np = min(64, P - p0);
  const int t = threadIdx.x;
  if (t == 0) blk_cnt = 0;
  const float *src = conf + ((size_t)b * P + p0) * C;
  for (int i = t; i < np * C; i += 256) s[i] = src[i];
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
// Block-wide radix select + sort.  key(i) for i in [0,n): 0 = not a candidate.  Selects the k largest keys,
// ties broken by lowest i, and leaves them sorted (key desc, i asc) in sh_key/sh_idx[0..k).
struct SelShared {
  unsigned hist[256];
  unsigned long long comp[SORT_N];
  unsigned prefix, krem, cnt_gt, cnt_eq, sel_eq;
  unsigned wave_tot[NT / 64];
};

template <typename KeyFn>
__device__ void block_topk_sorted(KeyFn key, int n, int k, SelShared &sh) {
  const int t = threadIdx.x;
  unsigned prefix = 0, mask = 0, krem = (unsigned)k, eq_total = 0;
  for (int shift = 24; shift >= 0; shift -= 8) {
    sh.hist[t] = 0;  // NT == 256 bins
    __syncthreads();
    for (int i = t; i < n; i += NT) {
      const unsigned kk = key(i);
      if (kk != 0 && (kk & mask) == prefix) atomicAdd(&sh.hist[(kk >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (t == 0) {
      unsigned acc = 0; int bin = 255;
      for (; bin > 0; --bin) { if (acc + sh.hist[bin] >= krem) break; acc += sh.hist[bin]; }
      sh.prefix = prefix | ((unsigned)bin << shift);
      sh.krem = krem - acc;
      sh.cnt_eq = sh.hist[bin];
    }
    __syncthreads();
    prefix = sh.prefix; krem = sh.krem; eq_total = sh.cnt_eq;
    mask |= 255u << shift;
    __syncthreads();
  }
  const unsigned T = prefix;  // key of the k-th largest; krem of the eq_total elements equal to T are taken
  sh.comp[t] = 0ull;          // SORT_N == NT
  if (t == 0) { sh.cnt_gt = 0; sh.sel_eq = 0; }
  __syncthreads();
  const unsigned n_gt = (unsigned)k - krem;
  if (eq_total == krem) {
    // common case: every element equal to T is taken; slot order is irrelevant (sorted below)
    for (int i = t; i < n; i += NT) {
      const unsigned kk = key(i);
      if (kk != 0 && kk >= T) {
        const unsigned slot = atomicAdd(&sh.cnt_gt, 1u);
        sh.comp[slot] = ((unsigned long long)kk << 32) | (unsigned)(0xffffffffu - (unsigned)i);
      }
    }
  } else {
    // tie on the threshold with more equals than slots: take the lowest indices, in index order
    unsigned base_eq = 0;  // equals seen in earlier chunks (uniform across the block)
    for (int c0 = 0; c0 < n; c0 += NT) {
      const int i = c0 + t;
      const unsigned kk = i < n ? key(i) : 0u;
      const bool gt = kk != 0 && kk > T, eq = kk != 0 && kk == T;
      const unsigned long long bal = __ballot(eq);
      const int lane = t & 63, w = t >> 6;
      if (lane == 0) sh.wave_tot[w] = (unsigned)__popcll(bal);
      __syncthreads();
      unsigned before = base_eq, tot = 0;
      for (int ww = 0; ww < NT / 64; ++ww) { if (ww < w) before += sh.wave_tot[ww]; tot += sh.wave_tot[ww]; }
      before += (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
      if (gt) {
        const unsigned slot = atomicAdd(&sh.cnt_gt, 1u);
        sh.comp[slot] = ((unsigned long long)kk << 32) | (unsigned)(0xffffffffu - (unsigned)i);
      } else if (eq && before < krem) {
        sh.comp[n_gt + before] = ((unsigned long long)kk << 32) | (unsigned)(0xffffffffu - (unsigned)i);
      }
      base_eq += tot;
      __syncthreads();
    }
  }
  __syncthreads();
  // bitonic sort, descending on the 64-bit composite (key desc, index asc)
  for (int size = 2; size <= SORT_N; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      const int partner = t ^ stride;
      if (partner > t) {
        const unsigned long long a = sh.comp[t], b = sh.comp[partner];
        const bool desc = (t & size) == 0;
        if (desc ? (a < b) : (a > b)) { sh.comp[t] = b; sh.comp[partner] = a; }
      }
      __syncthreads();
    }
  }
}

// K2: one block per (class, image)
__global__ __launch_bounds__(NT) void class_topk_nms_k(const float *__restrict__ scores,  // [B,nclass,P]
                                                       const int *__restrict__ keep, const int *__restrict__ num_keep,
                                                       const float *__restrict__ loc, const float *__restrict__ priors,
                                                       int P, int nclass, int top_k, float nms_thresh,
                                                       float *__restrict__ cand_score, int *__restrict__ cand_prior) {
  __shared__ SelShared sh;
  __shared__ float bx[SORT_N][4];
  const int c = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
  const int K = num_keep[b];
  float *cs = cand_score + ((size_t)b * nclass + c) * top_k;
  int *cp = cand_prior + ((size_t)b * nclass + c) * top_k;
  if (K == 0) {
    for (int i = t; i < top_k; i += NT) { cs[i] = -1.f; cp[i] = -1; }
    return;
  }
  const int k = K < top_k ? K : top_k;
  const float *sc = scores + ((size_t)b * nclass + c) * P;
  const int *kp = keep + (size_t)b * P;
  block_topk_sorted([&](int i) -> unsigned { return kp[i] ? f2key(sc[i]) : 0u; }, P, k, sh);

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
  if (t < top_k) {
    bool kept = false;
    if (t < k) {
      // iou_max[j] = max_{i<j} IoU(i, j) (0 for j = 0); jaccard(): inter / (area_i + area_j - inter)
      const float x1 = bx[t][0], y1 = bx[t][1], x2 = bx[t][2], y2 = bx[t][3];
      const float area_j = (x2 - x1) * (y2 - y1);
      float m = 0.f;
      bool nan = false;
      for (int i = 0; i < t; ++i) {
        const float ax1 = bx[i][0], ay1 = bx[i][1], ax2 = bx[i][2], ay2 = bx[i][3];
        float iw = fminf(ax2, x2) - fmaxf(ax1, x1);
        float ih = fminf(ay2, y2) - fmaxf(ay1, y1);
        iw = iw < 0.f ? 0.f : iw;
        ih = ih < 0.f ? 0.f : ih;
        const float inter = iw * ih;
        const float area_i = (ax2 - ax1) * (ay2 - ay1);
        const float iou = inter / ((area_i + area_j) - inter);
        if (iou != iou) nan = true;
        m = iou > m ? iou : m;
      }
      kept = !nan && (m <= nms_thresh);
    }
    cs[t] = kept ? score : -1.f;
    cp[t] = kept ? prior : -1;
  }
}

// K3: one block per image: best max_det over all per-class survivors (flattened class-major, rank-minor)
__global__ __launch_bounds__(NT) void final_topk_k(const float *__restrict__ cand_score, const int *__restrict__ cand_prior,
                                                   const float *__restrict__ loc, const float *__restrict__ priors,
                                                   const float *__restrict__ coef, const int *__restrict__ argmax,
                                                   int P, int D, int nclass, int top_k, int cap, int cross_class,
                                                   int *__restrict__ out_count, float *__restrict__ out_box,
                                                   float *__restrict__ out_score, long long *__restrict__ out_class,
                                                   float *__restrict__ out_coef, int *__restrict__ out_prior) {
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
  if (k == 0) return;
  block_topk_sorted([&](int i) -> unsigned { return cp[i] >= 0 ? f2key(cs[i]) : 0u; }, n, k, sh);
  for (int j = t; j < k; j += NT) {
    const int f = (int)(0xffffffffu - (unsigned)(sh.comp[j] & 0xffffffffull));
    const int prior = cp[f];
    const f32x4 bb = decode_box(loc + ((size_t)b * P + prior) * 4, priors + (size_t)prior * 4);
    float *ob = out_box + ((size_t)b * cap + j) * 4;
    ob[0] = bb[0]; ob[1] = bb[1]; ob[2] = bb[2]; ob[3] = bb[3];
    out_score[(size_t)b * cap + j] = cs[f];
    out_class[(size_t)b * cap + j] = cross_class ? (long long)argmax[(size_t)b * P + prior] : (long long)(f / top_k);
    out_prior[(size_t)b * cap + j] = prior;
  }
  // coefficient rows: D floats each, copied by all threads
  for (int i = t; i < k * D; i += NT) {
    const int j = i / D, e = i - j * D;
    const int f = (int)(0xffffffffu - (unsigned)(sh.comp[j] & 0xffffffffull));
    out_coef[((size_t)b * cap + j) * D + e] = coef[((size_t)b * P + cp[f]) * D + e];
  }
}

}  // namespace

extern "C" int ymi_detect_f32(const ymi_detect_desc *d, void *stream) {
  if (!d) return YMI_ENULL;
  if (!d->conf || !d->loc || !d->coef || !d->priors || !d->scores_t || !d->keep || !d->num_keep || !d->cand_score ||
      !d->cand_prior || !d->out_count || !d->out_box || !d->out_score || !d->out_class || !d->out_coef ||
      !d->out_prior || !d->maxsc || !d->argmax)
    return YMI_ENULL;
  if (d->B <= 0 || d->P <= 0 || d->C < 2 || d->D <= 0) return YMI_EARG;
  if (d->top_k <= 0 || d->top_k > SORT_N || d->max_det <= 0 || d->max_det > SORT_N) return YMI_EARG;
  if (d->B > 65535) return YMI_EARG;
  if ((size_t)64 * d->C * sizeof(float) > 60000) return YMI_ESHAPE;
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(d->num_keep, 0, sizeof(int32_t) * d->B, s);
  if (e != hipSuccess) return (int)e;
  const int nfg = d->C - 1;
  hipLaunchKernelGGL(softmax_keep_k, dim3((d->P + 63) / 64, d->B), dim3(256), 64 * d->C * sizeof(float), s, d->conf,
                     d->P, d->C, d->conf_is_logits, d->conf_thresh, d->scores_t, d->keep, d->num_keep, d->maxsc,
                     d->argmax);
  int rc = ymi_launch_status();
  if (rc) return rc;
  const int nclass = d->cross_class ? 1 : nfg;
  const float *sc = d->cross_class ? d->maxsc : d->scores_t;
  hipLaunchKernelGGL(class_topk_nms_k, dim3(nclass, d->B), dim3(NT), 0, s, sc, d->keep, d->num_keep, d->loc, d->priors,
                     d->P, nclass, d->top_k, d->nms_thresh, d->cand_score, d->cand_prior);
  rc = ymi_launch_status();
  if (rc) return rc;
  const int cap = d->cross_class ? d->top_k : d->max_det;
  hipLaunchKernelGGL(final_topk_k, dim3(d->B), dim3(NT), 0, s, d->cand_score, d->cand_prior, d->loc, d->priors, d->coef,
                     d->argmax, d->P, d->D, nclass, d->top_k, cap, d->cross_class, d->out_count, d->out_box,
                     d->out_score, (long long *)d->out_class, d->out_coef, d->out_prior);
  return ymi_launch_status();
}
