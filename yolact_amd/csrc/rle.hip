// COCO result wire format on device: the run-length encoding that eval.py's Detections.add_mask
// obtains from pycocotools.mask.encode(np.asfortranarray(mask.astype(np.uint8))) (eval.py:320-324), and the ASCII
// compression of the counts (pycocotools maskApi.c rleEncode / rleToString; restated in oracle/coco_rle.py).
// The reference copies N*h*w floats (121 MB per image at 550 x 550, N = 100) to the host and encodes them there; here
// only the counts / strings (a few KB per mask) leave the device.
//
// rle_counts_k: one block per mask.  The sequence is the COLUMN-major flattening of the row-major mask, so a thread owns
//   a (column, row band) segment and walks it downwards: a wave's 64 lanes read 64 consecutive columns of one row (256
//   coalesced bytes per step) and transitions of a segment come out already in sequence order.  Pass 1 counts the
//   transitions and remembers the last transition position per segment; a block scan over the segments in sequence order
//   (column outer, band inner) yields each segment's output offset (sum) and its predecessor transition (max); pass 2
//   (reads served by L2: a mask is 1.2 MB) writes the run lengths directly.  HBM-bound: N*h*w*4 bytes read once.
// rle_string_k: one block per mask; thread i turns count i (minus count i-2 for i > 2) into 1..7 characters (5 bits each,
//   0x20 = continuation, + 48), a block scan of the lengths places them.
#include "common.h"
#include "../../include/yolact_amd.h"

namespace {

constexpr int RB = 4;          // row bands per column
constexpr int NT = 1024;       // threads per block (16 waves)
constexpr int UNR = 8;         // independent loads in flight per thread

struct SegRange { int x, y0, y1; };

__device__ __forceinline__ SegRange seg_of(int q, int h, int w, int hb) {
  // iteration order q: band-major so that consecutive lanes read consecutive columns of the same row
  const int r = q / w, x = q - r * w;
  int y0 = r * hb, y1 = y0 + hb;
  if (y0 > h) y0 = h;
  if (y1 > h) y1 = h;
  return {x, y0, y1};
}

__device__ __forceinline__ int seg_prev(const float *m, int x, int y0, int h, int w) {
  if (y0 > 0) return m[(size_t)(y0 - 1) * w + x] != 0.f;
  if (x > 0) return m[(size_t)(h - 1) * w + (x - 1)] != 0.f;
  return 0;                                        // rleEncode starts with p = 0
}

// exclusive block scan of (sum, max) over NT per-thread partials
__device__ __forceinline__ void block_scan(uint32_t &sum, int &mx, uint32_t *ws_sum, int *ws_max) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t s = sum;
  int m = mx;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t so = __shfl_up(s, d);
    const int mo = __shfl_up(m, d);
    if (lane >= d) { s += so; m = m > mo ? m : mo; }
  }
  if (lane == 63) { ws_sum[wave] = s; ws_max[wave] = m; }
  __syncthreads();
  uint32_t base = 0;
  int bmax = -1;
  for (int k = 0; k < wave; ++k) { base += ws_sum[k]; bmax = bmax > ws_max[k] ? bmax : ws_max[k]; }
  // inclusive -> exclusive
  const uint32_t se = __shfl_up(s, 1);
  const int me = __shfl_up(m, 1);
  sum = base + (lane ? se : 0u);
  const int mprev = lane ? me : -1;
  mx = bmax > mprev ? bmax : mprev;
  __syncthreads();
}

__global__ __launch_bounds__(NT) void rle_counts_k(const float *__restrict__ masks, int h, int w, uint32_t *__restrict__ counts,
                                                  int32_t *__restrict__ nruns, int cap) {
  extern __shared__ uint32_t lds[];
  const int nseg = w * RB;
  uint32_t *cnt = lds;                              // [nseg] in sequence order (x * RB + r): transitions, then offsets
  int *last = reinterpret_cast<int *>(lds + nseg);  // [nseg] last transition position of the segment, then predecessor
  __shared__ uint32_t ws_sum[NT / 64];
  __shared__ int ws_max[NT / 64];
  __shared__ uint32_t total_s;
  __shared__ int lastpos_s;
  const float *m = masks + (size_t)blockIdx.x * h * w;
  const int hb = (h + RB - 1) / RB;

  for (int q = threadIdx.x; q < nseg; q += NT) {
    const SegRange sg = seg_of(q, h, w, hb);
    int prev = sg.y1 > sg.y0 ? seg_prev(m, sg.x, sg.y0, h, w) : 0;
    uint32_t c = 0;
    int lp = -1;
    int y = sg.y0;
    for (; y + UNR <= sg.y1; y += UNR) {
      float v[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) v[u] = m[(size_t)(y + u) * w + sg.x];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int b = v[u] != 0.f;
        if (b != prev) { ++c; lp = sg.x * h + y + u; prev = b; }
      }
    }
    for (; y < sg.y1; ++y) {
      const int b = m[(size_t)y * w + sg.x] != 0.f;
      if (b != prev) { ++c; lp = sg.x * h + y; prev = b; }
    }
    const int r = q / w;
    cnt[sg.x * RB + r] = c;
    last[sg.x * RB + r] = lp;
  }
  __syncthreads();

  // scan in sequence order: thread t owns the contiguous chunk [t*per, (t+1)*per)
  const int per = (nseg + NT - 1) / NT;
  const int s0 = threadIdx.x * per;
  const int s1 = s0 + per < nseg ? s0 + per : nseg;
  uint32_t psum = 0;
  int pmax = -1;
  for (int s = s0; s < s1; ++s) { psum += cnt[s]; pmax = pmax > last[s] ? pmax : last[s]; }
  uint32_t tsum = psum;
  int tmax = pmax;
  block_scan(psum, pmax, ws_sum, ws_max);           // now exclusive over threads
  if (threadIdx.x == NT - 1) { total_s = psum + tsum; lastpos_s = pmax > tmax ? pmax : tmax; }
  for (int s = s0; s < s1; ++s) {
    const uint32_t c = cnt[s];
    const int l = last[s];
    cnt[s] = psum;
    last[s] = pmax;
    psum += c;
    pmax = pmax > l ? pmax : l;
  }
  __syncthreads();

  uint32_t *out = counts + (size_t)blockIdx.x * cap;
  for (int q = threadIdx.x; q < nseg; q += NT) {
    const SegRange sg = seg_of(q, h, w, hb);
    if (sg.y1 <= sg.y0) continue;
    const int r = q / w;
    uint32_t k = cnt[sg.x * RB + r];
    int pp = last[sg.x * RB + r];                  // position of the previous transition (-1: none => run starts at 0)
    if (pp < 0) pp = 0;
    int prev = seg_prev(m, sg.x, sg.y0, h, w);
    int y = sg.y0;
    for (; y + UNR <= sg.y1; y += UNR) {
      float v[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) v[u] = m[(size_t)(y + u) * w + sg.x];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int b = v[u] != 0.f;
        if (b != prev) {
          const int p = sg.x * h + y + u;
          if (k < (uint32_t)cap) out[k] = (uint32_t)(p - pp);
          ++k; pp = p; prev = b;
        }
      }
    }
    for (; y < sg.y1; ++y) {
      const int b = m[(size_t)y * w + sg.x] != 0.f;
      if (b != prev) {
        const int p = sg.x * h + y;
        if (k < (uint32_t)cap) out[k] = (uint32_t)(p - pp);
        ++k; pp = p; prev = b;
      }
    }
  }
  if (threadIdx.x == 0) {
    const uint32_t k = total_s;                    // number of transitions; the final run closes the sequence
    const int lp = lastpos_s < 0 ? 0 : lastpos_s;
    if (k < (uint32_t)cap) out[k] = (uint32_t)(h * w - lp);
    nruns[blockIdx.x] = (int32_t)(k + 1);
  }
}

__global__ __launch_bounds__(256) void rle_string_k(const uint32_t *__restrict__ counts, const int32_t *__restrict__ nruns, int cap,
                                                   uint8_t *__restrict__ str, int32_t *__restrict__ nchars, int cap_chars) {
  __shared__ uint32_t ws[4];
  __shared__ uint32_t carry_s;
  const uint32_t *c = counts + (size_t)blockIdx.x * cap;
  uint8_t *o = str + (size_t)blockIdx.x * cap_chars;
  int m = nruns[blockIdx.x];
  if (m > cap) m = cap;                            // truncated encodings are reported through nruns > cap by the caller
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i0 = 0; i0 < m; i0 += 256) {
    const int i = i0 + threadIdx.x;
    uint64_t chars = 0;
    uint32_t len = 0;
    if (i < m) {
      long x = (long)c[i];
      if (i > 2) x -= (long)c[i - 2];
      bool more = true;
      while (more) {
        uint32_t ch = (uint32_t)(x & 0x1f);
        x >>= 5;
        more = (ch & 0x10) ? x != -1 : x != 0;
        if (more) ch |= 0x20;
        chars |= (uint64_t)(ch + 48) << (8 * len);
        ++len;
      }
    }
    uint32_t s = len;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t so = __shfl_up(s, d);
      if (lane >= d) s += so;
    }
    if (lane == 63) ws[wave] = s;
    __syncthreads();
    uint32_t off = carry_s + s - len;
    for (int k = 0; k < wave; ++k) off += ws[k];
    for (uint32_t j = 0; j < len; ++j)
      if (off + j < (uint32_t)cap_chars) o[off + j] = (uint8_t)(chars >> (8 * j));
    __syncthreads();
    if (threadIdx.x == 255) carry_s = off + len;
    __syncthreads();
  }
  if (threadIdx.x == 0) nchars[blockIdx.x] = (int32_t)carry_s;
}

}  // namespace

extern "C" int ymi_mask_rle_f32(const float *masks, int N, int h, int w, uint32_t *counts, int32_t *nruns, int cap,
                                void *stream) {
  if (N < 0 || h <= 0 || w <= 0 || cap <= 0) return YMI_EARG;
  if (N == 0) return YMI_OK;
  if (!masks || !counts || !nruns) return YMI_ENULL;
  if ((long)h * w >= (1L << 31) || w * RB * 8 > 65536) return YMI_ESHAPE;      // segment tables live in LDS: w <= 2048
  hipLaunchKernelGGL(rle_counts_k, dim3(N), dim3(NT), (size_t)w * RB * 8, (hipStream_t)stream, masks, h, w, counts, nruns, cap);
  return ymi_launch_status();
}

extern "C" int ymi_rle_to_string(const uint32_t *counts, const int32_t *nruns, int N, int cap, uint8_t *str, int32_t *nchars,
                                 int cap_chars, void *stream) {
  if (N < 0 || cap <= 0 || cap_chars <= 0) return YMI_EARG;
  if (N == 0) return YMI_OK;
  if (!counts || !nruns || !str || !nchars) return YMI_ENULL;
  hipLaunchKernelGGL(rle_string_k, dim3(N), dim3(256), 0, (hipStream_t)stream, counts, nruns, cap, str, nchars, cap_chars);
  return ymi_launch_status();
}
