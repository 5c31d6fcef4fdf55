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
//   The kernel is templated on where a mask's bits come from: a materialised [h,w] float mask (ymi_mask_rle_f32) or the
//   prototype-resolution sigmoid mask of postprocess, upsampled and thresholded on the fly (ymi_mask_rle_upsampled_f32):
//   then the full-resolution masks are never written and the kernel reads N*ph*pw*4 bytes (L1 / L2-resident per block).
// rle_string_k: one block per mask; thread i turns count i (minus count i-2 for i > 2) into 1..7 characters (5 bits each,
//   0x20 = continuation, + 48), a block scan of the lengths places them.
#include "common.h"
#include "upsample_math.h"
#include "../../include/yolact_amd.h"

namespace {

// Where the bits of a mask come from.  A source hands out a per-column context (the thread walks one column downwards)
// and the bit of a row of that column.
//   PlainSrc  a materialised [h,w] float mask (nonzero = foreground);
//   UpSrc     the low-resolution sigmoid mask of postprocess (output_utils.py:69-94): bit = bilinear(lo)[y,x] > thresh with
//             exactly mask_upsample's arithmetic (upsample_math.h), so the [N,h,w] float masks never exist — the COCO result
//             path reads N*ph*pw*4 bytes (7.6 MB at N = 100, 138 x 138) instead of writing and re-reading N*h*w*4 (121 MB).
struct PlainSrc {
  const float *m;
  int w;
  struct Col { int x; };
  __device__ __forceinline__ void prepare(uint32_t *, int) {}
  __device__ __forceinline__ Col col(int x) const { return {x}; }
  __device__ __forceinline__ int bit(Col &c, int y) const { return m[(size_t)y * w + c.x] != 0.f; }
};

struct UpSrc {
  const float *lo;
  int ph, pw;
  float sh, sw, thresh;
  const uint32_t *ytab;       // LDS: per output row {y0, y1, bits of ly}: the row coordinates are the same for every column
  // the column context caches the two source rows it last touched: walking down a column, a source row serves ~h/ph
  // consecutive output rows
  struct Col { int x0, x1; float lx; int cy0, cy1; float v00, v01, v10, v11; };
  __device__ __forceinline__ void prepare(uint32_t *extra, int h) {
    for (int y = threadIdx.x; y < h; y += blockDim.x) {
      int y0, y1; float ly;
      up_coord(y, sh, ph, y0, y1, ly);
      extra[3 * y] = (uint32_t)y0; extra[3 * y + 1] = (uint32_t)y1; extra[3 * y + 2] = __float_as_uint(ly);
    }
    ytab = extra;
    __syncthreads();
  }
  __device__ __forceinline__ Col col(int x) const {
    Col c;
    up_coord(x, sw, pw, c.x0, c.x1, c.lx);
    c.cy0 = c.cy1 = -1;
    c.v00 = c.v01 = c.v10 = c.v11 = 0.f;
    return c;
  }
  __device__ __forceinline__ int bit(Col &c, int y) const {
    const int y0 = (int)ytab[3 * y], y1 = (int)ytab[3 * y + 1];
    const float ly = __uint_as_float(ytab[3 * y + 2]);
    if (y0 != c.cy0) { c.cy0 = y0; c.v00 = lo[y0 * pw + c.x0]; c.v01 = lo[y0 * pw + c.x1]; }
    if (y1 != c.cy1) { c.cy1 = y1; c.v10 = lo[y1 * pw + c.x0]; c.v11 = lo[y1 * pw + c.x1]; }
    return up_lerp2(c.v00, c.v01, c.v10, c.v11, c.lx, ly) > thresh;
  }
};

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

template <typename Src>
__device__ __forceinline__ int seg_prev(const Src &src, int x, int y0, int h) {
  if (y0 > 0) { auto c = src.col(x); return src.bit(c, y0 - 1); }
  if (x > 0) { auto c = src.col(x - 1); return src.bit(c, h - 1); }
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

// SrcOf(mask index) -> the source of that mask
template <typename SrcOf>
__global__ __launch_bounds__(NT) void rle_counts_k(const SrcOf src_of, int h, int w, uint32_t *__restrict__ counts,
                                                  int32_t *__restrict__ nruns, int cap) {
  extern __shared__ uint32_t lds[];
  const int nseg = w * RB;
  uint32_t *cnt = lds;                              // [nseg] in sequence order (x * RB + r): transitions, then offsets
  int *last = reinterpret_cast<int *>(lds + nseg);  // [nseg] last transition position of the segment, then predecessor
  __shared__ uint32_t ws_sum[NT / 64];
  __shared__ int ws_max[NT / 64];
  __shared__ uint32_t total_s;
  __shared__ int lastpos_s;
  auto src = src_of(blockIdx.x);
  src.prepare(lds + 2 * nseg, h);                   // (the fused source tabulates its row coordinates behind the tables)
  const int hb = (h + RB - 1) / RB;

  for (int q = threadIdx.x; q < nseg; q += NT) {
    const SegRange sg = seg_of(q, h, w, hb);
    int prev = sg.y1 > sg.y0 ? seg_prev(src, sg.x, sg.y0, h) : 0;
    auto col = src.col(sg.x);
    uint32_t c = 0;
    int lp = -1;
    int y = sg.y0;
    for (; y + UNR <= sg.y1; y += UNR) {
      int v[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) v[u] = src.bit(col, y + u);
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int b = v[u];
        if (b != prev) { ++c; lp = sg.x * h + y + u; prev = b; }
      }
    }
    for (; y < sg.y1; ++y) {
      const int b = src.bit(col, y);
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
    int prev = seg_prev(src, sg.x, sg.y0, h);
    auto col = src.col(sg.x);
    int y = sg.y0;
    for (; y + UNR <= sg.y1; y += UNR) {
      int v[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) v[u] = src.bit(col, y + u);
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int b = v[u];
        if (b != prev) {
          const int p = sg.x * h + y + u;
          if (k < (uint32_t)cap) out[k] = (uint32_t)(p - pp);
          ++k; pp = p; prev = b;
        }
      }
    }
    for (; y < sg.y1; ++y) {
      const int b = src.bit(col, y);
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

struct PlainOf {
  const float *masks;
  int h, w;
  __device__ __forceinline__ PlainSrc operator()(int n) const { return {masks + (size_t)n * h * w, w}; }
};
struct UpOf {
  const float *lo;
  int ph, pw;
  float sh, sw, thresh;
  __device__ __forceinline__ UpSrc operator()(int n) const { return {lo + (size_t)n * ph * pw, ph, pw, sh, sw, thresh, nullptr}; }
};

}  // namespace

extern "C" int ymi_mask_rle_f32(const float *masks, int N, int h, int w, uint32_t *counts, int32_t *nruns, int cap,
                                void *stream) {
  if (N < 0 || h <= 0 || w <= 0 || cap <= 0) return YMI_EARG;
  if (N == 0) return YMI_OK;
  if (!masks || !counts || !nruns) return YMI_ENULL;
  if ((long)h * w >= (1L << 31) || w * RB * 8 > 65536) return YMI_ESHAPE;      // segment tables live in LDS: w <= 2048
  const PlainOf of{masks, h, w};
  hipLaunchKernelGGL(rle_counts_k<PlainOf>, dim3(N), dim3(NT), (size_t)w * RB * 8, (hipStream_t)stream, of, h, w, counts, nruns, cap);
  return ymi_launch_status();
}

extern "C" int ymi_mask_rle_upsampled_f32(const float *masks_lo, int N, int ph, int pw, int h, int w, float thresh,
                                          uint32_t *counts, int32_t *nruns, int cap, void *stream) {
  if (N < 0 || ph <= 0 || pw <= 0 || h <= 0 || w <= 0 || cap <= 0) return YMI_EARG;
  if (N == 0) return YMI_OK;
  if (!masks_lo || !counts || !nruns) return YMI_ENULL;
  if ((long)h * w >= (1L << 31) || (long)w * RB * 8 + (long)h * 12 > 65536 || (long)ph * pw >= (1L << 31)) return YMI_ESHAPE;
  const UpOf of{masks_lo, ph, pw, (float)ph / (float)h, (float)pw / (float)w, thresh};    // scales as mask_upsample computes them
  hipLaunchKernelGGL(rle_counts_k<UpOf>, dim3(N), dim3(NT), (size_t)w * RB * 8 + (size_t)h * 12, (hipStream_t)stream, of, h, w, counts,
                     nruns, cap);
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
