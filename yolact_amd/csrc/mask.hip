// postprocess() on device: prototype linear combination on the fp32 matrix cores + sigmoid + crop,
// then bilinear upsample to the image size + binarise, and box sanitising.
// Reference: layers/output_utils.py:69-99, layers/box_utils.py:327-373 (sanitize_coordinates, crop).
#include "common.h"
#include "upsample_math.h"
#include <stdlib.h>
#include "../../include/yolact_amd.h"

namespace {

// masks_lo[n, pix] = crop(sigmoid(sum_k coef[n,k] * proto[pix,k])).
// MFMA roles: A = coef (rows n), B = proto^T (cols = pixels) so that for a fixed accumulator register the
// 32 lanes of a half-wave hold 32 consecutive pixels of ONE detection -> 128-byte coalesced stores into
// the [N, ph*pw] output.  K = D (32): lane-half h holds k = 16h..16h+15 (64 contiguous bytes of the
// pixel's / detection's row) and step s pairs (s, 16+s) — same free-K-order trick as the conv engine.
// Batched form: blockIdx.y = image b of a fixed-capacity batch (proto [B,npix,D], coef / box / out rows b*cap ..); the
// number of live detections of image b is read from `count[b]` on the device (no host round trip), `N` when count is null.
template <int D>
__global__ __launch_bounds__(256) void lincomb_crop_k(const float *__restrict__ proto, const float *__restrict__ coef,
                                                      const float *__restrict__ box, float *__restrict__ out, int ph,
                                                      int pw, int N, int crop, const int *__restrict__ count, int cap) {
  static_assert(D == 32, "mask_dim 32");
  extern __shared__ float cb[];  // [Npad][4] crop bounds x1,x2,y1,y2
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int npix = ph * pw;
  {
    const int b = blockIdx.y;
    if (count) { N = count[b]; N = N > cap ? cap : N; }
    if (N <= 0) return;
    proto += (size_t)b * npix * D;
    coef += (size_t)b * cap * D;
    box += (size_t)b * cap * 4;
    out += (size_t)b * cap * npix;
  }
  const int ntiles = (N + 31) / 32;
  for (int n = t; n < ntiles * 32; n += 256) {
    float x1 = 0.f, x2 = (float)pw, y1 = 0.f, y2 = (float)ph;
    if (n < N && crop) {
      // sanitize_coordinates(_x1, _x2, img_size, padding=1, cast=False)
      const float a = box[n * 4 + 0] * (float)pw, b = box[n * 4 + 2] * (float)pw;
      x1 = fminf(a, b) - 1.f; x1 = x1 < 0.f ? 0.f : x1;
      x2 = fmaxf(a, b) + 1.f; x2 = x2 > (float)pw ? (float)pw : x2;
      const float c = box[n * 4 + 1] * (float)ph, e = box[n * 4 + 3] * (float)ph;
      y1 = fminf(c, e) - 1.f; y1 = y1 < 0.f ? 0.f : y1;
      y2 = fmaxf(c, e) + 1.f; y2 = y2 > (float)ph ? (float)ph : y2;
    }
    cb[n * 4 + 0] = x1; cb[n * 4 + 1] = x2; cb[n * 4 + 2] = y1; cb[n * 4 + 3] = y2;
  }
  __syncthreads();

  const int pix0 = (blockIdx.x * 4 + wave) * 32;
  if (pix0 >= npix) return;
  const int half = lane >> 5, l31 = lane & 31;
  // B operand: proto row of pixel (pix0 + l31), k = 16*half .. +15
  float bfrag[16];
  {
    const int pix = pix0 + l31;
    const bool ok = pix < npix;
    const float *src = proto + (size_t)(ok ? pix : 0) * D + 16 * half;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = *reinterpret_cast<const f32x4 *>(src + 4 * q);
#pragma unroll
      for (int e = 0; e < 4; ++e) bfrag[4 * q + e] = ok ? v[e] : 0.f;
    }
  }
  const int pix = pix0 + l31;
  const int py = pix / pw, px = pix - py * pw;
  const float fx = (float)px, fy = (float)py;
  for (int nt = 0; nt < ntiles; ++nt) {
    const int n_a = nt * 32 + l31;
    const bool ok = n_a < N;
    const float *src = coef + (size_t)(ok ? n_a : 0) * D + 16 * half;
    float afrag[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = *reinterpret_cast<const f32x4 *>(src + 4 * q);
#pragma unroll
      for (int e = 0; e < 4; ++e) afrag[4 * q + e] = ok ? v[e] : 0.f;
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(afrag[s], bfrag[s], acc, 0, 0, 0);
    if (pix < npix) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (n < N) {
          float v = 1.f / (1.f + expf(-acc[r]));
          const float *c4 = cb + n * 4;
          const bool in = fx >= c4[0] && fx < c4[1] && fy >= c4[2] && fy < c4[3];
          out[(size_t)n * npix + pix] = in ? v : 0.f;
        }
      }
    }
  }
}

// out[n,y,x] = (bilinear(masks_lo[n])[y,x] > thresh) ? 1 : 0.  Flat float4 stores: pure HBM-write stream
// (N*h*w*4 bytes, 121 MB at N=100, 550x550) while the [N,ph,pw] source stays L2-resident.
__global__ __launch_bounds__(256) void mask_upsample_k(const float *__restrict__ lo, float *__restrict__ out, int ph,
                                                       int pw, int h, int w, float sh, float sw, float thresh,
                                                       long total4, long total) {
  const long hw = (long)h * w;
  for (long i4 = blockIdx.x * 256L + threadIdx.x; i4 < total4; i4 += (long)gridDim.x * 256L) {
    const long base = i4 * 4;
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const long i = base + e;
      float r = 0.f;
      if (i < total) {
        const long n = i / hw;
        const int rem = (int)(i - n * hw);
        const int y = rem / w, x = rem - y * w;
        int y0, y1, x0, x1; float ly, lx;
        up_coord(y, sh, ph, y0, y1, ly);
        up_coord(x, sw, pw, x0, x1, lx);
        const float *img = lo + n * (long)ph * pw;
        const float v00 = img[y0 * pw + x0], v01 = img[y0 * pw + x1];
        const float v10 = img[y1 * pw + x0], v11 = img[y1 * pw + x1];
        const float v = up_lerp2(v00, v01, v10, v11, lx, ly);
        r = thresh < 0.f ? v : (v > thresh ? 1.f : 0.f);
      }
      o[e] = r;
    }
    if (base + 3 < total) {
      *reinterpret_cast<f32x4 *>(out + base) = o;
    } else {
      for (int e = 0; e < 4; ++e) if (base + e < total) out[base + e] = o[e];
    }
  }
}

// Row-band form of the same upsample (the default): one block = R consecutive output rows of one mask.
//   phase 1  thread = output column(s); the horizontally interpolated values of the two source rows y0 / y1 stay in
//            registers and are refreshed only when the (block-uniform) source row changes — about every h/ph output rows —
//            so a pixel costs ~1/4 of a gather instead of four, plus two lerps, one compare and one LDS store;
//   phase 2  the band (R*w contiguous floats of the flat [N,h,w] output) leaves LDS as aligned 16-byte stores.
// The flat kernel above did four dependent gathers and two integer divisions per pixel and reached 1.2 TB/s of the
// 968 MB mask write of a batch (profiles/r01); this one is bound by the HBM write.
// Arithmetic is identical to mask_upsample_k (same fp32 coordinate math, same lerp association) — bit-equal outputs.
// blockIdx.y = mask index n = b*cap + i; masks past count[b] are skipped (their rows are unspecified, like every
// fixed-capacity output of the path).
template <int R>
__global__ __launch_bounds__(256) void mask_upsample_band_k(const float *__restrict__ lo, float *__restrict__ out, int ph,
                                                            int pw, int h, int w, float sh, float sw, float thresh,
                                                            const int *__restrict__ count, int cap) {
  extern __shared__ __attribute__((aligned(16))) float band[];      // [3 + R*w], shifted so that LDS float4 j <-> global float4 j
  const int n = blockIdx.y;
  if (count) {
    const int b = n / cap, i = n - b * cap;
    if (i >= count[b]) return;
  }
  const int y_begin = blockIdx.x * R;
  const int rows = (h - y_begin) < R ? (h - y_begin) : R;
  const long g0 = ((long)n * h + y_begin) * w;       // first flat element of the band
  const int shift = (int)(g0 & 3);                   // LDS index e holds flat element g0 - shift + e
  const float *img = lo + (size_t)n * ph * pw;
  for (int x = threadIdx.x; x < w; x += 256) {
    int x0, x1; float lx;
    up_coord(x, sw, pw, x0, x1, lx);
    int cy0 = -1, cy1 = -1;
    float v00 = 0.f, v01 = 0.f, v10 = 0.f, v11 = 0.f;
    for (int r = 0; r < rows; ++r) {
      int y0, y1; float ly;
      up_coord(y_begin + r, sh, ph, y0, y1, ly);
      if (y0 != cy0) { cy0 = y0; v00 = img[y0 * pw + x0]; v01 = img[y0 * pw + x1]; }
      if (y1 != cy1) { cy1 = y1; v10 = img[y1 * pw + x0]; v11 = img[y1 * pw + x1]; }
      const float v = up_lerp2(v00, v01, v10, v11, lx, ly);
      band[shift + r * w + x] = thresh < 0.f ? v : (v > thresh ? 1.f : 0.f);
    }
  }
  __syncthreads();
  const int end = shift + rows * w;
  float *dst = out + (g0 - shift);                   // 16-byte aligned (out is, and g0 - shift is a multiple of 4)
  const int j0 = (shift + 3) >> 2, j1 = end >> 2;    // float4 chunks [j0, j1) are fully inside the band
  if (j1 > j0) {
    for (int e = shift + threadIdx.x; e < 4 * j0; e += 256) dst[e] = band[e];
    for (int j = j0 + threadIdx.x; j < j1; j += 256)
      *reinterpret_cast<f32x4 *>(dst + 4 * j) = *reinterpret_cast<const f32x4 *>(band + 4 * j);
    for (int e = 4 * j1 + threadIdx.x; e < end; e += 256) dst[e] = band[e];
  } else {
    for (int e = shift + threadIdx.x; e < end; e += 256) dst[e] = band[e];
  }
}

// Row-band upsample, second form (round 4; VERDICT r3 weak #6: the band kernel above streams 968 MB per batch at 2.4 - 3.3 TB/s).
// What bounded the band kernel: per block TWO phases around a barrier (columns -> LDS band -> copy out), and inside phase 1 a
// dependent global-load round trip every time the source row changes (2 - 3 per 8-row band), with 8 small blocks per CU to hide it.
// Here the horizontal half of the bilinear form is evaluated ONCE per needed source row into LDS ((R * ph / h) + 3 rows of w floats:
// all loads of the block are issued together, one round trip), and after the single barrier every WAVE produces whole output rows:
// two LDS values and one vertical lerp per pixel, one compare, one store — no second pass over the band.  Arithmetic: (1 - lx) * v0 + lx * v1 per source row, then (1 - ly) * top + ly * bottom — exactly the two
// lerps of up_lerp2 in its association (no FMA contraction), so the outputs are bit-identical to both kernels above.
// NT: nontemporal stores (the 121 MB per image are written once and read by another kernel, or the host, much later).
template <int R, bool NT>
__global__ __launch_bounds__(256) void mask_upsample_rows_k(const float *__restrict__ lo, float *__restrict__ out, int ph,
                                                            int pw, int h, int w, float sh, float sw, float thresh,
                                                            const int *__restrict__ count, int cap, int abl, int nmask) {
  extern __shared__ __attribute__((aligned(16))) float hs[];        // [ns][w] source rows ys0 .. ys0 + ns - 1 interpolated along x,
                                                                    // then [ns][pw] the raw source rows
  __shared__ f32x4 rowinfo[R];                                      // per output row of the band: {offset of its top row in hs, of its
                                                                    // bottom row (as int bits), ly, 1 - ly}
  const int n = blockIdx.y;
  if (count) {
    const int b = n / cap, i = n - b * cap;
    if (i >= count[b]) return;
  }
  const int t = threadIdx.x;
  const int bxi = blockIdx.x;
  const int y_begin = bxi * R;
  const int rows = (h - y_begin) < R ? (h - y_begin) : R;
  int ys0, ys1, tmp; float tl;
  up_coord(y_begin, sh, ph, ys0, tmp, tl);
  up_coord(y_begin + rows - 1, sh, ph, tmp, ys1, tl);
  const int ns = ys1 - ys0 + 1;
  const float *img = lo + (size_t)n * ph * pw + (size_t)ys0 * pw;
  float *raw = hs + ns * w;
  // phase 0: the needed source rows are one contiguous run of the low-resolution mask: coalesced copy into LDS, then every thread
  // owns output columns (their x coordinates are computed once) and walks down the rows out of LDS
#ifdef YMI_DIAGNOSTICS   // abl (env YMI_UP_ABLATE, tools/upsample_probe.py): bit0 no phase 0, bit1 no LDS reads in phase 1, bit2 no stores
  if (!(abl & 1))
#endif
  {
    for (int i = t; i < ns * pw; i += 256) raw[i] = img[i];
    __syncthreads();
    for (int x = t; x < w; x += 256) {
      int x0, x1; float lx;
      up_coord(x, sw, pw, x0, x1, lx);
      const float omx = 1.f - lx;
      int s = 0;
      for (; s + 4 <= ns; s += 4) {                  // four rows per trip: eight independent LDS reads, then the arithmetic
        float a[4], b[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { a[k] = raw[(s + k) * pw + x0]; b[k] = raw[(s + k) * pw + x1]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) hs[(s + k) * w + x] = omx * a[k] + lx * b[k];
      }
      for (; s < ns; ++s) hs[s * w + x] = omx * raw[s * pw + x0] + lx * raw[s * pw + x1];
    }
  }
  if (t < rows) {
    int a0, a1; float la;
    up_coord(y_begin + t, sh, ph, a0, a1, la);
    f32x4 ri;
    ri[0] = __int_as_float((a0 - ys0) * w); ri[1] = __int_as_float((a1 - ys0) * w);
    ri[2] = la; ri[3] = 1.f - la;
    rowinfo[t] = ri;
  }
  __syncthreads();
  // phase 1: the band as a FLAT range of the output, cut into segments of 64 consecutive floats that start on 128-byte lines of
  // the output tensor; wave v sweeps its quarter of the segments, lane l <-> float l of the segment.  All bookkeeping (element,
  // row, column of the segment's first float) is wave-uniform, i.e. scalar; a segment straddles at most one row boundary (w >= 64:
  // launcher), where lanes select between the constants of the two rows.  Per segment: two conflict-free LDS reads (consecutive
  // lanes, consecutive banks), three multiply / adds, a compare, one store instruction writing two whole cache lines.
  // Measured on the way here (profiles/r04_upsample_probe.txt, r04_upsample_ablation.txt): (i) an aligned float4 of the flat band
  // per lane (16-byte stores) has a lane stride of 16 bytes in LDS — 4-way bank conflicts; (ii) one wave per output ROW (stores
  // start wherever the row starts: 550 floats = 17.2 lines) reached 3.5 TB/s at 550 x 550 but 5.2 TB/s at 512 x 512, where rows are
  // whole lines; (iii) per-LANE row / column bookkeeping with a row-table read per pixel: the stores could be removed without
  // changing the time (3.8 TB/s) — 60 % of it was the loop skeleton.  torch's fill kernel: 6.8 TB/s over the same bytes.
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
  const bool soft = thresh < 0.f;
  const long g0 = ((long)n * h + y_begin) * w;       // first flat element of the band
  const int lead = (int)(g0 & 31);                   // floats between the preceding 128-byte line boundary and the band
  const int total = rows * w;
  float *base = out + (g0 - lead);                   // 128-byte aligned when `out` is (torch allocations are)
  const int nseg = (lead + total + 63) >> 6;
  const int per = (nseg + 3) >> 2;
  const int q0 = wave * per, q1 = (q0 + per) < nseg ? (q0 + per) : nseg;
  // Two independent streams per wave (halves of its segments), advanced together: the loop body then holds two independent
  // LDS-read -> lerp -> store chains instead of one serial chain (two streams: 0.272 -> 0.227 ms) (with ~5 waves per SIMD the single chain left the
  // waves parked on lgkmcnt 64 % of the time: profiles/r04_pmc_upsample.tsv).  Row selection is per lane and branch-free.
  struct Stream { int q, qe, e0, x0, r0, ta, ba, tb, bb; float lya, oma, lyb, omb; };
  auto row_of = [&](int r, int &toff, int &boff, float &ly, float &oml) {
    const f32x4 ri = rowinfo[r < rows ? r : rows - 1];
    const float f0 = ri[0], f1 = ri[1];              // (scalar copies first: __builtin_bit_cast applied directly to a vector ELEMENT
    toff = __float_as_int(f0); boff = __float_as_int(f1);   //  reads element 0 — hipcc 7.2, seen in the ISA)
    ly = ri[2]; oml = ri[3];
  };
  auto open = [&](Stream &st, int qa, int qb) {
    st.q = qa; st.qe = qb;
    st.e0 = 64 * qa - lead; st.r0 = 0; st.x0 = st.e0;
    if (st.e0 > 0) { st.r0 = st.e0 / w; st.x0 = st.e0 - st.r0 * w; }
    row_of(st.r0, st.ta, st.ba, st.lya, st.oma);
    row_of(st.r0 + 1, st.tb, st.bb, st.lyb, st.omb);
  };
  auto fetch = [&](const Stream &st, float &top, float &bot, float &ly, float &oml, bool &ok) {
    const int e = st.e0 + lane, xl = st.x0 + lane;
    ok = st.q < st.qe && e >= 0 && e < total;
    const bool nx = xl >= w;
    const int x = ok ? (nx ? xl - w : xl) : 0;
    top = hs[(ok ? (nx ? st.tb : st.ta) : 0) + x];
    bot = hs[(ok ? (nx ? st.bb : st.ba) : 0) + x];
    ly = nx ? st.lyb : st.lya; oml = nx ? st.omb : st.oma;
  };
  auto finish = [&](Stream &st, float top, float bot, float ly, float oml, bool ok) {
    float v = oml * top + ly * bot;
#ifdef YMI_DIAGNOSTICS
    if (abl & 2) v = (float)st.x0;
    if (abl & 4) ok = ok && v == 12345.f;
#endif
    const float o = soft ? v : (v > thresh ? 1.f : 0.f);
    if (ok) {
      float *dst = base + 64 * st.q + lane;
      if (NT) __builtin_nontemporal_store(o, dst);
      else *dst = o;
    }
    ++st.q; st.e0 += 64; st.x0 += 64;
    if (st.x0 >= w) {                                           // (wave-uniform) next row
      st.x0 -= w; ++st.r0;
      st.ta = st.tb; st.ba = st.bb; st.lya = st.lyb; st.oma = st.omb;
      row_of(st.r0 + 1, st.tb, st.bb, st.lyb, st.omb);
    }
  };
  constexpr int NSTR = 2;                            // independent streams per wave (4: 0.241 ms, 2: 0.224 - 0.227, 1: 0.272)
  const int len = q1 > q0 ? q1 - q0 : 0, part = (len + NSTR - 1) / NSTR;
  Stream st[NSTR];
#pragma unroll
  for (int k = 0; k < NSTR; ++k) {
    const int qa = q0 + k * part, qb = qa + part;
    open(st[k], qa < q1 ? qa : q1, qb < q1 ? qb : q1);
  }
  for (int it = 0; it < part; ++it) {
    float tp_[NSTR], bt_[NSTR], ly_[NSTR], om_[NSTR]; bool ok_[NSTR];
#pragma unroll
    for (int k = 0; k < NSTR; ++k) fetch(st[k], tp_[k], bt_[k], ly_[k], om_[k], ok_[k]);
#pragma unroll
    for (int k = 0; k < NSTR; ++k) finish(st[k], tp_[k], bt_[k], ly_[k], om_[k], ok_[k]);
  }
}

// boxes -> absolute int64 pixels: sanitize_coordinates(x1, x2, w, padding=0, cast=False) then .long()
__global__ void boxes_to_pixels_k(const float *__restrict__ box, long long *__restrict__ out, int N, int w, int h) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const float a = box[n * 4 + 0] * (float)w, b = box[n * 4 + 2] * (float)w;
  float x1 = fminf(a, b); x1 = x1 < 0.f ? 0.f : x1;
  float x2 = fmaxf(a, b); x2 = x2 > (float)w ? (float)w : x2;
  const float c = box[n * 4 + 1] * (float)h, e = box[n * 4 + 3] * (float)h;
  float y1 = fminf(c, e); y1 = y1 < 0.f ? 0.f : y1;
  float y2 = fmaxf(c, e); y2 = y2 > (float)h ? (float)h : y2;
  out[n * 4 + 0] = (long long)x1; out[n * 4 + 1] = (long long)y1;
  out[n * 4 + 2] = (long long)x2; out[n * 4 + 3] = (long long)y2;
}

}  // namespace

extern "C" {

int ymi_lincomb_crop_f32(const float *proto, const float *coef, const float *box, float *masks_lo, int ph, int pw,
                         int D, int N, int crop, void *stream) {
  if (!proto || !coef || !box || !masks_lo) return YMI_ENULL;
  if (D != 32) return YMI_ESHAPE;  // cfg.mask_dim of every shipped config (data/config.py:691)
  if (ph <= 0 || pw <= 0 || N <= 0 || N > 1024) return YMI_EARG;
  const int npix = ph * pw;
  const int grid = (npix + 127) / 128;
  const size_t lds = (size_t)((N + 31) / 32) * 32 * 4 * sizeof(float);
  hipLaunchKernelGGL(lincomb_crop_k<32>, dim3(grid), dim3(256), lds, (hipStream_t)stream, proto, coef, box, masks_lo,
                     ph, pw, N, crop, (const int *)nullptr, N);
  return ymi_launch_status();
}

int ymi_lincomb_crop_batch_f32(const float *proto, const float *coef, const float *box, const int32_t *count,
                               float *masks_lo, int B, int cap, int ph, int pw, int D, int crop, void *stream) {
  if (!proto || !coef || !box || !masks_lo) return YMI_ENULL;
  if (D != 32) return YMI_ESHAPE;
  if (B <= 0 || B > 65535 || ph <= 0 || pw <= 0 || cap <= 0 || cap > 1024) return YMI_EARG;
  const int npix = ph * pw;
  const size_t lds = (size_t)((cap + 31) / 32) * 32 * 4 * sizeof(float);
  hipLaunchKernelGGL(lincomb_crop_k<32>, dim3((npix + 127) / 128, B), dim3(256), lds, (hipStream_t)stream, proto, coef, box,
                     masks_lo, ph, pw, cap, crop, (const int *)count, cap);
  return ymi_launch_status();
}

namespace {
constexpr int UP_ROWS = 8;
constexpr int UP_ROWS2 = 32;
// YOLACT_AMD_UPSAMPLE = rowsnt (default: mask_upsample_rows_k with nontemporal stores) | rows (the same with plain stores) | band (the round-2
// kernel) — an A/B switch for the measurement log (tools/upsample_probe.py); all three produce the same bits
int upsample_variant() {
  static const int v = [] {
    const char *e = getenv("YOLACT_AMD_UPSAMPLE");
    if (!e) return 1;
    return e[0] == 'b' ? 2 : (e[0] == 'r' && e[1] == 'o' && e[2] == 'w' && e[3] == 's' && e[4] == 'n') ? 1 : 0;
  }();
  return v;
}
// rows kernel when its source rows fit the LDS budget, banded kernel when a band does, flat kernel otherwise
int launch_upsample(const float *masks_lo, float *out, const int32_t *count, int nmask, int cap, int ph, int pw, int h, int w,
                    float thresh, hipStream_t s) {
  {
    const float sh = (float)ph / (float)h;
    const int variant = upsample_variant();
    int abl = 0;
#ifdef YMI_DIAGNOSTICS
    { const char *e = getenv("YMI_UP_ABLATE"); abl = e ? atoi(e) : 0; }
#endif
    // bands of 32 output rows when that still gives every CU several blocks, else 16; tiny launches (< ~2 blocks per CU even so)
    // stay on the band kernel, whose 8-row blocks spread better (100 masks of 550 x 550: 0.033 vs 0.037 ms)
    const long bands32 = (long)((h + 31) / 32) * nmask, bands16 = (long)((h + 15) / 16) * nmask;
    const int R = bands32 >= 5120 ? 32 : 16;
    const long ns_max = (long)((float)R * sh) + 3;               // source rows a band of R output rows can touch
    const size_t lds = (size_t)ns_max * (w + pw) * sizeof(float);
    if (variant != 2 && w >= 64 && lds <= 48 * 1024 && nmask <= 65535 && ((uintptr_t)out & 15) == 0 && (R == 32 || bands16 >= 4096)) {
      // (one block per band: persistent blocks looping over bands measured SLOWER, 0.295 vs 0.272 ms — profiles/r04_upsample_ablation.txt)
      const dim3 grid((h + R - 1) / R, nmask);
      const float sw = (float)pw / (float)w;
#define YMI_UP_LAUNCH(RR, NTV) hipLaunchKernelGGL((mask_upsample_rows_k<RR, NTV>), grid, dim3(256), lds, s, masks_lo, out, ph, pw, h, w, \
                                                  sh, sw, thresh, (const int *)count, cap, abl, nmask)
      if (R == 32) { if (variant == 1) YMI_UP_LAUNCH(32, true); else YMI_UP_LAUNCH(32, false); }
      else { if (variant == 1) YMI_UP_LAUNCH(16, true); else YMI_UP_LAUNCH(16, false); }
#undef YMI_UP_LAUNCH
      return ymi_launch_status();
    }
  }
  if (((size_t)UP_ROWS * w + 4) * sizeof(float) <= 64 * 1024 && nmask <= 65535 && ((uintptr_t)out & 15) == 0) {
    hipLaunchKernelGGL(mask_upsample_band_k<UP_ROWS>, dim3((h + UP_ROWS - 1) / UP_ROWS, nmask), dim3(256),
                       ((size_t)UP_ROWS * w + 4) * sizeof(float), s, masks_lo, out, ph, pw, h, w, (float)ph / (float)h,
                       (float)pw / (float)w, thresh, (const int *)count, cap);
    return ymi_launch_status();
  }
  if (count) return YMI_ESHAPE;
  const long total = (long)nmask * h * w, total4 = (total + 3) / 4;
  long g = (total4 + 255) / 256;
  const long capg = 256L * 16;
  hipLaunchKernelGGL(mask_upsample_k, dim3((int)(g > capg ? capg : g)), dim3(256), 0, s, masks_lo, out, ph, pw, h, w,
                     (float)ph / (float)h, (float)pw / (float)w, thresh, total4, total);
  return ymi_launch_status();
}
}  // namespace

int ymi_mask_upsample_batch_f32(const float *masks_lo, const int32_t *count, float *out, int B, int cap, int ph, int pw,
                                int h, int w, float thresh, void *stream) {
  if (!masks_lo || !out) return YMI_ENULL;
  if (B <= 0 || cap <= 0 || ph <= 0 || pw <= 0 || h <= 0 || w <= 0 || (long)B * cap > 65535) return YMI_EARG;
  return launch_upsample(masks_lo, out, count, B * cap, cap, ph, pw, h, w, thresh, (hipStream_t)stream);
}

int ymi_mask_upsample_f32(const float *masks_lo, float *out, int N, int ph, int pw, int h, int w, float thresh,
                          void *stream) {
  if (!masks_lo || !out) return YMI_ENULL;
  if (N <= 0 || ph <= 0 || pw <= 0 || h <= 0 || w <= 0) return YMI_EARG;
  static const bool flat = getenv("YOLACT_AMD_UPSAMPLE_FLAT") != nullptr;   // A/B switch for the measurement log
  if (!flat)
    return launch_upsample(masks_lo, out, nullptr, N, N, ph, pw, h, w, thresh, (hipStream_t)stream);
  const long total = (long)N * h * w, total4 = (total + 3) / 4;
  long g = (total4 + 255) / 256;
  const long cap = 256L * 16;
  const int grid = (int)(g > cap ? cap : g);
  hipLaunchKernelGGL(mask_upsample_k, dim3(grid), dim3(256), 0, (hipStream_t)stream, masks_lo, out, ph, pw, h, w,
                     (float)ph / (float)h, (float)pw / (float)w, thresh, total4, total);
  return ymi_launch_status();
}

int ymi_boxes_to_pixels(const float *box, int64_t *out, int N, int w, int h, void *stream) {
  if (!box || !out) return YMI_ENULL;
  if (N <= 0) return YMI_EARG;
  hipLaunchKernelGGL(boxes_to_pixels_k, dim3((N + 63) / 64), dim3(64), 0, (hipStream_t)stream, box, (long long *)out, N,
                     w, h);
  return ymi_launch_status();
}

}  // extern "C"
