// Host emulation of ymi_jpeg_reconstruct_bgr_u8 for the CPU tests (tests/test_jpeg.py builds it with g++): loops over
// blocks / pixels calling the SAME inline arithmetic the GPU kernels call (yolact_amd/csrc/jpeg_math.h), with the same
// plane layout and component set-up as the launcher in jpeg.hip.  Test infrastructure — never part of the product.
#include "../include/yolact_amd.h"
#include "../yolact_amd/csrc/jpeg_math.h"

using namespace ymi_jpeg;

extern "C" int emul_jpeg_reconstruct_bgr_u8(const ymi_jpeg_info *info, const int16_t *coefs, const uint16_t *qt,
                                            uint8_t *planes_ws, uint8_t *out) {
  ColorArgs a;
  a.ncomp = info->ncomp; a.color = info->color; a.W = info->width; a.H = info->height;
  a.orientation = info->orientation;
  a.out_w = info->orientation >= 5 ? info->height : info->width;
  size_t off = 0;
  for (int i = 0; i < info->ncomp; ++i) {
    const int bw = info->bw[i], nblk = bw * info->bh[i];
    for (int blk = 0; blk < nblk; ++blk) {
      uint8_t px[64];
      idct_block(coefs + off + (size_t)blk * 64, qt + 64 * i, px);
      const int by = blk / bw, bx = blk - by * bw;
      for (int r = 0; r < 8; ++r)
        for (int c = 0; c < 8; ++c) planes_ws[off + ((size_t)(by * 8 + r) * bw + bx) * 8 + c] = px[r * 8 + c];
    }
    CompPlane &c = a.c[i];
    c.p = planes_ws + off; c.stride = bw * 8; c.dw = info->dw[i]; c.dh = info->dh[i]; c.hf = info->hf[i]; c.vf = info->vf[i];
    c.mode = upsample_mode(c.hf, c.vf, c.dw);
    off += (size_t)nblk * 64;
  }
  for (int y = 0; y < a.H; ++y)
    for (int x = 0; x < a.W; ++x) {
      const uint32_t v = pixel_bgr(a, x, y);
      int ox, oy;
      orient(a.orientation, a.W, a.H, x, y, ox, oy);
      uint8_t *o = out + ((size_t)oy * a.out_w + ox) * 3;
      o[0] = (uint8_t)v; o[1] = (uint8_t)(v >> 8); o[2] = (uint8_t)(v >> 16);
    }
  return 0;
}
