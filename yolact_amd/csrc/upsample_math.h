// F.interpolate(mode='bilinear', align_corners=False) arithmetic of the mask path (output_utils.py:91),
// shared by the upsample kernels (mask.hip) and the fused upsample + threshold + RLE kernel (rle.hip) so that both see
// bit-identical values: fp32 coordinate math, the same association of the two lerps, no FMA contraction
// (-ffp-contract=off in the Makefile).
#pragma once

__device__ __forceinline__ void up_coord(int dst, float scale, int in_size, int &i0, int &i1, float &l1) {
  float src = scale * ((float)dst + 0.5f) - 0.5f;
  src = src < 0.f ? 0.f : src;
  i0 = (int)src;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + ((i0 < in_size - 1) ? 1 : 0);
  l1 = src - (float)i0;
}

__device__ __forceinline__ float up_lerp2(float v00, float v01, float v10, float v11, float lx, float ly) {
  return (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
}
