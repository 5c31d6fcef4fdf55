// Version / error strings of the C ABI (no device code here).
#include "../../include/yolact_amd.h"
#include <hip/hip_runtime_api.h>

extern "C" {

int ymi_abi_version(void) { return YMI_ABI_VERSION; }

const char *ymi_strerror(int code) {
  switch (code) {
    case 0: return "ok";
    case -1: return "bad argument / unsupported configuration";
    case -2: return "shape or alignment constraint violated";
    case -3: return "null pointer";
    case YMI_EFORMAT: return "corrupt or truncated input stream";
    case YMI_EUNSUPPORTED: return "valid input outside the supported subset";
  }
  if (code > 0) return hipGetErrorString((hipError_t)code);
  return "unknown error";
}

// Workspace sizes in bytes: the one place the formulas of the header comments are written down as code, so that a host other than
// the Python shim does not re-derive them (SURVEY 8(b) "ownership": the caller allocates, the library never does).
int64_t ymi_workspace_bytes(int what, const void *desc) {
  auto cdiv = [](int64_t a, int64_t b) { return (a + b - 1) / b; };
  if (what == YMI_WS_AMAX_SLOT) return (int64_t)YMI_AMAX_SUB * YMI_AMAX_STRIDE * 4;
  if (!desc) return -3;
  switch (what) {
    case YMI_WS_WINO_V:
    case YMI_WS_WINO_M: {
      const ymi_wino_desc *d = (const ymi_wino_desc *)desc;
      const int m = d->m == 4 ? 4 : 2;
      if ((d->m != 0 && d->m != 2 && d->m != 4) || d->B < 1 || d->H < 1 || d->W < 1 || d->C < 1 || d->Cout < 1) return -1;
      const int64_t G = (int64_t)(m + 2) * (m + 2), T = (int64_t)d->B * cdiv(d->H, m) * cdiv(d->W, m);
      return 4 * G * T * (what == YMI_WS_WINO_V ? (int64_t)d->C : cdiv(d->Cout, 4) * 4);
    }
    case YMI_WS_SPLITK: {
      const ymi_conv_desc *d = (const ymi_conv_desc *)desc;
      if (d->B < 1 || d->Ho < 1 || d->Wo < 1 || d->Cout < 1 || d->split_k < 0) return -1;
      return d->split_k > 1 ? 4 * (int64_t)d->split_k * d->B * d->Ho * d->Wo * d->Cout : 0;
    }
    case YMI_WS_MASK_IOU: {
      const ymi_mask_iou_shape *d = (const ymi_mask_iou_shape *)desc;
      if (d->A < 0 || d->B < 0) return -1;
      return 4 * ((int64_t)d->A * d->B + d->A + d->B);
    }
    case YMI_WS_JPEG_COEFS: return 2 * ((const ymi_jpeg_info *)desc)->coef_count;
    case YMI_WS_JPEG_PLANES: return ((const ymi_jpeg_info *)desc)->plane_bytes;
    case YMI_WS_DETECT_SCORES_T:
    case YMI_WS_DETECT_PER_PRIOR:
    case YMI_WS_DETECT_CAND:
    case YMI_WS_DETECT_REC: {
      const ymi_detect_desc *d = (const ymi_detect_desc *)desc;
      if (d->B < 1 || d->P < 1 || d->C < 2 || d->D < 0 || d->top_k < 1 || d->max_det < 1) return -1;
      if (what == YMI_WS_DETECT_SCORES_T) return 4 * (int64_t)d->B * (d->C - 1) * d->P;
      if (what == YMI_WS_DETECT_PER_PRIOR) return 4 * (int64_t)d->B * d->P;
      if (what == YMI_WS_DETECT_CAND) return 4 * (int64_t)d->B * (d->C - 1) * d->top_k;
      const int64_t cap = d->cross_class ? d->top_k : d->max_det;
      return 4 * (int64_t)d->B * (1 + cap * (6 + d->D));
    }
    case YMI_WS_RLE_COUNTS: {
      const ymi_rle_shape *d = (const ymi_rle_shape *)desc;
      if (d->N < 0 || d->h < 1 || d->w < 1) return -1;
      const int64_t cap = d->cap > 0 ? d->cap : (int64_t)d->h * d->w + 1;
      return 4 * (int64_t)d->N * cap;
    }
  }
  return -1;
}

}
