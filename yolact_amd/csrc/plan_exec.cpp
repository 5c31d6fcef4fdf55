// Native executor of an engine plan (ABI 7): ONE call walks the flat op list of yolact_amd/engine.Plan — convolutions, Winograd
// layers, DCNv2, the pointwise chain, the stem, layout / pooling / resize passes, and the record / wait markers that order the
// plan's two HIP streams — instead of one Python -> ctypes round trip per launch.
//
// Why (round 5): a batch-8 step is ~130 kernels + ~25 event operations; issued from Python that is ~230 interpreter-level calls
// per step.  bench.py's host_issue_ms_per_step shows what that costs on a given box, and boxes with slow hosts have measured 20 %
// lower batch-1 numbers with identical kernels (the reference's own multi-stream inference has the same shape: eval.py:793-796
// primes the net once and then drives it from a thread pool).  The descriptors stay where they are — ctypes structures owned by
// the Python plan, patched per call (input pointer, prototype output) — this file only replaces the LOOP.
// No device memory is allocated here; events are HIP events the caller creates through ymi_event_create (not device memory).
#include "../../include/yolact_amd.h"
#include <hip/hip_runtime_api.h>

extern "C" {

int ymi_event_create(void **ev) {
  if (!ev) return -3;
  hipEvent_t e;
  const hipError_t rc = hipEventCreateWithFlags(&e, hipEventDisableTiming);
  if (rc != hipSuccess) return (int)rc;
  *ev = (void *)e;
  return 0;
}

int ymi_event_destroy(void *ev) {
  if (!ev) return -3;
  return (int)hipEventDestroy((hipEvent_t)ev);
}

int ymi_plan_run(const ymi_plan_op *ops, int first, int last, void *stream_a, void *stream_b, void *const *events, int overlap,
                 int skip_sections, int32_t *failed_op) {
  if (!ops || first < 0 || last < first) return -1;
  void *const sb = overlap ? stream_b : stream_a;
  for (int k = first; k < last; ++k) {
    const ymi_plan_op &o = ops[k];
    if (o.section && (skip_sections & (1 << o.section))) continue;       // e.g. the protonet's launches without the mask branch
    void *const s = o.stream == 1 ? sb : stream_a;
    int rc = 0;
    switch (o.kind) {
      case YMI_OP_NOP: break;
      case YMI_OP_CONV: rc = ymi_conv2d_nhwc_f32((const ymi_conv_desc *)o.desc, s); break;
      case YMI_OP_WINO: rc = ymi_conv3x3_winograd_f32((const ymi_wino_desc *)o.desc, s); break;
      case YMI_OP_DCN: rc = ymi_dcn_v2_forward_f32((const ymi_dcn_desc *)o.desc, s); break;
      case YMI_OP_CHAIN: rc = ymi_pointwise_chain_f32((const ymi_chain_desc *)o.desc, s); break;
      case YMI_OP_STEM: rc = ymi_stem_pool_f32((const ymi_stem_desc *)o.desc, s); break;
      case YMI_OP_INPUT:       /* p0 = x NCHW (patched per call), p1 = y NHWC4, p2 = amax slot or NULL; i = B, C, H, W */
        rc = o.p[2] ? ymi_nchw_to_nhwc4_amax_f32((const float *)o.p[0], (float *)o.p[1], (int)o.i[0], (int)o.i[1], (int)o.i[2], (int)o.i[3],
                                                 (float *)o.p[2], s)
                    : ymi_nchw_to_nhwc4_f32((const float *)o.p[0], (float *)o.p[1], (int)o.i[0], (int)o.i[1], (int)o.i[2], (int)o.i[3], s);
        break;
      case YMI_OP_BILINEAR:    /* p0 -> p1; i = B, Hi, Wi, C, Ho, Wo, relu; f = scale_h, scale_w */
        rc = ymi_bilinear_nhwc_f32((const float *)o.p[0], (float *)o.p[1], (int)o.i[0], (int)o.i[1], (int)o.i[2], (int)o.i[3], (int)o.i[4],
                                   (int)o.i[5], (float)o.f[0], (float)o.f[1], (int)o.i[6], s);
        break;
      case YMI_OP_MAXPOOL:     /* p0 -> p1; i = B, H, W, C, Ho, Wo */
        rc = ymi_maxpool3x3s2_nhwc_f32((const float *)o.p[0], (float *)o.p[1], (int)o.i[0], (int)o.i[1], (int)o.i[2], (int)o.i[3],
                                       (int)o.i[4], (int)o.i[5], s);
        break;
      case YMI_OP_BILINEAR_ADD: /* p1 += up(p0); p2 = amax slot or NULL; i = B, Hi, Wi, C, Ho, Wo */
        rc = ymi_bilinear_add_nhwc_f32((const float *)o.p[0], (float *)o.p[1], (int)o.i[0], (int)o.i[1], (int)o.i[2], (int)o.i[3],
                                       (int)o.i[4], (int)o.i[5], (float *)o.p[2], s);
        break;
      case YMI_OP_RECORD:      /* i0 = event index */
        if (overlap) rc = (int)hipEventRecord((hipEvent_t)events[o.i[0]], (hipStream_t)s);
        break;
      case YMI_OP_WAIT:
        if (overlap) rc = (int)hipStreamWaitEvent((hipStream_t)s, (hipEvent_t)events[o.i[0]], 0);
        break;
      case YMI_OP_MEMSET:      /* p0 = device pointer, i0 = bytes: zero (the magnitude-bound arena at the head of a run) */
        rc = (int)hipMemsetAsync(o.p[0], 0, (size_t)o.i[0], (hipStream_t)s);
        break;
      default: rc = -1;
    }
    if (rc != 0) {
      if (failed_op) *failed_op = k;
      return rc;
    }
  }
  return 0;
}

}
