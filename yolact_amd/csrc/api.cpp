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

}
