// ABI bookkeeping: version, status strings, last HIP error text.
#include "common.h"

namespace diffhe {
static thread_local hipError_t g_last = hipSuccess;
void set_last_error(hipError_t e) { g_last = e; }
}  // namespace diffhe

extern "C" int diffhe_abi_version(void) { return DIFFHE_ABI_VERSION; }

extern "C" const char* diffhe_status_string(int status) {
  switch (status) {
    case DIFFHE_OK: return "ok";
    case DIFFHE_E_BADARG: return "bad argument (null pointer, size or stride)";
    case DIFFHE_E_LAUNCH: return "HIP launch/runtime failure (see diffhe_last_hip_error)";
    case DIFFHE_E_TOOBIG: return "problem too large for this entry point";
    case DIFFHE_E_BATCHPAD: return "padded batch must be a power of two <= 64 or a multiple of 64";
    default: return "unknown status";
  }
}

extern "C" const char* diffhe_last_hip_error(void) { return hipGetErrorString(diffhe::g_last); }
