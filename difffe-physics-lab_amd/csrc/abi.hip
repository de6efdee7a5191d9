// ABI bookkeeping: version, status strings, last HIP error text.
#include <atomic>

#include "common.h"

namespace diffhe {
static thread_local hipError_t g_last = hipSuccess;
void set_last_error(hipError_t e) { g_last = e; }
static std::atomic<long long> g_bytes{0}, g_launches{0};
void account(double bytes) {
  g_bytes.fetch_add((long long)bytes, std::memory_order_relaxed);
  g_launches.fetch_add(1, std::memory_order_relaxed);
}
}  // namespace diffhe

extern "C" int diffhe_traffic_account(int reset, double* bytes, long long* launches) {
  if (bytes) *bytes = (double)diffhe::g_bytes.load();
  if (launches) *launches = diffhe::g_launches.load();
  if (reset) {
    diffhe::g_bytes.store(0);
    diffhe::g_launches.store(0);
  }
  return DIFFHE_OK;
}

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
