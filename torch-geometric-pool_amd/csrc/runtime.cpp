// Error reporting, version and device queries of the tgp HIP library.
#include "common.h"

#include <string.h>

namespace tgp {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace tgp

extern "C" int tgp_version(void) { return TGP_ABI_VERSION; }

extern "C" const char* tgp_last_error(void) { return tgp::g_err; }

extern "C" int tgp_device_cu_count(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
  return cus;
}
