// Error channel, version and device queries of the C ABI.
#include "dc_common.h"

namespace dc {
static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace dc

extern "C" const char* dc_last_error(void) { return dc::g_err; }
extern "C" int dc_version(void) { return 100; }

extern "C" int dc_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

extern "C" int dc_set_device(int device) {
  DC_HIP_CHECK(hipSetDevice(device));
  return DC_OK;
}
