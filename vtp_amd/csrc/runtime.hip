// Error plumbing shared by every entry point of libvtp_hip.so.
#include "common.h"
#include "vtp_hip.h"
#include <stdarg.h>
#include <stdio.h>

namespace vtp {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return (int)e;
  }
  return VTP_OK;
}
}  // namespace vtp

extern "C" int vtp_abi_version(void) { return VTP_ABI_VERSION; }
extern "C" const char* vtp_last_error(void) { return vtp::g_err; }
