// Error plumbing shared by every entry point of libvtp_hip.so.
#include "common.h"
#include "vtp_hip.h"
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

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

// diagnostics (tools/cu_thief.py, VTP_DIAG=1): `nwg` workgroups that each take a whole CU (160 KiB of LDS, 512 threads) and do nothing
// but hold it for `ticks` x 10 ns -- the footprint of a long-running communication kernel (RCCL channels) beside the step's GEMMs
namespace vtp {
__global__ __launch_bounds__(512) void cu_thief_kernel(int ticks, unsigned long long* out) {
  extern __shared__ char smem[];
  if (threadIdx.x == 0) smem[0] = 1;
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(64);
  if (out && threadIdx.x == 0) out[blockIdx.x] = wall_clock64() - t0 + smem[0];
}
}  // namespace vtp
extern "C" int vtp_cu_thief(int nwg, int ticks, void* out, void* stream) {
  const char* diag = getenv("VTP_DIAG");
  if (!(diag && diag[0] == '1')) {
    vtp::set_error("vtp_cu_thief: diagnostics hooks need VTP_DIAG=1 in the environment");
    return VTP_ERR_ARG;
  }
  if (nwg < 1 || nwg > 256 || ticks < 0 || ticks > 100000000) {
    vtp::set_error("vtp_cu_thief: 1..256 workgroups, 0..1 s");
    return VTP_ERR_ARG;
  }
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)vtp::cu_thief_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL(vtp::cu_thief_kernel, dim3(nwg), dim3(512), 160 * 1024, (hipStream_t)stream, ticks, (unsigned long long*)out);
  return vtp::check_launch("cu_thief");
}

extern "C" int vtp_abi_version(void) { return VTP_ABI_VERSION; }
extern "C" const char* vtp_last_error(void) { return vtp::g_err; }
