// Error plumbing + version for liblandiff_hip.so.
#include "ld_common.h"
#include "../../include/landiff_hip.h"
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

static thread_local char g_err[512] = "";

int ld_set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int ld_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return ld_set_error(LD_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return LD_OK;
}

int ld_ensure_dyn_smem(const void* kernel, size_t bytes, LdSmemCache* cache) {
  if (bytes <= 48 * 1024) return LD_OK;                       // the default limit needs no opt-in
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return ld_set_error(LD_ERR_LAUNCH, "hipGetDevice: %s", hipGetErrorString(e));
  const bool cached = dev >= 0 && dev < 16;
  if (cached && cache->bytes[dev] >= bytes) return LD_OK;
  e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess)
    return ld_set_error(LD_ERR_LAUNCH, "hipFuncSetAttribute(MaxDynamicSharedMemorySize=%zu) on device %d: %s", bytes, dev, hipGetErrorString(e));
  if (cached) cache->bytes[dev] = bytes;
  return LD_OK;
}

int ld_knob(const char* name, int dflt, int* cache) {
  static const int tuning = [] { const char* t = getenv("LD_TUNING"); return t ? atoi(t) : 0; }();
  if (!tuning && *cache != LD_KNOB_UNSET) return *cache;
  const char* e = getenv(name);
  const int v = e ? atoi(e) : dflt;
  *cache = v;                                                  // racing first calls write the same value
  return v;
}

LD_API int ld_version(void) { return LD_ABI_VERSION; }
LD_API const char* ld_last_error(void) { return g_err; }
