// Error reporting + build info for libpaths_hip.so.  No global mutable state besides the thread-local
// error string; no allocation, no synchronisation (include/paths_hip.h).
#include "common.h"

static thread_local char g_err[512] = "";

int paths_set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

extern "C" {
const char* paths_last_error(void) { return g_err; }
const char* paths_build_info(void) { return "paths_hip gfx950 fp32-mfma r1"; }
int paths_abi_version(void) { return 1; }
}
