// Error plumbing of the C-ABI (thread-local last-error string, negative return codes).
#include <cstdarg>
#include <cstdio>

#include "../../include/pixelhip.h"

namespace {
thread_local char g_err[1024] = "";
}

int pxl_set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

extern "C" const char* pxl_last_error(void) { return g_err; }
extern "C" int pxl_version(void) { return PXL_VERSION; }
