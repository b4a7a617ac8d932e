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

// launch-geometry knobs of the row-streaming kernels (target number of blocks); set by tools/eltwise_bench.py sweeps
namespace { int g_tune[PXL_TUNE_COUNT] = {1024, 2048, 2048, 2048, 16}; }
int pxl_tune_get(int key) { return (key >= 0 && key < PXL_TUNE_COUNT) ? g_tune[key] : 0; }
extern "C" int pxl_tune_set(int key, int value) {
  if (key < 0 || key >= PXL_TUNE_COUNT || value < 1) return pxl_set_error(PXL_ERR_ARG, "tune_set: bad key/value %d/%d", key, value);
  g_tune[key] = value;
  return PXL_OK;
}
