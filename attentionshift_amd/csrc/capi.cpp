// Error plumbing and version of libattnshift_hip.so (see include/attnshift.h).
#include <stdarg.h>

#include "common.h"

static thread_local char g_err[512] = "";

void as_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int as_version(void) { return AS_VERSION; }
extern "C" const char* as_last_error(void) { return g_err; }
