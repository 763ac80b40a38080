// Version and error reporting of the C ABI.
#include <string.h>

#include "common.h"

static thread_local char g_err[512] = "";

void mcdseg_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int mcdseg_version(void) { return MCDSEG_VERSION; }
extern "C" const char* mcdseg_last_error(void) { return g_err; }
