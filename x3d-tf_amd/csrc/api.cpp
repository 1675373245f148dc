// error channel + version of the C ABI
#include <stdarg.h>
#include <stdio.h>

#include "../../include/x3d_hip.h"

static thread_local char g_err[512] = "";

void x3d_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* x3d_last_error(void) { return g_err; }
extern "C" int x3d_version(void) { return 100; }
