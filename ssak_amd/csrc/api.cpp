// Status plumbing shared by every entry point of libssak_hip.so.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/ssak_hip.h"

static thread_local char g_err[512] = "";

void ssak_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int ssak_version(void) { return 100; }
extern "C" const char* ssak_last_error(void) { return g_err; }
