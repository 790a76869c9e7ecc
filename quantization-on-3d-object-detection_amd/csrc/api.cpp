// Error reporting + ABI version of libaq3d.so (host only).
#include <stdarg.h>
#include <stdio.h>
#include "../../include/aq3d.h"

static thread_local char g_err[512] = {0};

void aq3d_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *aq3d_last_error_string(void) { return g_err; }
extern "C" int aq3d_abi_version(void) { return 19; }
