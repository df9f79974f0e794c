// api.hip -- error reporting and version of the C ABI (include/clhip.h).
#include "common.h"

static thread_local char g_err[512] = "";

void clhip_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* clhip_last_error(void) { return g_err; }
extern "C" int clhip_version(void) { return 100; }
