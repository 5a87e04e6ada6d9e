#include <stdarg.h>
#include <stdio.h>
#include "../../include/dynscaler_hip.h"

static thread_local char g_err[512] = "";

void ds_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* ds_last_error(void) { return g_err; }
extern "C" int ds_abi_version(void) { return DS_ABI_VERSION; }
