// Thread-local error string and version of libgodot_hip.
#include <stdarg.h>
#include <stdio.h>

#include "godot_hip.h"

static thread_local char g_err[512] = "";

void godot_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *godot_last_error(void) { return g_err; }
extern "C" int godot_version(void) { return 100; }
