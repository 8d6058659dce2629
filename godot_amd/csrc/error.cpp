// Thread-local error string and version of libgodot_hip.
#include <stdarg.h>
#include <stdio.h>

#include "common.hpp"

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

static thread_local int64_t g_shape_photons = -1;

int64_t godot_shape_photons(int64_t nph) { return g_shape_photons >= 0 ? g_shape_photons : nph; }
GodotShapeHint::GodotShapeHint(int64_t photons_in_cells) { g_shape_photons = photons_in_cells; }
GodotShapeHint::~GodotShapeHint() { g_shape_photons = -1; }
