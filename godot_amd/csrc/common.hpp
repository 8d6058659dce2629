// Shared host/device helpers for libgodot_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <math.h>

#include "godot_hip.h"

// ---- host side -------------------------------------------------------------
void godot_set_error(const char *fmt, ...);

#define GODOT_REQUIRE(cond, code, ...)            \
    do {                                          \
        if (!(cond)) {                            \
            godot_set_error(__VA_ARGS__);         \
            return (code);                        \
        }                                         \
    } while (0)

#define GODOT_LAUNCH_CHECK(what)                                              \
    do {                                                                      \
        hipError_t e_ = hipGetLastError();                                    \
        if (e_ != hipSuccess) {                                               \
            godot_set_error("%s: %s", (what), hipGetErrorString(e_));         \
            return GODOT_E_LAUNCH;                                            \
        }                                                                     \
    } while (0)

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- device side -----------------------------------------------------------
constexpr int WAVE = 64;

// Butterfly reductions inside aligned groups of T lanes (T a power of two
// <= 64).  Every lane of the group ends with the bit-identical result, so
// data-dependent branches taken on it are uniform across the group.
template <int T>
__device__ __forceinline__ double team_sum(double v)
{
#pragma unroll
    for (int o = T / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}

template <int T>
__device__ __forceinline__ double team_max(double v)
{
#pragma unroll
    for (int o = T / 2; o > 0; o >>= 1) {
        double u = __shfl_xor(v, o, WAVE);
        v = (u > v) ? u : v;
    }
    return v;
}

// Python's max(a, b): b only if b > a (so a NaN second argument yields a).
__device__ __forceinline__ double pymax(double a, double b) { return (b > a) ? b : a; }
__device__ __forceinline__ double pymin(double a, double b) { return (b < a) ? b : a; }

__device__ __forceinline__ double qnan() { return __longlong_as_double(0x7ff8000000000000LL); }
