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

// First statement of every entry point that launches kernels: hipGetLastError keeps the
// error of whatever HIP call failed last in this host thread, the caller's own included
// (a device query before the runtime was up, say), and GODOT_LAUNCH_CHECK would report
// that stale error as this call's.
#define GODOT_ENTRY() (void)hipGetLastError()

#define GODOT_LAUNCH_CHECK(what)                                              \
    do {                                                                      \
        hipError_t e_ = hipGetLastError();                                    \
        if (e_ != hipSuccess) {                                               \
            godot_set_error("%s: %s", (what), hipGetErrorString(e_));         \
            return GODOT_E_LAUNCH;                                            \
        }                                                                     \
    } while (0)

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// The number of photons the cell kernels size their shape for (lanes and register slots per
// cell): the length of the weight array, unless the entry point was told how many photons lie
// inside the cells (the godot_*_shaped entry points; a subset of cells handed over with the
// whole weight array would otherwise get a shape for cells many times their size).
int64_t godot_shape_photons(int64_t nph);
struct GodotShapeHint {
    explicit GodotShapeHint(int64_t photons_in_cells);
    ~GodotShapeHint();
};

// ---- device side -----------------------------------------------------------
constexpr int WAVE = 64;

// Cross-lane moves by DPP (data-parallel primitives: no LDS round trip).
//   0xB1  quad_perm [1,0,3,2]   lane ^ 1
//   0x4E  quad_perm [2,3,0,1]   lane ^ 2
//   0x141 row_half_mirror       lane -> 7 - lane inside each group of 8
//   0x140 row_mirror            lane -> 15 - lane inside each row of 16
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v)
{
    return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, true);
}

// Reductions inside aligned groups of T lanes (T a power of two <= 64).
// Each step pairs two disjoint sub-groups whose lanes already agree, and
// + / max / * are commutative, so every lane of the group ends with the
// bit-identical result: branches taken on it are uniform across the group.
// Beyond sixteen lanes the rows' results (every lane of a row holds its row's by then) are
// read as scalars and combined in the butterfly's own order, (r0 op r1) op (r2 op r3): the
// same bits as exchanging them through the LDS crossbar (ds_bpermute), without its latency.
// Preconditions: a one-dimensional workgroup (a team is the aligned lane range
// threadIdx.x & ~(T - 1) .. + T - 1 of its wavefront), and for T >= 32 every lane of the team
// active at the call (v_readlane takes rows 0, 16, 32, 48 whether or not they are active; a team
// that calls with some lanes switched off reads what those lanes held last).
#define GODOT_TEAM_REDUCE(NAME, TYPE, MOVE, LANE, OP)                    \
    template <int T>                                                     \
    __device__ __forceinline__ TYPE NAME(TYPE v)                         \
    {                                                                    \
        if (T >= 2) { TYPE u = MOVE<0xB1>(v); v = OP(v, u); }            \
        if (T >= 4) { TYPE u = MOVE<0x4E>(v); v = OP(v, u); }            \
        if (T >= 8) { TYPE u = MOVE<0x141>(v); v = OP(v, u); }           \
        if (T >= 16) { TYPE u = MOVE<0x140>(v); v = OP(v, u); }          \
        if (T >= 32) {                                                   \
            const TYPE r0 = LANE(v, 0), r1 = LANE(v, 16), r2 = LANE(v, 32), r3 = LANE(v, 48); \
            const TYPE a = OP(r0, r1), b = OP(r2, r3);                   \
            if (T >= 64) v = OP(a, b);                                   \
            else v = (threadIdx.x & 32) ? b : a;                         \
        }                                                                \
        return v;                                                        \
    }
__device__ __forceinline__ double lane_f64(double v, int l)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int lane_i32(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
#define GODOT_OP_ADD(a, b) ((a) + (b))
#define GODOT_OP_MUL(a, b) ((a) * (b))
#define GODOT_OP_MAX(a, b) (((b) > (a)) ? (b) : (a))
GODOT_TEAM_REDUCE(team_sum, double, dpp_f64, lane_f64, GODOT_OP_ADD)
GODOT_TEAM_REDUCE(team_prod, double, dpp_f64, lane_f64, GODOT_OP_MUL)
GODOT_TEAM_REDUCE(team_max, double, dpp_f64, lane_f64, GODOT_OP_MAX)
GODOT_TEAM_REDUCE(team_sum_i32, int, dpp_i32, lane_i32, GODOT_OP_ADD)
#define GODOT_OP_FMAX(a, b) fmax((a), (b))
#define GODOT_OP_IMIN(a, b) (((b) < (a)) ? (b) : (a))
GODOT_TEAM_REDUCE(team_fmax, double, dpp_f64, lane_f64, GODOT_OP_FMAX)    // ignores NaN
GODOT_TEAM_REDUCE(team_min_i32, int, dpp_i32, lane_i32, GODOT_OP_IMIN)
GODOT_TEAM_REDUCE(team_max_i32, int, dpp_i32, lane_i32, GODOT_OP_MAX)

// 1/d for finite d away from 0 and from the ends of the exponent range:
// hardware estimate + two Newton steps (about 1 ulp).  No scaling, no
// special-case fix-up: callers guarantee the range.
__device__ __forceinline__ double rcp_fast(double d)
{
    double x = __builtin_amdgcn_rcp(d);
    double e = fma(-d, x, 1.0);
    x = fma(x, e, x);
    e = fma(-d, x, 1.0);
    return fma(x, e, x);
}

// One Newton step only: relative error about 2^-44, for intermediate passes.
__device__ __forceinline__ double rcp_fast1(double d)
{
    double x = __builtin_amdgcn_rcp(d);
    double e = fma(-d, x, 1.0);
    return fma(x, e, x);
}

// log(m * 2^e) for a mantissa m in [0.5, 1) as frexp delivers it (what the running
// products of the likelihood are kept as); a zero product gives -inf, NaN gives NaN.
// m is moved to [sqrt(1/2), sqrt(2)), s = (m - 1) / (m + 1), and
// log m = 2 s (1 + s^2/3 + s^4/5 + ... + s^16/17); |s| <= 0.172 leaves a truncation
// error below 1e-15 relative.  About a third of the instructions of the library log.
__device__ __forceinline__ double log_mant(double m, int e)
{
    if (!(m > 0)) return m == 0 ? -INFINITY : __longlong_as_double(0x7ff8000000000000LL);
    const bool low = m < 0.7071067811865476;
    m = low ? m + m : m;
    e = low ? e - 1 : e;
    const double f = m - 1.0;                   // exact
    const double s = f * rcp_fast(2.0 + f);
    const double z = s * s;
    double q = 1.0 / 17.0;
    q = fma(q, z, 1.0 / 15.0);
    q = fma(q, z, 1.0 / 13.0);
    q = fma(q, z, 1.0 / 11.0);
    q = fma(q, z, 1.0 / 9.0);
    q = fma(q, z, 1.0 / 7.0);
    q = fma(q, z, 1.0 / 5.0);
    q = fma(q, z, 1.0 / 3.0);
    const double s2 = s + s;
    return fma((double)e, 0.6931471805599453, fma(s2 * z, q, s2));
}

// log(x * 2^e) for a positive, normal x
__device__ __forceinline__ double log_scaled(double x, int e)
{
    return log_mant(__builtin_amdgcn_frexp_mant(x), e + __builtin_amdgcn_frexp_exp(x));
}

// Reciprocals of four positive numbers from ONE hardware reciprocal: invert the
// product, then peel the factors off again (nine multiplications).  The
// hardware reciprocal is a quarter-rate instruction, so this costs about 60 %
// of four separate rcp_fast1.  Relative error about 2^-46 plus three roundings.
// The product must stay inside the exponent range: |log10 d_i| < 75.
__device__ __forceinline__ void rcp4(double d0, double d1, double d2, double d3, double &x0, double &x1,
                                     double &x2, double &x3)
{
    const double ab = d0 * d1, cd = d2 * d3;
    const double x = rcp_fast1(ab * cd);
    const double iab = x * cd, icd = x * ab;
    x0 = iab * d1;
    x1 = iab * d0;
    x2 = icd * d3;
    x3 = icd * d2;
}

// Python's max(a, b): b only if b > a (so a NaN second argument yields a).
__device__ __forceinline__ double pymax(double a, double b) { return (b > a) ? b : a; }
__device__ __forceinline__ double pymin(double a, double b) { return (b < a) ? b : a; }

__device__ __forceinline__ double qnan() { return __longlong_as_double(0x7ff8000000000000LL); }
