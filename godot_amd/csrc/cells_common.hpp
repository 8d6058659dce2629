// Per-cell likelihood building blocks shared by the cell kernels (gfx950):
// how a team of lanes or a single lane holds one cell's weights, and the
// reference's fixed-schedule maximisers written once over that interface.
#pragma once
#include "common.hpp"

// sparse-cell fit kernel (cells_group.hip), launched from godot_cell_fit
namespace godot_detail {
bool fit_group_shape(int64_t nph, int64_t ncell, int &L, int &R, bool grid);
bool fit_group_launch(int L, int R, const double *we, int64_t nph, const int64_t *lo, const int64_t *hi,
                      const double *S, const double *B, int64_t ncell, const double *guess, double guess0,
                      const double *beta, double beta0, int swap, double *alpha, double *ts, int32_t *flags,
                      hipStream_t st);
// very large cells, one workgroup each (cells_block.hip)
bool block_cells(int64_t nph, int64_t ncell);
bool logpdf_block_cells(int64_t nph, int64_t ncell);
bool fit_block_cells(int64_t nph, int64_t ncell);
void fit_block_launch(const double *we, int64_t nph, const int64_t *lo, const int64_t *hi, const double *S,
                      const double *B, int64_t ncell, const double *guess, double guess0, const double *beta,
                      double beta0, int swap, double *alpha, double *ts, int32_t *flags, hipStream_t st);
void logpdf_block_launch(const double *we, int64_t nph, const int64_t *lo, const int64_t *hi, const double *S,
                         const double *B, int64_t ncell, const double *aopt, double dlogl, int npt,
                         int include_zero, int swap, double *dom, double *cod, int32_t *flags, hipStream_t st);
bool logpdf_group_launch(int L, int R, const double *we, int64_t nph, const int64_t *lo, const int64_t *hi,
                         const double *S, const double *B, int64_t ncell, const double *aopt, double dlogl,
                         int npt, int include_zero, int swap, double *dom, double *cod, int32_t *flags,
                         hipStream_t st);
}

namespace {


struct CellArgs {
    const double *we;
    int64_t nph;
    const int64_t *lo;
    const int64_t *hi;
    const double *S;
    const double *B;
    int64_t ncell;
    int swap;
};

// Weights of one cell as seen by one lane of its team.
template <int T, int R>
struct CellData {
    double w[R];
    const double *p;
    int64_t n;
    int lane;
    int nr;   // photons held in registers (0 when the cell is streamed)
    bool in_regs;
    bool swap;

    __device__ __forceinline__ double ld(int64_t i) const
    {
        double v = p[i];
        return swap ? 1.0 - v : v;
    }

    __device__ __forceinline__ void load(const double *base, int64_t n_, int lane_, bool swap_)
    {
        p = base;
        n = n_;
        lane = lane_;
        swap = swap_;
        in_regs = n <= (int64_t)T * R;
        nr = in_regs ? (int)n : 0;
        if (in_regs) {
            // branch-free: every lane loads R (clamped) addresses back to back and
            // zeroes what lies beyond the cell
            const int last = nr > 0 ? nr - 1 : 0;
#pragma unroll
            for (int r = 0; r < R; ++r) w[r] = 0.0;
            if (nr > 0) {
                // opaque copy: keeps the R slot indices from being hoisted out of an
                // enclosing loop and held in registers for its whole duration
                int ln = lane;
                asm volatile("" : "+v"(ln));
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int i = ln + r * T;
                    const double v = ld(i < last ? i : last);
                    w[r] = (i < nr) ? v : 0.0;
                }
            }
        }
    }

    // slot r of this lane's register copy holds a photon of the cell
    __device__ __forceinline__ bool has(int r) const { return lane + r * T < nr; }

    // acc[k] = sum over the cell's photons of what f adds, for k < NS.
    // ZS: f adds exactly nothing for a zero weight, so the zero padding of the
    // register copy needs no predication.
    template <int NS, bool ZS = false, class F>
    __device__ __forceinline__ void sums(F f, double (&acc)[NS]) const
    {
#pragma unroll
        for (int k = 0; k < NS; ++k) acc[k] = 0.0;
        if (in_regs) {
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (ZS || has(r)) f(w[r], acc);
        } else {
            // streamed cells: four loads of a lane in flight
            int64_t i = lane;
            for (; i + 3 * T < n; i += 4 * T) {
                const double w0 = ld(i), w1 = ld(i + T), w2 = ld(i + 2 * T), w3 = ld(i + 3 * T);
                f(w0, acc);
                f(w1, acc);
                f(w2, acc);
                f(w3, acc);
            }
            for (; i < n; i += T) f(ld(i), acc);
        }
#pragma unroll
        for (int k = 0; k < NS; ++k) acc[k] = team_sum<T>(acc[k]);
    }

    // acc[k] = sum over the cell's photons of what accf(w, d, 1/d, acc) adds, with
    // d = den(w) > 0.  Both must treat a zero weight as "adds nothing" (den(0) finite):
    // the zero-padded register copy is used without predication, four at a time.
    template <int NS, class DEN, class ACC>
    __device__ __forceinline__ void sums_rcp(DEN den, ACC accf, double (&acc)[NS]) const
    {
#pragma unroll
        for (int k = 0; k < NS; ++k) acc[k] = 0.0;
        if (in_regs) {
            if constexpr (R % 4 == 0) {
#pragma unroll
                for (int r = 0; r < R; r += 4) {
                    const double d0 = den(w[r]), d1 = den(w[r + 1]), d2 = den(w[r + 2]), d3 = den(w[r + 3]);
                    double x0, x1, x2, x3;
                    rcp4(d0, d1, d2, d3, x0, x1, x2, x3);
                    accf(w[r], d0, x0, acc);
                    accf(w[r + 1], d1, x1, acc);
                    accf(w[r + 2], d2, x2, acc);
                    accf(w[r + 3], d3, x3, acc);
                }
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const double d = den(w[r]);
                    accf(w[r], d, rcp_fast1(d), acc);
                }
            }
        } else {
            // streamed cells: four loads of a lane in flight, their reciprocals from one rcp
            int64_t i = lane;
            for (; i + 3 * T < n; i += 4 * T) {
                const double w0 = ld(i), w1 = ld(i + T), w2 = ld(i + 2 * T), w3 = ld(i + 3 * T);
                const double d0 = den(w0), d1 = den(w1), d2 = den(w2), d3 = den(w3);
                double x0, x1, x2, x3;
                rcp4(d0, d1, d2, d3, x0, x1, x2, x3);
                accf(w0, d0, x0, acc);
                accf(w1, d1, x1, acc);
                accf(w2, d2, x2, acc);
                accf(w3, d3, x3, acc);
            }
            for (; i < n; i += T) {
                const double wv = ld(i), d = den(wv);
                accf(wv, d, rcp_fast1(d), acc);
            }
        }
#pragma unroll
        for (int k = 0; k < NS; ++k) acc[k] = team_sum<T>(acc[k]);
    }

    // combines the lanes' partial products (each a mantissa in [0.5, 1) and an exponent)
    __device__ __forceinline__ void reduce_prod(double &pr, int &ex) const
    {
        pr = team_prod<T>(pr);   // factors in [0.5, 1): at worst 2^-64
        ex = team_sum_i32<T>(ex);
        ex += __builtin_amdgcn_frexp_exp(pr);
        pr = __builtin_amdgcn_frexp_mant(pr);
    }

    // sum over the cell of log(t_i) and of g_i, where f(w, t, g) yields the
    // positive t_i and the addend g_i of one photon.  The logarithms are taken
    // of the running product (mantissa renormalised into [0.5, 1) every eight
    // factors, exponent kept as an integer): one log per cell, not per photon.
    template <class F>
    __device__ __forceinline__ void logsum(F f, double &logs, double &gsum) const
    {
        double pr = 1.0, gs = 0.0;
        int ex = 0;
        auto renorm = [&]() {
            ex += __builtin_amdgcn_frexp_exp(pr);
            pr = __builtin_amdgcn_frexp_mant(pr);
        };
        if (in_regs) {
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (has(r)) {
                    double t, g;
                    f(w[r], t, g);
                    pr *= t;
                    gs += g;
                    if ((r & 7) == 7) renorm();
                }
        } else {
            int k = 0;
            for (int64_t i = lane; i < n; i += T) {
                double t, g;
                f(ld(i), t, g);
                pr *= t;
                gs += g;
                if ((++k & 7) == 0) renorm();
            }
        }
        renorm();
        pr = team_prod<T>(pr);   // factors in [0.5, 1): at worst 2^-64
        ex = team_sum_i32<T>(ex);
        gsum = team_sum<T>(gs);
        logs = log_scaled(pr, ex);
    }

    // logs = sum over the cell of log(t_i), gsum = sum of gf(w_i, 1/t_i), with
    // t_i = den(w_i) > 0; den(0) must be 1 and gf(0, .) must be 0 (zero padding).
    // The logarithm is taken of the running product, the reciprocals four at a time.
    template <class DEN, class GF>
    __device__ __forceinline__ void logsum_rcp(DEN den, GF gf, double &logs, double &gsum) const
    {
        double pr = 1.0, gs = 0.0;
        int ex = 0, k = 0;
        auto renorm = [&]() {
            ex += __builtin_amdgcn_frexp_exp(pr);
            pr = __builtin_amdgcn_frexp_mant(pr);
        };
        if (in_regs && R % 4 == 0) {
#pragma unroll
            for (int r = 0; r + 3 < R; r += 4) {
                const double d0 = den(w[r]), d1 = den(w[r + 1]), d2 = den(w[r + 2]), d3 = den(w[r + 3]);
                double x0, x1, x2, x3;
                rcp4(d0, d1, d2, d3, x0, x1, x2, x3);
                gs += gf(w[r], x0);
                gs += gf(w[r + 1], x1);
                gs += gf(w[r + 2], x2);
                gs += gf(w[r + 3], x3);
                pr *= (d0 * d1) * (d2 * d3);
                if ((r & 4) != 0) renorm();
            }
        } else if (in_regs) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const double d = den(w[r]);
                gs += gf(w[r], rcp_fast1(d));
                pr *= d;
                if ((r & 7) == 7) renorm();
            }
        } else {
            int64_t i = lane;
            for (; i + 3 * T < n; i += 4 * T) {
                const double w0 = ld(i), w1 = ld(i + T), w2 = ld(i + 2 * T), w3 = ld(i + 3 * T);
                const double d0 = den(w0), d1 = den(w1), d2 = den(w2), d3 = den(w3);
                double x0, x1, x2, x3;
                rcp4(d0, d1, d2, d3, x0, x1, x2, x3);
                gs += gf(w0, x0);
                gs += gf(w1, x1);
                gs += gf(w2, x2);
                gs += gf(w3, x3);
                pr *= (d0 * d1) * (d2 * d3);
                if ((++k & 1) == 0) renorm();
            }
            for (; i < n; i += T) {
                const double wv = ld(i), d = den(wv);
                gs += gf(wv, rcp_fast1(d));
                pr *= d;
            }
        }
        renorm();
        pr = team_prod<T>(pr);   // factors in [0.5, 1): at worst 2^-64
        ex = team_sum_i32<T>(ex);
        gsum = team_sum<T>(gs);
        logs = log_scaled(pr, ex);
    }

    // largest weight of the cell (-inf for an empty cell)
    __device__ __forceinline__ double wmax() const
    {
        double m = -INFINITY;
        if (in_regs) {
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (has(r)) m = fmax(m, w[r]);
        } else {
            for (int64_t i = lane; i < n; i += T) m = fmax(m, ld(i));
        }
        return team_max<T>(m);
    }

    // sum over the cell of log(t1_i) - log(t2_i), f(w, t1, t2): one log of the
    // ratio of the two renormalised products
    template <bool ZS = false, class F>
    __device__ __forceinline__ double logratio(F f) const
    {
        double p1 = 1.0, p2 = 1.0;
        int ex = 0;
        auto renorm = [&]() {
            ex += __builtin_amdgcn_frexp_exp(p1) - __builtin_amdgcn_frexp_exp(p2);
            p1 = __builtin_amdgcn_frexp_mant(p1);
            p2 = __builtin_amdgcn_frexp_mant(p2);
        };
        if (in_regs) {
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (ZS || has(r)) {
                    double t1, t2;
                    f(w[r], t1, t2);
                    p1 *= t1;
                    p2 *= t2;
                    if ((r & 7) == 7) renorm();
                }
        } else {
            int k = 0;
            for (int64_t i = lane; i < n; i += T) {
                double t1, t2;
                f(ld(i), t1, t2);
                p1 *= t1;
                p2 *= t2;
                if ((++k & 7) == 0) renorm();
            }
        }
        renorm();
        p1 = team_prod<T>(p1);
        p2 = team_prod<T>(p2);
        ex = team_sum_i32<T>(ex);
        return log_scaled(p1 * rcp_fast(p2), ex);
    }

    template <class F>
    __device__ __forceinline__ double maxof(F f) const
    {
        double m = -INFINITY;
        if (in_regs) {
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (has(r)) m = fmax(m, f(w[r]));
        } else {
            for (int64_t i = lane; i < n; i += T) m = fmax(m, f(ld(i)));
        }
        return team_max<T>(m);
    }
};

// A cell read from memory in every pass (no register copy): cheap to hand to
// out-of-line functions, used on the rare slow paths.
template <int T>
using StreamCell = CellData<T, 1>;

template <int T>
__device__ __forceinline__ void stream_load(StreamCell<T> &cd, const double *base, int64_t n, int lane,
                                            bool swap)
{
    cd.p = base;
    cd.n = n;
    cd.lane = lane;
    cd.swap = swap;
    cd.in_regs = false;
    cd.nr = 0;
    cd.w[0] = 0.0;
}

// ---------------------------------------------------------------------------
// get_max with the background level fixed: reference core.py:632-691.
// g = guess - 1, b = beta - 1 (the reference's zero-based parameters).
// ---------------------------------------------------------------------------

// f(a) = sum w/(t2 + a w) - S with t2 = 1 + b (1 - w)
template <class CD, bool B0 = false>
__device__ double score(const CD &cd, double a, double b, double S)
{
    double acc[1];
    cd.template sums_rcp<1>(
        [=](double w) { return fma(a, w, B0 ? 1.0 : 1.0 + b * (1.0 - w)); },
        [](double w, double, double x, double (&s)[1]) { s[0] = fma(w, x, s[0]); }, acc);
    return acc[0] - S;
}

// scipy.optimize.brentq (Zeros/brentq.c) on the score function with beta = 1,
// as the reference's get_max_numerical uses it (core.py:728-739).
template <class CD>
__device__ __noinline__ double max_numerical(const CD &cd, double S, int &flags)
{
    flags |= GODOT_F_BRENT;
    if (score<CD, true>(cd, -1.0, 0.0, S) < 0) {
        flags |= GODOT_F_AT_ZERO;
        return 0.0;
    }
    double a0 = -1.0, amax = 0.0;
    for (int i = 0; i < 12; ++i) {
        if (score<CD, true>(cd, amax, 0.0, S) < 0) break;
        a0 = amax;
        amax = 2 * amax + 1;
    }
    const double xtol = 1e-3, rtol = 8.881784197001252e-16;
    double xpre = a0, xcur = amax, xblk = 0, fblk = 0, spre = 0, scur = 0;
    double fpre = score<CD, true>(cd, xpre, 0.0, S);
    double fcur = score<CD, true>(cd, xcur, 0.0, S);
    if (fpre == 0) return xpre + 1;
    if (fcur == 0) return xcur + 1;
    if (signbit(fpre) == signbit(fcur)) {
        flags |= GODOT_F_FAILED;   // the reference raises ValueError here
        return qnan();
    }
    for (int it = 0; it < 100; ++it) {
        if (fpre != 0 && fcur != 0 && (signbit(fpre) != signbit(fcur))) {
            xblk = xpre;
            fblk = fpre;
            spre = scur = xcur - xpre;
        }
        if (fabs(fblk) < fabs(fcur)) {
            xpre = xcur; xcur = xblk; xblk = xpre;
            fpre = fcur; fcur = fblk; fblk = fpre;
        }
        double delta = (xtol + rtol * fabs(xcur)) / 2;
        double sbis = (xblk - xcur) / 2;
        if (fcur == 0 || fabs(sbis) < delta) return xcur + 1;
        if (fabs(spre) > delta && fabs(fcur) < fabs(fpre)) {
            double stry;
            if (xpre == xblk) {
                stry = -fcur * (xcur - xpre) / (fcur - fpre);
            } else {
                double dpre = (fpre - fcur) / (xpre - xcur);
                double dblk = (fblk - fcur) / (xblk - xcur);
                stry = -fcur * (fblk * dblk - fpre * dpre) / (dblk * dpre * (fblk - fpre));
            }
            if (2 * fabs(stry) < fmin(fabs(spre), 3 * fabs(sbis) - delta)) {
                spre = scur;
                scur = stry;
            } else {
                spre = sbis;
                scur = sbis;
            }
        } else {
            spre = sbis;
            scur = sbis;
        }
        xpre = xcur;
        fpre = fcur;
        if (fabs(scur) > delta) xcur += scur;
        else xcur += (sbis > 0 ? delta : -delta);
        fcur = score<CD, true>(cd, xcur, 0.0, S);
    }
    flags |= GODOT_F_FAILED;
    return qnan();
}

// log of a product kept as mantissa in [0.5, 1) and binary exponent
struct LogProd {
    double m;
    int e;
    bool valid;
};

template <class CD, bool B0, bool CHECK0 = true, bool WANT_L0 = false>
__device__ bool fit_attempt_impl(const CD &cd, double S, double g, double b, double &a_out, bool &at_zero,
                                 LogProd *l0 = nullptr);

// The flags a fitted cell leaves with: a maximum at the boundary is marked whichever way
// the schedule got there (the sign of the derivative at zero, or an iterate clamped to it).
__device__ __forceinline__ int fit_flags(int fl, double alpha)
{
    return alpha == 0.0 ? (fl | GODOT_F_AT_ZERO) : fl;
}

// One pass of the fixed schedule (core.py:639-680) from the zero-based start
// values g = guess - 1, b = beta - 1.  Returns true when the result `a` (zero-
// based) is final, false when the reference would call itself again with
// guess = a + 1 (core.py:684-689).
template <class CD>
__device__ bool fit_attempt(const CD &cd, double S, double g, double b, double &a_out,
                            bool &at_zero)
{
    // background level fixed at its nominal value (beta = 1): t2 = 1 exactly
    if (b == 0.0) return fit_attempt_impl<CD, true>(cd, S, g, 0.0, a_out, at_zero);
    return fit_attempt_impl<CD, false>(cd, S, g, b, a_out, at_zero);
}


// CHECK0 = false: the caller knows that the derivative at alpha = 0 is not
// negative (a restart: the first pass established it) and the pass that
// evaluates it is skipped.
// WANT_L0: the pass that looks at alpha = 0 also forms prod (1 - w_i), the
// likelihood at alpha = 0 that the test statistic needs, when it runs in its
// fused form (l0->valid says whether it did).
template <class CD, bool B0, bool CHECK0, bool WANT_L0>
__device__ bool fit_attempt_impl(const CD &cd, double S, double g, double b, double &a_out, bool &at_zero,
                                 LogProd *l0)
{
    double acc[3];
    at_zero = false;
    if (B0) b = 0.0;
    // t2 = 1 + b (1 - w)
    auto t2of = [=](double w) { return B0 ? 1.0 : 1.0 + b * (1.0 - w); };
    // derivative at alpha = 0 (core.py:639-644) fused with the sums of the
    // first Newton step when that starts from a = 0, b = 0 (w / 1 = w exactly)
    double a = g;
    if constexpr (WANT_L0) l0->valid = false;
    if (B0 && g == 0.0) {
        double pr = 1.0;
        int ex = 0, k = 0;
        cd.template sums_rcp<3>(
            [](double w) { return 1.0 - w; },
            [&](double w, double iw, double x, double (&s)[3]) {
                s[0] = fma(w, x, s[0]);
                s[1] += w;
                s[2] = fma(w, w, s[2]);
                if constexpr (WANT_L0) {
                    pr *= iw;
                    if ((++k & 7) == 0) {
                        ex += __builtin_amdgcn_frexp_exp(pr);
                        pr = __builtin_amdgcn_frexp_mant(pr);
                    }
                }
            }, acc);
        if constexpr (WANT_L0) {
            ex += __builtin_amdgcn_frexp_exp(pr);
            pr = __builtin_amdgcn_frexp_mant(pr);
            cd.reduce_prod(pr, ex);
            l0->m = pr;
            l0->e = ex;
            l0->valid = true;
        }
        if (acc[0] - S < 0) {
            at_zero = true;
            a_out = -1.0;
            return true;
        }
        a = a + (acc[1] - S) * rcp_fast(acc[2]);
    } else {
        if (CHECK0 && score<CD, B0>(cd, -1.0, b, S) < 0) {
            at_zero = true;
            a_out = -1.0;
            return true;
        }
        cd.template sums_rcp<2>(
            [=](double w) { return fma(g, w, t2of(w)); },
            [](double w, double, double x, double (&s)[2]) {
                const double r = w * x;
                s[0] += r;
                s[1] = fma(r, r, s[1]);
            }, reinterpret_cast<double (&)[2]>(acc));
        a = a + (acc[0] - S) * rcp_fast(acc[1]);
    }
    // Newton 1 keeps away from zero flux (core.py:653-655)
    if (a < 0 - 1) a = 0.2 - 1;
    auto pass2 = [&](double a_) {
        cd.template sums_rcp<2>(
            [=](double w) { return fma(a_, w, t2of(w)); },
            [](double w, double, double x, double (&s)[2]) {
                const double r = w * x;
                s[0] += r;
                s[1] = fma(r, r, s[1]);
            }, reinterpret_cast<double (&)[2]>(acc));
    };
    // Newton 2 (core.py:658-664)
    pass2(a);
    a = a + (acc[0] - S) * rcp_fast(acc[1]);
    if (a < 0.05 - 1) a = 0.05 - 1;
    // Newton 3 (core.py:667-671)
    pass2(a);
    a = pymax(0.0 - 1, a + (acc[0] - S) * rcp_fast(acc[1]));
    double alast = a;
    // Halley (core.py:674-680)
    {
        double a_ = a;
        cd.template sums_rcp<3>(
            [=](double w) { return fma(a_, w, t2of(w)); },
            [](double w, double, double x, double (&s)[3]) {
                const double r = w * x;
                const double r2 = r * r;
                s[0] += r;
                s[1] += r2;
                s[2] = fma(r2, r, s[2]);
            }, acc);
    }
    double f1 = acc[0] - S, f2 = acc[1], f3 = 2 * acc[2];
    a = pymax(0.0 - 1, a + 2 * f1 * f2 * rcp_fast(2 * f2 * f2 - f1 * f3));
    a_out = a;
    // slow convergence or a far-away start: the reference tries again
    return !((fabs(a - alast) > 0.05) || (fabs(g - a) > 10));
}

// get_max in place: used where a cell's weights are already loaded and no
// other team can take over its restarts (the profile-grid kernel).
//   hand_on   not null: when the restarts run out, the bracketing fallback is left to the
//             caller (*hand_on = true, NaN returned).  max_numerical is a function of its own
//             and takes the cell by reference, so a cell that is passed to it lives in memory
//             as well as in registers: for a cell held in tens of registers that is a copy
//             written to scratch memory by every wavefront, used by almost none.
template <class CD>
__device__ double fit_alpha(const CD &cd, double S, double guess, double beta, int &flags, bool *hand_on = nullptr)
{
    double g = guess - 1;
    double b = beta - 1;
    for (int depth = 0;; ++depth) {
        flags = (flags & ~GODOT_F_DEPTH_MASK) | (depth << 1);
        double a;
        bool at_zero;
        if (fit_attempt(cd, S, g, b, a, at_zero)) {
            if (at_zero) {
                flags |= GODOT_F_AT_ZERO;
                return 0.0;
            }
            return a + 1;
        }
        if (depth > 2) {
            if (hand_on) {
                *hand_on = true;
                return qnan();
            }
            return max_numerical(cd, S, flags);
        }
        g = (a + 1) - 1;
        b = (b + 1) - 1;
    }
}

// logL(alpha) = sum log(alpha w + (1 - w)) - S alpha   (core.py:353-364)
template <class CD>
__device__ double loglike(const CD &cd, double S, double alpha)
{
    // a negative factor makes the reference's sum of logarithms NaN (core.py:361-364);
    // the product of two of them would not show it, so they are counted
    double ls, nneg;
    cd.logsum([=](double w, double &t, double &g) {
        t = fma(w, alpha, 1.0 - w);
        g = t < 0.0 ? 1.0 : 0.0;
    }, ls, nneg);
    if (nneg > 0.0) ls = qnan();
    return ls - S * alpha;
}

// 2 (logL(aopt) - logL(0))   (core.py:982-994): the two sums of logarithms
// are taken together as one log of a ratio of renormalised products.
template <class CD>
__device__ double test_statistic(const CD &cd, double S, double aopt)
{
    if (S == 0 || aopt == 0) return 0.0;
    double dl = cd.template logratio<true>([=](double w, double &t1, double &t2) {
        t2 = 1.0 - w;
        t1 = fma(aopt, w, t2);
    });
    return 2 * (dl - S * aopt);
}

// Running products over a cell's wave-uniform weights (they arrive by scalar loads):
// pr[u] *= fac(w_j, u) for every photon j < n and every one of the lane's PPL grid points,
// mantissas renormalised after every eight factors.  The last n mod 8 weights are asked for
// in ONE request of eight, which reads on into whatever follows the cell but never beyond
// the `room` weights that exist from w[0] on; a request per weight makes the wavefront wait
// out the latency of a scalar load for each of them (3.5 times a cell on average).
template <int PPL, class FAC>
__device__ __forceinline__ void grid_products(const double *__restrict__ w, int n, int64_t room,
                                              double (&pr)[PPL], int (&ex)[PPL], FAC fac)
{
    int j = 0;
    for (; j + 8 <= n; j += 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const double wj = w[j + k];
#pragma unroll
            for (int u = 0; u < PPL; ++u) pr[u] *= fac(wj, u);
        }
#pragma unroll
        for (int u = 0; u < PPL; ++u) {
            ex[u] += __builtin_amdgcn_frexp_exp(pr[u]);
            pr[u] = __builtin_amdgcn_frexp_mant(pr[u]);
        }
    }
    const int rem = n - j;
    if (rem <= 0) return;
    if ((int64_t)j + 8 <= room) {
        double wt[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) wt[k] = w[j + k];
        // all eight before the first is used: keeps the compiler from splitting the request
        // into the part that is always needed and the part that may not be
        asm volatile("" : "+s"(wt[0]), "+s"(wt[1]), "+s"(wt[2]), "+s"(wt[3]), "+s"(wt[4]), "+s"(wt[5]),
                          "+s"(wt[6]), "+s"(wt[7]));
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            if (k < rem) {
#pragma unroll
                for (int u = 0; u < PPL; ++u) pr[u] *= fac(wt[k], u);
            }
        }
    } else {
        // the last cells of the photon array
        for (; j < n; ++j) {
            const double wj = w[j];
#pragma unroll
            for (int u = 0; u < PPL; ++u) pr[u] *= fac(wj, u);
        }
    }
}

struct Support {
    double amin, amax, llmax;
};

// Support of the profile (core.py:862-916): the interval on which logL stays
// within dlogl of its maximum, found by a few Newton steps on each side.
template <class CD>
__device__ Support support_search(const CD &cd, double S, double aopt, double dlogl,
                                  bool include_zero)
{
    Support r;
    double amin = 0.0, amax, llmax;
    double acc[2];
    // logL(a) + S a and its derivative + S; alpha w + (1 - w) formed as 1 + (alpha - 1) w.
    // A Newton step of the searches below can land on a flux at which some factor is
    // negative: the reference's sum of logarithms is NaN there and so is everything it
    // returns (core.py:876-878, 891-893).  The logarithm of the product would hide an even
    // number of such factors, so a negative factor (its reciprocal is negative) makes the
    // derivative sum NaN, which the step then carries into the end of the support.
    auto f0f1 = [&](double a) {
        const double am1 = a - 1.0;
        cd.logsum_rcp([=](double w) { return fma(am1, w, 1.0); },
                      [](double w, double x) { return x < 0.0 ? qnan() : w * x; }, acc[0], acc[1]);
    };
    if (aopt == 0) {   // core.py:869-880
        cd.logsum_rcp([](double w) { return 1.0 - w; },
                      [](double w, double x) { return w * x; }, acc[0], acc[1]);
        llmax = acc[0];
        amax = pymax(0.0, -(llmax + dlogl) * rcp_fast(acc[1] - S));
        for (int i = 0; i < 10; ++i) {
            f0f1(amax);
            double f0 = acc[0] - amax * S + dlogl - llmax;
            double f1 = acc[1] - S;
            amax = amax - f0 * rcp_fast(f1);
            if (fabs(f0) < 0.1) break;
        }
    } else {           // core.py:881-895
        const double am1 = aopt - 1.0;
        cd.logsum_rcp([=](double w) { return fma(am1, w, 1.0); },
                      [](double w, double x) {
                          const double q = w * x;
                          return q * q;
                      }, acc[0], acc[1]);
        llmax = acc[0] - S * aopt;
        const double half = sqrt(2 * dlogl * rcp_fast(fabs(acc[1])));
        amax = aopt + half;
        for (int i = 0; i < 5; ++i) {
            f0f1(amax);
            double f0 = acc[0] - amax * S + dlogl - llmax;
            double f1 = acc[1] - S;
            amax = amax - f0 * rcp_fast(f1);
            if (fabs(f0) < 0.1) break;
        }
        if (!include_zero && aopt > 0) {   // core.py:897-914
            // max over the photons of 1 - 1/w is 1 - 1/max(w)
            double min_a = (1 - 1. / cd.wmax()) + 1e-6;
            amin = pymax(min_a, aopt - half);
            for (int i = 0; i < 5; ++i) {
                f0f1(amin);
                double f0 = acc[0] - amin * S + dlogl - llmax;
                double f1 = acc[1] - S;
                amin = pymax(min_a, amin - f0 * rcp_fast(f1));
                if (fabs(f0) < 0.1) break;
            }
        }
    }
    amin = pymax(0.0, amin);   // core.py:916
    r.amin = amin;
    r.amax = amax;
    r.llmax = llmax;
    return r;
}

// numpy.linspace(amin, amax, npt)[i]
__device__ __forceinline__ double grid_point(double amin, double amax, double step, int i, int npt)
{
#pragma clang fp contract(off)   // numpy rounds the product before the sum
    if (i == npt - 1 && npt > 1) return amax;
    double prod = (double)i * step;
    return prod + amin;
}

template <int T>
__device__ __forceinline__ bool cell_setup(const CellArgs &p, int64_t &cell, int &lane,
                                           int64_t &lo, int64_t &n, double &S, double &B, bool &bad)
{
    lane = threadIdx.x & (T - 1);
    cell = (int64_t)blockIdx.x * (blockDim.x / T) + (threadIdx.x / T);
    if (cell >= p.ncell) return false;
    lo = p.lo[cell];
    int64_t hi = p.hi[cell];
    bad = (lo < 0) || (hi < lo) || (hi > p.nph);
    n = bad ? 0 : hi - lo;
    S = p.S[cell];
    B = p.B ? p.B[cell] : 0.0;
    if (p.swap) {
        double t = S; S = B; B = t;
    }
    return true;
}

// ---------------------------------------------------------------------------
// get_beta_max (core.py:741-823) and log_profile_likelihood (core.py:391-406)
// ---------------------------------------------------------------------------
template <class CD>
__device__ double fit_beta(const CD &cd, double B, double alpha, double guess, int &flags)
{
    if (isnan(alpha)) return 1.0;
    if (cd.n == 0) return 0.0;
    if (alpha == 0) return (double)cd.n / B;
    double al = alpha - 1;
    double g = guess - 1;
    for (int depth = 0;; ++depth) {
        flags = (flags & ~GODOT_F_DEPTH_MASK) | (depth << 1);
        double acc[3];
        cd.template sums<1>([](double w, double (&s)[1]) { s[0] += (1.0 - w) / w; },
                            reinterpret_cast<double (&)[1]>(acc));
        if (acc[0] < B * (1 + al)) {
            flags |= GODOT_F_AT_ZERO;
            return 0.0;
        }
        double b = g;
        const double lows[3] = {0.2 - 1, 0.05 - 1, 0.02 - 1};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            double b_ = b;
            cd.template sums<2>([=](double w, double (&s)[2]) {
                double iw = 1.0 - w;
                double r = iw / ((1.0 + al * w) + b_ * iw);
                s[0] += r;
                s[1] += r * r;
            }, reinterpret_cast<double (&)[2]>(acc));
            b = b + (acc[0] - B) / acc[1];
            b = pymax(lows[k], b);
        }
        double blast = b;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            double b_ = b;
            cd.template sums<3>([=](double w, double (&s)[3]) {
                double iw = 1.0 - w;
                double r = iw / ((1.0 + al * w) + b_ * iw);
                double r2 = r * r;
                s[0] += r;
                s[1] += r2;
                s[2] += r2 * r;
            }, acc);
            double f1 = acc[0] - B, f2 = acc[1], f3 = 2 * acc[2];
            double nb = pymax(0.0 - 1, b + 2 * f1 * f2 / (2 * f2 * f2 - f1 * f3));
            if (k == 0) {
                // step limiter of the first Halley iteration (core.py:799-803)
                if (fabs(nb - b) > 10) b = 2 * b + 1;
                else b = nb;
                blast = b;
            } else {
                b = nb;
            }
        }
        if ((fabs(b - blast) > 0.05) || (fabs(g - b) > 10) || (b == -1)) {
            if (depth > 2) {
                flags |= GODOT_F_FAILED;   // the reference raises ValueError
                return qnan();
            }
            al = (al + 1) - 1;
            g = (b + 1) - 1;
            continue;
        }
        return b + 1;
    }
}



// Team shape (T lanes x R registers) from the mean number of photons per cell:
// the smallest capacity T*R that holds the mean plus four (Poisson) sigma, so
// that almost every cell stays in registers, with as few lanes per cell as
// possible (fewer reduction steps, more cells per wavefront).
inline int pick_team(int64_t nph, int64_t ncell)
{
    double mean = ncell > 0 ? (double)godot_shape_photons(nph) / (double)ncell : 0.0;
    double need = mean + 4.0 * sqrt(mean) + 8.0;
    // T x R register slots, in steps of a quarter to a half: a shape twice too large
    // spends half of every pass on padding
    static const int slots[] = {32, 64, 96, 128, 160, 192, 256, 320, 384, 512, 640, 768, 1024, 1280, 1536};
    for (int k = 0; k < (int)(sizeof(slots) / sizeof(slots[0])); ++k)
        if (need <= slots[k]) return k;
    return 15;                    // 64 x 32, larger cells stream from L2
}

inline int check_cells(const void *we, int64_t nph, const void *lo, const void *hi, int64_t ncell,
                       const char *fn)
{
    GODOT_REQUIRE(ncell >= 0 && nph >= 0, GODOT_E_INVALID, "%s: negative size", fn);
    GODOT_REQUIRE(ncell < ((int64_t)1 << 31) * 4, GODOT_E_UNSUPPORTED, "%s: too many cells", fn);
    if (ncell == 0) return GODOT_OK;
    GODOT_REQUIRE(lo && hi, GODOT_E_INVALID, "%s: null cell range pointer", fn);
    GODOT_REQUIRE(we || nph == 0, GODOT_E_INVALID, "%s: null weights pointer", fn);
    return GODOT_OK;
}

#define DISPATCH_ONE(T_, R_, KERNEL, ...)                                                  \
    {                                                                                       \
        dim3 grid((unsigned)ceil_div(ncell, 256 / (T_)));                                   \
        hipLaunchKernelGGL((KERNEL<T_, R_>), grid, dim3(256), 0, st, __VA_ARGS__);          \
    }                                                                                       \
    break;
#define DISPATCH_TEAM(CFG_, KERNEL, GRID_ARGS, ...)                                         \
    do {                                                                                    \
        switch (CFG_) {                                                                     \
        case 0: DISPATCH_ONE(4, 8, KERNEL, __VA_ARGS__)                                     \
        case 1: DISPATCH_ONE(8, 8, KERNEL, __VA_ARGS__)                                     \
        case 2: DISPATCH_ONE(8, 12, KERNEL, __VA_ARGS__)                                    \
        case 3: DISPATCH_ONE(8, 16, KERNEL, __VA_ARGS__)                                    \
        case 4: DISPATCH_ONE(8, 20, KERNEL, __VA_ARGS__)                                    \
        case 5: DISPATCH_ONE(8, 24, KERNEL, __VA_ARGS__)                                    \
        case 6: DISPATCH_ONE(8, 32, KERNEL, __VA_ARGS__)                                    \
        case 7: DISPATCH_ONE(16, 20, KERNEL, __VA_ARGS__)                                   \
        case 8: DISPATCH_ONE(16, 24, KERNEL, __VA_ARGS__)                                   \
        case 9: DISPATCH_ONE(16, 32, KERNEL, __VA_ARGS__)                                   \
        case 10: DISPATCH_ONE(32, 20, KERNEL, __VA_ARGS__)                                  \
        case 11: DISPATCH_ONE(32, 24, KERNEL, __VA_ARGS__)                                  \
        case 12: DISPATCH_ONE(32, 32, KERNEL, __VA_ARGS__)                                  \
        case 13: DISPATCH_ONE(64, 20, KERNEL, __VA_ARGS__)                                  \
        case 14: DISPATCH_ONE(64, 24, KERNEL, __VA_ARGS__)                                  \
        default: DISPATCH_ONE(64, 32, KERNEL, __VA_ARGS__)                                  \
        }                                                                                   \
    } while (0)

}   // namespace
