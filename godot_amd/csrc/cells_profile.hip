// Likelihood with the background level free (gfx950): the joint (alpha, beta)
// maximum and the profile-likelihood grid of the reference's
// get_max(profile_background=True) / _get_logpdf_profile.
//
// These paths run hundreds of dependent passes per cell (every profile point
// is itself a fixed-schedule maximisation over beta), so they are organised
// for small code rather than register residency: a team of T lanes streams
// the cell's weights from L1/L2 in every pass and the pass bodies are
// out-of-line functions shared by all call sites.
#include "cells_common.hpp"

namespace {

// logL(alpha, beta) = sum log(alpha w + beta (1 - w)) - S alpha - B beta
// (log_full_likelihood, core.py:427-438, in one-based parameters)
template <int T>
__device__ __noinline__ double full_loglike(const StreamCell<T> &cd, double S, double B, double al, double be)
{
    double ls, unused;
    cd.logsum([=](double w, double &t, double &g) {
        t = fma(al, w, be * (1.0 - w));
        g = 0.0;
    }, ls, unused);
    return ls - S * al - B * be;
}

// log_profile_likelihood(alpha) (core.py:391-406): beta = get_beta_max(alpha)
template <int T>
__device__ __noinline__ double profile_loglike(const StreamCell<T> &cd, double S, double B, double alpha,
                                               bool &failed)
{
    int fl = 0;
    double bt = fit_beta(cd, B, alpha, 1.0, fl);
    if (fl & GODOT_F_FAILED) {
        failed = true;
        return qnan();
    }
    double b = bt - 1, a = alpha - 1;
    double ls, unused;
    cd.logsum([=](double w, double &t, double &g) {
        t = fma(w, a - b, 1.0 + b);
        g = 0.0;
    }, ls, unused);
    return ls - S * (1 + a) - B * (1 + b);
}

// Joint maximum over alpha >= 0, beta >= 0 by projected Newton steps with
// backtracking.  logL is concave in (alpha, beta), so this converges to the
// maximum the reference's bounded truncated-Newton search (scipy fmin_tnc,
// ftol 1e-3; core.py:552-630) approaches; results agree with it to the
// optimiser's own tolerance, not bit for bit.
template <int T>
__device__ __noinline__ void fit_joint(const StreamCell<T> &cd, double S, double B, double a0, double b0,
                                       double &alpha, double &beta, double &logl, int &flags)
{
    if (cd.n == 0) {
        alpha = beta = 0.0;
        logl = 0.0;
        return;
    }
    double al = a0 > 0 ? a0 : 0.0, be = b0 > 0 ? b0 : 0.0;
    if (al == 0 && be == 0) al = be = 0.5;
    double L = full_loglike(cd, S, B, al, be);
    int it;
    for (it = 0; it < 100; ++it) {
        double acc[5];
        cd.template sums<5>([=](double w, double (&s)[5]) {
            double iw = 1.0 - w;
            double r = rcp_fast(fma(al, w, be * iw));
            double u = w * r, v = iw * r;
            s[0] += u;
            s[1] += v;
            s[2] += u * u;
            s[3] += u * v;
            s[4] += v * v;
        }, acc);
        const double ga = acc[0] - S, gb = acc[1] - B;
        bool fa = !(al <= 0 && ga <= 0), fb = !(be <= 0 && gb <= 0);
        if (!fa && !fb) break;
        double da = 0.0, db = 0.0;
        if (fa && fb) {
            double det = acc[2] * acc[4] - acc[3] * acc[3];
            if (det > 1e-14 * acc[2] * acc[4]) {
                da = (acc[4] * ga - acc[3] * gb) / det;
                db = (acc[2] * gb - acc[3] * ga) / det;
            } else {   // (nearly) identical weights: only a combination is determined
                da = ga / acc[2];
                db = gb / acc[4];
            }
            // a coupled step that pushes a variable sitting on its bound outwards: hold it
            if (al <= 0 && da < 0) {
                da = 0.0;
                db = gb / acc[4];
            }
            if (be <= 0 && db < 0) {
                db = 0.0;
                da = ga / acc[2];
            }
        } else if (fa) {
            da = ga / acc[2];
        } else {
            db = gb / acc[4];
        }
        double x = al, y = be, Ln = L;
        bool moved = false;
        // Two tries: the Newton step, then the same gradient scaled by the diagonal of the
        // curvature alone.  A coupled step whose one component is cut off at its bound during
        // the search (a variable a hair above zero is not "on" the bound) can point downhill
        // in the other; the diagonal step keeps the sign of each gradient component, which
        // the projection onto the quadrant cannot turn around.
        for (int attempt = 0; attempt < 2 && !moved; ++attempt) {
            if (attempt == 1) {
                if (!(fa && fb)) break;             // the step was a diagonal one already
                da = ga / acc[2];
                db = gb / acc[4];
            }
            double s = 1.0;
            for (int h = 0; h < 40; ++h, s *= 0.5) {
                x = fmax(0.0, al + s * da);
                y = fmax(0.0, be + s * db);
                if (x == 0 && y == 0) continue;
                if (x == al && y == be) break;      // the step no longer changes the point
                Ln = full_loglike(cd, S, B, x, y);
                if (Ln >= L) {
                    moved = true;
                    break;
                }
            }
        }
        if (!moved) break;
        bool small = fabs(x - al) <= 1e-13 * (1 + al) && fabs(y - be) <= 1e-13 * (1 + be);
        al = x;
        be = y;
        L = Ln;
        if (small) break;
    }
    if (it >= 100) {
        // No convergence: the curvature is (nearly) singular -- one photon, or equal weights --
        // and the maximum sits on an edge, where Newton steps along the flat direction crawl.
        // logL is concave, so an edge point that meets the Kuhn-Tucker conditions IS the
        // maximum, and on an edge it has a closed form: beta = 0 gives alpha = n / S, alpha = 0
        // gives beta = n / B.
        double e[2];
        cd.template sums<2>([](double w, double (&s)[2]) {
            const double iw = 1.0 - w;
            s[0] += iw / w;
            s[1] += w / iw;
        }, e);
        const double a_edge = (double)cd.n / S, b_edge = (double)cd.n / B;
        if (e[0] / a_edge - B <= 0) {            // d logL / d beta at (n / S, 0)
            al = a_edge;
            be = 0.0;
            L = full_loglike(cd, S, B, al, be);
        } else if (e[1] / b_edge - S <= 0) {     // d logL / d alpha at (0, n / B)
            al = 0.0;
            be = b_edge;
            L = full_loglike(cd, S, B, al, be);
        } else {
            flags |= GODOT_F_FAILED;
        }
    }
    alpha = al;
    beta = be;
    logl = L;
}

// scipy.optimize.brentq (Zeros/brentq.c), default tolerances unless given
template <class F>
__device__ __forceinline__ double brentq_dev(F f, double xa, double xb, double xtol, bool &failed)
{
    const double rtol = 8.881784197001252e-16;
    double xpre = xa, xcur = xb, xblk = 0, fblk = 0, spre = 0, scur = 0;
    double fpre = f(xpre);
    double fcur = f(xcur);
    if (fpre == 0) return xpre;
    if (fcur == 0) return xcur;
    if (signbit(fpre) == signbit(fcur) || isnan(fpre) || isnan(fcur)) {
        failed = true;
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
        if (fcur == 0 || fabs(sbis) < delta) return xcur;
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
        fcur = f(xcur);
        if (isnan(fcur)) {
            failed = true;
            return qnan();
        }
    }
    failed = true;
    return qnan();
}

template <int T>
__global__ __launch_bounds__(256) void cell_fit_joint_kernel(CellArgs p, const double *guess_a, double ga0,
                                                             const double *guess_b, double gb0,
                                                             double *alpha, double *beta, double *logl,
                                                             int32_t *flags)
{
    int64_t cell, lo, n;
    int lane;
    double S, B;
    bool bad;
    if (!cell_setup<T>(p, cell, lane, lo, n, S, B, bad)) return;
    StreamCell<T> cd;
    stream_load(cd, p.we + lo, n, lane, p.swap != 0);
    int fl = 0;
    double a, b, L;
    if (bad) {
        fl = GODOT_F_FAILED;
        a = b = L = qnan();
    } else {
        fit_joint(cd, S, B, guess_a ? guess_a[cell] : ga0, guess_b ? guess_b[cell] : gb0, a, b, L, fl);
    }
    if (lane == 0) {
        alpha[cell] = a;
        beta[cell] = b;
        if (logl) logl[cell] = L;
        if (flags) flags[cell] = fl;
    }
}

// _get_logpdf_profile (core.py:934-974)
template <int T>
__global__ __launch_bounds__(256) void cell_logpdf_profile_kernel(CellArgs p, const double *aopt_in,
                                                                  double dlogl, int npt, double *dom,
                                                                  double *cod, int32_t *flags)
{
    int64_t cell, lo, n;
    int lane;
    double S, B;
    bool bad;
    if (!cell_setup<T>(p, cell, lane, lo, n, S, B, bad)) return;
    StreamCell<T> cd;
    stream_load(cd, p.we + lo, n, lane, p.swap != 0);
    int fl = 0;
    bool failed = bad;
    double aopt = 0, bopt = 0, llmax = 0, amin = 0, amax = 0;
    if (!failed) {
        if (aopt_in) {
            aopt = aopt_in[cell];
            bopt = fit_beta(cd, B, aopt, 1.0, fl);
            if (fl & GODOT_F_FAILED) failed = true;
            if (!failed) llmax = full_loglike(cd, S, B, aopt, bopt);
        } else {
            fit_joint(cd, S, B, 1.0, 1.0, aopt, bopt, llmax, fl);
            if (fl & GODOT_F_FAILED) failed = true;
        }
    }
    if (!failed) {
        auto f = [&](double a) { return profile_loglike(cd, S, B, a, failed) - llmax + dlogl; };
        double a0 = aopt;
        amax = pymax(5.0, a0);
        for (int i = 0; i < 4; ++i) {   // make sure the upper range contains the root
            if (f(amax) > 0) {
                a0 = amax;
                amax *= amax;
            }
        }
        if (!failed) amax = brentq_dev(f, a0, amax, 2e-12, failed);
        if (!failed && aopt > 0) {
            if (f(0.0) > 0) amin = 0;
            else amin = brentq_dev(f, 0.0, aopt, 2e-12, failed);
        }
    }
    double *d = dom + cell * npt, *c = cod + cell * npt;
    if (failed) {
        fl |= GODOT_F_FAILED;
        for (int i = lane; i < npt; i += T) d[i] = c[i] = qnan();
    } else {
        const double step = (amax - amin) / (double)(npt - 1);
        for (int i = 0; i < npt; ++i) {
            double a;
            {
#pragma clang fp contract(off)
                double prod = (double)i * step;
                a = (i == npt - 1) ? amax : prod + amin;
            }
            double v = profile_loglike(cd, S, B, a, failed) - llmax;
            if (lane == 0) {
                d[i] = a;
                c[i] = v;
            }
        }
        if (failed) fl |= GODOT_F_FAILED;
    }
    if (flags && lane == 0) flags[cell] = fl;
}

inline int pick_stream_team(int64_t nph, int64_t ncell)
{
    double mean = ncell > 0 ? (double)godot_shape_photons(nph) / (double)ncell : 0.0;
    if (mean <= 40) return 4;
    if (mean <= 400) return 16;
    return 64;
}

#define DISPATCH_STREAM(KERNEL, ...)                                                              \
    do {                                                                                          \
        switch (pick_stream_team(nph, ncell)) {                                                   \
        case 4:                                                                                   \
            hipLaunchKernelGGL((KERNEL<4>), dim3((unsigned)ceil_div(ncell, 64)), dim3(256), 0, st, \
                               __VA_ARGS__);                                                      \
            break;                                                                                \
        case 16:                                                                                  \
            hipLaunchKernelGGL((KERNEL<16>), dim3((unsigned)ceil_div(ncell, 16)), dim3(256), 0, st, \
                               __VA_ARGS__);                                                      \
            break;                                                                                \
        default:                                                                                  \
            hipLaunchKernelGGL((KERNEL<64>), dim3((unsigned)ceil_div(ncell, 4)), dim3(256), 0, st, \
                               __VA_ARGS__);                                                      \
            break;                                                                                \
        }                                                                                         \
    } while (0)

}   // namespace

extern "C" int godot_cell_fit_joint(const double *we, int64_t nph, const int64_t *lo, const int64_t *hi,
                                    const double *S, const double *B, int64_t ncell, const double *guess_a,
                                    double guess_a0, const double *guess_b, double guess_b0, int swap,
                                    double *alpha, double *beta, double *logl, int32_t *flags,
                                    godot_stream_t stream)
{
    GODOT_ENTRY();
    int rc = check_cells(we, nph, lo, hi, ncell, "godot_cell_fit_joint");
    if (rc) return rc;
    if (ncell == 0) return GODOT_OK;
    GODOT_REQUIRE(S && B && alpha && beta, GODOT_E_INVALID, "godot_cell_fit_joint: null argument");
    hipStream_t st = (hipStream_t)stream;
    CellArgs p{we, nph, lo, hi, S, B, ncell, swap};
    DISPATCH_STREAM(cell_fit_joint_kernel, p, guess_a, guess_a0, guess_b, guess_b0, alpha, beta, logl, flags);
    GODOT_LAUNCH_CHECK("godot_cell_fit_joint");
    return GODOT_OK;
}

extern "C" int godot_cell_logpdf_profile(const double *we, int64_t nph, const int64_t *lo, const int64_t *hi,
                                         const double *S, const double *B, int64_t ncell,
                                         const double *aopt, double dlogl, int npt, int swap, double *dom,
                                         double *cod, int32_t *flags, godot_stream_t stream)
{
    GODOT_ENTRY();
    int rc = check_cells(we, nph, lo, hi, ncell, "godot_cell_logpdf_profile");
    if (rc) return rc;
    GODOT_REQUIRE(npt >= 2 && npt <= 1024, GODOT_E_UNSUPPORTED,
                  "godot_cell_logpdf_profile: npt must be in [2, 1024]");
    if (ncell == 0) return GODOT_OK;
    GODOT_REQUIRE(S && B && dom && cod, GODOT_E_INVALID, "godot_cell_logpdf_profile: null argument");
    hipStream_t st = (hipStream_t)stream;
    CellArgs p{we, nph, lo, hi, S, B, ncell, swap};
    DISPATCH_STREAM(cell_logpdf_profile_kernel, p, aopt, dlogl, npt, dom, cod, flags);
    GODOT_LAUNCH_CHECK("godot_cell_logpdf_profile");
    return GODOT_OK;
}
