"""CPU oracle for the weighted-photon likelihood / periodogram hot path.

TEST INFRASTRUCTURE ONLY.  This module is a numpy restatement of the
reference algorithm (kerrm/godot ``core.py``) used as the checker in
``tests/``, in ``__graft_entry__.smoke()`` and as the ``cpu_baseline`` leg of
``bench.py``.  Nothing under ``godot_amd/`` may import it: the product path
is the HIP library and fails loudly without it.

Parity status: PINNED.  Every function below that restates a reference
function is checked against outputs of the reference itself (imported in the
build container by ``tools/make_golden.py``; the vectors live in
``tests/golden/*.npz``).  ``z2m`` / ``h_test`` have no counterpart in the
reference (it holds no Z^2_m or H-test function) and are "parity unpinned":
they are pinned only to the textbook definition evaluated in extended
precision.

Each function cites the reference lines it follows (paths relative to the
reference checkout).  Arithmetic is written so that the floating-point
operation order equals the reference's (same numpy ufunc sequence, pairwise
``np.sum``), which makes most results bit-identical to the reference's.
"""
import math

import numpy as np
from scipy.integrate import cumulative_trapezoid, simpson

MJDREF = 51910 + 7.428703703703703e-4


def met2mjd(times):
    """core.py:24-26 (extended precision on the host)."""
    return np.asarray(times, dtype=np.longdouble) * (1. / 86400) + MJDREF


# ---------------------------------------------------------------------------
# scipy.optimize.brentq restated (third-party: scipy 1.15.3,
# scipy/optimize/Zeros/brentq.c; the reference calls it at core.py:739, 849,
# 959, 964).  Pinned against scipy.optimize.brentq in tests/test_oracle.py.
# ---------------------------------------------------------------------------
def brentq(f, xa, xb, xtol=2e-12, rtol=8.881784197001252e-16, maxiter=100):
    xpre, xcur = float(xa), float(xb)
    xblk = fblk = spre = scur = 0.0
    fpre = f(xpre)
    fcur = f(xcur)
    if fpre == 0:
        return xpre
    if fcur == 0:
        return xcur
    if math.copysign(1.0, fpre) == math.copysign(1.0, fcur):
        raise ValueError('f(a) and f(b) must have different signs')
    for _ in range(maxiter):
        if fpre != 0 and fcur != 0 and \
                math.copysign(1.0, fpre) != math.copysign(1.0, fcur):
            xblk, fblk = xpre, fpre
            spre = scur = xcur - xpre
        if abs(fblk) < abs(fcur):
            xpre, xcur, xblk = xcur, xblk, xcur
            fpre, fcur, fblk = fcur, fblk, fcur
        delta = (xtol + rtol * abs(xcur)) / 2
        sbis = (xblk - xcur) / 2
        if fcur == 0 or abs(sbis) < delta:
            return xcur
        if abs(spre) > delta and abs(fcur) < abs(fpre):
            if xpre == xblk:
                stry = -fcur * (xcur - xpre) / (fcur - fpre)
            else:
                dpre = (fpre - fcur) / (xpre - xcur)
                dblk = (fblk - fcur) / (xblk - xcur)
                stry = -fcur * (fblk * dblk - fpre * dpre) / \
                    (dblk * dpre * (fblk - fpre))
            if 2 * abs(stry) < min(abs(spre), 3 * abs(sbis) - delta):
                spre, scur = scur, stry
            else:
                spre = scur = sbis
        else:
            spre = scur = sbis
        xpre, fpre = xcur, fcur
        if abs(scur) > delta:
            xcur += scur
        else:
            xcur += delta if sbis > 0 else -delta
        fcur = f(xcur)
    raise RuntimeError('brentq failed to converge')


# ---------------------------------------------------------------------------
# a3: per-cell likelihood set-up and evaluations
# ---------------------------------------------------------------------------
class CellLike(object):
    """Per-cell likelihood state: core.py:331-351 (CellLogLikelihood.__init__).

    ``we`` are photon weights of one cell, ``exp`` the cell's source-scaled
    exposure, ``SonB`` its source-to-background exposure ratio; ``a``/``b``
    are the baked-in weight scales (Cell._alpha / Cell._beta, core.py:176-177).
    """

    def __init__(self, we, exp, SonB, a=1, b=1, swap=False):
        w = np.asarray(we, dtype=np.float64)
        self.we = w * a / (w * a + b * (1 - w))
        self.iwe = 1 - self.we
        self.S = exp * a
        self.B = exp / SonB * b
        if swap:
            self.we, self.iwe = self.iwe, self.we
            self.S, self.B = self.B, self.S
        self.depth_max = 0      # deepest get_max recursion seen (diagnostic)
        self.used_brentq = False

    # core.py:353-364
    def log_likelihood(self, alpha):
        t = self.we * alpha
        t = t + self.iwe
        return np.sum(np.log(t)) - (self.S * alpha)

    # core.py:427-438
    def log_full_likelihood(self, alpha, beta):
        a = alpha - 1
        b = beta - 1
        t = (a - b) * self.we
        t = (1 + b) + t
        return np.sum(np.log(t)) - self.S * (1 + a) - self.B * (1 + b)

    # core.py:440-460 (value and gradient of -logL(alpha, beta))
    def neg_logl_grad(self, alpha, beta):
        a = alpha - 1
        b = beta - 1
        if (a == -1) and (b == -1):
            return np.inf, [np.inf, np.inf]
        t1 = self.we * (a - b)
        t1 = t1 + (1 + b)
        logl = np.sum(np.log(t1)) - self.S * a - self.B * b
        ga = np.sum(self.we / t1) - self.S
        gb = np.sum(self.iwe / t1) - self.B
        return -logl, [-ga, -gb]

    # ------------------------------------------------------------------
    # a4: fixed-schedule Newton/Halley maximiser, core.py:632-691
    # ------------------------------------------------------------------
    def get_max(self, guess=1, beta=1, depth=0):
        w, iw, S = self.we, self.iwe, self.S
        g = guess - 1
        b = beta - 1
        self.depth_max = max(self.depth_max, depth)

        t2 = 1 + b * iw
        t1 = w / (t2 + (-1) * w)
        if (np.sum(t1) - S) < 0:          # core.py:639-644
            return 0
        a = g

        t1 = w / (t2 + a * w)             # core.py:649-655
        f1 = np.sum(t1) - S
        t1 = t1 * t1
        f2 = np.sum(t1)
        a = a + f1 / f2
        if a < 0 - 1:
            a = 0.2 - 1

        t1 = w / (t2 + a * w)             # core.py:658-664
        f1 = np.sum(t1) - S
        t1 = t1 * t1
        f2 = np.sum(t1)
        a = a + f1 / f2
        if a < 0.05 - 1:
            a = 0.05 - 1

        t1 = w / (t2 + a * w)             # core.py:667-671
        f1 = np.sum(t1) - S
        t1 = t1 * t1
        f2 = np.sum(t1)
        alast = a = max(0 - 1, a + f1 / f2)

        t1 = w / (t2 + a * w)             # core.py:674-680 (Halley)
        f1 = np.sum(t1) - S
        t1 = t1 * t1
        f2 = np.sum(t1)
        t1 = t1 * (w / (t2 + a * w))
        f3 = 2 * np.sum(t1)
        a = max(0 - 1, a + 2 * f1 * f2 / (2 * f2 * f2 - f1 * f3))

        if (abs(a - alast) > 0.05) or (abs(g - a) > 10):   # core.py:684-689
            if depth > 2:
                return self.get_max_numerical()
            return self.get_max(guess=a + 1, beta=b + 1, depth=depth + 1)
        return a + 1

    # core.py:693-739 (non-profile branch); note the reference calls it with
    # its defaults guess=1, beta=1 whatever the caller's beta was (core.py:686)
    def get_max_numerical(self, guess=1, beta=1):
        w, iw, S = self.we, self.iwe, self.S
        self.used_brentq = True
        g = guess - 1
        b = beta - 1
        t2 = 1 + b * iw
        if (np.sum(w / (t2 + (-1) * w)) - S) < 0:
            return 0

        def f(a):
            return np.sum(w / (t2 + a * w)) - S

        a0 = -1
        amax = g
        for _ in range(12):
            if f(amax) < 0:
                break
            a0 = amax
            amax = 2 * amax + 1
        return brentq(f, a0, amax, xtol=1e-3) + 1

    # ------------------------------------------------------------------
    # a5: TS, core.py:982-994 (non-profile)
    # ------------------------------------------------------------------
    def get_ts(self, aopt):
        if self.S == 0:
            return 0
        if aopt == 0:
            return 0
        return 2 * (self.log_likelihood(aopt) - self.log_likelihood(0))

    def get_ts_profile(self, aopt):
        if self.S == 0 or aopt == 0:
            return 0
        return 2 * (self.log_profile_likelihood(aopt)
                    - self.log_profile_likelihood(0))

    # ------------------------------------------------------------------
    # a6: adaptive-support profile grid, core.py:851-932 (non-profile)
    # ------------------------------------------------------------------
    def get_logpdf(self, aopt=None, dlogl=20, npt=100, include_zero=False,
                   depth=0):
        if depth > 50:
            raise RuntimeError('get_logpdf re-centring did not terminate')
        if aopt is None:
            aopt = self.get_max()
        we, iw, S = self.we, self.iwe, self.S
        amin = 0

        if aopt == 0:                      # core.py:869-880
            llmax = np.log(iw).sum()
            amax = max(0, -(llmax + dlogl) / (np.sum(we / (1 - we)) - S))
            for _ in range(10):
                t = amax * we + iw
                f0 = np.log(t).sum() - amax * S + dlogl - llmax
                f1 = np.sum(we / t) - S
                amax = amax - f0 / f1
                if abs(f0) < 0.1:
                    break
        else:                              # core.py:881-895
            t = aopt * we + iw
            llmax = np.sum(np.log(t)) - S * aopt
            f2 = np.abs(np.sum((we / t) ** 2))
            amax = aopt + np.sqrt(2 * dlogl / f2)
            for _ in range(5):
                t = amax * we + iw
                f0 = np.log(t).sum() - amax * S + dlogl - llmax
                f1 = np.sum(we / t) - S
                amax = amax - f0 / f1
                if abs(f0) < 0.1:
                    break

        if (not include_zero) and (aopt > 0):   # core.py:897-914
            t = aopt * we + iw
            f2 = np.abs(np.sum((we / t) ** 2))
            min_a = np.max(1 - 1. / we) + 1e-6
            amin = max(min_a, aopt - np.sqrt(2 * dlogl / f2))
            for _ in range(5):
                t = amin * we + iw
                f0 = np.log(t).sum() - amin * S + dlogl - llmax
                f1 = np.sum(we / t) - S
                amin = max(min_a, amin - f0 / f1)
                if abs(f0) < 0.1:
                    break

        amin = max(0, amin)                # core.py:916

        dom = np.linspace(amin, amax, npt)
        cod = np.empty_like(dom)
        for i, a in enumerate(dom):
            cod[i] = self.log_likelihood(a)

        k = np.argmax(cod)                 # core.py:923-929
        if abs(cod[k] - llmax) > 0.05:
            return self.get_logpdf(aopt=dom[k], dlogl=dlogl, npt=npt,
                                   include_zero=include_zero, depth=depth + 1)
        cod -= llmax
        return dom, cod

    # core.py:976-980
    def get_pdf(self, aopt=None, dlogl=20, npt=100):
        dom, cod = self.get_logpdf(aopt=aopt, dlogl=dlogl, npt=npt)
        np.exp(cod, out=cod)
        return dom, cod * (1. / simpson(cod, x=dom))

    # ------------------------------------------------------------------
    # a7: background profile, core.py:741-823, 391-406
    # ------------------------------------------------------------------
    def get_beta_max(self, alpha, guess=1, depth=0):
        if np.isnan(alpha):
            return 1
        if len(self.we) == 0:
            return 0
        if alpha == 0:
            return len(self.we) / self.B
        al = alpha - 1
        g = guess - 1
        B, w, iw = self.B, self.we, self.iwe

        if np.sum(iw / w) < B * (1 + al):      # core.py:758-760
            return 0
        b = g

        t2 = 1 + al * w
        lows = (0.2 - 1, 0.05 - 1, 0.02 - 1)
        for lo in lows:                         # core.py:765-788 (3 Newton)
            t = iw / (t2 + b * iw)
            f1 = np.sum(t) - B
            t = t * t
            f2 = np.sum(t)
            b = b + f1 / f2
            b = max(lo, b)

        t = iw / (t2 + b * iw)                  # core.py:793-803 (Halley 1)
        f1 = np.sum(t) - B
        t = t * t
        f2 = np.sum(t)
        t = t * (iw / (t2 + b * iw))
        f3 = 2 * np.sum(t)
        newb = max(0 - 1, b + 2 * f1 * f2 / (2 * f2 * f2 - f1 * f3))
        if abs(newb - b) > 10:
            blast = b = 2 * b + 1
        else:
            blast = b = newb

        t = iw / (t2 + b * iw)                  # core.py:806-812 (Halley 2)
        f1 = np.sum(t) - B
        t = t * t
        f2 = np.sum(t)
        t = t * (iw / (t2 + b * iw))
        f3 = 2 * np.sum(t)
        b = max(0 - 1, b + 2 * f1 * f2 / (2 * f2 * f2 - f1 * f3))

        if (abs(b - blast) > 0.05) or (abs(g - b) > 10) or (b == -1):
            if depth > 2:                       # core.py:816-821
                raise ValueError('Did not converge for alpha=%.5f!' % (al + 1))
            return self.get_beta_max(al + 1, guess=b + 1, depth=depth + 1)
        return b + 1

    def log_profile_likelihood(self, alpha, beta_guess=1):
        b = self.get_beta_max(alpha, guess=beta_guess) - 1
        a = alpha - 1
        t = self.we * (a - b)
        t = (1. + b) + t
        return np.sum(np.log(t)) - self.S * (1 + a) - self.B * (1 + b)


    # ------------------------------------------------------------------
    # a7: joint (alpha, beta) maximum, core.py:552-630.  The optimiser is
    # third-party (scipy.optimize.fmin_tnc, ftol=1e-3, maxfun=200; the
    # reference pins no scipy version -- 1.15.3 is installed here): its
    # result is defined only to its own tolerance, so parity for this entry
    # is stated on logL (>= reference - 2e-3), not on (alpha, beta).
    # ------------------------------------------------------------------
    def get_max_profile(self, guess=1, beta=1):
        from scipy.optimize import fmin_tnc
        fn = lambda q: self.neg_logl_grad(q[0], q[1])
        kw = dict(bounds=[[0, None], [0, None]], disp=0, ftol=1e-3, maxfun=200)
        ok = lambda rc: not ((rc < 0) or (rc > 2))
        rvals, _, rc = fmin_tnc(fn, [float(guess), float(beta)], **kw)
        if ok(rc):
            return rvals
        guess, beta = rvals[0] + 1e-3, rvals[1] - 1e-3     # core.py:565-568
        rvals, _, rc = fmin_tnc(fn, [guess, beta], **kw)
        if ok(rc):
            return rvals
        oldrvals, oldlogl = rvals, fn(rvals)[0]
        if rc == 3:                                          # core.py:585-587
            newguess, newbeta = guess, beta
        else:                                                # core.py:589-607
            grid = np.asarray([0, 0.1, 0.3, 0.5, 1.0, 1.5, 2.0, 5.0, 10.0])
            cogrid = np.asarray([self.log_profile_likelihood(x) for x in grid])
            k = np.argmax(cogrid)
            lo, hi = max(0, k - 1), min(k + 1, len(grid) - 1)
            if abs(cogrid[hi] - cogrid[lo]) > 100:
                grid = np.linspace(grid[lo], grid[hi], 20)
                cogrid = np.asarray([self.log_profile_likelihood(x) for x in grid])
            newguess = float(grid[np.argmax(cogrid)])
            newbeta = float(max(0.1, self.get_beta_max(newguess)))
        rvals, _, rc = fmin_tnc(fn, [newguess, newbeta], **kw)
        if ok(rc) and fn(rvals)[0] > oldlogl:                # core.py:616-620
            return oldrvals
        return rvals

    # _get_logpdf_profile, core.py:934-974
    def get_logpdf_profile(self, aopt=None, dlogl=20, npt=100):
        if aopt is None:
            aopt, bopt = self.get_max_profile()
        else:
            bopt = self.get_beta_max(aopt)
        llmax = self.log_full_likelihood(aopt, bopt)
        f = lambda a: self.log_profile_likelihood(a) - llmax + dlogl
        a0 = aopt
        amin = 0
        amax = max(5, a0)
        for _ in range(4):
            if f(amax) > 0:
                a0 = amax
                amax *= amax
        amax = brentq(f, a0, amax)
        if aopt > 0:
            if f(0) > 0:
                amin = 0
            else:
                amin = brentq(f, 0, aopt)
        dom = np.linspace(amin, amax, npt)
        cod = np.empty_like(dom)
        for i, a in enumerate(dom):
            cod[i] = self.log_profile_likelihood(a)
        cod -= llmax
        return dom, cod


# ---------------------------------------------------------------------------
# a6 (batch): CellsLogLikelihood.__init__/get_flux/get_raw_lightcurve,
# core.py:1070-1098, 1227-1257, 1295-1336 (profile_background=False)
# ---------------------------------------------------------------------------
def flux_quantiles(dom, cod, conf):
    """core.py:1247-1256: pdf from the cached log profile, quantiles by
    trapezoid cdf + linear interpolation."""
    pdf = np.exp(cod)
    pdf *= (1. / simpson(pdf, x=dom))
    cdf = cumulative_trapezoid(pdf, dom, initial=0)
    cdf *= 1. / cdf[-1]
    idx = np.searchsorted(cdf, conf)
    ihi, ilo = idx, idx - 1
    m = (cdf[ihi] - cdf[ilo]) / (dom[ihi] - dom[ilo])
    return dom[ilo] + (np.asarray(conf) - cdf[ilo]) / m


def cells_logpdf(clls, npt=200, dlogl=30):
    """core.py:1087-1095: the (ncell, npt) domain / codomain cache."""
    dom = np.empty([len(clls), npt])
    cod = np.empty([len(clls), npt])
    for i, c in enumerate(clls):
        dom[i], cod[i] = c.get_logpdf(npt=npt, dlogl=dlogl)
    return dom, cod


def cell_flux(cll, dom, cod, conf):
    """core.py:1227-1257 for one cell with background fixed."""
    aguess = dom[np.argmax(cod)]
    aopt = cll.get_max(guess=aguess, beta=1)
    ts = cll.get_ts(aopt)
    return aopt, ts, flux_quantiles(dom, cod, conf)


def raw_lightcurve(clls, tstarts, tstops, dom, cod, tsmin=4):
    """core.py:1295-1336; rows [MJD mid, half width (d), alpha|UL, lo,
    hi|-1, TS]."""
    out = np.empty([len(clls), 6])
    for i, c in enumerate(clls):
        tmid = 0.5 * (tstarts[i] + tstops[i])
        terr = (tmid - tstarts[i]) / 86400
        tmjd = met2mjd(tmid)
        aopt, ts, xc = cell_flux(c, dom[i], cod[i], [0.159, 0.841])
        if ts <= tsmin:
            out[i] = tmjd, terr, xc[1], 0, -1, ts
        else:
            out[i] = tmjd, terr, aopt, aopt - xc[0], xc[1] - aopt, ts
    return out


# ---------------------------------------------------------------------------
# a1: Data.get_exposure, core.py:1792-1843
# ---------------------------------------------------------------------------
def exposure_at(TSTART, TSTOP, exposure, LIVETIME, S, B, E, times,
                deadtime_too=False, cumulative=True):
    times = np.asarray(times, dtype=np.float64)
    idx = np.searchsorted(TSTOP, times)
    frac = (times - TSTART[idx]) / (TSTOP[idx] - TSTART[idx])
    np.clip(frac, 0, 1, out=frac)
    frac = 1 - frac
    sexposure = (S / E) * exposure
    bexposure = (B / E) * exposure
    cexp = np.cumsum(exposure)[idx] - exposure[idx] * frac
    csexp = np.cumsum(sexposure)[idx] - sexposure[idx] * frac
    cbexp = np.cumsum(bexposure)[idx] - bexposure[idx] * frac
    cdead = None
    if deadtime_too:
        dead = (TSTOP - TSTART) - LIVETIME
        cdead = np.cumsum(dead)[idx] - dead[idx] * frac
    if not cumulative:
        cexp = cexp[1:] - cexp[:-1]
        csexp = csexp[1:] - csexp[:-1]
        cbexp = cbexp[1:] - cbexp[:-1]
        if cdead is not None:
            cdead = cdead[1:] - cdead[:-1]
    return cexp, csexp, cbexp, cdead


# ---------------------------------------------------------------------------
# a2: array tail of Data.get_cells, core.py:2063-2206 (use_barycenter=False,
# no randomize / scalers / injector)
# ---------------------------------------------------------------------------
def bin_cells(ti, we, TSTART, TSTOP, exposure, LIVETIME, S, B, E,
              tstart=None, tstop=None, tcell=86400., trim_zero_exposure=True,
              minimum_fractional_exposure=0):
    """Returns a dict with the per-cell arrays and the CSR layout of the
    photons: ``offsets[i]:offsets[i+1]`` indexes ``times``/``weights`` of
    cell i, exactly the slices the reference's loop takes (core.py:2197-2205).
    """
    if tstart is None:
        tstart = TSTART[0]
    tstart = max(tstart, TSTART[0])       # core.py:2047-2050
    if tstop is None:
        tstop = TSTOP[-1]
    tstop = min(tstop, TSTOP[-1])         # core.py:2055-2058

    ncell = int((tstop - tstart) / tcell)  # core.py:2066-2071
    if ncell == 0:
        tcell = tstop - tstart
        ncell = 1
    edges = tstart + np.arange(ncell + 1) * tcell
    assert edges[-1] <= tstop

    exp, sexp, bexp, dead = exposure_at(
        TSTART, TSTOP, exposure, LIVETIME, S, B, E, edges,
        deadtime_too=True, cumulative=False)
    starts, stops = edges[:-1], edges[1:]

    i0, i1 = np.searchsorted(ti, [starts[0], stops[-1]])   # core.py:2097
    times = ti[i0:i1]
    weights = we[i0:i1]

    if minimum_fractional_exposure > 0:    # core.py:2101-2108
        frac_exp = exp / exp[exp > 0].mean()
        m = frac_exp < minimum_fractional_exposure
        exp[m] = 0
        sexp[m] = 0
        bexp[m] = 0

    if trim_zero_exposure:                 # core.py:2110-2122
        keep = exp > 0
        ev = keep[np.searchsorted(stops, times)]
        times = times[ev]
        weights = weights[ev]
        starts = starts[keep]
        stops = stops[keep]
        exp, sexp, bexp, dead = exp[keep], sexp[keep], bexp[keep], dead[keep]

    event_idx = np.searchsorted(stops, times)              # core.py:2159
    counts = np.bincount(event_idx, minlength=len(starts))  # core.py:2161
    offsets = np.append(0, np.cumsum(counts))               # core.py:2178

    # core.py:2180-2183: extended-precision cumulative sums
    W1 = np.append(0, np.cumsum(weights.astype(np.longdouble)))
    W2 = np.append(0, np.cumsum(weights.astype(np.longdouble) ** 2))
    wsum = (W1[offsets[1:]] - W1[offsets[:-1]]).astype(float)
    w2sum = (W2[offsets[1:]] - W2[offsets[:-1]]).astype(float)

    return dict(starts=starts, stops=stops, exp=exp, sexp=sexp, bexp=bexp,
                dead=dead, times=times, weights=weights,
                event_idx=event_idx, counts=counts, offsets=offsets,
                wsum=wsum, w2sum=w2sum)


def cells_from_bins(b):
    """core.py:2197-2206: per-cell likelihood objects; Cell.exp is the
    source-scaled exposure and SonB = sexp/bexp."""
    out = []
    o = b['offsets']
    for i in range(len(b['starts'])):
        out.append(CellLike(b['weights'][o[i]:o[i + 1]], b['sexp'][i],
                            b['sexp'][i] / b['bexp'][i]))
    return out


def zero_low_exposure(ts, minimum_exposure):
    """CellTimeSeries.__init__ / _zero_weight, core.py:241-244, 279-285."""
    m = ~(ts['exp'] > minimum_exposure)
    for k in ('exp', 'sexp', 'bexp', 'counts', 'wsum', 'w2sum'):
        ts[k][m] = 0
    return ts


# ---------------------------------------------------------------------------
# a8: FFT power spectra, core.py:2598-2741 and 2743-2782
# ---------------------------------------------------------------------------
def power_spectrum_fft(starts, stops, counts, W, WW, sexp, bexp, dfgoal=None,
                       tweak_exp=False, exp_only=False, get_amps=False,
                       exposure_correction=None, no_zero_pad=False,
                       maxN=None):
    tsamp = stops[0] - starts[0]
    tspan = stops[-1] - starts[0]
    S = sexp
    if exposure_correction is not None:
        S = S * exposure_correction
    if tweak_exp:
        S = S * W.sum() / S.sum()
    Wb = counts - W
    WbWb = counts - 2 * W + WW
    B = bexp
    if tweak_exp:
        B = B * Wb.sum() / B.sum()
    if maxN is not None:
        W, WW, Wb, WbWb, S, B = [x[:maxN] for x in (W, WW, Wb, WbWb, S, B)]
    if exp_only:
        W = 2 * S - S.mean()

    if dfgoal is None:
        dfgoal = 0.2 / tspan
    nfft = 1. / (tsamp * dfgoal)
    l = 2 ** (int(np.log2(nfft)) + 1)
    if no_zero_pad:
        l = len(W)
    z = np.zeros(l - len(W))
    W, WW, S, Wb, WbWb, B = [np.append(x, z) for x in (W, WW, S, Wb, WbWb, B)]
    freqs = np.fft.rfftfreq(l) / tsamp
    nk = l // 4 + 1

    out = power_spectrum_from_parts(np.fft.rfft(W - S)[:nk], np.fft.rfft(Wb - B)[:nk], np.fft.rfft(WW)[::2],
                                    np.fft.rfft(WbWb)[::2], np.fft.rfft(W - WW)[::2], get_amps=get_amps)
    if get_amps:
        return (freqs[:nk],) + out
    if exp_only:
        return freqs[:nk], out[0]
    return (freqs[:nk],) + out


def power_spectrum_from_parts(f1, f2, g3, g4, g5, get_amps=False):
    """The estimators of power_spectrum_fft (core.py:2685-2741) from its five transforms:
    f1 = rfft(W - S)[:nk], f2 = rfft(Wb - B)[:nk] and the even bins of rfft(WW), rfft(WbWb)
    and rfft(W - WW) (complex arrays of nk entries).  Returns (P0, P1, Pb), or the eight
    amplitude arrays with get_amps."""
    WmS_cos, WmS_sin = np.real(f1), -np.imag(f1)
    WbmB_cos, WbmB_sin = np.real(f2), -np.imag(f2)

    def halves(f):
        s0 = f[0].real
        return 0.5 * (s0 + np.real(f)), 0.5 * (s0 - np.real(f))

    WW_cos, WW_sin = halves(g3)
    WbWb_cos, WbWb_sin = halves(g4)
    WWb_cos, WWb_sin = halves(g5)
    WW_sin[0] = 1.
    WbWb_sin[0] = 1.

    with np.errstate(all='ignore'):
        a_cos0 = WmS_cos / WW_cos
        a_sin0 = WmS_sin / WW_sin
        b_cos0 = WbmB_cos / WbWb_cos
        b_sin0 = WbmB_sin / WbWb_sin
        den_cos = 1. / (WW_cos * WbWb_cos - WWb_cos ** 2)
        den_sin = 1. / (WW_sin * WbWb_sin - WWb_sin ** 2)
        den_sin[0] = 0.
        a_cos = (WbWb_cos * WmS_cos - WWb_cos * WbmB_cos) * den_cos
        a_sin = (WbWb_sin * WmS_sin - WWb_sin * WbmB_sin) * den_sin
        b_cos = (WW_cos * WbmB_cos - WWb_cos * WmS_cos) * den_cos
        b_sin = (WW_sin * WbmB_sin - WWb_sin * WmS_sin) * den_sin
        if get_amps:
            return (a_cos, a_sin, b_cos, b_sin, a_cos0, a_sin0, b_cos0, b_sin0)
        dl_null = b_cos0 * WbmB_cos + b_sin0 * WbmB_sin
        dl_nobg = a_cos0 * WmS_cos + a_sin0 * WmS_sin
        dl = a_cos * WmS_cos + a_sin * WmS_sin + b_cos * WbmB_cos \
            + b_sin * WbmB_sin
    return dl_nobg, dl - dl_null, dl_null


def power_spectrum_simple(starts, stops, W, WW, sexp, get_amps=False):
    tsamp = stops[0] - starts[0]
    l = len(W)
    nk = l // 4 + 1
    f = np.fft.rfft(W - sexp)[:nk]
    c, s = np.real(f), -np.imag(f)
    f = np.fft.rfft(WW)
    s2 = f[0].real
    cc = 0.5 * (s2 + np.real(f)[::2])
    ss = 0.5 * (s2 - np.real(f)[::2])
    ss[0] = 1.
    with np.errstate(all='ignore'):
        dlogl = c ** 2 / cc + s ** 2 / ss
    freqs = np.fft.rfftfreq(l)[:nk] * (1. / tsamp)
    if get_amps:
        return freqs, dlogl, (c, s, cc, ss)
    return freqs, dlogl


# ---------------------------------------------------------------------------
# a9: direct trig sums, core.py:2396-2413 (flat layout), 2426-2488, 2533-2596
# ---------------------------------------------------------------------------
def flatten_cells(weights, offsets, exps, starts, stops):
    """CSR -> the flat photon layout of CellsLogLikelihoodOld."""
    counts = np.diff(offsets)
    alpha_indices = np.repeat(np.arange(len(counts)), counts)
    tmids = 0.5 * (np.asarray(starts) + np.asarray(stops))
    return dict(weights=np.asarray(weights, dtype=np.float64),
                alpha_indices=alpha_indices, exp=np.asarray(exps),
                tmids=tmids, nph=int(counts.sum()))


def fit_harmonics(flat, freq):
    tm = flat['tmids']
    ph = (2 * np.pi * freq) * (tm - tm.mean())
    cph, sph = np.cos(ph), np.sin(ph)
    ai, w, ex = flat['alpha_indices'], flat['weights'], flat['exp']
    tmp = cph[ai] * w
    cw1 = np.sum(tmp)
    cw2 = np.sum(tmp * tmp)
    ce = np.sum(ex * cph)
    tmp = sph[ai] * w
    sw1 = np.sum(tmp)
    sw2 = np.sum(tmp * tmp)
    se = np.sum(ex * sph)
    with np.errstate(all='ignore'):
        return ((cw1 - ce) / cw2, (sw1 - se) / sw2,
                (cw1 - ce) ** 2 / cw2 + (sw1 - se) ** 2 / sw2)


def power_spectrum_dft(flat, freqs, unweighted=False):
    tm = flat['tmids']
    basis = tm - tm[0]
    dt = tm[1] - tm[0]
    ai = flat['alpha_indices']
    if unweighted:
        w = np.ones_like(flat['weights'])
        ex = flat['exp'] * (float(flat['nph']) / np.sum(flat['weights']))
    else:
        w, ex = flat['weights'], flat['exp']
    grid = np.linspace(0, 2 * np.pi, 1001)[:-1]
    ctab, stab = np.cos(grid), np.sin(grid)
    out = np.empty(len(freqs))
    for i, fr in enumerate(freqs):
        ph = basis * fr
        corr = 1 - (dt * 2 * np.pi * fr) ** 2 / 24
        ph = ph - ph.astype(int)             # core.py:2568-2569
        k = (ph * 1000).astype(int)          # core.py:2570 (truncation)
        cph, sph = ctab[k], stab[k]
        tmp = cph[ai] * w
        cw1 = np.sum(tmp)
        cw2 = np.sum(tmp * tmp)
        ce = np.sum(cph * ex * corr)
        tmp = sph[ai] * w
        sw1 = np.sum(tmp)
        sw2 = np.sum(tmp * tmp)
        se = np.sum(sph * ex * corr)
        with np.errstate(all='ignore'):
            out[i] = (cw1 - ce) ** 2 / cw2 + (sw1 - se) ** 2 / sw2
    return out


def compute_ls(flat, counts, freqs, unweighted=False):
    """core.py:2490-2531."""
    if unweighted:
        flux = np.asarray(counts, dtype=float) / flat['exp'] ** 0.5
    else:
        off = np.append(0, np.cumsum(counts))
        flux = np.asarray([np.sum(flat['weights'][off[i]:off[i + 1]])
                           for i in range(len(counts))]) - flat['exp']
    flux = flux - flux.mean()
    flux = flux / flux.std()
    basis = flat['tmids'] - flat['tmids'][0]
    grid = np.linspace(0, 2 * np.pi, 1001)[:-1]
    ctab, stab = np.cos(grid), np.sin(grid)
    out = np.empty(len(freqs))
    for i, fr in enumerate(freqs):
        ph = basis * fr
        ph = ph - ph.astype(int)
        k = (ph * 1000).astype(int)
        out[i] = np.sum(ctab[k] * flux) ** 2 + np.sum(stab[k] * flux) ** 2
    return out / (0.5 * np.sum(flux ** 2))


def get_orbital_modulation(starts, stops, W, WW, sexp, freqs):
    """core.py:2828-2861."""
    ti = (starts + stops) * 0.5
    rvals = np.ones_like(ti)
    y = (W - sexp) / sexp.mean()
    pows = np.empty(len(freqs))
    for i in range(len(freqs)):
        ph = np.mod((ti - ti[0]) * freqs[i], 1) * (2 * np.pi)
        cph, sph = np.cos(ph), np.sin(ph)
        a = np.mean(cph * y)
        b = np.mean(sph * y)
        a2 = np.mean(cph ** 2 * WW)
        b2 = np.mean(sph ** 2 * WW)
        rvals += 2 * (a * cph + b * sph)
        pows[i] = (a ** 2 / a2 + b ** 2 / b2)
    return rvals, pows * sexp.sum() ** 2 / sexp.shape[0]


# ---------------------------------------------------------------------------
# a9 (new API, no reference counterpart -> "parity unpinned"):
# weighted Z^2_m and H-test on a uniform frequency grid f_j = f0 + j*df.
# Phases are formed in exact rational arithmetic for tiny cases
# (z2m_exact) and in double-double arithmetic for arrays (z2m).
# ---------------------------------------------------------------------------
def _two_prod(a, b):
    """Dekker/Veltkamp exact product a*b = p + e for float64 arrays."""
    p = a * b
    c = 134217729.0
    ah = a * c
    ah = ah - (ah - a)
    al = a - ah
    bh = b * c
    bh = bh - (bh - b)
    bl = b - bh
    e = ((ah * bh - p) + ah * bl + al * bh) + al * bl
    return p, e


def _frac_prod(a, b):
    """frac(a*b) in [-0.5, 0.5] cycles, accurate to ~1e-16 of a cycle."""
    p, e = _two_prod(np.asarray(a, dtype=np.float64),
                     np.asarray(b, dtype=np.float64))
    r = (p - np.rint(p)) + e
    return r - np.rint(r)


def z2m_trig_sums(t, w, f0, df, j, m):
    """Returns C[k-1, j], S[k-1, j] = sum_i w_i cos/sin(2 pi k (f0+j df) t_i)."""
    t = np.asarray(t, dtype=np.float64)
    w = np.asarray(w, dtype=np.float64)
    j = np.asarray(j, dtype=np.float64)
    a = _frac_prod(t, f0)                    # frac(t f0)
    uh, ul = _two_prod(t, np.float64(df))    # t*df as a double-double
    uh = uh - np.rint(uh)                    # reduce mod 1 (exact)
    C = np.empty((m, len(j)))
    Sn = np.empty((m, len(j)))
    for n, jj in enumerate(j):
        p, e = _two_prod(uh, np.float64(jj))
        x = (p - np.rint(p)) + (e + jj * ul) + a
        x = x - np.rint(x)
        for k in range(1, m + 1):
            ph = 2 * np.pi * ((k * x) - np.rint(k * x))
            C[k - 1, n] = np.sum(w * np.cos(ph))
            Sn[k - 1, n] = np.sum(w * np.sin(ph))
    return C, Sn


def z2m(t, w, f0, df, nf, m=2, j0=0):
    """Z^2_m(f_j) = 2/sum(w^2) * sum_{k<=m} (C_k^2 + S_k^2), j = j0..j0+nf-1."""
    C, Sn = z2m_trig_sums(t, w, f0, df, j0 + np.arange(nf), m)
    norm = 2. / np.sum(np.asarray(w, dtype=np.float64) ** 2)
    return norm * np.sum(C ** 2 + Sn ** 2, axis=0)


def h_test(t, w, f0, df, nf, mmax=20, j0=0):
    """H = max_{1<=m<=mmax} (Z^2_m - 4m + 4) per frequency."""
    C, Sn = z2m_trig_sums(t, w, f0, df, j0 + np.arange(nf), mmax)
    norm = 2. / np.sum(np.asarray(w, dtype=np.float64) ** 2)
    z = norm * np.cumsum(C ** 2 + Sn ** 2, axis=0)
    pen = 4. * np.arange(1, mmax + 1)[:, None] - 4
    return np.max(z - pen, axis=0)


def z2m_exact(t, w, f0, df, j, m):
    """Rational-arithmetic phases (pure Python; tiny inputs only)."""
    from fractions import Fraction
    num = 0.0
    for k in range(1, m + 1):
        c = s = 0.0
        for ti, wi in zip(t, w):
            ph = Fraction(float(ti)) * (Fraction(float(f0))
                                        + int(j) * Fraction(float(df))) * k
            x = float(ph - math.floor(ph))
            c += wi * math.cos(2 * math.pi * x)
            s += wi * math.sin(2 * math.pi * x)
        num += c * c + s * s
    return 2. * num / float(np.sum(np.asarray(w) ** 2))


# ---------------------------------------------------------------------------
# Synthetic inputs (deterministic; the generator the survey probed against
# the reference).  Lives here as well as in the package so tests can build
# inputs for the oracle without importing the product.
# ---------------------------------------------------------------------------
T0 = 239557417.0


# ----------------------------------------------------------------------
# Bayesian Blocks on the cached profiles: core.py:1108-1208, 1338-1389
# ----------------------------------------------------------------------
BB_NGRID = 1000


def bb_fitness(dom, cod, i0, i1):
    """core.py:1108-1136: for the cells i0 .. i1-1 the summed log profile of
    every block that ends with cell i1-1, on a 1000-point grid over the
    support of that last cell.  Column j of the result belongs to the
    block that starts at cell i0 + j: 2 * (maximum of the sum, alpha there)."""
    grid = np.linspace(dom[i1 - 1][0], dom[i1 - 1][-1], BB_NGRID)
    out = np.empty((2, i1 - i0))
    acc = np.zeros(BB_NGRID)
    for back in range(i1 - i0):
        c = i1 - 1 - back
        acc += np.interp(grid, dom[c], cod[c], left=-np.inf, right=-np.inf)
        k = np.argmax(acc)
        out[:, -(back + 1)] = acc[k], grid[k]
    return out * 2


def bb_partition(fit0, prior=2):
    """core.py:1146-1169 / 1183-1202: the dynamic programme over the first
    rows (fitness values) of bb_fitness(0, i+1), i = 0 .. ncell-1, and the
    walk back through the change points.  Returns indices, best[-1]."""
    ncell = len(fit0)
    best = np.asarray([-np.inf] * (ncell + 1))
    last = np.empty(ncell, dtype=int)
    last[0] = 0
    best[0] = 0
    best[1] = fit0[0][0] - prior
    for i in range(1, ncell):
        a = fit0[i] - prior + best[:i + 1]
        r = np.argmax(a)
        best[i + 1] = a[r]
        last[i] = r
    cps = []
    k = last[-1]
    while k > 0:
        cps.append(k)
        k = last[k - 1]
    indices = np.append(0, np.asarray(cps, dtype=int)[::-1])
    return indices, best[-1], best, last


def do_bb(dom, cod, prior=2):
    """core.py:1138-1177: indices, total fitness, variability TS and its
    degrees of freedom, and the fitness rows."""
    ncell = len(dom)
    rows = [bb_fitness(dom, cod, 0, i + 1) for i in range(ncell)]
    indices, best_fitness, _, _ = bb_partition([r[0] for r in rows], prior)
    var_dof = len(indices) - 1
    var_ts = (best_fitness - rows[-1][0][0]) + prior * var_dof
    return indices, best_fitness + len(indices) * prior, var_ts, var_dof, rows


def merge_cells(cells):
    """core.py:200-210 (cell_from_cells) on (tstart, tstop, exp, ti, we,
    SonB) tuples: one cell for several."""
    cells = sorted(cells, key=lambda c: c[0])
    we = np.concatenate([c[4] for c in cells])
    ti = np.concatenate([c[3] for c in cells])
    exp = np.asarray([c[2] for c in cells]).sum()
    bexp = np.asarray([c[2] / c[5] for c in cells]).sum()
    return cells[0][0], cells[-1][1], exp, ti, we, exp / bexp


def single_cell_flux(cl, conf=(0.05, 0.95)):
    """core.py:996-1026 (CellLogLikelihood.get_flux, background fixed)."""
    aopt = cl.get_max()
    dom, cod = cl.get_pdf(aopt=aopt)
    k = np.argmax(cod)
    if abs(dom[k] - aopt) > 0.1:
        aopt = cl.get_max(guess=dom[k])
        dom, cod = cl.get_pdf(aopt=aopt, npt=200)
        k = np.argmax(cod)
        if abs(dom[k] - aopt) > 0.1:
            aopt = dom[k]
    ts = cl.get_ts(aopt)
    cdf = cumulative_trapezoid(cod, dom, initial=0)
    cdf *= 1. / cdf[-1]
    idx = np.searchsorted(cdf, conf)
    ihi, ilo = idx, idx - 1
    m = (cdf[ihi] - cdf[ilo]) / (dom[ihi] - dom[ilo])
    return aopt, ts, dom[ilo] + (np.asarray(conf) - cdf[ilo]) / m


def bb_lightcurve(cells, dom, cod, tsmin=4, bb_prior=8):
    """core.py:1338-1389 for (tstart, tstop, exp, ti, we, SonB) tuples and
    their cached profiles; rows as in raw_lightcurve.  Returns rows and
    the block boundaries (cell indices, the last one = ncell)."""
    idx = do_bb(dom, cod, prior=bb_prior)[0]
    idx = np.append(idx, len(cells))
    rows = np.empty([len(idx) - 1, 6])
    for b, (i0, i1) in enumerate(zip(idx[:-1], idx[1:])):
        t0, t1, exp, ti, we, sonb = merge_cells(cells[i0:i1])
        cl = CellLike(we, exp, sonb)
        if cl.S == 0:
            rows[b] = np.nan
            continue
        tmid = 0.5 * (t0 + t1)
        terr = (tmid - t0) / 86400
        aopt, ts, xc = single_cell_flux(cl, conf=[0.16, 0.84])
        if ts <= tsmin:
            rows[b] = met2mjd(tmid), terr, xc[1], 0, -1, ts
        else:
            rows[b] = met2mjd(tmid), terr, aopt, aopt - xc[0], xc[1] - aopt, ts
    return rows, idx


def synth(nph, years=10., seed=7):
    rng = np.random.default_rng(seed)
    nrow = int(years * 365.25 * 86400 / 30)
    rows = np.arange(nrow)
    rows = rows[(rows % 190) < 150]
    TSTART = T0 + 30. * rows
    TSTOP = TSTART + 30.
    LIVETIME = np.full(len(rows), 27.)
    exposure = 27. * (3000 + 2000 * np.sin(2 * np.pi * (TSTART - T0) / 5700) ** 2)
    E = np.sum(exposure)
    idx = np.searchsorted(np.cumsum(exposure) / E, rng.random(nph))
    idx = np.minimum(idx, len(rows) - 1)
    ti = np.sort(TSTART[idx] + 30 * rng.random(nph))
    we = np.clip(rng.beta(0.3, 1.5, nph), 1e-6, 1 - 1e-6)
    S = np.sum(we)
    B = len(we) - S
    return dict(ti=ti, we=we, TSTART=TSTART, TSTOP=TSTOP, LIVETIME=LIVETIME,
                exposure=exposure, S=S, B=B, E=E)
