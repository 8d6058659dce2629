"""Host-side mirror of the reference's analysis API (its core.py) for the
likelihood / periodogram hot path.

Class and function names, argument order, defaults and error behaviour follow
the reference so that its callers can switch over; the arithmetic runs in the
HIP library (godot_amd.ops -> libgodot_hip.so).  There is no CPU path: every
compute method needs a ROCm device.

Device layout.  Many cells are held as one ``CellBatch``: photon weights (and
times) of all cells in one device array plus per-cell ranges ``lo``/``hi``
and per-cell ``starts, stops, sexp, bexp``.  ``Cell`` objects (the
reference's per-cell value type, core.py:161-187) are materialised from a
batch only on request.
"""
import time
import numpy as np
import torch

from . import ops
from ._lib import GodotError
from .scalers import PiecewiseScaler

__all__ = ['Cell', 'PhaseCell', 'cell_from_cells', 'CellBatch', 'CellTimeSeries',
           'CellLogLikelihood', 'CellsLogLikelihood', 'CellsLogLikelihoodOld', 'Data',
           'met2mjd', 'mjd2met', 'infer_met', 'power_spectrum_fft', 'power_spectrum_simple',
           'power_spectrum_dft', 'fit_harmonics', 'compute_ls', 'get_orbital_modulation', 'z2m',
           'h_test', 'PhaseCellsLogLikelihood', 'FitnessRows', 'PhaseData', 'bb_prior_tune']

MJDREF = 51910 + 7.428703703703703e-4
F64 = torch.float64
I64 = torch.int64


def met2mjd(times, mjdref=MJDREF):
    """Reference core.py:24-26 (extended precision on the host)."""
    return np.asarray(times, dtype=np.longdouble) * (1. / 86400) + mjdref


def mjd2met(times, mjdref=MJDREF):
    """Reference core.py:28-30."""
    return (np.asarray(times, dtype=np.longdouble) - mjdref) * 86400


def infer_met(time):
    """Reference core.py:32-35."""
    if time < 100000:
        return mjd2met(time)
    return time


def default_device():
    if not torch.cuda.is_available():
        raise RuntimeError('godot_amd needs a ROCm GPU: no device is visible and there is no '
                           'CPU fallback')
    return torch.device('cuda', torch.cuda.current_device())


def _to_dev(x, dev, dtype=F64):
    if isinstance(x, torch.Tensor):
        return x.to(device=dev, dtype=dtype).contiguous()
    return torch.from_numpy(np.ascontiguousarray(np.asarray(x), dtype={F64: np.float64,
                                                                       I64: np.int64}[dtype])).to(dev)


def _np(x):
    """Host copy as a numpy array.  Large device tensors go through page-locked memory
    (several times the rate of a copy into pageable memory); the array owns that buffer."""
    if not isinstance(x, torch.Tensor):
        return np.asarray(x)
    x = x.detach()
    if x.is_cuda and x.numel() * x.element_size() >= (1 << 20):
        x = x.contiguous()
        h = torch.empty(x.shape, dtype=x.dtype, pin_memory=True)
        h.copy_(x, non_blocking=True)
        torch.cuda.current_stream(x.device).synchronize()
        return h.numpy()
    return x.cpu().numpy()


# ---------------------------------------------------------------------------
# Cell value types (reference core.py:161-210)
# ---------------------------------------------------------------------------
class Cell(object):
    """A start/stop time, the exposure and the photons inside (host arrays)."""

    def __init__(self, tstart, tstop, exposure, photon_times, photon_weights,
                 source_to_background_ratio):
        self.tstart = tstart
        self.tstop = tstop
        self.exp = exposure
        self.ti = np.atleast_1d(photon_times)
        self.we = np.atleast_1d(photon_weights)
        self.SonB = source_to_background_ratio
        self._alpha = 1
        self._beta = 1

    def sanity_check(self):
        if self.exp == 0:
            assert(len(self.ti) == 0)
        assert(len(self.ti) == len(self.we))
        assert(np.all(self.ti >= self.tstart))
        assert(np.all(self.ti < self.tstop))

    def get_tmid(self):
        return 0.5 * (self.tstart + self.tstop)


class PhaseCell(Cell):

    def copy(self, phase_offset=0):
        return PhaseCell(self.tstart + phase_offset, self.tstop + phase_offset,
                         self.exp, self.ti, self.we, self.SonB)


def cell_from_cells(cells):
    """One Cell spanning several (reference core.py:200-210)."""
    cells = sorted(cells, key=lambda c: c.tstart)
    we = np.concatenate([c.we for c in cells])
    ti = np.concatenate([c.ti for c in cells])
    exp = np.asarray([c.exp for c in cells]).sum()
    bexp = np.asarray([c.exp / c.SonB for c in cells]).sum()
    return Cell(cells[0].tstart, cells[-1].tstop, exp, ti, we, exp / bexp)


class CellBatch(object):
    """Device-resident list of cells; behaves like the list of Cell objects
    the reference's Data.get_cells returns (len, indexing, iteration)."""

    def __init__(self, we, ti, lo, hi, starts, stops, sexp, bexp, phase=False, photons=None):
        self.we, self.ti, self.lo, self.hi = we, ti, lo, hi
        self.starts, self.stops, self.sexp, self.bexp = starts, stops, sexp, bexp
        self.device = we.device
        self.phase = bool(phase)      # cells of a folded light curve: indexing yields PhaseCell
        self._host = None
        # photons inside the cells (sum of hi - lo): what the kernels size their shape for; None
        # until someone knows it (ops.CellRanges counts it at the first fit and tells us)
        self._photons = None if photons is None else int(photons)

    @classmethod
    def from_cells(cls, cells, device=None):
        dev = device or default_device()
        counts = np.array([len(c.we) for c in cells], dtype=np.int64)
        off = np.append(0, np.cumsum(counts)).astype(np.int64)
        cat = (lambda xs: np.concatenate(xs) if len(xs) else np.empty(0))
        we = cat([np.asarray(c.we, dtype=np.float64) for c in cells])
        ti = cat([np.asarray(c.ti, dtype=np.float64) for c in cells])
        # the weight scales of Cell._alpha/_beta (reference core.py:341-345)
        if any((c._alpha != 1) or (c._beta != 1) for c in cells):
            a = np.repeat([c._alpha for c in cells], counts)
            b = np.repeat([c._beta for c in cells], counts)
            we = we * a / (we * a + b * (1 - we))
        sexp = np.array([c.exp * c._alpha for c in cells], dtype=np.float64)
        bexp = np.array([c.exp / c.SonB * c._beta for c in cells], dtype=np.float64)
        offd = _to_dev(off, dev, I64)
        return cls(_to_dev(we, dev), _to_dev(ti, dev), offd[:-1], offd[1:],
                   _to_dev([c.tstart for c in cells], dev), _to_dev([c.tstop for c in cells], dev),
                   _to_dev(sexp, dev), _to_dev(bexp, dev), photons=int(off[-1]))

    def ranges(self):
        cr = ops.CellRanges(self.we, self.lo.contiguous(), self.hi.contiguous(), self.sexp,
                            self.bexp, photons=self._photons)
        cr._owner = self              # a count made later is kept here too
        return cr

    def __len__(self):
        return self.lo.numel()

    def _hostcopy(self):
        if self._host is None:
            self._host = dict(lo=_np(self.lo), hi=_np(self.hi), starts=_np(self.starts),
                              stops=_np(self.stops), sexp=_np(self.sexp), bexp=_np(self.bexp))
        return self._host

    def __getitem__(self, i):
        if isinstance(i, slice):
            lo = self.lo[i].contiguous()
            # a slice is taken to be as densely filled as the whole, if that is known
            photons = None
            if self._photons is not None and len(self):
                photons = int(round(self._photons * (lo.numel() / len(self))))
            return CellBatch(self.we, self.ti, lo, self.hi[i].contiguous(),
                             self.starts[i].contiguous(), self.stops[i].contiguous(),
                             self.sexp[i].contiguous(), self.bexp[i].contiguous(), phase=self.phase,
                             photons=photons)
        h = self._hostcopy()
        n = len(h['lo'])
        if i < 0:
            i += n
        if not 0 <= i < n:
            raise IndexError('cell index out of range')
        l, u = int(h['lo'][i]), int(h['hi'][i])
        return (PhaseCell if self.phase else Cell)(h['starts'][i], h['stops'][i], h['sexp'][i],
                                                   _np(self.ti[l:u]), _np(self.we[l:u]),
                                                   h['sexp'][i] / h['bexp'][i])

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]

    def merge_blocks(self, edges):
        """One cell per run of cells edges[b] .. edges[b+1]-1, as cell_from_cells
        (reference core.py:200-210) makes them: photons concatenated, exposures
        summed, start of the earliest and stop of the latest member."""
        h = self._hostcopy()
        edges = np.asarray(edges, dtype=np.int64)
        nb = len(edges) - 1
        lo_h, hi_h = h['lo'], h['hi']
        contiguous = all(np.array_equal(hi_h[s:e - 1], lo_h[s + 1:e]) for s, e in zip(edges[:-1], edges[1:]))
        dev = self.device
        if contiguous:
            we, ti = self.we, self.ti
            lo = _to_dev(lo_h[edges[:-1]], dev, I64)
            hi = _to_dev(hi_h[edges[1:] - 1], dev, I64)
        else:
            # gather the members' photons into a new list, block after block
            cnt = hi_h - lo_h
            src = np.concatenate([np.arange(l, u) for l, u in zip(lo_h, hi_h)] or [np.empty(0, np.int64)])
            src = _to_dev(src.astype(np.int64), dev, I64)
            we, ti = ops.gather(self.we, src), ops.gather(self.ti, src)
            off = np.append(0, np.cumsum(cnt)).astype(np.int64)
            lo = _to_dev(off[edges[:-1]], dev, I64)
            hi = _to_dev(off[edges[1:]], dev, I64)
        starts = np.empty(nb)
        stops = np.empty(nb)
        sexp = np.empty(nb)
        bexp = np.empty(nb)
        for b, (s, e) in enumerate(zip(edges[:-1], edges[1:])):
            starts[b] = h['starts'][s:e].min()
            stops[b] = h['stops'][s:e][np.argmax(h['starts'][s:e])]
            sexp[b] = np.asarray(h['sexp'][s:e]).sum()
            bexp[b] = np.asarray(h['bexp'][s:e]).sum()
        return CellBatch(we, ti, lo, hi, _to_dev(starts, dev), _to_dev(stops, dev), _to_dev(sexp, dev),
                         _to_dev(bexp, dev), phase=self.phase)


def _as_batch(cells, device=None):
    if isinstance(cells, CellBatch):
        return cells
    return CellBatch.from_cells(list(cells), device=device)


# ---------------------------------------------------------------------------
# CellTimeSeries (reference core.py:212-327)
# ---------------------------------------------------------------------------
class CellTimeSeries(object):
    """Uniform-bin moments of the photon weights.  Arrays may be numpy arrays
    or device tensors; like the reference the constructor zeroes, IN PLACE,
    every moment of the bins with exp <= minimum_exposure."""

    _KEYS = ['starts', 'stops', 'exp', 'sexp', 'bexp', 'counts', 'weights', 'weights2',
             'deadtime']

    def __init__(self, starts, stops, exp, sexp, bexp, counts, weights, weights2, deadtime,
                 alt_starts=None, alt_stops=None, timesys='barycenter', minimum_exposure=3e4):
        self.starts = starts
        self.stops = stops
        self.exp = exp
        self.sexp = sexp
        self.bexp = bexp
        self.counts = counts
        self.weights = weights
        self.weights2 = weights2
        self.deadtime = deadtime
        self.alt_starts = alt_starts
        self.alt_stops = alt_stops
        self.timesys = timesys
        self.minimum_exposure = minimum_exposure
        mask = ~(exp > minimum_exposure)
        self._zero_weight(mask)

    def _zero_weight(self, mask):
        for k in ('exp', 'sexp', 'bexp', 'counts', 'weights', 'weights2'):
            getattr(self, k)[mask] = 0

    def save(self, fname, compress=True):
        keys = list(self._KEYS)
        if self.alt_starts is not None:
            keys += ['alt_starts']
        if self.alt_stops is not None:
            keys += ['alt_stops']
        d = dict((k, _np(getattr(self, k))) for k in keys)
        d['metadata'] = [str(self.minimum_exposure), self.timesys]
        (np.savez_compressed if compress else np.savez)(fname, **d)

    @staticmethod
    def load(fname, device=None):
        with np.load(fname) as q:
            minimum_exposure = float(q['metadata'][0])
            timesys = str(q['metadata'][1])
            alt_starts = q['alt_starts'] if 'alt_starts' in q.files else None
            alt_stops = q['alt_stops'] if 'alt_stops' in q.files else None
            args = [q[k] for k in CellTimeSeries._KEYS]
        if device is not None:
            args = [_to_dev(a, device, I64 if a.dtype.kind in 'iu' else F64) for a in args]
        return CellTimeSeries(*args, alt_starts=alt_starts, alt_stops=alt_stops,
                              timesys=timesys, minimum_exposure=minimum_exposure)

    def tsamp(self):
        known = getattr(self, '_tsamp', None)       # set by builders that know the edges on the host
        return float(self.stops[0] - self.starts[0]) if known is None else known

    def ps_nbin(self):
        return len(self.starts) // 4 + 1

    def tspan(self):
        known = getattr(self, '_tspan', None)
        return float(self.stops[-1] - self.starts[0]) if known is None else known

    def get_topo_bins(self):
        if self.timesys == 'barycenter':
            if self.alt_starts is None:
                raise Exception('Do not have topocentric bins.')
            return self.alt_starts, self.alt_stops
        return self.starts, self.stops

    def get_bary_bins(self):
        if self.timesys != 'barycenter':
            if self.alt_starts is None:
                raise Exception('Do not have barycentric bins.')
            return self.alt_starts, self.alt_stops
        return self.starts, self.stops

    def zero_weight(self, tstart, tstop):
        tstart = float(infer_met(tstart))
        tstop = float(infer_met(tstop))
        mask = (self.starts >= tstart) & (self.stops <= tstop)
        self._zero_weight(mask)


# ---------------------------------------------------------------------------
# Likelihood of one cell / of many cells (reference core.py:329-1044, 1047-1336)
# ---------------------------------------------------------------------------
def _scalar(t):
    return float(t.cpu().numpy().ravel()[0])


class CellLogLikelihood(object):
    """Single-cell view of the batched kernels (a batch of one cell)."""

    def __init__(self, cell, swap=False, device=None):
        self.cell = cell
        self.ti = cell.ti
        a, b, w = cell._alpha, cell._beta, np.asarray(cell.we, dtype=np.float64)
        self.we = w * a / (w * a + b * (1 - w))
        self.iwe = 1 - self.we
        self.S = cell.exp * cell._alpha
        self.B = cell.exp / cell.SonB * cell._beta
        self._swap = bool(swap)
        if swap:
            self.we, self.iwe = self.iwe, self.we
            self.S, self.B = self.B, self.S
        dev = device or default_device()
        # the kernels take the un-swapped layout plus the swap switch
        wd = _to_dev(self.iwe if swap else self.we, dev)
        off = torch.tensor([0, wd.numel()], dtype=I64, device=dev)
        S0, B0 = (self.B, self.S) if swap else (self.S, self.B)
        self._cr = ops.CellRanges.from_offsets(
            wd, off, torch.tensor([S0], dtype=F64, device=dev),
            torch.tensor([B0], dtype=F64, device=dev))
        self._dev = dev

    def _t(self, x):
        return torch.tensor([float(x)], dtype=F64, device=self._dev)

    def log_likelihood(self, alpha):
        return _scalar(ops.cell_loglike(self._cr, self._t(alpha), swap=self._swap))

    def get_max(self, guess=1, beta=1, profile_background=False):
        if profile_background:
            a, b, _, fl = ops.cell_fit_joint(self._cr, guess_a=float(guess), guess_b=float(beta),
                                             swap=self._swap)
            return np.array([_scalar(a), _scalar(b)])
        a, _, fl = ops.cell_fit(self._cr, guess=float(guess), beta=float(beta), swap=self._swap,
                                want_ts=False)
        a = _scalar(a)
        if np.isnan(a):
            raise ValueError('f(a) and f(b) must have different signs')
        # the reference returns the integer 0 on the boundary branch
        return 0 if (int(fl.cpu()[0]) & 1) else a

    def get_ts(self, aopt=None, profile_background=False):
        if self.S == 0:
            return 0
        if profile_background:
            if aopt is None:
                aopt = self.get_max(profile_background=True)[0]
            if aopt == 0:
                return 0
            return 2 * (self.log_profile_likelihood(aopt) - self.log_profile_likelihood(0))
        g = 1.0 if aopt is None else float(aopt)
        if aopt is None:
            a, ts, _ = ops.cell_fit(self._cr, swap=self._swap)
            return _scalar(ts)
        if aopt == 0:
            return 0
        l1 = self.log_likelihood(g)
        l0 = self.log_likelihood(0.0)
        return 2 * (l1 - l0)

    def get_logpdf(self, aopt=None, dlogl=20, npt=100, include_zero=False,
                   profile_background=False):
        at = None if aopt is None else self._t(aopt)
        if profile_background:
            dom, cod, fl = ops.cell_logpdf_profile(self._cr, aopt=at, dlogl=dlogl, npt=npt,
                                                   swap=self._swap)
            if int(fl.cpu()[0]) & 0x10:
                raise ValueError('profile likelihood grid failed (no bracket or no convergence)')
            return _np(dom[0]), _np(cod[0])
        dom, cod, fl = ops.cell_logpdf(self._cr, aopt=at, dlogl=dlogl, npt=npt,
                                       include_zero=include_zero, swap=self._swap)
        return _np(dom[0]), _np(cod[0])

    def get_pdf(self, aopt=None, dlogl=20, npt=100, profile_background=False):
        from scipy.integrate import simpson
        dom, cod = self.get_logpdf(aopt=aopt, dlogl=dlogl, npt=npt,
                                   profile_background=profile_background)
        np.exp(cod, out=cod)
        return dom, cod * (1. / simpson(cod, x=dom))

    def get_flux(self, conf=[0.05, 0.95], profile_background=False):
        """(aopt, ts, xconf) of the cell (reference core.py:996-1026)."""
        from scipy.integrate import cumulative_trapezoid
        aopt = self.get_max(profile_background=profile_background)
        if profile_background:
            aopt = aopt[0]
        dom, cod = self.get_pdf(aopt=aopt, profile_background=profile_background)
        amax = np.argmax(cod)
        if abs(dom[amax] - aopt) > 0.1:
            aopt = self.get_max(guess=dom[amax], profile_background=profile_background)
            if profile_background:
                aopt = aopt[0]
            dom, cod = self.get_pdf(aopt=aopt, profile_background=profile_background, npt=200)
            amax = np.argmax(cod)
            if abs(dom[amax] - aopt) > 0.1:
                aopt = dom[amax]
        ts = self.get_ts(aopt=aopt, profile_background=profile_background)
        cdf = cumulative_trapezoid(cod, dom, initial=0)
        cdf *= 1. / cdf[-1]
        indices = np.searchsorted(cdf, conf)
        ihi, ilo = indices, indices - 1
        m = (cdf[ihi] - cdf[ilo]) / (dom[ihi] - dom[ilo])
        return aopt, ts, dom[ilo] + (np.asarray(conf) - cdf[ilo]) / m

    def get_beta_max(self, alpha, guess=1):
        b, _, fl = ops.cell_beta_max(self._cr, float(alpha), guess=float(guess), swap=self._swap,
                                     want_logl=False)
        if int(fl.cpu()[0]) & 0x10:
            raise ValueError('Did not converge for alpha=%.5f!' % alpha)
        return _scalar(b)

    def log_profile_likelihood(self, alpha, beta_guess=1):
        b, lp, fl = ops.cell_beta_max(self._cr, float(alpha), guess=float(beta_guess),
                                      swap=self._swap)
        if int(fl.cpu()[0]) & 0x10:
            raise ValueError('Did not converge for alpha=%.5f!' % alpha)
        self._last_beta = _scalar(b) - 1
        return _scalar(lp)

    # ---- the reference's small per-cell functions, on the device sums of godot_cell_sums:
    # s = [sum log t, sum w/t, sum iw/t, sum (w/t)^2, sum (w/t)^3, sum (iw/t)^2,
    #      sum (w/t)(iw/t), sum (w/t)(iw/t)^2],  t = 1 + (beta-1) iw + (alpha-1) w
    def _sums(self, alpha, beta=1.0):
        return _np(ops.cell_sums(self._cr, alpha=float(alpha), beta=float(beta), swap=self._swap)[0])

    def get_likelihood_grid(self, amax=2, bmax=2, res=0.01):
        """logL on the grid [0, amax] x [0, bmax] of (alpha, beta) (core.py:408-425): one
        launch, every grid point a view of the same photons."""
        na = int(round(amax / res)) + 1
        nb = int(round(bmax / res)) + 1
        agrid = np.linspace(0, amax, na) - 1
        bgrid = np.linspace(0, bmax, nb) - 1
        cr, dev = self._cr, self._dev
        npt = na * nb
        grid = ops.CellRanges(cr.we, cr.lo.expand(npt).contiguous(), cr.hi.expand(npt).contiguous(),
                              cr.S.expand(npt).contiguous(), cr.B.expand(npt).contiguous())
        al = _to_dev(np.repeat(agrid + 1, nb), dev)
        be = _to_dev(np.tile(bgrid + 1, na), dev)
        s0 = _np(ops.cell_sums(grid, alpha=al, beta=be, swap=self._swap)[:, 0]).reshape(na, nb)
        with np.errstate(invalid='ignore'):
            rvals = s0 - self.B * (1 + bgrid)[None, :]
            rvals -= (self.S * (1 + agrid))[:, None]
        return agrid + 1, bgrid + 1, rvals

    def log_full_likelihood(self, p):
        """Likelihood for both source and background normalizations (core.py:427-438)."""
        alpha, beta = float(p[0]), float(p[1])
        return self._sums(alpha, beta)[0] - self.S * alpha - self.B * beta

    def fmin_tnc_func(self, p):
        """(-logL, [-dlogL/dalpha, -dlogL/dbeta]) as the reference hands it to fmin_tnc (core.py:440-460)."""
        alpha, beta = float(p[0]), float(p[1])
        if (alpha == 0) and (beta == 0):
            return np.inf, [np.inf, np.inf]     # would call log(0)
        s = self._sums(alpha, beta)
        logl = s[0] - self.S * (alpha - 1) - self.B * (beta - 1)
        return -logl, [-(s[1] - self.S), -(s[2] - self.B)]

    def fmin_fsolve(self, p):
        """The gradient as the reference hands it to fsolve (core.py:462-474); it prints its argument."""
        s = self._sums(float(p[0]), float(p[1]))
        grad_alpha, grad_beta = s[1] - self.S, s[2] - self.B
        print(p, grad_alpha, grad_beta)
        return [-grad_alpha, -grad_beta]

    def _get_logpdf_profile(self, aopt=None, dlogl=20, npt=100, include_zero=False):
        return self.get_logpdf(aopt=aopt, dlogl=dlogl, npt=npt, include_zero=include_zero,
                               profile_background=True)

    def fmin_fsolve_jac(self, p):
        """Declared and left empty in the reference (core.py:476-477); kept for callers that
        look the name up."""
        return None

    def f1(self, alpha):
        return self._sums(alpha)[1] - self.S

    def f2(self, alpha):
        return -self._sums(alpha)[3]

    def f3(self, alpha):
        return self._sums(alpha)[4]

    def f1_profile(self, alpha):
        """Derivative of the profile likelihood in its small-variation form (core.py:483-497)."""
        s = self._sums(alpha)
        T2 = s[5]
        beta_hat = (s[2] - self.B) / T2
        beta_hat_prime = (2 * beta_hat * s[7] - s[6]) / T2
        s2 = self._sums(alpha, 1 + beta_hat)
        return s2[1] + beta_hat_prime * s2[2] - self.S - beta_hat_prime * self.B

    def log_profile_likelihood_approx(self, alpha):
        """Profile over a background level assumed to vary by a few per cent (core.py:366-389)."""
        s = self._sums(alpha)
        beta_hat = (s[2] - self.B) / s[5]
        return self._sums(alpha, 1 + beta_hat)[0] - alpha * self.S - (1 + beta_hat) * self.B

    def nr(self, guess=1, niter=6):
        """Newton-Raphson solution to max (core.py:509-524)."""
        a = guess
        for i in range(niter):
            s = self._sums(a)
            a = max(0, a - (s[1] - self.S) / (-s[3]))
        return a

    def halley(self, guess=1, niter=5):
        """Halley's method solution to max (core.py:526-542)."""
        a = guess
        for i in range(niter):
            s = self._sums(a)
            f1, f2, f3 = s[1] - self.S, -s[3], 2 * s[4]
            a = max(0, a - 2 * f1 * f2 / (2 * f2 * f2 - f1 * f3))
        return a

    def get_max_numerical(self, guess=1, beta=1, profile_background=False):
        """The bracketing solution of the reference (core.py:693-739); the function values come
        from the device, the one-dimensional solver is scipy's, as in the reference."""
        if profile_background:
            from scipy.optimize import fmin_tnc
            rvals, nfeval, rc = fmin_tnc(self.fmin_tnc_func, [guess, beta], bounds=[[0, None], [0, None]],
                                         disp=0, ftol=1e-3)
            if (rc < 0) or (rc > 2):
                print('Warning, best guess probably wrong.')
            return rvals
        from scipy.optimize import brentq

        def f(a):
            return self._sums(a + 1, beta)[1] - self.S

        # the maximum is at zero flux if the derivative there is negative
        if f(-1) < 0:
            return 0
        a0, amax = -1, guess - 1
        for i in range(12):
            if f(amax) < 0:
                break
            a0 = amax
            amax = 2 * amax + 1
        return brentq(f, a0, amax, xtol=1e-3) + 1

    def get_beta_max_numerical(self, alpha, guess=1):
        """core.py:825-849."""
        from scipy.optimize import brentq

        def f(b):
            return self._sums(alpha, b + 1)[2] - self.B

        if f(-1) < 0:
            return 0
        b0, bmax = -1, guess - 1
        for i in range(8):
            if f(bmax) < 0:
                break
            b0 = bmax
            bmax = 2 * bmax + 1
        return brentq(f, b0, bmax, xtol=1e-3) + 1

    def get_profile_flux(self):
        """Flux estimator after profiling over the background, from the expansion for small
        deviations (core.py:1028-1044)."""
        N = len(self.we)
        w1, w2 = ops.cell_moments(self._cr.we, self._cr.lo, self._cr.hi)
        W, W2 = _scalar(w1), _scalar(w2)
        if self._swap:      # the kernel summed the un-swapped weights
            W, W2 = N - W, N - 2 * W + W2
        t1 = W2 - W
        t2 = t1 + (N - W)
        return ((W - self.S) * t2 + (N - self.B - W) * t1) / (W2 * t2 - t1 ** 2)


def _flag_summary(flags):
    """(any cell with GODOT_F_BAD_RANGE, any cell with GODOT_F_FAILED): ONE read from the device."""
    both = torch.stack([((flags & 0x80) != 0).any(), ((flags & 0x10) != 0).any()]).cpu()
    return bool(both[0]), bool(both[1])


def _raise_bad_ranges(flags, known=None):
    """A cell whose photon range lies outside the photon arrays (GODOT_F_BAD_RANGE) is a
    fault of whoever built the ranges, never of the data: no allow_failed_cells hides it."""
    if _flag_summary(flags)[0] if known is None else known:
        bad = (flags & 0x80) != 0
        raise GodotError('photon range of cell %d is not inside the photon arrays'
                         % int(torch.nonzero(bad)[0].cpu()))


def _raise_as_reference(flags, allow):
    """The reference fits cell after cell and stops at the first one it cannot fit: a
    ValueError from its root finder when the score has no sign change (core.py:728-739),
    a RecursionError when get_logpdf re-centres without end (core.py:923-929).  The
    kernels mark such cells (GODOT_F_FAILED) and go on; with allow=False the batch call
    then raises what the reference would have."""
    bad_range, failed = _flag_summary(flags)
    _raise_bad_ranges(flags, bad_range)
    if allow or not failed:
        return
    bad = (flags & 0x10) != 0
    fl = int(flags[bad][0].cpu())
    if fl & 0x08:
        raise ValueError('f(a) and f(b) must have different signs')
    raise RecursionError('the profile of a cell was re-centred 50 times without agreeing with its maximum')


def _raise_unconverged_beta(flags, allow, alpha=None):
    """get_beta_max raises a ValueError when its restarts run out (reference core.py:816-819);
    the kernels mark such cells GODOT_F_FAILED and go on."""
    bad_range, failed = _flag_summary(flags)
    _raise_bad_ranges(flags, bad_range)
    if allow or not failed:
        return
    bad = (flags & 0x10) != 0
    i = int(torch.nonzero(bad)[0].cpu())
    a = '' if alpha is None or not isinstance(alpha, torch.Tensor) else ' for alpha=%.5f' % float(alpha[i].cpu())
    raise ValueError('Did not converge%s!' % a)


class CellsLogLikelihood(object):
    """Batch driver: cached (ncell, 200) log-likelihood profiles, fluxes and
    the raw light curve (reference core.py:1070-1098, 1227-1336).

    allow_failed_cells=True keeps going where the reference raises out of a cell it
    cannot fit: such cells get NaN profiles, fluxes and light-curve rows."""

    def __init__(self, cells, profile_background=False, swap=False, reverse=False, device=None,
                 allow_failed_cells=False):
        self.allow_failed_cells = bool(allow_failed_cells)
        batch = _as_batch(cells, device=device)
        if reverse:
            batch = batch[::-1] if not isinstance(batch, CellBatch) else CellBatch(
                batch.we, batch.ti, batch.lo.flip(0).contiguous(), batch.hi.flip(0).contiguous(),
                batch.starts.flip(0).contiguous(), batch.stops.flip(0).contiguous(),
                batch.sexp.flip(0).contiguous(), batch.bexp.flip(0).contiguous())
        self.cells = batch
        self._swap = bool(swap)
        self.profile_background = bool(profile_background)
        self._cr = batch.ranges()
        if profile_background:
            self._dom_d, self._cod_d, self._flags = ops.cell_logpdf_profile(
                self._cr, dlogl=30, npt=200, swap=self._swap)
            if bool((self._flags & 0x10).any().cpu()):
                # the reference raises out of its constructor in this case (core.py:818)
                raise ValueError('Did not converge while profiling the background of a cell')
        else:
            self._dom_d, self._cod_d, self._flags = ops.cell_logpdf(self._cr, dlogl=30, npt=200,
                                                                    swap=self._swap)
            _raise_as_reference(self._flags, self.allow_failed_cells)
        # the rows of _dom_d are what the kernels wrote: numpy.linspace of their ends, to the bit
        self._rows_linspace = True
        self._dom_h = self._cod_h = None
        self.fitness_vals = None

    _rows_linspace = False

    @property
    def _dom(self):
        if self._dom_h is None:
            self._dom_h = _np(self._dom_d)
        return self._dom_h

    @property
    def _cod(self):
        if self._cod_h is None:
            self._cod_h = _np(self._cod_d)
        return self._cod_h

    @property
    def clls(self):
        return [CellLogLikelihood(c, swap=self._swap, device=self.cells.device)
                for c in self.cells]

    def __len__(self):
        return len(self.cells)

    def sanity_check(self):
        """Report cells with a suspiciously narrow support or a profile that does not peak at
        its maximum (reference core.py:1100-1106)."""
        dx = self._dom[:, -1] - self._dom[:, 0]
        bad_x = np.ravel(np.argwhere(np.abs(dx) < 1e-3))
        print('Indices with suspiciously narrow support ranges: ', bad_x)
        ymax = self._cod.max(axis=1)
        bad_y = np.ravel(np.argwhere(np.abs(ymax) > 0.2))
        print('Indices where there is a substantial disagreement in the optimized value of the log '
              'likelihood and the codomain: ', bad_y)
        return bad_x, bad_y

    def get_flux_all(self, conf=[0.05, 0.95]):
        """(aopt, ts, xlo, xhi) device tensors for every cell."""
        ag, xlo, xhi = ops.profile_quantiles(self._dom_d, self._cod_d, conf[0], conf[1],
                                             linspace=self._rows_linspace)
        if self.profile_background:
            # core.py:1231-1245: beta guess from the profile, joint refit, profile TS
            bg, _, bf = ops.cell_beta_max(self._cr, ag, swap=self._swap, want_logl=False)
            _raise_unconverged_beta(bf, self.allow_failed_cells, ag)
            # a joint fit that stops short is no error in the reference either (it prints a
            # warning, core.py:561, 619, 628): only ranges outside the photon arrays are
            aopt, _, _, jf = ops.cell_fit_joint(self._cr, guess_a=ag, guess_b=bg, swap=self._swap)
            _raise_bad_ranges(jf)
            _, lp1, bf1 = ops.cell_beta_max(self._cr, aopt, swap=self._swap)
            _, lp0, bf0 = ops.cell_beta_max(self._cr, 0.0, swap=self._swap)
            _raise_unconverged_beta(bf1 | bf0, self.allow_failed_cells, aopt)
            ts = 2 * (lp1 - lp0)
            s_eff = self._cr.B if self._swap else self._cr.S     # what get_ts tests: core.py:982-984, 343-345
            ts = torch.where((aopt == 0) | (s_eff == 0), torch.zeros_like(ts), ts)
            return aopt, ts, xlo, xhi
        aopt, ts, fl = ops.cell_fit(self._cr, guess=ag, beta=1.0, swap=self._swap)
        _raise_as_reference(fl, self.allow_failed_cells)
        return aopt, ts, xlo, xhi

    def get_flux(self, idx, conf=[0.05, 0.95]):
        sub = CellsLogLikelihood.__new__(CellsLogLikelihood)
        sub._cr = ops.CellRanges(self._cr.we, self._cr.lo[idx:idx + 1].contiguous(),
                                 self._cr.hi[idx:idx + 1].contiguous(),
                                 self._cr.S[idx:idx + 1].contiguous(),
                                 self._cr.B[idx:idx + 1].contiguous())
        sub._dom_d = self._dom_d[idx:idx + 1].contiguous()
        sub._cod_d = self._cod_d[idx:idx + 1].contiguous()
        sub._rows_linspace = self._rows_linspace
        sub._swap = self._swap
        sub.profile_background = self.profile_background
        sub.allow_failed_cells = self.allow_failed_cells
        aopt, ts, xlo, xhi = sub.get_flux_all(conf)
        return _scalar(aopt), _scalar(ts), np.array([_scalar(xlo), _scalar(xhi)])

    def log_likelihood(self, alpha, slow_but_sure=False):
        if len(alpha) != len(self.cells):
            raise ValueError('Must provide a value of alpha for all cells!')
        if not slow_but_sure:
            dom, cod = self._dom, self._cod
            alpha_h = _np(alpha)
            if np.all((alpha_h >= dom[:, 0]) & (alpha_h <= dom[:, -1])):
                return sum(np.interp(a, d, c) for a, d, c in zip(alpha_h, dom, cod))
        at = _to_dev(alpha, self.cells.device)
        if self.profile_background:
            _, lp, _ = ops.cell_beta_max(self._cr, at, swap=self._swap)
            return float(lp.sum().cpu())
        return float(ops.cell_loglike(self._cr, at, swap=self._swap).sum().cpu())

    # ---- Bayesian Blocks (reference core.py:1108-1225, 1338-1389) ------------
    def _fitness_rows(self):
        if self.fitness_vals is None:
            fv, fa = ops.bb_fitness(self._dom_d, self._cod_d, ngrid=1000)
            self.fitness_vals = FitnessRows(fv, fa, len(self.cells))
        return self.fitness_vals

    def fitness(self, i0, i1):
        """2 * (maximum, alpha at the maximum) of the summed profiles of the
        blocks i0 .. i1-1, i0+1 .. i1-1, ..., i1-1 (reference core.py:1108-1136)."""
        return self._fitness_rows()[i1 - 1][:, i0:]

    def do_bb(self, prior=2):
        """indices, total fitness, variability TS, its degrees of freedom, fitness
        rows (reference core.py:1138-1208; the rows are computed once)."""
        rows = self._fitness_rows()
        n = len(self.cells)
        best, last = ops.bb_partition(rows.fit_val, n, prior)
        best_fitness = float(best[-1].cpu())
        last = _np(last)
        cps = []
        k = last[-1]
        while k > 0:
            cps.append(k)
            k = last[k - 1]
        indices = np.append(0, np.asarray(cps, dtype=int)[::-1])
        var_dof = len(indices) - 1
        whole = float(rows.fit_val[(n - 1) * n // 2].cpu())      # fitness of the single block 0 .. n-1
        var_ts = (best_fitness - whole) + prior * var_dof
        return indices, best_fitness + len(indices) * prior, var_ts, var_dof, rows

    _do_bb_cache = do_bb

    def do_top_hat(self):
        """Top-hat filtered map of TS and flux (reference core.py:1210-1225)."""
        rows = self.do_bb()[-1]
        n = len(rows)
        out = np.full((2, n, n), np.nan)
        r = np.arange(n)
        for iv in range(n - 1):          # the full-length row lands on the diagonal, which ends as NaN
            v = rows[iv]
            out[:, r[:iv + 1], r[:iv + 1] + n - (iv + 1)] = v
        out[0] += (np.arange(n) * 2)[None, ::-1]
        out[1] *= 0.5
        return out

    def _block_fluxes(self, blocks):
        """CellLogLikelihood.get_flux(conf=[0.16, 0.84]) of every merged cell
        (reference core.py:996-1026), batched; the rare cells whose profile
        maximum and fitted maximum disagree take the reference's refinement
        one by one."""
        cr = blocks.ranges()
        if self.profile_background:
            aopt, _, _, fl = ops.cell_fit_joint(cr, swap=self._swap)
            _raise_bad_ranges(fl)
            dom, cod, pf = ops.cell_logpdf_profile(cr, aopt=aopt, dlogl=20, npt=100, swap=self._swap)
            if not self.allow_failed_cells and bool((pf & 0x10).any().cpu()):
                # get_beta_max out of restarts or brentq without a sign change inside
                # _get_logpdf_profile: the reference raises out of get_flux (core.py:816-819, 959-964)
                raise ValueError('Did not converge while profiling the background of a block')
            _, lp1, bf1 = ops.cell_beta_max(cr, aopt, swap=self._swap)
            _, lp0, bf0 = ops.cell_beta_max(cr, 0.0, swap=self._swap)
            _raise_unconverged_beta(bf1 | bf0, self.allow_failed_cells, aopt)
            ts = 2 * (lp1 - lp0)
            s_eff = cr.B if self._swap else cr.S                 # what get_ts tests: core.py:982-984, 343-345
            ts = torch.where((aopt == 0) | (s_eff == 0), torch.zeros_like(ts), ts)
        else:
            aopt, ts, fl = ops.cell_fit(cr, swap=self._swap)
            _raise_bad_ranges(fl)
            if bool((fl & 0x10).any().cpu()):
                raise ValueError('f(a) and f(b) must have different signs')
            dom, cod, _ = ops.cell_logpdf(cr, aopt=aopt, dlogl=20, npt=100, swap=self._swap)
        ag, xlo, xhi = ops.profile_quantiles(dom, cod, 0.16, 0.84, linspace=True)
        aopt, ts, xlo, xhi = _np(aopt), _np(ts), _np(xlo), _np(xhi)
        for b in np.flatnonzero(np.abs(_np(ag) - aopt) > 0.1):
            cl = CellLogLikelihood(blocks[int(b)], swap=self._swap, device=blocks.device)
            a, t, xc = cl.get_flux(conf=[0.16, 0.84], profile_background=self.profile_background)
            aopt[b], ts[b], xlo[b], xhi[b] = a, t, xc[0], xc[1]
        return aopt, ts, xlo, xhi

    def get_bb_lightcurve(self, tsmin=4, bb_prior=8, get_indices=False):
        """Bayesian-Blocks light curve, rows as in get_raw_lightcurve
        (reference core.py:1338-1389)."""
        plot_phase = isinstance(self, PhaseCellsLogLikelihood)
        bb_idx = self.do_bb(prior=bb_prior)[0]
        bb_idx = np.append(bb_idx, len(self.cells))
        blocks = self.cells.merge_blocks(bb_idx)
        aopt, ts, xlo, xhi = self._block_fluxes(blocks)
        h = blocks._hostcopy()
        S = h['bexp'] if self._swap else h['sexp']
        tmid = 0.5 * (h['starts'] + h['stops'])
        terr = tmid - h['starts']
        if not plot_phase:
            terr = terr / 86400
            tmid = met2mjd(tmid)
        rows = np.empty([len(bb_idx) - 1, 6])
        ul = ts <= tsmin
        rows[:, 0], rows[:, 1], rows[:, 5] = tmid, terr, ts
        rows[:, 2] = np.where(ul, xhi, aopt)
        rows[:, 3] = np.where(ul, 0, aopt - xlo)
        rows[:, 4] = np.where(ul, -1, xhi - aopt)
        rows[S == 0] = np.nan
        if get_indices:
            return rows, bb_idx
        return rows

    def get_raw_lightcurve(self, tsmin=4):
        """(ncell, 6) rows: MJD mid-time, half width (days), flux or upper
        limit, lower error, upper error (-1 for limits), TS."""
        aopt, ts, xlo, xhi = self.get_flux_all(conf=[0.159, 0.841])
        rows = ops.lightcurve_rows(self.cells.starts, self.cells.stops, aopt, ts, xlo, xhi,
                                   self._cr.S, tsmin)
        if self.allow_failed_cells:
            # cells the reference would have stopped at: no number in their rows
            rows[(self._flags & 0x10) != 0] = float('nan')
        # the six columns are put together on the device, so that the host receives the array
        # it hands out in one copy; the MJD column with the reference's extended-precision
        # roundings restated in integers (times the restatement does not cover come back
        # NaN and are done on the host, as the reference does all of them)
        tmid = 0.5 * (self.cells.starts + self.cells.stops)
        out_d = torch.empty((len(self.cells), 6), dtype=F64, device=rows.device)
        out_d[:, 1:] = rows
        mjd = ops.met2mjd(tmid, MJDREF)
        out_d[:, 0] = mjd
        # Whether any time was left to the host is asked on the device: a pass of the host over
        # a column of the page-locked result costs more than the copy itself (0.75 ms against
        # 0.40 ms for 4.4e5 rows).  (The same call in four parts, the rows of a part on their way
        # to the host while the quantiles of the next are taken, was measured: 1.05 ms against
        # 0.89 ms, the launches and slices of the parts cost more than the overlap gains.)
        left_any = torch.isnan(mjd).any()
        out = _np(out_d)
        if bool(left_any):
            left = np.isnan(out[:, 0])
            out[left, 0] = met2mjd(_np(tmid)[left])
        return out


class FitnessRows(object):
    """The reference's `fitness_vals` (a deque of (2, i+1) arrays, core.py:1148-1156)
    kept on the device as two packed lower triangles; rows are copied to the
    host when they are read."""

    def __init__(self, fit_val, fit_arg, ncell):
        self.fit_val, self.fit_arg, self.ncell = fit_val, fit_arg, ncell
        self._host = None

    def __len__(self):
        return self.ncell

    def __getitem__(self, i):
        if i < 0:
            i += self.ncell
        if not 0 <= i < self.ncell:
            raise IndexError('row index out of range')
        if self._host is None:
            self._host = (_np(self.fit_val), _np(self.fit_arg))
        off = i * (i + 1) // 2
        return np.stack([self._host[0][off:off + i + 1], self._host[1][off:off + i + 1]])

    def __iter__(self):
        for i in range(self.ncell):
            yield self[i]


class PhaseCellsLogLikelihood(CellsLogLikelihood):
    """Bayesian Blocks on a folded light curve: profiles and cells wrapped once
    on either side (reference core.py:1392-1408)."""

    def __init__(self, cells, device=None):
        super(PhaseCellsLogLikelihood, self).__init__(cells, device=device)
        b = self.cells
        self._orig_cells = b
        rep = lambda x: torch.cat([x, x, x]).contiguous()
        self._dom_d, self._cod_d = rep(self._dom_d), rep(self._cod_d)
        self._dom_h = self._cod_h = None
        self.cells = CellBatch(b.we, b.ti, rep(b.lo), rep(b.hi),
                               torch.cat([b.starts - 1, b.starts, b.starts + 1]).contiguous(),
                               torch.cat([b.stops - 1, b.stops, b.stops + 1]).contiguous(),
                               rep(b.sexp), rep(b.bexp), phase=True)
        self._cr = self.cells.ranges()
        self.profile_background = False


class CellsLogLikelihoodOld(object):
    """Flat photon layout for the direct trig sums (reference core.py:2388-2413)."""

    def __init__(self, cells, device=None):
        batch = _as_batch(cells, device=device)
        self.cells = batch
        self.nph = int((batch.hi - batch.lo).sum().cpu())
        self.exp = batch.sexp
        self.tmids = 0.5 * (batch.starts + batch.stops)
        self._cr = batch.ranges()

    def log_likelihood(self, alpha):
        """Alpha is a vector with one entry per cell (reference core.py:2415-2423)."""
        al = _to_dev(alpha, self._cr.device)
        # per cell: sum log(alpha w + 1 - w) - exp alpha (a cell's S is its exposure)
        return float(ops.cell_loglike(self._cr, al).sum().cpu())


# ---------------------------------------------------------------------------
# Data: the array tail of the reference's Data class (core.py:1792-1843, 1958-2206)
# ---------------------------------------------------------------------------
class Data(object):
    """Photon list + exposure table.  File ingest (FITS, IRFs) is not part of
    this package: construct from arrays."""

    def __init__(self, ti, we, TSTART, TSTOP, LIVETIME, exposure, timeref='LOCAL', device=None,
                 verbosity=1):
        dev = device or default_device()
        self.device = dev
        self._verbosity = verbosity
        self.timeref = timeref
        self.ti = _to_dev(ti, dev)
        self.we = _to_dev(we, dev)
        self.TSTART = _to_dev(TSTART, dev)
        self.TSTOP = _to_dev(TSTOP, dev)
        self.LIVETIME = _to_dev(LIVETIME, dev)
        self.exposure = _to_dev(exposure, dev)
        # totals as the reference forms them (core.py:1583-1585)
        if isinstance(we, torch.Tensor):
            self.S = float(self.we.sum().cpu())
        else:
            self.S = float(np.sum(np.asarray(we, dtype=np.float64)))
        self.B = self.we.numel() - self.S
        if isinstance(exposure, torch.Tensor):
            self.E = float(self.exposure.sum().cpu())
        else:
            self.E = float(np.sum(np.asarray(exposure, dtype=np.float64)))
        self._exposure_table = None
        self._checkpoints = None
        self._t0 = float(self.TSTART[0].cpu())
        self._t1 = float(self.TSTOP[-1].cpu())
        # per-row scaled exposures while scalers are applied (reference core.py:2282-2316)
        self.sexposure = None
        self.bexposure = None
        self._scaled_tables = {}
        self._scaled_key = None
        self._ft2_host = None

    def _vprint(self, s, level):
        if self._verbosity >= level:
            print(s)

    def get_exposure(self, times, deadtime_too=False, cumulative=True):
        """(exp, sexp, bexp, dead) at / between the given times."""
        t = _to_dev(times, self.device)
        if self.sexposure is not None:
            # scalers applied: running sums over the rescaled rows, kept per scaler pair
            tab = self._scaled_tables.get(self._scaled_key)
            if tab is None:
                tab = ops.exposure_table_rows(self.TSTART, self.TSTOP, self.exposure, self.sexposure,
                                              self.bexposure, self.LIVETIME)
                if self._scaled_key is not None:
                    self._scaled_tables[self._scaled_key] = tab
            return ops.exposure_at_rows(self.TSTART, self.TSTOP, self.exposure, self.sexposure,
                                        self.bexposure, self.LIVETIME, tab, t, cumulative=cumulative,
                                        deadtime_too=deadtime_too)
        return ops.exposure_at(self.TSTART, self.TSTOP, self.exposure, self.LIVETIME,
                               self.S / self.E, self.B / self.E, self._plain_exposure_table(), t,
                               cumulative=cumulative, deadtime_too=deadtime_too)

    def _plain_exposure_table(self):
        """Running sums of the exposure table (no scalers applied): built once, reused by
        every query."""
        if self._exposure_table is None:
            self._exposure_table = ops.exposure_table(
                self.TSTART, self.TSTOP, self.exposure, self.LIVETIME, self.S / self.E,
                self.B / self.E)
        return self._exposure_table

    def get_deadtime(self, times):
        """Cumulative dead time at the given times (reference core.py:1845-1847)."""
        return self.get_exposure(times, deadtime_too=True)[-1]

    def zero_weight(self, tstart, tstop):
        """Remove the exposure rows that lie inside [tstart, tstop] and the photons that fall in
        them; totals S, B, E follow (reference core.py:1593-1618)."""
        tstart = float(infer_met(tstart))
        tstop = float(infer_met(tstop))
        keep = ~((self.TSTART >= tstart) & (self.TSTOP <= tstop))
        row = ops.search_sorted(self.TSTOP, self.ti)
        if bool((row >= keep.numel()).any().cpu()):
            raise IndexError('a photon lies after the last exposure row')
        event_keep = keep[row]
        self.exposure = self.exposure[keep].contiguous()
        self.TSTART = self.TSTART[keep].contiguous()
        self.TSTOP = self.TSTOP[keep].contiguous()
        self.LIVETIME = self.LIVETIME[keep].contiguous()
        self.ti = self.ti[event_keep].contiguous()
        self.we = self.we[event_keep].contiguous()
        self.E = float(np.cumsum(_np(self.exposure))[-1])      # the reference's E is cumsum(exposure)[-1]
        self.S = float(np.sum(_np(self.we)))
        self.B = self.we.numel() - self.S
        self._t0 = float(self.TSTART[0].cpu())
        self._t1 = float(self.TSTOP[-1].cpu())
        self._exposure_table = None
        self._checkpoints = None
        self._scaled_tables = {}
        self._scaled_key = None
        self._ft2_host = None

    def _ft2(self):
        if self._ft2_host is None:
            self._ft2_host = (_np(self.TSTART), _np(self.TSTOP))
        return self._ft2_host

    def apply_scalers(self, src_scaler=None, bkg_scaler=None):
        """Rescale the rows' source / background exposures and the photon weights
        (reference core.py:2282-2306).  Returns what revert_scalers needs."""
        if (src_scaler is None) and (bkg_scaler is None):
            return None
        original = (self.we, self.sexposure, self.bexposure, self._scaled_key)
        key = (id(src_scaler), id(bkg_scaler))
        cached = self._scaled_tables.get(('rows',) + key)
        if cached is None:
            # the rows of the spacecraft table stay on the device; a scaler of another kind
            # (any callable of host arrays, as in the reference) gets host arrays
            dev_ok = all(s is None or isinstance(s, PiecewiseScaler) for s in (src_scaler, bkg_scaler))
            T0, T1 = (self.TSTART, self.TSTOP) if dev_ok else self._ft2()
            sexp = self.exposure * (self.S / self.E)
            bexp = self.exposure * (self.B / self.E)
            if src_scaler is not None:
                sexp = sexp * _to_dev(src_scaler(T0, T1), self.device)
            if bkg_scaler is not None:
                bexp = bexp * _to_dev(bkg_scaler(T0, T1), self.device)
            sscl = src_scaler(self.ti) if src_scaler is not None else 1.0
            bscl = bkg_scaler(self.ti) if bkg_scaler is not None else 1.0
            we = ops.rescale_weights(self.we, _to_dev(sscl, self.device) if not np.isscalar(sscl) else sscl,
                                     _to_dev(bscl, self.device) if not np.isscalar(bscl) else bscl)
            # the scaler objects are kept alive with their tables so that ids stay unique
            cached = (sexp.contiguous(), bexp.contiguous(), we, src_scaler, bkg_scaler)
            self._scaled_tables[('rows',) + key] = cached
        self.sexposure, self.bexposure, self.we = cached[0], cached[1], cached[2]
        self._scaled_key = key
        return original

    def revert_scalers(self, original_values):
        if original_values is None:
            return
        self.we, self.sexposure, self.bexposure, self._scaled_key = original_values

    def get_contiguous_exposures(self, tstart=None, tstop=None, max_interval=10):
        """Start / stop times of the stretches of uninterrupted exposure
        (reference core.py:1849-1860, py_exposure_p8.py:34-86); host arithmetic
        on the exposure table."""
        t0s, t1s = self._ft2()
        if tstart is not None:
            i = np.searchsorted(t1s, tstart)
            t0s, t1s = t0s[i:].copy(), t1s[i:].copy()
            if not (tstart < t0s[0]):
                t0s[0] = tstart
        if tstop is not None:
            i = np.searchsorted(t1s, tstop)
            t0s, t1s = t0s[:i].copy(), t1s[:i].copy()
            if tstop < t0s[-1]:
                t0s, t1s = t0s[:-1], t1s[:-1]
            else:
                t1s[-1] = tstop
        gap = (t0s[1:] - t1s[:-1]) > max_interval
        starts = np.concatenate([[t0s[0]], t0s[1:][gap]])
        stops = np.concatenate([t1s[:-1][gap], [t1s[-1]]])
        return starts, stops

    def get_cells_from_time_intervals(self, tstarts, tstops, randomize=False, seed=None):
        """Cells for given start / stop times (reference core.py:2208-2270).  With
        randomize the photons of the intervals are dealt out again in proportion
        to the exposures, with new uniform times (numpy's default_rng stream,
        generated on the host, so that the draw is the reference's)."""
        tstarts_h = np.asarray(_np(tstarts), dtype=np.float64)
        tstops_h = np.asarray(_np(tstops), dtype=np.float64)
        dev = self.device
        t0d, t1d = _to_dev(tstarts_h, dev), _to_dev(tstops_h, dev)
        exp = self.get_exposure(t1d)[1] - self.get_exposure(t0d)[1]
        lo = ops.search_sorted(self.ti, t0d)
        hi = ops.search_sorted(self.ti, t1d)
        bexp = exp * (self.B / self.S)          # Cell.SonB = S / B for every cell
        if not randomize:
            return CellBatch(self.we, self.ti, lo, hi, t0d, t1d, exp, bexp)
        rng = np.random.default_rng(seed)
        lo_h, hi_h = _np(lo), _np(hi)
        nphot = int(np.sum(hi_h) - np.sum(lo_h))
        ncell = len(tstarts_h)
        src = np.concatenate([np.arange(a, b) for a, b in zip(lo_h, hi_h)] or [np.empty(0, np.int64)])
        we = ops.gather(self.we, _to_dev(src.astype(np.int64), dev, I64))
        cexp = np.cumsum(_np(exp))
        cexp /= cexp[-1]
        new_idx = ops.search_sorted(_to_dev(cexp, dev), _to_dev(rng.random(nphot), dev))
        order = torch.sort(new_idx, stable=True)[1]
        new_idx = new_idx[order]
        we = we[order].contiguous()
        dts = (t1d - t0d)[new_idx]
        ti = t0d[new_idx] + _to_dev(rng.random(nphot), dev) * dts
        counts = torch.bincount(new_idx, minlength=ncell)
        off = torch.cat([torch.zeros(1, dtype=I64, device=dev), torch.cumsum(counts, 0)])
        return CellBatch(we, ti.contiguous(), off[:-1].contiguous(), off[1:].contiguous(), t0d, t1d, exp, bexp)

    def get_contiguous_exposure_cells(self, tstart=None, tstop=None, randomize=False, seed=None):
        """One cell per stretch of uninterrupted exposure (reference core.py:2272-2280)."""
        starts, stops = self.get_contiguous_exposures(tstart=tstart, tstop=tstop)
        return self.get_cells_from_time_intervals(starts, stops, randomize=randomize, seed=seed)

    def get_photon_cells(self, tstart, tstop, max_interval=30):
        """One cell per photon (reference core.py:1862-1956): the cell boundaries lie
        half way IN EXPOSURE between neighbouring photons, and a gap in the exposure
        table longer than max_interval seconds ends the cell before it and starts the
        next one after it.  Exposures, their midpoints and the inversion back to times
        run on the device; the fix-up at the (few) gaps is done on the host."""
        dev = self.device
        tstart, tstop = float(tstart), float(tstop)
        if tstart < self._t0:
            raise ValueError('Start time precedes start of exposure!')
        if tstop > self._t1:
            raise ValueError('Stop time exceeds stop of exposure!')
        span = _to_dev(np.array([tstart, tstop]), dev)
        e0, e1 = (int(x) for x in ops.search_sorted(self.ti, span).cpu())
        ti = self.ti[e0:e1].contiguous()
        we = self.we[e0:e1].contiguous()
        nph = e1 - e0
        print('have %d photons' % (nph))

        # snap tstart/stop to edges of exposure
        i0, i1 = (int(x) for x in ops.search_sorted(self.TSTOP, span).cpu())
        tstart = max(tstart, float(self.TSTART[i0].cpu()))
        if tstop < float(self.TSTART[i1].cpu()):
            i1 -= 1
        tstop = min(tstop, float(self.TSTOP[i1].cpu()))

        # exposure at start, photon times and stop; its midpoints, inverted to times
        times = torch.cat([_to_dev(np.array([tstart]), dev), ti, _to_dev(np.array([tstop]), dev)])
        exposure = self.get_exposure(times)[0]
        mp_exposure = 0.5 * (exposure[1:] + exposure[:-1])
        n_ft2 = self.TSTART.numel()
        c = self._exposure_table[:32 * n_ft2].view(F64).view(n_ft2, 4)[:, 0].contiguous()
        idx = ops.search_sorted(c, mp_exposure)
        frac = 1 - (c[idx] - mp_exposure) / self.exposure[idx]
        t0r = self.TSTART[idx]
        mp_times = t0r + frac * (self.TSTOP[idx] - t0r)

        cell_starts = mp_times[:-1].cpu().numpy().copy()
        cell_starts[0] = tstart
        cell_stops = mp_times[1:].cpu().numpy().copy()
        cell_stops[-1] = tstop

        # gaps in the exposure rows between the snapped start and stop
        t0s = self.TSTART[i0:i1 + 1].cpu().numpy()
        t1s = self.TSTOP[i0:i1 + 1].cpu().numpy()
        is_gap = (t0s[1:] - t1s[:-1]) > max_interval
        gap_begin, gap_end = t1s[:-1][is_gap], t0s[1:][is_gap]
        ti_h = ti.cpu().numpy()
        k = np.searchsorted(cell_stops, gap_end)       # the cell each gap falls into
        # photon after the gap: the previous cell ends where the gap begins and this one
        # starts where it ends; photon before the gap: this cell ends at the gap, the
        # next one starts after it (numpy's assignment order decides repeated cells)
        after = ((ti_h[k] - gap_end) > 0) | (k == nph - 1)
        cell_stops[k[after] - 1] = gap_begin[after]
        cell_starts[k[after]] = gap_end[after]
        before = ~after
        cell_stops[k[before]] = gap_begin[before]
        cell_starts[k[before] + 1] = gap_end[before]

        starts, stops = _to_dev(cell_starts, dev), _to_dev(cell_stops, dev)
        exp = self.get_exposure(stops)[0] - self.get_exposure(starts)[0]
        sexp = exp * (self.S / self.E)
        bexp = sexp / (self.S / self.B)
        lo = torch.arange(nph, dtype=I64, device=dev)
        return CellBatch(we, ti, lo, lo + 1, starts, stops, sexp, bexp)

    def _clip_span(self, tstart, tstop, quiet):
        """The requested span as METs, clipped to the exposure table (reference
        core.py:2040-2057)."""
        if tstart is None:
            tstart = self._t0
        tstart = float(infer_met(tstart))
        if tstart < self._t0:
            if not quiet:
                print('Warning: Start time precedes start of exposure.')
                print('Will clip to MET=%.2f.' % (self._t0))
            tstart = self._t0
        if tstop is None:
            tstop = self._t1
        tstop = float(infer_met(tstop))
        if tstop > self._t1:
            if not quiet:
                print('Warning: Stop time follows end of exposure.')
                print('Will clip to MET=%.2f.' % (self._t1))
            tstop = self._t1
        return tstart, tstop

    def _uniform_cells(self, tstart, tstop, tcell, quiet):
        """(first edge, last time, cell length, number of cells) of get_cells without
        barycentring (reference core.py:2064-2071): whole cells only, one cell when the span
        is shorter than a cell."""
        tstart, tstop = self._clip_span(tstart, tstop, quiet)
        ncell = int((tstop - tstart) / tcell)
        if ncell == 0:
            tcell = tstop - tstart
            ncell = 1
        assert(tstart + ncell * tcell <= tstop)
        return tstart, tstop, tcell, ncell

    def get_cells(self, tstart=None, tstop=None, tcell=None, trim_zero_exposure=True,
                  time_series_only=False, use_barycenter=True, randomize=False, seed=None,
                  src_scaler=None, bkg_scaler=None, src_injector=None, minimum_exposure=3e4,
                  minimum_fractional_exposure=0, quiet=False):
        if use_barycenter and not (hasattr(self, '_t2b') and hasattr(self, '_b2t')):
            raise NotImplementedError(
                'use_barycenter=True needs the time converters that the reference builds from '
                'its astropy ephemeris (bary.py); hand them in with '
                'set_barycenter_converters(topo2bary, bary2topo), or pass use_barycenter=False')
        ft1_is_bary = self.timeref == 'SOLARSYSTEM'
        tstart, tstop = self._clip_span(tstart, tstop, quiet)

        if use_barycenter:
            tstart, tstop = (float(x) for x in self._topo2bary([tstart, tstop]))

        if tcell is None:
            return self.get_photon_cells(tstart, tstop)

        ncell = int((tstop - tstart) / tcell)
        if ncell == 0:
            tcell = tstop - tstart
            ncell = 1
        if use_barycenter:
            edges_h = tstart + np.arange(ncell + 1) * tcell
            assert(edges_h[-1] <= tstop)
            # uniform cells at the barycentre: nonuniform edges at the observatory, which
            # is where exposure (and, for topocentric photon times, the binning) is done
            topo_h = np.array(self._bary2topo(edges_h), dtype=np.float64)
            assert(abs(topo_h[0] - self._t0) < 1e-6)
            topo_h[0] = self._t0   # fix tiny numerical error
            assert(topo_h[-1] <= self._t1)
            bary_h = edges_h
            self._topo_edges_v = topo_h.copy()
        else:
            # tstart + arange(ncell + 1) * tcell (core.py:2071) formed on the device: the same
            # two roundings per edge (i * tcell, then + tstart), so the same bits, without
            # half a million edges made on the host and uploaded
            assert(tstart + ncell * tcell <= tstop)
            topo_h = bary_h = torch.arange(ncell + 1, dtype=F64, device=self.device) * tcell + tstart
            self._topo_edges_v = topo_h

        # apply scales if provided (reverted before returning, as the reference does)
        ov = self.apply_scalers(src_scaler=src_scaler, bkg_scaler=bkg_scaler)
        try:
            return self._cells_on_edges(topo_h, bary_h, ft1_is_bary, use_barycenter, trim_zero_exposure,
                                        time_series_only, randomize, seed, src_injector,
                                        minimum_fractional_exposure, first_edge_dev=tstart)
        finally:
            self.revert_scalers(ov)

    @property
    def _topo_edges(self):
        """Topocentric cell edges of the last get_cells call, as a host array (the reference
        keeps them under this name, core.py:2081)."""
        v = self.__dict__.get('_topo_edges_v')
        if v is None:
            raise AttributeError('_topo_edges')
        if isinstance(v, torch.Tensor):
            v = self._topo_edges_v = _np(v)
        return v

    def set_barycenter_converters(self, topo2bary, bary2topo):
        """Callables mapping arrays of topocentric METs to barycentric ones and back: what
        the reference's get_topo2bary_interpolator / get_bary2topo_interpolator return
        (core.py:1784-1790) and caches under these same attribute names."""
        self._t2b = topo2bary
        self._b2t = bary2topo

    def _topo2bary(self, topo_times):
        return np.asarray(self._t2b(np.asarray(topo_times, dtype=np.float64)), dtype=np.float64)

    def _bary2topo(self, bary_times):
        return np.asarray(self._b2t(np.asarray(bary_times, dtype=np.float64)), dtype=np.float64)

    def _cells_on_edges(self, topo_h, bary_h, ft1_is_bary, use_barycenter, trim_zero_exposure,
                        time_series_only, randomize, seed, src_injector, minimum_fractional_exposure,
                        first_edge_dev=None):
        """The array tail of get_cells (reference core.py:2087-2206).  topo_h: cell edges
        at the observatory (exposure); bary_h: the same edges at the barycentre (equal to
        topo_h without barycentring).  Photons are binned on the edges of their own time
        system; the cells are labelled with the barycentric edges when barycentring."""
        dev = self.device
        on_device = isinstance(topo_h, torch.Tensor)      # uniform edges made on the device
        topo = topo_h if on_device else torch.from_numpy(np.ascontiguousarray(topo_h)).to(dev)
        # always use topocentric times to manage the exposure calculation
        exp, sexp, bexp, dead = self.get_exposure(topo, deadtime_too=True, cumulative=False)
        edges_h = bary_h if ft1_is_bary else topo_h
        edges = topo if edges_h is topo_h else torch.from_numpy(np.ascontiguousarray(edges_h)).to(dev)
        first_edge = first_edge_dev if on_device else float(edges_h[0])
        starts, stops = edges[:-1].contiguous(), edges[1:].contiguous()
        relabel = use_barycenter and not ft1_is_bary
        if relabel or use_barycenter:
            bary = torch.from_numpy(np.ascontiguousarray(bary_h)).to(dev)
            label_starts, label_stops = bary[:-1].contiguous(), bary[1:].contiguous()
            alt_starts, alt_stops = topo[:-1].contiguous(), topo[1:].contiguous()

        if minimum_fractional_exposure > 0:
            frac = exp / exp[exp > 0].mean()
            m = frac < minimum_fractional_exposure
            exp[m] = 0
            sexp[m] = 0
            bexp[m] = 0

        lo, hi, counts = ops.bin_photons(self.ti, first_edge, stops)

        if trim_zero_exposure:
            keep = ops.select_cells(exp, 0.0)
            cols = [starts, stops, exp, sexp, bexp, dead, lo, hi, counts]
            if use_barycenter:
                cols += [label_starts, label_stops, alt_starts, alt_stops]
            cols = ops.gather_columns([c.contiguous() for c in cols], keep)     # one launch for the table
            starts, stops, exp, sexp, bexp, dead, lo, hi, counts = cols[:9]
            if use_barycenter:
                label_starts, label_stops, alt_starts, alt_stops = cols[9:]

        we, ti = self.we, self.ti
        if src_injector is not None:
            randomize = True
        if randomize:
            # Deal the photons out again following the (possibly rescaled) source and
            # background exposures (reference core.py:2129-2156).  The two uniform draws
            # come from numpy's default_rng on the host so that they are the reference's;
            # inside a cell the photons keep their time order (numpy's unstable argsort
            # leaves them in another order: sums over a cell agree to rounding).
            dev = self.device
            n_all = int(counts.sum().cpu())
            src = torch.repeat_interleave(lo - torch.cumsum(counts, 0) + counts, counts) + \
                torch.arange(n_all, dtype=I64, device=dev)
            weights = ops.gather(we, src)
            times = ops.gather(ti, src)
            sexp_h, bexp_h = _np(sexp), _np(bexp)
            if src_injector is not None:
                # the injector is handed ALL barycentric edges, as in the reference (2135)
                bh = _np(bary_h) if on_device else bary_h
                csexp = np.cumsum(src_injector(bh[:-1], bh[1:]) * sexp_h)
            else:
                csexp = np.cumsum(sexp_h)
            csexp *= 1. / csexp[-1]
            cbexp = np.cumsum(bexp_h)
            cbexp *= 1. / cbexp[-1]
            rng = np.random.default_rng(seed)
            is_bkg = _to_dev(rng.random(n_all), dev) >= weights
            rv = _to_dev(rng.random(n_all), dev)
            idx = torch.where(is_bkg, ops.search_sorted(_to_dev(cbexp, dev), rv),
                              ops.search_sorted(_to_dev(csexp, dev), rv))
            order = torch.sort(idx, stable=True)[1]
            ti, we = times[order].contiguous(), weights[order].contiguous()
            counts = torch.bincount(idx, minlength=starts.numel())
            if bool((exp[counts > 0] <= 0).any().cpu()):
                raise AssertionError('photons were dealt to cells without exposure')
            off = torch.cat([torch.zeros(1, dtype=I64, device=dev), torch.cumsum(counts, 0)])
            lo, hi = off[:-1].contiguous(), off[1:].contiguous()

        # exposure and event selection are done; with barycentring the cells carry the
        # (uniform) barycentric edges
        if relabel:
            starts, stops = label_starts, label_stops

        if time_series_only:
            w1, w2 = ops.cell_moments(we, lo, hi)
            if use_barycenter:
                return CellTimeSeries(starts, stops, exp, sexp, bexp, counts, w1, w2, dead,
                                      alt_starts=alt_starts, alt_stops=alt_stops,
                                      timesys='barycenter', minimum_exposure=0)
            return CellTimeSeries(starts, stops, exp, sexp, bexp, counts, w1, w2, dead,
                                  timesys='topocentric', minimum_exposure=0)
        return CellBatch(we, ti, lo, hi, starts, stops, sexp, bexp)


class SourceBatch(object):
    """The cells of many sources as ONE device-resident list (multi-source batches: a
    thousand sky positions, each with its own photon list, exposure table and S, B, E;
    per source this is what Data.get_cells(...) returns, reference core.py:1958-2206).

    cells         CellBatch over the concatenated photon lists (cells of source 0 first)
    source        int64 device tensor, the source of every cell
    cell_offsets  host array [nsource + 1]: cells[cell_offsets[k]:cell_offsets[k + 1]]
                  are the cells of source k, in time order

    One fit launch serves all sources: ops.cell_fit(batch.cells.ranges()), or
    CellsLogLikelihood(batch.cells) for profiles and light curves; split(rows) cuts a
    per-cell result into the per-source pieces."""

    def __init__(self, cells, source, cell_offsets):
        self.cells, self.source, self.cell_offsets = cells, source, cell_offsets

    def __len__(self):
        return len(self.cell_offsets) - 1

    def split(self, rows):
        """Per-source pieces of an array or tensor with one entry (or row) per cell."""
        o = self.cell_offsets
        return [rows[int(o[k]):int(o[k + 1])] for k in range(len(o) - 1)]


def build_exposure_tables(sources, nstream=4):
    """The running-sum tables of many Data objects (what the first get_exposure / get_cells of
    each would build), issued over `nstream` HIP streams.  The exact running sums are a chain of
    six launches per source, two of them a handful of workgroups that walk the table in order
    while the rest of the chip idles; tables of different sources do not depend on one another,
    so the walks of some overlap the streaming launches of others.  Same tables, bit for bit,
    as one after another.  Returns `sources`."""
    todo = [d for d in sources if d._exposure_table is None]
    if not todo:
        return sources
    dev = todo[0].device
    here = torch.cuda.current_stream(dev)
    streams = [torch.cuda.Stream(dev) for _ in range(max(1, min(int(nstream), len(todo))))]
    for st in streams:
        st.wait_stream(here)
    for k, d in enumerate(todo):
        with torch.cuda.stream(streams[k % len(streams)]):
            d._plain_exposure_table().record_stream(here)
    for st in streams:
        here.wait_stream(st)
    return sources


def build_exposure_checkpoints(sources, chunk=250):
    """Compact running sums (ops.exposure_checkpoints: every sixteenth row, 2 bytes a row) for
    the Data objects that have neither them nor a full table yet, `chunk` sources to a handful
    of launches.  get_cells_batch and get_timeseries_batch read cells from them with the same
    bits as from the full tables (80 bytes a row, which a thousand sources over one spacecraft
    file cannot afford).  Returns `sources`."""
    todo = [d for d in sources if d._checkpoints is None and d._exposure_table is None]
    for at in range(0, len(todo), max(1, int(chunk))):
        part = todo[at:at + max(1, int(chunk))]
        built = ops.exposure_checkpoints([dict(tstart=d.TSTART, tstop=d.TSTOP, exposure=d.exposure,
                                               livetime=d.LIVETIME, s_over_e=d.S / d.E, b_over_e=d.B / d.E)
                                          for d in part])
        for d, ck in zip(part, built):
            d._checkpoints = ck
    return sources


# full tables still to build beyond this many bytes: the compact ones are built instead
_FULL_TABLES_LIMIT = 4 << 30


def _batch_tables(sources, compact_tables):
    """Builds the exposure tables the sources still lack: compact ones if asked for, or (None)
    if full ones would take more than _FULL_TABLES_LIMIT bytes or some source has compact ones."""
    if compact_tables is None:
        missing = sum(80 * d.TSTART.numel() for d in sources if d._exposure_table is None)
        compact_tables = missing > _FULL_TABLES_LIMIT or any(d._checkpoints is not None for d in sources)
    if compact_tables:
        build_exposure_checkpoints(sources)
    else:
        build_exposure_tables(sources)              # over several streams
    return sources


def _batch_descriptors(sources, spans, compact_tables):
    """What ops.bin_sources needs of every source, with its exposure table built on the way: the
    full one where the source has it already (or compact_tables is False), else the compact one."""
    _batch_tables(sources, compact_tables)
    desc = []
    for d, (t0, t1, tc, nc) in zip(sources, spans):
        s = dict(tstart=d.TSTART, tstop=d.TSTOP, exposure=d.exposure, livetime=d.LIVETIME, ti=d.ti,
                 s_over_e=d.S / d.E, b_over_e=d.B / d.E, t0=t0, tcell=tc, ncell=nc)
        if d._exposure_table is not None:
            s['table'] = d._exposure_table
        else:
            s['checkpoints'] = d._checkpoints
        desc.append(s)
    return desc


def _kept_per_source(keep, ncl):
    """How many of every source's cells a selection kept: `keep` holds the ascending indices
    of the kept cells, source k owns the ncl[k] cells after those of the sources before it."""
    edges = torch.from_numpy(np.concatenate([[0], np.cumsum(ncl)]).astype(np.int64)).to(keep.device)
    return np.diff(torch.searchsorted(keep, edges).cpu().numpy())


def _joined(parts):
    """The arrays of `parts` one after another as ONE tensor: a view when they already lie
    like that in one allocation (pack_sources puts them there), else a copy."""
    full = [t for t in parts if t.numel() > 0]
    if not full:
        return torch.cat(parts)
    first = full[0]
    store = first.untyped_storage()
    at = first.data_ptr()
    for t in full:
        if not t.is_contiguous() or t.dtype != first.dtype or t.data_ptr() != at \
                or t.untyped_storage().data_ptr() != store.data_ptr():
            return torch.cat(parts)
        at += t.numel() * t.element_size()
    total = sum(t.numel() for t in full)
    return torch.empty(0, dtype=first.dtype, device=first.device).set_(store, first.storage_offset(), (total,))


def pack_sources(sources):
    """Move the photon lists of the Data objects into one allocation, source after source
    (every object's ti and we become views into it; values and order unchanged), so that
    get_cells_batch hands the kernels the concatenated lists without copying them at every
    call: 16 bytes a photon, a gigabyte for 64 sources of 10^6 photons.  Returns `sources`."""
    for name in ('ti', 'we'):
        parts = [getattr(d, name) for d in sources]
        if not all(t.is_contiguous() for t in parts):
            parts = [t.contiguous() for t in parts]
        whole = torch.cat(parts)
        at = 0
        for d, t in zip(sources, parts):
            setattr(d, name, whole[at:at + t.numel()])
            at += t.numel()
    return sources


def get_cells_batch(sources, tstart=None, tstop=None, tcell=None, trim_zero_exposure=True, quiet=True,
                    per_source_calls=False, compact_tables=None):
    """Data.get_cells(tstart, tstop, tcell, use_barycenter=False) for every Data object of
    `sources`, as one SourceBatch.  Every source is binned on its own edges and exposure
    table, all of them by the same two launches (ops.bin_sources); the zero-exposure cells of
    all of them are then dropped by ONE selection, so the host waits for the device once per
    batch, not once per source, and the number of launches does not grow with the number of
    sources (a thousand sources at ten launches each would be bound by issuing them).
    per_source_calls=True, and any source with scalers applied, takes the calls one source
    at a time instead: the same bits.  compact_tables: True reads the exposures from
    build_exposure_checkpoints' running sums (2 bytes a table row), False from the full tables
    (80 bytes a row), None picks the compact ones when the full tables still to build would
    take more than 4 GiB; the same bits either way."""
    if tcell is None:
        raise ValueError('get_cells_batch needs a cell length')
    if not sources:
        raise ValueError('no sources')
    dev = sources[0].device
    spans = None
    if not (per_source_calls or any(d.sexposure is not None for d in sources)):
        spans = [d._uniform_cells(tstart, tstop, tcell, quiet) for d in sources]
        if any(not tc > 0 or nc < 1 for _, _, tc, nc in spans):
            spans = None         # a span that ends before it begins: whatever get_cells makes of it
    if spans is None:
        # one source after another, each trimmed by its own call (on its exposure, as below)
        per = [d.get_cells(tstart=tstart, tstop=tstop, tcell=tcell, trim_zero_exposure=trim_zero_exposure,
                           use_barycenter=False, quiet=quiet) for d in sources]
        nph = np.array([b.we.numel() for b in per], dtype=np.int64)
        ncl = np.array([len(b) for b in per], dtype=np.int64)
        poff = np.concatenate([[0], np.cumsum(nph)])
        cat = lambda name: torch.cat([getattr(b, name) for b in per])
        shift = torch.repeat_interleave(torch.from_numpy(poff[:-1]).to(dev), torch.from_numpy(ncl).to(dev))
        lo, hi = cat('lo') + shift, cat('hi') + shift
        starts, stops, sexp, bexp = cat('starts'), cat('stops'), cat('sexp'), cat('bexp')
        source = torch.repeat_interleave(torch.arange(len(per), dtype=I64, device=dev), torch.from_numpy(ncl).to(dev))
        we, ti = _joined([d.we for d in sources]), _joined([d.ti for d in sources])
    else:
        desc = _batch_descriptors(sources, spans, compact_tables)
        starts, stops, exp, sexp, bexp, _, lo, hi, source = ops.bin_sources(desc)
        ncl = np.array([s['ncell'] for s in desc], dtype=np.int64)
        we, ti = _joined([d.we for d in sources]), _joined([d.ti for d in sources])
        if trim_zero_exposure:
            # on the exposure itself, as the reference does (core.py:2110-2122): exposure and source
            # exposure are differences of two separately rounded running sums, so that beside a gap
            # one of them can be zero and the other an ulp (and with S = 0 every sexp is zero)
            keep = ops.select_cells(exp, 0.0)                      # the one wait of the batch
            lo, hi, starts, stops, sexp, bexp, source = ops.gather_columns(
                [c.contiguous() for c in (lo, hi, starts, stops, sexp, bexp, source)], keep)
            ncl = _kept_per_source(keep, ncl)
    cells = CellBatch(we, ti, lo.contiguous(), hi.contiguous(), starts, stops, sexp, bexp)
    return SourceBatch(cells, source, np.concatenate([[0], np.cumsum(ncl)]).astype(np.int64))


def _timeseries_columns(sources, tstart, tstop, tcell, trim_zero_exposure, quiet, compact_tables):
    """The nine columns of CellTimeSeries for all sources side by side, the offsets of the sources
    in them and the spans binned; None where the batched binning does not apply."""
    spans = None
    if not any(d.sexposure is not None for d in sources):
        spans = [d._uniform_cells(tstart, tstop, tcell, quiet) for d in sources]
        if any(not tc > 0 or nc < 1 for _, _, tc, nc in spans):
            spans = None         # a span that ends before it begins: whatever get_cells makes of it
    if spans is None:
        return None
    desc = _batch_descriptors(sources, spans, compact_tables)
    starts, stops, exp, sexp, bexp, dead, lo, hi, source = ops.bin_sources(desc, deadtime_too=True)
    we = _joined([d.we for d in sources])
    w1, w2 = ops.cell_moments(we, lo, hi)
    counts = hi - lo
    ncl = np.array([s['ncell'] for s in desc], dtype=np.int64)
    cols = [starts, stops, exp, sexp, bexp, counts, w1, w2, dead]
    if trim_zero_exposure:
        keep = ops.select_cells(exp, 0.0)                      # the one wait of the batch
        cols = ops.gather_columns([c.contiguous() for c in cols], keep)
        ncl = _kept_per_source(keep, ncl)
    else:
        # what the constructor does to cells without exposure (minimum_exposure = 0), once for all
        mask = ~(cols[2] > 0)
        for c in cols[2:8]:
            c[mask] = 0
    return cols, np.concatenate([[0], np.cumsum(ncl)]).astype(np.int64), spans


def _span_sampling(span):
    """tsamp and tspan of an untrimmed series on the edges i * tcell + t0, rounded as the kernels
    (and numpy) round them: what CellTimeSeries.tsamp() / tspan() read off the device."""
    t0, _, tc, nc = span
    return (1.0 * tc + t0) - (0.0 * tc + t0), (float(nc) * tc + t0) - (0.0 * tc + t0)


def get_timeseries_batch(sources, tstart=None, tstop=None, tcell=None, trim_zero_exposure=True, quiet=True,
                         compact_tables=None):
    """Data.get_cells(tstart, tstop, tcell, time_series_only=True, use_barycenter=False) for
    every Data object of `sources` (reference core.py:1958-2206 with the return of 2185-2192):
    a list of CellTimeSeries, one per source, the same bits as the per-source calls.  The
    edges, exposures, dead times and photon ranges of all sources come from the two launches
    of ops.bin_sources, the weight moments of all cells from one more, the trimming from one
    selection: the input of a power spectrum per source (power_spectrum_fft) without a
    dozen small launches per source.  compact_tables as for get_cells_batch."""
    if tcell is None:
        raise ValueError('get_timeseries_batch needs a cell length')
    if not sources:
        raise ValueError('no sources')
    made = _timeseries_columns(sources, tstart, tstop, tcell, trim_zero_exposure, quiet, compact_tables)
    if made is None:
        return [d.get_cells(tstart=tstart, tstop=tstop, tcell=tcell, trim_zero_exposure=trim_zero_exposure,
                            time_series_only=True, use_barycenter=False, quiet=quiet) for d in sources]
    cols, off, spans = made
    out = []
    for k in range(len(sources)):
        ts = object.__new__(CellTimeSeries)
        for name, c in zip(CellTimeSeries._KEYS, cols):
            setattr(ts, name, c[int(off[k]):int(off[k + 1])])
        ts.alt_starts = ts.alt_stops = None
        ts.timesys = 'topocentric'
        ts.minimum_exposure = 0
        if not trim_zero_exposure:
            # untrimmed: the first and last edges are those of the span; no device read for
            # tsamp() / tspan()
            ts._tsamp, ts._tspan = _span_sampling(spans[k])
        out.append(ts)
    return out


class SourceResults(object):
    """What run_sources returns.  batch: the SourceBatch of the fitted cells; alpha, ts, flags:
    ops.cell_fit's per-cell results for all sources (batch.split cuts them per source);
    peak_power[k, j], peak_freq[k, j]: the largest value of source k's spectrum j (P_0, P_1,
    P_b of power_spectrum_fft) above zero frequency and the frequency [Hz] it stands at;
    spectra: per source (l, tsamp, P_0, P_1, P_b) with device tensors when asked for, else
    None."""

    def __init__(self, batch, alpha, ts, flags, peak_power, peak_freq, spectra):
        self.batch, self.alpha, self.ts, self.flags = batch, alpha, ts, flags
        self.peak_power, self.peak_freq, self.spectra = peak_power, peak_freq, spectra


def run_sources(sources, tcell_fit, tcell_ps, tstart=None, tstop=None, chunk=32, dfgoal=None,
                keep_spectra=False, compact_tables=None, timings=None, nstream=3):
    """Many sources as one job: for every Data object the cells of length tcell_fit fitted
    (Data.get_cells + the per-cell maximisation of CellsLogLikelihood, reference
    core.py:1958, 1070) and the power spectra of its time series of tcell_ps bins
    (get_cells(time_series_only=True) + power_spectrum_fft, core.py:2598).  The exposure
    tables of all sources are built by a handful of launches, the cells of all sources are
    binned by two and fitted by one; the time series are binned `chunk` sources at a time
    (a source's series at 300-s bins is 10^6 bins of nine columns) and transformed one source
    after another, dealt over `nstream` HIP streams: a spectrum is half a dozen launches of
    40-110 us, each of which ramps up and drains on its own, and the launches of the next
    source's spectrum fill what one leaves idle.  The host reads results back once, at the end.
    `timings`: a dict that receives the seconds spent per stage (the device is then waited for
    after every stage)."""
    if not sources:
        raise ValueError('no sources')
    dev = sources[0].device
    clock = [time.perf_counter()]

    def lap(stage):
        if timings is not None:
            torch.cuda.synchronize(dev)
            now = time.perf_counter()
            timings[stage] = timings.get(stage, 0.0) + now - clock[0]
            clock[0] = now

    _batch_tables(sources, compact_tables)
    lap('exposure_tables')
    batch = get_cells_batch(sources, tstart=tstart, tstop=tstop, tcell=tcell_fit, compact_tables=compact_tables)
    lap('cells')
    alpha, ts_, flags = ops.cell_fit(batch.cells.ranges())
    lap('fit')
    peaks, df, spectra = [], [], ([] if keep_spectra else None)
    chunk = max(1, int(chunk))
    here = torch.cuda.current_stream(dev)
    side = [torch.cuda.Stream(dev) for _ in range(max(1, int(nstream)))] if int(nstream) > 1 else [here]
    work = [None] * len(side)                    # a transform workspace per stream, kept over the job

    def fork():
        for st in side:
            if st is not here:
                st.wait_stream(here)

    def join():
        for st in side:
            if st is not here:
                here.wait_stream(st)

    for at in range(0, len(sources), chunk):
        part = sources[at:at + chunk]
        spans = None
        if not any(d.sexposure is not None for d in part):
            spans = [d._uniform_cells(tstart, tstop, tcell_ps, True) for d in part]
            if any(not tc > 0 or nc < 1 for _, _, tc, nc in spans):
                spans = None
        if spans is None:
            # sources the batched binning does not take: one call after the other
            for d in part:
                ser = d.get_cells(tstart=tstart, tstop=tstop, tcell=tcell_ps, trim_zero_exposure=False,
                                  time_series_only=True, use_barycenter=False, quiet=True)
                l, tsamp, p0, p1, pb, _ = _power_spectrum_parts(ser, dfgoal=dfgoal)
                v = [torch.max(p[1:], 0) for p in (p0, p1, pb)]
                peaks.append(torch.stack([torch.stack([m.values for m in v]),
                                          torch.stack([m.indices for m in v]).to(F64)]).unsqueeze(0))
                df.append((1.0 / l, tsamp))
                if keep_spectra:
                    spectra.append((l, tsamp, p0, p1, pb))
            lap('spectra')
            continue
        # the bins of all sources of the chunk from three launches, the six series that
        # power_spectrum_fft transforms (core.py:2178-2183, 241-244, 2654-2661) from one more
        desc = _batch_descriptors(part, spans, compact_tables)
        _, _, exp, sexp, bexp, _, lo, hi, _ = ops.bin_sources(desc, want_edges=False)
        W, WW, Wb, WbWb = ops.cell_series(_joined([d.we for d in part]), lo, hi, exp, sexp, bexp)
        off = np.concatenate([[0], np.cumsum([sp[3] for sp in spans])]).astype(np.int64)
        del exp, lo, hi
        lap('time_series')
        # padded length of every source's transform, as power_spectrum_fft chooses it (core.py:2667-2671)
        ls, samp = [], []
        for sp in spans:
            tsamp, tspan = _span_sampling(sp)
            goal = 0.2 / tspan if dfgoal is None else dfgoal
            ls.append(2 ** (int(np.log2(1. / (tsamp * goal))) + 1))
            samp.append(tsamp)
        for k, l in enumerate(ls):
            if l < int(off[k + 1] - off[k]):
                raise NotImplementedError('dfgoal gives a transform shorter than the series')
        same = all(l == ls[0] for l in ls)
        buf = torch.empty((len(part), 3, ls[0] // 4 + 1), dtype=F64, device=dev) if same else None
        fork()
        for k in range(len(part)):
            j = k % len(side)
            st = side[j]
            c0, c1 = int(off[k]), int(off[k + 1])
            with torch.cuda.stream(st):
                if work[j] is None or work[j].numel() < ops._lib.lib.godot_power_spectrum_workspace_bytes(ls[k]):
                    work[j] = ops.power_spectrum_workspace(ls[k], dev)
                out = (buf[k, 0], buf[k, 1], buf[k, 2]) if same else None
                p0, p1, pb, _ = ops.power_spectrum_fft(W[c0:c1], WW[c0:c1], sexp[c0:c1], Wb[c0:c1], WbWb[c0:c1],
                                                       bexp[c0:c1], ls[k], out=out, workspace=work[j])
                if not same:
                    v = [torch.max(p[1:], 0) for p in (p0, p1, pb)]
                    peaks.append(torch.stack([torch.stack([m.values for m in v]),
                                              torch.stack([m.indices for m in v]).to(F64)]).unsqueeze(0))
            df.append((1.0 / ls[k], samp[k]))
            if keep_spectra:
                spectra.append((ls[k], samp[k], p0, p1, pb))
        join()
        if same:
            # one reduction for the chunk: the peak of every spectrum beyond its zero-frequency bin
            mv, mi = ops.row_peaks(buf.view(len(part) * 3, -1), first=1)
            peaks.append(torch.stack([mv.view(len(part), 3), mi.view(len(part), 3).to(F64)], 1))
        # (what the side streams read and wrote belongs to this stream's allocator; join() made
        # this stream wait for them, so it may be handed out again from here on)
        del W, WW, Wb, WbWb, sexp, bexp, buf
        lap('spectra')
    pk = torch.cat(peaks).cpu().numpy()           # the one read of the job
    lap('read_back')
    df = np.array(df)
    # bin i stands at (i * (1 / l)) / tsamp, as power_spectrum_fft forms its frequencies
    freq = ((pk[:, 1] + 1) * df[:, :1]) / df[:, 1:]
    return SourceResults(batch, alpha, ts_, flags, pk[:, 0], freq, spectra)


class PhaseData(object):
    """Pulse phases instead of times, exposure ignored (reference core.py:2319-2386).
    Constructed from arrays of phases in [0, 1) and weights (the reference reads
    them from FT1 files)."""

    def __init__(self, phases, weights, phase_shift=None, device=None):
        dev = device or default_device()
        ph = np.asarray(_np(phases), dtype=np.float64).copy()
        we = np.asarray(_np(weights), dtype=np.float64)
        if phase_shift is not None:
            ph += phase_shift
            np.mod(ph, 1, out=ph)
        a = np.argsort(ph)
        self.device = dev
        self.ti = _to_dev(ph[a], dev)
        self.we = _to_dev(we[a], dev)
        self.E = 1
        self.S = float(np.sum(we[a]))
        self.B = len(we) - self.S

    def get_cells(self, ncell=100, get_converse=False, randomize=False, seed=None):
        dev = self.device
        edges = np.linspace(0, 1, ncell + 1)
        starts, stops = _to_dev(edges[:-1], dev), _to_dev(edges[1:], dev)
        times, weights = self.ti, self.we
        if get_converse:
            weights = 1 - weights
        if randomize:
            # new uniform phases from numpy's default_rng (host), sorted; no ties among them
            times, order = torch.sort(_to_dev(np.random.default_rng(seed).random(times.numel()), dev))
            weights = weights[order].contiguous()
        idx = ops.search_sorted(stops, times)
        counts = torch.bincount(idx, minlength=ncell)[:ncell]
        off = torch.cat([torch.zeros(1, dtype=I64, device=dev), torch.cumsum(counts, 0)])
        sexp = torch.full((ncell,), self.S / ncell, dtype=F64, device=dev)
        bexp = sexp / (self.S / self.B)
        return CellBatch(weights.contiguous(), times.contiguous(), off[:-1].contiguous(), off[1:].contiguous(),
                         starts, stops, sexp, bexp, phase=True)


def bb_prior_tune(data, tcell, bb_priors=[2, 3, 4, 5, 6, 7, 8, 9, 10], ntrial=10, orbital=False,
                  **cell_kwargs):
    """Number of Bayesian blocks found in randomised realisations of the data for a
    range of prior parameters (reference core.py:2784-2817).  Returns
    (counts [len(bb_priors), ntrial], number of cells)."""
    rvals = np.empty((len(bb_priors), ntrial), dtype=int)
    for itrial in range(ntrial):
        if orbital:
            cells = data.get_contiguous_exposure_cells(randomize=True, **cell_kwargs)
        elif isinstance(data, PhaseData):
            cells = data.get_cells(tcell, randomize=True, **cell_kwargs)
        else:
            cells = data.get_cells(tcell=tcell, randomize=True, **cell_kwargs)
        cll = CellsLogLikelihood(cells)
        for iprior, prior in enumerate(bb_priors):
            rvals[iprior, itrial] = len(cll.do_bb(prior=prior)[0])
    return rvals, len(cells)


# ---------------------------------------------------------------------------
# Power spectra (reference core.py:2598-2782) and trig sums (2426-2596)
# ---------------------------------------------------------------------------
def _ts_dev(ts, name, dev):
    return _to_dev(getattr(ts, name), dev)


def _power_spectrum_parts(timeseries, dfgoal=None, tweak_exp=False, exp_only=False, get_amps=False,
                          exposure_correction=None, no_zero_pad=False, maxN=None, device=None):
    """power_spectrum_fft without the trip to the host: (l, tsamp, P_0, P_1, P_b, amplitudes)
    with the spectra as device tensors of l // 4 + 1 bins."""
    cells = timeseries
    dev = device or (cells.weights.device if isinstance(cells.weights, torch.Tensor)
                     else default_device())
    W = _ts_dev(cells, 'weights', dev)
    WW = _ts_dev(cells, 'weights2', dev)
    S = _ts_dev(cells, 'sexp', dev)
    counts = _to_dev(cells.counts, dev, F64) if not (isinstance(cells.counts, torch.Tensor)
                                                    and cells.counts.dtype == F64) \
        else cells.counts.to(dev)
    if exposure_correction is not None:
        S = S * _to_dev(exposure_correction, dev)
    if tweak_exp:
        S = S * W.sum() / S.sum()
    Wb = counts - W
    WbWb = counts - 2 * W + WW
    B = _ts_dev(cells, 'bexp', dev)
    if tweak_exp:
        B = B * Wb.sum() / B.sum()
    if maxN is not None:
        W, WW, Wb, WbWb, S, B = [x[:maxN].contiguous() for x in (W, WW, Wb, WbWb, S, B)]
    if exp_only:
        W = 2 * S - S.mean()

    tsamp = cells.tsamp()
    if dfgoal is None:
        dfgoal = 0.2 / cells.tspan()
    nfft = 1. / (tsamp * dfgoal)
    l = 2 ** (int(np.log2(nfft)) + 1)
    if no_zero_pad:
        l = W.numel()
    if l < W.numel():
        raise NotImplementedError('dfgoal gives a transform shorter than the series')
    p0, p1, pb, amps = ops.power_spectrum_fft(W.contiguous(), WW.contiguous(), S.contiguous(),
                                              Wb.contiguous(), WbWb.contiguous(), B.contiguous(),
                                              l, want_amps=get_amps)
    return l, tsamp, p0, p1, pb, amps


def power_spectrum_fft(timeseries, dfgoal=None, tweak_exp=False, exp_only=False, get_amps=False,
                       exposure_correction=None, no_zero_pad=False, maxN=None, device=None):
    """FFT evaluation of the first-order likelihood power spectra.  Returns
    (frequencies [Hz], P_0, P_1, P_b) as numpy arrays, or the amplitude tuple
    of get_amps=True, exactly as the reference orders them."""
    l, tsamp, p0, p1, pb, amps = _power_spectrum_parts(
        timeseries, dfgoal=dfgoal, tweak_exp=tweak_exp, exp_only=exp_only, get_amps=get_amps,
        exposure_correction=exposure_correction, no_zero_pad=no_zero_pad, maxN=maxN, device=device)
    # np.fft.rfftfreq(l) / tsamp cut to l // 4 + 1 bins (core.py:2679) without forming the other half of
    # rfftfreq's array: the same products and quotients, element for element, formed on the device
    # (the host takes 12 ms over the 8.4e6 bins of l = 2^25, fifteen times the transforms)
    freqs = _np(ops.spectrum_frequencies(l // 4 + 1, 1.0 / l, tsamp, True, p0.device))
    if get_amps:
        a = _np(amps)
        return (freqs,) + tuple(a[i] for i in range(8))
    if exp_only:
        return freqs, _np(p0)
    return freqs, _np(p0), _np(p1), _np(pb)


def power_spectrum_simple(timeseries, get_amps=False, device=None):
    ts = timeseries
    dev = device or (ts.weights.device if isinstance(ts.weights, torch.Tensor) else default_device())
    W, WW, S = _ts_dev(ts, 'weights', dev), _ts_dev(ts, 'weights2', dev), _ts_dev(ts, 'sexp', dev)
    l = W.numel()
    dl, mom = ops.power_spectrum_simple(W, WW, S, want_moments=get_amps)
    # rfftfreq(l)[:l // 4 + 1] * (1 / tsamp), same products
    freqs = _np(ops.spectrum_frequencies(l // 4 + 1, 1.0 / l, 1. / ts.tsamp(), False, dl.device))
    if get_amps:
        m = _np(mom)
        return freqs, _np(dl), (m[0], m[1], m[2], m[3])
    return freqs, _np(dl)


def power_spectrum_dft(cll, freqs, unweighted=False):
    """First-order likelihood spectrum at arbitrary frequencies on possibly
    uneven cells (cll: a CellsLogLikelihoodOld)."""
    cr = cll._cr
    dev = cr.device
    w1, w2 = ops.cell_moments(cr.we, cr.lo, cr.hi)
    scale = 1.0
    if unweighted:
        # unit weights: both moments are the counts; exposure rescaled by nph / sum(w)
        scale = float(cll.nph) / float(w1.sum().cpu())
        w1 = (cr.hi - cr.lo).to(F64)
        w2 = w1.clone()
    out = ops.trigsum_cells(w1, w2, cll.exp.contiguous(), cll.tmids.contiguous(),
                            _to_dev(freqs, dev), exp_scale=scale)
    return _np(out)


def fit_harmonics(cll, freq):
    """Best-fitting cos/sin amplitudes and TS at one frequency (or an array of
    frequencies: then an (n, 3) array)."""
    cr = cll._cr
    dev = cr.device
    w1, w2 = ops.cell_moments(cr.we, cr.lo, cr.hi)
    f = np.atleast_1d(np.asarray(freq, dtype=np.float64))
    tmean = float(np.mean(_np(cll.tmids)))
    out = _np(ops.fit_harmonics(w1, w2, cll.exp.contiguous(), cll.tmids.contiguous(), tmean,
                                _to_dev(f, dev)))
    if np.ndim(freq) == 0:
        return out[0, 0], out[0, 1], out[0, 2]
    return out


def z2m(times, weights, f0, df, nf, m=2, j0=0, device=None):
    """Weighted Z^2_m on the grid f0 + (j0 + j) df (device tensor of nf values).
    Not part of the reference: see godot_z2m in include/godot_hip.h."""
    dev = device or (times.device if isinstance(times, torch.Tensor) and times.is_cuda
                     else default_device())
    z, _ = ops.z2m(_to_dev(times, dev), _to_dev(weights, dev), f0, df, nf, m=m, j0=j0)
    return z


def h_test(times, weights, f0, df, nf, mmax=20, j0=0, device=None):
    """Weighted H statistic max_{m<=mmax}(Z^2_m - 4m + 4) on the same grid."""
    dev = device or (times.device if isinstance(times, torch.Tensor) and times.is_cuda
                     else default_device())
    _, h = ops.z2m(_to_dev(times, dev), _to_dev(weights, dev), f0, df, nf, m=mmax, j0=j0,
                   want_z2=False, want_h=True)
    return h


def compute_ls(cll, freqs, unweighted=False):
    """Lomb-Scargle periodogram of the per-cell flux (reference core.py:2490-2531)."""
    cr = cll._cr
    dev = cr.device
    if unweighted:
        flux = (cr.hi - cr.lo).to(F64) / cll.exp ** 0.5
    else:
        w1, _ = ops.cell_moments(cr.we, cr.lo, cr.hi)
        flux = w1 - cll.exp
    flux = flux - flux.mean()
    flux = flux / flux.std(unbiased=False)
    out = ops.ls_cells(flux.contiguous(), cll.tmids.contiguous(), _to_dev(freqs, dev))
    scale = 0.5 * float((flux ** 2).sum().cpu())
    return _np(out) / scale


def get_orbital_modulation(ts, freqs, device=None):
    """Time-domain sinusoid model at the given frequencies and its powers
    (reference core.py:2828-2861); the first return value can be passed to
    power_spectrum_fft as exposure_correction."""
    dev = device or (ts.weights.device if isinstance(ts.weights, torch.Tensor) else default_device())
    starts, stops = _ts_dev(ts, 'starts', dev), _ts_dev(ts, 'stops', dev)
    W, WW, S = _ts_dev(ts, 'weights', dev), _ts_dev(ts, 'weights2', dev), _ts_dev(ts, 'sexp', dev)
    rvals = torch.ones_like(W)
    y = ((W - S) / S.mean()).contiguous()
    pows = np.empty(len(freqs))
    for i, f in enumerate(freqs):
        m = _np(ops.orbital_modulation_step(starts, stops, y, WW, float(f), rvals))
        pows[i] = m[0] ** 2 / m[2] + m[1] ** 2 / m[3]
    return _np(rvals), pows * float(S.sum().cpu()) ** 2 / S.numel()
