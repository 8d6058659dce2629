"""Tensor-level wrappers over the C ABI.

Inputs and outputs are torch tensors on a ROCm device (torch is used for
device memory and streams only).  Every function enqueues work on torch's
current stream of the tensors' device and returns without synchronising.
"""
import ctypes

import numpy as np

import torch

from . import _lib

F64 = torch.float64
I64 = torch.int64
I32 = torch.int32


def _dev_check(t, dtype, name):
    if t is None:
        return
    if not isinstance(t, torch.Tensor):
        raise TypeError('%s must be a torch.Tensor' % name)
    if not t.is_cuda:
        raise ValueError('%s must live on the GPU (got %s); there is no CPU path' % (name, t.device))
    if t.dtype != dtype:
        raise TypeError('%s must have dtype %s (got %s)' % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError('%s must be contiguous' % name)


def _p(t):
    return 0 if t is None else t.data_ptr()


def _stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


class CellRanges(object):
    """Device view of many cells: weights + per-cell photon ranges + S, B.

    photons   how many photons lie inside the cells (sum of hi - lo), which is what the fit and
              profile kernels size their shape for; None: counted on the device at the first
              call that needs it (one small read, once per object).  A subset of cells keeps the
              whole weight array and carries its own count, so it is fitted with the shape of
              its own cells, not of cells as large as the array is long."""

    def __init__(self, we, lo, hi, S, B, photons=None):
        _dev_check(we, F64, 'we')
        _dev_check(lo, I64, 'lo')
        _dev_check(hi, I64, 'hi')
        _dev_check(S, F64, 'S')
        _dev_check(B, F64, 'B')
        n = lo.numel()
        if hi.numel() != n or S.numel() != n or (B is not None and B.numel() != n):
            raise ValueError('lo, hi, S, B must have one entry per cell')
        self.we, self.lo, self.hi, self.S, self.B = we, lo, hi, S, B
        self.ncell = n
        self.nph = we.numel()
        self.device = we.device
        self._photons = None if photons is None else int(photons)

    @property
    def photons(self):
        if self._photons is None:
            self._photons = int((self.hi - self.lo).sum().item()) if self.ncell else 0
            owner = getattr(self, '_owner', None)
            if owner is not None:
                owner._photons = self._photons
        return self._photons

    @classmethod
    def from_offsets(cls, we, offsets, S, B):
        _dev_check(offsets, I64, 'offsets')
        return cls(we, offsets[:-1], offsets[1:], S, B)

    def _head(self):
        return (_p(self.we), self.nph, _p(self.lo), _p(self.hi))

    def shard(self, c0, c1, photons=None):
        """Cells [c0, c1) as a view of their own, over the same weight array.  photons: how
        many photons the shard's cells hold if the caller knows (a split balanced by photon
        count does); otherwise the shard is taken to be as densely filled as the whole.
        Nothing is read back from the device."""
        n = max(0, min(c1, self.ncell) - max(c0, 0))
        if photons is None and self._photons is not None and self.ncell:
            photons = int(round(self._photons * (n / self.ncell)))
        return CellRanges(self.we, self.lo[c0:c1].contiguous(), self.hi[c0:c1].contiguous(),
                          self.S[c0:c1].contiguous(), None if self.B is None else self.B[c0:c1].contiguous(),
                          photons=photons)


def _per_cell(x, n, dev, name):
    """Scalar -> (None, scalar); tensor -> (tensor, 0.0)."""
    if x is None:
        return None, 0.0
    if isinstance(x, torch.Tensor):
        _dev_check(x, F64, name)
        if x.numel() != n:
            raise ValueError('%s must have one entry per cell' % name)
        return x, 0.0
    return None, float(x)


def cell_fit(cr, guess=1.0, beta=1.0, swap=False, want_ts=True):
    n, dev = cr.ncell, cr.device
    gt, g0 = _per_cell(guess, n, dev, 'guess')
    bt, b0 = _per_cell(beta, n, dev, 'beta')
    alpha = torch.empty(n, dtype=F64, device=dev)
    ts = torch.empty(n, dtype=F64, device=dev) if want_ts else None
    flags = torch.empty(n, dtype=I32, device=dev)
    with torch.cuda.device(dev):
        _lib.call('godot_cell_fit_shaped', *cr._head(), _p(cr.S), _p(cr.B), n, cr.photons, _p(gt), g0, _p(bt), b0,
                  int(bool(swap)), _p(alpha), _p(ts), _p(flags), _stream(dev))
    return alpha, ts, flags


def cell_loglike(cr, alpha, swap=False):
    n, dev = cr.ncell, cr.device
    _dev_check(alpha, F64, 'alpha')
    if alpha.numel() != n:
        raise ValueError('alpha must have one entry per cell')
    out = torch.empty(n, dtype=F64, device=dev)
    with torch.cuda.device(dev):
        _lib.call('godot_cell_loglike', *cr._head(), _p(cr.S), _p(cr.B), n, _p(alpha),
                  int(bool(swap)), _p(out), _stream(dev))
    return out


def cell_sums(cr, alpha=1.0, beta=1.0, swap=False):
    """(ncell, 8) sums over each cell's photons at flux factors alpha, beta, with
    t = 1 + (beta-1)(1-w) + (alpha-1)w: sum log t, sum w/t, sum (1-w)/t, sum (w/t)^2,
    sum (w/t)^3, sum ((1-w)/t)^2, sum (w/t)((1-w)/t), sum (w/t)((1-w)/t)^2."""
    n, dev = cr.ncell, cr.device
    at, a0 = _per_cell(alpha, n, dev, 'alpha')
    bt, b0 = _per_cell(beta, n, dev, 'beta')
    out = torch.empty((n, 8), dtype=F64, device=dev)
    with torch.cuda.device(dev):
        _lib.call('godot_cell_sums', *cr._head(), n, _p(at), a0, _p(bt), b0, int(bool(swap)), _p(out),
                  _stream(dev))
    return out


def cell_logpdf(cr, aopt=None, dlogl=20., npt=100, include_zero=False, swap=False):
    n, dev = cr.ncell, cr.device
    if aopt is not None:
        _dev_check(aopt, F64, 'aopt')
        if aopt.numel() != n:
            raise ValueError('aopt must have one entry per cell')
    dom = torch.empty((n, npt), dtype=F64, device=dev)
    cod = torch.empty((n, npt), dtype=F64, device=dev)
    flags = torch.empty(n, dtype=I32, device=dev)
    with torch.cuda.device(dev):
        _lib.call('godot_cell_logpdf_shaped', *cr._head(), _p(cr.S), _p(cr.B), n, cr.photons, _p(aopt), float(dlogl),
                  int(npt), int(bool(include_zero)), int(bool(swap)), _p(dom), _p(cod), _p(flags),
                  _stream(dev))
    return dom, cod, flags


def profile_quantiles(dom, cod, conf_lo, conf_hi, linspace=False):
    """Arg-max abscissa and the two quantiles of exp(cod) for every row.  linspace=True: the
    rows of dom are numpy.linspace of their ends to the bit (what cell_logpdf and
    cell_logpdf_profile write); only the ends are read then, half the bytes."""
    _dev_check(dom, F64, 'dom')
    _dev_check(cod, F64, 'cod')
    n, npt = dom.shape
    dev = dom.device
    ag = torch.empty(n, dtype=F64, device=dev)
    xlo = torch.empty(n, dtype=F64, device=dev)
    xhi = torch.empty(n, dtype=F64, device=dev)
    with torch.cuda.device(dev):
        _lib.call('godot_profile_quantiles_linspace' if linspace else 'godot_profile_quantiles', _p(dom), _p(cod),
                  n, npt, float(conf_lo), float(conf_hi), _p(ag), _p(xlo), _p(xhi), _stream(dev))
    return ag, xlo, xhi


def lightcurve_rows(tstart, tstop, alpha, ts, xlo, xhi, S, tsmin):
    n, dev = alpha.numel(), alpha.device
    for t, nm in ((tstart, 'tstart'), (tstop, 'tstop'), (alpha, 'alpha'), (ts, 'ts'), (xlo, 'xlo'),
                  (xhi, 'xhi'), (S, 'S')):
        _dev_check(t, F64, nm)
    rows = torch.empty((n, 5), dtype=F64, device=dev)
    with torch.cuda.device(dev):
        _lib.call('godot_lightcurve_rows', _p(tstart), _p(tstop), _p(alpha), _p(ts), _p(xlo),
                  _p(xhi), _p(S), n, float(tsmin), _p(rows), _stream(dev))
    return rows


def row_peaks(a, first=0):
    """(values, indices) of the largest entry of every row of a contiguous 2-D tensor from column
    `first` on, the index counted from `first`: the first of equal entries, a NaN before any
    number (numpy's argmax)."""
    _dev_check(a, F64, 'a')
    if a.dim() != 2 or not a.is_contiguous():
        raise ValueError('a must be a contiguous 2-D tensor')
    nrow, ncol = a.shape
    dev = a.device
    values = torch.empty(nrow, dtype=F64, device=dev)
    indices = torch.empty(nrow, dtype=I64, device=dev)
    ws = _workspace(_lib.lib.godot_row_peaks_workspace_bytes(nrow, ncol), dev)
    with torch.cuda.device(dev):
        _lib.call('godot_row_peaks', _p(a), nrow, ncol, int(first), _p(values), _p(indices), _p(ws), ws.numel(),
                  _stream(dev))
    return values, indices


def spectrum_frequencies(nk, step, scale, divide, device):
    """(i * step) / scale (divide) or (i * step) * scale for i < nk: the frequency axis of
    power_spectrum_fft / power_spectrum_simple with numpy's roundings (core.py:2677-2679, 2765-2766)."""
    out = torch.empty(int(nk), dtype=F64, device=device)
    with torch.cuda.device(device):
        _lib.call('godot_spectrum_frequencies', int(nk), float(step), float(scale), int(bool(divide)), _p(out),
                  _stream(device))
    return out


def met2mjd(met, mjdref):
    """MJD of mission elapsed times with the reference's extended-precision roundings
    (core.py:24-26); NaN where the integer restatement does not apply."""
    _dev_check(met, F64, 'met')
    out = torch.empty_like(met)
    with torch.cuda.device(met.device):
        _lib.call('godot_met2mjd', _p(met), met.numel(), float(mjdref), _p(out), _stream(met.device))
    return out


def cell_beta_max(cr, alpha, guess=1.0, swap=False, want_logl=True):
    n, dev = cr.ncell, cr.device
    at, a0 = _per_cell(alpha, n, dev, 'alpha')
    gt, g0 = _per_cell(guess, n, dev, 'guess')
    beta = torch.empty(n, dtype=F64, device=dev)
    lp = torch.empty(n, dtype=F64, device=dev) if want_logl else None
    flags = torch.empty(n, dtype=I32, device=dev)
    with torch.cuda.device(dev):
        _lib.call('godot_cell_beta_max', *cr._head(), _p(cr.S), _p(cr.B), n, _p(at), a0, _p(gt), g0,
                  int(bool(swap)), _p(beta), _p(lp), _p(flags), _stream(dev))
    return beta, lp, flags


def cell_fit_joint(cr, guess_a=1.0, guess_b=1.0, swap=False):
    """Joint (alpha, beta) maximum of every cell -> (alpha, beta, logL, flags)."""
    n, dev = cr.ncell, cr.device
    at, a0 = _per_cell(guess_a, n, dev, 'guess_a')
    bt, b0 = _per_cell(guess_b, n, dev, 'guess_b')
    alpha = torch.empty(n, dtype=F64, device=dev)
    beta = torch.empty(n, dtype=F64, device=dev)
    logl = torch.empty(n, dtype=F64, device=dev)
    flags = torch.empty(n, dtype=I32, device=dev)
    with torch.cuda.device(dev):
        _lib.call('godot_cell_fit_joint', *cr._head(), _p(cr.S), _p(cr.B), n, _p(at), a0, _p(bt), b0,
                  int(bool(swap)), _p(alpha), _p(beta), _p(logl), _p(flags), _stream(dev))
    return alpha, beta, logl, flags


def cell_logpdf_profile(cr, aopt=None, dlogl=20., npt=100, swap=False):
    n, dev = cr.ncell, cr.device
    if aopt is not None:
        _dev_check(aopt, F64, 'aopt')
        if aopt.numel() != n:
            raise ValueError('aopt must have one entry per cell')
    dom = torch.empty((n, npt), dtype=F64, device=dev)
    cod = torch.empty((n, npt), dtype=F64, device=dev)
    flags = torch.empty(n, dtype=I32, device=dev)
    with torch.cuda.device(dev):
        _lib.call('godot_cell_logpdf_profile', *cr._head(), _p(cr.S), _p(cr.B), n, _p(aopt),
                  float(dlogl), int(npt), int(bool(swap)), _p(dom), _p(cod), _p(flags), _stream(dev))
    return dom, cod, flags


def cell_moments(we, lo, hi):
    _dev_check(we, F64, 'we')
    _dev_check(lo, I64, 'lo')
    _dev_check(hi, I64, 'hi')
    n, dev = lo.numel(), we.device
    w1 = torch.empty(n, dtype=F64, device=dev)
    w2 = torch.empty(n, dtype=F64, device=dev)
    with torch.cuda.device(dev):
        _lib.call('godot_cell_moments', _p(we), we.numel(), _p(lo), _p(hi), n, _p(w1), _p(w2),
                  _stream(dev))
    return w1, w2


# ---------------------------------------------------------------------------
# binning
# ---------------------------------------------------------------------------
def cell_series(we, lo, hi, exp, sexp, bexp):
    """The series power_spectrum_fft transforms, from the bins of a time series: returns
    (W, WW, Wb, WbWb); bins without exposure are emptied, and sexp / bexp are zeroed there IN
    PLACE (what CellTimeSeries does to them)."""
    for t, nm, dt in ((we, 'we', F64), (lo, 'lo', I64), (hi, 'hi', I64), (exp, 'exp', F64), (sexp, 'sexp', F64),
                      (bexp, 'bexp', F64)):
        _dev_check(t, dt, nm)
    n, dev = lo.numel(), we.device
    if any(t.numel() != n for t in (hi, exp, sexp, bexp)):
        raise ValueError('lo, hi, exp, sexp, bexp must have one entry per bin')
    out = [torch.empty(n, dtype=F64, device=dev) for _ in range(4)]
    with torch.cuda.device(dev):
        _lib.call('godot_cell_series', _p(we), we.numel(), _p(lo), _p(hi), _p(exp), n, _p(sexp), _p(bexp),
                  _p(out[0]), _p(out[1]), _p(out[2]), _p(out[3]), _stream(dev))
    return tuple(out)


def _workspace(nbytes, dev):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=dev)


def exposure_table(tstart, tstop, exposure, livetime, s_over_e, b_over_e):
    """Left-to-right running sums of the four exposure series (device blob)."""
    for t, nm in ((tstart, 'TSTART'), (tstop, 'TSTOP'), (exposure, 'exposure'),
                  (livetime, 'LIVETIME')):
        _dev_check(t, F64, nm)
    n_ft2, dev = tstart.numel(), tstart.device
    if not (tstop.numel() == exposure.numel() == livetime.numel() == n_ft2):
        raise ValueError('TSTART, TSTOP, exposure, LIVETIME must have equal lengths')
    table = _workspace(_lib.lib.godot_exposure_table_bytes(n_ft2), dev)
    with torch.cuda.device(dev):
        _lib.call('godot_exposure_table', _p(tstart), _p(tstop), _p(exposure), _p(livetime), n_ft2,
                  float(s_over_e), float(b_over_e), _p(table), table.numel(), _stream(dev))
    return table


def exposure_at(tstart, tstop, exposure, livetime, s_over_e, b_over_e, table, times,
                cumulative=True, deadtime_too=True):
    _dev_check(times, F64, 'times')
    n_ft2, n_t, dev = tstart.numel(), times.numel(), times.device
    n_out = n_t if cumulative else max(n_t - 1, 0)
    outs = [torch.empty(n_out, dtype=F64, device=dev) for _ in range(4 if deadtime_too else 3)]
    ws = _workspace(_lib.lib.godot_exposure_workspace_bytes(n_t), dev)
    with torch.cuda.device(dev):
        _lib.call('godot_exposure_at', _p(tstart), _p(tstop), _p(exposure), _p(livetime), n_ft2,
                  float(s_over_e), float(b_over_e), _p(table), _p(times), n_t,
                  int(bool(cumulative)), _p(outs[0]), _p(outs[1]), _p(outs[2]),
                  _p(outs[3]) if deadtime_too else 0, _p(ws), ws.numel(), _stream(dev))
    return outs[0], outs[1], outs[2], (outs[3] if deadtime_too else None)


def exposure_table_rows(tstart, tstop, exposure, sexposure, bexposure, livetime):
    """Running sums when the scaled exposures are per-row arrays (apply_scalers)."""
    n_ft2, dev = tstart.numel(), tstart.device
    for t, nm in ((tstart, 'TSTART'), (tstop, 'TSTOP'), (exposure, 'exposure'), (sexposure, 'sexposure'),
                  (bexposure, 'bexposure'), (livetime, 'LIVETIME')):
        _dev_check(t, F64, nm)
        if t.numel() != n_ft2:
            raise ValueError('the exposure table columns must have equal lengths')
    table = _workspace(_lib.lib.godot_exposure_table_bytes(n_ft2), dev)
    with torch.cuda.device(dev):
        _lib.call('godot_exposure_table_rows', _p(tstart), _p(tstop), _p(exposure), _p(sexposure),
                  _p(bexposure), _p(livetime), n_ft2, _p(table), table.numel(), _stream(dev))
    return table


def exposure_at_rows(tstart, tstop, exposure, sexposure, bexposure, livetime, table, times,
                     cumulative=True, deadtime_too=True):
    _dev_check(times, F64, 'times')
    n_ft2, n_t, dev = tstart.numel(), times.numel(), times.device
    n_out = n_t if cumulative else max(n_t - 1, 0)
    outs = [torch.empty(n_out, dtype=F64, device=dev) for _ in range(4 if deadtime_too else 3)]
    ws = _workspace(_lib.lib.godot_exposure_workspace_bytes(n_t), dev)
    with torch.cuda.device(dev):
        _lib.call('godot_exposure_at_rows', _p(tstart), _p(tstop), _p(exposure), _p(sexposure),
                  _p(bexposure), _p(livetime), n_ft2, _p(table), _p(times), n_t, int(bool(cumulative)),
                  _p(outs[0]), _p(outs[1]), _p(outs[2]), _p(outs[3]) if deadtime_too else 0, _p(ws),
                  ws.numel(), _stream(dev))
    return outs[0], outs[1], outs[2], (outs[3] if deadtime_too else None)


def search_sorted(sorted_vals, values, side='left'):
    """numpy.searchsorted on the device -> int64 indices."""
    _dev_check(sorted_vals, F64, 'sorted_vals')
    _dev_check(values, F64, 'values')
    if side not in ('left', 'right'):
        raise ValueError("side must be 'left' or 'right'")
    out = torch.empty(values.numel(), dtype=I64, device=values.device)
    with torch.cuda.device(values.device):
        _lib.call('godot_search_sorted', _p(sorted_vals), sorted_vals.numel(), _p(values), values.numel(),
                  int(side == 'right'), _p(out), _stream(values.device))
    return out


def piecewise_average(starts, stops, scales, prod, cprod, t0, t1):
    """Mean of the piecewise-constant function over each interval t0[k] .. t1[k]."""
    for x, nm in ((starts, 'starts'), (stops, 'stops'), (scales, 'scales'), (prod, 'prod'), (cprod, 'cprod'),
                  (t0, 't0'), (t1, 't1')):
        _dev_check(x, F64, nm)
    n = stops.numel()
    if n == 0 or any(x.numel() != n for x in (starts, scales, prod, cprod)):
        raise ValueError('the segment arrays must be non-empty and of equal length')
    if t0.numel() != t1.numel():
        raise ValueError('t0 and t1 must have the same length')
    out = torch.empty(t0.numel(), dtype=F64, device=t0.device)
    with torch.cuda.device(t0.device):
        _lib.call('godot_piecewise_average', _p(starts), _p(stops), _p(scales), _p(prod), _p(cprod), n, _p(t0),
                  _p(t1), t0.numel(), _p(out), _stream(t0.device))
    return out


def piecewise_lookup(stops, scales, t):
    """scales[searchsorted(stops, t)] for event times t."""
    for x, nm in ((stops, 'stops'), (scales, 'scales'), (t, 't')):
        _dev_check(x, F64, nm)
    if stops.numel() != scales.numel() or stops.numel() == 0:
        raise ValueError('stops and scales must be non-empty and of equal length')
    out = torch.empty(t.numel(), dtype=F64, device=t.device)
    with torch.cuda.device(t.device):
        _lib.call('godot_piecewise_lookup', _p(stops), _p(scales), stops.numel(), _p(t), t.numel(), _p(out),
                  _stream(t.device))
    return out


def rescale_weights(w, a=1.0, b=1.0):
    """a w / (a w + b (1 - w)); a, b scalars or per-photon tensors."""
    _dev_check(w, F64, 'w')
    at, a0 = _per_cell(a, w.numel(), w.device, 'a')
    bt, b0 = _per_cell(b, w.numel(), w.device, 'b')
    out = torch.empty_like(w)
    with torch.cuda.device(w.device):
        _lib.call('godot_rescale_weights', _p(w), _p(at), a0, _p(bt), b0, w.numel(), _p(out), _stream(w.device))
    return out


def bin_photons(ti, start0, stops):
    _dev_check(ti, F64, 'ti')
    _dev_check(stops, F64, 'stops')
    n, dev = stops.numel(), ti.device
    lo = torch.empty(n, dtype=I64, device=dev)
    hi = torch.empty(n, dtype=I64, device=dev)
    counts = torch.empty(n, dtype=I64, device=dev)
    with torch.cuda.device(dev):
        _lib.call('godot_bin_photons', _p(ti), ti.numel(), float(start0), _p(stops), n, _p(lo),
                  _p(hi), _p(counts), _stream(dev))
    return lo, hi, counts


# one record per source, field for field the C struct godot_source_cells (all fields 8 bytes)
SOURCE_CELLS = np.dtype([(n, 'u8') for n in ('tstart', 'tstop', 'exposure', 'livetime', 'sexposure', 'bexposure',
                                             'table', 'ti')]
                        + [('n_ft2', 'i8'), ('nph', 'i8'), ('s_over_e', 'f8'), ('b_over_e', 'f8'), ('t0', 'f8'),
                           ('tcell', 'f8'), ('ncell', 'i8'), ('cell_offset', 'i8'), ('photon_offset', 'i8'),
                           ('checkpoints', 'i8')])
# one record per table, field for field the C struct godot_exposure_rows
EXPOSURE_ROWS = np.dtype([(n, 'u8') for n in ('tstart', 'tstop', 'exposure', 'livetime', 'sexposure', 'bexposure')]
                         + [('n_ft2', 'i8'), ('s_over_e', 'f8'), ('b_over_e', 'f8'), ('checkpoints', 'u8')])


def _row_pointers(s, k, dev, col):
    """Checks the exposure-table columns of descriptor `s` and appends their addresses to
    `col`; returns the number of rows."""
    n_ft2 = s['tstart'].numel()
    for nm in ('tstart', 'tstop', 'exposure', 'livetime', 'sexposure', 'bexposure'):
        t = s.get(nm)
        if t is None:
            if nm in ('sexposure', 'bexposure'):
                col[nm].append(0)
                continue
            raise ValueError('source %d: no %s' % (k, nm))
        _dev_check(t, F64, nm)
        if t.numel() != n_ft2 or t.device != dev:
            raise ValueError('source %d: the exposure table columns must have equal lengths and live on '
                             'the device of the batch' % k)
        col[nm].append(t.data_ptr())
    if (col['sexposure'][-1] == 0) != (col['bexposure'][-1] == 0):
        raise ValueError('source %d: sexposure and bexposure come together' % k)
    if n_ft2 < 1:
        raise ValueError('source %d: empty exposure table' % k)
    return n_ft2


def exposure_checkpoints(tables):
    """Running sums of many exposure tables, kept for every sixteenth row (2 bytes a row where
    exposure_table keeps 80), built by launches whose number does not depend on how many tables
    there are.  `tables`: one dict per table with the device tensors 'tstart', 'tstop',
    'exposure', 'livetime' (optionally 'sexposure' and 'bexposure') and the numbers 's_over_e',
    'b_over_e'.  Returns one uint8 tensor per table (views of one allocation) for the
    'checkpoints' entry of bin_sources' descriptors."""
    if not tables:
        return []
    dev = tables[0]['tstart'].device
    col = {n: [] for n in EXPOSURE_ROWS.names}
    sizes = []
    for k, s in enumerate(tables):
        n_ft2 = _row_pointers(s, k, dev, col)
        col['n_ft2'].append(n_ft2)
        col['s_over_e'].append(float(s.get('s_over_e', 0.0)))
        col['b_over_e'].append(float(s.get('b_over_e', 0.0)))
        sizes.append(int(_lib.lib.godot_exposure_checkpoints_bytes(n_ft2)))
    whole = torch.empty(sum(sizes), dtype=torch.uint8, device=dev)
    out, at = [], 0
    for b in sizes:
        out.append(whole[at:at + b])
        at += b
    col['checkpoints'] = [t.data_ptr() for t in out]
    rec = np.zeros(len(tables), dtype=EXPOSURE_ROWS)
    for nm in EXPOSURE_ROWS.names:
        rec[nm] = col[nm]
    ws = _workspace(_lib.lib.godot_exposure_checkpoints_workspace_bytes(rec.ctypes.data, len(tables)), dev)
    with torch.cuda.device(dev):
        _lib.call('godot_exposure_checkpoints', rec.ctypes.data, len(tables), _p(ws), ws.numel(), _stream(dev))
    return out


def bin_sources(sources, deadtime_too=False, want_edges=True):
    """Uniform cells of many sources in two launches.  `sources`: one dict per source with the
    device tensors 'tstart', 'tstop', 'exposure', 'livetime', 'table' (exposure_table of those
    rows) or 'checkpoints' (exposure_checkpoints of them), 'ti' (time-sorted photon times),
    optionally 'sexposure' and 'bexposure' (with the exposure_table_rows table, or checkpoints
    built with them), and the numbers 's_over_e', 'b_over_e', 't0', 'tcell', 'ncell'.
    The cells of the sources follow one another in the outputs, and lo/hi index the
    concatenation of the photon lists.  Returns (starts, stops, exp, sexp, bexp, dead or None,
    lo, hi, source) over all cells; want_edges=False leaves starts, stops and source out (None)."""
    if not sources:
        raise ValueError('no sources')
    dev = sources[0]['ti'].device
    col = {n: [] for n in SOURCE_CELLS.names}       # filled column by column: one numpy call a field
    cells = photons = 0
    for k, s in enumerate(sources):
        n_ft2 = _row_pointers(s, k, dev, col)
        ti, table, ck = s['ti'], s.get('table'), s.get('checkpoints')
        _dev_check(ti, F64, 'ti')
        if ti.device != dev:
            raise ValueError('source %d: all sources of a batch live on one device' % k)
        if ck is not None:
            table = ck
            need = _lib.lib.godot_exposure_checkpoints_bytes(n_ft2)
        else:
            need = _lib.lib.godot_exposure_table_bytes(n_ft2)
        if table is None:
            raise ValueError("source %d: neither 'table' nor 'checkpoints'" % k)
        if table.device != dev or table.dtype != torch.uint8 or table.numel() < need:
            raise ValueError('source %d: the table does not belong to these %d rows' % (k, n_ft2))
        col['checkpoints'].append(int(ck is not None))
        ncell = int(s['ncell'])
        if ncell < 0 or not float(s['tcell']) > 0:
            raise ValueError('source %d: needs ncell >= 0 and tcell > 0' % k)
        col['table'].append(table.data_ptr())
        col['ti'].append(ti.data_ptr())
        col['n_ft2'].append(n_ft2)
        col['nph'].append(ti.numel())
        for nm in ('s_over_e', 'b_over_e', 't0', 'tcell'):
            col[nm].append(float(s[nm]))
        col['ncell'].append(ncell)
        col['cell_offset'].append(cells)
        col['photon_offset'].append(photons)
        cells += ncell
        photons += ti.numel()
    rec = np.zeros(len(sources), dtype=SOURCE_CELLS)
    for nm in SOURCE_CELLS.names:
        rec[nm] = col[nm]
    f = lambda: torch.empty(cells, dtype=F64, device=dev)
    i = lambda: torch.empty(cells, dtype=I64, device=dev)
    starts, stops = (f(), f()) if want_edges else (None, None)
    exp, sexp, bexp = f(), f(), f()
    dead = f() if deadtime_too else None
    lo, hi = i(), i()
    source = i() if want_edges else None
    ws = _workspace(_lib.lib.godot_bin_sources_workspace_bytes(len(sources), cells), dev)
    with torch.cuda.device(dev):
        _lib.call('godot_bin_sources', rec.ctypes.data, len(sources), cells, _p(starts), _p(stops), _p(exp),
                  _p(sexp), _p(bexp), _p(dead) if deadtime_too else 0, _p(lo), _p(hi), _p(source), _p(ws),
                  ws.numel(), _stream(dev))
    return starts, stops, exp, sexp, bexp, dead, lo, hi, source


def select_cells(exp, threshold=0.0):
    """Ascending indices of the cells with exp > threshold (synchronises once
    to learn how many there are)."""
    _dev_check(exp, F64, 'exp')
    n, dev = exp.numel(), exp.device
    keep = torch.empty(n, dtype=I64, device=dev)
    nk = torch.zeros(1, dtype=I64, device=dev)
    ws = _workspace(_lib.lib.godot_select_workspace_bytes(n), dev)
    with torch.cuda.device(dev):
        _lib.call('godot_select_cells', _p(exp), n, float(threshold), _p(keep), _p(nk), _p(ws),
                  ws.numel(), _stream(dev))
    return keep[:int(nk.item())]


def select_cells_static(exp, threshold, keep, n_keep):
    """select_cells into the caller's tensors, through the entry point that allocates
    nothing (keep: int64 [>= exp.numel()], n_keep: int64 [1]); nothing is synchronised."""
    _dev_check(exp, F64, 'exp')
    _dev_check(keep, I64, 'keep')
    _dev_check(n_keep, I64, 'n_keep')
    if keep.numel() < exp.numel() or n_keep.numel() < 1:
        raise ValueError('keep must hold one index per cell and n_keep one count')
    with torch.cuda.device(exp.device):
        _lib.call('godot_select_cells_static', _p(exp), exp.numel(), float(threshold), _p(keep), _p(n_keep),
                  _stream(exp.device))


def gather(src, idx):
    _dev_check(idx, I64, 'idx')
    n, dev = idx.numel(), src.device
    dst = torch.empty(n, dtype=src.dtype, device=dev)
    if src.dtype == F64:
        name = 'godot_gather_f64'
    elif src.dtype == I64:
        name = 'godot_gather_i64'
    else:
        raise TypeError('gather supports float64 and int64')
    _dev_check(src, src.dtype, 'src')
    with torch.cuda.device(dev):
        _lib.call(name, _p(src), _p(idx), n, _p(dst), _stream(dev))
    return dst


def gather_columns(columns, idx):
    """[c[idx] for c in columns] in one launch (float64 and int64 columns of one length,
    at most 16 a launch): the trimming of a cell table by one selection."""
    _dev_check(idx, I64, 'idx')
    n = idx.numel()
    out = []
    for c0 in range(0, len(columns), 16):
        cols = columns[c0:c0 + 16]
        dev = cols[0].device
        for c in cols:
            if c.dtype not in (F64, I64):
                raise TypeError('gather_columns supports float64 and int64')
            _dev_check(c, c.dtype, 'column')
        dst = [torch.empty(n, dtype=c.dtype, device=dev) for c in cols]
        PtrArray = ctypes.c_void_p * len(cols)
        with torch.cuda.device(dev):
            _lib.call('godot_gather_columns', PtrArray(*[c.data_ptr() for c in cols]),
                      PtrArray(*[d.data_ptr() for d in dst]), len(cols), _p(idx), n, _stream(dev))
        out += dst
    return out


# ---------------------------------------------------------------------------
# power spectra and trig sums
# ---------------------------------------------------------------------------
def rfft(x, l):
    """numpy.fft.rfft(x, n=l) for a power-of-two l: (l/2+1, 2) tensor of (re, im)."""
    _dev_check(x, F64, 'x')
    dev = x.device
    out = torch.empty((l // 2 + 1, 2), dtype=F64, device=dev)
    ws = _workspace(_lib.lib.godot_rfft_workspace_bytes(l), dev)
    with torch.cuda.device(dev):
        _lib.call('godot_rfft_f64', _p(x), x.numel(), l, _p(out), _p(ws), ws.numel(), _stream(dev))
    return out


def power_spectrum_workspace(l, dev):
    """A workspace for power_spectrum_fft calls of padded length up to l: a caller that makes
    many calls on one stream may allocate it once and hand it to every call."""
    return _workspace(_lib.lib.godot_power_spectrum_workspace_bytes(l), dev)


def power_spectrum_fft(W, WW, S, Wb, WbWb, B, l, want_amps=False, out=None, workspace=None):
    """out: three contiguous float64 tensors of l // 4 + 1 entries to take P_0, P_1, P_b (rows
    of a larger array, say); workspace: power_spectrum_workspace(l or more) on the same device,
    not in use by a call on another stream."""
    for t, nm in ((W, 'W'), (WW, 'WW'), (S, 'S'), (Wb, 'Wb'), (WbWb, 'WbWb'), (B, 'B')):
        _dev_check(t, F64, nm)
        if t.numel() != W.numel():
            raise ValueError('the six series must have equal lengths')
    n, dev = W.numel(), W.device
    nk = l // 4 + 1
    if out is None:
        p0 = torch.empty(nk, dtype=F64, device=dev)
        p1 = torch.empty(nk, dtype=F64, device=dev)
        pb = torch.empty(nk, dtype=F64, device=dev)
    else:
        p0, p1, pb = out
        for t, nm in ((p0, 'out[0]'), (p1, 'out[1]'), (pb, 'out[2]')):
            _dev_check(t, F64, nm)
            if t.numel() != nk or t.device != dev:
                raise ValueError('%s must hold l // 4 + 1 entries on the device of the series' % nm)
    amps = torch.empty((8, nk), dtype=F64, device=dev) if want_amps else None
    need = _lib.lib.godot_power_spectrum_workspace_bytes(l)
    ws = workspace if workspace is not None else _workspace(need, dev)
    if ws.device != dev or ws.dtype != torch.uint8 or ws.numel() < need:
        raise ValueError('the workspace is too small for l = %d, or not on the device of the series' % l)
    with torch.cuda.device(dev):
        _lib.call('godot_power_spectrum_fft', _p(W), _p(WW), _p(S), _p(Wb), _p(WbWb), _p(B), n, l,
                  _p(p0), _p(p1), _p(pb), _p(amps), _p(ws), ws.numel(), _stream(dev))
    return p0, p1, pb, amps


def power_spectrum_parts_supported(n, l, nparts):
    """Whether power_spectrum_columns / power_spectrum_rows serve a series of n bins padded to l,
    dealt over nparts GPUs."""
    return bool(_lib.lib.godot_power_spectrum_parts_supported(int(n), int(l), int(nparts)))


def power_spectrum_part_shapes(l, nparts, nrows=None):
    """Shapes (rows, columns) of the five arrays of complex numbers that one part of the columns
    (nrows None: all 8 193 stored rows, a nparts-th of the columns) or a range of nrows whole rows
    (nparts = 1) consists of: WW, WbWb, W - WW, then W - S, Wb - B."""
    rows = int(_lib.lib.godot_power_spectrum_part_rows()) if nrows is None else int(nrows)
    B = l // 16384
    return [(rows, B // 2 // nparts)] * 3 + [(rows, B // nparts)] * 2


def _part_views(buf, shapes):
    out, at = [], 0
    for r, c in shapes:
        out.append(buf[at:at + 2 * r * c].view(r, c, 2))
        at += 2 * r * c
    return out


def power_spectrum_columns(W, WW, S, Wb, WbWb, B, l, part, nparts):
    """Columns-pass of power_spectrum_fft for part `part` of nparts GPUs.  Returns (buffer,
    views): one float64 allocation holding the five arrays of this part back to back, and the
    five (8 193, columns, 2) views of it."""
    for t, nm in ((W, 'W'), (WW, 'WW'), (S, 'S'), (Wb, 'Wb'), (WbWb, 'WbWb'), (B, 'B')):
        _dev_check(t, F64, nm)
        if t.numel() != W.numel():
            raise ValueError('the six series must have equal lengths')
    n, dev = W.numel(), W.device
    if not power_spectrum_parts_supported(n, l, nparts):
        raise ValueError('a series of %d bins padded to %d is not transformed in %d parts' % (n, l, nparts))
    shapes = power_spectrum_part_shapes(l, nparts)
    buf = torch.empty(sum(2 * r * c for r, c in shapes), dtype=F64, device=dev)
    ws = _workspace(_lib.lib.godot_power_spectrum_part_workspace_bytes(l, nparts), dev)
    with torch.cuda.device(dev):
        _lib.call('godot_power_spectrum_columns', _p(W), _p(WW), _p(S), _p(Wb), _p(WbWb), _p(B), n, l, int(part),
                  int(nparts), _p(buf), _p(ws), ws.numel(), _stream(dev))
    return buf, _part_views(buf, shapes)


def power_spectrum_rows(v_rows, dc, l, m_first, m_end, want_amps=False):
    """Rows-pass of power_spectrum_fft for rows m_first .. m_end - 1: v_rows is one float64
    tensor with the five arrays of whole rows back to back (power_spectrum_part_shapes(l, 1,
    m_end - m_first)), dc the sums of WW, WbWb and W - WW (device, 3 doubles).  Returns
    (p0, p1, pb, amps) of l // 4 + 1 bins with the bins of these rows filled and zeros elsewhere."""
    _dev_check(v_rows, F64, 'v_rows')
    _dev_check(dc, F64, 'dc')
    dev = v_rows.device
    need = sum(2 * r * c for r, c in power_spectrum_part_shapes(l, 1, m_end - m_first))
    if v_rows.numel() < need or dc.numel() < 3:
        raise ValueError('v_rows holds %d numbers, rows %d .. %d of l = %d need %d' % (v_rows.numel(), m_first, m_end,
                                                                                      l, need))
    nk = l // 4 + 1
    p0, p1, pb = (torch.zeros(nk, dtype=F64, device=dev) for _ in range(3))
    amps = torch.zeros((8, nk), dtype=F64, device=dev) if want_amps else None
    with torch.cuda.device(dev):
        _lib.call('godot_power_spectrum_rows', _p(v_rows), _p(dc), l, int(m_first), int(m_end), _p(p0), _p(p1), _p(pb),
                  _p(amps), _stream(dev))
    return p0, p1, pb, amps


def power_spectrum_from_parts(f1, f2, g3, g4, g5, want_amps=False):
    """(p0, p1, pb, amps) of power_spectrum_fft from its five transforms, each an (nk, 2)
    tensor of (re, im): rfft(W - S)[:nk], rfft(Wb - B)[:nk] and the even bins
    rfft(.)[0 : 2 nk : 2] of WW, WbWb and W - WW."""
    nk = f1.shape[0]
    for t, nm in ((f1, 'f1'), (f2, 'f2'), (g3, 'g3'), (g4, 'g4'), (g5, 'g5')):
        _dev_check(t, F64, nm)
        if tuple(t.shape) != (nk, 2):
            raise ValueError('%s must have shape (nk, 2)' % nm)
    dev = f1.device
    p0 = torch.empty(nk, dtype=F64, device=dev)
    p1 = torch.empty(nk, dtype=F64, device=dev)
    pb = torch.empty(nk, dtype=F64, device=dev)
    amps = torch.empty((8, nk), dtype=F64, device=dev) if want_amps else None
    with torch.cuda.device(dev):
        _lib.call('godot_power_spectrum_from_parts', _p(f1), _p(f2), _p(g3), _p(g4), _p(g5), nk, _p(p0), _p(p1),
                  _p(pb), _p(amps), _stream(dev))
    return p0, p1, pb, amps


def power_spectrum_simple(W, WW, S, want_moments=False):
    for t, nm in ((W, 'W'), (WW, 'WW'), (S, 'S')):
        _dev_check(t, F64, nm)
    n, dev = W.numel(), W.device
    nk = n // 4 + 1
    dl = torch.empty(nk, dtype=F64, device=dev)
    mom = torch.empty((4, nk), dtype=F64, device=dev) if want_moments else None
    ws = _workspace(_lib.lib.godot_power_spectrum_workspace_bytes(n), dev)
    with torch.cuda.device(dev):
        _lib.call('godot_power_spectrum_simple', _p(W), _p(WW), _p(S), n, _p(dl), _p(mom), _p(ws),
                  ws.numel(), _stream(dev))
    return dl, mom


def trigsum_cells(w1, w2, exp, tmid, freqs, exp_scale=1.0):
    for t, nm in ((w1, 'w1'), (w2, 'w2'), (exp, 'exp'), (tmid, 'tmid'), (freqs, 'freqs')):
        _dev_check(t, F64, nm)
    dev = w1.device
    out = torch.empty(freqs.numel(), dtype=F64, device=dev)
    with torch.cuda.device(dev):
        _lib.call('godot_trigsum_cells', _p(w1), _p(w2), _p(exp), _p(tmid), w1.numel(), _p(freqs),
                  freqs.numel(), float(exp_scale), _p(out), _stream(dev))
    return out


def fit_harmonics(w1, w2, exp, tmid, tmean, freqs):
    for t, nm in ((w1, 'w1'), (w2, 'w2'), (exp, 'exp'), (tmid, 'tmid'), (freqs, 'freqs')):
        _dev_check(t, F64, nm)
    dev = w1.device
    out = torch.empty((freqs.numel(), 3), dtype=F64, device=dev)
    with torch.cuda.device(dev):
        _lib.call('godot_fit_harmonics', _p(w1), _p(w2), _p(exp), _p(tmid), w1.numel(), float(tmean),
                  _p(freqs), freqs.numel(), _p(out), _stream(dev))
    return out


def z2m(t, w, f0, df, nf, m=2, j0=0, want_z2=True, want_h=False, workspace=None):
    """Weighted Z^2_m (and/or H) on the grid f0 + (j0 + j) df, j < nf."""
    _dev_check(t, F64, 't')
    _dev_check(w, F64, 'w')
    if t.numel() != w.numel():
        raise ValueError('t and w must have equal lengths')
    dev = t.device
    need = _lib.lib.godot_z2m_workspace_bytes(t.numel(), nf, m)
    ws = workspace if workspace is not None else _workspace(need, dev)
    z = torch.empty(nf, dtype=F64, device=dev) if want_z2 else None
    h = torch.empty(nf, dtype=F64, device=dev) if want_h else None
    with torch.cuda.device(dev):
        _lib.call('godot_z2m', _p(t), _p(w), t.numel(), float(f0), float(df), int(j0), int(nf), int(m),
                  _p(z), _p(h), _p(ws), ws.numel(), _stream(dev))
    return z, h


def z2m_sums(t, w, f0, df, nf, m=2, j0=0, workspace=None):
    """Trig sums of one photon list: (nf, m, 2) tensor of (cos, sin) sums and the
    device scalar sum(w^2).  Sums of disjoint photon lists add."""
    _dev_check(t, F64, 't')
    _dev_check(w, F64, 'w')
    if t.numel() != w.numel():
        raise ValueError('t and w must have equal lengths')
    dev = t.device
    need = _lib.lib.godot_z2m_workspace_bytes(t.numel(), nf, m)
    ws = workspace if workspace is not None else _workspace(need, dev)
    sums = torch.empty((nf, m, 2), dtype=F64, device=dev)
    sumw2 = torch.empty(1, dtype=F64, device=dev)
    with torch.cuda.device(dev):
        _lib.call('godot_z2m_sums', _p(t), _p(w), t.numel(), float(f0), float(df), int(j0), int(nf), int(m),
                  _p(sums), _p(sumw2), _p(ws), ws.numel(), _stream(dev))
    return sums, sumw2


def z2m_finish(sums, sumw2, want_z2=True, want_h=False):
    """Z^2_m and/or H from total trig sums (nf, m, 2) and the total sum(w^2)."""
    _dev_check(sums, F64, 'sums')
    _dev_check(sumw2, F64, 'sumw2')
    nf, m = sums.shape[0], sums.shape[1]
    dev = sums.device
    z = torch.empty(nf, dtype=F64, device=dev) if want_z2 else None
    h = torch.empty(nf, dtype=F64, device=dev) if want_h else None
    with torch.cuda.device(dev):
        _lib.call('godot_z2m_finish', _p(sums), _p(sumw2), int(nf), int(m), _p(z), _p(h), _stream(dev))
    return z, h


def ls_cells(flux, tmid, freqs):
    for t, nm in ((flux, 'flux'), (tmid, 'tmid'), (freqs, 'freqs')):
        _dev_check(t, F64, nm)
    dev = flux.device
    out = torch.empty(freqs.numel(), dtype=F64, device=dev)
    with torch.cuda.device(dev):
        _lib.call('godot_ls_cells', _p(flux), _p(tmid), flux.numel(), _p(freqs), freqs.numel(), _p(out),
                  _stream(dev))
    return out


def orbital_modulation_step(starts, stops, y, z, freq, rvals):
    """One frequency: returns the 4 means (device) and updates rvals in place."""
    for t, nm in ((starts, 'starts'), (stops, 'stops'), (y, 'y'), (z, 'z'), (rvals, 'rvals')):
        _dev_check(t, F64, nm)
    dev = y.device
    means = torch.empty(4, dtype=F64, device=dev)
    ws = _workspace(_lib.lib.godot_orbital_workspace_bytes(), dev)
    with torch.cuda.device(dev):
        _lib.call('godot_orbital_modulation_step', _p(starts), _p(stops), _p(y), _p(z), y.numel(),
                  float(freq), _p(means), _p(rvals), _p(ws), ws.numel(), _stream(dev))
    return means


# ---------------------------------------------------------------------------
# Bayesian Blocks
# ---------------------------------------------------------------------------
def bb_fitness(dom, cod, ngrid=1000, rows=None):
    """Fitness rows (packed lower triangle) -> (fit_val, fit_arg).  rows = (first,
    last_exclusive) restricts the work to those rows; the result then holds
    only them, packed one after the other."""
    _dev_check(dom, F64, 'dom')
    _dev_check(cod, F64, 'cod')
    if dom.dim() != 2 or dom.shape != cod.shape:
        raise ValueError('dom and cod must be [ncell, npt] arrays of one shape')
    n, npt = dom.shape
    r0, r1 = (0, n) if rows is None else (int(rows[0]), int(rows[1]))
    if not 0 <= r0 <= r1 <= n:
        raise ValueError('rows must satisfy 0 <= first <= last <= ncell')
    dev = dom.device
    ntri = r1 * (r1 + 1) // 2 - r0 * (r0 + 1) // 2
    fv = torch.empty(ntri, dtype=F64, device=dev)
    fa = torch.empty(ntri, dtype=F64, device=dev)
    ws = torch.empty(max(int(_lib.lib.godot_bb_workspace_bytes(n, npt)), 8), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.call('godot_bb_fitness', _p(dom.contiguous()), _p(cod.contiguous()), n, int(npt), int(ngrid),
                  r0, r1, _p(ws), _p(fv), _p(fa), _stream(dev))
    return fv, fa


def bb_partition(fit_val, ncell, prior):
    """The dynamic programme over the fitness rows -> (best [ncell+1], last [ncell])."""
    _dev_check(fit_val, F64, 'fit_val')
    if fit_val.numel() != ncell * (ncell + 1) // 2:
        raise ValueError('fit_val must hold ncell (ncell + 1) / 2 values')
    dev = fit_val.device
    best = torch.empty(ncell + 1, dtype=F64, device=dev)
    last = torch.empty(ncell, dtype=I64, device=dev)
    with torch.cuda.device(dev):
        _lib.call('godot_bb_partition', _p(fit_val), int(ncell), float(prior), _p(best), _p(last), _stream(dev))
    return best, last

