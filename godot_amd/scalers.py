"""Piecewise-constant flux scalers for Data.get_cells(src_scaler=..., bkg_scaler=...)
(reference scalers.py:3-85).  The interval bookkeeping is host arithmetic on a
handful of intervals; evaluation at the photon times (tens of millions of
lookups) runs on the device."""
import numpy as np
import torch

from . import ops


class PiecewiseScaler(object):
    """Flux multiplier that is constant on given intervals and keeps the first /
    last value outside them (reference scalers.py:5-29)."""

    def __init__(self, starts, stops, scales):
        starts, stops, scales = (np.asarray(x, dtype=np.float64) for x in (starts, stops, scales))
        pad = lambda a, v0, v1: np.concatenate([[v0], a, [v1]])
        self._starts = pad(starts, 0, stops[-1])
        self._stops = pad(stops, starts[0], np.inf)
        self._scales = pad(scales, scales[0], scales[-1])
        self._tbin = self._stops - self._starts
        x = self._tbin.astype(np.longdouble)
        y = self._scales.astype(np.longdouble)
        self._prod = (x * y).astype(np.float64)
        self._cprod = np.cumsum(x * y).astype(np.float64)
        self._dev = {}

    def _on(self, device):
        key = str(device)
        if key not in self._dev:
            self._dev[key] = (torch.from_numpy(self._stops).to(device), torch.from_numpy(self._scales).to(device))
        return self._dev[key]

    def __call__(self, met_start, met_stop=None):
        """Scale at event times (one argument) or averaged over the intervals
        met_start .. met_stop (reference scalers.py:31-85)."""
        if met_stop is None and isinstance(met_start, torch.Tensor) and met_start.is_cuda:
            stops, scales = self._on(met_start.device)
            return ops.piecewise_lookup(stops, scales, met_start)
        to_np = lambda v: v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v, dtype=np.float64)
        T0 = np.atleast_1d(to_np(met_start))
        idx0 = np.searchsorted(self._stops, T0)
        if met_stop is None:
            return self._scales[idx0]
        T1 = np.atleast_1d(to_np(met_stop))
        idx1 = np.searchsorted(self._stops, T1)
        # an interval that starts exactly on a boundary belongs to the next segment
        idx0[T0 == self._stops[idx0]] += 1
        span = idx0 != idx1
        if not np.any(span):
            return self._scales[idx0]
        t0 = self._stops[idx0][span]
        t1 = self._starts[idx1][span]
        F0 = self._scales[idx0][span]
        F1 = self._scales[idx1][span]
        assert np.all(t0 > T0[span]) and np.all(T1[span] > t1)
        edges = F1 * (T1[span] - t1) + F0 * (t0 - T0[span])
        i0 = idx0[span] + 1
        i1 = idx1[span]
        inner = np.zeros(len(i0))
        a = np.flatnonzero((i1 - i0) == 1)
        inner[a] = self._prod[i0[a]]
        a = np.flatnonzero((i1 - i0) > 1)
        inner[a] = self._cprod[i1[a]] - self._cprod[i0[a]]
        out = np.empty(len(span), dtype=float)
        out[span] = (edges + inner) / (T1[span] - T0[span])
        out[~span] = self._scales[idx0[~span]]
        return out
