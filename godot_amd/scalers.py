"""Piecewise-constant flux scalers for Data.get_cells(src_scaler=..., bkg_scaler=...)
(reference scalers.py:3-85).  Device tensors are evaluated by kernels (the
spacecraft table has millions of rows, the photon list tens of millions of
events); host arrays by the same arithmetic in numpy."""
import numpy as np
import torch

from . import ops


class PiecewiseScaler(object):
    """Flux multiplier that is constant on given intervals and keeps the first /
    last value outside them (reference scalers.py:5-29)."""

    def __init__(self, starts, stops, scales):
        starts, stops, scales = (np.asarray(x, dtype=np.float64) for x in (starts, stops, scales))
        # one segment in front (from 0) and one behind (to infinity) carry the end values
        self._starts = np.concatenate([[0.0], starts, [stops[-1]]])
        self._stops = np.concatenate([[starts[0]], stops, [np.inf]])
        self._scales = np.concatenate([[scales[0]], scales, [scales[-1]]])
        self._tbin = self._stops - self._starts
        area = self._tbin.astype(np.longdouble) * self._scales.astype(np.longdouble)
        self._prod = area.astype(np.float64)
        self._cprod = np.cumsum(area).astype(np.float64)
        self._dev = {}

    def _on(self, device):
        key = str(device)
        if key not in self._dev:
            self._dev[key] = tuple(torch.from_numpy(np.ascontiguousarray(a)).to(device) for a in (
                self._starts, self._stops, self._scales, self._prod, self._cprod))
        return self._dev[key]

    def __call__(self, met_start, met_stop=None):
        """Scale at event times (one argument) or averaged over the intervals
        met_start .. met_stop (reference scalers.py:31-85)."""
        on_gpu = isinstance(met_start, torch.Tensor) and met_start.is_cuda
        if on_gpu:
            starts, stops, scales, prod, cprod = self._on(met_start.device)
            if met_stop is None:
                return ops.piecewise_lookup(stops, scales, met_start)
            if isinstance(met_stop, torch.Tensor) and met_stop.is_cuda:
                return ops.piecewise_average(starts, stops, scales, prod, cprod, met_start.contiguous(),
                                             met_stop.contiguous())
        host = lambda v: v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v, dtype=np.float64)
        begin = np.atleast_1d(host(met_start))
        first = np.searchsorted(self._stops, begin)
        if met_stop is None:
            return self._scales[first]
        finish = np.atleast_1d(host(met_stop))
        final = np.searchsorted(self._stops, finish)
        # an interval that begins exactly where a segment ends belongs to the next segment
        first = first + (begin == self._stops[first])
        top = len(self._scales) - 1
        a, b = np.minimum(first, top), np.minimum(final, top)
        # the partial segments at the two ends, the whole segments between them (one: its own
        # product; several: a difference of the running sums), over the length of the interval
        with np.errstate(all='ignore'):
            ends = self._scales[b] * (finish - self._starts[b]) + self._scales[a] * (self._stops[a] - begin)
            nxt = np.minimum(a + 1, top)
            between = final - first - 1
            whole = np.where(between == 1, self._prod[nxt],
                             np.where(between > 1, self._cprod[b] - self._cprod[nxt], 0.0))
            mean = (ends + whole) / (finish - begin)
        return np.where(first == final, self._scales[a], mean)
