#!/usr/bin/env python3
"""The C2 light curve end to end (1e7 photons, 600-s cells): the profile grid (ops.cell_logpdf), the
quantile kernel on whole rows and on log-densities + row ends, CellsLogLikelihood(cells) and
get_raw_lightcurve.  HIP events on the launch stream, mean of ten calls; wall clock for the
two calls that end with a copy to the host."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from godot_amd import core, ops, synth

s = synth.synth(10_000_000, years=10.0, seed=11)
d = core.Data(s['ti'], s['we'], s['TSTART'], s['TSTOP'], s['LIVETIME'], s['exposure'])
cells = d.get_cells(tcell=600., use_barycenter=False, quiet=True)
cr = cells.ranges()
stream = torch.cuda.current_stream()


def events(f, n=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(stream)
        f()
        b.record(stream)
    torch.cuda.synchronize()
    return float(np.mean([a.elapsed_time(b) for a, b in ev]))


def wall(f, n=5):
    f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        f()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3


print('%d cells, %.1f photons a cell' % (cr.ncell, cr.nph / cr.ncell))
print('grid, rows written (cell_logpdf)                         %.3f ms' % events(lambda: ops.cell_logpdf(cr, dlogl=30, npt=200)))
dom, cod, _ = ops.cell_logpdf(cr, dlogl=30, npt=200)
print('quantile kernel on whole rows                            %.3f ms' % events(lambda: ops.profile_quantiles(dom, cod, 0.159, 0.841)))
print('quantile kernel on log-densities and row ends            %.3f ms' %
      events(lambda: ops.profile_quantiles(dom, cod, 0.159, 0.841, linspace=True)))
del dom, cod
print('CellsLogLikelihood(cells)                                %.3f ms (wall clock)' % wall(lambda: core.CellsLogLikelihood(cells)))
cll = core.CellsLogLikelihood(cells)
print('get_raw_lightcurve()                                     %.3f ms (wall clock, rows copied to the host)' % wall(cll.get_raw_lightcurve))
print('both                                                     %.3f ms (wall clock)' %
      wall(lambda: core.CellsLogLikelihood(cells).get_raw_lightcurve()))
