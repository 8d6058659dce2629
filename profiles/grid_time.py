#!/usr/bin/env python3
"""Times the 200-point profile grid on the 1e7 C2 photons in cells of several lengths
(HIP events on the launch stream, mean of ten calls).  Usage: grid_time.py [tcell ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from godot_amd import core, ops, synth

s = synth.synth(10_000_000, years=10.0, seed=11)
d = core.Data(s['ti'], s['we'], s['TSTART'], s['TSTOP'], s['LIVETIME'], s['exposure'])
stream = torch.cuda.current_stream()
for tcell in [float(x) for x in sys.argv[1:]] or [600.]:
    cr = d.get_cells(tcell=tcell, use_barycenter=False, quiet=True).ranges()
    for _ in range(3):
        out = ops.cell_logpdf(cr, dlogl=30, npt=200)
    del out
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for a, b in ev:
        a.record(stream)
        out = ops.cell_logpdf(cr, dlogl=30, npt=200)
        b.record(stream)
        del out
    torch.cuda.synchronize()
    ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    print('tcell %g: %d cells, %.1f photons a cell: grid %.3f ms' % (tcell, cr.ncell, cr.nph / cr.ncell, ms), flush=True)
