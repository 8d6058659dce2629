#!/usr/bin/env python3
"""The 200-point profile grid of the C2 cells (1e7 photons, ten-minute cells), three calls,
for kernel-trace and counter passes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from godot_amd import core, ops, synth

s = synth.synth(10_000_000, years=10.0, seed=11)
d = core.Data(s['ti'], s['we'], s['TSTART'], s['TSTOP'], s['LIVETIME'], s['exposure'])
cr = d.get_cells(tcell=600., use_barycenter=False, quiet=True).ranges()
for _ in range(3):
    ops.cell_logpdf(cr, dlogl=30, npt=200)
torch.cuda.synchronize()
print('ncell', cr.ncell, 'bytes written per call', 2 * 8 * 200 * cr.ncell)
