#!/usr/bin/env python3
"""Two launches of the Z^2_2 search exactly as bench.py's C4 leg issues them (1e6 photons of
one year, 2^21 trial frequencies a launch) for a counter pass; writes the number of
photon x frequency evaluations of one launch to the file named on the command line."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from godot_amd import ops, synth

s = synth.synth(1_000_000, years=1.0, seed=17)
t = torch.from_numpy(s['ti'] - s['ti'][0]).cuda()
w = torch.from_numpy(s['we']).cuda()
nf = 1 << 21
span = float(s['ti'][-1] - s['ti'][0])
ws = torch.empty(ops._lib.lib.godot_z2m_workspace_bytes(t.numel(), nf, 2), dtype=torch.uint8, device='cuda')
for j0 in (0, nf):
    ops.z2m(t, w, 1e-4, 1.0 / (4 * span), nf, m=2, j0=j0, workspace=ws)
torch.cuda.synchronize()
json.dump({'evals_per_launch': float(t.numel()) * nf, 'nph': t.numel(), 'nf': nf, 'm': 2, 'launches': 2},
          open(sys.argv[1], 'w'))
