#!/usr/bin/env python3
"""One Z^2_2 search (1e6 photons x 2^18 frequencies) for a counter pass; writes the
number of photon x frequency evaluations of the launch to the file named on the command line."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from godot_amd import ops, synth

s = synth.synth(1_000_000, years=10.0, seed=17)
t = torch.from_numpy(s['ti']).cuda()
w = torch.from_numpy(s['we']).cuda()
nf = 512 * 256 * 20          # whole residency rounds: one launch over all photons
span = float(s['ti'][-1] - s['ti'][0])
z, _ = ops.z2m(t, w, 1e-4, 1.0 / (4 * span), nf, m=2)
torch.cuda.synchronize()
json.dump({'evals': float(t.numel()) * nf, 'nph': t.numel(), 'nf': nf, 'm': 2}, open(sys.argv[1], 'w'))
