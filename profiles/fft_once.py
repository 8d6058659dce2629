#!/usr/bin/env python3
"""power_spectrum_fft at the C3 size (3 786 912 bins of 100 s -> l = 2^25), three calls, for
kernel-trace and counter passes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from godot_amd import ops

dev = torch.device('cuda:0')
nb, l = 3_786_912, 1 << 25
g = torch.Generator(device=dev)
g.manual_seed(13)
cnt = torch.poisson(torch.full((nb,), 2.6, dtype=torch.float64, device=dev), generator=g)
W, WW = cnt * 0.17, cnt * 0.08
S = torch.full((nb,), 0.44, dtype=torch.float64, device=dev)
B = S * 5
Wb, WbWb = cnt - W, cnt - 2 * W + WW
for _ in range(3):
    ops.power_spectrum_fft(W, WW, S, Wb, WbWb, B, l)
torch.cuda.synchronize()
print('algorithmic bytes per call', 5 * (8 * nb + 16 * (l // 4 + 1)))
