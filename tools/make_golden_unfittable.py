"""Finds, with the CPU oracle alone, the cell of a randomised light curve on which the reference's
get_max ends in its root finder without a bracket (ValueError, core.py:728-739) and stores it as a fixture:
weights, S, B and the guess with which get_max was called (1 from get_logpdf, the profile's peak from get_flux)."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import godot_oracle as orc
T0 = 2.4e8
out = []
for conf in (5, 8):
    rng = np.random.default_rng(5200 * 1 + conf)
    nrow = int(rng.choice([3000, 30000]))
    lens = np.full(nrow, 30.)
    gaps = np.where(rng.random(nrow) < 0.05, rng.choice([30., 900., 2400., 40000.], nrow), 0.0)
    TSTART = T0 + np.cumsum(gaps) + np.append(0, np.cumsum(lens[:-1]))
    TSTOP = TSTART + lens
    LIVETIME = lens * rng.uniform(0.7, 0.95, nrow)
    exposure = LIVETIME * (3000 + 2000 * np.sin(2 * np.pi * (TSTART - T0) / 5700) ** 2)
    span = TSTOP[-1] - TSTART[0]
    nph = int(rng.choice([3000, 30000, 200000]))
    u = rng.random(nph)
    flare = rng.random(nph) < 0.4
    tt = np.where(flare, 1 / 3 + u / 3, u)
    ti = np.sort(TSTART[0] + span * tt)
    mid = (ti > TSTART[0] + span / 3) & (ti < TSTART[0] + 2 * span / 3)
    we = np.clip(np.where(mid & (rng.random(nph) < 0.5), rng.beta(3, 1, nph), rng.beta(0.3, 1.5, nph)), 1e-6, 1 - 1e-6)
    ncell_goal = int(rng.choice([20, 150, 600]))
    tcell = span / (ncell_goal + 0.37)
    S, B, E = np.sum(we), len(we) - np.sum(we), np.sum(exposure)
    b = orc.bin_cells(ti, we, TSTART, TSTOP, exposure, LIVETIME, S, B, E, tcell=tcell)
    clls = orc.cells_from_bins(b)
    with np.errstate(all='ignore'):
        for i, c in enumerate(clls):
            guess = 1.0
            try:
                dom, cod = c.get_logpdf(npt=200, dlogl=30)      # starts with get_max() (core.py:855)
                guess = dom[np.argmax(cod)]
                c.get_max(guess=guess, beta=1)
            except ValueError as e:
                print('conf', conf, 'cell', i, 'n', len(c.we), 'S', c.S, 'B', c.B, 'guess', guess, e)
                out.append((c.we.copy(), c.S, c.B, guess))
d = {}
for k, (w, S, B, g) in enumerate(out):
    d['we%d' % k], d['S%d' % k], d['B%d' % k], d['guess%d' % k] = w, S, B, g
d['n'] = len(out)
np.savez(os.path.join(ROOT, 'tests', 'golden', 'unfittable_cells.npz'), **d)
print('saved', len(out))
