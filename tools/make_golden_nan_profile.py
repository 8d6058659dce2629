"""Finds, with the CPU oracle alone, the cell of a randomised light curve whose profile the reference fills
with NaN without raising (alpha-hat = 0; the Newton search for the upper end of the support, core.py:869-880,
steps to a flux where a factor of the likelihood is negative) and stores it as a fixture: weights, S, B, cell
edges and the row of get_raw_lightcurve the reference's arithmetic gives for it."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import godot_oracle as orc
T0 = 2.4e8
rng = np.random.default_rng(5200 * 41 + 5)
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
    dom, cod = orc.cells_logpdf(clls)
    ref = orc.raw_lightcurve(clls, b['starts'], b['stops'], dom, cod, tsmin=4)
idx = np.flatnonzero(np.isnan(dom).all(axis=1))
assert len(idx) == 1, idx
i = int(idx[0])
sel = slice(i - 1, i + 2)          # with its two neighbours, which are ordinary cells
np.savez(os.path.join(ROOT, 'tests', 'golden', 'nan_profile_cell.npz'),
         we=np.concatenate([c.we for c in clls[sel]]), counts=np.array([len(c.we) for c in clls[sel]]),
         S=np.array([c.S for c in clls[sel]]), B=np.array([c.B for c in clls[sel]]),
         starts=b['starts'][sel], stops=b['stops'][sel], rows=ref[sel], dom=dom[sel], cod=cod[sel])
print('cell', i, 'of', len(clls), 'rows', ref[sel])
