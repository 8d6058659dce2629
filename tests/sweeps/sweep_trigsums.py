"""Randomised comparison of power_spectrum_dft / fit_harmonics / compute_ls / get_orbital_modulation with the oracle."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from godot_amd import core
from oracle import godot_oracle as orc
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
nconf = int(sys.argv[2]) if len(sys.argv) > 2 else 40
d = torch.device('cuda:0')
def rel(a, b, floor):
    a, b = np.asarray(a), np.asarray(b)
    ok = (np.abs(a - b) <= 1e-9 * np.maximum(np.abs(b), floor)) | (~np.isfinite(a) & ~np.isfinite(b))
    return None if ok.all() else 'max err %.2e at %s' % (np.nanmax(np.abs(a - b)), np.argwhere(~ok)[:3].ravel())
nbad = 0
for conf in range(nconf):
    rng = np.random.default_rng(8800 * seed0 + conf)
    ncell = int(rng.choice([2, 3, 17, 200, 3000]))
    tcell = float(rng.choice([600., 86400., 7200.]))
    uneven = rng.random() < 0.4
    starts = 2.4e8 + np.arange(ncell) * tcell
    if uneven:
        starts = starts + np.cumsum(rng.choice([0, 0, tcell, 3 * tcell], ncell))
    stops = starts + tcell
    counts = rng.poisson(rng.choice([0.5, 5, 60]), ncell).astype(np.int64)
    if counts.sum() == 0: counts[0] = 3
    off = np.append(0, np.cumsum(counts)).astype(np.int64)
    we = np.clip(rng.beta(0.3, 1.5, int(off[-1])), 1e-6, 1 - 1e-6)
    exp = counts.mean() * 0.2 * rng.uniform(0.5, 1.5, ncell)
    nfreq = int(rng.choice([1, 5, 64, 300]))
    freqs = np.sort(rng.uniform(1e-7, 0.5 / tcell, nfreq))
    offd = torch.from_numpy(off).to(d)
    wd = torch.from_numpy(we).to(d)
    batch = core.CellBatch(wd, wd, offd[:-1], offd[1:], torch.from_numpy(starts).to(d), torch.from_numpy(stops).to(d),
                           torch.from_numpy(exp).to(d), torch.from_numpy(exp * 5).to(d))
    old = core.CellsLogLikelihoodOld(batch)
    flat = orc.flatten_cells(we, off, exp, starts, stops)
    what = []
    with np.errstate(all='ignore'):
        for unw in (False, True):
            ref = orc.power_spectrum_dft(flat, freqs, unweighted=unw)
            got = core.power_spectrum_dft(old, freqs, unweighted=unw)
            fl = np.nanmax(np.abs(ref[np.isfinite(ref)])) * 1e-6 if np.isfinite(ref).any() else 0
            r = rel(got, ref, fl)
            if r: what.append('dft unw=%d: %s' % (unw, r))
            ref = orc.compute_ls(flat, counts, freqs, unweighted=unw)
            got = core.compute_ls(old, freqs, unweighted=unw)
            fl = np.nanmax(np.abs(ref[np.isfinite(ref)])) * 1e-6 if np.isfinite(ref).any() else 0
            r = rel(got, ref, fl)
            if r: what.append('ls unw=%d: %s' % (unw, r))
        ref = np.array([orc.fit_harmonics(flat, f) for f in freqs[:16]])
        got = core.fit_harmonics(old, freqs[:16])
        fin = np.isfinite(ref)
        fl = np.where(fin, np.abs(ref), 0).max(axis=0, keepdims=True) * 1e-6
        r = rel(got, ref, fl)
        if r: what.append('harmonics: %s' % r)
    nbad += bool(what)
    print('conf %d ncell %d uneven %d nfreq %d: %s' % (conf, ncell, uneven, nfreq, what or 'ok'), flush=True)
print('TOTAL BAD', nbad)
