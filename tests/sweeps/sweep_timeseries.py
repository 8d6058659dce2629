"""Randomised comparison of power_spectrum_fft / power_spectrum_simple (through the reference-shaped API, on time
series binned on the device) with the oracle's numpy transforms."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from godot_amd import core
from oracle import godot_oracle as orc
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
nconf = int(sys.argv[2]) if len(sys.argv) > 2 else 40
RTOL = 1e-9
def n_(t): return t.cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
def close(got, ref):
    got, ref = np.asarray(got), np.asarray(ref)
    if got.shape != ref.shape: return 'shape %s vs %s' % (got.shape, ref.shape)
    fin = np.isfinite(ref)
    if not np.array_equal(np.isfinite(got), fin): return 'finite masks differ at %s' % np.argwhere(np.isfinite(got) != fin)[:3].ravel()
    if not fin.any(): return None
    tol = RTOL * np.maximum(np.abs(ref[fin]), 1e-3 * np.abs(ref[fin]).max())
    e = np.abs(got[fin] - ref[fin])
    return None if (e <= tol).all() else 'max err %.2e (tol %.2e)' % (e.max(), tol[np.argmax(e)])
nbad = 0
T0 = 2.4e8
for conf in range(nconf):
    rng = np.random.default_rng(3100 * seed0 + conf)
    nrow = int(rng.choice([300, 3000, 30000]))
    lens = np.full(nrow, 30.)
    gaps = np.where(rng.random(nrow) < 0.05, rng.choice([30., 900., 2400.], nrow), 0.0)
    TSTART = T0 + np.cumsum(gaps) + np.append(0, np.cumsum(lens[:-1]))
    TSTOP = TSTART + lens
    LIVETIME = lens * rng.uniform(0.7, 0.95, nrow)
    exposure = LIVETIME * (3000 + 2000 * np.sin(2 * np.pi * (TSTART - T0) / 5700) ** 2)
    span = TSTOP[-1] - TSTART[0]
    nph = int(rng.choice([200, 20000, 300000]))
    ti = np.sort(TSTART[0] + span * rng.random(nph))
    we = np.clip(rng.beta(0.3, 1.5, nph), 1e-6, 1 - 1e-6)
    d = core.Data(ti, we, TSTART, TSTOP, LIVETIME, exposure)
    tcell = float(rng.choice([30., 100., 300., 1000., span / 37.3]))
    ts = d.get_cells(tcell=tcell, time_series_only=True, trim_zero_exposure=False, use_barycenter=False, quiet=True)
    args = [n_(ts.starts), n_(ts.stops), n_(ts.counts), n_(ts.weights), n_(ts.weights2), n_(ts.sexp), n_(ts.bexp)]
    n = len(args[0])
    kw = {}
    if rng.random() < 0.3: kw['tweak_exp'] = True
    if rng.random() < 0.2: kw['exp_only'] = True
    if rng.random() < 0.3: kw['no_zero_pad'] = True
    if rng.random() < 0.2: kw['dfgoal'] = float(rng.choice([0.05, 1.0]) / span)
    if rng.random() < 0.2: kw['maxN'] = int(max(4, n * rng.uniform(0.3, 1.0)))
    if rng.random() < 0.2 and 'exp_only' not in kw: kw['get_amps'] = True
    if rng.random() < 0.2: kw['exposure_correction'] = rng.uniform(0.8, 1.2, n)
    what = []
    with np.errstate(all='ignore'):
        try:
            ref = orc.power_spectrum_fft(*args, **kw); oerr = None
        except Exception as e:
            oerr = e
    try:
        got = core.power_spectrum_fft(ts, **kw); gerr = None
    except Exception as e:
        gerr = e
    if oerr is not None or gerr is not None:
        if (oerr is None) != (gerr is None): what.append('errors differ: oracle %r here %r' % (oerr, gerr))
    else:
        if len(got) != len(ref): what.append('tuple length')
        else:
            if not np.array_equal(got[0], ref[0]): what.append('freqs')
            for k in range(1, len(ref)):
                r = close(got[k], ref[k])
                if r: what.append('out[%d]: %s' % (k, r))
    with np.errstate(all='ignore'):
        rf, rdl = orc.power_spectrum_simple(args[0], args[1], args[3], args[4], args[5])
    f, dl = core.power_spectrum_simple(ts)
    if not np.array_equal(f, rf): what.append('simple freqs')
    r = close(dl, rdl)
    if r: what.append('simple: %s' % r)
    nbad += bool(what)
    print('conf %d n %d nph %d tcell %g kw %s: %s' % (conf, n, nph, tcell, {k: (v if np.isscalar(v) else '...') for k, v in kw.items()}, what or 'ok'), flush=True)
print('TOTAL BAD', nbad)
