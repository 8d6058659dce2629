#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the reference implementation.

Runs ONLY in the build container (needs /root/reference).  The reference is
imported as the package ``reference`` with empty stand-in modules for
``astropy`` (not installed; the hot path never calls it).  A ``Data`` object
is made with ``Data.__new__`` and its array attributes set by hand, which
bypasses the FITS constructor.  Only data (inputs and the reference's
outputs) is written; no reference source is stored.

Usage:  python tools/make_golden.py [outdir]
"""
import contextlib
import io
import os
import sys
import types
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
sys.path.insert(0, '/root')
os.environ.setdefault('MPLBACKEND', 'Agg')


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def import_reference():
    a = _stub('astropy')
    aio = _stub('astropy.io')
    fits = _stub('astropy.io.fits')
    col = _stub('astropy.io.fits.column', _ColumnFormat=object)
    a.io = aio
    aio.fits = fits
    fits.column = col
    for n in ('constants', 'coordinates', 'time', 'units'):
        setattr(a, n, _stub('astropy.' + n))
    import reference.core as core
    return core


def make_data(core, d):
    obj = core.Data.__new__(core.Data)
    for k in ('ti', 'we', 'TSTART', 'TSTOP', 'LIVETIME', 'exposure',
              'S', 'B', 'E'):
        v = d[k]
        setattr(obj, k, v.copy() if isinstance(v, np.ndarray) else v)
    obj.sexposure = None
    obj.bexposure = None
    obj.timeref = 'LOCAL'
    obj.ra = obj.dec = 0.
    obj._verbosity = 0
    obj.minimum_exposure = 3e4
    return obj


@contextlib.contextmanager
def quiet():
    with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
        warnings.simplefilter('ignore')
        with np.errstate(all='ignore'):
            yield


def inputs_of(d):
    return {('in_' + k): d[k] for k in
            ('ti', 'we', 'TSTART', 'TSTOP', 'LIVETIME', 'exposure')} | \
        {'in_S': d['S'], 'in_B': d['B'], 'in_E': d['E']}


def cell_outputs(core, cells, logpdf_cells=None, lightcurve=False):
    """Per-cell outputs of the reference's likelihood objects."""
    n = len(cells)
    out = dict(
        starts=np.array([c.tstart for c in cells]),
        stops=np.array([c.tstop for c in cells]),
        exp=np.array([c.exp for c in cells]),
        SonB=np.array([c.SonB for c in cells]),
        counts=np.array([len(c.we) for c in cells], dtype=np.int64),
        we=np.concatenate([c.we for c in cells]) if n else np.empty(0),
        ti=np.concatenate([c.ti for c in cells]) if n else np.empty(0),
    )
    amax = np.empty(n)
    ts = np.empty(n)
    ll1 = np.empty(n)
    for i, c in enumerate(cells):
        cl = core.CellLogLikelihood(c)
        ll1[i] = cl.log_likelihood(1.3)
        try:
            a = cl.get_max()
        except ValueError:       # reference's own brentq bracket failure
            amax[i] = ts[i] = np.nan
            continue
        amax[i] = a
        ts[i] = cl.get_ts(aopt=a)
    out.update(alpha=amax, ts=ts, logl_1p3=ll1)
    if logpdf_cells is None:
        logpdf_cells = n
    dom = np.empty((logpdf_cells, 200))
    cod = np.empty((logpdf_cells, 200))
    for i in range(logpdf_cells):
        cl = core.CellLogLikelihood(cells[i])
        if np.isnan(amax[i]):
            dom[i] = cod[i] = np.nan
            continue
        dom[i], cod[i] = cl.get_logpdf(npt=200, dlogl=30)
    out.update(dom=dom, cod=cod)
    if lightcurve:
        cll = core.CellsLogLikelihood(cells)
        out['lightcurve'] = cll.get_raw_lightcurve()
        assert np.array_equal(cll._dom, dom) and np.array_equal(cll._cod, cod)
    return out


def main(outdir):
    from oracle import godot_oracle as orc
    core = import_reference()
    os.makedirs(outdir, exist_ok=True)

    # ---- G1: dense cells (1-hour cells, 20 days) -------------------------
    d = orc.synth(40000, years=20 / 365.25, seed=7)
    data = make_data(core, d)
    with quiet():
        cells = data.get_cells(tcell=3600., use_barycenter=False)
        o = cell_outputs(core, cells, lightcurve=True)
    np.savez_compressed(os.path.join(outdir, 'cells_dense.npz'),
                        tcell=3600., **inputs_of(d), **o)
    print('cells_dense: %d cells, %d photons' % (len(cells), len(o['we'])))

    # ---- G2: sparse cells (10-minute cells; many restarts) ---------------
    d = orc.synth(6000, years=20 / 365.25, seed=11)
    data = make_data(core, d)
    with quiet():
        cells = data.get_cells(tcell=600., use_barycenter=False)
        o = cell_outputs(core, cells, logpdf_cells=300, lightcurve=False)
        # light curve on a slice (uses its own profile cache)
        sub = cells[:300]
        cll = core.CellsLogLikelihood(sub)
        o['lightcurve'] = cll.get_raw_lightcurve()
    np.savez_compressed(os.path.join(outdir, 'cells_sparse.npz'),
                        tcell=600., **inputs_of(d), **o)
    print('cells_sparse: %d cells, %d photons' % (len(cells), len(o['we'])))

    # ---- G3: hand-made edge cells ---------------------------------------
    rng = np.random.default_rng(3)
    edge_cells = []
    edge_cells.append(core.Cell(0., 10., 3.0, np.array([]), np.array([]), 0.2))  # empty
    edge_cells.append(core.Cell(0., 10., 0.5, np.array([1.]), np.array([0.9]), 0.2))
    edge_cells.append(core.Cell(0., 10., 0.01, np.array([1., 2.]),
                                np.array([1 - 1e-6, 1 - 1e-6]), 0.2))  # strong source
    edge_cells.append(core.Cell(0., 10., 50., np.arange(5.),
                                np.full(5, 1e-6), 0.2))               # all background
    w = np.clip(rng.beta(0.3, 1.5, 700), 1e-6, 1 - 1e-6)
    edge_cells.append(core.Cell(0., 10., w.sum() * 6, np.arange(700.), w, 0.2))   # alpha ~ 1/6
    edge_cells.append(core.Cell(0., 10., w.sum() / 40, np.arange(700.), w, 0.2))  # alpha ~ 40 (restart)
    w = np.clip(rng.beta(0.3, 1.5, 5000), 1e-6, 1 - 1e-6)
    edge_cells.append(core.Cell(0., 10., w.sum(), np.arange(5000.), w, 0.2))      # big cell
    w = np.clip(rng.beta(2, 2, 64), 1e-6, 1 - 1e-6)
    edge_cells.append(core.Cell(0., 10., w.sum() * 1.02, np.arange(64.), w, 0.2))
    w = np.clip(rng.beta(2, 2, 65), 1e-6, 1 - 1e-6)
    edge_cells.append(core.Cell(0., 10., w.sum() / 300, np.arange(65.), w, 0.2))  # alpha ~ 300
    w = np.clip(rng.beta(0.3, 1.5, 513), 1e-6, 1 - 1e-6)
    edge_cells.append(core.Cell(0., 10., w.sum() / 3000, np.arange(513.), w, 0.2))  # alpha ~ 3000
    with quiet():
        o = cell_outputs(core, edge_cells, lightcurve=False)
        # restart statistics: run get_max from different guesses
        g_list = np.array([0.05, 0.5, 3.0, 30.0])
        ag = np.empty((len(edge_cells), len(g_list)))
        for i, c in enumerate(edge_cells):
            for j, g in enumerate(g_list):
                try:
                    ag[i, j] = core.CellLogLikelihood(c).get_max(guess=g)
                except ValueError:
                    ag[i, j] = np.nan
        o['guesses'] = g_list
        o['alpha_from_guess'] = ag
    np.savez_compressed(os.path.join(outdir, 'cells_edge.npz'), **o)
    print('cells_edge: %d cells' % len(edge_cells))

    # ---- G4: background profile (get_beta_max, log_profile_likelihood) ---
    d = orc.synth(40000, years=20 / 365.25, seed=7)
    data = make_data(core, d)
    with quiet():
        cells = data.get_cells(tcell=7200., use_barycenter=False)[:60]
        o = cell_outputs(core, cells, logpdf_cells=0)
        alphas = np.array([0., 0.05, 0.5, 1.0, 1.7, 4.0, 12.0])
        bm = np.full((len(cells), len(alphas)), np.nan)
        lp = np.full((len(cells), len(alphas)), np.nan)
        for i, c in enumerate(cells):
            cl = core.CellLogLikelihood(c)
            for j, al in enumerate(alphas):
                try:
                    bm[i, j] = cl.get_beta_max(al)
                    lp[i, j] = cl.log_profile_likelihood(al)
                except ValueError:
                    pass
        ab = np.empty((len(cells), 2))
        abll = np.empty(len(cells))
        for i, c in enumerate(cells):
            cl = core.CellLogLikelihood(c)
            r = np.asarray(cl.get_max(profile_background=True), dtype=float)
            ab[i] = r
            abll[i] = cl.log_full_likelihood(list(r))
        # profile-likelihood grids centred on the reference's own joint maxima
        # (given aopt the routine is deterministic: no third-party optimiser inside)
        nprof = 16
        pdom = np.full((nprof, 50), np.nan)
        pcod = np.full((nprof, 50), np.nan)
        for i in range(nprof):
            cl = core.CellLogLikelihood(cells[i])
            try:
                pdom[i], pcod[i] = cl.get_logpdf(aopt=float(ab[i, 0]), dlogl=30, npt=50,
                                                 profile_background=True)
            except ValueError:
                pass
        # and the full profile-background driver (optimiser inside: compared loosely)
        cllp = core.CellsLogLikelihood(cells[:24], profile_background=True)
        lcp = cllp.get_raw_lightcurve()
    np.savez_compressed(os.path.join(outdir, 'profile_bkg.npz'),
                        alphas=alphas, beta_max=bm, logl_profile=lp,
                        joint_ab=ab, joint_logl=abll, prof_dom=pdom, prof_cod=pcod,
                        prof_lightcurve=lcp, prof_dom_full=cllp._dom, prof_cod_full=cllp._cod,
                        **{k: o[k] for k in ('starts', 'stops', 'exp', 'SonB',
                                             'counts', 'we')})
    print('profile_bkg: %d cells' % len(cells))

    # ---- G5: time series + FFT power spectra ----------------------------
    d = orc.synth(30000, years=12 / 365.25, seed=13)
    data = make_data(core, d)
    with quiet():
        ts = data.get_cells(tcell=100., time_series_only=True,
                            trim_zero_exposure=False, use_barycenter=False)
        o = dict(starts=ts.starts, stops=ts.stops, exp=ts.exp, sexp=ts.sexp,
                 bexp=ts.bexp, counts=ts.counts, weights=ts.weights,
                 weights2=ts.weights2, deadtime=ts.deadtime)
        f, p0, p1, pb = core.power_spectrum_fft(ts)
        o.update(ps_freqs=f, ps_p0=p0, ps_p1=p1, ps_pb=pb)
        amps = core.power_spectrum_fft(ts, get_amps=True)
        o['ps_amps'] = np.array(amps[1:])
        omf = np.array([1. / 5700., 2.3e-5])
        omr, omp = core.get_orbital_modulation(ts, omf)
        o.update(om_freqs=omf, om_rvals=omr, om_pows=omp)
        f, p0c, p1c, pbc = core.power_spectrum_fft(ts, exposure_correction=omr)
        o.update(ps_corr=np.array([p0c, p1c, pbc]))
        f, p0t, p1t, pbt = core.power_spectrum_fft(ts, tweak_exp=True)
        o.update(ps_tweak=np.array([p0t, p1t, pbt]))
        f, pe = core.power_spectrum_fft(ts, exp_only=True)
        o['ps_exp_only'] = pe
        f, p0n, p1n, pbn = core.power_spectrum_fft(ts, no_zero_pad=True)
        o.update(ps_nzp_freqs=f, ps_nzp=np.array([p0n, p1n, pbn]))
        f, dl = core.power_spectrum_simple(ts)
        o.update(pss_freqs=f, pss_dlogl=dl)
        # a power-of-two length series for the simple spectrum
        ts2 = data.get_cells(tstop=d['TSTART'][0] + 8192 * 100., tcell=100.,
                             time_series_only=True, trim_zero_exposure=False,
                             use_barycenter=False)
        f, dl = core.power_spectrum_simple(ts2)
        o.update(pss2_n=len(ts2.starts), pss2_freqs=f, pss2_dlogl=dl)
    np.savez_compressed(os.path.join(outdir, 'timeseries_fft.npz'),
                        tcell=100., **inputs_of(d), **o)
    print('timeseries_fft: %d bins, %d freqs' % (len(ts.starts), len(o['ps_p0'])))

    # ---- G6: direct trig sums -------------------------------------------
    d = orc.synth(20000, years=30 / 365.25, seed=17)
    data = make_data(core, d)
    with quiet():
        cells = data.get_cells(tcell=7200., use_barycenter=False)
        old = core.CellsLogLikelihoodOld(cells)
        freqs = np.linspace(1e-6, 6e-5, 64)
        pd = core.power_spectrum_dft(old, freqs)
        pdu = core.power_spectrum_dft(old, freqs, unweighted=True)
        fh = np.array([core.fit_harmonics(old, f) for f in freqs[:16]])
        ls = core.compute_ls(old, freqs)
        lsu = core.compute_ls(old, freqs, unweighted=True)
    np.savez_compressed(
        os.path.join(outdir, 'trig_sums.npz'), freqs=freqs, dft=pd,
        dft_unweighted=pdu, harmonics=fh, ls=ls, ls_unweighted=lsu,
        starts=np.array([c.tstart for c in cells]),
        stops=np.array([c.tstop for c in cells]),
        exp=np.array([c.exp for c in cells]),
        counts=np.array([len(c.we) for c in cells], dtype=np.int64),
        we=np.concatenate([c.we for c in cells]))
    print('trig_sums: %d cells' % len(cells))

    # ---- G7: exposure lookup --------------------------------------------
    d = orc.synth(1000, years=12 / 365.25, seed=5)
    data = make_data(core, d)
    rng = np.random.default_rng(5)
    tq = np.sort(rng.uniform(d['TSTART'][0], d['TSTOP'][-1], 500))
    tq = np.concatenate([[d['TSTART'][0]], tq, d['TSTOP'][100:104],
                         d['TSTART'][200:204], [d['TSTOP'][-1]]])
    tq.sort()
    with quiet():
        c = data.get_exposure(tq.copy(), deadtime_too=True, cumulative=True)
        data2 = make_data(core, d)
        nc = data2.get_exposure(tq.copy(), deadtime_too=True, cumulative=False)
    np.savez_compressed(os.path.join(outdir, 'exposure.npz'), times=tq,
                        cumulative=np.array(c), differential=np.array(nc),
                        **inputs_of(d))
    print('exposure: %d query times' % len(tq))


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1
         else os.path.join(REPO, 'tests', 'golden'))
