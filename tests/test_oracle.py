"""Pins the CPU oracle to outputs of the reference implementation
(tests/golden/*.npz, written by tools/make_golden.py) and to scipy."""
import numpy as np
import pytest
import scipy.optimize

from oracle import godot_oracle as orc


def _cells(g):
    off = np.append(0, np.cumsum(g['counts']))
    return [orc.CellLike(g['we'][off[i]:off[i + 1]], g['exp'][i], g['SonB'][i])
            for i in range(len(g['counts']))], off


def test_brentq_matches_scipy():
    rng = np.random.default_rng(0)
    for k in range(200):
        c = rng.uniform(-3, 3, 4)
        f = lambda x: c[0] + c[1] * x + c[2] * x ** 3 + np.tanh(c[3] * x) - 0.1
        a, b = -5., 7.
        if np.sign(f(a)) == np.sign(f(b)):
            continue
        for xtol in (1e-3, 2e-12):
            assert orc.brentq(f, a, b, xtol=xtol) == \
                scipy.optimize.brentq(f, a, b, xtol=xtol)


@pytest.mark.parametrize('name', ['cells_dense.npz', 'cells_sparse.npz',
                                  'cells_edge.npz'])
def test_fit_ts_logpdf_match_reference(golden, name):
    g = golden(name)
    clls, off = _cells(g)
    with np.errstate(all='ignore'):
        for i, c in enumerate(clls):
            if np.isnan(g['alpha'][i]):
                with pytest.raises(ValueError):
                    c.get_max()
                continue
            a = c.get_max()
            assert a == g['alpha'][i]
            assert c.get_ts(a) == g['ts'][i]
            assert c.log_likelihood(1.3) == g['logl_1p3'][i]
        for i in range(len(g['dom'])):
            if np.isnan(g['alpha'][i]):
                continue
            dom, cod = clls[i].get_logpdf(npt=200, dlogl=30)
            assert np.array_equal(dom, g['dom'][i])
            assert np.array_equal(cod, g['cod'][i])


def test_guess_restarts_match_reference(golden):
    g = golden('cells_edge.npz')
    clls, _ = _cells(g)
    with np.errstate(all='ignore'):
        for i, c in enumerate(clls):
            for j, gs in enumerate(g['guesses']):
                ref = g['alpha_from_guess'][i, j]
                if np.isnan(ref):
                    continue
                assert c.get_max(guess=gs) == ref


@pytest.mark.parametrize('name,n', [('cells_dense.npz', None),
                                    ('cells_sparse.npz', 300)])
def test_lightcurve_matches_reference(golden, name, n):
    g = golden(name)
    clls, _ = _cells(g)
    n = n or len(clls)
    with np.errstate(all='ignore'):
        lc = orc.raw_lightcurve(clls[:n], g['starts'][:n], g['stops'][:n],
                                g['dom'][:n], g['cod'][:n])
    assert np.array_equal(lc, g['lightcurve'], equal_nan=True)


@pytest.mark.parametrize('name', ['cells_dense.npz', 'cells_sparse.npz'])
def test_binning_matches_reference(golden, name):
    g = golden(name)
    b = orc.bin_cells(g['in_ti'], g['in_we'], g['in_TSTART'], g['in_TSTOP'],
                      g['in_exposure'], g['in_LIVETIME'], float(g['in_S']),
                      float(g['in_B']), float(g['in_E']),
                      tcell=float(g['tcell']))
    assert np.array_equal(b['starts'], g['starts'])
    assert np.array_equal(b['stops'], g['stops'])
    assert np.array_equal(b['counts'], g['counts'])
    assert np.array_equal(b['weights'], g['we'])
    assert np.array_equal(b['times'], g['ti'])
    assert np.array_equal(b['sexp'], g['exp'])
    assert np.array_equal(b['sexp'] / b['bexp'], g['SonB'])


def test_timeseries_and_fft_match_reference(golden):
    g = golden('timeseries_fft.npz')
    b = orc.bin_cells(g['in_ti'], g['in_we'], g['in_TSTART'], g['in_TSTOP'],
                      g['in_exposure'], g['in_LIVETIME'], float(g['in_S']),
                      float(g['in_B']), float(g['in_E']),
                      tcell=float(g['tcell']), trim_zero_exposure=False)
    orc.zero_low_exposure(b, 0)
    for k, gk in (('starts', 'starts'), ('stops', 'stops'), ('exp', 'exp'),
                  ('sexp', 'sexp'), ('bexp', 'bexp'), ('counts', 'counts'),
                  ('wsum', 'weights'), ('w2sum', 'weights2'),
                  ('dead', 'deadtime')):
        assert np.array_equal(b[k], g[gk]), k
    args = (b['starts'], b['stops'], b['counts'], b['wsum'], b['w2sum'],
            b['sexp'], b['bexp'])
    f, p0, p1, pb = orc.power_spectrum_fft(*args)
    assert np.array_equal(f, g['ps_freqs'])
    for x, k in ((p0, 'ps_p0'), (p1, 'ps_p1'), (pb, 'ps_pb')):
        assert np.array_equal(x, g[k], equal_nan=True)
    amps = orc.power_spectrum_fft(*args, get_amps=True)
    assert np.array_equal(np.array(amps[1:]), g['ps_amps'], equal_nan=True)
    tw = orc.power_spectrum_fft(*args, tweak_exp=True)
    assert np.array_equal(np.array(tw[1:]), g['ps_tweak'], equal_nan=True)
    eo = orc.power_spectrum_fft(*args, exp_only=True)
    assert np.array_equal(eo[1], g['ps_exp_only'], equal_nan=True)
    nz = orc.power_spectrum_fft(*args, no_zero_pad=True)
    assert np.array_equal(nz[0], g['ps_nzp_freqs'])
    assert np.array_equal(np.array(nz[1:]), g['ps_nzp'], equal_nan=True)
    omr, omp = orc.get_orbital_modulation(b['starts'], b['stops'], b['wsum'], b['w2sum'], b['sexp'],
                                          g['om_freqs'])
    assert np.array_equal(omr, g['om_rvals']) and np.array_equal(omp, g['om_pows'])
    pc = orc.power_spectrum_fft(*args, exposure_correction=omr)
    assert np.array_equal(np.array(pc[1:]), g['ps_corr'], equal_nan=True)
    f, dl = orc.power_spectrum_simple(b['starts'], b['stops'], b['wsum'],
                                      b['w2sum'], b['sexp'])
    assert np.array_equal(f, g['pss_freqs'])
    assert np.array_equal(dl, g['pss_dlogl'], equal_nan=True)


def test_trig_sums_match_reference(golden):
    g = golden('trig_sums.npz')
    off = np.append(0, np.cumsum(g['counts']))
    flat = orc.flatten_cells(g['we'], off, g['exp'], g['starts'], g['stops'])
    with np.errstate(all='ignore'):
        assert np.array_equal(orc.power_spectrum_dft(flat, g['freqs']), g['dft'],
                              equal_nan=True)
        assert np.array_equal(orc.power_spectrum_dft(flat, g['freqs'], True),
                              g['dft_unweighted'], equal_nan=True)
        fh = np.array([orc.fit_harmonics(flat, f) for f in g['freqs'][:16]])
    assert np.array_equal(fh, g['harmonics'], equal_nan=True)
    assert np.array_equal(orc.compute_ls(flat, g['counts'], g['freqs']), g['ls'])
    assert np.array_equal(orc.compute_ls(flat, g['counts'], g['freqs'], True), g['ls_unweighted'])


def test_exposure_matches_reference(golden):
    g = golden('exposure.npz')
    a = (g['in_TSTART'], g['in_TSTOP'], g['in_exposure'], g['in_LIVETIME'],
         float(g['in_S']), float(g['in_B']), float(g['in_E']), g['times'])
    c = orc.exposure_at(*a, deadtime_too=True, cumulative=True)
    assert np.array_equal(np.array(c), g['cumulative'])
    d = orc.exposure_at(*a, deadtime_too=True, cumulative=False)
    assert np.array_equal(np.array(d), g['differential'])


def test_profile_background_matches_reference(golden):
    g = golden('profile_bkg.npz')
    clls, _ = _cells(g)
    with np.errstate(all='ignore'):
        for i, c in enumerate(clls):
            for j, al in enumerate(g['alphas']):
                ref = g['beta_max'][i, j]
                if np.isnan(ref):
                    with pytest.raises(ValueError):
                        c.get_beta_max(al)
                    continue
                assert c.get_beta_max(al) == ref
                assert c.log_profile_likelihood(al) == g['logl_profile'][i, j]
            ab = g['joint_ab'][i]
            assert c.log_full_likelihood(ab[0], ab[1]) == g['joint_logl'][i]
            # the joint fit goes through scipy's fmin_tnc with the reference's call sequence
            assert np.array_equal(np.asarray(c.get_max_profile(), dtype=float), ab)
        for i in range(len(g['prof_dom'])):
            if np.isnan(g['prof_dom'][i, 0]):
                continue
            dom, cod = clls[i].get_logpdf_profile(aopt=float(g['joint_ab'][i, 0]), dlogl=30, npt=50)
            assert np.array_equal(dom, g['prof_dom'][i])
            assert np.array_equal(cod, g['prof_cod'][i])


def test_empty_cell_known_answers():
    c = orc.CellLike(np.array([]), 3.0, 0.2)
    assert c.get_max() == 0 and c.get_ts(0) == 0 and c.get_beta_max(1.0) == 0
    dom, cod = c.get_logpdf(npt=200, dlogl=30)
    assert np.allclose(dom, np.linspace(0, 10, 200), rtol=0, atol=1e-12)
    assert np.allclose(cod, -3.0 * dom, rtol=0, atol=1e-12)


def test_z2m_against_exact_phases():
    rng = np.random.default_rng(1)
    t = np.sort(rng.uniform(0, 3.0e8, 150))
    w = rng.uniform(0.01, 0.99, 150)
    f0, df = 1.234567e-3, 7.7e-10
    j = np.array([0, 1, 17, 123456, 99999999])
    C, S = orc.z2m_trig_sums(t, w, f0, df, j, 2)
    z = 2. / np.sum(w ** 2) * np.sum(C ** 2 + S ** 2, axis=0)
    for n, jj in enumerate(j):
        ze = orc.z2m_exact(t, w, f0, df, jj, 2)
        assert abs(z[n] - ze) <= 1e-11 * max(1., abs(ze))


def test_bayesian_blocks_restatement_equals_reference(golden):
    """fitness / do_bb / do_top_hat / get_bb_lightcurve (core.py:1108-1225, 1338-1389)."""
    g = golden('bayesian_blocks.npz')
    dom, cod = g['dom'], g['cod']
    n = len(dom)
    idx8, total8, vts8, vdof8, rows = orc.do_bb(dom, cod, prior=8)
    assert np.array_equal(np.concatenate([r[0] for r in rows]), g['fit_val'])
    assert np.array_equal(np.concatenate([r[1] for r in rows]), g['fit_arg'])
    assert np.array_equal(idx8, g['bb8_indices'])
    assert np.array_equal(np.array([total8, vts8, vdof8], dtype=float), g['bb8_scalars'])
    idx2, best2, _, _ = orc.bb_partition([r[0] for r in rows], prior=2)
    assert np.array_equal(idx2, g['bb2_indices'])
    assert best2 + len(idx2) * 2 == g['bb2_scalars'][0]
    assert np.array_equal(orc.bb_fitness(dom, cod, 5, 40), g['fitness_5_40'])
    off = np.append(0, np.cumsum(g['counts']))
    cells = [(g['starts'][i], g['stops'][i], g['exp'][i], g['ti'][off[i]:off[i + 1]],
              g['we'][off[i]:off[i + 1]], g['SonB'][i]) for i in range(n)]
    with np.errstate(all='ignore'):
        lc, edges = orc.bb_lightcurve(cells, dom, cod, tsmin=4, bb_prior=8)
    assert np.array_equal(lc, g['bb_rows'], equal_nan=True)
    assert np.array_equal(edges, g['bb_idx'])
    # do_top_hat (core.py:1210-1225): three diagonals of the two maps
    top = np.empty((2, n, n))
    ix = np.arange(n)
    for iv, v in enumerate(rows):
        top[:, ix[:iv + 1], ix[:iv + 1] + n - (iv + 1)] = v
        top[:, ix[:iv + 1] + n - (iv + 1), ix[:iv + 1]] = np.nan
    top[0] += (np.arange(n) * 2)[None, ::-1]
    top[1] *= 0.5
    for k in (0, 1):
        for j, o in enumerate((0, 1, 7)):
            assert np.array_equal(np.diagonal(top[k], offset=o)[: n - 7], g['top_hat_diag'][k, j], equal_nan=True)

