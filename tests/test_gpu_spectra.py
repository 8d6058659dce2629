"""GPU parity of the FFT power spectra, the direct trig sums and the Z^2_m
search against the reference's golden vectors and the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import godot_oracle as orc

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _n(t):
    return t.cpu().numpy()


def _ts(g):
    from godot_amd import core
    d = torch.device(DEV)
    f = lambda k, dt=torch.float64: torch.from_numpy(np.array(g[k])).to(device=d, dtype=dt)
    return core.CellTimeSeries(f('starts'), f('stops'), f('exp'), f('sexp'), f('bexp'),
                               f('counts', torch.int64), f('weights'), f('weights2'), f('deadtime'),
                               timesys='topocentric', minimum_exposure=0)


@pytest.mark.parametrize('l,n', [(2, 2), (3, 3), (5, 4), (6, 6), (7, 9), (12, 12), (1000, 1000),
                                 (10362, 10362), (99991, 60000), (786432, 786432), (4, 3), (8, 8), (16, 5), (32, 32), (64, 33), (256, 256), (1024, 700),
                                 (4096, 4096), (1 << 13, 5000), (1 << 16, 1 << 16), (1 << 17, 99999),
                                 (1 << 20, 300000), (1 << 14, 1 << 14), (1 << 15, 20000), (1 << 18, 1 << 18), (1 << 19, 400000),
                                 (1 << 21, 1 << 21), (1 << 22, 3000000), (1 << 23, 1 << 23), (1 << 25, 1 << 25)])
def test_rfft_matches_numpy(l, n):
    from godot_amd import ops
    rng = np.random.default_rng(l + n)
    x = rng.normal(size=n) * rng.uniform(0.1, 10)
    out = _n(ops.rfft(torch.from_numpy(x).to(DEV), l))
    ref = np.fft.rfft(x, n=l)
    tol = 1e-9 * np.abs(ref).max()      # the bar: abs <= 1e-9 max|X|
    assert np.abs(out[:, 0] - ref.real).max() <= tol and np.abs(out[:, 1] - ref.imag).max() <= tol
    # in practice far tighter
    assert np.abs(out[:, 0] + 1j * out[:, 1] - ref).max() <= 1e-13 * np.abs(ref).max() * np.log2(l)
    # a real series: zero-frequency and Nyquist bins are real exactly, as numpy returns them
    # (the reference divides the sine amplitude of bin 0 by 1 and expects 0, core.py:2690-2700)
    assert out[0, 1] == 0.0
    if l % 2 == 0:
        assert out[l // 2, 1] == 0.0


def test_rfft_linearity_and_impulse():
    from godot_amd import ops
    l = 1 << 18
    x = torch.zeros(l, dtype=torch.float64, device=DEV)
    x[1] = 1.0
    out = _n(ops.rfft(x, l))
    k = np.arange(l // 2 + 1)
    assert np.abs(out[:, 0] - np.cos(2 * np.pi * k / l)).max() < 1e-14
    assert np.abs(out[:, 1] + np.sin(2 * np.pi * k / l)).max() < 1e-14


def _rel(a, b, floor):
    return np.abs(a - b) <= 1e-9 * np.maximum(np.abs(b), floor)


def test_power_spectrum_fft_matches_reference(golden):
    from godot_amd import core
    g = golden('timeseries_fft.npz')
    ts = _ts(g)
    f, p0, p1, pb = core.power_spectrum_fft(ts)
    assert np.array_equal(f, g['ps_freqs'])
    for got, key in ((p0, 'ps_p0'), (p1, 'ps_p1'), (pb, 'ps_pb')):
        ref = g[key]
        ok = _rel(got, ref, floor=np.nanmax(np.abs(ref)) * 1e-6) | (np.isnan(got) & np.isnan(ref))
        assert ok.all(), (key, np.argwhere(~ok)[:5].ravel(), got[~ok][:5], ref[~ok][:5])
    amps = core.power_spectrum_fft(ts, get_amps=True)
    for i in range(8):
        ref = g['ps_amps'][i]
        got = amps[1 + i]
        ok = _rel(got, ref, floor=np.nanmax(np.abs(ref[np.isfinite(ref)])) * 1e-6) \
            | (~np.isfinite(ref) & ~np.isfinite(got))
        assert ok.all(), (i, np.argwhere(~ok)[:5].ravel(), got[~ok][:5], ref[~ok][:5])
    tw = core.power_spectrum_fft(ts, tweak_exp=True)
    for i in range(3):
        ref = g['ps_tweak'][i]
        assert _rel(tw[1 + i], ref, floor=np.abs(ref).max() * 1e-6).all(), i
    omr, omp = core.get_orbital_modulation(ts, g['om_freqs'])
    assert np.abs(omr - g['om_rvals']).max() <= 1e-9
    assert _rel(omp, g['om_pows'], floor=0).all()
    pc = core.power_spectrum_fft(ts, exposure_correction=omr)
    for i in range(3):
        ref = g['ps_corr'][i]
        assert _rel(pc[1 + i], ref, floor=np.abs(ref).max() * 1e-6).all(), i
    nz = core.power_spectrum_fft(ts, no_zero_pad=True)
    assert np.array_equal(nz[0], g['ps_nzp_freqs'])
    for i in range(3):
        ref = g['ps_nzp'][i]
        assert _rel(nz[1 + i], ref, floor=np.abs(ref).max() * 1e-6).all(), ('no_zero_pad', i)
    eo = core.power_spectrum_fft(ts, exp_only=True)
    assert _rel(eo[1], g['ps_exp_only'], floor=np.abs(g['ps_exp_only']).max() * 1e-6).all()


def test_power_spectrum_simple_matches_reference(golden):
    from godot_amd import core
    g = golden('timeseries_fft.npz')
    n = int(g['pss2_n'])
    ts = _ts({k: (g[k][:n] if k in ('starts', 'stops', 'exp', 'sexp', 'bexp', 'counts', 'weights',
                                    'weights2', 'deadtime') else g[k]) for k in g})
    f, dl = core.power_spectrum_simple(ts)
    assert np.array_equal(f, g['pss2_freqs'])
    ref = g['pss2_dlogl']
    assert _rel(dl, ref, floor=np.abs(ref).max() * 1e-6).all()
    # any length (chirp-z path)
    f, dl = core.power_spectrum_simple(_ts(g))
    assert np.array_equal(f, g['pss_freqs'])
    ref = g['pss_dlogl']
    assert _rel(dl, ref, floor=np.abs(ref).max() * 1e-6).all()


def test_trig_sums_match_reference(golden):
    from godot_amd import core, ops
    g = golden('trig_sums.npz')
    d = torch.device(DEV)
    off = np.append(0, np.cumsum(g['counts'])).astype(np.int64)
    offd = torch.from_numpy(off).to(d)
    batch = core.CellBatch(torch.from_numpy(g['we']).to(d), torch.from_numpy(g['we']).to(d), offd[:-1],
                           offd[1:], torch.from_numpy(g['starts']).to(d),
                           torch.from_numpy(g['stops']).to(d), torch.from_numpy(g['exp']).to(d),
                           torch.from_numpy(g['exp'] * 5).to(d))
    old = core.CellsLogLikelihoodOld(batch)
    assert old.nph == len(g['we'])
    got = core.power_spectrum_dft(old, g['freqs'])
    assert _rel(got, g['dft'], floor=np.abs(g['dft']).max() * 1e-6).all()
    got = core.power_spectrum_dft(old, g['freqs'], unweighted=True)
    assert _rel(got, g['dft_unweighted'], floor=np.abs(g['dft_unweighted']).max() * 1e-6).all()
    fh = core.fit_harmonics(old, g['freqs'][:16])
    ref = g['harmonics']
    assert _rel(fh, ref, floor=np.abs(ref).max(axis=0, keepdims=True) * 1e-6).all()
    for unw, key in ((False, 'ls'), (True, 'ls_unweighted')):
        got = core.compute_ls(old, g['freqs'], unweighted=unw)
        assert _rel(got, g[key], floor=np.abs(g[key]).max() * 1e-6).all(), key
    c, s, t = core.fit_harmonics(old, float(g['freqs'][3]))
    assert _rel(np.array([c, s, t]), ref[3], floor=np.abs(ref).max(axis=0) * 1e-6).all()


@pytest.mark.parametrize('m', [1, 2, 3, 4, 7, 10, 20])
def test_z2m_matches_oracle(m):
    from godot_amd import ops
    rng = np.random.default_rng(m)
    nph = 5000
    t = np.sort(rng.uniform(0, 3.2e8, nph))
    w = np.clip(rng.beta(0.3, 1.5, nph), 1e-6, 1 - 1e-6)
    f0, df, nf, j0 = 2.5e-3, 6.1e-10, 300, 123456
    z, h = ops.z2m(torch.from_numpy(t).to(DEV), torch.from_numpy(w).to(DEV), f0, df, nf, m=m, j0=j0,
                   want_h=True)
    ref = orc.z2m(t, w, f0, df, nf, m=m, j0=j0)
    assert np.abs(_n(z) - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())
    href = orc.h_test(t, w, f0, df, nf, mmax=m, j0=j0)
    assert np.abs(_n(h) - href).max() <= 1e-12 * max(1.0, np.abs(href).max())


@pytest.mark.parametrize('nf', [300, 20 * 256 * 3 + 7, 20 * 256 * 512])
def test_z2m_negative_weight_takes_general_kernel(nf):
    """m = 2 carries sqrt(w) in the phasor; a table with a negative weight must
    give the same sums through the general kernel (single launch and photon split)."""
    from godot_amd import ops
    rng = np.random.default_rng(21)
    nph = 4100
    t = np.sort(rng.uniform(0, 3.2e8, nph))
    w = rng.uniform(0.0, 1.0, nph)
    f0, df = 1.5e-3, 7.3e-10
    zpos, _ = ops.z2m(torch.from_numpy(t).to(DEV), torch.from_numpy(w).to(DEV), f0, df, nf, m=2)
    ref = orc.z2m(t, w, f0, df, min(nf, 400), m=2)
    assert np.abs(_n(zpos)[:len(ref)] - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())
    w[17] = -0.25
    w[3000] = -0.5
    zneg, hneg = ops.z2m(torch.from_numpy(t).to(DEV), torch.from_numpy(w).to(DEV), f0, df, nf, m=2, want_h=True)
    ref = orc.z2m(t, w, f0, df, min(nf, 400), m=2)
    assert np.abs(_n(zneg)[:len(ref)] - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())
    assert np.isfinite(_n(zneg)).all() and np.isfinite(_n(hneg)).all()


@pytest.mark.parametrize('m', [1, 2, 5])
def test_z2m_sums_of_photon_shards_add(m):
    """Trig sums of two photon shards add up to the whole list's statistic, and
    the sums themselves match the oracle's."""
    from godot_amd import ops
    rng = np.random.default_rng(30 + m)
    nph = 6001
    t = np.sort(rng.uniform(0, 3.2e8, nph))
    w = np.clip(rng.beta(0.3, 1.5, nph), 1e-6, 1 - 1e-6)
    f0, df, nf, j0 = 2.5e-3, 6.1e-10, 333, 4567
    td, wd = torch.from_numpy(t).to(DEV), torch.from_numpy(w).to(DEV)
    cut = 2500
    s1, q1 = ops.z2m_sums(td[:cut].contiguous(), wd[:cut].contiguous(), f0, df, nf, m=m, j0=j0)
    s2, q2 = ops.z2m_sums(td[cut:].contiguous(), wd[cut:].contiguous(), f0, df, nf, m=m, j0=j0)
    C, Sn = orc.z2m_trig_sums(t[:cut], w[:cut], f0, df, j0 + np.arange(nf), m)
    scale = np.sqrt(np.sum(w[:cut] ** 2))
    assert np.abs(_n(s1)[:, :, 0] - C.T).max() <= 1e-12 * scale
    assert np.abs(_n(s1)[:, :, 1] - Sn.T).max() <= 1e-12 * scale
    assert abs(float(q1) - np.sum(w[:cut] ** 2)) <= 1e-12 * np.sum(w ** 2)
    z, h = ops.z2m_finish(s1 + s2, q1 + q2, want_h=True)
    zw, hw = ops.z2m(td, wd, f0, df, nf, m=m, j0=j0, want_h=True)
    assert np.abs(_n(z) - _n(zw)).max() <= 1e-12 * max(1.0, _n(zw).max())
    assert np.abs(_n(h) - _n(hw)).max() <= 1e-12 * max(1.0, np.abs(_n(hw)).max())
    ref = orc.z2m(t, w, f0, df, nf, m=m, j0=j0)
    assert np.abs(_n(z) - ref).max() <= 1e-12 * max(1.0, ref.max())


def test_z2m_signal_and_exactness_at_large_phase():
    """A pulsed signal is recovered at its frequency after 1e10 cycles; the
    single-launch and photon-split paths agree."""
    from godot_amd import ops
    rng = np.random.default_rng(9)
    nph = 20000
    fsig = 31.41592653589793
    t = np.sort(rng.uniform(0, 3.0e8, nph))
    # pulsed: keep photons near phase 0
    ph = (t * fsig) % 1.0
    keep = rng.random(nph) < 0.5 * (1 + np.cos(2 * np.pi * ph))
    t = t[keep]
    w = np.full(len(t), 0.7)
    df = 1e-10
    f0 = fsig - 50 * df
    z, _ = ops.z2m(torch.from_numpy(t).to(DEV), torch.from_numpy(w).to(DEV), f0, df, 101, m=2)
    z = _n(z)
    assert np.argmax(z) in (49, 50, 51) and z.max() > 1000
    ze = orc.z2m_exact(t[:300], w[:300], f0, df, 50, 2)
    z3, _ = ops.z2m(torch.from_numpy(t[:300].copy()).to(DEV), torch.from_numpy(w[:300].copy()).to(DEV),
                    f0, df, 51, m=2)
    assert abs(_n(z3)[50] - ze) <= 1e-9 * max(1, ze)
    # many frequencies -> one launch over all photons; must equal the split path
    nf = 256 * 20 * 1024     # whole residency rounds: not split
    zbig, _ = ops.z2m(torch.from_numpy(t).to(DEV), torch.from_numpy(w).to(DEV), f0, df, nf, m=2)
    assert np.abs(_n(zbig[:101]) - z).max() <= 1e-9 * z.max()


def test_spectrum_from_separately_transformed_series():
    """power_spectrum_fft put together from five stand-alone transforms (what ranks that
    each own some of the transforms exchange) equals the fused call and the oracle."""
    from godot_amd import ops, dist as gd
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(31)
    nbin, l = 5000, 1 << 15
    cnt = rng.poisson(3.0, nbin).astype(float)
    W = cnt * rng.uniform(0.1, 0.3, nbin)
    WW = W * rng.uniform(0.2, 0.6, nbin)
    S = np.full(nbin, W.mean() * 1.05)
    B = np.full(nbin, (cnt - W).mean() * 0.95)
    Wb, WbWb = cnt - W, cnt - 2 * W + WW
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    series = [d(W - S), d(Wb - B), d(WW), d(WbWb), d(W - WW)]
    p0, p1, pb, amps = gd.power_spectrum_fft_sharded(
        lambda i: series[i], l, ops.rfft, lambda *parts: ops.power_spectrum_from_parts(*parts, want_amps=True))
    q0, q1, qb, qamps = ops.power_spectrum_fft(d(W), d(WW), d(S), d(Wb), d(WbWb), d(B), l, want_amps=True)
    nk = l // 4 + 1
    f = [np.fft.rfft(x, n=l) for x in (W - S, Wb - B, WW, WbWb, W - WW)]
    r0, r1, rb = orc.power_spectrum_from_parts(f[0][:nk], f[1][:nk], f[2][::2], f[3][::2], f[4][::2])
    for got, fused, ref in ((p0, q0, r0), (p1, q1, r1), (pb, qb, rb)):
        got, fused = got.cpu().numpy(), fused.cpu().numpy()
        sc = np.abs(ref).max()
        assert np.abs(got - ref).max() <= 1e-9 * sc and np.abs(got - fused).max() <= 1e-9 * sc
    assert np.allclose(amps.cpu().numpy()[:, 1:], qamps.cpu().numpy()[:, 1:], rtol=1e-7, atol=1e-12)


@pytest.mark.parametrize('lg,frac', [(17, 0.9), (18, 0.113), (19, 1.0), (20, 0.5), (21, 0.113), (23, 0.05)])
def test_spectrum_whose_last_pass_forms_the_estimators(lg, frac):
    """Padded lengths from 2^17 on take the kernel that runs the last pass of all five
    transforms, splits them into the real transforms' bins and evaluates the estimators
    without writing a transform out: every bin (the first, the last at l/4, those of
    columns 0 and S/2 that pair with themselves) against numpy's transforms, with the
    amplitudes, for series that fill the padded length and series that leave most of it
    empty."""
    from godot_amd import ops
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(100 + lg)
    l = 1 << lg
    nbin = max(8, int(l * frac))
    cnt = rng.poisson(3.0, nbin).astype(float)
    W = cnt * rng.uniform(0.1, 0.3, nbin)
    WW = W * rng.uniform(0.2, 0.6, nbin)
    S = rng.uniform(0.5, 1.5, nbin) * W.mean()
    B = rng.uniform(0.5, 1.5, nbin) * (cnt - W).mean()
    Wb, WbWb = cnt - W, cnt - 2 * W + WW
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    q0, q1, qb, qamps = ops.power_spectrum_fft(d(W), d(WW), d(S), d(Wb), d(WbWb), d(B), l, want_amps=True)
    nk = l // 4 + 1
    f = [np.fft.rfft(x, n=l) for x in (W - S, Wb - B, WW, WbWb, W - WW)]
    parts = (f[0][:nk], f[1][:nk], f[2][::2], f[3][::2], f[4][::2])
    with np.errstate(all='ignore'):
        r0, r1, rb = orc.power_spectrum_from_parts(*parts)
        ramps = orc.power_spectrum_from_parts(*parts, get_amps=True)
    assert q0.numel() == nk
    for got, ref in ((q0, r0), (q1, r1), (qb, rb)):
        got = got.cpu().numpy()
        sc = np.abs(ref[1:]).max()
        assert np.abs(got[1:] - ref[1:]).max() <= 1e-9 * sc
        assert np.isnan(got[0]) == np.isnan(ref[0]) and (np.isnan(ref[0]) or abs(got[0] - ref[0]) <= 1e-9 * sc)
    got = qamps.cpu().numpy()
    ramps = np.asarray(ramps)
    sc = np.abs(ramps[:, 1:]).max(axis=1, keepdims=True)
    assert (np.abs(got[:, 1:] - ramps[:, 1:]) <= 1e-8 * sc).all()


@pytest.mark.parametrize('lg,nbin', [(23, 1_051_920), (23, 1 << 22), (23, 900), (23, 524_289),
                                     (24, 1_893_456), (24, 700), (24, 1 << 23), (24, 1_048_577),
                                     (25, 3_786_912), (25, 1 << 24), (25, 5000)])
def test_long_series_transformed_by_columns_and_rows(lg, nbin):
    """Padded lengths 2^23, 2^24 and 2^25 take the path of two trips through memory (columns of the
    padded series, then rows with the estimators): every bin and the amplitudes against numpy's
    transforms, and against the three-trip path of the same library, for series that fill a
    tenth of the padded length (the default padding), half of it (the most the path takes: every
    sixteenth of a column holds samples), a short series (not a shorter one: the lowest bins of
    a few samples in 2^24 are a quotient of two rounding errors in numpy as well), and one
    sample past a sixteenth."""
    import os
    from godot_amd import ops
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(500 + lg + nbin % 97)
    l = 1 << lg
    cnt = rng.poisson(3.0, nbin).astype(float)
    W = cnt * rng.uniform(0.1, 0.3, nbin)
    WW = W * rng.uniform(0.2, 0.6, nbin)
    S = rng.uniform(0.5, 1.5, nbin) * max(W.mean(), 0.1)
    B = rng.uniform(0.5, 1.5, nbin) * max((cnt - W).mean(), 0.1)
    Wb, WbWb = cnt - W, cnt - 2 * W + WW
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    args = (d(W), d(WW), d(S), d(Wb), d(WbWb), d(B), l)
    q0, q1, qb, qamps = ops.power_spectrum_fft(*args, want_amps=True)
    os.environ['GODOT_FFT_THREE_TRIPS'] = '1'
    try:
        o0, o1, ob, oamps = ops.power_spectrum_fft(*args, want_amps=True)
    finally:
        del os.environ['GODOT_FFT_THREE_TRIPS']
    nk = l // 4 + 1
    f = [np.fft.rfft(x, n=l) for x in (W - S, Wb - B, WW, WbWb, W - WW)]
    parts = (f[0][:nk], f[1][:nk], f[2][::2], f[3][::2], f[4][::2])
    del f
    with np.errstate(all='ignore'):
        r0, r1, rb = orc.power_spectrum_from_parts(*parts)
        ramps = np.asarray(orc.power_spectrum_from_parts(*parts, get_amps=True))
    assert q0.numel() == nk
    for got, old, ref in ((q0, o0, r0), (q1, o1, r1), (qb, ob, rb)):
        got, old = got.cpu().numpy(), old.cpu().numpy()
        sc = np.abs(ref[1:]).max()
        assert np.abs(got[1:] - ref[1:]).max() <= 1e-9 * sc
        assert np.abs(got[1:] - old[1:]).max() <= 1e-9 * sc
        assert np.isnan(got[0]) == np.isnan(ref[0]) and (np.isnan(ref[0]) or abs(got[0] - ref[0]) <= 1e-9 * sc)
    got = qamps.cpu().numpy()
    sc = np.abs(ramps[:, 1:]).max(axis=1, keepdims=True)
    assert (np.abs(got[:, 1:] - ramps[:, 1:]) <= 1e-8 * sc).all()


@pytest.mark.parametrize('lg,nbin,nparts', [(23, 1_051_920, 8), (24, 1_893_456, 4), (25, 3_786_912, 2), (23, 700, 2)])
def test_transform_in_parts_equals_the_whole(lg, nbin, nparts):
    """The columns-then-rows transform done part by part on one GPU -- the columns of every part
    (godot_power_spectrum_columns), the rows of every range (godot_power_spectrum_rows) after the
    pieces have been put side by side as an all-to-all between GPUs would put them -- gives the
    bits of godot_power_spectrum_fft for P_0, P_1, P_b up to the zero-frequency sums (formed from
    the series instead of from row 0: 1e-12), every bin written by exactly one range."""
    from godot_amd import ops, dist as gd
    dev = torch.device('cuda:0')
    l = 1 << lg
    g = torch.Generator(device=dev)
    g.manual_seed(lg * 1000 + nparts)
    cnt = torch.poisson(torch.full((nbin,), 2.6, dtype=torch.float64, device=dev), generator=g)
    W = cnt * torch.rand(nbin, dtype=torch.float64, device=dev, generator=g) * 0.3
    WW = W * 0.4
    S = torch.full((nbin,), float(W.mean()) * 1.1, dtype=torch.float64, device=dev)
    B = S * 5
    Wb, WbWb = cnt - W, cnt - 2 * W + WW
    assert ops.power_spectrum_parts_supported(nbin, l, nparts)
    assert not ops.power_spectrum_parts_supported(nbin, l, 3) and not ops.power_spectrum_parts_supported(nbin, 1 << 20, 2)
    whole = ops.power_spectrum_fft(W, WW, S, Wb, WbWb, B, l)[:3]
    parts = [ops.power_spectrum_columns(W, WW, S, Wb, WbWb, B, l, p, nparts)[1] for p in range(nparts)]
    dc = torch.stack([WW.sum(), WbWb.sum(), (W - WW).sum()])
    edges = gd.four_step_rows(nparts)
    assert edges[0] == 0 and edges[-1] == 8193 and all(e % 16 == 0 for e in edges[:-1])
    total = [torch.zeros(l // 4 + 1, dtype=torch.float64, device=dev) for _ in range(3)]
    written = torch.zeros(l // 4 + 1, dtype=torch.int32, device=dev)
    for r in range(nparts):
        m0, m1 = edges[r], edges[r + 1]
        rows = [torch.cat([parts[p][i][m0:m1] for p in range(nparts)], dim=1) for i in range(5)]
        v_rows = torch.cat([x.reshape(-1) for x in rows])
        got = ops.power_spectrum_rows(v_rows, dc, l, m0, m1)[:3]
        for t, x in zip(total, got):
            t += x
        written += (got[1] != 0).to(torch.int32)
    assert int(written.max()) == 1
    for t, w in zip(total, whole):
        fin = torch.isfinite(w)
        assert torch.equal(torch.isfinite(t), fin)
        scale = w[fin].abs().max()
        # (a series of a few hundred bins: the estimators near zero frequency are differences of
        # nearly equal sums, and the last bits of the zero-frequency values show)
        assert float((t[fin] - w[fin]).abs().max()) <= (1e-12 if nbin > 10_000 else 1e-7) * float(scale)
    with pytest.raises(Exception):
        ops.power_spectrum_rows(v_rows[:100], dc, l, edges[-2], edges[-1])


@pytest.mark.gpu
@pytest.mark.parametrize('l,tsamp', [(1 << 14, 300.0), (1 << 23, 299.99999999), (1 << 25, 100.0), (48211, 86400.0 / 7)])
def test_frequency_axis_is_numpys_to_the_bit(l, tsamp):
    """np.fft.rfftfreq(l)[:l // 4 + 1] / tsamp (power_spectrum_fft, reference core.py:2677-2679) and
    * (1 / tsamp) (power_spectrum_simple, core.py:2765-2766), formed on the device."""
    from godot_amd import ops
    dev = torch.device('cuda:0')
    nk = l // 4 + 1
    ref = np.fft.rfftfreq(l)[:nk]
    got = ops.spectrum_frequencies(nk, 1.0 / l, tsamp, True, dev).cpu().numpy()
    assert np.array_equal(got, ref / tsamp)
    got = ops.spectrum_frequencies(nk, 1.0 / l, 1. / tsamp, False, dev).cpu().numpy()
    assert np.array_equal(got, ref * (1. / tsamp))
    assert ops.spectrum_frequencies(0, 1.0 / l, tsamp, True, dev).numel() == 0


@pytest.mark.gpu
def test_row_peaks_are_numpys_max_and_argmax():
    """godot_row_peaks: the first maximum of every row from a first column on, a NaN before any number."""
    from godot_amd import ops
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(5)
    for nrow, ncol, first in [(1, 1, 0), (3, 2, 1), (7, 1000, 1), (96, 70001, 1), (5, 2_097_153, 1), (2, 300000, 17)]:
        a = rng.standard_normal((nrow, ncol))
        if ncol > 100:
            a[0, [5, 90]] = a[0].max() + 1.0                  # equal maxima: the first counts
            a[nrow - 1, ncol - 1] = 1e9                       # the last entry of the last row
            if nrow > 2:
                a[1, [40, 7]] = np.nan                        # a NaN before any number, the first of them
                a[2, 0] = np.inf                              # before the first column: not looked at
                a[2, first:] = -np.inf                        # nothing but -inf: the first entry
        v, i = ops.row_peaks(torch.from_numpy(a).to(dev), first=first)
        ref_i = np.argmax(a[:, first:], axis=1)
        ref_v = a[np.arange(nrow), ref_i + first]
        assert np.array_equal(i.cpu().numpy(), ref_i)
        assert np.array_equal(v.cpu().numpy(), ref_v, equal_nan=True)
    with pytest.raises(Exception):
        ops.row_peaks(torch.zeros((2, 3), dtype=torch.float64, device=dev), first=3)
