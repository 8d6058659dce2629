"""GPU parity of the Bayesian-Blocks path (fitness rows, dynamic programme,
block light curve) against the reference's golden vectors and the oracle."""
import numpy as np
import pytest
import torch

from oracle import godot_oracle as orc

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')


def _d(x, dt=torch.float64):
    return torch.from_numpy(np.ascontiguousarray(x)).to(device=DEV, dtype=dt)


def _batch(g):
    from godot_amd import core
    off = np.append(0, np.cumsum(g['counts'])).astype(np.int64)
    offd = _d(off, torch.int64)
    sexp = g['exp']
    bexp = g['exp'] / g['SonB']
    return core.CellBatch(_d(g['we']), _d(g['ti']), offd[:-1].contiguous(), offd[1:].contiguous(), _d(g['starts']),
                          _d(g['stops']), _d(sexp), _d(bexp))


def test_fitness_rows_equal_reference_bit_for_bit(golden):
    """On the reference's own profiles the interpolated sums are the reference's
    to the last bit (same slopes, same order of additions)."""
    from godot_amd import ops
    g = golden('bayesian_blocks.npz')
    fv, fa = ops.bb_fitness(_d(g['dom']), _d(g['cod']), ngrid=1000)
    assert np.array_equal(fv.cpu().numpy(), g['fit_val'])
    assert np.array_equal(fa.cpu().numpy(), g['fit_arg'])


@pytest.mark.parametrize('prior,key', [(8, 'bb8'), (2, 'bb2')])
def test_partition_equals_reference(golden, prior, key):
    from godot_amd import ops
    g = golden('bayesian_blocks.npz')
    n = len(g['dom'])
    best, last = ops.bb_partition(_d(g['fit_val']), n, prior)
    idx, best_ref, best_all, last_ref = orc.bb_partition(
        [g['fit_val'][i * (i + 1) // 2:(i + 1) * (i + 2) // 2] for i in range(n)], prior)
    assert np.array_equal(last.cpu().numpy(), last_ref)
    assert np.array_equal(best.cpu().numpy(), best_all)
    assert np.array_equal(idx, g[key + '_indices'])


@pytest.mark.parametrize('variant', ['ties', 'smooth', 'nan'])
def test_partition_over_several_blocks_of_rows(variant):
    """More rows than one block of the dynamic programme holds (the candidates below a
    block are reduced by a kernel of their own): best and last against the oracle, with
    many exact ties, -inf entries and a NaN."""
    from godot_amd import ops
    rng = np.random.default_rng(len(variant))
    n = 2500
    tri = n * (n + 1) // 2
    if variant == 'ties':
        fv = np.round(rng.normal(0, 3, tri), 0)          # integers: ties everywhere
        fv[rng.random(tri) < 0.05] = -np.inf
    else:
        fv = rng.normal(0, 3, tri) + 0.01 * np.arange(tri) / n
    if variant == 'nan':
        fv[1800 * 1801 // 2 + 700] = np.nan
    rows = [fv[i * (i + 1) // 2:(i + 1) * (i + 2) // 2] for i in range(n)]
    with np.errstate(all='ignore'):
        idx, best_ref, best_all, last_ref = orc.bb_partition(rows, 2.5)
    best, last = ops.bb_partition(_d(fv), n, 2.5)
    assert np.array_equal(last.cpu().numpy(), last_ref)
    assert np.array_equal(best.cpu().numpy(), best_all, equal_nan=True)


def test_bb_pipeline_matches_reference(golden):
    """Profiles, fitness rows, partition, merged cells and their fluxes all on the
    device, against the reference's get_bb_lightcurve."""
    from godot_amd import core
    g = golden('bayesian_blocks.npz')
    cll = core.CellsLogLikelihood(_batch(g))
    rows, idx = cll.get_bb_lightcurve(tsmin=4, bb_prior=8, get_indices=True)
    assert np.array_equal(idx, g['bb_idx'])
    ref = g['bb_rows']
    assert np.array_equal(rows[:, 4] == -1, ref[:, 4] == -1)
    assert np.abs(rows[:, 0] - ref[:, 0]).max() <= 1e-12 * np.abs(ref[:, 0]).max()
    tol = 1e-9 * np.maximum(np.abs(ref[:, 1:]), 1.0)
    tol[:, 4] *= np.maximum(1.0, np.abs(ref[:, 5]))      # TS: differences of sums of logarithms
    assert (np.abs(rows[:, 1:] - ref[:, 1:]) <= tol).all(), np.abs(rows[:, 1:] - ref[:, 1:]).max(0)
    # do_bb with another prior reuses the rows
    i2, tot2, vts2, vdof2, fr = cll.do_bb(prior=2)
    assert np.array_equal(i2, g['bb2_indices'])
    assert abs(tot2 - g['bb2_scalars'][0]) <= 1e-9 * abs(g['bb2_scalars'][0])
    assert vdof2 == g['bb2_scalars'][2] and (vts2 == g['bb2_scalars'][1] or np.isclose(vts2, g['bb2_scalars'][1]))
    f = cll.fitness(5, 40)
    r = g['fitness_5_40']
    fin = np.isfinite(r)
    assert np.array_equal(np.isfinite(f), fin)
    assert np.abs(f[fin] - r[fin]).max() <= 1e-9 * np.maximum(np.abs(r[fin]), 1).max()
    top = cll.do_top_hat()
    n = len(g['dom'])
    for k in (0, 1):
        for j, o in enumerate((0, 1, 7)):
            a, b = np.diagonal(top[k], offset=o)[: n - 7], g['top_hat_diag'][k, j]
            fin = np.isfinite(b)
            assert np.array_equal(np.isnan(a), np.isnan(b))
            assert np.abs(a[fin] - b[fin]).max() <= 1e-9 * max(1.0, np.abs(b[fin]).max()) if fin.any() else True


def test_fitness_matches_oracle_on_random_profiles():
    """Ragged supports (many -inf sums), NaN rows, non-uniform abscissae, a grid that is not
    1000 points."""
    from godot_amd import ops
    rng = np.random.default_rng(5)
    n, npt = 300, 64
    lo = rng.uniform(0, 2, n)
    hi = lo + rng.uniform(0.05, 3, n)
    dom = np.linspace(lo, hi, npt).T.copy()
    # every fifth cell on abscissae that are not a linspace: the kernel must then load them
    # (it computes the abscissae of cells whose rows equal numpy.linspace to the bit)
    warped = np.arange(n) % 5 == 3
    u = np.linspace(0, 1, npt) ** 1.3
    dom[warped] = lo[warped, None] + (hi - lo)[warped, None] * u[None, :]
    peak = rng.uniform(lo, hi)
    cod = -((dom - peak[:, None]) * rng.uniform(1, 6, n)[:, None]) ** 2
    cod -= cod.max(axis=1, keepdims=True)
    dom[17] = np.nan
    cod[17] = np.nan
    for ngrid in (1000, 333):
        old = orc.BB_NGRID
        orc.BB_NGRID = ngrid
        try:
            with np.errstate(all='ignore'):
                ref = [orc.bb_fitness(dom, cod, 0, i + 1) for i in range(n)]
        finally:
            orc.BB_NGRID = old
        fv, fa = ops.bb_fitness(_d(dom), _d(cod), ngrid=ngrid)
        fv, fa = fv.cpu().numpy(), fa.cpu().numpy()
        rv = np.concatenate([r[0] for r in ref])
        ra = np.concatenate([r[1] for r in ref])
        # rows that touch the NaN cell hold whatever numpy's search made of NaN abscissae
        clean = np.concatenate([np.arange(i + 1) > (17 if i >= 17 else -1) for i in range(n)])
        assert np.array_equal(fv[clean], rv[clean])
        assert np.array_equal(fa[clean], ra[clean])
        assert np.isnan(fv[~clean]).all()


@pytest.mark.parametrize('poison', [None, 'plus_inf'])
def test_fitness_with_disjoint_supports(poison):
    """A flaring source: runs of cells whose supports do not overlap.  Grid points that left
    some cell's support stay at -inf and whole rows end early; a +inf among the values makes
    -inf + inf = NaN in numpy and must switch both shortcuts off for what it touches."""
    from godot_amd import ops
    rng = np.random.default_rng(3)
    n, npt = 180, 50
    state = (np.arange(n) // 37) % 2                    # quiet, flare, quiet, ...
    lo = np.where(state == 0, rng.uniform(0.0, 0.3, n), rng.uniform(4.0, 4.5, n))
    hi = lo + np.where(state == 0, rng.uniform(0.8, 1.5, n), rng.uniform(2.0, 3.0, n))
    dom = np.linspace(lo, hi, npt).T.copy()
    peak = rng.uniform(lo, hi)
    cod = -((dom - peak[:, None]) * rng.uniform(1, 6, n)[:, None]) ** 2
    cod -= cod.max(axis=1, keepdims=True)
    if poison == 'plus_inf':
        cod[60, 10] = np.inf
    with np.errstate(all='ignore'):
        ref = [orc.bb_fitness(dom, cod, 0, i + 1) for i in range(n)]
    fv, fa = ops.bb_fitness(_d(dom), _d(cod))
    fv, fa = fv.cpu().numpy(), fa.cpu().numpy()
    rv = np.concatenate([r[0] for r in ref])
    ra = np.concatenate([r[1] for r in ref])
    assert np.isneginf(rv).mean() > 0.3                 # the shortcuts have something to do
    assert np.array_equal(fv, rv, equal_nan=True)
    assert np.array_equal(fa, ra, equal_nan=True)


def test_fitness_row_ranges_tile_the_triangle(golden):
    """Rows computed in ranges (the multi-GPU split) concatenate to the whole."""
    from godot_amd import ops, dist as gd
    g = golden('bayesian_blocks.npz')
    dom, cod = _d(g['dom']), _d(g['cod'])
    n = dom.shape[0]
    parts = [ops.bb_fitness(dom, cod, rows=gd.bb_row_shard(n, r, 3)) for r in range(3)]
    fv = torch.cat([p[0] for p in parts]).cpu().numpy()
    fa = torch.cat([p[1] for p in parts]).cpu().numpy()
    assert np.array_equal(fv, g['fit_val']) and np.array_equal(fa, g['fit_arg'])
    e = ops.bb_fitness(dom, cod, rows=(7, 7))
    assert e[0].numel() == 0

