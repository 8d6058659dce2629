import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name), allow_pickle=False) as f:
        return {k: f[k] for k in f.files}


@pytest.fixture(scope='session')
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]
    return get


def assert_same(got, want, what=''):
    """Bit-for-bit equality (NaN == NaN) with a short failure message."""
    got = np.asarray(got)
    want = np.asarray(want)
    assert got.shape == want.shape, '%s: shape %s vs %s' % (what, got.shape, want.shape)
    bad = ~((got == want) | ((got != got) & (want != want)))
    if bad.any():
        i = np.argwhere(bad)[:4]
        raise AssertionError('%s: %d of %d differ, first at %s: got %r want %r' % (
            what, bad.sum(), bad.size, i.tolist(), got[bad][:4].tolist(), want[bad][:4].tolist()))
