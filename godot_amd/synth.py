"""Deterministic synthetic photon lists and exposure tables (host, numpy).

Exposure table: 30-s rows over the span, rows with (row % 190) >= 150 removed
(a 95-minute "orbit" with a 20-minute gap), LIVETIME 27 s, exposure
27 * (3000 + 2000 sin^2(2 pi (TSTART - T0) / 5700)) cm^2 s.  Photons follow the
exposure; weights are clip(Beta(0.3, 1.5), 1e-6, 1 - 1e-6).
"""
import numpy as np

T0 = 239557417.0


def synth(nph, years=10., seed=7):
    rng = np.random.default_rng(seed)
    nrow = int(years * 365.25 * 86400 / 30)
    rows = np.arange(nrow)
    rows = rows[(rows % 190) < 150]
    TSTART = T0 + 30. * rows
    TSTOP = TSTART + 30.
    LIVETIME = np.full(len(rows), 27.)
    exposure = 27. * (3000 + 2000 * np.sin(2 * np.pi * (TSTART - T0) / 5700) ** 2)
    E = np.sum(exposure)
    idx = np.searchsorted(np.cumsum(exposure) / E, rng.random(nph))
    idx = np.minimum(idx, len(rows) - 1)
    ti = np.sort(TSTART[idx] + 30 * rng.random(nph))
    we = np.clip(rng.beta(0.3, 1.5, nph), 1e-6, 1 - 1e-6)
    S = np.sum(we)
    B = len(we) - S
    return dict(ti=ti, we=we, TSTART=TSTART, TSTOP=TSTOP, LIVETIME=LIVETIME,
                exposure=exposure, S=S, B=B, E=E)
