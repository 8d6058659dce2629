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


def synth_sources_device(nsrc, device, nph=1_000_000, years=10., seed=1000, indices=None):
    """Many sources over ONE spacecraft file, made on the GPU (a thousand host-made sources
    take minutes): TSTART, TSTOP, LIVETIME as synth() makes them, shared by all sources; every
    source has its own exposure column (that of synth() times 1 + k / 1000: its own sky
    position), photons drawn in proportion to exposure (uniform inside their row) and weights
    clip(u^3.3, 1e-6, 1 - 1e-6), u uniform: mostly background with a tail of source photons,
    like the Beta(0.3, 1.5) of synth().  Returns a list of dicts of device tensors with the
    keys of synth() (without the totals).  indices: make only these of the nsrc sources (a
    rank's share of a job dealt over several GPUs); source k is the same whoever makes it."""
    import torch
    base = synth(1, years=years, seed=seed)
    TSTART, TSTOP, LIVETIME, exposure = [torch.from_numpy(base[k]).to(device)
                                         for k in ('TSTART', 'TSTOP', 'LIVETIME', 'exposure')]
    cdf = torch.cumsum(exposure, 0)
    cdf = cdf / cdf[-1]
    g = torch.Generator(device=device)
    out = []
    for k in (range(nsrc) if indices is None else indices):
        g.manual_seed(seed + k)
        u = lambda: torch.rand(nph, dtype=torch.float64, device=device, generator=g)
        idx = torch.searchsorted(cdf, u()).clamp_(max=cdf.numel() - 1)
        ti = torch.sort(TSTART[idx] + 30 * u())[0]
        we = u().pow_(3.3).clamp_(1e-6, 1 - 1e-6)
        out.append(dict(ti=ti, we=we, TSTART=TSTART, TSTOP=TSTOP, LIVETIME=LIVETIME,
                        exposure=exposure * (1.0 + 0.001 * k)))
    return out
