#!/usr/bin/env python3
"""Generate tests/golden/cell_variants.npz by running the reference's
get_cells variants: flux scalers (PiecewiseScaler), the randomised deal of
photons, cells from given time intervals and from contiguous exposures.

Runs ONLY in the build container (needs /root/reference); see
make_golden.py for how the reference is imported.  Only data is written.

Usage:  python tools/make_golden_variants.py [outdir]
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg   # noqa: E402


def per_cell(cells):
    """Order-independent description of a list of reference cells."""
    return dict(starts=np.array([c.tstart for c in cells]), stops=np.array([c.tstop for c in cells]),
                exp=np.array([c.exp for c in cells]), SonB=np.array([c.SonB for c in cells]),
                counts=np.array([len(c.we) for c in cells], dtype=np.int64),
                wsum=np.array([np.sum(np.sort(c.we)) for c in cells]),
                w2sum=np.array([np.sum(np.sort(c.we) ** 2) for c in cells]),
                tsum=np.array([np.sum(np.sort(c.ti - c.tstart)) for c in cells]))


def main(outdir):
    from oracle import godot_oracle as orc
    core = mg.import_reference()
    import reference.scalers as rsc
    d = orc.synth(30000, years=12 / 365.25, seed=23)
    t0 = d['TSTART'][0]
    out = dict(mg.inputs_of(d))
    data = mg.make_data(core, d)
    # scalers: three flux states for the source, two for the background, edges off the cell grid
    s_starts = t0 + np.array([0.0, 3.3, 7.1]) * 86400
    s_stops = t0 + np.array([3.3, 7.1, 12.5]) * 86400
    s_scales = np.array([1.0, 2.5, 0.4])
    b_starts = t0 + np.array([0.0, 5.05]) * 86400
    b_stops = t0 + np.array([5.05, 12.5]) * 86400
    b_scales = np.array([1.0, 1.3])
    out.update(s_starts=s_starts, s_stops=s_stops, s_scales=s_scales, b_starts=b_starts, b_stops=b_stops,
               b_scales=b_scales)
    with mg.quiet():
        src = rsc.PiecewiseScaler(s_starts, s_stops, s_scales)
        bkg = rsc.PiecewiseScaler(b_starts, b_stops, b_scales)
        cells = data.get_cells(tcell=3600., use_barycenter=False, src_scaler=src, bkg_scaler=bkg)
        for k, v in per_cell(cells).items():
            out['scaled_' + k] = v
        out['scaled_we'] = np.concatenate([c.we for c in cells])
        ts = data.get_cells(tcell=3600., use_barycenter=False, src_scaler=src, time_series_only=True)
        out.update(scaled_ts_sexp=ts.sexp, scaled_ts_bexp=ts.bexp, scaled_ts_weights=ts.weights,
                   scaled_ts_weights2=ts.weights2, scaled_ts_counts=np.asarray(ts.counts, dtype=np.int64))
        # scaler values themselves
        out['scaler_rows'] = src(d['TSTART'], d['TSTOP'])
        out['scaler_events'] = src(d['ti'])
        # the randomised deal
        cells = data.get_cells(tcell=3600., use_barycenter=False, randomize=True, seed=3)
        for k, v in per_cell(cells).items():
            out['rand_' + k] = v
        cells = data.get_cells(tcell=3600., use_barycenter=False, randomize=True, seed=4, src_scaler=src)
        for k, v in per_cell(cells).items():
            out['randscaled_' + k] = v
        # contiguous exposures and cells on them
        gs, ge = data.get_contiguous_exposures()
        out.update(contig_starts=gs, contig_stops=ge)
        gs2, ge2 = data.get_contiguous_exposures(tstart=t0 + 1.2345 * 86400, tstop=t0 + 9.8765 * 86400)
        out.update(contig2_starts=gs2, contig2_stops=ge2)
        cells = data.get_contiguous_exposure_cells()
        for k, v in per_cell(cells).items():
            out['contig_' + k] = v
        cells = data.get_contiguous_exposure_cells(randomize=True, seed=5)
        for k, v in per_cell(cells).items():
            out['contigrand_' + k] = v
    os.makedirs(outdir, exist_ok=True)
    np.savez_compressed(os.path.join(outdir, 'cell_variants.npz'), **out)
    print('cell_variants: %d scaled cells, %d contiguous stretches' % (len(out['scaled_exp']), len(gs)))


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(mg.REPO, 'tests', 'golden'))
