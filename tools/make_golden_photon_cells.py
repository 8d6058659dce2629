#!/usr/bin/env python3
"""Generate tests/golden/photon_cells.npz by running the reference's
get_photon_cells (one cell per photon, boundaries at the exposure midpoints,
cells broken at gaps in the exposure table).

Runs ONLY in the build container (needs /root/reference); see
make_golden.py for how the reference is imported.  Only data is written.

Usage:  python tools/make_golden_photon_cells.py [outdir]
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg   # noqa: E402


def main(outdir):
    from oracle import godot_oracle as orc
    core = mg.import_reference()
    d = orc.synth(4000, years=6 / 365.25, seed=31)
    t0 = d['TSTART'][0]
    out = dict(mg.inputs_of(d))
    data = mg.make_data(core, d)
    data.cexposure = np.cumsum(data.exposure)    # what Data.__init__ sets (core.py:1608)
    cases = {'all': (d['TSTART'][0], d['TSTOP'][-1]),
             'part': (t0 + 0.8123 * 86400, t0 + 4.4567 * 86400)}
    for tag, (a, b) in cases.items():
        with mg.quiet():
            cells = data.get_photon_cells(a, b)
        out[tag + '_range'] = np.array([a, b])
        out[tag + '_starts'] = np.array([c.tstart for c in cells])
        out[tag + '_stops'] = np.array([c.tstop for c in cells])
        out[tag + '_exp'] = np.array([c.exp for c in cells])
        out[tag + '_SonB'] = np.array([c.SonB for c in cells])
        out[tag + '_ti'] = np.array([np.ravel(c.ti)[0] for c in cells])
        out[tag + '_we'] = np.array([np.ravel(c.we)[0] for c in cells])
    # through get_cells(tcell=None)
    with mg.quiet():
        cells = data.get_cells(tcell=None, use_barycenter=False)
    out['via_get_cells_starts'] = np.array([c.tstart for c in cells])
    os.makedirs(outdir, exist_ok=True)
    np.savez_compressed(os.path.join(outdir, 'photon_cells.npz'), **out)
    print('wrote', os.path.join(outdir, 'photon_cells.npz'), len(out), 'arrays;',
          len(out['all_starts']), len(out['part_starts']), 'cells')


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(HERE), 'tests', 'golden'))
