"""A slice of the multi-source job: 256 sources x 1e6 photons (16 distinct photon lists, each
used by 16 sources with its own brightness), daily cells: binning, fit, profile grid."""
import time, torch, numpy as np
from godot_amd import core, ops, synth
dev = torch.device('cuda:0')
raw = [synth.synth(1_000_000, years=10.0, seed=1000 + k) for k in range(16)]
srcs = []
for j in range(16):
    for k, ss in enumerate(raw):
        d = core.Data(ss['ti'], ss['we'], ss['TSTART'], ss['TSTOP'], ss['LIVETIME'], ss['exposure'], device=dev)
        d.S = ss['S'] * (1.0 + 0.05 * j)
        srcs.append(d)
torch.cuda.synchronize()
t0 = time.perf_counter()
for d in srcs[:32]:
    d._plain_exposure_table()
torch.cuda.synchronize()
t1 = time.perf_counter()
core.build_exposure_tables(srcs[32:64], nstream=4)
torch.cuda.synchronize()
t2 = time.perf_counter()
print('exposure tables of 32 sources (8.3e6 rows each): one after another %.1f ms, over four streams %.1f ms' %
      ((t1 - t0) * 1e3, (t2 - t1) * 1e3), flush=True)
core.build_exposure_tables(srcs, nstream=4)
torch.cuda.synchronize()
print('sources', len(srcs), flush=True)

def timed(f, n=3):
    f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        r = f()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3, r

ms_b, batch = timed(lambda: core.get_cells_batch(srcs, tcell=86400.))
core.pack_sources(srcs)
ms_p, batch = timed(lambda: core.get_cells_batch(srcs, tcell=86400.))
print('binning after core.pack_sources (photon lists in one allocation, no copy per call): %.2f ms' % ms_p, flush=True)
cr = batch.cells.ranges()
ms_f, _ = timed(lambda: ops.cell_fit(cr))
ms_g, _ = timed(lambda: ops.cell_logpdf(cr, dlogl=30, npt=200))
print('256 sources x 1e6 photons, %d daily cells: binning %.2f ms, fit %.3f ms, 200-point grid %.2f ms' %
      (cr.ncell, ms_b, ms_f, ms_g), flush=True)
print('sources/s binning+fit: %.3g' % (len(srcs) / ((ms_b + ms_f) * 1e-3)), flush=True)
