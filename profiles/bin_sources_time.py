import time, torch, numpy as np
from godot_amd import core, ops, synth
dev = torch.device('cuda:0')
nsrc = 64
srcs = []
for k in range(nsrc):
    ss = synth.synth(1_000_000, years=10.0, seed=1000 + k)
    srcs.append(core.Data(ss['ti'], ss['we'], ss['TSTART'], ss['TSTOP'], ss['LIVETIME'], ss['exposure'], device=dev))
for d in srcs:
    d._plain_exposure_table()
torch.cuda.synchronize()
for n in (16, 64):
    for per in (True, False):
        core.get_cells_batch(srcs[:n], tcell=86400., per_source_calls=per)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            b = core.get_cells_batch(srcs[:n], tcell=86400., per_source_calls=per)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        print('sources %d per_source_calls=%s: %.3f ms (min of 5), cells %d' % (n, per, min(ts) * 1e3, len(b.cells.lo)), flush=True)
core.pack_sources(srcs)
for n in (16, 64):
    core.get_cells_batch(srcs[:n], tcell=86400.)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        b = core.get_cells_batch(srcs[:n], tcell=86400.)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    print('sources %d packed (core.pack_sources: the photon lists in one allocation): %.3f ms (min of 5)' % (n, min(ts) * 1e3), flush=True)
# without the concatenation of the photon lists (what a caller that keeps them concatenated pays)
desc = []
for d in srcs:
    t0, t1, tc, nc = d._uniform_cells(None, None, 86400., True)
    desc.append(dict(tstart=d.TSTART, tstop=d.TSTOP, exposure=d.exposure, livetime=d.LIVETIME,
                     table=d._plain_exposure_table(), ti=d.ti, s_over_e=d.S / d.E, b_over_e=d.B / d.E,
                     t0=t0, tcell=tc, ncell=nc))
ops.bin_sources(desc)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    t0 = time.perf_counter()
    ops.bin_sources(desc)
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
print('ops.bin_sources of 64 sources alone: %.3f ms' % (min(ts) * 1e3), flush=True)
# the input of a power spectrum per source: 300-s bins (1.05e6 a source)
for n in (16,):
    for name, f in (('per source', lambda: [d.get_cells(tcell=300., time_series_only=True, use_barycenter=False,
                                                         quiet=True) for d in srcs[:n]]),
                    ('batch', lambda: core.get_timeseries_batch(srcs[:n], tcell=300.))):
        f()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            r = f()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        print('time series of %d sources in 300-s bins, %s: %.3f ms (min of 5), bins %d' %
              (n, name, min(ts) * 1e3, sum(len(x.exp) for x in r)), flush=True)
