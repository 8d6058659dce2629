import time, torch
from godot_amd import core, synth
dev = torch.device('cuda:0')
ss = synth.synth(100_000, years=10.0, seed=1000)
srcs = [core.Data(ss['ti'], ss['we'], ss['TSTART'], ss['TSTOP'], ss['LIVETIME'], ss['exposure'] * (1 + 0.01 * k), device=dev)
        for k in range(32)]
for n in (1, 1, 2, 4, 8, 16, 32):
    for d in srcs:
        d._exposure_table = None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    core.build_exposure_tables(srcs, nstream=n)
    torch.cuda.synchronize()
    print('32 tables of %d rows over %d streams: %.1f ms' % (srcs[0].TSTART.numel(), n, (time.perf_counter() - t0) * 1e3), flush=True)
