"""CPU-only checks of the boundary: the shared library loads, exports every
function include/godot_hip.h declares, validates arguments before touching a
device, and the Python layer refuses to run without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(REPO, 'include', 'godot_hip.h')


def _declared():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(godot_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    from godot_amd import _lib
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(_lib.lib, n), 'libgodot_hip.so does not export %s' % n
    # and the ctypes table binds exactly the declared set
    assert sorted(_lib.SIGNATURES) == names
    assert _lib.MISSING == []
    assert _lib.lib.godot_version() >= 100


def test_arguments_are_validated_before_any_device_work():
    from godot_amd import _lib
    lib = _lib.lib
    # negative sizes / null pointers -> GODOT_E_INVALID (-1) with a message
    rc = lib.godot_cell_fit(0, -1, 0, 0, 0, 0, 5, 0, 1.0, 0, 1.0, 0, 0, 0, 0, 0)
    assert rc == -1 and 'negative' in _lib.last_error()
    rc = lib.godot_cell_fit(0, 10, 0, 0, 0, 0, 5, 0, 1.0, 0, 1.0, 0, 0, 0, 0, 0)
    assert rc == -1 and 'null' in _lib.last_error()
    with pytest.raises(_lib.GodotError):
        _lib.call('godot_cell_logpdf', 0, 0, 0, 0, 0, 0, 0, 0, 30.0, 5000, 0, 0, 0, 0, 0, 0)
    # unsupported sizes are refused, not approximated
    rc = lib.godot_rfft_f64(0, 10, 1, 0, 0, 0, 0)
    assert rc == -1 and 'l >= 2' in _lib.last_error()
    rc = lib.godot_z2m(0, 0, 0, 1.0, 1.0, 0, 10, 21, 0, 0, 0, 0, 0)
    assert rc == -4
    # workspace queries are pure host functions
    assert lib.godot_rfft_workspace_bytes(1 << 20) >= 2 * 16 * (1 << 19)
    assert lib.godot_exposure_table_bytes(1000) >= 32000
    assert lib.godot_z2m_workspace_bytes(1000, 100, 2) >= 64000
    assert lib.godot_power_spectrum_workspace_bytes(1 << 10) > 0
    # empty problems succeed without a device
    assert lib.godot_cell_fit(0, 0, 0, 0, 0, 0, 0, 0, 1.0, 0, 1.0, 0, 0, 0, 0, 0) == 0


def test_python_layer_has_no_cpu_path():
    import torch
    from godot_amd import core, ops
    if torch.cuda.is_available():
        pytest.skip('a GPU is visible')
    with pytest.raises(RuntimeError, match='no CPU fallback|needs a ROCm GPU'):
        core.Data(np.arange(5.), np.full(5, .5), np.arange(5.), np.arange(5.) + 1, np.ones(5),
                  np.ones(5))
    with pytest.raises(ValueError, match='must live on the GPU'):
        ops.CellRanges(torch.zeros(3, dtype=torch.float64), torch.zeros(1, dtype=torch.int64),
                       torch.ones(1, dtype=torch.int64), torch.ones(1, dtype=torch.float64),
                       torch.ones(1, dtype=torch.float64))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(REPO, 'godot_amd')
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.hpp', '.cpp', '.h')):
                txt = open(os.path.join(root, f)).read()
                assert 'oracle' not in txt, '%s mentions the oracle' % f


def test_host_side_value_types():
    from godot_amd import core
    c = core.Cell(0., 10., 3.0, [1., 2.], [0.2, 0.9], 0.25)
    c.sanity_check()
    assert c.get_tmid() == 5.0 and c._alpha == 1 and c._beta == 1
    m = core.cell_from_cells([core.Cell(10., 20., 1.0, [11.], [0.5], 0.25), c])
    assert m.tstart == 0. and m.tstop == 20. and m.exp == 4.0 and len(m.we) == 3
    assert abs(m.SonB - 0.25) < 1e-15
    ts = core.CellTimeSeries(np.arange(4.) * 100, np.arange(1., 5.) * 100, np.array([5e4, 1e4, 6e4, 0.]),
                             np.ones(4), np.ones(4), np.array([3, 2, 1, 0]), np.ones(4), np.ones(4),
                             np.zeros(4))
    assert ts.tsamp() == 100. and ts.tspan() == 400. and ts.ps_nbin() == 2
    # the exposure mask zeroes the moments in place (minimum_exposure = 3e4)
    assert list(ts.counts) == [3, 0, 1, 0] and list(ts.weights) == [1, 0, 1, 0]
    with pytest.raises(Exception):
        ts.get_topo_bins()
    assert abs(float(core.met2mjd(239557417.0)) - 54682.65603) < 1e-4
    assert core.infer_met(3e8) == 3e8


def test_piecewise_scaler_host_logic_matches_reference(golden):
    """The interval arithmetic of PiecewiseScaler runs on the host (no GPU needed)."""
    from godot_amd.scalers import PiecewiseScaler
    g = golden('cell_variants.npz')
    src = PiecewiseScaler(g['s_starts'], g['s_stops'], g['s_scales'])
    assert np.array_equal(src(g['in_TSTART'], g['in_TSTOP']), g['scaler_rows'])
    assert np.array_equal(src(g['in_ti']), g['scaler_events'])


def test_cell_timeseries_file_format_is_the_references(tmp_path):
    """CellTimeSeries.save/load: a file written by the reference loads here, and the
    file written here has the reference's keys, dtypes and contents."""
    from godot_amd import core
    ref = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'cell_timeseries_ref.npz')
    ts = core.CellTimeSeries.load(ref)
    assert ts.timesys == 'barycenter' and ts.minimum_exposure == 0.0
    mine = str(tmp_path / 'mine.npz')
    ts.save(mine)
    with np.load(ref, allow_pickle=False) as q, np.load(mine, allow_pickle=False) as m:
        assert sorted(q.files) == sorted(m.files)
        for k in q.files:
            if k == 'metadata':   # [str(minimum_exposure), timesys]; '0' comes back as '0.0', as in the reference
                assert float(q[k][0]) == float(m[k][0]) and q[k][1] == m[k][1] and m[k].dtype.kind == 'U'
                continue
            assert q[k].dtype == m[k].dtype and np.array_equal(q[k], m[k]), k
        for k in core.CellTimeSeries._KEYS + ['alt_starts', 'alt_stops']:
            assert np.array_equal(np.asarray(getattr(ts, k)), q[k]), k
    topo = ts.get_topo_bins()
    assert np.array_equal(topo[0], ts.alt_starts) and ts.get_bary_bins()[0] is ts.starts
    # the uncompressed variant and a topocentric series without alternative bins
    ts2 = core.CellTimeSeries(ts.starts, ts.stops, ts.exp, ts.sexp, ts.bexp, ts.counts, ts.weights, ts.weights2,
                              ts.deadtime, timesys='topocentric', minimum_exposure=0)
    f2 = str(tmp_path / 'plain.npz')
    ts2.save(f2, compress=False)
    back = core.CellTimeSeries.load(f2)
    assert back.timesys == 'topocentric' and back.alt_starts is None
    with pytest.raises(Exception):
        back.get_bary_bins()


def test_every_entry_point_rejects_null_pointers_without_a_gpu():
    """All error-returning entry points, called with null pointers and sizes 0, 1, -1 and 5:
    none crashes or reaches a kernel launch; a positive size with null pointers is an error,
    size 0 is either nothing to do (0) or an error."""
    import ctypes
    from godot_amd import _lib
    ncalls = 0
    for name, (rt, args) in _lib.SIGNATURES.items():
        if rt is not _lib.c_int or name == 'godot_version' or name.endswith('_supported'):   # predicates: 0 or 1
            continue
        fn = getattr(_lib.lib, name)
        for fill in (0, 1, -1, 5):
            call = [None if a is ctypes.c_void_p else (float(max(fill, 0)) if a is ctypes.c_double else fill)
                    for a in args]
            rc = fn(*call)
            ncalls += 1
            assert rc <= 0, (name, fill, rc)
            if rc:
                assert _lib.last_error(), (name, fill)
            if fill == 1:
                assert rc < 0, (name, 'accepted null pointers for one element')
    assert ncalls >= 120


def test_source_descriptor_layout_is_the_headers(tmp_path):
    """ops.SOURCE_CELLS and ops.EXPOSURE_ROWS, the numpy records the Python layer fills, have the sizes
    and the field offsets that a C compiler gives godot_source_cells and godot_exposure_rows of
    include/godot_hip.h."""
    import shutil
    import subprocess
    from godot_amd import ops
    if shutil.which('gcc') is None:
        pytest.skip('no C compiler')
    records = (('godot_source_cells', ops.SOURCE_CELLS), ('godot_exposure_rows', ops.EXPOSURE_ROWS))
    src = tmp_path / 'layout.c'
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "godot_hip.h"\nint main(void){\n'
                   + ''.join('printf("%%zu\\n", sizeof(%s));\n' % ctype
                             + ''.join('printf("%%zu\\n", offsetof(%s, %s));\n' % (ctype, n) for n in rec.names)
                             for ctype, rec in records)
                   + 'return 0;}\n')
    exe = tmp_path / 'layout'
    subprocess.run(['gcc', '-I', os.path.dirname(HEADER), str(src), '-o', str(exe)], check=True)
    out = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    for ctype, rec in records:
        mine = [rec.itemsize] + [rec.fields[n][1] for n in rec.names]
        assert out[:len(mine)] == mine, ctype
        out = out[len(mine):]
    assert out == []
    # and the call refuses a table it cannot index before any device work
    from godot_amd import _lib
    rec = np.zeros(1, dtype=ops.SOURCE_CELLS)
    rec['n_ft2'], rec['ncell'], rec['tcell'] = 10, 5, 1.0
    assert _lib.lib.godot_bin_sources(rec.ctypes.data, 1, 5, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0) == -1
    assert 'null pointer' in _lib.last_error()
    assert _lib.lib.godot_bin_sources_workspace_bytes(3, 100) >= 3 * ops.SOURCE_CELLS.itemsize + 32 * 103
