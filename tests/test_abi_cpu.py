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
