"""ctypes binding of libgodot_hip.so (the C ABI declared in include/godot_hip.h).

There is no fallback: if the shared library is missing the import fails with
instructions to build it, and every call that returns a non-zero code raises
GodotError with the library's message.
"""
import ctypes
import os

# torch first: it brings its own copy of the HIP runtime (libamdhip64), and the library's
# dependency on that name must resolve to the copy torch uses.  Loaded the other way round
# the process holds two runtimes, and the library's launches see no device while torch's
# tensors, whose pointers and streams it is handed, live in the other one.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# GODOT_HIP_LIBRARY names another build of the same library (e.g. one under test)
LIB_PATH = os.environ.get('GODOT_HIP_LIBRARY') or os.path.join(_HERE, 'libgodot_hip.so')

c_f64p = ctypes.c_void_p      # device pointers travel as integers
c_i64p = ctypes.c_void_p
c_i32p = ctypes.c_void_p
c_i64 = ctypes.c_int64
c_f64 = ctypes.c_double
c_int = ctypes.c_int
c_size = ctypes.c_size_t
c_stream = ctypes.c_void_p
c_ws = ctypes.c_void_p


class GodotError(RuntimeError):
    pass


if not os.path.exists(LIB_PATH):
    raise ImportError(
        'godot_amd: %s is missing. Build it with `make -C godot_amd/csrc` or '
        '`python -c "import __graft_entry__ as g; g.build()"` (needs hipcc; '
        'cross-compiles for gfx950 without a GPU).' % LIB_PATH)

lib = ctypes.CDLL(LIB_PATH)

_CELLS = [c_f64p, c_i64, c_i64p, c_i64p]          # we, nph, lo, hi

# name -> (restype, argtypes); int-returning functions are error-checked
SIGNATURES = {
    'godot_last_error': (ctypes.c_char_p, []),
    'godot_version': (c_int, []),
    'godot_cell_fit': (c_int, _CELLS + [c_f64p, c_f64p, c_i64, c_f64p, c_f64, c_f64p, c_f64,
                                        c_int, c_f64p, c_f64p, c_i32p, c_stream]),
    'godot_cell_fit_shaped': (c_int, _CELLS + [c_f64p, c_f64p, c_i64, c_i64, c_f64p, c_f64, c_f64p, c_f64,
                                               c_int, c_f64p, c_f64p, c_i32p, c_stream]),
    'godot_cell_logpdf_shaped': (c_int, _CELLS + [c_f64p, c_f64p, c_i64, c_i64, c_f64p, c_f64, c_int, c_int,
                                                  c_int, c_f64p, c_f64p, c_i32p, c_stream]),
    'godot_cell_loglike': (c_int, _CELLS + [c_f64p, c_f64p, c_i64, c_f64p, c_int, c_f64p, c_stream]),
    'godot_cell_logpdf': (c_int, _CELLS + [c_f64p, c_f64p, c_i64, c_f64p, c_f64, c_int, c_int,
                                           c_int, c_f64p, c_f64p, c_i32p, c_stream]),
    'godot_profile_quantiles': (c_int, [c_f64p, c_f64p, c_i64, c_int, c_f64, c_f64, c_f64p, c_f64p,
                                        c_f64p, c_stream]),
    'godot_profile_quantiles_linspace': (c_int, [c_f64p, c_f64p, c_i64, c_int, c_f64, c_f64, c_f64p, c_f64p,
                                        c_f64p, c_stream]),
    'godot_lightcurve_rows': (c_int, [c_f64p] * 7 + [c_i64, c_f64, c_f64p, c_stream]),
    'godot_met2mjd': (c_int, [c_f64p, c_i64, c_f64, c_f64p, c_stream]),
    'godot_cell_beta_max': (c_int, _CELLS + [c_f64p, c_f64p, c_i64, c_f64p, c_f64, c_f64p, c_f64,
                                             c_int, c_f64p, c_f64p, c_i32p, c_stream]),
    'godot_cell_fit_joint': (c_int, _CELLS + [c_f64p, c_f64p, c_i64, c_f64p, c_f64, c_f64p, c_f64,
                                              c_int, c_f64p, c_f64p, c_f64p, c_i32p, c_stream]),
    'godot_cell_logpdf_profile': (c_int, _CELLS + [c_f64p, c_f64p, c_i64, c_f64p, c_f64, c_int, c_int,
                                                   c_f64p, c_f64p, c_i32p, c_stream]),
    'godot_exposure_table_bytes': (c_size, [c_i64]),
    'godot_exposure_table': (c_int, [c_f64p] * 4 + [c_i64, c_f64, c_f64, c_ws, c_size, c_stream]),
    'godot_exposure_workspace_bytes': (c_size, [c_i64]),
    'godot_exposure_at': (c_int, [c_f64p] * 4 + [c_i64, c_f64, c_f64, c_ws, c_f64p, c_i64, c_int]
                          + [c_f64p] * 4 + [c_ws, c_size, c_stream]),
    'godot_bin_photons': (c_int, [c_f64p, c_i64, c_f64, c_f64p, c_i64, c_i64p, c_i64p, c_i64p,
                                  c_stream]),
    'godot_bin_sources_workspace_bytes': (c_size, [c_i64, c_i64]),
    'godot_bin_sources': (c_int, [ctypes.c_void_p, c_i64, c_i64] + [c_f64p] * 6 + [c_i64p] * 3
                          + [c_ws, c_size, c_stream]),
    'godot_cell_moments': (c_int, _CELLS + [c_i64, c_f64p, c_f64p, c_stream]),
    'godot_cell_series': (c_int, _CELLS + [c_f64p, c_i64] + [c_f64p] * 6 + [c_stream]),
    'godot_select_workspace_bytes': (c_size, [c_i64]),
    'godot_select_cells': (c_int, [c_f64p, c_i64, c_f64, c_i64p, c_i64p, c_ws, c_size, c_stream]),
    'godot_select_cells_static_max': (c_i64, []),
    'godot_select_cells_static': (c_int, [c_f64p, c_i64, c_f64, c_i64p, c_i64p, c_stream]),
    'godot_gather_f64': (c_int, [c_f64p, c_i64p, c_i64, c_f64p, c_stream]),
    'godot_gather_i64': (c_int, [c_i64p, c_i64p, c_i64, c_i64p, c_stream]),
    'godot_gather_columns': (c_int, [ctypes.c_void_p, ctypes.c_void_p, c_int, c_i64p, c_i64, c_stream]),
    'godot_rfft_workspace_bytes': (c_size, [c_i64]),
    'godot_rfft_f64': (c_int, [c_f64p, c_i64, c_i64, c_f64p, c_ws, c_size, c_stream]),
    'godot_power_spectrum_workspace_bytes': (c_size, [c_i64]),
    'godot_power_spectrum_parts_supported': (c_int, [c_i64, c_i64, c_int]),
    'godot_power_spectrum_part_rows': (c_i64, []),
    'godot_power_spectrum_part_bytes': (c_size, [c_i64, c_int]),
    'godot_power_spectrum_part_workspace_bytes': (c_size, [c_i64, c_int]),
    'godot_power_spectrum_columns': (c_int, [c_f64p] * 6 + [c_i64, c_i64, c_int, c_int, c_ws, c_ws, c_size, c_stream]),
    'godot_power_spectrum_rows': (c_int, [c_ws, c_f64p, c_i64, c_i64, c_i64] + [c_f64p] * 4 + [c_stream]),
    'godot_power_spectrum_fft': (c_int, [c_f64p] * 6 + [c_i64, c_i64] + [c_f64p] * 4
                                 + [c_ws, c_size, c_stream]),
    'godot_power_spectrum_from_parts': (c_int, [c_f64p] * 5 + [c_i64] + [c_f64p] * 4 + [c_stream]),
    'godot_row_peaks_workspace_bytes': (c_size, [c_i64, c_i64]),
    'godot_row_peaks': (c_int, [c_f64p, c_i64, c_i64, c_i64, c_f64p, c_i64p, c_ws, c_size, c_stream]),
    'godot_spectrum_frequencies': (c_int, [c_i64, c_f64, c_f64, c_int, c_f64p, c_stream]),
    'godot_power_spectrum_simple': (c_int, [c_f64p] * 3 + [c_i64, c_f64p, c_f64p, c_ws, c_size,
                                                           c_stream]),
    'godot_trigsum_cells': (c_int, [c_f64p] * 4 + [c_i64, c_f64p, c_i64, c_f64, c_f64p, c_stream]),
    'godot_fit_harmonics': (c_int, [c_f64p] * 4 + [c_i64, c_f64, c_f64p, c_i64, c_f64p, c_stream]),
    'godot_ls_cells': (c_int, [c_f64p, c_f64p, c_i64, c_f64p, c_i64, c_f64p, c_stream]),
    'godot_orbital_workspace_bytes': (c_size, []),
    'godot_orbital_modulation_step': (c_int, [c_f64p] * 4 + [c_i64, c_f64, c_f64p, c_f64p, c_ws, c_size,
                                                             c_stream]),
    'godot_z2m_workspace_bytes': (c_size, [c_i64, c_i64, c_int]),
    'godot_z2m': (c_int, [c_f64p, c_f64p, c_i64, c_f64, c_f64, c_i64, c_i64, c_int, c_f64p, c_f64p,
                          c_ws, c_size, c_stream]),
    'godot_cell_sums': (c_int, [c_f64p, c_i64, c_i64p, c_i64p, c_i64, c_f64p, c_f64, c_f64p, c_f64, c_int,
                                c_f64p, c_stream]),
    'godot_z2m_sums': (c_int, [c_f64p, c_f64p, c_i64, c_f64, c_f64, c_i64, c_i64, c_int, c_f64p, c_f64p,
                               c_ws, c_size, c_stream]),
    'godot_z2m_finish': (c_int, [c_f64p, c_f64p, c_i64, c_int, c_f64p, c_f64p, c_stream]),
    'godot_exposure_checkpoints_bytes': (c_size, [c_i64]),
    'godot_exposure_checkpoints_workspace_bytes': (c_size, [ctypes.c_void_p, c_i64]),
    'godot_exposure_checkpoints': (c_int, [ctypes.c_void_p, c_i64, c_ws, c_size, c_stream]),
    'godot_exposure_table_rows': (c_int, [c_f64p] * 6 + [c_i64, c_ws, c_size, c_stream]),
    'godot_exposure_at_rows': (c_int, [c_f64p] * 6 + [c_i64, c_ws, c_f64p, c_i64, c_int, c_f64p, c_f64p,
                                                       c_f64p, c_f64p, c_ws, c_size, c_stream]),
    'godot_search_sorted': (c_int, [c_f64p, c_i64, c_f64p, c_i64, c_int, c_i64p, c_stream]),
    'godot_piecewise_lookup': (c_int, [c_f64p, c_f64p, c_i64, c_f64p, c_i64, c_f64p, c_stream]),
    'godot_piecewise_average': (c_int, [c_f64p, c_f64p, c_f64p, c_f64p, c_f64p, c_i64, c_f64p, c_f64p, c_i64, c_f64p,
                                        c_stream]),
    'godot_rescale_weights': (c_int, [c_f64p, c_f64p, c_f64, c_f64p, c_f64, c_i64, c_f64p, c_stream]),
    'godot_bb_workspace_bytes': (c_size, [c_i64, c_int]),
    'godot_bb_fitness': (c_int, [c_f64p, c_f64p, c_i64, c_int, c_int, c_i64, c_i64, c_ws, c_f64p, c_f64p,
                                 c_stream]),
    'godot_debug_spread_spin_limit': (c_int, [ctypes.c_long]),
    'godot_bb_partition': (c_int, [c_f64p, c_i64, c_f64, c_f64p, c_i64p, c_stream]),
}

MISSING = []
for _name, (_res, _args) in SIGNATURES.items():
    try:
        _fn = getattr(lib, _name)
    except AttributeError:
        MISSING.append(_name)
        continue
    _fn.restype = _res
    _fn.argtypes = _args


def last_error():
    return lib.godot_last_error().decode('utf-8', 'replace')


def call(name, *args):
    """Call an int-returning entry point; raise GodotError on failure."""
    if name in MISSING:
        raise GodotError('%s is not exported by %s' % (name, LIB_PATH))
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise GodotError('%s failed (%d): %s' % (name, rc, last_error()))
    return rc
