"""The entry point that allocates nothing (godot_select_cells_static) against
godot_select_cells, through the C ABI: the same indices and count, bit for bit, on two
calls in a row, the second with a smaller input (the file-scope device array keeps what
the first call left in it)."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _both(lib_mod, exp, threshold):
    """(indices, count) from the workspace entry point and from the static one."""
    lib = lib_mod.lib
    dev = exp.device
    n = exp.numel()
    stream = torch.cuda.current_stream(dev).cuda_stream
    out = []
    ws = torch.empty(max(int(lib.godot_select_workspace_bytes(n)), 1), dtype=torch.uint8, device=dev)
    for static in (False, True):
        keep = torch.full((max(n, 1),), -7, dtype=torch.int64, device=dev)
        nk = torch.full((1,), -7, dtype=torch.int64, device=dev)
        if static:
            rc = lib.godot_select_cells_static(exp.data_ptr(), n, float(threshold), keep.data_ptr(), nk.data_ptr(),
                                               stream)
        else:
            rc = lib.godot_select_cells(exp.data_ptr(), n, float(threshold), keep.data_ptr(), nk.data_ptr(),
                                        ws.data_ptr(), ws.numel(), stream)
        assert rc == 0, lib_mod.last_error()
        torch.cuda.synchronize()
        k = int(nk.item())
        out.append((keep.cpu().numpy(), k))
    return out


def test_static_entry_point_equals_the_workspace_one_on_two_calls_in_a_row():
    from godot_amd import _lib
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    rng = np.random.default_rng(8)
    # first a large input (many tiles of the scan), then a smaller one, then a tiny one
    for n, thr in ((3_000_001, 0.0), (70_000, 0.0), (5, 0.5), (2048, 0.0)):
        e = rng.uniform(0.0, 1.0, n)
        e[rng.random(n) < 0.3] = 0.0
        exp = torch.from_numpy(e).to(dev)
        (ka, na), (kb, nb) = _both(_lib, exp, thr)
        want = np.flatnonzero(e > thr)
        assert na == nb == len(want)
        assert np.array_equal(ka[:na], want) and np.array_equal(kb[:nb], want)
        assert np.array_equal(ka, kb)          # what lies beyond the count is untouched in both


def test_static_entry_point_rejects_what_its_scratch_cannot_hold_and_handles_empty_input():
    from godot_amd import _lib
    lib = _lib.lib
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    nmax = int(lib.godot_select_cells_static_max())
    assert nmax == 16_777_216
    one = torch.zeros(1, dtype=torch.float64, device=dev)
    keep = torch.zeros(1, dtype=torch.int64, device=dev)
    nk = torch.full((1,), 5, dtype=torch.int64, device=dev)
    # the size is checked before any pointer is looked at
    rc = lib.godot_select_cells_static(one.data_ptr(), nmax + 1, 0.0, keep.data_ptr(), nk.data_ptr(), stream)
    assert rc == -4 and 'at most' in _lib.last_error()
    rc = lib.godot_select_cells_static(one.data_ptr(), 0, 0.0, keep.data_ptr(), nk.data_ptr(), stream)
    torch.cuda.synchronize()
    assert rc == 0 and int(nk.item()) == 0


def test_static_entry_point_allocates_nothing():
    """torch's allocator sees no new device memory across the calls (the library itself has
    no allocation call in this path: the scratch is a symbol of its code object)."""
    from godot_amd import _lib, ops
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    exp = torch.rand(500_000, dtype=torch.float64, device=dev)
    keep = torch.empty(exp.numel(), dtype=torch.int64, device=dev)
    nk = torch.zeros(1, dtype=torch.int64, device=dev)
    ops.select_cells_static(exp, 0.25, keep, nk)      # first call on the device: looks the symbol up
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info(dev)
    before = torch.cuda.memory_allocated(dev)
    for _ in range(3):
        ops.select_cells_static(exp, 0.25, keep, nk)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info(dev)
    assert torch.cuda.memory_allocated(dev) == before and free1 == free0
    k = int(nk.item())
    assert np.array_equal(keep[:k].cpu().numpy(), np.flatnonzero(exp.cpu().numpy() > 0.25))
