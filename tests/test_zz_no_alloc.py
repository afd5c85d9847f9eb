"""kry_apply_delta_noalloc: the state update of ImplicitEuler::timestep
(include/time_disc.hpp:281-286) for callers that cannot allocate device memory.
It must give exactly what kry_ie_apply_delta gives (updated temperatures bit
for bit, the same maximum with the reference's std::max fold semantics) on two
calls in a row, the second on a smaller mesh (the scratch array keeps the first
call's partial results), and allocate nothing."""
import ctypes as C
import math

import numpy as np
import pytest

from krylov_amd import capi, host, problems

pytestmark = pytest.mark.gpu


def _existing(ctx, cx, cy, delta):
    dev = host.ImplicitEuler(ctx, **problems.s_iso(cx, cy), m=3)
    T0 = dev.field(0)
    return T0, dev.apply_delta(delta), dev.field(0)


def _same(a, b):
    return a == b or (math.isnan(a) and math.isnan(b))


def test_no_alloc_state_update_equals_existing_entry_point(ctx):
    import torch
    L = capi.lib()
    rng = np.random.default_rng(7)
    free_before = None
    for call, (cx, cy, nan_cells) in enumerate([(700, 300, [(350, 17), (600, 299)]), (90, 41, []), (33, 20, [(32, 19)])]):
        delta = rng.uniform(-2.0, 2.0, (cx + 2, cy + 2))
        for (i, j) in nan_cells:
            delta[i + 1, j + 1] = np.nan
        delta[0, :] = np.nan  # ghost entries are ignored
        T0, want_max, want_T = _existing(ctx, cx, cy, delta)
        T_dev = torch.from_numpy(T0.copy()).to("cuda:0")
        d_dev = torch.from_numpy(delta).to("cuda:0")
        out_dev = torch.full((1,), -7.0, dtype=torch.float64, device="cuda:0")
        torch.cuda.synchronize()
        if call == 1:
            free_before = torch.cuda.mem_get_info()[0]
        capi.check(L.kry_apply_delta_noalloc(ctx.h, cx, cy, C.c_void_p(T_dev.data_ptr()), C.c_void_p(d_dev.data_ptr()),
                                             C.c_void_p(out_dev.data_ptr())))
        ctx.synchronize()
        if call == 1:
            assert torch.cuda.mem_get_info()[0] == free_before  # nothing was allocated on the device
        got_max = float(out_dev.cpu()[0])
        assert _same(got_max, want_max), (call, got_max, want_max)
        assert np.array_equal(T_dev.cpu().numpy(), want_T, equal_nan=True)
    # last case: the last interior cell is NaN, so is the reference's maximum
    assert math.isnan(got_max)


def test_no_alloc_entry_point_under_graph_capture(ctx):
    """The callers this entry point is for include code under hipGraph capture.  The call is recorded between
    hipStreamBeginCapture and hipStreamEndCapture on the context's stream (any allocation, free or wait for the
    device inside it would end the capture with an error) and the graph is replayed twice with new inputs in the
    same buffers; every replay must give what kry_ie_apply_delta gives."""
    import torch
    L = capi.lib()
    cx, cy = 260, 120
    rng = np.random.default_rng(11)
    stream = torch.cuda.ExternalStream(int(ctx.stream), device="cuda:0")
    T_dev = torch.zeros((cx + 2, cy + 2), dtype=torch.float64, device="cuda:0")
    d_dev = torch.zeros((cx + 2, cy + 2), dtype=torch.float64, device="cuda:0")
    out_dev = torch.full((1,), -7.0, dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=stream):
        capi.check(L.kry_apply_delta_noalloc(ctx.h, cx, cy, C.c_void_p(T_dev.data_ptr()), C.c_void_p(d_dev.data_ptr()),
                                             C.c_void_p(out_dev.data_ptr())))
    for nan_cells in ([], [(100, 50), (259, 119)]):
        delta = rng.uniform(-1.0, 1.0, (cx + 2, cy + 2))
        for (i, j) in nan_cells:
            delta[i + 1, j + 1] = np.nan
        T0, want_max, want_T = _existing(ctx, cx, cy, delta)
        T_dev.copy_(torch.from_numpy(T0))
        d_dev.copy_(torch.from_numpy(delta))
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        got_max = float(out_dev.cpu()[0])
        assert _same(got_max, want_max), (got_max, want_max)
        assert np.array_equal(T_dev.cpu().numpy(), want_T, equal_nan=True)
    assert math.isnan(got_max)


def test_no_alloc_rejects_oversized_meshes_and_null_pointers(ctx):
    L = capi.lib()
    fake = C.c_void_p(256)
    assert L.kry_apply_delta_noalloc(ctx.h, 50000, 50000, fake, fake, fake) == capi.ERR_INVALID
    assert b"padded cells" in L.kry_last_error()
    assert L.kry_apply_delta_noalloc(ctx.h, 0, 4, fake, fake, fake) == capi.ERR_INVALID
    assert L.kry_apply_delta_noalloc(ctx.h, 4, 4, None, fake, fake) == capi.ERR_INVALID
    assert L.kry_apply_delta_noalloc(None, 4, 4, fake, fake, fake) == capi.ERR_INVALID
