"""vsv_encode_windows_noalloc -- the one entry point that allocates nothing (workspace = a device array built
into the library) -- through the C ABI against vsv_encode_windows, bit for bit, on two calls in a row, the
second with a smaller input (the shared array keeps the first call's contents); then the size limit; and under a
HIP graph capture after vsv_init_device().
Collected last (test_zz_*): the round's end runs pytest -x."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import vsv_oracle as vo

pytestmark = pytest.mark.gpu


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def test_noalloc_under_graph_capture_after_init():
    """The caller the entry point exists for: a HIP graph capture.  vsv_init_device() resolves the scratch array's
    address eagerly (the lookup must not happen inside a capture); the captured call then only enqueues kernels, and
    every replay -- on new inputs written into the same buffers -- equals the workspace entry point bit for bit."""
    from vsv import lib as vlib
    from vsv.engine import Engine
    L, X = 12, 3
    cfg = vo.make_config(L, X)
    eng = Engine(cfg, vo.init_weights(cfg, 4, 2.0, np.float32), "cuda:0")
    lib = eng.lib
    vlib.check(lib.vsv_init_device(), "vsv_init_device")
    vlib.check(lib.vsv_init_device(), "vsv_init_device")          # idempotent
    rng = np.random.default_rng(5)
    n_win, T = 120, 20
    vol = _t(0.2 + 0.1 * rng.standard_normal((n_win + T - 1, 5, 5)))
    ex = _t(0.01 * rng.standard_normal((n_win + T - 1, X)))
    out = torch.full((n_win, T, L), float("nan"), dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    free_before = torch.cuda.mem_get_info()[0]
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        rc = lib.vsv_encode_windows_noalloc(C.byref(eng.dims), eng.packed.data_ptr(), 1, vol.data_ptr(), ex.data_ptr(),
                                            n_win, T, out.data_ptr(), st)
    vlib.check(rc, "vsv_encode_windows_noalloc (captured)")
    for k in range(2):
        if k:
            vol.copy_(_t(0.25 + 0.1 * rng.standard_normal((n_win + T - 1, 5, 5))))
            ex.copy_(_t(0.01 * rng.standard_normal((n_win + T - 1, X))))
        out.fill_(float("nan"))
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, eng.encode_windows(1, vol, ex, n_win, T)), k
    del graph
    assert torch.cuda.mem_get_info()[0] >= free_before - (8 << 20)   # (the graph object itself; no workspace)


def test_noalloc_encode_windows_equals_workspace_entry_point():
    from vsv import lib as vlib
    from vsv.engine import Engine
    L, X = 12, 3
    cfg = vo.make_config(L, X)
    eng = Engine(cfg, vo.init_weights(cfg, 4, 2.0, np.float32), "cuda:0")
    lib = eng.lib
    rng = np.random.default_rng(3)
    n_cal = 600
    vol = _t(0.2 + 0.1 * rng.standard_normal((n_cal, 5, 5)))
    ex = _t(0.01 * rng.standard_normal((n_cal, X)))
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    calls = 0
    for which, nh in ((0, 2 * L), (1, L)):
        for n_win, T in ((300, 60), (45, 7)):                   # big first, then small: stale scratch behind it
            want = eng.encode_windows(which, vol, ex, n_win, T)
            out = torch.full((n_win, T, nh), float("nan"), dtype=torch.float32, device="cuda:0")
            free_before = torch.cuda.mem_get_info()[0]
            rc = lib.vsv_encode_windows_noalloc(C.byref(eng.dims), eng.packed.data_ptr(), which, vol.data_ptr(),
                                                ex.data_ptr(), n_win, T, out.data_ptr(), st)
            vlib.check(rc, "vsv_encode_windows_noalloc")
            torch.cuda.synchronize()
            if calls > 0:                                         # (the first call on a device looks the array up)
                assert torch.cuda.mem_get_info()[0] >= free_before, "the call took device memory"
            calls += 1
            assert torch.equal(out, want), (which, n_win, T)
    # empty request: nothing to do
    assert lib.vsv_encode_windows_noalloc(C.byref(eng.dims), eng.packed.data_ptr(), 0, vol.data_ptr(), ex.data_ptr(),
                                          0, 5, None, st) == 0
    # beyond the built-in buffer: refused with a clear message, nothing launched
    big_vol = _t(np.zeros((4100, 5, 5)))
    big_ex = _t(np.zeros((4100, X)))
    out = torch.empty((4000, 60, L), dtype=torch.float32, device="cuda:0")
    rc = lib.vsv_encode_windows_noalloc(C.byref(eng.dims), eng.packed.data_ptr(), 1, big_vol.data_ptr(), big_ex.data_ptr(),
                                        4000, 60, out.data_ptr(), st)
    assert rc == -6 and b"scratch" in lib.vsv_last_error()
