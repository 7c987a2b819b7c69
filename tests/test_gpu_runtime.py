"""The library's runtime pieces that are not a stage: the arena of temporaries (csrc/util.hip)."""
import ctypes as C

import pytest

pytestmark = pytest.mark.gpu


def _arena(lib, st):
    live, res = C.c_int64(-1), C.c_int64(-1)
    assert lib.xmap_debug_arena(st, C.byref(live), C.byref(res)) == 0
    return int(live.value), int(res.value)


def test_arena_resets_after_an_error_return_and_trims():
    """An entry point that returns between a malloc and its free (XM_HIP / XM_ARG / `if (rc) return rc`) must not leave
    the arena's live count above zero (from then on every call would bump-allocate fresh HBM), and an idle arena keeps at
    most 256 MiB."""
    import torch
    from xmap.engine import hipabi as abi
    lib = abi.lib
    st = C.c_void_p(torch.cuda.current_stream("cuda:0").cuda_stream)
    assert lib.xmap_debug_arena_call(st, C.c_int64(1 << 20), 0) == 0
    live, res0 = _arena(lib, st)
    assert live == 0
    # the error path: two temporaries taken, then XM_ARG fails
    assert lib.xmap_debug_arena_call(st, C.c_int64(1 << 20), 1) == abi.ERR_ARG
    live, res1 = _arena(lib, st)
    assert live == 0 and res1 == res0
    # the same memory is handed out again (no growth) ...
    for _ in range(4):
        assert lib.xmap_debug_arena_call(st, C.c_int64(1 << 20), 1) == abi.ERR_ARG
    assert _arena(lib, st) == (0, res0)
    # ... a high-water temporary does not stay pinned once the arena is idle ...
    assert lib.xmap_debug_arena_call(st, C.c_int64(600 << 20), 0) == 0
    live, res2 = _arena(lib, st)
    assert live == 0 and res2 <= (256 << 20)
    # ... and xmap_trim returns the rest
    assert lib.xmap_trim() == 0
    assert _arena(lib, st) == (0, 0)
