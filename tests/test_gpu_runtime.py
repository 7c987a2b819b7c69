"""The library's runtime pieces that are not a stage: the arena of temporaries (csrc/util.hip)."""
import ctypes as C
import os

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


def test_bench_gpus_2_starts_its_own_ranks():
    """`python3 bench.py --gpus 2` with no launcher environment: the parent starts two ranks (gloo here: both on the one GPU
    of the test box; on a multi-GPU node the default backend is RCCL), ONE line comes back with n_gpus = 2 and the counts
    of the one-rank line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    base = [sys.executable, os.path.join(root, "bench.py"), "--workload", "c1", "--steps", "1", "--warmup", "1", "--no-cpu", "--no-extra"]

    def line(extra, **more):
        p = subprocess.run(base + extra, env=dict(env, **more), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-2000:]
        rows = [x for x in p.stdout.splitlines() if x.strip()]
        assert len(rows) == 1, p.stdout
        return json.loads(rows[0])
    one = line(["--gpus", "1"])
    two = line(["--gpus", "2"], XMAP_DIST_BACKEND="gloo")
    assert one["n_gpus"] == 1 and one["rccl_ranks"] == 0
    assert two["n_gpus"] == 2 and two["rccl_ranks"] == 0 and two["dist_backend"] == "gloo"
    for key in ("D_pairs_evaluated", "D_pairs_kept", "paths", "P_contributions"):
        assert two["config"][key] == one["config"][key], key
    assert two["alterego_rows"] == one["alterego_rows"] and two["profiles"] == one["profiles"]
    assert set(two["kernel_ms_per_rank"]["exchange"]) and len(two["kernel_ms_per_rank"]["stage_b"]) == 2
