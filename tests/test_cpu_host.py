"""CPU-side tests (no GPU): the C-ABI library loads and exports every symbol include/xmap_hip.h declares,
the host logic (id predicates, RDD-like container, pyspark stand-in, shard planning, RNG draws) behaves."""
import os
import re
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_exports_every_declared_symbol():
    """the PRODUCT library exports exactly what the header declares outside #ifdef XMAP_CROSSCHECK; the cross-check library
    (same sources + the test formulations) exports those and the guarded ones; the product does not carry the latter"""
    import ctypes
    hdr = open(os.path.join(ROOT, "include", "xmap_hip.h")).read()
    guarded = "".join(re.findall(r"#ifdef XMAP_CROSSCHECK.*?#endif /\* XMAP_CROSSCHECK \*/", hdr, flags=re.S))
    names = set(re.findall(r"\b(xmap_[a-z0-9_]+)\s*\(", hdr))
    xnames = set(re.findall(r"\b(xmap_[a-z0-9_]+)\s*\(", guarded))
    assert len(names) >= 19 and len(xnames) >= 8 and xnames < names
    from xmap.engine import hipabi      # loads libxmap_hip.so (no compute call is made without a GPU)
    for n in sorted(names - xnames):
        assert hasattr(hipabi.lib, n), n
    assert set(hipabi.EXPORTS) == names - xnames and set(hipabi.XCHECK_EXPORTS) == xnames
    for n in sorted(xnames):
        assert not hasattr(hipabi.lib, n), "test formulation %s in the product library" % n
    X = hipabi.xlib()
    for n in sorted(names):
        assert hasattr(X, n), n
    assert hipabi.lib.xmap_version() >= 100
    # struct layouts match the header (field order / count)
    assert [f[0] for f in hipabi.Ratings._fields_] == re.findall(
        r"(?:const\s+)?\w+\s+\*?(\w+);", hdr[hdr.index("typedef struct xmap_ratings {"):hdr.index("} xmap_ratings;")])
    assert ctypes.sizeof(hipabi.Sim) == 8 * 8
    # every export carries argtypes generated from the header: a mis-typed or missing argument raises in ctypes
    assert set(hipabi.PROTOTYPES) == names
    for n in sorted(names):
        f = getattr(X if n in xnames else hipabi.lib, n)
        assert f.argtypes is not None and len(f.argtypes) == len(hipabi.PROTOTYPES[n]), n
    assert hipabi.PROTOTYPES["xmap_exclusive_scan_i64"] == [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                                            ctypes.c_void_p]
    with pytest.raises((ctypes.ArgumentError, TypeError)):
        hipabi.lib.xmap_exclusive_scan_i64(None, None, None)            # too few arguments
    for st_name, cls in (("xmap_ext_tables", hipabi.ExtTables), ("xmap_path_units", hipabi.PathUnits),
                         ("xmap_path_rows", hipabi.PathRows), ("xmap_path_out", hipabi.PathOut)):
        body = hdr[hdr.index("typedef struct %s {" % st_name):hdr.index("} %s;" % st_name)]
        assert [f[0] for f in cls._fields_] == re.findall(r"\*?(\w+)\s*[;,]", body.split("{", 1)[1]), st_name


def test_item_attrs_predicates():
    from xmap.engine import ids
    iids = ["00aS:", "B0xT:", "T:zS:", "00T:bT:", "1xqS:1:", "S:pT:"]
    pre, suf, mask, flags = ids.item_attrs(iids)
    for a, sa in enumerate(iids):
        assert bool(flags[a] & 1) == ("S:" in sa) and bool(flags[a] & 2) == ("T:" in sa)
        for b, sb in enumerate(iids):
            assert (pre[a] != pre[b]) == (sa[:2] != sb[:2])                  # baselinerSim.py:191
            assert bool((mask[b] >> suf[a]) & 1) == (sa[-2:] in sb)          # extender.py:29-35


def test_local_rdd_and_pyspark_shim(tmp_path):
    from pyspark import SparkContext, SparkConf
    from pyspark.sql import SQLContext, Row
    sc = SparkContext(conf=SparkConf().setAppName("t").set("a", "b"))
    r = sc.parallelize([("a", 1), ("b", 2), ("a", 3)], 4)
    assert r.reduceByKey(lambda x, y: x + y).collect() == [("a", 4), ("b", 2)]
    assert r.map(lambda kv: (kv[0], [kv[1]])).reduceByKey(lambda x, y: x + y).collectAsMap() == {"a": [1, 3], "b": [2]}
    assert r.combineByKey(lambda v: [v], lambda c, v: c + [v], lambda a, b: a + b).collect() == [("a", [1, 3]), ("b", [2])]
    assert r.join(sc.parallelize([("a", "x")])).collect() == [("a", (1, "x")), ("a", (3, "x"))]
    assert r.filter(lambda kv: kv[1] > 1).keys().collect() == ["b", "a"]
    assert r.flatMap(lambda kv: [kv[0]] * kv[1]).count() == 6
    assert r.mapPartitions(lambda it: (v for _, v in it)).reduce(lambda a, b: a + b) == 6
    assert sc.broadcast([1]).value == [1]
    p = tmp_path / "x.txt"
    p.write_text("l1\nl2\n")
    assert sc.textFile("file:" + str(p), 30).collect() == ["l1", "l2"]
    rows = sc.parallelize([Row(id1="a", label=1), Row(id1="a", label=1), Row(id1="b", label=0)])
    rows.toDF().registerTempTable("sim_table")
    out = SQLContext(sc).sql("SELECT DISTINCT id1 FROM sim_table WHERE label = 1").map(lambda l: l.id1).collect()
    assert out == ["a"]


def test_api_surface_matches_reference_names():
    import inspect
    from xmap.utils import assist
    from xmap.core.baselinerSim import BaselinerSim
    from xmap.core.extender import ExtendSim
    from xmap.core.generator import Generator
    import xmap.core as core
    assert list(inspect.signature(assist.baseliner_calculate_sim_pipeline).parameters) == ["sc", "itemsim_tool", "trainRDD"]
    assert list(inspect.signature(assist.extender_pipeline).parameters) == [
        "sc", "sqlContext", "itemsim_tool", "extendsim_tool", "item2item_simRDD"]
    assert list(inspect.signature(assist.generator_pipeline).parameters) == [
        "privatemap_tool", "trainRDD", "extended_simRDD", "private"]
    assert core.extender_pipeline is assist.extender_pipeline
    t = BaselinerSim("adjust_cosine", 50)
    assert (t.method, t.num_atleast) == ("adjust_cosine", 50)
    assert t.significance_weighting(0.5, 10) == 1.0 * 0.5 * 10 / 50 and t.cosine(1.0, 0) == 0.0
    assert ExtendSim(10).top_k == 10
    g = Generator(1, 0.6, "cosine", 0.1)
    assert (g.mapping_range, g.privacy_epsilon, g.sim_method, g.rpo, g.global_sentivity()) == (1, 0.6, "cosine", 0.1, 1)
    assert Generator(1, 0.6, "adjust_cosine", 0.1).global_sentivity() == 2
    # unknown method: the reference ends in `None.cache()` (assist.py:75)
    with pytest.raises(AttributeError):
        assist.baseliner_calculate_sim_pipeline(None, BaselinerSim("pearson", 50), [])


def test_balanced_ranges():
    from xmap.engine.sharded import balanced_ranges
    w = np.array([5, 1, 1, 1, 1, 1, 5, 5])
    for world in (1, 2, 3, 4, 8):
        rs = balanced_ranges(w, world)
        assert rs[0][0] == 0 and rs[-1][1] == len(w)
        assert all(a[1] == b[0] for a, b in zip(rs, rs[1:]))
    assert balanced_ranges(w, 2) == [(0, 6), (6, 8)]
    assert balanced_ranges(np.zeros(0), 3) == [(0, 0)] * 3


def test_unit_cuts_of_the_sharded_stage_a():
    """light units over the ranks: contiguous, complete, equal rater steps; the heavy rows count against rank 0."""
    import torch
    from xmap.engine.sharded import unit_cuts
    steps = torch.tensor([3.0, 1, 1, 1, 2, 2, 2, 4, 4, 4, 1, 1, 1, 1, 4], dtype=torch.float64)
    c = torch.cumsum(steps, 0)
    for world in (1, 2, 3, 4, 8):
        for wh in (None, torch.tensor(0.0, dtype=torch.float64), torch.tensor(8.0, dtype=torch.float64), torch.tensor(1e9, dtype=torch.float64)):
            cuts = unit_cuts(c, wh, world)
            assert len(cuts) == world + 1 and cuts[0] == 0 and cuts[-1] == len(steps)
            assert all(a <= b for a, b in zip(cuts, cuts[1:]))
    assert unit_cuts(c, None, 1).tolist() == [0, 15]
    even = unit_cuts(c, None, 2).tolist()
    assert abs(float(c[even[1] - 1]) - 16.0) <= 4.0                      # half of the 32 steps, to within one unit
    shifted = unit_cuts(c, torch.tensor(8.0, dtype=torch.float64), 2).tolist()
    assert shifted[1] < even[1]                                          # rank 0 gives light units away
    assert unit_cuts(c, torch.tensor(1e9, dtype=torch.float64), 2).tolist() == [0, 0, 15]   # heavier than everything
    assert unit_cuts(torch.zeros(0, dtype=torch.float64), None, 3).tolist() == [0, 0, 0, 0]


def test_draw_picks_matches_sequential_reference_draws():
    from xmap.engine import hipabi  # noqa: F401  (device module needs the library)
    from xmap.engine.device import draw_picks
    n_top = np.array([0, 4, 2, 3, 0, 4, 4, 2])
    np.random.seed(11)
    want = [np.random.randint(0, int(m) - 1) if m else 0 for m in n_top]   # generator.py:110, one call per start
    np.random.seed(11)
    assert draw_picks(n_top).tolist() == want
    with pytest.raises(ValueError):
        draw_picks(np.array([3, 1]))     # singleton candidate list: randint(0, 0) raises in the reference


def test_synth_is_deterministic_and_lexicographic():
    from xmap.engine import synth
    a, b = synth.make_two_domain(3, 500, 80, 90), synth.make_two_domain(3, 500, 80, 90)
    assert np.array_equal(a.item, b.item) and np.array_equal(a.rating, b.rating) and np.array_equal(a.time, b.time)
    iids = a.item_ids()
    assert iids == sorted(iids) and len(set(iids)) == a.n_items
    d = np.diff(a.user_ptr)
    assert d.min() >= 1 and set(np.unique(a.rating)) <= {1., 2., 3., 4., 5.}
    for u in range(0, 500, 50):
        prof = a.item[a.user_ptr[u]:a.user_ptr[u + 1]]
        assert len(set(prof.tolist())) == len(prof)


# ---- bench.py --gpus N outside a launcher: the parent starts the ranks itself (reference README.md:53-66: one command)
def _bench_module():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("xmap_bench", os.path.join(root, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)          # importing must not redirect stdout or touch a GPU: both happen in main()
    return mod


def test_bench_launcher_argv_and_when_it_applies():
    b = _bench_module()
    argv = b.launcher_argv(8, 29999, ["--gpus", "8", "--steps", "20", "--warmup", "5"])
    assert argv[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert argv[argv.index("--nproc-per-node") + 1] == "8"
    assert argv[argv.index("--master-addr") + 1] == "127.0.0.1" and argv[argv.index("--master-port") + 1] == "29999"
    assert argv[-7].endswith("bench.py") and argv[-6:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    assert b.needs_self_launch(8, {}) and b.needs_self_launch(2, {"HOME": "/"})
    assert not b.needs_self_launch(1, {})
    assert not b.needs_self_launch(8, {"WORLD_SIZE": "8", "RANK": "3"})      # under torch.distributed.run: this IS a rank
    assert not b.needs_self_launch(8, {"XMAP_FORCE_DIST": "1"})               # one-rank rehearsal of the collectives


def test_bench_parent_starts_the_ranks_without_touching_the_gpu(monkeypatch):
    """the parent of a self-started run: builds the launcher command, relays rank 0's line, returns the child's exit code --
    and has not initialised CUDA when it starts the ranks (a process that has must not; the ranks own the devices)"""
    import io
    import subprocess
    import torch
    b = _bench_module()
    seen = {}

    class FakeChild(object):
        def __init__(self, cmd, stdout=None, env=None, text=None):
            seen["cmd"], seen["env"], seen["cuda"] = cmd, env, torch.cuda.is_initialized()
            self.stdout = io.StringIO('NCCL version banner\n{"metric": "item_sim_pairs_per_s", "n_gpus": 2}\n')

        def wait(self):
            return seen.get("rc", 0)

    monkeypatch.setattr(subprocess, "Popen", FakeChild)
    monkeypatch.setenv("XMAP_DIST_BACKEND", "gloo")      # (nccl refuses up front when fewer devices than ranks are visible)
    out = io.StringIO()
    monkeypatch.setattr(b, "_REAL_STDOUT", out)
    assert b.self_launch(2, ["--gpus", "2", "--workload", "c1"]) == 0
    assert seen["cuda"] is False and not torch.cuda.is_initialized()
    assert seen["cmd"][:3] == [sys.executable, "-m", "torch.distributed.run"] and seen["cmd"][-4:] == ["--gpus", "2", "--workload", "c1"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert out.getvalue().strip() == '{"metric": "item_sim_pairs_per_s", "n_gpus": 2}'     # the banner went to the log
    seen["rc"] = 7                                        # a failing rank: non-zero exit, no line
    out.seek(0); out.truncate()
    assert b.self_launch(2, ["--gpus", "2"]) == 7 and out.getvalue() == ""
    monkeypatch.setenv("XMAP_DIST_BACKEND", "nccl")
    if torch.cuda.device_count() < 2:
        assert b.self_launch(2, ["--gpus", "2"]) == 2    # RCCL needs one device per rank
