"""Drop-in boundary test: the reference's pipeline API (xmap.utils.assist / xmap.core.*) driven exactly the
way twodomain_demo.py:87-105 drives it, on the golden inputs, compared with the reference's own outputs
(string ids, python tuples).  Runs on the GPU box."""
import datetime

import numpy as np
import pytest

from golden_util import METHODS, CAP, Golden

pytestmark = pytest.mark.gpu


def ts2dt(ts):
    return datetime.datetime.utcfromtimestamp(int(ts))


def records(gold):
    out = []
    for u, uid in enumerate(gold.uids):
        a, b = gold.ptr[u], gold.ptr[u + 1]
        out.append((uid, [(gold.iids[gold.item[e]], float(gold.g["train_rating"][e]), ts2dt(gold.time[e]))
                          for e in range(a, b)]))
    return out


@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("case", ["kat7", "small", "mixed", "multilabel"])
def test_three_pipelines(case, method):
    import torch
    assert torch.cuda.is_available()
    from pyspark import SparkContext, SparkConf
    from pyspark.sql import SQLContext
    from xmap.core.baselinerSim import BaselinerSim
    from xmap.core.extender import ExtendSim
    from xmap.core.generator import Generator
    from xmap.utils.assist import baseliner_calculate_sim_pipeline, extender_pipeline, generator_pipeline, map_to_dict
    gold = Golden(case)
    sc = SparkContext(conf=SparkConf().setAppName("parity"))
    sqlContext = SQLContext(sc)
    trainRDD = sc.parallelize(records(gold), 30).cache()
    tool = BaselinerSim(method, CAP)
    item2item_simRDD = baseliner_calculate_sim_pipeline(sc, tool, trainRDD)
    got = dict(item2item_simRDD.collect())
    iids = gold.iids
    exp = {(iids[a], iids[b]): (v, lab) for a, b, v, lab in zip(
        gold[method + ".sim_i"], gold[method + ".sim_j"], gold[method + ".sim_val"], gold[method + ".sim_label"])}
    assert set(got) == set(exp)
    for key, (sim, mutu, frac, label) in got.items():
        v, lab = exp[key]
        assert mutu == v[1] and frac == v[2] and label == lab
        assert sim == pytest.approx(v[0], rel=1e-11, abs=0)
    # downstream consumer shape (recommenderSim.py:20-27): map + reduceByKey on the handle
    per_item = item2item_simRDD.map(lambda kv: (kv[0][0], 1)).reduceByKey(lambda a, b: a + b).collectAsMap()
    assert sum(per_item.values()) == len(exp)
    for k in gold.ks(method):
        tag = "%s.k%d" % (method, k)
        ext_tool = ExtendSim(k)
        ext = extender_pipeline(sc, sqlContext, tool, ext_tool, item2item_simRDD)
        # lazy handle: a Generator consumes the per-start candidate arrays; no (start, end) list exists until the RDD
        # is iterated
        assert not ext.materialised and ext.E.xs_end is None
        if "priv" in gold.gen_tags(method, k):
            generator_pipeline(Generator(1, 0.6, method, 0.1), trainRDD, ext, True).collect()
            assert not ext.materialised
        gx = {s: dict(lst) for s, lst in ext.collect()}
        assert ext.materialised
        # the reference's own orchestration of the stage (assist.py:80-102) against xmap.core.*: classified lists ->
        # extract_siminfo -> sim_extend -> get_final_extension
        from xmap.utils.assist import extract_siminfo
        bridges = sorted({key[0] for key, val in item2item_simRDD.collect() if val[3] == 1})
        classified = ext_tool.find_knn_items(tool.get_item_sim(item2item_simRDD), sc.broadcast(bridges)).cache()
        BB_info, NB_info, knn_BB_bd, knn_NB_bd = extract_siminfo(sc, classified)
        assert set(knn_BB_bd.value) == set(bridges) & {i for i, _ in BB_info.collect()}
        ext_b = ext_tool.get_final_extension(ext_tool.sim_extend(BB_info, NB_info, knn_BB_bd, knn_NB_bd)).cache()
        assert {s: dict(lst) for s, lst in ext_b.collect()} == gx
        xh, xv = gold[tag + ".xsim_head"], gold[tag + ".xsim_val"]
        ex = {}
        for (s, e), v in zip(xh, xv):
            ex.setdefault(iids[s], {})[iids[e]] = v
        assert {s: set(d) for s, d in gx.items()} == {s: set(d) for s, d in ex.items()}
        for s in ex:
            for e in ex[s]:
                assert gx[s][e] == pytest.approx(ex[s][e], rel=1e-9, abs=1e-300)
        # the same stage fed a generic, re-ordered copy of its input (ids and frac taken from the records)
        refed = sc.parallelize(sorted(item2item_simRDD.collect(), key=lambda kv: kv[0], reverse=True))
        gx2 = {s: dict(lst) for s, lst in extender_pipeline(sc, sqlContext, tool, ext_tool, refed).collect()}
        assert gx2 == gx
        gen_tool = Generator(1, 0.6, method, 0.1)
        for gt in gold.gen_tags(method, k):
            gtag = tag + "." + gt
            private = gt == "priv"
            if not private:
                np.random.seed(int(gt[2:]))
            if gold.has(gtag + ".raises"):
                with pytest.raises(ValueError):
                    generator_pipeline(gen_tool, trainRDD, ext, private)
                continue
            rows = generator_pipeline(gen_tool, trainRDD, ext, private).collect()
            eh, er, et = gold[gtag + ".ae_head"], gold[gtag + ".ae_rating"], gold[gtag + ".ae_time"]
            assert len(rows) == len(eh)
            for (u, i, r, t), (eu, ei), rr, tt in zip(rows, eh, er, et):
                assert u == gold.uids[eu] and i == iids[ei] and t == ts2dt(tt)
                assert float(r) == rr            # np.float64 means, bit for bit
            assert all("T:" in i for (_, i, _, _) in rows)
            # Generator's own mapping methods + map_to_dict agree with the golden choices
            if not private:
                np.random.seed(int(gt[2:]))
            mapped = (gen_tool.cross_private_mapping(ext) if private else gen_tool.cross_nonprivate_mapping(ext))
            ch = gold[gtag + ".choice"]
            assert [(s, str(c)) for s, c in mapped.collect()] == [(iids[s], iids[c]) for s, c in ch]
            want = {}
            for s, c in ch:
                want[iids[c]] = iids[s]
            assert map_to_dict(mapped) == want


def test_end_to_end_driver(tmp_path):
    """clean -> split -> GPU hot path -> recommender -> MAE through the drop-in API (examples/run_twodomain.py)."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("run_twodomain", os.path.join(root, "examples", "run_twodomain.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mae = mod.main(["--users", "1500", "--items", "300", "--workdir", str(tmp_path)])
    a, b = [float(x) for x in mae.split(";")]
    assert 0.0 < a < 2.5 and 0.0 < b < 2.5
    assert os.path.isdir(os.path.join(str(tmp_path), "data", "output", "runs"))


def test_multidomain_driver(tmp_path):
    """two source domains ("S:1:", "S:2:") against one target through the drop-in API (examples/run_multidomain.py)."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("run_multidomain", os.path.join(root, "examples", "run_multidomain.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mae = mod.main(["--users", "1200", "--items", "250", "--workdir", str(tmp_path)])
    a, b = [float(x) for x in mae.split(";")]
    assert 0.0 < a < 2.5 and 0.0 < b < 2.5


def test_handles_release_their_buffers_when_dropped():
    """a dropped pipeline handle gives its HBM buffers back at once (no reference cycle inside the handle: at BASELINE
    configs[1] an extension holds 13 GB, which must not wait for the cyclic garbage collector)"""
    import gc
    import torch
    from pyspark import SparkContext, SparkConf
    from pyspark.sql import SQLContext
    from xmap.core.baselinerSim import BaselinerSim
    from xmap.core.extender import ExtendSim
    from xmap.core.generator import Generator
    from xmap.utils.assist import baseliner_calculate_sim_pipeline, extender_pipeline, generator_pipeline
    gold = Golden("small")
    sc = SparkContext(conf=SparkConf().setAppName("release"))
    trainRDD = sc.parallelize(records(gold), 4).cache()
    tool = BaselinerSim("cosine", CAP)
    gc.collect()
    gc.disable()
    try:
        sim = baseliner_calculate_sim_pipeline(sc, tool, trainRDD)
        ext = extender_pipeline(sc, SQLContext(sc), tool, ExtendSim(5), sim)      # (allocates the engine's persistent scratch)
        ae = generator_pipeline(Generator(1, 0.6, "cosine", 0.1), trainRDD, ext, True)
        del ext, ae
        base = torch.cuda.memory_allocated()
        for _ in range(2):
            ext = extender_pipeline(sc, SQLContext(sc), tool, ExtendSim(5), sim)
            ae = generator_pipeline(Generator(1, 0.6, "cosine", 0.1), trainRDD, ext, True)
            assert torch.cuda.memory_allocated() > base
            del ext, ae
            assert torch.cuda.memory_allocated() <= base
    finally:
        gc.enable()
