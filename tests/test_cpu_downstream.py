"""Stages around the hot path (clean, split, recommender sim / privacy / prediction): the
build's host-side implementations against vectors captured from the reference
(tests/golden/small_downstream.json.gz <- oracle/ref_harness/make_golden_downstream.py).  No GPU needed."""
import datetime
import gzip
import json
import os
import time

import numpy as np
import pytest

os.environ["TZ"] = "UTC"
time.tzset()

from pyspark import SparkContext, SparkConf  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "small_downstream.json.gz")


@pytest.fixture(scope="module")
def gold():
    with gzip.open(GOLD, "rt") as f:
        return json.load(f)


def dt(ts):
    return datetime.datetime.utcfromtimestamp(int(ts))


def test_clean_matches_reference(gold, tmp_path):
    from xmap.core.baselinerClean import BaselinerClean
    from xmap.utils.assist import baseliner_clean_data_pipeline
    g = gold["clean"]
    p = tmp_path / "raw.txt"
    p.write_text("\n".join(g["lines"]) + "\n")
    sc = SparkContext(conf=SparkConf())
    tool = BaselinerClean(*g["params"])
    cleaned = baseliner_clean_data_pipeline(sc, tool, "file:" + str(p), False, 30).collect()
    want = [(u, [(i, r, dt(t)) for (i, r, t) in prof]) for u, prof in g["cleaned"]]
    assert cleaned == want
    debug = baseliner_clean_data_pipeline(sc, tool, str(p), True, 30).collect()
    assert [u for u, _ in debug] == g["partial"] and len(debug) <= g["params"][1]


def test_native_feeder_matches_reference(gold, tmp_path, monkeypatch):
    """csrc/feeder.hip (text -> id tables + CSR + predicate arrays in C++) against the reference's own cleaned output: the
    same users in the same order, the same (item, rating, time) entries; then the engine's conventions on top of it (items in
    lexicographic order, the four predicate arrays of xmap.engine.ids); and the pipeline switch XMAP_NATIVE_FEED=1."""
    from xmap.core.baselinerClean import BaselinerClean
    from xmap.engine import feeder, ids
    from xmap.utils.assist import baseliner_clean_data_pipeline
    g = gold["clean"]
    text = "\n".join(g["lines"]) + "\n"
    want = [(u, [(i, r, dt(t)) for (i, r, t) in prof]) for u, prof in g["cleaned"]]
    f = feeder.Feed.from_text(text, g["params"][2], g["params"][3], g["params"][4], g["params"][0])
    assert f.n_lines == len(g["lines"]) and f.records() == want
    ptr, item, rating, when, attrs = f.arrays()
    iids = f.ids(1)
    assert iids == sorted({i for _, prof in want for (i, _, _) in prof}) and f.ids(0) == [u for u, _ in want]
    assert [iids[k] for k in item] == [i for _, prof in want for (i, _, _) in prof]
    for a, b in zip(attrs, ids.item_attrs(iids)):
        assert np.array_equal(a, b)
    # lines with leading / trailing blanks and CRs split like re.split(r"\s+"); short lines are errors like the IndexError
    messy = text.replace("\t", "  \t ").replace("\n", " \r\n")
    assert feeder.Feed.from_text(messy, *g["params"][2:5], g["params"][0]).records() == want
    with pytest.raises(Exception):
        feeder.Feed.from_text("u1 i1 5.0\n", 2012, 2013, "S:", 1)
    # two domains merged: users of the first, then the rest; items stay sorted; predicates of the union
    f2 = feeder.Feed.from_text(text, g["params"][2], g["params"][3], "T:", g["params"][0])
    m = f.merge(f2)
    assert m.n_users == f.n_users and m.n_items == 2 * f.n_items and m.nnz == 2 * f.nnz
    mi = m.ids(1)
    assert mi == sorted(mi)
    for a, b in zip(m.arrays()[4], ids.item_attrs(mi)):
        assert np.array_equal(a, b)
    rec = m.records()
    assert [(i, r, w) for (i, r, w) in rec[0][1][:len(want[0][1])]] == want[0][1]
    # ... and the same feed from both texts in one call (users that reappear in the second text: the grouping by user is a
    # stable counting sort there), also with the lines of the first text shuffled (nothing grouped by user at all)
    m2 = feeder.Feed.from_texts([(text, g["params"][4]), (text, "T:")], g["params"][2], g["params"][3], g["params"][0])
    assert m2.records() == rec and m2.ids(1) == mi
    for a, b in zip(m2.arrays(), m.arrays()):
        assert all(np.array_equal(x, y) for x, y in zip(a, b)) if isinstance(a, tuple) else np.array_equal(a, b)
    rng = np.random.default_rng(0)
    lines = list(g["lines"])
    perm = rng.permutation(len(lines))
    shuffled = "\n".join(lines[k] for k in perm) + "\n"
    tool = BaselinerClean(*g["params"])
    sc0 = SparkContext(conf=SparkConf())
    ref = tool.clean_data(tool.filter_data(tool.parse_data(sc0.parallelize([lines[k] for k in perm], 1)))).collect()
    assert feeder.Feed.from_text(shuffled, *g["params"][2:5], g["params"][0]).records() == ref
    # the pipeline switch
    p = tmp_path / "raw.txt"
    p.write_text(text)
    monkeypatch.setenv("XMAP_NATIVE_FEED", "1")
    sc = SparkContext(conf=SparkConf())
    rdd = baseliner_clean_data_pipeline(sc, BaselinerClean(*g["params"]), "file:" + str(p), False, 30)
    assert isinstance(rdd, feeder.FeedRDD) and rdd._items is None
    assert rdd.collect() == want


def test_native_feeder_reads_tokens_like_the_reference():
    """parse_line (baselinerClean.py:46-52) converts the timestamp of every line and float(rating) only of the lines inside
    the period; float() takes '1_0', 'inf', '1e1' and refuses hex floats; remove_invalid keeps one rating per (user, item),
    the strictly later one in place -- also for a profile far longer than any scan per rating could take."""
    from xmap.engine import feeder
    T = 1356998400 + 86400 * 40            # inside 2013 (UTC)
    OUT = 1262304000 + 86400 * 40          # inside 2010
    ok = feeder.Feed.from_text("u1 i1 4.0 %d\nu1 i2 bad %d\nu1 i3 1_0 %d\nu1 i4 1e1 %d\n" % (T, OUT, T, T), 2012, 2013, "S:", 1)
    assert [(i, r) for (i, r, _) in ok.records()[0][1]] == [("i1S:", 4.0), ("i3S:", 10.0), ("i4S:", 10.0)]
    assert ok.n_lines == 4
    for bad in ("u1 i1 bad %d\n" % T, "u1 i1 0x10 %d\n" % T, "u1 i1 1__0 %d\n" % T, "u1 i1 _1 %d\n" % T,
                "u1 i1 nan(1) %d\n" % T, "u1 i1 4.0 0x5\n", "u1 i1 4.0 later\n"):
        with pytest.raises(Exception):
            feeder.Feed.from_text(bad, 2012, 2013, "S:", 1)
    inf = feeder.Feed.from_text("u1 i1 -Infinity %d\n" % T, 2012, 2013, "S:", 1).records()[0][1][0][1]
    assert inf == float("-inf")
    # a crawler account: 40 000 ratings over 5 000 items, every item rated eight times -- the latest (strictly) wins in place
    rng = np.random.default_rng(5)
    it = rng.integers(0, 5000, 40000)
    when = T + rng.integers(0, 86400 * 200, 40000)
    val = rng.integers(1, 6, 40000).astype(float)
    text = "".join("crawler i%d %.1f %d\n" % (a, v, w) for a, v, w in zip(it, val, when)) + "u2 i1 3.0 %d\n" % T
    best, order = {}, []
    for a, v, w in zip(it, val, when):
        k = "i%dS:" % a
        if k not in best:
            order.append(k); best[k] = (v, w)
        elif w > best[k][1]:
            best[k] = (v, w)
    t0 = time.time()
    rec = feeder.Feed.from_text(text, 2012, 2014, "S:", 1).records()
    assert time.time() - t0 < 5.0
    assert rec[0][0] == "crawler" and [(i, r) for (i, r, _) in rec[0][1]] == [(k, best[k][0]) for k in order]
    assert [w for (_, _, w) in rec[0][1]] == [dt(best[k][1]) for k in order]


def test_split_protocol():
    from xmap.core.baselinerSplit import BaselinerSplit
    from xmap.utils.assist import baseliner_split_data_pipeline
    sc = SparkContext(conf=SparkConf())
    src = sc.parallelize([("u%d" % u, [("s%d_%dS:" % (u, k), 4.0, k) for k in range(3)]) for u in range(0, 60)])
    tgt = sc.parallelize([("u%d" % u, [("t%d_%dT:" % (u, k), 3.0, k) for k in range(4)]) for u in range(30, 90)])
    tool = BaselinerSplit(1, 0.2, 0.8, 666666)
    train, test = baseliner_split_data_pipeline(sc, tool, src, tgt)
    train, test = train.collect(), dict(test.collect())
    overlap = {"u%d" % u for u in range(30, 60)}
    assert set(test) <= overlap and 0 < len(test) < len(overlap)
    by_user = {}
    for u, prof in train:
        by_user.setdefault(u, []).extend(prof)
    for u in overlap:
        items = {i for i, _, _ in by_user[u]}
        if u in test:       # num_left target ratings stay, the rest are hidden and form the test profile
            hidden = {i for i, _, _ in test[u]}
            assert len(hidden) == 3 and not (hidden & items)
            assert sum("T:" in i for i in items) == 1 and sum("S:" in i for i in items) == 3
        else:
            assert len(items) == 7
    assert all(len(by_user["u%d" % u]) == 3 for u in range(0, 30)) and all(len(by_user["u%d" % u]) == 4 for u in range(60, 90))


@pytest.mark.parametrize("method", ["cosine_item", "adjust_cosine_item"])
def test_recommender_stages_match_reference(gold, method):
    from xmap.core.recommenderSim import RecommenderSim
    from xmap.core.recommenderPrivacy import RecommenderPrivacy
    from xmap.core.recommenderPrediction import RecommenderPrediction
    from xmap.utils.assist import (recommender_calculate_sim_pipeline, recommender_privacy_pipeline,
                                   recommender_prediction_pipeline)
    sc = SparkContext(conf=SparkConf())
    rows = [(u, i, r, dt(t)) for (u, i, r, t) in gold["downstream_input"]["rows"]]
    test = [(u, [(i, r, dt(t)) for (i, r, t) in prof]) for u, prof in gold["downstream_input"]["test"]]
    g = gold[method]
    sim_tool = RecommenderSim(method, 50)
    sim_tool.calculate_sim = sim_tool.calculate_sim_host     # the per-pair Python statement (the product runs it on the GPU)
    res = recommender_calculate_sim_pipeline(sc, sim_tool, sc.parallelize(rows))
    user_based, item_based, ubd, ibd, uinfo, iinfo, sim = res
    assert sorted((k, [float(x) for x in v]) for k, v in iinfo.value.items()) == [(k, v) for k, v in g["item_info"]]
    assert sorted((k, [float(x) for x in v]) for k, v in uinfo.value.items()) == [(k, v) for k, v in g["user_info"]]
    pairs = sim.collect()
    got = [([a, b], [float(v[0]), float(v[1])]) for (a, b), v in pairs]
    assert len(got) == len(g["sim"])
    for (ka, va), (kb, vb) in zip(got, g["sim"]):
        assert ka == kb
        assert np.array_equal(np.array(va), np.array(vb), equal_nan=True)
    for name, private in (("nonprivate", False), ("private", True)):
        exp = g[name]
        np.random.seed(exp["seed"])
        sel = recommender_privacy_pipeline(RecommenderPrivacy(10, 0.6, 0.1), sc.parallelize(pairs), private).collect()
        sel = [(i, [(n, float(v)) for n, v in lst]) for i, lst in sel]
        assert [(i, [[n, v] for n, v in lst]) for i, lst in sel] == [(i, lst) for i, lst in exp["selected"]]
        pred_tool = RecommenderPrediction(0.03, method)
        test_rdd = sc.parallelize(test)
        pred = pred_tool.item_based_recommendation(test_rdd, ibd, sc.broadcast(dict(sel)), iinfo).collect()
        assert [(u, [list(p) for p in lst]) for u, lst in pred] == [(u, lst) for u, lst in exp["predicted"]]
        mae = recommender_prediction_pipeline(pred_tool, sim_tool, test_rdd, sc.broadcast(dict(sel)), ubd, ibd, uinfo, iinfo)
        assert mae == exp["mae"]
