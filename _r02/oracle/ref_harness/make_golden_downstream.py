"""Golden vectors for the stages around the hot path: BaselinerClean, RecommenderSim,
RecommenderPrivacy, RecommenderPrediction -- captured by IMPORTING the reference's modules (build container only)
and driving them through the list-backed RDD stand-in.  TEST INFRASTRUCTURE ONLY; writes data only
(tests/golden/*_downstream.json.gz).

Usage: TZ=UTC PYTHONHASHSEED=0 python oracle/ref_harness/make_golden_downstream.py
"""
import gzip
import json
import os
import sys
import time

os.environ["TZ"] = "UTC"
time.tzset()
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference/code")

import numpy as np  # noqa: E402
from minirdd import MiniRDD, MiniSC, install_pyspark_stub  # noqa: E402

install_pyspark_stub()
from xmap.core.baselinerClean import BaselinerClean  # noqa: E402  (reference)
from xmap.core.recommenderSim import RecommenderSim  # noqa: E402  (reference)
from xmap.core.recommenderPrivacy import RecommenderPrivacy  # noqa: E402  (reference)
from xmap.core.recommenderPrediction import RecommenderPrediction  # noqa: E402  (reference)
from xmap.utils import assist  # noqa: E402  (reference)
import make_golden as mg  # noqa: E402  (build-owned harness: cases, ts2dt)

OUT = os.path.join(REPO, "tests", "golden")


def raw_lines(seed=12):
    """Amazon-format text lines with duplicates, out-of-period timestamps and sparse users."""
    r = mg.synth.make_two_domain(seed, 120, 40, 40, overlap=0.4)
    rng = np.random.default_rng(seed)
    lines = []
    for u in range(r.n_users):
        for e in range(r.user_ptr[u], r.user_ptr[u + 1]):
            it = int(r.item[e])
            if it >= r.n_src_items:
                continue                      # one domain only (the clean stage runs per domain)
            ts = int(r.time[e])
            if rng.random() < 0.1:
                ts -= 3 * 365 * 86400         # outside 2012-2013
            lines.append("A%05d\t%010d\t%.1f\t%d" % (u, it, float(r.rating[e]), ts))
            if rng.random() < 0.15:           # the same (user, item) rated again, earlier or later
                lines.append("A%05d  %010d %.1f %d" % (u, it, float(rng.integers(1, 6)), ts + int(rng.integers(-5, 6)) * 86400))
    return lines


def main():
    sc = MiniSC()
    out = {}
    # ---- BaselinerClean
    lines = raw_lines()
    tool = BaselinerClean(5, 30, 2012, 2013, "S:")
    parsed = tool.parse_data(MiniRDD(lines, sc))
    rdd = MiniRDD(parsed.collect(), sc)
    # the reference groups with aggregateByKey; the stand-in offers combineByKey with the same (ordered) semantics
    grouped = rdd.combineByKey(lambda v: [v], lambda a, v: a + [v], lambda a, b: a + b)
    filtered = grouped.mapPartitions(tool.remove_invalid)
    cleaned = tool.clean_data(filtered).collect()
    out["clean"] = dict(
        lines=lines, params=[5, 30, 2012, 2013, "S:"],
        cleaned=[(u, [(i, r, mg.dt2ts(t)) for (i, r, t) in prof]) for u, prof in cleaned],
        partial=[u for u, _ in tool.take_partial_data(MiniRDD(cleaned, sc))])
    # ---- recommender stages on the AlterEgo profile of the 'small' case (private mapping, cosine, k=5)
    g = np.load(os.path.join(OUT, "small.npz"))
    uids = [str(s) for s in g["uids"]]
    iids = [str(s) for s in g["iids"]]
    tag = "cosine.k5.priv"
    rows = [(uids[u], iids[i], float(r), mg.ts2dt(t)) for (u, i), r, t in
            zip(g[tag + ".ae_head"], g[tag + ".ae_rating"], g[tag + ".ae_time"])]
    # a test set: for every 3rd user that has target ratings, its own target ratings
    per_user = {}
    for (u, i, r, t) in rows[: int((g[tag + ".ae_head"][:, 0] >= 0).sum())]:
        per_user.setdefault(u, []).append((i, r, t))
    test = [(u, prof[:4]) for k, (u, prof) in enumerate(sorted(per_user.items())) if k % 3 == 0 and "T:" in prof[0][0]]
    out["downstream_input"] = dict(
        rows=[(u, i, r, mg.dt2ts(t)) for (u, i, r, t) in rows],
        test=[(u, [(i, r, mg.dt2ts(t)) for (i, r, t) in prof]) for u, prof in test])
    for method in ("cosine_item", "adjust_cosine_item"):
        sim_tool = RecommenderSim(method, 50)
        res = assist.recommender_calculate_sim_pipeline(sc, sim_tool, MiniRDD(rows, sc))
        user_based, item_based, ubd, ibd, uinfo, iinfo, sim = res
        pairs = sim.collect()
        key = method
        out[key] = dict(
            item_info=[(i, [float(x) for x in v]) for i, v in sorted(iinfo.value.items())],
            user_info=[(u, [float(x) for x in v]) for u, v in sorted(uinfo.value.items())],
            sim=[([a, b], [float(v[0]), float(v[1])]) for (a, b), v in pairs])
        for private, seed in ((False, 0), (True, 3)):
            pol = RecommenderPrivacy(10, 0.6, 0.1)
            np.random.seed(seed)
            sel = assist.recommender_privacy_pipeline(pol, MiniRDD(pairs, sc), private).collect()
            sel = [(i, [(n, float(v)) for n, v in lst]) for i, lst in sel]
            pred_tool = RecommenderPrediction(0.03, method)
            test_rdd = MiniRDD([(u, [(i, r, mg.ts2dt(t)) for (i, r, t) in prof])
                                for u, prof in out["downstream_input"]["test"]], sc)
            pred = pred_tool.item_based_recommendation(test_rdd, ibd, sc.broadcast(dict(sel)), iinfo).collect()
            import io
            import contextlib
            with contextlib.redirect_stdout(io.StringIO()):
                mae = assist.recommender_prediction_pipeline(
                    pred_tool, sim_tool, test_rdd, sc.broadcast(dict(sel)), ubd, ibd, uinfo, iinfo)
            out[key]["private" if private else "nonprivate"] = dict(
                seed=seed, selected=sel,
                predicted=[(u, [list(p) for p in lst]) for u, lst in pred], mae=mae)
    # ---- RecommenderSim on non-integer ratings (AlterEgo ratings are means): pins the float behaviour of the
    # similarity and of the leave-one-out local sensitivity (items with a single rater included)
    frac = (0.0, 1.0 / 3.0, 0.5, 0.25, 2.0 / 3.0)
    rows_f = [(u, i, float(np.float32(float(r) - frac[k % 5])) if float(r) > 1 else float(r), t)   # fp32 values: the
              for k, (u, i, r, t) in enumerate(rows)]                                                  # engine's rating type
    sim_tool = RecommenderSim("cosine_item", 50)
    res = assist.recommender_calculate_sim_pipeline(sc, sim_tool, MiniRDD(rows_f, sc))
    out["cosine_item_float"] = dict(
        rows=[(u, i, r, mg.dt2ts(t)) for (u, i, r, t) in rows_f],
        item_info=[(i, [float(x) for x in v]) for i, v in sorted(res[5].value.items())],
        sim=[([a, b], [float(v[0]), float(v[1])]) for (a, b), v in res[6].collect()])
    path = os.path.join(OUT, "small_downstream.json.gz")
    with gzip.open(path, "wt") as f:
        json.dump(out, f)
    print("wrote", path, os.path.getsize(path) // 1024, "KB;", {k: (v.get("nonprivate", {}).get("mae"), v.get("private", {}).get("mae")) for k, v in out.items() if "item" in k})


if __name__ == "__main__":
    main()
