"""Multi-domain run (BASELINE configs[3] shape): N source domains -> one target.  As in the reference's
code/multidomain_demo.py:101-128, every source domain is an independent two-domain problem against the same target
(item ids labelled "S:1:", "S:2:", ... and "T:"); the AlterEgo profiles are united before the recommender stages.
The reference's own multi-domain driver cannot run as shipped; this is the build's driver over
the drop-in API.  On a multi-GPU node the per-domain problems are independent and can be placed on different GPUs
(domain-parallel replicas).

    python examples/run_multidomain.py [--users 2000] [--items 400] [--sources 2] [--workdir /tmp/xmap_multi]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "x-map_amd"))

from pyspark import SparkContext, SparkConf  # noqa: E402
from pyspark.sql import SQLContext  # noqa: E402

from xmap.core.baselinerClean import BaselinerClean  # noqa: E402
from xmap.core.baselinerSplit import BaselinerSplit  # noqa: E402
from xmap.core.baselinerSim import BaselinerSim  # noqa: E402
from xmap.core.extender import ExtendSim  # noqa: E402
from xmap.core.generator import Generator  # noqa: E402
from xmap.core.recommenderSim import RecommenderSim  # noqa: E402
from xmap.core.recommenderPrivacy import RecommenderPrivacy  # noqa: E402
from xmap.core.recommenderPrediction import RecommenderPrediction  # noqa: E402
from xmap.utils import assist  # noqa: E402
from xmap.engine import synth  # noqa: E402


def write_inputs(workdir, users, items, n_sources, seed):
    """source_<d>.txt for each source domain and target.txt; all users of a source overlap candidates share uids."""
    raw = os.path.join(workdir, "data", "raw")
    os.makedirs(raw, exist_ok=True)
    paths = []
    tgt_written = False
    for d in range(n_sources):
        r = synth.make_two_domain(seed, users, items, items, overlap=0.5) if d == 0 else \
            synth.make_two_domain(seed + d, users, items, items, overlap=0.5)
        p = os.path.join(raw, "source_%d.txt" % (d + 1))
        with open(p, "w") as fs, open(os.path.join(raw, "target.txt"), "a" if tgt_written else "w") as ft:
            for u in range(r.n_users):
                for e in range(r.user_ptr[u], r.user_ptr[u + 1]):
                    it = int(r.item[e])
                    if it < r.n_src_items:
                        fs.write("A%013d\t%02d%08d\t%.1f\t%d\n" % (u, d, r.src_numbers[it], float(r.rating[e]), int(r.time[e])))
                    elif not tgt_written:
                        ft.write("A%013d\tB0%08d\t%.1f\t%d\n" % (u, r.tgt_numbers[it - r.n_src_items], float(r.rating[e]), int(r.time[e])))
        tgt_written = True
        paths.append(p)
    return paths, os.path.join(raw, "target.txt")


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--users", type=int, default=2000)
    ap.add_argument("--items", type=int, default=400)
    ap.add_argument("--sources", type=int, default=2)
    ap.add_argument("--seed", type=int, default=41)
    ap.add_argument("--topk", type=int, default=10)
    ap.add_argument("--workdir", default="/tmp/xmap_multi")
    args = ap.parse_args(argv)
    src_paths, tgt_path = write_inputs(args.workdir, args.users, args.items, args.sources, args.seed)
    sc = SparkContext(conf=SparkConf().setAppName("xmap multi-domain on MI355X"))
    sqlContext = SQLContext(sc)
    t0 = time.time()
    targetRDD = assist.baseliner_clean_data_pipeline(sc, BaselinerClean(5, 6666, 2012, 2013, domain_label="T:"), tgt_path, False, 30)
    sources = [assist.baseliner_clean_data_pipeline(
        sc, BaselinerClean(5, 6666, 2012, 2013, domain_label="S:%d:" % (d + 1)), p, False, 30)
        for d, p in enumerate(src_paths)]
    split = BaselinerSplit(0, 0.2, 0.8, 666666)
    if len(sources) == 2:
        train1, train2, testRDD = assist.baseliner_split_multidomain_data_pipeline(sc, split, sources[0], sources[1], targetRDD)
        trains = [train1, train2]
    else:   # one two-domain split per source, evaluated on the first split's test users
        trains, testRDD = [], None
        for s in sources:
            tr, te = assist.baseliner_split_data_pipeline(sc, split, s, targetRDD)
            trains.append(tr)
            testRDD = testRDD or te
    sim_tool = BaselinerSim("adjust_cosine", 50)
    gen_tool = Generator(1, 0.6, "adjust_cosine", 0.1)
    alterEgo = None
    for d, trainRDD in enumerate(trains):       # independent two-domain problems
        trainRDD = trainRDD.filter(lambda rec: len(rec[1]) > 0).cache()
        sim = assist.baseliner_calculate_sim_pipeline(sc, sim_tool, trainRDD)
        ext = assist.extender_pipeline(sc, sqlContext, sim_tool, ExtendSim(args.topk), sim)
        profile = assist.generator_pipeline(gen_tool, trainRDD, ext, True)
        n_starts = int((ext.E.n_cand > 0).sum().item())     # the lazy handle's candidate counts (ext.count() would build the lists)
        print("source %d: %d sim pairs, %d start items, %d AlterEgo rows" % (d + 1, sim.count(), n_starts, profile.count()))
        alterEgo = profile if alterEgo is None else alterEgo.union(profile)
    rsim = RecommenderSim("cosine_item", 50)
    _, _, ubd, ibd, uinfo, iinfo, alterEgo_sim = assist.recommender_calculate_sim_pipeline(sc, rsim, alterEgo.distinct())
    kept = assist.recommender_privacy_pipeline(RecommenderPrivacy(10, 0.6, 0.1), alterEgo_sim, False)
    mae = assist.recommender_prediction_pipeline(RecommenderPrediction(0.03, "cosine_item"), rsim, testRDD,
                                                 sc.broadcast(kept.collectAsMap()), ubd, ibd, uinfo, iinfo)
    print("MAE (no decay; decay):", mae, " [%.1f s]" % (time.time() - t0))
    return mae


if __name__ == "__main__":
    main()
