"""End-to-end two-domain run on the MI355X engine: clean -> split -> [item-item sim -> X-Sim extension -> AlterEgo
generation] (GPU) -> recommender sim / privacy / prediction -> MAE.

This is the build's own driver over the drop-in API (the same calls, in the same order, as the reference's
code/twodomain_demo.py:31-140, which runs unmodified against x-map_amd/ when its hard-coded
/home/tlin/notebooks paths exist -- see INTEGRATION.md).  Data: synthetic Amazon-format text files written to a
work directory (the reference ships none).

    python examples/run_twodomain.py [--users 3000] [--items 600] [--workdir /tmp/xmap_demo] [--private]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "x-map_amd"))

import yaml  # noqa: E402
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


def write_inputs(workdir, users, items, seed):
    """book.txt (source) / movie.txt (target) in `uid \\t iid \\t rating \\t unix_ts` + parameters.yaml"""
    raw = os.path.join(workdir, "data", "raw")
    os.makedirs(raw, exist_ok=True)
    r = synth.make_two_domain(seed, users, items, items, overlap=0.35)
    with open(os.path.join(raw, "book.txt"), "w") as fb, open(os.path.join(raw, "movie.txt"), "w") as fm:
        for u in range(r.n_users):
            for e in range(r.user_ptr[u], r.user_ptr[u + 1]):
                it = int(r.item[e])
                src = it < r.n_src_items
                rid = "%010d" % r.src_numbers[it] if src else "B0%08d" % r.tgt_numbers[it - r.n_src_items]
                (fb if src else fm).write("A%013d\t%s\t%.1f\t%d\n" % (u, rid, float(r.rating[e]), int(r.time[e])))
    para = {
        "init": {"path_hdfs": "file:" + os.path.join(workdir, "data"), "path_movie": "raw/movie.txt",
                 "path_book": "raw/book.txt", "is_debug": False, "seed": 666666, "num_partition": 30},
        "baseliner": {"num_atleast_rating": 5, "size_subset": 6666, "date_from": 2012, "date_to": 2013, "num_left": 0,
                      "ratio_split": 0.2, "ratio_both": 0.8, "calculate_baseline_sim_method": "adjust_cosine",
                      "calculate_baseline_weighting": 50},
        "extender": {"extend_among_topk": 10},
        "generator": {"private_flag": False, "mapping_range": 1, "private_epsilon": 0.6, "private_rpo": 0.1},
        "recommender": {"calculate_xmap_sim_method": "cosine_item", "calculate_xmap_weighting": 50, "mapping_range": 10,
                        "private_flag": False, "private_epsilon": 0.6, "private_rpo": 0.1, "decay_alpha": 0.03},
    }
    path = os.path.join(workdir, "parameters.yaml")
    with open(path, "w") as f:
        yaml.safe_dump(para, f)
    return path


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--users", type=int, default=3000)
    ap.add_argument("--items", type=int, default=600)
    ap.add_argument("--seed", type=int, default=31)
    ap.add_argument("--workdir", default="/tmp/xmap_demo")
    ap.add_argument("--private", action="store_true")
    args = ap.parse_args(argv)
    para = assist.load_parameter(write_inputs(args.workdir, args.users, args.items, args.seed))
    if args.private:
        para["generator"]["private_flag"] = True
        para["recommender"]["private_flag"] = True
    sc = SparkContext(conf=SparkConf().setAppName("xmap two-domain on MI355X"))
    sqlContext = SQLContext(sc)
    b, g, rc = para["baseliner"], para["generator"], para["recommender"]
    t = {}

    def timed(name, f, *a):
        t0 = time.time()
        out = f(*a)
        t[name] = time.time() - t0
        return out
    clean_s = BaselinerClean(b["num_atleast_rating"], b["size_subset"], b["date_from"], b["date_to"], domain_label="S:")
    clean_t = BaselinerClean(b["num_atleast_rating"], b["size_subset"], b["date_from"], b["date_to"], domain_label="T:")
    split = BaselinerSplit(b["num_left"], b["ratio_split"], b["ratio_both"], para["init"]["seed"])
    sim_tool = BaselinerSim(b["calculate_baseline_sim_method"], b["calculate_baseline_weighting"])
    hdfs = para["init"]["path_hdfs"]
    sourceRDD = timed("clean_source", assist.baseliner_clean_data_pipeline, sc, clean_s,
                      os.path.join(hdfs, para["init"]["path_book"]), para["init"]["is_debug"], para["init"]["num_partition"])
    targetRDD = timed("clean_target", assist.baseliner_clean_data_pipeline, sc, clean_t,
                      os.path.join(hdfs, para["init"]["path_movie"]), para["init"]["is_debug"], para["init"]["num_partition"])
    trainRDD, testRDD = timed("split", assist.baseliner_split_data_pipeline, sc, split, sourceRDD, targetRDD)
    item2item_simRDD = timed("A_item_sim", assist.baseliner_calculate_sim_pipeline, sc, sim_tool, trainRDD)
    ext_tool = ExtendSim(para["extender"]["extend_among_topk"])
    extendedsimRDD = timed("B_extend", assist.extender_pipeline, sc, sqlContext, sim_tool, ext_tool, item2item_simRDD)
    gen_tool = Generator(g["mapping_range"], g["private_epsilon"], b["calculate_baseline_sim_method"], g["private_rpo"])
    alterEgo_profile = timed("C_generate", assist.generator_pipeline, gen_tool, trainRDD, extendedsimRDD, g["private_flag"])
    rsim = RecommenderSim(rc["calculate_xmap_sim_method"], rc["calculate_xmap_weighting"])
    rpriv = RecommenderPrivacy(rc["mapping_range"], rc["private_epsilon"], rc["private_rpo"])
    rpred = RecommenderPrediction(rc["decay_alpha"], rc["calculate_xmap_sim_method"])
    _, _, ubd, ibd, uinfo, iinfo, alterEgo_sim = timed(
        "recommender_sim", assist.recommender_calculate_sim_pipeline, sc, rsim, alterEgo_profile)
    kept = timed("recommender_privacy", assist.recommender_privacy_pipeline, rpriv, alterEgo_sim, rc["private_flag"])
    simpair_bd = sc.broadcast(kept.collectAsMap())
    mae = timed("recommender_prediction", assist.recommender_prediction_pipeline, rpred, rsim, testRDD, simpair_bd,
                ubd, ibd, uinfo, iinfo)
    assist.write_to_disk({"mae": mae}, para, os.path.join(args.workdir, "data", "output"))
    sc.stop()
    print("train users %d, test users %d, sim pairs %d, AlterEgo rows %d" % (
        trainRDD.count(), testRDD.count(), item2item_simRDD.count(), alterEgo_profile.count()))
    print("MAE (no decay; decay):", mae)
    print("seconds:", {k: round(v, 3) for k, v in t.items()})
    return mae


if __name__ == "__main__":
    main()
