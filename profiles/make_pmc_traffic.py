"""HBM bytes per launch from two rocprofv3 PMC passes -> profiles/pmc_traffic.json (read by bench.py for roofline.traffic).

    python profiles/make_pmc_traffic.py <FETCH_SIZE counter_collection.csv> <WRITE_SIZE counter_collection.csv> <tag> \
           [<dense FETCH_SIZE csv> <dense WRITE_SIZE csv>]      (passes of `bench.py --workload dense`)

The passes are `rocprofv3 --pmc FETCH_SIZE ...` and `rocprofv3 --pmc WRITE_SIZE ...` of `python3 bench.py --steps 1
--warmup 0 --no-cpu` (separate runs, one row per dispatch).  Counter values are KiB.  gfx950 correction
(MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 64 B per 128-B request, so bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024.
Kernels launched several times per step (template instances of one logical pass) are summed per step.
"""
import collections
import csv
import json
import os
import re
import sys


def per_kernel(path, counter):
    acc = collections.defaultdict(float)
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] != counter:
            continue
        name = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "").replace("xmap::", "").replace("ts::", "").strip()
        acc[name] += float(row["Counter_Value"]) * 1024.0
    return acc


def main(fetch_csv, write_csv, tag, dense_fetch=None, dense_write=None):
    f, w = per_kernel(fetch_csv, "FETCH_SIZE"), per_kernel(write_csv, "WRITE_SIZE")
    if dense_fetch:
        f.update(per_kernel(dense_fetch, "FETCH_SIZE"))
        w.update(per_kernel(dense_write, "WRITE_SIZE"))
    raw = {k: {"FETCH_SIZE_bytes": f.get(k, 0.0), "WRITE_SIZE_bytes": w.get(k, 0.0)} for k in sorted(set(f) | set(w))
           if k.startswith("k_")}
    tot = lambda pred: sum(2.0 * v["FETCH_SIZE_bytes"] + v["WRITE_SIZE_bytes"] for k, v in raw.items() if pred(k))
    out = {"_comment": "HBM bytes per step at BASELINE configs[1] from rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate "
                       "passes, profiles/%s_c2_pmc_*.csv); bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 correction); "
                       "k_pair_tri = sum over its table-class launches (one logical pass)" % tag,
           "k_pair_tri": tot(lambda k: k.startswith("k_pair_tri")),
           "k_paths4": tot(lambda k: k.startswith("k_paths4")),
           # lower reading for k_paths4: FETCH_SIZE taken as is (exact for 32- and 64-byte random reads, measured with
           # profiles/rand_rmw_grp.hip under --pmc: README, "FETCH_SIZE correction")
           "k_paths4_fetch_x1": sum(v["FETCH_SIZE_bytes"] + v["WRITE_SIZE_bytes"] for k, v in raw.items() if k.startswith("k_paths4")), "k_paths2": tot(lambda k: k == "k_paths2"), "k_scatter": tot(lambda k: k == "k_scatter"),
           "k_knn_classify": tot(lambda k: k == "k_knn_classify"), "k_sort_profiles": tot(lambda k: k == "k_sort_profiles"),
           "k_csc_fill": tot(lambda k: k == "k_csc_fill"),
           # round 3: the mirror = level A over the COO (+ own halves), level B, tiles, slices of the large keys; the
           # transposition of the ratings = the two levels over the sort records, tiles, slices
           "mirror": tot(lambda k: k.startswith("k_ts_bin<3") or k.startswith("k_mir_")),
           "rater_records": tot(lambda k: k.startswith("k_ts_bin<2") or k.startswith("k_rc_")),
           "k_count3": tot(lambda k: k == "k_count3"), "k_sort_profiles3": tot(lambda k: k.startswith("k_sort_profiles3")),
           "k_item_stats3": tot(lambda k: k.startswith("k_item_")), "k_dense_topk": tot(lambda k: k.startswith("k_dense_topk")) or None,
           "raw": raw}
    here = os.path.dirname(os.path.abspath(__file__))
    old = {}
    try:
        old = json.load(open(os.path.join(here, "pmc_traffic.json")))
    except Exception:
        pass
    if out["k_dense_topk"] is None and old.get("k_dense_topk"):
        out["k_dense_topk"] = old["k_dense_topk"]
    json.dump(out, open(os.path.join(here, "pmc_traffic.json"), "w"), indent=1)
    print({k: ("%.3e" % v if isinstance(v, float) else v) for k, v in out.items() if k not in ("raw", "_comment")})


if __name__ == "__main__":
    main(*sys.argv[1:6])
