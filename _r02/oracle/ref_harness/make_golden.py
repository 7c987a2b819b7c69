"""Generate golden vectors for the hot path by IMPORTING the reference's own
modules (read-only, from /root/reference/code) and driving them through the
build-owned list-backed RDD stand-in (minirdd.py).

TEST INFRASTRUCTURE ONLY; runs in the build container only (the reference does
not travel to the GPU box).  Only data (inputs + expected outputs, in index
space plus the id tables) is written to tests/golden/*.npz; no reference source
or bytecode is copied (sys.dont_write_bytecode is set).

Canonical order: one partition; each stage is fed the
canonically sorted output of the previous stage:
  * item2item_simRDD sorted by (iid1, iid2)          -> top-k tie-break = ascending iid2
  * extended_simRDD: records sorted by start id, each candidate list sorted by end id
  * np.random.seed(seed) right before the non-private generator call.

Usage:  PYTHONHASHSEED=0 python oracle/ref_harness/make_golden.py [case ...]
"""
import os
import sys
import time
import datetime

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference/code")

import numpy as np  # noqa: E402
from minirdd import MiniRDD, MiniSC, MiniSQL, install_pyspark_stub  # noqa: E402

install_pyspark_stub()
from xmap.core.baselinerSim import BaselinerSim  # noqa: E402  (reference)
from xmap.core.extender import ExtendSim  # noqa: E402  (reference)
from xmap.core.generator import Generator  # noqa: E402  (reference)
from xmap.utils import assist  # noqa: E402  (reference)

import importlib.util  # noqa: E402
_spec = importlib.util.spec_from_file_location(
    "xmap_synth", os.path.join(REPO, "x-map_amd", "xmap", "engine", "synth.py"))
synth = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(synth)

OUT = os.path.join(REPO, "tests", "golden")
CAP = 50            # parameters.yaml:18 calculate_baseline_weighting
EPS = 0.6           # parameters.yaml:24
RPO = 0.1           # parameters.yaml:25
MAPPING_RANGE = 1   # parameters.yaml:23


def ts2dt(ts):
    return datetime.datetime.utcfromtimestamp(int(ts))


def dt2ts(dt):
    return int((dt - datetime.datetime(1970, 1, 1)).total_seconds())


# ---------------------------------------------------------------------------
# cases: name -> train records [(uid, [(iid, rating, unix_ts)*])*]
# ---------------------------------------------------------------------------
def case_kat7():
    """The 7-user worked example of SURVEY.md Appendix A.6."""
    d0 = 1325376000

    def t(d):
        return d0 + 86400 * d
    return [
        ("u1", [("00aS:", 5.0, t(1)), ("00bS:", 3.0, t(2)), ("B0xT:", 4.0, t(3)), ("B0yT:", 2.0, t(4))]),
        ("u2", [("00aS:", 4.0, t(1)), ("B0xT:", 5.0, t(2))]),
        ("u3", [("00aS:", 2.0, t(1)), ("00bS:", 5.0, t(2)), ("00cS:", 4.0, t(5))]),
        ("u4", [("00bS:", 1.0, t(1)), ("00cS:", 2.0, t(2))]),
        ("u5", [("B0xT:", 3.0, t(1)), ("B0yT:", 4.0, t(2)), ("B0zT:", 5.0, t(3))]),
        ("u6", [("B0yT:", 1.0, t(1)), ("B0zT:", 2.0, t(2))]),
        ("u7", [("00cS:", 5.0, t(1))]),
    ]


def case_synth(seed, users, isrc, itgt, **kw):
    return synth.make_two_domain(seed, users, isrc, itgt, **kw).train_records()


def case_mixed_prefix():
    """Adversarial ids: the 2-char prefix does NOT coincide with the domain
    (pins quirk A.5-1: label uses iid[:2]); also a raw id containing 'T:'."""
    recs = case_synth(11, 160, 60, 60, overlap=0.4)
    def remap(iid):
        n = int(iid[2:10]) if iid.startswith("B0") else int(iid[:10])
        if iid.endswith("S:"):
            pre = ("00", "B0", "1x")[n % 3]
            return "%s%05dS:" % (pre, n)
        pre = ("B0", "00", "T:")[n % 3]
        return "%s%05dT:" % (pre, n)
    return [(u, [(remap(i), r, t) for (i, r, t) in prof]) for (u, prof) in recs]


def case_multilabel():
    """Labels of the multi-domain demo (multidomain_demo.py:62-72): one source domain "S:1:" against "T:".
    The knn classification uses iid[-2:] = "1:" / "T:" as substring tests (extender.py:29-35)."""
    recs = case_synth(13, 300, 100, 100, overlap=0.35)
    return [(u, [((i[:-2] + "S:1:") if i.endswith("S:") else i, r, t) for (i, r, t) in prof]) for (u, prof) in recs]


CASES = {
    "kat7": (case_kat7, dict(ks=[2], seeds=[7])),
    "tiny": (lambda: case_synth(3, 60, 30, 30, overlap=0.5), dict(ks=[2, 5], seeds=[5])),
    "small": (lambda: case_synth(4, 400, 150, 150, overlap=0.3), dict(ks=[2, 5, 10], seeds=[5, 6])),
    "medium": (lambda: case_synth(5, 1500, 400, 400, overlap=0.25), dict(ks=[2, 5], seeds=[5])),
    "mixed": (case_mixed_prefix, dict(ks=[3], seeds=[5])),
    "multilabel": (case_multilabel, dict(ks=[4], seeds=[5])),
}


# ---------------------------------------------------------------------------
def run_case(name, train, ks, seeds):
    sc = MiniSC()
    sql = MiniSQL(sc)
    uids = [u for u, _ in train]
    iids = sorted({i for _, prof in train for (i, _, _) in prof})
    iidx = {s: k for k, s in enumerate(iids)}
    uidx = {s: k for k, s in enumerate(uids)}
    out = {}
    out["uids"] = np.array(uids)
    out["iids"] = np.array(iids)
    ptr = np.zeros(len(train) + 1, np.int64)
    items, ratings, times = [], [], []
    for k, (_, prof) in enumerate(train):
        ptr[k + 1] = ptr[k] + len(prof)
        for (i, r, t) in prof:
            items.append(iidx[i]); ratings.append(r); times.append(t)
    out["train_ptr"] = ptr
    out["train_item"] = np.array(items, np.int32)
    out["train_rating"] = np.array(ratings, np.float64)
    out["train_time"] = np.array(times, np.int64)

    def fresh_train():
        return MiniRDD([(u, [(i, r, ts2dt(t)) for (i, r, t) in prof])
                        for (u, prof) in train], sc)

    timing = {}
    for method in ("cosine", "adjust_cosine"):
        tool = BaselinerSim(method, CAP)
        t0 = time.time()
        sim_rdd = assist.baseliner_calculate_sim_pipeline(sc, tool, fresh_train())
        sim = sorted(sim_rdd.collect(), key=lambda kv: kv[0])
        timing["%s.sim" % method] = time.time() - t0
        # stats (A2/A3)
        uinfo = tool.get_universal_user_info(fresh_train()).collectAsMap()
        iinfo = tool.get_universal_item_info(
            fresh_train(), sc.broadcast(uinfo)).collectAsMap()
        out["%s.user_info" % method] = np.array([uinfo[u] for u in uids], np.float64)
        out["%s.item_info" % method] = np.array([iinfo[i] for i in iids], np.float64)
        out["%s.sim_i" % method] = np.array([iidx[k[0]] for k, _ in sim], np.int32)
        out["%s.sim_j" % method] = np.array([iidx[k[1]] for k, _ in sim], np.int32)
        out["%s.sim_val" % method] = np.array([[v[0], v[1], v[2]] for _, v in sim], np.float64).reshape(-1, 3)
        out["%s.sim_label" % method] = np.array([v[3] for _, v in sim], np.int8)

        for k in ks:
            ext_tool = ExtendSim(k)
            tag = "%s.k%d" % (method, k)
            t0 = time.time()
            # -- intermediate: BB list + classified knn lists (B1-B3)
            df = tool.build_sim_DF(MiniRDD(sim, sc))
            df.registerTempTable("sim_table")
            bb = sql.sql(MiniSQL.QUERY).map(lambda l: l.id1).collect()
            out[tag + ".bb"] = np.array(sorted(iidx[b] for b in bb), np.int32)
            item_sim = tool.get_item_sim(MiniRDD(sim, sc))
            cls = ext_tool.find_knn_items(item_sim, sc.broadcast(bb)).collect()
            rows = []
            for iid, bbinfo, nbinfo in cls:
                lists = ((0, bbinfo[0]), (1, bbinfo[1])) if bbinfo is not None \
                    else ((2, nbinfo[0]), (3, nbinfo[1]))
                for lid, lst in lists:
                    for pos, (nbr, s, m, f) in enumerate(lst):
                        rows.append((iidx[iid], lid, pos, iidx[nbr], s, m, f))
            out[tag + ".knn_head"] = np.array([r[:4] for r in rows], np.int32).reshape(-1, 4)
            out[tag + ".knn_val"] = np.array([r[4:] for r in rows], np.float64).reshape(-1, 3)
            out[tag + ".knn_items"] = np.array(
                [(iidx[iid], 1 if b is not None else 2) for iid, b, n in cls], np.int32).reshape(-1, 2)
            # -- the pipeline proper (B0)
            ext = assist.extender_pipeline(sc, sql, tool, ext_tool, MiniRDD(sim, sc)).collect()
            timing[tag + ".ext"] = time.time() - t0
            ext = sorted([(s, sorted(lst, key=lambda p: p[0])) for s, lst in ext],
                         key=lambda kv: kv[0])
            xs = [(iidx[s], iidx[e], float(v)) for s, lst in ext for (e, v) in lst]
            out[tag + ".xsim_head"] = np.array([x[:2] for x in xs], np.int32).reshape(-1, 2)
            out[tag + ".xsim_val"] = np.array([x[2] for x in xs], np.float64)

            gen_tool = Generator(MAPPING_RANGE, EPS, method, RPO)
            for private in (True, False):
                for seed in (seeds if not private else [0]):
                    gtag = "%s.%s" % (tag, "priv" if private else "np%d" % seed)
                    np.random.seed(seed)
                    try:
                        if private:
                            mapped = gen_tool.cross_private_mapping(MiniRDD(ext, sc)).collect()
                        else:
                            mapped = gen_tool.cross_nonprivate_mapping(MiniRDD(ext, sc)).collect()
                    except ValueError as e:
                        out[gtag + ".raises"] = np.array([str(e)])
                        continue
                    out[gtag + ".choice"] = np.array(
                        [(iidx[s], iidx[str(c)]) for s, c in mapped], np.int32).reshape(-1, 2)
                    np.random.seed(seed)
                    rows = assist.generator_pipeline(
                        gen_tool, fresh_train(), MiniRDD(ext, sc), private).collect()
                    out[gtag + ".ae_head"] = np.array(
                        [(uidx[u], iidx[str(i)]) for (u, i, r, t) in rows], np.int32).reshape(-1, 2)
                    out[gtag + ".ae_rating"] = np.array([float(r) for (u, i, r, t) in rows], np.float64)
                    out[gtag + ".ae_time"] = np.array([dt2ts(t) for (u, i, r, t) in rows], np.int64)
    for k, v in sorted(timing.items()):
        print("   %-28s %8.2f s" % (k, v))
    return out


def main(argv):
    names = argv[1:] or list(CASES)
    os.makedirs(OUT, exist_ok=True)
    for name in names:
        fn, opt = CASES[name]
        t0 = time.time()
        train = fn()
        print("[%s] users=%d nnz=%d" % (name, len(train), sum(len(p) for _, p in train)))
        out = run_case(name, train, **opt)
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **out)
        print("[%s] wrote %s (%.1f KB) in %.1f s" % (
            name, path, os.path.getsize(path) / 1024.0, time.time() - t0))


if __name__ == "__main__":
    main(sys.argv)
