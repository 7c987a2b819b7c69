"""Pins the CPU oracle (oracle/xmap_oracle.c) against vectors captured from the reference itself
(tests/golden/*.npz <- oracle/ref_harness/make_golden.py importing /root/reference/code/xmap)."""
import numpy as np
import pytest

from golden_util import CASES, METHODS, CAP, Golden, csr_to_pairs
from oracle import xmap_oracle as xo


@pytest.fixture(scope="module", params=CASES)
def gold(request):
    return Golden(request.param)


@pytest.mark.parametrize("method", METHODS)
def test_stage_a(gold, method):
    T = gold.oracle_train()
    uavg, unorm = xo.user_info(T)
    exp_u = gold[method + ".user_info"]
    assert np.array_equal(uavg, exp_u[:, 0])
    assert np.array_equal(unorm, exp_u[:, 1])
    info = xo.item_info(T, uavg)
    exp_i = gold[method + ".item_info"]
    assert np.array_equal(info[:, [0, 1, 3]], exp_i[:, [0, 1, 3]])
    np.testing.assert_allclose(info[:, 2], exp_i[:, 2], rtol=1e-14, atol=0)   # exact-sum canonical value
    S = xo.item_sim(T, method, CAP, uavg, info, nthreads=2)
    rows, cols = csr_to_pairs(S.row_ptr, S.col)
    assert np.array_equal(rows, gold[method + ".sim_i"])
    assert np.array_equal(cols, gold[method + ".sim_j"])
    val = gold[method + ".sim_val"]
    assert np.array_equal(S.mutu.astype(np.float64), val[:, 1])
    frac = S.mutu / (info[rows, 3] + info[cols, 3] - S.nij)
    assert np.array_equal(frac, val[:, 2])
    if method == "cosine":
        assert np.array_equal(S.sim, val[:, 0])   # integer-exact sums: bit-identical to the reference
    else:
        # canonical adjusted dot = exact sum of the reference's fp64 terms; np.sum's pairwise rounding
        # differs by <= 1e-13 relative here (ill-conditioned sums of mixed-sign terms)
        np.testing.assert_allclose(S.sim, val[:, 0], rtol=1e-11, atol=0)
    lab = (T.prefix_cls[rows] != T.prefix_cls[cols]).astype(np.int8)
    assert np.array_equal(lab, gold[method + ".sim_label"])
    xo.sim_free(S)


@pytest.mark.parametrize("method", METHODS)
def test_stage_b_c(gold, method):
    T = gold.oracle_train()
    S = xo.item_sim(T, method, CAP, nthreads=2)
    for k in gold.ks(method):
        tag = "%s.k%d" % (method, k)
        X = xo.extend(T, S, k)
        assert np.array_equal(np.nonzero(X.bb)[0], gold[tag + ".bb"])
        # classified items + knn lists (ids exact, values bit-exact)
        ki = gold[tag + ".knn_items"]
        got_items = np.nonzero(X.cls)[0]
        assert np.array_equal(got_items, np.sort(ki[:, 0]))
        assert np.array_equal(X.cls[ki[:, 0]], ki[:, 1])
        head, val = gold[tag + ".knn_head"], gold[tag + ".knn_val"]
        assert int(X.cnt[X.cls > 0].sum()) == len(head)
        it, lid, pos, nbr = head.T
        l01 = lid % 2
        assert np.array_equal(X.col[it, l01, pos], nbr)
        if method == "cosine":
            assert np.array_equal(X.val[it, l01, pos], val)
        else:
            np.testing.assert_allclose(X.val[it, l01, pos], val, rtol=1e-11, atol=0)
        # X-Sim: (start,end) set exact, values to 1e-12 (np.dot/BLAS order is not reproducible)
        st, en = csr_to_pairs(X.xs_ptr, X.xs_end)
        xh = gold[tag + ".xsim_head"]
        assert np.array_equal(st, xh[:, 0]) and np.array_equal(en, xh[:, 1])
        np.testing.assert_allclose(X.xs_val, gold[tag + ".xsim_val"], rtol=1e-10, atol=1e-300)
        for gt in gold.gen_tags(method, k):
            gtag = tag + "." + gt
            private = gt == "priv"
            picks = None
            n_top, _, _ = xo.select(T, X, private, None)
            if not private:
                seed = int(gt[2:])
                if gold.has(gtag + ".raises"):
                    with pytest.raises(ValueError):
                        xo.draw_picks(n_top, seed)
                    continue
                picks = xo.draw_picks(n_top, seed)
            n_top, choice, m = xo.select(T, X, private, picks)
            exp = gold[gtag + ".choice"]
            starts = np.nonzero(n_top)[0]
            assert np.array_equal(starts, exp[:, 0])
            assert np.array_equal(choice[starts], exp[:, 1])
            ae = xo.alterego(T, m)
            eh = gold[gtag + ".ae_head"]
            nt = ae["n_target_rows"]
            # pass-through rows: same order as the reference; AlterEgo rows: same per-user order
            assert np.array_equal(ae["user"][:nt], eh[:nt, 0]) and np.array_equal(ae["item"][:nt], eh[:nt, 1])
            assert np.array_equal(ae["user"], eh[:, 0]) and np.array_equal(ae["item"], eh[:, 1])
            assert np.array_equal(ae["rating"], gold[gtag + ".ae_rating"])
            assert np.array_equal(ae["time"], gold[gtag + ".ae_time"])
        xo.ext_free(X)
    xo.sim_free(S)


# ---------------------------------------------------------------------------------------------------------------
# RecommenderSim: oracle vs the reference's own output on the AlterEgo profile of the 'small' case,
# with integer ratings and with non-integer ones (AlterEgo ratings are means)
REC_RTOL = 1e-9   # exact sums (inner product, squared norms) vs python sum() / np.sum roundings


@pytest.mark.parametrize("key,rows_key", [("cosine_item", None), ("adjust_cosine_item", None), ("cosine_item_float", "rows")])
def test_rec_sim_oracle_matches_reference(key, rows_key):
    import gzip
    import json
    import os
    from golden_util import rows_to_csr
    from oracle import xmap_oracle as xo
    with gzip.open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "small_downstream.json.gz"), "rt") as f:
        g = json.load(f)
    rows = g[key][rows_key] if rows_key else g["downstream_input"]["rows"]
    uids, iids, ptr, item, rating = rows_to_csr(rows)
    # the oracle sees the reference's own fp64 values (4 - 1/3, ... in the float case: no float32 holds them)
    by_user = {}
    for r in rows:
        by_user.setdefault(r[0], []).append(float(r[2]))
    assert np.array_equal(rating, np.array([x for u in uids for x in by_user[u]], np.float64))
    if rows_key:
        assert np.any(rating != rating.astype(np.float32).astype(np.float64))
    R = xo.rec_sim(ptr, item, rating, len(iids), 50)
    got = {}
    for i in range(len(iids)):
        for p in range(R.row_ptr[i], R.row_ptr[i + 1]):
            got[(iids[i], iids[R.col[p]])] = (R.sim[p], R.ls[p])
    want = {(a, b): (v[0], v[1]) for (a, b), v in g[key]["sim"]}
    assert set(got) == set(want)
    for kk, (s, l) in want.items():
        gs, gl = got[kk]
        assert np.isnan(l) == np.isnan(gl)
        assert gs == pytest.approx(s, rel=REC_RTOL, abs=1e-15)
        if not np.isnan(l):
            assert gl == pytest.approx(l, rel=REC_RTOL, abs=1e-13)
    info = dict((i, v) for i, v in g[key]["item_info"])
    for i, name in enumerate(iids):
        assert R.norm[i] == pytest.approx(info[name][1], rel=1e-14)


def test_oracle_extension_of_a_sample_of_starts():
    """xo.extend(starts=...) enumerates only the paths that start at the given items: their X-Sim lists equal the ones of the
    full enumeration (which is pinned against the golden vectors above), the other starts get none, and the path counts of
    disjoint samples add up to the full count."""
    from oracle import xmap_oracle as xo
    from xmap.engine import synth
    r = synth.make_two_domain(5, 1500, 400, 400, overlap=0.3)
    T = xo.Train(r.user_ptr, r.item, r.rating, r.time, r.n_items, *r.item_attrs())
    S = xo.item_sim(T, "adjust_cosine", 50)
    full = xo.extend(T, S, 5)
    has = np.nonzero(np.diff(full.xs_ptr))[0]
    assert len(has) > 50
    rng = np.random.default_rng(1)
    sample = np.sort(rng.choice(has, size=40, replace=False))
    rest = np.setdiff1d(np.arange(r.n_items), sample)
    a, b = xo.extend(T, S, 5, starts=sample), xo.extend(T, S, 5, starts=rest)
    assert a.n_paths + b.n_paths == full.n_paths and 0 < a.n_paths < full.n_paths
    for s in range(r.n_items):
        lo, hi = full.xs_ptr[s], full.xs_ptr[s + 1]
        x = a if s in set(sample.tolist()) else b
        y = b if x is a else a
        assert x.xs_ptr[s + 1] - x.xs_ptr[s] == hi - lo and y.xs_ptr[s + 1] == y.xs_ptr[s]
        assert np.array_equal(x.xs_end[x.xs_ptr[s]:x.xs_ptr[s + 1]], full.xs_end[lo:hi])
        assert np.array_equal(x.xs_val[x.xs_ptr[s]:x.xs_ptr[s + 1]], full.xs_val[lo:hi])
    for x in (a, b, full):
        xo.ext_free(x)
    xo.sim_free(S)
