"""GPU parity tests (run on the MI355X box: `pytest -m gpu`).  Everything goes through the C ABI of
libxmap_hip.so (xmap.engine.device -> ctypes); results are compared with
  * the golden vectors captured from the reference (tests/golden/*.npz), and
  * the CPU oracle (oracle/xmap_oracle.c) on seeded synthetic inputs (BASELINE configs[0] size),
ids / index sets bit-exact, similarities to the tolerance stated next to each assert.
"""
import os

import numpy as np
import pytest

from golden_util import CASES, METHODS, CAP, Golden, csr_to_pairs

pytestmark = pytest.mark.gpu

SIM_RTOL = 1e-11   # adjusted-cosine: exact (error-free) sum of the fp64 terms vs the reference's np.sum rounding
XSIM_RTOL = 1e-9   # sum over paths: exact sum vs the reference's BLAS dot / pairwise rounding


@pytest.fixture(scope="module")
def dev():
    import torch
    assert torch.cuda.is_available(), "needs the MI355X"
    from xmap.engine import device  # raises if libxmap_hip.so is missing: no CPU fallback
    return device


def _engine(dev, ptr, item, rating, time, I, attrs):
    R = dev.DeviceRatings(ptr, item, rating, time, I, attrs)
    return dev.Engine(R)


def _sorted_sim(S):
    row_ptr = S.row_ptr.cpu().numpy()
    rows = np.repeat(np.arange(len(row_ptr) - 1, dtype=np.int64), np.diff(row_ptr))
    col = S.col.cpu().numpy().astype(np.int64)
    o = np.lexsort((col, rows))
    return rows[o], col[o], S.sim.cpu().numpy()[o], S.mutu.cpu().numpy()[o], S.nij.cpu().numpy()[o]


def _xsim_lists(E, I):
    n_cand = E.n_cand.cpu().numpy()[:I]
    off = E.xs_off.cpu().numpy()[:I]
    xe = E.xs_end.cpu().numpy()
    xv = E.xs_val.cpu().numpy()
    st, en, va = [], [], []
    for s in np.nonzero(n_cand)[0]:
        e = xe[off[s]:off[s] + n_cand[s]]
        v = xv[off[s]:off[s] + n_cand[s]]
        o = np.argsort(e)
        st.append(np.full(len(e), s)); en.append(e[o]); va.append(v[o])
    if not st:
        return np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0)
    return np.concatenate(st), np.concatenate(en), np.concatenate(va)


@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("case", CASES)
def test_golden_all_stages(dev, case, method):
    gold = Golden(case)
    eng = _engine(dev, gold.ptr, gold.item, gold.rating, gold.time, gold.I, gold.attrs)
    exact = method == "cosine"
    # ---- stage A
    S = eng.item_sim(method, CAP)
    exp_u = gold[method + ".user_info"]
    assert np.array_equal(S.u_avg.cpu().numpy()[:len(exp_u)], exp_u[:, 0])
    assert np.array_equal(S.u_norm.cpu().numpy()[:len(exp_u)], exp_u[:, 1])
    info = S.info.cpu().numpy()
    exp_i = gold[method + ".item_info"]
    assert np.array_equal(info[:, [0, 1, 3]], exp_i[:, [0, 1, 3]])
    np.testing.assert_allclose(info[:, 2], exp_i[:, 2], rtol=1e-14)
    rows, cols, sim, mutu, nij = _sorted_sim(S)
    assert np.array_equal(rows, gold[method + ".sim_i"])
    assert np.array_equal(cols, gold[method + ".sim_j"])
    val = gold[method + ".sim_val"]
    assert np.array_equal(mutu.astype(np.float64), val[:, 1])
    assert np.array_equal(mutu / (info[rows, 3] + info[cols, 3] - nij), val[:, 2])
    if exact:
        assert np.array_equal(sim, val[:, 0])
    else:
        np.testing.assert_allclose(sim, val[:, 0], rtol=SIM_RTOL, atol=0)
    # symmetric bit for bit
    back = {(int(a), int(b)): s for a, b, s in zip(rows, cols, sim)}
    assert all(back[(b, a)] == s for (a, b), s in back.items())
    # ---- stage B / C
    for k in gold.ks(method):
        tag = "%s.k%d" % (method, k)
        E = eng.extend(S, k, full=True)
        assert np.array_equal(np.nonzero(E.bb.cpu().numpy()[:gold.I])[0], gold[tag + ".bb"])
        ki = gold[tag + ".knn_items"]
        cls = E.cls.cpu().numpy()[:gold.I]
        assert np.array_equal(np.nonzero(cls)[0], np.sort(ki[:, 0]))
        assert np.array_equal(cls[ki[:, 0]], ki[:, 1])
        head, kv = gold[tag + ".knn_head"], gold[tag + ".knn_val"]
        kcnt = E.kcnt.cpu().numpy()[:gold.I]
        assert int(kcnt[cls > 0].sum()) == len(head)
        it, lid, pos, nbr = head.T
        assert np.array_equal(E.kcol.cpu().numpy()[it, lid % 2, pos], nbr)
        got = E.kval.cpu().numpy()[it, lid % 2, pos]
        if exact:
            assert np.array_equal(got, kv)
        else:
            np.testing.assert_allclose(got, kv, rtol=SIM_RTOL, atol=0)
        st, en, va = _xsim_lists(E, gold.I)
        xh = gold[tag + ".xsim_head"]
        assert np.array_equal(st, xh[:, 0]) and np.array_equal(en, xh[:, 1])
        np.testing.assert_allclose(va, gold[tag + ".xsim_val"], rtol=XSIM_RTOL, atol=1e-300)
        for gt in gold.gen_tags(method, k):
            gtag = tag + "." + gt
            private = gt == "priv"
            picks = None
            if not private:
                n_top, _, _ = eng.select(E, False, None)
                np.random.seed(int(gt[2:]))
                if gold.has(gtag + ".raises"):
                    with pytest.raises(ValueError):
                        dev.draw_picks(n_top.cpu().numpy()[:gold.I])
                    continue
                picks = dev.draw_picks(n_top.cpu().numpy()[:gold.I])
            n_top, choice, mp = eng.select(E, private, picks)
            exp = gold[gtag + ".choice"]
            n_top = n_top.cpu().numpy()[:gold.I]
            starts = np.nonzero(n_top)[0]
            assert np.array_equal(starts, exp[:, 0])
            assert np.array_equal(choice.cpu().numpy()[starts], exp[:, 1])
            G = eng.alterego(mp)
            eh = gold[gtag + ".ae_head"]
            assert np.array_equal(G.user.cpu().numpy(), eh[:, 0])
            assert np.array_equal(G.item.cpu().numpy(), eh[:, 1])
            assert np.array_equal(G.rating.cpu().numpy(), gold[gtag + ".ae_rating"])      # fp64 means, bit for bit
            assert np.array_equal(G.time.cpu().numpy(), gold[gtag + ".ae_time"])


def _check_all_stages(dev, r, method, k, private=True, picks_seed=None, **sim_kw):
    """every stage of one pass against the CPU oracle, bit for bit (ratings of the AlterEgo rows: fp32, atol 1e-5)"""
    from oracle import xmap_oracle as xo
    attrs = r.item_attrs()
    eng = _engine(dev, r.user_ptr, r.item, r.rating, r.time, r.n_items, attrs)
    T = xo.Train(r.user_ptr, r.item, r.rating, r.time, r.n_items, *attrs)
    So = xo.item_sim(T, method, CAP, nthreads=8)
    S = eng.item_sim_tri(method, CAP, **sim_kw) if sim_kw else eng.item_sim(method, CAP)
    assert S.n_eval == So.n_eval and S.n_contrib == So.n_contrib
    rows, cols, sim, mutu, nij = _sorted_sim(S)
    orow, ocol = csr_to_pairs(So.row_ptr, So.col)
    assert np.array_equal(rows, orow) and np.array_equal(cols, ocol)
    assert np.array_equal(mutu, So.mutu) and np.array_equal(nij, So.nij)
    # both sides sum error-free (cosine: integer-exact) -> bit-identical in both modes
    assert np.array_equal(S.info.cpu().numpy(), So.info)
    assert np.array_equal(sim, So.sim)
    Xo = xo.extend(T, So, k)
    E = eng.extend(S, k, full=True)
    assert np.array_equal(E.bb.cpu().numpy()[:T.I], Xo.bb)
    assert np.array_equal(E.cls.cpu().numpy()[:T.I], Xo.cls)
    assert np.array_equal(E.kcnt.cpu().numpy()[:T.I], Xo.cnt)
    kc = E.kcol.cpu().numpy()[:T.I]
    msk = np.arange(k)[None, None, :] < Xo.cnt[:, :, None]
    assert np.array_equal(kc[msk], Xo.col[msk])
    assert E.n_paths == Xo.n_paths
    st, en, va = _xsim_lists(E, T.I)
    ost, oen = csr_to_pairs(Xo.xs_ptr, Xo.xs_end)
    assert np.array_equal(st, ost) and np.array_equal(en, oen)
    # path values are bit-identical and both sides sum them error-free (double-double)
    assert np.array_equal(va, Xo.xs_val)
    picks = None
    if not private:
        n_top_probe, _, _ = xo.select(T, Xo, True, None)
        np.random.seed(picks_seed)
        try:
            picks = dev.draw_picks(np.minimum(n_top_probe, 4))
        except ValueError:       # a start with a single candidate: the reference's randint(0, 0) raises (generator.py:109)
            picks = None
            private = True
    n_top_o, choice_o, m_o = xo.select(T, Xo, private, picks)
    n_top, choice, mp = eng.select(E, private, picks)
    assert np.array_equal(n_top.cpu().numpy()[:T.I], n_top_o)
    assert np.array_equal(choice.cpu().numpy()[:T.I], choice_o)
    assert np.array_equal(mp.cpu().numpy()[:T.I], m_o)
    G = eng.alterego(mp)
    ae = xo.alterego(T, m_o)
    assert np.array_equal(G.user.cpu().numpy(), ae["user"]) and np.array_equal(G.item.cpu().numpy(), ae["item"])
    assert np.array_equal(G.rating.cpu().numpy(), ae["rating"])
    assert np.array_equal(G.time.cpu().numpy(), ae["time"])
    assert eng.n_profiles(G) == ae["n_profiles"]
    xo.ext_free(Xo)
    xo.sim_free(So)


@pytest.mark.parametrize("method", METHODS)
def test_c1_vs_oracle(dev, method):
    """BASELINE configs[0] size (10k users / 2x5k items): every stage against the CPU oracle."""
    from xmap.engine import synth
    _check_all_stages(dev, synth.config_c1(), method, 5)


# a sweep over shapes: sparse / dense, skewed, tiny, one domain much larger, heavy rows forced, many partitions
SWEEP = [
    dict(seed=101, users=40, src=15, tgt=15, overlap=0.6, k=2),
    dict(seed=102, users=300, src=40, tgt=200, overlap=0.3, k=3),
    dict(seed=103, users=800, src=300, tgt=60, overlap=0.5, k=7, mu=1.6),
    dict(seed=104, users=1500, src=120, tgt=120, overlap=0.9, k=4, zipf=1.3),
    dict(seed=105, users=2500, src=900, tgt=900, overlap=0.15, k=10),
    dict(seed=106, users=600, src=80, tgt=80, overlap=0.4, k=5, mu=2.4, sigma=0.8, sim_kw=dict(ch_min=64)),
    dict(seed=107, users=1200, src=250, tgt=250, overlap=0.35, k=6, sim_kw=dict(slot_target=32)),
    dict(seed=108, users=200, src=500, tgt=500, overlap=0.5, k=3, mu=3.0),
    dict(seed=109, users=900, src=260, tgt=260, overlap=0.5, k=100, mu=1.4),     # BASELINE configs[3]'s top-k
    dict(seed=110, users=500, src=150, tgt=150, overlap=0.6, k=64),
]


@pytest.mark.parametrize("cfg", SWEEP, ids=lambda c: "s%d" % c["seed"])
@pytest.mark.parametrize("method", METHODS)
def test_shape_sweep_vs_oracle(dev, method, cfg):
    from xmap.engine import synth
    kw = {n: cfg[n] for n in ("overlap", "mu", "sigma", "zipf") if n in cfg}
    r = synth.make_two_domain(cfg["seed"], cfg["users"], cfg["src"], cfg["tgt"], **kw)
    _check_all_stages(dev, r, method, cfg["k"], private=(cfg["seed"] % 2 == 0), picks_seed=cfg["seed"],
                      **cfg.get("sim_kw", {}))


def test_edge_cases(dev):
    """empty input, a single user, users with one rating only (no pairs)."""
    attrs1 = (np.zeros(3, np.int32), np.zeros(3, np.int32), np.ones(3, np.uint32), np.ones(3, np.uint8))
    # users with a single rating each -> no pairs at all
    eng = _engine(dev, np.array([0, 1, 2, 3]), np.array([0, 1, 2]), np.array([5., 4., 3.]),
                  np.array([1, 2, 3]), 3, attrs1)
    S = eng.item_sim("cosine", CAP)
    assert S.n_kept == 0 and S.n_eval == 0
    E = eng.extend(S, 3, full=True)
    assert E.n_out == 0 and E.n_paths == 0
    n_top, choice, mp = eng.select(E, True)
    assert int(n_top.sum()) == 0 and (mp.cpu().numpy()[:3] == -1).all()
    G = eng.alterego(mp)
    assert G.n_rows == 0
    # one user, two items of one domain: one symmetric pair, no bridge
    eng = _engine(dev, np.array([0, 2]), np.array([0, 1]), np.array([5., 3.]), np.array([1, 2]), 3, attrs1)
    S = eng.item_sim("cosine", CAP)
    assert S.n_kept == 2 and S.n_eval == 2
    E = eng.extend(S, 3, full=True)
    assert int(E.bb.sum()) == 0 and E.n_out == 0


@pytest.mark.parametrize("method", METHODS)
def test_long_profiles_and_popular_items(dev, method):
    """Profiles of up to 1100 ratings (all three profile-sort paths: packed 16-lane groups, one wave, counted ranks
    through LDS and through the scratch array) and items rated by most users (LDS-privatised CSC build, heavy rows)."""
    from oracle import xmap_oracle as xo
    from xmap.engine import synth
    r = synth.make_two_domain(21, 600, 1500, 1500, overlap=0.5, mu=4.0, sigma=1.6)
    d = np.diff(r.user_ptr)
    assert d.max() > 1024 and (d > 64).sum() > 100 and (d <= 16).sum() > 10
    attrs = r.item_attrs()
    eng = _engine(dev, r.user_ptr, r.item, r.rating, r.time, r.n_items, attrs)
    T = xo.Train(r.user_ptr, r.item, r.rating, r.time, r.n_items, *attrs)
    So = xo.item_sim(T, method, CAP, nthreads=8)
    orow, ocol = csr_to_pairs(So.row_ptr, So.col)
    for S in (eng.item_sim(method, CAP), eng.item_sim_tri(method, CAP, ch_min=64), eng.item_sim(method, CAP, algo="rows")):
        assert S.n_eval == So.n_eval and S.n_contrib == So.n_contrib
        rows, cols, sim, mutu, nij = _sorted_sim(S)
        assert np.array_equal(rows, orow) and np.array_equal(cols, ocol)
        assert np.array_equal(mutu, So.mutu) and np.array_equal(nij, So.nij)
        assert np.array_equal(S.info.cpu().numpy(), So.info)
        assert np.array_equal(sim, So.sim)
    # the on-device CSC is a permutation of the ratings grouped by item
    R = eng.R
    ip = R.item_ptr.cpu().numpy()
    assert np.array_equal(np.diff(ip), np.bincount(r.item, minlength=r.n_items))
    iu, ir = R.item_user.cpu().numpy(), R.item_rating.cpu().numpy()
    users = np.repeat(np.arange(r.n_users), d)
    o = np.lexsort((users, r.item))
    for i in (0, int(np.argmax(np.diff(ip))), r.n_items - 1):
        a, b = ip[i], ip[i + 1]
        q = np.argsort(iu[a:b])
        assert np.array_equal(iu[a:b][q], users[o][a:b]) and np.array_equal(ir[a:b][q], r.rating[o][a:b])
    xo.sim_free(So)


@pytest.mark.parametrize("method", METHODS)
def test_keys_larger_than_a_tile(dev, method):
    """The two transpositions of stage A (rater records by item, kept pairs by heavier item: csrc/tilesort.h) with keys
    far larger than a tile: two items rated by every user (8000 raters; each is the partner of every lighter row, so its
    mirrored half has thousands of entries too), next to thousands of items with a handful.  Both sequences of the stage
    (round 3: one transposition, tile-sorted mirror; round 2: CSC + cursor-atomic mirror) against the oracle, bit for bit."""
    from oracle import xmap_oracle as xo
    from xmap.engine import synth
    r = synth.make_two_domain(31, 8000, 4000, 4000, overlap=0.4)
    # every user also rates source item 7 and target item n_src + 11 (appended to the profile if absent)
    hubs = (7, r.n_src_items + 11)
    rng = np.random.default_rng(5)
    ptr, item, rating, time = [0], [], [], []
    for u in range(r.n_users):
        a, b = int(r.user_ptr[u]), int(r.user_ptr[u + 1])
        it, ra, ti = list(r.item[a:b]), list(r.rating[a:b]), list(r.time[a:b])
        for h in hubs:
            if h not in it:
                it.append(h); ra.append(float(rng.integers(1, 6))); ti.append(int(rng.integers(synth.T0, synth.T1)))
        item += it; rating += ra; time += ti
        ptr.append(len(item))
    ptr, item = np.asarray(ptr, np.int64), np.asarray(item, np.int32)
    rating, time = np.asarray(rating, np.float32), np.asarray(time, np.int64)
    assert np.bincount(item, minlength=r.n_items)[list(hubs)].tolist() == [r.n_users, r.n_users]
    attrs = r.item_attrs()
    eng = _engine(dev, ptr, item, rating, time, r.n_items, attrs)
    T = xo.Train(ptr, item, rating, time, r.n_items, *attrs)
    So = xo.item_sim(T, method, CAP, nthreads=8)
    orow, ocol = csr_to_pairs(So.row_ptr, So.col)
    assert np.diff(So.row_ptr)[list(hubs)].min() > 3000
    old = os.environ.get("XMAP_A_V2")
    try:
        for v2 in ("0", "1"):
            os.environ["XMAP_A_V2"] = v2
            for ch_min in (1024, 64):
                S = eng.item_sim_tri(method, CAP, ch_min=ch_min)
                assert S.n_eval == So.n_eval and S.n_contrib == So.n_contrib
                rows, cols, sim, mutu, nij = _sorted_sim(S)
                assert np.array_equal(rows, orow) and np.array_equal(cols, ocol)
                assert np.array_equal(mutu, So.mutu) and np.array_equal(nij, So.nij)
                assert np.array_equal(S.info.cpu().numpy(), So.info)
                assert np.array_equal(S.u_avg.cpu().numpy()[:r.n_users], xo.user_info(T)[0])
                assert np.array_equal(sim, So.sim)
    finally:
        if old is None:
            os.environ.pop("XMAP_A_V2", None)
        else:
            os.environ["XMAP_A_V2"] = old
    # the rater counts of large inputs come from the partitioned count (k_cb_*: bucket histogram, scatter, LDS windows),
    # forced here on this input: the hub items are 8000 same-word increments of one window
    os.environ["XMAP_COUNT_PART_MIN"] = "1"
    try:
        S = eng.item_sim_tri(method, CAP)
        rows, cols, sim, mutu, nij = _sorted_sim(S)
        assert np.array_equal(rows, orow) and np.array_equal(cols, ocol) and np.array_equal(sim, So.sim)
        assert np.array_equal(S.info.cpu().numpy(), So.info)
    finally:
        os.environ.pop("XMAP_COUNT_PART_MIN", None)
    xo.sim_free(So)


def test_partitioned_count_on_random_shapes(dev):
    """xmap_sim3_layout's partitioned rater count (large inputs) forced on small random shapes, incl. fewer items than
    buckets and items beyond a bucket boundary: stage A against the oracle, bit for bit."""
    from oracle import xmap_oracle as xo
    from xmap.engine import synth
    os.environ["XMAP_COUNT_PART_MIN"] = "1"
    try:
        for seed, (U, Is, It) in enumerate(((300, 40, 50), (2500, 900, 700), (6000, 2100, 1900))):
            r = synth.make_two_domain(100 + seed, U, Is, It, overlap=0.3)
            attrs = r.item_attrs()
            eng = _engine(dev, r.user_ptr, r.item, r.rating, r.time, r.n_items, attrs)
            T = xo.Train(r.user_ptr, r.item, r.rating, r.time, r.n_items, *attrs)
            So = xo.item_sim(T, "adjust_cosine", CAP, nthreads=8)
            S = eng.item_sim_tri("adjust_cosine", CAP)
            rows, cols, sim, mutu, nij = _sorted_sim(S)
            orow, ocol = csr_to_pairs(So.row_ptr, So.col)
            assert np.array_equal(rows, orow) and np.array_equal(cols, ocol) and np.array_equal(sim, So.sim)
            assert np.array_equal(S.info.cpu().numpy(), So.info)
            xo.sim_free(So)
    finally:
        os.environ.pop("XMAP_COUNT_PART_MIN", None)


def test_reverse_lists_fused_count_equals_separate_passes(dev, monkeypatch):
    """attach and rnn lists are counted in ONE pass over the matrix (xmap_reverse_count_att_rnn; its byte per entry serves both
    fill passes); XMAP_REV_SEPARATE=1 counts them in a pass each as rounds 1-4a did (mode 2 then writes its own bit): the
    three reverse adjacencies come out identical, also with every row walked by the 16-wave kernel of the long rows"""
    from xmap.engine import synth
    r = synth.make_two_domain(23, 3000, 700, 700, overlap=0.15)
    eng = _engine(dev, r.user_ptr, r.item, r.rating, r.time, r.n_items, r.item_attrs())
    S = eng.item_sim("adjust_cosine", CAP)

    def lists():
        E = eng.ext_tables(S, 10)
        return [t.cpu().numpy() for name in ("att", "src", "rnn") for t in getattr(E, name)[:4]]
    fused = lists()
    monkeypatch.setenv("XMAP_REV_SEPARATE", "1")
    separate = lists()
    monkeypatch.setenv("XMAP_REV_LONG", "8")           # (rows of more than 8 entries: k_reverse_long)
    separate_long = lists()
    monkeypatch.delenv("XMAP_REV_SEPARATE")
    fused_long = lists()
    assert sum(len(x) for x in fused) > 1000
    for other in (separate, separate_long, fused_long):
        for x, y in zip(fused, other):
            assert np.array_equal(x, y)


def test_middle_lists_in_column_ranges(dev, monkeypatch):
    """k_mid_rows keeps the tile counters of XMAP_MID_ROWS_SPAN columns in LDS and builds wider rows range by range: with a
    span of 37 columns (several ranges per row here) the extension equals the one-range build and the oracle, bit for bit."""
    from oracle import xmap_oracle as xo
    from xmap.engine import synth
    r = synth.make_two_domain(17, 3000, 700, 700, overlap=0.15)
    attrs = r.item_attrs()
    eng = _engine(dev, r.user_ptr, r.item, r.rating, r.time, r.n_items, attrs)
    S = eng.item_sim("adjust_cosine", CAP)
    E0 = eng.extend(S, 10, full=True)
    assert E0.mid is not None and E0.mid.n_nb > 100
    monkeypatch.setenv("XMAP_MID_ROWS_SPAN", "37")
    E1 = eng.extend(S, 10, full=True)
    monkeypatch.delenv("XMAP_MID_ROWS_SPAN")
    assert E1.mid.n_tiles == E0.mid.n_tiles and E1.mid.n_records == E0.mid.n_records
    assert np.array_equal(E1.mid.dir_ptr.cpu().numpy(), E0.mid.dir_ptr.cpu().numpy())
    d0, d1 = E0.mid.dir.cpu().numpy().reshape(-1, 3), E1.mid.dir.cpu().numpy().reshape(-1, 3)
    assert np.array_equal(d0[:E0.mid.n_tiles], d1[:E1.mid.n_tiles])          # same tiles, same x order, same offsets
    assert E1.n_paths == E0.n_paths and E1.n_out == E0.n_out
    a, b = _xsim_lists(E0, r.n_items), _xsim_lists(E1, r.n_items)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    T = xo.Train(r.user_ptr, r.item, r.rating, r.time, r.n_items, *attrs)
    So = xo.item_sim(T, "adjust_cosine", CAP, nthreads=8)
    Xo = xo.extend(T, So, 10)
    assert E1.n_paths == Xo.n_paths
    ost, oen = csr_to_pairs(Xo.xs_ptr, Xo.xs_end)
    assert np.array_equal(b[0], ost) and np.array_equal(b[1], oen) and np.array_equal(b[2], Xo.xs_val)
    xo.ext_free(Xo)
    xo.sim_free(So)


def test_determinism_and_partitions(dev):
    """two runs give identical bytes; the result does not depend on the table partitioning."""
    from xmap.engine import synth
    r = synth.make_two_domain(9, 3000, 600, 600)
    eng = _engine(dev, r.user_ptr, r.item, r.rating, r.time, r.n_items, r.item_attrs())
    a = _sorted_sim(eng.item_sim("adjust_cosine", CAP))
    b = _sorted_sim(eng.item_sim("adjust_cosine", CAP))
    c = _sorted_sim(eng.item_sim("adjust_cosine", CAP, slot_target=48))  # many partitions per item
    d = _sorted_sim(eng.item_sim("adjust_cosine", CAP, algo="rows"))     # first formulation: complete rows
    e = _sorted_sim(eng.item_sim("adjust_cosine", CAP, algo="rows", slot_target=48))
    f = _sorted_sim(eng.item_sim_tri("adjust_cosine", CAP, ch_min=64))   # forces heavy rows + rater chunks
    for x, y, z, w, v, t in zip(a, b, c, d, e, f):
        assert np.array_equal(x, y) and np.array_equal(x, z) and np.array_equal(x, w) and np.array_equal(x, v)
        assert np.array_equal(x, t)
    for m in ("cosine",):
        g = _sorted_sim(eng.item_sim(m, CAP))
        h = _sorted_sim(eng.item_sim(m, CAP, algo="rows"))
        k2 = _sorted_sim(eng.item_sim_tri(m, CAP, ch_min=64))
        for x, y, z in zip(g, h, k2):
            assert np.array_equal(x, y) and np.array_equal(x, z)
    # stage B: heavy starts split into chunks with dedicated rows + device merge == one wave per start
    S = eng.item_sim("adjust_cosine", CAP)
    E1 = eng.extend(S, 5, full=True, chunk=1 << 40)
    E2 = eng.extend(S, 5, full=True, chunk=64, n_slots=64)
    E3 = eng.extend(S, 5, full=True, algo="mid")               # middle lists + register-tile accumulation
    E4 = eng.extend(S, 5, full=True, algo="mid", chunk=64)
    E6 = eng.extend(S, 5, full=True, algo="enum")              # one accumulate per path
    os.environ["XMAP_MID_TABLE"] = "1"                        # middle lists through the dense tile table (n_nb > 40 000 form)
    try:
        E5 = eng.extend(S, 5, full=True, algo="mid")
    finally:
        del os.environ["XMAP_MID_TABLE"]
    os.environ["XMAP_SLOW_DIV"] = "1"                         # k_paths4<false>: the IEEE division of calculate_path_confidence
    try:
        E7 = eng.extend(S, 5, full=True)
    finally:
        del os.environ["XMAP_SLOW_DIV"]
    assert E1.fast_div == 1                                   # stage A's output always meets div_mid's precondition
    assert E5.mid.n_records == E3.mid.n_records and E5.mid.n_tiles == E3.mid.n_tiles
    assert np.array_equal(E5.mid.dir.view(-1, 3)[:, 0].cpu().numpy(), E3.mid.dir.view(-1, 3)[:, 0].cpu().numpy())   # (x, ne) per tile
    for Ex in (E3, E4, E5, E6, E7):
        assert Ex.n_paths == E1.n_paths and Ex.n_out == E1.n_out
        for x, y in zip(_xsim_lists(E1, r.n_items), _xsim_lists(Ex, r.n_items)):
            assert np.array_equal(x, y)
        assert np.array_equal(E1.top_end.cpu().numpy(), Ex.top_end.cpu().numpy())
        assert np.array_equal(E1.top_val.cpu().numpy(), Ex.top_val.cpu().numpy())
    assert E1.units.n_heavy == 0 and E2.units.n_heavy > 10
    assert E1.n_paths == E2.n_paths and E1.n_out == E2.n_out
    for x, y in zip(_xsim_lists(E1, r.n_items), _xsim_lists(E2, r.n_items)):
        assert np.array_equal(x, y)
    assert np.array_equal(E1.top_end.cpu().numpy(), E2.top_end.cpu().numpy())
    assert np.array_equal(E1.top_val.cpu().numpy(), E2.top_val.cpu().numpy())
    assert np.array_equal(E1.n_cand.cpu().numpy(), E2.n_cand.cpu().numpy())


def test_heavy_starts_with_long_candidate_lists(dev):
    """Starts split over dedicated rows whose candidate lists are long enough for every wave of the merge block to cut
    its running selection back several times (k_merge_groups + k_merge + finalize_slice with 16 waves) against the
    one-wave-per-start form (finalize_start), which test_c1_vs_oracle pins to the oracle at the same size."""
    from xmap.engine import synth
    r = synth.config_c1()
    eng = _engine(dev, r.user_ptr, r.item, r.rating, r.time, r.n_items, r.item_attrs())
    S = eng.item_sim("adjust_cosine", CAP)
    E1 = eng.extend(S, 10, full=True, chunk=1 << 40)
    assert E1.units.n_heavy == 0
    assert int(E1.n_cand.max().item()) > 16 * 128          # slices longer than the selection buffer
    for chunk in (1 << 14, 1 << 11):                       # few rows per start / more than one merge group
        E2 = eng.extend(S, 10, full=True, chunk=chunk)
        assert E2.units.n_heavy > 10
        g = E2.units.unit_G.cpu().numpy()[:E2.units.n_units]
        if chunk == 1 << 11:
            assert g.max() > 12                            # more than one merge group
        assert E1.n_paths == E2.n_paths and E1.n_out == E2.n_out
        assert np.array_equal(E1.n_cand.cpu().numpy(), E2.n_cand.cpu().numpy())
        assert np.array_equal(E1.top_end.cpu().numpy(), E2.top_end.cpu().numpy())
        assert np.array_equal(E1.top_val.cpu().numpy(), E2.top_val.cpu().numpy())
        for x, y in zip(_xsim_lists(E1, r.n_items), _xsim_lists(E2, r.n_items)):
            assert np.array_equal(x, y)


def test_k50_vs_oracle(dev):
    """the list length of BASELINE configs[1] (k = 50) against the CPU oracle, every stage bit for bit: 9.5e7 paths, the
    largest case the one-thread oracle enumerates in about half a minute"""
    from xmap.engine import synth
    _check_all_stages(dev, synth.make_two_domain(11, 2000, 1000, 1000), "adjust_cosine", 50)


@pytest.mark.parametrize("method", METHODS)
def test_long_rows_of_the_reverse_lists(dev, method, monkeypatch):
    """every row of more than 64 entries through the 16-wave form of k_reverse (XMAP_REV_LONG): all stages against the
    oracle."""
    monkeypatch.setenv("XMAP_REV_LONG", "64")
    from xmap.engine import synth
    _check_all_stages(dev, synth.make_two_domain(5, 2000, 400, 400), method, 5)


def test_rows_longer_than_one_knn_chunk(dev):
    """similarity rows of several thousand entries (k_knn_classify streams what follows its first 2048-entry chunk
    against the lists' thresholds): knn tables and everything downstream against the oracle."""
    from xmap.engine import synth
    r = synth.make_two_domain(17, 2500, 3000, 3000, overlap=0.5, mu=3.2, sigma=1.0)
    eng = _engine(dev, r.user_ptr, r.item, r.rating, r.time, r.n_items, r.item_attrs())
    S = eng.item_sim("cosine", CAP)
    ln = np.diff(S.row_ptr.cpu().numpy())
    assert (ln > 2048).sum() > 200 and ln.max() > 4096
    _check_all_stages(dev, r, "cosine", 3)
