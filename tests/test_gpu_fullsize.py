"""Parity at BASELINE configs[1] size (1 M users / 200k+200k items) through size-independent properties and a
bounded oracle sample: the two independent stage-A formulations agree bit for bit, the CSR is symmetric, the
counters match closed forms computed on the host, a row sample matches the CPU oracle exactly, the two stage-B
formulations agree, and stage C conserves rows."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
CAP = 50


@pytest.fixture(scope="module")
def c2():
    import torch
    assert torch.cuda.is_available()
    from xmap.engine import synth, device
    r = synth.config_c2()
    eng = device.Engine(device.DeviceRatings(r.user_ptr, r.item, r.rating, r.time, r.n_items, r.item_attrs()))
    return r, eng


def _csr(S):
    rp = S.row_ptr.cpu().numpy()
    rows = np.repeat(np.arange(len(rp) - 1, dtype=np.int64), np.diff(rp))
    col = S.col.cpu().numpy().astype(np.int64)
    o = np.lexsort((col, rows))
    return rp, rows[o], col[o], S.sim.cpu().numpy()[o], S.mutu.cpu().numpy()[o], S.nij.cpu().numpy()[o]


def test_stage_a_full_size(c2):
    from oracle import xmap_oracle as xo
    r, eng = c2
    method = "adjust_cosine"
    S = eng.item_sim(method, CAP)                 # each unordered pair once + mirror
    rp, rows, cols, sim, mutu, nij = _csr(S)
    d = np.diff(r.user_ptr)
    assert S.n_contrib == int((d * (d - 1)).sum())                   # P = sum d(d-1)
    assert rp[-1] == S.n_kept and S.n_kept % 2 == 0 and S.n_eval % 2 == 0 and S.n_kept <= S.n_eval <= S.n_contrib
    assert (np.diff(rp) >= 0).all()
    # symmetry, bit for bit: the transposed entry list is the same list
    o2 = np.lexsort((rows, cols))
    assert np.array_equal(rows[o2], cols) and np.array_equal(cols[o2], rows)
    assert np.array_equal(sim[o2], sim) and np.array_equal(mutu[o2], mutu) and np.array_equal(nij[o2], nij)
    assert (sim != 0).all() and (mutu > 0).all() and (mutu <= nij).all() and (np.abs(sim) <= 1 + 1e-12).all()
    n = np.bincount(r.item, minlength=r.n_items)
    assert (nij <= np.minimum(n[rows], n[cols])).all()
    # the complete-rows formulation (independent kernel, hash partitions, no mirroring) gives the same matrix
    S2 = eng.item_sim(method, CAP, algo="rows")
    assert S2.n_eval == S.n_eval and S2.n_kept == S.n_kept
    rp2, rows2, cols2, sim2, mutu2, nij2 = _csr(S2)
    assert np.array_equal(rp, rp2) and np.array_equal(cols, cols2) and np.array_equal(sim, sim2)
    assert np.array_equal(mutu, mutu2) and np.array_equal(nij, nij2)
    del S2
    # the CPU oracle over ALL rows (8 threads: seconds at this size): the whole matrix, bit for bit
    attrs = r.item_attrs()
    T = xo.Train(r.user_ptr, r.item, r.rating, r.time, r.n_items, *attrs)
    So = xo.item_sim(T, method, CAP, nthreads=8)
    assert np.array_equal(S.info.cpu().numpy(), So.info)
    assert S.n_eval == So.n_eval and S.n_contrib == So.n_contrib
    assert np.array_equal(rp, So.row_ptr)
    assert np.array_equal(cols, So.col) and np.array_equal(sim, So.sim)
    assert np.array_equal(mutu, So.mutu) and np.array_equal(nij, So.nij)
    xo.sim_free(So)


def test_stage_a_full_size_cosine(c2):
    """the plain-cosine branch (fp64 LDS atomic sums, integer-exact) at BASELINE configs[1]: the whole matrix against the oracle"""
    from oracle import xmap_oracle as xo
    r, eng = c2
    S = eng.item_sim("cosine", CAP)
    rp, rows, cols, sim, mutu, nij = _csr(S)
    T = xo.Train(r.user_ptr, r.item, r.rating, r.time, r.n_items, *r.item_attrs())
    So = xo.item_sim(T, "cosine", CAP, nthreads=8)
    assert S.n_eval == So.n_eval and np.array_equal(rp, So.row_ptr)
    assert np.array_equal(cols, So.col) and np.array_equal(sim, So.sim)
    assert np.array_equal(mutu, So.mutu) and np.array_equal(nij, So.nij)
    xo.sim_free(So)


def _candidate_properties(r, E, I):
    """size-independent properties of the per-start candidate arrays"""
    n_cand = E.n_cand.cpu().numpy()[:I]
    top_end = E.top_end.cpu().numpy()[:I]
    top_val = E.top_val.cpu().numpy()[:I]
    flags = r.item_attrs()[3]
    has = n_cand > 0
    assert int(n_cand.sum()) == E.n_out
    # starts are target-side, candidates source-side; candidate lists are sorted by (|xsim| desc, end asc)
    assert (flags[np.nonzero(has)[0]] & 2).all()
    m = np.minimum(n_cand, 10)
    for q in range(10):
        sel = m > q
        assert (top_end[sel, q] >= 0).all() and (flags[top_end[sel, q]] & 1).all()
        assert (top_end[~sel, q] == -1).all()
        if q:
            a, b = np.abs(top_val[sel, q - 1]), np.abs(top_val[sel, q])
            assert (a >= b).all()
            tie = a == b
            assert (top_end[sel, q - 1][tie] < top_end[sel, q][tie]).all()
    return n_cand, top_end, top_val, flags, has


@pytest.mark.parametrize("k", [10, 50])
def test_stage_b_c_full_size(c2, k):
    """BASELINE configs[1] at its own list length (k = 50) and at k = 10: the column form (default), the per-path
    enumeration (independent formulation: one accumulate per path) and, at k = 10, the round-1 tile-major form agree
    bit for bit on every per-start output."""
    r, eng = c2
    I = r.n_items
    S = eng.item_sim("cosine" if k == 10 else "adjust_cosine", CAP)
    E1 = eng.extend(S, k)                         # algo "cols"
    E2 = eng.extend(S, k, algo="enum")            # one accumulate per path: independent formulation
    others = [E2] + ([eng.extend(S, k, algo="mid")] if k == 10 else [])
    for Ex in others:
        assert E1.n_paths == Ex.n_paths == E1.units.total and E1.n_out == Ex.n_out
        assert np.array_equal(E1.n_cand.cpu().numpy(), Ex.n_cand.cpu().numpy())
        assert np.array_equal(E1.top_end.cpu().numpy(), Ex.top_end.cpu().numpy())
        assert np.array_equal(E1.top_val.cpu().numpy(), Ex.top_val.cpu().numpy())
    del others, E2
    n_cand, top_end, top_val, flags, has = _candidate_properties(r, E1, I)
    # stage C: private mapping = candidate 0; map is the max start per chosen source; rows are conserved
    n_top, choice, mp_d = eng.select(E1, True)
    choice, mp = choice.cpu().numpy()[:I], mp_d.cpu().numpy()[:I]
    assert np.array_equal(choice[has], top_end[has, 0]) and (choice[~has] == -1).all()
    want = np.full(I, -1, np.int64)
    np.maximum.at(want, choice[has], np.nonzero(has)[0])
    assert np.array_equal(mp, want)
    G = eng.alterego(mp_d)
    G2 = eng.alterego(mp_d)
    u, it, rt = G.user.cpu().numpy(), G.item.cpu().numpy(), G.rating.cpu().numpy()
    assert np.array_equal(u, G2.user.cpu().numpy()) and np.array_equal(rt, G2.rating.cpu().numpy())   # idempotent
    n_t = int((flags[r.item] & 2).astype(bool).sum())
    assert G.n_target_rows == n_t and (flags[it] & 2).all()
    users = np.repeat(np.arange(r.n_users), np.diff(r.user_ptr))
    mapped = mp[r.item] >= 0
    key = users[mapped] * np.int64(I) + mp[r.item][mapped]
    assert G.n_rows - n_t == len(np.unique(key))                      # one AlterEgo row per (user, mapped target)
    assert (rt >= 1).all() and (rt <= 5).all()
    if k == 50:
        # stage C against the oracle on the same replacement map: all 5.2 M rows (users, items, fp64 means, times), exactly
        from oracle import xmap_oracle as xo
        T = xo.Train(r.user_ptr, r.item, r.rating, r.time, r.n_items, *r.item_attrs())
        ae = xo.alterego(T, mp)
        assert np.array_equal(u, ae["user"]) and np.array_equal(it, ae["item"])
        assert np.array_equal(rt, ae["rating"]) and np.array_equal(G.time.cpu().numpy(), ae["time"])
        assert eng.n_profiles(G) == ae["n_profiles"] and G.n_target_rows == ae["n_target_rows"]


def test_stage_b_against_the_oracle_at_full_size(c2):
    """BASELINE configs[1], k = 50, against the CPU oracle (pinned to the reference by tests/test_oracle_golden.py):
    bridge flags and the classified top-k lists of EVERY item, the attach lists derived from them, and -- for a sample
    of starts, whose paths the oracle can enumerate -- the exact path counts, the candidate counts and the ten best
    candidates with their X-Sim values, bit for bit."""
    r, eng = c2
    S = eng.item_sim("adjust_cosine", CAP)
    E = eng.extend(S, 50)
    stage_b_against_the_oracle(r, S, E, 50, "adjust_cosine", n_small=90, n_mid=6, at_least=40)


def stage_b_against_the_oracle(r, S, E, k, method, n_small, n_mid, at_least):
    """(shared with tests/test_gpu_s1.py) S, E: the engine's similarity matrix and extension of r at list length k"""
    from oracle import xmap_oracle as xo
    I = r.n_items
    T = xo.Train(r.user_ptr, r.item, r.rating, r.time, r.n_items, *r.item_attrs())
    So = xo.item_sim(T, method, CAP, nthreads=16)
    assert So.n_eval == S.n_eval and int(So.row_ptr[-1]) == S.n_kept
    Xo = xo.extend(T, So, k, do_paths=False)
    # B1-B4: every item
    assert np.array_equal(E.bb.cpu().numpy()[:I], Xo.bb)
    cls = E.cls.cpu().numpy()[:I]
    assert np.array_equal(cls, Xo.cls)
    kcnt = E.kcnt.cpu().numpy()[:I]
    assert np.array_equal(kcnt, Xo.cnt)
    msk = np.arange(k)[None, None, :] < Xo.cnt[:, :, None]
    kcol = E.kcol.cpu().numpy()[:I]
    assert np.array_equal(kcol[msk], Xo.col[msk])
    assert np.array_equal(E.kval.cpu().numpy()[:I][msk], Xo.val[msk])
    # B5a: attach(b) = the non-bridge records x with b in NB_BB(x), from the oracle's lists
    nb = np.nonzero(Xo.cls == 2)[0]
    c0 = Xo.cnt[nb, 0]
    xs = np.repeat(nb, c0)
    q = np.arange(int(c0.sum())) - np.repeat(np.cumsum(c0) - c0, c0)
    b = Xo.col[xs, 0, q].astype(np.int64)
    o = np.lexsort((xs, b))
    aptr = np.zeros(I + 1, np.int64)
    np.cumsum(np.bincount(b, minlength=I), out=aptr[1:])
    g_ptr, g_idx = E.att[0].cpu().numpy(), E.att[1].cpu().numpy()
    assert np.array_equal(g_ptr[:I + 1], aptr)
    rows = np.repeat(np.arange(I, dtype=np.int64), np.diff(aptr))
    go = np.lexsort((g_idx[:aptr[-1]], rows))
    assert np.array_equal(g_idx[:aptr[-1]][go], xs[o])
    # B5b-B6: a sample of starts
    P = E.units.P.cpu().numpy()[:I]
    assert int(P.sum()) == E.n_paths
    rng = np.random.default_rng(11)
    small = np.nonzero((P > 0) & (P <= 2000000))[0]
    mid = np.nonzero((P > 2000000) & (P <= 30000000))[0]
    sample = np.sort(np.concatenate([rng.choice(small, size=min(n_small, len(small)), replace=False),
                                     rng.choice(mid, size=min(n_mid, len(mid)), replace=False)]))
    assert len(sample) >= at_least
    Xs = xo.extend(T, So, k, starts=sample)
    assert Xs.n_paths == int(P[sample].sum())
    n_cand = E.n_cand.cpu().numpy()[:I]
    top_end, top_val = E.top_end.cpu().numpy()[:I], E.top_val.cpu().numpy()[:I]
    for s in sample:
        lo, hi = int(Xs.xs_ptr[s]), int(Xs.xs_ptr[s + 1])
        ends, vals = Xs.xs_end[lo:hi], Xs.xs_val[lo:hi]
        assert n_cand[s] == hi - lo
        order = np.lexsort((ends, -np.abs(vals)))[:10]
        m = len(order)
        assert np.array_equal(top_end[s, :m], ends[order]) and (top_end[s, m:] == -1).all()
        assert np.array_equal(top_val[s, :m], vals[order])
    xo.ext_free(Xs)
    xo.ext_free(Xo)
    xo.sim_free(So)


def test_full_lists_at_full_size(c2):
    """full=True (what iterating the lazy extended_simRDD triggers) at BASELINE configs[1], k = 50: 4.6e9 (start, end)
    pairs in buffers sized exactly from the candidate counts of the first pass -- no capacity retry --, checked through
    the list invariants and, on a sample of starts, against the candidate arrays."""
    import torch
    r, eng = c2
    I = r.n_items
    S = eng.item_sim("adjust_cosine", CAP)
    E = eng.extend(S, 50)
    n0, t0, v0 = E.n_cand.clone(), E.top_end.clone(), E.top_val.clone()
    assert E.xs_end is None
    eng.extend_lists(E)
    assert np.array_equal(n0.cpu().numpy(), E.n_cand.cpu().numpy())
    assert np.array_equal(t0.cpu().numpy(), E.top_end.cpu().numpy()) and np.array_equal(v0.cpu().numpy(), E.top_val.cpu().numpy())
    n_cand = E.n_cand[:I].long()
    total = int(n_cand.sum().item())
    assert total == E.n_out and int(E.xs_end.numel()) == total            # exact allocation
    off = E.xs_off[:I]
    has = n_cand > 0
    # the lists tile the buffer: offsets are a permutation-free packing of the counts
    o = torch.sort(off[has]).indices
    so, sn = off[has][o], n_cand[has][o]
    assert int(so[0].item()) == 0 and bool((so[1:] == so[:-1] + sn[:-1]).all()) and int((so[-1] + sn[-1]).item()) == total
    flags = torch.from_numpy(r.item_attrs()[3]).to(E.xs_end.device)
    for lo in range(0, total, 1 << 28):                                   # every end is a source-side item, every value finite
        hi = min(total, lo + (1 << 28))
        assert bool((flags[E.xs_end[lo:hi].long()] & 1).all()) and bool(torch.isfinite(E.xs_val[lo:hi]).all())
    rng = np.random.default_rng(5)
    starts = np.nonzero(has.cpu().numpy())[0]
    te, tv = E.top_end.cpu().numpy(), E.top_val.cpu().numpy()
    for s in rng.choice(starts, 300, replace=False):
        a, n = int(off[s].item()), int(n_cand[s].item())
        e, v = E.xs_end[a:a + n].cpu().numpy(), E.xs_val[a:a + n].cpu().numpy()
        assert len(np.unique(e)) == n                                     # distinct ends
        order = np.lexsort((e, -np.abs(v)))[:10]
        m = min(n, 10)
        assert np.array_equal(e[order], te[s, :m]) and np.array_equal(v[order], tv[s, :m])


def test_rec_sim_full_size_against_the_oracle(c2):
    """RecommenderSim over the 5.2 M AlterEgo rows the hot path produces at BASELINE configs[1] (k = 50,
    private mapping; fp64 means, items held twice): every pair's weighted cosine, local sensitivity and count against the
    CPU oracle over the same rows, bit for bit -- the tile-sorted mirror with the sixth column, the self pairs and the
    mirrored counts taken from the COO at full size."""
    import torch
    from oracle import xmap_oracle as xo
    from xmap.engine import device, ids
    r, eng = c2
    S = eng.item_sim("adjust_cosine", CAP)
    E = eng.extend(S, 50)
    _, _, mp = eng.select(E, True)
    G = eng.alterego(mp)
    u, it, ra = G.user.cpu().numpy(), G.item.cpu().numpy(), G.rating.cpu().numpy()
    del S, E, G
    torch.cuda.empty_cache()
    o = np.argsort(u, kind="stable")
    uu, uinv = np.unique(u[o], return_inverse=True)
    ii, iinv = np.unique(it[o], return_inverse=True)
    ptr = np.zeros(len(uu) + 1, np.int64)
    np.cumsum(np.bincount(uinv, minlength=len(uu)), out=ptr[1:])
    item, rating = iinv.astype(np.int32), ra[o].astype(np.float64)
    rating[::97] /= 3.0            # means no float32 holds (the stage takes np.float64 values: build_alterEgo's np.mean results)
    assert len(item) > 4_000_000 and np.any(rating != rating.astype(np.float32).astype(np.float64))
    all_ids = r.item_ids()
    iids = [all_ids[x] for x in ii]
    R = device.DeviceRatings(ptr, item, rating, np.zeros(len(item), np.int64), len(iids), ids.item_attrs(iids), "cuda:0", rating64=True)
    e2 = device.Engine(R)
    S2 = e2.rec_sim(CAP)
    rp = S2.row_ptr.cpu().numpy()
    rows = np.repeat(np.arange(len(iids), dtype=np.int64), np.diff(rp))
    col = S2.col.cpu().numpy().astype(np.int64)
    od = np.lexsort((col, rows))
    O = xo.rec_sim(ptr, item, rating, len(iids), CAP)
    orow = np.repeat(np.arange(len(iids), dtype=np.int64), np.diff(O.row_ptr))
    assert len(rows) == len(orow) > 30_000_000
    assert np.array_equal(rows[od], orow) and np.array_equal(col[od], O.col.astype(np.int64))
    assert np.array_equal(S2.nij.cpu().numpy()[od], O.nij)
    assert np.array_equal(S2.sim.cpu().numpy()[od].view(np.uint64), O.sim.view(np.uint64))
    assert np.array_equal(S2.ls.cpu().numpy()[od].view(np.uint64), O.ls.view(np.uint64))
    assert (rows == col).any()                                           # an item held twice pairs with itself
    assert np.array_equal(S2.norm.cpu().numpy(), O.norm)
    del S2, e2, R
    torch.cuda.empty_cache()


def test_k100_full_size(c2):
    """the list length of BASELINE configs[3] (k = 100) at configs[1] size, through the size-independent properties"""
    r, eng = c2
    S = eng.item_sim("adjust_cosine", CAP)
    E = eng.extend(S, 100)
    assert E.n_paths == E.units.total
    kc = E.kcnt.cpu().numpy()
    assert kc.max() <= 100 and (kc[:, 0] > 50).any()
    _candidate_properties(r, E, r.n_items)


def test_dense_full_size():
    """BASELINE configs[4]: 200k x 200k item factors of dimension 128, k = 50 -- a row sample against the oracle (bit for
    bit: same k-ordered fp32 chain), every row through the order invariants"""
    import torch
    from oracle import xmap_oracle as xo
    from xmap.engine import synth, device
    r = synth.make_two_domain(3, 60, 30, 30, overlap=0.5)
    eng = device.Engine(device.DeviceRatings(r.user_ptr, r.item, r.rating, r.time, r.n_items, r.item_attrs()))
    n, K, k = 200000, 128, 50
    g = torch.Generator(device="cuda").manual_seed(1)
    Ft = torch.randn(n, K, device="cuda", generator=g)
    Fs = torch.randn(n, K, device="cuda", generator=g)
    idx, val = eng.dense_topk(Ft, Fs, k)
    a = val.abs()
    assert bool((idx >= 0).all()) and bool((idx < n).all())
    assert bool((a[:, :-1] >= a[:, 1:]).all())
    tie = a[:, :-1] == a[:, 1:]
    assert bool((idx[:, :-1][tie] < idx[:, 1:][tie]).all())
    rows = np.random.default_rng(2).choice(n, 48, replace=False)
    oi, ov = xo.dense_topk(xo.dense_normalize(Ft[rows].cpu().numpy()), xo.dense_normalize(Fs.cpu().numpy()), k, nthreads=8)
    assert np.array_equal(idx[rows].cpu().numpy(), oi)
    assert np.array_equal(val[rows].cpu().numpy().view(np.uint32), ov.view(np.uint32))


def test_c4_shape_full_size():
    """BASELINE configs[3]: 4 source domains -> one target, 1.25 M users per two-domain problem, k = 100, private mapping,
    through the multi-domain driver (one rank: domain after domain).  Size-independent properties of every domain's part of
    the union, and one domain recomputed on its own must give exactly its rows."""
    import torch
    from xmap.engine import synth, device, multidomain
    doms = synth.config_c4()
    n_dom = len(doms)
    cache = {}

    def make_engine(d):
        cache.clear()                       # one domain resident at a time
        torch.cuda.empty_cache()
        r = doms[d]
        cache[d] = device.Engine(device.DeviceRatings(r.user_ptr, r.item, r.rating, r.time, r.n_items, r.item_attrs()))
        return cache[d], r.n_src_items

    out = multidomain.run_multidomain(make_engine, n_dom, "adjust_cosine", CAP, 100, True)
    assert len(out["user"]) == int(out["n_rows"].sum()) and (out["n_paths"] > 0).all() and (out["n_rows"] > 0).all()
    assert np.array_equal(np.unique(out["domain"]), np.arange(n_dom))
    n_tgt = doms[0].n_items - doms[0].n_src_items
    assert (out["item"] >= 0).all() and (out["item"] < n_tgt).all()           # the common target numbering
    assert (out["rating"] >= 1).all() and (out["rating"] <= 5).all()
    for d in range(n_dom):
        sel = out["domain"] == d
        assert int(sel.sum()) == int(out["n_rows"][d])
        # at least the users' own target ratings are there: one row per target rating of the domain's users
        flags = doms[d].item_attrs()[3]
        assert int(sel.sum()) >= int((flags[doms[d].item] & 2).astype(bool).sum())
    # domain 2 on its own
    d = 2
    eng, n_src = make_engine(d)
    S = eng.item_sim("adjust_cosine", CAP)
    E = eng.extend(S, 100)
    assert E.n_paths == E.units.total == int(out["n_paths"][d])
    _candidate_properties(doms[d], E, doms[d].n_items)
    _, _, mp = eng.select(E, True)
    G = eng.alterego(mp)
    sel = out["domain"] == d
    assert np.array_equal(G.user.cpu().numpy(), out["user"][sel]) and np.array_equal(G.item.cpu().numpy() - n_src, out["item"][sel])
    assert np.array_equal(G.rating.cpu().numpy(), out["rating"][sel])
