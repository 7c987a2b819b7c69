"""Host-side plumbing between the reference's record formats and the device engine.

  trainRDD records   (uid, [(iid, rating, time)*])           -> DeviceRatings (index space, HBM)
  item2item_simRDD   ((iid1, iid2), (sim, mutu, frac, label)) <-> SimResult
  extended_simRDD    (start_iid, [(end_iid, xsim)*])          <-> ExtResult
  alterEgo_profile   (uid, iid, rating, time)                 <-  GenResult

The id dictionary built from a trainRDD is shared by the three stages; engines are
cached per trainRDD object so that generator_pipeline reuses the one stage A uploaded.
"""
import weakref

import numpy as np

from . import ids as xids
from .localrdd import LocalRDD, records_of

_engines = {}   # id(trainRDD) -> (weakref or None, TrainState, fingerprint or None)


class TrainState(object):
    def __init__(self, records):
        from . import device
        self.idt = xids.IdTable.from_records(records)
        iidx = self.idt.iidx
        n = sum(len(p) for _, p in records)
        ptr = np.zeros(len(records) + 1, np.int64)
        item = np.empty(n, np.int32)
        rating = np.empty(n, np.float32)
        self.times = []          # original time objects, device carries their position
        self.ratings = []        # original rating objects (pass-through rows keep them)
        e = 0
        for u, (_, prof) in enumerate(records):
            for (iid, r, t) in prof:
                item[e] = iidx[iid]
                rating[e] = r
                self.times.append(t)
                self.ratings.append(r)
                e += 1
            ptr[u + 1] = e
        self.R = device.DeviceRatings(ptr, item, rating, np.arange(n, dtype=np.int64), len(self.idt.iids),
                                      self.idt.attrs)
        self.engine = device.Engine(self.R)


def _fingerprint(records):
    """content key of a record list that cannot be weak-referenced: every (uid, iid, rating, time) takes part, so ratings edited
    in place -- or a new list of the same shape behind a recycled id(), such as a CV fold with perturbed ratings -- never
    hit the cache of the previous upload.  O(ratings) of host work per call, paid by plain-list inputs only (an RDD
    object is keyed by identity through a weak reference)."""
    h, n = 1469598103934665603, 0
    for rec in records:
        uid, prof = rec[0], rec[1]
        ph = hash((uid, len(prof), tuple((e[0], float(e[1]), e[2]) for e in prof)))      # (times too: AlterEgo rows and the decay read them)
        h = ((h ^ (ph & 0xFFFFFFFFFFFFFFFF)) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        n += len(prof)
    return (len(records), n, h)


def train_state(trainRDD):
    """The TrainState (id dictionary + ratings in HBM + engine) of a trainRDD, shared by the three stages.  Cached per
    trainRDD object through a weak reference; inputs that cannot be weak-referenced (plain lists) are keyed on a content
    fingerprint and only the most recent one is kept, so a recycled id() can neither leak the previous engine nor be
    mistaken for it.  release() drops everything."""
    key = id(trainRDD)
    hit = _engines.get(key)
    if hit is not None:
        ref, st, fp = hit
        if ref is not None and ref() is trainRDD:
            return st
        if ref is None and fp == _fingerprint(records_of(trainRDD)):
            return st
        _engines.pop(key, None)
    feed = getattr(trainRDD, "feed", None)
    if feed is not None and getattr(trainRDD, "_items", None) is None:
        from . import feeder                       # native records (xmap.engine.feeder.FeedRDD): no Python loop over them
        recs = None
        st = feeder.train_state_from_feed(feed)
    else:
        recs = records_of(trainRDD)
        st = TrainState(recs)
    try:
        ref = weakref.ref(trainRDD, lambda _r, k=key: _engines.pop(k, None))
        fp = None
    except TypeError:
        ref, fp = None, _fingerprint(recs)
        for k in [k for k, v in _engines.items() if v[0] is None]:      # one strong entry at most
            _engines.pop(k, None)
    _engines[key] = (ref, st, fp)
    return st


def release(trainRDD=None):
    """drop the cached engine of one trainRDD (or of all): frees its HBM buffers and accumulator scratch"""
    if trainRDD is None:
        _engines.clear()
    else:
        _engines.pop(id(trainRDD), None)


# ---------------------------------------------------------------------------------------------
class SimPairsRDD(LocalRDD):
    """item2item_simRDD: ((iid1, iid2), (sim, mutu, frac_mutu, label)) -- rows live in HBM."""

    def __init__(self, state, S, ctx=None):
        LocalRDD.__init__(self, None, ctx)
        self.state, self.S = state, S

    def _rows(self):
        S, idt = self.S, self.state.idt
        row_ptr = S.row_ptr.cpu().numpy()
        rows = np.repeat(np.arange(len(row_ptr) - 1), np.diff(row_ptr))
        col = S.col.cpu().numpy()
        o = np.lexsort((col, rows))        # canonical order: (iid1, iid2) ascending
        rows, col = rows[o], col[o]
        sim = S.sim.cpu().numpy()[o]
        mutu = S.mutu.cpu().numpy()[o].astype(np.float64)
        info = S.info.cpu().numpy()
        frac = mutu / (info[rows, 3] + info[col, 3] - S.nij.cpu().numpy()[o])
        pre = self.state.idt.attrs[0]
        label = (pre[rows] != pre[col]).astype(int)
        iids = idt.iids
        return [((iids[a], iids[b]), (float(s), float(m), float(f), int(lab)))
                for a, b, s, m, f, lab in zip(rows, col, sim, mutu, frac, label)]


class RecSimRDD(LocalRDD):
    """alterEgo_sim of recommender_calculate_sim_pipeline: ((iid1, iid2), [sim, local sensitivity]) -- rows live in
    HBM (Engine.rec_sim); both directions of every pair, an item paired with itself once."""

    def __init__(self, S, iids, ctx=None, engine=None):
        LocalRDD.__init__(self, None, ctx)
        self.S, self.iids, self.engine = S, iids, engine

    def select_neighbors(self, keep):
        """nonprivate_neighbor_selection on the device: [(iid, [(nid, [sim, ls])*])*], items in id order"""
        cnt, col, sim, ls = [x.cpu().numpy() for x in self.engine.rec_select(self.S, int(keep))]
        iids = self.iids
        return [(iids[i], [(iids[col[i, t]], [float(sim[i, t]), float(ls[i, t])]) for t in range(cnt[i])])
                for i in range(len(iids)) if cnt[i]]

    def _rows(self):
        S, iids = self.S, self.iids
        row_ptr = S.row_ptr.cpu().numpy()
        rows = np.repeat(np.arange(len(row_ptr) - 1), np.diff(row_ptr))
        col = S.col.cpu().numpy()
        o = np.lexsort((col, rows))        # canonical order: (iid1, iid2) ascending
        sim, ls = S.sim.cpu().numpy()[o], S.ls.cpu().numpy()[o]
        return [((iids[a], iids[b]), [float(s), float(l)]) for a, b, s, l in zip(rows[o], col[o], sim, ls)]


def rec_sim_from_profiles(user_profiles, cap, ctx=None):
    """user_profiles: [(uid, [(iid, rating, time)*])*] (the user-based AlterEgo profile).  Builds the index space
    (items in lexicographic id order), uploads the CSR and runs Engine.rec_sim."""
    import torch
    from . import device
    recs = records_of(user_profiles)
    iids = sorted({t[0] for _, prof in recs for t in prof})
    iidx = {s: k for k, s in enumerate(iids)}
    ptr = np.zeros(len(recs) + 1, np.int64)
    item, rating = [], []
    for k, (_, prof) in enumerate(recs):
        ptr[k + 1] = ptr[k] + len(prof)
        item.extend(iidx[t[0]] for t in prof)
        rating.extend(float(t[1]) for t in prof)
    dev = "cuda:%d" % torch.cuda.current_device()
    # fp64 all the way: AlterEgo ratings are np.float64 means (reference core/generator.py:123-138), and that is what
    # RecommenderSim multiplies (core/recommenderSim.py:64-133)
    R = device.DeviceRatings(ptr, np.asarray(item, np.int32), np.asarray(rating, np.float64),
                             np.zeros(len(item), np.int64), len(iids), xids.item_attrs(iids), dev, rating64=True)
    eng = device.Engine(R)
    S = eng.rec_sim(cap)
    return RecSimRDD(S, iids, ctx, eng)


class ExtendedSimRDD(LocalRDD):
    """extended_simRDD: (start_iid, [(end_iid, xsim)*]) -- a LAZY handle.  The pass behind it keeps, per start item, the
    number of candidates and the XMAP_TOPC best by |xsim| (all a Generator reads: generator.py:85,109) in HBM.  The full
    lists (4.6e9 pairs at BASELINE configs[1]) are only produced when somebody iterates / collects this RDD: the
    enumeration then runs once more with list buffers sized exactly from the candidate counts."""

    def __init__(self, state, E, ctx=None):
        LocalRDD.__init__(self, None, ctx)
        self.state, self.E = state, E

    @property
    def materialised(self):
        return getattr(self.E, "xs_end", None) is not None

    def _rows(self):
        self.state.engine.extend_lists(self.E)
        E, iids = self.E, self.state.idt.iids
        I = len(iids)
        n_cand = E.n_cand.cpu().numpy()[:I]
        off = E.xs_off.cpu().numpy()[:I]
        xe, xv = E.xs_end.cpu().numpy(), E.xs_val.cpu().numpy()
        out = []
        for s in np.nonzero(n_cand)[0]:
            e = xe[off[s]:off[s] + n_cand[s]]
            v = xv[off[s]:off[s] + n_cand[s]]
            o = np.argsort(e)               # canonical order: end id ascending
            out.append((iids[s], [(iids[j], float(x)) for j, x in zip(e[o], v[o])]))
        return out


class AlterEgoRDD(LocalRDD):
    """alterEgo_profile: (uid, iid, rating, time) rows -- generator.py:140-157."""

    def __init__(self, state, G, ctx=None):
        LocalRDD.__init__(self, None, ctx)
        self.state, self.G = state, G

    def _rows(self):
        G, st = self.G, self.state
        u = G.user.cpu().numpy()
        it = G.item.cpu().numpy()
        r = G.rating.cpu().numpy()
        pos = G.time.cpu().numpy()          # position of the source row in trainRDD order
        nt = G.n_target_rows
        uids, iids = st.idt.uids, st.idt.iids
        out = []
        for q in range(len(u)):
            # pass-through rows keep the caller's rating object; AlterEgo rows carry the mean
            rating = st.ratings[pos[q]] if q < nt else np.float64(r[q])
            out.append((uids[u[q]], iids[it[q]], rating, st.times[pos[q]]))
        return out


# ---------------------------------------------------------------------------------------------
def sim_from_records(state, records):
    """Device SimResult from generic ((iid1,iid2),(sim,mutu,frac,label)) records (any order)."""
    iidx = state.idt.iidx
    I = len(state.idt.iids)
    n = len(records)
    a = np.fromiter((iidx[k[0]] for k, _ in records), np.int64, n)
    b = np.fromiter((iidx[k[1]] for k, _ in records), np.int32, n)
    sim = np.fromiter((v[0] for _, v in records), np.float64, n)
    mutu = np.fromiter((v[1] for _, v in records), np.float64, n)
    frac = np.fromiter((v[2] for _, v in records), np.float64, n)
    o = np.lexsort((b, a))
    row_ptr = np.zeros(I + 1, np.int64)
    np.cumsum(np.bincount(a, minlength=I), out=row_ptr[1:])
    info = np.zeros((I, 4))
    return state.engine.sim_from_host(row_ptr, b[o], sim[o], mutu[o].astype(np.int32), None, info, frac=frac[o])


def ext_from_records(state, records):
    """Device candidate arrays from generic (start, [(end, xsim)*]) records."""
    import torch
    from . import device, hipabi as abi
    eng = state.engine
    iidx = state.idt.iidx
    I = len(state.idt.iids)
    st = np.fromiter((iidx[s] for s, lst in records for _ in lst), np.int64)
    en = np.fromiter((iidx[e] for _, lst in records for (e, _) in lst), np.int32)
    va = np.fromiter((v for _, lst in records for (_, v) in lst), np.float64)
    o = np.lexsort((en, st))
    xs_ptr = np.zeros(I + 1, np.int64)
    np.cumsum(np.bincount(st, minlength=I), out=xs_ptr[1:])
    d = eng.dev
    E = device.ExtResult()
    pad = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a if len(a) else np.zeros(1), dt)).to(d)
    E.xs_ptr = torch.from_numpy(xs_ptr).to(d)
    E.xs_off = E.xs_ptr[:-1].contiguous()
    E.xs_end, E.xs_val = pad(en[o], np.int32), pad(va[o], np.float64)
    E.n_cand = torch.zeros(max(I, 1), dtype=torch.int32, device=d)
    E.top_end = torch.full((max(I, 1), abi.TOPC), -1, dtype=torch.int32, device=d)
    E.top_val = torch.zeros((max(I, 1), abi.TOPC), dtype=torch.float64, device=d)
    abi.check(abi.lib.xmap_topc_from_lists(device._stream(d), abi.i32(I), abi.vp(E.xs_ptr), abi.vp(E.xs_end),
                                           abi.vp(E.xs_val), abi.vp(E.n_cand), abi.vp(E.top_end), abi.vp(E.top_val)))
    return E
