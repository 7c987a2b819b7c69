# -*- coding: utf-8 -*-
"""ExtendSim: cross-domain top-k similarity extension (mirror of reference core/extender.py:8-217)."""
import numpy as np

from xmap.engine.localrdd import LocalRDD, records_of


class ExtendSim:
    def __init__(self, top_k):
        """reference core/extender.py:9-14"""
        self.top_k = top_k

    def extend(self, state, S, full=False):
        """B1-B6 on the device: bridge flags, knn classification, reverse adjacencies, streamed path
        enumeration with fused top-10 (reference find_knn_items + sim_extend + get_final_extension)."""
        return state.engine.extend(S, self.top_k, full=full)

    def find_knn_items(self, rdd, BB_items_bd):
        """(iid, (BB_BB, BB_NB), None) | (iid, None, (NB_BB, NB_NN)) records -- reference :16-44.
        rdd: (iid, [(iid2, sim, mutu, frac_mutu)*])*; BB_items_bd.value: list of bridge item ids."""
        from xmap.engine import session, ids as xids, device
        recs = records_of(rdd)
        pairs = [((i, j), (s, m, f, 0)) for i, lst in recs for (j, s, m, f) in lst]
        st = _items_state(sorted({i for i, _ in recs} | {j for _, lst in recs for (j, _, _, _) in lst}
                                 | set(BB_items_bd.value)))
        S = session.sim_from_records(st, pairs)
        import torch
        bb = np.zeros(len(st.idt.iids), np.uint8)
        for b in BB_items_bd.value:
            bb[st.idt.iidx[b]] = 1
        E = st.engine.knn(S, self.top_k, bb=torch.from_numpy(bb).to(st.engine.dev))
        cls = E.cls.cpu().numpy()
        kcnt, kcol, kval = E.kcnt.cpu().numpy(), E.kcol.cpu().numpy(), E.kval.cpu().numpy()
        iids = st.idt.iids
        out = []
        for i, _ in recs:
            a = st.idt.iidx[i]
            if cls[a] == 0:
                continue
            lists = []
            for l in (0, 1):
                lists.append([(iids[kcol[a, l, q]], float(kval[a, l, q, 0]), float(kval[a, l, q, 1]),
                               float(kval[a, l, q, 2])) for q in range(kcnt[a, l])])
            out.append((i, tuple(lists), None) if cls[a] == 1 else (i, None, tuple(lists)))
        return LocalRDD(out, getattr(rdd, "ctx", None))

    def sim_extend(self, BB_info, NB_info, knn_BB_bd, knn_NB_bd):
        """cross-domain path extension over the classified lists -- reference :46-182.  The reference materialises one
        record per path; here the records stay implicit: the returned handle holds the knn tables and reverse
        adjacencies in HBM, and get_final_extension runs the enumeration.
        BB_info: (bridge iid, (BB_BB, BB_NB))*, NB_info: (non-bridge iid, (NB_BB, NB_NN))* as find_knn_items /
        extract_siminfo produce them (the two broadcasts repeat that information and are not needed)."""
        bb, nb = records_of(BB_info), records_of(NB_info)
        ids = {i for i, _ in bb} | {i for i, _ in nb}
        for _, lists in list(bb) + list(nb):
            for lst in lists:
                ids.update(e[0] for e in lst)
        st = _items_state(sorted(ids))
        I, k = len(st.idt.iids), int(self.top_k)
        cls = np.zeros(I, np.uint8)
        kcnt = np.zeros((I, 2), np.int32)
        kcol = np.zeros((I, 2, k), np.int32)
        kval = np.zeros((I, 2, k, 3), np.float64)
        for recs, c in ((bb, 1), (nb, 2)):
            for iid, lists in recs:
                a = st.idt.iidx[iid]
                cls[a] = c
                for l, lst in enumerate(lists):
                    if len(lst) > k:
                        raise ValueError("a neighbour list is longer than top_k")
                    kcnt[a, l] = len(lst)
                    for q, e in enumerate(lst):
                        kcol[a, l, q] = st.idt.iidx[e[0]]
                        kval[a, l, q] = e[1:4]
        E = st.engine.ext_tables_from_knn(k, cls, kcnt, kcol, kval)
        return CrossExtendedHandle(st, E, getattr(BB_info, "ctx", None))

    def get_final_extension(self, cross_extended):
        """(start iid, [(end iid, xsim)*])* with xsim = sum(s_p c_p) / sum(c_p) over all paths of a pair -- reference
        :184-217.  Returns the lazy extended_simRDD (see xmap.engine.session.ExtendedSimRDD)."""
        from xmap.engine import session
        if not isinstance(cross_extended, CrossExtendedHandle):
            raise TypeError("get_final_extension expects the handle returned by sim_extend")
        st = cross_extended.state
        E = st.engine.extend_tables(cross_extended.E, full=False)
        return session.ExtendedSimRDD(st, E, cross_extended.ctx)


class CrossExtendedHandle(object):
    """what sim_extend returns: the stage-B tables of one pass in HBM (the reference's RDD of per-path records is
    never built)"""

    def __init__(self, state, E, ctx=None):
        self.state, self.E, self.ctx = state, E, ctx

    def cache(self):
        return self


class _ItemsState(object):
    pass


def _items_state(iids_sorted):
    """Engine over an item dictionary only (no ratings): used when a stage is fed generic records."""
    from xmap.engine import ids as xids, device
    st = _ItemsState()
    st.idt = xids.IdTable([], iids_sorted)
    R = device.DeviceRatings(np.zeros(1, np.int64), np.zeros(0, np.int32), np.zeros(0, np.float32),
                             np.zeros(0, np.int64), len(iids_sorted), st.idt.attrs)
    st.R = R
    st.engine = device.Engine(R)
    st.times, st.ratings = [], []
    return st
