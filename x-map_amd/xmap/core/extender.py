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
        """The reference materialises one record per path here (:46-182); this engine streams the
        paths inside extender_pipeline and never builds that RDD."""
        raise NotImplementedError("paths are enumerated on the GPU inside xmap.utils.assist.extender_pipeline")

    def get_final_extension(self, cross_extended):
        raise NotImplementedError("see xmap.utils.assist.extender_pipeline")


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
