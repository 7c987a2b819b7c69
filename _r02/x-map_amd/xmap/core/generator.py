# -*- coding: utf-8 -*-
"""Generator: replacement selection + AlterEgo profiles (mirror of reference core/generator.py:6-157)."""
import numpy as np

from xmap.engine.localrdd import LocalRDD, records_of


class Generator:
    def __init__(self, mapping_range, privacy_epsilon, sim_method, rpo):
        """reference core/generator.py:7-19"""
        self.mapping_range = mapping_range
        self.privacy_epsilon = privacy_epsilon
        self.sim_method = sim_method
        self.rpo = rpo

    def global_sentivity(self):
        """reference :21-25"""
        return 1 if self.sim_method == "cosine" else 2

    # -- C2 / C3 on the device ---------------------------------------------------------------
    def select(self, state, E, private):
        """Per start item one replacement + the {source: target} map array (device tensors).
        private    : arg-max |xsim| (what cross_private_mapping degenerates to under Python 3)
        non-private: top4[np.random.randint(0, len(top)-1)], drawn here from the GLOBAL NumPy RNG in
                     ascending start order; a singleton candidate list raises ValueError (generator.py:110)."""
        from xmap.engine import device
        eng = state.engine
        picks = None
        if not private:
            n_top, _, _ = eng.select(E, False, None)
            picks = device.draw_picks(n_top.cpu().numpy()[:len(state.idt.iids)])
        return eng.select(E, private, picks)

    def _mapping(self, rdd, private):
        from xmap.engine import session
        from xmap.core.extender import _items_state
        if isinstance(rdd, session.ExtendedSimRDD):
            st, E = rdd.state, rdd.E
        else:
            recs = records_of(rdd)
            st = _items_state(sorted({s for s, _ in recs} | {e for _, lst in recs for (e, _) in lst}))
            E = session.ext_from_records(st, recs)
        n_top, choice, _ = self.select(st, E, private)
        n_top, choice = n_top.cpu().numpy(), choice.cpu().numpy()
        iids = st.idt.iids
        return LocalRDD([(iids[s], np.str_(iids[choice[s]])) for s in np.nonzero(n_top[:len(iids)])[0]],
                        getattr(rdd, "ctx", None))

    def cross_private_mapping(self, rdd):
        """(start iid, chosen iid)* -- reference :27-98"""
        return self._mapping(rdd, True)

    def cross_nonprivate_mapping(self, rdd, topn=4):
        """(start iid, chosen iid)* -- reference :100-111 (topn is fixed at 4 on the device)"""
        if topn != 4:
            raise NotImplementedError("topn != 4")
        return self._mapping(rdd, False)

    def mapping_item(self, line, mapping_dict):
        """reference :113-121"""
        return (line[0], mapping_dict[line[1]], line[2], line[3]) if line[1] in mapping_dict else None

    def build_alterEgo(self, trainRDD, mapping_dict):
        """(uid, iid, rating, time)* -- reference :140-157; mapping_dict = {source item: target item}."""
        from xmap.engine import session
        import torch
        st = session.train_state(trainRDD)
        mp = np.full(max(len(st.idt.iids), 1), -1, np.int32)
        for src, tgt in mapping_dict.items():
            if src in st.idt.iidx:
                if str(tgt) not in st.idt.iidx:
                    raise KeyError(tgt)
                mp[st.idt.iidx[src]] = st.idt.iidx[str(tgt)]
        G = st.engine.alterego(torch.from_numpy(mp).to(st.engine.dev))
        return session.AlterEgoRDD(st, G, getattr(trainRDD, "ctx", None))
