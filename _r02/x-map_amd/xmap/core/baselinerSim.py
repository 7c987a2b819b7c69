# -*- coding: utf-8 -*-
"""BaselinerSim: item-item similarity tool (mirror of reference core/baselinerSim.py:11-244).

Same constructor, attributes and method names as the reference class; the RDD-shaped work runs
on the MI355X through xmap.engine (no Spark shuffle, no CPU fallback)."""

from xmap.engine.localrdd import LocalRDD, LocalDF, records_of


class BaselinerSim:
    def __init__(self, method, num_atleast):
        """reference core/baselinerSim.py:12-15"""
        self.method = method
        self.num_atleast = num_atleast

    # -- A2 / A3 -------------------------------------------------------------------------
    def get_universal_user_info(self, dataRDD):
        """(uid, (average, norm2))*  -- reference :17-38"""
        from xmap.engine import session
        st = session.train_state(dataRDD)
        u_avg, u_norm, _, _, _ = st.engine.stats()
        a, n = u_avg.cpu().numpy(), u_norm.cpu().numpy()
        return LocalRDD([(uid, (float(a[k]), float(n[k]))) for k, uid in enumerate(st.idt.uids)])

    def get_universal_item_info(self, dataRDD, user_info=None):
        """(iid, (average, norm2, adjusted norm2, count))*  -- reference :40-82"""
        from xmap.engine import session
        st = session.train_state(dataRDD)
        info = st.engine.stats()[2].cpu().numpy()
        return LocalRDD([(iid, tuple(float(x) for x in info[k])) for k, iid in enumerate(st.idt.iids)])

    # -- scalar helpers (reference :84-95) --------------------------------------------------
    def significance_weighting(self, sim, count):
        return 1.0 * sim * min(count, self.num_atleast) / self.num_atleast

    def cosine(self, dot_product, norm2_product):
        return 1.0 * dot_product / (norm2_product) if norm2_product else 0.0

    # -- per-pair helpers on host records (reference :97-185).  The device computes all pairs at once
    #    (calculate_item2item_sim); these mirror the reference's public per-pair methods for callers that use them
    #    directly.  item_info / user_info: broadcast-like objects with a .value dict.
    def retrieve_path_info(self, item_pair, rating_pairs, item_info):
        """mutuality: co-raters whose two ratings lie on the same side of the two item averages -- reference :97-113"""
        avg1, avg2 = item_info.value[item_pair[0]][0], item_info.value[item_pair[1]][0]
        agree = sum(1 for r1, r2, _ in rating_pairs if (r1 >= avg1 and r2 >= avg2) or (r1 < avg1 and r2 < avg2))
        return float(agree)

    def _pair_result(self, item_pair, rating_pairs, item_info, inner_product, norm_col):
        n = len(rating_pairs)
        a, b = item_info.value[item_pair[0]], item_info.value[item_pair[1]]
        sim = self.significance_weighting(self.cosine(inner_product, a[norm_col] * b[norm_col]), n)
        mutu = self.retrieve_path_info(item_pair, rating_pairs, item_info)
        return item_pair, (sim, mutu, 1.0 * mutu / (a[3] + b[3] - n))

    def calculate_cosine_sim(self, item_pair, rating_pairs, item_info):
        """(item pair, (weighted cosine, mutu, frac_mutu)) -- reference :115-142"""
        dot = sum(1.0 * r1 * r2 for r1, r2, _ in rating_pairs)
        return self._pair_result(item_pair, rating_pairs, item_info, dot, 1)

    def calculate_adjusted_cosine_sim(self, item_pair, rating_pairs, item_info, user_info):
        """(item pair, (weighted adjusted cosine, mutu, frac_mutu)) -- reference :144-174"""
        import numpy as np
        r1 = np.array([p[0] for p in rating_pairs])
        r2 = np.array([p[1] for p in rating_pairs])
        avg = np.array([user_info.value[p[2]][0] for p in rating_pairs])
        return self._pair_result(item_pair, rating_pairs, item_info, np.sum((r1 - avg) * (r2 - avg)), 2)

    def produce_pairwise_items(self, dataRDD):
        """((iid1, iid2), [(rating1, rating2, uid)])* for both orders of every two entries of a profile -- reference
        :176-185"""
        from itertools import combinations
        out = []
        for uid, profile in records_of(dataRDD):
            if len(profile) < 2:
                continue
            for e1, e2 in combinations(profile, 2):
                out.append(((e1[0], e2[0]), [(e1[1], e2[1], uid)]))
                out.append(((e2[0], e1[0]), [(e2[1], e1[1], uid)]))
        return LocalRDD(out, getattr(dataRDD, "ctx", None))

    # -- A4-A9 --------------------------------------------------------------------------------
    def calculate_item2item_sim(self, dataRDD, item_info=None, user_info=None):
        """((iid1, iid2), (sim, mutu, frac_mutu, label))* -- reference :187-216.
        Like the reference, an unknown method yields None."""
        if self.method not in ("cosine", "adjust_cosine"):
            return None
        from xmap.engine import session
        st = session.train_state(dataRDD)
        S = st.engine.item_sim(self.method, self.num_atleast)
        return session.SimPairsRDD(st, S, getattr(dataRDD, "ctx", None))

    def get_item_sim(self, dataRDD):
        """(iid1, [(iid2, sim, mutu, frac_mutu)*])* -- reference :218-233 (one direction per record)."""
        acc = {}
        for (iid1, iid2), (sim, mutu, frac_mutu, label) in records_of(dataRDD):
            acc.setdefault(iid1, []).append((iid2, sim, mutu, frac_mutu))
        return LocalRDD(list(acc.items()), getattr(dataRDD, "ctx", None))

    def build_sim_DF(self, sim_pairsRDD):
        """reference :235-244"""
        from pyspark.sql import Row
        rows = [Row(id1=iid[0], id2=iid[1], sim=float(info[0]), mutu=info[1],
                    frac_mutu=float(info[2]), label=info[3]) for iid, info in records_of(sim_pairsRDD)]
        return LocalDF(rows, getattr(sim_pairsRDD, "ctx", None))
