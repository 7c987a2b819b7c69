# -*- coding: utf-8 -*-
"""RecommenderPrediction: item-based kNN prediction with temporal decay + MAE (mirror of reference
core/recommenderPrediction.py:5-139).  Evaluation stage downstream of the hot path."""
import numpy as np


class RecommenderPrediction:
    def __init__(self, alpha, method):
        self.alpha = alpha
        self.method = method

    def bound_rating(self, rating):
        """round half up, clamp to [0, 5] (reference :17-23)"""
        return 1.0 * max(0, min(int(rating + 0.5), 5))

    def _rank_by_time(self, triples):
        """(a, b, time) -> (a, b, rank): equal times share a rank, ranks start at 1 (reference :35-48)"""
        ordered = sorted(triples, key=lambda t: t[2])
        out, rank = [], 0
        for pos, t in enumerate(ordered):
            if pos == 0 or t[2] != ordered[pos - 1][2]:
                rank += 1
            out.append((t[0], t[1], rank))
        return out

    def _decayed_ratio(self, triples):
        """sum(w*x)/sum(w*y) with w = exp(-alpha (now - rank)), now = max rank + 1 (reference :50-66)"""
        ranked = self._rank_by_time(triples)
        now = max(t[2] for t in ranked) + 1
        weighted = [(t[0] * np.exp(- self.alpha * (now - t[2])), t[1] * np.exp(- self.alpha * (now - t[2])))
                    for t in ranked]
        return sum(w[0] for w in weighted) / sum(w[1] for w in weighted)

    def item_based_prediction(self, line, rating_bd, sim_bd, item_bd):
        """(uid, [(iid, real, predicted without decay, predicted with decay) | ()]) -- reference :25-105.
        rating_bd: {iid: [(uid, rating, time)*]}, sim_bd: {iid: [(iid, sim)*]}, item_bd: {iid: (avg, norm, n)}.
        Like the reference, a neighbour's rating counts when `uid in rater_id` (substring test)."""
        uid, pairs = line
        out = []
        for pair in pairs:
            iid, real = pair[0], pair[1]
            if iid not in sim_bd.value.keys():
                out.append(())
                continue
            base = item_bd.value[iid][0]
            evidence = []
            for niid, nsim in sim_bd.value[iid]:
                navg = item_bd.value[niid][0]
                for rater, rating, when in rating_bd.value[niid]:
                    if uid in rater:
                        evidence.append((nsim * (rating - navg), abs(nsim), when))
            if evidence:
                plain = base + sum(ev[0] for ev in evidence) / sum(ev[1] for ev in evidence)
                decayed = base + self._decayed_ratio(evidence)
            else:
                plain = decayed = base
            out.append((iid, real, self.bound_rating(plain), self.bound_rating(decayed)))
        return uid, out

    def item_based_recommendation(self, test_dataRDD, item_based_dict_bd, itembased_sim_pair_dict_bd, item_info_bd):
        return test_dataRDD.map(lambda line: self.item_based_prediction(
            line, item_based_dict_bd, itembased_sim_pair_dict_bd, item_info_bd))

    def calculate_mae(self, rdd):
        """'<MAE without decay>; <MAE with decay>' (or a single MAE for user-based methods) -- reference :107-139"""
        def errors(line, index):
            return [abs(p[1] - p[index]) for p in line[1] if p != ()]

        def mae(index):
            tot = rdd.map(lambda line: errors(line, index)).map(
                lambda errs: np.array([sum(errs), len(errs)])).reduce(lambda a, b: a + b)
            return tot[0] / tot[1]
        if "user" in self.method:
            return str(mae(2))
        return str(1.0 * mae(2)) + '; ' + str(mae(3))
