# -*- coding: utf-8 -*-
"""RecommenderPrivacy: (private) neighbour selection + (Laplace) perturbation on the AlterEgo similarities
(mirror of reference core/recommenderPrivacy.py:9-189).  Downstream of the hot path."""
from math import log, e

import numpy as np


class RecommenderPrivacy:
    def __init__(self, mapping_range, privacy_epsilon, rpo):
        self.mapping_range = mapping_range
        self.privacy_epsilon = privacy_epsilon / 2      # half for selection, half for perturbation (reference :19)
        self.rpo = rpo

    def find_neighbor(self, dataRDD):
        """((id1, id2), [sim, LS])* -> (id1, [(id2, [sim, LS])*]) (one direction per record, reference :22-35)"""
        return dataRDD.map(lambda rec: (rec[0][0], [(rec[0][1], rec[1])])).reduceByKey(lambda a, b: a + b)

    # -- private selection (reference :37-139) ---------------------------------------------------
    def prepare_private_selection(self, lines, max_replacement_selection):
        k = self.mapping_range
        k_sim = lines[k - 1][1][0] if len(lines) >= k else lines[-1][1][0]
        count = len(lines)
        if count > k:
            w = min(k_sim, (2 * k * max_replacement_selection / self.privacy_epsilon) *
                    log(k * (count - k) / self.rpo, e))
        else:
            w = k_sim
        inside = [ln for ln in lines if ln[1][0] >= k_sim - w]
        outside = [ln for ln in lines if ln not in inside]
        truncated = [(ln[0], [max(ln[1][0], ln[1][0] - w), ln[1][1]]) for ln in lines]
        return [inside, outside], truncated

    def get_private_neighbor(self, lines):
        """exponential-mechanism pick among the neighbours sorted by |sim| (reference :80-139).  Under Python 3
        `np.count_nonzero(map(...))` is 1, so at most one neighbour is drawn unless mapping_range is 1."""
        lines = sorted(lines, key=lambda x: - abs(x[1][0]))
        sens = [(ln[0], ln[1][1]) for ln in lines]
        max_sens = sorted(sens, key=lambda x: - abs(x[1]))[0][1]
        _, truncated = self.prepare_private_selection(lines, max_sens)
        weights = [1.0 * np.exp(self.privacy_epsilon * t[1][0] / (2 * self.mapping_range * s[1]))
                   for t, s in zip(truncated, sens)]
        total = sum(weights)
        probs = [wgt / total for wgt in weights]
        nnz = np.count_nonzero(map(lambda p: p, probs))
        n_picks = self.mapping_range if nnz >= self.mapping_range else nnz
        return map(lambda ind: lines[ind], self._weighted_pick(probs, n_picks))

    def _weighted_pick(self, weights, n_picks):
        cum = np.cumsum(weights)
        total = np.sum(weights)
        picked = list(set(np.searchsorted(cum, np.random.rand(n_picks) * total)))
        missing = n_picks - len(picked)
        return picked + (self._weighted_pick(weights, missing) if missing else [])

    def private_neighbor_selection(self, rdd):
        return self.find_neighbor(rdd).map(lambda rec: (rec[0], self.get_private_neighbor(rec[1])))

    # -- non-private selection (reference :141-152) -------------------------------------------------
    def get_nonprivate_neighbor(self, pairs):
        return sorted(pairs, key=lambda x: - abs(x[1][0]))[: self.mapping_range]

    def nonprivate_neighbor_selection(self, rdd):
        """the mapping_range most similar neighbours per item.  On the RecommenderSim handle this is a per-row top-k
        on the GPU (Engine.rec_select; ties in ascending neighbour id); on any other RDD the reference's Python."""
        if hasattr(rdd, "select_neighbors") and self.mapping_range <= 64:
            from ..engine.localrdd import LocalRDD
            return LocalRDD(rdd.select_neighbors(self.mapping_range), getattr(rdd, "ctx", None))
        return self.find_neighbor(rdd).map(lambda rec: (rec[0], self.get_nonprivate_neighbor(rec[1])))

    # -- perturbation (reference :154-189) ------------------------------------------------------------
    def noise_perturbation(self, rdd):
        """sim + Laplace(0, |LS| / epsilon) per selected neighbour"""
        def perturb(rec):
            key, pairs = rec
            return key, [(nid, info[0] + np.random.laplace(0, abs(info[1]) / self.privacy_epsilon))
                         for nid, info in pairs]
        return rdd.map(perturb)

    def nonnoise_perturbation(self, rdd):
        return rdd.map(lambda rec: (rec[0], [(nid, info[0]) for nid, info in rec[1]]))
