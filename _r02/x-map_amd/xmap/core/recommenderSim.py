# -*- coding: utf-8 -*-
"""RecommenderSim: second item-item similarity over the AlterEgo profile, with per-pair local sensitivity
(mirror of reference core/recommenderSim.py:9-195).  calculate_sim runs on the GPU
(Engine.rec_sim: the stage-A pair machinery with a second walk for the leave-one-out local sensitivity); the
profile / info helpers around it are host-side Python over RDD-like objects, as in the reference.  cosine_sim and
adjusted_cosine_sim are kept as the readable per-pair statement of what the kernel computes (used by the CPU tests)."""
from itertools import combinations

import numpy as np


class RecommenderSim:
    def __init__(self, method, num_atleast):
        self.method = method
        self.num_atleast = num_atleast

    def build_sthbased_profile(self, rdd, profile):
        """rows (uid, iid, rating, time) -> (uid, [(iid, rating, time)*]) or (iid, [(uid, rating, time)*])
        (reference :15-27)"""
        if "user" in profile:
            return rdd.map(lambda l: (l[0], [(l[1], l[2], l[3])])).reduceByKey(lambda a, b: a + b)
        elif "item" in profile:
            return rdd.map(lambda l: (l[1], [(l[0], l[2], l[3])])).reduceByKey(lambda a, b: a + b)

    def get_info(self, dataRDD):
        """(id, (average, norm2, count)) -- reference :29-63"""
        def describe(rec):
            key, ratings = rec
            values = [r[1] for r in ratings]
            return key, (1.0 * np.average(values), np.sqrt(np.sum([v ** 2 for v in values])), len(ratings))
        return dataRDD.map(describe)

    def produce_pairwise(self, dataRDD):
        """((id1, id2), [(r1, r2, owner)*]) for both directions of every co-occurring pair (reference :65-76)"""
        def expand(records):
            for owner, ratings in records:
                for a, b in combinations(ratings, 2):
                    yield (a[0], b[0]), [(a[1], b[1], owner)]
                    yield (b[0], a[0]), [(b[1], a[1], owner)]
        return dataRDD.filter(lambda rec: len(rec[1]) >= 2).mapPartitions(expand).reduceByKey(lambda x, y: x + y)

    def significance_weighting(self, sim, count):
        return 1.0 * sim * min(count, self.num_atleast) / self.num_atleast

    def cosine(self, dot_product, norm2_product):
        return 1.0 * dot_product / (norm2_product) if norm2_product else 0.0

    def cosine_sim(self, line, info):
        """((id1, id2), [weighted cosine, local sensitivity]) -- reference :90-133.  The local sensitivity is the
        largest change of the weighted cosine when one co-rating is removed from either side."""
        (id1, id2), rating_pairs = line
        n = len(rating_pairs)
        prods = [(rp[0], rp[1], rp[0] * rp[1]) for rp in rating_pairs]
        inner = sum(p[2] for p in prods)
        norm_x, norm_y = info.value[id1][1], info.value[id2][1]
        sim = self.significance_weighting(self.cosine(inner, norm_x * norm_y), n)
        variants = []
        for r0, r1, r01 in prods:
            rest = inner - r0 * r1
            without_x = np.sqrt((norm_x ** 2 - r0 ** 2) * (norm_y ** 2))
            without_y = np.sqrt((norm_x ** 2) * (norm_y ** 2 - r1 ** 2))
            variants.append(self.significance_weighting(self.cosine(rest, without_x), n - 1))
            variants.append(self.significance_weighting(self.cosine(rest, without_y), n - 1))
        return (id1, id2), [sim, max(abs(np.array(variants) - sim))]

    def adjusted_cosine_sim(self, line, info):
        """((id1, id2), [weighted adjusted cosine, local sensitivity]) -- reference :135-186.  As in the reference
        the leave-one-out loop zeroes the entries IN PLACE (the arrays are aliased), so removals accumulate."""
        (id1, id2), rating_pairs = line
        n = len(rating_pairs)
        rx = np.array([rp[0] for rp in rating_pairs])
        ry = np.array([rp[1] for rp in rating_pairs])
        avg = np.array([info.value[rp[2]][0] for rp in rating_pairs])
        inner = np.sum((rx - avg) * (ry - avg))
        sim = self.significance_weighting(
            self.cosine(inner, np.sqrt(np.sum((rx - avg) ** 2)) * np.sqrt(np.sum((ry - avg) ** 2))), n)
        variants = []
        for i in range(len(rx)):
            avg[i], rx[i], ry[i] = 0, 0, 0
            mx = np.sqrt(np.sum((rx - avg) ** 2))
            my = np.sqrt(np.sum((ry - avg) ** 2))
            variants.append(self.significance_weighting(
                self.cosine(sum((rx - avg) * (ry - avg)), mx * my), n - 1))
        return (id1, id2), [sim, max(abs(np.array(variants) - sim))]

    def calculate_sim(self, item_profile, user_profile, item_info, user_info):
        """reference :188-195.  `"cosine_item" in "adjust_cosine_item"` is true, so (as in the reference) both
        method names take the cosine branch."""
        if "cosine_item" in self.method:
            from ..engine import session      # raises if libxmap_hip.so is missing: no CPU fallback
            return session.rec_sim_from_profiles(user_profile, self.num_atleast, getattr(user_profile, "ctx", None))
        elif "adjust_cosine_item" in self.method:
            return self.produce_pairwise(user_profile).map(lambda line: self.adjusted_cosine_sim(line, user_info))

    def calculate_sim_host(self, item_profile, user_profile, item_info, user_info):
        """the per-pair Python statement of calculate_sim (reference :188-195), for tests without a GPU"""
        return self.produce_pairwise(user_profile).map(lambda line: self.cosine_sim(line, item_info))
