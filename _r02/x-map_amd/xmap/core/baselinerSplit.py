# -*- coding: utf-8 -*-
"""BaselinerSplit: train / test protocol over the overlap users (mirror of reference
core/baselinerSplit.py:6-146).  Host-side.  Spark's randomSplit draws from a per-partition
XORShift stream that cannot be reproduced without Spark; the local RDD's randomSplit uses Python's `random`
with the same seed, so the split is statistically, not bitwise, the reference's."""
import random


class BaselinerSplit:
    def __init__(self, num_left, ratio_split, ratio_both, seed):
        """reference :7-32 (seeds Python's global RNG, like the reference)"""
        self.num_left = num_left
        self.ratio_split = ratio_split
        self.ratio_both = ratio_both
        self.seed = seed
        random.seed(seed)

    def find_overlap_user(self, sourceRDD, targetRDD):
        return sourceRDD.keys().intersection(targetRDD.keys())

    def find_overlap_user_multidomain(self, sourceRDD1, sourceRDD2, targetRDD):
        return sourceRDD1.keys().intersection(sourceRDD2.keys()).intersection(targetRDD.keys())

    def distinguish_data(self, overlap_userRDD_bd, dataRDD):
        """(overlap records, non-overlap records) -- reference :48-64"""
        overlap = set(overlap_userRDD_bd.value)
        return (dataRDD.filter(lambda rec: rec[0] in overlap),
                dataRDD.filter(lambda rec: rec[0] not in overlap))

    def determine_remaining(self, iterators):
        """(uid, source lines, kept target lines, hidden target lines) per test user: num_left target ratings stay
        in training, the rest are hidden for evaluation (reference :66-88)"""
        for uid, lines in iterators:
            source = [ln for ln in lines if "S:" in ln[0]]
            target = [ln for ln in lines if ln not in source]
            keep = random.sample(target, self.num_left)
            hidden = [ln for ln in target if ln not in keep]
            yield uid, source, keep, hidden

    def _three_way(self, unionRDD):
        return unionRDD.randomSplit(
            [self.ratio_split, self.ratio_both, 1 - self.ratio_split - self.ratio_both], seed=self.seed)

    def split_data(self, non_overlap_sourceRDD, overlap_sourceRDD, non_overlap_targetRDD, overlap_targetRDD):
        """(trainingRDD, testRDD) -- reference :90-113"""
        merged = overlap_sourceRDD.union(overlap_targetRDD).reduceByKey(lambda a, b: a + b)
        parts = self._three_way(merged)
        test_part = parts[0].mapPartitions(self.determine_remaining).cache()
        testRDD = test_part.map(lambda rec: (rec[0], rec[3]))
        trainRDD = non_overlap_sourceRDD.union(non_overlap_targetRDD).union(parts[1]).union(parts[2]).union(
            test_part.map(lambda rec: (rec[0], rec[1] + rec[2])))
        return trainRDD, testRDD

    def split_data_multipledomain(self, nonoverlap_source1RDD, overlap_source1RDD, nonoverlap_source2RDD,
                                  overlap_source2RDD, non_overlap_targetRDD, overlap_targetRDD):
        """(training source 1, training source 2, testRDD) -- reference :115-146 (the reference's version maps with
        two-argument lambdas and cannot run; this one does what it describes)"""
        merged = overlap_source1RDD.union(overlap_source2RDD).union(overlap_targetRDD).reduceByKey(lambda a, b: a + b)
        parts = self._three_way(merged)
        test_part = parts[0].mapPartitions(self.determine_remaining).cache()
        testRDD = test_part.map(lambda rec: (rec[0], rec[3]))
        overlap_train = test_part.map(lambda rec: (rec[0], rec[1] + rec[2])).union(parts[1]).union(parts[2])

        def restrict(tag):
            return overlap_train.map(
                lambda rec: (rec[0], [ln for ln in rec[1] if tag in ln[0] or "T:" in ln[0]]))
        train1 = nonoverlap_source1RDD.union(restrict("S:1:")).union(non_overlap_targetRDD)
        train2 = nonoverlap_source2RDD.union(restrict("S:2:")).union(non_overlap_targetRDD)
        return train1, train2, testRDD
