# -*- coding: utf-8 -*-
"""BaselinerClean: parse / filter / clean the raw `uid iid rating unix_ts` lines (mirror of reference
core/baselinerClean.py:7-97).  Host-side text ETL upstream of the hot path; plain Python
over RDD-like objects, nothing here runs on the GPU."""
import re
from datetime import datetime

_WS = re.compile(r"\s+")


class BaselinerClean:
    def __init__(self, num_atleast_rating, size_subset, date_from, date_to, domain_label):
        """reference :8-29"""
        self.num_atleast_rating = num_atleast_rating
        self.size_subset = size_subset
        self.period = range(date_from, date_to + 1)
        self.label = domain_label

    def parse_time(self, s):
        """unix timestamp (string) -> local-time datetime (reference :31-38)"""
        return datetime.fromtimestamp(float(s))

    def parse_line(self, iterators):
        """(uid, (iid + label, rating, datetime)) for the lines whose year lies in the period (reference :40-54)"""
        for text in iterators:
            tok = _WS.split(text)
            when = self.parse_time(tok[3])
            if when.year in self.period:
                yield tok[0], (tok[1] + self.label, float(tok[2]), when)

    def parse_data(self, originalRDD):
        return originalRDD.mapPartitions(self.parse_line)

    def take_partial_data(self, dataRDD):
        """first `size_subset` cleaned profiles (reference :60-62)"""
        return dataRDD.take(self.size_subset)

    def remove_invalid(self, iterators):
        """per user keep one rating per item: the latest one (strictly later wins), first-seen item order
        (reference :64-87)"""
        for uid, ratings in iterators:
            latest = {}
            for entry in ratings:
                seen = latest.get(entry[0])
                if seen is None or entry[2] > seen[2]:
                    latest[entry[0]] = entry
            yield uid, list(latest.values())

    def filter_data(self, dataRDD):
        """group by user, drop duplicate / superseded ratings (reference :89-96)"""
        grouped = dataRDD.aggregateByKey([], lambda acc, v: acc + [v], lambda a, b: a + b)
        return grouped.mapPartitions(self.remove_invalid)

    def clean_data(self, filteredRDD):
        """drop users with fewer than num_atleast_rating ratings (reference :98-101)"""
        return filteredRDD.filter(lambda rec: len(rec[1]) >= self.num_atleast_rating)
