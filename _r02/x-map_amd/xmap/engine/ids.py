"""String id <-> dense index dictionary and the per-item string predicates.

All of the reference's hot-path logic runs on id *strings* (SURVEY.md A.1): the
cross-domain label compares `iid[:2]` (core/baselinerSim.py:189-191), the knn
classification tests `iid[-2:] in other_iid` (core/extender.py:29-35) and the
path/AlterEgo stages test `"S:" in iid` / `"T:" in iid` (core/extender.py:68,79,
174-175; core/generator.py:157).  The engine evaluates those predicates once per
item on the host and ships them as small integer arrays; items are indexed in
lexicographic id order so that index order == the canonical tie-break order of
SURVEY.md Appendix B.
"""
import numpy as np


def item_attrs(iids):
    """iids: sequence of id strings (already in index order).
    Returns prefix_cls int32[I], suffix_cls int32[I], contains_mask uint32[I], flags uint8[I]."""
    n = len(iids)
    pre = {}
    suf = {}
    prefix_cls = np.empty(n, np.int32)
    suffix_cls = np.empty(n, np.int32)
    for k, s in enumerate(iids):
        prefix_cls[k] = pre.setdefault(s[:2], len(pre))
        suffix_cls[k] = suf.setdefault(s[-2:], len(suf))
    if len(suf) > 32:
        raise ValueError("more than 32 distinct 2-char id suffixes (domain labels): %d" % len(suf))
    labels = sorted(suf.items(), key=lambda kv: kv[1])
    contains_mask = np.zeros(n, np.uint32)
    flags = np.zeros(n, np.uint8)
    for k, s in enumerate(iids):
        m = 0
        for lab, c in labels:
            if lab in s:
                m |= (1 << c)
        contains_mask[k] = m
        flags[k] = (1 if "S:" in s else 0) | (2 if "T:" in s else 0)
    return prefix_cls, suffix_cls, contains_mask, flags


class IdTable(object):
    """Lexicographically sorted item ids + user ids in trainRDD order."""

    def __init__(self, uids, iids_sorted):
        self.uids = list(uids)
        self.iids = list(iids_sorted)
        self.uidx = {s: k for k, s in enumerate(self.uids)}
        self.iidx = {s: k for k, s in enumerate(self.iids)}
        self.attrs = item_attrs(self.iids)

    @classmethod
    def from_records(cls, records):
        """records: [(uid, [(iid, rating, time)*])*]"""
        uids = [u for u, _ in records]
        iids = sorted({i for _, prof in records for (i, _, _) in prof})
        return cls(uids, iids)
