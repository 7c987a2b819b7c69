"""Native feeder (csrc/feeder.hip): raw `uid iid rating unix_ts` text -> id tables + CSR + predicate arrays in C++, with
the clean stage of the reference (core/baselinerClean.py:40-101) in between.  Host code only (no GPU needed); the device
engine uploads the arrays as they are, Python objects are made only if somebody iterates the records."""
import ctypes as C
from datetime import datetime

import numpy as np

from . import hipabi as abi
from .localrdd import LocalRDD

lib = abi.lib


class Feed(object):
    """owner of one xmap_feed handle"""

    def __init__(self, handle):
        self._h = handle
        sz = (C.c_int64 * 7)()
        abi.check(lib.xmap_feed_sizes(self._h, sz))
        (self.n_users, self.n_items, self.nnz, self._ub, self._ib, self.n_lines, self.n_in_period) = [int(x) for x in sz]
        self._arrays = None
        self._ids = {}

    @classmethod
    def from_text(cls, text, year_from, year_to, label, min_ratings):
        """text: bytes (or str) of newline-separated lines"""
        if isinstance(text, str):
            text = text.encode("utf-8")
        h = C.c_void_p()
        abi.check(lib.xmap_feed_text(text, C.c_int64(len(text)), int(year_from), int(year_to), label.encode("utf-8"),
                                     int(min_ratings), C.byref(h)))
        return cls(h)

    @classmethod
    def from_texts(cls, parts, year_from, year_to, min_ratings):
        """parts: [(text, label)*], the domains of one problem, in one call: users in first-seen order over the parts, a
        user's entries of an earlier part first (= the merge of the parts' feeds)"""
        texts = [t.encode("utf-8") if isinstance(t, str) else t for t, _ in parts]
        n = len(parts)
        tp = (C.c_char_p * n)(*texts)
        lens = (C.c_int64 * n)(*[len(t) for t in texts])
        labs = (C.c_char_p * n)(*[lab.encode("utf-8") for _, lab in parts])
        h = C.c_void_p()
        abi.check(lib.xmap_feed_texts(n, tp, lens, labs, int(year_from), int(year_to), int(min_ratings), C.byref(h)))
        return cls(h)

    @classmethod
    def from_file(cls, path, year_from, year_to, label, min_ratings):
        if path.startswith("file:"):
            path = path[5:]
        with open(path, "rb") as f:
            return cls.from_text(f.read(), year_from, year_to, label, min_ratings)

    def merge(self, other):
        """this feed's users, then the users only `other` has; a user both have gets this feed's entries first"""
        h = C.c_void_p()
        abi.check(lib.xmap_feed_merge(self._h, other._h, C.byref(h)))
        return Feed(h)

    def arrays(self):
        """(user_ptr i64, item i32, rating f64, when f64, (prefix_cls, suffix_cls, contains_mask, flags))"""
        if self._arrays is None:
            n, I = max(self.nnz, 1), max(self.n_items, 1)
            ptr = np.zeros(self.n_users + 1, np.int64)
            item, rating, when = np.zeros(n, np.int32), np.zeros(n, np.float64), np.zeros(n, np.float64)
            pre, suf = np.zeros(I, np.int32), np.zeros(I, np.int32)
            msk, flg = np.zeros(I, np.uint32), np.zeros(I, np.uint8)
            p = lambda a: a.ctypes.data_as(C.c_void_p)
            abi.check(lib.xmap_feed_arrays(self._h, p(ptr), p(item), p(rating), p(when), p(pre), p(suf), p(msk), p(flg)))
            self._arrays = (ptr, item[:self.nnz], rating[:self.nnz], when[:self.nnz],
                            (pre[:self.n_items], suf[:self.n_items], msk[:self.n_items], flg[:self.n_items]))
        return self._arrays

    def ids(self, which):
        """the user (0) or item (1) id strings"""
        if which not in self._ids:
            n, nb = (self.n_items, self._ib) if which else (self.n_users, self._ub)
            buf = C.create_string_buffer(max(nb + n, 1))
            abi.check(lib.xmap_feed_ids(self._h, int(which) | 2, buf, None))      # one id per line: split in C, not per id
            self._ids[which] = buf.raw[:nb + n].decode("utf-8").split("\n")[:n] if n else []
        return self._ids[which]

    def records(self):
        """[(uid, [(iid, rating, datetime)*])*]: what BaselinerClean's three steps leave (local-time datetimes)"""
        ptr, item, rating, when, _ = self.arrays()
        uids, iids = self.ids(0), self.ids(1)
        out = []
        for u in range(self.n_users):
            a, b = int(ptr[u]), int(ptr[u + 1])
            out.append((uids[u], [(iids[item[e]], float(rating[e]), datetime.fromtimestamp(float(when[e]))) for e in range(a, b)]))
        return out

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            lib.xmap_feed_free(h)


class FeedRDD(LocalRDD):
    """cleaned ratings of one domain (or a merged train set) as an RDD whose records exist natively: the device engine takes
    the arrays (session.train_state), Python tuples are made only when the RDD is iterated (e.g. by the split stage)"""

    def __init__(self, feed, ctx=None):
        LocalRDD.__init__(self, None, ctx)
        self.feed = feed

    def _rows(self):
        return self.feed.records()

    def cache(self):
        return self


class _LazyTimes(object):
    """positions -> the reference's time objects (datetime.fromtimestamp), made on demand"""

    def __init__(self, when):
        self.when = when

    def __getitem__(self, k):
        return datetime.fromtimestamp(float(self.when[k]))

    def __len__(self):
        return len(self.when)


class _LazyRatings(object):
    def __init__(self, rating):
        self.rating = rating

    def __getitem__(self, k):
        return float(self.rating[k])

    def __len__(self):
        return len(self.rating)


def train_state_from_feed(feed):
    """session.TrainState without a Python loop over the ratings"""
    from . import device, ids as xids, session
    ptr, item, rating, when, attrs = feed.arrays()
    st = session.TrainState.__new__(session.TrainState)
    idt = xids.IdTable.__new__(xids.IdTable)
    idt.uids, idt.iids = feed.ids(0), feed.ids(1)
    idt.uidx = {s: k for k, s in enumerate(idt.uids)}
    idt.iidx = {s: k for k, s in enumerate(idt.iids)}
    idt.attrs = attrs
    st.idt = idt
    st.times, st.ratings = _LazyTimes(when), _LazyRatings(rating)
    n = len(item)
    st.R = device.DeviceRatings(ptr, item, rating.astype(np.float32), np.arange(n, dtype=np.int64), len(idt.iids), attrs)
    st.engine = device.Engine(st.R)
    return st
