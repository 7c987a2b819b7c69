"""Helpers shared by the parity tests: load tests/golden/*.npz (vectors captured from the
reference by oracle/ref_harness/make_golden.py) into index-space inputs."""
import os

import numpy as np

from xmap.engine import ids as xids

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["kat7", "tiny", "mixed", "multilabel", "small", "medium"]
METHODS = ["cosine", "adjust_cosine"]
CAP = 50


class Golden(object):
    def __init__(self, name):
        self.name = name
        self.g = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        self.iids = [str(s) for s in self.g["iids"]]
        self.uids = [str(s) for s in self.g["uids"]]
        self.I = len(self.iids)
        self.attrs = xids.item_attrs(self.iids)
        self.ptr = self.g["train_ptr"]
        self.item = self.g["train_item"]
        self.rating = self.g["train_rating"].astype(np.float32)
        self.time = self.g["train_time"]

    def __getitem__(self, k):
        return self.g[k]

    def has(self, k):
        return k in self.g.files

    def ks(self, method):
        out = set()
        for f in self.g.files:
            p = f.split(".")
            if p[0] == method and len(p) > 2 and p[1].startswith("k"):
                out.add(int(p[1][1:]))
        return sorted(out)

    def gen_tags(self, method, k):
        pre = "%s.k%d." % (method, k)
        tags = set()
        for f in self.g.files:
            if f.startswith(pre):
                t = f[len(pre):].split(".")[0]
                if t == "priv" or t.startswith("np"):
                    tags.add(t)
        return sorted(tags)

    def oracle_train(self):
        from oracle import xmap_oracle as xo
        return xo.Train(self.ptr, self.item, self.rating, self.time, self.I, *self.attrs)


def csr_to_pairs(row_ptr, col):
    rows = np.repeat(np.arange(len(row_ptr) - 1, dtype=np.int64), np.diff(row_ptr))
    return rows, col.astype(np.int64)


def rows_to_csr(rows):
    """(uid, iid, rating, ts)* -> (uids, iids, user_ptr, item, rating) with users in order of first appearance and items
    in lexicographic id order (the index space of the engine and of the oracle)."""
    import numpy as np
    uids, useen = [], {}
    for r in rows:
        if r[0] not in useen:
            useen[r[0]] = len(uids)
            uids.append(r[0])
    iids = sorted({r[1] for r in rows})
    iidx = {s: k for k, s in enumerate(iids)}
    per = [[] for _ in uids]
    for r in rows:
        per[useen[r[0]]].append((iidx[r[1]], float(r[2])))
    ptr = np.zeros(len(uids) + 1, np.int64)
    item, rating = [], []
    for k, prof in enumerate(per):
        ptr[k + 1] = ptr[k] + len(prof)
        item += [p[0] for p in prof]
        rating += [p[1] for p in prof]
    return uids, iids, ptr, np.array(item, np.int32), np.array(rating, np.float32)
