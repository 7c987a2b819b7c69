"""Order-preserving, list-backed stand-in for the slice of the PySpark 1.6 RDD /
SQLContext surface that X-MAP's hot path touches.

TEST INFRASTRUCTURE ONLY.  This module exists so that the *reference's own*
modules (imported read-only from /root/reference/code in this container) can be
driven end to end without Spark, to capture golden vectors.  It is build-owned
code; nothing here is taken from the reference.  It is never imported by the
product path (x-map_amd/), only by oracle/ref_harness/make_golden.py.

Semantics pinned here: one partition, insertion-ordered
reduceByKey/combineByKey, left-ordered join.
"""
import sys
import types


class MiniRDD(object):
    def __init__(self, data, ctx=None):
        self._data = list(data)
        self.ctx = ctx

    # -- transformations ----------------------------------------------------
    def map(self, f):
        return MiniRDD([f(x) for x in self._data], self.ctx)

    def flatMap(self, f):
        out = []
        for x in self._data:
            out.extend(f(x))
        return MiniRDD(out, self.ctx)

    def filter(self, f):
        return MiniRDD([x for x in self._data if f(x)], self.ctx)

    def mapPartitions(self, f):
        return MiniRDD(list(f(iter(self._data))), self.ctx)

    def reduceByKey(self, f):
        acc = {}
        for k, v in self._data:
            if k in acc:
                acc[k] = f(acc[k], v)
            else:
                acc[k] = v
        return MiniRDD(list(acc.items()), self.ctx)

    def combineByKey(self, create, merge_value, merge_combiners):
        acc = {}
        for k, v in self._data:
            if k in acc:
                acc[k] = merge_value(acc[k], v)
            else:
                acc[k] = create(v)
        return MiniRDD(list(acc.items()), self.ctx)

    def join(self, other):
        right = {}
        for k, v in other._data:
            right.setdefault(k, []).append(v)
        out = []
        for k, v in self._data:
            for w in right.get(k, ()):
                out.append((k, (v, w)))
        return MiniRDD(out, self.ctx)

    def union(self, other):
        return MiniRDD(self._data + other._data, self.ctx)

    def keys(self):
        return MiniRDD([k for k, _ in self._data], self.ctx)

    def intersection(self, other):
        s = set(other._data)
        seen = set()
        out = []
        for x in self._data:
            if x in s and x not in seen:
                seen.add(x)
                out.append(x)
        return MiniRDD(out, self.ctx)

    def cache(self):
        return self

    # -- actions --------------------------------------------------------------
    def collect(self):
        return list(self._data)

    def collectAsMap(self):
        return dict(self._data)

    def take(self, n):
        return self._data[:n]

    def count(self):
        return len(self._data)

    def reduce(self, f):
        it = iter(self._data)
        acc = next(it)
        for x in it:
            acc = f(acc, x)
        return acc

    # -- the one DataFrame hop on the path (assist.py:82-87) -------------------
    def toDF(self):
        return MiniDF(self._data, self.ctx)


class MiniDF(object):
    def __init__(self, rows, ctx):
        self._rows = rows
        self.ctx = ctx

    def registerTempTable(self, name):
        self.ctx._tables[name] = self._rows


class Broadcast(object):
    def __init__(self, value):
        self.value = value


class MiniSC(object):
    def __init__(self):
        self._tables = {}

    def broadcast(self, v):
        return Broadcast(v)

    def parallelize(self, data, n=None):
        return MiniRDD(data, self)


class MiniSQL(object):
    """Answers exactly the literal query the path issues."""
    QUERY = "SELECT DISTINCT id1 FROM sim_table WHERE label = 1"

    def __init__(self, sc):
        self.sc = sc

    def sql(self, q):
        if " ".join(q.split()) != self.QUERY:
            raise NotImplementedError(q)
        seen = {}
        for r in self.sc._tables["sim_table"]:
            if r.label == 1 and r.id1 not in seen:
                seen[r.id1] = True
        return MiniRDD([_Row(id1=k) for k in seen], self.sc)


class _Row(object):
    def __init__(self, **kw):
        self.__dict__.update(kw)


def install_pyspark_stub():
    """Put a minimal `pyspark.sql.Row` into sys.modules so the reference's
    baselinerSim.py (`from pyspark.sql import Row`) imports."""
    if "pyspark" in sys.modules:
        return
    pk = types.ModuleType("pyspark")
    sq = types.ModuleType("pyspark.sql")
    sq.Row = _Row
    pk.sql = sq
    sys.modules["pyspark"] = pk
    sys.modules["pyspark.sql"] = sq
