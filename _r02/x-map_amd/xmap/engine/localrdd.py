"""List-backed, order-preserving RDD-like container (single partition).

The hot path returns handles that own HBM buffers; they behave like the RDDs the reference's
downstream code expects (recommenderSim.py:20-27 does rdd.map(...).reduceByKey(...)) by
materialising Python tuples lazily on first access.  The same class backs the minimal local
`pyspark` runtime shipped in x-map_amd/pyspark.
"""


class LocalRDD(object):
    def __init__(self, data=None, ctx=None, thunk=None):
        self._items = None if data is None else list(data)
        self._thunk = thunk
        self.ctx = ctx

    # lazily materialised payload -------------------------------------------------
    @property
    def _data(self):
        if self._items is None:
            # handles (subclasses owning HBM buffers) define _rows(); it is looked up here rather than stored as a bound
            # method: self -> bound method -> self would be a reference cycle, and the buffers of a dropped handle (13 GB
            # for an extension at BASELINE configs[1]) would stay allocated until the cyclic collector happens to run
            self._items = list((self._thunk or self._rows)())
        return self._items

    def _rows(self):
        raise ValueError("LocalRDD without data")

    def _new(self, data):
        return LocalRDD(data, self.ctx)

    # transformations ------------------------------------------------------------
    def map(self, f):
        return self._new([f(x) for x in self._data])

    def flatMap(self, f):
        out = []
        for x in self._data:
            out.extend(f(x))
        return self._new(out)

    def filter(self, f):
        return self._new([x for x in self._data if f(x)])

    def mapPartitions(self, f):
        return self._new(list(f(iter(self._data))))

    def reduceByKey(self, f):
        acc = {}
        for key, v in self._data:
            acc[key] = f(acc[key], v) if key in acc else v
        return self._new(list(acc.items()))

    def combineByKey(self, create, merge_value, merge_combiners):
        acc = {}
        for key, v in self._data:
            acc[key] = merge_value(acc[key], v) if key in acc else create(v)
        return self._new(list(acc.items()))

    def aggregateByKey(self, zero, seq_op, comb_op):
        import copy
        acc = {}
        for key, v in self._data:
            acc[key] = seq_op(acc[key] if key in acc else copy.deepcopy(zero), v)
        return self._new(list(acc.items()))

    def groupByKey(self):
        acc = {}
        for key, v in self._data:
            acc.setdefault(key, []).append(v)
        return self._new(list(acc.items()))

    def join(self, other):
        right = {}
        for key, v in other._data:
            right.setdefault(key, []).append(v)
        return self._new([(key, (v, w)) for key, v in self._data for w in right.get(key, ())])

    def union(self, other):
        return self._new(self._data + list(other._data))

    def keys(self):
        return self._new([key for key, _ in self._data])

    def values(self):
        return self._new([v for _, v in self._data])

    def distinct(self):
        seen, out = set(), []
        for x in self._data:
            if x not in seen:
                seen.add(x)
                out.append(x)
        return self._new(out)

    def intersection(self, other):
        s = set(other._data)
        return self._new([x for x in self.distinct()._data if x in s])

    def sortBy(self, keyfunc, ascending=True):
        return self._new(sorted(self._data, key=keyfunc, reverse=not ascending))

    def randomSplit(self, weights, seed=None):
        import random
        rng = random.Random(seed)
        tot = float(sum(weights))
        cuts, acc = [], 0.0
        for w in weights:
            acc += w / tot
            cuts.append(acc)
        parts = [[] for _ in weights]
        for x in self._data:
            r = rng.random()
            for k, c in enumerate(cuts):
                if r < c or k == len(cuts) - 1:
                    parts[k].append(x)
                    break
        return [self._new(p) for p in parts]

    def cache(self):
        return self

    def persist(self, *a):
        return self

    def unpersist(self):
        return self

    # actions ----------------------------------------------------------------------
    def collect(self):
        return list(self._data)

    def collectAsMap(self):
        return dict(self._data)

    def take(self, n):
        return self._data[:n]

    def first(self):
        return self._data[0]

    def count(self):
        return len(self._data)

    def reduce(self, f):
        it = iter(self._data)
        acc = next(it)
        for x in it:
            acc = f(acc, x)
        return acc

    def foreach(self, f):
        for x in self._data:
            f(x)

    def __iter__(self):
        return iter(self._data)

    def toDF(self):
        return LocalDF(self._data, self.ctx)


class LocalDF(object):
    def __init__(self, rows, ctx):
        self._rows = list(rows)
        self.ctx = ctx

    def registerTempTable(self, name):
        tables = getattr(self.ctx, "_tables", None)
        if tables is None:
            raise RuntimeError("no SparkContext attached to this DataFrame")
        tables[name] = self._rows

    def collect(self):
        return list(self._rows)

    def map(self, f):  # Spark 1.x DataFrame.map (assist.py:84-86)
        return LocalRDD([f(r) for r in self._rows], self.ctx)


def records_of(rdd_or_list):
    """Records of an RDD-like object or a plain iterable."""
    if hasattr(rdd_or_list, "collect"):
        return rdd_or_list.collect()
    return list(rdd_or_list)
