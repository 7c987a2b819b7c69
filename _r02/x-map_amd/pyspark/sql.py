"""pyspark.sql stand-in: Row and the one SQLContext query the path issues (assist.py:84-85)."""
import re


class Row(object):
    def __init__(self, **kw):
        self.__dict__.update(kw)

    def asDict(self):
        return dict(self.__dict__)

    def __repr__(self):
        return "Row(%s)" % ", ".join("%s=%r" % kv for kv in sorted(self.__dict__.items()))


class SQLContext(object):
    _Q = re.compile(r"^\s*SELECT\s+DISTINCT\s+(\w+)\s+FROM\s+(\w+)\s+WHERE\s+(\w+)\s*=\s*(-?\d+)\s*$", re.I)

    def __init__(self, sparkContext):
        self.sc = sparkContext

    def sql(self, query):
        from xmap.engine.localrdd import LocalDF
        m = self._Q.match(query)
        if not m:
            raise NotImplementedError("local SQLContext only answers SELECT DISTINCT c FROM t WHERE c2 = n: %r" % query)
        col, table, wcol, wval = m.group(1), m.group(2), m.group(3), int(m.group(4))
        seen, rows = set(), []
        for r in self.sc._tables[table]:
            if getattr(r, wcol) == wval:
                v = getattr(r, col)
                if v not in seen:
                    seen.add(v)
                    rows.append(Row(**{col: v}))
        return LocalDF(rows, self.sc)
