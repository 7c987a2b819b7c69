"""Minimal local stand-in for the slice of PySpark 1.6 that X-MAP's demos touch
(twodomain_demo.py:5-6,31-43): SparkConf, SparkContext (textFile / parallelize / broadcast / stop).
Single process, single partition, order preserving.  Spark itself is NOT part of this engine: the
hot path runs on the GPU behind xmap.utils.assist; this module only lets the reference's driver
scripts run where no Spark installation exists.  If a real PySpark is installed,
put it ahead of x-map_amd on sys.path and it is used instead.
"""
from xmap.engine.localrdd import LocalRDD


class SparkConf(object):
    def __init__(self):
        self._conf = {}

    def setAppName(self, name):
        self._conf["spark.app.name"] = name
        return self

    def setMaster(self, master):
        self._conf["spark.master"] = master
        return self

    def set(self, key, value):
        self._conf[key] = value
        return self

    def get(self, key, default=None):
        return self._conf.get(key, default)


class Broadcast(object):
    def __init__(self, value):
        self.value = value

    def unpersist(self):
        pass


class SparkContext(object):
    def __init__(self, master=None, appName=None, conf=None, **kw):
        self.conf = conf or SparkConf()
        self._tables = {}

    def textFile(self, path, minPartitions=None):
        if path.startswith("file:"):
            path = path[len("file:"):]
        with open(path, "r") as f:
            return LocalRDD([line.rstrip("\n") for line in f], self)

    def parallelize(self, data, numSlices=None):
        return LocalRDD(data, self)

    def broadcast(self, value):
        return Broadcast(value)

    def stop(self):
        pass
