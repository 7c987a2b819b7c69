# -*- coding: utf-8 -*-
"""RecommenderPrediction: item-based kNN prediction with temporal decay + MAE (mirror of reference
core/recommenderPrediction.py:5-139).  Evaluation stage downstream of the hot path."""
import numpy as np


class RecommenderPrediction:
    def __init__(self, alpha, method):
        self.alpha = alpha
        self.method = method

    def bound_rating(self, rating):
        """round half up, clamp to [0, 5] (reference :17-23)"""
        return 1.0 * max(0, min(int(rating + 0.5), 5))

    def _rank_by_time(self, triples):
        """(a, b, time) -> (a, b, rank): equal times share a rank, ranks start at 1 (reference :35-48)"""
        ordered = sorted(triples, key=lambda t: t[2])
        out, rank = [], 0
        for pos, t in enumerate(ordered):
            if pos == 0 or t[2] != ordered[pos - 1][2]:
                rank += 1
            out.append((t[0], t[1], rank))
        return out

    def _decayed_ratio(self, triples):
        """sum(w*x)/sum(w*y) with w = exp(-alpha (now - rank)), now = max rank + 1 (reference :50-66)"""
        ranked = self._rank_by_time(triples)
        now = max(t[2] for t in ranked) + 1
        weighted = [(t[0] * np.exp(- self.alpha * (now - t[2])), t[1] * np.exp(- self.alpha * (now - t[2])))
                    for t in ranked]
        return sum(w[0] for w in weighted) / sum(w[1] for w in weighted)

    def _predict_pair(self, uid, pair, rating_bd, sim_bd, item_bd):
        """one (iid, real) pair of a test user -- reference :68-99"""
        iid, real = pair[0], pair[1]
        if iid not in sim_bd.value.keys():
            return ()
        base = item_bd.value[iid][0]
        evidence = []
        for niid, nsim in sim_bd.value[iid]:
            navg = item_bd.value[niid][0]
            for rater, rating, when in rating_bd.value[niid]:
                if uid in rater:
                    evidence.append((nsim * (rating - navg), abs(nsim), when))
        if evidence:
            plain = base + sum(ev[0] for ev in evidence) / sum(ev[1] for ev in evidence)
            decayed = base + self._decayed_ratio(evidence)
        else:
            plain = decayed = base
        return (iid, real, self.bound_rating(plain), self.bound_rating(decayed))

    def item_based_prediction(self, line, rating_bd, sim_bd, item_bd):
        """(uid, [(iid, real, predicted without decay, predicted with decay) | ()]) -- reference :25-105.
        rating_bd: {iid: [(uid, rating, time)*]}, sim_bd: {iid: [(iid, sim)*]}, item_bd: {iid: (avg, norm, n)}.
        Like the reference, a neighbour's rating counts when `uid in rater_id` (substring test)."""
        uid, pairs = line
        return uid, [self._predict_pair(uid, pair, rating_bd, sim_bd, item_bd) for pair in pairs]

    def item_based_recommendation(self, test_dataRDD, item_based_dict_bd, itembased_sim_pair_dict_bd, item_info_bd):
        """reference :101-105.  On a machine with the HIP library and a GPU the pairs are predicted by `xmap_predict`
        (csrc/stage_e.hip: one thread per test pair, the same operations in the same order; the decay weights come from
        np.exp on the host) whenever the reference's substring test `uid in rater_id` is an equality test, i.e. all user ids
        have one length; otherwise -- and for the rare pair with more than 64 evidence entries -- by the Python statement
        above.  Both give the same tuples (tests/test_gpu_recsim.py)."""
        out = self._device_recommendation(test_dataRDD, item_based_dict_bd, itembased_sim_pair_dict_bd, item_info_bd)
        if out is not None:
            return out
        return test_dataRDD.map(lambda line: self.item_based_prediction(
            line, item_based_dict_bd, itembased_sim_pair_dict_bd, item_info_bd))

    @staticmethod
    def _time_key(when):
        """a number with the order and the ties of the time objects (naive datetimes: seconds since 1970 by subtraction,
        which is monotone; aware ones: timestamp())"""
        import datetime
        if isinstance(when, datetime.datetime):
            if when.tzinfo is None:
                return (when - datetime.datetime(1970, 1, 1)).total_seconds()
            return when.timestamp()
        return float(when)

    def _device_recommendation(self, test_dataRDD, rating_bd, sim_bd, item_bd):
        try:
            import torch
            if not torch.cuda.is_available():
                return None
            from ..engine import hipabi as abi
            from ..engine.localrdd import LocalRDD, records_of
        except ImportError:
            return None
        import ctypes as C
        recs = records_of(test_dataRDD)
        ratings, sims, info = rating_bd.value, sim_bd.value, item_bd.value
        raters = {r[0] for lst in ratings.values() for r in lst}
        ids = raters | {uid for uid, _ in recs}
        if not ids or not all(isinstance(x, str) for x in ids) or len({len(x) for x in ids}) != 1:
            return None                     # `uid in rater_id` is a genuine substring test here
        try:
            uidx = {u: k for k, u in enumerate(sorted(raters))}
            items = set(ratings) | set(sims) | {n for lst in sims.values() for n, _ in lst}
            iidx = {i: k for k, i in enumerate(sorted(items))}
            I = len(iidx)
            avg = np.zeros(max(I, 1))
            for i in sims:                                      # averages are read for items with a list and for neighbours
                avg[iidx[i]] = info[i][0]
            for lst in sims.values():
                for n, _ in lst:
                    avg[iidx[n]] = info[n][0]
                    ratings[n]                                  # the reference raises KeyError here too
            nb_ptr = np.zeros(I + 1, np.int64)
            for i, lst in sims.items():
                nb_ptr[iidx[i] + 1] = len(lst)
            np.cumsum(nb_ptr, out=nb_ptr)
            nb_item = np.zeros(max(int(nb_ptr[-1]), 1), np.int32)
            nb_sim = np.zeros(max(int(nb_ptr[-1]), 1), np.float64)
            for i, lst in sims.items():
                a = int(nb_ptr[iidx[i]])
                for q, (n, sv) in enumerate(lst):
                    nb_item[a + q] = iidx[n]
                    nb_sim[a + q] = sv
            rt_ptr = np.zeros(I + 1, np.int64)
            for i, lst in ratings.items():
                rt_ptr[iidx[i] + 1] = len(lst)
            np.cumsum(rt_ptr, out=rt_ptr)
            nr = int(rt_ptr[-1])
            rt_user = np.zeros(max(nr, 1), np.int32)
            rt_rating = np.zeros(max(nr, 1), np.float64)
            rt_time = np.zeros(max(nr, 1), np.float64)
            for i, lst in ratings.items():
                a = int(rt_ptr[iidx[i]])
                us = np.fromiter((uidx[r[0]] for r in lst), np.int64, len(lst))
                o = np.argsort(us, kind="stable")                # a user's ratings of an item keep their list order
                rt_user[a:a + len(lst)] = us[o]
                rt_rating[a:a + len(lst)] = np.asarray([float(lst[q][1]) for q in o])
                rt_time[a:a + len(lst)] = np.asarray([self._time_key(lst[q][2]) for q in o])
        except (KeyError, TypeError, ValueError):
            return None                     # let the Python statement raise what the reference raises
        n_w = 66
        wtab = np.asarray([np.exp(- self.alpha * d) for d in range(n_w)], np.float64)      # scalar calls, like the reference's
        tu, ti, where = [], [], []
        for ru, (uid, pairs) in enumerate(recs):
            for rp, pair in enumerate(pairs):
                tu.append(uidx.get(uid, -1))
                ti.append(iidx[pair[0]] if pair[0] in sims else -1)
                where.append((ru, rp))
        T = len(tu)
        dev = "cuda:%d" % torch.cuda.current_device()
        to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        d = [to(np.asarray(tu, np.int32).reshape(-1)), to(np.asarray(ti, np.int32).reshape(-1)), to(nb_ptr), to(nb_item), to(nb_sim),
             to(rt_ptr), to(rt_user), to(rt_rating), to(rt_time), to(avg), to(wtab)]
        plain = torch.zeros(max(T, 1), dtype=torch.float64, device=dev)
        decay = torch.zeros(max(T, 1), dtype=torch.float64, device=dev)
        status = torch.zeros(max(T, 1), dtype=torch.int32, device=dev)
        vp = lambda t: C.c_void_p(t.data_ptr())
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        abi.check(abi.lib.xmap_predict(st, C.c_int64(T), *[vp(x) for x in d[:10]], vp(d[10]), C.c_int32(n_w), vp(plain), vp(decay),
                                       vp(status)))
        plain, decay, status = plain.cpu().numpy(), decay.cpu().numpy(), status.cpu().numpy()
        out = [(uid, [None] * len(pairs)) for uid, pairs in recs]
        for q, (ru, rp) in enumerate(where):
            uid, pairs = recs[ru]
            if status[q] == 1:
                out[ru][1][rp] = ()
            elif status[q] == 0:
                out[ru][1][rp] = (pairs[rp][0], pairs[rp][1], float(plain[q]), float(decay[q]))
            else:
                out[ru][1][rp] = self._predict_pair(uid, pairs[rp], rating_bd, sim_bd, item_bd)
        return LocalRDD(out, getattr(test_dataRDD, "ctx", None))

    def calculate_mae(self, rdd):
        """'<MAE without decay>; <MAE with decay>' (or a single MAE for user-based methods) -- reference :107-139"""
        def errors(line, index):
            return [abs(p[1] - p[index]) for p in line[1] if p != ()]

        def mae(index):
            tot = rdd.map(lambda line: errors(line, index)).map(
                lambda errs: np.array([sum(errs), len(errs)])).reduce(lambda a, b: a + b)
            return tot[0] / tot[1]
        if "user" in self.method:
            return str(mae(2))
        return str(1.0 * mae(2)) + '; ' + str(mae(3))
